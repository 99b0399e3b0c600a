/*
 * oracle/ko_fft.h -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Power-of-two double-precision FFTs with the calling conventions the WORLD
 * vocoder's FFT wrapper exposes (FFTW-like: forward r2c, UNNORMALISED c2r,
 * complex forward/backward).  WORLD itself bundles Ooura's FFT behind that
 * wrapper; this file is an independent restatement of the same transforms
 * (results agree to rounding, not bit-for-bit).
 *
 * Nothing under kwiiyatta_amd/ may include, link or call this file.
 */
#ifndef KO_FFT_H_
#define KO_FFT_H_

#ifdef __cplusplus
extern "C" {
#endif

/* in-place complex FFT of n (power of two) points; re/im interleaved.
 * sign = -1: X[k] = sum x[j] exp(-2 pi i jk/n);  sign = +1: exp(+...).
 * No normalisation in either direction. */
void ko_cfft(double *z, int n, int sign);

/* real -> complex: out has n/2+1 interleaved complex bins. */
void ko_rfft(const double *x, int n, double *out);

/* complex (n/2+1 bins, Hermitian implied) -> real, unnormalised:
 * ko_irfft(ko_rfft(x)) == n * x   (same as FFTW c2r / WORLD's wrapper). */
void ko_irfft(const double *spec, int n, double *x);

#ifdef __cplusplus
}
#endif
#endif
