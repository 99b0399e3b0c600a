// How many bytes of kernel arguments does a captured HIP graph take before hipStreamEndCapture / instantiate breaks?
// N launches of a kernel with a BYTES-byte by-value struct, captured on one stream (optionally forked over two).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/graph_kernarg tools/ubench/graph_kernarg.hip && /tmp/graph_kernarg
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/wait.h>
#include <unistd.h>

template <int BYTES>
struct blob { char b[BYTES]; };

template <int BYTES>
__global__ void k_blob(blob<BYTES> a, int *out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(out, (int)a.b[BYTES - 1]);
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(3); } } while (0)

template <int BYTES>
static int run(int n, int streams) {
  hipStream_t s[2];
  CK(hipStreamCreate(&s[0])); CK(hipStreamCreate(&s[1]));
  int *d; CK(hipMalloc(&d, 4)); CK(hipMemset(d, 0, 4));
  blob<BYTES> a; for (int i = 0; i < BYTES; ++i) a.b[i] = 1;
  hipGraph_t g; hipGraphExec_t ge; hipEvent_t ev, ev2;
  CK(hipEventCreate(&ev)); CK(hipEventCreate(&ev2));
  CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeGlobal));
  if (streams == 2) { CK(hipEventRecord(ev, s[0])); CK(hipStreamWaitEvent(s[1], ev, 0)); }
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_blob<BYTES>, dim3(1), dim3(64), 0, s[streams == 2 ? i & 1 : 0], a, d);
  if (streams == 2) { CK(hipEventRecord(ev2, s[1])); CK(hipStreamWaitEvent(s[0], ev2, 0)); }
  CK(hipStreamEndCapture(s[0], &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, s[0]));
  CK(hipStreamSynchronize(s[0]));
  int h = 0; CK(hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost));
  return h == n ? 0 : 4;
}

int main() {
  const int ns[] = {16, 64, 128, 256, 512, 1024, 2048};
  for (int streams = 1; streams <= 2; ++streams)
    for (int bytes : {256, 1024, 3072})
      for (int n : ns) {
        fflush(stdout);
        pid_t p = fork();      // (a fresh process per case: a crash must not end the sweep; fork happens before any HIP call)
        if (p == 0) {
          int rc = bytes == 256 ? run<256>(n, streams) : bytes == 1024 ? run<1024>(n, streams) : run<3072>(n, streams);
          _exit(rc);
        }
        int st = 0; waitpid(p, &st, 0);
        printf("streams %d  bytes %4d  launches %4d  total %7d B : %s\n", streams, bytes, n, bytes * n,
               WIFSIGNALED(st) ? "SIGNAL" : (WEXITSTATUS(st) == 0 ? "ok" : "FAILED"));
      }
  return 0;
}
