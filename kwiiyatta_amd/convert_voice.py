"""`kwiiyatta`: train a converter on a parallel corpus (--source / --target directories with equally named wav
files) and convert wav files with it.  Command line and outputs of the reference's kwiiyatta/convert_voice.py:
for every input <name>.wav a <name>.diff.wav (the input waveform through the differential MLSA filter) and a
<name>.synth.wav (WORLD synthesis from the converted mel-cepstrum).  Additions: `--no-diffvc` skips the first;
`--converter-model FILE` keeps the trained converter between runs; `--batch` renders the .synth.wav outputs of all
input files -- and the .diff.wav outputs -- through the HBM-resident batch path (kwiiyatta_amd.corpus.convert_batch:
waves of 16 files in lockstep, wav in -> 16-bit PCM out on the device) instead of file by file."""
import pathlib

OUTPUTS = (('diff', True), ('synth', False))          # suffix, differential?


def convert(conf, converter, src_path, diffvc=True):
    """one converted waveform.  The file is analysed afresh per call, as the reference does."""
    import kwiiyatta_amd as k
    source = conf.create_analyzer(src_path, Analyzer=k.analyze_wav)
    converted = converter.convert(source.mel_cepstrum, diff=diffvc)
    if diffvc:
        return k.apply_mlsa_filter(source, converted)
    rendered = k.feature(source)
    rendered.mel_cepstrum = converted                   # takes over from the analysed envelope
    return rendered.synthesize()


class _Pcm16:
    """the 16-bit samples of a finished waveform as the batch path hands them over: `save` writes them as they are
    (what Wavdata.save(normalize=True) would write for the waveform -- kwy_finish_pcm16_batch_dev)"""

    def __init__(self, fs, pcm):
        self.fs, self.pcm = fs, pcm

    def save(self, wav):
        from scipy.io import wavfile
        wavfile.write(wav, self.fs, self.pcm)


def convert_synth_batch(conf, converter, paths, diffvc=False):
    """{(path, differential?): object with .save(file)} of the .synth.wav outputs -- with diffvc=True of the .diff.wav
    outputs too.  Files whose sampling rate or frame period differ from the converter's go through `convert` one by
    one (the batch path has no resampling stage).  The others go through the device WAV IN -> PCM OUT: f0 (DIO +
    StoneMask), analysis, conversion, synthesis / the MLSA filter of the differential conversion, the post-step of
    `synthesize` and `save`'s normalisation and 16-bit truncation all run on the GPU
    (corpus.convert_batch(pcm=True, diff=...)); the host reads the wav files and writes 2 bytes per sample."""
    import kwiiyatta_amd as k
    from . import corpus
    from .converter.delta import DeltaFeatureConverter
    out, batch = {}, []
    period = next((s.frame_period for s in _stages(converter) if isinstance(s, DeltaFeatureConverter)), None)
    for path in paths:
        a = conf.create_analyzer(path, Analyzer=k.analyze_wav)
        if a.fs != converter.fs or a.mel_cepstrum_order != converter.order or \
                (period is not None and a.frame_period != period):
            out[path, False] = convert(conf, converter, path, diffvc=False)
        else:
            batch.append((path, a))
    if batch:
        fs = batch[0][1].fs
        waves = [a.wavdata.data for _, a in batch]
        res = corpus.convert_batch(waves, fs, converter.gmm, order=converter.order,
                                   frame_period=float(batch[0][1].frame_period), pcm=True, diff=diffvc)
        for k, (path, a) in enumerate(batch):
            out[path, False] = _Pcm16(fs, res[1][k].cpu().numpy())
            if diffvc:
                out[path, True] = _Pcm16(fs, res[3][k].cpu().numpy())
    return out


def _stages(converter):
    stage = converter
    while stage is not None:
        yield stage
        stage = getattr(stage, '__dict__', {}).get('base')


def main():
    import kwiiyatta_amd as k
    conf = k.Config()
    conf.add_argument('--result-dir', type=str, help='Path to write result wav files')
    conf.add_argument('files', type=str, nargs='+', help='Wav files to convert voice')
    conf.add_argument('--no-diffvc', action='store_true', help='Write only the .synth.wav outputs')
    conf.add_argument('--batch', action='store_true',
                      help='Render the outputs of all files through the GPU-resident batch path')
    conf.add_converter_arguments()
    conf.parse_args()
    converter = conf.train_converter(use_delta=True)
    batched = convert_synth_batch(conf, converter, [pathlib.Path(n) for n in conf.files],
                                  diffvc=not conf.no_diffvc) if conf.batch else {}
    for name in conf.files:
        wav_path = pathlib.Path(name)
        stem = wav_path if conf.result_dir is None else pathlib.Path(conf.result_dir) / wav_path.name
        stem.parent.mkdir(parents=True, exist_ok=True)
        for suffix, differential in OUTPUTS:
            if differential and conf.no_diffvc:
                continue
            out = stem.with_suffix(f'.{suffix}.wav')
            print(f'{suffix} MLPG: {out!s}')
            if (wav_path, differential) in batched:
                batched[wav_path, differential].save(out)
            else:
                convert(conf, converter, wav_path, diffvc=differential).save(out)


if __name__ == '__main__':
    main()
