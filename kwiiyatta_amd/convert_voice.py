"""`kwiiyatta`: train a converter on a parallel corpus (--source / --target directories with equally named wav
files) and convert wav files with it.  Command line and outputs of the reference's kwiiyatta/convert_voice.py:
for every input <name>.wav a <name>.diff.wav (the input waveform through the differential MLSA filter) and a
<name>.synth.wav (WORLD synthesis from the converted mel-cepstrum).  `--no-diffvc` (an addition) skips the first."""
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


def main():
    import kwiiyatta_amd as k
    conf = k.Config()
    conf.add_argument('--result-dir', type=str, help='Path to write result wav files')
    conf.add_argument('files', type=str, nargs='+', help='Wav files to convert voice')
    conf.add_argument('--no-diffvc', action='store_true', help='Write only the .synth.wav outputs')
    conf.add_converter_arguments()
    conf.parse_args()
    converter = conf.train_converter(use_delta=True)
    for name in conf.files:
        wav_path = pathlib.Path(name)
        stem = wav_path if conf.result_dir is None else pathlib.Path(conf.result_dir) / wav_path.name
        stem.parent.mkdir(parents=True, exist_ok=True)
        for suffix, differential in OUTPUTS:
            if differential and conf.no_diffvc:
                continue
            out = stem.with_suffix(f'.{suffix}.wav')
            print(f'{suffix} MLPG: {out!s}')
            convert(conf, converter, wav_path, diffvc=differential).save(out)


if __name__ == '__main__':
    main()
