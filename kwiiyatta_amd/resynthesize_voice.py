"""`kwiieiya`: analyse a wav file and synthesise it again -- optionally through its mel-cepstrum (--mcep), at another
sampling rate (--result-fs), or on the timing and f0 of a second recording (--carrier; with --diffvc the carrier's
own waveform is filtered towards the source's spectrum instead of being re-synthesised).  Command line of the
reference's kwiiyatta/resynthesize_voice.py; its Qt dialog (started when no source file is given) is not part of
this build."""
import copy
import pathlib


def render(conf, source):
    """the result waveform for the parsed options"""
    import kwiiyatta_amd as k
    if conf.carrier is None:
        picture = k.feature(source)
    else:
        carrier = conf.create_analyzer(conf.carrier, Analyzer=k.analyze_wav)
        picture = k.align(source, carrier)               # the source's features on the carrier's frames
        if conf.diffvc:
            difference = copy.copy(picture.mel_cepstrum)
            difference.data -= carrier.mel_cepstrum.data
            return k.apply_mlsa_filter(carrier.wavdata, difference)
        picture.f0 = carrier.f0
    if conf.mcep:
        picture.extract_mel_cepstrum()
        picture.spectrum_envelope = None                 # from here on the mel-cepstrum is the envelope
    if conf.result_fs is not None:
        picture.resample(conf.result_fs)
    return picture.synthesize()


def main():
    import kwiiyatta_amd as k
    conf = k.Config()
    conf.add_argument('source', type=str, default=None, nargs='?', help='Source wav file of voice resynthesis')
    conf.add_argument('--result-dir', type=str, help='Path to write result wav files')
    conf.add_argument('--mcep', action='store_true', help='Use mel-cepstrum to resynthesize')
    conf.add_argument('--play', action='store_true', help='Play result wavform')
    conf.add_argument('--no-save', action='store_true', help='Not to write result wav file, and play wavform')
    conf.add_argument('--carrier', type=str, help='Wav file to use for carrier')
    conf.add_argument('--diffvc', action='store_true', help='Use difference MelCepstrum synthesis')
    conf.add_argument('--result-fs', type=int, help='Result waveform sampling rate')
    conf.parse_args()
    if conf.source is None:
        conf.parser.error('a source wav file is required (the Qt dialog of the reference is not part of this build)')
    source_path = pathlib.Path(conf.source).resolve()
    wav = render(conf, conf.create_analyzer(source_path, Analyzer=k.analyze_wav))
    if not conf.no_save:
        target = (source_path.with_suffix('.resynth.wav') if conf.result_dir is None
                  else pathlib.Path(conf.result_dir) / source_path.name)
        target.parent.mkdir(parents=True, exist_ok=True)
        wav.save(target)
    if conf.play or conf.no_save:
        wav.play()


if __name__ == '__main__':
    main()
