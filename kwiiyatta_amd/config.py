"""Options shared by the command-line tools, and the factories that apply them.  API of kwiiyatta.config.Config
(/root/reference/kwiiyatta/config.py): the options are parsed INTO the Config object (it is the argparse namespace),
analyzers and converters are created through it so that frame period, mel-cepstrum order, component count and seed
follow the command line."""
import argparse
import functools
import pathlib
import sys

VOCODER_OPTIONS = (
    ('--frame-period', dict(type=int, default=5, help='Frame period milli-seconds of vocoder')),
    ('--mcep-order', dict(type=int, default=24, help='Mel-cepstrum order for spectrum envelope')),
)
CONVERTER_OPTIONS = (
    ('--source', dict(type=str, help='Source data-set path of voice conversion')),
    ('--target', dict(type=str, help='Target data-set path of voice conversion')),
    ('--max-files', dict(type=int, help='File num to train feature converter')),
    ('--skip-files', dict(type=int, help='Skip file num to train feature converter')),
    ('--mcep-fs', dict(type=int, help='Sampling rate of training mel cepstrum')),
    ('--converter-components', dict(type=int, default=64, help='Components num for feature converter')),
    ('--converter-seed', dict(type=int, help='Random seed for feature converter')),
    # an addition to the reference's options: keep the trained converter between runs
    ('--converter-model', dict(type=str, help='File of the trained converter: loaded when it exists (no training, '
                                              '--source/--target not needed), written after training otherwise')),
)


def _pkg():
    import kwiiyatta_amd
    return kwiiyatta_amd


class Config:
    def __init__(self, argparser=None):
        self.parser = argparser or argparse.ArgumentParser()
        self._declare(VOCODER_OPTIONS)

    def _declare(self, options):
        for flag, spec in options:
            self.parser.add_argument(flag, **spec)

    def add_converter_arguments(self):
        self._declare(CONVERTER_OPTIONS)

    def add_argument(self, *args, **kwargs):
        self.parser.add_argument(*args, **kwargs)

    def parse_args(self, args=None):
        """`args` come BEFORE the process arguments, which therefore win (the reference's order)"""
        self.parser.parse_args(args=list(args or []) + sys.argv[1:], namespace=self)

    # ---- factories --------------------------------------------------------------------------------------
    def create_analyzer(self, *args, Analyzer=None, **kwargs):
        make = Analyzer or _pkg().Analyzer
        return make(*args, **dict(kwargs, frame_period=self.frame_period, mcep_order=self.mcep_order))

    def create_converter(self, Converter=None, **kwargs):
        make = Converter or _pkg().MelCepstrumConverter
        chosen = dict(random_state=self.converter_seed, components=self.converter_components)
        if self.mcep_fs is not None:
            chosen['mcep_fs'] = self.mcep_fs
        return make(**dict(chosen, **kwargs))

    def _required_dir(self, flag):
        # the reference tests `source` for both flags (config.py:79); kept, since it decides when --target alone errors
        if self.source is None:
            self.parser.error(f'the following arguments are required: --{flag}')
        return pathlib.Path(getattr(self, flag))

    source_path = property(lambda self: self._required_dir('source'))
    target_path = property(lambda self: self._required_dir('target'))

    def load_dataset(self):
        """the aligned parallel training set of --source / --target"""
        k = _pkg()
        analyze = functools.partial(self.create_analyzer, Analyzer=k.analyze_wav)
        sides = [k.WavFileDataset(path, Analyzer=analyze) for path in (self.source_path, self.target_path)]
        return k.align(*sides)

    def train_converter(self, **kwargs):
        converter = self.create_converter(**kwargs)
        model = getattr(self, 'converter_model', None)
        if model is not None and pathlib.Path(model).is_file():
            return converter.load(model)
        converter = self._train(converter)
        if model is not None:
            converter.save(model)
        return converter

    def _train(self, converter):
        dataset = self.load_dataset()
        keys = sorted(dataset.keys())[slice(self.skip_files, None)]
        converter.train(dataset, keys[:self.max_files])
        return converter
