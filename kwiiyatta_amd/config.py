"""Command-line configuration shared by the two CLIs (mirrors
/root/reference/kwiiyatta/config.py:9-104): an argparse parser that parses INTO
the Config object, plus factories that inject the parsed options."""
import argparse
import functools
import pathlib
import sys

import kwiiyatta_amd as kwiiyatta


class Config:
    def __init__(self, argparser=None):
        self.parser = argparser if argparser is not None else argparse.ArgumentParser()
        self.parser.add_argument('--frame-period', type=int, default=5,
                                 help='Frame period milli-seconds of vocoder')
        self.parser.add_argument('--mcep-order', type=int, default=24,
                                 help='Mel-cepstrum order for spectrum envelope')

    def add_converter_arguments(self):
        add = self.parser.add_argument
        add('--source', type=str, help='Source data-set path of voice conversion')
        add('--target', type=str, help='Target data-set path of voice conversion')
        add('--max-files', type=int, help='File num to train feature converter')
        add('--skip-files', type=int, help='Skip file num to train feature converter')
        add('--mcep-fs', type=int, help='Sampling rate of training mel cepstrum')
        add('--converter-components', type=int, default=64,
            help='Components num for feature converter')
        add('--converter-seed', type=int, help='Random seed for feature converter')

    def add_argument(self, *args, **kwargs):
        self.parser.add_argument(*args, **kwargs)

    def parse_args(self, args=None):
        # explicit args are PREPENDED to the process arguments, so sys.argv wins
        argv = sys.argv[1:] if args is None else args + sys.argv[1:]
        self.parser.parse_args(args=argv, namespace=self)

    def create_analyzer(self, *args, Analyzer=None, **kwargs):
        if Analyzer is None:
            Analyzer = kwiiyatta.Analyzer
        kwargs.update(frame_period=self.frame_period, mcep_order=self.mcep_order)
        return Analyzer(*args, **kwargs)

    def create_converter(self, Converter=None, **kwargs):
        if Converter is None:
            Converter = kwiiyatta.MelCepstrumConverter
        kwargs.setdefault('random_state', self.converter_seed)
        kwargs.setdefault('components', self.converter_components)
        if 'mcep_fs' not in kwargs and self.mcep_fs is not None:
            kwargs['mcep_fs'] = self.mcep_fs
        return Converter(**kwargs)

    @property
    def source_path(self):
        if self.source is None:
            self.parser.error('the following arguments are required: --source')
        return pathlib.Path(self.source)

    @property
    def target_path(self):
        if self.source is None:   # (sic) the reference tests `source` here too
            self.parser.error('the following arguments are required: --target')
        return pathlib.Path(self.target)

    def load_dataset(self):
        analyzer = functools.partial(self.create_analyzer, Analyzer=kwiiyatta.analyze_wav)
        src = kwiiyatta.WavFileDataset(self.source_path, Analyzer=analyzer)
        tgt = kwiiyatta.WavFileDataset(self.target_path, Analyzer=analyzer)
        return kwiiyatta.align(src, tgt)

    def train_converter(self, **kwargs):
        converter = self.create_converter(**kwargs)
        dataset = self.load_dataset()
        keys = sorted(dataset.keys())
        if self.skip_files is not None:
            keys = keys[self.skip_files:]
        if self.max_files is not None:
            keys = keys[:self.max_files]
        converter.train(dataset, keys)
        return converter
