"""CPU-side checks of the drop-in boundary: libkwy.so loads without a GPU and
exports every symbol include/kwy.h declares; the ctypes table mirrors it."""
import os
import re
import subprocess

import pytest

from conftest import ROOT


def header_prototypes():
    text = open(os.path.join(ROOT, 'include', 'kwy.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(kwy_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_entry_points():
    names = header_prototypes()
    for must in ('kwy_ctx_create', 'kwy_cheaptrick', 'kwy_d4c', 'kwy_synthesize'):
        assert must in names


def test_library_exports_every_declared_symbol():
    so = os.path.join(ROOT, 'kwiiyatta_amd', 'libkwy.so')
    assert os.path.exists(so), 'build first: python -c "import __graft_entry__ as g; g.build()"'
    out = subprocess.check_output(['nm', '-D', '--defined-only', so]).decode()
    exported = set(re.findall(r' T (kwy_[a-z0-9_]+)', out))
    missing = [n for n in header_prototypes() if n not in exported]
    assert not missing, f'declared in include/kwy.h but not exported: {missing}'


def test_ctypes_table_matches_header():
    from kwiiyatta_amd import _lib
    declared = set(header_prototypes())
    assert declared <= set(_lib.SIGNATURES), declared - set(_lib.SIGNATURES)
    assert not [m for m in _lib.MISSING if m in declared]


def test_size_helpers_need_no_device():
    from kwiiyatta_amd import _lib
    assert _lib.lib.kwy_cheaptrick_fft_size(48000, 71.0) == 2048
    assert _lib.lib.kwy_cheaptrick_fft_size(16000, 71.0) == 1024
    assert _lib.lib.kwy_dio_frames(48000, 480000, 5.0) == 2001
    assert _lib.lib.kwy_synth_length(2001, 5.0, 48000) == 480240


def test_no_cpu_fallback_without_gpu():
    """Without a HIP device the product refuses to run (no silent CPU path)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    from kwiiyatta_amd import _lib
    with pytest.raises(RuntimeError, match='no HIP device'):
        _lib.Context(0)


def test_product_never_imports_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'kwiiyatta_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.hpp', '.cpp', '.h', '.sh')):
                src = open(os.path.join(dirpath, f), errors='ignore').read()
                if re.search(r'\boracle\b|liboracle|ko_[a-z]+\(', src):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad
