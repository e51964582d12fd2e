"""The C-ABI library loads on a CPU-only box and exports every symbol include/nimg.h declares (no compute calls)."""
import ctypes
import os
import re

from neural_imaging_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, 'include', 'nimg.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(nimg_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_entry_points():
    syms = _header_symbols()
    assert 'nimg_djpeg_fwd' in syms and 'nimg_conv2d_fwd' in syms and len(syms) >= 30


def test_library_exports_every_declared_symbol():
    assert os.path.isfile(_lib.LIB_PATH), 'build libnimg.so first (__graft_entry__.build())'
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in _header_symbols() if not hasattr(lib, s)]
    assert not missing, 'declared in include/nimg.h but not exported: {}'.format(missing)


def test_python_prototypes_cover_the_header():
    syms = set(_header_symbols())
    assert syms == set(_lib.PROTOTYPES.keys()), syms ^ set(_lib.PROTOTYPES.keys())
    lib = _lib.load()
    header = open(os.path.join(ROOT, 'include', 'nimg.h')).read()
    declared = int(re.search(r'#define\s+NIMG_ABI_VERSION\s+(\d+)', header).group(1))
    assert lib.nimg_abi_version() == declared == _lib.ABI_VERSION       # library, header and binding agree (load() checks too)


def test_no_oracle_import_in_product():
    """The product path must never route through the oracle or a CPU fallback."""
    pkg = os.path.join(ROOT, 'neural-imaging_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', src, flags=re.M), os.path.join(dirpath, f)
