"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/rdfc_hip.h declares.  No compute calls (no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'rdfc_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(rdfc_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_header():
    from rdfc_gan_amd import build, hip
    build.build(verbose=False)
    lib = hip.lib()
    names = declared_symbols()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f'declared in rdfc_hip.h but not exported: {missing}'
    assert lib.rdfc_version() >= 100


def test_argument_validation_without_gpu():
    """Rejected descriptors return an error code + message before any launch."""
    from rdfc_gan_amd import hip
    lib = hip.lib()
    d = hip.ConvDesc(1, 4, 4, 8, 4, 4, 8, 3, 3, 1, 1, 1, 1, 1, 1, 4, 8, 0, 0, 0.0, 0)   # ldx < Ci
    rc = lib.rdfc_conv2d_forward(ctypes.byref(d), None, None, None, None, None)
    assert rc != 0 and b'ld smaller' in lib.rdfc_last_error()
    md = hip.MDConvDesc(1, 3, 4, 4, 4, 3, 3, 1, 1, 1, 1, 1, 1, 2, 1)                        # C % group != 0
    rc = lib.rdfc_mdconv_forward(ctypes.byref(md), None, None, None, None, None, None, None)
    assert rc != 0 and b'must divide group' in lib.rdfc_last_error()


def test_product_does_not_import_oracle():
    """The product package must never route through the oracle / a CPU fallback."""
    pkg = os.path.join(ROOT, 'rdfc_gan_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', txt, flags=re.M), f'{f} imports oracle'


def test_loaded_library_was_built_from_these_sources():
    """rdfc_build_hash() is compiled into the library; it must name the kernel sources that lie next to it (a stale binary
    beside fresh sources would otherwise pass every source-level check of measurement artefacts)."""
    from rdfc_gan_amd import build, hip
    build.build(verbose=False)
    assert hip.build_hash() == build.source_hash()
