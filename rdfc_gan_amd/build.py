"""Ahead-of-time build of librdfc_hip.so for gfx950 (hipcc cross-compiles without a GPU).

    python -m rdfc_gan_amd.build [--force]

Objects go to rdfc_gan_amd/csrc/build/, the library to rdfc_gan_amd/librdfc_hip.so
(in-tree so it travels to the GPU box; git-ignored so history stays source-only).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, 'build')
LIB = os.path.join(HERE, 'librdfc_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-Wno-unused-value',
         '-Wno-pass-failed']


def source_hash():
    """sha256 (first 16 hex digits) over the kernel sources: stamps measurement artefacts (profiles/*_pmc_traffic.json) with
    the code they were taken from, so that bench.py can refuse counters of another build"""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(('.hip', '.h'))]
    files.append(os.path.join(os.path.dirname(HERE), 'include', 'rdfc_hip.h'))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hdrs.append(os.path.join(os.path.dirname(HERE), 'include', 'rdfc_hip.h'))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, hdr_mtime, force):
    obj = os.path.join(OBJ, src[:-4] + '.o')
    spath = os.path.join(CSRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(spath), hdr_mtime):
        return obj, False
    subprocess.check_call([HIPCC] + FLAGS + ['-c', spath, '-o', obj])
    return obj, True


def _hash_object(force):
    """rdfc_build_hash(): the hash of the kernel sources compiled INTO the library (bench.py compares measurement artefacts with
    what is loaded, not with the files next to it; tests/test_abi_cpu.py requires it to equal source_hash(), so a stale
    binary beside fresh sources is caught).  -> (object, rebuilt)"""
    h = source_hash()
    src, obj = os.path.join(OBJ, 'build_hash.cpp'), os.path.join(OBJ, 'build_hash.o')
    text = f'extern "C" const char *rdfc_build_hash(void) {{ return "{h}"; }}\n'
    if not force and os.path.exists(obj) and os.path.exists(src) and open(src).read() == text:
        return obj, False
    with open(src, 'w') as f:
        f.write(text)
    subprocess.check_call([os.environ.get('CXX', 'g++'), '-O2', '-fPIC', '-c', src, '-o', obj])
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hdr = _deps_mtime()
    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 2)) as ex:
        res = list(ex.map(lambda s: _compile(s, hdr, force), _sources()))
    res.append(_hash_object(force))
    objs = [o for o, _ in res]
    if force or any(ch for _, ch in res) or not os.path.exists(LIB):
        subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs)
        if verbose:
            print(f'[rdfc_gan_amd.build] linked {LIB}')
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
