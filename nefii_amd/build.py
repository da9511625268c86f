"""Build libnefii_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m nefii_amd.build [--force]

The shared library is written next to the sources (nefii_amd/csrc/) so that it travels with the
repo snapshot to the GPU box; it is git-ignored.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
SOURCES = ['nefii_mlp.hip', 'nefii_tracer.hip', 'nefii_shading.hip']
HEADERS = ['mlp_tile.h', os.path.join('..', '..', 'include', 'nefii_amd.h')]
OUT = os.path.join(CSRC, 'libnefii_hip.so')
HOST_SRC = os.path.join(CSRC, 'exr_huf.c')          # host-only helper of utils/exr.py (PIZ Huffman loop)
HOST_OUT = os.path.join(CSRC, 'libnefii_host.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-ffp-contract=off', '-Wall', '-Wno-unused-function']


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_host(force=False, verbose=True):
    if not force and os.path.exists(HOST_OUT) and os.path.getmtime(HOST_OUT) >= os.path.getmtime(HOST_SRC):
        return HOST_OUT
    cmd = [os.environ.get('CC', 'gcc'), '-O2', '-shared', '-fPIC', '-o', HOST_OUT, HOST_SRC]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return HOST_OUT


def build(force=False, verbose=True):
    build_host(force, verbose)
    if not force and not _stale():
        return OUT
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    extra = os.environ.get('NEFII_EXTRA_HIPCC_FLAGS', '').split()
    objs = []
    for s in SOURCES:
        o = os.path.join(CSRC, s.replace('.hip', '.o'))
        cmd = [hipcc] + FLAGS + extra + ['-c', os.path.join(CSRC, s), '-o', o]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(o)
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
