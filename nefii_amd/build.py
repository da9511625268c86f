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
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-ffp-contract=off', '-Wall', '-Wno-unused-function',
         # no packed-fp32 VALU instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32): on gfx950 they compute wrong
         # results while their wave shares a SIMD with MFMA-streaming waves of another kernel (csrc/mlp_tile.h,
         # NEFII_CLAIM_SIMD; tools/concurrency_probe.py).  The host pass does not know the feature and says so: -Wno-...
         '-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']


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


def _run_quietly(cmd):
    """check_call that drops the host pass's note about the device-only target feature"""
    r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    for ln in r.stderr.splitlines():
        if 'packed-fp32-ops' not in ln:
            print(ln, file=sys.stderr)
    if r.returncode:
        raise subprocess.CalledProcessError(r.returncode, cmd)


CANARY_OUT = os.path.join(CSRC, 'libnefii_canary.so')


def build_canary(force=False, verbose=True):
    """TEST INFRASTRUCTURE, never loaded by nefii_amd: the shading kernels compiled WITH packed-fp32 instructions - the
    form in which nefii_mis_sample computed wrong directions beside the tracer's evaluators (mlp_tile.h, NEFII_CLAIM_SIMD).
    tests/test_gpu_concurrency.py runs it beside every evaluator of the product library and demands bit-identical results:
    it fails on gfx950 if an evaluator stops claiming its SIMDs."""
    src = os.path.join(CSRC, 'nefii_shading.hip')
    forms = os.path.join(HERE, '..', 'tests', 'canary', 'pk_forms.hip')       # packed-fp32 instruction forms, one kernel each
    srcs = [src] + ([forms] if os.path.exists(forms) else [])
    if not force and os.path.exists(CANARY_OUT) and os.path.getmtime(CANARY_OUT) >= max(
            [os.path.getmtime(f) for f in srcs] + [os.path.getmtime(os.path.abspath(__file__))]):
        return CANARY_OUT
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc] + [f for f in FLAGS if f not in ('-Xclang', '-target-feature', '-packed-fp32-ops')] + \
        ['-shared'] + srcs + ['-o', CANARY_OUT]
    if verbose:
        print(' '.join(cmd), flush=True)
    _run_quietly(cmd)
    return CANARY_OUT


def build(force=False, verbose=True):
    build_host(force, verbose)
    build_canary(force, verbose)
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
        _run_quietly(cmd)
        objs.append(o)
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
