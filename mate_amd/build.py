"""Builds the HIP engine in-tree: mate_amd/lib/libmate_engine.so (gfx950 only).

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the numerics
contract (see csrc/device_math.hpp); explicit fma() marks the fused products.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'csrc', 'mate_engine.hip')
DEPS = [SRC] + [os.path.join(HERE, 'csrc', f) for f in ('engine_kernels.hpp', 'reset_kernels.hpp', 'policy_kernels.hpp', 'device_math.hpp')] + [
    os.path.join(os.path.dirname(HERE), 'include', 'mate_engine.h')]
OUT = os.path.join(HERE, 'lib', 'libmate_engine.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared',
         # MachineLICM hoists every literal (polynomial coefficients, masks) out of the phase loops and pins
         # ~50 VGPRs for the whole kernel; without it step_kernel needs 52 VGPRs instead of 100
         '-mllvm', '-disable-machine-licm']


def needs_build():
    if not os.path.exists(OUT):
        return True
    built = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > built for d in DEPS)


def build_engine(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc] + FLAGS + ['-o', OUT, SRC]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    build_engine(force='--force' in sys.argv, verbose=True)
