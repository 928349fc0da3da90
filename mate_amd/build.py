"""Builds the HIP engine in-tree: mate_amd/lib/libmate_engine.so (gfx950 only).

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the numerics
contract (see csrc/device_math.hpp); explicit fma() marks the fused products.
"""
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'csrc', 'mate_engine.hip')
DEPS = [SRC] + [os.path.join(HERE, 'csrc', f) for f in ('engine_kernels.hpp', 'reset_kernels.hpp', 'policy_kernels.hpp', 'aux_kernels.hpp', 'device_math.hpp', 'experiments.hpp',
                                                          'shape_groups.hpp', 'shape_group.inc', 'shape_group.hip')] + [
    os.path.join(os.path.dirname(HERE), 'include', 'mate_engine.h')]
OUT = os.path.join(HERE, 'lib', 'libmate_engine.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared',
         # MachineLICM hoists every literal (polynomial coefficients, masks) out of the phase loops and pins
         # ~50 VGPRs for the whole kernel; without it step_kernel needs 52 VGPRs instead of 100
         '-mllvm', '-disable-machine-licm']


STAMP = os.path.join(HERE, 'lib', 'build_stamp.json')


def source_digest():
    """sha256 over the sources and the flags the library is built from."""
    import hashlib
    h = hashlib.sha256(' '.join(FLAGS).encode())
    for d in DEPS:
        with open(d, 'rb') as fh:
            h.update(os.path.basename(d).encode() + b'\0' + fh.read())
    return h.hexdigest()


def needs_build():
    """True unless the library in the tree was built from exactly these sources.  By CONTENT, not by modification time: a snapshot of
    the tree copied to another machine (gpurun, the driver's GPU tier) does not keep the order of the files' time stamps, and a
    time-based rule rebuilt the whole library there -- 40 s in front of every smoke() -- for nothing."""
    if not os.path.exists(OUT) or not os.path.exists(STAMP):
        return True
    try:
        with open(STAMP) as fh:
            return json.load(fh).get('sources') != source_digest()
    except (OSError, ValueError):
        return True


RESOURCES = os.path.join(HERE, 'lib', 'kernel_resources.json')
_REMARK = re.compile(r'remark:\s+(Function Name|TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Dynamic Stack|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill): (\S+)')


def parse_resources(text):
    """Per-kernel register / scratch / occupancy figures from the compiler's kernel-resource-usage remarks."""
    kernels, current = {}, None
    for key, value in _REMARK.findall(text):
        if key == 'Function Name':
            current = kernels.setdefault(value, {})
        elif current is not None:
            current[key.split(' [')[0]] = value if key == 'Dynamic Stack' else int(value)
    return kernels


N_SHAPE_GROUPS = 6      # csrc/shape_groups.hpp: the compiled scenario shapes in six groups, one translation unit each


def _compile_and_link(out, extra=(), verbose=False, remarks=False):
    """The engine as seven translation units compiled in parallel -- mate_engine.hip (host side, generic kernels, reset) and one
    shape_group.hip per group of compiled scenario shapes -- and linked into `out`.  Returns the compilers' stderr (remarks)."""
    import concurrent.futures
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objdir = os.path.join(os.path.dirname(OUT), 'obj', os.path.splitext(os.path.basename(out))[0])
    os.makedirs(objdir, exist_ok=True)
    compile_flags = [f for f in FLAGS if f != '-shared'] + list(extra) + ['-DMATE_SPLIT_BUILD'] + (['-Rpass-analysis=kernel-resource-usage'] if remarks else [])
    units = [(SRC, [], os.path.join(objdir, 'mate_engine.o'))]
    units += [(os.path.join(HERE, 'csrc', 'shape_group.hip'), [f'-DMATE_SHAPE_GROUP={k}'], os.path.join(objdir, f'shape_group_{k}.o')) for k in range(N_SHAPE_GROUPS)]

    def compile_unit(unit):
        src, defs, obj = unit
        cmd = [hipcc] + compile_flags + defs + ['-c', '-o', obj, src]
        if verbose:
            print(' '.join(cmd), flush=True)
        done = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
        return cmd, done

    workers = max(1, min(len(units), int(os.environ.get('MATE_BUILD_JOBS', '0')) or (os.cpu_count() or 2)))
    with concurrent.futures.ThreadPoolExecutor(max_workers=workers) as pool:
        results = list(pool.map(compile_unit, units))
    text = ''
    for cmd, done in results:
        if done.returncode != 0:
            sys.stderr.write(done.stderr)
            raise subprocess.CalledProcessError(done.returncode, cmd)
        text += done.stderr
    link = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + [u[2] for u in units]
    if verbose:
        print(' '.join(link), flush=True)
    done = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    if done.returncode != 0:
        sys.stderr.write(done.stderr)
        raise subprocess.CalledProcessError(done.returncode, link)
    return text


def build_engine(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    kernels = parse_resources(_compile_and_link(OUT, verbose=verbose, remarks=True))
    with open(RESOURCES, 'w') as f:
        json.dump(kernels, f, indent=1, sort_keys=True)
    # A kernel that needs private scratch memory (a spilled register, an outlined helper that takes the environment
    # context by reference) costs ~5 us more per LAUNCH than one that does not -- a third of the headline step.
    bad = [k for k, r in kernels.items() if r.get('ScratchSize', 0) != 0 or r.get('Dynamic Stack') != 'False']
    if bad:
        os.remove(OUT)
        raise RuntimeError('kernels with private scratch memory: ' + ', '.join(bad))
    with open(STAMP, 'w') as f:
        json.dump({'sources': source_digest()}, f)
    return OUT


def build_profiling(verbose=False):
    """The profiling build (per-wave s_memtime stamps at the phase boundaries, -DMATE_PHASE_CLOCKS): lib/libmate_engine_prof.so,
    selected with MATE_ENGINE_LIB by tools/*_phases.py.  Never loaded by the package itself."""
    out = os.path.join(HERE, 'lib', 'libmate_engine_prof.so')
    _compile_and_link(out, extra=['-DMATE_PHASE_CLOCKS'], verbose=verbose)
    return out


def build_tools(verbose=False):
    """The stand-alone measurement programs under tools/ (store rooflines, issue rates): one hipcc call each, binaries next to
    the sources (git-ignored; they travel to the GPU box with the tree)."""
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools')
    built = []
    for name in ('store_roof', 'store_bits', 'store_contig', 'store_vmm', 'region_bw', 'valu_rates', 'chunk_order', 'dispatch_probe', 'va_reuse', 'store_layout'):
        src, out = os.path.join(tools, name + '.hip'), os.path.join(tools, name)
        if not os.path.exists(src) or (os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src)):
            continue
        cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-I' + os.path.dirname(SRC), '-o', out, src]
        if verbose:
            print(' '.join(cmd))
        subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        built.append(out)
    return built


def build_variant(name, defines, verbose=False):
    """An experiment build, lib/libmate_engine_<name>.so with extra -D flags (A/B partners of tools/rollout_ab.py and of
    MATE_ENGINE_LIB=...; never loaded by the package itself)."""
    out = os.path.join(HERE, 'lib', 'libmate_engine_%s.so' % name)
    _compile_and_link(out, extra=list(defines), verbose=verbose)
    return out


if __name__ == '__main__':
    if '--variant' in sys.argv:      # python -m mate_amd.build --variant NAME -DFLAG[=V] ...
        build_variant(sys.argv[sys.argv.index('--variant') + 1], [a for a in sys.argv if a.startswith('-D')], verbose=True)
        sys.exit(0)
    build_engine(force='--force' in sys.argv, verbose=True)
    if '--tools' in sys.argv:
        build_tools(verbose=True)
    if '--prof' in sys.argv:
        build_profiling(verbose=True)
