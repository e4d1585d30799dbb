"""Builds libairvision_hip.so in-tree with hipcc for gfx950 (no JIT cache, no torch extension).

    python -m uav_airvision_amd.build [--force]

Flags that matter for parity: -ffp-contract=off (no FMA fusion: float expressions round as written,
the same rule the CPU oracle is compiled with) and correctly rounded fp32 divide/sqrt -- for the image kernels, whose
results are compared bit for bit.  The fp64 filter (msckf.hip) is compared at 1e-6 against numpy / LAPACK, which fuse
and reorder as they please: it is built with -ffp-contract=fast (v_fma_f64 instead of v_mul_f64 + v_add_f64).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(HERE, 'libairvision_hip.so')
SOURCES = ['ops_api.hip', 'pyramid.hip', 'lk.hip', 'fast.hip', 'frontend.hip', 'msckf.hip', 'png_read.hip']
HEADERS = ['av_common.h', 'msckf_batch.inc', 'msckf_mfma.inc', 'msckf_store.h', 'msckf_dev.inc', 'msckf_dev_host.inc', os.path.join('..', '..', 'include', 'airvision.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden',
         '-ffp-contract=off', '-fhip-fp32-correctly-rounded-divide-sqrt', '-fno-fast-math', '-fopenmp',
         '-Wall', '-Wno-unused-function']
FLAGS_BY_SOURCE = {'msckf.hip': {'-ffp-contract=off': '-ffp-contract=fast'}}      # per-source replacements


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objdir = os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    objs, procs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(objdir, src.replace('.hip', '.o'))
        objs.append(op)
        if force or _stale(op, [sp] + hdrs + [os.path.abspath(__file__)]):
            swap = FLAGS_BY_SOURCE.get(src, {})
            cmd = [_hipcc()] + [swap.get(f, f) for f in FLAGS] + ['-c', sp, '-o', op]
            if verbose:
                print(' '.join(cmd))
            procs.append((src, subprocess.Popen(cmd)))
    failed = [s for s, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError('hipcc failed for: ' + ', '.join(failed))
    if force or procs or _stale(OUT, objs):
        cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-fopenmp', '-Wl,-rpath,/opt/rocm/lib/llvm/lib', '-o', OUT] + objs + ['-lz', '-ldl']
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
