"""Build the C-ABI HIP library (libcprhip.so) for gfx950, in-tree.

``python -m pointtinybenchmark_amd.build`` or ``__graft_entry__.build()``.  hipcc cross-compiles
without a GPU; the .so is git-ignored but travels with the gpurun snapshot.
"""
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
LIB = os.path.join(CSRC, 'libcprhip.so')
SOURCES = ['conv_mfma.hip', 'conv1x1_stream.hip', 'conv_mfma_bf16.hip', 'conv_bf16_dma.hip', 'conv_bf16_pp.hip', 'stem_bf16.hip', 'stem_f32.hip', 'conv_wgrad_bf16.hip', 'conv_wgrad_bf16_tn.hip', 'conv_wgrad.hip', 'norm_pool.hip', 'cpr_points.hip', 'assign.hip',
           'postproc.hip', 'backward.hip', 'preprocess.hip', 'pack.hip', 'project.hip', 'conv_wino.hip', 'conv_wino32.hip', 'conv_wino_wgrad.hip']
HEADERS = ['common.h', 'conv_bf16_dma.h']
# Kernels whose integer / mask / index outputs are held bit-exact against the reference's CPU arithmetic restate it
# operation by operation.  hipcc's default -ffp-contract=fast fuses a*b+c into one fma EVEN ACROSS the __fmul_rn/__fadd_rn
# intrinsics (plain operators in the HIP headers), which changes the last bit (round 1 shipped a Hungarian cost whose
# pos - neg had become fma(t, q, -neg): 24 % of the entries 1 ulp off).  These files are compiled with contraction off;
# where the reference itself uses an fma, the source says __fmaf_rn explicitly.
EXACT_SOURCES = {'cpr_points.hip', 'assign.hip', 'postproc.hip', 'preprocess.hip'}


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


BENCH_LIB = os.path.join(CSRC, 'libcprhip_bench.so')
HOOK_SOURCES = {'conv_mfma.hip', 'conv_wgrad.hip', 'conv_wino.hip', 'conv_mfma_bf16.hip', 'conv_bf16_dma.hip', 'conv_bf16_pp.hip', 'assign.hip', 'conv_wino32.hip'}     # the files that carry #ifdef CPR_BENCH_HOOKS code


def build(force=False, verbose=True, bench_hooks=False):
    """bench_hooks=True builds libcprhip_bench.so (-DCPR_BENCH_HOOKS: forced tiles, schedule A/B, loop ablations) for
    tools/*.py next to the product library; the product library never contains those switches."""
    lib = BENCH_LIB if bench_hooks else LIB
    if not force and not (needs_build() if not bench_hooks else (
            not os.path.exists(lib) or any(os.path.getmtime(os.path.join(CSRC, s)) > os.path.getmtime(lib)
                                           for s in SOURCES + HEADERS))):
        return lib
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            continue
        hooked = bench_hooks and s in HOOK_SOURCES
        obj = os.path.join(CSRC, s.replace('.hip', '.bench.o' if hooked else '.o'))
        if bench_hooks and not hooked and os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(src) \
                and all(os.path.getmtime(obj) >= os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS):
            objs.append(obj)          # identical object as in the product build
            continue
        cmd = [_hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC'] + \
            (['-ffp-contract=off'] if s in EXACT_SOURCES else []) + (['-DCPR_BENCH_HOOKS'] if hooked else []) + \
            ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
    if '--bench-hooks' in sys.argv:
        print(build(force='--force' in sys.argv, bench_hooks=True))
