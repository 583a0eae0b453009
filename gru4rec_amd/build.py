"""Builds gru4rec_amd/libgru4rec_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'csrc', 'g4r_api.hip')
SRC_HOST = os.path.join(HERE, 'csrc', 'g4r_io.cpp')      # host-only translation unit (event-table loader)
OUT = os.path.join(HERE, 'libgru4rec_hip.so')
DEPS = [os.path.join(HERE, 'csrc', f) for f in ('g4r_api.hip', 'g4r_device.cuh', 'g4r_gemm.cuh', 'g4r_step_kernels.cuh',
                                                 'g4r_eval_kernels.cuh', 'g4r_sync_kernels.cuh', 'g4r_micro_kernels.cuh', 'g4r_io.cpp')] + \
       [os.path.join(os.path.dirname(HERE), 'include', 'gru4rec_hip.h')]


def build(force=False, verbose=False, out=None, defs=()):
    """`out` / `defs`: a second library with extra -D flags next to the product one (kernel experiments, selected with G4R_LIB)."""
    if out is not None:
        return _compile(out, list(defs), verbose)
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    return _compile(OUT, [], verbose)


def _compile(OUT, defs, verbose):
    rocm = os.environ.get('ROCM_PATH', '/opt/rocm')
    obj = os.path.join(HERE, 'csrc', 'g4r_io.o')      # host-only unit: plain g++, linked into the same library
    host = [os.environ.get('CXX', 'g++'), '-O3', '-std=c++17', '-fPIC', '-pthread', '-c', SRC_HOST, '-o', obj]
    if verbose:
        print(' '.join(host))
    subprocess.check_call(host)
    trace = ['-DG4R_CLK_TRACE'] if os.environ.get('G4R_BUILD_CLK') else []      # in-kernel phase traces for tools/clk*.py
    cmd = [os.path.join(rocm, 'bin', 'hipcc'), '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC'] + trace + ['-D' + d for d in defs] + [
           '-I' + os.path.join(rocm, 'include'), '-o', OUT, SRC, '-Wl,' + obj, '-pthread', '-L' + os.path.join(rocm, 'lib'), '-lrccl',
           '-Wl,-rpath,' + os.path.join(rocm, 'lib')]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    # python -m gru4rec_amd.build [--force] | --variant <path.so> NAME=VALUE ...
    if '--variant' in sys.argv:
        k = sys.argv.index('--variant')
        print(build(out=os.path.abspath(sys.argv[k + 1]), defs=sys.argv[k + 2:], verbose=True))
    else:
        print(build(force='--force' in sys.argv, verbose=True))
