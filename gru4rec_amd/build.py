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


MUTANTS = {1: 'sparse accumulator increments x 1.01', 2: 'sparse Adagrad steps x 1.01', 3: 'dense accumulator increments x 1.01'}


def mutant_path(k):
    return os.path.join(HERE, '_variants', 'libgru4rec_hip_mut%d.so' % k)


def build_mutants(force=False, verbose=False):
    """The deliberately wrong libraries of tests/test_gpu_mutation.py (-DG4R_MUTATE=k, g4r_device.cuh): the parity suite has to
    turn red on each of them.  Test infrastructure: the product never loads them.  Built concurrently (one hipcc each)."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(os.path.join(HERE, '_variants'), exist_ok=True)
    todo = [k for k in MUTANTS if force or not os.path.exists(mutant_path(k)) or
            any(os.path.getmtime(mutant_path(k)) < os.path.getmtime(d) for d in DEPS)]
    if not todo:
        return [mutant_path(k) for k in MUTANTS]
    _host_object(verbose)
    with ThreadPoolExecutor(len(todo)) as ex:
        list(ex.map(lambda k: _device(mutant_path(k), ['G4R_MUTATE=%d' % k], verbose), todo))
    return [mutant_path(k) for k in MUTANTS]


def _host_object(verbose):
    obj = os.path.join(HERE, 'csrc', 'g4r_io.o')      # host-only unit: plain g++, linked into the same library
    host = [os.environ.get('CXX', 'g++'), '-O3', '-std=c++17', '-fPIC', '-pthread', '-c', SRC_HOST, '-o', obj]
    if verbose:
        print(' '.join(host))
    subprocess.check_call(host)
    return obj


def _device(OUT, defs, verbose):
    rocm = os.environ.get('ROCM_PATH', '/opt/rocm')
    obj = os.path.join(HERE, 'csrc', 'g4r_io.o')
    trace = ['-DG4R_CLK_TRACE'] if os.environ.get('G4R_BUILD_CLK') else []      # in-kernel phase traces for tools/clk*.py
    cmd = [os.path.join(rocm, 'bin', 'hipcc'), '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC'] + trace + ['-D' + d for d in defs] + [
           '-I' + os.path.join(rocm, 'include'), '-o', OUT, SRC, '-Wl,' + obj, '-pthread', '-L' + os.path.join(rocm, 'lib'), '-lrccl',
           '-Wl,-rpath,' + os.path.join(rocm, 'lib')]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return OUT


def _compile(OUT, defs, verbose):
    _host_object(verbose)
    return _device(OUT, defs, verbose)


if __name__ == '__main__':
    # python -m gru4rec_amd.build [--force] | --variant <path.so> NAME=VALUE ...
    if '--variant' in sys.argv:
        k = sys.argv.index('--variant')
        print(build(out=os.path.abspath(sys.argv[k + 1]), defs=sys.argv[k + 2:], verbose=True))
    else:
        print(build(force='--force' in sys.argv, verbose=True))
