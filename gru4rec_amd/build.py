"""Builds gru4rec_amd/libgru4rec_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'csrc', 'g4r_api.hip')
SRC_HOST = os.path.join(HERE, 'csrc', 'g4r_io.cpp')      # host-only translation unit (event-table loader)
OUT = os.path.join(HERE, 'libgru4rec_hip.so')
DEPS = [os.path.join(HERE, 'csrc', f) for f in ('g4r_api.hip', 'g4r_device.cuh', 'g4r_gemm.cuh', 'g4r_step_kernels.cuh',
                                                 'g4r_eval_kernels.cuh', 'g4r_sync_kernels.cuh', 'g4r_micro_kernels.cuh', 'g4r_wide_kernels.cuh', 'g4r_io.cpp',
                                                 'g4r_fwd_kernels.cuh', 'g4r_loss_kernel.cuh', 'g4r_bwd_kernels.cuh', 'g4r_update_kernels.cuh',
                                                 'g4r_host_model.hpp', 'g4r_host_create.hpp', 'g4r_host_plan.hpp', 'g4r_host_step.hpp', 'g4r_host_predict.hpp',
                                                 'g4r_host_comm.hpp', 'g4r_host_sync.hpp', 'g4r_host_debug.hpp')] + \
       [os.path.join(os.path.dirname(HERE), 'include', 'gru4rec_hip.h')]


def build(force=False, verbose=False, out=None, defs=(), flags=()):
    """`out` / `defs` / `flags`: a second library with extra -D definitions / raw hipcc flags next to the product one (kernel
    experiments, selected with G4R_LIB)."""
    if out is not None:
        return _compile(out, list(defs), verbose, list(flags))
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    return _compile(OUT, [], verbose)


# raw hipcc flags of every build.  Kernarg preload (gfx940+): the first 16 dwords of a kernel's arguments -- the descriptor and
# step-state pointers every step kernel takes -- arrive in scalar registers with the wave instead of behind an s_load from the
# kernarg segment: one scalar memory round trip (~0.4 us) less at the head of EVERY launch.  Measured (profiles/r04_experiments.md):
# cfg #2 kernel sum 43.8 -> 41.8 us, 21.5 -> 21.9 K mb/s; cfg #4 5.62 -> 5.75 K.  Code objects keep the compatibility prologue hipcc
# emits for firmware without the feature.
COMMON_FLAGS = ['-mllvm', '-amdgpu-kernarg-preload-count=16']


MUTANTS = {1: 'sparse accumulator increments x 1.01', 2: 'sparse Adagrad steps x 1.01', 3: 'dense accumulator increments x 1.01',
           4: "round 3's stale-register pipeline of gemm_tile2k (tied wait operands in two branches): fails the ISA audit, so it is the "
              "one library built with audit=False",
           5: 'the 1 / nranks factor of the exact-replica joint update (REDUCE / MEAN forms) x 1.01',
           6: 'the Adagrad step of ONE item row per step (the item of score column 0) x 1.5: a single wrong row must not pass'}


def mutant_path(k):
    return os.path.join(HERE, '_variants', 'libgru4rec_hip_mut%d.so' % k)


def build_mutants(force=False, verbose=False):
    """The deliberately wrong libraries of tests/test_gpu_mutation.py (-DG4R_MUTATE=k, g4r_device.cuh): the parity suite has to
    turn red on each of them.  Test infrastructure: the product never loads them.  Built concurrently (one hipcc each)."""
    import fcntl
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(os.path.join(HERE, '_variants'), exist_ok=True)

    def stale():
        return [k for k in MUTANTS if force or not os.path.exists(mutant_path(k)) or
                any(os.path.getmtime(mutant_path(k)) < os.path.getmtime(d) for d in DEPS)]
    if not stale():
        return [mutant_path(k) for k in MUTANTS]
    # one builder at a time (pytest-xdist workers share the tree: two of them building into the same _build/ directory lose each
    # other's intermediates); whoever comes second finds the libraries fresh
    with open(os.path.join(HERE, '_variants', '.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        todo = stale()
        if todo:
            _host_object(verbose)
            with ThreadPoolExecutor(min(len(todo), 2)) as ex:      # two hipcc at a time (each peaks at ~2 GB)
                list(ex.map(lambda k: _device(mutant_path(k), ['G4R_MUTATE=%d' % k], verbose, audit=(k != 4)), todo))
    return [mutant_path(k) for k in MUTANTS]


def _host_object(verbose):
    obj = os.path.join(HERE, 'csrc', 'g4r_io.o')      # host-only unit: plain g++, linked into the same library
    host = [os.environ.get('CXX', 'g++'), '-O3', '-std=c++17', '-fPIC', '-pthread', '-c', SRC_HOST, '-o', obj]
    if verbose:
        print(' '.join(host))
    subprocess.check_call(host)
    return obj


class AuditError(RuntimeError):
    """The generated code uses the destination of a hand-counted asm load before the wait that retires it (isa_audit.py)."""


def hipcc_version():
    rocm = os.environ.get('ROCM_PATH', '/opt/rocm')
    try:
        out = subprocess.run([os.path.join(rocm, 'bin', 'hipcc'), '--version'], capture_output=True, text=True).stdout
    except OSError:
        return 'unknown'
    hip = [ln for ln in out.splitlines() if ln.startswith('HIP version')]
    return hip[0].split(':', 1)[1].strip() if hip else 'unknown'


def build_dir(OUT):
    """Where the device listing (`*.s`) and the per-kernel resource table of library OUT are kept (git-ignored)."""
    return os.path.join(HERE, '_build', os.path.splitext(os.path.basename(OUT))[0])


def kernels_without_device_code(lib, device_kernels):
    """Launch stubs of the host code (`__device_stub__<kernel>` symbols) whose kernel is not among the device listing's: hipcc instantiates a
    __global__ template on the device side only where an explicit instantiation asks for it, the host side wherever a launch names it --
    such a library links, loads and fails at the first launch of that kernel (`invalid device function`)."""
    import re
    blob = open(lib, 'rb').read()
    out = []
    for mt in set(re.findall(rb'_Z(\d+)__device_stub__(\w+)', blob)):
        n = int(mt[0]) - len('__device_stub__')
        name, rest = mt[1][:n].decode(), mt[1][n:].decode()
        sym = '_Z%d%s%s' % (n, name, rest)
        if sym not in device_kernels:
            out.append(sym)
    return sorted(out)


def _device(OUT, defs, verbose, audit=True, flags=()):
    """hipcc -> OUT.  The device listing hipcc assembles into the library (-save-temps) is kept under _build/ and AUDITED
    (isa_audit.audit): a library whose code names the destination register of a hand-counted asm load before the wait that
    retires it is deleted and AuditError raised -- another compiler version cannot silently produce a wrong library (round 3's
    stale-register bug passed 250 parity tests).  `audit=False` exists for the test build of exactly that bug (mutant 4)."""
    import glob
    import json
    import shutil
    from . import isa_audit
    rocm = os.environ.get('ROCM_PATH', '/opt/rocm')
    obj = os.path.join(HERE, 'csrc', 'g4r_io.o')
    trace = ['-DG4R_CLK_TRACE'] if os.environ.get('G4R_BUILD_CLK') else []      # in-kernel phase traces for tools/clk*.py
    tmp = build_dir(OUT)
    shutil.rmtree(tmp, ignore_errors=True)
    os.makedirs(tmp)
    ver = hipcc_version()
    cmd = [os.path.join(rocm, 'bin', 'hipcc'), '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-save-temps=obj',
           '-DG4R_HIPCC_VERSION="%s"' % ver] + COMMON_FLAGS + list(flags) + trace + ['-D' + d for d in defs] + [
           '-I' + os.path.join(rocm, 'include'), '-o', os.path.join(tmp, os.path.basename(OUT)), SRC, '-Wl,' + obj, '-pthread',
           '-L' + os.path.join(rocm, 'lib'), '-lrccl', '-Wl,-rpath,' + os.path.join(rocm, 'lib')]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    listing = glob.glob(os.path.join(tmp, '*amdgcn*gfx950.s'))
    if len(listing) != 1:
        raise AuditError('no device listing next to the library (%s): cannot audit the asm loads' % tmp)
    asm = open(listing[0]).read()
    findings, counted = isa_audit.audit(asm)
    with open(os.path.join(tmp, 'resources.json'), 'w') as f:
        json.dump({'hipcc': ver, 'validated_hipcc': isa_audit.VALIDATED_HIPCC, 'asm_register_loads': counted,
                   'audit_findings': [list(x) for x in findings], 'kernels': isa_audit.resources(asm)}, f, indent=1, sort_keys=True)
    for junk in glob.glob(os.path.join(tmp, '*')):      # keep the listing and the table, drop the other intermediates
        if not (junk.endswith('gfx950.s') or junk.endswith('resources.json') or junk.endswith('.so')):
            os.remove(junk)
    built = os.path.join(tmp, os.path.basename(OUT))
    missing = kernels_without_device_code(built, isa_audit.resources(asm))
    if missing:
        os.remove(built)
        if os.path.exists(OUT):
            os.remove(OUT)
        raise AuditError('the host code launches %d kernel(s) the device code does not hold (a __global__ template is only emitted for an '
                         'explicit instantiation: g4r_eval_kernels.cuh), first: %s -- library NOT installed' % (len(missing), missing[0]))
    if findings and audit:
        os.remove(built)
        if os.path.exists(OUT):
            os.remove(OUT)
        raise AuditError('%d premature use(s) of an asm load destination in the code %s generated, first: %s line %d `%s` names '
                         'in-flight v%s -- library NOT installed (gru4rec_amd/isa_audit.py)' % ((len(findings), ver) + tuple(findings[0])))
    if not any('k_score_bwd2' in k for k in counted) and audit:
        os.remove(built)
        raise AuditError('the audit no longer sees the asm loads of k_score_bwd2 it was written for: %s' % sorted(counted))
    os.replace(built, OUT)
    return OUT


def _compile(OUT, defs, verbose, flags=()):
    _host_object(verbose)
    return _device(OUT, defs, verbose, flags=flags)


if __name__ == '__main__':
    # python -m gru4rec_amd.build [--force] | --variant <path.so> NAME=VALUE ... [-flag ...]
    if '--variant' in sys.argv:
        k = sys.argv.index('--variant')
        rest = sys.argv[k + 2:]
        print(build(out=os.path.abspath(sys.argv[k + 1]), defs=[a for a in rest if not a.startswith('-')],
                    flags=[a for a in rest if a.startswith('-')], verbose=True))
    else:
        print(build(force='--force' in sys.argv, verbose=True))
