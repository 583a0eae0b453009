"""Audit of the hand-counted register loads in the gfx950 code hipcc generates for the library (guide section 5.7, item 1).

An asm `global_load_*` whose destination is a VGPR is invisible to hipcc's s_waitcnt bookkeeping AND to its idea of when the
destination is written: the compiler regards the register as defined when the asm statement ends and may copy, read or reuse it
before the data lands.  The kernels that hide loads from hipcc (gemm_tile2k's two-chunks-ahead pipeline, the stream-K fix-up, the
peer-memory all-reduce) wait for them with their own `s_waitcnt vmcnt(N)` statements; this audit walks every kernel of the
device assembly in program order and reports any COMPILER-generated instruction that names a destination register of an asm load
between that load and the asm wait that retires it (loads return in order: an asm `vmcnt(N)` retires all but the newest N asm
loads; any compiler `s_waitcnt vmcnt(0)` retires everything).  Program order is the linear order of the listing -- exact for the
loop shapes in this library, where the state at a loop's back edge equals the state at its entry.

    python tools/isa_audit.py [file.s]        (without a file: compiles gru4rec_amd/csrc/g4r_api.hip with --cuda-device-only -S)

gru4rec_amd/build.py runs `audit` on the assembly of EVERY library it links (hipcc -save-temps: the listing that is assembled
into the shipped code object) and refuses to return a library with a finding: a different hipcc cannot silently produce a wrong
library.  The compiler the pattern was last validated against is recorded in VALIDATED_HIPCC; another one still builds (the
audit is what decides), the version string it was built with is in g4r_version().
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VALIDATED_HIPCC = 'HIP version: 7.2.26015-fc0010cf6a'      # ROCm 7.2.0 (clang 22.0.0git roc-7.2.0 26014)
REG = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')


def device_asm(path=None, defs=()):
    if path:
        return open(path).read()
    out = os.path.join(tempfile.mkdtemp(prefix='g4r_isa_'), 'g4r.s')
    hipcc = os.path.join(os.environ.get('ROCM_PATH', '/opt/rocm'), 'bin', 'hipcc')
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '--cuda-device-only', '-S', '-mllvm', '-amdgpu-kernarg-preload-count=16'] + ['-D' + d for d in defs] + [
                           os.path.join(ROOT, 'gru4rec_amd', 'csrc', 'g4r_api.hip'), '-o', out], stderr=subprocess.DEVNULL)
    return open(out).read()


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def audit(asm):
    """[(kernel, line number within the listing, instruction, in-flight registers it names)], {kernel: number of asm loads}"""
    findings, counted = [], {}
    kernel, in_asm, flight = None, False, []      # flight: [(set of destination registers, line)] oldest first
    for ln, raw in enumerate(asm.splitlines(), 1):
        line = raw.strip()
        m = re.match(r'^(_Z\w+|k_\w+):\s*(;.*)?$', line)
        if m:
            kernel, in_asm, flight = m.group(1), False, []
            continue
        if kernel is None or not line:
            continue
        if line.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if line.startswith(';;#ASMEND'):
            in_asm = False
            continue
        if line.startswith(';') or line.startswith('.'):
            if line.startswith('.size') or line.startswith('.Lfunc_end'):
                kernel = None
            continue
        ins = line.split(';')[0].strip()
        op = ins.split()[0] if ins else ''
        if in_asm:
            if op.startswith('global_load') or op.startswith('buffer_load') or op.startswith('flat_load'):
                if ' lds' in ins or op.startswith('global_load_lds'):
                    continue                      # LDS-DMA: no register destination
                dst = ins[len(op):].split(',')[0]
                flight.append((regs(dst), ln))
                counted[kernel] = counted.get(kernel, 0) + 1
            else:
                w = re.match(r's_waitcnt\s+.*vmcnt\((\d+)\)', ins)
                if w:
                    n = int(w.group(1))
                    flight = flight[len(flight) - n:] if n else []
            continue
        w = re.match(r's_waitcnt\s+.*vmcnt\((\d+)\)', ins)
        if w and int(w.group(1)) == 0:
            flight = []
            continue
        if not flight or op in ('s_endpgm',):
            if op == 's_endpgm':
                flight = []
            continue
        hot = set().union(*[r for r, _ in flight])
        named = regs(ins) & hot
        if named:
            findings.append((kernel, ln, ins, sorted(named)))
    return findings, counted


def resources(asm):
    """{kernel symbol: {vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, scratch, lds}} from the code-object metadata at the end of the
    listing (what `tools/kernel_resources.sh` reads from the compiler remarks; here it comes with the build)."""
    out, cur = {}, None
    keys = {'.agpr_count': 'agpr', '.vgpr_count': 'vgpr', '.sgpr_count': 'sgpr', '.vgpr_spill_count': 'vgpr_spill',
            '.sgpr_spill_count': 'sgpr_spill', '.private_segment_fixed_size': 'scratch', '.group_segment_fixed_size': 'lds'}
    i = asm.rfind('amdhsa.kernels:')
    if i < 0:
        return out
    for raw in asm[i:].splitlines()[1:]:
        line = raw.strip()
        if line.startswith('- .') or line.startswith('-   .'):
            if cur is not None and 'name' in cur:
                out[cur.pop('name')] = cur
            cur = {}
            line = line[1:].strip()
        if cur is None:
            continue
        m = re.match(r'^(\.\w+):\s+(\S+)$', line)
        if m and m.group(1) in keys and raw.startswith('    .') or (m and raw.startswith('  - .') and m.group(1) in keys):
            cur[keys[m.group(1)]] = int(m.group(2))
        elif m and m.group(1) == '.name' and raw.startswith('    .'):
            cur['name'] = m.group(2)
        if line.startswith('amdhsa.target'):
            break
    if cur is not None and 'name' in cur:
        out[cur.pop('name')] = cur
    return out


def main(argv=None):
    argv = sys.argv if argv is None else argv
    f, c = audit(device_asm(argv[1] if len(argv) > 1 else None))
    for k, n in sorted(c.items()):
        print('%-60s %3d asm register loads' % (k[:60], n))
    for k, ln, ins, named in f:
        print('VIOLATION %s line %d: `%s` names in-flight v%s' % (k, ln, ins, named))
    print('%d violation(s)' % len(f))
    sys.exit(1 if f else 0)


if __name__ == '__main__':
    main()
