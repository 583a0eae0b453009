"""Static audit of the asm register loads in the generated gfx950 code (tools/isa_audit.py): no compiler-generated instruction may
name the destination of an asm `global_load` between the load and the asm `s_waitcnt` that retires it.  Background: round 3's
first two-chunks-ahead pipeline of gemm_tile2k passed every parity test on cache-resident tables and committed its last K chunk from
stale registers on a 10 GB table under load -- hipcc had placed the copies of a tied `"+v"` wait operand in front of the wait."""
import os
import shutil
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
import isa_audit      # noqa: E402

BAD = """
_Z5k_badv:
	;;#ASMSTART
	global_load_dwordx4 v[42:45], v[2:3], off
	;;#ASMEND
	v_mfma_f32_32x32x2_f32 v[2:17], v0, v50, v[2:17]
	v_mov_b64_e32 v[0:1], v[42:43]
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	v_cndmask_b32_e64 v8, 0, v0, s[54:55]
	s_endpgm
.Lfunc_end0:
"""
GOOD = """
_Z6k_goodv:
	;;#ASMSTART
	global_load_dwordx4 v[34:37], v[2:3], off
	;;#ASMEND
	;;#ASMSTART
	global_load_dwordx4 v[42:45], v[4:5], off
	;;#ASMEND
	v_mfma_f32_32x32x2_f32 v[2:17], v0, v50, v[2:17]
	;;#ASMSTART
	s_waitcnt vmcnt(1)
	;;#ASMEND
	v_cndmask_b32_e64 v8, 0, v34, s[54:55]
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	v_cndmask_b32_e64 v9, 0, v42, s[54:55]
	s_endpgm
.Lfunc_end1:
"""


def test_the_audit_flags_a_copy_in_front_of_the_wait():
    f, c = isa_audit.audit(BAD)
    assert c == {'_Z5k_badv': 1} and len(f) == 1 and f[0][3] == [42, 43]


def test_the_audit_accepts_counted_waits():
    f, c = isa_audit.audit(GOOD)
    assert c == {'_Z6k_goodv': 2} and not f
    # the same listing with the first consumer moved above its wait is flagged
    moved = GOOD.replace("\tv_cndmask_b32_e64 v8, 0, v34, s[54:55]\n", "").replace(
        "\t;;#ASMSTART\n\ts_waitcnt vmcnt(1)", "\tv_cndmask_b32_e64 v8, 0, v34, s[54:55]\n\t;;#ASMSTART\n\ts_waitcnt vmcnt(1)")
    assert len(isa_audit.audit(moved)[0]) == 1


@pytest.mark.skipif(shutil.which('hipcc') is None and not os.path.exists('/opt/rocm/bin/hipcc'), reason='needs hipcc')
def test_library_code_has_no_premature_use_of_an_asm_load():
    f, c = isa_audit.audit(isa_audit.device_asm())
    assert any('k_score_bwd2' in k for k in c), 'the audit no longer sees the pipeline it was written for: %s' % sorted(c)
    assert not f, '\n'.join('%s line %d: %s names in-flight v%s' % x for x in f[:20])


@pytest.mark.skipif(shutil.which('hipcc') is None and not os.path.exists('/opt/rocm/bin/hipcc'), reason='needs hipcc')
def test_the_build_refuses_the_stale_register_pipeline(tmp_path):
    """Mutant 4 = round 3's faulty gemm_tile2k (tied wait operands in two branches).  build._device audits the listing hipcc
    assembles into the library: it must raise and leave NO library behind; with audit=False (how the GPU stress test gets its red
    build) the same source links."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from gru4rec_amd import build
    out = str(tmp_path / 'libg4r_mut4_audit.so')
    build._host_object(False)
    with pytest.raises(build.AuditError) as e:
        build._device(out, ['G4R_MUTATE=4'], False, audit=True)
    assert 'k_score_bwd2' in str(e.value) and not os.path.exists(out)
    import json
    info = json.load(open(os.path.join(build.build_dir(out), 'resources.json')))
    assert len(info['audit_findings']) >= 4 and info['hipcc'] != 'unknown'


def test_the_shipped_library_names_its_compiler_and_has_a_clean_audit_record():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from gru4rec_amd import _native, build
    if not os.path.exists(_native.LIB_PATH):
        pytest.skip('library not built')
    v = _native.lib().g4r_version().decode()
    assert 'hipcc' in v and 'isa-audited' in v and 'unknown' not in v
    rec = os.path.join(build.build_dir(build.OUT), 'resources.json')
    if os.path.exists(rec):      # written by the build that produced the library (not shipped to the GPU box)
        import json
        info = json.load(open(rec))
        assert info['audit_findings'] == [] and any('k_score_bwd2' in k for k in info['asm_register_loads'])


def test_every_kernel_the_host_code_launches_has_device_code():
    """A __global__ template the host code launches without an explicit instantiation links, loads and fails at its first launch;
    build._device refuses such a library (kernels_without_device_code).  The shipped library: no such kernel; the check itself: a
    kernel taken out of the device table is reported by its symbol."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from gru4rec_amd import _native, build
    rec = os.path.join(build.build_dir(build.OUT), 'resources.json')
    if not (os.path.exists(_native.LIB_PATH) and os.path.exists(rec)):
        pytest.skip('library / build record not here')
    import json
    kernels = json.load(open(rec))['kernels']
    assert build.kernels_without_device_code(_native.LIB_PATH, kernels) == []
    victim = next(k for k in kernels if 'k_gru_p2' in k)
    assert build.kernels_without_device_code(_native.LIB_PATH, {k: v for k, v in kernels.items() if k != victim}) == [victim]
