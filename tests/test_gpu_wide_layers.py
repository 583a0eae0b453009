"""The K-sliced kernels of wide GRU layers (gru4rec_amd/csrc/g4r_wide_kernels.cuh: k_gru_p1s + k_gru_gate, k_gru_bwd_bw, k_dense_grad2 with
its row-finishing workgroups; the K slices of a tile are left as partial sums that the consuming kernel adds up in slice order)
against the oracle, against the round-1 kernels they replace (G4R_WIDE2=0; the switch is read per model), one kernel at a time, under
the default policy, and against themselves (graph replay == eager launches, bit for bit).

Tolerances: test_gpu_parity.py (fp32 both sides, different summation orders).  The old and the new kernels differ ONLY in summation
order (k-ordered 16x16x4 chains over the whole K against 32x32x2 chains over slices), so their losses are compared at rtol 2e-5."""
import os

import numpy as np
import pytest

from test_gpu_parity import close, compare_params, make_pair, random_plan, report

pytestmark = pytest.mark.gpu

ALL = 1 | 8 | 16
SHAPES = {
    # name: (I, B, ns, T, kwargs)      B = 96: a full and a half row tile; B = 240: BASELINE configs[2]'s 3.75 row tiles
    'd256_b96_bprmax': (3000, 96, 512, 8, dict(loss='bpr-max', final_act='elu-0.5', constrained_embedding=True, layers=(256,),
                                               learning_rate=0.1, bpreg=1.0, momentum=0.1)),
    'd512_b240_xe_logq_drop': (5000, 240, 1024, 6, dict(loss='cross-entropy', final_act='softmax', constrained_embedding=True, layers=(512,),
                                                       learning_rate=0.065, logq=1.0, sample_alpha=0.5, dropout_p_embed=0.45, dropout_p_hidden=0.2)),
    'two_layers_256_320': (2000, 130, 256, 6, dict(loss='top1-max', final_act='elu-0.5', constrained_embedding=True, layers=(256, 320),
                                                   learning_rate=0.1, dropout_p_embed=0.2, dropout_p_hidden=0.1)),
    'separate_embedding_128_to_384': (2000, 64, 256, 6, dict(loss='bpr-max', final_act='elu-0.5', constrained_embedding=False, embedding=128,
                                                             layers=(384,), learning_rate=0.1, bpreg=0.5)),
    'rmsprop_generic_path': (2000, 96, 256, 6, dict(loss='bpr-max', final_act='elu-0.5', constrained_embedding=True, layers=(256,),
                                                    learning_rate=0.05, bpreg=1.0, adapt='rmsprop', adapt_params=[0.9], dropout_p_embed=0.1)),
}


def _env(**kv):
    class _E:
        def __enter__(self):
            self.old = {k: os.environ.get(k) for k in kv}
            for k, v in kv.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = str(v)

        def __exit__(self, *a):
            for k, v in self.old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    return _E()


def _plan(I, B, T, o, tail):
    plan = random_plan(I, B, T, seed=77, tail=tail)
    if o.ST is not None and len(o.ST):      # repeated items inside the batch and against the negatives
        plan['in_idx'][:, :6] = o.ST[0][:6]
        plan['out_idx'][:, 6:12] = plan['in_idx'][:, :6]
    return plan


@pytest.mark.parametrize('name', sorted(SHAPES))
@pytest.mark.parametrize('tail', [False, True])
def test_wide_kernels_against_the_oracle(name, tail):
    I, B, ns, T, kw = SHAPES[name]
    with _env(G4R_WIDE2=ALL):
        o, m = make_pair(I, B, ns, store_rows=T + 2, **kw)
    assert int(m.get_debug('wide_mask', 1)[0]) == ALL, 'the wide-layer kernels did not engage at this shape'
    plan = _plan(I, B, T, o, tail)
    m.set_plan(plan)
    want = [o.train_step(plan['in_idx'][t], plan['out_idx'][t], int(plan['M'][t]), plan['reset'][t]) for t in range(T)]
    m.train_steps(0, T)
    errs = []
    report('--- wide %s tail=%s' % (name, tail))
    close('loss curve', m.get_losses(0, T), np.array(want), atol=5e-6, rtol=5e-4, errs=errs)
    compare_params(o, m, errs, 'wide-' + name, Mrows=int(plan['M'].min()))
    m.close()
    assert not errs, errs


@pytest.mark.parametrize('mask', [1, 8, 16, 24, ALL, -1])
def test_each_wide_kernel_against_the_kernel_it_replaces(mask):
    """One new kernel at a time, 8 + 16, all of them, and -1: the default policy, next to the round-1 kernels on the same plan: the
    losses agree to summation order."""
    I, B, ns, T, kw = SHAPES['d512_b240_xe_logq_drop']
    runs = {}
    for mk in (0, mask):
        with _env(G4R_WIDE2=None if mk < 0 else mk):
            o, m = make_pair(I, B, ns, store_rows=T + 2, **kw)
            if mk != 0:
                assert int(m.get_debug('wide_mask', 1)[0]) == (ALL if mk < 0 else mk)      # (the default policy takes all three at this shape)
        plan = _plan(I, B, T, o, tail=True)
        m.set_plan(plan)
        m.train_steps(0, T)
        runs[mk] = (m.get_losses(0, T).copy(), m.get_param('Wx', (512, 1536), 0).copy(), m.get_param('Wh', (512, 512), 0).copy(),
                    m.get_param('Wy', (I, 512)).copy(), m.get_param('acc_Wy', (I, 512)).copy())
        m.close()
    assert np.isfinite(runs[mask][0]).all()
    np.testing.assert_allclose(runs[mask][0], runs[0][0], rtol=2e-5, atol=1e-6)
    for a, b, nm in zip(runs[mask][1:], runs[0][1:], ('Wx', 'Wh', 'Wy', 'acc_Wy')):
        scale = np.abs(b).max()
        assert np.abs(a - b).max() <= 2e-4 * scale, (nm, float(np.abs(a - b).max()), float(scale))


def test_deep_chunk_policy_by_shape():
    """8 waves x 256-deep chunks for k_gru_p2 / k_gru_bwd_a from 384 units on while the tiles leave CUs idle (BASELINE configs[2]: 512 units,
    B = 240); BASELINE configs[3] (256 units) and a batch of 1024 rows at 512 units (512 tiles) stay on 4 waves x 128."""
    for (B, D), want in (((240, 512), 3), ((512, 256), 0), ((1024, 512), 0)):
        o, m = make_pair(3000, B, 1024, store_rows=2, loss='bpr-max', final_act='elu-0.5', constrained_embedding=True, layers=(D,), learning_rate=0.1)
        assert int(m.get_debug('deep_geo', 1)[0]) == want, (B, D)
        m.close()


def test_default_policy_by_shape():
    """BASELINE configs[3]'s shape (D = 256, 2 B + n_sample = 9216 rows): the merged k_update stays (no k_dense_grad2, no k_gru_p1s below
    D = 512); dy is sliced at every wide layer -- layer 0's slices are added up by k_finish_rows in front of k_update."""
    o, m = make_pair(3000, 512, 8192, store_rows=2, loss='bpr-max', final_act='elu-0.5', constrained_embedding=True, layers=(256,), learning_rate=0.1)
    assert int(m.get_debug('wide_mask', 1)[0]) == 8
    m.close()
    o, m = make_pair(3000, 512, 8192, store_rows=2, loss='bpr-max', final_act='elu-0.5', constrained_embedding=True, layers=(256, 256), learning_rate=0.1)
    assert int(m.get_debug('wide_mask', 1)[0]) == 8
    m.close()


@pytest.mark.parametrize('ks', [(64, 128), (96, 256), (128, 768)])
def test_slice_lengths(ks):
    """Other K-slice geometries than the default: same results."""
    I, B, ns, T, kw = SHAPES['d512_b240_xe_logq_drop']
    with _env(G4R_WIDE2=ALL, G4R_P1_KS=ks[0], G4R_BB_KS=ks[1]):
        o, m = make_pair(I, B, ns, store_rows=T + 2, **kw)
    plan = _plan(I, B, T, o, tail=False)
    m.set_plan(plan)
    want = [o.train_step(plan['in_idx'][t], plan['out_idx'][t], B, plan['reset'][t]) for t in range(T)]
    m.train_steps(0, T)
    errs = []
    report('--- wide slices of %d / %d' % ks)
    close('loss curve', m.get_losses(0, T), np.array(want), atol=5e-6, rtol=5e-4, errs=errs)
    compare_params(o, m, errs, 'wide-ks%d' % ks[0])
    m.close()
    assert not errs, errs


def test_wide_graph_replay_is_bit_identical_to_eager():
    """32 steps as two graph replays against 32 eager steps (one launch after the other): identical bits in every loss and parameter
    (slices are added in slice order, by whichever workgroup gets there), and a second identical run reproduces the first."""
    I, B, ns, T, kw = 4000, 240, 1024, 32, SHAPES['d512_b240_xe_logq_drop'][4]
    out = []
    for use_graph in (0, 1, 1):
        with _env(G4R_WIDE2=ALL):
            o, m = make_pair(I, B, ns, store_rows=T + 2, use_graph=use_graph, **kw)
        plan = _plan(I, B, T, o, tail=False)
        m.set_plan(plan)
        m.train_steps(0, T)
        out.append((m.get_losses(0, T).copy(), m.get_param('Wy', (I, 512)).copy(), m.get_param('Wx', (512, 1536), 0).copy(),
                    m.get_param('Bh', (1536,), 0).copy()))
        m.close()
    for other in out[1:]:
        for a, b in zip(out[0], other):
            np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize('geo', [0, 1])
@pytest.mark.parametrize('name', ['d512_b240_xe_logq_drop', 'two_layers_256_320', 'd256_b96_bprmax'])
def test_phase2_geometries(name, geo):
    """k_gru_p2 / k_gru_bwd_a<threads, chunk depth> (G4R_P2_GEO / G4R_BA_GEO: 0 = 4 waves x 128, 1 = 8 waves x 256) at K = 512, 256, and
    320 (whole chunks and partial ones) against the oracle: losses and every parameter."""
    I, B, ns, T, kw = SHAPES[name]
    with _env(G4R_P2_GEO=geo, G4R_BA_GEO=geo):
        o, m = make_pair(I, B, ns, store_rows=T + 2, **kw)
    assert int(m.get_debug('deep_geo', 1)[0]) == 3 * geo
    plan = _plan(I, B, T, o, tail=True)
    m.set_plan(plan)
    want = [o.train_step(plan['in_idx'][t], plan['out_idx'][t], int(plan['M'][t]), plan['reset'][t]) for t in range(T)]
    m.train_steps(0, T)
    errs = []
    report('--- phase-2 geometry %d at %s' % (geo, name))
    close('loss curve', m.get_losses(0, T), np.array(want), atol=5e-6, rtol=5e-4, errs=errs)
    compare_params(o, m, errs, 'p2geo%d-%s' % (geo, name), Mrows=int(plan['M'].min()))
    m.close()
    assert not errs, errs
