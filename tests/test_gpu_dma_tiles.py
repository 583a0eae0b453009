"""Oracle parity of the LDS-DMA scoring tile (gemm_tile3, `k_score_fwd<64, 32, 3>`) on the shapes that exercise its edges: a
batch that is not a whole number of 64-row tiles, score rows that end inside a 64-column tile, a top layer that is a multiple of
32 but not of 64 (odd number of ring stages), a K shorter than the ring (D = 32: one stage), and steps whose live batch M shrinks
below B -- rows past M and the -1 items of finished sessions read the zero row instead of being masked.  Also the slab-count
heuristic of k_score_bwd2 (tile count = whole rounds of CUs) at a shape where it picks something else than 17 slabs.

Tolerances as in test_gpu_baseline_configs.py: per-step cost rtol 5e-4 + atol 5e-6, parameters as updates / accumulators against
their own scale (test_gpu_parity.compare_params)."""
import numpy as np
import pytest

from test_gpu_parity import close, compare_params, make_pair, random_plan, report

pytestmark = pytest.mark.gpu


def _run(tag, I, B, ns, T, store_rows, tail=True, **kw):
    o, m = make_pair(I, B, ns, store_rows=store_rows, **kw)
    plan = random_plan(I, B, T, seed=77, tail=tail)
    if tail:
        plan['M'][T // 2:] = max(1, B - 37)      # ends inside a 64-row tile, then inside the first one
        plan['M'][-2:] = 5
    plan['in_idx'][:, :6] = o.ST[0][:6]          # items repeated between input and negatives
    plan['out_idx'][:, 6:12] = plan['in_idx'][:, :6]
    m.set_plan(plan)
    want = [o.train_step(plan['in_idx'][t], plan['out_idx'][t], int(plan['M'][t]), plan['reset'][t]) for t in range(T)]
    m.train_steps(0, T)
    errs = []
    report('--- %s (score_fwd on %s)' % (tag, 'gemm_tile3'))
    close('loss curve', m.get_losses(0, T), np.array(want), atol=5e-6, rtol=5e-4, errs=errs)
    compare_params(o, m, errs, tag, Mrows=int(plan['M'][-1]))
    m.close()
    assert not errs, errs


@pytest.mark.parametrize('D', [256, 288, 512])
def test_wide_top_layer_ragged_batch_and_score_row(D):
    """B = 100 (1.56 row tiles), 1000 negatives (ldSc = 1104 = 17.25 column tiles): the wide-layer route into the DMA tile."""
    _run('dma D=%d' % D, I=9000, B=100, ns=1000, T=8, store_rows=10, loss='cross-entropy', final_act='softmax', constrained_embedding=True,
         layers=(D,), learning_rate=0.07, logq=1.0, sample_alpha=0.5)


@pytest.mark.parametrize('D', [32, 96, 160])
def test_long_score_rows_big_batch_short_k(D):
    """B = 300, 4000 negatives (the `wide_scores` route) with top layers of 1, 3 and 5 ring stages; D = 96 / 160 are not multiples
    of 64, so the backward stays on the round-1 tiles while the forward takes the DMA tile."""
    _run('dma wide D=%d' % D, I=12000, B=300, ns=4000, T=6, store_rows=8, loss='bpr-max', final_act='elu-0.5', constrained_embedding=True,
         layers=(D,), learning_rate=0.1, bpreg=0.5)


def test_slab_count_that_fills_whole_rounds():
    """B = 256, 3840 negatives, D = 128: role A has 64 x 2 = 128 tiles, a slab adds 8; the heuristic must not change results whatever
    count it picks (the slabs are summed in a fixed order by k_gru_bwd_pre / k_gru_bwd_fused)."""
    _run('slabs', I=15000, B=256, ns=3840, T=6, store_rows=8, tail=False, loss='top1-max', final_act='tanh', constrained_embedding=True,
         layers=(128,), learning_rate=0.1)


@pytest.mark.parametrize('loss,fa', [('bpr-max', 'elu-0.5'), ('cross-entropy', 'softmax'), ('top1-max', 'tanh'), ('xe_logit', 'softmax_logit')])
def test_score_rows_longer_than_two_lds_copies(loss, fa):
    """B + n_sample = 20,032 score columns: two copies of a row (yhat and dL/dyhat) no longer fit the 160 KB of LDS, so
    k_loss_rows<true> keeps the second one in the score row itself.  Same tolerances as the other shapes."""
    _run('long rows %s' % loss, I=40000, B=32, ns=20000, T=4, store_rows=6, tail=False, loss=loss, final_act=fa, constrained_embedding=True,
         layers=(32,), learning_rate=0.05, logq=1.0 if loss == 'cross-entropy' else 0.0, bpreg=0.5)


@pytest.mark.parametrize('B,ns,D', [(240, 2048, 128), (300, 4000, 48), (256, 3840, 64)])
def test_tile_or_slab_that_starts_on_an_inactive_in_batch_column(B, ns, D):
    """The tail of an epoch (M < B): the in-batch columns [M, B) are inactive (-1 items).  A 64-column tile of the scoring forward
    (`gemm_tile2`, D = 48) or a split-K slab of dh in `k_score_bwd2` (`gemm_tile2k`, D a multiple of 64 from B = 192 on) that STARTS
    inside that range used its own first row as the address masked slots load from -- a null pointer there: round 3's first
    `fit` at B = 240 faulted at address 0 (the exact-shape tests always ran full batches).  M = 5 puts every boundary below B
    into the inactive range."""
    _run('inactive head B=%d D=%d' % (B, D), I=9000, B=B, ns=ns, T=8, store_rows=10, loss='bpr-max', final_act='elu-0.5',
         constrained_embedding=True, layers=(D,), learning_rate=0.1, bpreg=0.5)
