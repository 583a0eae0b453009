"""Oracle parity of the macro-tile scoring kernels.  Forward (k_score_mt, g4r_score_mt.cuh): the score matrix of B = 512, 8192 negatives cut
into 256 tiles of 64 rows x 272 columns, one per compute unit.  Backward (k_score_bmt, g4r_score_bmt.cuh; D = 256): 256 tiles of 272 x 32
of dS and 256 of 64 x 128 x 16 slabs of dh, the raw gradient rows finished by extra workgroups of k_gru_bwd_a, the bias gradient folded
into the dS tiles -- same edges: batch rows past M read the zero row (K of dS), inactive columns, items that repeat (accumulator in place
only for single occurrences).  Edges: steps whose live batch M ends inside a 64-row tile, inside a
wave's 32-row block and inside the 16-row strip blocks (rows past M read the zero row and are not stored), the -1 items of the
in-batch columns [M, B), items repeated between input and negatives, K = 64 (fewer stages than the ring holds: the counted waits of
the prologue and of the tail) and K = 256 / 512 (steady iterations); cross-entropy with the logQ correction (the epilogue's second
gather) and BPR-max.  The 64 x 64 tiles it replaced stay reachable (G4R_NO_MT=1) and must keep passing the exact-shape test.

Tolerances as in test_gpu_dma_tiles.py."""
import os
import subprocess
import sys

import numpy as np
import pytest

from test_gpu_parity import close, compare_params, make_pair, random_plan, report

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tag, I, T, store_rows, D, **kw):
    B, ns = 512, 8192
    o, m = make_pair(I, B, ns, store_rows=store_rows, layers=(D,), constrained_embedding=True, **kw)
    assert m.get_debug('score_mt', 1)[0] == 272, 'the macro-tile kernel was not selected at B = 512, N = 8704'
    assert (m.get_debug('score_bmt', 1)[0] == 16) == (D == 256), 'macro-tile backward: 16 slabs at D = 256, k_score_bwd2 elsewhere'
    plan = random_plan(I, B, T, seed=31, tail=True)
    plan['M'][:] = B
    plan['M'][1] = B - 37          # ends inside a 64-row tile and a 32-row block
    plan['M'][2] = 64 + 16 + 5     # inside the second strip block of row tile 1
    plan['M'][-1] = 5
    plan['in_idx'][:, :6] = o.ST[0][:6]          # items repeated between input and negatives
    plan['out_idx'][:, 6:12] = plan['in_idx'][:, :6]
    m.set_plan(plan)
    want = [o.train_step(plan['in_idx'][t], plan['out_idx'][t], int(plan['M'][t]), plan['reset'][t]) for t in range(T)]
    m.train_steps(0, T)
    errs = []
    report('--- %s (score_fwd on k_score_mt)' % tag)
    close('loss curve', m.get_losses(0, T), np.array(want), atol=5e-6, rtol=5e-4, errs=errs)
    compare_params(o, m, errs, tag, Mrows=int(plan['M'][-1]))
    m.close()
    assert not errs, errs


@pytest.mark.parametrize('D', [64, 256, 512])
def test_macro_tiles_bprmax_ragged_batches(D):
    _run('mt bpr-max D=%d' % D, I=30000, T=4, store_rows=6, D=D, loss='bpr-max', final_act='elu-0.5', learning_rate=0.1, bpreg=1.0)


@pytest.mark.parametrize('kw', [dict(adapt='rmsprop', adapt_params=[0.9], learning_rate=0.02), dict(momentum=0.1, learning_rate=0.1), dict(lmbd=1e-4, learning_rate=0.1)],
                         ids=['rmsprop', 'adagrad_momentum', 'adagrad_l2'])
def test_macro_tile_backward_with_the_other_update_rules(kw):
    """k_score_bmt leaves RAW gradient rows behind and `score_fin_rows` (riding on k_gru_bwd_a) finishes them: the generic optimizers take the raw
    gradient as the step, Adagrad with momentum / an L2 term takes the Adagrad step and leaves the rest to the update launch."""
    _run('mt ' + '/'.join(sorted(kw)), I=30000, T=3, store_rows=5, D=256, loss='bpr-max', final_act='elu-0.5', bpreg=1.0, **kw)


def test_macro_tiles_cross_entropy_with_logq():
    _run('mt xe logq', I=30000, T=4, store_rows=6, D=128, loss='cross-entropy', final_act='softmax', learning_rate=0.07, logq=1.0, sample_alpha=0.5)


def test_the_replaced_tiles_still_pass_the_exact_shape_test():
    env = dict(os.environ, G4R_NO_MT='1', G4R_NO_BMT='1')
    r = subprocess.run([sys.executable, '-m', 'pytest', 'tests/test_gpu_baseline_configs.py::test_cfg4_exact_shape', '-x', '-q', '-p', 'no:cacheprovider'],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
