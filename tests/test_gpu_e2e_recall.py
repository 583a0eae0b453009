"""End-to-end parity of the metric north_star names: Recall@20 / MRR@20 after a whole epoch on identical inputs.

An RSC15-shaped synthetic click stream (2,500 items, 24,000 sessions, the BASELINE configs[1] model: layers=[100],
batch=128, 2048 negatives, BPR-max) is trained for one epoch by the product (public class -> C ABI -> HIP kernels) and by
the NumPy oracle driven by the literal restatement of the reference's fit loop; both then rank the same test sessions.
Bar (north_star): |Recall@20 - Recall@20_oracle| <= 0.002 and the same for MRR@20 (absolute).  Loss curve: with a smooth
final activation (linear) every per-step cost of the epoch agrees to rtol 2e-3 (the oracle's own fp32-vs-fp64 gap on this run is
4e-5).  With BASELINE's elu-0.5 the derivative of the activation jumps from 1 to 0.5 at s = 0, and a score that cancels to
|s| ~ 1e-8 lands on either side depending on the summation order of the fp32 dot product (measured: the oracle has
s = -1.49e-8 at step 34 where the MFMA chain has s >= 0): two or three such elements per epoch, each a 2x difference in ONE of
278 K gradient entries, which the training dynamics amplify to 1e-2 relative per-step cost differences later on.  That is a
property of the loss surface, not of either implementation (the reference's own GPU and CPU paths differ the same way), so for
elu-0.5 the per-step bound is 5e-2 and the bars that matter are the epoch loss (rtol 1e-3) and Recall / MRR.

A second group pins `evaluate_gpu` -- the streaming path that never materialises the score matrix as well as the
materialised softmax path -- against `oracle.driver.oracle_evaluate` (evaluation.py:77-147 restated) on the SAME weights,
in all tie modes: there the only admissible difference is a rank flipping on a near-tie of two fp32 scores."""
import numpy as np
import pytest

from gru4rec_amd import evaluation, synth
from gru4rec_amd.gru4rec import GRU4Rec
from oracle.driver import oracle_evaluate, oracle_fit
from oracle.model import OracleGRU4Rec

pytestmark = pytest.mark.gpu

PARAMS = dict(loss='bpr-max', final_act='elu-0.5', layers=[100], batch_size=128, n_sample=2048, constrained_embedding=True,
              learning_rate=0.1, bpreg=1.0, momentum=0.0, sample_alpha=0.75, n_epochs=1)
STORE = 2048 * 640


def _train_both(final_act):
    data = synth.make_sessions(24000, n_items=2500, seed=17)
    train, test = synth.train_test_split(data, test_frac=0.1)
    p = dict(PARAMS, final_act=final_act)
    gru = GRU4Rec(**p)
    gru.fit(train.copy(), sample_store=STORE)
    p['layers'] = tuple(p['layers'])
    run = oracle_fit(train.copy(), p, STORE, seed=gru.seed)
    return gru, run, test


@pytest.fixture(scope='module')
def epoch():
    return _train_both('elu-0.5')


def test_epoch_loss_curve_smooth_activation():
    gru, run, _ = _train_both('linear')
    got = np.concatenate(gru.step_costs)
    assert len(got) == len(run.costs) and len(got) > 400
    np.testing.assert_allclose(got, run.costs, rtol=2e-3, atol=1e-5)
    assert abs(gru.loss_history[0] - run.epoch_loss[0]) <= 1e-4 * abs(run.epoch_loss[0])
    gru.close()


def test_epoch_loss_curve(epoch):
    gru, run, _ = epoch
    got = np.concatenate(gru.step_costs)
    assert len(got) == len(run.costs) and len(got) > 400
    np.testing.assert_allclose(got[:30], run.costs[:30], rtol=2e-5, atol=1e-6)      # before the first kink event
    np.testing.assert_allclose(got, run.costs, rtol=5e-2, atol=1e-4)                # see the module docstring
    assert abs(gru.loss_history[0] - run.epoch_loss[0]) <= 1e-3 * abs(run.epoch_loss[0])


@pytest.mark.parametrize('mode', ['standard', 'conservative', 'median'])
def test_recall_mrr_after_one_epoch(epoch, mode):
    gru, run, test = epoch
    rec, mrr = evaluation.evaluate_gpu(gru, test.copy(), cut_off=[1, 5, 20], batch_size=100, mode=mode)
    orec, omrr = oracle_evaluate(run.model, run.itemidmap, test.copy(), cut_off=[1, 5, 20], batch_size=100, mode=mode)
    print('mode %s  HIP recall %s mrr %s | oracle recall %s mrr %s' % (mode, rec, mrr, orec, omrr))
    assert orec[-1] > 0.2, 'the synthetic stream must be learnable, otherwise the comparison says nothing'
    np.testing.assert_allclose(rec, orec, rtol=0, atol=2e-3)      # north_star: +-0.2 % absolute
    np.testing.assert_allclose(mrr, omrr, rtol=0, atol=2e-3)


def _oracle_from(gru):
    """An oracle holding the product's trained weights (prediction only)."""
    o = OracleGRU4Rec(n_items=gru.n_items, layers=tuple(gru.layers), batch_size=gru.batch_size, loss=gru.loss,
                      final_act=gru.final_act, hidden_act=gru.hidden_act, n_sample=gru.n_sample,
                      constrained_embedding=gru.constrained_embedding, embedding=gru.embedding, dtype=np.float32, seed=gru.seed)
    for i in range(len(gru.layers)):
        o.Wx[i], o.Wh[i], o.Wrz[i], o.Bh[i] = gru.Wx[i], gru.Wh[i], gru.Wrz[i], gru.Bh[i]
    o.Wy, o.By = gru.Wy, gru.By.reshape(-1)
    if getattr(gru, 'E', None) is not None and not gru.constrained_embedding and gru.embedding:
        o.E = gru.E
    return o


@pytest.mark.parametrize('batch', [100, 37])
@pytest.mark.parametrize('mode', ['standard', 'conservative', 'median', 'tiebreaking'])
def test_streaming_evaluation_against_the_oracle(epoch, mode, batch):
    """Same weights on both sides: hit counts may differ only where two fp32 scores are a rounding apart."""
    gru, _, test = epoch
    o = _oracle_from(gru)
    cuts = [1, 5, 20]
    rec, mrr = evaluation.evaluate_gpu(gru, test.copy(), cut_off=cuts, batch_size=batch, mode=mode)
    orec, omrr = oracle_evaluate(o, gru.itemidmap, test.copy(), cut_off=cuts, batch_size=batch, mode=mode)
    n_events = len(test) - test.SessionId.nunique()
    np.testing.assert_allclose(rec, orec, rtol=0, atol=3.0 / n_events)      # at most 3 flipped hits out of ~5 K events
    np.testing.assert_allclose(mrr, omrr, rtol=0, atol=3.0 / n_events)


@pytest.mark.parametrize('mode', ['standard', 'conservative', 'median', 'tiebreaking'])
def test_materialised_softmax_evaluation_against_the_oracle(mode):
    """Softmax final activation keeps the score matrix (row max / sum first); fp32 softmax underflow makes real ties."""
    data = synth.make_sessions(6000, n_items=1200, seed=23)
    train, test = synth.train_test_split(data, test_frac=0.15)
    gru = GRU4Rec(loss='cross-entropy', final_act='softmax', layers=[64], batch_size=64, n_sample=512, constrained_embedding=True,
                  learning_rate=0.1, logq=1.0, sample_alpha=0.5, n_epochs=2)
    gru.fit(train.copy(), sample_store=512 * 400)
    o = _oracle_from(gru)
    cuts = [1, 5, 20]
    rec, mrr = evaluation.evaluate_gpu(gru, test.copy(), cut_off=cuts, batch_size=50, mode=mode)
    orec, omrr = oracle_evaluate(o, gru.itemidmap, test.copy(), cut_off=cuts, batch_size=50, mode=mode)
    n_events = len(test) - test.SessionId.nunique()
    np.testing.assert_allclose(rec, orec, rtol=0, atol=3.0 / n_events)
    np.testing.assert_allclose(mrr, omrr, rtol=0, atol=3.0 / n_events)
    if mode == 'tiebreaking':
        # the mode exists for saturated softmax outputs: targets tied at an underflowed score must no longer all win
        srec, _ = evaluation.evaluate_gpu(gru, test.copy(), cut_off=cuts, batch_size=50, mode='standard')
        crec, _ = evaluation.evaluate_gpu(gru, test.copy(), cut_off=cuts, batch_size=50, mode='conservative')
        assert crec[-1] - 1e-12 <= rec[-1] <= srec[-1] + 1e-12
    gru.close()
