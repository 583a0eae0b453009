"""The reference's published best parameter sets (paramfiles/*.py, values restated here) run through the public class on
an MI355X, and a few steps of each are checked against the oracle at full layer width: these are the shapes real users
run (layers 100 / 224 / 480 / 512, batches 32 / 48 / 80 / 128 / 144 / 240, 2048 negatives, momentum, both dropouts)."""
import numpy as np
import pytest

from gru4rec_amd import _native, synth
from gru4rec_amd.gru4rec import GRU4Rec
from oracle.model import OracleGRU4Rec, parse_act

pytestmark = pytest.mark.gpu

PARAMFILES = {
    'coveo_bprmax_shared_best': dict(loss='bpr-max', constrained_embedding=True, embedding=0, final_act='elu-1', layers=[512],
                                     batch_size=144, dropout_p_embed=0.35, dropout_p_hidden=0.0, learning_rate=0.05,
                                     momentum=0.4, n_sample=2048, sample_alpha=0.2, bpreg=1.85, logq=0.0),
    'diginetica_bprmax_shared_best': dict(loss='bpr-max', constrained_embedding=True, embedding=0, final_act='elu-1', layers=[512],
                                          batch_size=128, dropout_p_embed=0.5, dropout_p_hidden=0.3, learning_rate=0.05,
                                          momentum=0.15, n_sample=2048, sample_alpha=0.3, bpreg=0.9, logq=0.0),
    'rees46_xe_shared_best': dict(loss='cross-entropy', constrained_embedding=True, embedding=0, final_act='softmax', layers=[512],
                                  batch_size=240, dropout_p_embed=0.45, dropout_p_hidden=0.0, learning_rate=0.065,
                                  momentum=0.0, n_sample=2048, sample_alpha=0.5, bpreg=0.0, logq=1.0),
    'retailrocket_bprmax_shared_best': dict(loss='bpr-max', constrained_embedding=True, embedding=0, final_act='elu-0.5',
                                            layers=[224], batch_size=80, dropout_p_embed=0.5, dropout_p_hidden=0.05,
                                            learning_rate=0.05, momentum=0.4, n_sample=2048, sample_alpha=0.4, bpreg=1.95,
                                            logq=0.0),
    'rsc15_xe_shared_100_best': dict(loss='cross-entropy', constrained_embedding=True, embedding=0, final_act='softmax',
                                     layers=[100], batch_size=32, dropout_p_embed=0.0, dropout_p_hidden=0.4,
                                     learning_rate=0.2, momentum=0.2, n_sample=2048, sample_alpha=0.5, bpreg=0.0, logq=1.0),
    'yoochoose_xe_shared_best': dict(loss='cross-entropy', constrained_embedding=True, embedding=0, final_act='softmax',
                                     layers=[480], batch_size=48, dropout_p_embed=0.0, dropout_p_hidden=0.2,
                                     learning_rate=0.07, momentum=0.0, n_sample=2048, sample_alpha=0.2, bpreg=0.0, logq=1.0),
}


@pytest.mark.parametrize('name', sorted(PARAMFILES))
def test_paramfile_trains_through_the_public_class(name):
    p = dict(PARAMFILES[name])
    data = synth.make_sessions(3000, n_items=1500, seed=3)
    gru = GRU4Rec()
    gru.set_params(n_epochs=2, **p)
    gru.fit(data, sample_store=2048 * 64)
    assert not gru.error_during_train
    h = gru.loss_history
    assert len(h) == 2 and np.isfinite(h).all(), h
    # rsc15_xe (learning rate 0.2, momentum 0.2, batch 32) is unstable on this small synthetic set: the fp32 and the fp64
    # oracle end its second epoch at 11.35 and 14.16; "the loss went down" says something for the other five only
    if name != 'rsc15_xe_shared_100_best':
        assert h[1] < h[0], h


@pytest.mark.parametrize('name', sorted(PARAMFILES))
def test_paramfile_steps_match_the_oracle(name):
    """12 steps at the paramfile's full width / batch / 2048 negatives against the NumPy oracle: costs against the fp32 oracle; the
    parameters against the fp64 oracle, within a bound bracketed by the fp32 oracle's own distance from it.  (Round 6: on
    rsc15_xe_shared_100_best one element of Wy -- a first-touch Adagrad step with |g| ~ sqrt(eps) -- is 6.3e-4 off in the FP32 ORACLE, 2.4e-4
    in the round-2 kernels and 1.2e-4 in the round-6 kernels: a fixed tolerance against the fp32 run failed the more accurate kernels.)"""
    p = dict(PARAMFILES[name])
    I, T, rows = 3000, 12, 8
    B, ns, D = p['batch_size'], p['n_sample'], p['layers'][0]
    o = OracleGRU4Rec(n_items=I, layers=tuple(p['layers']), batch_size=B, loss=p['loss'], final_act=p['final_act'], n_sample=ns,
                      sample_alpha=p['sample_alpha'], learning_rate=p['learning_rate'], momentum=p['momentum'], bpreg=p['bpreg'],
                      logq=p['logq'], dropout_p_hidden=p['dropout_p_hidden'], dropout_p_embed=p['dropout_p_embed'],
                      constrained_embedding=True, dtype=np.float32, seed=11)
    o64 = OracleGRU4Rec(n_items=I, layers=tuple(p['layers']), batch_size=B, loss=p['loss'], final_act=p['final_act'], n_sample=ns,
                        sample_alpha=p['sample_alpha'], learning_rate=p['learning_rate'], momentum=p['momentum'], bpreg=p['bpreg'],
                        logq=p['logq'], dropout_p_hidden=p['dropout_p_hidden'], dropout_p_embed=p['dropout_p_embed'],
                        constrained_embedding=True, dtype=np.float64, seed=11)
    rng = np.random.RandomState(5)
    pop = rng.randint(1, 60, size=I)
    for q in (o, o64):
        q.set_popularity(pop)
        q.make_sample_store(rows * ns)
    o64.Wx[0], o64.Wh[0], o64.Wrz[0] = (x.astype(np.float64) for x in (o.Wx[0], o.Wh[0], o.Wrz[0]))      # the same fp32 initial weights
    o64.Wy, o64.By, o64.Bh[0] = o.Wy.astype(np.float64), o.By.astype(np.float64), o.Bh[0].astype(np.float64)
    fa = parse_act(p['final_act'])
    m = _native.Model(n_items=I, layers=[D], batch_size=B, n_sample=ns, loss=_native.LOSS_IDS[p['loss']],
                      final_act=_native.ACT_IDS[fa[0]], final_act_p0=fa[1], final_act_p1=fa[2], hidden_act=_native.ACT_IDS['tanh'],
                      embed_mode=_native.EMBED_CONSTRAINED, embedding=0, learning_rate=p['learning_rate'], momentum=p['momentum'],
                      lmbd=0.0, bpreg=p['bpreg'], logq=p['logq'], sample_alpha=p['sample_alpha'],
                      dropout_p_hidden=p['dropout_p_hidden'], dropout_p_embed=p['dropout_p_embed'], sample_store=rows * ns,
                      seed=11, device=0, rank=0, nranks=1, use_graph=1)
    m.set_param('Wx', o.Wx[0], 0); m.set_param('Wh', o.Wh[0], 0); m.set_param('Wrz', o.Wrz[0], 0); m.set_param('Bh', o.Bh[0], 0)
    m.set_param('Wy', o.Wy); m.set_param('By', o.By)
    m.set_popularity(o.P, o.lq_tgt if o.logq else None, o.lq_smp if o.logq else None)
    plan = dict(in_idx=rng.randint(0, I, size=(T, B)).astype(np.int32), out_idx=rng.randint(0, I, size=(T, B)).astype(np.int32),
                reset=(rng.rand(T, B) < 0.2).astype(np.uint8), M=np.full(T, B, dtype=np.int32), T=T, n_compact=0)
    m.set_plan(plan)
    want = [o.train_step(plan['in_idx'][t], plan['out_idx'][t], B, plan['reset'][t].astype(bool)) for t in range(T)]
    for t in range(T):
        o64.train_step(plan['in_idx'][t], plan['out_idx'][t], B, plan['reset'][t].astype(bool))
    m.train_steps(0, T)
    got = m.get_losses(0, T)
    np.testing.assert_allclose(got, want, rtol=1e-3, atol=1e-5)
    for name_, gpu, f32, f64 in (('Wy', m.get_param('Wy', (I, D)), o.Wy, o64.Wy), ('Wh', m.get_param('Wh', (D, D), 0), o.Wh[0], o64.Wh[0])):
        gap = float(np.abs(f32.astype(np.float64) - f64).max())      # what fp32 arithmetic costs the oracle itself on this run
        assert gap < 2e-3, (name_, gap)                               # (the run is not chaotic: the bracket below means something)
        np.testing.assert_allclose(gpu.astype(np.float64), f64, rtol=5e-3, atol=max(2e-4, 2.0 * gap), err_msg='%s (fp32 oracle vs fp64: %.2e)' % (name_, gap))
    m.close()
