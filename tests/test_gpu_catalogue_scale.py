"""Oracle parity at the REAL catalogue sizes of BASELINE.json configs[2] / configs[3]: 3,000,000 x 512 and 10,000,000 x 256 item
tables (6.1 GB / 10.2 GB each for Wy and its accumulator), i.e. with row indices far beyond 2^21 x D floats, where every gather
and scatter of the step does 64-bit address arithmetic (reference gather semantics: custom_theano_ops.py:505-519; the rows a
step may touch: gru4rec.py:335-340,428-431).

A dense NumPy oracle of that size would need ~40 GB and minutes, but a training step can only ever read or write the rows named
by the plan and by the sample store.  So the oracle holds exactly those rows: item ids are renumbered to 0..n-1 in ascending
order of the real id (a monotone map: duplicate structure, occurrence order and the popularity head are unchanged), the oracle
runs on the compact table, the device on the full one whose touched rows carry the oracle's initial values and whose other rows
carry a fixed pattern.  Checked after 10 steps: per-step costs, every touched row of Wy / By (as updates) and of their
accumulators with the bounds of test_gpu_parity.compare_params, the dense GRU parameters -- and that EVERY untouched row of
the full tables kept its bits (a scatter that wraps or truncates an index would land there; a gather that does would read the
pattern instead of the oracle's row and the costs would differ).

The ids include the corners: 0, n_items - 1, and the rows on either side of byte offsets 2^31, 2^32 and 2^33 of the table."""
import numpy as np
import pytest

from gru4rec_amd import _native
from oracle.model import OracleGRU4Rec, parse_act

from test_gpu_parity import close, close_rel, report, snapshot

pytestmark = pytest.mark.gpu


def _mem_available_gb():
    try:
        for line in open('/proc/meminfo'):
            if line.startswith('MemAvailable:'):
                return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def _run(tag, I, B, ns, T, store_rows, D, **kw):
    need = 4.0 * I * D * 4 / 1e9
    if _mem_available_gb() < need + 8:
        pytest.skip('needs ~%.0f GB of host memory for the full-table images' % need)
    rng = np.random.RandomState(7)
    # ---- real item ids: a pool spread over the whole catalogue + the address-arithmetic corners, popularity-weighted draws
    row_bytes = D * 4
    corners = [0, 1, I - 2, I - 1]
    for p in (31, 32, 33):
        r = (1 << p) // row_bytes
        corners += [r - 1, r, r + 1]
    corners = np.array(sorted(set(c for c in corners if 0 <= c < I)), dtype=np.int64)
    pool = np.unique(np.concatenate([rng.randint(0, I, size=60000).astype(np.int64), corners]))
    w = 1.0 / (np.arange(len(pool)) + 20.0)
    rng.shuffle(w)
    w /= w.sum()
    draw = lambda size: pool[rng.choice(len(pool), size=size, p=w)]
    in_idx = draw((T, B))
    out_idx = draw((T, B))
    store = draw((store_rows, ns))
    # the corners sit in every role: input, target, negative; and some items repeat across the roles within a step
    nc = len(corners)
    in_idx[0, :nc] = corners
    out_idx[1, :nc] = corners
    store[0, :nc] = corners
    in_idx[:, nc:nc + 8] = store[:T, 100:108]
    out_idx[:, nc + 8:nc + 16] = in_idx[:, nc:nc + 8]
    touched = np.unique(np.concatenate([in_idx.ravel(), out_idx.ravel(), store.ravel()]))
    n = len(touched)
    assert touched[-1] == I - 1 and touched[0] == 0
    comp = lambda a: np.searchsorted(touched, a).astype(np.int32)
    # ---- the oracle on the compact table
    o = OracleGRU4Rec(n_items=n, batch_size=B, n_sample=ns, dtype=np.float32, seed=3, layers=(D,), constrained_embedding=True, **kw)
    support_c = rng.randint(1, 40, size=n)
    o.set_popularity(support_c)
    o.By = (rng.randn(n) * 0.1).astype(np.float32)
    o.Bh[0] = (rng.randn(3 * D) * 0.1).astype(np.float32)
    o.H[0] = (rng.randn(B, D) * 0.3).astype(np.float32)
    o.init0 = snapshot(o)
    # ---- the device on the full table
    fa = parse_act(kw.get('final_act', 'linear'))
    m = _native.Model(n_items=I, layers=[D], batch_size=B, n_sample=ns, loss=_native.LOSS_IDS[o.loss], final_act=_native.ACT_IDS[fa[0]],
                      final_act_p0=fa[1], final_act_p1=fa[2], hidden_act=_native.ACT_IDS['tanh'], embed_mode=0, embedding=0,
                      learning_rate=o.learning_rate, momentum=o.momentum, lmbd=0.0, bpreg=o.bpreg, logq=o.logq, smoothing=0.0,
                      adapt=_native.ADAPT_IDS['adagrad'], sample_alpha=o.sample_alpha, dropout_p_hidden=o.dropout_p_hidden,
                      dropout_p_embed=o.dropout_p_embed, sample_store=store_rows * ns, seed=3, device=0, rank=0, nranks=1, use_graph=1)
    block = (rng.randn(4096, D) * 0.05).astype(np.float32)
    reps, rest = divmod(I, 4096)
    full = np.empty((I, D), dtype=np.float32)
    full[:reps * 4096].reshape(reps, 4096, D)[:] = block
    full[reps * 4096:] = block[:rest]
    full[touched] = o.Wy
    m.set_param('Wy', full)
    by_full = np.full(I, 0.0123, dtype=np.float32)
    by_full[touched] = o.By
    m.set_param('By', by_full)
    m.set_param('Wx', o.Wx[0]); m.set_param('Wh', o.Wh[0]); m.set_param('Wrz', o.Wrz[0]); m.set_param('Bh', o.Bh[0]); m.set_param('H', o.H[0])
    lq_t = lq_s = None
    if o.logq:
        lq_t = np.zeros(I, dtype=np.float32); lq_s = np.zeros(I, dtype=np.float32)
        lq_t[touched] = o.lq_tgt; lq_s[touched] = o.lq_smp
    cum = np.linspace(0, 1, I, dtype=np.float32)      # the store is given explicitly (frozen); the table only has to exist
    cum[-1] = 1.0
    m.set_popularity(cum, lq_t, lq_s)
    m.set_sample_store(store.astype(np.int32))
    plan = dict(in_idx=in_idx.astype(np.int32), out_idx=out_idx.astype(np.int32), reset=(rng.rand(T, B) < 0.25).astype(np.uint8),
                M=np.full(T, B, dtype=np.int32), T=T, n_compact=0, compact_steps=np.zeros(0, dtype=np.int64),
                compact_maps=np.zeros((0, B), dtype=np.int32))
    m.set_plan(plan)
    # ---- 10 steps on both sides
    cin, cout, cst = comp(in_idx), comp(out_idx), comp(store)
    want = [o.train_step(cin[t], cout[t], B, plan['reset'][t], samples=cst[t % store_rows].astype(np.int64)) for t in range(T)]
    m.train_steps(0, T)
    errs = []
    report('--- catalogue scale %s: %d items x %d, %d touched rows, largest byte offset %.2f GB' % (tag, I, D, n, (I - 1) * row_bytes / 1e9))
    close('loss curve', m.get_losses(0, T), np.array(want), atol=5e-6, rtol=5e-4, errs=errs)
    PR, PA, AR, AA = 1e-3, 1e-4, 2e-4, 1e-5

    def table(name, full0, want_rows, init_rows, acc):
        """touched rows against the oracle; then those rows are put back to their uploaded values and the whole table must equal
        the uploaded image bit for bit (untouched rows kept their bits)."""
        got = m.get_param(name, full0.shape)
        rows = got[touched].astype(np.float64)
        if acc:
            close_rel('%s %s' % (tag, name), rows, want_rows, AR, AA, errs)
        else:
            w0 = np.asarray(init_rows, dtype=np.float64)
            floor = 4.0 * float(np.spacing(np.float32(np.abs(want_rows).max())))
            close_rel('%s d%s' % (tag, name), rows - w0, np.asarray(want_rows, dtype=np.float64) - w0, PR, PA, errs, floor)
        assert (np.abs(rows - np.asarray(init_rows if not acc else 0.0, dtype=np.float64)).reshape(n, -1).max(axis=1) > 0).mean() > 0.9, \
            'the touched rows must actually have moved'
        got[touched] = full0[touched]
        same = np.array_equal(got, full0)
        report('%s %-10s untouched rows bit-identical: %s' % (tag, name, same))
        if not same:
            errs.append('%s %s: an untouched row changed' % (tag, name))
        del got

    table('Wy', full, o.Wy, o.init0['Wy'], False)
    del full
    zeros = np.zeros((I, D), dtype=np.float32)
    table('acc_Wy', zeros, o.acc['Wy'], None, True)
    del zeros
    table('By', by_full, o.By, o.init0['By'], False)
    table('acc_By', np.zeros(I, dtype=np.float32), o.acc['By'], None, True)
    for nm, shape in (('Wx', (D, 3 * D)), ('Wh', (D, D)), ('Wrz', (D, 2 * D)), ('Bh', (3 * D,))):
        w0 = o.init0[nm][0].astype(np.float64)
        wantp = getattr(o, nm)[0].astype(np.float64)
        floor = 4.0 * float(np.spacing(np.float32(np.abs(wantp).max())))
        close_rel('%s d%s' % (tag, nm), m.get_param(nm, shape, 0).astype(np.float64) - w0, wantp - w0, PR, PA, errs, floor)
        close_rel('%s acc_%s' % (tag, nm), m.get_param('acc_' + nm, shape, 0), o.acc[nm][0], AR, AA, errs)
    m.close()
    assert not errs, errs


def test_cfg4_real_catalogue_10M_items():
    """configs[3]: 10 M items x 256, batch 512, 8192 negatives, BPR-max (per-GPU shape of the 8-GPU run)."""
    _run('cfg4@10M', I=10_000_000, B=512, ns=8192, T=10, store_rows=10, D=256, loss='bpr-max', final_act='elu-0.5',
         learning_rate=0.1, bpreg=1.0)


def test_cfg3_real_catalogue_3M_items():
    """configs[2]: 3 M items x 512, batch 240, 2048 negatives, cross-entropy + logQ, embedding dropout 0.45."""
    _run('cfg3@3M', I=3_000_000, B=240, ns=2048, T=10, store_rows=10, D=512, loss='cross-entropy', final_act='softmax',
         learning_rate=0.065, logq=1.0, sample_alpha=0.5, dropout_p_embed=0.45)
