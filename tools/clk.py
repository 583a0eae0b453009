"""In-kernel phase timing (debug): G4R_BUILD_CLK=1 python -m gru4rec_amd.build --force; G4R_CLK=1 python tools/clk.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
cfg = bench.CONFIGS[os.environ.get('CFG', 'cfg2')]
plan, support = bench.make_plan(cfg, 400, 0, 1)
m = bench.create_model(cfg, support, 0, 1, 0, None, use_graph=False)
for k in ('in_idx', 'out_idx', 'reset', 'M'):
    plan[k] = plan[k][:400]
plan['T'] = 400; plan['n_compact'] = 0
m.set_plan(plan); m.reset_hidden()
m.train_steps(0, 200)
for rep in range(5):
    m.train_steps(200 + rep, 1)
    R = 2 * cfg['batch_size'] + cfg['n_sample']
    raw = m.get_debug('dbgclk', (2 * (64 + 8 * R),)).view(np.int64)
    g = raw[0:6]; s = raw[16:21]
    f = raw[0:11]
    print('k_gru_fwd_fused tile(1,1) us: requests %.2f  commits %.2f  barrier %.2f  gather issue + A1 %.2f  barrier %.2f  chunk swap + y rows + barrier %.2f  A2 %.2f  epilogue A + barrier %.2f  stage B %.2f  join+epilogue %.2f' % tuple(np.diff(f) / 100.))
    b = raw[16:25]
    print('k_gru_bwd_fused tile(1,1) us: requests %.2f  Wh/Wx->LDS %.2f  stage0 %.2f  barrier %.2f  stage1 %.2f  barrier %.2f  stage2 MFMA %.2f  join+epilogue %.2f' % tuple(np.diff(b) / 100.))
    mx = raw[32]
    tr = raw[64:].reshape(R, 8)
    tr = tr[(tr[:, 4] > 0) & (tr[:, 6] == 200 + rep)]
    t0 = tr[:, 0].min()
    d = (tr[:, 4] - tr[:, 0]) / 100.0
    pc = lambda x: np.round(np.percentile(x, [0, 50, 90, 99, 100]), 1)
    print('  sparse role: span %.1f us; wave start pct(0,50,90,99,100) %s; wave duration pct %s' % (
        (tr[:, 4].max() - t0) / 100.0, pc((tr[:, 0] - t0) / 100.0), pc(d)))
    ph = np.stack([tr[:, 1] - tr[:, 0], tr[:, 2] - tr[:, 1], tr[:, 3] - tr[:, 2], tr[:, 4] - tr[:, 3]], 1) / 100.0
    cnt = tr[:, 5] & 0xFFFFF
    print('  mean phase us: row fetch %.2f  stage list %.2f  own dups %.2f  hot loops + store %.2f ; owners %d of %d, dup owners %d, hot owners(>8) %d' % (
        *ph.mean(0), (cnt > 0).sum(), len(tr), (cnt > 1).sum(), (cnt > 9).sum()))
    for q in np.argsort(-d)[:5]:
        a = tr[q]
        hr = [(a[7] >> sh) & 0xFFFF for sh in (0, 16, 32, 48)]
        print('   wave count %3d: start +%.1f | row fetch %.1f  stage list %.1f  own dups %.1f  hot loops + store %.1f | last hot round: +%.2f scan %.2f rows %.2f wait %.2f combine %.2f -> end %.2f' % (
            cnt[q], (a[0] - t0) / 100.0, *ph[q], (a[5] >> 20) / 100.0, *[x / 100.0 for x in hr],
            (a[4] - a[3] - (a[5] >> 20) - sum(hr)) / 100.0))
