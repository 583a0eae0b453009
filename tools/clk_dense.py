"""In-kernel phase timing of the dense-gradient tiles (debug):  G4R_BUILD_CLK=1 python -m gru4rec_amd.build --force; G4R_CLK=1 G4R_NO_MERGE=1 CFG=cfg3 python tools/clk_dense.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
cfg = bench.CONFIGS[os.environ.get('CFG', 'cfg3')]
plan, support = bench.make_plan(cfg, 300, 0, 1)
m = bench.create_model(cfg, support, 0, 1, 0, None, use_graph=False)
for k in ('in_idx', 'out_idx', 'reset', 'M'):
    plan[k] = plan[k][:300]
plan['T'] = 300; plan['n_compact'] = 0
m.set_plan(plan); m.reset_hidden()
m.train_steps(0, 100)
nt = int(m.get_debug('ntiles', (1,))[0])
for rep in range(3):
    m.train_steps(100 + rep, 1)
    tr = m.get_debug('dbgtile', (2 * 8 * nt,)).view(np.int64).reshape(nt, 8)
    tr = tr[tr[:, 5] == 100 + rep]
    t0 = tr[:, 0].min()
    pc = lambda x: np.round(np.percentile(x, [0, 10, 50, 90, 100]), 2)
    ph = np.diff(tr[:, 0:5], axis=1) / 100.0
    print('tiles %d  span %.1f us | start pct(0,10,50,90,100) %s | end pct %s' % (len(tr), (tr[:, 4].max() - t0) / 100.0, pc((tr[:, 0] - t0) / 100.0), pc((tr[:, 4] - t0) / 100.0)))
    print('   phase us (median / p90): ctx+descriptor %.2f / %.2f   K loop %.2f / %.2f   LDS join %.2f / %.2f   epilogue+stores %.2f / %.2f   total %.2f / %.2f' % (
        *[v for i in range(4) for v in (np.median(ph[:, i]), np.percentile(ph[:, i], 90))], np.median(ph.sum(1)), np.percentile(ph.sum(1), 90)))
    order = np.argsort(tr[:, 0])
    q = len(tr) // 4
    for name, sel in (('first quarter', order[:q]), ('last quarter', order[-q:])):
        print('   %s: start %.1f..%.1f  total median %.2f  (ctx %.2f  K %.2f  join %.2f  epi %.2f)' % (name, (tr[sel, 0].min() - t0) / 100.0, (tr[sel, 0].max() - t0) / 100.0,
              np.median(ph[sel].sum(1)), *np.median(ph[sel], axis=0)))
