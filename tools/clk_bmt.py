"""In-kernel trace of the macro-tile scoring backward (k_score_bmt):
     G4R_BUILD_CLK=1 python -m gru4rec_amd.build --variant tmp_var/libbmtclk.so;  G4R_LIB=tmp_var/libbmtclk.so G4R_CLK=1 python tools/clk_bmt.py
Per workgroup (wave 0): start, setup done, first stage landed, K loop done, stores retired, role, the CU it ran on -> phases per role and which
roles share a CU."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
cfg = bench.CONFIGS[os.environ.get('CFG', 'cfg4')]
plan, support = bench.make_plan(cfg, 300, 0, 1)
m = bench.create_model(cfg, support, 0, 1, 0, None, use_graph=False)
for k in ('in_idx', 'out_idx', 'reset', 'M'):
    plan[k] = plan[k][:300]
plan['T'] = 300; plan['n_compact'] = 0
m.set_plan(plan); m.reset_hidden()
m.train_steps(0, 100)
for rep in range(3):
    m.train_steps(100 + rep, 1)
    allt = m.get_debug('dbgtile', (2 * 8 * 8192,)).view(np.int64).reshape(8192, 8)
    tr = allt[6144:]
    tr = tr[(tr[:, 5] == 100 + rep) & (tr[:, 4] > 0)]
    t0 = tr[:, 0].min()
    ph = np.diff(tr[:, 0:5], axis=1) / 100.0
    print('workgroups %d  span %.1f us' % (len(tr), (tr[:, 4].max() - t0) / 100.0))
    for role, name in ((0, 'role A (dS tiles)'), (1, 'role B (dh slabs)')):
        s = tr[:, 6] == role
        print('   %s: %d, end median %.2f max %.2f | setup %.2f  first stage %.2f  K loop %.2f (p90 %.2f)  epilogue %.2f' % (
            name, s.sum(), np.median(tr[s, 4] - t0) / 100, (tr[s, 4] - t0).max() / 100, *np.median(ph[s, :2], axis=0), np.median(ph[s, 2]), np.percentile(ph[s, 2], 90), np.median(ph[s, 3])))
    hw = tr[:, 7] & 0xFFFFFFFF; xcc = (tr[:, 7] >> 32) & 0xF
    key = ((xcc * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 0xF)
    pairs = {}
    for k in np.unique(key):
        r = tuple(sorted(int(x) for x in tr[key == k, 6]))
        pairs[r] = pairs.get(r, 0) + 1
    print('   roles per CU:', {''.join('AB'[x] for x in k): v for k, v in sorted(pairs.items())})
    for r in sorted(pairs):
        if len(r) == 2:
            ks = [k for k in np.unique(key) if tuple(sorted(int(x) for x in tr[key == k, 6])) == r]
            ends = [(tr[key == k, 4].max() - t0) / 100 for k in ks]
            print('      CUs with %s: last end median %.2f max %.2f' % (''.join('AB'[x] for x in r), np.median(ends), np.max(ends)))
