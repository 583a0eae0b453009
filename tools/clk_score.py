"""In-kernel phase timing of the k_score_fwd tiles (gemm_tile2 path; debug):  G4R_BUILD_CLK=1 python -m gru4rec_amd.build --force; G4R_CLK=1 CFG=cfg4 python tools/clk_score.py"""
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
    which = os.environ.get('KERNEL', 'fwd')
    tr = allt[4096:6144] if which == 'fwd' else allt[6144:]
    tr = tr[(tr[:, 5] == 100 + rep) & (tr[:, 4] > 0)]
    if which != 'fwd':
        for role, name in ((0, 'role A (dSy tiles)'), (1, 'role B (dh slabs)')):
            r = tr[tr[:, 6] == role]
            if len(r):
                t00 = tr[:, 0].min()
                php = np.diff(r[:, 0:5], axis=1) / 100.0
                print('   %s: %d tiles, start median %.2f max %.2f, end median %.2f max %.2f | setup %.2f first chunk %.2f K loop %.2f epilogue %.2f' % (
                    name, len(r), np.median(r[:, 0] - t00) / 100, (r[:, 0] - t00).max() / 100, np.median(r[:, 4] - t00) / 100, (r[:, 4] - t00).max() / 100,
                    *np.median(php, axis=0)))
    if which == 'fwd' and tr[:, 7].max() > 0:      # stream-K workers: placement (HW_ID: cu bits 8..11, sh 12, se 13..15 -> xcc in XCC_ID, not here)
        hw = tr[:, 6]
        cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
        key = se * 100 + sh * 20 + cu
        xcd = (hw >> 16) & 0xF        # XCC_ID
        uniq, cnt = np.unique(key * 10 + xcd, return_counts=True)
        print('   stream-K: %d workers on %d distinct (xcd, se, sh, cu); workers per CU histogram %s; units per worker %d..%d' % (
            len(tr), len(uniq), dict(zip(*np.unique(cnt, return_counts=True))), tr[:, 7].min(), tr[:, 7].max()))
    t0 = tr[:, 0].min()
    pc = lambda x: np.round(np.percentile(x, [0, 10, 50, 90, 100]), 2)
    ph = np.diff(tr[:, 0:5], axis=1) / 100.0
    print('tiles %d  span %.1f us | start pct(0,10,50,90,100) %s | end pct %s' % (len(tr), (tr[:, 4].max() - t0) / 100.0, pc((tr[:, 0] - t0) / 100.0), pc((tr[:, 4] - t0) / 100.0)))
    print('   phase us (median / p90): ctx+items %.2f / %.2f   first chunk %.2f / %.2f   K loop %.2f / %.2f   epilogue %.2f / %.2f   total %.2f / %.2f' % (
        *[v for i in range(4) for v in (np.median(ph[:, i]), np.percentile(ph[:, i], 90))], np.median(ph.sum(1)), np.percentile(ph.sum(1), 90)))
