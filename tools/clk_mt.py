"""In-kernel trace of the macro-tile scoring forward (k_score_mt):
     G4R_BUILD_CLK=1 python -m gru4rec_amd.build --variant tmp_var/libmtclk.so;  G4R_LIB=tmp_var/libmtclk.so G4R_CLK=1 python tools/clk_mt.py
Per workgroup (wave 0): start, context + column items known, first stage landed, K loop done, stores retired; time spent in the DMA
waits and in the barriers of the K loop; the CU it ran on."""
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
pc = lambda x: np.round(np.percentile(x, [0, 10, 50, 90, 100]), 2)
for rep in range(3):
    m.train_steps(100 + rep, 1)
    allt = m.get_debug('dbgtile', (2 * 8 * 8192,)).view(np.int64).reshape(8192, 8)
    tr = allt[4096:6144]
    tr = tr[(tr[:, 5] == 100 + rep) & (tr[:, 4] > 0)]
    t0 = tr[:, 0].min()
    ph = np.diff(tr[:, 0:5], axis=1) / 100.0
    print('tiles %d  span %.1f us | start pct(0,10,50,90,100) %s | end pct %s' % (len(tr), (tr[:, 4].max() - t0) / 100.0, pc((tr[:, 0] - t0) / 100.0), pc((tr[:, 4] - t0) / 100.0)))
    print('   phase us (median / p90 / max): ctx+items %.2f / %.2f / %.2f   first stage %.2f / %.2f / %.2f   K loop %.2f / %.2f / %.2f   epilogue %.2f / %.2f / %.2f' % (
        *[v for i in range(4) for v in (np.median(ph[:, i]), np.percentile(ph[:, i], 90), ph[:, i].max())],))
    cyc = tr[:, 7].astype(np.float64)
    print('   K loop in s_memtime ticks: median %.0f -> %.3f ticks per ns of s_memrealtime' % (np.median(cyc), np.median(cyc / (ph[:, 2] * 1000.0))))
    hw = tr[:, 6] & 0xFFFFFFFF; xcc = (tr[:, 6] >> 32) & 0xF
    cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    uniq, cnt = np.unique(key, return_counts=True)
    print('   placement: %d distinct (xcc, se, sh, cu); workgroups per CU histogram %s' % (len(uniq), dict(zip(*np.unique(cnt, return_counts=True)))))
    for x in range(8):
        sel = xcc == x
        if sel.any():
            print('   xcc %d: %3d tiles, K loop median %.2f max %.2f, end median %.2f max %.2f' % (x, sel.sum(), np.median(ph[sel, 2]), ph[sel, 2].max(),
                  np.median(tr[sel, 4] - t0) / 100, (tr[sel, 4] - t0).max() / 100))
