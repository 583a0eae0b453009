"""In-kernel phase timing of k_gru_p1 tile (1,1) (split GRU path, wide layers; debug):
G4R_BUILD_CLK=1 python -m gru4rec_amd.build --force; G4R_CLK=1 CFG=cfg3 python tools/clk_p1.py"""
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
for rep in range(4):
    m.train_steps(100 + rep, 1)
    R = 2 * cfg['batch_size'] + cfg['n_sample']
    raw = m.get_debug('dbgclk', (2 * (64 + 8 * R),)).view(np.int64)
    c = raw[0:6].astype(np.float64) / 100.0
    # order in time: [4] context, [5] row indices in LDS, [0] loads issued, [1] first chunk in LDS, [2] MFMA loop done, [3] epilogue issued
    seq = [c[4], c[5], c[0], c[1], c[2], c[3]]
    print('k_gru_p1 tile(1,1) us: ctx->indices %.2f  ->loads issued %.2f  ->first chunk in LDS %.2f  ->K loop done %.2f  ->epilogue %.2f | total %.2f' % (
        *np.diff(seq), seq[-1] - seq[0]))
    ch = raw[6:14].astype(np.float64) / 100.0
    ch = ch[ch > c[1] - 1e-9]
    print('     chunk starts after first-chunk-in-LDS (us): %s ; K loop end %.2f' % (np.round(ch - c[1], 2), c[2] - c[1]))
