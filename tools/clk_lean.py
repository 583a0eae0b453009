"""In-kernel phase stamps of the lean GRU kernels (debug build: python -m gru4rec_amd.build --variant tmp_var/libclk.so G4R_CLK_TRACE;
G4R_LIB=tmp_var/libclk.so G4R_CLK=1 python tools/clk_lean.py)"""
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
    def show(name, v, idx):
        v = np.asarray(v, dtype=np.int64)
        t0 = v[idx[0]]
        print('%-9s stamps (us after stamp %d): ' % (name, idx[0]) + '  '.join('[%d] %.2f' % (i, (v[i] - t0) / 100.) for i in idx[1:]))
    show('k_gru_v', raw[0:16], [1, 2, 3, 4, 5, 6, 7])
    show('k_gru_h', raw[16:32], [1, 2, 5, 6, 7])
    show('k_gru_da', raw[32:48], [1, 2, 3, 4, 5, 6])
    show('k_gru_dy', raw[48:64], [6, 7])
