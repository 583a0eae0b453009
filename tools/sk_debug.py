"""stream-K scoring forward: a few eager steps at the cfg4 shape (small catalogue), timed per call, against the tile launch.
    timeout 120 python tools/sk_debug.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
cfgname = os.environ.get('CFG', 'cfg4s')
cfg = bench.CONFIGS[cfgname]
T = 6
plan, support = bench.make_plan(cfg, T + 8, 0, 1)
for k in ('in_idx', 'out_idx', 'reset', 'M'):
    plan[k] = plan[k][:T]
plan['T'] = T; plan['n_compact'] = 0
res = {}
for sk in ('0', '1'):
    os.environ['G4R_STREAMK'] = sk
    m = bench.create_model(cfg, support, 0, 1, 0, None, use_graph=False, sample_store=cfg['n_sample'] * 16)
    print('streamk env', sk, 'workers', m.get_debug('streamk_workers', (1,))[0], flush=True)
    m.set_plan(plan); m.reset_hidden()
    ts = []
    for t in range(T):
        t0 = time.time(); m.train_steps(t, 1); ts.append(time.time() - t0)
    L = m.get_losses(0, T)
    print('  ms per call', np.round(1000 * np.array(ts), 2), 'losses', L, flush=True)
    res[sk] = (L, m.get_param('Wy', (cfg['n_items'], cfg['layers'][-1]))[:2000].copy())
    m.close()
print('max |loss diff|', np.abs(res['0'][0] - res['1'][0]).max(), ' max |Wy diff|', np.abs(res['0'][1] - res['1'][1]).max())
