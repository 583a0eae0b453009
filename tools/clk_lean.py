"""In-kernel phase stamps of the lean GRU kernels (debug build: python -m gru4rec_amd.build --variant tmp_var/libclk.so G4R_CLK_TRACE;
G4R_LIB=tmp_var/libclk.so G4R_CLK=1 python tools/clk_lean.py)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
cfg = bench.CONFIGS[os.environ.get('CFG', 'cfg2')]
plan, support = bench.make_plan(cfg, 400, 0, 1)
GRAPH = bool(os.environ.get('GRAPH'))
m = bench.create_model(cfg, support, 0, 1, 0, None, use_graph=GRAPH)
for k in ('in_idx', 'out_idx', 'reset', 'M'):
    plan[k] = plan[k][:400]
plan['T'] = 400; plan['n_compact'] = 0
m.set_plan(plan); m.reset_hidden()
m.train_steps(0, 200)
for rep in range(5):
    if GRAPH:
        m.train_steps(200 + 16 * rep, 16)      # one replay of the 16-step graph: the stamps are those of its last step
    else:
        m.train_steps(200 + rep, 1)
    R = 2 * cfg['batch_size'] + cfg['n_sample']
    raw = m.get_debug('dbgclk', (2 * (64 + 8 * R),)).view(np.int64)
    def show(name, v, idx):
        v = np.asarray(v, dtype=np.int64)
        t0 = v[idx[0]]
        print('%s: stamps (us after stamp %d): ' % (name, idx[0]) + '  '.join('[%d] %.2f' % (i, (v[i] - t0) / 100.) for i in idx[1:]))
    show('k_gru_v (args pinned | y requested | m0 < M | masks | MFMAs | barrier | end)', raw[0:16], [1, 2, 3, 4, 5, 6, 7])
    show('k_gru_h (args pinned | m0 < M | MFMAs | barrier | end)', raw[16:32], [1, 2, 5, 6, 7])
    show('k_gru_dy wave 0 (start | args pinned | gathers requested | state | MFMAs)', raw[32:40], [0, 1, 2, 3, 4])
    show('k_gru_dy wave 8 (start | args pinned | H requested | MFMAs)', raw[40:48], [0, 1, 2, 3])
    show('k_gru_dy epilogue', raw[48:64], [6, 7])
    show('k_score_s (args pinned | gathers requested | m0 < M | MFMAs | barrier | end)', raw[56:64], [0, 1, 2, 3, 4, 5])
    tl_all = m.get_debug('dbgtile', (2 * 8 * 8192,)).view(np.int64).reshape(8192, 8)
    B_, N_, D_ = cfg['batch_size'], cfg['batch_size'] + cfg['n_sample'], cfg['layers'][-1]
    ld = (N_ + 15) // 16 * 16
    regions = [('k_gru_v', 1024, 3 * ((D_ + 15) // 16) * ((B_ + 15) // 16)), ('k_gru_h', 1280, ((D_ + 15) // 16) * ((B_ + 15) // 16)),
               ('k_score_s', 4096, ((ld + 31) // 32) * ((B_ + 31) // 32)), ('k_score_b', 2048, ((ld + 15) // 16) * ((D_ + 64) // 64) + ((ld + 127) // 128) * ((B_ + 15) // 16) * ((D_ + 63) // 64)),
               ('k_update_l', 2700, 6 + 82 + (2 * B_ + cfg['n_sample'] + 7) // 8),
               ('k_gru_da', 1400, ((D_ + 15) // 16) * ((B_ + 15) // 16)), ('k_gru_dy', 1500, ((D_ + 15) // 16) * ((B_ + 15) // 16))]
    t0 = None
    for name, base, n in regions:
        t = tl_all[base:base + min(n, 2040)]
        t = t[t[:, 1] > 0]
        if not len(t):
            continue
        if t0 is None:
            t0 = t[:, 0].min()
        d = (t[:, 1] - t[:, 0]) / 100.
        print('   %-10s %4d workgroups: first stamp at %+6.2f .. %+6.2f us, last stamp at %+6.2f .. %+6.2f us (after k_gru_v began); own duration median %.2f max %.2f' % (
            name, len(t), (t[:, 0].min() - t0) / 100., (t[:, 0].max() - t0) / 100., (t[:, 1].min() - t0) / 100., (t[:, 1].max() - t0) / 100., np.median(d), d.max()))
    if True:
        nt = 82
        nbk = 1 + (ld + 511) // 512
        t = tl_all[2700:2700 + nbk + nt + (2 * B_ + cfg['n_sample'] + 7) // 8]
        ok = t[:, 1] > 0
        t0u = t[ok][:, 0].min()
        for nm, sl in (('bookkeeping', slice(0, nbk)), ('dense tiles', slice(nbk, nbk + nt)), ('row workgroups (wave 0)', slice(nbk + nt, None))):
            tt = t[sl]; tt = tt[tt[:, 1] > 0]
            if len(tt):
                print('      k_update_l %-24s %4d: start %+5.2f .. %+5.2f, end %+5.2f .. %+5.2f us; own duration median %.2f max %.2f' % (
                    nm, len(tt), (tt[:, 0].min() - t0u) / 100., (tt[:, 0].max() - t0u) / 100., (tt[:, 1].min() - t0u) / 100., (tt[:, 1].max() - t0u) / 100.,
                    np.median((tt[:, 1] - tt[:, 0]) / 100.), ((tt[:, 1] - tt[:, 0]) / 100.).max()))
