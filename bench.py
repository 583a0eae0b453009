#!/usr/bin/env python
"""Benchmark of the GRU4Rec session-parallel training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one mini-batch of the reference's `train_function` (gru4rec.py:623) at BASELINE.json config #2:
RSC15-shaped synthetic sessions (I = 37,483 items), layers=[100], batch=128, n_sample=2048, BPR-max,
constrained embedding, Adagrad.  The timed region is K plan steps with the plan, weights and sample store
already resident in HBM; it includes sample-store refills (as the reference's epoch timing does) and
excludes plan building / upload.  metric = mini-batches/s exactly as gru4rec.py:661 prints it (steps / seconds);
events/s is reported next to it.  For N > 1 (one process per GPU: `bench.py --gpus N` spawns its ranks itself, or
`python -m torch.distributed.run` provides RANK / LOCAL_RANK / WORLD_SIZE) sessions are sharded over ranks, dense GRU gradients
are all-reduced by RCCL every step, value = sum over ranks INCLUDING the reconciliation of the GPU-local item tables every
sync_every steps (the bare step is reported next to it as value_step_only).  No torch anywhere: the ranks meet through gru4rec_amd/launch.py
(file rendezvous of the RCCL unique id) and synchronise through the communicator (g4r_comm_max_i64).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

A30_PUBLISHED_MBS = 1240.0   # BPR-max, layers=[100], batch 128, n_sample 2048 on an A30 (Theano)

CONFIGS = {
    # BASELINE.json configs[0] (the reference's CPU-runnable plumbing case; in-batch negatives only)
    'cfg1': dict(n_items=37483, layers=[100], batch_size=32, n_sample=0, loss='cross-entropy', final_act='softmax',
                 bpreg=0.0, learning_rate=0.1, momentum=0.0, sample_alpha=0.75, logq=0.0, dropout_p_embed=0.0,
                 dropout_p_hidden=0.0, constrained_embedding=True),
    # BASELINE.json configs[1]
    'cfg2': dict(n_items=37483, layers=[100], batch_size=128, n_sample=2048, loss='bpr-max', final_act='elu-0.5',
                 bpreg=1.0, learning_rate=0.1, momentum=0.0, sample_alpha=0.75, logq=0.0, dropout_p_embed=0.0,
                 dropout_p_hidden=0.0, constrained_embedding=True),
    # BASELINE.json configs[2] (Rees46 shape, HBM-bound gather)
    'cfg3': dict(n_items=3000000, layers=[512], batch_size=240, n_sample=2048, loss='cross-entropy', final_act='softmax',
                 bpreg=0.0, learning_rate=0.065, momentum=0.0, sample_alpha=0.5, logq=1.0, dropout_p_embed=0.45,
                 dropout_p_hidden=0.0, constrained_embedding=True),
    # BASELINE.json configs[3] (per-GPU shape of the 8-GPU synthetic 10M-item catalogue)
    'cfg4': dict(n_items=10000000, layers=[256], batch_size=512, n_sample=8192, loss='bpr-max', final_act='elu-0.5',
                 bpreg=1.0, learning_rate=0.1, momentum=0.0, sample_alpha=0.75, logq=0.0, dropout_p_embed=0.0,
                 dropout_p_hidden=0.0, constrained_embedding=True),
    # diagnostic: the cfg4 step over a cache-resident catalogue (separates GEMM time from the cost of touching a 10 GB table)
    'cfg4s': dict(n_items=100000, layers=[256], batch_size=512, n_sample=8192, loss='bpr-max', final_act='elu-0.5',
                  bpreg=1.0, learning_rate=0.1, momentum=0.0, sample_alpha=0.75, logq=0.0, dropout_p_embed=0.0,
                  dropout_p_hidden=0.0, constrained_embedding=True),
    # configs[3]'s shape over a catalogue that is built in seconds (2 GB table, 16 x the Infinity Cache): the large-catalogue legs of the default line
    'cfg4m': dict(n_items=2000000, layers=[256], batch_size=512, n_sample=8192, loss='bpr-max', final_act='elu-0.5',
                  bpreg=1.0, learning_rate=0.1, momentum=0.0, sample_alpha=0.75, logq=0.0, dropout_p_embed=0.0,
                  dropout_p_hidden=0.0, constrained_embedding=True),
    # BASELINE.json configs[4]
    'cfg5': dict(n_items=37483, layers=[100, 100], batch_size=128, n_sample=2048, loss='top1-max', final_act='elu-0.5',
                 bpreg=1.0, learning_rate=0.1, momentum=0.0, sample_alpha=0.75, logq=0.0, dropout_p_embed=0.2,
                 dropout_p_hidden=0.0, constrained_embedding=True),
}


def algorithmic_cost(cfg):
    """Per-launch algorithmic bytes / flops of the step kernels."""
    B, ns, D = cfg['batch_size'], cfg['n_sample'], cfg['layers'][-1]
    N, R = B + ns, 2 * B + ns
    mom = 2 if cfg['momentum'] > 0 else 0
    return {
        # gradient row read + param r/w + accumulator r/w (+ velocity r/w) per occurrence, By path, index list
        'k_sparse_update': dict(bound='hbm', bytes=(5 + mom) * R * D * 4 + 5 * N * 4 + R * 4),
        # gathered Wy rows + h + score write ; 2*B*D*N flops
        'k_score_fwd': dict(bound='mfma', flops=2.0 * B * D * N, bytes=N * D * 4 + B * D * 4 + B * N * 4),
        # dSy = ds^T h and dh = ds Sy
        'k_score_bwd': dict(bound='mfma', flops=4.0 * B * D * N, bytes=2 * B * N * 4 + 2 * N * D * 4 + B * D * 4),
        'k_loss_rows': dict(bound='hbm', bytes=2 * B * N * 4),
        'k_gru_p1': dict(bound='mfma', flops=2.0 * B * 5 * D * D, bytes=B * D * 4 * 6 + 5 * D * D * 4),
        'k_gru_p2': dict(bound='mfma', flops=2.0 * B * D * D, bytes=B * D * 4 * 6 + D * D * 4),
        # k_gru_p1 + k_gru_p2 in one launch (r for all D columns is recomputed by the column tiles of a row block: not counted)
        'k_gru_fwd': dict(bound='mfma', flops=2.0 * B * 6 * D * D, bytes=B * D * 4 * 10 + 6 * D * D * 4),
        'k_gru_bwd_pre': dict(bound='hbm', bytes=B * D * 4 * 6),
        'k_gru_bwd_a': dict(bound='mfma', flops=2.0 * B * D * D, bytes=B * D * 4 * 4 + D * D * 4),
        'k_gru_bwd_b': dict(bound='mfma', flops=2.0 * B * 3 * D * D, bytes=B * D * 4 * 4 + 3 * D * D * 4),
        # round 6, narrow layers (g4r_lean_kernels.cuh): the forward as k_gru_v (V = [y | H] [Wx ; 0 | Wrz], r, Hr, z) + k_gru_h (candidate, h), the
        # backward as k_gru_da (da, dz', K-slice planes of dr') + k_gru_dy (dy = dV Wx^T, Adagrad pieces of the input rows)
        'k_gru_v': dict(bound='mfma', flops=2.0 * B * 5 * D * D, bytes=B * D * 4 * 6 + 5 * D * D * 4),
        'k_gru_h': dict(bound='mfma', flops=2.0 * B * D * D, bytes=B * D * 4 * 6 + D * D * 4),
        'k_gru_da': dict(bound='mfma', flops=2.0 * B * D * D, bytes=B * D * 4 * 6 + D * D * 4),
        'k_gru_dy': dict(bound='mfma', flops=2.0 * B * 3 * D * D, bytes=B * D * 4 * 6 + 3 * D * D * 4),
        # the three stages above in one launch (stage 0 / 1 are repeated by the column tiles of a row block: not counted)
        'k_gru_bwd': dict(bound='mfma', flops=2.0 * B * 4 * D * D, bytes=B * D * 4 * 10 + 4 * D * D * 4),
        'k_dense_grad': dict(bound='mfma', flops=2.0 * B * 6 * D * D, bytes=B * D * 4 * 6 + 3 * 6 * D * D * 4),
        # single GPU: dense-gradient tiles + sparse row update share one launch; its HBM-side work is the sparse
        # gather/scatter plus the dense parameters / accumulators read and written once
        'k_update': dict(bound='hbm', bytes=(5 + mom) * R * D * 4 + 5 * N * 4 + R * 4 + 4 * 6 * D * D * 4 + B * D * 4 * 6),
    }


PEAK = {'hbm': (8000.0, 'GB/s'), 'mfma': (157.3, 'TFLOP/s')}   # MI355X_MICROARCH.md: HBM3E 8 TB/s, fp32 MFMA 157.3 TF


def mom_planes(cfg):
    return 2 if cfg['momentum'] > 0 else 0


def newest_profile(suffix):
    """profiles/rNN_<suffix> of the newest round that has one (the files are written by tools/final_profile.sh on the GPU box)."""
    import glob
    hits = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_' + suffix)))
    return hits[-1] if hits else os.path.join(ROOT, 'profiles', 'none_' + suffix)


# device kernel name (rocprofv3) -> the launch slot bench.py / g4r_profile report it under
KERNEL_ALIAS = {'k_score_s': 'k_score_fwd', 'k_score_mt': 'k_score_fwd', 'k_score_b': 'k_score_bwd', 'k_score_bmt': 'k_score_bwd', 'k_gru_fwd_fused': 'k_gru_fwd',
                'k_gru_bwd_fused': 'k_gru_bwd', 'k_score_bwd2': 'k_score_bwd', 'k_gru_p1w': 'k_gru_p1', 'k_gru_p1s': 'k_gru_p1', 'k_gru_p2w': 'k_gru_p2',
                'k_gru_bwd_aw': 'k_gru_bwd_a', 'k_gru_bwd_bw': 'k_gru_bwd_b', 'k_dense_grad2': 'k_dense_grad'}


def rocprof_means(config):
    """{bench kernel name: mean us} from the tracked rocprofv3 --kernel-trace --stats CSV of this config (+ '__file__')."""
    import csv
    path = newest_profile('kernel_stats_rocprofv3_%s.csv' % config)
    if not os.path.exists(path):
        return {}
    out = {'__file__': os.path.relpath(path, ROOT)}
    alias = KERNEL_ALIAS
    acc = {}
    for r in csv.DictReader(open(path)):
        name = r['Name'].split('(')[0].replace('void ', '').split('<')[0].strip()
        name = alias.get(name, name)
        t = acc.setdefault(name, [0.0, 0])
        t[0] += float(r['TotalDurationNs']); t[1] += int(r['Calls'])
    for name, (ns, n) in acc.items():
        if n:
            out[name] = ns / n / 1000.0
    return out


def make_plan(cfg, n_steps, rank, nranks, seed=42, session_items=0):
    """RSC15-shaped sessions -> (plan, number of sessions).  Enough sessions for n_steps full-batch steps.
    session_items: distinct items the SESSIONS are drawn from (0 = the whole catalogue); a smaller set is spread over the catalogue
    by a fixed random injection (round 1-3 used 200,000 for the 3 M / 10 M-item configs: input / target rows were more
    cache-resident than a real stream's; the negatives always covered the full catalogue)."""
    from gru4rec_amd import _native, synth
    B = cfg['batch_size']
    n_sessions = int(n_steps * B / 2.6) + 8 * B           # ~2.9 scoring events per session
    n_items = min(cfg['n_items'], session_items) if session_items > 0 else cfg['n_items']
    # the generator is deterministic; its output for a 10 M-item catalogue takes ~20 s of host time, so consecutive invocations on
    # one box (profile passes) share it through a file in the temp directory
    import tempfile
    cache = os.path.join(tempfile.gettempdir(), 'g4r_synth_%d_%d_%d_v1.npz' % (n_sessions * nranks, n_items, seed))
    if os.path.exists(cache):
        z = np.load(cache)
        sizes, ids, items_all = z['sizes'], z['ids'], z['items_all']
    else:
        data = synth.make_sessions(n_sessions * nranks, n_items=n_items, seed=seed)
        sizes = data.groupby('SessionId').size().values
        ids, inv = np.unique(data.ItemId.values, return_inverse=True)
        items_all = inv.astype(np.int32)
        try:
            tmp = cache + '.%d.npz' % os.getpid()
            np.savez(tmp, sizes=sizes, ids=ids, items_all=items_all)
            os.replace(tmp, cache)
        except OSError:
            pass
    offs = np.zeros(len(sizes) + 1, dtype=np.int64)
    offs[1:] = np.cumsum(sizes)
    if cfg['n_items'] > len(ids):
        # the sessions name len(ids) distinct items (np.unique made them consecutive): spread them over the full catalogue with a
        # fixed random injection, so that their rows lie all over the table and not in its first pages
        rng = np.random.RandomState(seed + 1)
        spread = np.sort(rng.choice(cfg['n_items'], size=len(ids), replace=False)).astype(np.int32)
        items_all = spread[items_all]
    mine = np.arange(rank, len(sizes), nranks)
    lens = sizes[mine]
    sub_off = np.zeros(len(mine) + 1, dtype=np.int32)
    sub_off[1:] = np.cumsum(lens)
    idx = np.concatenate([np.arange(offs[s], offs[s + 1]) for s in mine])
    items = items_all[idx]
    support = np.bincount(items_all, minlength=cfg['n_items']).astype(np.float64) + 1.0
    plan = _native.build_plan(sub_off, np.arange(len(mine)), items, B, cfg['n_sample'])
    return plan, support


def create_model(cfg, support, rank, nranks, device, unique_id, use_graph=True, seed=12345, sample_store=10000000, sparse_exact=False, defer=False):
    from gru4rec_amd import _native
    from gru4rec_amd.gru4rec import _parse_act
    fa = _parse_act(cfg['final_act'], True)
    m = _native.Model(
        n_items=cfg['n_items'], layers=cfg['layers'], batch_size=cfg['batch_size'], n_sample=cfg['n_sample'],
        loss=_native.LOSS_IDS[cfg['loss']], final_act=fa[0], final_act_p0=fa[1], final_act_p1=fa[2],
        hidden_act=_native.ACT_IDS['tanh'], embed_mode=_native.EMBED_CONSTRAINED, embedding=0,
        learning_rate=cfg['learning_rate'], momentum=cfg['momentum'], lmbd=0.0, bpreg=cfg['bpreg'], logq=cfg['logq'],
        sample_alpha=cfg['sample_alpha'], dropout_p_hidden=cfg['dropout_p_hidden'],
        dropout_p_embed=cfg['dropout_p_embed'], sample_store=sample_store, seed=seed + (0 if sparse_exact else 7919 * rank), device=device,
        rank=rank, nranks=nranks, use_graph=1 if use_graph else 0, sparse_exact=3 if sparse_exact else 0, defer_updates=1 if defer else 0)
    if nranks > 1:
        m.comm_init(unique_id, nranks, rank)
    elif os.environ.get('G4R_FORCE_STAGED'):      # diagnostic: the N > 1 data path with a one-rank communicator
        m.comm_init(_native.comm_unique_id(), 1, 0)
    # reference initialisation (gru4rec.py:252-294): uniform(-s, s), s = sqrt(6 / (fan_in + fan_out)) per block
    rng = np.random.RandomState(42)

    def init(shape):
        s = np.sqrt(6.0 / (shape[0] + shape[1]))
        return (rng.rand(*shape) * 2 * s - s).astype(np.float32)
    L = cfg['layers']
    for i, D in enumerate(L):
        n_in = L[i - 1] if i > 0 else L[-1]
        m.set_param('Wx', np.hstack([init((n_in, D)) for _ in range(3)]), i)
        m.set_param('Wh', init((D, D)), i)
        m.set_param('Wrz', np.hstack([init((D, D)) for _ in range(2)]), i)
    m.set_param('Wy', init((cfg['n_items'], L[-1])))
    pop = support ** cfg['sample_alpha']
    pop = pop.cumsum() / pop.sum()
    pop[-1] = 1
    lq_t = lq_s = None
    if cfg['logq']:
        p0 = support.astype(np.float32)
        lq_t, lq_s = np.log(p0), np.log(p0 ** np.float32(cfg['sample_alpha']))
    m.set_popularity(pop.astype(np.float32), lq_t, lq_s)
    return m


def cpu_baseline(cfg, plan, support, budget_s=15.0):
    """The NumPy oracle (CPU restatement of the reference step; Theano itself is not installable) timed on
    this box's host cores over the first steps of the same plan."""
    from oracle.model import OracleGRU4Rec
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get('num_threads', 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count()
    o = OracleGRU4Rec(n_items=cfg['n_items'], layers=tuple(cfg['layers']), batch_size=cfg['batch_size'],
                      loss=cfg['loss'], final_act=cfg['final_act'], n_sample=cfg['n_sample'],
                      sample_alpha=cfg['sample_alpha'], learning_rate=cfg['learning_rate'], momentum=cfg['momentum'],
                      bpreg=cfg['bpreg'], logq=cfg['logq'], dropout_p_hidden=cfg['dropout_p_hidden'],
                      dropout_p_embed=cfg['dropout_p_embed'], constrained_embedding=True)
    o.set_popularity(support)
    o.make_sample_store(cfg['n_sample'] * 64)       # a short store: the refill cost is not what is being timed
    for t in range(3):
        o.train_step(plan['in_idx'][t], plan['out_idx'][t], int(plan['M'][t]), plan['reset'][t])
    # three consecutive windows of budget_s / 3 each, the BEST one reported: the host is a shared 256-CPU box and a single 15 s
    # sample swung 28 - 38 mini-batches/s between rounds on the same code (context, not a comparator: the roofline fraction is)
    t = 3
    wins = []
    for w in range(3):
        t0 = time.time()
        n = ev = 0
        while time.time() - t0 < budget_s / 3.0 and t < plan['T']:
            M = int(plan['M'][t])
            o.train_step(plan['in_idx'][t], plan['out_idx'][t], M, plan['reset'][t])
            n += 1
            ev += M
            t += 1
        dt = time.time() - t0
        if n:
            wins.append((n / dt, n, ev, dt))
    if not wins:      # (a plan too short for a single step per window)
        return dict(value=0.0, unit='mini-batches/s', cores=int(threads), kind='port', sample='no step ran (plan of %d steps)' % plan['T'], host_cpus=os.cpu_count())
    best = max(wins)
    return dict(value=best[0], unit='mini-batches/s', cores=int(threads), kind='port',
                sample='best of %d windows of %.0f s: %d steps (%d events) of the same plan, NumPy/BLAS fp32 oracle, %.1f s; all windows: %s' % (
                    len(wins), budget_s / 3.0, best[1], best[2], best[3], ', '.join('%.1f' % w[0] for w in wins)),
                events_per_s=best[2] / best[3], host_cpus=os.cpu_count())


def large_catalogue_legs(timeout=150):
    """`bench.py --config cfg4m` without / with --defer in child processes -> {'immediate': {...}, 'deferred': {...}} (mini-batches/s, us per
    step, the update launch, and for the deferred run the flush launch priced on the bytes it moves)."""
    import subprocess
    res = {'config': 'cfg4m: B = 512, n_sample = 8192, layers = [256], BPR-max over a 2,000,000-item catalogue (item table 2 GB + 2 GB accumulators)'}
    for key, extra in (('immediate', []), ('deferred', ['--defer'])):
        cmd = [sys.executable, os.path.abspath(__file__), '--config', 'cfg4m', '--steps', '600', '--warmup', '100', '--profile-steps', '128', '--long-steps', '0',
               '--no-cpu-baseline', '--no-micro', '--no-defer-leg'] + extra
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
            if not line:
                res[key] = {'error': (r.stderr or r.stdout)[-300:]}
                continue
            o = json.loads(line[-1])
            g = o.get('roofline_gather_scatter') or {}
            kk = o.get('kernels') or {}
            gemm = {n: {'avg_us': kk[n].get('avg_us'), 'achieved': kk[n].get('achieved'), 'unit': kk[n].get('unit'), 'frac_of_fp32_mfma_peak': kk[n].get('frac')}
                    for n in ('k_score_fwd', 'k_score_bwd') if n in kk and kk[n].get('bound') == 'mfma'}
            res[key] = {'value': o['value'], 'unit': o['unit'], 'us_per_step': 1000.0 * o['ms_per_step'], 'steps': o['steps'],
                        'update_launch': {'kernel': g.get('kernel'), 'avg_us': g.get('avg_us'), 'frac_of_8TBps_on_survey_8d_bytes': g.get('frac')},
                        'scoring_gemms': gemm,      # the macro-tile scoring kernels (k_score_mt / k_score_bmt), HIP events on the dispatches
                        'deferred_flush': g.get('deferred_flush')}
        except Exception as e:      # noqa: BLE001
            res[key] = {'error': '%s: %s' % (type(e).__name__, e)}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=6000)
    ap.add_argument('--warmup', type=int, default=600)
    ap.add_argument('--config', default='cfg2', choices=sorted(CONFIGS))
    ap.add_argument('--profile-steps', type=int, default=300)
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-micro', action='store_true', help='skip the row gather / scatter micro-benchmark object')
    ap.add_argument('--defer', action='store_true', help='deferred row updates (g4r_config::defer_updates): one flush launch per window of 16 steps; bit-identical '
                    'results, the flush launch is priced in roofline_gather_scatter.deferred_flush')
    ap.add_argument('--sparse-exact', action='store_true', help='N > 1 (or G4R_FORCE_STAGED=1): the exact-replica mode (REDUCE form) instead of GPU-local '
                    'item rows + reconciliation: per-occurrence gradient rows all-gathered every step, nothing to reconcile')
    ap.add_argument('--session-items', type=int, default=0, help='distinct items the synthetic sessions are drawn from (0 = the whole catalogue of the '
                    'config; rounds 1-3 used 200000 for cfg3 / cfg4)')
    ap.add_argument('--no-defer-leg', action='store_true', help='default line only: skip the two short large-catalogue legs (BASELINE configs[3] shape over a '
                    '2 M-item catalogue, without / with deferred row updates) that fill roofline_gather_scatter.deferred_flush and large_catalogue')
    ap.add_argument('--long-steps', type=int, default=2000, help='a run of --steps below this also times that many steps behind the timed region and '
                    'reports them as "long_run" (a 20-step window is ~1 ms; 0 = off)')
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    from gru4rec_amd import launch
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher (one process per GPU, as torch.distributed.run would)
        raise SystemExit(launch.spawn(__file__, sys.argv[1:], args.gpus))
    rank, world, local_rank = launch.layout()
    if world != args.gpus:
        raise SystemExit('WORLD_SIZE (%d) != --gpus (%d)' % (world, args.gpus))
    from gru4rec_amd import _native
    if _native.device_count() <= 0:
        raise SystemExit('bench.py needs an MI355X: the hot path has no CPU fallback')
    if world > 1 and _native.device_count() <= local_rank:
        raise SystemExit('rank %d: LOCAL_RANK %d but only %d GPU(s) visible' % (rank, local_rank, _native.device_count()))
    unique_id = launch.unique_id(rank, world)      # rank 0 creates the RCCL id, the others read it (file rendezvous; no torch)
    # per-kernel durations behind the timed region: every rank runs those steps (the all-reduce is collective), rank 0 reports
    n_profile = args.profile_steps
    n_long = args.long_steps if (0 < args.steps < args.long_steps) else 0
    total_steps = args.warmup + args.steps + n_long + n_profile + max(8, n_profile // 4) + 8 + 64 + (min(args.steps, 2000) if (world > 1 or os.environ.get('G4R_FORCE_STAGED')) else 0)
    plan, support = make_plan(cfg, total_steps, rank, world, session_items=args.session_items)
    assert plan['T'] >= total_steps, 'synthetic plan too short: %d < %d' % (plan['T'], total_steps)
    assert (plan['M'][:total_steps] == cfg['batch_size']).all()
    m = create_model(cfg, support, rank, world, local_rank if world > 1 else 0, unique_id, use_graph=not args.no_graph, sparse_exact=args.sparse_exact, defer=args.defer)
    n_ranks = m.comm_nranks()      # what RCCL reports for the communicator (1 without one): n_gpus in the output is THIS number
    if n_ranks != world:
        raise SystemExit('RCCL communicator has %d rank(s), expected %d' % (n_ranks, world))
    for k in ('in_idx', 'out_idx', 'reset', 'M'):
        plan[k] = plan[k][:total_steps]
    plan['T'] = total_steps
    plan['n_compact'] = 0
    m.set_plan(plan)
    m.reset_hidden()
    # N > 1 with GPU-local item rows: fit() reconciles the item tables every `sync_every` steps -- the model does not train without
    # it --, so the TIMED region runs with it wherever the library does it on the stream inside g4r_train_steps
    # (item tables up to 64 MB; the one-rank communicator of G4R_FORCE_STAGED=1 takes the same path, as for 8 ranks)
    staged_mode = (world > 1 or bool(os.environ.get('G4R_FORCE_STAGED'))) and not args.sparse_exact
    sync_in_timed = 0
    if staged_mode:
        from gru4rec_amd.gru4rec import GRU4Rec
        g0 = GRU4Rec(layers=list(cfg['layers']))
        g0.n_items = cfg['n_items']
        k0 = g0.sync_steps(world if world > 1 else 8)
        try:
            if k0 and m.set_sync_every(k0):
                sync_in_timed = int(k0)
        except Exception:
            sync_in_timed = 0
    m.train_steps(0, args.warmup)

    def barrier():
        if world > 1:
            launch.barrier(m)      # a max-reduce on the communicator: returns when every rank has entered it
    barrier()
    launch.cleanup(rank)
    t0 = time.perf_counter()
    m.train_steps(args.warmup, args.steps)        # synchronous at return (hipStreamSynchronize inside)
    dt_own = time.perf_counter() - t0
    barrier()
    dt = dt_own
    rank_dts = [dt_own]
    if world > 1:
        rank_dts = launch.gather_us(m, rank, world, dt_own)
        dt = max(rank_dts)                        # MAX over ranks
    losses = m.get_losses(args.warmup, args.steps)
    long_run = None
    if n_long:
        # the same measurement over >= 2000 steps right behind the timed region (the driver's --steps 20 is a ~1 ms window)
        barrier()
        t1 = time.perf_counter()
        m.train_steps(args.warmup + args.steps, n_long)
        dl = time.perf_counter() - t1
        barrier()
        if world > 1:
            dl = launch.max_over_ranks_us(m, dl)
        long_run = {'steps': n_long, 'value': n_long * world / dl, 'unit': 'mini-batches/s', 'ms_per_step': 1000.0 * dl / n_long,
                    'note': 'same workload, the next %d plan steps, timed the same way (barrier / synchronize on both sides, max over ranks)' % n_long}
    events = int(plan['M'][args.warmup:args.warmup + args.steps].sum()) * world
    out = {
        'metric': 'mini-batches/sec (gru4rec.py:661), RSC15-shaped batch=128 n_sample=2048 BPR-max' if args.config == 'cfg2'
        else 'mini-batches/sec (gru4rec.py:661), %s: batch=%d n_sample=%d %s' % (args.config, cfg['batch_size'], cfg['n_sample'], cfg['loss']),
        'value': args.steps * world / dt, 'unit': 'mini-batches/s', 'n_gpus': n_ranks, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': 1000.0 * dt / args.steps, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': (args.steps * world / dt) / A30_PUBLISHED_MBS if args.config == 'cfg2' else None,
        'baseline_note': 'BASELINE.md: ~1240 mb/s, Theano on an NVIDIA A30 (closest published point; no MI355X number exists)',
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'BASELINE.json configs[1]: RSC15-shaped synthetic sessions (%d items), layers=%s, batch=%d, '
                               'n_sample=%d, %s, constrained_embedding, Adagrad' % (
                                   cfg['n_items'], cfg['layers'], cfg['batch_size'], cfg['n_sample'], cfg['loss'])
                   if args.config == 'cfg2' else args.config,
                   'global_batch': cfg['batch_size'] * world, 'parallelism': 'session-sharded dp%d' % world,
                   'item_rows': 'exact replicas (REDUCE form: gradient rows exchanged every step)' if args.sparse_exact else ('gpu-local, reconciled every sync_every steps' if world > 1 else 'single GPU'),
                   'hip_graph': not args.no_graph,
                   'session_items': int(min(cfg['n_items'], args.session_items) if args.session_items > 0 else cfg['n_items']),
                   'distinct_items_in_plan': int(len(np.unique(np.concatenate([plan['in_idx'].ravel(), plan['out_idx'].ravel()]))))},
        'events_per_s': events / dt, 'loss_first': float(losses[0]), 'loss_last': float(losses[-1]),
        'loss_finite': bool(np.isfinite(losses).all()),
    }
    out['config']['defer_updates'] = bool(args.defer)
    if long_run:
        out['long_run'] = long_run
    if sync_in_timed:
        m.set_sync_every(0)      # the per-kernel passes below time the bare step
    kt = {}
    defer_rows = None
    if n_profile > 0:
        # per-kernel durations: HIP events on the library's own stream, eager launches over the next plan steps
        d0 = m.get_debug('defer_stats', 4) if args.defer else None
        m.profile(True)
        m.train_steps(args.warmup + args.steps + n_long, n_profile)
        m.profile(False)
        kt = m.kernel_times()
        if args.defer:
            d1 = m.get_debug('defer_stats', 4)
            defer_rows = (float(d1[0] - d0[0]), float(d1[1] - d0[1])) if d1[2] else None
    staged = (world > 1 or bool(os.environ.get('G4R_FORCE_STAGED'))) and not args.sparse_exact
    kt_split = {}
    if n_profile > 0 and world == 1 and not staged:
        # the same again with the update launch split into its two roles (g4r_profile(m, 2)): the embedding gather / scatter -- the
        # kernel north_star prices against the HBM roofline -- timed ALONE, next to the merged launch the step actually runs
        n_split = max(8, n_profile // 4)
        m.profile(2)
        m.train_steps(args.warmup + args.steps + n_long + n_profile, n_split)
        m.profile(False)
        kt_split = m.kernel_times()
    reconcile = None
    if staged:
        # The timed region is the training step north_star defines for N > 1 (dense-gradient all-reduce every step, item rows
        # GPU-local).  GRU4Rec.fit also reconciles the item tables every `sync_every` steps (without it the
        # replicas drift apart); that cost is measured here, separately, over the rows the next `sync_every` steps touch, and
        # reported next to `value` -- never inside it, never hidden.
        from gru4rec_amd.gru4rec import GRU4Rec
        g_ = GRU4Rec(layers=list(cfg['layers']))
        g_.n_items = cfg['n_items']
        K = g_.sync_steps(world if world > 1 else 8) or 16      # what fit() uses at this rank count (one-rank communicator: as for 8) and catalogue size ('auto')
        try:
            t_sync = []
            base_t = args.warmup + args.steps + n_long + n_profile + max(8, n_profile // 4)
            spare = total_steps - base_t - 1
            reps = max(0, min(3, spare // max(K, 1)))
            m.comm_sync_sparse()                      # rows of the whole run so far: not timed
            for r in range(reps):
                m.train_steps(base_t + r * K, K)
                barrier()
                t2 = time.perf_counter()
                m.comm_sync_sparse()
                t_sync.append(time.perf_counter() - t2)
            measured = None
            n_rec = min(args.steps, 2000) // K * K
            step_only = None
            if t_sync and n_rec >= K and spare - reps * K >= n_rec and (sync_in_timed or m.set_sync_every(K)):
                # small item tables: the library reconciles inside g4r_train_steps (every K steps, on the stream).  When the timed
                # region already ran that way, this window times the BARE step instead (value_step_only); else it times the step
                # with the reconciliation inside
                t_rec0 = base_t + reps * K
                barrier()
                t3 = time.perf_counter()
                m.train_steps(t_rec0, n_rec)
                d_rec = time.perf_counter() - t3
                barrier()
                if world > 1:
                    d_rec = launch.max_over_ranks_us(m, d_rec)
                m.set_sync_every(0)
                win = {'steps': n_rec, 'value': n_rec * world / d_rec, 'ms_per_step': 1000.0 * d_rec / n_rec}
                if sync_in_timed:
                    step_only = win
                else:
                    measured = dict(win, reconciliations_inside=int(m.get_debug('dev_syncs', (1,))[0]))
            if t_sync:
                ms = 1000.0 * (launch.max_over_ranks_us(m, min(t_sync)) if world > 1 else min(t_sync))
                step_ms = 1000.0 * dt / args.steps
                reconcile = {'sync_every': K, 'ms_per_reconciliation': ms, 'ms_per_step_amortised': ms / K,
                             'value_with_reconciliation': None if sync_in_timed else args.steps * world / (dt + args.steps * ms / K / 1000.0),
                             'measured_with_reconciliation_inside_train_steps': measured, 'step_only_window': step_only,
                             'timed_region_includes_reconciliation': bool(sync_in_timed),
                             'note': 'g4r_comm_sync_sparse over the item rows %d steps touch (best of %d, max over ranks); fit() runs one every %d '
                                     'steps: step %.4f ms + %.4f ms amortised' % (K, len(t_sync), K, step_ms, ms / K)}
        except Exception as e:      # the reconciliation must not take the step measurement down with it
            reconcile = {'error': str(e)}
    if sync_in_timed:
        # the timed region itself ran with the reconciliation every sync_every steps on the stream: `value` is what fit() achieves
        out['value_source'] = ('the timed --steps window itself: g4r_train_steps reconciles the GPU-local item tables every %d steps on the stream '
                               '(barrier / synchronize on both sides, max over ranks); the bare step is value_step_only' % sync_in_timed)
        so = (reconcile or {}).get('step_only_window') if reconcile and 'error' not in reconcile else None
        if so:
            out['value_step_only'], out['ms_per_step_step_only'] = so['value'], so['ms_per_step']
    elif world > 1 and reconcile and 'error' not in reconcile:
        # N > 1: the headline is the throughput fit() achieves -- WITH the reconciliation of the item tables the model cannot train
        # without.  Measured directly when the library reconciles inside g4r_train_steps (small tables), else the
        # timed step plus the measured cost of a reconciliation amortised over sync_every steps.  The bare step (what north_star
        # literally describes: all-reduce every step, item rows GPU-local) stays next to it as value_step_only.
        meas = reconcile.get('measured_with_reconciliation_inside_train_steps')
        out['value_step_only'] = out['value']
        out['ms_per_step_step_only'] = out['ms_per_step']
        if meas:
            out['value'], out['ms_per_step'] = meas['value'], meas['ms_per_step']
            out['value_source'] = ('%d steps with a reconciliation every %d steps inside g4r_train_steps (barrier / synchronize on both sides, max '
                                   'over ranks); the --steps window without reconciliation is value_step_only' % (meas['steps'], reconcile['sync_every']))
        else:
            out['value'] = reconcile['value_with_reconciliation']
            out['ms_per_step'] = 1000.0 * world / out['value']
            out['value_source'] = ('timed --steps window + measured ms_per_reconciliation / sync_every (host-driven packed-parts exchange: the '
                                   'item tables are too large for the on-stream dense form)')
        out['events_per_s'] = out['events_per_s'] * out['value'] / out['value_step_only']
        if args.config == 'cfg2':
            out['vs_baseline'] = out['value'] / A30_PUBLISHED_MBS
    if world > 1:
        def us(name):
            return 1000.0 * kt[name][0] / max(kt[name][1], 1) if name in kt else None
        out['multi_gpu'] = {
            'ncclCommCount': n_ranks, 'rank_ms_per_step': [1000.0 * x / args.steps for x in rank_dts],
            'rccl_allreduce_us': us('rccl_allreduce'), 'k_dense_apply_us': us('k_dense_apply'),
            'allreduce': 'p2p one-shot (k_p2p_allreduce, G4R_P2P=1)' if m.p2p_active() else 'rccl',
            'dense_gradient_bytes': 4 * int(m.get_debug('dense_count', (1,))[0]),
            'step_graph_holds_allreduce': bool(m.get_debug('graph_mode', (1,))[0] == 1.0),
            'kernel_us_rank0': {k: 1000.0 * v[0] / max(v[1], 1) for k, v in kt.items()}, 'reconciliation': reconcile,
            'note': 'rank_ms_per_step: every rank\'s own wall time of the timed region / steps (value uses the max); all-reduce / dense '
                    'apply: HIP events around the eager launches of %d profile steps on rank 0 (every step synchronises there, so the '
                    'all-reduce time includes the skew between ranks)' % n_profile}
    if world == 1 and reconcile is not None:
        out['reconciliation_one_rank_communicator'] = reconcile
    if world == 1 and os.environ.get('G4R_FORCE_STAGED'):
        # What 8 GPUs would do, from the pieces one GPU can measure + the two numbers it cannot (stated, with a range): the latency of
        # the 8-rank collectives over xGMI.  The driver's SCALE run replaces this with a measurement.
        step_us = 1e6 * dt / args.steps
        if sync_in_timed and reconcile and reconcile.get('step_only_window'):
            step_us = 1000.0 * reconcile['step_only_window']['ms_per_step']      # (the timed region holds the one-rank reconciliation: price the bare step)
        dense_bytes = 4 * int(m.get_debug('dense_count', (1,))[0])
        R, D = 2 * cfg['batch_size'] + cfg['n_sample'], cfg['layers'][-1]
        proj = {'n_gpus': 8, 'one_gpu_fused_step_us_reference': None, 'measured_on_this_gpu': {'step_us_with_one_rank_collectives': step_us},
                'assumed': {'allreduce_dense_gradients_8_ranks_us': [10.0, 20.0], 'dense_gradient_bytes': dense_bytes,
                            'note': 'a %d KB all-reduce is latency bound (2 x 7 ring hops or one LL round); the one-rank collective inside the measured '
                                    'step already costs its launch' % (dense_bytes // 1024)}}
        if args.sparse_exact:
            # ONE collective per step: the all-gather of the ranks' blocks (occurrence list, gradient rows, bias gradients, raw dense
            # gradients); every rank receives 7 blocks over its 7 xGMI links
            blk = (R * D + 2 * R) * 4 + dense_bytes
            rates = [100.0, 50.0]      # GB/s per link actually sustained by a message of this size (peak 153)
            ag = [blk / (r * 1e3) + 5.0 for r in rates]      # each link carries one block; + launch / handshake
            proj['assumed'] = {'allgather_block_bytes_per_rank': blk, 'allgather_bytes_received_per_rank': 7 * blk,
                               'per_link_GBps': rates, 'allgather_8_ranks_us': ag,
                               'note': 'no separate all-reduce in this mode: the dense gradients ride in the block and every rank sums the eight copies itself; '
                                       'the one-rank all-gather inside the measured step already costs its launch'}
            # the joint update walks 8 x 2B + n_sample list entries at eight ranks where the one-rank run walks 2B + n_sample, and sums eight
            # rows per shared negative: scale the measured launch with the list length
            ksp = 1000.0 * kt['k_sparse_update'][0] / max(kt['k_sparse_update'][1], 1) if 'k_sparse_update' in kt else 0.0
            grow = (8 * 2 * cfg['batch_size'] + cfg['n_sample']) / float(R) - 1.0
            proj['measured_on_this_gpu']['joint_update_us_one_rank'] = ksp
            proj['assumed']['joint_update_extra_us_8_ranks'] = ksp * grow
            lo = step_us + ag[0] + ksp * grow
            hi = step_us + ag[1] + ksp * grow
            proj['mode'] = 'exact replicas (REDUCE form), one all-gather per step'
        else:
            per_call_ms = reconcile.get('ms_per_reconciliation') if reconcile and 'error' not in reconcile else None
            K = reconcile.get('sync_every') if reconcile and 'error' not in reconcile else None
            tab = cfg['n_items'] * (2 * D + 3) * 4
            proj['measured_on_this_gpu']['reconciliation_ms_per_call_one_rank'] = per_call_ms
            proj['measured_on_this_gpu']['sync_every'] = K
            if tab <= 64 * 1024 * 1024:
                # small item tables: the on-stream dense form, one all-reduce of [n_items][widths + 1] floats per call
                proj['assumed']['dense_reconciliation_allreduce_8_ranks_ms'] = [tab / 200e9 * 1e3, tab / 100e9 * 1e3]
                proj['assumed']['reconciliation_bytes'] = tab
                am = [(per_call_ms or 0.0) * 1000.0 / max(K or 16, 1) + x * 1000.0 / max(K or 16, 1) for x in proj['assumed']['dense_reconciliation_allreduce_8_ranks_ms']]
            else:
                # catalogues beyond the dense form: the packed-parts exchange moves the rows touched since the last call -- on a
                # catalogue this size nearly every gathered row is a new one, so the volume per STEP does not shrink with the interval
                row_bytes = (2 * D + 2) * 4
                vol = R * row_bytes
                proj['assumed']['exchange_bytes_per_rank_and_step'] = vol
                proj['assumed']['allgather_rate_GBps'] = [200.0, 400.0]
                proj['assumed']['note_exchange'] = ('rows touched per step x (parameter + accumulator row): every rank receives 7 x that; only a reconciliation '
                                                    'rare enough for the touched set to saturate (epoch end) amortises it -- DESIGN.md section 7')
                am = [7 * vol / (r * 1e3) for r in (400.0, 200.0)]
            lo = step_us + 10.0 + am[0]
            hi = step_us + 20.0 + am[1]
            proj['mode'] = 'gpu-local item rows, reconciled every %s steps' % K
        proj['step_us_8_gpus'] = [lo, hi]
        proj['mini_batches_per_s_8_gpus'] = [8e6 / hi, 8e6 / lo]
        out['projection_8_gpus'] = proj
    if rank == 0 and world == 1 and n_profile > 0:
        alg = algorithmic_cost(cfg)
        kern = {}
        for name, (ms, n) in kt.items():
            us = 1000.0 * ms / max(n, 1)
            e = {'avg_us': us, 'launches_per_step': n / n_profile}
            a = alg.get(name)
            if a:
                per = n / n_profile
                if a['bound'] == 'hbm':
                    e.update(bound='hbm', achieved=a['bytes'] / per / (us * 1e-6) / 1e9, unit='GB/s')
                else:
                    e.update(bound='mfma', achieved=a['flops'] / per / (us * 1e-6) / 1e12, unit='TFLOP/s',
                             bytes_GBps=a['bytes'] / per / (us * 1e-6) / 1e9)
                e['frac'] = e['achieved'] / PEAK[e['bound']][0]
            kern[name] = e
        # the second clock: mean duration per kernel in the tracked rocprofv3 --kernel-trace --stats summary of this command
        # (STATIC file under profiles/, written by tools/final_profile.sh on an earlier call; the profiler's own clock reads up to 10 %
        # more than the HIP events on the big launches, and its fraction is the one DESIGN.md quotes)
        rp = rocprof_means(args.config)
        for name, e in kern.items():
            r_us = rp.get(name)
            if r_us:
                e['rocprofv3_avg_us'] = r_us
                if 'frac' in e:
                    e['frac_rocprofv3'] = e['frac'] * e['avg_us'] / r_us
        out['kernels'] = kern
        out['kernel_clock_note'] = ('avg_us / frac: HIP events attached to the dispatches in THIS run; rocprofv3_avg_us / frac_rocprofv3: %s '
                                    '(static file of an earlier rocprofv3 --kernel-trace --stats pass of the same command)' % rp.get('__file__', 'no tracked rocprofv3 summary found'))
        out['kernel_time_sum_us_per_step'] = sum(1000.0 * ms / n_profile for ms, n in kt.values())
        dom = max(kern.items(), key=lambda kv: kv[1]['avg_us'] * kv[1]['launches_per_step'])
        out['dominant_kernel'] = dom[0]
        pmc_path = newest_profile('pmc_traffic_%s.json' % args.config)
        pmc = json.load(open(pmc_path))['kernels'] if os.path.exists(pmc_path) else {}

        def traffic_of(name):      # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (tools/pmc_summary.py), bytes per launch
            for k, v in pmc.items():
                if k == name or KERNEL_ALIAS.get(k) == name or k.startswith(name):
                    return v.get('traffic_bytes')
            return None
        pmc2_path = newest_profile('pmc_traffic_%s_no_merge.json' % args.config)
        pmc2 = json.load(open(pmc2_path))['kernels'] if os.path.exists(pmc2_path) else {}

        def traffic_split(name):      # the same counters from the passes run with G4R_NO_MERGE=1 (the update launch as its two roles)
            for k, v in pmc2.items():
                if k == name or KERNEL_ALIAS.get(k) == name or k.startswith(name):
                    return v.get('traffic_bytes')
            return None
        # the roofline entry: the DOMINANT kernel of the step (largest time per step), priced with its algorithmic flops / bytes
        dk, dv = dom
        if 'bound' in dv:
            a = alg[dk]
            out['roofline'] = {'kernel': dk, 'bound': dv['bound'], 'achieved': dv['achieved'], 'peak': PEAK[dv['bound']][0],
                               'unit': dv['unit'], 'frac': dv['frac'], 'traffic': traffic_of(dk), 'traffic_unit': 'bytes per launch',
                               'algorithmic_flops': a.get('flops'), 'algorithmic_bytes': a.get('bytes'),
                               'traffic_over_algorithmic_bytes': (traffic_of(dk) / a['bytes']) if (traffic_of(dk) and a.get('bytes')) else None,
                               'avg_us': dv['avg_us'],
                               'traffic_source': ('%s (STATIC file: 2 x FETCH_SIZE + WRITE_SIZE from separate rocprofv3 --pmc passes of this '
                                                  'command, tools/pmc_traffic.sh; not measured by the run that prints this line)' %
                                                  os.path.relpath(pmc_path, ROOT)) if pmc else None,
                               'rocprofv3_avg_us': dv.get('rocprofv3_avg_us'), 'frac_rocprofv3': dv.get('frac_rocprofv3'),
                               'avg_us_source': 'HIP events attached to the dispatches (hipExtLaunchKernelGGL) in this run; rocprofv3_avg_us / '
                                                'frac_rocprofv3: the tracked rocprofv3 --kernel-trace summary of the same command (kernel_clock_note)',
                               'note': 'dominant kernel of the step by time; achieved = algorithmic %s per launch / mean launch duration '
                                       '(HIP events attached to the dispatches, %d steps); traffic = 2 x FETCH_SIZE + WRITE_SIZE from separate '
                                       'rocprofv3 --pmc passes (profiles/%s) when that file is present' % (
                                           'flops' if dv['bound'] == 'mfma' else 'bytes', n_profile, os.path.basename(pmc_path))}
        # the launch that holds the embedding gather / scatter north_star names: SPARSE bytes only (gradient row read +
        # parameter r/w + accumulator r/w per gathered occurrence, By path, index list); the dense parameters it also updates on a
        # single GPU are listed separately, not counted
        rk = 'k_update' if 'k_update' in kern else ('k_sparse_update' if 'k_sparse_update' in kern else None)
        if rk:
            k = kern[rk]
            sparse_bytes = alg['k_sparse_update']['bytes']
            gbps = sparse_bytes / (k['avg_us'] * 1e-6) / 1e9
            alone = None
            if 'k_sparse_update' in kt_split:
                ms_a, n_a = kt_split['k_sparse_update']
                us_a = 1000.0 * ms_a / max(n_a, 1)
                ms_d, n_d = kt_split.get('k_dense_grad', (0.0, 0))
                # bytes this kernel REALLY moves since round 4: single-occurrence rows (accumulator written in place by the producer
                # of the step row) cost a step-row read and a parameter read + write = 3 row transfers, not 8d's 5
                moved = (3 + mom_planes(cfg)) * (2 * cfg['batch_size'] + cfg['n_sample']) * cfg['layers'][-1] * 4
                alone = {'kernel': 'k_sparse_update', 'avg_us': us_a, 'launches': n_a,
                         # first the bytes the kernel MOVES (three row transfers per occurrence since round 4), then SURVEY 8d's five
                         'rows_moved_bytes': moved, 'achieved_on_rows_moved': moved / (us_a * 1e-6) / 1e9,
                         'frac_on_rows_moved': moved / (us_a * 1e-6) / 1e9 / 8000.0,
                         'survey_8d_bytes': sparse_bytes, 'achieved': sparse_bytes / (us_a * 1e-6) / 1e9,
                         'frac': sparse_bytes / (us_a * 1e-6) / 1e9 / 8000.0, 'unit': 'GB/s',
                         'traffic': traffic_split('k_sparse_update'),
                         'k_dense_grad_alone_us': (1000.0 * ms_d / n_d) if n_d else None,
                         'note': 'g4r_profile(m, 2): the sparse row update as a launch of its own (k_sparse_update; the dense-gradient tiles '
                                 'run as k_dense_grad next to it), HIP events on the dispatches; achieved / frac price it with SURVEY 8d\'s '
                                 'sparse bytes like the merged launch; rows_moved_bytes is what it reads and writes if every row is a '
                                 'single occurrence (3 row transfers: the accumulator is updated in place by the gradient producer)'}
            # Deferred row updates (k_sparse_flush): the row transfers of every update that could wait for the end of its window of
            # steps, in ONE launch per window.  Priced on the bytes the launch MOVES (step row read + parameter row read + write per
            # applied row, bias entry, and its scan of the window's pending list), with SURVEY 8d's five-transfer figure beside it.
            flush = None
            if 'k_sparse_flush' in kt and defer_rows:
                ms_f, n_f = kt['k_sparse_flush']
                us_f = 1000.0 * ms_f / max(n_f, 1)
                rows_per, bias_per = defer_rows[0] / max(n_f, 1), defer_rows[1] / max(n_f, 1)
                W_ = cfg['layers'][-1]
                moved_f = rows_per * 3 * W_ * 4 + bias_per * 12 + (n_profile / max(n_f, 1)) * (2 * cfg['batch_size'] + cfg['n_sample']) * 8
                flush = {'kernel': 'k_sparse_flush', 'avg_us': us_f, 'launches': n_f, 'rows_per_launch': rows_per,
                         'share_of_the_steps_rows': defer_rows[0] / max(1.0, n_profile * (2 * cfg['batch_size'] + cfg['n_sample'])),
                         'moved_bytes_per_launch': moved_f, 'achieved': moved_f / (us_f * 1e-6) / 1e9, 'unit': 'GB/s',
                         'frac': moved_f / (us_f * 1e-6) / 1e9 / 8000.0, 'survey_8d_bytes_per_launch': rows_per * 5 * W_ * 4,
                         'frac_on_survey_8d_bytes': rows_per * 5 * W_ * 4 / (us_f * 1e-6) / 1e9 / 8000.0,
                         'rocprofv3_avg_us': rp.get('k_sparse_flush'),
                         'note': 'one launch per window of <= 16 steps (a replay of the step graph): rows whose item is not gathered again inside '
                                 'the window (k_defer_scan, from the plan and the sample store) are applied here instead of in their step\'s '
                                 'k_update -- bit-identical results (tests/test_gpu_defer.py); avg_us: HIP events around the launch'}
            out['roofline_gather_scatter'] = {
                'kernel': rk, 'bound': 'hbm', 'achieved': gbps, 'peak': 8000.0, 'unit': 'GB/s', 'frac': gbps / 8000.0,
                'traffic': traffic_of(rk), 'sparse_bytes': sparse_bytes, 'avg_us': k['avg_us'],
                'rocprofv3_avg_us': k.get('rocprofv3_avg_us'), 'sparse_role_alone': alone, 'deferred_flush': flush,
                'dense_param_bytes_same_launch': (alg['k_update']['bytes'] - sparse_bytes) if rk == 'k_update' else 0,
                'note': 'sparse bytes per launch = (5 R D + 5 N + R) * 4, R = 2B + n_sample gathered rows, N = B + n_sample score columns '
                        '(SURVEY 8d); on one GPU the same launch also holds the dense-gradient tiles (their parameter / accumulator bytes '
                        'are listed, not counted)%s' % ('; the %d-item table (%.0f MB) is Infinity-Cache resident at this config' % (
                            cfg['n_items'], cfg['n_items'] * cfg['layers'][-1] * 4 / 1e6) if cfg['n_items'] * cfg['layers'][-1] * 4 < 2.5e8 else '')}
        if not args.no_micro:
            # the access pattern alone (tools/micro_rows.py for the full sweep): R random rows of a table far beyond the Infinity Cache
            W = max(4, (cfg['layers'][-1] + 3) // 4 * 4)
            R = 2 * cfg['batch_size'] + cfg['n_sample']
            n_big = int(6.2e9 / (4 * W))
            micro = []
            for mode, name, streams in ((1, 'gather_fused', 1), (2, 'adagrad_scatter', 5)):
                for mult in (1, 16):
                    k_us, _ = _native.bench_rows(n_big, W, R * mult, launches=100 if mult == 1 else 40, mode=mode)
                    nb = streams * R * mult * W * 4 + R * mult * 4
                    micro.append({'mode': name, 'rows_per_launch': R * mult, 'steps_batched': mult, 'width': W, 'table_GB': n_big * W * 4 / 1e9 * (2 if mode == 2 else 1),
                                  'kernel_us': k_us, 'GBps': nb / k_us / 1e3, 'frac_of_8TBps': nb / k_us / 1e3 / 8000.0})
            out['gather_scatter_micro'] = micro
        if args.config == 'cfg2' and not args.no_defer_leg and not args.defer and not args.no_cpu_baseline and 'roofline_gather_scatter' in out:
            # The embedding gather / scatter at the sizes where it IS bound by the HBM (north_star's >= 60 % claim): BASELINE configs[3]'s shape
            # over a 2 M-item catalogue (2 GB table + 2 GB of accumulators: 16 x the Infinity Cache; the full 10 M-item catalogue costs a
            # minute of host-side weight initialisation, tools/final_profile.sh runs it), once without and once with deferred row updates,
            # in child processes (their failure must not take this line down).  ~20 s each.
            out['large_catalogue'] = large_catalogue_legs()
            fl = (out['large_catalogue'].get('deferred') or {}).get('deferred_flush')
            if fl:
                out['roofline_gather_scatter']['deferred_flush'] = dict(fl, config='cfg4m: BASELINE configs[3] shape (B = 512, 8192 negatives, D = 256) over a 2 M-item catalogue')
        if not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(cfg, plan, support)
    barrier()
    m.close()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
