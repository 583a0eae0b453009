#!/usr/bin/env python
"""Recall@20 / MRR@20 of N virtual ranks against the single-rank run (gru4rec_amd/virtual_ranks.py): the A/B of the N > 1 designs.

Part A, the end-to-end workload of tests/test_gpu_e2e_recall.py (24 K sessions, 2.5 K items, the BASELINE configs[1] model): the
single-rank bars (B = 128 and the global batches), the GPU-local mode at its default reconciliation, the three forms of the
exact-replica mode (REDUCE / MEAN / SUM), weak scaling (N x 128) at 1 and 3 epochs and strong scaling (N x 128 / N).
Part B (--large), a configs[3]-like shape -- >= 1 M items, layers [256], B = 512, 8192 negatives: the GPU-local mode against
sync_every, the exact-replica REDUCE form, the single-rank bars; with the rows a reconciliation moves per call.
Writes gpurun_out/r04_virtual_ranks[_large].json (copy to profiles/).  Needs an MI355X.
      python tools/virtual_ranks_study.py [--quick] [--large [--items N] [--sessions N] [--ranks N]]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from gru4rec_amd import evaluation, synth  # noqa: E402
from gru4rec_amd.virtual_ranks import fit_virtual_ranks  # noqa: E402

PARAMS = dict(loss='bpr-max', final_act='elu-0.5', layers=[100], batch_size=128, n_sample=2048, constrained_embedding=True,
              learning_rate=0.1, bpreg=1.0, momentum=0.0, sample_alpha=0.75, n_epochs=1)


def arg(name, default):
    return type(default)(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


def study(params, train, test, store, plan, out_name, workload, eval_batch=100):
    rows = []

    def run(tag, n, sync_every='default', exact=False, **over):
        p = dict(params, **over)
        t0 = time.time()
        try:
            grus, st = fit_virtual_ranks(p, train, n, sample_store=store, sync_every=sync_every, sparse_exact=exact)
        except FloatingPointError:
            rows.append(dict(tag=tag, nranks=n, diverged=True, batch_per_rank=p['batch_size'], epochs=p['n_epochs']))
            print(json.dumps(rows[-1]), flush=True)
            return
        rec, mrr = evaluation.evaluate_gpu(grus[0], test.copy(), cut_off=[5, 20], batch_size=eval_batch, mode='standard')
        k = grus[0].sync_steps(n) if sync_every == 'default' else sync_every
        for g in grus:
            g.close()
        r = dict(tag=tag, nranks=n, batch_per_rank=p['batch_size'], epochs=p['n_epochs'], mode=('exact-' + str(exact)) if exact else 'gpu-local',
                 sync_every=None if exact else k, steps=st['steps'], events=st['events'], loss=st['loss'], recall20=float(rec[1]),
                 mrr20=float(mrr[1]), recall5=float(rec[0]), mrr5=float(mrr[0]), reconciled_rows=st['sync_rows'], reconciliations=st['syncs'],
                 rows_per_reconciliation=(st['sync_rows'] / st['syncs']) if st['syncs'] else 0.0, seconds=time.time() - t0)
        rows.append(r)
        print(json.dumps(r), flush=True)

    plan(run)
    base = rows[0]
    for r in rows:
        if 'recall20' in r:
            r['d_recall20'] = r['recall20'] - base['recall20']
            r['d_mrr20'] = r['mrr20'] - base['mrr20']
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', out_name), 'w') as f:
        json.dump(dict(workload=workload, rows=rows), f, indent=1)
    print('%-66s %8s %8s %9s %9s %9s %12s' % ('run', 'R@20', 'MRR@20', 'dR@20', 'dMRR@20', 'loss', 'rows/reconc.'))
    for r in rows:
        if 'recall20' in r:
            print('%-66s %8.4f %8.4f %+9.4f %+9.4f %9.4f %12.0f' % (r['tag'], r['recall20'], r['mrr20'], r['d_recall20'], r['d_mrr20'], r['loss'][-1], r['rows_per_reconciliation']))
        else:
            print('%-66s diverged (NaN cost)' % r['tag'])


def part_a(quick):
    data = synth.make_sessions(24000, n_items=2500, seed=17)
    train, test = synth.train_test_split(data, test_frac=0.1)

    def plan(run):
        run('1 rank, B=128 (the bar)', 1)
        for b in (256, 1024):
            run('1 rank at the global batch B=%d' % b, 1, batch_size=b)
        for n in (2, 4, 8):
            run('%d x 128, GPU-local rows, default sync_every' % n, n)
            run('%d x 128, exact-replica REDUCE' % n, n, exact='reduce')
        for n in (2, 4, 8):
            run('strong scaling %d x %d, exact-replica REDUCE' % (n, 128 // n), n, exact='reduce', batch_size=128 // n)
        run('strong scaling 4 x 32, exact-replica MEAN', 4, exact='mean', batch_size=32)
        run('strong scaling 4 x 32, exact-replica SUM', 4, exact='sum', batch_size=32)
        run('strong scaling 4 x 32, GPU-local rows, default sync_every', 4, batch_size=32)
        run('8 x 128, exact-replica MEAN', 8, exact='mean')
        run('8 x 128, exact-replica SUM', 8, exact='sum')
        if not quick:
            run('1 rank, 3 epochs', 1, n_epochs=3)
            run('1 rank at B=1024, 3 epochs', 1, n_epochs=3, batch_size=1024)
            run('8 x 128, 3 epochs, GPU-local rows, default sync_every', 8, n_epochs=3)
            run('8 x 128, 3 epochs, exact-replica REDUCE', 8, n_epochs=3, exact='reduce')
            run('8 x 16, 3 epochs, exact-replica REDUCE (strong scaling)', 8, n_epochs=3, exact='reduce', batch_size=16)
    study(PARAMS, train, test, 2048 * 640, plan, 'r04_virtual_ranks.json',
          'synth.make_sessions(24000, n_items=2500, seed=17), 10 % test split; BASELINE configs[1] model')


def part_b():
    items, sessions, n = arg('--items', 1000000), arg('--sessions', 240000), arg('--ranks', 8)
    params = dict(PARAMS, layers=[256], batch_size=512, n_sample=8192)
    data = synth.make_sessions(sessions, n_items=items, seed=23)
    train, test = synth.train_test_split(data, test_frac=0.05)
    print('configs[3]-like: %d items, %d train events, %d test events' % (items, len(train), len(test)), flush=True)

    def plan(run):
        run('1 rank, B=512 (the bar)', 1)
        run('1 rank at the global batch B=%d' % (512 * n), 1, batch_size=512 * n)
        for k in (4, 16, 64, None):
            run('%d x 512, GPU-local rows, reconcile %s' % (n, 'at epoch end' if k is None else 'every %d steps' % k), n, sync_every=k)
        run('%d x 512, exact-replica REDUCE' % n, n, exact='reduce')
        run('strong scaling %d x %d, exact-replica REDUCE' % (n, 512 // n), n, exact='reduce', batch_size=512 // n)
    study(params, train, test, 8192 * 64, plan, 'r04_virtual_ranks_large.json',
          'synth.make_sessions(%d, n_items=%d, seed=23), 5 %% test split; layers [256], B = 512, 8192 negatives (configs[3] shape at %d items), %d ranks' % (
              sessions, items, items, n), eval_batch=256)


if __name__ == '__main__':
    if '--large' in sys.argv:
        part_b()
    else:
        part_a('--quick' in sys.argv)
