#!/usr/bin/env python
"""Recall@20 / MRR@20 of N virtual ranks against the single-rank run (gru4rec_amd/virtual_ranks.py) on the end-to-end workload of
tests/test_gpu_e2e_recall.py, plus the single-rank run at the GLOBAL batch size (what N ranks x batch B amount to if the item rows
were shared) and reconciliation every K steps instead of once per epoch.  Writes gpurun_out/r03_virtual_ranks.json (copy to
profiles/).  Needs an MI355X.      python tools/virtual_ranks_study.py [--quick]"""
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
STORE = 2048 * 640


def main():
    quick = '--quick' in sys.argv
    data = synth.make_sessions(24000, n_items=2500, seed=17)
    train, test = synth.train_test_split(data, test_frac=0.1)
    rows = []

    def run(tag, n, sync_every=None, rule=None, **over):
        p = dict(PARAMS, **over)
        t0 = time.time()
        grus, st = fit_virtual_ranks(p, train, n, sample_store=STORE, sync_every=sync_every, rule=rule)
        rec, mrr = evaluation.evaluate_gpu(grus[0], test.copy(), cut_off=[5, 20], batch_size=100, mode='standard')
        for g in grus:
            g.close()
        r = dict(tag=tag, nranks=n, batch_per_rank=p['batch_size'], epochs=p['n_epochs'], sync_every=sync_every, rule=rule, steps=st['steps'],
                 events=st['events'], loss=st['loss'], recall20=float(rec[1]), mrr20=float(mrr[1]), recall5=float(rec[0]), mrr5=float(mrr[0]),
                 reconciled_rows=st['sync_rows'], reconciliations=st['syncs'], seconds=time.time() - t0)
        rows.append(r)
        print(json.dumps(r), flush=True)

    run('1 rank, B=128 (the bar)', 1)
    for b in (256, 512, 1024):
        run('1 rank at the global batch B=%d' % b, 1, batch_size=b)
    # rule = (parameters, optimizer statistics) of the reconciliation: sum / mean of the deltas of the ranks that touched a row
    for rule in (('sum', 'sum'), ('mean', 'sum'), ('mean', 'mean')):
        for n in (2, 4, 8):
            for k in (None, 64, 16, 4):
                run('%d ranks, P %s / A %s, reconcile %s' % (n, rule[0], rule[1], 'at epoch end' if k is None else 'every %d steps' % k), n, sync_every=k, rule=rule)
    if not quick:
        run('1 rank, 3 epochs', 1, n_epochs=3)
        for rule in (('mean', 'sum'), ('mean', 'mean')):
            for k in (None, 16):
                run('8 ranks, 3 epochs, P %s / A %s, reconcile %s' % (rule[0], rule[1], 'at epoch end' if k is None else 'every %d steps' % k), 8, sync_every=k, rule=rule, n_epochs=3)
    base = rows[0]
    for r in rows:
        r['d_recall20'] = r['recall20'] - base['recall20']
        r['d_mrr20'] = r['mrr20'] - base['mrr20']
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'r03_virtual_ranks.json'), 'w') as f:
        json.dump(dict(workload='synth.make_sessions(24000, n_items=2500, seed=17), 10 % test split; BASELINE configs[1] model', rows=rows), f, indent=1)
    print('%-62s %8s %8s %9s %9s %9s' % ('run', 'R@20', 'MRR@20', 'dR@20', 'dMRR@20', 'loss'))
    for r in rows:
        print('%-62s %8.4f %8.4f %+9.4f %+9.4f %9.4f' % (r['tag'], r['recall20'], r['mrr20'], r['d_recall20'], r['d_mrr20'], r['loss'][-1]))


if __name__ == '__main__':
    main()
