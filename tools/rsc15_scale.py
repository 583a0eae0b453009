"""The whole pipeline at the size of RSC15 (7.97 M sessions, ~31.4 M events, 37,483 items) through the public entry points:
TSV on disk -> eventio.read_events (native parser) vs pandas.read_csv -> GRU4Rec.fit (one epoch, the reference's own
mb/s line: steps / epoch wall time, gru4rec.py:661) -> evaluate_gpu on the last day's sessions.
python tools/rsc15_scale.py [n_sessions]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pandas as pd
from gru4rec_amd import eventio, evaluation
from gru4rec_amd.gru4rec import GRU4Rec

n_sessions = int(sys.argv[1]) if len(sys.argv) > 1 else 7_970_000
tmp = os.environ.get('TMPDIR', '/tmp')
rng = np.random.default_rng(42)
t0 = time.time()
lens = np.minimum(1 + rng.geometric(0.34, n_sessions), 200)
n = int(lens.sum())
sid = np.repeat(np.arange(1, n_sessions + 1, dtype=np.int64), lens)
# RSC15's head is flat (top item ~0.5 % of the events): Zipf with exponent 0.8 over 37,483 items
w = 1.0 / np.arange(1, 37484) ** 0.8
items = rng.choice(37483, size=n, p=w / w.sum()) + 214500000
start = np.sort(rng.integers(1396300000, 1412000000, n_sessions))
pos = np.arange(n) - np.repeat(np.cumsum(lens) - lens, lens)
frame = pd.DataFrame({'SessionId': sid, 'ItemId': items, 'Time': np.repeat(start, lens) + 37 * pos})
cut = np.quantile(start, 1.0 - 1.0 / 30)
is_test = np.repeat(start >= cut, lens)
paths = {k: os.path.join(tmp, 'rsc15_like_%s.tsv' % k) for k in ('train', 'test')}
frame[~is_test].to_csv(paths['train'], sep='\t', index=False)
frame[is_test].to_csv(paths['test'], sep='\t', index=False)
print('generated %d sessions, %d events (%.0f MB train file) in %.1f s' % (n_sessions, n, os.path.getsize(paths['train']) / 1e6, time.time() - t0))
del frame, sid, items

t0 = time.time(); ref = pd.read_csv(paths['train'], sep='\t', usecols=['SessionId', 'ItemId', 'Time'], dtype={'SessionId': 'int32', 'ItemId': 'str'})
t_pandas = time.time() - t0
t0 = time.time(); ids = ref['ItemId'].unique(); idx = pd.Series(np.arange(len(ids)), index=ids)[ref['ItemId'].values].values
t_pandas_idx = time.time() - t0
t0 = time.time(); train = eventio.read_events(paths['train']); t_native = time.time() - t0
same = np.array_equal(train['SessionId'].values, ref['SessionId'].values) and np.array_equal(train['ItemId'].cat.codes.values, idx)
print('read train events: pandas.read_csv %.2f s (+ unique/itemidmap lookup %.2f s) | native %.2f s (%d cores) | identical: %s' % (
    t_pandas, t_pandas_idx, t_native, os.cpu_count(), same))
del ref, idx
test = eventio.read_events(paths['test'])

gru = GRU4Rec(loss='bpr-max', final_act='elu-0.5', layers=[100], batch_size=128, n_sample=2048, sample_alpha=0.75, bpreg=1.0,
              learning_rate=0.1, constrained_embedding=True, n_epochs=1)
t0 = time.time(); gru.prepare(train); t_prep = time.time() - t0
t0 = time.time(); gru.run_epoch(0); t_epoch = time.time() - t0
st = gru.last_epoch_stats
print('prepare (item map, offsets, weights, device model, sample store) %.2f s | epoch %.2f s: %d steps, %.0f mb/s, %.0f events/s' % (
    t_prep, t_epoch, st['steps'], st['steps'] / st['seconds'], st['events'] / st['seconds']))
gru._download_weights()
t0 = time.time()
rec, mrr = evaluation.evaluate_gpu(gru, test, cut_off=[1, 5, 20], batch_size=512)
print('evaluate_gpu: %d test events in %.2f s  recall@20 %.4f mrr@20 %.4f' % (len(test), time.time() - t0, rec[-1], mrr[-1]))
