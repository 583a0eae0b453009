"""End-to-end check at scale through the public class: synthetic RSC15-shaped sessions -> GRU4Rec.fit (epochs, as the
reference prints them) -> evaluate_gpu.  python tools/epoch_scale.py [n_sessions] [n_epochs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gru4rec_amd import synth, evaluation
from gru4rec_amd.gru4rec import GRU4Rec

n_sessions = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
n_epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
t0 = time.time()
data = synth.make_sessions(n_sessions, n_items=37483, seed=42)
train, test = synth.train_test_split(data)
print('data: %d train events, %d test events, %d items (%.1f s)' % (len(train), len(test), train.ItemId.nunique(), time.time() - t0))
gru = GRU4Rec(loss='bpr-max', final_act='elu-0.5', layers=[100], batch_size=128, n_sample=2048, sample_alpha=0.75, bpreg=1.0,
              learning_rate=0.1, constrained_embedding=True, n_epochs=n_epochs)
t0 = time.time()
gru.fit(train)
t_fit = time.time() - t0
print('fit wall %.2f s' % t_fit)
t0 = time.time()
rec, mrr = evaluation.evaluate_gpu(gru, test, cut_off=[1, 5, 20], batch_size=512)
print('eval wall %.2f s  recall@20 %.4f mrr@20 %.4f' % (time.time() - t0, rec[-1] if isinstance(rec, (list, tuple, np.ndarray)) else rec, mrr[-1] if isinstance(mrr, (list, tuple, np.ndarray)) else mrr))
