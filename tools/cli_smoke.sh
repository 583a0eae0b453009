#!/bin/bash
# run.py end to end on an MI355X: TSV in, train, save, load, evaluate (the reference's CLI flow, run.py:80-140)
set -e
cd "$(dirname "$0")/.."
python - <<'PY'
from gru4rec_amd import synth
d = synth.make_sessions(20000, n_items=3000, seed=5)
tr, te = synth.train_test_split(d)
tr.to_csv('/tmp/g4r_train.tsv', sep='\t', index=False)
te.to_csv('/tmp/g4r_test.tsv', sep='\t', index=False)
print(len(tr), len(te))
PY
python run.py /tmp/g4r_train.tsv -ps loss=bpr-max,layers=64,constrained_embedding=True,n_sample=512,batch_size=64,n_epochs=3,final_act=elu-0.5 -s /tmp/g4r_model.pickle -t /tmp/g4r_test.tsv -m 1 5 20
python run.py /tmp/g4r_model.pickle -l -t /tmp/g4r_test.tsv -m 20 -e conservative
python run.py /tmp/g4r_train.tsv -ps loss=cross-entropy,layers=32,n_sample=256,batch_size=32,n_epochs=1,final_act=softmax -t /tmp/g4r_test.tsv -m 20 -lpm 2>&1 | tail -4
