"""The command line the reference's users type (run.py:1-155 of the reference): train from a TSV, save, reload with -l,
evaluate with every tie mode, with the device sampler and with --sample_store_on_cpu.  The checks are on the contract that
scripts depend on: exit code 0, the progress lines, one `Recall@N: x MRR@N: y` line per cut-off, the `PRIMARY METRIC:` line
paropt.py scrapes -- and that a reloaded model evaluates to exactly the numbers printed right after training."""
import os
import re
import subprocess
import sys

import pytest

from gru4rec_amd import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PS = 'loss=bpr-max,final_act=elu-0.5,layers=48,batch_size=32,n_sample=256,constrained_embedding=True,learning_rate=0.1,n_epochs=2,bpreg=0.5,sample_alpha=0.5'
LINE = re.compile(r'Recall@(\d+): ([0-9.]+) MRR@\d+: ([0-9.]+)')


def run_cli(*args, env=None):
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'run.py')] + list(args), capture_output=True, text=True, timeout=600,
                       env=None if env is None else dict(os.environ, **env))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


@pytest.fixture(scope='module')
def tsvs(tmp_path_factory):
    d = tmp_path_factory.mktemp('cli')
    data = synth.make_sessions(8000, n_items=600, seed=3)      # > 512 test sessions: run.py evaluates with batch_size=512
    train, test = synth.train_test_split(data, test_frac=0.1)
    paths = (str(d / 'train.tsv'), str(d / 'test.tsv'))
    train.to_csv(paths[0], sep='\t', index=False)
    test.to_csv(paths[1], sep='\t', index=False)
    return paths + (str(d),)


def metrics(out):
    return {int(c): (float(r), float(m)) for c, r, m in LINE.findall(out)}


def test_train_save_load_evaluate(tsvs):
    train, test, d = tsvs
    model = os.path.join(d, 'model.pickle')
    out = run_cli(train, '-ps', PS, '-t', test, '-m', '1', '5', '20', '-s', model, '-ss', str(256 * 64), '-pm', 'mrr', '-lpm')
    assert 'Started training' in out and out.count('Epoch') == 2 and 'Saving trained model to: ' + model in out
    first = metrics(out)
    assert sorted(first) == [1, 5, 20] and first[20][0] > 0.1
    assert first[1][0] <= first[5][0] <= first[20][0]
    primary = float(re.search(r'PRIMARY METRIC: ([0-9.eE+-]+)', out).group(1))
    assert abs(primary - first[1][1]) < 1e-6                       # -pm mrr: MRR at the FIRST cut-off (run.py:153 of the reference)
    again = metrics(run_cli(model, '-l', '-t', test, '-m', '1', '5', '20'))
    assert again == first                                          # the pickle holds everything prediction needs


@pytest.mark.parametrize('mode', ['conservative', 'median', 'tiebreaking'])
def test_eval_modes_from_a_saved_model(tsvs, mode):
    train, test, d = tsvs
    model = os.path.join(d, 'model.pickle')
    if not os.path.exists(model):
        run_cli(train, '-ps', PS, '-s', model, '-ss', str(256 * 64))
    std = metrics(run_cli(model, '-l', '-t', test))
    got = metrics(run_cli(model, '-l', '-t', test, '-e', mode))
    assert 'using {} mode'.format(mode) in run_cli(model, '-l', '-t', test, '-e', mode)
    assert got[20][0] <= std[20][0] + 1e-9                         # no mode ranks a target better than `standard`
    assert got[20][0] >= std[20][0] - 0.01                         # and real-valued scores have almost no ties


def test_host_sampler_flag(tsvs):
    train, test, _ = tsvs
    out = run_cli(train, '-ps', PS, '-t', test, '--sample_store_on_cpu', '-ss', str(256 * 64))
    assert 'WARNING! The sample store is set to be on the CPU' in out
    assert '(type=CPU)' in out
    assert metrics(out)[20][0] > 0.1


def test_gpus_flag_rank_path_with_a_one_rank_communicator(tsvs):
    """`run.py --gpus N` (not in the reference): every rank calls set_distributed, creates its communicator inside fit, takes part in
    the per-chunk / per-epoch collectives and in the reconciliation of the item tables; rank 0 saves and evaluates.  A 1-GPU box runs
    that path as ONE rank (G4R_FORCE_STAGED=1: staged dense gradients -> RCCL all-reduce captured in the step graph -> dense apply),
    and the model it trains must evaluate like the fused single-GPU run (same math, other summation order in the dense apply)."""
    train, test, d = tsvs
    plain = metrics(run_cli(train, '-ps', PS, '-t', test, '-ss', str(256 * 64)))
    out = run_cli(train, '-ps', PS, '-t', test, '-ss', str(256 * 64), '--gpus', '1', env={'G4R_FORCE_STAGED': '1', 'RANK': '0', 'WORLD_SIZE': '1', 'LOCAL_RANK': '0'})
    got = metrics(out)
    assert out.count('Epoch') == 2
    assert abs(got[20][0] - plain[20][0]) <= 0.01 and abs(got[20][1] - plain[20][1]) <= 0.01


def test_gpus_flag_spawns_its_ranks(tsvs):
    """`run.py --gpus 2` on a box with one GPU: the launcher starts two ranks, rank 1 finds no second device and exits non-zero, the
    launcher stops rank 0 (which would otherwise wait in RCCL forever) and returns a non-zero code -- no hang."""
    from gru4rec_amd import _native
    if _native.device_count() >= 2:
        pytest.skip('needs a box with ONE GPU (on more, this is the real 2-GPU run)')
    train, test, _ = tsvs
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'run.py'), train, '-ps', PS, '-ss', str(256 * 64), '--gpus', '2'],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert 'rank 1 exited' in r.stderr


def test_bench_gpus_flag_on_a_one_gpu_box_fails_fast_and_loud():
    """`python bench.py --gpus 2` where only one GPU exists: rank 1 refuses (LOCAL_RANK 1, one device), the launcher stops rank 0 and
    returns non-zero -- no JSON line that claims two GPUs, no hang in RCCL."""
    from gru4rec_amd import _native
    if _native.device_count() >= 2:
        pytest.skip('needs a box with ONE GPU')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '20', '--warmup', '5', '--no-cpu-baseline', '--no-micro'],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert 'rank 1' in r.stderr and '"n_gpus": 2' not in r.stdout
