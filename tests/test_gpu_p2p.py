"""The one-shot all-reduce of the dense GRU gradients through peer memory (k_p2p_allreduce; include/gru4rec_hip.h g4r_p2p_*), the
switch next to the RCCL all-reduce.  RCCL refuses two ranks on one device, the peer-memory path does not care where its peers
live: two PROCESSES on the one MI355X of the test box map each other's exchange regions (hipIpc) and train as ranks 0 / 1 -- a real
N > 1 run of the step graph with stamped cross-process hand-offs -- and must end with exactly the bits of two virtual ranks
(g4r_virtual_train_steps: the same per-rank kernels, the gradients summed in rank order in process)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from gru4rec_amd import _native

from p2p_worker import B, CASE, I, NS, T, rank_model, results
from test_gpu_parity import CASES, make_pair, random_plan

pytestmark = pytest.mark.gpu
WORKER = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'p2p_worker.py')


def spawn(tmp_path, modes, env_extra=None, timeout=150):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', **(env_extra or {}))
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(len(modes)), str(tmp_path), mode], env=env) for r, mode in enumerate(modes)]
    try:
        return [p.wait(timeout=timeout) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()


def test_one_rank_matches_the_rccl_path(monkeypatch):
    """nranks = 1 (staged dense path): the peer-memory kernel with no peers hands the gradient through, as RCCL's one-rank
    all-reduce does -- same losses, same parameters; and it replays from the step graph."""
    monkeypatch.setenv('G4R_FORCE_STAGED', '1')
    outs = []
    for p2p in (0, 1):
        _, m = make_pair(I, B, NS, store_rows=200, use_graph=1, **dict(CASES[CASE]))
        if p2p:
            m.p2p_attach([m.p2p_export()], 1, 0)
            assert m.p2p_active()
        else:
            m.comm_init(_native.comm_unique_id(), 1, 0)
        m.set_plan(random_plan(I, B, T, seed=17))
        m.train_steps(0, T)
        assert m.get_debug('graph_mode', (1,))[0] == 1.0
        outs.append(results(m))
        m.close()
    for k in outs[0]:
        np.testing.assert_array_equal(outs[0][k], outs[1][k], err_msg=k)


def test_two_processes_on_one_device_match_two_virtual_ranks(tmp_path):
    rcs = spawn(tmp_path, ['train', 'train'], dict(G4R_P2P_TIMEOUT_MS='30000'))
    errs = [open(os.path.join(tmp_path, f)).read() for f in sorted(os.listdir(tmp_path)) if f.startswith('error')]
    assert rcs == [0, 0] and not errs, (rcs, errs)
    ms = [rank_model(q, 2) for q in range(2)]
    _native.virtual_train_steps(ms, 0, T)
    want = [results(m) for m in ms]
    for m in ms:
        m.close()
    for q in range(2):
        got = np.load(os.path.join(tmp_path, 'out%d.npz' % q))
        assert got['graph_mode'][0] == 1.0, 'the peer-memory all-reduce was not captured into the step graph'
        for k in want[q]:
            np.testing.assert_array_equal(got[k], want[q][k], err_msg='rank %d %s' % (q, k))
    # both ranks applied the same summed gradients: identical dense parameters; their item tables are their own
    np.testing.assert_array_equal(want[0]['Wh'], want[1]['Wh'])
    assert not np.array_equal(want[0]['Wy'], want[1]['Wy'])


def test_a_dead_peer_is_an_error_not_a_hang(tmp_path):
    rcs = spawn(tmp_path, ['train', 'idle'], dict(G4R_P2P_TIMEOUT_MS='300'), timeout=120)
    assert rcs == [0, 0]
    err = open(os.path.join(tmp_path, 'error0')).read()
    assert 'p2p all-reduce' in err and not os.path.exists(os.path.join(tmp_path, 'out0.npz'))
