"""One rank of tests/test_gpu_p2p.py: a process of its own that owns a model handle, maps the other ranks' exchange regions
(g4r_p2p_export / g4r_p2p_attach, handles carried through files) and trains its plan with the peer-memory all-reduce.
usage: p2p_worker.py <rank> <nranks> <directory> <train | idle>"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from test_gpu_parity import CASES, make_pair, random_plan      # noqa: E402

I, B, NS, T = 80, 12, 24, 70
CASE = 'bprmax_mom_drop'


def rank_model(rank, nranks):
    _, m = make_pair(I, B, NS, store_rows=200, use_graph=1, rank=rank, nranks=nranks, **dict(CASES[CASE]))
    m.set_plan(random_plan(I, B, T, seed=17 + rank))
    return m


def results(m):
    return dict(losses=m.get_losses(0, T), Wy=m.get_param('Wy', (I, 16)), By=m.get_param('By', (I,)), Wx=m.get_param('Wx', (16, 48), 0),
                Wh=m.get_param('Wh', (16, 16), 0), Bh=m.get_param('Bh', (48,), 0), acc_Wh=m.get_param('acc_Wh', (16, 16), 0),
                acc_Wy=m.get_param('acc_Wy', (I, 16)))


def publish(path, blob):
    with open(path + '.tmp', 'wb') as f:
        f.write(blob)
    os.replace(path + '.tmp', path)


def collect(paths, size, timeout=60.0):
    t0, out = time.time(), []
    for p in paths:
        while True:
            try:
                with open(p, 'rb') as f:
                    blob = f.read()
                if len(blob) == size:
                    out.append(blob)
                    break
            except OSError:
                pass
            if time.time() - t0 > timeout:
                raise RuntimeError('no %s after %.0f s' % (p, timeout))
            time.sleep(0.01)
    return out


def main():
    rank, nranks, d, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    m = rank_model(rank, nranks)
    publish(os.path.join(d, 'handle%d' % rank), m.p2p_export())
    m.p2p_attach(collect([os.path.join(d, 'handle%d' % q) for q in range(nranks)], 64), nranks, rank)
    assert m.p2p_active()
    publish(os.path.join(d, 'ready%d' % rank), b'1')
    collect([os.path.join(d, 'ready%d' % q) for q in range(nranks)], 1)
    if mode == 'idle':
        # a rank that never steps: its peers must give up (G4R_P2P_TIMEOUT_MS), not hang; it keeps its region mapped until they did
        collect([os.path.join(d, 'done%d' % q) for q in range(nranks) if q != rank], 1)
        m.close()
        return
    try:
        m.train_steps(0, T)
        np.savez(os.path.join(d, 'out%d.npz' % rank), graph_mode=m.get_debug('graph_mode', (1,)), **results(m))
    except Exception as e:      # the parent asserts on what happened
        publish(os.path.join(d, 'error%d' % rank), str(e).encode() or b'?')
    publish(os.path.join(d, 'done%d' % rank), b'1')
    m.close()


if __name__ == '__main__':
    main()
