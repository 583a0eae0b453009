"""The hand-off protocol of the peer-memory all-reduce (k_p2p_allreduce, g4r_sync_kernels.cuh) as a model that a scheduler can
interleave any way it likes: every rank, for step s, (1) writes its slice into its own buffer s & 1, (2) publishes stamp s + 1,
(3) waits until every peer's stamp is >= s + 1, (4) reads every peer's buffer s & 1.  Two buffers are enough only because a rank
can start step s + 2 -- the next writer of buffer s & 1 -- after every peer has published s + 2, which a peer does after its reads
of step s.  The model runs N ranks under exhaustive (small) and random (larger) schedules and checks that every read returns the
value written for exactly that step; a one-buffer variant and a variant that publishes before it writes must fail, so the check
is known to be able to."""
import itertools
import random

import pytest


def run(n, steps, pick, nbuf=2, publish_first=False):
    """pick(runnable ranks) -> the rank that executes its next micro-operation.  Returns None or a description of the first bad read."""
    data = [[None] * nbuf for _ in range(n)]
    flag = [0] * n
    pc = [(0, 0, 0)] * n          # (step, phase, peer cursor)
    done = [False] * n
    while not all(done):
        runnable = []
        for r in range(n):
            if done[r]:
                continue
            s, ph, q = pc[r]
            if ph == 2:           # waiting for peer q's stamp
                while q < n and (q == r or flag[q] >= s + 1):
                    q += 1
                pc[r] = (s, ph, q)
                if q < n:
                    continue      # blocked
            runnable.append(r)
        if not runnable:
            return 'deadlock at %s' % (pc,)
        r = pick(runnable)
        s, ph, q = pc[r]
        write_phase = 1 if publish_first else 0      # the faulty variant publishes its stamp before it writes its data
        if ph in (0, 1):
            if ph == write_phase:
                data[r][s % nbuf] = (r, s)
            else:
                flag[r] = s + 1
            pc[r] = (s, ph + 1, 0)
        elif ph == 2:             # all stamps seen (cursor ran past the last peer): move on to the reads
            pc[r] = (s, 3, 0)
        elif ph == 3:             # one peer read per micro-operation
            if q == r:
                q += 1
            if q < n:
                if data[q][s % nbuf] != (q, s):
                    return 'rank %d step %d read %s from rank %d' % (r, s, data[q][s % nbuf], q)
                q += 1
            if q >= n or (q == r and q + 1 >= n):
                s += 1
                pc[r] = (s, 0, 0)
                if s == steps:
                    done[r] = True
            else:
                pc[r] = (s, 3, q)
    return None


def test_every_schedule_of_two_ranks_reads_the_right_step():
    # exhaustive over schedules encoded as a choice sequence: 2 ranks x 3 steps is ~30 micro-operations, branch only where both can run
    bad = []

    def explore(prefix):
        choices = iter(prefix)
        forks = []

        def pick(runnable):
            if len(runnable) == 1:
                return runnable[0]
            try:
                return runnable[next(choices)]
            except StopIteration:
                forks.append(len(runnable))
                return runnable[0]
        res = run(2, 3, pick)
        if res:
            bad.append((prefix, res))
        return len(forks)
    frontier, seen = [()], 0
    while frontier and seen < 20000:
        prefix = frontier.pop()
        seen += 1
        extra = explore(prefix)
        if extra:                     # extend the prefix at the first undecided fork
            frontier.append(prefix + (0,))
            frontier.append(prefix + (1,))
    assert not bad, bad[:3]
    assert seen > 1000


@pytest.mark.parametrize('n,steps', [(3, 6), (8, 5)])
def test_random_schedules(n, steps):
    for seed in range(300):
        rng = random.Random(seed)
        # biased schedulers too: one rank far ahead, one rank starved
        fav = rng.randrange(n)
        mode = seed % 3
        pick = (lambda rs: rng.choice(rs)) if mode == 0 else (lambda rs: fav if fav in rs and rng.random() < 0.9 else rng.choice(rs)) if mode == 1 else \
               (lambda rs: rng.choice([r for r in rs if r != fav] or rs))
        assert run(n, steps, pick) is None


def test_the_model_can_fail():
    # one buffer: a fast rank overwrites what a slow peer has not read yet
    assert any(run(3, 6, random.Random(seed).choice, nbuf=1) for seed in range(200))
    # stamp before data: a peer reads the previous contents
    assert any(run(3, 6, random.Random(seed).choice, publish_first=True) for seed in range(200))
