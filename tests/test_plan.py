"""Host scheduler (C++ `g4r_build_plan`) vs the oracle's literal restatement of the reference loop
(gru4rec.py:594-651): integer work, must be bit-exact, including tail shrinkage and the
`n_valid < 2 and n_sample == 0` stop rule."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from gru4rec_amd import _native
from oracle.scheduler import fit_schedule


def oracle_plan(off, order, items, B, ns):
    ins, outs, rs, Ms, csteps, cmaps = [], [], [], [], [], []
    t = 0
    for ev in fit_schedule(off, order, items, B, ns):
        if ev[0] == 'step':
            _, i, o, M, r = ev
            row_i = np.zeros(B, dtype=np.int32); row_o = np.zeros(B, dtype=np.int32); row_r = np.zeros(B, dtype=np.uint8)
            row_i[:M] = i; row_o[:M] = o; row_r[:M] = r
            ins.append(row_i); outs.append(row_o); rs.append(row_r); Ms.append(M)
            t += 1
        else:
            valid = ev[1]
            mp = np.full(B, -1, dtype=np.int32)
            keep = np.nonzero(valid)[0]
            mp[:len(keep)] = keep
            csteps.append(t); cmaps.append(mp)
    return ins, outs, rs, Ms, csteps, cmaps


def check(lens, B, ns, seed, time_sorted):
    rng = np.random.RandomState(seed)
    lens = np.asarray(lens)
    off = np.zeros(len(lens) + 1, dtype=np.int32)
    off[1:] = np.cumsum(lens)
    items = rng.randint(0, 1000, size=off[-1]).astype(np.int32)
    order = rng.permutation(len(lens)) if not time_sorted else np.arange(len(lens))
    p = _native.build_plan(off, order, items, B, ns)
    ins, outs, rs, Ms, csteps, cmaps = oracle_plan(off, order, items, B, ns)
    assert p['T'] == len(Ms)
    if len(Ms):
        np.testing.assert_array_equal(p['M'], np.array(Ms))
        np.testing.assert_array_equal(p['in_idx'], np.array(ins))
        np.testing.assert_array_equal(p['out_idx'], np.array(outs))
        np.testing.assert_array_equal(p['reset'], np.array(rs))
    assert p['n_compact'] == len(csteps)
    if csteps:
        np.testing.assert_array_equal(p['compact_steps'], np.array(csteps))
        np.testing.assert_array_equal(p['compact_maps'], np.array(cmaps))


@settings(max_examples=150, deadline=None)
@given(st.lists(st.integers(1, 9), min_size=8, max_size=60), st.integers(1, 8), st.sampled_from([0, 4]),
       st.integers(0, 10000), st.booleans())
def test_plan_matches_reference_loop(lens, B, ns, seed, time_sorted):
    check(lens, B, ns, seed, time_sorted)


def test_plan_edge_cases():
    check([2] * 8, 8, 0, 0, True)              # everything finishes at once
    check([1, 1, 1, 5, 1, 2, 1, 1], 2, 4, 1, True)   # length-1 sessions never produce a step
    check([9] + [2] * 20, 3, 0, 2, False)      # one long session keeps its slot, others churn
    check([3, 3, 3], 3, 0, 3, True)            # n_sessions == batch_size


def test_too_few_sessions_is_an_error():
    off = np.array([0, 2, 4], dtype=np.int32)
    with pytest.raises(_native.NativeError):
        _native.build_plan(off, np.arange(2), np.arange(4, dtype=np.int32), 4, 0)


def test_train_plan_builder_reproduces_the_evaluation_schedule():
    """evaluation.py:96-139 is the loop of fit (gru4rec.py:594-651): the C++ plan builder with n_sample = 1 (run until no
    session is left) must emit exactly the steps / zeroed rows / dropped rows of the restated evaluation loop."""
    from oracle.scheduler import eval_schedule
    rng = np.random.RandomState(3)
    for B, n_sess in ((4, 9), (7, 40), (16, 16), (5, 23)):
        lens = rng.randint(1, 9, size=n_sess)
        lens[rng.randint(0, n_sess, size=3)] = 1          # sessions without a target are skipped over by both loops
        offs = np.zeros(n_sess + 1, dtype=np.int32)
        offs[1:] = np.cumsum(lens)
        items = rng.randint(0, 50, size=offs[-1]).astype(np.int32)
        if (offs[1:B + 1] - offs[:B]).min() < 1:
            continue
        plan = _native.build_plan(offs, np.arange(n_sess), items, B, 1)
        t = 0
        nrows = B
        ci = 0
        stepped = False
        for ev in eval_schedule(offs, items, B):
            if ev[0] == 'step':
                _, cur_in, cur_out, M = ev
                assert plan['M'][t] == M == nrows
                np.testing.assert_array_equal(plan['in_idx'][t, :M], cur_in)
                np.testing.assert_array_equal(plan['out_idx'][t, :M], cur_out)
                t += 1
                stepped = True
            else:
                _, zero_mask, valid = ev
                # rows zeroed by the evaluation loop carry the reset flag of the step that ended their session (a slot that
                # was just refilled with a one-event session is zeroed again without a step in between: still zero)
                if stepped:
                    assert (plan['reset'][t - 1, :nrows][zero_mask] == 1).all()
                stepped = False
                if not valid.all():
                    assert plan['compact_steps'][ci] == t
                    keep = np.nonzero(valid)[0]
                    np.testing.assert_array_equal(plan['compact_maps'][ci][:len(keep)], keep)
                    ci += 1
                nrows = int(valid.sum())
        assert t == plan['T'] and ci == plan['n_compact']
