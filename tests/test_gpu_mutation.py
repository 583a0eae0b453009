"""The parity suite must be able to FAIL.  Three deliberately wrong builds of the library (gru4rec_amd/build.py: build_mutants,
-DG4R_MUTATE=k in g4r_device.cuh) are run through a handful of the parity tests in a child process with G4R_LIB pointing at the
mutant; every one of those runs has to come back red, and the tensor it names has to be the one the mutation damages:

  mutant 1  every per-occurrence sparse accumulator increment x 1.01   -> acc_Wy / acc_By (1 % of an accumulator of ~1e-7)
  mutant 2  every sparse Adagrad step x 1.01                             -> dWy / dBy (1 % of a step)
  mutant 3  every dense accumulator increment x 1.01                     -> acc_Wx / acc_Wh / ...
  mutant 5  the 1 / nranks factor of the exact-replica joint update x 1.01 -> the REDUCE / MEAN oracle-as-replicas tests
  mutant 6  the Adagrad step of ONE item row per step x 1.5 (the item of score column 0) -> dWy: a single wrong row must fail, also in
            the exact-shape tests that compare the rows of kink items apart (round 5 left those rows out; now they are bounded)

(Round 2's `atol = 1e-4` on every tensor let an accumulator that is wrong by 100 x pass.)  The same selection runs green on the
product library in the ordinary suite."""
import os
import subprocess
import sys

import pytest

from gru4rec_amd import build as g4r_build

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SELECTION = ['tests/test_gpu_parity.py::test_first_step_intermediates[bprmax_elu]',
             'tests/test_gpu_baseline_configs.py::test_cfg4_exact_shape',
             'tests/test_gpu_parity.py::test_baseline_config2_shape_few_steps',
             'tests/test_gpu_golden.py::test_product_reproduces_reference_run[bprmax_constrained]']
# one step from zero accumulators: what must fail and what must still pass (p1 = the first-step test's tag)
FIRST_STEP = {1: (('p1 acc_Wy', 'p1 acc_By'), ('p1 acc_Wx0', 'p1 acc_Wh0', 'p1 dWx0')),
              2: (('p1 dWy', 'p1 dBy'), ('p1 acc_Wy', 'p1 acc_By', 'p1 acc_Wx0', 'p1 dWx0')),
              3: (('p1 acc_Wx0', 'p1 acc_Wh0', 'p1 acc_Wrz0', 'p1 acc_Bh0'), ('p1 acc_Wy', 'p1 acc_By', 'p1 dWy')),
              6: (('p1 dWy',), ('p1 acc_Wy', 'p1 acc_By', 'p1 acc_Wx0', 'p1 dWx0'))}


@pytest.fixture(scope='module')
def mutants():
    paths = g4r_build.build_mutants()
    assert all(os.path.exists(p) for p in paths)
    return dict(zip(sorted(g4r_build.MUTANTS), paths))


@pytest.mark.parametrize('k', sorted(FIRST_STEP))      # (mutant 4, the stale-register pipeline, has its own test: test_gpu_stress.py)
def test_mutant_turns_the_parity_tests_red(mutants, k, tmp_path):
    for i, sel in enumerate(SELECTION):
        rep = str(tmp_path / ('report%d.txt' % i))
        env = dict(os.environ, G4R_LIB=mutants[k], G4R_PARITY_REPORT=rep)
        r = subprocess.run([sys.executable, '-m', 'pytest', sel, '-x', '-q', '-p', 'no:cacheprovider'], cwd=ROOT, env=env,
                           capture_output=True, text=True, timeout=900)
        out = r.stdout + r.stderr
        assert r.returncode == 1, 'mutant %d (%s) passed %s:\n%s' % (k, g4r_build.MUTANTS[k], sel, out[-3000:])
        if i == 0:
            # the per-tensor report of the child: the damaged tensors are red, the others still green
            lines = open(rep).read().splitlines()
            state = {ln[:28].strip(): ln.rstrip().endswith('FAIL') for ln in lines if 'worst/tol' in ln}
            must_fail, must_pass = FIRST_STEP[k]
            assert all(state[n] for n in must_fail), (k, {n: state[n] for n in must_fail})
            assert not any(state[n] for n in must_pass), (k, {n: state[n] for n in must_pass})


EXACT_SELECTION = ['tests/test_gpu_exact_replicas.py::test_reduce_form_against_the_oracle_run_as_replicas[bprmax_constrained]',
                   'tests/test_gpu_exact_replicas.py::test_mean_form_against_the_oracle_run_as_replicas[bprmax_constrained]']


def test_mutant_5_turns_the_exact_replica_parity_red(mutants, tmp_path):
    """The 1 / nranks factor of the joint update x 1.01 (REDUCE: the gradient scale; MEAN: the mean over the touching ranks): the
    oracle-as-replicas tests of the forms that carry it must fail, the SUM form (no such factor) must still pass."""
    for i, (sel, want_rc) in enumerate([(s_, 1) for s_ in EXACT_SELECTION] +
                                       [('tests/test_gpu_exact_replicas.py::test_exact_mode_against_the_oracle_run_as_replicas[bprmax_constrained]', 0)]):
        env = dict(os.environ, G4R_LIB=mutants[5], G4R_PARITY_REPORT=str(tmp_path / ('x%d.txt' % i)))
        r = subprocess.run([sys.executable, '-m', 'pytest', sel, '-x', '-q', '-p', 'no:cacheprovider'], cwd=ROOT, env=env,
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == want_rc, 'mutant 5 on %s: rc %d\n%s' % (sel, r.returncode, (r.stdout + r.stderr)[-3000:])


def test_product_library_is_not_a_mutant():
    from gru4rec_amd import _native
    assert 'G4R_LIB' not in os.environ or '_variants' not in os.environ['G4R_LIB']
    assert '_variants' not in _native.LIB_PATH
