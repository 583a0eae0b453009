"""Narrow layers run the lean launches of g4r_lean_kernels.cuh (k_gru_v / k_gru_h / k_gru_da / k_gru_dy, k_score_s / k_score_b) by default; the
single-launch kernels they replaced (k_gru_fwd_fused / k_gru_bwd_fused, k_score_fwd / k_score_bwd on LDS-staged tiles) stay in the library
behind G4R_NO_LEAN=1 (A/B runs) and must keep passing the same oracle parity: this file runs a selection of the parity suite in a child
process with that switch (the library reads it once per process), and checks that the default process really took the lean path."""
import os
import subprocess
import sys

import numpy as np
import pytest

from gru4rec_amd import _native

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SELECTION = ['tests/test_gpu_parity.py::test_first_step_intermediates',
             'tests/test_gpu_parity.py::test_baseline_config2_shape_few_steps',
             'tests/test_gpu_golden.py',
             'tests/test_gpu_baseline_configs.py::test_cfg5_exact_shape']


def test_the_replaced_kernels_still_pass_the_parity_suite():
    env = dict(os.environ, G4R_NO_LEAN='1')
    r = subprocess.run([sys.executable, '-m', 'pytest'] + SELECTION + ['-x', '-q', '-p', 'no:cacheprovider'], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]


def _kernel_names(extra_env):
    code = ("import numpy as np, json; from gru4rec_amd import _native\n"
            "rng = np.random.RandomState(0); I, B, ns, T = 300, 32, 64, 6\n"
            "m = _native.Model(n_items=I, layers=[32], batch_size=B, n_sample=ns, loss=1, final_act=4, final_act_p0=0.5, hidden_act=2, embed_mode=0,\n"
            "                  learning_rate=0.1, momentum=0.0, bpreg=1.0, sample_alpha=0.75, sample_store=ns * 10, seed=9, device=0, rank=0, nranks=1, use_graph=0)\n"
            "m.set_param('Wy', (rng.rand(I, 32) * 0.2 - 0.1).astype(np.float32)); m.set_popularity(np.cumsum(np.ones(I)) / I)\n"
            "plan = dict(in_idx=rng.randint(0, I, size=(T, B)).astype(np.int32), out_idx=rng.randint(0, I, size=(T, B)).astype(np.int32),\n"
            "            reset=np.zeros((T, B), dtype=np.uint8), M=np.full(T, B, dtype=np.int32), n_compact=0, compact_steps=None, compact_maps=None)\n"
            "m.set_plan(plan); m.reset_hidden(); m.profile(True); m.train_steps(0, T); m.profile(False)\n"
            "print('KN', json.dumps(sorted(m.kernel_times())))\n")
    r = subprocess.run([sys.executable, '-c', code], cwd=ROOT, env=dict(os.environ, **extra_env), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    import json
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('KN ')][-1][3:])


def test_default_takes_the_lean_launches_and_the_switch_the_fused_ones():
    lean = _kernel_names({})
    assert {'k_gru_v', 'k_gru_h', 'k_gru_da', 'k_gru_dy'} <= set(lean) and 'k_gru_fwd' not in lean, lean
    fused = _kernel_names({'G4R_NO_LEAN': '1'})
    assert {'k_gru_fwd', 'k_gru_bwd'} <= set(fused) and 'k_gru_v' not in fused, fused
