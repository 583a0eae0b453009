"""bench.py's N > 1 code path, driven on the CPU with a stand-in for the device model.

No multi-GPU box has ever run `bench.py --gpus N`: the driver's SCALE run would be the first.  Everything the RCCL ranks need below
`bench.main()` -- the model handle and the collectives on it -- is replaced by a fake that answers as rank 0 of a world of two (every
collective returns this rank's own value), so that the Python of the path is executed: plan per rank, timed region with the
on-stream reconciliation, per-kernel pass, the reconciliation measurement, `value` / `value_step_only` / `value_source`, the
`multi_gpu` object, the ONE JSON line on stdout.  What it cannot check is RCCL itself."""
import json
import sys

import numpy as np
import pytest

import bench
from gru4rec_amd import _native, launch


class FakeModel:
    """The methods bench.py calls on _native.Model, with plausible answers and no device."""
    instances = []

    def __init__(self, **kw):
        self.kw = kw
        self.sync_k = 0
        self.calls = []
        self.profile_mode = 0
        FakeModel.instances.append(self)

    def comm_init(self, uid, nranks, rank): self.comm = (uid, nranks, rank)
    def comm_nranks(self): return self.kw['nranks']
    def set_param(self, *a, **k): pass
    def set_popularity(self, *a, **k): pass
    def set_plan(self, plan): self.T = int(plan['T'])
    def reset_hidden(self): pass

    def train_steps(self, t0, n):
        assert 0 <= t0 and t0 + n <= self.T, 'step range outside the plan'
        self.calls.append((t0, n, self.sync_k, self.profile_mode))

    def get_losses(self, t0, n): return np.full(n, 0.7, dtype=np.float32)
    def profile(self, mode): self.profile_mode = int(mode)

    def kernel_times(self):
        names = ('k_gru_fwd', 'k_score_fwd', 'k_loss_rows', 'k_score_bwd', 'k_gru_bwd', 'k_update', 'rccl_allreduce', 'k_dense_apply')
        return {n: (0.005 * 300, 300) for n in names}

    def set_sync_every(self, k):
        self.sync_k = int(k)
        return True      # small item tables: the library reconciles on the stream

    def comm_sync_sparse(self): self.calls.append(('sync',))
    def comm_max(self, v): return int(v)
    def p2p_active(self): return False

    def get_debug(self, name, shape):
        return np.array([{'dense_count': 60300.0, 'graph_mode': 1.0, 'dev_syncs': 3.0}.get(name, 0.0)], dtype=np.float32)

    def close(self): pass


@pytest.mark.parametrize('extra', [[], ['--sparse-exact']], ids=['gpu_local_rows', 'exact_replicas'])
def test_bench_main_as_rank_0_of_2(monkeypatch, capsys, extra):
    FakeModel.instances.clear()
    monkeypatch.setenv('RANK', '0')
    monkeypatch.setenv('LOCAL_RANK', '0')
    monkeypatch.setenv('WORLD_SIZE', '2')
    monkeypatch.delenv('G4R_FORCE_STAGED', raising=False)
    monkeypatch.setattr(_native, 'device_count', lambda: 2)
    monkeypatch.setattr(_native, 'Model', FakeModel)
    monkeypatch.setattr(launch, 'unique_id', lambda rank, world, **kw: b'\0' * 128)
    monkeypatch.setattr(launch, 'cleanup', lambda rank: None)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '2', '--steps', '20', '--warmup', '5', '--no-cpu-baseline', '--no-micro',
                                      '--profile-steps', '16'] + extra)
    bench.main()
    lines = [l for l in capsys.readouterr().out.splitlines() if l.strip()]
    assert len(lines) == 1, 'rank 0 prints ONE JSON line'
    out = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config'):
        assert k in out
    assert out['n_gpus'] == 2 and out['steps'] == 20 and out['warmup'] == 5 and out['scaling'] == 'weak' and out['unit'] == 'mini-batches/s'
    assert out['config']['global_batch'] == 2 * bench.CONFIGS['cfg2']['batch_size']
    assert out['value'] > 0 and np.isfinite(out['value'])
    assert 'multi_gpu' in out and out['multi_gpu']['ncclCommCount'] == 2 and len(out['multi_gpu']['rank_ms_per_step']) == 2
    m = FakeModel.instances[0]
    assert m.kw['rank'] == 0 and m.kw['nranks'] == 2
    timed = [c for c in m.calls if c[0] == 5 and c[1] == 20]
    assert len(timed) == 1, 'the timed region is ONE call over exactly --steps plan steps behind the warmup'
    if extra:
        assert m.kw['sparse_exact'] == 3 and timed[0][2] == 0 and 'value_source' not in out      # nothing to reconcile in this mode
        assert out['config']['item_rows'].startswith('exact replicas')
    else:
        # GPU-local rows: the timed region runs WITH the reconciliation every sync_every steps (4 at two ranks), the bare step is reported next to it
        assert m.kw['sparse_exact'] == 0 and timed[0][2] == 4
        assert 'reconciles the GPU-local item tables every 4 steps' in out['value_source']
        assert out['value_step_only'] > 0
        assert out['multi_gpu']['reconciliation']['timed_region_includes_reconciliation'] is True
        assert all(c[2] == 0 for c in m.calls if len(c) == 4 and c[3] != 0), 'the per-kernel passes time the bare step'
