"""bench.py's workload generator (host only): every rank of an N-GPU run must get a plan that is long enough and full-batch over
the measured range -- a rank that asserts here would leave the others waiting in the all-reduce."""
import numpy as np
import pytest

import bench


@pytest.mark.parametrize('world', [1, 2, 4, 8])
@pytest.mark.parametrize('config', ['cfg2', 'cfg1'])
def test_every_rank_gets_a_full_plan(config, world):
    cfg = bench.CONFIGS[config]
    total = 260
    seen = []
    for rank in range(world):
        plan, support = bench.make_plan(cfg, total, rank, world)
        assert plan['T'] >= total
        assert (plan['M'][:total] == cfg['batch_size']).all()
        assert plan['in_idx'].shape[1] == cfg['batch_size'] and plan['in_idx'].max() < cfg['n_items']
        assert support.shape == (cfg['n_items'],) and support.min() >= 1.0
        seen.append(plan['in_idx'][:total].copy())
    if world > 1:      # ranks train on different session shards
        assert not np.array_equal(seen[0], seen[1])


def test_algorithmic_cost_covers_every_kernel_name():
    """bench.py prices the kernels the library reports by name (g4r_kernel_time)."""
    cost = bench.algorithmic_cost(bench.CONFIGS['cfg2'])
    for name in ('k_gru_fwd', 'k_gru_bwd', 'k_gru_p1', 'k_gru_p2', 'k_score_fwd', 'k_loss_rows', 'k_score_bwd', 'k_gru_bwd_pre',
                 'k_gru_bwd_a', 'k_gru_bwd_b', 'k_dense_grad', 'k_sparse_update', 'k_update'):
        assert name in cost and cost[name]['bound'] in bench.PEAK
    assert cost['k_update']['bytes'] == 5927936      # the figure DESIGN.md section 6 quotes for cfg #2
