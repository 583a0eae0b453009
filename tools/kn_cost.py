"""What each launch costs the captured step: the step timed with that launch left out (G4R_SKIP_KN, g4r_host_step.hpp).
   python tools/kn_cost.py [cfg2]   (runs bench.py in child processes, one per kernel slot; results of the skipped runs are garbage by design)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
KN = ["k_gru_p1", "k_gru_p2", "k_score_fwd", "k_loss_rows", "k_score_bwd", "k_gru_bwd_pre", "k_gru_bwd_a", "k_gru_bwd_b", "k_dense_grad", "rccl_allreduce",
      "k_dense_apply", "k_sparse_update", "k_update", "k_gru_bwd", "k_gru_fwd", "k_gru_gate", "k_sparse_flush", "k_defer_scan", "k_finish_rows",
      "k_gru_v", "k_gru_h", "k_gru_da", "k_gru_dy"]
def run(mask):
    env = dict(os.environ)
    if mask: env['G4R_SKIP_KN'] = hex(mask)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--config', cfg, '--steps', '3000', '--warmup', '300', '--no-cpu-baseline', '--no-micro',
                          '--profile-steps', '0', '--long-steps', '0'], env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith('{')]
    if not line:
        return None
    return json.loads(line[-1])['ms_per_step'] * 1000.0
base = run(0)
print('%s full step: %.2f us' % (cfg, base))
used = sys.argv[2].split(',') if len(sys.argv) > 2 else ['k_gru_v', 'k_gru_h', 'k_score_fwd', 'k_loss_rows', 'k_score_bwd', 'k_gru_da', 'k_gru_dy', 'k_update']
tot = 0.0
for name in used:
    t = run(1 << KN.index(name))
    if t is None:
        print('  %-16s failed' % name); continue
    print('  without %-16s %.2f us  -> costs %.2f us' % (name, t, base - t)); tot += base - t
print('  sum of the launches\' costs %.2f us of %.2f' % (tot, base))
