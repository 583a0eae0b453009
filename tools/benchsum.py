"""Print a one-screen summary of a bench.py JSON line (file argument or gpurun_out/bench.log)."""
import json
import sys
path = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/bench.log'
for line in open(path):
    if line.startswith('{'):
        d = json.loads(line)
        print('value %.0f %s | %.1f us/step | kernel sum %.1f us' % (d['value'], d['unit'], 1000 * d['ms_per_step'],
                                                                   d.get('kernel_time_sum_us_per_step', 0)))
        print({k: round(v['avg_us'], 1) for k, v in d.get('kernels', {}).items()})
        if 'cpu_baseline' in d:
            print('cpu', d['cpu_baseline'])
