"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950):

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out_f -- python bench.py --steps 300 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-graph
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out_w -- python bench.py ... (same)
  python tools/pmc_summary.py out_f/*/*counter_collection.csv out_w/*/*counter_collection.csv profiles/r02_pmc_traffic_<cfg>.json

Corrections (MI355X_MICROARCH.md, HBM section): the counters are in KiB-like units of 1024 B; on gfx950 FETCH_SIZE
tallies the 128-byte requests of wide (16 B/lane) loads at 64 B, so it is doubled; WRITE_SIZE is taken as reported
(uncalibrated).  Infinity-Cache hits are counted, i.e. this is L2 <-> fabric traffic, an upper bound of DRAM traffic."""
import collections
import csv
import json
import sys


def mean_by_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            acc[r['Kernel_Name'].split('(')[0].replace('void ', '').split('<')[0]].append(float(r['Counter_Value']))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items() if k.startswith('k_') and len(v) >= 5}


def main():
    f = mean_by_kernel(sys.argv[1], 'FETCH_SIZE')
    w = mean_by_kernel(sys.argv[2], 'WRITE_SIZE')
    out = {'units': 'bytes per launch', 'correction': 'FETCH_SIZE x 2 (gfx950 wide loads), WRITE_SIZE as reported; both x 1024',
           'kernels': {}}
    for k in sorted(f):
        fb, wb = 2.0 * f[k][0] * 1024.0, w.get(k, (0.0, 0))[0] * 1024.0
        out['kernels'][k] = {'fetch_bytes': fb, 'write_bytes': wb, 'traffic_bytes': fb + wb, 'launches': f[k][1]}
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
    for k, v in out['kernels'].items():
        print('%-18s fetch %9.0f  write %9.0f  total %9.0f' % (k, v['fetch_bytes'], v['write_bytes'], v['traffic_bytes']))


if __name__ == '__main__':
    main()
