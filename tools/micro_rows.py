"""Row gather / scatter micro-benchmark sweep (g4r_bench_rows): tables far beyond the 256 MiB Infinity Cache, the row counts of
the BASELINE configs (R = 2B + n_sample gathered rows per step) and batched multiples of them.

    python tools/micro_rows.py [out.json]

Algorithmic bytes per launch: gather = rows * W * 4 (+ 4 per index); mode 0 also writes the compact copy (reported as
2x); Adagrad scatter = 5 * rows * W * 4.  GB/s = bytes / mean kernel duration; frac = GB/s / 8000 (HBM3E peak)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gru4rec_amd import _native

PEAK = 8000.0
MODES = {0: ('gather_copy', 2), 1: ('gather_fused', 1), 2: ('adagrad_scatter', 5)}


def sweep(quick=False):
    out = []
    shapes = [(6_500_000, 256, 9216, 'cfg4 (10M-item class table, D=256, R=9216)'), (3_000_000, 512, 2528, 'cfg3 (3M items, D=512, R=2528)')]
    mults = (1, 16) if quick else (1, 4, 16, 64)
    for n_items, W, R, label in shapes:
        for mode, (name, streams) in MODES.items():
            for mult in mults:
                rows = R * mult
                launches = 200 if mult <= 4 else 50
                k_us, w_us = _native.bench_rows(n_items, W, rows, launches=launches, mode=mode)
                nbytes = streams * rows * W * 4 + rows * 4
                out.append(dict(shape=label, table_gb=n_items * W * 4 / 1e9 * (2 if mode == 2 else 1), width=W, rows_per_launch=rows,
                                steps_batched=mult, mode=name, bytes_per_launch=nbytes, kernel_us=k_us, wall_us_per_launch=w_us,
                                gbps=nbytes / k_us / 1e3, frac_of_8TBps=nbytes / k_us / 1e3 / PEAK))
    return out


if __name__ == '__main__':
    res = sweep()
    for r in res:
        print('%-44s %-16s rows %7d  kernel %8.2f us  wall %8.2f us  %7.0f GB/s  %.3f of peak' % (
            r['shape'], r['mode'], r['rows_per_launch'], r['kernel_us'], r['wall_us_per_launch'], r['gbps'], r['frac_of_8TBps']))
    if len(sys.argv) > 1:
        json.dump({'peak_GBps': PEAK, 'results': res}, open(sys.argv[1], 'w'), indent=1)
