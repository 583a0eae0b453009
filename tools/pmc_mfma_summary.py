"""MFMA-pipe utilisation of the step kernels from a rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
SQ_WAVES) joined with the kernel trace of the same pass (durations):
    python tools/pmc_mfma_summary.py out.json <dir with *counter_collection.csv and *kernel_trace.csv> [cfg name]
Per kernel: mean counters per launch, mean duration, effective clock = GRBM_GUI_ACTIVE / duration (GUI_ACTIVE is reported per XCD
and summed by rocprofv3 when it exceeds any plausible clock: divided by 8 then), MFMA busy fraction = MFMA_BUSY / (4 SIMD x 256 CU x
active cycles), and -- for the scoring GEMMs -- the MFMA issue cycles the algorithm needs (tiles x waves x MFMAs x 64) next to the
counter, which calibrates the counter's unit."""
import collections, csv, glob, json, os, sys
out_path, d = sys.argv[1], sys.argv[2]
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(path)):
        k = r['Kernel_Name'].replace('void ', '').split('(')[0]
        if k.startswith('k_'):
            cnt[k][r['Counter_Name']].append(float(r['Counter_Value']))
dur = collections.defaultdict(list)
for path in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(path)):
        k = r['Kernel_Name'].replace('void ', '').split('(')[0]
        if k.startswith('k_'):
            dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000.0)
out = {}
for k in sorted(cnt):
    c = {n: sum(v) / len(v) for n, v in cnt[k].items()}
    us = sum(dur[k]) / len(dur[k]) if dur.get(k) else None
    e = dict(counters=c, launches=len(next(iter(cnt[k].values()))), avg_us_under_pmc=us)
    gui = c.get('GRBM_GUI_ACTIVE')
    if gui and us:
        ghz = gui / (us * 1e3)
        div = 1
        while ghz / div > 3.0:
            div *= 2
        e['gui_active_divisor'] = div
        e['effective_clock_GHz'] = ghz / div
        act = gui / div
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in c:
            e['mfma_busy_frac_of_4x256_simd_cycles'] = c['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * 256 * act)
        if 'SQ_BUSY_CYCLES' in c:
            e['sq_busy_per_active_cycle'] = c['SQ_BUSY_CYCLES'] / act
    out[k] = e
json.dump(dict(config=sys.argv[3] if len(sys.argv) > 3 else None, note=__doc__, kernels=out), open(out_path, 'w'), indent=1)
for k, e in out.items():
    print('%-22s us %-8s clock %-6s mfma_busy %-7s  %s' % (k, '%.2f' % e['avg_us_under_pmc'] if e['avg_us_under_pmc'] else '-',
          '%.2f' % e.get('effective_clock_GHz', 0), '%.3f' % e.get('mfma_busy_frac_of_4x256_simd_cycles', 0),
          ' '.join('%s=%.4g' % kv for kv in sorted(e['counters'].items()))))
