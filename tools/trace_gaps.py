"""Analyse a rocprofv3 kernel_trace.csv: per-kernel duration and the idle gap before each kernel (graph replay)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
prev_end = None
for r in rows:
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    dur[name].append(e - s)
    if prev_end is not None:
        gap[name].append(s - prev_end)
    prev_end = e
tot = 0
for k in dur:
    if len(dur[k]) < 50: continue
    d = sorted(dur[k]); g = sorted(gap[k])
    md, mg = d[len(d) // 2], g[len(g) // 2]
    tot += md + mg
    print('%-24s n=%5d  dur median %7.2f us  (p10 %6.2f p90 %6.2f)   gap-before median %6.2f us' % (k, len(d), md / 1e3, d[len(d) // 10] / 1e3, d[9 * len(d) // 10] / 1e3, mg / 1e3))
print('sum of medians (dur+gap): %.1f us' % (tot / 1e3))
