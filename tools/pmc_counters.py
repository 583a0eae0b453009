"""Mean per-launch value of every collected counter, per kernel, from rocprofv3 --pmc counter_collection CSVs.
   python tools/pmc_counters.py out.json a/*counter_collection.csv b/*counter_collection.csv ..."""
import collections, csv, json, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        k = r['Kernel_Name'].replace('void ', '').split('(')[0]
        if k.startswith('k_'):
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
json.dump(out, open(sys.argv[1], 'w'), indent=1)
for k in sorted(out):
    print(k, ' '.join('%s=%.4g' % (c, v) for c, v in sorted(out[k].items())))
