"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (dev tool)."""
import csv, collections, sys
rows = csv.DictReader(open(sys.argv[1]))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r['Kernel_Name']
    if 'usf::' not in k: continue
    k = k[k.index('usf::'):][:48]
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    n = len(next(iter(d.values())))
    print(k, f"(n={n})")
    for c, v in sorted(d.items()):
        print(f"   {c:32s} {sum(v)/len(v):16.1f}")
