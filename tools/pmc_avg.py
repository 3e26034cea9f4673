"""Mean counter values per dispatch of the kernels whose name contains argv[2] (rocprofv3 counter_collection.csv)."""
import collections
import csv
import sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r['Kernel_Name']]
acc = collections.defaultdict(list)
for r in rows:
    acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    v = v[2:] if len(v) > 4 else v                       # drop the warm-ups
    print(f'{k:40s} {sum(v) / len(v):18.1f}  (n={len(v)})')
