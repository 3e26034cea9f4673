#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection CSV: mean counter value per dispatch, per kernel."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in rows:
    k = (r['Kernel_Name'][:44], int(r['Grid_Size']) // max(1, int(r['Workgroup_Size'])))
    a = agg[k][r['Counter_Name']]
    a[0] += 1
    a[1] += float(r['Counter_Value'])
for k, cs in sorted(agg.items(), key=lambda kv: -sum(v[1] for v in kv[1].values())):
    print(k[0], 'wgs', k[1], 'dispatches', max(v[0] for v in cs.values()))
    for name, (n, tot) in sorted(cs.items()):
        print(f'    {name:28s} {tot / n:16.1f}')
