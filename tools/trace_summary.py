#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel-trace CSV per (kernel, grid): calls, average duration, share."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = (r['Kernel_Name'][:46], int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), int(r['Grid_Size_Y']),
         r['LDS_Block_Size'], r['VGPR_Count'], r.get('Accum_VGPR_Count', ''))
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    a = agg.setdefault(k, [0, 0])
    a[0] += 1
    a[1] += d
tot = sum(a[1] for a in agg.values())
print(f'total kernel time {tot / 1e6:.2f} ms')
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'{k[0]:48s} wgs=({k[1]},{k[2]}) lds={k[3]:>6} vgpr={k[4]}+{k[5]} calls={a[0]:5d} avg_us={a[1] / a[0] / 1e3:9.1f} '
          f'share={a[1] / tot * 100:5.1f}%')
