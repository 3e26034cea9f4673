#!/usr/bin/env python3
"""Per (kernel, grid): mean duration, effective shader clock (GRBM_GUI_ACTIVE / 8 XCDs / duration) and the share of
those cycles in which the matrix pipes were busy (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs), from ONE rocprofv3 run with
--kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES ... (kernel_trace.csv, counter_collection.csv)."""
import collections
import csv
import sys

dur, key = {}, {}
for r in csv.DictReader(open(sys.argv[1])):
    dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    key[r['Dispatch_Id']] = (r['Kernel_Name'][:46], int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), int(r['Grid_Size_Y']))
vals = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[2])):
    vals[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for i, d in dur.items():
    v = vals.get(i, {})
    if 'GRBM_GUI_ACTIVE' not in v:
        continue
    a = agg[key[i]]
    a[0] += 1; a[1] += d; a[2] += v['GRBM_GUI_ACTIVE'] / 8; a[3] += v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / 1024
tot = sum(a[1] for a in agg.values())
print(f'{"kernel":48s} {"grid":>12s} {"calls":>6s} {"avg us":>9s} {"share":>6s} {"GHz":>6s} {"mfma busy":>9s}')
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if a[1] / tot < 0.002:
        continue
    print(f'{k[0]:48s} {f"({k[1]},{k[2]})":>12s} {a[0]:6d} {a[1] / a[0]:9.1f} {a[1] / tot * 100:5.1f}% {a[2] / a[1] / 1e3:6.3f} {a[3] / max(a[2], 1):9.3f}')
