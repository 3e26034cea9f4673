#!/usr/bin/env python3
"""Per-LAYER durations from a rocprofv3 kernel-trace CSV of bench.py run with FNN_NO_PIPELINE=1 (one stream).

The launches are sorted by start time and cut into forwards at every stem kernel; the forwards of the most common
length are laid over each other and the median duration per position is printed - one line per launch of a forward of
32 patches, in network order.
"""
import collections
import csv
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
fw, cur = [], None
for r in rows:
    name = r['Kernel_Name']
    if 'stem_mfma' in name or 'stem_row_kernel' in name:
        if cur:
            fw.append(cur)
        cur = []
    if cur is not None:
        if 'gather_head' in name or 'rocclr' in name:
            continue
        cur.append(r)
if cur:
    fw.append(cur)


def key(r):
    return (r['Kernel_Name'][:52], int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), int(r['Grid_Size_Y']))


sig = collections.Counter(tuple(key(r) for r in f) for f in fw)
best = sig.most_common(1)[0][0]
sel = [f for f in fw if tuple(key(r) for r in f) == best]
print(f'{len(fw)} forwards, {len(sel)} with the common launch sequence of {len(best)} launches')
tot = 0.0
for i, k in enumerate(best):
    if 'stats_finalize' in k[0]:
        continue
    d = statistics.median(int(f[i]['End_Timestamp']) - int(f[i]['Start_Timestamp']) for f in sel) / 1e3
    tot += d
    print(f'{i:3d} {k[0]:54s} wgs=({k[1]},{k[2]}) {d:9.1f} us')
print(f'sum of medians {tot / 1e3:.3f} ms per forward')
