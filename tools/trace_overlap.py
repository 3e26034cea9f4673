#!/usr/bin/env python3
"""What runs beside what when several batches are in flight: a rocprofv3 kernel-trace CSV of a plain bench.py run.
usage: python tools/trace_overlap.py <kernel_trace.csv> [span_ms_from_end]
Takes the last `span` ms of the trace (default 400: two timed steps), and prints (1) the share of that span with 0 / 1 / 2 / 3+
kernels running, (2) per kernel family the time it runs alone and beside each other family, (3) per kernel the mean duration
here against the shortest quartile's mean (what it takes when nothing shares the chip with it)."""
import csv
import sys
from collections import defaultdict


def family(name):
    for key, fam in (('conv_row_stem', 'level0'), ('conv_row_kernel', 'level0'), ('stem_row', 'level0'), ('conv3d_zsw', 'level0'),
                     ('conv3d_zr_kernel', 'zr'), ('conv3d_zq12', 'mid'), ('conv3d_s2', 'mid'), ('conv3d_lds', 'mid'),
                     ('tconv', 'tconv'), ('gather', 'gather'), ('stats_finalize', 'stats')):
        if key in name:
            return fam
    return 'other'


rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1]))]
span = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 400e6
t_end = max(e for _, e, _ in rows)
rows = [r for r in rows if r[0] >= t_end - span]
ev = []
for i, (s, e, _) in enumerate(rows):
    ev.append((s, 1, i))
    ev.append((e, 0, i))
ev.sort()
live, last = set(), ev[0][0]
depth = defaultdict(int)
beside = defaultdict(int)
for t, kind, i in ev:
    dt = t - last
    if dt > 0:
        depth[min(len(live), 3)] += dt
        fams = sorted({family(rows[j][2]) for j in live})
        if len(live) == 1:
            beside[(fams[0], 'alone')] += dt
        else:
            for a in fams:
                for b in fams:
                    if a != b or len(fams) == 1:
                        beside[(a, b)] += dt
    last = t
    if kind:
        live.add(i)
    else:
        live.discard(i)
tot = sum(depth.values())
print(f'span {tot / 1e6:.1f} ms, {len(rows)} launches')
print('kernels running: ' + ', '.join(f'{k}{"+" if k == 3 else ""}: {100.0 * v / tot:.1f} %' for k, v in sorted(depth.items())))
fams = sorted({a for a, _ in beside})
print('family      ' + ''.join(f'{b:>9}' for b in ['alone'] + fams) + '   (ms of the span in which the family runs beside ...)')
for a in fams:
    print(f'{a:<12}' + ''.join(f'{beside.get((a, b), 0) / 1e6:9.1f}' for b in ['alone'] + fams))
dur = defaultdict(list)
for s, e, n in rows:
    dur[n[:58]].append(e - s)
print('kernel: launches, mean us here, mean us of its shortest quartile')
for n, d in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:16]:
    d.sort()
    q = d[:max(1, len(d) // 4)]
    print(f'  {n:<58} {len(d):5d} {sum(d) / len(d) / 1e3:9.1f} {sum(q) / len(q) / 1e3:9.1f}')
