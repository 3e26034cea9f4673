#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 kernel-trace CSV (bench.py run with FNN_NO_PIPELINE=1: one stream).
usage: python tools/trace_gaps.py <kernel_trace.csv>
Prints the distribution of the gaps (end of kernel i -> start of kernel i + 1) shorter than 100 us (longer ones are host
pauses between steps) and what share of the traced span they are: what back-to-back dependent launches cost on this stack."""
import csv
import sys

rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1]))))
gaps, busy = [], 0
for (s0, e0, _), (s1, e1, _) in zip(rows, rows[1:]):
    busy += e0 - s0
    g = s1 - e0
    if 0 <= g < 100_000:
        gaps.append(g)
gaps.sort()
n = len(gaps)
if not n:
    sys.exit('no gaps found')
tot = sum(gaps)
print(f'{n} gaps below 100 us: median {gaps[n // 2] / 1e3:.2f} us, mean {tot / n / 1e3:.2f} us, p90 {gaps[int(n * 0.9)] / 1e3:.2f} us, '
      f'sum {tot / 1e6:.2f} ms = {100.0 * tot / (tot + busy):.2f} % of kernel time + gaps')
