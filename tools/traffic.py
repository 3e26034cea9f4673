#!/usr/bin/env python3
"""HBM traffic per kernel family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

    python tools/traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> [out.json]

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads
(MI355X_MICROARCH.md, HBM section), so reads are doubled; WRITE_SIZE is exact for 16-byte-per-lane stores and
uncalibrated for the 8-byte-per-lane channels-last conv stores (stated with the numbers).
"""
import collections
import csv
import json
import sys

FAMILIES = {'conv3d_': 'conv3d_mfma', 'conv_thin': 'conv3d_mfma', 'conv_row': 'conv3d_mfma', 'stem_mfma': 'stem', 'stem_row': 'stem', 'gather_head': 'seg_head_accumulate', 'tconv_mfma': 'tconv', 'seg_head': 'seg_head_accumulate',
            'finalize': 'finalize', 'stats_finalize': 'stats_finalize'}


def family(name):
    for k, v in FAMILIES.items():
        if k in name:
            return v
    return 'other'


def load(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        a = agg[family(r['Kernel_Name'])]
        a[0] += 1
        a[1] += float(r['Counter_Value'])
    return agg


def load_by_kernel(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        a = agg[(r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '')[:44], r.get('Grid_Size', ''))]
        a[0] += 1
        a[1] += float(r['Counter_Value'])
    return agg


fetch, write = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
# per kernel (name, grid): mean HBM bytes of one launch - next to a kernel's duration this is its real memory rate
fk, wk = load_by_kernel(sys.argv[1], 'FETCH_SIZE'), load_by_kernel(sys.argv[2], 'WRITE_SIZE')
print('# per kernel and grid: launches, mean read GB (FETCH_SIZE x 2), mean written GB per launch', file=sys.stderr)
for k in sorted(set(fk) | set(wk), key=lambda k: -(2 * fk[k][1] + wk[k][1])):
    n = max(fk[k][0], wk[k][0], 1)
    print(f'# {k[0]:44s} grid {k[1]:>10s}  n={n:4d}  read {2 * fk[k][1] * 1024 / n / 1e9:8.3f} GB  written {wk[k][1] * 1024 / n / 1e9:8.3f} GB', file=sys.stderr)
out = {}
for fam in sorted(set(fetch) | set(write)):
    n = max(fetch[fam][0], write[fam][0])
    rd, wr = 2.0 * fetch[fam][1] * 1024, write[fam][1] * 1024
    out[fam] = {'launches': n, 'read_bytes_per_launch': rd / max(1, n), 'write_bytes_per_launch': wr / max(1, n),
                'bytes_per_launch': (rd + wr) / max(1, n), 'total_GB': (rd + wr) / 1e9}
print(json.dumps(out, indent=1))
if len(sys.argv) > 3:
    # argv[3] = the per-round traffic file (profiles/rNN_traffic.json), argv[4] = the workload key of bench.traffic_key();
    # captures on other kernel sources are dropped (bench.py quotes a figure only when csrc_sha256 matches)
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    sha = bench.csrc_sha256()
    key = sys.argv[4] if len(sys.argv) > 4 else 'bone_turbo_r2|f16|mirror=0|fp16|vol=512|batch=32'
    doc = {}
    if os.path.isfile(sys.argv[3]):
        try:
            doc = json.load(open(sys.argv[3]))
        except Exception:
            doc = {}
    if doc.get('csrc_sha256') != sha:
        doc = {'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), reads x2 per the gfx950 correction; '
                       'bench.py --steps 1 --warmup 0 per workload key', 'csrc_sha256': sha, 'workloads': {}}
    doc['workloads'][key] = {'families': out}
    json.dump(doc, open(sys.argv[3], 'w'), indent=1)
