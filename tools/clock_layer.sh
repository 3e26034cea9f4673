#!/bin/bash
# usage (GPU box): bash tools/clock_layer.sh N CIN COUT D H W [cin2] - effective shader clock of a conv layer: GRBM_GUI_ACTIVE / 8 / duration
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/clock
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp FNN_KNOBS=1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $out/p -- python3 $root/tools/layer_time.py "$@" > $out/log.txt 2>&1
python3 - <<EOF
import csv, glob, collections
kt = glob.glob('$out/p/**/*kernel_trace.csv', recursive=True)
cc = glob.glob('$out/p/**/*counter_collection.csv', recursive=True)
dur = {}
for r in csv.DictReader(open(kt[0])):
    if 'conv3d' in r['Kernel_Name']:
        dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
vals = collections.defaultdict(dict)
for r in csv.DictReader(open(cc[0])):
    if r['Dispatch_Id'] in dur:
        vals[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
ids = sorted(dur, key=int)[3:]
d = sum(dur[i] for i in ids) / len(ids)
g = sum(vals[i]['GRBM_GUI_ACTIVE'] for i in ids) / len(ids)
m = sum(vals[i]['SQ_VALU_MFMA_BUSY_CYCLES'] for i in ids) / len(ids)
print(f'{len(ids)} dispatches: mean {d:.1f} us, GRBM_GUI_ACTIVE/8 {g / 8:.0f} cycles -> {g / 8 / d / 1e3:.3f} GHz; MFMA busy {m / 1024 / (g / 8):.3f} of the cycles')
EOF
rm -rf $out/p
