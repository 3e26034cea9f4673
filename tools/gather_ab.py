"""Diagnostic: the gather kernel alone on the benchmark's kept activations, under several knob settings in ONE process
(interleaved rounds, HIP-event time of fnn_gather_box, median / min).  The 600 patches of the 512^3 bench volume are
produced once (fnn_patch_features into torch buffers), then every arm re-runs the gather over them.
usage (GPU box): python tools/gather_ab.py ["" "FNN_GATHER_PF=0" "FNN_GATHER_V1=1" ...] [--accum fp16_autocast] [--labels] [--check]
--check: every arm's output must equal the first arm's bit for bit."""
import os
import sys
import statistics
os.environ.setdefault('FNN_KNOBS', '1')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from fast_nnunet_amd import capi

args = [a for a in sys.argv[1:] if not a.startswith('--')]
accum = sys.argv[sys.argv.index('--accum') + 1] if '--accum' in sys.argv else 'fp16'
if '--accum' in sys.argv:
    args.remove(accum)
labels = '--labels' in sys.argv
check = '--check' in sys.argv
arms = args or ['', 'FNN_GATHER_PF=0', 'FNN_GATHER_V1=1']
dev = torch.device('cuda', 0)
predictor, sd, info = bench.build_predictor('bone_turbo_r2', dev, 32, accum)
vol = bench.synthetic_volume(512, dev)
eng, patch = predictor._engine, info['patch']
padded, pad_lo, origins = capi.plan_volume(patch, vol.shape[1:], 0.5)
n = int(origins.shape[0])
C = eng.feature_channels
opts = predictor._opts()
feat = torch.empty((1, n, *patch, C), dtype=torch.half, device=dev)
fss = torch.empty((1, n, 2, C), dtype=torch.float32, device=dev)
eng.patch_features(vol.data_ptr(), vol.shape, opts, list(range(n)), feat.data_ptr(), fss.data_ptr(), fold=0, slot0=0, n_slots=n)
table = np.arange(n, dtype=np.int32)
out = torch.empty((info['heads'], *vol.shape[1:]), dtype=torch.half, device=dev) if not labels else None
lab = torch.zeros(vol.shape[1:], dtype=torch.uint8, device=dev) if labels else None
if labels:
    eng.set_label_rule(None, uint16=False)
lo, hi = (0, 0, 0), tuple(int(v) for v in vol.shape[1:])


def run():
    eng.gather_box(feat.data_ptr(), fss.data_ptr(), table, vol.shape, opts, lo, hi, logits_ptr=out.data_ptr() if out is not None else None,
                   labels_ptr=lab.data_ptr() if lab is not None else None, fold=0, n_slots=n)


def set_arm(arm):
    for kv in arm.split():
        k, v = kv.split('=', 1)
        os.environ[k] = v


def clear_arm(arm):
    for kv in arm.split():
        os.environ.pop(kv.split('=', 1)[0], None)


times = {a: [] for a in arms}
ref = None
for rnd in range(6):
    for a in arms:
        set_arm(a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run()
        e1.record()
        torch.cuda.synchronize()
        clear_arm(a)
        if rnd > 0:
            times[a].append(e0.elapsed_time(e1))
        if check and rnd == 0:
            cur = (out if out is not None else lab).clone()
            if ref is None:
                ref = cur
            else:
                same = torch.equal(cur.view(torch.int16) if out is not None else cur, ref.view(torch.int16) if out is not None else ref)
                print(f'[{a}] bits equal to the first arm: {same}')
for a in arms:
    t = times[a]
    print(f'gather {accum}{" labels" if labels else ""} [{a or "default"}]: median {statistics.median(t):.2f} ms, min {min(t):.2f} ms  ({len(t)} rounds)')
