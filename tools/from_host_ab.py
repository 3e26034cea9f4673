#!/usr/bin/env python3
"""The benchmark step from a host tensor against the resident one, alternating, for several slab sizes of the engine's
upload (FNN_UPLOAD_SLAB_BYTES is read per call; needs FNN_KNOBS=1):  python tools/from_host_ab.py [steps]"""
import os
import sys
import time

os.environ.setdefault('FNN_KNOBS', '1')
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device('cuda', 0)
p, sd, info = bench.build_predictor('bone_turbo_r2', dev, 32, 'fp16')
vol = bench.synthetic_volume(512, dev)
cpu = vol.cpu()
pin = cpu.pin_memory()


def run(x, k=steps):
    out = p.predict_sliding_window_return_logits(x); del out
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        out = p.predict_sliding_window_return_logits(x); del out
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


print(f'resident {run(vol):.2f} ms')
for mb in (4, 8, 16, 32, 64, 128, 512):
    os.environ['FNN_UPLOAD_SLAB_BYTES'] = str(mb << 20)
    r = [run(vol), run(pin), run(cpu), run(vol), run(pin), run(cpu)]
    print(f'slab {mb:4d} MiB: resident {r[0]:.2f} / {r[3]:.2f}  pinned {r[1]:.2f} / {r[4]:.2f}  pageable {r[2]:.2f} / {r[5]:.2f} ms', flush=True)
