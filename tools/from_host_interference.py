#!/usr/bin/env python3
"""Where the from-host step's extra milliseconds go: the RESIDENT step (nothing waits for data) beside a 512 MiB pinned host-to-device copy on a
side stream - what the copy traffic alone costs the kernels - next to the resident step alone and the real from-host step (tools/from_host_ab.py).
    python tools/from_host_interference.py [steps]"""
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
pin = vol.cpu().pin_memory()
dst = torch.empty_like(vol)
side = torch.cuda.Stream()


def run(x, copy=False, k=steps):
    out = p.predict_sliding_window_return_logits(x); del out
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        if copy:
            with torch.cuda.stream(side):
                dst.copy_(pin, non_blocking=True)           # 512 MiB over the link while the step runs
        out = p.predict_sliding_window_return_logits(x); del out
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


for r in range(3):
    a, b, c = run(vol), run(vol, copy=True), run(pin)
    print(f'resident {a:.2f} ms   resident + side copy {b:.2f} ms (+{b - a:.2f})   from pinned host {c:.2f} ms (+{c - a:.2f})', flush=True)
