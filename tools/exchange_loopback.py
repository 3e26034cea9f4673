"""Loopback measurement of one rank's patch-activation exchange at the 8-rank decomposition of the bench volume, on ONE GPU
(VERDICT r3 item 3b): rank 0 computes its own 75 patches, packs its real send lists (fnn_pack_regions: one launch per peer),
the messages are copied device-to-device (the stand-in for RCCL over xGMI: what is measured is everything AROUND the wire),
and messages of the real receive sizes are landed in the foreign slots (fnn_unpack_regions) - host ms (time until the calls
return, stream not synchronised) and device ms (HIP events), with and without test-time mirroring.  Also the Python
geometry (Decomposition + ExchangePlan: paid once per volume shape since round 4, per step before).
usage (GPU box): python tools/exchange_loopback.py > profiles/r04_exchange_loopback.txt"""
import os
import sys
import time
os.environ.setdefault('FNN_KNOBS', '1')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from fast_nnunet_amd import capi
from fast_nnunet_amd.dist import Decomposition, ExchangePlan, mirror_flips

dev = torch.device('cuda', 0)
WORLD, RANK = 8, 0
for mirror in (False, True):
    predictor, sd, info = bench.build_predictor('bone_turbo_r2', dev, 32, 'fp16', 'f16', mirror)
    vol = bench.synthetic_volume(512, dev)
    eng, patch = predictor._engine, info['patch']
    t0 = time.perf_counter()
    padded, pad_lo, origins = capi.plan_volume(patch, vol.shape[1:], 0.5)
    steps = [sorted(set(int(v) for v in origins[:, d])) for d in range(3)]
    dec = Decomposition.build(patch, padded, steps, WORLD)
    boundary, interior = dec.split_patches_for_features(RANK, patch, origins)
    _, recvs = dec.feature_transfers(RANK, patch, origins)
    slot_of = {pid: i for i, pid in enumerate(boundary + interior)}
    for _, pid, _ in recvs:
        slot_of.setdefault(pid, len(slot_of))
    n_slots, C = len(slot_of), eng.feature_channels
    flips = mirror_flips((0, 1, 2) if mirror else None)
    plan = ExchangePlan(dec, RANK, patch, origins, slot_of, flips, C, dev, n_slots)
    t_geo = (time.perf_counter() - t0) * 1e3
    E = len(flips)
    feat = torch.empty((E, n_slots, *patch, C), dtype=torch.half, device=dev)
    fss = torch.empty((E, n_slots, 2, C), dtype=torch.float32, device=dev)
    opts = predictor._opts()
    eng.patch_features(vol.data_ptr(), vol.shape, opts, boundary + interior, feat.data_ptr(), fss.data_ptr(), fold=0, slot0=0, n_slots=n_slots)
    torch.cuda.synchronize()
    st = torch.cuda.current_stream().cuda_stream
    rows2d = fss.reshape(-1, 2, C)
    send_bufs = [torch.empty(m['numel'], dtype=torch.half, device=dev) for m in plan.send]
    wire = [torch.empty(m['numel'], dtype=torch.half, device=dev) for m in plan.send]
    recv_bufs = [torch.randn(m['numel'], device=dev).half() for m in plan.recv]

    def one_exchange():
        for m, buf, w in zip(plan.send, send_bufs, wire):
            eng.pack_regions(feat.data_ptr(), n_slots, m['table'].data_ptr(), len(m['recs']), buf.data_ptr(), st)
            rows = rows2d.index_select(0, m['rows'])
            w.copy_(buf)                                     # stand-in for the wire
        for m, buf in zip(plan.recv, recv_bufs):
            eng.unpack_regions(feat.data_ptr(), n_slots, m['table'].data_ptr(), len(m['recs']), buf.data_ptr(), st)
            rows2d.index_copy_(0, m['rows'], rows2d.index_select(0, m['rows']))

    one_exchange()
    torch.cuda.synchronize()
    host, devt = [], []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        one_exchange()
        e1.record()
        host.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
        devt.append(e0.elapsed_time(e1))
    sent = sum(m['numel'] for m in plan.send) * 2
    recd = sum(m['numel'] for m in plan.recv) * 2
    print(f'mirroring {"(0, 1, 2): 8 evaluations" if mirror else "off"}: rank {RANK} of {WORLD} (grid {dec.grid}), {len(boundary)} boundary + {len(interior)} interior '
          f'patches, {n_slots - len(boundary) - len(interior)} foreign slots; {len(plan.send)} peers to send to / {len(plan.recv)} to receive from, '
          f'{sum(len(m["recs"]) for m in plan.send)} + {sum(len(m["recs"]) for m in plan.recv)} regions, {sent / 1e6:.1f} MB out, {recd / 1e6:.1f} MB in')
    print(f'  pack + device copy + unpack: host {np.median(host):.3f} ms (min {min(host):.3f}), device {np.median(devt):.3f} ms (min {min(devt):.3f}); '
          f'geometry + tables once per volume shape: {t_geo:.1f} ms of Python')
    del predictor, feat, fss
    torch.cuda.empty_cache()
