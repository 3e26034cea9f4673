"""Multi-GPU sharding logic on CPU: the decomposition is checked exhaustively as integer geometry, and the
halo exchange runs for real with world_size 2 and 3 over the gloo backend (one process per rank) against the
single-process result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import sliding_window as osw


def _geometry(shape, patch, step):
    pads, _ = osw.pad_to_patch(shape, patch)
    padded = [s + a + b for s, (a, b) in zip(shape, pads)]
    steps = osw.tile_starts(padded, patch, step)
    return padded, [p[0] for p in pads], steps


CASES = [((512, 512, 512), (160, 96, 96), 0.5), ((512, 512, 512), (128, 128, 128), 0.5), ((40, 36, 44), (16, 16, 16), 0.5),
         ((11, 30, 9), (16, 16, 16), 0.5), ((100, 33, 70), (32, 32, 24), 0.3), ((64, 64, 64), (64, 64, 64), 0.5)]


@pytest.mark.parametrize('shape,patch,step', CASES)
@pytest.mark.parametrize('world', [1, 2, 3, 4, 8])
def test_decomposition_is_a_partition(shape, patch, step, world):
    from fast_nnunet_amd.dist import Decomposition
    padded, _, steps = _geometry(shape, patch, step)
    dec = Decomposition.build(patch, padded, steps, world)
    n_patches = int(np.prod([len(s) for s in steps]))
    # every patch is run by exactly one rank
    all_ids = sorted(i for ids in dec.patch_ids for i in ids)
    assert all_ids == list(range(n_patches))
    # owned boxes tile the padded volume
    owner = np.full(padded, -1, dtype=np.int16) if np.prod(padded) <= 4e6 else None
    vol = 0
    for r, ob in enumerate(dec.owned):
        if ob is None:
            assert dec.patch_ids[r] == []
            continue
        vol += int(np.prod([h - l for l, h in zip(*ob)]))
        bx = dec.boxes[r]
        assert all(bx[0][d] <= ob[0][d] and ob[1][d] <= bx[1][d] for d in range(3)), 'owned box must lie in the accumulator box'
        if owner is not None:
            sl = tuple(slice(ob[0][d], ob[1][d]) for d in range(3))
            assert (owner[sl] == -1).all()
            owner[sl] = r
    assert vol == int(np.prod(padded))
    # every patch lies inside its rank's accumulator box
    ny, nz = len(steps[1]), len(steps[2])
    for r, ids in enumerate(dec.patch_ids):
        for i in ids:
            o = (steps[0][i // (ny * nz)], steps[1][(i // nz) % ny], steps[2][i % nz])
            assert all(dec.boxes[r][0][d] <= o[d] and o[d] + patch[d] <= dec.boxes[r][1][d] for d in range(3))
    # whatever a rank accumulates outside its own box of ownership is sent to exactly the owner
    for r in range(world):
        if dec.boxes[r] is None:
            continue
        sends, _ = dec.transfers(r)
        sent = sum(int(np.prod([h - l for l, h in zip(*reg)])) for _, reg in sends)
        own = int(np.prod([h - l for l, h in zip(*dec.owned[r])]))
        box = int(np.prod([h - l for l, h in zip(*dec.boxes[r])]))
        assert sent + own == box
    # sends and receives pair up
    for r in range(world):
        for peer, reg in dec.transfers(r)[0]:
            assert (r, reg) in dec.transfers(peer)[1]


def test_benchmark_decomposition_is_balanced_and_cheap():
    from fast_nnunet_amd.dist import Decomposition
    padded, _, steps = _geometry((512,) * 3, (160, 96, 96), 0.5)
    dec = Decomposition.build((160, 96, 96), padded, steps, 8)
    assert sorted(len(i) for i in dec.patch_ids) == [75] * 8
    # halo traffic per rank (61+1 -> 64 channels fp32) stays far below a full-buffer all-reduce (57 GB)
    worst = max(dec.halo_voxels(r) for r in range(8)) * 64 * 4 / 1e9
    assert worst < 8.0, worst


def test_boundary_patches_are_exactly_those_that_feed_another_rank():
    from fast_nnunet_amd.dist import Decomposition, _intersect
    patch = (160, 96, 96)
    padded, pad_lo, steps = _geometry((512, 512, 512), patch, 0.5)
    origins = [(x, y, z) for x in steps[0] for y in steps[1] for z in steps[2]]
    for world in (1, 2, 4, 8):
        dec = Decomposition.build(patch, padded, steps, world)
        for r in range(world):
            boundary, interior = dec.split_patches(r, patch, origins)
            assert sorted(boundary + interior) == sorted(dec.patch_ids[r]) and not set(boundary) & set(interior)
            sends = dec.transfers(r)[0]
            for i in dec.patch_ids[r]:
                ext = (tuple(origins[i]), tuple(origins[i][d] + patch[d] for d in range(3)))
                touches = any(_intersect(ext, reg) is not None for _, reg in sends)
                assert touches == (i in boundary)
            if world == 1:
                assert not boundary
            if world == 8:                       # 3 x 5 x 5 patches per rank, one layer per cut face is boundary
                assert len(dec.patch_ids[r]) == 75 and len(interior) == 2 * 4 * 4


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _toy_logits(origin, patch, heads):
    """Deterministic per-patch 'network output' that depends on the absolute position and on the patch."""
    gx = torch.arange(patch[0]).view(-1, 1, 1) + origin[0]
    gy = torch.arange(patch[1]).view(1, -1, 1) + origin[1]
    gz = torch.arange(patch[2]).view(1, 1, -1) + origin[2]
    base = torch.sin(gx * 0.37) + torch.cos(gy * 0.23) * torch.sin(gz * 0.11) + 0.01 * float(sum(origin))
    return torch.stack([base * (h + 1) - h for h in range(heads)], -1)          # [px,py,pz,heads]


def _accumulate(ids, origins, box, patch, heads, hp, gauss):
    dims = [box[1][d] - box[0][d] for d in range(3)]
    acc = torch.zeros((*dims, hp), dtype=torch.float32)
    for i in ids:
        o = [int(v) for v in origins[i]]
        sl = tuple(slice(o[d] - box[0][d], o[d] - box[0][d] + patch[d]) for d in range(3))
        acc[sl][..., :heads] += _toy_logits(o, patch, heads) * gauss[..., None]
        acc[sl][..., heads] += gauss
    return acc


def _worker(rank, world, port, shape, patch, step, heads, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from fast_nnunet_amd.dist import Decomposition, HaloExchange, exchange_halos, gather_owned_boxes, unpadded
        padded, pad_lo, steps = _geometry(shape, patch, step)
        origins = [(x, y, z) for x in steps[0] for y in steps[1] for z in steps[2]]
        hp = (heads + 1 + 7) // 8 * 8
        gauss = osw.gaussian_weight(patch).float()
        dec = Decomposition.build(patch, padded, steps, world)
        box = dec.boxes[rank]
        out = torch.full((heads, *shape), float('nan'))       # what no rank owns or gathers would stay NaN
        labels = torch.full(shape, 255, dtype=torch.uint8)
        if box is not None:
            # the order ShardedPredictor uses: boundary patches, sends leave, interior patches, receives are added
            boundary, interior = dec.split_patches(rank, patch, origins)
            assert sorted(boundary + interior) == sorted(dec.patch_ids[rank])
            acc = _accumulate(boundary, origins, box, patch, heads, hp, gauss)
            hx = HaloExchange(acc, dec, rank, None).start()
            acc += _accumulate(interior, origins, box, patch, heads, hp, gauss)
            hx.finish()
            own = unpadded(dec.owned[rank], pad_lo, shape)
            if own is not None:
                ob = dec.owned[rank]
                sl_acc = tuple(slice(own[0][d] + pad_lo[d] - box[0][d], own[1][d] + pad_lo[d] - box[0][d]) for d in range(3))
                part = acc[sl_acc]
                sl_out = tuple(slice(own[0][d], own[1][d]) for d in range(3))
                out[(slice(None), *sl_out)] = (part[..., :heads] / part[..., heads:heads + 1]).permute(3, 0, 1, 2)
                labels[sl_out] = out[(slice(None), *sl_out)].argmax(0).to(torch.uint8)
        else:
            exchange_halos(torch.empty(0), dec, rank, None)
        # the assembly step of ShardedPredictor: every rank ends up with the whole volume (logits: leading head axis;
        # labels: none), although ranks own boxes of different sizes and some may own nothing
        owns = [None if b is None else unpadded(b, pad_lo, shape) for b in dec.owned]
        gather_owned_boxes(out, owns, rank, None)
        gather_owned_boxes(labels, owns, rank, None)
        q.put((rank, out.numpy(), labels.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3, 7])
def test_halo_exchange_matches_single_process_gloo(world):
    shape, patch, step, heads = (30, 41, 26), (16, 16, 16), 0.5, 3
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, shape, patch, step, heads, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r for r, _, _ in results) == list(range(world))
    got = results[0][1]
    for _, o, l in results[1:]:                               # every rank holds the same assembled volume
        assert np.array_equal(o, got) and np.array_equal(l, results[0][2])
    assert np.array_equal(results[0][2], got.argmax(0).astype(np.uint8))
    # single-process reference
    padded, pad_lo, steps = _geometry(shape, patch, step)
    origins = [(x, y, z) for x in steps[0] for y in steps[1] for z in steps[2]]
    hp = (heads + 1 + 7) // 8 * 8
    full = ((0, 0, 0), tuple(padded))
    acc = _accumulate(range(len(origins)), origins, full, patch, heads, hp, osw.gaussian_weight(patch).float())
    sl = tuple(slice(pad_lo[d], pad_lo[d] + shape[d]) for d in range(3))
    want = (acc[sl][..., :heads] / acc[sl][..., heads:heads + 1]).permute(3, 0, 1, 2).numpy()
    assert np.allclose(got, want, rtol=1e-5, atol=1e-5)


# ---------------------------------------------------------------------------------------------------------------
# gather path: ranks exchange the parts of their patches' last activations that reach into a neighbour's owned box
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('shape,patch,step', CASES)
@pytest.mark.parametrize('world', [2, 3, 8])
def test_feature_transfers_cover_every_voxel_of_every_owned_box(shape, patch, step, world):
    """Integer geometry: with its own patches and the received regions a rank holds, for every voxel of the box it owns,
    the activation of EVERY patch that covers the voxel; sends and receives pair up."""
    from fast_nnunet_amd.dist import Decomposition, _intersect
    padded, _, steps = _geometry(shape, patch, step)
    origins = [(x, y, z) for x in steps[0] for y in steps[1] for z in steps[2]]
    dec = Decomposition.build(patch, padded, steps, world)
    owner = dec.rank_of_patch()
    assert sorted(owner) == sorted(r for r, ids in enumerate(dec.patch_ids) for _ in ids) and min(owner) >= 0
    for r in range(world):
        sends, recvs = dec.feature_transfers(r, patch, origins)
        for peer, pid, reg in sends:
            assert (r, pid, reg) in dec.feature_transfers(peer, patch, origins)[1]
        if dec.owned[r] is None:
            assert not sends and not recvs
            continue
        have = {pid: [] for pid in dec.patch_ids[r]}
        for _, pid, reg in recvs:
            have.setdefault(pid, []).append(reg)
        for pid, o in enumerate(origins):
            ext = (tuple(o), tuple(o[d] + patch[d] for d in range(3)))
            need = _intersect(ext, dec.owned[r])
            if need is None:
                continue
            assert pid in have
            if owner[pid] != r:
                assert need in have[pid]
        boundary, interior = dec.split_patches_for_features(r, patch, origins)
        assert sorted(boundary + interior) == sorted(dec.patch_ids[r]) and {i for _, i, _ in sends} == set(boundary)


def _gather_worker(rank, world, port, shape, patch, step, heads, mirror, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from fast_nnunet_amd.dist import Decomposition, FeatureExchange, gather_owned_boxes, mirror_flips, unpadded
        padded, pad_lo, steps = _geometry(shape, patch, step)
        origins = [(x, y, z) for x in steps[0] for y in steps[1] for z in steps[2]]
        gauss = osw.gaussian_weight(patch).float()
        dec = Decomposition.build(patch, padded, steps, world)
        flips = mirror_flips(mirror)
        E = len(flips)

        def stored(pid, f):
            """Evaluation f of patch pid as the engine keeps it: value toy + 100 f, in the network's (flipped) coordinates."""
            v = _toy_logits(origins[pid], patch, heads) + 100.0 * f
            return torch.flip(v, list(flips[f])) if flips[f] else v

        out = torch.full((heads, *shape), float('nan'))
        if dec.owned[rank] is not None:
            boundary, interior = dec.split_patches_for_features(rank, patch, origins)
            _, recvs = dec.feature_transfers(rank, patch, origins)
            slot_of = {pid: i for i, pid in enumerate(boundary + interior)}
            for _, pid, _ in recvs:
                slot_of.setdefault(pid, len(slot_of))
            feat = torch.full((E, len(slot_of), *patch, heads), float('nan'))
            fss = torch.zeros((E, len(slot_of), 2, heads))
            for pid in boundary:                                   # the order ShardedPredictor uses
                for f in range(E):
                    feat[f, slot_of[pid]] = stored(pid, f)
                    fss[f, slot_of[pid], 0] = float(pid * 10 + f)
            fx = FeatureExchange(feat, fss, dec, rank, patch, origins, slot_of, None, flips).start()
            for pid in interior:
                for f in range(E):
                    feat[f, slot_of[pid]] = stored(pid, f)
                    fss[f, slot_of[pid], 0] = float(pid * 10 + f)
            fx.finish()
            assert all(float(fss[f, sl, 0, 0]) == pid * 10 + f for pid, sl in slot_of.items() for f in range(E))   # rows arrived with their slots
            assert fx.bytes_sent > 0 or world == 1
            # what gather.hip does over the owned box: every covering patch, ascending; per visit the mean of the
            # evaluations, each read at the flipped voxel
            ob = dec.owned[rank]
            own = unpadded(ob, pad_lo, shape)
            if own is not None:
                lo = [own[0][d] + pad_lo[d] for d in range(3)]
                hi = [own[1][d] + pad_lo[d] for d in range(3)]
                num = torch.zeros((*[hi[d] - lo[d] for d in range(3)], heads))
                den = torch.zeros([hi[d] - lo[d] for d in range(3)])
                for pid, o in enumerate(origins):
                    a = [max(lo[d], o[d]) for d in range(3)]
                    b = [min(hi[d], o[d] + patch[d]) for d in range(3)]
                    if any(b[d] <= a[d] for d in range(3)):
                        continue
                    dst = tuple(slice(a[d] - lo[d], b[d] - lo[d]) for d in range(3))
                    src = tuple(slice(a[d] - o[d], b[d] - o[d]) for d in range(3))
                    logit = sum((torch.flip(feat[f, slot_of[pid]], list(flips[f])) if flips[f] else feat[f, slot_of[pid]])[src]
                                for f in range(E)) / E
                    num[dst] += logit * gauss[src][..., None]
                    den[dst] += gauss[src]
                sl = tuple(slice(own[0][d], own[1][d]) for d in range(3))
                out[(slice(None), *sl)] = (num / den[..., None]).permute(3, 0, 1, 2)
        else:
            FeatureExchange(torch.empty((E, 0, *patch, heads)), torch.empty((E, 0, 2, heads)), dec, rank, patch, origins, {}, None,
                            flips).start().finish()
        owns = [None if b is None else unpadded(b, pad_lo, shape) for b in dec.owned]
        gather_owned_boxes(out, owns, rank, None)
        q.put((rank, out.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,mirror', [(2, None), (3, None), (8, None), (2, (0, 2)), (4, (0, 1, 2))])
def test_feature_exchange_matches_single_process_gloo(world, mirror):
    """Worlds 2 / 3 / 8 (the node has eight GPUs) and test-time mirroring: the 2^k evaluations' activations
    travel as flipped sub-blocks and land where the gather kernel reads them."""
    shape, patch, step, heads = (30, 41, 26), (16, 16, 16), 0.5, 3
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, shape, patch, step, heads, mirror, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    padded, pad_lo, steps = _geometry(shape, patch, step)
    origins = [(x, y, z) for x in steps[0] for y in steps[1] for z in steps[2]]
    hp = (heads + 1 + 7) // 8 * 8
    acc = _accumulate(range(len(origins)), origins, ((0, 0, 0), tuple(padded)), patch, heads, hp, osw.gaussian_weight(patch).float())
    sl = tuple(slice(pad_lo[d], pad_lo[d] + shape[d]) for d in range(3))
    want = (acc[sl][..., :heads] / acc[sl][..., heads:heads + 1]).permute(3, 0, 1, 2).numpy()
    n_eval = 1 if mirror is None else 2 ** len(mirror)
    want = want + 100.0 * (n_eval - 1) / 2                       # evaluation f carries toy + 100 f: the mean over f
    for _, got in results:
        assert not np.isnan(got).any() and np.allclose(got, want, rtol=1e-4, atol=1e-3)


# ------------------------------------------------------------------------------- the path decision is a collective one
def _decision_worker(rank, world, port, mode, q):
    """ShardedPredictor._use_gather_all with a stand-in predictor (no engine: the decision is host logic): rank 1 sees a
    volume where 70 tiles of one axis lie over one voxel (beyond the 64 a wave of the gather kernel holds; `counts` is what
    dist.tile_cover returns - an axis with more than 64 POSITIONS is fine since round 5 as long as 64 consecutive ones reach)."""
    import types
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from fast_nnunet_amd.dist import ShardedPredictor
        p = types.SimpleNamespace(_spec=types.SimpleNamespace(features=[16, 32], patch=(16, 16, 16)), use_mirroring=False,
                                  allowed_mirroring_axes=None, device=torch.device('cpu'))
        sp = ShardedPredictor(p, None, mode=mode)
        counts = [70, 3, 3] if rank == 1 else [6, 3, 3]
        try:
            use = sp._use_gather_all(None, counts)
            q.put((rank, 'ok', use, sp.last_mode))
        except NotImplementedError as ex:
            q.put((rank, 'raised', str(ex), None))
        dist.barrier()                                        # every rank got here: nobody was left waiting in the all-reduce
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('mode', ['auto', 'gather'])
def test_gather_path_decision_is_taken_by_every_rank_together(mode):
    """a rank that cannot take the gather path used to raise BEFORE the group's all-reduce and left the others
    waiting in it.  Now every rank reduces first: 'auto' falls back to the accumulate path everywhere (and says so in
    `last_mode`, what bench.py prints), 'gather' raises on every rank."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_decision_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    if mode == 'auto':
        assert [g[1:] for g in got] == [('ok', False, 'accumulate')] * 2
    else:
        assert all(g[1] == 'raised' for g in got)
        assert 'tiles of one axis' in got[1][2]                  # the rank that cannot says why; the other one reports the group's verdict
