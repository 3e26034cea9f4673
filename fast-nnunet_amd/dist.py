"""Patches of ONE volume sharded over the GPUs of a node.

Not in the reference: its only inference parallelism is case-level
(``-num_parts/-part_id``, predict_from_raw_data.py:918-925).  Here the patch grid
of a volume (x-major list of ``compute_steps_for_sliding_window`` positions) is
cut into a 3-D grid of rank blocks.  Every rank

1. runs its own patches into an accumulator (fp16 like the reference's, or fp32) that
   covers only the bounding box of those patches (channels-last ``[bx, by, bz, HP]``,
   channel ``heads`` is the weight sum - so logits and weights travel together);
2. exchanges ONLY the overlap regions with the ranks whose boxes intersect its
   own: each voxel of the padded volume has exactly one owner, and every other
   rank that touched it sends its partial sums to the owner (point-to-point over
   RCCL/xGMI, ``batch_isend_irecv``).  The patches that touch a region the rank
   sends run first, so the transfers overlap the rest of its patches;  A full-buffer all-reduce would move
   ~57 GB per GPU for 61 classes on 512^3; the halos are a few GB;
3. normalises and writes the box it owns.

The geometry (`Decomposition`) is pure integer logic and is unit-tested on CPU
with the gloo backend against the single-process result.
"""
from __future__ import annotations

import itertools
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import collections
import numpy as np
import torch
import torch.distributed as dist

Box = Tuple[Tuple[int, int, int], Tuple[int, int, int]]          # (lo, hi) in padded-volume coordinates


def _split(n: int, parts: int) -> List[Tuple[int, int]]:
    """`n` items into `parts` contiguous, balanced ranges."""
    return [(n * i // parts, n * (i + 1) // parts) for i in range(parts)]


def tile_cover(steps: Sequence[Sequence[int]], patch: Sequence[int]) -> List[int]:
    """Per axis: the largest number of tiles that reach one coordinate (along the last axis: one 64-voxel run of the gather
    kernel) - what csrc/gather.hip gather_tile_windows bounds by 64.  An axis with <= 64 tile positions is its count."""
    out = []
    patch3 = [1] * (3 - len(patch)) + [int(v) for v in patch]
    for d in range(3):
        st, ext, run = list(steps[d]), patch3[d], (64 if d == 2 else 1)
        if len(st) <= 64:
            out.append(len(st))
            continue
        worst, b = 0, 0
        for c in range(st[-1] + ext):
            while b < len(st) and st[b] + ext <= c:
                b += 1
            e = b
            while e < len(st) and st[e] < c + run:
                e += 1
            worst = max(worst, e - b)
        out.append(worst)
    return out


def rank_grid(world: int, counts: Sequence[int]) -> Tuple[int, int, int]:
    """Factorisation (gx, gy, gz) of at most `world` ranks over the patch grid `counts`: fewest patches on
    the busiest rank first, then the smallest total halo surface."""
    best, best_key = (1, 1, 1), None
    for gx in range(1, world + 1):
        for gy in range(1, world // gx + 1):
            gz = world // (gx * gy)
            g = (gx, gy, gz)
            if gx * gy * gz > world or any(gi > c for gi, c in zip(g, counts)):
                continue
            busiest = int(np.prod([-(-c // gi) for c, gi in zip(counts, g)]))
            cuts = sum(gi - 1 for gi in g)
            key = (busiest, -(gx * gy * gz), cuts)
            if best_key is None or key < best_key:
                best, best_key = g, key
    return best


def _intersect(a: Box, b: Box) -> Optional[Box]:
    lo = tuple(max(a[0][d], b[0][d]) for d in range(3))
    hi = tuple(min(a[1][d], b[1][d]) for d in range(3))
    return (lo, hi) if all(h > l for l, h in zip(lo, hi)) else None


@dataclass
class Decomposition:
    world: int
    grid: Tuple[int, int, int]
    patch_ids: List[List[int]]              # per rank
    boxes: List[Optional[Box]]              # accumulator box per rank (None = idle rank)
    owned: List[Optional[Box]]              # disjoint boxes that tile the padded volume

    @staticmethod
    def build(patch: Sequence[int], padded: Sequence[int], steps: Sequence[Sequence[int]], world: int) -> 'Decomposition':
        counts = [len(s) for s in steps]
        grid = rank_grid(world, counts)
        ranges = [_split(counts[d], grid[d]) for d in range(3)]
        # per axis: accumulator interval and ownership cut points of every block
        acc_iv, own_iv = [], []
        for d in range(3):
            iv = [(steps[d][b], steps[d][e - 1] + patch[d]) for b, e in ranges[d]]
            cuts = [0]
            for a in range(len(iv) - 1):
                lo_next, hi_cur = iv[a + 1][0], iv[a][1]
                cuts.append((lo_next + hi_cur) // 2 if hi_cur > lo_next else hi_cur)
            cuts.append(padded[d])
            acc_iv.append(iv)
            own_iv.append([(cuts[a], cuts[a + 1]) for a in range(len(iv))])
        patch_ids, boxes, owned = [], [], []
        ny, nz = counts[1], counts[2]
        for r in range(world):
            if r >= grid[0] * grid[1] * grid[2]:
                patch_ids.append([]); boxes.append(None); owned.append(None)
                continue
            rz, ry, rx = r % grid[2], (r // grid[2]) % grid[1], r // (grid[2] * grid[1])
            blk = (rx, ry, rz)
            ids = [(ix * ny + iy) * nz + iz
                   for ix in range(*ranges[0][rx]) for iy in range(*ranges[1][ry]) for iz in range(*ranges[2][rz])]
            patch_ids.append(ids)
            boxes.append((tuple(acc_iv[d][blk[d]][0] for d in range(3)), tuple(acc_iv[d][blk[d]][1] for d in range(3))))
            owned.append((tuple(own_iv[d][blk[d]][0] for d in range(3)), tuple(own_iv[d][blk[d]][1] for d in range(3))))
        return Decomposition(world, grid, patch_ids, boxes, owned)

    def transfers(self, rank: int):
        """-> (sends, recvs): lists of (peer, region) with region in padded coordinates.
        A rank sends what it accumulated inside another rank's owned box."""
        sends, recvs = [], []
        if self.boxes[rank] is None:
            return sends, recvs
        for peer in range(self.world):
            if peer == rank or self.boxes[peer] is None:
                continue
            out = _intersect(self.boxes[rank], self.owned[peer])
            if out is not None:
                sends.append((peer, out))
            inc = _intersect(self.boxes[peer], self.owned[rank])
            if inc is not None:
                recvs.append((peer, inc))
        return sends, recvs

    def split_patches_for_features(self, rank: int, patch: Sequence[int], origins) -> Tuple[List[int], List[int]]:
        """-> (boundary, interior) for the gather path: boundary patches reach into another rank's OWNED box."""
        need = {i for _, i, _ in self.feature_transfers(rank, patch, origins)[0]}
        ids = self.patch_ids[rank]
        return [i for i in ids if i in need], [i for i in ids if i not in need]

    def split_patches(self, rank: int, patch: Sequence[int], origins) -> Tuple[List[int], List[int]]:
        """-> (boundary, interior) patch ids of `rank` in visiting order: a boundary patch touches a region this
        rank sends to another rank; once they are accumulated the sends can leave while the interior computes."""
        sends, _ = self.transfers(rank)
        boundary, interior = [], []
        for i in self.patch_ids[rank]:
            o = [int(v) for v in origins[i]]
            ext: Box = (tuple(o), tuple(o[d] + patch[d] for d in range(3)))
            (boundary if any(_intersect(ext, reg) is not None for _, reg in sends) else interior).append(i)
        return boundary, interior

    def halo_voxels(self, rank: int) -> int:
        return sum(int(np.prod([h - l for l, h in zip(*reg)])) for _, reg in self.transfers(rank)[0])

    # ---- gather path: what travels is the part of a patch's last activation that reaches into another rank's owned box
    def rank_of_patch(self) -> List[int]:
        n = sum(len(i) for i in self.patch_ids)
        out = [-1] * n
        for r, ids in enumerate(self.patch_ids):
            for i in ids:
                out[i] = r
        return out

    def feature_transfers(self, rank: int, patch: Sequence[int], origins):
        """-> (sends, recvs): lists of (peer, patch id, region) with the region in padded-volume coordinates, ordered
        by (peer, patch id) on both sides.  A rank sends, for each of its patches, the intersection of the patch with
        every OTHER rank's owned box; it receives the parts of foreign patches that reach into the box it owns."""
        owner = self.rank_of_patch()
        sends, recvs = [], []
        if self.owned[rank] is None:
            return sends, recvs
        for peer in range(self.world):
            if peer == rank or self.owned[peer] is None:
                continue
            for i in self.patch_ids[rank]:
                o = [int(v) for v in origins[i]]
                reg = _intersect((tuple(o), tuple(o[d] + patch[d] for d in range(3))), self.owned[peer])
                if reg is not None:
                    sends.append((peer, i, reg))
            for i in self.patch_ids[peer]:
                o = [int(v) for v in origins[i]]
                reg = _intersect((tuple(o), tuple(o[d] + patch[d] for d in range(3))), self.owned[rank])
                if reg is not None:
                    recvs.append((peer, i, reg))
        assert all(owner[i] == rank for _, i, _ in sends)
        return sends, recvs


def _view(acc: torch.Tensor, box: Box, region: Box) -> torch.Tensor:
    sl = tuple(slice(region[0][d] - box[0][d], region[1][d] - box[0][d]) for d in range(3))
    return acc[sl]


class HaloExchange:
    """The exchange in two halves so that compute can run in between: ``start()`` posts every send and receive
    (the sends read `acc` as the current stream leaves it - call it once every patch that touches a region this
    rank sends has been accumulated, see ``Decomposition.split_patches``), ``finish()`` waits and adds the other
    ranks' contributions to the part of `acc` this rank owns.  `acc` is [bx, by, bz, HP] over ``dec.boxes[rank]``."""

    def __init__(self, acc: torch.Tensor, dec: Decomposition, rank: int, group=None):
        self.acc, self.dec, self.rank, self.group = acc, dec, rank, group
        self.reqs, self.landing, self.keep = [], [], []

    def start(self) -> 'HaloExchange':
        sends, recvs = self.dec.transfers(self.rank)
        if not sends and not recvs:
            return self
        box, group, acc = self.dec.boxes[self.rank], self.group, self.acc
        ops = []
        for peer, region in sends:
            buf = _view(acc, box, region).contiguous()
            self.keep.append(buf)
            ops.append(dist.P2POp(dist.isend, buf, dist.get_global_rank(group, peer) if group is not None else peer, group))
        for peer, region in recvs:
            shape = tuple(region[1][d] - region[0][d] for d in range(3)) + (acc.shape[3],)
            buf = torch.empty(shape, dtype=acc.dtype, device=acc.device)
            ops.append(dist.P2POp(dist.irecv, buf, dist.get_global_rank(group, peer) if group is not None else peer, group))
            self.landing.append((region, buf))
        self.reqs = dist.batch_isend_irecv(ops)
        return self

    def finish(self) -> None:
        for req in self.reqs:
            req.wait()
        box = self.dec.boxes[self.rank]
        # fixed (peer-rank) order -> the sums are reproducible run to run
        for region, buf in self.landing:
            _view(self.acc, box, region).add_(buf)
        self.reqs, self.landing, self.keep = [], [], []


def exchange_halos(acc: torch.Tensor, dec: Decomposition, rank: int, group=None) -> None:
    """Adds the other ranks' contributions to the part of `acc` this rank owns (start + finish in one go)."""
    HaloExchange(acc, dec, rank, group).start().finish()


def mirror_flips(mirror_axes) -> List[Tuple[int, ...]]:
    """The evaluations of one patch under test-time mirroring as tuples of flipped axes, the un-mirrored one first, then
    the non-empty subsets of `mirror_axes` by size, then lexicographically - the order of
    _internal_maybe_mirror_and_predict (predict_from_raw_data.py:541-557) and of the engine's evaluation index."""
    if not mirror_axes:
        return [()]
    axes = list(mirror_axes)
    return [()] + [c for k in range(len(axes)) for c in itertools.combinations(axes, k + 1)]


class ExchangePlan:
    """What ``FeatureExchange`` moves, derived ONCE per decomposition (the lists depend on the volume's shape and the rank
    grid only, not on its values): per peer and direction the (evaluation, slot, sub-block) records - as the device tables
    of ``fnn_pack_regions`` / ``fnn_unpack_regions`` (include/fnn.h) on a GPU - the message sizes, and the row numbers of
    the InstanceNorm rows that travel with the blocks.  Both sides derive the same lists, so no metadata travels."""

    def __init__(self, dec: Decomposition, rank: int, patch, origins, slot_of, flips, channels: int, device, n_slots: int):
        self.patch, self.flips, self.C, self.n_slots = tuple(patch), [tuple(f) for f in flips], int(channels), int(n_slots)
        self.origins, self.slot_of = origins, slot_of
        sends, recvs = dec.feature_transfers(rank, patch, origins)
        self.send, self.recv = self._tables(sends, device), self._tables(recvs, device)

    def local(self, pid: int, region: Box, flip=()):
        """The block of `region` in the slot's own coordinates: [P - hi, P - lo) along every flipped axis."""
        o = [int(v) for v in self.origins[pid]]
        lo = [region[0][d] - o[d] for d in range(3)]
        hi = [region[1][d] - o[d] for d in range(3)]
        return tuple((self.patch[d] - hi[d], self.patch[d] - lo[d]) if d in flip else (lo[d], hi[d]) for d in range(3))

    def _tables(self, items, device):
        per = {}
        for peer, pid, reg in items:
            per.setdefault(peer, []).append((pid, reg))
        out = []
        for peer, lst in sorted(per.items()):
            recs, rows, off = [], [], 0                             # off: elements; the device table wants 16-byte units (C % 8 == 0 there)
            for f, fl in enumerate(self.flips):                     # evaluation-major, then the (patch, region) list: both sides alike
                for pid, reg in lst:
                    loc = self.local(pid, reg, fl)
                    recs.append([f, self.slot_of[pid], loc[0][0], loc[1][0], loc[2][0], loc[0][1], loc[1][1], loc[2][1], off // 8, 0])
                    rows.append(f * self.n_slots + self.slot_of[pid])
                    off += int(np.prod([b - a for a, b in loc])) * self.C
            table = torch.tensor(recs, dtype=torch.int32).reshape(-1, 10)
            out.append(dict(peer=peer, items=lst, recs=recs, numel=off,
                            table=table.to(device) if device is not None and torch.device(device).type == 'cuda' else table,
                            rows=torch.tensor(rows, dtype=torch.int64, device=device)))
        return out


class FeatureExchange:
    """Gather-path exchange (SURVEY.md 8e with csrc/gather.hip): `feat` is [n_eval, n_slots, PD, PH, PW, C] (fp16 on the
    GPUs; a 5-D tensor is taken as n_eval = 1), `fss` [n_eval, n_slots, 2, C]; `slot_of[pid]` says where a patch sits.
    ``start()`` sends the regions of this rank's patches that other ranks' owned boxes need (call it once those patches
    are computed), ``finish()`` lands the foreign regions in their slots.  One packed buffer per peer and direction; both
    sides derive the same (peer, patch, region) lists from the decomposition, so no metadata travels.  `flips`: the
    evaluations under test-time mirroring (``mirror_flips``): the activation of a mirrored evaluation is stored in the
    network's coordinates, so the block of a region sits at [P - hi, P - lo) along every flipped axis - on both sides.
    Round 4: on a GPU the packing and the landing are ONE launch per peer and direction (``fnn_pack_regions`` /
    ``fnn_unpack_regions`` through `engine`, tables from a cached ``ExchangePlan``) instead of a strided torch copy per
    (evaluation, patch, region); host tensors (the gloo tests of this logic) take the torch copies."""

    def __init__(self, feat: torch.Tensor, fss: torch.Tensor, dec: Decomposition, rank: int, patch, origins, slot_of, group=None,
                 flips=((),), engine=None, plan: Optional[ExchangePlan] = None):
        if feat.ndim == 5:
            feat, fss = feat[None], fss[None]
        self.feat, self.fss, self.group, self.engine = feat, fss, group, engine
        assert feat.shape[0] == len(flips)
        if feat.is_cuda and engine is None:
            raise RuntimeError('FeatureExchange on device tensors needs the engine (fnn_pack_regions / fnn_unpack_regions): no torch fallback on a GPU')
        self.plan = plan if plan is not None else ExchangePlan(dec, rank, patch, origins, slot_of, flips, feat.shape[-1],
                                                               feat.device if feat.is_cuda else None, feat.shape[1])
        self.reqs, self.landing, self.keep = [], [], []
        self.bytes_sent = self.bytes_received = 0

    def _block(self, rec):
        f, slot, l0, l1, l2, h0, h1, h2 = rec[:8]
        return (f, slot, slice(l0, h0), slice(l1, h1), slice(l2, h2))

    def start(self) -> 'FeatureExchange':
        pl = self.plan
        if not pl.send and not pl.recv:
            return self
        g = self.group
        dst = (lambda r: dist.get_global_rank(g, r)) if g is not None else (lambda r: r)
        ops = []
        C = self.feat.shape[-1]
        rows2d = self.fss.reshape(-1, 2, C)
        for m in pl.send:
            buf = torch.empty(m['numel'], dtype=self.feat.dtype, device=self.feat.device)
            if self.feat.is_cuda:
                self.engine.pack_regions(self.feat.data_ptr(), pl.n_slots, m['table'].data_ptr(), len(m['recs']), buf.data_ptr(),
                                         torch.cuda.current_stream(self.feat.device).cuda_stream)
            else:
                off = 0
                for rec in m['recs']:
                    blk = self.feat[self._block(rec)].reshape(-1)
                    buf[off:off + blk.numel()] = blk
                    off += blk.numel()
            rows = rows2d.index_select(0, m['rows'])
            self.keep += [buf, rows]
            self.bytes_sent += buf.numel() * buf.element_size() + rows.numel() * rows.element_size()
            ops += [dist.P2POp(dist.isend, buf, dst(m['peer']), g), dist.P2POp(dist.isend, rows, dst(m['peer']), g)]
        for m in pl.recv:
            buf = torch.empty(m['numel'], dtype=self.feat.dtype, device=self.feat.device)
            rows = torch.empty((len(m['recs']), 2, C), dtype=self.fss.dtype, device=self.fss.device)
            self.bytes_received += buf.numel() * buf.element_size() + rows.numel() * rows.element_size()
            ops += [dist.P2POp(dist.irecv, buf, dst(m['peer']), g), dist.P2POp(dist.irecv, rows, dst(m['peer']), g)]
            self.landing.append((m, buf, rows))
        self.reqs = dist.batch_isend_irecv(ops)
        return self

    def finish(self) -> None:
        for req in self.reqs:
            req.wait()
        pl = self.plan
        C = self.feat.shape[-1]
        rows2d = self.fss.reshape(-1, 2, C)
        for m, buf, rows in self.landing:
            if self.feat.is_cuda:
                self.engine.unpack_regions(self.feat.data_ptr(), pl.n_slots, m['table'].data_ptr(), len(m['recs']), buf.data_ptr(),
                                           torch.cuda.current_stream(self.feat.device).cuda_stream)
            else:
                off = 0
                for rec in m['recs']:
                    tgt = self.feat[self._block(rec)]
                    self.feat[self._block(rec)] = buf[off:off + tgt.numel()].view(tgt.shape)
                    off += tgt.numel()
            rows2d.index_copy_(0, m['rows'], rows)
        self.reqs, self.landing, self.keep = [], [], []


def unpadded(box: Box, pad_lo: Sequence[int], shape_sp: Sequence[int]) -> Optional[Box]:
    lo = tuple(max(0, box[0][d] - pad_lo[d]) for d in range(3))
    hi = tuple(min(shape_sp[d], box[1][d] - pad_lo[d]) for d in range(3))
    return (lo, hi) if all(h > l for l, h in zip(lo, hi)) else None


def gather_owned_boxes(full: torch.Tensor, owns: Sequence[Optional[Box]], rank: int, group=None) -> torch.Tensor:
    """Assembles `full` ([..., X, Y, Z]; every rank has written only the box it owns, ``owns[rank]`` in un-padded
    coordinates, None = nothing) on EVERY rank: each rank packs its box into a flat buffer of the largest box's size,
    one ``all_gather`` moves the slabs (RCCL over xGMI on the GPUs, gloo in the CPU tests), every rank unpacks the
    other ranks' boxes.  The owned boxes are disjoint and tile the volume, so nothing is added: the bytes that travel
    are the volume itself, once ((world - 1) / world of it per rank), not a reduction over full-size buffers."""
    world = len(owns)
    lead = tuple(full.shape[:-3])
    nlead = int(np.prod(lead)) if lead else 1

    def nvox(b):
        return 0 if b is None else int(np.prod([b[1][d] - b[0][d] for d in range(3)]))

    cap = max(nvox(b) for b in owns) * nlead
    if cap == 0:
        return full
    mine = torch.zeros(cap, dtype=full.dtype, device=full.device)
    b = owns[rank]
    if b is not None:
        sl = (Ellipsis, *[slice(b[0][d], b[1][d]) for d in range(3)])
        mine[:nvox(b) * nlead] = full[sl].reshape(-1)
    slabs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(slabs, mine, group=group)
    for r, br in enumerate(owns):
        if r == rank or br is None:
            continue
        shape_r = lead + tuple(br[1][d] - br[0][d] for d in range(3))
        sl = (Ellipsis, *[slice(br[0][d], br[1][d]) for d in range(3)])
        full[sl] = slabs[r][:nvox(br) * nlead].view(shape_r)
    return full


class ShardedPredictor:
    """The predictor's whole-volume entry points with the patches of ONE volume sharded over the ranks of `group`.

    Every rank passes the same (replicated) preprocessed volume.
    ``predict_sliding_window_return_logits`` returns ``(logits, owned_box)``: a full-size fp16 tensor in which this
    rank has written the box it owns (un-padded coordinates; ``None`` on an idle rank, whose tensor holds nothing) -
    or, with ``gather=True``, the assembled logits on every rank.  ``predict_segmentation_from_preprocessed_data``
    labels each owned box on its owner and gathers the uint8 / uint16 slabs (128 MiB for a 512^3 volume instead of
    15 GiB of fp16 logits for 61 classes): the assembled label map on every rank.  With several folds loaded both
    form the ensemble mean like the single-GPU predictor.
    """

    def __init__(self, predictor, group=None, mode: str = 'auto'):
        """mode: 'gather' - ranks keep patch activations and exchange the parts that reach into a neighbour's box
        (csrc/gather.hip: no accumulators, results bit-identical to one GPU, test-time mirroring included: the 2^k
        evaluations' activations travel like any other); 'accumulate' - ranks exchange partial sums of read-modify-write
        accumulators; 'auto' - gather where the engine's gather kernel applies and the kept activations fit in HBM."""
        self.p = predictor
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        assert mode in ('auto', 'gather', 'accumulate')
        self.mode = mode
        # per (volume shape, patch, step size, world): decomposition, slots, exchange tables - the 8 most recently used
        # (real datasets have another shape for nearly every case: an unbounded cache grows for ever); `_cur` = the entry of
        # the volume being predicted, only ever used through an identity check against the decomposition in hand
        self._plans, self._cur = collections.OrderedDict(), None
    PLAN_CACHE = 8

    def reset_plans(self):
        """Drop the cached decompositions (the predictor was re-initialised with another network)."""
        self._plans.clear()
        self._cur = None

    def _flips(self):
        p = self.p
        return mirror_flips(p.allowed_mirroring_axes if p.use_mirroring else None)

    def _gather_fits(self, n_slots: Optional[int] = None, counts: Optional[Sequence[int]] = None) -> Tuple[bool, str]:
        """Does the gather path apply on THIS rank: the engine's gather kernel does (<= 32 channels at full resolution,
        <= 8 evaluations per patch, <= 64 tiles of one axis over a voxel (`counts`: tile_cover); more than 63 classes run as
        passes over the heads) AND
        the kept activations fit - `n_slots` patch slots per evaluation against 80 % of the free HBM (the single-GPU engine
        bounds the same buffers, csrc/engine.hip gather_plan).  Never raises: the decision is a collective one."""
        p = self.p
        n_eval = len(self._flips())
        if not (n_eval <= 8 and p._spec.features[0] <= 32 and (counts is None or max(counts) <= 64)):
            return False, ('the gather path needs <= 32 channels at full resolution, <= 8 evaluations per patch and <= 64 tiles '
                           'of one axis over a voxel')
        if n_slots is not None and p.device.type == 'cuda':
            C = p._engine.feature_channels
            need = n_eval * n_slots * (int(np.prod(p._spec.patch)) * C * 2 + 2 * C * 4)
            free = torch.cuda.mem_get_info(p.device)[0]
            if need > 0.8 * free:
                return False, f'the kept patch activations need {need / 2 ** 30:.1f} GiB, {free / 2 ** 30:.1f} GiB are free'
        return True, ''

    last_mode = None                                # 'gather' | 'accumulate': the path the last step really took

    def _use_gather_all(self, slots, counts=None) -> bool:
        """The same path on every rank: a rank whose activations do not fit sends the whole group to the accumulate path
        (the two exchanges do not pair up).  mode='gather' raises - on EVERY rank, after the collective, so that no rank is
        left waiting in it - when one rank cannot."""
        local, why = self._gather_fits(None if slots is None else len(slots[2]), counts)
        flag_all = local
        if self.world > 1:
            flag = torch.tensor([1 if local else 0], dtype=torch.int32, device=self.p.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            flag_all = bool(int(flag.item()))
        if self.mode == 'gather' and not flag_all:
            raise NotImplementedError(why or 'another rank of the group cannot take the gather path')
        use = flag_all and self.mode != 'accumulate'
        self.last_mode = 'gather' if use else 'accumulate'
        return use

    def _slots(self, dec, origins):
        """(boundary, interior, slot_of) of this rank: own patches first (boundary, then interior), then the foreign ones."""
        if self._cur is not None and self._cur['dec'] is dec and self._cur['slots'] is not None:
            return self._cur['slots']
        out = self._slots_uncached(dec, origins)
        if self._cur is not None and self._cur['dec'] is dec:
            self._cur['slots'] = out
        return out

    def _slots_uncached(self, dec, origins):
        patch = self.p._spec.patch
        boundary, interior = dec.split_patches_for_features(self.rank, patch, origins)
        _, recvs = dec.feature_transfers(self.rank, patch, origins)
        slot_of = {pid: i for i, pid in enumerate(boundary + interior)}
        for _, pid, _ in recvs:
            slot_of.setdefault(pid, len(slot_of))
        return boundary, interior, slot_of

    def _features_fold(self, x, dec, origins, opts, fold, slots=None):
        """(feat [n_eval, n_slots, PD, PH, PW, C], fss, slot table over all patches, n_slots) with this rank's own patches
        and the foreign regions its owned box needs; None on an idle rank."""
        p, eng, patch = self.p, self.p._engine, self.p._spec.patch
        n_patches = int(origins.shape[0])
        table = np.full(n_patches, -1, np.int32)
        flips = self._flips()
        ph = self.phases
        if dec.owned[self.rank] is None:
            FeatureExchange(torch.empty((len(flips), 0, *patch, 1)), torch.empty((len(flips), 0, 2, 1)), dec, self.rank, patch, origins,
                            {}, self.group, flips).start().finish()
            return None
        boundary, interior, slot_of = slots if slots is not None else self._slots(dec, origins)
        C, n_slots = eng.feature_channels, len(slot_of)
        # no zero fill: own slots are written whole by fnn_patch_features, a foreign slot exactly where it overlaps this
        # rank's owned box - the only part of it fnn_gather_box reads
        feat = torch.empty((len(flips), n_slots, *patch, C), dtype=torch.half, device=p.device)
        fss = torch.empty((len(flips), n_slots, 2, C), dtype=torch.float32, device=p.device)
        xplan = None
        if self._cur is not None and self._cur['dec'] is dec:
            xkey = (tuple(flips), C, n_slots)
            xplan = self._cur['xplan'].get(xkey)
            if xplan is None:
                xplan = self._cur['xplan'][xkey] = ExchangePlan(dec, self.rank, patch, origins, slot_of, flips, C, p.device, n_slots)
        fx = FeatureExchange(feat, fss, dec, self.rank, patch, origins, slot_of, self.group, flips, engine=eng, plan=xplan)
        t0 = self._tick()
        if boundary:
            eng.patch_features(x.data_ptr(), x.shape, opts, boundary, feat.data_ptr(), fss.data_ptr(), fold=fold, slot0=0, n_slots=n_slots)
        t1 = self._tick()
        fx.start()
        t2 = self._tick()
        if interior:
            eng.patch_features(x.data_ptr(), x.shape, opts, interior, feat.data_ptr(), fss.data_ptr(), fold=fold,
                               slot0=len(boundary), n_slots=n_slots)
        t3 = self._tick()
        fx.finish()
        t4 = self._tick()
        if ph is not None:
            ph['boundary_patches_ms'] += (t1 - t0) * 1e3
            ph['exchange_post_ms'] += (t2 - t1) * 1e3
            ph['interior_patches_ms'] += (t3 - t2) * 1e3
            ph['exchange_wait_ms'] += (t4 - t3) * 1e3
            ph['n_boundary_patches'] += len(boundary)
            ph['n_interior_patches'] += len(interior)
            ph['n_foreign_patches'] += n_slots - len(boundary) - len(interior)
            ph['exchange_bytes_sent'] += fx.bytes_sent
            ph['exchange_bytes_received'] += fx.bytes_received
        for pid, sl in slot_of.items():
            table[pid] = sl
        return feat, fss, table, n_slots

    # ---- per-phase wall times of one step (bench.py --gpus N, tools/sharded_phases.py): `phases` is None (off) or a dict
    # that the step fills; every boundary then synchronises the device, so a profiled step is slower than a timed one
    phases = None

    def start_phases(self):
        self.phases = {k: 0.0 for k in ('boundary_patches_ms', 'exchange_post_ms', 'interior_patches_ms', 'exchange_wait_ms',
                                        'gather_box_ms', 'assemble_all_gather_ms')}
        self.phases.update({k: 0 for k in ('n_boundary_patches', 'n_interior_patches', 'n_foreign_patches',
                                           'exchange_bytes_sent', 'exchange_bytes_received')})
        return self.phases

    def _tick(self):
        if self.phases is None:
            return 0.0
        import time
        if self.p.device.type == 'cuda':
            torch.cuda.synchronize(self.p.device)
        return time.perf_counter()

    def _plan(self, x):
        """(decomposition, patch origins, un-padded owned boxes) of a volume shape - cached: the Python geometry (every
        patch against every rank's box) is milliseconds per step at 8 ranks and does not depend on the voxels."""
        from . import capi
        p = self.p
        key = (tuple(int(v) for v in x.shape[1:]), float(p.tile_step_size), self.world,
               tuple(int(v) for v in p._spec.patch), int(p._spec.spatial_dims),
               int(p._spec.features[0]), int(p._spec.num_heads))      # the spec's content, not its id() (ids are reused)
        hit = self._plans.get(key)
        if hit is not None:
            self._plans.move_to_end(key)
        if hit is None:
            patch = p._spec.patch
            padded, pad_lo, origins = capi.plan_volume(patch[3 - p._spec.spatial_dims:], x.shape[1:], p.tile_step_size)
            steps = [sorted(set(int(v) for v in origins[:, d])) for d in range(3)]
            dec = Decomposition.build(patch, padded, steps, self.world)
            owns = [None if b is None else unpadded(b, pad_lo, x.shape[1:]) for b in dec.owned]
            hit = self._plans[key] = dict(dec=dec, origins=origins, owns=owns, counts=tile_cover(steps, patch), slots=None, xplan={})
            while len(self._plans) > self.PLAN_CACHE:
                self._plans.popitem(last=False)
        self._cur = hit
        return hit['dec'], hit['origins'], hit['owns']

    def _accumulate_fold(self, x, dec, origins, opts, fold):
        """This rank's accumulator box with every contribution to the part it owns (its own patches + the halos)."""
        p, eng, patch = self.p, self.p._engine, self.p._spec.patch
        box = dec.boxes[self.rank]
        if box is None:
            exchange_halos(torch.empty(0), dec, self.rank, self.group)
            return None
        acc_dtype = torch.float32 if p.accumulate_in == 'fp32' else torch.half
        dims = tuple(box[1][d] - box[0][d] for d in range(3))
        acc = torch.zeros((*dims, eng.accumulator_channels), dtype=acc_dtype, device=p.device)
        # patches that feed another rank's box first; their halos travel over xGMI while the interior computes
        boundary, interior = dec.split_patches(self.rank, patch, origins)
        hx = HaloExchange(acc, dec, self.rank, self.group)
        if boundary:
            eng.accumulate_patches(x.data_ptr(), x.shape, opts, boundary, box[0], box[1], acc.data_ptr(), fold=fold)
        hx.start()
        if interior:
            eng.accumulate_patches(x.data_ptr(), x.shape, opts, interior, box[0], box[1], acc.data_ptr(), fold=fold)
        hx.finish()
        return acc

    @torch.inference_mode()
    def predict_sliding_window_return_logits(self, input_image: torch.Tensor, out: Optional[torch.Tensor] = None,
                                             gather: bool = False, folds: Optional[Sequence[int]] = None):
        p = self.p
        p._check_input(input_image)
        eng = p._engine
        folds = [p._active_fold] if folds is None else list(folds)
        with torch.cuda.device(p.device):
            x = input_image.to(device=p.device, dtype=torch.float32).contiguous()
            dec, origins, owns = self._plan(x)
            opts = p._opts()                                  # accumulator dtype follows predictor.accumulate_in
            if out is None:
                out = torch.empty((p._spec.num_heads, *x.shape[1:]), dtype=torch.half, device=p.device)
            box, own = dec.boxes[self.rank], owns[self.rank]
            part = None
            slots = self._slots(dec, origins) if dec.owned[self.rank] is not None else None
            use_gather = self._use_gather_all(slots, self._cur['counts'])
            for i, f in enumerate(folds):
                if use_gather:
                    got = self._features_fold(x, dec, origins, opts, f, slots)
                    if own is None:
                        continue
                    t0 = self._tick()
                    eng.gather_box(got[0].data_ptr(), got[1].data_ptr(), got[2], x.shape, opts, own[0], own[1],
                                   logits_ptr=out.data_ptr(), fold=f, n_slots=got[3])
                    if self.phases is not None:
                        self.phases['gather_box_ms'] += (self._tick() - t0) * 1e3
                else:
                    acc = self._accumulate_fold(x, dec, origins, opts, f)
                    if own is None:
                        continue
                    eng.normalize_box(acc.data_ptr(), x.shape, opts, box[0], box[1], own[0], own[1], out.data_ptr())
                if len(folds) > 1:                            # fp16 sum over the folds, then / n (:494-500)
                    sl = (slice(None), *[slice(own[0][d], own[1][d]) for d in range(3)])
                    part = out[sl].clone() if i == 0 else part.add_(out[sl])
            if part is not None:
                out[(slice(None), *[slice(own[0][d], own[1][d]) for d in range(3)])] = part.div_(len(folds))
            if gather:
                t0 = self._tick()
                res = gather_owned_boxes(out, owns, self.rank, self.group)
                if self.phases is not None:
                    self.phases['assemble_all_gather_ms'] += (self._tick() - t0) * 1e3
                return res
        return out, own

    @torch.inference_mode()
    def predict_logits_from_preprocessed_data(self, data: torch.Tensor) -> torch.Tensor:
        """Ensemble mean over the loaded folds, assembled on every rank (device tensor)."""
        return self.predict_sliding_window_return_logits(data, gather=True, folds=range(self.p._n_folds))

    @torch.inference_mode()
    def predict_segmentation_from_preprocessed_data(self, data: torch.Tensor, gather: bool = True) -> torch.Tensor:
        """Label map [X, Y, Z] on every rank: the owner labels its box straight from its accumulators
        (``fnn_labels_box``), the slabs are gathered.  Several folds: ensemble logits of the owned box first."""
        p = self.p
        p._check_input(data)
        eng = p._engine
        order, u16 = p._label_rule()
        with torch.cuda.device(p.device):
            x = data.to(device=p.device, dtype=torch.float32).contiguous()
            eng.set_label_rule(order, uint16=u16)
            labels = torch.zeros(x.shape[1:], dtype=torch.int16 if u16 else torch.uint8, device=p.device)
            if p._n_folds > 1 or p._spec.num_heads > 63:          # (> 63 classes: the gather kernel forms the logits in passes over the heads)
                logits, own = self.predict_sliding_window_return_logits(x, folds=range(p._n_folds) if p._n_folds > 1 else None)
                if own is not None:
                    sl = tuple(slice(own[0][d], own[1][d]) for d in range(3))
                    labels[sl] = p.convert_logits_to_segmentation(logits[(slice(None), *sl)].contiguous()).to(labels.dtype)
                dec, origins, owns = self._plan(x)
            else:
                dec, origins, owns = self._plan(x)
                opts = p._opts()
                box, own = dec.boxes[self.rank], owns[self.rank]
                slots = self._slots(dec, origins) if dec.owned[self.rank] is not None else None
                if self._use_gather_all(slots, self._cur['counts']):
                    got = self._features_fold(x, dec, origins, opts, p._active_fold, slots)
                    if own is not None:
                        t0 = self._tick()
                        eng.gather_box(got[0].data_ptr(), got[1].data_ptr(), got[2], x.shape, opts, own[0], own[1],
                                       labels_ptr=labels.data_ptr(), fold=p._active_fold, n_slots=got[3])
                        if self.phases is not None:
                            self.phases['gather_box_ms'] += (self._tick() - t0) * 1e3
                else:
                    acc = self._accumulate_fold(x, dec, origins, opts, p._active_fold)
                    if own is not None:
                        eng.labels_box(acc.data_ptr(), x.shape, opts, box[0], box[1], own[0], own[1], labels.data_ptr())
            if gather:
                t0 = self._tick()
                gather_owned_boxes(labels, owns, self.rank, self.group)
                if self.phases is not None:
                    self.phases['assemble_all_gather_ms'] += (self._tick() - t0) * 1e3
            if u16:
                labels = labels.to(torch.int32) & 0xffff
        return labels
