"""CPU restatement of the reference's sliding-window predictor (test oracle).

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.  Every function names the
reference lines it restates (paths relative to
``/root/reference/distillation/nnunetv2/inference/``).

The numerical contract that is reproduced bit-for-bit:

* tile starts: float64 arithmetic, round-half-to-even;
* Gaussian map: float64 ``scipy.ndimage.gaussian_filter`` of a unit impulse,
  rescaled to max 10, cast to fp16, zeros replaced by the smallest non-zero;
* accumulation: fp16 accumulators, each visit computes in fp32 and rounds to
  fp16 once (torch's mixed ``half += float`` semantics); the weight sum is an
  fp16 + fp16 add; the final divide is half / half -> half;
* patch visit order: x slowest, z fastest.
"""
from __future__ import annotations

import itertools
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch
from scipy.ndimage import gaussian_filter


# --------------------------------------------------------------------------
# geometry
# --------------------------------------------------------------------------
def tile_starts(image_size: Sequence[int], patch: Sequence[int], step: float) -> List[List[int]]:
    """Start coordinates of the tiles along every axis.

    Restates ``compute_steps_for_sliding_window``
    (sliding_window_prediction.py:30-54): at most ``patch*step`` apart, evenly
    re-spaced so that the last tile ends exactly at the image border.
    """
    if not (0 < step <= 1):
        raise AssertionError('step_size must be larger than 0 and smaller or equal to 1')
    out = []
    for size, p in zip(image_size, patch):
        target = p * step                                  # float
        n = int(np.ceil((size - p) / target)) + 1
        span = size - p
        if n > 1:
            stride = span / (n - 1)
            out.append([int(np.round(stride * i)) for i in range(n)])
        else:
            out.append([0])
    return out


def pad_to_patch(shape_sp: Sequence[int], patch: Sequence[int]) -> Tuple[List[Tuple[int, int]], Tuple[slice, ...]]:
    """Per-axis (below, above) zero padding that grows an image to >= patch.

    Restates the way the reference calls ``acvl_utils.pad_nd_image`` at
    predict_from_raw_data.py:657-659 (constant 0, no divisibility
    constraint): the missing voxels are split evenly, the odd one goes to the
    high side.  Returns the pads and the slicer that undoes them.
    """
    pads, undo = [], []
    lead = len(shape_sp) - len(patch)          # 2-D configuration: new_shape applies to the trailing axes only
    for ax, size in enumerate(shape_sp):
        p = patch[ax - lead] if ax >= lead else size
        missing = max(p, size) - size
        lo = missing // 2
        hi = missing // 2 + missing % 2
        pads.append((lo, hi))
        undo.append(slice(lo, lo + size))
    return pads, tuple(undo)


def patch_slicers(shape_sp: Sequence[int], patch: Sequence[int], step: float) -> List[Tuple[slice, ...]]:
    """All patch windows in the reference's visit order (x, then y, then z).

    Restates ``_internal_get_sliding_window_slicers``
    (predict_from_raw_data.py:506-538), both branches.
    """
    out = []
    if len(patch) < len(shape_sp):
        # 2-D configuration (:508-524): every slice of the first axis, tiles over the other two
        assert len(patch) == len(shape_sp) - 1, 'if tile_size has less entries than image_size, len(tile_size) ' \
                                                'must be one shorter than len(image_size)'
        starts = tile_starts(shape_sp[1:], patch, step)
        for d in range(shape_sp[0]):
            for sx in starts[0]:
                for sy in starts[1]:
                    out.append((slice(None), d, slice(sx, sx + patch[0]), slice(sy, sy + patch[1])))
        return out
    starts = tile_starts(shape_sp, patch, step)
    for sx in starts[0]:
        for sy in starts[1]:
            for sz in starts[2]:
                out.append((slice(None),
                            slice(sx, sx + patch[0]),
                            slice(sy, sy + patch[1]),
                            slice(sz, sz + patch[2])))
    return out


# --------------------------------------------------------------------------
# Gaussian importance map
# --------------------------------------------------------------------------
def gaussian_weight(patch: Sequence[int], sigma_scale: float = 1. / 8, peak: float = 10.0) -> torch.Tensor:
    """fp16 Gaussian importance map for one patch.

    Restates ``compute_gaussian`` (sliding_window_prediction.py:10-27) as the
    predictor calls it (``value_scaling_factor=10``,
    predict_from_raw_data.py:592-595).
    """
    impulse = np.zeros(tuple(patch), dtype=np.float64)
    impulse[tuple(p // 2 for p in patch)] = 1.0
    g = gaussian_filter(impulse, [p * sigma_scale for p in patch], 0, mode='constant', cval=0)
    # torch (not numpy) does the rescale and the float64 -> fp16 cast so that the
    # rounding is the one the reference gets (the two differ in rare ties).
    gt = torch.from_numpy(g)
    gt = gt / (gt.max() / peak)
    g16 = gt.to(torch.float16)
    zero = g16 == 0
    g16[zero] = g16[~zero].min()
    return g16


# --------------------------------------------------------------------------
# test-time augmentation
# --------------------------------------------------------------------------
def mirror_combinations(mirror_axes: Optional[Sequence[int]]) -> List[Tuple[int, ...]]:
    """Non-empty subsets of the spatial mirror axes in the reference's order
    (by size, then lexicographic; predict_from_raw_data.py:551-553)."""
    if mirror_axes is None:
        return []
    axes = list(mirror_axes)
    return [c for k in range(len(axes)) for c in itertools.combinations(axes, k + 1)]


def predict_with_mirroring(net: Callable[[torch.Tensor], torch.Tensor], x: torch.Tensor,
                           mirror_axes: Optional[Sequence[int]]) -> torch.Tensor:
    """``net(x)`` averaged with its mirrored evaluations.

    Restates ``_internal_maybe_mirror_and_predict``
    (predict_from_raw_data.py:541-557).  ``x`` is ``[1, C, X, Y, Z]``,
    ``mirror_axes`` index the spatial axes (0..2).
    """
    pred = net(x)
    combos = mirror_combinations(mirror_axes)
    if mirror_axes is not None:
        assert max(mirror_axes) <= x.ndim - 3, 'mirror_axes does not match the dimension of the input!'
        for c in combos:
            dims = [a + 2 for a in c]
            pred = pred + torch.flip(net(torch.flip(x, dims)), dims)
        pred = pred / (len(combos) + 1)
    return pred


# --------------------------------------------------------------------------
# the hot loop
# --------------------------------------------------------------------------
@torch.inference_mode()
def sliding_window_logits(net: Callable[[torch.Tensor], torch.Tensor], image: torch.Tensor,
                          patch: Sequence[int], num_heads: int, step: float = 0.5,
                          use_gaussian: bool = True, mirror_axes: Optional[Sequence[int]] = None,
                          accum: str = 'fp16') -> torch.Tensor:
    """Logits ``[heads, X, Y, Z]`` for a preprocessed image ``[C, X, Y, Z]``.

    Restates ``predict_sliding_window_return_logits`` +
    ``_internal_predict_sliding_window_return_logits``
    (predict_from_raw_data.py:560-680) on the CPU path (no autocast, fp32
    network).

    ``accum='fp16'`` reproduces the reference's half accumulators bit for
    bit; ``accum='fp32'`` is the exact blend used to measure how far the fp16
    accumulators themselves are from the truth and
    returns fp32.
    """
    assert isinstance(image, torch.Tensor)
    assert image.ndim == 4, 'input_image must be a 4D np.ndarray or torch.Tensor (c, x, y, z)'
    pads, undo = pad_to_patch(image.shape[1:], patch)
    flat = [v for lo_hi in reversed(pads) for v in lo_hi]
    data = torch.nn.functional.pad(image, flat, mode='constant', value=0) if any(flat) else image
    slicers = patch_slicers(data.shape[1:], patch, step)

    acc_dtype = torch.half if accum == 'fp16' else torch.float32
    acc = torch.zeros((num_heads, *data.shape[1:]), dtype=acc_dtype)
    wsum = torch.zeros(data.shape[1:], dtype=acc_dtype)
    if use_gaussian:
        g = gaussian_weight(tuple(patch))
        if accum != 'fp16':
            g = g.float()
    else:
        g = 1

    for sl in slicers:
        x = data[sl][None].contiguous()
        pred = predict_with_mirroring(net, x, mirror_axes)[0]
        if use_gaussian:
            pred = pred * g                 # fp32 * fp16 -> fp32, as `pred *= gaussian`
        acc[sl] += pred                     # half += float: fp32 add, one RNE to half
        wsum[sl[1:]] += g
    out = acc / wsum
    if torch.any(torch.isinf(out)):
        raise RuntimeError('Encountered inf in predicted array. Aborting... If this problem persists, '
                           'reduce value_scaling_factor in compute_gaussian or increase the dtype of '
                           'predicted_logits to fp32')
    return out[(slice(None), *undo)]


@torch.inference_mode()
def sliding_window_logits_box(net: Callable[[torch.Tensor], torch.Tensor], image: torch.Tensor,
                              patch: Sequence[int], num_heads: int, box: Sequence[Tuple[int, int]],
                              step: float = 0.5, use_gaussian: bool = True,
                              mirror_axes: Optional[Sequence[int]] = None, accum: str = 'fp16') -> torch.Tensor:
    """``sliding_window_logits(...)[:, box]`` computed from only the patches that touch ``box``.

    The same statements as ``sliding_window_logits`` (predict_from_raw_data.py:560-680); patches that do not
    intersect ``box = ((x0, x1), (y0, y1), (z0, z1))`` (coordinates of the padded image = of the image when no
    axis is smaller than the patch) are skipped - they never touch a voxel of the box, and the patches that do
    are visited in the reference's order, so every value inside the box goes through the same arithmetic.  The
    accumulators cover only the hull of the visited patches.  For full-size volumes whose complete
    evaluation would take the CPU half an hour."""
    assert image.ndim == 4 and all(s >= p for s, p in zip(image.shape[1:], patch)), 'box form: no padding'
    slicers = [sl for sl in patch_slicers(image.shape[1:], patch, step)
               if all(s.start < hi and s.stop > lo for s, (lo, hi) in zip(sl[1:], box))]
    lo = [min(sl[1 + a].start for sl in slicers) for a in range(3)]
    hi = [max(sl[1 + a].stop for sl in slicers) for a in range(3)]
    acc_dtype = torch.half if accum == 'fp16' else torch.float32
    acc = torch.zeros((num_heads, *[h - l for l, h in zip(lo, hi)]), dtype=acc_dtype)
    wsum = torch.zeros(acc.shape[1:], dtype=acc_dtype)
    g = gaussian_weight(tuple(patch)) if use_gaussian else 1
    if use_gaussian and accum != 'fp16':
        g = g.float()
    for sl in slicers:
        x = image[sl][None].contiguous()
        pred = predict_with_mirroring(net, x, mirror_axes)[0]
        if use_gaussian:
            pred = pred * g
        rel = (slice(None), *[slice(s.start - l, s.stop - l) for s, l in zip(sl[1:], lo)])
        acc[rel] += pred
        wsum[rel[1:]] += g
    inner = tuple(slice(b0 - l, b1 - l) for (b0, b1), l in zip(box, lo))
    out = acc[(slice(None), *inner)] / wsum[inner]
    return out, len(slicers)


@torch.inference_mode()
def ensemble_logits(nets: Sequence[Callable[[torch.Tensor], torch.Tensor]], image: torch.Tensor,
                    patch: Sequence[int], num_heads: int, **kw) -> torch.Tensor:
    """Mean of the per-fold sliding-window logits.

    Restates ``predict_logits_from_preprocessed_data``
    (predict_from_raw_data.py:471-504): folds are summed in the accumulator
    dtype on the host and divided by the fold count.
    """
    total = None
    for net in nets:
        cur = sliding_window_logits(net, image, patch, num_heads, **kw)
        total = cur if total is None else total.add_(cur)
    if len(nets) > 1:
        total /= len(nets)
    return total


def logits_to_labels(logits: torch.Tensor, regions_class_order: Optional[Sequence[int]] = None) -> torch.Tensor:
    """Label map from logits.

    Restates ``LabelManager.convert_logits_to_segmentation``
    (utilities/label_handling/label_handling.py:144-195): argmax over heads
    for plain labels; for region training sigmoid > 0.5 painted in
    ``regions_class_order``.
    """
    if regions_class_order is None:
        return torch.from_numpy(logits.float().numpy().argmax(0))
    prob = torch.sigmoid(logits.float())
    seg = torch.zeros(logits.shape[1:], dtype=torch.int16)
    for i, c in enumerate(regions_class_order):
        seg[prob[i] > 0.5] = c
    return seg
