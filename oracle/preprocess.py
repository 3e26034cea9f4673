"""CPU restatement of the steps around the sliding window that touch whole volumes (test oracle).

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.  Restates, in numpy and in the reference's own order of
operations (paths relative to ``/root/reference/distillation/nnunetv2/``):

* ``DefaultPreprocessor.run_case_npy`` up to the resampling call
  (preprocessing/preprocessors/default_preprocessor.py:45-93): float32 copy, ``transpose_forward``,
  ``crop_to_nonzero`` (preprocessing/cropping/cropping.py:7-39), per-channel intensity normalisation
  (preprocessing/normalization/default_normalization_schemes.py:27-109);
* the label half of ``convert_predicted_logits_to_segmentation_with_correct_shape``
  (inference/export_prediction.py:43-53): dtype rule, revert cropping, ``transpose_backward``.

``acvl_utils.get_bbox_from_mask`` is an absent third-party helper: its published behaviour (first / last index
with any foreground per axis, the full extent for an empty mask) is restated here.  Resampling
(preprocessing/resampling/default_resampling.py, skimage) is NOT restated: parity unpinned, not built.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
from scipy.ndimage import binary_fill_holes


def nonzero_bbox(data: np.ndarray) -> List[List[int]]:
    """Bounding box of ``create_nonzero_mask(data)`` (cropping.py:7-17, :29-30).  ``binary_fill_holes`` only
    fills enclosed background, so it never changes the box and is skipped."""
    assert data.ndim == 4, 'data must have shape (C, X, Y, Z)'
    mask = data[0] != 0
    for c in range(1, data.shape[0]):
        mask |= data[c] != 0
    out = []
    for ax in range(3):
        idx = np.where(np.any(mask, axis=tuple(a for a in range(3) if a != ax)))[0]
        out.append([int(idx[0]), int(idx[-1]) + 1] if idx.size else [0, int(mask.shape[ax])])
    return out


def filled_nonzero_mask(data: np.ndarray) -> np.ndarray:
    """``create_nonzero_mask`` (cropping.py:7-17): any channel non-zero, holes filled."""
    mask = data[0] != 0
    for c in range(1, data.shape[0]):
        mask |= data[c] != 0
    return binary_fill_holes(mask)


def normalize_channel(image: np.ndarray, scheme: str, props: Optional[dict], mask: Optional[np.ndarray] = None) -> np.ndarray:
    """One channel, in place semantics of the reference's ``ImageNormalization.run``; ``mask`` = the filled non-zero
    mask of the cropped image when ``use_mask_for_norm`` is set for the channel (only ZScore looks at it)."""
    image = image.astype(np.float32, copy=True)
    if scheme == 'CTNormalization':                                                      # :53-67
        np.clip(image, props['percentile_00_5'], props['percentile_99_5'], out=image)
        image -= props['mean']
        image /= max(props['std'], 1e-8)
    elif scheme == 'ZScoreNormalization' and mask is not None:                           # :36-44, use_mask_for_norm
        mean = image[mask].mean()
        std = image[mask].std()
        image[mask] = (image[mask] - mean) / (max(std, 1e-8))
    elif scheme == 'ZScoreNormalization':                                                # :45-49
        mean = image.mean()
        std = image.std()
        image -= mean
        image /= (max(std, 1e-8))
    elif scheme == 'NoNormalization':                                                    # :70-74
        pass
    elif scheme == 'RescaleTo01Normalization':                                           # :77-84
        image -= image.min()
        image /= np.clip(image.max(), a_min=1e-8, a_max=None)
    elif scheme == 'RGBTo01Normalization':                                               # :87-98
        assert image.min() >= 0 and image.max() <= 255
        image /= 255.
    else:
        raise RuntimeError(f"Unable to locate class '{scheme}' for normalization")
    return image


def preprocess_case(data: np.ndarray, transpose_forward: Sequence[int], schemes: Sequence[str],
                    intensity_props: Dict[str, dict], use_mask: Optional[Sequence[bool]] = None
                    ) -> Tuple[np.ndarray, List[List[int]], Tuple[int, ...]]:
    """-> (cropped + normalised float32 data, bbox_used_for_cropping, shape_before_cropping)."""
    data = data.astype(np.float32)
    data = data.transpose([0, *[i + 1 for i in transpose_forward]])
    shape_before_cropping = tuple(data.shape[1:])
    bbox = nonzero_bbox(data)
    crop = (slice(None), *[slice(lo, hi) for lo, hi in bbox])
    mask = filled_nonzero_mask(data)[crop[1:]] if use_mask is not None and any(use_mask) else None
    data = data[crop]
    out = np.empty(data.shape, np.float32)
    for c in range(data.shape[0]):
        out[c] = normalize_channel(data[c], schemes[c], intensity_props.get(str(c)),
                                   mask if (use_mask is not None and use_mask[c]) else None)
    return out, bbox, shape_before_cropping


def revert_labels(seg: np.ndarray, bbox: Sequence[Sequence[int]], shape_before_cropping: Sequence[int],
                  transpose_backward: Sequence[int], n_foreground_labels: int) -> np.ndarray:
    """export_prediction.py:43-53: zeros of the uncropped shape (uint8 below 255 foreground labels, else uint16),
    the segmentation inserted at the crop box, axes transposed back."""
    full = np.zeros(tuple(shape_before_cropping), dtype=np.uint8 if n_foreground_labels < 255 else np.uint16)
    full[tuple(slice(lo, hi) for lo, hi in bbox)] = seg
    return full.transpose(list(transpose_backward))


def export_with_probabilities(logits: np.ndarray, bbox: Sequence[Sequence[int]], shape_before_cropping: Sequence[int],
                              transpose_backward: Sequence[int], n_foreground_labels: int,
                              regions_class_order: Optional[Sequence[int]] = None):
    """export_prediction.py:36-70 with ``return_probabilities=True`` on logits of the cropped grid (after the
    resampling step): ``apply_inference_nonlin`` (label_handling.py:125-139: fp32 softmax over the heads, sigmoid
    for regions), the label rule ON THE PROBABILITIES (``convert_probabilities_to_segmentation``,
    label_handling.py:163-181), revert cropping of the labels (zeros outside the box) and of the probabilities
    (``revert_cropping_on_probabilities``, label_handling.py:197-221: background probability 1 outside the box for
    plain labels, all zeros for regions), transposes back.  -> (segmentation, float32 probabilities)."""
    import torch
    lg = torch.from_numpy(np.asarray(logits)).float()
    if regions_class_order is None:
        probs = torch.softmax(lg, 0)
        seg = probs.argmax(0).numpy()
    else:
        probs = torch.sigmoid(lg)
        seg = np.zeros(probs.shape[1:], np.int64)
        for i, c in enumerate(regions_class_order):
            seg[(probs[i] > 0.5).numpy()] = c
    probs = probs.numpy()
    seg_full = revert_labels(seg, bbox, shape_before_cropping, transpose_backward, n_foreground_labels)
    full = np.zeros((probs.shape[0], *shape_before_cropping), np.float32)
    if regions_class_order is None:
        full[0] = 1
    full[(slice(None), *[slice(lo, hi) for lo, hi in bbox])] = probs
    return seg_full, full.transpose([0] + [i + 1 for i in transpose_backward])
