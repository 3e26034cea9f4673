"""CPU restatement of the reference's resampling step (test oracle).

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.  **Parity unpinned**: the reference resamples with
``skimage.transform.resize`` (preprocessing/resampling/default_resampling.py:9,150,176-188), a third-party
package that is neither vendored nor installed here (and un-pinned in distillation/setup.py), so the reference's
own resampler cannot produce vectors in this container.  This file restates

* ``resample_data_or_seg`` for images / logits (``is_seg=False``; default_resampling.py:113-196) and
  ``determine_do_sep_z_and_axis`` / ``compute_new_shape`` (:14-71) line by line, and
* skimage's published ``resize(image, shape, order, mode='edge', anti_aliasing=False)`` (scikit-image >= 0.19,
  transform/_warps.py): ``scipy.ndimage.zoom(image, out/in, order, mode='nearest', grid_mode=True)`` on the
  float64 image followed by a clip to the input's [min, max] - scipy IS installed and is what skimage calls.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np
from scipy import ndimage as ndi

ANISO_THRESHOLD = 3           # nnunetv2/configuration.py


def compute_new_shape(old_shape, old_spacing, new_spacing):
    assert len(old_spacing) == len(old_shape) and len(old_shape) == len(new_spacing)
    return [int(round(i / j * k)) for i, j, k in zip(old_spacing, new_spacing, old_shape)]


def get_do_separate_z(spacing, anisotropy_threshold=ANISO_THRESHOLD) -> bool:
    return (np.max(spacing) / np.min(spacing)) > anisotropy_threshold


def get_lowres_axis(new_spacing):
    return np.where(max(new_spacing) / np.array(new_spacing) == 1)[0]


def determine_do_sep_z_and_axis(force_separate_z: Optional[bool], current_spacing, new_spacing,
                                separate_z_anisotropy_threshold: float = ANISO_THRESHOLD) -> Tuple[bool, Optional[int]]:
    """default_resampling.py:34-71."""
    if force_separate_z is not None:
        do_separate_z = force_separate_z
        axis = get_lowres_axis(current_spacing) if force_separate_z else None
    else:
        if get_do_separate_z(current_spacing, separate_z_anisotropy_threshold):
            do_separate_z, axis = True, get_lowres_axis(current_spacing)
        elif get_do_separate_z(new_spacing, separate_z_anisotropy_threshold):
            do_separate_z, axis = True, get_lowres_axis(new_spacing)
        else:
            do_separate_z, axis = False, None
    if axis is not None:
        if len(axis) == 3 or len(axis) == 2:
            do_separate_z, axis = False, None
        else:
            axis = int(axis[0])
    return do_separate_z, axis


def skimage_resize(image: np.ndarray, output_shape: Sequence[int], order: int) -> np.ndarray:
    """``skimage.transform.resize(image, output_shape, order, mode='edge', anti_aliasing=False)`` for a float64
    image (clip=True, preserve_range irrelevant for floats)."""
    image = np.asarray(image, dtype=np.float64)
    zoom = [o / i for o, i in zip(output_shape, image.shape)]
    out = ndi.zoom(image, zoom, order=order, mode='nearest', grid_mode=True)
    assert out.shape == tuple(output_shape)
    np.clip(out, np.min(image), np.max(image), out=out)
    return out


def resample_data(data: np.ndarray, new_shape: Sequence[int], axis: Optional[int] = None, order: int = 3,
                  do_separate_z: bool = False, order_z: int = 0) -> np.ndarray:
    """``resample_data_or_seg(data, new_shape, is_seg=False, ...)`` (default_resampling.py:113-196)."""
    assert data.ndim == 4 and len(new_shape) == 3
    shape = np.array(data[0].shape)
    new_shape = np.array(new_shape)
    dtype_out = data.dtype
    out = np.zeros((data.shape[0], *new_shape), dtype=dtype_out)
    if not np.any(shape != new_shape):
        return data
    data = data.astype(float, copy=False)
    if do_separate_z:
        assert axis is not None
        new_shape_2d = [new_shape[i] for i in range(3) if i != axis]
        for c in range(data.shape[0]):
            tmp = list(new_shape)
            tmp[axis] = shape[axis]
            here = np.zeros(tmp)
            for s in range(shape[axis]):
                sl = [slice(None)] * 3
                sl[axis] = s
                here[tuple(sl)] = skimage_resize(data[c][tuple(sl)], new_shape_2d, order)
            if shape[axis] != new_shape[axis]:
                rows, cols, dim = new_shape
                orows, ocols, odim = here.shape
                mr, mc, md = np.mgrid[:rows, :cols, :dim]
                coords = np.array([float(orows) / rows * (mr + 0.5) - 0.5, float(ocols) / cols * (mc + 0.5) - 0.5,
                                   float(odim) / dim * (md + 0.5) - 0.5])
                out[c] = ndi.map_coordinates(here, coords, order=order_z, mode='nearest')
            else:
                out[c] = here
    else:
        for c in range(data.shape[0]):
            out[c] = skimage_resize(data[c], new_shape, order)
    return out


def resize_segmentation(segmentation: np.ndarray, new_shape: Sequence[int], order: int = 3) -> np.ndarray:
    """``batchgenerators.augmentations.utils.resize_segmentation`` (third-party, un-vendored, absent here: published
    algorithm restated; the reference calls it at default_resampling.py:143-146 for ``is_seg=True``): order 0 resizes the
    label image itself; otherwise every label's mask is resized with `order` and the voxels where it reaches 0.5 take
    the label, labels in ascending order (a later label overwrites an earlier one; voxels no mask claims stay 0)."""
    tpe = segmentation.dtype
    assert len(segmentation.shape) == len(new_shape)
    if order == 0:
        return skimage_resize(segmentation.astype(float), new_shape, 0).astype(tpe)
    out = np.zeros(tuple(new_shape), dtype=tpe)
    for c in np.unique(segmentation):
        out[skimage_resize((segmentation == c).astype(float), new_shape, order) >= 0.5] = c
    return out


def resample_seg(seg: np.ndarray, new_shape: Sequence[int], axis: Optional[int] = None, order: int = 1,
                 do_separate_z: bool = False, order_z: int = 0) -> np.ndarray:
    """``resample_data_or_seg(seg, new_shape, is_seg=True, ...)`` (default_resampling.py:113-196) for ``order_z == 0``
    (the only value the reference's plans use, default_experiment_planner.py:170-181): per-slice
    ``resize_segmentation`` + nearest-neighbour along the anisotropic axis, or one 3-D ``resize_segmentation``."""
    assert seg.ndim == 4 and len(new_shape) == 3 and order_z == 0
    shape = np.array(seg[0].shape)
    new_shape = np.array(new_shape)
    if not np.any(shape != new_shape):
        return seg
    out = np.zeros((seg.shape[0], *new_shape), dtype=seg.dtype)
    data = seg.astype(float, copy=False)
    for c in range(seg.shape[0]):
        if do_separate_z:
            new_shape_2d = [new_shape[i] for i in range(3) if i != axis]
            tmp = list(new_shape)
            tmp[axis] = shape[axis]
            here = np.zeros(tmp)
            for s in range(shape[axis]):
                sl = [slice(None)] * 3
                sl[axis] = s
                here[tuple(sl)] = resize_segmentation(data[c][tuple(sl)], new_shape_2d, order)
            if shape[axis] != new_shape[axis]:
                rows, cols, dim = new_shape
                orows, ocols, odim = here.shape
                mr, mc, md = np.mgrid[:rows, :cols, :dim]
                coords = np.array([float(orows) / rows * (mr + 0.5) - 0.5, float(ocols) / cols * (mc + 0.5) - 0.5,
                                   float(odim) / dim * (md + 0.5) - 0.5])
                out[c] = ndi.map_coordinates(here, coords, order=0, mode='nearest')
            else:
                out[c] = here
        else:
            out[c] = resize_segmentation(data[c], new_shape, order)
    return out
