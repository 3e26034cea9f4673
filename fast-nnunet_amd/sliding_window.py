"""Host-side mirror of the reference's ``sliding_window_prediction`` module.

Same function names and argument meaning as
``distillation/nnunetv2/inference/sliding_window_prediction.py`` so callers can
switch imports.  ``compute_steps_for_sliding_window`` runs in the C-ABI library
(integer logic, no GPU needed); ``compute_gaussian`` stays on the host in
Python because its definition *is* a ``scipy.ndimage.gaussian_filter`` call
followed by torch's float64->fp16 cast, and the engine must be fed exactly
those bits (it is computed once per predictor and uploaded).
"""
from __future__ import annotations

from functools import lru_cache
from typing import List, Sequence, Tuple, Union

import numpy as np
import torch
from scipy.ndimage import gaussian_filter

from . import capi


@lru_cache(maxsize=2)
def compute_gaussian(tile_size: Union[Tuple[int, ...], List[int]], sigma_scale: float = 1. / 8,
                     value_scaling_factor: float = 1, dtype=torch.float16, device=torch.device('cpu')) -> torch.Tensor:
    """Gaussian importance map (reference: sliding_window_prediction.py:10-27)."""
    tmp = np.zeros(tile_size)
    tmp[tuple(i // 2 for i in tile_size)] = 1
    blurred = torch.from_numpy(gaussian_filter(tmp, [i * sigma_scale for i in tile_size], 0, mode='constant', cval=0))
    blurred /= (torch.max(blurred) / value_scaling_factor)
    g = blurred.to(device=device, dtype=dtype)
    zero = g == 0
    g[zero] = torch.min(g[~zero])
    return g


def compute_steps_for_sliding_window(image_size: Sequence[int], tile_size: Sequence[int],
                                     tile_step_size: float) -> List[List[int]]:
    """Tile start positions per axis (reference: sliding_window_prediction.py:30-54)."""
    assert 0 < tile_step_size <= 1, 'step_size must be larger than 0 and smaller or equal to 1'
    return [capi.compute_steps(i, t, tile_step_size) for i, t in zip(image_size, tile_size)]
