"""Import shims that let the *reference's own* predictor run in this container.

Used ONLY by ``tests/golden/make_golden.py`` (which runs only where
``/root/reference`` exists).  The reference's hot-path modules import a few
third-party packages that are not installed here.  Two kinds of shim:

* small real implementations, written for this repo, of the handful of helper
  functions the path actually calls (``batchgenerators`` file helpers,
  ``acvl_utils`` ``pad_nd_image``);
* ``MagicMock`` stand-ins for packages that are only touched at import time
  on this path (``dynamic_network_architectures``, ``SimpleITK``, ...).

Nothing here is reference code and none of it ships to the GPU box's tests.
"""
from __future__ import annotations

import importlib.abc
import importlib.machinery
import json
import os
import pickle
import sys
import types
from typing import List, Tuple, Union  # noqa: F401  (re-exported through `import *`)
from unittest.mock import MagicMock

import numpy as np
import torch

REFERENCE_ROOT = '/root/reference/distillation'


# ---- batchgenerators.utilities.file_and_folder_operations ------------------
def _ffo_module():
    m = types.ModuleType('batchgenerators.utilities.file_and_folder_operations')
    join, isfile, isdir = os.path.join, os.path.isfile, os.path.isdir

    def load_json(path):
        with open(path) as f:
            return json.load(f)

    def save_json(obj, path, indent=4, sort_keys=True):
        with open(path, 'w') as f:
            json.dump(obj, f, indent=indent, sort_keys=sort_keys)

    def load_pickle(path, mode='rb'):
        with open(path, mode) as f:
            return pickle.load(f)

    def save_pickle(obj, path, mode='wb'):
        with open(path, mode) as f:
            pickle.dump(obj, f)

    def maybe_mkdir_p(d):
        os.makedirs(d, exist_ok=True)

    def _listing(folder, want_dir, join_=True, prefix=None, suffix=None, sort=True):
        res = []
        for e in os.listdir(folder):
            full = os.path.join(folder, e)
            if (os.path.isdir(full) if want_dir else os.path.isfile(full)) \
                    and (prefix is None or e.startswith(prefix)) and (suffix is None or e.endswith(suffix)):
                res.append(full if join_ else e)
        return sorted(res) if sort else res

    def subdirs(folder, join=True, prefix=None, suffix=None, sort=True):
        return _listing(folder, True, join, prefix, suffix, sort)

    def subfiles(folder, join=True, prefix=None, suffix=None, sort=True):
        return _listing(folder, False, join, prefix, suffix, sort)

    for k, v in dict(os=os, List=List, Tuple=Tuple, Union=Union, join=join, isfile=isfile, isdir=isdir,
                     load_json=load_json, save_json=save_json, load_pickle=load_pickle, save_pickle=save_pickle,
                     write_pickle=save_pickle, maybe_mkdir_p=maybe_mkdir_p, subdirs=subdirs, subfiles=subfiles,
                     np=np).items():
        setattr(m, k, v)
    m.__all__ = [k for k in vars(m) if not k.startswith('_')]
    return m


# ---- acvl_utils.cropping_and_padding.padding.pad_nd_image -------------------
def pad_nd_image(image, new_shape=None, mode='constant', kwargs=None, return_slicer=False,
                 shape_must_be_divisible_by=None):
    """Own implementation of the published behaviour of acvl_utils' helper:
    centre the image in ``max(new_shape, shape)`` over the trailing axes; the
    odd voxel of an uneven pad goes to the high side."""
    kwargs = kwargs or {}
    old = np.array(image.shape)
    if new_shape is None:
        new_shape = old
    new_shape = list(new_shape)
    lead = len(old) - len(new_shape)
    target = np.array(list(old[:lead]) + [max(a, b) for a, b in zip(new_shape, old[lead:])])
    if shape_must_be_divisible_by is not None:
        div = shape_must_be_divisible_by
        if not isinstance(div, (list, tuple, np.ndarray)):
            div = [div] * len(new_shape)
        div = [1] * lead + list(div)
        target = np.array([t if t % d == 0 else t + d - t % d for t, d in zip(target, div)])
    diff = target - old
    below = diff // 2
    above = diff // 2 + diff % 2
    if isinstance(image, torch.Tensor):
        flat = [int(v) for b, a in zip(below[::-1], above[::-1]) for v in (b, a)]
        res = torch.nn.functional.pad(image, flat, mode=mode, **kwargs) if diff.any() else image
    else:
        res = np.pad(image, [(int(b), int(a)) for b, a in zip(below, above)], mode,
                     **({'constant_values': kwargs['value']} if 'value' in kwargs else kwargs)) if diff.any() else image
    if not return_slicer:
        return res
    slicer = tuple(slice(int(b), int(b) + int(o)) for b, o in zip(below, old))
    return res, slicer


class _MockFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    ROOTS = ('dynamic_network_architectures', 'SimpleITK', 'nibabel', 'blosc2', 'tifffile', 'skimage',
             'batchgeneratorsv2', 'batchgenerators', 'acvl_utils', 'nnunetv2.imageio', 'matplotlib', 'seaborn',
             'tqdm_unused')

    def find_spec(self, name, path=None, target=None):
        if name in sys.modules:
            return None
        if any(name == r or name.startswith(r + '.') for r in self.ROOTS):
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = MagicMock(name=spec.name)
        m.__name__ = spec.name
        m.__path__ = []
        m.__spec__ = spec
        m.__all__ = []
        return m

    def exec_module(self, module):
        pass


def install():
    if getattr(install, 'done', False):
        return
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError('the reference tree is not present; golden vectors can only be regenerated '
                           'in the build container')
    real = {
        'batchgenerators.utilities.file_and_folder_operations': _ffo_module(),
    }
    dl = types.ModuleType('batchgenerators.dataloading.data_loader')
    dl.DataLoader = type('DataLoader', (), {'__init__': lambda self, *a, **k: None})
    real['batchgenerators.dataloading.data_loader'] = dl
    mt = types.ModuleType('batchgenerators.dataloading.multi_threaded_augmenter')
    mt.MultiThreadedAugmenter = type('MultiThreadedAugmenter', (), {})
    real['batchgenerators.dataloading.multi_threaded_augmenter'] = mt
    pad = types.ModuleType('acvl_utils.cropping_and_padding.padding')
    pad.pad_nd_image = pad_nd_image
    real['acvl_utils.cropping_and_padding.padding'] = pad
    # acvl_utils.cropping_and_padding.bounding_boxes (used by preprocessing/cropping/cropping.py:3 and
    # inference/export_prediction.py:5): own restatement of the package's published helpers
    bb = types.ModuleType('acvl_utils.cropping_and_padding.bounding_boxes')

    def get_bbox_from_mask(mask):
        """[[lo, hi), ...] of the True region per axis; the full extent for an empty mask."""
        out = []
        for ax in range(mask.ndim):
            proj = np.any(mask, axis=tuple(a for a in range(mask.ndim) if a != ax))
            idx = np.where(proj)[0]
            out.append([int(idx[0]), int(idx[-1]) + 1] if idx.size else [0, int(mask.shape[ax])])
        return out

    def bounding_box_to_slice(bbox):
        return tuple(slice(*i) for i in bbox)

    def insert_crop_into_image(image, crop, bbox):
        """Writes `crop` into `image` at `bbox` (bbox indexes the trailing axes of `image`)."""
        lead = image.ndim - len(bbox)
        sl = tuple([slice(None)] * lead + [slice(b[0], b[1]) for b in bbox])
        image[sl] = crop
        return image

    bb.get_bbox_from_mask, bb.bounding_box_to_slice, bb.insert_crop_into_image = (
        get_bbox_from_mask, bounding_box_to_slice, insert_crop_into_image)
    real['acvl_utils.cropping_and_padding.bounding_boxes'] = bb
    # the two one-line lookups plans_handler.py:22 needs for old-format plans
    hp = types.ModuleType('dynamic_network_architectures.building_blocks.helper')
    hp.convert_dim_to_conv_op = lambda dim: {1: torch.nn.Conv1d, 2: torch.nn.Conv2d, 3: torch.nn.Conv3d}[dim]
    hp.get_matching_instancenorm = lambda conv_op=None, dimension=None: {
        1: torch.nn.InstanceNorm1d, 2: torch.nn.InstanceNorm2d, 3: torch.nn.InstanceNorm3d}[dimension]
    real['dynamic_network_architectures.building_blocks.helper'] = hp
    sys.modules.update(real)
    sys.meta_path.insert(0, _MockFinder())
    sys.path.insert(0, REFERENCE_ROOT)
    install.done = True
