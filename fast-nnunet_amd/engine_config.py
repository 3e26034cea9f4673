"""The reference's deployment front-end: an ``.ini`` file describes one exported model and how to run it.

Mirrors the use of ``FastnnUNet::Engine`` in the reference's engine/fast_nnunet.cpp:16-26
(``set_config(ini)``, ``set_workspace(models)``, ``infer(raw image) -> mask``; the class itself is not in the
tree) over the keys of engine/config/fast_nnunet_bone_turbo.ini:

    [model]          file_name, input_name, output_name, num_class
    [input]          depth, height, width, patch_size, target_spacing
    [preprocessing]  mean, std_dev, lower_bound, upper_bound        (CTNormalization's four numbers)
    [inference]      use_mirroring, step_size, use_gaussian

The reference's ``file_name`` is a TensorRT engine; here the workspace is an nnU-Net model folder (plans.json,
dataset.json, fold_k/checkpoint_*.pth) whose network must agree with the ini (patch size, number of classes).
The ini is authoritative for everything it states: target spacing, normalisation numbers, inference options.
"""
from __future__ import annotations

import configparser
import os
from copy import deepcopy
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from .plans import PlansManager

_BOOL = {'true': True, 'false': False, '1': True, '0': False, 'yes': True, 'no': False, 'on': True, 'off': False}


@dataclass
class EngineConfig:
    file_name: str
    input_name: str
    output_name: str
    num_class: int
    patch_size: List[int]
    target_spacing: List[float]
    mean: float
    std_dev: float
    lower_bound: float
    upper_bound: float
    use_mirroring: bool
    step_size: float
    use_gaussian: bool


def _floats(text: str) -> List[float]:
    return [float(t) for t in text.replace(';', ',').split(',') if t.strip()]


def _bool(text: str, key: str) -> bool:
    if text.strip().lower() not in _BOOL:
        raise ValueError(f'{key}: expected true/false, got {text!r}')
    return _BOOL[text.strip().lower()]


def load_engine_config(path: str) -> EngineConfig:
    """Parse and validate an engine ini (keys of engine/config/fast_nnunet_bone_turbo.ini:1-23)."""
    if not os.path.isfile(path):
        raise FileNotFoundError(path)
    cp = configparser.ConfigParser()
    cp.read(path)
    for sec in ('model', 'input', 'preprocessing', 'inference'):
        if not cp.has_section(sec):
            raise ValueError(f'{path}: section [{sec}] is missing')
    m, i, pp, inf = cp['model'], cp['input'], cp['preprocessing'], cp['inference']
    patch = [int(round(v)) for v in _floats(i['patch_size'])] if 'patch_size' in i else \
        [int(i['depth']), int(i['height']), int(i['width'])]
    if len(patch) != 3 or min(patch) <= 0:
        raise ValueError(f'{path}: patch_size must be three positive integers, got {patch}')
    for key, want in zip(('depth', 'height', 'width'), patch):
        if key in i and int(i[key]) != want:
            raise ValueError(f'{path}: [input] {key} = {i[key]} disagrees with patch_size = {patch}')
    spacing = _floats(i['target_spacing'])
    if len(spacing) != 3 or min(spacing) <= 0:
        raise ValueError(f'{path}: target_spacing must be three positive numbers, got {spacing}')
    cfg = EngineConfig(file_name=m.get('file_name', ''), input_name=m.get('input_name', 'input'),
                       output_name=m.get('output_name', 'output'), num_class=int(m['num_class']),
                       patch_size=patch, target_spacing=spacing,
                       mean=float(pp['mean']), std_dev=float(pp['std_dev']),
                       lower_bound=float(pp['lower_bound']), upper_bound=float(pp['upper_bound']),
                       use_mirroring=_bool(inf.get('use_mirroring', 'false'), 'use_mirroring'),
                       step_size=float(inf.get('step_size', '0.5')),
                       use_gaussian=_bool(inf.get('use_gaussian', 'true'), 'use_gaussian'))
    if cfg.num_class < 1:
        raise ValueError(f'{path}: num_class must be positive')
    if not (0 < cfg.step_size <= 1):
        raise ValueError(f'{path}: step_size must be in (0, 1], got {cfg.step_size}')
    if cfg.std_dev <= 0 or cfg.upper_bound < cfg.lower_bound:
        raise ValueError(f'{path}: std_dev must be positive and lower_bound <= upper_bound')
    return cfg


def plans_with_engine_config(plans: dict, configuration_name: str, cfg: EngineConfig) -> dict:
    """A copy of ``plans`` in which ``configuration_name`` states what the ini states: target spacing,
    CTNormalization with the ini's four numbers on channel 0."""
    out = deepcopy(plans)
    conf = out['configurations'][configuration_name]
    conf['spacing'] = list(cfg.target_spacing)
    conf['normalization_schemes'] = ['CTNormalization']
    conf['use_mask_for_norm'] = [False]
    props = dict(out.get('foreground_intensity_properties_per_channel') or {})
    ch0 = dict(props.get('0', {}))
    ch0.update({'mean': cfg.mean, 'std': cfg.std_dev, 'percentile_00_5': cfg.lower_bound,
                'percentile_99_5': cfg.upper_bound})
    props['0'] = ch0
    out['foreground_intensity_properties_per_channel'] = props
    return out


class Engine:
    """``set_config`` -> ``set_workspace`` -> ``infer``: raw CT in, label mask out, every step on the GPU."""

    def __init__(self, device: torch.device = torch.device('cuda'), patches_per_forward: int = 8, verbose: bool = False):
        self.device = device
        self.patches_per_forward = patches_per_forward
        self.verbose = verbose
        self.config: Optional[EngineConfig] = None
        self.predictor = None

    def set_config(self, ini_path: str) -> EngineConfig:
        self.config = load_engine_config(ini_path)
        self.predictor = None
        return self.config

    def set_workspace(self, model_training_output_dir: str, use_folds: Union[Tuple[Union[int, str], ...], None] = None,
                      checkpoint_name: str = 'checkpoint_final.pth'):
        """Loads the model folder and checks it against the ini (engine/fast_nnunet.cpp:20 ``set_workspace``)."""
        from .predictor import nnUNetPredictor
        if self.config is None:
            raise RuntimeError('set_config() must be called before set_workspace()')
        cfg = self.config
        p = nnUNetPredictor(tile_step_size=cfg.step_size, use_gaussian=cfg.use_gaussian, use_mirroring=cfg.use_mirroring,
                            perform_everything_on_device=True, device=self.device, verbose=self.verbose,
                            allow_tqdm=False, patches_per_forward=self.patches_per_forward)
        p.initialize_from_trained_model_folder(model_training_output_dir, use_folds, checkpoint_name)
        if [int(v) for v in p.configuration_manager.patch_size] != list(cfg.patch_size):
            raise RuntimeError(f'the model was trained with patch size {list(p.configuration_manager.patch_size)}, '
                               f'the ini says {cfg.patch_size}')
        if p.label_manager.num_segmentation_heads != cfg.num_class:
            raise RuntimeError(f'the model has {p.label_manager.num_segmentation_heads} segmentation heads, '
                               f'the ini says num_class = {cfg.num_class}')
        if len(p.dataset_json.get('channel_names', p.dataset_json.get('modality', {}))) != 1:
            raise RuntimeError('the engine ini describes a single-channel (CT) model')
        name = p._configuration_name
        pm = PlansManager(plans_with_engine_config(p.plans_manager.plans, name, cfg))
        p.plans_manager = pm
        p.configuration_manager = pm.get_configuration(name)
        self.predictor = p

    def infer(self, image: np.ndarray, spacing: Sequence[float]) -> np.ndarray:
        """``image``: ``[s0, s1, s2]`` or ``[1, s0, s1, s2]`` raw intensities in the axis order the model was
        trained with; ``spacing``: the three voxel sizes in that order.  Returns the label mask on the same grid
        (uint8, or uint16 for >= 255 foreground labels) - engine/fast_nnunet.cpp:27 ``Engine->infer``."""
        if self.predictor is None:
            raise RuntimeError('set_config() and set_workspace() must be called before infer()')
        img = np.asarray(image)
        if img.ndim == 3:
            img = img[None]
        if img.ndim != 4 or img.shape[0] != 1:
            raise ValueError(f'image must be [s0, s1, s2] or [1, s0, s1, s2], got shape {tuple(np.asarray(image).shape)}')
        if len(spacing) != 3:
            raise ValueError('spacing must have three entries')
        return self.predictor.predict_single_npy_array(img, {'spacing': [float(s) for s in spacing]})
