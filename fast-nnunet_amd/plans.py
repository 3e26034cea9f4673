"""Readers for the reference's input formats on the hot path: ``plans.json`` and
``dataset.json``.

Mirrors the parts of the reference's ``PlansManager`` / ``ConfigurationManager``
(utilities/plans_handling/plans_handler.py:31-341) and ``LabelManager``
(utilities/label_handling/label_handling.py:21-311) that the predictor and
its callers read.  Same attribute names; only inference-relevant members.
"""
from __future__ import annotations

import json
import warnings
from copy import deepcopy
from typing import Dict, List, Optional, Sequence, Tuple, Union

_OLD_KEYS = ('UNet_class_name', 'UNet_base_num_features', 'n_conv_per_stage_encoder', 'n_conv_per_stage_decoder',
             'num_pool_per_axis', 'pool_op_kernel_sizes', 'conv_kernel_sizes', 'unet_max_num_features')
_CLASS_PATHS = {
    'PlainConvUNet': 'dynamic_network_architectures.architectures.unet.PlainConvUNet',
    'ResidualEncoderUNet': 'dynamic_network_architectures.architectures.residual_unet.ResidualEncoderUNet',
}


def _upgrade_old_format(cfg: dict) -> dict:
    """Old plans (no 'architecture' block) -> new block, as plans_handler.py:36-97 does."""
    name = cfg['UNet_class_name']
    if name not in _CLASS_PATHS:
        raise RuntimeError(f'Unknown architecture {name}. This conversion only supports '
                           f'PlainConvUNet and ResidualEncoderUNet')
    dim = len(cfg['patch_size'])
    n_stages = len(cfg['n_conv_per_stage_encoder'])
    count_key = 'n_conv_per_stage' if name == 'PlainConvUNet' else 'n_blocks_per_stage'
    kwargs = {
        'n_stages': n_stages,
        'features_per_stage': [min(cfg['UNet_base_num_features'] * 2 ** i, cfg['unet_max_num_features'])
                               for i in range(n_stages)],
        'conv_op': f'torch.nn.modules.conv.Conv{dim}d',
        'kernel_sizes': deepcopy(cfg['conv_kernel_sizes']),
        'strides': deepcopy(cfg['pool_op_kernel_sizes']),
        count_key: deepcopy(cfg['n_conv_per_stage_encoder']),
        'n_conv_per_stage_decoder': deepcopy(cfg['n_conv_per_stage_decoder']),
        'conv_bias': True,
        'norm_op': f'torch.nn.modules.instancenorm.InstanceNorm{dim}d',
        'norm_op_kwargs': {'eps': 1e-05, 'affine': True},
        'dropout_op': None, 'dropout_op_kwargs': None,
        'nonlin': 'torch.nn.LeakyReLU', 'nonlin_kwargs': {'inplace': True},
    }
    out = {k: v for k, v in cfg.items() if k not in _OLD_KEYS}
    out['architecture'] = {'network_class_name': _CLASS_PATHS[name], 'arch_kwargs': kwargs,
                           '_kw_requires_import': ['conv_op', 'norm_op', 'dropout_op', 'nonlin']}
    return out


class ConfigurationManager:
    def __init__(self, configuration_dict: dict):
        if 'architecture' not in configuration_dict:
            warnings.warn('Detected old nnU-Net plans format. Attempting to reconstruct network architecture '
                          'parameters.')
            configuration_dict = _upgrade_old_format(configuration_dict)
        self.configuration = configuration_dict

    def __repr__(self):
        return repr(self.configuration)

    @property
    def patch_size(self) -> List[int]:
        return self.configuration['patch_size']

    @property
    def spacing(self) -> List[float]:
        return self.configuration['spacing']

    @property
    def batch_size(self) -> int:
        return self.configuration['batch_size']

    @property
    def data_identifier(self) -> str:
        return self.configuration['data_identifier']

    @property
    def preprocessor_name(self) -> str:
        return self.configuration['preprocessor_name']

    @property
    def normalization_schemes(self) -> List[str]:
        return self.configuration['normalization_schemes']

    @property
    def use_mask_for_norm(self) -> List[bool]:
        return self.configuration['use_mask_for_norm']

    # resampling (plans_handler.py:100-120 builds partials of default_resampling functions from these)
    @property
    def resampling_fn_data_name(self) -> str:
        return self.configuration.get('resampling_fn_data', 'resample_data_or_seg_to_shape')

    @property
    def resampling_fn_data_kwargs(self) -> dict:
        return self.configuration.get('resampling_fn_data_kwargs', {'is_seg': False, 'order': 3, 'order_z': 0,
                                                                       'force_separate_z': None})

    @property
    def resampling_fn_probabilities_name(self) -> str:
        return self.configuration.get('resampling_fn_probabilities', 'resample_data_or_seg_to_shape')

    @property
    def resampling_fn_probabilities_kwargs(self) -> dict:
        return self.configuration.get('resampling_fn_probabilities_kwargs', {'is_seg': False, 'order': 1, 'order_z': 0,
                                                                                'force_separate_z': None})

    @property
    def network_arch_class_name(self) -> str:
        return self.configuration['architecture']['network_class_name']

    @property
    def network_arch_init_kwargs(self) -> dict:
        return self.configuration['architecture']['arch_kwargs']

    @property
    def network_arch_init_kwargs_req_import(self):
        return self.configuration['architecture']['_kw_requires_import']

    @property
    def pool_op_kernel_sizes(self):
        return self.configuration['architecture']['arch_kwargs']['strides']

    @property
    def previous_stage_name(self) -> Optional[str]:
        return self.configuration.get('previous_stage')

    @property
    def next_stage_names(self) -> Optional[List[str]]:
        nxt = self.configuration.get('next_stage')
        return [nxt] if isinstance(nxt, str) else nxt


class LabelManager:
    """Segmentation-head bookkeeping (label_handling.py:21-311, inference part)."""

    def __init__(self, label_dict: dict, regions_class_order: Optional[Sequence[int]], force_use_labels: bool = False):
        if 'background' not in label_dict:
            raise RuntimeError('Background label not declared (remember that this should be label 0!)')
        bg = label_dict['background']
        if isinstance(bg, (tuple, list)):
            raise RuntimeError(f'Background label must be 0. Not a list. Not a tuple. Your background label: {bg}')
        assert int(bg) == 0, f'Background label must be 0. Your background label: {bg}'
        self.label_dict = label_dict
        self.regions_class_order = regions_class_order
        self.has_regions = (not force_use_labels) and any(
            isinstance(v, (tuple, list)) and len(v) > 1 for v in label_dict.values())
        self.ignore_label = self._find_ignore_label()
        self.all_labels = self._collect_labels()
        if self.has_ignore_label:
            assert self.ignore_label == max(self.all_labels) + 1, \
                'If you use the ignore label it must have the highest label value!'

    def _find_ignore_label(self):
        ign = self.label_dict.get('ignore')
        if ign is not None:
            assert isinstance(ign, int), 'Ignore label has to be an integer. It cannot be a region'
        return ign

    def _collect_labels(self) -> List[int]:
        vals = set()
        for name, v in self.label_dict.items():
            if name == 'ignore':
                continue
            if isinstance(v, (tuple, list)):
                vals.update(int(i) for i in v)
            else:
                vals.add(int(v))
        return sorted(vals)

    @property
    def has_ignore_label(self) -> bool:
        return self.ignore_label is not None

    @property
    def all_regions(self):
        if not self.has_regions:
            return None
        out = []
        for name, v in self.label_dict.items():
            if name == 'ignore':
                continue
            if v == 0 or (isinstance(v, (tuple, list)) and len(set(v)) == 1 and list(set(v))[0] == 0):
                continue                                  # pure background is not a region
            out.append(tuple(v) if isinstance(v, (tuple, list)) else int(v))
        return out

    @property
    def foreground_labels(self) -> List[int]:
        return [l for l in self.all_labels if l != 0]

    @property
    def foreground_regions(self):
        return self.all_regions

    @property
    def num_segmentation_heads(self) -> int:
        return len(self.foreground_regions) if self.has_regions else len(self.all_labels)


class PlansManager:
    def __init__(self, plans_file_or_dict: Union[str, dict]):
        if isinstance(plans_file_or_dict, dict):
            self.plans = plans_file_or_dict
        else:
            with open(plans_file_or_dict) as f:
                self.plans = json.load(f)
        self._cache: Dict[str, ConfigurationManager] = {}

    def __repr__(self):
        return repr(self.plans)

    def _resolve(self, name: str, visited: Tuple[str, ...] = ()) -> dict:
        cfgs = self.plans['configurations']
        if name not in cfgs:
            raise ValueError(f'The configuration {name} does not exist in the plans I have. Valid '
                             f'configuration names are {list(cfgs.keys())}.')
        cfg = deepcopy(cfgs[name])
        parent = cfg.get('inherits_from')
        if parent is not None:
            if parent in visited or parent == name:
                raise RuntimeError(f'Circular dependency detected while resolving {name}: {visited}')
            base = self._resolve(parent, (*visited, name))
            base.update(cfg)
            cfg = base
        return cfg

    def get_configuration(self, configuration_name: str) -> ConfigurationManager:
        if configuration_name not in self.plans['configurations']:
            raise RuntimeError(f'Requested configuration {configuration_name} not found in plans. '
                               f"Available configurations: {list(self.plans['configurations'].keys())}")
        if configuration_name not in self._cache:
            self._cache[configuration_name] = ConfigurationManager(self._resolve(configuration_name))
        return self._cache[configuration_name]

    @property
    def available_configurations(self) -> List[str]:
        return list(self.plans['configurations'].keys())

    @property
    def dataset_name(self) -> str:
        return self.plans['dataset_name']

    @property
    def plans_name(self) -> str:
        return self.plans['plans_name']

    @property
    def transpose_forward(self) -> List[int]:
        return self.plans['transpose_forward']

    @property
    def transpose_backward(self) -> List[int]:
        return self.plans['transpose_backward']

    @property
    def original_median_spacing_after_transp(self) -> List[float]:
        return self.plans['original_median_spacing_after_transp']

    @property
    def foreground_intensity_properties_per_channel(self) -> dict:
        if 'foreground_intensity_properties_per_channel' not in self.plans \
                and 'foreground_intensity_properties_by_modality' in self.plans:
            return self.plans['foreground_intensity_properties_by_modality']
        return self.plans['foreground_intensity_properties_per_channel']

    def get_label_manager(self, dataset_json: dict, **kwargs) -> LabelManager:
        return LabelManager(dataset_json['labels'], dataset_json.get('regions_class_order'), **kwargs)


def determine_num_input_channels(plans_manager: PlansManager,
                                 configuration_or_config_manager: Union[str, ConfigurationManager],
                                 dataset_json: dict) -> int:
    """#modalities (+ #foreground labels for a cascade stage); label_handling.py:294-311."""
    cm = plans_manager.get_configuration(configuration_or_config_manager) \
        if isinstance(configuration_or_config_manager, str) else configuration_or_config_manager
    mods = dataset_json['modality'] if 'modality' in dataset_json else dataset_json['channel_names']
    n = len(mods)
    if cm.previous_stage_name is not None:
        n += len(plans_manager.get_label_manager(dataset_json).foreground_labels)
    return n
