"""Case definitions shared by ``make_golden.py`` (generator, build container)
and the tests that replay the fixtures.  No reference import happens here."""
from __future__ import annotations

import os
import sys

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from oracle.topology import UNetSpec  # noqa: E402
from oracle.unet import OracleUNet, synthetic_state_dict  # noqa: E402

# (image, patch, step)
STEP_CASES = [
    ([110], [64], 0.5), ([512, 512, 512], [128, 128, 128], 0.5), ([512, 512, 512], [160, 160, 160], 0.5),
    ([512, 512, 512], [160, 96, 96], 0.5), ([128, 128, 128], [128, 128, 128], 0.5), ([129, 128, 200], [128, 128, 128], 0.5),
    ([40, 36, 44], [16, 16, 16], 0.5), ([40, 36, 44], [16, 16, 16], 1.0), ([40, 36, 44], [16, 16, 16], 0.25),
    ([33, 47, 21], [20, 28, 20], 0.5), ([300, 211, 97], [96, 160, 80], 0.5), ([300, 211, 97], [96, 160, 80], 0.75),
    ([122, 101, 30], [122, 101, 30], 0.5), ([65, 65], [64, 64], 0.5), ([1000, 777], [512, 448], 0.5),
    ([17], [16], 0.5), ([18], [16], 0.5), ([19], [16], 0.3), ([100], [7], 0.5), ([100], [7], 1.0), ([101], [10], 0.1),
    ([255, 256, 257], [128, 128, 128], 0.5), ([511], [160], 0.5), ([513], [160], 0.5), ([320], [160], 0.5),
    ([321], [160], 0.5), ([240], [160], 0.5), ([241], [160], 0.5), ([400, 400, 400], [160, 96, 96], 0.5),
    ([160, 96, 96], [160, 96, 96], 0.5), ([161, 97, 98], [160, 96, 96], 0.5), ([96, 512, 512], [48, 192, 192], 0.5),
    ([73], [24], 0.5), ([74], [24], 0.5), ([75], [24], 0.5), ([76], [24], 0.5), ([77], [24], 0.5), ([78], [24], 0.5),
    ([79], [24], 0.5), ([80], [24], 0.5), ([81], [24], 0.5), ([82], [24], 0.5), ([83], [24], 0.5), ([84], [24], 0.5),
    ([85], [24], 0.5), ([86], [24], 0.5), ([87], [24], 0.5), ([88], [24], 0.5), ([89], [24], 0.5), ([90], [24], 0.5),
    ([45, 45, 45], [30, 30, 30], 0.5), ([46, 46, 46], [30, 30, 30], 0.5),
]

GAUSS_FULL = [[8, 8, 8], [16, 12, 10], [20, 28, 20], [32, 32, 32], [16, 16, 16], [24, 16, 20]]
GAUSS_SUMMARY = [[128, 128, 128], [160, 96, 96], [160, 160, 160]]

# spacing, patch  (C1..C5)
TOPOLOGY_CASES = [
    ([1.0, 1.0, 1.0], [128, 128, 128]), ([2.0, 0.9765625, 0.9765625], [160, 96, 96]),
    ([1.0, 1.0, 1.0], [160, 160, 160]), ([3.0, 0.8, 0.8], [40, 224, 192]), ([5.0, 0.7, 0.7], [16, 320, 320]),
    ([1.5, 1.5, 1.5], [96, 160, 160]), ([1.0, 1.0, 1.0], [64, 64, 64]), ([2.5, 2.5, 1.0], [48, 48, 192]),
]

_ARCH = {
    'network_class_name': 'dynamic_network_architectures.architectures.unet.PlainConvUNet',
    'arch_kwargs': {'n_stages': 6, 'features_per_stage': [32, 64, 128, 256, 320, 320],
                    'conv_op': 'torch.nn.modules.conv.Conv3d',
                    'kernel_sizes': [[1, 3, 3], [3, 3, 3], [3, 3, 3], [3, 3, 3], [3, 3, 3], [3, 3, 3]],
                    'strides': [[1, 1, 1], [1, 2, 2], [2, 2, 2], [2, 2, 2], [2, 2, 2], [2, 1, 1]],
                    'n_conv_per_stage': [2, 2, 2, 2, 2, 2], 'n_conv_per_stage_decoder': [2, 2, 2, 2, 2],
                    'conv_bias': True, 'norm_op': 'torch.nn.modules.instancenorm.InstanceNorm3d',
                    'norm_op_kwargs': {'eps': 1e-05, 'affine': True}, 'dropout_op': None, 'dropout_op_kwargs': None,
                    'nonlin': 'torch.nn.LeakyReLU', 'nonlin_kwargs': {'inplace': True}},
    '_kw_requires_import': ['conv_op', 'norm_op', 'dropout_op', 'nonlin']}

PLANS_NEW = {
    'dataset_name': 'Dataset999_Golden', 'plans_name': 'nnUNetPlans',
    'original_median_spacing_after_transp': [2.0, 0.9765625, 0.9765625],
    'original_median_shape_after_transp': [300, 512, 512], 'image_reader_writer': 'SimpleITKIO',
    'transpose_forward': [0, 1, 2], 'transpose_backward': [0, 1, 2],
    'experiment_planner_used': 'ExperimentPlanner', 'label_manager': 'LabelManager',
    'foreground_intensity_properties_per_channel': {'0': {'mean': 418.68, 'std': 412.19, 'percentile_00_5': -60.0,
                                                          'percentile_99_5': 3068.0}},
    'configurations': {
        '3d_lowres': {'data_identifier': 'nnUNetPlans_3d_lowres', 'preprocessor_name': 'DefaultPreprocessor',
                      'batch_size': 2, 'patch_size': [128, 128, 128], 'spacing': [3.0, 2.0, 2.0],
                      'normalization_schemes': ['CTNormalization'], 'use_mask_for_norm': [False],
                      'architecture': _ARCH, 'next_stage': '3d_cascade_fullres'},
        '3d_fullres': {'data_identifier': 'nnUNetPlans_3d_fullres', 'preprocessor_name': 'DefaultPreprocessor',
                       'batch_size': 2, 'patch_size': [160, 96, 96], 'spacing': [2.0, 0.9765625, 0.9765625],
                       'normalization_schemes': ['CTNormalization'], 'use_mask_for_norm': [False],
                       'architecture': _ARCH},
        '3d_fullres_bs4': {'inherits_from': '3d_fullres', 'batch_size': 4, 'patch_size': [96, 96, 96]},
        '3d_cascade_fullres': {'inherits_from': '3d_fullres', 'previous_stage': '3d_lowres'},
    }}

PLANS_OLD = {
    'dataset_name': 'Dataset998_GoldenOld', 'plans_name': 'nnUNetPlans',
    'original_median_spacing_after_transp': [1.0, 1.0, 1.0], 'original_median_shape_after_transp': [200, 200, 200],
    'image_reader_writer': 'SimpleITKIO', 'transpose_forward': [0, 1, 2], 'transpose_backward': [0, 1, 2],
    'experiment_planner_used': 'ExperimentPlanner', 'label_manager': 'LabelManager',
    'foreground_intensity_properties_per_channel': {'0': {'mean': 0.0, 'std': 1.0}},
    'configurations': {
        '3d_fullres': {'data_identifier': 'nnUNetPlans_3d_fullres', 'preprocessor_name': 'DefaultPreprocessor',
                       'batch_size': 2, 'patch_size': [128, 128, 128], 'spacing': [1.0, 1.0, 1.0],
                       'normalization_schemes': ['ZScoreNormalization'], 'use_mask_for_norm': [False],
                       'UNet_class_name': 'PlainConvUNet', 'UNet_base_num_features': 32,
                       'n_conv_per_stage_encoder': [2, 2, 2, 2, 2, 2], 'n_conv_per_stage_decoder': [2, 2, 2, 2, 2],
                       'num_pool_per_axis': [5, 5, 5],
                       'pool_op_kernel_sizes': [[1, 1, 1], [2, 2, 2], [2, 2, 2], [2, 2, 2], [2, 2, 2], [2, 2, 2]],
                       'conv_kernel_sizes': [[3, 3, 3]] * 6, 'unet_max_num_features': 320},
    }}

DATASET_JSONS = {
    'labels3': {'labels': {'background': 0, 'a': 1, 'b': 2}, 'channel_names': {'0': 'CT'}, 'file_ending': '.nii.gz'},
    'two_mod': {'labels': {'background': 0, 'a': 1, 'b': 2, 'c': 3}, 'channel_names': {'0': 'T1', '1': 'T2'},
                'file_ending': '.nii.gz'},
    'regions': {'labels': {'background': 0, 'whole': [1, 2, 3], 'core': [2, 3], 'enh': 3},
                'regions_class_order': [1, 2, 3], 'channel_names': {'0': 'T1'}, 'file_ending': '.nii.gz'},
    # >= 255 foreground labels: uint16 label maps (export_prediction.py:45-46); class values above 255
    'regions_u16': {'labels': {'background': 0, 'whole': list(range(1, 301)), 'core': list(range(100, 301)), 'enh': 300},
                    'regions_class_order': [1, 100, 300], 'channel_names': {'0': 'T1'}, 'file_ending': '.nii.gz'},
}


def _c(name, kind, shape, patch, heads=3, channels=1, step=0.5, gaussian=True, mirror=None, folds=1, seed=0,
       act=None, **kw):
    d = dict(name=name, kind=kind, shape=list(shape), patch=list(patch), heads=heads, channels=channels, step=step,
             gaussian=gaussian, mirror=mirror, folds=folds, seed=seed, act=act)
    d.update(kw)
    return d


SW_CASES = [
    _c('exact_basic', 'exact', (40, 36, 44), (16, 16, 16)),
    _c('exact_nogauss', 'exact', (40, 36, 44), (16, 16, 16), gaussian=False, seed=1),
    _c('exact_step1', 'exact', (40, 36, 44), (16, 16, 16), step=1.0, seed=2),
    _c('exact_mirror0', 'exact', (33, 30, 21), (16, 24, 16), mirror=[0], seed=3, heads=4),
    _c('exact_mirror012', 'exact', (28, 30, 33), (16, 16, 24), mirror=[0, 1, 2], seed=4, heads=2, channels=2),
    _c('exact_mirror12_lrelu', 'exact', (24, 30, 33), (16, 16, 16), mirror=[1, 2], seed=5, act='lrelu_half'),
    _c('exact_smaller_than_patch', 'exact', (11, 30, 9), (16, 16, 16), seed=6, heads=5),
    _c('exact_equal_patch', 'exact', (16, 16, 16), (16, 16, 16), seed=7),
    _c('exact_3folds', 'exact', (30, 24, 27), (16, 16, 16), folds=3, seed=8, heads=3),
    _c('exact_1fold_via_folds', 'exact', (30, 24, 27), (16, 16, 16), folds=1, seed=9, via_folds=True),
    _c('exact_aniso_patch', 'exact', (48, 20, 37), (32, 8, 16), seed=10, heads=3, channels=2, act='lrelu_half'),
    _c('unet_basic', 'unet', (40, 36, 44), (16, 16, 32), heads=3, seed=11),
    _c('unet_mirror_folds', 'unet', (24, 40, 24), (16, 32, 16), heads=2, channels=2, mirror=[0, 1, 2], folds=2, seed=12),
]


# The reference's GPU numerics (network output fp16 under autocast: every step of the accumulation in half precision)
# produced by the reference's own predictor on a CPU through networks that return fp16; fixtures in sliding_window_half.npz
SW_CASES_HALF = [
    _c('half_basic', 'exact', (40, 36, 44), (16, 16, 16), act='scale03', half_out=True, seed=30),
    _c('half_nogauss', 'exact', (40, 36, 44), (16, 16, 16), gaussian=False, act='scale03', half_out=True, seed=31),
    _c('half_step03', 'exact', (30, 36, 29), (16, 16, 16), step=0.3, act='scale03', half_out=True, seed=32, heads=4),
    _c('half_mirror0', 'exact', (33, 30, 21), (16, 24, 16), mirror=[0], act='scale03', half_out=True, seed=33, heads=4),
    _c('half_mirror012', 'exact', (28, 30, 33), (16, 16, 24), mirror=[0, 1, 2], act='scale03', half_out=True, seed=34,
       heads=2, channels=2),
    _c('half_smaller_than_patch', 'exact', (11, 30, 9), (16, 16, 16), act='scale03', half_out=True, seed=35, heads=5),
    _c('half_3folds', 'exact', (30, 24, 27), (16, 16, 16), folds=3, act='scale03', half_out=True, seed=36, heads=3),
    _c('half_plain_lrelu', 'exact', (24, 30, 33), (16, 16, 16), mirror=[1, 2], act='lrelu_half', half_out=True, seed=37),
]


# 2-D configurations (patch_size with two entries: every slice of the first axis is tiled, predict_from_raw_data.py
# :508-524); fixtures in sliding_window_2d.npz
SW_CASES_2D = [
    _c('exact2d_basic', 'exact', (5, 36, 44), (16, 16), seed=20),
    _c('exact2d_mirror01', 'exact', (3, 30, 21), (24, 16), mirror=[0, 1], seed=21, heads=4, channels=2),
    _c('exact2d_smaller_than_patch', 'exact', (4, 11, 30), (16, 16), seed=22, heads=2, act='lrelu_half'),
    _c('exact2d_2folds_nogauss', 'exact', (3, 24, 27), (16, 16), folds=2, seed=23, gaussian=False, step=0.3),
    _c('unet2d_basic', 'unet', (6, 40, 44), (16, 32), heads=3, seed=24),
    _c('unet2d_mirror1', 'unet', (2, 20, 70), (16, 32), heads=2, channels=2, mirror=[1], seed=25),
]


class ExactConvNet(nn.Module):
    """Zero-padded 3x3x3 (or 3x3) conv (+ optional LeakyReLU(1/2)) with dyadic weights:
    every fp32 sum is exact, so results do not depend on summation order.
    ``half_out``: the logits leave as fp16 - what the reference's network does under ``torch.autocast`` on a GPU
    (predict_from_raw_data.py:591-593), so the predictor's own code then runs its half-precision arithmetic
    (mirror sums, ``prediction *= gaussian``, ``+=``) on a CPU too.  ``scale03`` multiplies the exact sums by 0.3f
    first (one IEEE operation: still independent of the summation order) so that the rounding to fp16 is not trivial."""

    def __init__(self, cin, heads, act=None, nd=3, half_out=False):
        super().__init__()
        self.conv = (nn.Conv3d if nd == 3 else nn.Conv2d)(cin, heads, 3, padding=1, bias=True)
        self.act = act
        self.half_out = half_out

    def forward(self, x):
        y = self.conv(x)
        if self.act == 'lrelu_half':
            y = nn.functional.leaky_relu(y, 0.5)
        elif self.act == 'scale03':
            y = y * 0.3
        return y.half() if self.half_out else y


def exact_state_dict(cin, heads, seed, nd=3):
    g = torch.Generator().manual_seed(seed)
    w = torch.randint(-4, 5, (heads, cin) + (3,) * nd, generator=g).float() / 8
    b = torch.randint(-8, 9, (heads,), generator=g).float() / 4
    return {'conv.weight': w, 'conv.bias': b}


def toy_unet_spec_2d(cin, heads):
    return UNetSpec('plain', cin, heads, [8, 16, 16], [(3, 3)] * 3, [(1, 1), (2, 2), (1, 2)], [2, 2, 2], [2, 2])


def toy_unet_spec(cin, heads):
    return UNetSpec('plain', cin, heads, [8, 16, 16], [(3, 3, 3)] * 3, [(1, 1, 1), (2, 2, 2), (1, 2, 2)],
                    [2, 2, 2], [2, 2])


def make_case_inputs(case):
    g = torch.Generator().manual_seed(1000 + case['seed'])
    shape = (case['channels'], *case['shape'])
    if case['kind'] == 'exact':
        return torch.randint(-16, 17, shape, generator=g).float() / 8
    return torch.randn(shape, generator=g)


def make_case_networks(case):
    """-> ([net per fold], [state_dict per fold])"""
    nets, params = [], []
    for f in range(case['folds']):
        seed = 77 * case['seed'] + f
        nd = len(case['patch'])
        if case['kind'] == 'exact':
            net = ExactConvNet(case['channels'], case['heads'], case['act'], nd, case.get('half_out', False))
            sd = exact_state_dict(case['channels'], case['heads'], seed, nd)
        else:
            spec = (toy_unet_spec if nd == 3 else toy_unet_spec_2d)(case['channels'], case['heads'])
            net = OracleUNet(spec)
            sd = synthetic_state_dict(spec, seed)
        net.load_state_dict(sd)
        nets.append(net.eval())
        params.append(sd)
    return nets, params


def label_rule_inputs():
    """Logits for the label-rule golden vectors (tests/golden/label_rules.npz), rebuilt from fixed seeds."""
    import numpy as np
    import torch
    allh = torch.arange(0, 65536, dtype=torch.int32).to(torch.int16).view(torch.half)          # every fp16 bit pattern
    g = torch.Generator().manual_seed(123)
    heads = [allh[torch.randperm(65536, generator=g)] for _ in range(3)]
    regions_f16 = torch.stack(heads).reshape(3, 64, 32, 32)
    # fp32 logits on both sides of torch's sigmoid(x) > 0.5 threshold (1.5 * 2^-24), plus ordinary values
    bits = np.arange(0x33bffff0, 0x33c00010, dtype=np.uint32)
    near = torch.from_numpy(bits.view(np.float32).copy())
    body = torch.randn(3, 4096 - near.numel(), generator=g) * 1e-3
    f32 = torch.cat([torch.stack([near, near.flip(0), -near]), body], 1).reshape(3, 16, 16, 16)
    # argmax: 5 heads of coarse values (many ties), a few NaNs
    am = (torch.randint(-3, 4, (5, 16, 16, 16), generator=g).float() * 0.5).half()
    am[2, 0, 0, :4] = float('nan')
    am[4, 1, 0, :4] = float('nan')
    am[0, 2, 0, :4] = float('nan')
    return {'regions_f16': regions_f16, 'regions_f32': f32, 'argmax_f16': am}


# ---- preprocessing without resampling (f-2) and label un-cropping (f-3): fixtures in preprocess.npz
PREP_CASES = [
    dict(name='ct_identity', shape=(1, 24, 20, 28), margins=((3, 2), (0, 4), (5, 1)), tf=(0, 1, 2), schemes=['CTNormalization'],
         props={'0': {'mean': 418.6798400878906, 'std': 412.1883239746094, 'percentile_00_5': -60.0, 'percentile_99_5': 3068.0}},
         scale=900.0, offset=300.0, seed=1),
    dict(name='zscore_rescale_transposed', shape=(2, 18, 22, 26), margins=((0, 5), (2, 2), (1, 0)), tf=(2, 0, 1),
         schemes=['ZScoreNormalization', 'RescaleTo01Normalization'], props={'0': {}, '1': {}}, scale=50.0, offset=10.0, seed=2),
    dict(name='rgb_none', shape=(3, 8, 30, 33), margins=((1, 1), (4, 0), (0, 6)), tf=(1, 2, 0),
         schemes=['RGBTo01Normalization', 'NoNormalization', 'RGBTo01Normalization'], props={'0': {}, '1': {}, '2': {}},
         scale=None, offset=None, seed=3),
    dict(name='all_zero', shape=(1, 6, 7, 9), margins=None, tf=(0, 1, 2), schemes=['NoNormalization'], props={'0': {}},
         scale=0.0, offset=0.0, seed=4),
    dict(name='zscore_masked_holes', shape=(2, 20, 24, 22), margins=((2, 3), (1, 4), (3, 2)), tf=(1, 2, 0),
         schemes=['ZScoreNormalization', 'ZScoreNormalization'], props={'0': {}, '1': {}}, scale=40.0, offset=90.0, seed=6,
         use_mask=[True, False], blobs=True),
    dict(name='ct_single_voxel_holes', shape=(1, 12, 12, 12), margins=((4, 4), (3, 5), (2, 2)), tf=(0, 2, 1),
         schemes=['CTNormalization'],
         props={'0': {'mean': 10.0, 'std': 0.0, 'percentile_00_5': -5.0, 'percentile_99_5': 5.0}}, scale=8.0, offset=0.0,
         seed=5, holes=True),
]


def prep_case_input(case):
    """Raw image [C, s0, s1, s2] (float32) with exact-zero margins (and optionally zero holes inside)."""
    import numpy as np
    rng = np.random.default_rng(4000 + case['seed'])
    shape = case['shape']
    if case['scale'] is None:                                   # uint8-like RGB
        x = rng.integers(1, 256, shape).astype(np.float32)
    else:
        x = (rng.standard_normal(shape) * case['scale'] + case['offset']).astype(np.float32)
        x[x == 0] = 1.0
    if case['margins'] is None:
        return np.zeros(shape, np.float32)
    m = case['margins']
    keep = np.zeros(shape[1:], bool)
    keep[m[0][0]:shape[1] - m[0][1], m[1][0]:shape[2] - m[1][1], m[2][0]:shape[3] - m[2][1]] = True
    if case.get('holes'):
        keep[shape[1] // 2, shape[2] // 2 - 1:shape[2] // 2 + 1, shape[3] // 2] = False
    if case.get('blobs'):
        # an enclosed zero blob (a hole: stays inside the mask), a zero tunnel that reaches the margin (outside), and an
        # enclosed blob that is zero in channel 0 only (non-zero somewhere: foreground)
        keep[8:12, 9:14, 9:13] = False
        keep[5:7, 0:12, 6:8] = False
        keep[14:16, 16:19, 5:17] = False
        keep[15, 17, 17:shape[3]] = False
    x[:, ~keep] = 0
    if case.get('blobs'):
        x[0, 3:6, 15:18, 12:15] = 0                                  # zero in one channel only
    return x


def prep_label_input(case, cropped_shape, n_labels=7):
    import numpy as np
    rng = np.random.default_rng(5000 + case['seed'])
    return rng.integers(0, n_labels, cropped_shape).astype(np.uint8)


# ---- resampling decisions (pure-numpy functions of preprocessing/resampling/default_resampling.py): resample_logic.json
RESAMPLE_LOGIC_CASES = [
    # (shape, current spacing, new spacing, force_separate_z)
    ((30, 44, 36), (0.8, 3.0, 0.8), (1.0, 2.0, 1.0), None),
    ((160, 96, 96), (2.0, 0.9765625, 0.9765625), (2.0, 0.9765625, 0.9765625), None),
    ((55, 512, 512), (5.0, 0.7, 0.7), (2.5, 0.8, 0.8), None),
    ((55, 512, 512), (1.0, 0.7, 0.7), (4.0, 0.8, 0.8), None),
    ((100, 100, 100), (1.0, 1.0, 1.0), (1.5, 1.5, 1.5), None),
    ((100, 100, 100), (0.24, 1.25, 1.25), (1.0, 1.0, 1.0), None),       # two low-res axes: no separate z
    ((40, 41, 43), (3.0, 3.0, 3.0), (1.0, 1.0, 1.0), True),             # forced, but all axes equal
    ((40, 41, 43), (4.0, 1.0, 1.1), (1.0, 1.0, 1.0), True),
    ((40, 41, 43), (4.0, 1.0, 1.1), (1.0, 1.0, 1.0), False),
    ((7, 9, 5), (1.0, 1.0, 2.5), (2.0, 1.0, 1.0), None),               # round-half-even in compute_new_shape
    ((101, 51, 7), (1.0, 1.0, 2.5), (2.0, 1.0, 1.0), None),
]


# ---- export with probabilities (export_prediction.py:16-70 with return_probabilities=True, no resampling): export.npz
EXPORT_CASES = [
    dict(name='labels_transposed', dataset='two_mod', cropped=(10, 12, 9), bbox=[[2, 12], [1, 13], [3, 12]],
         before=(14, 15, 13), tf=(2, 0, 1), seed=1),
    dict(name='regions_identity', dataset='regions', cropped=(8, 9, 10), bbox=[[0, 8], [2, 11], [1, 11]],
         before=(8, 12, 11), tf=(0, 1, 2), seed=2),
    dict(name='labels_no_crop', dataset='labels3', cropped=(6, 7, 8), bbox=[[0, 6], [0, 7], [0, 8]],
         before=(6, 7, 8), tf=(1, 2, 0), seed=3),
]


def export_case_logits(case, heads):
    """fp16 logits [heads, *cropped] with a few exact ties between heads."""
    import numpy as np
    rng = np.random.default_rng(6000 + case['seed'])
    x = (rng.standard_normal((heads, *case['cropped'])) * 3).astype(np.float16)
    x[1, 0] = x[0, 0]                                           # ties: the first maximum wins
    return x
