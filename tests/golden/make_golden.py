#!/usr/bin/env python3
"""Generate the golden vectors in this directory FROM THE REFERENCE ITSELF.

Run in the build container only (needs ``/root/reference``):

    python tests/golden/make_golden.py

It imports the reference's own ``nnUNetPredictor``, ``compute_gaussian``,
``compute_steps_for_sliding_window``, ``PlansManager``, ``LabelManager`` and
``get_pool_and_conv_props`` (through the import shims in ``ref_shims.py``) and
stores inputs-by-seed + expected outputs as small ``.json`` / ``.npz`` files.
The fixtures are data only; no reference source travels.

Two kinds of sliding-window case:

* ``exact`` - the "network" is a zero-padded 3x3x3 convolution (optionally
  followed by a LeakyReLU with slope 1/2) whose weights and inputs are dyadic
  rationals, so every fp32 sum is exact whatever the summation order / CPU
  ISA.  Expected logits are compared BIT FOR BIT.
* ``unet`` - a seeded 3-stage PlainConvUNet (this repo's ``oracle.unet``
  restatement, handed to the reference predictor through
  ``manual_initialization``).  InstanceNorm makes the result depend on the
  CPU's reduction order in the last ulp, so these are compared with a
  tolerance; they pin the predictor logic around a realistic network.
"""
from __future__ import annotations

import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import ref_shims  # noqa: E402

ref_shims.install()

from nnunetv2.inference.predict_from_raw_data import nnUNetPredictor  # noqa: E402
from nnunetv2.inference.sliding_window_prediction import (compute_gaussian,  # noqa: E402
                                                          compute_steps_for_sliding_window)
from nnunetv2.utilities.plans_handling.plans_handler import PlansManager  # noqa: E402
from nnunetv2.utilities.label_handling.label_handling import LabelManager, determine_num_input_channels  # noqa: E402
from nnunetv2.experiment_planning.experiment_planners.network_topology import get_pool_and_conv_props  # noqa: E402

from golden_cases import (SW_CASES, SW_CASES_2D, SW_CASES_HALF, STEP_CASES, GAUSS_FULL, GAUSS_SUMMARY, TOPOLOGY_CASES, PLANS_NEW,  # noqa: E402
                          PLANS_OLD, DATASET_JSONS, make_case_inputs, make_case_networks)


def bits(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def gen_steps():
    out = []
    for image, patch, step in STEP_CASES:
        out.append({'image': image, 'patch': patch, 'step': step,
                    'steps': compute_steps_for_sliding_window(tuple(image), tuple(patch), step)})
    with open(os.path.join(HERE, 'steps.json'), 'w') as f:
        json.dump(out, f)


def gen_gaussian():
    arrays, summary = {}, []
    for p in GAUSS_FULL:
        compute_gaussian.cache_clear()
        g = compute_gaussian(tuple(p), sigma_scale=1. / 8, value_scaling_factor=10, device=torch.device('cpu'))
        arrays['g_' + '_'.join(map(str, p))] = bits(g)
    for p in GAUSS_SUMMARY:
        compute_gaussian.cache_clear()
        g = compute_gaussian(tuple(p), sigma_scale=1. / 8, value_scaling_factor=10, device=torch.device('cpu'))
        b = bits(g)
        c = [i // 2 for i in p]
        summary.append({'patch': p, 'sha256': hashlib.sha256(b.tobytes()).hexdigest(),
                        'min_bits': int(b.min()), 'max_bits': int(b.max()),
                        'count_at_min': int((b == b.min()).sum()),
                        'line0': b[:, c[1], c[2]].tolist(), 'line1': b[c[0], :, c[2]].tolist(),
                        'line2': b[c[0], c[1], :].tolist()})
    np.savez_compressed(os.path.join(HERE, 'gaussian.npz'), **arrays)
    with open(os.path.join(HERE, 'gaussian_summary.json'), 'w') as f:
        json.dump(summary, f)


def gen_sliding_window_2d():
    gen_sliding_window(SW_CASES_2D, 'sliding_window_2d.npz', '2d')


def gen_sliding_window_half():
    gen_sliding_window(SW_CASES_HALF, 'sliding_window_half.npz')


def gen_sliding_window(cases=None, fname='sliding_window.npz', config='3d_fullres'):
    arrays = {}
    cases = SW_CASES if cases is None else cases
    for case in cases:
        name = case['name']
        image = make_case_inputs(case)
        nets, params = make_case_networks(case)
        plans = PlansManager({'dataset_name': 'Dataset999_Golden', 'plans_name': 'nnUNetPlans',
                              'configurations': {config: {'patch_size': list(case['patch']),
                                                                'architecture': {'network_class_name': 'toy',
                                                                                 'arch_kwargs': {},
                                                                                 '_kw_requires_import': []}}}})
        dataset_json = {'labels': {('background' if i == 0 else f'c{i}'): i for i in range(case['heads'])},
                        'channel_names': {str(i): 'CT' for i in range(case['channels'])}, 'file_ending': '.nii.gz'}
        pred = nnUNetPredictor(tile_step_size=case['step'], use_gaussian=case['gaussian'],
                               use_mirroring=case['mirror'] is not None, perform_everything_on_device=False,
                               device=torch.device('cpu'), verbose=False, allow_tqdm=False)
        pred.manual_initialization(nets[0], plans, plans.get_configuration(config), params, dataset_json,
                                   'nnUNetTrainer', tuple(case['mirror']) if case['mirror'] is not None else None)
        compute_gaussian.cache_clear()
        torch.set_num_threads(4)
        if case['folds'] > 1 or case.get('via_folds', False):
            out = pred.predict_logits_from_preprocessed_data(image)
        else:
            out = pred.predict_sliding_window_return_logits(image)
        assert out.dtype == torch.half and out.shape == (case['heads'], *image.shape[1:])
        arrays[name] = bits(out)
        lm = pred.label_manager
        arrays[name + '__seg'] = lm.convert_logits_to_segmentation(out).numpy().astype(np.int16)
        print(name, tuple(out.shape), float(out.float().abs().max()))
    if fname != 'sliding_window.npz':
        np.savez_compressed(os.path.join(HERE, fname), **arrays)
        return
    # region-based label conversion on one case's logits
    case = SW_CASES[0]
    logits = torch.from_numpy(arrays[case['name']].view(np.int16)).view(torch.half)
    lm = LabelManager(DATASET_JSONS['regions']['labels'], DATASET_JSONS['regions']['regions_class_order'])
    assert lm.num_segmentation_heads == logits.shape[0]
    arrays['regions__seg'] = lm.convert_logits_to_segmentation(logits).numpy().astype(np.int16)
    np.savez_compressed(os.path.join(HERE, 'sliding_window.npz'), **arrays)


def gen_plans():
    out = {}
    for tag, plans in (('new', PLANS_NEW), ('old', PLANS_OLD)):
        import copy
        import warnings
        pm = PlansManager(copy.deepcopy(plans))
        res = {}
        for cfg in pm.available_configurations:
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                cm = pm.get_configuration(cfg)
            res[cfg] = {'patch_size': list(cm.patch_size),
                        'network_class_name': cm.network_arch_class_name,
                        'arch_kwargs': json.loads(json.dumps(cm.network_arch_init_kwargs, default=str)),
                        'pool_op_kernel_sizes': json.loads(json.dumps(cm.pool_op_kernel_sizes)),
                        'previous_stage': cm.previous_stage_name}
            for dj_name, dj in DATASET_JSONS.items():
                lm = pm.get_label_manager(dj)
                res[cfg][f'heads__{dj_name}'] = lm.num_segmentation_heads
                res[cfg][f'cin__{dj_name}'] = determine_num_input_channels(pm, cm, dj)
        out[tag] = res
    with open(os.path.join(HERE, 'plans_expected.json'), 'w') as f:
        json.dump(out, f, indent=1)


def gen_topology():
    out = []
    for spacing, patch in TOPOLOGY_CASES:
        npool, strides, kernels, new_patch, div = get_pool_and_conv_props(tuple(spacing), tuple(patch), 4, 999999)
        out.append({'spacing': spacing, 'patch': patch, 'num_pool': [int(i) for i in npool],
                    'strides': [list(map(int, s)) for s in strides], 'kernels': [list(map(int, k)) for k in kernels],
                    'patch_out': [int(i) for i in new_patch], 'divisible_by': [int(i) for i in div]})
    with open(os.path.join(HERE, 'topology.json'), 'w') as f:
        json.dump(out, f)


def gen_label_rules():
    """LabelManager.convert_logits_to_segmentation on crafted logits: every fp16 bit pattern in every head (the
    sigmoid > 0.5 threshold sits between the two smallest positive fp16 values), fp32 logits around that threshold,
    and the plain argmax with ties / NaNs.  Inputs are rebuilt from seeds by golden_cases.label_rule_inputs()."""
    from golden_cases import label_rule_inputs
    arrays = {}
    inp = label_rule_inputs()
    lm = LabelManager(DATASET_JSONS['regions']['labels'], DATASET_JSONS['regions']['regions_class_order'])
    arrays['regions_f16'] = lm.convert_logits_to_segmentation(inp['regions_f16']).numpy().astype(np.int16)
    arrays['regions_f32'] = lm.convert_logits_to_segmentation(inp['regions_f32']).numpy().astype(np.int16)
    lm2 = LabelManager(DATASET_JSONS['regions_u16']['labels'], DATASET_JSONS['regions_u16']['regions_class_order'])
    arrays['regions_u16'] = lm2.convert_logits_to_segmentation(inp['regions_f16']).numpy().astype(np.int16)
    lm3 = LabelManager({('background' if i == 0 else f'c{i}'): i for i in range(5)}, None)
    arrays['argmax_f16'] = lm3.convert_logits_to_segmentation(inp['argmax_f16']).numpy().astype(np.int16)
    np.savez_compressed(os.path.join(HERE, 'label_rules.npz'), **arrays)
    print('label rules', {k: (v.shape, int(v.max())) for k, v in arrays.items()})


def gen_preprocess():
    """crop_to_nonzero + the normalisation scheme classes + the label half of export_prediction, called in the order
    DefaultPreprocessor.run_case_npy / convert_predicted_logits_to_segmentation_with_correct_shape use them."""
    from golden_cases import PREP_CASES, prep_case_input, prep_label_input
    from nnunetv2.preprocessing.cropping.cropping import crop_to_nonzero
    from nnunetv2.preprocessing.normalization import default_normalization_schemes as dns
    from acvl_utils.cropping_and_padding.bounding_boxes import insert_crop_into_image
    arrays = {}
    for case in PREP_CASES:
        raw = prep_case_input(case)
        data = raw.astype(np.float32)                                             # default_preprocessor.py:49
        tf = list(case['tf'])
        data = data.transpose([0, *[i + 1 for i in tf]])                          # :57
        shape_before = data.shape[1:]
        data, seg, bbox = crop_to_nonzero(data, None)                             # :66
        for c in range(data.shape[0]):                                            # :228-240
            cls = getattr(dns, case['schemes'][c])
            norm = cls(use_mask_for_norm=bool(case.get('use_mask', [False] * data.shape[0])[c]),
                       intensityproperties=case['props'][str(c)])
            data[c] = norm.run(data[c], seg[0])
        arrays[case['name'] + '__data'] = data.astype(np.float32)
        arrays[case['name'] + '__bbox'] = np.asarray(bbox, np.int64)
        arrays[case['name'] + '__shape_before'] = np.asarray(shape_before, np.int64)
        # export_prediction.py:43-53 on a label map of the cropped shape
        tb = [int(i) for i in np.argsort(tf)]
        for n_fg, tag in ((6, 'u8'), (300, 'u16')):
            lab = prep_label_input(case, data.shape[1:])
            full = np.zeros(shape_before, dtype=np.uint8 if n_fg < 255 else np.uint16)
            full = insert_crop_into_image(full, lab, bbox)
            arrays[case['name'] + '__labels_' + tag] = full.transpose(tb)
    np.savez_compressed(os.path.join(HERE, 'preprocess.npz'), **arrays)
    print('preprocess', {k: v.shape for k, v in arrays.items() if k.endswith('__data')})


def gen_export():
    """convert_predicted_logits_to_segmentation_with_correct_shape(..., return_probabilities=True) on logits that need
    no resampling (the resize itself is skimage's, absent here): apply_inference_nonlin, the label rule on the
    probabilities, revert cropping of labels and probabilities (background probability 1 outside the box for plain
    labels, 0 for regions), both transposes back."""
    from golden_cases import EXPORT_CASES, export_case_logits
    from nnunetv2.inference.export_prediction import convert_predicted_logits_to_segmentation_with_correct_shape
    arrays = {}
    for case in EXPORT_CASES:
        dj = DATASET_JSONS[case['dataset']]
        tf = list(case['tf'])
        tb = [int(i) for i in np.argsort(tf)]
        plans = {'dataset_name': 'Dataset998_Export', 'plans_name': 'nnUNetPlans', 'transpose_forward': tf,
                 'transpose_backward': tb, 'label_manager': 'LabelManager',
                 'configurations': {'3d_fullres': {
                     'patch_size': [8, 8, 8], 'spacing': [1.0, 1.0, 1.0],
                     'resampling_fn_probabilities': 'resample_data_or_seg_to_shape',
                     'resampling_fn_probabilities_kwargs': {'is_seg': False, 'order': 1, 'order_z': 0, 'force_separate_z': None},
                     'architecture': {'network_class_name': 'x', 'arch_kwargs': {}, '_kw_requires_import': []}}}}
        pm = PlansManager(plans)
        cm = pm.get_configuration('3d_fullres')
        lm = pm.get_label_manager(dj)
        logits = export_case_logits(case, lm.num_segmentation_heads)
        props = {'spacing': [1.0, 1.0, 1.0], 'shape_before_cropping': tuple(case['before']),
                 'bbox_used_for_cropping': [list(b) for b in case['bbox']],
                 'shape_after_cropping_and_before_resampling': tuple(case['cropped'])}
        seg, probs = convert_predicted_logits_to_segmentation_with_correct_shape(
            torch.from_numpy(logits), pm, cm, lm, props, return_probabilities=True)
        seg_only = convert_predicted_logits_to_segmentation_with_correct_shape(
            torch.from_numpy(logits), pm, cm, lm, props, return_probabilities=False)
        arrays[case['name'] + '__seg'] = np.asarray(seg)
        arrays[case['name'] + '__seg_from_logits'] = np.asarray(seg_only)
        arrays[case['name'] + '__probs'] = np.asarray(probs, dtype=np.float32)
    np.savez_compressed(os.path.join(HERE, 'export.npz'), **arrays)
    print('export', {k: (v.shape, str(v.dtype)) for k, v in arrays.items()})


def gen_resample_logic():
    from golden_cases import RESAMPLE_LOGIC_CASES
    from nnunetv2.preprocessing.resampling.default_resampling import compute_new_shape, determine_do_sep_z_and_axis
    out = []
    for shape, cur, new, force in RESAMPLE_LOGIC_CASES:
        do_sep, axis = determine_do_sep_z_and_axis(force, cur, new)
        out.append({'shape': list(shape), 'current': list(cur), 'new': list(new), 'force': force,
                    'new_shape': [int(i) for i in compute_new_shape(shape, cur, new)],
                    'do_separate_z': bool(do_sep), 'axis': None if axis is None else int(axis)})
    json.dump(out, open(os.path.join(HERE, 'resample_logic.json'), 'w'), indent=1)
    print('resample logic', [(o['do_separate_z'], o['axis'], o['new_shape']) for o in out])


if __name__ == '__main__':
    if len(sys.argv) > 1:                                   # regenerate single fixtures: labels, plans
        for what in sys.argv[1:]:
            {'labels': gen_label_rules, 'plans': gen_plans, 'sw2d': gen_sliding_window_2d, 'swhalf': gen_sliding_window_half, 'prep': gen_preprocess, 'resample': gen_resample_logic, 'export': gen_export}[what]()
        sys.exit(0)
    gen_label_rules()
    gen_steps()
    gen_gaussian()
    gen_topology()
    gen_plans()
    gen_sliding_window()
    gen_sliding_window_2d()
    gen_sliding_window_half()
    gen_preprocess()
    gen_resample_logic()
    gen_export()
    print('golden vectors written to', HERE)
