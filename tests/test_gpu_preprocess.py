"""f-2 / f-3 on the device: crop-to-nonzero box, transpose, intensity normalisation and the label revert against
vectors produced by the reference's own functions (tests/golden/preprocess.npz) and against the oracle on
larger random cases.  CT / RescaleTo01 / RGBTo01 / NoNormalization are bit-exact (three fp32 roundings in the
reference's order); ZScoreNormalization uses double-precision statistics where numpy sums pairwise in fp32:
tolerance 2e-6 relative to the data range."""
import os

import numpy as np
import pytest
import torch

from golden_cases import PREP_CASES, prep_case_input, prep_label_input
from oracle import preprocess as opre

pytestmark = pytest.mark.gpu


class _PM:                                      # the three things the preprocessor reads from a PlansManager
    def __init__(self, tf, props):
        self.transpose_forward = list(tf)
        self.transpose_backward = [int(i) for i in np.argsort(tf)]
        self.foreground_intensity_properties_per_channel = props


class _CM:
    def __init__(self, schemes, spacing=(1.0, 1.0, 1.0)):
        self.normalization_schemes = list(schemes)
        self.use_mask_for_norm = [False] * len(schemes)
        self.spacing = list(spacing)


class _LM:
    def __init__(self, n):
        self.foreground_labels = list(range(1, n + 1))


def _check_data(name, got, ref, schemes):
    assert got.dtype == np.float32 and got.shape == ref.shape
    for c, sch in enumerate(schemes):
        if sch == 'ZScoreNormalization':
            assert np.abs(got[c] - ref[c]).max() <= 2e-6 * max(1.0, np.abs(ref[c]).max()), name
        else:
            assert np.array_equal(got[c].view(np.uint32), ref[c].view(np.uint32)), (name, sch)


@pytest.mark.parametrize('case', PREP_CASES, ids=lambda c: c['name'])
def test_preprocess_matches_reference_golden(case, golden_dir):
    from fast_nnunet_amd.preprocess import DevicePreprocessor
    z = np.load(os.path.join(golden_dir, 'preprocess.npz'))
    pp = DevicePreprocessor(torch.device('cuda', 0))
    props = {'spacing': [1.0, 1.0, 1.0]}
    pm, cm = _PM(case['tf'], case['props']), _CM(case['schemes'])
    data, seg, props = pp.run_case_npy(prep_case_input(case), None, props, pm, cm)
    assert seg is None and data.is_cuda
    assert np.array_equal(np.asarray(props['bbox_used_for_cropping']), z[case['name'] + '__bbox'])
    assert list(props['shape_before_cropping']) == z[case['name'] + '__shape_before'].tolist()
    assert list(props['shape_after_cropping_and_before_resampling']) == list(z[case['name'] + '__data'].shape[1:])
    _check_data(case['name'], data.cpu().numpy(), z[case['name'] + '__data'], case['schemes'])
    for n_fg, tag in ((6, 'u8'), (300, 'u16')):
        lab = prep_label_input(case, tuple(data.shape[1:]))
        full = pp.revert_labels(torch.from_numpy(lab.astype(np.int32)).cuda(), props, pm, _LM(n_fg)).cpu().numpy()
        want = z[case['name'] + '__labels_' + tag]
        assert full.shape == want.shape and np.array_equal(full.astype(np.int64), want.astype(np.int64))


@pytest.mark.parametrize('tf', [(0, 1, 2), (1, 0, 2), (2, 1, 0), (1, 2, 0)])
def test_preprocess_matches_oracle_on_a_larger_volume(tf):
    from fast_nnunet_amd.preprocess import DevicePreprocessor
    rng = np.random.default_rng(7)
    raw = (rng.standard_normal((2, 70, 45, 83)) * 400 + 100).astype(np.float32)
    raw[:, :9] = 0; raw[:, :, -4:] = 0; raw[:, :, :, :2] = 0; raw[:, 60:] = 0
    raw[1, 30:40, 10:20, 50:60] = 0                                   # zero block in one channel only: still inside
    schemes = ['CTNormalization', 'ZScoreNormalization']
    ip = {'0': {'mean': 80.5, 'std': 377.25, 'percentile_00_5': -900.0, 'percentile_99_5': 1300.0}, '1': {}}
    want, bbox, before = opre.preprocess_case(raw, tf, schemes, ip)
    pp = DevicePreprocessor(torch.device('cuda', 0))
    props = {'spacing': [1.0, 1.0, 1.0]}
    got, _, props = pp.run_case_npy(torch.from_numpy(raw), None, props, _PM(tf, ip), _CM(schemes))
    assert props['bbox_used_for_cropping'] == bbox and tuple(props['shape_before_cropping']) == tuple(before)
    _check_data(str(tf), got.cpu().numpy(), want, schemes)
    lab = rng.integers(0, 5, want.shape[1:]).astype(np.uint8)
    tb = [int(i) for i in np.argsort(tf)]
    full = pp.revert_labels(torch.from_numpy(lab).cuda(), props, _PM(tf, ip), _LM(4)).cpu().numpy()
    assert np.array_equal(full, opre.revert_labels(lab, bbox, before, tb, 4))


def test_preprocess_refuses_what_it_does_not_implement():
    from fast_nnunet_amd.preprocess import DevicePreprocessor
    pp = DevicePreprocessor(torch.device('cuda', 0))
    raw = torch.ones(1, 8, 8, 8)
    with pytest.raises(NotImplementedError, match='resampling'):
        pp.run_case_npy(raw, None, {'spacing': [2.0, 1.0, 1.0]}, _PM((0, 1, 2), {'0': {}}), _CM(['NoNormalization']))
    cm = _CM(['ZScoreNormalization'])
    cm.use_mask_for_norm = [True]
    with pytest.raises(NotImplementedError, match='use_mask_for_norm'):
        pp.run_case_npy(raw, None, {'spacing': [1.0, 1.0, 1.0]}, _PM((0, 1, 2), {'0': {}}), cm)
    with pytest.raises(RuntimeError, match='Unable to locate class'):
        pp.run_case_npy(raw, None, {'spacing': [1.0, 1.0, 1.0]}, _PM((0, 1, 2), {'0': {}}), _CM(['FancyNorm']))
    with pytest.raises(RuntimeError, match='no CPU path'):
        DevicePreprocessor(torch.device('cpu'))
