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
    cm.use_mask_for_norm = list(case.get('use_mask', cm.use_mask_for_norm))
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
    with pytest.raises(RuntimeError, match='Unable to locate class'):
        pp.run_case_npy(raw, None, {'spacing': [1.0, 1.0, 1.0]}, _PM((0, 1, 2), {'0': {}}), _CM(['FancyNorm']))
    with pytest.raises(RuntimeError, match='no CPU path'):
        DevicePreprocessor(torch.device('cpu'))


# ---------------------------------------------------------------------------------------------------------------
# resampling (f-2 / f-3): device kernels vs the scipy-based restatement of skimage.transform.resize (oracle/resample.py;
# parity unpinned against skimage itself, which is not installed).  Both sides compute in fp64, so the results are the
# same fp32 / fp16 numbers except on rounding boundaries: tolerance 4 fp32 ulps of the data range / 1 fp16 ulp.
# ---------------------------------------------------------------------------------------------------------------
RESAMPLE_CASES = [
    # shape (C, ...), new_shape, order, separate axis
    ((2, 19, 23, 17), (25, 20, 30), 3, None),
    ((1, 40, 12, 33), (18, 12, 50), 3, None),            # one axis unchanged
    ((1, 7, 20, 22), (12, 31, 17), 3, 0),                # anisotropic: per-slice resize + order-0 along axis 0
    ((2, 16, 18, 6), (24, 11, 9), 3, 2),
    ((1, 9, 14, 11), (9, 21, 16), 3, 0),                 # separate axis keeps its length
    ((3, 13, 10, 12), (20, 15, 9), 1, None),
    ((1, 11, 11, 11), (7, 16, 5), 0, None),
]


@pytest.mark.parametrize('shape,new_shape,order,axis', RESAMPLE_CASES)
def test_resample_matches_scipy_restatement(shape, new_shape, order, axis):
    _resample_case(shape, new_shape, order, axis)


def _random_resample_case(seed):
    rs = np.random.RandomState(900 + seed)
    shape = (int(rs.randint(1, 4)),) + tuple(int(rs.randint(3, 41)) for _ in range(3))
    new_shape = tuple(int(max(2, round(d * rs.uniform(0.4, 2.2)))) for d in shape[1:])
    order = int(rs.choice([0, 1, 3, 3]))
    axis = None if rs.rand() < 0.5 else int(rs.randint(3))
    return shape, new_shape, order, axis


@pytest.mark.parametrize('seed', range(24))
def test_resample_matches_scipy_restatement_on_random_shapes(seed):
    """Up- and down-sampling by 0.4-2.2 per axis, orders 0 / 1 / 3, with and without the separate axis, drawn from a seed."""
    case = _random_resample_case(seed)
    print(case)
    _resample_case(*case)


def _resample_case(shape, new_shape, order, axis):
    from fast_nnunet_amd import capi
    from oracle import resample as ores
    rng = np.random.default_rng(hash((shape, new_shape, order)) % 2 ** 31)
    x = (rng.standard_normal(shape) * 50 + 10).astype(np.float32)
    want = ores.resample_data(x, new_shape, axis=axis, order=order, do_separate_z=axis is not None)
    xd = torch.from_numpy(x).cuda()
    out = torch.empty((shape[0], *new_shape), dtype=torch.float32, device='cuda')
    capi.resample(xd.data_ptr(), shape, new_shape, order, axis, False, out.data_ptr())
    got = out.cpu().numpy()
    assert np.abs(got - want).max() <= 4 * 6e-8 * np.abs(want).max(), np.abs(got - want).max()
    # fp16 logits (the probabilities path): output rounded to fp16 like reshaped_final.dtype = data.dtype
    xh = x.astype(np.float16)
    wanth = ores.resample_data(xh, new_shape, axis=axis, order=order, do_separate_z=axis is not None)
    assert wanth.dtype == np.float16
    outh = torch.empty((shape[0], *new_shape), dtype=torch.half, device='cuda')
    capi.resample(torch.from_numpy(xh).cuda().data_ptr(), shape, new_shape, order, axis, True, outh.data_ptr())
    goth = outh.cpu().numpy()
    ulp = np.spacing(np.abs(wanth).astype(np.float16)).astype(np.float32)
    assert (np.abs(goth.astype(np.float32) - wanth.astype(np.float32)) <= ulp).all()
    assert (goth != wanth).mean() < 5e-3            # exact half-ulp ties of the fp16 inputs' linear blends


def test_run_case_npy_with_resampling_and_export_round_trip():
    """The whole f-2 -> hot path stand-in -> f-3 chain against the oracle: transpose, crop, CT normalisation,
    order-3 resampling to the target spacing; then logits -> order-1 resampling back -> argmax -> un-crop -> transpose."""
    from fast_nnunet_amd.preprocess import DevicePreprocessor
    from oracle import resample as ores
    rng = np.random.default_rng(3)
    raw = (rng.standard_normal((1, 30, 44, 36)) * 300 + 200).astype(np.float32)
    raw[:, :4] = 0; raw[:, :, -6:] = 0
    tf = (1, 0, 2)
    ip = {'0': {'mean': 150.0, 'std': 280.0, 'percentile_00_5': -500.0, 'percentile_99_5': 900.0}}
    spacing = [3.0, 0.8, 0.8]                                  # of the raw axes; transposed: [0.8, 3.0, 0.8]
    pm, cm = _PM(tf, ip), _CM(['CTNormalization'], spacing=(1.0, 2.0, 1.0))
    props = {'spacing': spacing}
    pp = DevicePreprocessor(torch.device('cuda', 0))
    data, _, props = pp.run_case_npy(raw, None, props, pm, cm)
    ref, bbox, before = opre.preprocess_case(raw, tf, ['CTNormalization'], ip)
    sp_t = [spacing[i] for i in tf]
    new_shape = ores.compute_new_shape(ref.shape[1:], sp_t, cm.spacing)
    do_sep, axis = ores.determine_do_sep_z_and_axis(None, sp_t, cm.spacing)
    assert do_sep and axis == 1                                  # 3.0 / 0.8 > 3
    want = ores.resample_data(ref, new_shape, axis=axis, order=3, do_separate_z=do_sep)
    assert tuple(data.shape[1:]) == tuple(new_shape)
    assert np.abs(data.cpu().numpy() - want).max() <= 1e-5 * max(1.0, np.abs(want).max())

    # f-3: fake logits on the network grid -> label map on the raw grid
    class _P:
        label_manager = _LM(3)

        def convert_logits_to_segmentation(self, lg):
            return lg.float().argmax(0).to(torch.uint8)
    logits = torch.randn(4, *new_shape, generator=torch.Generator().manual_seed(1)).half()
    seg = pp.convert_predicted_logits_to_segmentation_with_correct_shape(logits.cuda(), _P(), pm, cm, props).cpu().numpy()
    back = ores.resample_data(logits.numpy(), ref.shape[1:], axis=axis, order=1, do_separate_z=do_sep)
    lab = back.astype(np.float32).argmax(0).astype(np.uint8)
    full = opre.revert_labels(lab, bbox, before, [int(i) for i in np.argsort(tf)], 3)
    assert seg.shape == raw.shape[1:] and (seg != full).mean() < 2e-3       # ties flip on fp16 rounding boundaries only


def test_masked_zscore_with_a_maze_of_background():
    """use_mask_for_norm on a volume whose background winds through the foreground: the sweeps have to carry
    "outside" around many corners before they converge; enclosed chambers stay inside (binary_fill_holes)."""
    from fast_nnunet_amd.preprocess import DevicePreprocessor
    rng = np.random.default_rng(12)
    shape = (1, 48, 40, 44)
    raw = (rng.standard_normal(shape) * 30 + 200).astype(np.float32)
    raw[raw == 0] = 1
    # a serpentine zero corridor entering from the border, plus sealed zero rooms
    for i in range(4, 44, 8):
        raw[0, i:i + 2, 2:38, 20:23] = 0
        raw[0, i:i + 10, (36 if (i // 8) % 2 == 0 else 2):(38 if (i // 8) % 2 == 0 else 4), 20:23] = 0
    raw[0, 0:6, 2:4, 20:23] = 0
    raw[0, 20:24, 10:14, 30:36] = 0                                  # sealed room
    raw[0, 30:33, 25:28, 5:9] = 0                                    # sealed room
    want, bbox, before = opre.preprocess_case(raw, (0, 1, 2), ['ZScoreNormalization'], {'0': {}}, [True])
    cm = _CM(['ZScoreNormalization'])
    cm.use_mask_for_norm = [True]
    pp = DevicePreprocessor(torch.device('cuda', 0))
    got, _, props = pp.run_case_npy(torch.from_numpy(raw), None, {'spacing': [1.0, 1.0, 1.0]}, _PM((0, 1, 2), {'0': {}}), cm)
    assert props['bbox_used_for_cropping'] == bbox
    g = got.cpu().numpy()
    assert np.abs(g - want).max() <= 2e-6 * max(1.0, np.abs(want).max())
    # outside the mask the zeros are untouched; inside (sealed rooms included) everything was normalised - an exact zero
    # there needs a voxel equal to the mean, which the oracle's fp32 mean and the kernel's fp64 mean may disagree on
    outside = ~opre.filled_nonzero_mask(raw)
    assert outside.sum() > 1000 and (g[0][outside] == 0).all() and (want[0][outside] == 0).all()
    assert (g[0][~outside] == 0).sum() <= 1 and (want[0][~outside] == 0).sum() <= 1


# ---- f-3 with return_probabilities=True (export_prediction.py:36-70)
@pytest.mark.parametrize('half', [True, False])
def test_export_probabilities_match_reference_golden(golden_dir, half):
    """Probabilities and labels on the original grid against vectors made by the reference's
    convert_predicted_logits_to_segmentation_with_correct_shape(..., return_probabilities=True).  The device's expf and
    division are not torch's CPU kernels: probabilities within 5e-7; labels equal wherever the top two reference
    probabilities differ by more than that (and in the crafted exact ties, where the first maximum wins)."""
    from golden_cases import DATASET_JSONS, EXPORT_CASES, export_case_logits
    from fast_nnunet_amd import capi
    from fast_nnunet_amd.plans import LabelManager
    z = np.load(os.path.join(golden_dir, 'export.npz'))
    for case in EXPORT_CASES:
        dj = DATASET_JSONS[case['dataset']]
        lm = LabelManager(dj['labels'], dj.get('regions_class_order'))
        H = lm.num_segmentation_heads
        logits = export_case_logits(case, H)
        lg = torch.from_numpy(logits).cuda()
        if not half:
            lg = lg.float()
        tb = [int(i) for i in np.argsort(case['tf'])]
        grid = [case['before'][j] for j in tb]
        probs = torch.empty((H, *grid), dtype=torch.float32, device='cuda')
        labels = torch.empty(grid, dtype=torch.uint8, device='cuda')
        capi.export_probabilities(lg.data_ptr(), half, H, dj.get('regions_class_order'), case['bbox'], case['before'], tb,
                                  probs.data_ptr(), labels.data_ptr(), False, torch.cuda.current_stream().cuda_stream)
        got_p, got_l = probs.cpu().numpy(), labels.cpu().numpy()
        # the uint16 label map (class values above 255: fnn_export_probabilities(..., FNN_LABEL_U16)) must hold the same labels
        probs16, labels16 = torch.empty_like(probs), torch.empty(grid, dtype=torch.int16, device='cuda')
        capi.export_probabilities(lg.data_ptr(), half, H, dj.get('regions_class_order'), case['bbox'], case['before'], tb,
                                  probs16.data_ptr(), labels16.data_ptr(), True, torch.cuda.current_stream().cuda_stream)
        assert torch.equal(probs16, probs) and np.array_equal(labels16.cpu().numpy().view(np.uint16), got_l.astype(np.uint16)), case['name']
        ref_p, ref_l = z[case['name'] + '__probs'], z[case['name'] + '__seg']
        assert np.abs(got_p - ref_p).max() <= 5e-7, case['name']
        if dj.get('regions_class_order') is None:
            top2 = np.sort(ref_p, 0)[-2:]
            decided = (top2[1] - top2[0] > 1e-6) | (top2[1] == top2[0])
        else:
            decided = (np.abs(ref_p - 0.5) > 1e-6).all(0)
        assert decided.mean() > 0.99 and np.array_equal(got_l[decided], ref_l[decided]), case['name']


def test_export_probabilities_rejects_bad_arguments():
    from fast_nnunet_amd import capi
    lg = torch.zeros((2, 4, 4, 4), dtype=torch.half, device='cuda')
    probs = torch.empty((2, 6, 6, 6), dtype=torch.float32, device='cuda')
    labels = torch.empty((6, 6, 6), dtype=torch.uint8, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    with pytest.raises(AssertionError, match='bbox outside'):
        capi.export_probabilities(lg.data_ptr(), True, 2, None, [[3, 7], [0, 4], [0, 4]], (6, 6, 6), (0, 1, 2), probs.data_ptr(),
                                  labels.data_ptr(), False, st)
    with pytest.raises(AssertionError, match='permutation'):
        capi.export_probabilities(lg.data_ptr(), True, 2, None, [[0, 4], [0, 4], [0, 4]], (6, 6, 6), (0, 1, 1), probs.data_ptr(),
                                  labels.data_ptr(), False, st)
    with pytest.raises(AssertionError, match='device pointers'):
        capi.export_probabilities(np.zeros(128, np.float16).ctypes.data, True, 2, None, [[0, 4], [0, 4], [0, 4]], (6, 6, 6),
                                  (0, 1, 2), probs.data_ptr(), labels.data_ptr(), False, st)


# ---------------------------------------------------------------------------------------------------------------
# segmentation resampling + cascade input (f-2 / f-4): resample_data_or_seg(is_seg=True) with batchgenerators'
# resize_segmentation (third-party, absent: oracle/resample.py restates its published algorithm - parity unpinned)
# ---------------------------------------------------------------------------------------------------------------
def _blobs(shape, n_labels, seed):
    """A label image of smooth random regions (argmax of smoothed noise fields)."""
    from scipy import ndimage as ndi
    rng = np.random.default_rng(seed)
    fields = np.stack([ndi.gaussian_filter(rng.standard_normal(shape), 2.5) for _ in range(n_labels + 1)])
    return fields.argmax(0).astype(np.int16)


@pytest.mark.parametrize('shape,new_shape,order,axis', [((20, 24, 18), (31, 20, 27), 1, None), ((9, 26, 22), (14, 40, 17), 1, 0),
                                                        ((16, 16, 12), (24, 11, 19), 0, None), ((12, 20, 7), (18, 30, 7), 3, 2)])
def test_segmentation_resampling_matches_the_restatement(shape, new_shape, order, axis):
    from fast_nnunet_amd.preprocess import DevicePreprocessor
    from oracle import resample as ores
    seg = _blobs(shape, 4, hash((shape, order)) % 1000)[None]
    want = ores.resample_seg(seg, new_shape, axis=axis, order=order, do_separate_z=axis is not None)
    pp = DevicePreprocessor(torch.device('cuda', 0))
    # spacings that make determine_do_sep_z_and_axis choose the wanted path
    cur = [1.0, 1.0, 1.0]
    if axis is not None:
        cur[axis] = 4.0
    got = pp.resample_seg(torch.from_numpy(seg).cuda(), new_shape, cur, cur,
                          {'is_seg': True, 'order': order, 'order_z': 0, 'force_separate_z': None}).cpu().numpy()
    assert got.shape == want.shape and got.dtype == np.int16
    # both sides resize in fp64 and threshold at 0.5.  A linearly interpolated binary mask lands on 0.5 at whole families
    # of rational positions; there the two summation orders round to either side (the device also stores the resized mask
    # in fp32 first): a few voxels per ten thousand may differ, nowhere else
    assert (got != want).mean() <= 1e-3, (got != want).mean()
    assert set(np.unique(got)) <= set(np.unique(seg))


@pytest.mark.parametrize('n_fg', [3, 9])
def test_cascade_input_one_hot_channels_through_the_predictor(n_fg):
    """predict_single_npy_array(image, props, segmentation_previous_stage): the previous stage's labels are cropped and
    resampled with the image and enter the network as one-hot channels (data_iterators.py:195-204); the result must equal
    running the predictor on the hand-built [image, one-hot] input.  9 foreground labels = 10 input channels: more than
    the stem stages at once (two channel groups)."""
    from fast_nnunet_amd import nnUNetPredictor
    from fast_nnunet_amd.plans import PlansManager
    from fast_nnunet_amd.preprocess import DevicePreprocessor
    from oracle import resample as ores
    from oracle.topology import UNetSpec
    from oracle.unet import synthetic_state_dict
    spec = UNetSpec('plain', 1 + n_fg, n_fg + 1, [16, 32], [(3, 3, 3)] * 2, [(1, 1, 1), (2, 2, 2)], [2, 2], [2])
    patch = (16, 16, 32)
    pm = PlansManager({'dataset_name': 'Dataset998_Cascade', 'plans_name': 'nnUNetPlans', 'transpose_forward': [0, 1, 2],
                       'transpose_backward': [0, 1, 2],
                       'foreground_intensity_properties_per_channel': {'0': {'mean': 0.0, 'std': 1.0, 'percentile_00_5': -5.0,
                                                                             'percentile_99_5': 5.0}},
                       'configurations': {'3d_lowres': {'patch_size': list(patch), 'spacing': [2.0, 2.0, 2.0]},
                                          '3d_cascade_fullres': {'patch_size': list(patch), 'spacing': [1.0, 1.0, 1.5],
                                                                 'previous_stage': '3d_lowres',
                                                                 'normalization_schemes': ['CTNormalization'],
                                                                 'use_mask_for_norm': [False],
                                                                 'resampling_fn_data_kwargs': {'is_seg': False, 'order': 3, 'order_z': 0, 'force_separate_z': None},
                                                                 'resampling_fn_seg_kwargs': {'is_seg': True, 'order': 1, 'order_z': 0, 'force_separate_z': None},
                                                                 'resampling_fn_probabilities_kwargs': {'is_seg': False, 'order': 1, 'order_z': 0, 'force_separate_z': None},
                                                                 'architecture': {'network_class_name': 'PlainConvUNet', 'arch_kwargs': {},
                                                                                  '_kw_requires_import': []}}}})
    cm = pm.get_configuration('3d_cascade_fullres')
    dj = {'labels': {'background': 0, **{f'l{i}': i for i in range(1, n_fg + 1)}}, 'channel_names': {'0': 'CT'}, 'file_ending': '.nii.gz'}
    p = nnUNetPredictor(tile_step_size=0.5, use_gaussian=True, use_mirroring=False, device=torch.device('cuda', 0),
                        allow_tqdm=False, patches_per_forward=2)
    p.manual_initialization(None, pm, cm, [synthetic_state_dict(spec, 12)], dj, 'nnUNetTrainer', None)
    rng = np.random.default_rng(5)
    image = rng.standard_normal((1, 30, 34, 40)).astype(np.float32)
    image[:, :3] = 0
    prev = _blobs(image.shape[1:], n_fg, 9)[None]
    props = {'spacing': [1.0, 1.0, 1.0]}
    labels = p.predict_single_npy_array(image, dict(props), prev)
    assert labels.shape == image.shape[1:] and labels.dtype == np.uint8
    # by hand: the same preprocessing, then the one-hot channels from the ORACLE's segmentation resampler
    pp = DevicePreprocessor(torch.device('cuda', 0))
    data, seg, pr = pp.run_case_npy(image, prev, dict(props), pm, cm, dj)
    bbox = pr['bbox_used_for_cropping']
    crop = prev[(slice(None), *[slice(lo, hi) for lo, hi in bbox])]
    want_seg = ores.resample_seg(crop, data.shape[1:], order=1)
    assert (seg.cpu().numpy() != want_seg).mean() <= 1e-3
    onehot = torch.stack([(seg[0] == l) for l in range(1, n_fg + 1)]).float()
    logits = p.predict_logits_from_preprocessed_data(torch.cat((data, onehot), 0))
    want = pp.convert_predicted_logits_to_segmentation_with_correct_shape(logits.cuda(), p, pm, cm, pr).cpu().numpy()
    assert np.array_equal(labels, want)
    # and the image-only call on a cascade network fails loudly instead of feeding one channel into a four-channel stem
    with pytest.raises((RuntimeError, AssertionError)):
        p.predict_single_npy_array(image, dict(props))
