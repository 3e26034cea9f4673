"""The oracle (CPU restatement) replayed against golden vectors that were
produced by the reference's own code (tests/golden/make_golden.py).  CPU only."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from golden_cases import SW_CASES, SW_CASES_2D, SW_CASES_HALF, DATASET_JSONS, make_case_inputs, make_case_networks
from oracle import sliding_window as osw
from oracle import topology as otopo


def _bits(t):
    return t.contiguous().view(torch.int16).numpy().view(np.uint16)


def test_tile_starts_match_reference(golden_dir):
    cases = json.load(open(os.path.join(golden_dir, 'steps.json')))
    assert len(cases) >= 50
    for c in cases:
        assert osw.tile_starts(c['image'], c['patch'], c['step']) == c['steps'], c


def test_tile_starts_known_answers():
    # the worked example in the reference's own comment (sliding_window_prediction.py:35-36)
    assert osw.tile_starts([110], [64], 0.5) == [[0, 23, 46]]
    assert osw.tile_starts([512], [128], 0.5) == [[0, 64, 128, 192, 256, 320, 384]]
    assert osw.tile_starts([512], [160], 0.5) == [[0, 70, 141, 211, 282, 352]]
    assert osw.tile_starts([128], [128], 0.5) == [[0]]


def test_gaussian_full_maps_bit_exact(golden_dir):
    z = np.load(os.path.join(golden_dir, 'gaussian.npz'))
    for key in z.files:
        patch = tuple(int(i) for i in key.split('_')[1:])
        got = _bits(osw.gaussian_weight(patch))
        assert np.array_equal(got, z[key]), key


def test_gaussian_large_maps_hash(golden_dir):
    for s in json.load(open(os.path.join(golden_dir, 'gaussian_summary.json'))):
        b = _bits(osw.gaussian_weight(tuple(s['patch'])))
        assert int(b.max()) == s['max_bits'] and int(b.min()) == s['min_bits']
        assert int((b == b.min()).sum()) == s['count_at_min']
        c = [i // 2 for i in s['patch']]
        assert b[:, c[1], c[2]].tolist() == s['line0']
        assert b[c[0], :, c[2]].tolist() == s['line1']
        assert b[c[0], c[1], :].tolist() == s['line2']
        assert hashlib.sha256(b.tobytes()).hexdigest() == s['sha256']


def test_gaussian_clamp_value():
    g = osw.gaussian_weight((128, 128, 128))
    assert float(g.max()) == 10.0
    assert float(g.min()) == pytest.approx(5.96e-8, rel=1e-2)      # smallest fp16 subnormal


def test_topology_planner_matches_reference(golden_dir):
    for c in json.load(open(os.path.join(golden_dir, 'topology.json'))):
        strides, kernels, npool = otopo.plan_pool_and_kernels(c['spacing'], c['patch'])
        assert [list(s) for s in strides] == c['strides'], c
        assert [list(k) for k in kernels] == c['kernels'], c
        assert npool == c['num_pool']


def _run_case(case, accum='fp16'):
    torch.set_num_threads(4)
    image = make_case_inputs(case)
    nets, _ = make_case_networks(case)
    kw = dict(step=case['step'], use_gaussian=case['gaussian'], mirror_axes=case['mirror'], accum=accum)
    if case['folds'] > 1 or case.get('via_folds', False):
        return osw.ensemble_logits(nets, image, case['patch'], case['heads'], **kw)
    return osw.sliding_window_logits(nets[0], image, case['patch'], case['heads'], **kw)


@pytest.mark.parametrize('case', [c for c in SW_CASES if c['kind'] == 'exact'], ids=lambda c: c['name'])
def test_sliding_window_exact_cases_bit_identical(case, golden_dir):
    z = np.load(os.path.join(golden_dir, 'sliding_window.npz'))
    out = _run_case(case)
    assert out.dtype == torch.half
    assert np.array_equal(_bits(out), z[case['name']])
    seg = osw.logits_to_labels(out).numpy().astype(np.int16)
    assert np.array_equal(seg, z[case['name'] + '__seg'])


@pytest.mark.parametrize('case', SW_CASES_HALF, ids=lambda c: c['name'])
def test_sliding_window_half_logit_cases_bit_identical(case, golden_dir):
    """The reference's GPU numerics (fp16 network output under autocast, predict_from_raw_data.py:591-593: mirror sums,
    `prediction *= gaussian` and the accumulation all in half precision), produced by the reference's own predictor on a
    CPU through fp16-output networks: the oracle's driver, given the same networks, matches bit for bit."""
    z = np.load(os.path.join(golden_dir, 'sliding_window_half.npz'))
    out = _run_case(case)
    assert out.dtype == torch.half
    assert np.array_equal(_bits(out), z[case['name']])
    seg = osw.logits_to_labels(out).numpy().astype(np.int16)
    assert np.array_equal(seg, z[case['name'] + '__seg'])


@pytest.mark.parametrize('case', [c for c in SW_CASES if c['kind'] == 'unet'], ids=lambda c: c['name'])
def test_sliding_window_unet_cases_close(case, golden_dir):
    z = np.load(os.path.join(golden_dir, 'sliding_window.npz'))
    out = _run_case(case).float().numpy()
    ref = torch.from_numpy(z[case['name']].view(np.int16)).view(torch.half).float().numpy()
    # same algorithm, same torch primitives; only the CPU's reduction order may differ.
    # Outside the 5.96e-8-weight border (where the reference's own fp16 accumulators
    # quantise to +-0.5) agreement is at fp16 resolution.
    err = np.abs(out - ref)
    assert np.median(err) < 1e-3
    assert (err > 0.02 * max(1.0, np.abs(ref).max())).mean() < 2e-3


def test_region_label_conversion(golden_dir):
    z = np.load(os.path.join(golden_dir, 'sliding_window.npz'))
    logits = torch.from_numpy(z[SW_CASES[0]['name']].view(np.int16)).view(torch.half)
    seg = osw.logits_to_labels(logits, DATASET_JSONS['regions']['regions_class_order'])
    assert np.array_equal(seg.numpy(), z['regions__seg'])


def test_fp32_accumulator_is_close_to_fp16_reference_away_from_border():
    case = SW_CASES[0]
    a = _run_case(case, 'fp16').float()
    b = _run_case(case, 'fp32')
    assert b.dtype == torch.float32
    inner = (slice(None), slice(4, -4), slice(4, -4), slice(4, -4))
    assert (a[inner] - b[inner]).abs().max() < 0.05


def test_inf_check_raises():
    net = lambda x: torch.full((1, 2, *x.shape[2:]), 7e4)           # overflows fp16 after weighting
    with pytest.raises(RuntimeError, match='Encountered inf'):
        osw.sliding_window_logits(net, torch.zeros(1, 16, 16, 16), (8, 8, 8), 2)


def test_ndim_assert():
    with pytest.raises(AssertionError):
        osw.sliding_window_logits(lambda x: x, torch.zeros(16, 16, 16), (8, 8, 8), 1)


def test_label_rules_match_reference_golden(golden_dir):
    """LabelManager.convert_logits_to_segmentation on every fp16 bit pattern, on fp32 logits around the
    sigmoid > 0.5 threshold, with class values above 255, and the plain argmax with ties and NaNs."""
    from golden_cases import label_rule_inputs
    z = np.load(os.path.join(golden_dir, 'label_rules.npz'))
    inp = label_rule_inputs()
    order = DATASET_JSONS['regions']['regions_class_order']
    assert np.array_equal(osw.logits_to_labels(inp['regions_f16'], order).numpy(), z['regions_f16'])
    assert np.array_equal(osw.logits_to_labels(inp['regions_f32'], order).numpy(), z['regions_f32'])
    order16 = DATASET_JSONS['regions_u16']['regions_class_order']
    assert np.array_equal(osw.logits_to_labels(inp['regions_f16'], order16).numpy(), z['regions_u16'])
    assert np.array_equal(osw.logits_to_labels(inp['argmax_f16']).numpy(), z['argmax_f16'])
    # the closed form the HIP kernels use: sigmoid(float(x)) > 0.5  <=>  x > 1.5 * 2^-24
    x = inp['regions_f16'].float()
    assert torch.equal(torch.sigmoid(x) > 0.5, x > 1.5 * 2.0 ** -24)
    x = inp['regions_f32']
    assert torch.equal(torch.sigmoid(x) > 0.5, x > 1.5 * 2.0 ** -24)


@pytest.mark.parametrize('case', SW_CASES_2D, ids=lambda c: c['name'])
def test_sliding_window_2d_configuration_matches_reference(case, golden_dir):
    """`2d` configurations (patch_size with two entries): every slice of the first axis is tiled
    (predict_from_raw_data.py:508-524), padding and Gaussian are 2-D, mirror axes index (y, z)."""
    z = np.load(os.path.join(golden_dir, 'sliding_window_2d.npz'))
    out = _run_case(case)
    assert out.dtype == torch.half and tuple(out.shape) == (case['heads'], *case['shape'])
    if case['kind'] == 'exact':
        assert np.array_equal(_bits(out), z[case['name']])
        assert np.array_equal(osw.logits_to_labels(out).numpy().astype(np.int16), z[case['name'] + '__seg'])
    else:
        ref = torch.from_numpy(z[case['name']].view(np.int16)).view(torch.half).float().numpy()
        err = np.abs(out.float().numpy() - ref)
        assert np.median(err) < 1e-3
        assert (err > 0.02 * max(1.0, np.abs(ref).max())).mean() < 2e-3


def test_preprocess_and_label_revert_match_reference_golden(golden_dir):
    """f-2 / f-3 restatements against vectors made by the reference's crop_to_nonzero, normalisation classes
    and export_prediction steps."""
    from golden_cases import PREP_CASES, prep_case_input, prep_label_input
    from oracle import preprocess as opre
    z = np.load(os.path.join(golden_dir, 'preprocess.npz'))
    for case in PREP_CASES:
        data, bbox, before = opre.preprocess_case(prep_case_input(case), case['tf'], case['schemes'], case['props'],
                                                  case.get('use_mask'))
        assert np.array_equal(np.asarray(bbox), z[case['name'] + '__bbox']), case['name']
        assert list(before) == z[case['name'] + '__shape_before'].tolist()
        ref = z[case['name'] + '__data']
        assert data.dtype == np.float32 and data.shape == ref.shape
        assert np.array_equal(data.view(np.uint32), ref.view(np.uint32)), case['name']
        tb = [int(i) for i in np.argsort(case['tf'])]
        for n_fg, tag in ((6, 'u8'), (300, 'u16')):
            lab = prep_label_input(case, data.shape[1:])
            full = opre.revert_labels(lab, bbox, before, tb, n_fg)
            want = z[case['name'] + '__labels_' + tag]
            assert full.dtype == want.dtype and np.array_equal(full, want)


def test_resampling_decisions_match_reference_golden(golden_dir):
    """compute_new_shape / determine_do_sep_z_and_axis of the oracle AND of the product's host mirror against the
    reference's own functions (default_resampling.py:14-71)."""
    from oracle import resample as ores
    from fast_nnunet_amd import preprocess as ppre
    for c in json.load(open(os.path.join(golden_dir, 'resample_logic.json'))):
        for mod in (ores, ppre):
            assert list(mod.compute_new_shape(c['shape'], c['current'], c['new'])) == c['new_shape'], c
            do_sep, axis = mod.determine_do_sep_z_and_axis(c['force'], c['current'], c['new'])
            assert bool(do_sep) == c['do_separate_z'] and (None if axis is None else int(axis)) == c['axis'], c


def test_resize_restatement_is_the_cubic_spline_it_claims():
    """The order-3 path of oracle/resample.py (scipy zoom, grid_mode) against an independent evaluation of the same
    definition: 12 edge-padded samples, B-spline coefficients by the truncated impulse response of the recursive
    filter (what the HIP kernel does), 4 taps per axis at x = (o + 0.5) in/out - 0.5."""
    from oracle import resample as ores
    rng = np.random.default_rng(5)
    x = rng.standard_normal((9, 12, 7)) * 10
    out_shape = (13, 8, 11)
    z = np.sqrt(3.0) - 2.0
    K = 30
    h = np.array([(-6 * z / (1 - z * z)) * z ** abs(k) for k in range(-K, K + 1)])
    p = np.pad(x, 12, mode='edge')
    for ax in range(3):
        n = p.shape[ax]
        idx = np.mod(np.arange(n)[:, None] + np.arange(-K, K + 1)[None, :], 2 * n - 2)
        idx = np.where(idx >= n, 2 * n - 2 - idx, idx)
        p = np.moveaxis((np.moveaxis(p, ax, -1)[..., idx] * h).sum(-1), -1, ax)
    res = np.zeros(out_shape)
    tabs = []
    for d in range(3):
        zoom = x.shape[d] / out_shape[d]
        c = np.arange(out_shape[d]) * zoom + 0.5 * zoom - 0.5 + 12
        f = np.floor(c)
        t = c - f
        tabs.append((f.astype(int) - 1, np.stack([(1 - t) ** 3 / 6, (3 * t ** 3 - 6 * t ** 2 + 4) / 6,
                                                  (-3 * t ** 3 + 3 * t ** 2 + 3 * t + 1) / 6, t ** 3 / 6], -1)))
    for a in range(4):
        for b in range(4):
            for c in range(4):
                res += (tabs[0][1][:, a][:, None, None] * tabs[1][1][:, b][None, :, None] * tabs[2][1][:, c][None, None, :]) * \
                    p[np.ix_(tabs[0][0] + a, tabs[1][0] + b, tabs[2][0] + c)]
    res = np.clip(res, x.min(), x.max())
    assert np.abs(res - ores.skimage_resize(x, out_shape, 3)).max() < 1e-10


def test_export_with_probabilities_matches_reference_golden(golden_dir):
    """f-3 with ``return_probabilities=True``: vectors made by the reference's
    convert_predicted_logits_to_segmentation_with_correct_shape (no resampling needed)."""
    from golden_cases import DATASET_JSONS, EXPORT_CASES, export_case_logits
    from fast_nnunet_amd.plans import LabelManager
    from oracle import preprocess as opre
    z = np.load(os.path.join(golden_dir, 'export.npz'))
    for case in EXPORT_CASES:
        dj = DATASET_JSONS[case['dataset']]
        lm = LabelManager(dj['labels'], dj.get('regions_class_order'))
        logits = export_case_logits(case, lm.num_segmentation_heads)
        tb = [int(i) for i in np.argsort(case['tf'])]
        seg, probs = opre.export_with_probabilities(logits, case['bbox'], case['before'], tb, len(lm.foreground_labels),
                                                    dj.get('regions_class_order'))
        assert seg.dtype == z[case['name'] + '__seg'].dtype and np.array_equal(seg, z[case['name'] + '__seg']), case['name']
        ref = z[case['name'] + '__probs']
        assert probs.shape == ref.shape and probs.dtype == np.float32
        assert np.abs(probs - ref).max() <= 2e-7, case['name']          # torch's softmax kernel varies with the CPU's ISA
        outside = np.ones(ref.shape[1:], bool)
        tf = list(case['tf'])
        outside_t = np.ones(case['before'], bool)
        outside_t[tuple(slice(lo, hi) for lo, hi in case['bbox'])] = False
        outside = outside_t.transpose(tb)
        assert np.array_equal(probs[:, outside], ref[:, outside])


def test_box_restricted_driver_equals_the_full_driver_on_the_box():
    """``sliding_window_logits_box`` (used by the full-size GPU tests, where the whole volume would cost the CPU half
    an hour) must be the full driver restricted to the box, bit for bit - corner, interior and ragged boxes."""
    import torch
    from oracle import sliding_window as osw
    g = torch.Generator().manual_seed(3)
    w = torch.randn(3, 1, 3, 3, 3, generator=g)

    def net(x):
        return torch.nn.functional.conv3d(x, w, padding=1)

    image = torch.randn(1, 40, 36, 50, generator=g)
    patch = (16, 12, 20)
    full = osw.sliding_window_logits(net, image, patch, 3, step=0.5, accum='fp16')
    for box in (((0, 8), (0, 6), (0, 10)), ((13, 22), (9, 17), (21, 33)), ((30, 40), (0, 36), (45, 50))):
        got, n = osw.sliding_window_logits_box(net, image, patch, 3, box, step=0.5, accum='fp16')
        want = full[(slice(None), *[slice(a, b) for a, b in box])]
        assert n >= 1 and got.dtype == torch.half
        assert torch.equal(got.view(torch.int16), want.contiguous().view(torch.int16))
