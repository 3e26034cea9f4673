"""End-to-end parity on a real MI355X, through the drop-in ``nnUNetPredictor`` and
the C ABI underneath it.

* network forward vs the fp32 CPU oracle (fp16-MFMA tolerance, stated below);
* the sliding-window driver vs the oracle driver fed with the ENGINE's own
  per-patch logits: pad / tile starts / visit order / Gaussian / fp16
  round-to-nearest-even accumulation / normalise / un-pad / mirroring / folds
  must then agree BIT FOR BIT in reference-rounding mode;
* the reference's own golden volumes (tests/golden, produced by the reference
  predictor) within the fp16 tolerance, label maps compared where the logit
  margin exceeds the measured error;
* edge cases the reference handles: image smaller than the patch, image equal
  to the patch, step 1.0, no Gaussian, inf detection, argument assertions.

Tolerance (network in fp16 on the matrix cores vs fp32 on the CPU):
max |err| <= 6e-3 * max|ref| and relative RMSE <= 3.5e-3 (MAX_REL, RMSE_REL below).
Measured on MI355X: max ratio 0.9e-3 .. 1.9e-3, relative RMSE 0.8e-3 .. 1.8e-3 over
the small topologies below, 2.2-3.4e-3 / 1.7-2.4e-3 on the BASELINE configurations at
full patch size; values are printed with -s.
"""
import os

import numpy as np
import pytest
import torch

from golden_cases import SW_CASES, SW_CASES_2D, make_case_inputs, make_case_networks, toy_unet_spec, toy_unet_spec_2d
from oracle import sliding_window as osw
from oracle.topology import UNetSpec, student_spec
from oracle.unet import build as build_oracle, synthetic_state_dict

pytestmark = pytest.mark.gpu

# measured on the BASELINE configurations at full patch size: max ratio 2.2-3.4e-3, relative RMSE 1.7-2.4e-3
MAX_REL, RMSE_REL = 6e-3, 3.5e-3


def _bits(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def _plans(patch):
    from fast_nnunet_amd.plans import PlansManager
    return PlansManager({'dataset_name': 'Dataset999_Golden', 'plans_name': 'nnUNetPlans',
                         'configurations': {'3d_fullres': {'patch_size': list(patch), 'architecture': {
                             'network_class_name': 'PlainConvUNet', 'arch_kwargs': {}, '_kw_requires_import': []}}}})


def _predictor(spec: UNetSpec, patch, state_dicts, mirror=None, step=0.5, gaussian=True, accumulate_in='fp16',
               batch=3, on_device=True, dataset_json=None):
    from fast_nnunet_amd import nnUNetPredictor
    pm = _plans(patch)
    cm = pm.get_configuration('3d_fullres')
    dj = dataset_json or {'labels': {('background' if i == 0 else f'c{i}'): i for i in range(spec.num_heads)},
                          'channel_names': {str(i): 'CT' for i in range(spec.in_channels)}, 'file_ending': '.nii.gz'}
    p = nnUNetPredictor(tile_step_size=step, use_gaussian=gaussian, use_mirroring=mirror is not None,
                        perform_everything_on_device=on_device, device=torch.device('cuda', 0), verbose=False,
                        allow_tqdm=False, accumulate_in=accumulate_in, patches_per_forward=batch)
    p.manual_initialization(None, pm, cm, list(state_dicts), dj, 'nnUNetTrainer',
                            tuple(mirror) if mirror is not None else None)
    return p


def _report(name, got, ref):
    err = (got - ref).abs()
    mx, rmse = float(err.max()), float(err.pow(2).mean().sqrt())
    scale, rms_ref = float(ref.abs().max()), float(ref.pow(2).mean().sqrt())
    print(f'[{name}] max|err| {mx:.4g} (ref max {scale:.4g}, ratio {mx / scale:.3g})  '
          f'rmse {rmse:.4g} (rel {rmse / rms_ref:.3g})')
    return mx / scale, rmse / rms_ref


SPECS = {
    'toy3': (toy_unet_spec(1, 3), (16, 16, 32)),
    'toy3_2ch': (toy_unet_spec(2, 2), (16, 32, 16)),
    'aniso5': (UNetSpec('plain', 1, 5, [16, 32, 64, 64], [(1, 3, 3), (3, 3, 3), (3, 3, 3), (3, 3, 3)],
                        [(1, 1, 1), (1, 2, 2), (2, 2, 2), (2, 1, 1)], [2, 2, 2, 2], [2, 2, 2]), (16, 32, 32)),
    'r6_odd_channels': (UNetSpec('plain', 1, 4, [8, 10, 21], [(3, 3, 3)] * 3, [(1, 1, 1), (2, 2, 2), (2, 2, 2)],
                                 [2, 2, 2], [2, 2]), (16, 16, 16)),
    'heads61': (UNetSpec('plain', 1, 61, [16, 32], [(3, 3, 3)] * 2, [(1, 1, 1), (2, 2, 2)], [2, 2], [2]), (16, 16, 32)),
    'one_conv_per_stage': (UNetSpec('plain', 1, 2, [16, 32, 32], [(3, 3, 3)] * 3, [(1, 1, 1), (2, 2, 2), (2, 2, 2)],
                                    [1, 1, 1], [1, 1]), (16, 16, 16)),
    # ResidualEncoderUNet: projections with and without pooling, identity skips, strided identity skip
    'resenc4': (UNetSpec('resenc', 1, 3, [16, 32, 32, 48], [(1, 3, 3), (3, 3, 3), (3, 3, 3), (3, 3, 3)],
                         [(1, 1, 1), (1, 2, 2), (2, 2, 2), (2, 1, 1)], [1, 3, 2, 2], [1, 1, 1]), (16, 32, 32)),
    # more input channels than the stem stages at once (a cascade stage: image + one one-hot channel per foreground label,
    # label_handling.py:294-311): groups of 8 channels, 11 = 8 + 3, 17 = 8 + 8 + 1; an anisotropic stem kernel too
    'in11': (UNetSpec('plain', 11, 3, [16, 32], [(3, 3, 3)] * 2, [(1, 1, 1), (2, 2, 2)], [2, 2], [2]), (16, 16, 32)),
    'in17_aniso': (UNetSpec('plain', 17, 2, [16, 32], [(1, 3, 3), (3, 3, 3)], [(1, 1, 1), (1, 2, 2)], [2, 2], [2]), (8, 32, 32)),
    'resenc_2ch_odd': (UNetSpec('resenc', 2, 4, [8, 10, 21], [(3, 3, 3)] * 3, [(1, 1, 1), (2, 2, 2), (2, 2, 2)],
                                [2, 2, 1], [1, 1]), (16, 16, 16)),
}


@pytest.mark.parametrize('name', list(SPECS))
def test_network_forward_matches_fp32_oracle(name):
    spec, patch = SPECS[name]
    sd = synthetic_state_dict(spec, 1234)
    net = build_oracle(spec, sd)
    p = _predictor(spec, patch, [sd])
    g = torch.Generator().manual_seed(5)
    x = torch.randn(5, spec.in_channels, *patch, generator=g)      # 5 patches, batch 3 -> ragged last batch
    got = p.forward_patches(x).cpu()
    torch.set_num_threads(4)
    with torch.inference_mode():
        ref = net(x)
    mr, rr = _report(name, got, ref)
    assert mr <= MAX_REL and rr <= RMSE_REL
    # batching must not change results beyond the statistics' summation order
    single = p.forward_patches(x[3:4]).cpu()
    assert (single - got[3:4]).abs().max() <= 1e-3 * float(ref.abs().max())


def _random_spec(seed):
    """A U-Net of the family the reference builds from a plans file (PlainConvUNet / ResidualEncoderUNet: get_network_from_plans),
    drawn from a seed: stages, widths (the r = 3 / 6 students' odd ones too), anisotropic kernels and strides, convs per stage."""
    rs = np.random.RandomState(700 + seed)
    kind = 'resenc' if rs.rand() < 0.25 else 'plain'
    stages = int(rs.randint(2, 6))
    base = int(rs.choice([8, 10, 16, 16, 24, 32]))
    cap = int(rs.choice([64, 96, 160]))
    feats = [min(int(base * 2 ** i * (1.0 if rs.rand() < 0.7 else 1.06)), cap) for i in range(stages)]
    strides = [(1, 1, 1)] + [[(2, 2, 2), (2, 2, 2), (1, 2, 2), (2, 1, 1)][rs.randint(4)] for _ in range(stages - 1)]
    kernels = [(1, 3, 3) if i < 2 and rs.rand() < 0.3 else (3, 3, 3) for i in range(stages)]
    enc = [int(rs.randint(1, 4)) for _ in range(stages)]
    dec = [int(rs.randint(1, 3)) for _ in range(stages - 1)]
    spec = UNetSpec(kind, int(rs.choice([1, 1, 2, 4])), int(rs.choice([2, 3, 5, 9, 17])), feats, kernels, strides, enc, dec)
    total = [int(np.prod([s[a] for s in strides])) for a in range(3)]
    patch = [t * int(rs.randint(2, 9)) for t in total]       # the bottleneck keeps at least 2 voxels per axis (nnU-Net's own plans: 4)
    while patch[0] * patch[1] * patch[2] > 150000:
        a = int(np.argmax([patch[i] / total[i] for i in range(3)]))
        if patch[a] <= 2 * total[a]:
            break
        patch[a] -= total[a]
    return spec, tuple(patch)


@pytest.mark.parametrize('seed', range(20))
def test_random_topologies_match_fp32_oracle(seed):
    """Networks nobody wrote a launch rule for: whatever the plan picks per layer, the logits are the oracle's within the
    tolerance of the fixed topologies above, and a patch's result does not depend on its batch."""
    spec, patch = _random_spec(seed)
    sd = synthetic_state_dict(spec, 4000 + seed)
    net = build_oracle(spec, sd)
    p = _predictor(spec, patch, [sd])
    x = torch.randn(4, spec.in_channels, *patch, generator=torch.Generator().manual_seed(seed))     # batch 3 -> ragged last batch
    got = p.forward_patches(x).cpu()
    torch.set_num_threads(8)
    with torch.inference_mode():
        ref = net(x)
    mr, rr = _report(f'seed {seed}: {spec.kind} {spec.features} k {spec.kernels} s {spec.strides} convs {spec.n_conv_enc} / '
                     f'{spec.n_conv_dec}, {spec.in_channels} -> {spec.num_heads}, patch {patch}', got, ref)
    # measured over the 20 seeds: max ratio 0.8-3.0e-3, relative RMSE 0.7-2.5e-3, and 4.6e-3 / 3.5e-3 for seed 13 (19 convs, three per
    # encoder stage, over a bottleneck of 108 voxels: InstanceNorm over so few voxels amplifies the fp16 rounding of its input -
    # nnU-Net's own plans keep 4 voxels per axis and two convs per stage): 1.5 x the tolerance of the fixed topologies
    assert mr <= 1.5 * MAX_REL and rr <= 1.5 * RMSE_REL
    single = p.forward_patches(x[3:4]).cpu()
    assert (single - got[3:4]).abs().max() <= 1e-3 * float(ref.abs().max())


def test_forward_through_the_depth_shift_strided_kernel_matches_oracle():
    """An anisotropic net (first down-sampling (1, 2, 2), like the benchmark's) at a patch large enough for
    conv3d_zs_kernel to take that conv (it needs >= 768 tiles; the small cases above keep the linear-tap kernels):
    64 x 96 x 128 -> 8 x 12 x 8 tiles x batch 2, plus a ragged second patch size."""
    spec = UNetSpec('plain', 1, 3, [16, 32], [(1, 3, 3), (3, 3, 3)], [(1, 1, 1), (1, 2, 2)], [2, 2], [2])
    sd = synthetic_state_dict(spec, 77)
    net = build_oracle(spec, sd)
    torch.set_num_threads(8)
    for patch in ((64, 96, 128), (40, 112, 184)):
        p = _predictor(spec, patch, [sd], batch=2)
        x = torch.randn(2, 1, *patch, generator=torch.Generator().manual_seed(3))
        got = p.forward_patches(x).cpu()
        with torch.inference_mode():
            ref = net(x)
        mr, rr = _report(f'zs_{patch}', got, ref)
        assert mr <= MAX_REL and rr <= RMSE_REL


def test_c1_sized_student_forward_matches_oracle():
    """BASELINE config 1 topology (PlainConv r=2, 6 stages) at a reduced 64^3 patch so the CPU oracle is quick."""
    spec = student_spec((1.0, 1.0, 1.0), (128, 128, 128), 1, 2, reduction=2)
    assert spec.features == [16, 32, 64, 128, 160, 160]
    patch = (64, 64, 64)
    sd = synthetic_state_dict(spec, 1234)
    p = _predictor(spec, patch, [sd], batch=2)
    x = torch.randn(2, 1, *patch, generator=torch.Generator().manual_seed(0))
    got = p.forward_patches(x).cpu()
    torch.set_num_threads(8)
    with torch.inference_mode():
        ref = build_oracle(spec, sd)(x)
    mr, rr = _report('c1_64', got, ref)
    assert mr <= MAX_REL and rr <= RMSE_REL
    assert (got.argmax(1) != ref.argmax(1)).float().mean() < 5e-3


def test_forward_is_bit_stable_while_another_engine_shares_the_gpu():
    """Two engines driven from two host threads on two streams: every forward must reproduce the engine's
    solo result bit for bit.  (Found a stem-conv build whose accumulators went wrong when other kernels
    shared the CU - DESIGN.md section 3; the engine's own two-batches-in-flight pipelining relies on this.)"""
    import threading
    spec, patch = SPECS['toy3']
    sd = synthetic_state_dict(spec, 50)
    ps = [_predictor(spec, patch, [sd], batch=3) for _ in range(2)]
    xs = [torch.randn(3, spec.in_channels, *patch, generator=torch.Generator().manual_seed(9 + i)) for i in range(2)]
    solo = [ps[i].forward_patches(xs[i]).cpu() for i in range(2)]
    bad = [0, 0]

    def work(i):
        with torch.cuda.stream(torch.cuda.Stream()):
            for _ in range(40):
                bad[i] += int(not torch.equal(ps[i].forward_patches(xs[i]).cpu(), solo[i]))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert bad == [0, 0]


def _driver_case(case, mode, spec, patch):
    sds = [synthetic_state_dict(spec, 50 + f) for f in range(case['folds'])]
    p = _predictor(spec, patch, sds, mirror=case['mirror'], step=case['step'], gaussian=case['gaussian'], accumulate_in=mode)
    image = torch.randn(1, *case['shape'], generator=torch.Generator().manual_seed(9))

    def engine_net(fold):
        def f(x):
            p._active_fold = fold
            y = p.forward_patches(x).cpu()
            return y.half() if mode == 'fp16_autocast' else y
        return f

    nets = [engine_net(f) for f in range(case['folds'])]
    kw = dict(step=case['step'], use_gaussian=case['gaussian'], mirror_axes=case['mirror'], accum='fp16')
    if case['folds'] > 1:
        want = osw.ensemble_logits(nets, image, patch, spec.num_heads, **kw)
        got = p.predict_logits_from_preprocessed_data(image)
        assert got.device.type == 'cpu'
    else:
        want = osw.sliding_window_logits(nets[0], image, patch, spec.num_heads, **kw)
        p._active_fold = 0
        got = p.predict_sliding_window_return_logits(image)
        assert got.device.type == 'cuda'
    assert got.dtype == torch.half and tuple(got.shape) == tuple(want.shape)
    gb, wb = _bits(got), _bits(want)
    same = (gb == wb).mean()
    print(f'bit-identical fraction {same:.6f}')
    assert same == 1.0


DRIVER_CASES = [
    dict(shape=(40, 36, 44), mirror=None, step=0.5, gaussian=True, folds=1),
    dict(shape=(40, 36, 44), mirror=None, step=1.0, gaussian=False, folds=1),
    dict(shape=(11, 30, 9), mirror=None, step=0.5, gaussian=True, folds=1),        # smaller than the patch -> padded
    dict(shape=(16, 16, 32), mirror=None, step=0.5, gaussian=True, folds=1),       # exactly one patch
    dict(shape=(24, 33, 40), mirror=[0], step=0.5, gaussian=True, folds=1),
    dict(shape=(20, 18, 47), mirror=[0, 1, 2], step=0.5, gaussian=True, folds=1),
    dict(shape=(33, 20, 37), mirror=[1, 2], step=0.3, gaussian=True, folds=3),
    # (image - patch) / (patch * step) lands on an integer: ceil() of the quotient formed with a float-rounded step
    # gives one tile position more (72: 6 instead of 5; 44 and 88: 6 instead of 5) - fnn_opts carries a double
    dict(shape=(72, 20, 40), mirror=None, step=0.7, gaussian=True, folds=1),
    dict(shape=(44, 16, 88), mirror=None, step=0.35, gaussian=True, folds=1),
]


@pytest.mark.parametrize('mode', ['fp16', 'fp16_autocast'])
@pytest.mark.parametrize('case', DRIVER_CASES, ids=lambda c: f"{c['shape']}-m{c['mirror']}-s{c['step']}-f{c['folds']}")
def test_driver_bit_identical_to_oracle_driver_on_engine_logits(case, mode):
    """'fp16': the reference without autocast (fp32 logits; pinned by tests/golden/sliding_window.npz).
    'fp16_autocast': the reference on a GPU - the network returns fp16, so the oracle's driver (the reference's own
    torch statements, pinned for fp16-output networks by tests/golden/sliding_window_half.npz) rounds the mirror sums,
    the Gaussian product and the accumulation to fp16; the engine's FNN_ACC_FP16_AUTOCAST must give the same bits."""
    _driver_case(case, mode, *SPECS['toy3'])


def _random_driver_case(seed):
    rs = np.random.RandomState(300 + seed)
    name = 'heads61' if rs.rand() < 0.35 else 'toy3'
    patch = SPECS[name][1]
    shape = tuple(int(rs.randint(max(3, q // 2), int(3.2 * q) + 1)) for q in patch)
    axes = [a for a in range(3) if rs.rand() < 0.5]
    return name, dict(shape=shape, mirror=axes if axes and rs.rand() < 0.6 else None,
                      step=float(rs.choice([0.25, 0.3, 0.4, 0.5, 0.5, 0.6, 0.75, 1.0])), gaussian=bool(rs.rand() < 0.85),
                      folds=int(rs.choice([1, 1, 2]))), ['fp16', 'fp16_autocast'][rs.randint(2)]


@pytest.mark.parametrize('seed', range(24))
def test_driver_bit_identical_to_oracle_driver_on_random_volumes(seed):
    """Volume shapes from half a patch to 3.2 patches per axis, tile steps 0.25-1.0, every subset of mirror axes, one or two
    folds, 3 or 61 heads, both accumulation arithmetics, drawn from a seed: the same bits as the oracle's driver."""
    name, case, mode = _random_driver_case(seed)
    print(name, case, mode)
    _driver_case(case, mode, *SPECS[name])


@pytest.mark.parametrize('shape,folds,accum', [((24, 40, 64), 1, 'fp16'), ((19, 23, 72), 2, 'fp16'), ((17, 16, 40), 1, 'fp32')])
def test_driver_with_61_heads_uses_the_tiled_finalize(shape, folds, accum):
    """61 heads -> 64-channel accumulator rows: the vectorised seg head and the LDS-tiled finalize kernel
    (aligned and ragged z extents, fold ensembling through the add mode) against the oracle driver."""
    spec, patch = SPECS['heads61']
    sds = [synthetic_state_dict(spec, 70 + f) for f in range(folds)]
    p = _predictor(spec, patch, sds, accumulate_in=accum)
    image = torch.randn(1, *shape, generator=torch.Generator().manual_seed(13))

    def engine_net(fold):
        def f(x):
            p._active_fold = fold
            return p.forward_patches(x).cpu()
        return f

    nets = [engine_net(f) for f in range(folds)]
    if folds > 1:
        want = osw.ensemble_logits(nets, image, patch, spec.num_heads, accum=accum)
        got = p.predict_logits_from_preprocessed_data(image).cpu()
    else:
        want = osw.sliding_window_logits(nets[0], image, patch, spec.num_heads, accum=accum)
        p._active_fold = 0
        got = p.predict_sliding_window_return_logits(image).cpu()
    if accum == 'fp16':
        assert (_bits(got) == _bits(want)).all()
    else:
        assert (got.float() - want.float()).abs().max() <= 2e-3 * float(want.abs().max()) + 1e-3


@pytest.mark.parametrize('accum', ['fp16', 'fp32'])
@pytest.mark.parametrize('heads,features0,shape', [(15, 16, (21, 27, 50)), (61, 16, (24, 40, 64)), (70, 16, (17, 16, 40)),
                                                   (130, 16, (17, 16, 40)), (3, 48, (21, 27, 50))])
def test_accumulate_path_kernels_against_the_oracle_driver(heads, features0, shape, accum):
    """The whole-volume accumulator path (FNN_NO_GATHER; what a network whose last stage has more than 32 channels takes by
    itself: 48 here - the multi-k-step `seg_head_acc_kernel`) against the oracle driver on the engine's logits: the
    cooperative label kernel at 2 / 8 / 16 / 32 lanes per voxel (15 / 61 / 70 / 130 classes), the LDS-tiled finalize kernel for
    64-channel accumulator rows with an aligned z extent (61 classes - since the gather path became the default no test
    reached it), both buffer arithmetics, two folds through the add mode."""
    spec = UNetSpec('plain', 1, heads, [features0, 32], [(3, 3, 3)] * 2, [(1, 1, 1), (2, 2, 2)], [2, 2], [2])
    patch = (16, 16, 32)
    sds = [synthetic_state_dict(spec, 610 + f) for f in range(2)]
    os.environ['FNN_NO_GATHER'] = '1'
    try:
        p, p1 = _predictor(spec, patch, sds, accumulate_in=accum), _predictor(spec, patch, sds[:1], accumulate_in=accum)
    finally:
        os.environ.pop('FNN_NO_GATHER', None)
    image = torch.randn(1, *shape, generator=torch.Generator().manual_seed(43))

    def net(q, fold):
        def f(x):
            q._active_fold = fold
            return q.forward_patches(x).cpu()
        return f

    want1 = osw.sliding_window_logits(net(p1, 0), image, patch, heads, accum=accum)
    want2 = osw.ensemble_logits([net(p, 0), net(p, 1)], image, patch, heads, accum=accum)
    got1, got2 = p1.predict_sliding_window_return_logits(image).cpu(), p.predict_logits_from_preprocessed_data(image).cpu()
    if accum == 'fp16':
        assert (_bits(got1) == _bits(want1)).all() and (_bits(got2) == _bits(want2)).all()
    else:
        assert (got1.float() - want1.float()).abs().max() <= 2e-3 * float(want1.abs().max()) + 1e-3
        assert (got2.float() - want2.float()).abs().max() <= 2e-3 * float(want2.abs().max()) + 1e-3
    labels = p1.predict_segmentation_from_preprocessed_data(image).cpu().long()
    if accum == 'fp16':
        assert torch.equal(labels, osw.logits_to_labels(want1.float()).long())
    else:                                                               # fp32 sums: only ties of the rounded logits may differ
        assert float((labels != osw.logits_to_labels(want1.float()).long()).float().mean()) <= 2e-3


@pytest.mark.parametrize('gather', [True, False])
@pytest.mark.parametrize('heads,accum', [(3, 'fp32'), (61, 'fp32'), (61, 'fp16')])
def test_fp32_logits_through_the_c_abi_round_to_the_fp16_logits(heads, accum, gather):
    """fnn_opts.out_dtype = FNN_OUT_F32 (the Python predictor always asks for the reference's fp16 logits, so only the C ABI reaches
    it): the quotient sum / weight sum before its rounding to fp16 (fp32 sums), or the fp16 logits widened (fp16 sums: they ARE the
    reference's values).  Either way fp16(out32) must be the fp16 output's bits.  The gather kernel writes fp16 only: an engine on the gather
    path takes the accumulators for such a call - the plain and the LDS-tiled (61 classes, aligned z) finalize kernels."""
    from fast_nnunet_amd import capi
    spec = UNetSpec('plain', 1, heads, [16, 32], [(3, 3, 3)] * 2, [(1, 1, 1), (2, 2, 2)], [2, 2], [2])
    patch = (16, 16, 32)
    if not gather:
        os.environ['FNN_NO_GATHER'] = '1'
    try:
        p = _predictor(spec, patch, [synthetic_state_dict(spec, 650)], accumulate_in=accum)
    finally:
        os.environ.pop('FNN_NO_GATHER', None)
    image = torch.randn(1, 24, 40, 64, generator=torch.Generator().manual_seed(47))
    want = p.predict_sliding_window_return_logits(image)
    x = image.to('cuda', torch.float32).contiguous()
    out = torch.empty((heads, *image.shape[1:]), dtype=torch.float32, device='cuda')
    o = p._opts()
    o.out_dtype = capi.FNN_OUT_F32
    p._engine.predict_volume(x.data_ptr(), x.shape, o, out.data_ptr())
    torch.cuda.synchronize()
    assert (_bits(out.half()) == _bits(want)).all()
    if accum == 'fp32':
        assert not torch.equal(out, out.half().float())                   # really un-rounded


@pytest.mark.parametrize('heads', [15, 16, 63, 64, 130])
def test_driver_with_head_counts_around_a_block_boundary(heads):
    """The weight-sum channel is a seg head of its own (zero weights, bias 1) right after the last class: with 15 / 63
    classes it fills the last slot of a 16-channel head block, with 16 / 64 it opens a new block; 64 and 130 classes run
    the gather kernel in two / three passes of <= 63 heads over the kept activations (round 3: the reference has no class
    limit, predict_from_raw_data.py:587-590) and take the labels from those logits.  Driver vs the oracle driver on the engine's logits, bit for bit;
    labels straight from the accumulators vs argmax of those logits."""
    spec = UNetSpec('plain', 1, heads, [16, 32], [(3, 3, 3)] * 2, [(1, 1, 1), (2, 2, 2)], [2, 2], [2])
    patch = (16, 16, 32)
    p = _predictor(spec, patch, [synthetic_state_dict(spec, 90 + heads)])
    image = torch.randn(1, 21, 27, 50, generator=torch.Generator().manual_seed(17))
    want = osw.sliding_window_logits(lambda x: p.forward_patches(x).cpu(), image, patch, heads, accum='fp16')
    got = p.predict_sliding_window_return_logits(image).cpu()
    assert (_bits(got) == _bits(want)).all()
    labels = p.predict_segmentation_from_preprocessed_data(image).cpu()
    assert torch.equal(labels.long(), osw.logits_to_labels(want.float()).long())


def test_fp32_accumulators_match_exact_blend():
    spec, patch = SPECS['toy3']
    sd = synthetic_state_dict(spec, 3)
    p = _predictor(spec, patch, [sd], accumulate_in='fp32')
    image = torch.randn(1, 40, 36, 44, generator=torch.Generator().manual_seed(2))
    want = osw.sliding_window_logits(lambda x: p.forward_patches(x).cpu(), image, patch, spec.num_heads, accum='fp32')
    got = p.predict_sliding_window_return_logits(image).float().cpu()
    assert (got - want).abs().max() <= 2e-3 * float(want.abs().max()) + 1e-3      # only the final fp16 store


@pytest.mark.parametrize('case', [c for c in SW_CASES if c['kind'] == 'unet'], ids=lambda c: c['name'])
def test_reference_golden_volumes(case, golden_dir):
    """Outputs of the reference's own predictor (fp32 CPU network) vs the HIP engine."""
    z = np.load(os.path.join(golden_dir, 'sliding_window.npz'))
    ref = torch.from_numpy(z[case['name']].view(np.int16)).view(torch.half).float()
    spec = toy_unet_spec(case['channels'], case['heads'])
    _, params = make_case_networks(case)
    p = _predictor(spec, case['patch'], params, mirror=case['mirror'], step=case['step'], gaussian=case['gaussian'])
    image = make_case_inputs(case)
    if case['folds'] > 1:
        got = p.predict_logits_from_preprocessed_data(image).float()
    else:
        got = p.predict_sliding_window_return_logits(image).float().cpu()
    # the reference's fp16 accumulators quantise to +-0.5 where the summed weight is at the 5.96e-8 clamp
    # (volume corners); compare on the well-conditioned interior and report the rest
    m = 3
    inner = (slice(None), slice(m, -m), slice(m, -m), slice(m, -m))
    mr, rr = _report(case['name'] + ' interior', got[inner], ref[inner])
    assert mr <= MAX_REL and rr <= RMSE_REL
    err = float((got - ref).abs().max())
    # label maps: identical wherever the top-1/top-2 margin exceeds twice the measured logit error
    top2 = ref.topk(2, 0).values
    safe = (top2[0] - top2[1]) > 2 * err
    seg_ref = torch.from_numpy(z[case['name'] + '__seg'].astype(np.int64))
    assert (got.argmax(0)[safe] == seg_ref[safe]).all()
    print(f'label agreement overall {(got.argmax(0) == seg_ref).float().mean():.5f}, '
          f'safe voxels {safe.float().mean():.3f}')


def test_argmax_labels_on_device_match_numpy():
    spec, patch = SPECS['heads61']
    sd = synthetic_state_dict(spec, 8)
    p = _predictor(spec, patch, [sd])
    image = torch.randn(1, 24, 20, 40, generator=torch.Generator().manual_seed(4))
    logits = p.predict_sliding_window_return_logits(image)
    labels = p.predict_segmentation_from_preprocessed_data(image).cpu().numpy()
    assert np.array_equal(labels, logits.cpu().numpy().argmax(0))
    assert labels.max() > 0


def test_label_rules_match_reference_golden(golden_dir):
    """f-1: LabelManager.convert_logits_to_segmentation on the device - regions (sigmoid > 0.5 in
    regions_class_order) on every fp16 bit pattern and on fp32 logits around the threshold, uint16 labels for
    >= 255 foreground labels, argmax with ties and NaNs - bit-exact against vectors made by the reference."""
    from golden_cases import DATASET_JSONS, label_rule_inputs
    z = np.load(os.path.join(golden_dir, 'label_rules.npz'))
    inp = label_rule_inputs()
    spec, patch = SPECS['toy3']                                         # 3 heads = the 3 regions
    sd = synthetic_state_dict(spec, 1)
    p = _predictor(spec, patch, [sd], dataset_json=DATASET_JSONS['regions'])
    got = p.convert_logits_to_segmentation(inp['regions_f16'].cuda())
    assert got.dtype == torch.uint8 and np.array_equal(got.cpu().numpy(), z['regions_f16'])
    got = p.convert_logits_to_segmentation(inp['regions_f32'].cuda())
    assert np.array_equal(got.cpu().numpy(), z['regions_f32'])
    p16 = _predictor(spec, patch, [sd], dataset_json=DATASET_JSONS['regions_u16'])
    got = p16.convert_logits_to_segmentation(inp['regions_f16'].cuda())
    assert np.array_equal(got.cpu().numpy(), z['regions_u16']) and int(got.max()) == 300
    spec5, patch5 = SPECS['aniso5']
    p5 = _predictor(spec5, patch5, [synthetic_state_dict(spec5, 1)])
    got = p5.convert_logits_to_segmentation(inp['argmax_f16'].cuda())
    assert np.array_equal(got.cpu().numpy(), z['argmax_f16'])


@pytest.mark.parametrize('path', ['gather', 'accumulate', 'accumulate_fp32'])
@pytest.mark.parametrize('n_folds', [1, 2])
def test_region_labels_straight_from_the_accumulators(n_folds, path):
    """Region labels (the highest head above the sigmoid threshold, mapped through regions_class_order - values above 255 make the
    map uint16) from the gather kernel and from the whole-volume accumulators (FNN_NO_GATHER: the cooperative uint8 kernel and
    the one-lane uint16 kernel, fp16 and fp32 sums)."""
    from golden_cases import DATASET_JSONS
    spec, patch = SPECS['toy3']
    sds = [synthetic_state_dict(spec, 3 + i) for i in range(n_folds)]
    for dj_name in ('regions', 'regions_u16'):
        if path != 'gather':
            os.environ['FNN_NO_GATHER'] = '1'
        try:
            p = _predictor(spec, patch, sds, dataset_json=DATASET_JSONS[dj_name], accumulate_in='fp32' if path == 'accumulate_fp32' else 'fp16')
        finally:
            os.environ.pop('FNN_NO_GATHER', None)
        image = torch.randn(1, 24, 20, 40, generator=torch.Generator().manual_seed(4))
        logits = p.predict_logits_from_preprocessed_data(image)
        want = osw.logits_to_labels(logits.cpu(), DATASET_JSONS[dj_name]['regions_class_order'])
        got = p.predict_segmentation_from_preprocessed_data(image).cpu()
        assert torch.equal(got.to(torch.int64), want.to(torch.int64))
        assert len(torch.unique(got)) >= 3


def test_results_on_cpu_when_not_everything_on_device():
    spec, patch = SPECS['toy3']
    p = _predictor(spec, patch, [synthetic_state_dict(spec, 1)], on_device=False)
    out = p.predict_sliding_window_return_logits(torch.zeros(1, 16, 16, 32))
    assert out.device.type == 'cpu' and out.dtype == torch.half


def test_argument_errors_match_reference():
    spec, patch = SPECS['toy3']
    p = _predictor(spec, patch, [synthetic_state_dict(spec, 1)])
    with pytest.raises(AssertionError):
        p.predict_sliding_window_return_logits(torch.zeros(16, 16, 32))            # ndim != 4
    with pytest.raises(AssertionError):
        p.predict_sliding_window_return_logits(np.zeros((1, 16, 16, 32), np.float32))   # not a tensor
    with pytest.raises(AssertionError):
        p.predict_sliding_window_return_logits(torch.zeros(2, 16, 16, 32))         # wrong channel count
    p.tile_step_size = 1.5
    with pytest.raises(AssertionError):
        p.predict_sliding_window_return_logits(torch.zeros(1, 16, 16, 32))


def test_inf_in_prediction_raises_like_reference():
    spec, patch = SPECS['toy3']
    sd = synthetic_state_dict(spec, 1)
    sd = {k: v.clone() for k, v in sd.items()}
    sd['decoder.seg_layers.1.bias'] += 7e4           # logits beyond the fp16 range after weighting
    p = _predictor(spec, patch, [sd])
    with pytest.raises(RuntimeError, match='Encountered inf'):
        p.predict_sliding_window_return_logits(torch.zeros(1, 16, 16, 32))


def test_idempotence_and_linearity_of_the_blend_at_full_patch_size():
    """Size-independent properties at a BASELINE-sized patch (160x96x96, 61 heads): predicting twice gives the
    same bits; with a constant-logit network the blend returns that constant wherever it is representable."""
    spec = student_spec((2.0, 0.9765625, 0.9765625), (160, 96, 96), 1, 61, reduction=2)
    sd = synthetic_state_dict(spec, 1234)
    # zero the seg-head weights -> logits == bias everywhere
    sd['decoder.seg_layers.4.weight'].zero_()
    bias = torch.linspace(-3, 3, 61)
    sd['decoder.seg_layers.4.bias'].copy_(bias)
    p = _predictor(spec, (160, 96, 96), [sd], batch=2, accumulate_in='fp32')
    image = torch.randn(1, 200, 120, 130, generator=torch.Generator().manual_seed(1))
    a = p.predict_sliding_window_return_logits(image)
    b = p.predict_sliding_window_return_logits(image)
    assert torch.equal(a, b)
    want = bias.half().float()[:, None, None, None].expand_as(a)
    assert (a.float().cpu() - want).abs().max() <= 2e-3


@pytest.mark.parametrize('acc_mode', ['fp32', 'fp16'])
@pytest.mark.parametrize('world', [2, 4])
def test_sharded_boxes_through_c_abi_match_single_gpu(world, acc_mode):
    """The multi-GPU building blocks (fnn_accumulate_patches / fnn_normalize_box) driven for `world` virtual
    ranks on one GPU, with the halo exchange done locally: must reproduce the single-engine fp32 result."""
    from fast_nnunet_amd import capi
    from fast_nnunet_amd.dist import Decomposition, _view, unpadded
    spec, patch = SPECS['toy3']
    sd = synthetic_state_dict(spec, 21)
    p = _predictor(spec, patch, [sd], accumulate_in=acc_mode)
    image = torch.randn(1, 37, 30, 70, generator=torch.Generator().manual_seed(6))
    want = p.predict_sliding_window_return_logits(image)
    x = image.cuda().float().contiguous()
    padded, pad_lo, origins = capi.plan_volume(patch, x.shape[1:], 0.5)
    steps = [sorted(set(int(v) for v in origins[:, d])) for d in range(3)]
    dec = Decomposition.build(patch, padded, steps, world)
    opts = p._opts()
    hp = p._engine.accumulator_channels
    adt = torch.float32 if acc_mode == 'fp32' else torch.half
    accs = []
    for r in range(world):
        box = dec.boxes[r]
        dims = tuple(box[1][d] - box[0][d] for d in range(3))
        acc = torch.zeros((*dims, hp), dtype=adt, device='cuda')
        # ShardedPredictor's order: the patches that feed another rank first, then the interior (two calls)
        boundary, interior = dec.split_patches(r, patch, origins)
        assert boundary and sorted(boundary + interior) == sorted(dec.patch_ids[r])
        for ids in (boundary, interior):
            if ids:
                p._engine.accumulate_patches(x.data_ptr(), x.shape, opts, ids, box[0], box[1], acc.data_ptr())
        accs.append(acc)
    torch.cuda.synchronize()
    for r in range(world):                       # local stand-in for exchange_halos
        for peer, region in dec.transfers(r)[1]:
            _view(accs[r], dec.boxes[r], region).add_(_view(accs[peer], dec.boxes[peer], region))
    got = torch.zeros_like(want)
    for r in range(world):
        own = unpadded(dec.owned[r], pad_lo, x.shape[1:])
        p._engine.normalize_box(accs[r].data_ptr(), x.shape, opts, dec.boxes[r][0], dec.boxes[r][1], own[0], own[1],
                                got.data_ptr())
    torch.cuda.synchronize()
    # partial sums are added in a different order than on one GPU: fp32 agrees up to the final fp16 store; fp16
    # accumulators round differently per visit (same error class as the reference's own fp16 accumulation, whose
    # +-0.5 quantisation at the 5.96e-8-weight volume border is excluded)
    m = 3
    inner = (slice(None), slice(m, -m), slice(m, -m), slice(m, -m))
    diff = (got.float() - want.float())[inner].abs().max()
    assert diff <= (2e-3 if acc_mode == 'fp32' else 2e-2) * float(want.float().abs().max()), diff
    assert (got.argmax(0) != want.argmax(0))[inner[1:]].float().mean() < (1e-4 if acc_mode == 'fp32' else 2e-3)


def test_sharded_predictor_single_rank_process_group():
    """ShardedPredictor end to end with a one-rank RCCL process group on the GPU: owned box, gathered logits,
    gathered labels (fnn_labels_box + all_gather) and the fold ensemble must equal the single-GPU predictor's."""
    import torch.distributed as dist
    from fast_nnunet_amd.dist import ShardedPredictor
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        spec, patch = SPECS['toy3']
        sds = [synthetic_state_dict(spec, 22 + f) for f in range(2)]
        for accum in ('fp32', 'fp16'):
            p = _predictor(spec, patch, sds[:1], accumulate_in=accum)
            image = torch.randn(1, 20, 40, 33, generator=torch.Generator().manual_seed(7))
            want = p.predict_sliding_window_return_logits(image)
            for mode in ('gather', 'accumulate'):
                sp = ShardedPredictor(p, mode=mode)
                got, own = sp.predict_sliding_window_return_logits(image)
                assert own == ((0, 0, 0), (20, 40, 33))
                assert torch.equal(got, want)                 # one rank: the reference's visiting order, bit for bit
                assert torch.equal(sp.predict_sliding_window_return_logits(image, gather=True), want)
                labels = sp.predict_segmentation_from_preprocessed_data(image)
                assert labels.dtype == torch.uint8 and torch.equal(labels, p.predict_segmentation_from_preprocessed_data(image))
                assert torch.equal(labels.long(), want.float().argmax(0))
        # test-time mirroring (the reference's default): the gather mode keeps all 2^k evaluations, 'auto' takes it
        pm = _predictor(spec, patch, sds[:1], mirror=(0, 1, 2))
        want_m = pm.predict_sliding_window_return_logits(image)
        for mode in ('auto', 'gather', 'accumulate'):
            spm = ShardedPredictor(pm, mode=mode)
            ph = spm.start_phases()
            got_m, _ = spm.predict_sliding_window_return_logits(image)
            if mode != 'accumulate':
                assert torch.equal(got_m, want_m) and ph['gather_box_ms'] > 0 and ph['n_interior_patches'] > 0
            else:                                             # partial sums in another visiting order: the reference's error class
                assert (got_m.float() - want_m.float()).abs().max() <= 2e-2 * float(want_m.float().abs().max())
            spm.phases = None
        p2 = _predictor(spec, patch, sds)
        sp2 = ShardedPredictor(p2)
        assert torch.equal(sp2.predict_logits_from_preprocessed_data(image).cpu(), p2.predict_logits_from_preprocessed_data(image))
        assert torch.equal(sp2.predict_segmentation_from_preprocessed_data(image),
                           p2.predict_segmentation_from_preprocessed_data(image))
    finally:
        dist.destroy_process_group()


def test_resenc_sliding_window_bit_identical_and_close_to_oracle():
    spec, patch = SPECS['resenc4']
    sd = {'network.' + k: v for k, v in synthetic_state_dict(spec, 31).items()}        # student checkpoints carry `network.`
    p = _predictor(spec, patch, [sd], mirror=[0, 2])
    image = torch.randn(1, 24, 40, 50, generator=torch.Generator().manual_seed(3))
    got = p.predict_sliding_window_return_logits(image).cpu()
    want = osw.sliding_window_logits(lambda x: p.forward_patches(x).cpu(), image, patch, spec.num_heads, mirror_axes=[0, 2])
    assert torch.equal(got, want)
    net = build_oracle(spec, synthetic_state_dict(spec, 31))
    ref = osw.sliding_window_logits(net, image, patch, spec.num_heads, mirror_axes=[0, 2], accum='fp32')
    inner = (slice(None), slice(3, -3), slice(3, -3), slice(3, -3))
    mr, rr = _report('resenc4 volume', got.float()[inner], ref[inner])
    assert mr <= MAX_REL and rr <= RMSE_REL


@pytest.mark.parametrize('name', ['resenc4', 'resenc_2ch_odd'])
def test_resenc_stage_closing_add_writes_the_pooled_skip_tensor_too(name):
    """Round 5: the block output that closes a stage is average-pooled by the next stage's skip path; combine_pool_kernel
    forms the block output and the pooled tensor in one pass (the pooled values from the fp16-rounded outputs, summed in
    avgpool_kernel's order) instead of avgpool_kernel re-reading the tensor just written.  Bit-identical to the two
    launches (FNN_NO_POOL_FUSE), and no avgpool_kernel behind a combine launch."""
    spec, patch = SPECS[name]
    sd = synthetic_state_dict(spec, 77)
    x = torch.randn(3, spec.in_channels, *patch, generator=torch.Generator().manual_seed(5))
    os.environ.pop('FNN_NO_POOL_FUSE', None)
    p = _predictor(spec, patch, [sd])
    p._engine.set_profiling(True)
    fused = p.forward_patches(x).cpu()
    log = p._engine.kernel_log()
    p._engine.set_profiling(False)
    # (resenc4's last stage pools with stride (2, 1, 1): not a form the fused kernel takes - that one avgpool_kernel stays)
    assert 'combine_pool_kernel' in log and log.count('avgpool_kernel') == (1 if name == 'resenc4' else 0), log
    os.environ['FNN_NO_POOL_FUSE'] = '1'
    try:
        q = _predictor(spec, patch, [sd])
    finally:
        os.environ.pop('FNN_NO_POOL_FUSE', None)
    q._engine.set_profiling(True)
    plain = q.forward_patches(x).cpu()
    assert 'avgpool_kernel' in q._engine.kernel_log() and 'combine_pool_kernel' not in q._engine.kernel_log()
    q._engine.set_profiling(False)
    assert torch.equal(fused, plain)


def test_initialize_from_trained_model_folder_distilled_student(tmp_path):
    """The reference's model-folder layout (plans.json, dataset.json, fold_k/checkpoint_*.pth with the
    nnUNetTrainer.save_checkpoint schema, nnUNetTrainer.py:1159-1169) for a DISTILLED student - which the
    reference's own predictor cannot rebuild: folds auto-detected, weights wrapped the way real
    checkpoints are, ensemble == oracle."""
    import json
    from fast_nnunet_amd import nnUNetPredictor
    r = 2
    teacher_feats = [32, 64, 64]
    kernels = [[3, 3, 3]] * 3
    strides = [[1, 1, 1], [2, 2, 2], [1, 2, 2]]
    spec = UNetSpec('plain', 1, 3, [max(f // r, 8) for f in teacher_feats], [tuple(k) for k in kernels],
                    [tuple(s) for s in strides], [2, 2, 2], [2, 2])
    patch = (16, 16, 32)
    plans = {'dataset_name': 'Dataset123_Toy', 'plans_name': 'nnUNetPlans', 'transpose_forward': [0, 1, 2],
             'transpose_backward': [0, 1, 2], 'label_manager': 'LabelManager',
             'configurations': {
                 '3d_fullres': {'patch_size': list(patch), 'spacing': [1.0, 1.0, 1.0], 'batch_size': 2,
                                'architecture': {'network_class_name': 'dynamic_network_architectures.architectures.unet.PlainConvUNet',
                                                 'arch_kwargs': {'n_stages': 3, 'features_per_stage': teacher_feats,
                                                                 'kernel_sizes': kernels, 'strides': strides,
                                                                 'n_conv_per_stage': [2, 2, 2], 'n_conv_per_stage_decoder': [2, 2],
                                                                 'conv_bias': True, 'norm_op_kwargs': {'eps': 1e-5, 'affine': True}},
                                                 '_kw_requires_import': []}},
                 '3d_fullres_student': {'inherits_from': '3d_fullres'}}}
    dataset_json = {'labels': {'background': 0, 'liver': 1, 'tumour': 2}, 'channel_names': {'0': 'CT'}, 'file_ending': '.nii.gz'}
    folder = tmp_path / 'nnUNetDistillationTrainer__nnUNetPlans__3d_fullres_student'
    folder.mkdir()
    (folder / 'plans.json').write_text(json.dumps(plans))
    (folder / 'dataset.json').write_text(json.dumps(dataset_json))
    sds = []
    for fold in (0, 1):
        sd = synthetic_state_dict(spec, 70 + fold)
        sds.append(sd)
        wrapped = {}
        for k, v in sd.items():                               # aliases real checkpoints carry
            wrapped[k] = v
            if '.conv.' in k:
                wrapped[k.replace('.conv.', '.all_modules.0.')] = v
        (folder / f'fold_{fold}').mkdir()
        torch.save({'network_weights': wrapped, 'optimizer_state': None, 'current_epoch': 1000,
                    'init_args': {'configuration': '3d_fullres_student', 'feature_reduction_factor': r, 'fold': fold},
                    'trainer_name': 'nnUNetDistillationTrainer', 'inference_allowed_mirroring_axes': (0, 1, 2)},
                   folder / f'fold_{fold}' / 'checkpoint_best.pth')
    (folder / 'fold_all').mkdir()                             # must be ignored by auto-detection
    p = nnUNetPredictor(tile_step_size=0.5, use_gaussian=True, use_mirroring=False, device=torch.device('cuda', 0),
                        allow_tqdm=False, patches_per_forward=2)
    p.initialize_from_trained_model_folder(str(folder), use_folds=None, checkpoint_name='checkpoint_best.pth')
    assert p.trainer_name == 'nnUNetDistillationTrainer' and p.allowed_mirroring_axes == (0, 1, 2)
    assert p.label_manager.num_segmentation_heads == 3 and len(p.list_of_parameters) == 2
    assert p.configuration_manager.patch_size == list(patch)
    image = torch.randn(1, 20, 24, 40, generator=torch.Generator().manual_seed(11))
    got = p.predict_logits_from_preprocessed_data(image).float()
    nets = [build_oracle(spec, sd) for sd in sds]
    ref = osw.ensemble_logits(nets, image, patch, 3, accum='fp32')
    inner = (slice(None), slice(3, -3), slice(3, -3), slice(3, -3))
    mr, rr = _report('model folder', got[inner], ref[inner])
    assert mr <= MAX_REL and rr <= RMSE_REL
    # a wrong reduction factor in init_args is caught against the plans
    ck = torch.load(folder / 'fold_0' / 'checkpoint_best.pth', weights_only=False)
    ck['init_args']['feature_reduction_factor'] = 4
    torch.save(ck, folder / 'fold_0' / 'checkpoint_best.pth')
    with pytest.raises(RuntimeError, match='do not match'):
        p.initialize_from_trained_model_folder(str(folder), use_folds=(0,), checkpoint_name='checkpoint_best.pth')


# ---------------------------------------------------------------------------------------------------------------
# `2d` configurations: Conv2d network, patch_size with two entries, every slice of the first axis is tiled
# (predict_from_raw_data.py:508-524).  The engine runs them as depth-1 3-D patches.
# ---------------------------------------------------------------------------------------------------------------
def test_2d_network_forward_matches_fp32_oracle():
    spec = UNetSpec('plain', 2, 5, [16, 32, 64, 64], [(3, 3)] * 4, [(1, 1), (2, 2), (2, 2), (1, 2)], [2, 2, 2, 2], [2, 2, 2])
    patch = (48, 64)
    sd = synthetic_state_dict(spec, 31)
    p = _predictor(spec, patch, [sd], batch=5)
    assert p._spec.spatial_dims == 2 and tuple(p._spec.patch) == (1, 48, 64)
    x = torch.randn(5, 2, *patch, generator=torch.Generator().manual_seed(0))
    got = p.forward_patches(x).cpu()
    with torch.inference_mode():
        ref = build_oracle(spec, sd)(x)
    mr, rr = _report('2d forward', got, ref)
    assert mr <= MAX_REL and rr <= RMSE_REL


@pytest.mark.parametrize('shape,mirror,step,folds', [((5, 36, 44), None, 0.5, 1), ((3, 30, 21), [0, 1], 0.5, 1),
                                                     ((4, 11, 30), [1], 0.3, 2), ((1, 16, 32), None, 1.0, 1)])
def test_2d_driver_bit_identical_to_oracle_driver_on_engine_logits(shape, mirror, step, folds):
    spec, patch = toy_unet_spec_2d(1, 3), (16, 32)
    sds = [synthetic_state_dict(spec, 90 + f) for f in range(folds)]
    p = _predictor(spec, patch, sds, mirror=mirror, step=step)
    image = torch.randn(1, *shape, generator=torch.Generator().manual_seed(9))

    def engine_net(fold):
        def f(x):
            p._active_fold = fold
            return p.forward_patches(x).cpu()
        return f

    nets = [engine_net(f) for f in range(folds)]
    kw = dict(step=step, mirror_axes=mirror, accum='fp16')
    if folds > 1:
        want = osw.ensemble_logits(nets, image, patch, spec.num_heads, **kw)
        got = p.predict_logits_from_preprocessed_data(image)
    else:
        want = osw.sliding_window_logits(nets[0], image, patch, spec.num_heads, **kw)
        p._active_fold = 0
        got = p.predict_sliding_window_return_logits(image)
    assert got.dtype == torch.half and tuple(got.shape) == tuple(want.shape)
    assert (_bits(got) == _bits(want)).all()
    # the slicer list of the mirror class is the reference's 2-D branch
    padded = [s + a + b for s, (a, b) in zip(image.shape[1:], osw.pad_to_patch(image.shape[1:], patch)[0])]
    assert p._internal_get_sliding_window_slicers(tuple(padded)) == osw.patch_slicers(padded, patch, step)


@pytest.mark.parametrize('case', [c for c in SW_CASES_2D if c['kind'] == 'unet'], ids=lambda c: c['name'])
def test_2d_reference_golden_volumes(case, golden_dir):
    """Outputs of the reference's own predictor on a `2d` configuration (fp32 CPU Conv2d network) vs the HIP engine."""
    z = np.load(os.path.join(golden_dir, 'sliding_window_2d.npz'))
    ref = torch.from_numpy(z[case['name']].view(np.int16)).view(torch.half).float()
    spec = toy_unet_spec_2d(case['channels'], case['heads'])
    _, params = make_case_networks(case)
    p = _predictor(spec, case['patch'], params, mirror=case['mirror'], step=case['step'], gaussian=case['gaussian'])
    got = p.predict_sliding_window_return_logits(make_case_inputs(case)).float().cpu()
    m = 3
    inner = (slice(None), slice(None), slice(m, -m), slice(m, -m))
    mr, rr = _report(case['name'] + ' interior', got[inner], ref[inner])
    assert mr <= MAX_REL and rr <= RMSE_REL


# ---------------------------------------------------------------------------------------------------------------
# raw image -> label map on the raw grid, everything on the device (predict_single_npy_array,
# predict_from_raw_data.py:423-468): f-2 -> hot path -> f-3 against the oracle chain
# ---------------------------------------------------------------------------------------------------------------
def _raw_case(spacing_cfg, transpose=(0, 1, 2), heads=3):
    from fast_nnunet_amd import nnUNetPredictor
    from fast_nnunet_amd.plans import PlansManager
    spec, patch = toy_unet_spec(1, heads), (16, 16, 32)
    ip = {'0': {'mean': 100.0, 'std': 250.0, 'percentile_00_5': -400.0, 'percentile_99_5': 800.0}}
    tb = [int(i) for i in np.argsort(transpose)]
    pm = PlansManager({'dataset_name': 'Dataset999_Golden', 'plans_name': 'nnUNetPlans', 'transpose_forward': list(transpose),
                       'transpose_backward': tb, 'foreground_intensity_properties_per_channel': ip,
                       'configurations': {'3d_fullres': {
                           'patch_size': list(patch), 'spacing': list(spacing_cfg), 'normalization_schemes': ['CTNormalization'],
                           'use_mask_for_norm': [False],
                           'architecture': {'network_class_name': 'PlainConvUNet', 'arch_kwargs': {}, '_kw_requires_import': []}}}})
    dj = {'labels': {('background' if i == 0 else f'c{i}'): i for i in range(heads)}, 'channel_names': {'0': 'CT'},
          'file_ending': '.nii.gz'}
    sd = synthetic_state_dict(spec, 17)
    p = nnUNetPredictor(tile_step_size=0.5, use_gaussian=True, use_mirroring=False, perform_everything_on_device=True,
                        device=torch.device('cuda', 0), verbose=False, allow_tqdm=False, patches_per_forward=3)
    p.manual_initialization(None, pm, pm.get_configuration('3d_fullres'), [sd], dj, 'nnUNetTrainer', None)
    return p, spec, patch, ip, tb


@pytest.mark.parametrize('spacing_raw,spacing_cfg,transpose', [
    ((1.0, 1.0, 1.0), (1.0, 1.0, 1.0), (0, 1, 2)),             # no resampling: labels straight from the accumulators
    ((1.0, 1.0, 1.0), (1.0, 1.0, 1.0), (2, 0, 1)),
    ((1.5, 0.8, 0.8), (1.0, 1.0, 1.0), (0, 1, 2)),             # isotropic-ish: order-3 resize in, order-1 back
    ((0.8, 4.0, 0.8), (1.0, 2.0, 1.0), (1, 0, 2)),             # anisotropic after the transpose: per-slice path
])
def test_predict_single_npy_array_matches_the_oracle_chain(spacing_raw, spacing_cfg, transpose):
    from oracle import preprocess as opre
    from oracle import resample as ores
    p, spec, patch, ip, tb = _raw_case(spacing_cfg, transpose)
    rng = np.random.default_rng(11)
    raw = (rng.standard_normal((1, 34, 40, 52)) * 300 + 150).astype(np.float32)
    raw[:, :3] = 0; raw[:, :, -5:] = 0; raw[:, :, :, :2] = 0
    got = p.predict_single_npy_array(raw, {'spacing': list(spacing_raw)})
    assert got.dtype == np.uint8 and got.shape == raw.shape[1:]

    data, bbox, before = opre.preprocess_case(raw, transpose, ['CTNormalization'], ip)
    sp_t = [spacing_raw[i] for i in transpose]
    new_shape = ores.compute_new_shape(data.shape[1:], sp_t, spacing_cfg)
    do_sep, axis = ores.determine_do_sep_z_and_axis(None, sp_t, spacing_cfg)
    net_in = ores.resample_data(data, new_shape, axis=axis, order=3, do_separate_z=do_sep)
    logits = osw.sliding_window_logits(lambda t: p.forward_patches(t).cpu(), torch.from_numpy(net_in), patch,
                                       spec.num_heads, accum='fp16')
    do_sep_b, axis_b = ores.determine_do_sep_z_and_axis(None, spacing_cfg, sp_t)
    back = ores.resample_data(logits.numpy(), data.shape[1:], axis=axis_b, order=1, do_separate_z=do_sep_b)
    lab = osw.logits_to_labels(torch.from_numpy(back.astype(np.float32))).numpy().astype(np.uint8)
    want = opre.revert_labels(lab, bbox, before, tb, spec.num_heads - 1)
    mismatch = (got != want).mean()
    print(f'label mismatch {mismatch:.5f}')
    # identical when nothing is resampled; otherwise the network input differs in the last fp32 bit and a few
    # near-tie voxels may flip
    assert mismatch == 0.0 if list(new_shape) == list(data.shape[1:]) else mismatch < 5e-3
    assert len(np.unique(got)) >= 2

    # save_or_return_probabilities=True: (labels, fp32 probabilities) on the raw grid, like the reference's export
    seg2, probs = p.predict_single_npy_array(raw, {'spacing': list(spacing_raw)}, save_or_return_probabilities=True)
    want_seg2, want_probs = opre.export_with_probabilities(back, bbox, before, tb, spec.num_heads - 1)
    assert probs.dtype == np.float32 and probs.shape == (spec.num_heads, *raw.shape[1:])
    assert np.abs(probs.sum(0) - 1).max() < 1e-5
    assert (seg2 != got).mean() < 1e-4                         # argmax of probabilities vs argmax of logits: ties only
    same = list(new_shape) == list(data.shape[1:])
    assert np.abs(probs - want_probs).mean() < (1e-6 if same else 2e-3)
    assert (seg2 != want_seg2).mean() < (1e-4 if same else 5e-3)


# ---------------------------------------------------------------------------------------------------------------
# engine .ini front-end: set_config -> set_workspace -> infer, the call order of the
# reference's engine/fast_nnunet.cpp:16-27
# ---------------------------------------------------------------------------------------------------------------
def _toy_model_folder(tmp_path, patch, num_heads, plans_spacing, seed=5):
    import json
    teacher_feats = [32, 64, 64]
    kernels = [[3, 3, 3]] * 3
    strides = [[1, 1, 1], [2, 2, 2], [1, 2, 2]]
    spec = UNetSpec('plain', 1, num_heads, [16, 32, 32], [tuple(k) for k in kernels], [tuple(s) for s in strides],
                    [2, 2, 2], [2, 2])
    plans = {'dataset_name': 'Dataset124_Toy', 'plans_name': 'nnUNetPlans', 'transpose_forward': [0, 1, 2],
             'transpose_backward': [0, 1, 2], 'label_manager': 'LabelManager',
             'foreground_intensity_properties_per_channel': {'0': {'mean': 0.0, 'std': 1.0, 'percentile_00_5': -1.0,
                                                                   'percentile_99_5': 1.0}},
             'configurations': {'3d_fullres': {
                 'patch_size': list(patch), 'spacing': list(plans_spacing), 'batch_size': 2,
                 'normalization_schemes': ['ZScoreNormalization'], 'use_mask_for_norm': [False],
                 'resampling_fn_data': 'resample_data_or_seg_to_shape',
                 'resampling_fn_data_kwargs': {'is_seg': False, 'order': 3, 'order_z': 0, 'force_separate_z': None},
                 'resampling_fn_probabilities': 'resample_data_or_seg_to_shape',
                 'resampling_fn_probabilities_kwargs': {'is_seg': False, 'order': 1, 'order_z': 0, 'force_separate_z': None},
                 'architecture': {'network_class_name': 'dynamic_network_architectures.architectures.unet.PlainConvUNet',
                                  'arch_kwargs': {'n_stages': 3, 'features_per_stage': teacher_feats, 'kernel_sizes': kernels,
                                                  'strides': strides, 'n_conv_per_stage': [2, 2, 2],
                                                  'n_conv_per_stage_decoder': [2, 2], 'conv_bias': True,
                                                  'norm_op_kwargs': {'eps': 1e-5, 'affine': True}},
                                  '_kw_requires_import': []}}}}
    dataset_json = {'labels': {('background' if i == 0 else f'c{i}'): i for i in range(num_heads)},
                    'channel_names': {'0': 'CT'}, 'file_ending': '.nii.gz'}
    folder = tmp_path / 'nnUNetDistillationTrainer__nnUNetPlans__3d_fullres'
    folder.mkdir()
    (folder / 'plans.json').write_text(json.dumps(plans))
    (folder / 'dataset.json').write_text(json.dumps(dataset_json))
    sd = synthetic_state_dict(spec, seed)
    (folder / 'fold_0').mkdir()
    torch.save({'network_weights': sd, 'init_args': {'configuration': '3d_fullres', 'feature_reduction_factor': 2, 'fold': 0},
                'trainer_name': 'nnUNetDistillationTrainer', 'inference_allowed_mirroring_axes': (0, 1, 2)},
               folder / 'fold_0' / 'checkpoint_final.pth')
    return folder, plans, dataset_json, sd, spec


_INI = """[model]
file_name = toy.trt
input_name = input
output_name = output
num_class = {num_class}

[input]
depth = {p[0]}
height = {p[1]}
width = {p[2]}
patch_size = {p[0]}, {p[1]}, {p[2]}
target_spacing = {s[0]}, {s[1]}, {s[2]}

[preprocessing]
mean = 150.0
std_dev = 300.0
lower_bound = -400.0
upper_bound = 700.0

[inference]
use_mirroring = {mirror}
step_size = 0.5
use_gaussian = true
"""


@pytest.mark.parametrize('spacing_raw,target,mirror', [((1.0, 1.0, 1.0), (1.0, 1.0, 1.0), False),
                                                       ((1.4, 0.8, 0.8), (1.0, 1.0, 1.0), True)])
def test_engine_ini_front_end_matches_the_predictor_it_configures(tmp_path, spacing_raw, target, mirror):
    from fast_nnunet_amd import nnUNetPredictor
    from fast_nnunet_amd.engine_config import Engine, load_engine_config, plans_with_engine_config
    from fast_nnunet_amd.plans import PlansManager
    patch = (16, 16, 32)
    folder, plans, dj, sd, spec = _toy_model_folder(tmp_path, patch, 4, plans_spacing=(3.0, 3.0, 3.0))
    ini = tmp_path / 'toy.ini'
    ini.write_text(_INI.format(num_class=4, p=patch, s=target, mirror=str(mirror).lower()))
    eng = Engine(device=torch.device('cuda', 0), patches_per_forward=3)
    eng.set_config(str(ini))
    eng.set_workspace(str(folder))
    rng = np.random.default_rng(3)
    raw = (rng.standard_normal((30, 36, 44)) * 300 + 150).astype(np.float32)
    raw[:2] = 0; raw[:, -4:] = 0
    got = eng.infer(raw, spacing_raw)
    assert got.dtype == np.uint8 and got.shape == raw.shape and len(np.unique(got)) >= 2

    # the same model driven through the predictor API with plans that state what the ini states
    pm = PlansManager(plans_with_engine_config(plans, '3d_fullres', load_engine_config(str(ini))))
    p = nnUNetPredictor(tile_step_size=0.5, use_gaussian=True, use_mirroring=mirror, device=torch.device('cuda', 0),
                        allow_tqdm=False, patches_per_forward=3)
    p.manual_initialization(None, pm, pm.get_configuration('3d_fullres'), [sd], dj, 'nnUNetDistillationTrainer',
                            (0, 1, 2) if mirror else None)
    want = p.predict_single_npy_array(raw[None], {'spacing': list(spacing_raw)})
    assert np.array_equal(got, want)
    # and not what the untouched plans (3 mm spacing, z-score) would have produced
    p0 = nnUNetPredictor(tile_step_size=0.5, use_gaussian=True, use_mirroring=mirror, device=torch.device('cuda', 0),
                         allow_tqdm=False, patches_per_forward=3)
    p0.initialize_from_trained_model_folder(str(folder), use_folds=(0,))
    other = p0.predict_single_npy_array(raw[None], {'spacing': list(spacing_raw)})
    assert other.shape == got.shape and (other != got).mean() > 0.01


def test_engine_ini_front_end_rejects_a_model_that_disagrees_with_the_ini(tmp_path):
    from fast_nnunet_amd.engine_config import Engine
    patch = (16, 16, 32)
    folder, *_ = _toy_model_folder(tmp_path, patch, 4, plans_spacing=(1.0, 1.0, 1.0))
    for kw, err in ((dict(num_class=5, p=patch), 'segmentation heads'), (dict(num_class=4, p=(16, 16, 16)), 'patch size')):
        ini = tmp_path / 'bad.ini'
        ini.write_text(_INI.format(s=(1.0, 1.0, 1.0), mirror='false', **kw))
        eng = Engine(device=torch.device('cuda', 0))
        eng.set_config(str(ini))
        with pytest.raises(RuntimeError, match=err):
            eng.set_workspace(str(folder))


@pytest.mark.parametrize('accum', ['fp16', 'fp32'])
@pytest.mark.parametrize('shape,heads', [((40, 36, 70), 3), ((21, 27, 50), 61), ((16, 16, 32), 3), ((11, 30, 9), 3)])
def test_gather_path_is_bit_identical_to_the_accumulate_path(shape, heads, accum):
    """gather.hip keeps every patch's last activation and forms each voxel's weighted sum in registers, in visiting
    order, instead of read-modify-writing whole-volume accumulators (FNN_NO_GATHER=1): logits (fp16 and fp32
    accumulation), the inf flag and the labels must agree bit for bit - ragged z runs, volumes smaller than the patch
    (padding), one-patch volumes, 3 and 61 heads (1 and 4 head blocks), 2 folds (the add mode)."""
    spec = UNetSpec('plain', 1, heads, [16, 32], [(3, 3, 3)] * 2, [(1, 1, 1), (2, 2, 2)], [2, 2], [2])
    patch = (16, 16, 32)
    sds = [synthetic_state_dict(spec, 300 + f) for f in range(2)]
    os.environ.pop('FNN_NO_GATHER', None)
    g = _predictor(spec, patch, sds, accumulate_in=accum)
    os.environ['FNN_NO_GATHER'] = '1'
    try:
        a = _predictor(spec, patch, sds, accumulate_in=accum)
    finally:
        os.environ.pop('FNN_NO_GATHER', None)
    image = torch.randn(1, *shape, generator=torch.Generator().manual_seed(23))
    for fold in (0, 1):
        g._active_fold = a._active_fold = fold
        assert torch.equal(g.predict_sliding_window_return_logits(image), a.predict_sliding_window_return_logits(image))
    assert torch.equal(g.predict_logits_from_preprocessed_data(image), a.predict_logits_from_preprocessed_data(image))
    g1, a1 = _predictor(spec, patch, sds[:1], accumulate_in=accum), None
    os.environ['FNN_NO_GATHER'] = '1'
    try:
        a1 = _predictor(spec, patch, sds[:1], accumulate_in=accum)
    finally:
        os.environ.pop('FNN_NO_GATHER', None)
    assert torch.equal(g1.predict_segmentation_from_preprocessed_data(image), a1.predict_segmentation_from_preprocessed_data(image))


@pytest.mark.parametrize('kind,shape', [('3d_x', (530, 16, 40)), ('3d_z', (16, 20, 1070)), ('2d', (70, 20, 40))])
def test_gather_path_with_more_than_64_tile_positions_on_an_axis(kind, shape):
    """the gather kernel holds 64 tile starts of an axis, one per lane; an axis with more positions - every
    slice of a 2-D configuration, a small patch in a long volume - used to drop to the accumulate path (and the autocast
    arithmetic to FNN_E_UNSUPPORTED).  Now the 64 starts are a WINDOW from the first tile that reaches the wave's
    coordinate (GatherParams::base_x, csrc/gather.hip gather_tile_windows): 66 positions along x, 65 along z (the axis
    of the 64-voxel runs), 70 slices.  The gather kernel must run and agree bit for bit with the accumulate path
    (logits, labels), and in the autocast arithmetic with the oracle driver."""
    if kind == '2d':
        spec, patch = toy_unet_spec_2d(1, 3), (16, 32)
    else:
        spec, patch = UNetSpec('plain', 1, 3, [16, 32], [(3, 3, 3)] * 2, [(1, 1, 1), (2, 2, 2)], [2, 2], [2]), (16, 16, 32)
    sds = [synthetic_state_dict(spec, 610)]
    image = torch.randn(1, *shape, generator=torch.Generator().manual_seed(61))
    os.environ.pop('FNN_NO_GATHER', None)
    g = _predictor(spec, patch, sds, batch=8)
    g._engine.set_profiling(True)                             # (the kernel log is kept while profiling)
    got = g.predict_sliding_window_return_logits(image)
    log = g._engine.kernel_log()
    g._engine.set_profiling(False)
    assert any(k.startswith('gather_head_kernel') for k in log), log
    os.environ['FNN_NO_GATHER'] = '1'
    try:
        a = _predictor(spec, patch, sds, batch=8)
    finally:
        os.environ.pop('FNN_NO_GATHER', None)
    assert torch.equal(got, a.predict_sliding_window_return_logits(image))
    assert torch.equal(g.predict_segmentation_from_preprocessed_data(image), a.predict_segmentation_from_preprocessed_data(image))
    pa = _predictor(spec, patch, sds, batch=8, accumulate_in='fp16_autocast')
    want = osw.sliding_window_logits(lambda x: pa.forward_patches(x).cpu().half(), image, patch, spec.num_heads, step=0.5,
                                     use_gaussian=True, mirror_axes=None, accum='fp16')
    assert np.array_equal(_bits(pa.predict_sliding_window_return_logits(image)), _bits(want))


@pytest.mark.parametrize('accum', ['fp16', 'fp32'])
@pytest.mark.parametrize('heads,mirror', [(3, (0,)), (20, None), (20, (2,)), (61, (1,))])
def test_gather_kernel_variants_against_the_accumulate_path(heads, mirror, accum):
    """gather_head_kernel<HB, ACCM, LABELS, TTA> is 36 kernels; a trace of the whole GPU suite (round 3) showed 19 of them
    never launched - two head blocks (17-31 classes) beyond the plain case, fp32 sums with mirroring, labels with
    mirroring.  Logits, the fold ensemble's add mode and the labels of the gather path against the accumulate path
    (FNN_NO_GATHER: per-patch buffers, `seg_head_acc*` / `patch_acc_kernel` / `labels_from_acc*` - variants of their own
    that the same trace had not seen either) for 1 / 2 / 4 head blocks, both buffer arithmetics, with mirroring."""
    spec = UNetSpec('plain', 1, heads, [16, 32], [(3, 3, 3)] * 2, [(1, 1, 1), (2, 2, 2)], [2, 2], [2])
    patch = (16, 16, 32)
    sds = [synthetic_state_dict(spec, 520 + f) for f in range(2)]
    image = torch.randn(1, 21, 27, 50, generator=torch.Generator().manual_seed(37))
    os.environ.pop('FNN_NO_GATHER', None)
    g, g1 = _predictor(spec, patch, sds, mirror=mirror, accumulate_in=accum), _predictor(spec, patch, sds[:1], mirror=mirror, accumulate_in=accum)
    os.environ['FNN_NO_GATHER'] = '1'
    try:
        a, a1 = _predictor(spec, patch, sds, mirror=mirror, accumulate_in=accum), _predictor(spec, patch, sds[:1], mirror=mirror, accumulate_in=accum)
    finally:
        os.environ.pop('FNN_NO_GATHER', None)
    assert torch.equal(g1.predict_sliding_window_return_logits(image), a1.predict_sliding_window_return_logits(image))
    assert torch.equal(g.predict_logits_from_preprocessed_data(image), a.predict_logits_from_preprocessed_data(image))
    assert torch.equal(g1.predict_segmentation_from_preprocessed_data(image), a1.predict_segmentation_from_preprocessed_data(image))


@pytest.mark.parametrize('heads,mirror', [(20, None), (20, (0,)), (61, (2,))])
def test_gather_kernel_variants_in_the_autocast_arithmetic(heads, mirror):
    """The same variants for FNN_ACC_FP16_AUTOCAST (packed fp16 sums; served by the gather path only): logits and labels
    against the oracle driver on the engine's own fp16 network output."""
    spec = UNetSpec('plain', 1, heads, [16, 32], [(3, 3, 3)] * 2, [(1, 1, 1), (2, 2, 2)], [2, 2], [2])
    patch = (16, 16, 32)
    p = _predictor(spec, patch, [synthetic_state_dict(spec, 530)], mirror=mirror, accumulate_in='fp16_autocast')
    image = torch.randn(1, 21, 27, 50, generator=torch.Generator().manual_seed(41))
    want = osw.sliding_window_logits(lambda x: p.forward_patches(x).cpu().half(), image, patch, heads, step=0.5, use_gaussian=True,
                                     mirror_axes=mirror, accum='fp16')
    assert np.array_equal(_bits(p.predict_sliding_window_return_logits(image)), _bits(want))
    assert torch.equal(p.predict_segmentation_from_preprocessed_data(image).long().cpu(), osw.logits_to_labels(want).long())


@pytest.mark.parametrize('accum', ['fp16', 'fp16_autocast'])
@pytest.mark.parametrize('heads,scale', [(61, 1.0), (61, 1e-4), (61, 300.0), (20, 1.0), (16, 1.0), (3, 30.0)])
def test_gather_labels_without_the_quotients_on_near_ties(heads, scale, accum):
    """Round 5: the label form of the gather kernel takes ONE quotient per voxel (the largest sum's), derives the smallest
    sum whose logit rounds to the same fp16 value and compares every head's sum with it, instead of dividing all 64 sums
    and running the argmax chain (gather.hip, "argmax labels without the 64 quotients").  A seg head built to tie: pairs of
    identical classes (the first must win), classes whose biases differ by a few fp16 ulps of the logit (the division by a
    weight sum > 1 merges them: the lower index must win exactly when the quotients are equal), a class that is the maximum
    almost everywhere; logits scaled down to the fp16 subnormals and up towards the range's end (both leave the fast route).
    Labels must equal (a) the chain's (FNN_GATHER_IEEE forces it), (b) the oracle's labels of the engine's own logits."""
    spec = UNetSpec('plain', 1, heads, [16, 32], [(3, 3, 3)] * 2, [(1, 1, 1), (2, 2, 2)], [2, 2], [2])
    patch = (16, 16, 32)
    sd = synthetic_state_dict(spec, 710 + heads)
    w, b = sd['decoder.seg_layers.0.weight'].clone(), sd['decoder.seg_layers.0.bias'].clone()
    for h in range(1, heads, 4):                              # identical to the class in front of it
        w[h], b[h] = w[h - 1], b[h - 1]
    for h in range(2, heads, 4):                              # the same class with a bias a hair larger / smaller
        w[h] = w[h - 2]
        b[h] = b[h - 2] + (1e-3 if h % 8 == 2 else -2e-4) * (1 + h % 3)
    sd['decoder.seg_layers.0.weight'], sd['decoder.seg_layers.0.bias'] = w * scale, b * scale
    image = torch.randn(1, 40, 36, 70, generator=torch.Generator().manual_seed(71))
    os.environ.pop('FNN_GATHER_IEEE', None)
    p = _predictor(spec, patch, [sd], accumulate_in=accum)
    fast = p.predict_segmentation_from_preprocessed_data(image)
    os.environ['FNN_GATHER_IEEE'] = '1'
    try:
        chain = p.predict_segmentation_from_preprocessed_data(image)
    finally:
        os.environ.pop('FNN_GATHER_IEEE', None)
    assert torch.equal(fast, chain)
    logits = p.predict_sliding_window_return_logits(image)
    assert torch.equal(fast.long().cpu(), osw.logits_to_labels(logits.cpu()).long())
    ties = (logits.float().topk(2, dim=0).values.diff(dim=0) == 0).float().mean().item()
    print(f'[labels fast path] heads {heads} scale {scale}: {ties:.3f} of the voxels have equal top-2 logits')


@pytest.mark.parametrize('mirror', [None, (0, 2)])
@pytest.mark.parametrize('accum', ['fp16', 'fp16_autocast'])
@pytest.mark.parametrize('heads', [3, 20, 61])
def test_gather_k16_head_is_the_k32_head_bit_for_bit(heads, accum, mirror):
    """Round 4: a last layer of 16 channels runs the gather kernel's head as v_mfma_f32_16x16x16_f16 (a lane holds 4
    channels of its voxel) instead of a K = 32 operand whose upper half is zero.  tools/hw_probe.cpp found the two MFMA
    forms bit-identical on 16.8 M random values; here the whole driver: logits and labels with the K = 32 kernels forced
    (FNN_GATHER_K32, which also keeps those variants - what 32-channel networks run - under test on small volumes).
    Round 5: with test-time mirroring too (the mirrored evaluations used to take the K = 32 form whatever the channels)."""
    spec = UNetSpec('plain', 1, heads, [16, 32], [(3, 3, 3)] * 2, [(1, 1, 1), (2, 2, 2)], [2, 2], [2])
    patch = (16, 16, 32)
    p = _predictor(spec, patch, [synthetic_state_dict(spec, 540)], accumulate_in=accum, mirror=mirror)
    image = torch.randn(1, 37, 30, 70, generator=torch.Generator().manual_seed(43))
    os.environ.pop('FNN_GATHER_K32', None)
    logits, labels = p.predict_sliding_window_return_logits(image), p.predict_segmentation_from_preprocessed_data(image)
    p._engine.set_profiling(True)
    p.predict_sliding_window_return_logits(image)
    assert any(k.startswith('gather_head_kernel') and k.endswith(',1>') for k in p._engine.kernel_log())
    os.environ['FNN_GATHER_K32'] = '1'
    try:
        p.predict_sliding_window_return_logits(image)
        assert any(k.startswith('gather_head_kernel') and k.endswith(',0>') for k in p._engine.kernel_log())
        p._engine.set_profiling(False)
        assert np.array_equal(_bits(p.predict_sliding_window_return_logits(image)), _bits(logits))
        assert torch.equal(p.predict_segmentation_from_preprocessed_data(image), labels)
    finally:
        os.environ.pop('FNN_GATHER_K32', None)
        p._engine.set_profiling(False)


@pytest.mark.parametrize('mirror', [None, (0, 1, 2), (1,)])
def test_gather_ring_and_mirroring_are_bit_identical_to_the_accumulate_path(mirror):
    """The gather path with test-time mirroring (the 2^k evaluations' logits summed per visit, predict_from_raw_data.py:
    541-557) and with a RING of x layers instead of every patch of the volume (FNN_GATHER_RING: what a volume whose
    patch activations exceed the memory budget gets) against the accumulate path with its per-patch fp32 buffers."""
    spec, patch = SPECS['toy3']
    sd = synthetic_state_dict(spec, 77)
    image = torch.randn(1, 50, 37, 70, generator=torch.Generator().manual_seed(29))       # 6 x 4 x 4 tile positions
    os.environ['FNN_NO_GATHER'] = '1'
    try:
        want = _predictor(spec, patch, [sd], mirror=mirror).predict_sliding_window_return_logits(image)
    finally:
        os.environ.pop('FNN_NO_GATHER', None)
    p = _predictor(spec, patch, [sd], mirror=mirror)
    assert torch.equal(p.predict_sliding_window_return_logits(image), want)
    for ring in (2, 3):
        os.environ['FNN_GATHER_RING'] = str(ring)
        try:
            assert torch.equal(p.predict_sliding_window_return_logits(image), want)
            assert torch.equal(p.predict_segmentation_from_preprocessed_data(image).long(), want.float().argmax(0))
        finally:
            os.environ.pop('FNN_GATHER_RING', None)


@pytest.mark.parametrize('heads,mirror', [(3, None), (61, None), (3, (0, 2)), (130, None), (64, (0, 2))])
def test_autocast_accumulation_whole_ring_labels_and_refusals(heads, mirror):
    """FNN_ACC_FP16_AUTOCAST beyond the small driver cases: 61 heads (four head blocks), the ring of x layers, labels
    written by the gather kernel, the fold ensemble; the accumulator entry points and a forced accumulate path refuse
    the mode instead of silently computing the other arithmetic."""
    spec = UNetSpec('plain', 1, heads, [16, 32], [(3, 3, 3)] * 2, [(1, 1, 1), (2, 2, 2)], [2, 2], [2])
    patch = (16, 16, 32)
    sds = [synthetic_state_dict(spec, 410 + f) for f in range(2)]
    p = _predictor(spec, patch, sds, mirror=mirror, accumulate_in='fp16_autocast')
    image = torch.randn(1, 50, 37, 70, generator=torch.Generator().manual_seed(31))

    def net(fold):
        def f(x):
            p._active_fold = fold
            return p.forward_patches(x).cpu().half()
        return f

    kw = dict(step=0.5, use_gaussian=True, mirror_axes=mirror, accum='fp16')
    want = osw.sliding_window_logits(net(0), image, patch, heads, **kw)
    p._active_fold = 0
    got = p.predict_sliding_window_return_logits(image)
    assert np.array_equal(_bits(got), _bits(want))
    p1 = _predictor(spec, patch, sds[:1], mirror=mirror, accumulate_in='fp16_autocast')      # one fold: labels of `want`
    for ring in (2, 3):
        os.environ['FNN_GATHER_RING'] = str(ring)
        try:
            assert torch.equal(p.predict_sliding_window_return_logits(image), got)
            assert torch.equal(p1.predict_segmentation_from_preprocessed_data(image).long().cpu(), osw.logits_to_labels(want).long())
        finally:
            os.environ.pop('FNN_GATHER_RING', None)
    assert torch.equal(p1.predict_segmentation_from_preprocessed_data(image).long().cpu(), osw.logits_to_labels(want).long())
    want2 = osw.ensemble_logits([net(0), net(1)], image, patch, heads, **kw)
    assert np.array_equal(_bits(p.predict_logits_from_preprocessed_data(image)), _bits(want2))
    # differs from the no-autocast arithmetic (else the test above proves nothing)
    q = _predictor(spec, patch, sds, mirror=mirror, accumulate_in='fp16')
    assert not torch.equal(q.predict_sliding_window_return_logits(image), got)
    os.environ['FNN_NO_GATHER'] = '1'
    try:
        r = _predictor(spec, patch, sds, mirror=mirror, accumulate_in='fp16_autocast')
        with pytest.raises(NotImplementedError, match='AUTOCAST'):
            r.predict_sliding_window_return_logits(image)
    finally:
        os.environ.pop('FNN_NO_GATHER', None)


@pytest.mark.parametrize('heads,world,mirror', [(3, 2, None), (3, 4, None), (61, 8, None), (3, 4, (0, 1, 2)), (61, 2, (1,)), (70, 2, None)])
def test_sharded_gather_path_through_c_abi_is_bit_identical_to_single_gpu(heads, world, mirror):
    """fnn_patch_features / fnn_gather_box for `world` virtual ranks on one GPU, the feature exchange done with local
    copies of exactly the regions FeatureExchange would send: logits and labels of every owned box must be the bits of
    the single-GPU predictor (the gather kernel visits a voxel's covering patches in the reference's order whoever
    computed them).  With test-time mirroring (the reference's default, predict_from_raw_data.py:42) the 2^k
    evaluations' activations travel too, each as the flipped sub-block the kernel reads."""
    from fast_nnunet_amd import capi
    from fast_nnunet_amd.dist import Decomposition, mirror_flips, unpadded
    spec = UNetSpec('plain', 1, heads, [16, 32], [(3, 3, 3)] * 2, [(1, 1, 1), (2, 2, 2)], [2, 2], [2])
    patch = (16, 16, 32)
    p = _predictor(spec, patch, [synthetic_state_dict(spec, 33)], mirror=mirror)
    image = torch.randn(1, 37, 30, 70, generator=torch.Generator().manual_seed(6))
    want = p.predict_sliding_window_return_logits(image)
    want_labels = p.predict_segmentation_from_preprocessed_data(image)
    x = image.cuda().float().contiguous()
    padded, pad_lo, origins = capi.plan_volume(patch, x.shape[1:], 0.5)
    steps = [sorted(set(int(v) for v in origins[:, d])) for d in range(3)]
    dec = Decomposition.build(patch, padded, steps, world)
    opts, eng = p._opts(), p._engine
    C = eng.feature_channels
    flips = mirror_flips(mirror)
    E = len(flips)
    feats = []
    for r in range(world):                                           # every rank: its own patches (boundary first)
        if dec.owned[r] is None:
            feats.append(None)
            continue
        boundary, interior = dec.split_patches_for_features(r, patch, origins)
        slot_of = {pid: i for i, pid in enumerate(boundary + interior)}
        for _, pid, _ in dec.feature_transfers(r, patch, origins)[1]:
            slot_of.setdefault(pid, len(slot_of))
        n_slots = len(slot_of)
        feat = torch.zeros((E, n_slots, *patch, C), dtype=torch.half, device='cuda')
        fss = torch.zeros((E, n_slots, 2, C), dtype=torch.float32, device='cuda')
        for ids, off in ((boundary, 0), (interior, len(boundary))):
            if ids:
                eng.patch_features(x.data_ptr(), x.shape, opts, ids, feat.data_ptr(), fss.data_ptr(), slot0=off, n_slots=n_slots)
        feats.append((feat, fss, slot_of))
    torch.cuda.synchronize()
    for r in range(world):                                           # local stand-in for FeatureExchange
        if feats[r] is None:
            continue
        feat, fss, slot_of = feats[r]
        for peer, pid, reg in dec.feature_transfers(r, patch, origins)[1]:
            pf, ps, pslot = feats[peer]
            o = [int(v) for v in origins[pid]]
            for f, fl in enumerate(flips):
                loc = tuple(slice(patch[d] - (reg[1][d] - o[d]), patch[d] - (reg[0][d] - o[d])) if d in fl
                            else slice(reg[0][d] - o[d], reg[1][d] - o[d]) for d in range(3))
                feat[(f, slot_of[pid], *loc)] = pf[(f, pslot[pid], *loc)]
                fss[f, slot_of[pid]] = ps[f, pslot[pid]]
    got = torch.zeros_like(want)
    labels = torch.full_like(want_labels, 255)
    for r in range(world):
        if feats[r] is None:
            continue
        feat, fss, slot_of = feats[r]
        table = np.full(origins.shape[0], -1, np.int32)
        for pid, sl in slot_of.items():
            table[pid] = sl
        own = unpadded(dec.owned[r], pad_lo, x.shape[1:])
        if own is not None:                                          # (labels from fnn_gather_box: <= 63 classes, one pass over the heads)
            eng.gather_box(feat.data_ptr(), fss.data_ptr(), table, x.shape, opts, own[0], own[1], logits_ptr=got.data_ptr(),
                           labels_ptr=labels.data_ptr() if heads <= 63 else None, n_slots=len(slot_of))
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert torch.equal(labels, want_labels) if heads <= 63 else torch.equal(want_labels.long(), got.float().argmax(0))


@pytest.mark.parametrize('mirror', [None, (0, 2)])
def test_pack_and_unpack_regions_are_the_strided_copies_they_replace(mirror):
    """fnn_pack_regions / fnn_unpack_regions (ABI 4): one launch per peer and direction instead of a torch slice per
    (evaluation, patch, region).  For every rank of an 8-rank decomposition: the packed message equals the concatenation of
    the sub-blocks FeatureExchange's host path would send, and unpacking a message into zeroed slots writes exactly those
    sub-blocks - with mirrored evaluations (blocks at [P - hi, P - lo) along a flipped axis) too."""
    from fast_nnunet_amd import capi
    from fast_nnunet_amd.dist import Decomposition, ExchangePlan, mirror_flips
    spec = UNetSpec('plain', 1, 3, [16, 32], [(3, 3, 3)] * 2, [(1, 1, 1), (2, 2, 2)], [2, 2], [2])
    patch = (16, 16, 32)
    p = _predictor(spec, patch, [synthetic_state_dict(spec, 35)], mirror=mirror)
    eng = p._engine
    C = eng.feature_channels
    padded, pad_lo, origins = capi.plan_volume(patch, (37, 30, 70), 0.5)
    steps = [sorted(set(int(v) for v in origins[:, d])) for d in range(3)]
    dec = Decomposition.build(patch, padded, steps, 8)
    flips = mirror_flips(mirror)
    g = torch.Generator().manual_seed(5)
    checked = 0
    for r in range(8):
        if dec.owned[r] is None:
            continue
        boundary, interior = dec.split_patches_for_features(r, patch, origins)
        slot_of = {pid: i for i, pid in enumerate(boundary + interior)}
        for _, pid, _ in dec.feature_transfers(r, patch, origins)[1]:
            slot_of.setdefault(pid, len(slot_of))
        n_slots = len(slot_of)
        feat = torch.randn((len(flips), n_slots, *patch, C), generator=g).half().cuda()
        plan = ExchangePlan(dec, r, patch, origins, slot_of, flips, C, feat.device, n_slots)
        for m in plan.send + plan.recv:
            want = torch.cat([feat[(rec[0], rec[1], slice(rec[2], rec[5]), slice(rec[3], rec[6]), slice(rec[4], rec[7]))].reshape(-1)
                              for rec in m['recs']])
            assert want.numel() == m['numel']
            buf = torch.full((m['numel'],), float('nan'), dtype=torch.half, device='cuda')
            eng.pack_regions(feat.data_ptr(), n_slots, m['table'].data_ptr(), len(m['recs']), buf.data_ptr(),
                             torch.cuda.current_stream().cuda_stream)
            assert torch.equal(buf, want)
            land = torch.zeros_like(feat)
            eng.unpack_regions(land.data_ptr(), n_slots, m['table'].data_ptr(), len(m['recs']), buf.data_ptr(),
                               torch.cuda.current_stream().cuda_stream)
            ref = torch.zeros_like(feat)
            for rec in m['recs']:
                sl = (rec[0], rec[1], slice(rec[2], rec[5]), slice(rec[3], rec[6]), slice(rec[4], rec[7]))
                ref[sl] = feat[sl]
            assert torch.equal(land, ref)
            checked += 1
    assert checked >= 8
