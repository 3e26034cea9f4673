"""Size-independent properties at BASELINE's full size (configs[1]: bone_turbo-like student, patch 160x96x96,
61 classes, 512^3 volume, 600 patches) - where the CPU oracle would need half an hour.

A network whose convolutions are all zero outputs the seg head's bias c at every voxel, so the sliding window must
return c everywhere: sum_k(c g_k) / sum_k(g_k) = c whatever the Gaussian weights, the number of visits (1..8) or the
patch a voxel falls into.  That checks, on the benchmark's own geometry, that every voxel of the volume is visited and
normalised by the matching weight sum, with the fp16 accumulators' rounding as the only error: each visit rounds the
product and the sum to fp16 (2^-11 relative), the weight sum likewise, the quotient once - bound used below:
12 roundings * 2^-11 = 6e-3 relative, 16 voxels away from the volume faces.  Closer to the faces the Gaussian weight
(10 e^-8 on a face, 10 e^-16 on an edge, clamped to 5.96e-8 in the corners) is an fp16 subnormal, c g is a handful of
5.96e-8 quanta and the quotient is a ratio of small integers - the reference's own fp16 accumulators do the same
(a corner voxel of c = 1.5 comes back as 2.0); there only "finite and within half of max(|c|, 1)" is asserted.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _constant_predictor(batch=32, accumulate_in='fp16'):
    import bench
    made = {}
    real = bench.synthetic_checkpoint

    def constant(features, kernels, strides, in_ch, heads, seed=1234):
        sd = real(features, kernels, strides, in_ch, heads, seed)
        g = torch.Generator().manual_seed(77)
        for k, v in sd.items():
            if k.endswith('norm.weight'):
                sd[k] = torch.ones_like(v)
            else:
                sd[k] = torch.zeros_like(v)
        last = max(int(k.split('.')[2]) for k in sd if k.startswith('decoder.seg_layers.'))
        c = (torch.rand(heads, generator=g) * 6 - 3).half().float()            # fp16-representable logits in [-3, 3)
        c[5] = c[2] = c.max() + 0.5                                             # a tie for the maximum: first wins
        sd[f'decoder.seg_layers.{last}.bias'] = c
        made['c'] = c
        return sd

    bench.synthetic_checkpoint = constant
    try:
        p, _, info = bench.build_predictor('bone_turbo_r2', torch.device('cuda', 0), batch, accumulate_in)
    finally:
        bench.synthetic_checkpoint = real
    return p, made['c'], info


def test_full_size_volume_constant_network_returns_the_head_bias_everywhere():
    p, c, info = _constant_predictor()
    assert tuple(info['patch']) == (160, 96, 96) and info['heads'] == 61
    vol = torch.randn((1, 512, 512, 512), generator=torch.Generator().manual_seed(0)).cuda()
    out = p.predict_sliding_window_return_logits(vol)
    assert out.shape == (61, 512, 512, 512) and out.dtype == torch.half and out.is_cuda
    worst, worst_border = 0.0, 0.0
    m = 16
    for h in range(61):
        o = out[h].float()
        assert bool(torch.isfinite(o).all())
        worst_border = max(worst_border, float((o - c[h]).abs().max()) / max(abs(float(c[h])), 1.0))
        worst = max(worst, float((o[m:-m, m:-m, m:-m] - c[h]).abs().max()) / max(abs(float(c[h])), 0.25))
    print(f'worst relative deviation from the head bias: interior {worst:.2e}, whole volume {worst_border:.2e}')
    assert worst <= 6e-3 and worst_border <= 0.51
    # a second run over the same volume reproduces the first bit for bit (several batches in flight, 600 patches)
    again = p.predict_sliding_window_return_logits(vol)
    assert torch.equal(out, again)
    del out, again
    # labels straight from the accumulators: the first of the two tied maxima, at every voxel
    labels = p.predict_segmentation_from_preprocessed_data(vol)
    assert labels.shape == (512, 512, 512) and int(labels.min()) == 2 and int(labels.max()) == 2


def test_full_size_volume_fp32_accumulators_are_exact_for_the_constant_network():
    """With fp32 accumulators the only roundings left are fp32: c comes back to 1e-6 relative (then one fp16 store)."""
    p, c, _ = _constant_predictor(accumulate_in='fp32')
    vol = torch.zeros((1, 512, 512, 512), device='cuda')
    out = p.predict_sliding_window_return_logits(vol)
    for h in (0, 2, 30, 60):
        assert torch.equal(out[h], torch.full_like(out[h], float(c[h])))           # c is fp16-representable


def test_full_size_volume_is_reproduced_bit_for_bit_with_three_batches_in_flight():
    """The benchmark network with its random weights: every kernel family of the hot path (stem, persistent thin and
    strided convs, ZR convs, transposed convs, seg head, finalize) runs next to the other two streams' kernels.  A
    kernel that is sensitive to what shares its CU ("stem under concurrency") corrupts different
    voxels every run, so three runs that agree bit for bit - and with a fourth at another batch size, which changes
    what overlaps what - rule that out."""
    import bench
    dev = torch.device('cuda', 0)
    p, _, _ = bench.build_predictor('bone_turbo_r2', dev, 32, 'fp16')
    vol = torch.randn((1, 512, 512, 512), generator=torch.Generator().manual_seed(0)).cuda()
    first = p.predict_sliding_window_return_logits(vol)
    assert bool(torch.isfinite(first[::7].float()).all())
    for _ in range(2):
        again = p.predict_sliding_window_return_logits(vol)
        assert torch.equal(first, again)
        del again
    p.patches_per_forward = 20                    # other batch boundaries (what overlaps what), same kernels per layer
    other = p.predict_sliding_window_return_logits(vol)
    assert torch.equal(first, other)
    del other
    # an engine PLANNED for another batch size may pick other kernel variants per layer (other summation orders):
    # same result within the fp16 network tolerance, not bit for bit
    q, _, _ = bench.build_predictor('bone_turbo_r2', dev, 20, 'fp16')
    other = q.predict_sliding_window_return_logits(vol)
    scale = float(first[::5].float().abs().max())
    m = 16                                                    # away from the subnormal-weight border (module docstring)
    worst = max(float((first[h, m:-m, m:-m, m:-m].float() - other[h, m:-m, m:-m, m:-m].float()).abs().max())
                for h in range(0, 61, 6))
    print(f'planned batch 32 vs 20: max |diff| {worst:.3g} of max |logit| {scale:.3g}')
    assert worst <= 1e-2 * scale
