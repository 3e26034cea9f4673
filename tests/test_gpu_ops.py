"""Per-kernel parity on a real MI355X: every HIP kernel is called through the C ABI
(fnn_op_*) and compared with torch's own CPU fp32 ops on the same fp16-rounded
operands.  Tolerances: the kernels compute fp16 x fp16 products exactly and
accumulate in fp32, the only other error is the final fp16 store, so
|err| <= 2^-10 * |ref| + small absolute slack."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _h(a):
    """round to fp16 like the kernel's operands"""
    return torch.as_tensor(a).half().float()


def _check(y, ref, what):
    ref = ref.numpy()
    err = np.abs(y - ref)
    tol = 2e-3 * np.abs(ref) + 2e-3 * max(1.0, float(np.abs(ref).max())) * 0.5
    assert (err <= tol).all(), f'{what}: max err {err.max():.4g} (ref max {np.abs(ref).max():.4g})'


CONV_CASES = [
    # n, cin, cout, dims, k, stride
    (1, 16, 16, (8, 16, 16), (3, 3, 3), (1, 1, 1)),
    (2, 32, 16, (8, 8, 16), (3, 3, 3), (1, 1, 1)),
    (1, 16, 32, (8, 16, 16), (3, 3, 3), (2, 2, 2)),
    (1, 16, 16, (6, 12, 20), (1, 3, 3), (1, 1, 1)),
    (1, 16, 32, (8, 12, 12), (1, 3, 3), (1, 2, 2)),
    (2, 32, 48, (10, 6, 6), (3, 3, 3), (2, 1, 1)),
    (1, 64, 64, (4, 4, 4), (3, 3, 3), (1, 1, 1)),
    (1, 160, 160, (5, 3, 3), (3, 3, 3), (1, 1, 1)),
    (1, 8, 24, (7, 9, 11), (3, 3, 3), (1, 1, 1)),          # channel padding (8 -> 16, 24 -> 32), ragged tiles
    (1, 21, 10, (5, 9, 8), (3, 3, 3), (1, 1, 1)),          # odd channel counts (r = 3, 6 students)
    (1, 16, 16, (4, 8, 8), (1, 1, 1), (1, 1, 1)),
    (3, 48, 80, (4, 6, 6), (3, 1, 3), (1, 1, 1)),
    (1, 320, 160, (4, 4, 4), (3, 3, 3), (1, 1, 1)),
    # depth stride 1, in-plane stride 2 with enough tiles for the depth-shift strided kernel (conv3d_zs_kernel): whole
    # tiles; tiles ragged in every dimension on odd input sizes; two 16-channel chunks and 64 output channels
    (2, 16, 32, (64, 96, 128), (3, 3, 3), (1, 2, 2)),
    (4, 16, 32, (20, 91, 94), (3, 3, 3), (1, 2, 2)),
    (2, 32, 64, (24, 63, 96), (3, 3, 3), (1, 2, 2)),
]


@pytest.mark.parametrize('n,cin,cout,dims,k,stride', CONV_CASES)
def test_conv3d_identity_input(n, cin, cout, dims, k, stride):
    from fast_nnunet_amd import capi
    g = torch.Generator().manual_seed(hash((n, cin, cout, dims, k)) % 2 ** 31)
    x = _h(torch.randn(n, cin, *dims, generator=g))
    w = _h(torch.randn(cout, cin, *k, generator=g) / (cin * k[0] * k[1] * k[2]) ** 0.5)
    b = torch.randn(cout, generator=g)
    y, stats = capi.op_conv3d(x.numpy(), w.numpy(), b.numpy(), k, stride, want_stats=True)
    ref = F.conv3d(x, w, b, stride, [(i - 1) // 2 for i in k])
    _check(y, ref, 'conv3d')
    # epilogue statistics = sums over the fp16-rounded outputs
    y64 = y.astype(np.float64)
    assert np.allclose(stats[..., 0], y64.sum((2, 3, 4)), rtol=1e-6, atol=1e-3)
    assert np.allclose(stats[..., 1], (y64 ** 2).sum((2, 3, 4)), rtol=1e-6, atol=1e-3)


# Layers large enough for the persistent / row / walking variants the launchers keep for them (the launcher picks by shape
# and planned batch; a trace of the whole GPU suite in round 3 found these variants compiled but never launched).  Each case
# names the variant it must run (fnn_op_last_kernels): n, cin, cin2 (second source), cout, dims, k, expected kernel.
VARIANT_CASES = [
    (16, 16, 0, 16, (16, 64, 80), (1, 3, 3), 'conv3d_persist_kernel<1,4,1,5,1,4,0>'),     # thin (1,3,3) layer, rows of 80: no row kernel
    (16, 16, 16, 16, (16, 64, 80), (1, 3, 3), 'conv3d_persist_kernel<1,4,1,5,2,4,0>'),    # ... two sources
    (16, 48, 0, 16, (16, 64, 80), (1, 3, 3), 'conv3d_persist_kernel<1,8,1,5,0,8,0>'),     # ... three chunks
    (18, 16, 0, 16, (4, 96, 80), (1, 3, 3), 'conv3d_persist_kernel<1,4,1,5,0,8,0>'),      # ... fewer than 8 planes
    (16, 16, 0, 16, (16, 64, 80), (3, 3, 1), 'conv3d_persist_kernel<1,4,1,5,1,4,0>'),     # nine taps the other way round
    (16, 16, 0, 16, (16, 64, 80), (1, 1, 1), 'conv3d_persist_kernel<1,8,1,0,0,8,0>'),     # 1x1x1 (a ResEnc skip projection at full size)
    (18, 16, 0, 16, (4, 96, 80), (1, 1, 1), 'conv3d_persist_kernel<1,4,1,0,0,8,0>'),
    (13, 176, 0, 16, (16, 64, 80), (1, 3, 3), 'conv3d_persist_kernel<1,8,0,5,0,8,0>'),    # weights too large to stay resident
    (18, 176, 0, 16, (4, 96, 80), (1, 3, 3), 'conv3d_persist_kernel<1,4,0,5,0,8,0>'),
    (8, 16, 0, 16, (16, 64, 48), (1, 3, 3), 'conv3d_lds_kernel<1,8,8>'),                   # too few tiles for the persistent form
    (2, 16, 16, 16, (8, 16, 128), (1, 3, 3), 'conv_row_kernel<8,2,0>'),                    # two-source rows of 128 / 160 voxels
    (2, 16, 16, 16, (8, 16, 160), (1, 3, 3), 'conv_row_kernel<10,2,0>'),
    (4, 16, 0, 32, (64, 64, 64), (3, 3, 3), 'conv3d_zrw_kernel<2>'),                       # walking depth-shift kernel, two cout blocks
]


@pytest.mark.parametrize('n,cin,cin2,cout,dims,k,kernel', VARIANT_CASES, ids=lambda v: str(v).replace(' ', ''))
def test_conv3d_large_layer_variants(n, cin, cin2, cout, dims, k, kernel):
    from fast_nnunet_amd import capi
    g = torch.Generator().manual_seed(hash((n, cin, cin2, cout, dims, k)) % 2 ** 31)
    x = _h(torch.randn(n, cin, *dims, generator=g) * 2 - 0.5)
    gamma, beta = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.1
    x2 = _h(torch.randn(n, cin2, *dims, generator=g)) if cin2 else None
    w = _h(torch.randn(cout, cin + cin2, *k, generator=g) / ((cin + cin2) * k[0] * k[1] * k[2]) ** 0.5)
    b = torch.randn(cout, generator=g)
    y, stats = capi.op_conv3d(x.numpy(), w.numpy(), b.numpy(), k, (1, 1, 1), gamma=gamma.numpy(), beta=beta.numpy(), slope=0.01,
                              x2=None if x2 is None else x2.numpy(), want_stats=True)
    assert capi.op_last_kernels() == [kernel]
    torch.set_num_threads(max(8, torch.get_num_threads()))
    xn = _h(F.leaky_relu(F.instance_norm(x, weight=gamma, bias=beta, eps=1e-5), 0.01))
    ref = F.conv3d(xn if x2 is None else torch.cat((xn, x2), 1), w, b, 1, [(i - 1) // 2 for i in k])
    err = np.abs(y - ref.numpy())
    assert err.max() <= 6e-3 * max(1.0, float(ref.abs().max())), err.max()
    y64 = y.astype(np.float64)
    assert np.allclose(stats[..., 0], y64.sum((2, 3, 4)), rtol=1e-6, atol=1e-2)
    assert np.allclose(stats[..., 1], (y64 ** 2).sum((2, 3, 4)), rtol=1e-6, atol=1e-2)


def test_conv3d_linear_taps_on_a_layer_too_large_for_the_depth_shift_kernels():
    """3 x 3 x 3, stride 1, 9.4 M voxels per item: beyond the depth-shift kernels' 24-bit voxel arithmetic (conv3d_zr.hip: zr_pick),
    so the layer keeps the linear tap order and runs the persistent kernel with 14 unrolled k-steps and travelling weights."""
    from fast_nnunet_amd import capi
    g = torch.Generator().manual_seed(77)
    dims = (256, 192, 192)
    x = _h(torch.randn(1, 16, *dims, generator=g))
    w = _h(torch.randn(16, 16, 3, 3, 3, generator=g) / 432 ** 0.5)
    b = torch.randn(16, generator=g)
    y = capi.op_conv3d(x.numpy(), w.numpy(), b.numpy(), (3, 3, 3), (1, 1, 1))
    assert capi.op_last_kernels() == ['conv3d_persist_kernel<1,8,0,14,0,8,0>']
    torch.set_num_threads(max(8, torch.get_num_threads()))
    sl = (slice(None), slice(None), slice(100, 140), slice(0, 64), slice(120, 192))       # a slab incl. two faces: the CPU conv of it all takes a minute
    ref = F.conv3d(x[:, :, 99:141, 0:65, 119:192], w, b, 1, 1)[:, :, 1:-1, :-1, 1:]
    _check(y[sl], ref, 'conv3d 256x192x192')


@pytest.mark.parametrize('cout', [16, 32, 64])
def test_conv3d_generic_kernel_behind_every_launcher(cout):
    """conv3d_mfma_kernel<1 | 2 | 4>: the form every layer shape can fall back to when no specialised launcher takes it (no
    BASELINE workload does: tests/test_gpu_bench_oracle.py).  FNN_CONV_V1 (read per call) sends a layer there."""
    import os
    from fast_nnunet_amd import capi
    g = torch.Generator().manual_seed(5 + cout)
    for k, stride, dims in (((3, 3, 3), (1, 1, 1), (7, 9, 20)), ((1, 3, 3), (1, 2, 2), (5, 12, 18)), ((1, 1, 1), (1, 1, 1), (3, 8, 9))):
        x = _h(torch.randn(2, 24, *dims, generator=g))
        w = _h(torch.randn(cout, 24, *k, generator=g) / (24 * k[0] * 9) ** 0.5)
        b = torch.randn(cout, generator=g)
        os.environ['FNN_CONV_V1'] = '1'
        try:
            y = capi.op_conv3d(x.numpy(), w.numpy(), b.numpy(), k, stride)
        finally:
            del os.environ['FNN_CONV_V1']
        assert capi.op_last_kernels() == [f'conv3d_mfma_kernel<{cout // 16 if cout < 64 else 4}> (generic fallback)']
        _check(y, F.conv3d(x, w, b, stride, [(i - 1) // 2 for i in k]), f'generic conv {k} {stride}')


@pytest.mark.parametrize('n,cin,cout,dims,k,stride', CONV_CASES[:6] + CONV_CASES[-2:])
def test_conv3d_with_fused_instancenorm_lrelu_on_load(n, cin, cout, dims, k, stride):
    from fast_nnunet_amd import capi
    g = torch.Generator().manual_seed(7 + cin + cout)
    x = _h(torch.randn(n, cin, *dims, generator=g) * 3 + 1.5)
    gamma = torch.rand(cin, generator=g) + 0.5
    beta = torch.randn(cin, generator=g) * 0.1
    w = _h(torch.randn(cout, cin, *k, generator=g) / (cin * k[0] * k[1] * k[2]) ** 0.5)
    b = torch.randn(cout, generator=g)
    y = capi.op_conv3d(x.numpy(), w.numpy(), b.numpy(), k, stride, gamma=gamma.numpy(), beta=beta.numpy(), slope=0.01)
    xn = F.leaky_relu(F.instance_norm(x, weight=gamma, bias=beta, eps=1e-5), 0.01)
    ref = F.conv3d(xn, w, b, stride, [(i - 1) // 2 for i in k])
    # the kernel rounds the normalised activation to fp16 before the MFMA
    ref16 = F.conv3d(_h(xn), w, b, stride, [(i - 1) // 2 for i in k])
    err = np.abs(y - ref16.numpy())
    assert err.max() <= 6e-3 * max(1.0, float(ref.abs().max())), err.max()


def test_conv3d_two_sources_replaces_concat():
    from fast_nnunet_amd import capi
    g = torch.Generator().manual_seed(3)
    n, c1, c2, cout, dims = 2, 16, 16, 16, (8, 8, 16)
    up = _h(torch.randn(n, c1, *dims, generator=g))                 # transposed-conv output: identity on load
    skip = _h(torch.randn(n, c2, *dims, generator=g) * 2 - 1)       # raw conv output: norm + lrelu on load
    gamma, beta = torch.rand(c2, generator=g) + 0.5, torch.randn(c2, generator=g) * 0.1
    w = _h(torch.randn(cout, c1 + c2, 3, 3, 3, generator=g) / 30)
    b = torch.randn(cout, generator=g)
    y = capi.op_conv3d(up.numpy(), w.numpy(), b.numpy(), (3, 3, 3), (1, 1, 1),
                       x2=skip.numpy(), gamma2=gamma.numpy(), beta2=beta.numpy(), slope2=0.01)
    cat = torch.cat((up, _h(F.leaky_relu(F.instance_norm(skip, weight=gamma, bias=beta, eps=1e-5), 0.01))), 1)
    ref = F.conv3d(cat, w, b, 1, 1)
    assert np.abs(y - ref.numpy()).max() <= 6e-3 * max(1.0, float(ref.abs().max()))


def test_conv3d_two_sources_with_padded_channels():
    from fast_nnunet_amd import capi
    g = torch.Generator().manual_seed(4)
    n, c1, c2, cout, dims = 1, 8, 8, 8, (4, 8, 8)
    a = _h(torch.randn(n, c1, *dims, generator=g))
    bsrc = _h(torch.randn(n, c2, *dims, generator=g))
    w = _h(torch.randn(cout, c1 + c2, 3, 3, 3, generator=g) / 20)
    y = capi.op_conv3d(a.numpy(), w.numpy(), None, (3, 3, 3), (1, 1, 1), x2=bsrc.numpy())
    ref = F.conv3d(torch.cat((a, bsrc), 1), w, None, 1, 1)
    _check(y, ref, 'conv3d 2-src padded')


TCONV_CASES = [
    (1, 32, 16, (4, 8, 8), (2, 2, 2)),
    (2, 160, 160, (5, 3, 3), (2, 1, 1)),
    (1, 64, 32, (4, 6, 6), (1, 2, 2)),
    (1, 16, 8, (3, 5, 7), (2, 2, 2)),
    (1, 80, 64, (2, 4, 4), (2, 2, 2)),
    (1, 21, 10, (3, 4, 5), (2, 2, 2)),
]


@pytest.mark.parametrize('n,cin,cout,dims,stride', TCONV_CASES)
def test_conv_transpose3d(n, cin, cout, dims, stride):
    from fast_nnunet_amd import capi
    g = torch.Generator().manual_seed(11 + cin)
    x = _h(torch.randn(n, cin, *dims, generator=g) * 2 + 0.5)
    gamma, beta = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.1
    w = _h(torch.randn(cin, cout, *stride, generator=g) / cin ** 0.5)
    b = torch.randn(cout, generator=g)
    y = capi.op_conv_transpose3d(x.numpy(), w.numpy(), b.numpy(), stride, gamma=gamma.numpy(), beta=beta.numpy(),
                                 slope=0.01)
    xn = _h(F.leaky_relu(F.instance_norm(x, weight=gamma, bias=beta, eps=1e-5), 0.01))
    ref = F.conv_transpose3d(xn, w, b, stride)
    assert np.abs(y - ref.numpy()).max() <= 6e-3 * max(1.0, float(ref.abs().max()))
    y2 = capi.op_conv_transpose3d(x.numpy(), w.numpy(), b.numpy(), stride)
    _check(y2, F.conv_transpose3d(x, w, b, stride), 'tconv identity')


def test_mfma_operand_map_with_asymmetric_integer_data():
    """A = I style check with exact integers: catches transposed / permuted fragment maps."""
    from fast_nnunet_amd import capi
    cin = cout = 16
    x = torch.zeros(1, cin, 4, 8, 8)
    for c in range(cin):
        x[0, c] = (torch.arange(4 * 8 * 8).reshape(4, 8, 8) % 7) + c          # small ints, channel-dependent
    w = torch.zeros(cout, cin, 3, 3, 3)
    for co in range(cout):
        w[co, (co * 5 + 3) % cin, co % 3, (co // 3) % 3, (co + 1) % 3] = 1.0   # one tap, asymmetric permutation
    y = capi.op_conv3d(x.numpy(), w.numpy(), None, (3, 3, 3), (1, 1, 1))
    ref = F.conv3d(x, w, None, 1, 1)
    assert np.array_equal(y, ref.numpy())


# ---- conv3d_zr_kernel (depth-shift operand reuse): shapes large enough for the launcher to pick it
ZR_CASES = [
    # n, cin, cout, dims                       variant
    (4, 32, 32, (61, 45, 43)),               # <2, 8>, ragged tiles on every axis
    (6, 16, 16, (64, 40, 40)),               # <1, 8>
    (3, 16, 16, (36, 48, 50)),               # <1, 4>
    (2, 32, 64, (20, 56, 56)),               # <2, 4>, two workgroup columns
    (5, 40, 24, (33, 41, 47)),               # channel padding on both sides (3 chunks, 2 cout blocks)
]


@pytest.mark.parametrize('n,cin,cout,dims', ZR_CASES)
def test_conv3d_zr_variants(n, cin, cout, dims):
    from fast_nnunet_amd import capi
    g = torch.Generator().manual_seed(11 + cin + cout + dims[0])
    x = _h(torch.randn(n, cin, *dims, generator=g) * 2 + 0.5)
    gamma = torch.rand(cin, generator=g) + 0.5
    beta = torch.randn(cin, generator=g) * 0.1
    w = _h(torch.randn(cout, cin, 3, 3, 3, generator=g) / (cin * 27) ** 0.5)
    b = torch.randn(cout, generator=g)
    y, stats = capi.op_conv3d(x.numpy(), w.numpy(), b.numpy(), (3, 3, 3), (1, 1, 1), want_stats=True)
    _check(y, F.conv3d(x, w, b, 1, 1), 'conv3d zr')
    y64 = y.astype(np.float64)
    assert np.allclose(stats[..., 0], y64.sum((2, 3, 4)), rtol=1e-6, atol=1e-3)
    assert np.allclose(stats[..., 1], (y64 ** 2).sum((2, 3, 4)), rtol=1e-6, atol=1e-3)
    y = capi.op_conv3d(x.numpy(), w.numpy(), b.numpy(), (3, 3, 3), (1, 1, 1), gamma=gamma.numpy(), beta=beta.numpy(),
                       slope=0.01)
    xn = _h(F.leaky_relu(F.instance_norm(x, weight=gamma, bias=beta, eps=1e-5), 0.01))
    ref = F.conv3d(xn, w, b, 1, 1)
    assert np.abs(y - ref.numpy()).max() <= 6e-3 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize('n,cin,cin2,cout,dims', [(32, 160, 0, 160, (20, 6, 6)), (32, 48, 32, 64, (30, 5, 7)), (96, 32, 0, 64, (10, 3, 3))])
def test_conv3d_zr_six_row_tiles(n, cin, cin2, cout, dims, monkeypatch):
    """conv3d_zr_kernel<2, 10, 6> (round 5): planes of at most 6 x 8 voxels in layers whose depth is a multiple of 10 - the
    160-channel stages of the benchmark net (20 x 6 x 6, 10 x 3 x 3) - as tiles of 10 x 6 x 8 on three waves.  Full and ragged
    planes, two sources, 1 - 5 cout groups; the OUTPUT BITS must be the 8 x 8 x 8 kernel's (FNN_NO_ZR6), the statistics the
    same sums."""
    from fast_nnunet_amd import capi
    monkeypatch.delenv('FNN_NO_ZR6', raising=False)
    g = torch.Generator().manual_seed(17 + cin + dims[1])
    x = _h(torch.randn(n, cin, *dims, generator=g) * 2 + 0.5)
    gamma, beta = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.1
    w = _h(torch.randn(cout, cin + cin2, 3, 3, 3, generator=g) / ((cin + cin2) * 27) ** 0.5)
    b = torch.randn(cout, generator=g)
    kw = dict(gamma=gamma.numpy(), beta=beta.numpy(), slope=0.01, want_stats=True)
    x2 = None
    if cin2:
        x2 = _h(torch.randn(n, cin2, *dims, generator=g))
        kw.update(x2=x2.numpy())
    y, stats = capi.op_conv3d(x.numpy(), w.numpy(), b.numpy(), (3, 3, 3), (1, 1, 1), **kw)
    assert capi.op_last_kernels() == ['conv3d_zr_kernel<2,10,6>'], capi.op_last_kernels()
    xn = _h(F.leaky_relu(F.instance_norm(x, weight=gamma, bias=beta, eps=1e-5), 0.01))
    ref = F.conv3d(xn if x2 is None else torch.cat((xn, x2), 1), w, b, 1, 1)
    assert np.abs(y - ref.numpy()).max() <= 6e-3 * max(1.0, float(ref.abs().max()))
    y64 = y.astype(np.float64)
    assert np.allclose(stats[..., 0], y64.sum((2, 3, 4)), rtol=1e-6, atol=1e-3)
    assert np.allclose(stats[..., 1], (y64 ** 2).sum((2, 3, 4)), rtol=1e-6, atol=1e-3)
    monkeypatch.setenv('FNN_NO_ZR6', '1')
    y_old, stats_old = capi.op_conv3d(x.numpy(), w.numpy(), b.numpy(), (3, 3, 3), (1, 1, 1), **kw)
    assert capi.op_last_kernels() != ['conv3d_zr_kernel<2,10,6>']
    monkeypatch.delenv('FNN_NO_ZR6')
    if any(k.startswith('conv3d_zr_kernel') for k in capi.op_last_kernels()):      # the same arithmetic per output value
        assert np.array_equal(y.view(np.uint16), y_old.view(np.uint16))
    assert np.allclose(stats, stats_old, rtol=2e-6, atol=1e-3)


@pytest.mark.parametrize('n,cin,cout,dims', [(1, 16, 16, (45, 16, 16)),     # few windows: three d-segments per window, the last tile ragged
                                             (2, 8, 32, (37, 24, 9)),       # two cout blocks, padded input channels, ragged everywhere
                                             (3, 16, 16, (64, 40, 40))])    # one segment of eight tiles per window
def test_conv3d_zr_walking_kernel_for_single_chunk_layers(n, cin, cout, dims):
    """conv3d_zrw_kernel (Cin <= 16, >= 4 tiles along d): a workgroup walks the d-tiles of an 8 x 8 window with the
    weights resident and the halo planes in a ring - exact on integer data (ring slots, plane order, segment starts, the
    zero planes at both ends), within the op tolerance with the fused norm, and its per-tile statistics rows add up."""
    from fast_nnunet_amd import capi
    base = (torch.arange(dims[0] * dims[1] * dims[2]).reshape(dims) * 5 % 19).float()
    x = torch.stack([torch.stack([base + ch + 2 * i for ch in range(cin)]) for i in range(n)])
    w = torch.zeros(cout, cin, 3, 3, 3)
    for co in range(cout):
        w[co, (co * 5 + 3) % cin, co % 3, (co // 3) % 3, (co + 1) % 3] = 1.0
        w[co, (co * 3 + 1) % cin, (co + 2) % 3, (co + 1) % 3, co % 3] += 2.0
    y = capi.op_conv3d(x.numpy(), w.numpy(), None, (3, 3, 3), (1, 1, 1))
    assert np.array_equal(y, F.conv3d(x, w, None, 1, 1).numpy())
    g = torch.Generator().manual_seed(21 + cin + cout)
    x = _h(torch.randn(n, cin, *dims, generator=g) * 2 + 0.5)
    gamma, beta = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.1
    w = _h(torch.randn(cout, cin, 3, 3, 3, generator=g) / (cin * 27) ** 0.5)
    b = torch.randn(cout, generator=g)
    y, stats = capi.op_conv3d(x.numpy(), w.numpy(), b.numpy(), (3, 3, 3), (1, 1, 1), gamma=gamma.numpy(), beta=beta.numpy(),
                              slope=0.01, want_stats=True)
    xn = _h(F.leaky_relu(F.instance_norm(x, weight=gamma, bias=beta, eps=1e-5), 0.01))
    ref = F.conv3d(xn, w, b, 1, 1)
    assert np.abs(y - ref.numpy()).max() <= 6e-3 * max(1.0, float(ref.abs().max()))
    y64 = y.astype(np.float64)
    assert np.allclose(stats[..., 0], y64.sum((2, 3, 4)), rtol=1e-6, atol=1e-3)
    assert np.allclose(stats[..., 1], (y64 ** 2).sum((2, 3, 4)), rtol=1e-6, atol=1e-3)


def test_conv3d_zr_two_sources():
    from fast_nnunet_amd import capi
    g = torch.Generator().manual_seed(5)
    n, c1, c2, cout, dims = 6, 32, 32, 32, (32, 48, 48)
    up = _h(torch.randn(n, c1, *dims, generator=g))
    skip = _h(torch.randn(n, c2, *dims, generator=g) * 2 - 1)
    gamma, beta = torch.rand(c2, generator=g) + 0.5, torch.randn(c2, generator=g) * 0.1
    w = _h(torch.randn(cout, c1 + c2, 3, 3, 3, generator=g) / 40)
    b = torch.randn(cout, generator=g)
    y = capi.op_conv3d(up.numpy(), w.numpy(), b.numpy(), (3, 3, 3), (1, 1, 1),
                       x2=skip.numpy(), gamma2=gamma.numpy(), beta2=beta.numpy(), slope2=0.01)
    cat = torch.cat((up, _h(F.leaky_relu(F.instance_norm(skip, weight=gamma, bias=beta, eps=1e-5), 0.01))), 1)
    ref = F.conv3d(cat, w, b, 1, 1)
    assert np.abs(y - ref.numpy()).max() <= 6e-3 * max(1.0, float(ref.abs().max()))


def test_conv3d_zr_operand_map_with_exact_integers():
    """one-hot taps on integer data: any permuted tap pair / depth shift / fragment lane shows up as inequality"""
    from fast_nnunet_amd import capi
    n, c, dims = 6, 16, (64, 40, 40)
    base = (torch.arange(dims[0] * dims[1] * dims[2]).reshape(dims) * 7 % 23).float()
    x = torch.stack([torch.stack([base + ch + 3 * i for ch in range(c)]) for i in range(n)])
    w = torch.zeros(c, c, 3, 3, 3)
    for co in range(c):
        w[co, (co * 5 + 3) % c, co % 3, (co // 3) % 3, (co + 1) % 3] = 1.0
        w[co, (co * 3 + 1) % c, (co + 2) % 3, (co + 1) % 3, co % 3] += 2.0
    y = capi.op_conv3d(x.numpy(), w.numpy(), None, (3, 3, 3), (1, 1, 1))
    assert np.array_equal(y, F.conv3d(x, w, None, 1, 1).numpy())


@pytest.mark.parametrize('stride', [(1, 2, 2), (2, 2, 2)])
@pytest.mark.parametrize('cin,cout', [(16, 32), (32, 64)])
def test_conv3d_strided_persistent_variant(stride, cin, cout):
    """Strided convs with enough tiles for the persistent strided kernel (tile ranges per workgroup, cross-tile
    prefetch; one chunk = templated, two chunks = generic form with two cout groups): ragged tiles on every axis, fused
    InstanceNorm + LeakyReLU on load, statistics."""
    from fast_nnunet_amd import capi
    g = torch.Generator().manual_seed(21 + stride[0])
    n, dims = 8, (61, 93, 90)
    x = _h(torch.randn(n, cin, *dims, generator=g) * 2 + 0.5)
    gamma = torch.rand(cin, generator=g) + 0.5
    beta = torch.randn(cin, generator=g) * 0.1
    w = _h(torch.randn(cout, cin, 3, 3, 3, generator=g) / (cin * 27) ** 0.5)
    b = torch.randn(cout, generator=g)
    y, stats = capi.op_conv3d(x.numpy(), w.numpy(), b.numpy(), (3, 3, 3), stride, want_stats=True)
    _check(y, F.conv3d(x, w, b, stride, 1), 'conv3d strided persistent')
    y64 = y.astype(np.float64)
    assert np.allclose(stats[..., 0], y64.sum((2, 3, 4)), rtol=1e-6, atol=1e-3)
    assert np.allclose(stats[..., 1], (y64 ** 2).sum((2, 3, 4)), rtol=1e-6, atol=1e-3)
    y = capi.op_conv3d(x.numpy(), w.numpy(), b.numpy(), (3, 3, 3), stride, gamma=gamma.numpy(), beta=beta.numpy(), slope=0.01)
    xn = _h(F.leaky_relu(F.instance_norm(x, weight=gamma, bias=beta, eps=1e-5), 0.01))
    ref = F.conv3d(xn, w, b, stride, 1)
    assert np.abs(y - ref.numpy()).max() <= 6e-3 * max(1.0, float(ref.abs().max()))


# ----------------------------------------------------------------------------------------------- conv3d_row.hip
ROW_CASES = [
    # n, sources, dims (d, h, w): (1, 3, 3) convs of 16 channels at full rows of 64 / 96 / 128 voxels
    (3, 1, (5, 24, 64)),        # one strip of 6 steps
    (2, 1, (7, 96, 96)),        # two strips of 48 rows (the benchmark's stage-0 geometry)
    (2, 1, (3, 40, 128)),
    (2, 2, (4, 48, 96)),        # two sources = torch.cat((up, skip), 1) without the fusion
    (1, 1, (2, 8, 64)),         # a strip of two steps: the group stream is mostly prologue and tail
    (2, 1, (3, 24, 160)),       # wide rows: the k-loop runs in two halves of the row's column blocks
    (1, 2, (2, 16, 192)),
]


@pytest.mark.parametrize('n,nsrc,dims', ROW_CASES)
def test_conv3d_row_streaming_kernel(n, nsrc, dims):
    """conv3d_row.hip (16 -> 16 channels, (1, 3, 3), full rows): against torch on the same fp16 operands with identity
    input, with InstanceNorm + LeakyReLU on load, and its statistics; zero padding along h at strip and plane borders."""
    from fast_nnunet_amd import capi
    g = torch.Generator().manual_seed(31 + dims[1] + dims[2] + nsrc)
    k, stride = (1, 3, 3), (1, 1, 1)
    x = _h(torch.randn(n, 16, *dims, generator=g) * 2 + 0.5)
    x2 = _h(torch.randn(n, 16, *dims, generator=g) - 0.25) if nsrc == 2 else None
    cin = 16 * nsrc
    w = _h(torch.randn(16, cin, *k, generator=g) / (cin * 9) ** 0.5)
    b = torch.randn(16, generator=g)
    gamma, beta = torch.rand(16, generator=g) + 0.5, torch.randn(16, generator=g) * 0.1
    kw = dict(x2=x2.numpy()) if nsrc == 2 else {}
    y, stats = capi.op_conv3d(x.numpy(), w.numpy(), b.numpy(), k, stride, want_stats=True, **kw)
    xin = torch.cat((x, x2), 1) if nsrc == 2 else x
    _check(y, F.conv3d(xin, w, b, stride, (0, 1, 1)), 'conv3d row')
    y64 = y.astype(np.float64)
    assert np.allclose(stats[..., 0], y64.sum((2, 3, 4)), rtol=1e-6, atol=1e-3)
    assert np.allclose(stats[..., 1], (y64 ** 2).sum((2, 3, 4)), rtol=1e-6, atol=1e-3)
    if nsrc == 2:
        y = capi.op_conv3d(x.numpy(), w.numpy(), b.numpy(), k, stride, x2=x2.numpy(), gamma2=gamma.numpy(), beta2=beta.numpy(), slope2=0.01)
        xn = torch.cat((x, _h(F.leaky_relu(F.instance_norm(x2, weight=gamma, bias=beta, eps=1e-5), 0.01))), 1)
    else:
        y = capi.op_conv3d(x.numpy(), w.numpy(), b.numpy(), k, stride, gamma=gamma.numpy(), beta=beta.numpy(), slope=0.01)
        xn = _h(F.leaky_relu(F.instance_norm(x, weight=gamma, bias=beta, eps=1e-5), 0.01))
    ref = F.conv3d(xn, w, b, stride, (0, 1, 1))
    assert np.abs(y - ref.numpy()).max() <= 6e-3 * max(1.0, float(ref.abs().max()))


def test_conv3d_row_operand_map_with_exact_integers():
    """one-hot taps on integer data: a wrong tap / column block / ring row / channel half shows up as inequality"""
    from fast_nnunet_amd import capi
    n, c, dims = 2, 16, (3, 96, 96)
    base = (torch.arange(dims[0] * dims[1] * dims[2]).reshape(dims) * 7 % 23).float()
    x = torch.stack([torch.stack([base + ch + 3 * i for ch in range(c)]) for i in range(n)])
    w = torch.zeros(c, c, 1, 3, 3)
    for co in range(c):
        w[co, (co * 5 + 3) % c, 0, co % 3, (co // 3) % 3] = 1.0
        w[co, (co * 3 + 1) % c, 0, (co + 1) % 3, (co + 2) % 3] += 2.0
    y = capi.op_conv3d(x.numpy(), w.numpy(), None, (1, 3, 3), (1, 1, 1))
    assert np.array_equal(y, F.conv3d(x, w, None, 1, (0, 1, 1)).numpy())


@pytest.mark.parametrize('cin,cout,n,dims', [(32, 64, 8, (45, 61, 58)), (64, 128, 8, (45, 61, 58)), (48, 192, 8, (45, 61, 58)),
                                             (48, 160, 8, (45, 61, 58)), (32, 96, 8, (45, 61, 58)),
                                             (128, 160, 32, (40, 12, 12)), (48, 160, 32, (38, 11, 9))])
def test_conv3d_stride2_grouped_kernel(cin, cout, n, dims):
    """conv3d_s2.hip (3x3x3, stride (2,2,2), whole groups of 64 output channels per staged halo, 512-thread persistent
    workgroups): ragged 4 x 8 x 8 tiles on every axis, 2 - 4 chunks, 1 - 3 cout groups, InstanceNorm + LeakyReLU on load,
    statistics across units and batch items.  Round 5: a last group of two cout blocks (160 = 64 + 64 + 32 channels), and the
    <13, 3> form for output planes of at most 6 x 6 (13 x 13 of the 17 x 17 halo staged, three of the four column blocks)."""
    from fast_nnunet_amd import capi
    g = torch.Generator().manual_seed(41 + cin)
    stride = (2, 2, 2)
    kernel = 'conv3d_s2_kernel<13,3>' if dims[1] <= 12 else 'conv3d_s2_kernel'
    x = _h(torch.randn(n, cin, *dims, generator=g) * 2 + 0.5)
    gamma = torch.rand(cin, generator=g) + 0.5
    beta = torch.randn(cin, generator=g) * 0.1
    w = _h(torch.randn(cout, cin, 3, 3, 3, generator=g) / (cin * 27) ** 0.5)
    b = torch.randn(cout, generator=g)
    y, stats = capi.op_conv3d(x.numpy(), w.numpy(), b.numpy(), (3, 3, 3), stride, want_stats=True)
    assert capi.op_last_kernels() == [kernel], capi.op_last_kernels()
    _check(y, F.conv3d(x, w, b, stride, 1), 'conv3d s2')
    y64 = y.astype(np.float64)
    assert np.allclose(stats[..., 0], y64.sum((2, 3, 4)), rtol=1e-6, atol=1e-3)
    assert np.allclose(stats[..., 1], (y64 ** 2).sum((2, 3, 4)), rtol=1e-6, atol=1e-3)
    y = capi.op_conv3d(x.numpy(), w.numpy(), b.numpy(), (3, 3, 3), stride, gamma=gamma.numpy(), beta=beta.numpy(), slope=0.01)
    xn = _h(F.leaky_relu(F.instance_norm(x, weight=gamma, bias=beta, eps=1e-5), 0.01))
    ref = F.conv3d(xn, w, b, stride, 1)
    assert np.abs(y - ref.numpy()).max() <= 6e-3 * max(1.0, float(ref.abs().max()))


def test_conv3d_persistent_depth_shift_kernel_with_two_cout_groups():
    """conv3d_zsp_kernel (single 16-channel chunk, stride (1, 2, 2), >= 4096 tiles) with 64 output channels: two cout
    groups walk the same tiles (grid.y = 2); ragged tiles, fused InstanceNorm + LeakyReLU on load, statistics rows."""
    from fast_nnunet_amd import capi
    g = torch.Generator().manual_seed(77)
    n, cin, cout, dims, stride = 8, 16, 64, (61, 93, 90), (1, 2, 2)
    x = _h(torch.randn(n, cin, *dims, generator=g) * 2 + 0.5)
    gamma = torch.rand(cin, generator=g) + 0.5
    beta = torch.randn(cin, generator=g) * 0.1
    w = _h(torch.randn(cout, cin, 3, 3, 3, generator=g) / (cin * 27) ** 0.5)
    b = torch.randn(cout, generator=g)
    y, stats = capi.op_conv3d(x.numpy(), w.numpy(), b.numpy(), (3, 3, 3), stride, want_stats=True)
    assert 'conv3d_zsw_kernel' in capi.op_last_kernels()              # round 4: eight tiles along d -> the walking form ...
    os.environ['FNN_NO_ZSW'] = '1'
    try:                                                              # ... whose per-tile arithmetic is conv3d_zsp_kernel's: the same bits
        y_p, stats_p = capi.op_conv3d(x.numpy(), w.numpy(), b.numpy(), (3, 3, 3), stride, want_stats=True)
        assert 'conv3d_zsp_kernel' in capi.op_last_kernels()
    finally:
        os.environ.pop('FNN_NO_ZSW', None)
    assert np.array_equal(y, y_p) and np.array_equal(stats, stats_p)
    _check(y, F.conv3d(x, w, b, stride, 1), 'conv3d zsp two groups')
    y64 = y.astype(np.float64)
    assert np.allclose(stats[..., 0], y64.sum((2, 3, 4)), rtol=1e-6, atol=1e-3)
    assert np.allclose(stats[..., 1], (y64 ** 2).sum((2, 3, 4)), rtol=1e-6, atol=1e-3)
    y = capi.op_conv3d(x.numpy(), w.numpy(), b.numpy(), (3, 3, 3), stride, gamma=gamma.numpy(), beta=beta.numpy(), slope=0.01)
    xn = _h(F.leaky_relu(F.instance_norm(x, weight=gamma, bias=beta, eps=1e-5), 0.01))
    ref = F.conv3d(xn, w, b, stride, 1)
    assert np.abs(y - ref.numpy()).max() <= 6e-3 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize('n,cin,cout,dims', [(8, 128, 128, (40, 12, 12)), (16, 48, 64, (21, 11, 9)), (32, 32, 32, (17, 12, 10)),
                                             (64, 32, 32, (6, 12, 10))])
def test_conv3d_zr_whole_plane_tiles(n, cin, cout, dims, monkeypatch):
    """Planes of 9 .. 12 voxels per axis, whole planes per tile: conv3d_zq12_kernel (round 5: 8 x 12 x 12 tiles, eight waves
    with 5 + 4 column blocks per SIMD) where the layer is at least 8 deep, conv3d_zr12_kernel (4 x 12 x 12, nine one-block
    waves) below that.  Full and ragged planes, ragged depth, 2 - 8 chunks, 1 - 4 cout groups; identity input with
    statistics, then fused InstanceNorm + LeakyReLU on load; the two kernels' OUTPUT BITS must be equal."""
    from fast_nnunet_amd import capi
    monkeypatch.delenv('FNN_NO_ZQ12', raising=False)
    zq = dims[0] >= 8 and ((dims[0] + 7) // 8) * 8 * 10 <= ((dims[0] + 3) // 4) * 4 * 11      # conv3d_zq12_ok's depth rule: 40 and 21 yes, 17 and 6 no
    want_kernel = 'conv3d_zq12_kernel' if zq else 'conv3d_zr12_kernel<4>'
    g = torch.Generator().manual_seed(91 + cin + dims[1])
    x = _h(torch.randn(n, cin, *dims, generator=g) * 2 + 0.5)
    gamma = torch.rand(cin, generator=g) + 0.5
    beta = torch.randn(cin, generator=g) * 0.1
    w = _h(torch.randn(cout, cin, 3, 3, 3, generator=g) / (cin * 27) ** 0.5)
    b = torch.randn(cout, generator=g)
    y, stats = capi.op_conv3d(x.numpy(), w.numpy(), b.numpy(), (3, 3, 3), (1, 1, 1), want_stats=True)
    assert capi.op_last_kernels() == [want_kernel], capi.op_last_kernels()
    _check(y, F.conv3d(x, w, b, 1, 1), 'conv3d zr12')
    y64 = y.astype(np.float64)
    assert np.allclose(stats[..., 0], y64.sum((2, 3, 4)), rtol=1e-6, atol=1e-3)
    assert np.allclose(stats[..., 1], (y64 ** 2).sum((2, 3, 4)), rtol=1e-6, atol=1e-3)
    yn = capi.op_conv3d(x.numpy(), w.numpy(), b.numpy(), (3, 3, 3), (1, 1, 1), gamma=gamma.numpy(), beta=beta.numpy(), slope=0.01)
    xn = _h(F.leaky_relu(F.instance_norm(x, weight=gamma, bias=beta, eps=1e-5), 0.01))
    ref = F.conv3d(xn, w, b, 1, 1)
    assert np.abs(yn - ref.numpy()).max() <= 6e-3 * max(1.0, float(ref.abs().max()))
    if zq:
        monkeypatch.setenv('FNN_NO_ZQ12', '1')
        y_old, stats_old = capi.op_conv3d(x.numpy(), w.numpy(), b.numpy(), (3, 3, 3), (1, 1, 1), want_stats=True)
        assert capi.op_last_kernels() == ['conv3d_zr12_kernel<4>'], capi.op_last_kernels()
        yn_old = capi.op_conv3d(x.numpy(), w.numpy(), b.numpy(), (3, 3, 3), (1, 1, 1), gamma=gamma.numpy(), beta=beta.numpy(), slope=0.01)
        monkeypatch.delenv('FNN_NO_ZQ12')
        assert np.array_equal(y.view(np.uint16), y_old.view(np.uint16)) and np.array_equal(yn.view(np.uint16), yn_old.view(np.uint16))
        assert np.allclose(stats, stats_old, rtol=2e-6, atol=1e-3)


def test_conv3d_zr_whole_plane_operand_map_with_exact_integers():
    """one-hot taps on integer data through conv3d_zr12_kernel: any wrong block / tap / depth shift shows up as inequality"""
    from fast_nnunet_amd import capi
    n, c, dims = 32, 32, (20, 12, 12)
    base = (torch.arange(dims[0] * dims[1] * dims[2]).reshape(dims) * 7 % 23).float()
    x = torch.stack([torch.stack([base + ch + 3 * i for ch in range(c)]) for i in range(n)])
    w = torch.zeros(c, c, 3, 3, 3)
    for co in range(c):
        w[co, (co * 5 + 3) % c, co % 3, (co // 3) % 3, (co + 1) % 3] = 1.0
        w[co, (co * 3 + 1) % c, (co + 2) % 3, (co + 1) % 3, co % 3] += 2.0
    y = capi.op_conv3d(x.numpy(), w.numpy(), None, (3, 3, 3), (1, 1, 1))
    assert np.array_equal(y, F.conv3d(x, w, None, 1, 1).numpy())


# ----------------------------------------------------------------------------------------------- chunk-major layout
def test_conv_ops_with_chunk_major_tensors(monkeypatch):
    """Tensors of more than 16 channels stored [C / 16][voxels][16] (fnn_device.h, SrcDesc: the engine's layout between
    conv kernels) on both sides of every conv kernel family - generic, LDS-pipelined, persistent, ZR (8 x 8 x 8,
    whole-plane, depth-shift strided and its persistent form), stride-2 grouped, two sources - and of the transposed
    conv: the same cases as above, repacked by the op wrapper (FNN_OP_CHUNK_MAJOR=1)."""
    monkeypatch.setenv('FNN_OP_CHUNK_MAJOR', '1')
    for case in (CONV_CASES[1], CONV_CASES[5], CONV_CASES[6], CONV_CASES[7], CONV_CASES[8], CONV_CASES[11], CONV_CASES[12], CONV_CASES[15]):
        test_conv3d_identity_input(*case)
    test_conv3d_with_fused_instancenorm_lrelu_on_load(*CONV_CASES[5])
    test_conv3d_zr_variants(*ZR_CASES[-1])
    test_conv3d_zr_two_sources()
    test_conv3d_zr_whole_plane_tiles(16, 48, 64, (21, 11, 9), monkeypatch)
    test_conv3d_stride2_grouped_kernel(64, 128, 8, (45, 61, 58))
    test_conv3d_persistent_depth_shift_kernel_with_two_cout_groups()
    for case in TCONV_CASES[:3] + TCONV_CASES[4:]:
        test_conv_transpose3d(*case)


def test_gather_quotient_matches_ieee_division_on_every_fp16_pair():
    """The seg-head gather divides by the weight sum through one shared reciprocal per 16 voxels
    (gather.hip: quot_rcp / quot_fast) instead of IEEE division per value (predict_from_raw_data.py:619:
    predicted_logits /= n_predictions on half tensors).  Exhaustive: all 2^16 sums x all 2^15 non-negative
    weight sums - the fp16 bits must be those of fp32 division rounded to fp16 wherever the fast route is taken."""
    from fast_nnunet_amd import capi
    diff, fast, example = capi.op_quotient_check()
    assert diff == 0, f'{diff} pairs differ, e.g. a = {example[0]:#06x}, b = {example[1]:#06x}'
    # finite a, finite b > 0 whose quotient stays below fp16's overflow boundary: ~1.78e9 of the 2^31 pairs
    assert 1.7e9 < fast < 1.85e9


def _random_layer(seed):
    """A conv layer of the kind an nnU-Net plans file can ask for, drawn from a seed: the shapes nobody tuned a launch rule on."""
    rs = np.random.RandomState(1000 + seed)
    k = [(3, 3, 3), (3, 3, 3), (3, 3, 3), (1, 3, 3), (1, 1, 1), (3, 1, 3)][rs.randint(6)]
    stride = [(1, 1, 1), (1, 1, 1), (1, 1, 1), (2, 2, 2), (1, 2, 2), (2, 1, 1)][rs.randint(6)]
    cin = int(rs.choice([8, 16, 21, 32, 48, 64, 96, 128, 160]))
    two = stride == (1, 1, 1) and k == (3, 3, 3) and cin % 16 == 0 and cin <= 64 and rs.rand() < 0.4
    cout = int(rs.choice([10, 16, 24, 32, 48, 64, 96, 128, 160, 320]))
    n = int(rs.choice([1, 2, 3, 8, 16, 32]))
    dims = [int(rs.randint(2, 41)), int(rs.randint(3, 35)), int(rs.randint(3, 35))]
    macs = lambda: n * dims[0] * dims[1] * dims[2] * k[0] * k[1] * k[2] * cin * (2 if two else 1) * cout
    while macs() > 1.5e10:                                   # what torch's CPU conv does in about a second
        i = int(np.argmax(dims))
        if dims[i] > 6:
            dims[i] = dims[i] * 2 // 3
        elif n > 1:
            n = max(1, n // 2)
        else:
            cout = max(16, cout // 2)
    return n, cin, two, cout, tuple(dims), k, stride, bool(rs.rand() < 0.5)


@pytest.mark.parametrize('seed', range(48))
def test_conv3d_random_layers_through_the_launch_rules(seed):
    """Whatever kernel the launch rules pick for a layer nobody timed (odd planes, ragged tiles, channel padding, one or two
    sources, statistics, InstanceNorm + LeakyReLU on load), the output and the statistics are torch's."""
    from fast_nnunet_amd import capi
    n, cin, two, cout, dims, k, stride, norm = _random_layer(seed)
    g = torch.Generator().manual_seed(seed)
    x = _h(torch.randn(n, cin, *dims, generator=g) * 2 + 0.5)
    x2 = _h(torch.randn(n, cin, *dims, generator=g) - 0.25) if two else None
    ctot = cin * (2 if two else 1)
    w = _h(torch.randn(cout, ctot, *k, generator=g) / (ctot * k[0] * k[1] * k[2]) ** 0.5)
    b = torch.randn(cout, generator=g)
    pad = [(i - 1) // 2 for i in k]
    what = f'seed {seed}: n {n}, {cin}{"+" + str(cin) if two else ""} -> {cout}, {dims}, k {k}, stride {stride}, norm {norm}'
    if norm:
        gamma, beta = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.1
        kw = dict(x2=x2.numpy(), gamma2=gamma.numpy(), beta2=beta.numpy(), slope2=0.01) if two else \
            dict(gamma=gamma.numpy(), beta=beta.numpy(), slope=0.01)
        y = capi.op_conv3d(x.numpy(), w.numpy(), b.numpy(), k, stride, **kw)
        src = x2 if two else x
        xn = _h(F.leaky_relu(F.instance_norm(src, weight=gamma, bias=beta, eps=1e-5), 0.01))
        ref = F.conv3d(torch.cat((x, xn), 1) if two else xn, w, b, stride, pad)
        print(what, '->', capi.op_last_kernels())
        assert np.abs(y - ref.numpy()).max() <= 6e-3 * max(1.0, float(ref.abs().max())), what
    else:
        y, stats = capi.op_conv3d(x.numpy(), w.numpy(), b.numpy(), k, stride, x2=None if x2 is None else x2.numpy(),
                                  want_stats=True)
        print(what, '->', capi.op_last_kernels())
        _check(y, F.conv3d(torch.cat((x, x2), 1) if two else x, w, b, stride, pad), what)
        y64 = y.astype(np.float64)
        assert np.allclose(stats[..., 0], y64.sum((2, 3, 4)), rtol=1e-6, atol=1e-3), what
        assert np.allclose(stats[..., 1], (y64 ** 2).sum((2, 3, 4)), rtol=1e-6, atol=1e-3), what


def _random_big_layer(seed):
    """A layer of a network's size at a batch of patches: the (cin, cout, kernel, stride) pairs U-Nets are made of, on planes and
    depths drawn from a seed - enough workgroups for the depth-shift, whole-plane, stride-2, walking and row-streaming kernels."""
    rs = np.random.RandomState(5000 + seed)
    cin, cin2, cout, k, stride = [
        (16, 0, 16, (3, 3, 3), (1, 1, 1)), (16, 0, 16, (1, 3, 3), (1, 1, 1)), (16, 16, 16, (1, 3, 3), (1, 1, 1)),
        (16, 0, 32, (3, 3, 3), (1, 2, 2)), (16, 0, 32, (3, 3, 3), (2, 2, 2)), (32, 0, 32, (3, 3, 3), (1, 1, 1)),
        (32, 32, 32, (3, 3, 3), (1, 1, 1)), (32, 0, 64, (3, 3, 3), (2, 2, 2)), (64, 0, 64, (3, 3, 3), (1, 1, 1)),
        (64, 64, 64, (3, 3, 3), (1, 1, 1)), (64, 0, 128, (3, 3, 3), (2, 2, 2)), (128, 0, 128, (3, 3, 3), (1, 1, 1)),
        (128, 128, 128, (3, 3, 3), (1, 1, 1)), (128, 0, 160, (3, 3, 3), (2, 2, 2)), (160, 0, 160, (3, 3, 3), (1, 1, 1)),
        (48, 0, 96, (3, 3, 3), (1, 1, 1)), (32, 0, 48, (3, 3, 3), (2, 2, 2)), (64, 0, 320, (3, 3, 3), (1, 1, 1)),
    ][seed % 18]
    n = int(rs.choice([8, 16, 32]))
    small = cin >= 128
    dims = [int(rs.randint(6, 25 if small else 65)), int(rs.randint(5, 17 if small else 65)), int(rs.randint(5, 17 if small else 65))]
    macs = lambda: n * dims[0] * dims[1] * dims[2] * k[0] * k[1] * k[2] * (cin + cin2) * cout / (stride[0] * stride[1] * stride[2])
    byts = lambda: n * dims[0] * dims[1] * dims[2] * (cin + cin2 + cout) * 4
    while macs() > 3e11 or byts() > 3e9:
        i = int(np.argmax(dims))
        dims[i] = dims[i] * 3 // 4
    return n, cin, cin2, cout, tuple(dims), k, stride, bool(rs.rand() < 0.6)


@pytest.mark.parametrize('seed', range(36))
def test_conv3d_random_network_sized_layers_through_the_launch_rules(seed):
    """The same for layers large enough to reach the kernels a network runs on (reference: torch's fp32 conv on the GPU, fp16-rounded
    operands): random depths and planes - ragged tiles, planes the whole-plane and six-row forms were not written for."""
    from fast_nnunet_amd import capi
    n, cin, cin2, cout, dims, k, stride, norm = _random_big_layer(seed)
    g = torch.Generator().manual_seed(100 + seed)
    x = _h(torch.randn(n, cin, *dims, generator=g) * 2 + 0.5)
    x2 = _h(torch.randn(n, cin2, *dims, generator=g) - 0.25) if cin2 else None
    ctot = cin + cin2
    w = _h(torch.randn(cout, ctot, *k, generator=g) / (ctot * k[0] * k[1] * k[2]) ** 0.5)
    b = torch.randn(cout, generator=g)
    pad = [(i - 1) // 2 for i in k]
    what = f'seed {seed}: n {n}, {cin}{"+" + str(cin2) if cin2 else ""} -> {cout}, {dims}, k {k}, stride {stride}, norm {norm}'
    dev = torch.device('cuda:0')
    conv = lambda t: F.conv3d(t.to(dev), w.to(dev), b.to(dev), stride, pad).cpu()
    if norm:
        gamma, beta = torch.rand(cin2 or cin, generator=g) + 0.5, torch.randn(cin2 or cin, generator=g) * 0.1
        kw = dict(x2=x2.numpy(), gamma2=gamma.numpy(), beta2=beta.numpy(), slope2=0.01) if cin2 else \
            dict(gamma=gamma.numpy(), beta=beta.numpy(), slope=0.01)
        y = capi.op_conv3d(x.numpy(), w.numpy(), b.numpy(), k, stride, **kw)
        print(what, '->', capi.op_last_kernels())
        src = x2 if cin2 else x
        xn = _h(F.leaky_relu(F.instance_norm(src.to(dev), weight=gamma.to(dev), bias=beta.to(dev), eps=1e-5), 0.01).cpu())
        ref = conv(torch.cat((x, xn), 1) if cin2 else xn)
        assert np.abs(y - ref.numpy()).max() <= 6e-3 * max(1.0, float(ref.abs().max())), what
    else:
        y, stats = capi.op_conv3d(x.numpy(), w.numpy(), b.numpy(), k, stride, x2=None if x2 is None else x2.numpy(),
                                  want_stats=True)
        print(what, '->', capi.op_last_kernels())
        _check(y, conv(torch.cat((x, x2), 1) if cin2 else x), what)
        y64 = y.astype(np.float64)
        assert np.allclose(stats[..., 0], y64.sum((2, 3, 4)), rtol=1e-6, atol=2e-3), what
        assert np.allclose(stats[..., 1], (y64 ** 2).sum((2, 3, 4)), rtol=1e-6, atol=2e-3), what


def _random_tconv(seed):
    rs = np.random.RandomState(2000 + seed)
    stride = [(2, 2, 2), (2, 2, 2), (1, 2, 2), (2, 1, 1)][rs.randint(4)]
    cin = int(rs.choice([16, 21, 32, 48, 64, 96, 128, 160, 320]))
    cout = int(rs.choice([8, 10, 16, 32, 48, 64, 128, 160]))
    n = int(rs.choice([1, 2, 5, 16, 32]))
    dims = [int(rs.randint(2, 33)), int(rs.randint(2, 33)), int(rs.randint(2, 33))]
    while n * dims[0] * dims[1] * dims[2] * cin * cout * 8 > 2e10 or n * dims[0] * dims[1] * dims[2] * cout * 8 * 4 > 2e9:
        i = int(np.argmax(dims))
        if dims[i] > 3:
            dims[i] = dims[i] * 2 // 3
        else:
            n = max(1, n // 2)
    return n, cin, cout, tuple(dims), stride


@pytest.mark.parametrize('seed', range(20))
def test_conv_transpose3d_random_layers(seed):
    """Transposed convs (kernel = stride, as in every nnU-Net decoder) on shapes drawn from a seed: both launch forms, ragged rows."""
    from fast_nnunet_amd import capi
    n, cin, cout, dims, stride = _random_tconv(seed)
    g = torch.Generator().manual_seed(seed)
    x = _h(torch.randn(n, cin, *dims, generator=g) * 2 + 0.5)
    gamma, beta = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.1
    w = _h(torch.randn(cin, cout, *stride, generator=g) / cin ** 0.5)
    b = torch.randn(cout, generator=g)
    what = f'seed {seed}: n {n}, {cin} -> {cout}, {dims}, stride {stride}'
    y = capi.op_conv_transpose3d(x.numpy(), w.numpy(), b.numpy(), stride, gamma=gamma.numpy(), beta=beta.numpy(), slope=0.01)
    print(what, '->', capi.op_last_kernels())
    xn = _h(F.leaky_relu(F.instance_norm(x, weight=gamma, bias=beta, eps=1e-5), 0.01))
    ref = F.conv_transpose3d(xn, w, b, stride)
    assert np.abs(y - ref.numpy()).max() <= 6e-3 * max(1.0, float(ref.abs().max())), what
    y2 = capi.op_conv_transpose3d(x.numpy(), w.numpy(), b.numpy(), stride)
    _check(y2, F.conv_transpose3d(x, w, b, stride), what)


# ---------------------------------------------------------------------------- conv2d_zp_kernel: (1, 3, 3) layers of >= 32 channels
# n, cin, cin2 (second source), cout, dims, expected variant <rows per wave, waves along the columns>
ZP_CASES = [
    (2, 32, 0, 32, (1, 64, 64), '8,4'),          # 2-D plane, whole 8 x 64 tiles
    (3, 64, 0, 32, (1, 37, 83), '8,4'),          # ragged rows and columns, two chunks
    (2, 32, 32, 64, (2, 24, 40), '8,4'),         # two sources, depth 2, two cout pairs
    (2, 48, 16, 32, (3, 21, 30), '8,2'),         # sources of 48 and 16 channels: half-empty last chunks, 16 x 32 tiles
    (4, 128, 0, 64, (1, 32, 32), '8,2'),
    (4, 96, 0, 96, (1, 16, 16), '4,1'),          # 16 x 16 tiles, three cout pairs
    (8, 160, 160, 64, (1, 8, 8), '4,1'),         # plane smaller than the tile
    (6, 64, 0, 64, (2, 4, 4), '4,1'),
    (2, 16, 0, 32, (2, 33, 70), '8,4,2,half'),   # a single 16-channel source (a stem on the conv kernels): the half image
    (1, 4, 0, 32, (1, 40, 48), '8,4,2,half'),    # channel padding 4 -> 16
    # one cout block per workgroup (16 output channels - the full-resolution level of a `2d` r = 2 student - or an odd block count)
    (2, 16, 0, 16, (1, 64, 96), '8,4,1,half'),
    (3, 16, 16, 16, (1, 37, 83), '8,4,1,half'),  # decoder conv of that level: two 16-channel sources = two half chunks, ragged tiles
    (2, 1, 0, 16, (2, 24, 40), '8,4,1,half'),    # a 1-channel stem on the conv kernels, depth 2
    (3, 16, 0, 16, (1, 20, 30), '8,2,1,half'),
    (3, 16, 0, 32, (1, 12, 14), '4,1,2,half'),
    (2, 16, 0, 32, (1, 20, 30), '8,2,2,half'),
    (3, 16, 0, 16, (2, 10, 12), '4,1,1,half'),
    (2, 32, 0, 16, (1, 40, 64), '8,4,1'),        # one cout block, a whole 32-channel chunk
    (4, 64, 0, 48, (1, 32, 20), '8,2,1'),        # three cout blocks
    (4, 32, 0, 16, (3, 9, 12), '4,1,1'),
]


@pytest.mark.parametrize('n,cin,cin2,cout,dims,variant', ZP_CASES, ids=lambda v: str(v).replace(' ', ''))
def test_conv2d_plane_kernel(n, cin, cin2, cout, dims, variant):
    """Round 6: conv2d_zp_kernel against torch's fp32 conv on the same fp16-rounded operands - identity input (statistics
    checked), then InstanceNorm + LeakyReLU on load; two sources; the variant the launch rule must pick."""
    from fast_nnunet_amd import capi
    k, st = (1, 3, 3), (1, 1, 1)
    g = torch.Generator().manual_seed(31 + cin + cin2 + cout + dims[1])
    x = _h(torch.randn(n, cin, *dims, generator=g) * 2 + 0.5)
    x2 = _h(torch.randn(n, cin2, *dims, generator=g) * 1.5 - 0.3) if cin2 else None
    w = _h(torch.randn(cout, cin + cin2, *k, generator=g) / ((cin + cin2) * 9) ** 0.5)
    b = torch.randn(cout, generator=g)
    y, stats = capi.op_conv3d(x.numpy(), w.numpy(), b.numpy(), k, st, x2=None if x2 is None else x2.numpy(), want_stats=True)
    assert capi.op_last_kernels() == [f'conv2d_zp_kernel<{variant}>'], capi.op_last_kernels()
    cat = x if x2 is None else torch.cat((x, x2), 1)
    _check(y, F.conv3d(cat, w, b, 1, (0, 1, 1)), 'conv2d zp')
    y64 = y.astype(np.float64)
    assert np.allclose(stats[..., 0], y64.sum((2, 3, 4)), rtol=1e-6, atol=1e-3)
    assert np.allclose(stats[..., 1], (y64 ** 2).sum((2, 3, 4)), rtol=1e-6, atol=1e-3)
    gamma, beta = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.1
    kw = {}
    xn = _h(F.leaky_relu(F.instance_norm(x, weight=gamma, bias=beta, eps=1e-5), 0.01))
    if x2 is not None:
        gamma2, beta2 = torch.rand(cin2, generator=g) + 0.5, torch.randn(cin2, generator=g) * 0.1
        kw = dict(x2=x2.numpy(), gamma2=gamma2.numpy(), beta2=beta2.numpy(), slope2=0.01)
        xn = torch.cat((xn, _h(F.leaky_relu(F.instance_norm(x2, weight=gamma2, bias=beta2, eps=1e-5), 0.01))), 1)
    y = capi.op_conv3d(x.numpy(), w.numpy(), b.numpy(), k, st, gamma=gamma.numpy(), beta=beta.numpy(), slope=0.01, **kw)
    ref = F.conv3d(xn, w, b, 1, (0, 1, 1))
    assert np.abs(y - ref.numpy()).max() <= 6e-3 * max(1.0, float(ref.abs().max()))


def test_conv2d_plane_kernel_operand_map_with_exact_integers():
    """Small integers (every product and sum exact in fp16 / fp32): a wrong tap, channel or output-channel permutation cannot
    hide behind a tolerance."""
    from fast_nnunet_amd import capi
    g = torch.Generator().manual_seed(5)
    n, cin, cout, dims = 2, 64, 64, (2, 19, 70)
    x = torch.randint(-3, 4, (n, cin, *dims), generator=g).float()
    w = torch.randint(-2, 3, (cout, cin, 1, 3, 3), generator=g).float()
    b = torch.randint(-5, 6, (cout,), generator=g).float()
    y = capi.op_conv3d(x.numpy(), w.numpy(), b.numpy(), (1, 3, 3), (1, 1, 1))
    assert capi.op_last_kernels() == ['conv2d_zp_kernel<8,4>']
    ref = F.conv3d(x, w, b, 1, (0, 1, 1))
    assert float(ref.abs().max()) < 2048
    assert np.array_equal(y, ref.numpy())


# n, cin, cin2, cout, input dims, expected variant <cout blocks per workgroup, waves along the columns>
ZPS_CASES = [
    (2, 32, 0, 64, (1, 64, 64), '4,2'),          # 2-D down-sampling conv: 32 x 32 outputs, 64 output channels in one group
    (3, 32, 0, 32, (2, 37, 83), '2,2'),          # odd input sizes (ragged rows / columns), one cout pair
    (2, 64, 0, 128, (1, 128, 96), '4,2'),        # several tiles per plane, two chunks, two cout groups
    (2, 48, 16, 96, (2, 30, 44), '2,2'),         # two sources with half-empty last chunks, three cout pairs
    (4, 128, 0, 256, (1, 32, 32), '4,1'),        # 16 x 16 outputs
    (8, 96, 0, 64, (1, 9, 7), '4,1'),            # a plane smaller than the tile, odd sizes
    (2, 16, 0, 32, (3, 40, 66), '2,2,half'),     # a single 16-channel source: the half image, two workgroups per CU
    (3, 9, 0, 32, (1, 20, 18), '2,1,half'),      # ... 16 x 16 tiles, channel padding 9 -> 16
    (2, 16, 0, 64, (2, 34, 40), '4,2'),          # 16 channels into 64: the whole image with four cout blocks
    (3, 64, 0, 96, (2, 20, 18), '2,1'),          # 16 x 16 tiles, cout blocks in pairs (96 = 3 x 32)
]


@pytest.mark.parametrize('n,cin,cin2,cout,dims,variant', ZPS_CASES, ids=lambda v: str(v).replace(' ', ''))
def test_conv2d_plane_kernel_strided(n, cin, cin2, cout, dims, variant):
    """conv2d_zps_kernel ((1, 3, 3) taps, stride (1, 2, 2)) against torch's fp32 conv on the same fp16-rounded operands."""
    from fast_nnunet_amd import capi
    k, st = (1, 3, 3), (1, 2, 2)
    g = torch.Generator().manual_seed(57 + cin + cin2 + cout + dims[1])
    x = _h(torch.randn(n, cin, *dims, generator=g) * 2 + 0.5)
    x2 = _h(torch.randn(n, cin2, *dims, generator=g) * 1.5 - 0.3) if cin2 else None
    w = _h(torch.randn(cout, cin + cin2, *k, generator=g) / ((cin + cin2) * 9) ** 0.5)
    b = torch.randn(cout, generator=g)
    y, stats = capi.op_conv3d(x.numpy(), w.numpy(), b.numpy(), k, st, x2=None if x2 is None else x2.numpy(), want_stats=True)
    assert capi.op_last_kernels() == [f'conv2d_zps_kernel<{variant}>'], capi.op_last_kernels()
    cat = x if x2 is None else torch.cat((x, x2), 1)
    _check(y, F.conv3d(cat, w, b, st, (0, 1, 1)), 'conv2d zps')
    y64 = y.astype(np.float64)
    assert np.allclose(stats[..., 0], y64.sum((2, 3, 4)), rtol=1e-6, atol=1e-3)
    assert np.allclose(stats[..., 1], (y64 ** 2).sum((2, 3, 4)), rtol=1e-6, atol=1e-3)
    gamma, beta = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.1
    kw = {}
    xn = _h(F.leaky_relu(F.instance_norm(x, weight=gamma, bias=beta, eps=1e-5), 0.01))
    if x2 is not None:
        gamma2, beta2 = torch.rand(cin2, generator=g) + 0.5, torch.randn(cin2, generator=g) * 0.1
        kw = dict(x2=x2.numpy(), gamma2=gamma2.numpy(), beta2=beta2.numpy(), slope2=0.01)
        xn = torch.cat((xn, _h(F.leaky_relu(F.instance_norm(x2, weight=gamma2, bias=beta2, eps=1e-5), 0.01))), 1)
    y = capi.op_conv3d(x.numpy(), w.numpy(), b.numpy(), k, st, gamma=gamma.numpy(), beta=beta.numpy(), slope=0.01, **kw)
    ref = F.conv3d(xn, w, b, st, (0, 1, 1))
    assert np.abs(y - ref.numpy()).max() <= 6e-3 * max(1.0, float(ref.abs().max()))


def test_conv2d_plane_kernel_strided_operand_map_with_exact_integers():
    from fast_nnunet_amd import capi
    g = torch.Generator().manual_seed(6)
    n, cin, cout, dims = 2, 64, 128, (2, 21, 70)
    x = torch.randint(-3, 4, (n, cin, *dims), generator=g).float()
    w = torch.randint(-2, 3, (cout, cin, 1, 3, 3), generator=g).float()
    b = torch.randint(-5, 6, (cout,), generator=g).float()
    y = capi.op_conv3d(x.numpy(), w.numpy(), b.numpy(), (1, 3, 3), (1, 2, 2))
    assert capi.op_last_kernels() == ['conv2d_zps_kernel<4,2>']
    ref = F.conv3d(x, w, b, (1, 2, 2), (0, 1, 1))
    assert float(ref.abs().max()) < 2048
    assert np.array_equal(y, ref.numpy())
