"""PyTorch-CPU restatement of the reference's networks (test oracle).

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.  "parity unpinned": the
network bodies the reference runs live in the third-party package
``dynamic_network_architectures`` (imported at
training/nnUNetTrainer/variants/nnUNetDistillationTrainer.py:67-69 and
utilities/get_network_from_plans.py:17-38; un-vendored, un-pinned,
distillation/setup.py:7-10).  This file restates that package's published
module structure from the constructor arguments at the reference's call sites
(nnUNetDistillationTrainer.py:141-173, 248-266) so that parameter names match
a real checkpoint:

    encoder.stages.{s}.0.convs.{i}.conv.{weight,bias} / .norm.{weight,bias}
    decoder.transpconvs.{d}.{weight,bias}
    decoder.stages.{d}.convs.{i}.conv / .norm
    decoder.seg_layers.{d}.{weight,bias}

Arithmetic is torch's own CPU kernels in fp32: Conv3d(padding=(k-1)//2) ->
InstanceNorm3d(eps, affine, no running stats) -> LeakyReLU(0.01);
ConvTranspose3d(kernel=stride); torch.cat((upsampled, skip), 1).  A spec whose
kernels have two entries builds the 2-D network of a `2d` configuration
(Conv2d / InstanceNorm2d / ConvTranspose2d, get_network_from_plans.py:17-38
with conv_op = torch.nn.modules.conv.Conv2d).
"""
from __future__ import annotations

from typing import Dict, List

import torch
from torch import nn

from .topology import UNetSpec


_CONV = {2: nn.Conv2d, 3: nn.Conv3d}
_NORM = {2: nn.InstanceNorm2d, 3: nn.InstanceNorm3d}
_TCONV = {2: nn.ConvTranspose2d, 3: nn.ConvTranspose3d}
_POOL = {2: nn.AvgPool2d, 3: nn.AvgPool3d}


class ConvNormAct(nn.Module):
    def __init__(self, cin, cout, k, stride, bias, eps, slope, act=True):
        super().__init__()
        nd = len(k)
        self.conv = _CONV[nd](cin, cout, tuple(k), tuple(stride), padding=[(i - 1) // 2 for i in k], bias=bias)
        self.norm = _NORM[nd](cout, eps=eps, affine=True)
        self.slope = slope
        self.act = act

    def forward(self, x):
        x = self.norm(self.conv(x))
        return nn.functional.leaky_relu(x, self.slope) if self.act else x


class ConvStack(nn.Module):
    """`n` conv blocks, the first one strided (StackedConvBlocks)."""

    def __init__(self, n, cin, cout, k, stride, bias, eps, slope):
        super().__init__()
        self.convs = nn.Sequential(*[
            ConvNormAct(cin if i == 0 else cout, cout, k, stride if i == 0 else (1,) * len(k), bias, eps, slope)
            for i in range(n)])

    def forward(self, x):
        return self.convs(x)


class PlainEncoder(nn.Module):
    def __init__(self, spec: UNetSpec):
        super().__init__()
        stages, cin = [], spec.in_channels
        for s in range(spec.n_stages):
            stages.append(nn.Sequential(ConvStack(spec.n_conv_enc[s], cin, spec.features[s], spec.kernels[s],
                                                  spec.strides[s], spec.conv_bias, spec.eps, spec.slope)))
            cin = spec.features[s]
        self.stages = nn.Sequential(*stages)

    def forward(self, x):
        skips = []
        for st in self.stages:
            x = st(x)
            skips.append(x)
        return skips


class ResBlock(nn.Module):
    """BasicBlockD: conv-norm-act, conv-norm, + projected skip, act."""

    def __init__(self, cin, cout, k, stride, bias, eps, slope):
        super().__init__()
        self.conv1 = ConvNormAct(cin, cout, k, stride, bias, eps, slope, act=True)
        self.conv2 = ConvNormAct(cout, cout, k, (1,) * len(k), bias, eps, slope, act=False)
        self.slope = slope
        ops = []
        if any(s != 1 for s in stride):
            ops.append(_POOL[len(k)](tuple(stride), tuple(stride)))
        if cin != cout:
            ops.append(ConvNormAct(cin, cout, (1,) * len(k), (1,) * len(k), False, eps, slope, act=False))
        self.skip = nn.Sequential(*ops) if ops else nn.Identity()

    def forward(self, x):
        return nn.functional.leaky_relu(self.conv2(self.conv1(x)) + self.skip(x), self.slope)


class ResBlockStack(nn.Module):
    def __init__(self, n, cin, cout, k, stride, bias, eps, slope):
        super().__init__()
        self.blocks = nn.Sequential(*[
            ResBlock(cin if i == 0 else cout, cout, k, stride if i == 0 else (1,) * len(k), bias, eps, slope)
            for i in range(n)])

    def forward(self, x):
        return self.blocks(x)


class ResEncoder(nn.Module):
    def __init__(self, spec: UNetSpec):
        super().__init__()
        f0 = spec.features[0]
        self.stem = ConvStack(1, spec.in_channels, f0, spec.kernels[0], (1,) * len(spec.kernels[0]), spec.conv_bias, spec.eps,
                              spec.slope)
        stages, cin = [], f0
        for s in range(spec.n_stages):
            stages.append(ResBlockStack(spec.n_conv_enc[s], cin, spec.features[s], spec.kernels[s],
                                        spec.strides[s], spec.conv_bias, spec.eps, spec.slope))
            cin = spec.features[s]
        self.stages = nn.Sequential(*stages)

    def forward(self, x):
        x = self.stem(x)
        skips = []
        for st in self.stages:
            x = st(x)
            skips.append(x)
        return skips


class Decoder(nn.Module):
    def __init__(self, spec: UNetSpec):
        super().__init__()
        n = spec.n_stages
        tconvs, stages, segs = [], [], []
        for d in range(n - 1):
            below, skip = spec.features[-(d + 1)], spec.features[-(d + 2)]
            st = spec.strides[-(d + 1)]
            nd = len(st)
            tconvs.append(_TCONV[nd](below, skip, tuple(st), tuple(st), bias=spec.conv_bias))
            stages.append(ConvStack(spec.n_conv_dec[d], 2 * skip, skip, spec.kernels[-(d + 2)], (1,) * nd,
                                    spec.conv_bias, spec.eps, spec.slope))
            segs.append(_CONV[nd](skip, spec.num_heads, 1, 1, 0, bias=True))
        self.transpconvs = nn.ModuleList(tconvs)
        self.stages = nn.ModuleList(stages)
        self.seg_layers = nn.ModuleList(segs)
        self.deep_supervision = spec.deep_supervision

    def forward(self, skips):
        x = skips[-1]
        outs = []
        for d in range(len(self.stages)):
            x = self.transpconvs[d](x)
            x = torch.cat((x, skips[-(d + 2)]), 1)
            x = self.stages[d](x)
            if self.deep_supervision or d == len(self.stages) - 1:
                outs.append(self.seg_layers[d](x))
        outs = outs[::-1]
        return outs if self.deep_supervision else outs[0]


class OracleUNet(nn.Module):
    """PlainConvUNet / ResidualEncoderUNet restatement (root attrs encoder, decoder)."""

    def __init__(self, spec: UNetSpec):
        super().__init__()
        self.spec = spec
        self.encoder = ResEncoder(spec) if spec.kind == 'resenc' else PlainEncoder(spec)
        self.decoder = Decoder(spec)

    def forward(self, x):
        return self.decoder(self.encoder(x))


def synthetic_state_dict(spec: UNetSpec, seed: int = 1234, affine_jitter: bool = True) -> Dict[str, torch.Tensor]:
    """Seeded random weights in the checkpoint key schema.

    Conv / transposed-conv weights ``kaiming_normal_(a=0.01)`` as the
    reference's ``InitWeights_He`` (utilities/network_initialization.py:4-12),
    but with small random biases and InstanceNorm gamma~U(0.5,1.5),
    beta~N(0,0.1) so that every parameter is exercised.
    """
    g = torch.Generator().manual_seed(seed)
    net = OracleUNet(spec)
    sd = net.state_dict()
    out = {}
    for k, v in sd.items():
        if v.ndim >= 4:
            w = torch.empty_like(v)
            fan_in = v.shape[1] * v[0, 0].numel()
            if 'transpconvs' in k:
                fan_in = v.shape[0] * v[0, 0].numel()
            std = (2.0 / (1 + 0.01 ** 2)) ** 0.5 / fan_in ** 0.5
            out[k] = w.normal_(0, std, generator=g)
        elif '.norm.weight' in k:
            out[k] = (torch.rand(v.shape, generator=g) + 0.5) if affine_jitter else torch.ones_like(v)
        elif '.norm.bias' in k:
            out[k] = (torch.randn(v.shape, generator=g) * 0.1) if affine_jitter else torch.zeros_like(v)
        else:
            out[k] = (torch.randn(v.shape, generator=g) * 0.05) if affine_jitter else torch.zeros_like(v)
    return out


def build(spec: UNetSpec, state_dict=None, seed: int = 1234) -> OracleUNet:
    net = OracleUNet(spec)
    net.load_state_dict(state_dict if state_dict is not None else synthetic_state_dict(spec, seed))
    return net.eval()
