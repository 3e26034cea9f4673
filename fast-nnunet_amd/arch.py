"""checkpoint / plans -> engine topology (``fnn_arch_desc``) + canonical weight blob.

The reference rebuilds the network object from ``plans.json`` through
``get_network_from_plans`` (utilities/get_network_from_plans.py:9-43) and, for a
distilled student, through ``nnUNetDistillationTrainer.build_network_architecture``
(training/nnUNetTrainer/variants/nnUNetDistillationTrainer.py:605-758, rule
``max(f // r, 8)`` at :678) - which its own predictor cannot call.
The engine does not need a module object, only shapes, so the topology is read
from the state dict itself: feature widths, kernel sizes, conv counts and (from
the transposed-conv kernels) the strides.  That is robust for any reduction
factor and is cross-checked against the plans when they are available.

State-dict key schema; aliases (`all_modules.*`,
`decoder.encoder.*`) and wrapper prefixes (`module.`, `_orig_mod.`,
`network.`) are canonicalised away.
"""
from __future__ import annotations

import re
from dataclasses import dataclass, field
from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import numpy as np

from . import capi

_PREFIXES = ('module.', '_orig_mod.', 'network.')


@dataclass
class ArchSpec:
    kind: int
    in_channels: int
    num_heads: int
    features: List[int]
    kernels: List[Tuple[int, int, int]]
    strides: List[Tuple[int, int, int]]
    n_conv_enc: List[int]
    n_conv_dec: List[int]
    patch: Tuple[int, int, int]
    eps: float = 1e-5
    slope: float = 0.01
    spatial_dims: int = 3                      # 2: `2d` configuration, run as patch (1, py, pz) with kernels (1, k, k)
    precision: int = 0                         # capi.FNN_PREC_*: operand format of the 3x3x3 stride-1 convs

    @property
    def n_stages(self) -> int:
        return len(self.features)

    def to_desc(self) -> capi.ArchDesc:
        if self.n_stages > capi.FNN_MAX_STAGES:
            raise NotImplementedError(f'{self.n_stages} stages > {capi.FNN_MAX_STAGES}')
        d = capi.ArchDesc()
        d.kind, d.n_stages, d.in_channels, d.num_heads = self.kind, self.n_stages, self.in_channels, self.num_heads
        for s in range(self.n_stages):
            d.features[s] = self.features[s]
            d.n_conv_enc[s] = self.n_conv_enc[s]
            for a in range(3):
                d.kernels[s][a] = self.kernels[s][a]
                d.strides[s][a] = self.strides[s][a]
        for s in range(self.n_stages - 1):
            d.n_conv_dec[s] = self.n_conv_dec[s]
        for a in range(3):
            d.patch[a] = int(self.patch[a])
        d.eps, d.slope = self.eps, self.slope
        d.spatial_dims = self.spatial_dims
        d.precision = self.precision
        return d


def canonical_state_dict(state_dict: Mapping[str, object]) -> Dict[str, np.ndarray]:
    """Strip wrapper prefixes, drop alias keys, convert to float32 numpy."""
    out = {}
    for k, v in state_dict.items():
        changed = True
        while changed:
            changed = False
            for p in _PREFIXES:
                if k.startswith(p):
                    k, changed = k[len(p):], True
        if '.all_modules.' in k or k.startswith('decoder.encoder.'):
            continue
        arr = v.detach().cpu().float().numpy() if hasattr(v, 'detach') else np.asarray(v, dtype=np.float32)
        out[k] = np.ascontiguousarray(arr, dtype=np.float32)
    return out


def _count(sd, pattern: str) -> int:
    rx = re.compile(pattern)
    idx = {int(m.group(1)) for k in sd for m in [rx.match(k)] if m}
    return (max(idx) + 1) if idx else 0


def spec_from_state_dict(state_dict: Mapping[str, object], patch: Sequence[int], eps: float = 1e-5,
                         slope: float = 0.01) -> ArchSpec:
    sd = canonical_state_dict(state_dict)
    resenc = any(k.startswith('encoder.stem.') for k in sd)
    n = _count(sd, r'encoder\.stages\.(\d+)\.')
    if n < 2:
        raise RuntimeError('state dict does not look like a PlainConvUNet / ResidualEncoderUNet (no encoder.stages.*)')
    first = (lambda s: f'encoder.stages.{s}.blocks.0.conv1.conv.weight') if resenc else \
        (lambda s: f'encoder.stages.{s}.0.convs.0.conv.weight')
    feats, kernels, n_enc = [], [], []
    nd = sd[first(0)].ndim - 2                                   # Conv3d weights are 5-D, Conv2d (`2d` configurations) 4-D
    if nd not in (2, 3):
        raise NotImplementedError('only Conv2d / Conv3d networks are supported')
    lift = (lambda t: (1, *t)) if nd == 2 else (lambda t: tuple(t))      # 2-D runs as depth-1 3-D
    for s in range(n):
        w = sd[first(s)]
        feats.append(int(w.shape[0]))
        kernels.append(lift(tuple(int(i) for i in w.shape[2:])))
        n_enc.append(_count(sd, rf'encoder\.stages\.{s}\.blocks\.(\d+)\.') if resenc
                     else _count(sd, rf'encoder\.stages\.{s}\.0\.convs\.(\d+)\.'))
    strides = [(1, 1, 1)] * n
    n_dec = []
    for d in range(n - 1):
        tw = sd[f'decoder.transpconvs.{d}.weight']
        strides[n - 1 - d] = lift(tuple(int(i) for i in tw.shape[2:]))
        n_dec.append(_count(sd, rf'decoder\.stages\.{d}\.convs\.(\d+)\.'))
    patch = tuple(int(i) for i in patch)
    if len(patch) != nd:
        raise RuntimeError(f'patch_size {patch} does not match the {nd}-D network of the checkpoint')
    w0 = sd['encoder.stem.convs.0.conv.weight'] if resenc else sd['encoder.stages.0.0.convs.0.conv.weight']
    if resenc and (int(w0.shape[0]) != feats[0] or lift(tuple(int(i) for i in w0.shape[2:])) != kernels[0]):
        raise NotImplementedError('stem with a width / kernel different from stage 0 is not supported')
    in_ch = int(w0.shape[1])
    heads = int(sd[f'decoder.seg_layers.{n - 2}.weight'].shape[0])
    return ArchSpec(capi.FNN_NET_RESENC if resenc else capi.FNN_NET_PLAIN, in_ch, heads, feats, kernels, strides,
                    n_enc, n_dec, lift(patch), eps, slope, nd)


def ops_from_plans(arch_kwargs: dict) -> Tuple[float, float]:
    """-> (eps, negative_slope) of the plans' architecture block, after checking that its operators are the ones
    the engine implements: affine InstanceNorm{2,3}d + LeakyReLU, no dropout (what the reference's planner writes,
    experiment_planning/experiment_planners/default_experiment_planner.py:288-300; a checkpoint trained with
    BatchNorm or ReLU would otherwise load and give silently wrong logits)."""
    def name(v):
        return v if isinstance(v, str) else getattr(v, '__name__', str(v))
    norm = arch_kwargs.get('norm_op')
    if norm is not None and 'InstanceNorm' not in name(norm):
        raise NotImplementedError(f'norm_op {name(norm)}: the engine implements InstanceNorm2d / InstanceNorm3d only')
    nkw = arch_kwargs.get('norm_op_kwargs') or {}
    if norm is not None and not nkw.get('affine', True):
        raise NotImplementedError('InstanceNorm without affine parameters is not supported')
    nonlin = arch_kwargs.get('nonlin')
    if nonlin is not None and 'LeakyReLU' not in name(nonlin):
        raise NotImplementedError(f'nonlin {name(nonlin)}: the engine implements LeakyReLU only')
    drop = arch_kwargs.get('dropout_op')
    if drop is not None and float((arch_kwargs.get('dropout_op_kwargs') or {}).get('p', 0.0)) != 0.0:
        pass                                                   # dropout is the identity at inference time
    slope = float((arch_kwargs.get('nonlin_kwargs') or {}).get('negative_slope', 0.01))
    if not 0.0 <= slope <= 1.0:
        raise NotImplementedError(f'LeakyReLU negative_slope {slope} outside [0, 1]')
    return float(nkw.get('eps', 1e-5)), slope


def check_against_plans(spec: ArchSpec, arch_kwargs: dict, reduction: Optional[int] = None):
    """Raise if the checkpoint disagrees with the plans' architecture block."""
    n = int(arch_kwargs['n_stages'])
    if n != spec.n_stages:
        raise RuntimeError(f'checkpoint has {spec.n_stages} stages, plans say {n}')
    nd = spec.spatial_dims
    lift = (lambda t: (1, *t)) if nd == 2 else (lambda t: tuple(t))
    ks = arch_kwargs['kernel_sizes']
    plan_k = [tuple(k) if not isinstance(k, int) else (k,) * nd for k in (ks if not isinstance(ks, int) else [ks] * n)]
    plan_k = [lift(tuple(int(i) for i in k)) for k in plan_k]
    if [tuple(k) for k in spec.kernels] != plan_k:
        raise RuntimeError(f'kernel sizes differ: checkpoint {spec.kernels} vs plans {plan_k}')
    plan_s = [lift(tuple(int(i) for i in (s if not isinstance(s, int) else (s,) * nd))) for s in arch_kwargs['strides']]
    if list(spec.strides) != plan_s:
        raise RuntimeError(f'strides differ: checkpoint {spec.strides} vs plans {plan_s}')
    if reduction is not None:
        want = [max(int(f) // reduction, 8) for f in arch_kwargs['features_per_stage']]
        if want != spec.features:
            raise RuntimeError(f'features {spec.features} do not match max(f // {reduction}, 8) = {want}')


def weight_blob(spec: ArchSpec, state_dict: Mapping[str, object]) -> np.ndarray:
    """Flatten the parameters in the order ``fnn_load_weights`` documents."""
    full = canonical_state_dict(state_dict)
    used = set()

    class _Tracked(dict):                                      # remembers which keys the blob consumed
        def __getitem__(self, k):
            used.add(k)
            return dict.__getitem__(self, k)

        def get(self, k, default=None):
            if k in self:
                used.add(k)
            return dict.get(self, k, default)

    sd = _Tracked(full)
    parts: List[np.ndarray] = []

    def conv_block(prefix):
        w = sd[prefix + '.conv.weight']
        parts.append(w.reshape(-1))
        b = sd.get(prefix + '.conv.bias')
        parts.append(b if b is not None else np.zeros(w.shape[0], np.float32))
        parts.append(sd[prefix + '.norm.weight'])
        parts.append(sd[prefix + '.norm.bias'])

    n = spec.n_stages
    if spec.kind == capi.FNN_NET_RESENC:
        conv_block('encoder.stem.convs.0')
        cin = spec.features[0]
        for s in range(n):
            for b in range(spec.n_conv_enc[s]):
                pre = f'encoder.stages.{s}.blocks.{b}'
                conv_block(pre + '.conv1')
                conv_block(pre + '.conv2')
                if cin != spec.features[s]:
                    # skip = Sequential([AvgPool3d if strided], conv1x1 + norm): the projection is the last entry
                    idx = _count(sd, rf'encoder\.stages\.{s}\.blocks\.{b}\.skip\.(\d+)\.') - 1
                    parts.append(sd[f'{pre}.skip.{idx}.conv.weight'].reshape(-1))
                    parts.append(sd[f'{pre}.skip.{idx}.norm.weight'])
                    parts.append(sd[f'{pre}.skip.{idx}.norm.bias'])
                cin = spec.features[s]
    else:
        for s in range(n):
            for i in range(spec.n_conv_enc[s]):
                conv_block(f'encoder.stages.{s}.0.convs.{i}')
    for d in range(n - 1):
        tw = sd[f'decoder.transpconvs.{d}.weight']
        parts.append(tw.reshape(-1))
        tb = sd.get(f'decoder.transpconvs.{d}.bias')
        parts.append(tb if tb is not None else np.zeros(tw.shape[1], np.float32))
        for i in range(spec.n_conv_dec[d]):
            conv_block(f'decoder.stages.{d}.convs.{i}')
    parts.append(sd[f'decoder.seg_layers.{n - 2}.weight'].reshape(-1))
    parts.append(sd[f'decoder.seg_layers.{n - 2}.bias'])
    # Anything left over that is not a deep-supervision head (decoder.seg_layers.<d>, unused at inference:
    # the reference switches deep supervision off, predict_from_raw_data.py:118) is a module the engine does not
    # implement - e.g. BatchNorm's running_mean / running_var - and would be dropped silently.
    left = [k for k in full if k not in used and not k.startswith('decoder.seg_layers.')]
    if left:
        raise NotImplementedError(f'state dict has parameters / buffers the engine does not implement: {sorted(left)[:6]}'
                                  f'{" ..." if len(left) > 6 else ""}')
    return np.concatenate([p.astype(np.float32, copy=False).reshape(-1) for p in parts])


def spec_from_plans(arch_class_name: str, arch_kwargs: dict, in_channels: int, num_heads: int, patch: Sequence[int],
                    reduction: int = 1) -> ArchSpec:
    """Topology from the plans alone (teacher: reduction 1; student: features ``max(f // r, 8)``)."""
    resenc = 'Residual' in arch_class_name or 'ResEnc' in arch_class_name
    n = int(arch_kwargs['n_stages'])
    feats = [max(int(f) // reduction, 8) if reduction != 1 else int(f) for f in arch_kwargs['features_per_stage']]
    nd = len(patch)
    lift = (lambda t: (1, *t)) if nd == 2 else (lambda t: tuple(t))      # `2d` configurations run as depth-1 3-D
    ks = arch_kwargs['kernel_sizes']
    kernels = [lift((int(ks),) * nd)] * n if isinstance(ks, int) else [
        lift(tuple(int(i) for i in (k[0] if isinstance(k[0], (list, tuple)) else k))) for k in ks]
    strides = [lift(tuple(int(i) for i in (s if not isinstance(s, int) else (s,) * nd))) for s in arch_kwargs['strides']]
    enc = arch_kwargs['n_blocks_per_stage' if resenc else 'n_conv_per_stage']
    enc = [int(enc)] * n if isinstance(enc, int) else [int(i) for i in enc]
    dec = arch_kwargs['n_conv_per_stage_decoder']
    dec = [int(dec)] * (n - 1) if isinstance(dec, int) else [int(i) for i in dec]
    eps = float((arch_kwargs.get('norm_op_kwargs') or {}).get('eps', 1e-5))
    return ArchSpec(capi.FNN_NET_RESENC if resenc else capi.FNN_NET_PLAIN, in_channels, num_heads, feats, kernels,
                    strides, enc, dec, lift(tuple(int(i) for i in patch)), eps, 0.01, nd)
