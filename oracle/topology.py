"""plans.json -> network description (test oracle; see ``oracle/__init__.py``).

Restates, independently of the product's own reader in
``fast-nnunet_amd/arch.py``:

* the architecture block of a configuration
  (utilities/plans_handling/plans_handler.py:31-156, schema from
  experiment_planning/experiment_planners/default_experiment_planner.py:279-298);
* the distilled-student reduction rule
  (training/nnUNetTrainer/variants/nnUNetDistillationTrainer.py:678,685-708);
* the pooling / kernel planner used to derive benchmark topologies
  (experiment_planning/experiment_planners/network_topology.py:30-108).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np


@dataclass
class UNetSpec:
    kind: str                                  # 'plain' | 'resenc'
    in_channels: int
    num_heads: int
    features: List[int]
    kernels: List[Tuple[int, int, int]]
    strides: List[Tuple[int, int, int]]
    n_conv_enc: List[int]                      # convs (plain) or residual blocks (resenc) per stage
    n_conv_dec: List[int]
    conv_bias: bool = True
    eps: float = 1e-5
    slope: float = 0.01
    deep_supervision: bool = False

    @property
    def n_stages(self) -> int:
        return len(self.features)


def _t3(v) -> Tuple[int, int, int]:
    if isinstance(v, int):
        return (v, v, v)
    v = list(v)
    if isinstance(v[0], (list, tuple)):
        v = list(v[0])
    return tuple(int(i) for i in v)


def spec_from_arch_kwargs(class_name: str, kw: dict, in_channels: int, num_heads: int,
                          reduction: int = 1, block_strategy: str = 'keep') -> UNetSpec:
    """Teacher (reduction=1) or distilled student spec from ``arch_kwargs``."""
    resenc = 'Residual' in class_name or 'ResEnc' in class_name
    n = int(kw['n_stages'])
    feats = [int(f) for f in kw['features_per_stage']]
    if reduction != 1:
        feats = [max(f // reduction, 8) for f in feats]
    ks = kw['kernel_sizes']
    kernels = [_t3(ks)] * n if isinstance(ks, int) else [_t3(k) for k in ks]
    strides = [_t3(s) for s in kw['strides']]
    if resenc:
        blocks = [int(b) for b in kw.get('n_blocks_per_stage', [1, 3, 4, 6, 6, 6][:n])]
        if reduction != 1:
            if block_strategy == 'reduce':
                blocks = [max(b // 2, 1) for b in blocks]
            elif block_strategy == 'increase':
                blocks = [min(b + 1, 8) for b in blocks]
            elif block_strategy == 'adaptive':
                orig = [int(f) for f in kw['features_per_stage']]
                blocks = [min(b + max(0, int((o / f) / 4)), 8) for b, o, f in zip(blocks, orig, feats)]
        enc = blocks
    else:
        enc = kw['n_conv_per_stage']
        enc = [int(enc)] * n if isinstance(enc, int) else [int(e) for e in enc]
    dec = kw['n_conv_per_stage_decoder']
    dec = [int(dec)] * (n - 1) if isinstance(dec, int) else [int(e) for e in dec]
    nk = kw.get('norm_op_kwargs') or {}
    return UNetSpec('resenc' if resenc else 'plain', in_channels, num_heads, feats, kernels, strides,
                    enc, dec, bool(kw.get('conv_bias', True)), float(nk.get('eps', 1e-5)))


def plan_pool_and_kernels(spacing: Sequence[float], patch: Sequence[int], min_edge: int = 4,
                          max_pool: int = 999999):
    """Pooling strides and conv kernels per stage for a spacing / patch size.

    Restates ``get_pool_and_conv_props`` (network_topology.py:30-108): an axis
    is pooled while its edge stays >= 2*min_edge and its spacing is within 2x
    of the finest poolable axis; an axis's kernel becomes 3 once its spacing is
    within 2x of the finest axis and stays 3.
    """
    dim = len(spacing)
    sp = [float(s) for s in spacing]
    size = [float(p) for p in patch]
    strides = [[1] * dim]
    kernels = []
    pooled = [0] * dim
    k = [1] * dim
    while True:
        ok = [i for i in range(dim) if size[i] >= 2 * min_edge]
        if not ok:
            break
        finest = min(sp[i] for i in ok)
        ok = [i for i in ok if sp[i] / finest < 2]
        ok = [i for i in ok if pooled[i] < max_pool]
        if len(ok) == 1 and not size[ok[0]] >= 3 * min_edge:
            break
        if not ok:
            break
        for d in range(dim):
            if k[d] != 3 and sp[d] / min(sp) < 2:
                k[d] = 3
        st = [1] * dim
        for i in ok:
            st[i] = 2
            pooled[i] += 1
            sp[i] *= 2
            size[i] = float(np.ceil(size[i] / 2))
        strides.append(st)
        kernels.append(list(k))
    kernels.append([3] * dim)
    return [tuple(s) for s in strides], [tuple(kk) for kk in kernels], pooled


def default_features(n_stages: int, base: int = 32, cap: int = 320) -> List[int]:
    """``min(cap, base * 2**i)`` (default_experiment_planner.py:234-236)."""
    return [min(cap, base * 2 ** i) for i in range(n_stages)]


def student_spec(spacing, patch, in_channels, num_heads, reduction: int = 2) -> UNetSpec:
    """PlainConv student spec for a spacing / patch, via the planner rules."""
    strides, kernels, _ = plan_pool_and_kernels(spacing, patch)
    n = len(strides)
    feats = [max(f // reduction, 8) for f in default_features(n)]
    return UNetSpec('plain', in_channels, num_heads, feats, kernels, strides, [2] * n, [2] * (n - 1))
