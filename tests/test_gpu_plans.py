"""The plans of the sweep (tools/plans/*.json, tools/plan_sweep.py: realistic nnU-Net configurations off the BASELINE shapes):
whatever kernels the launch rules pick for them at the sweep's planned batch, one whole-patch forward is the fp32 oracle's
within the suite's tolerance, and no plan reaches the generic fallback kernel."""
import glob
import json
import os

import numpy as np
import pytest
import torch

import bench
from oracle.topology import UNetSpec
from oracle.unet import build as build_oracle
from test_gpu_predictor import MAX_REL, RMSE_REL, _report

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLANS = sorted(glob.glob(os.path.join(ROOT, 'tools', 'plans', '*.json')))


class _Args:
    def __init__(self, plan):
        self.plan, self.batch, self.batch_given = plan, 32, False


@pytest.mark.parametrize('plan', PLANS, ids=[os.path.basename(p)[:-5] for p in PLANS])
def test_plan_forward_matches_fp32_oracle(plan):
    args = _Args(plan)
    w = bench.resolve_workload(args)
    p, sd, info = bench.build_predictor(w, torch.device('cuda', 0), args.batch, 'fp16')
    n = len(info['features'])
    nd = len(info['patch'])
    lift = (lambda t: (1, *t)) if nd == 2 else tuple
    spec = UNetSpec('resenc' if info['resenc'] else 'plain', info['in_channels'], info['heads'], info['features'],
                    [lift(k) for k in info['kernels']], [lift(s) for s in info['strides']],
                    list(info['blocks']) if info['resenc'] else [2] * n, [1 if info['resenc'] else 2] * (n - 1))
    if nd == 2:                      # the oracle runs a `2d` network as depth-1 3-D: Conv2d weights get the depth axis
        sd = {k: (v.unsqueeze(2) if v.ndim == 4 else v) for k, v in sd.items()}
    net = build_oracle(spec, sd)
    x = torch.randn(1, info['in_channels'], *info['patch'], generator=torch.Generator().manual_seed(3))
    p._engine.set_profiling(True)
    got = p.forward_patches(x).cpu()
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    with torch.inference_mode():
        ref = net(x.unsqueeze(2) if nd == 2 else x)
    if nd == 2:
        ref = ref[:, :, 0]
    mr, rr = _report(os.path.basename(plan), got, ref)
    # 3-D plans (6-7 stages, <= 31 layers): the suite's gate; measured 2.1-3.5e-3 / 1.8-2.9e-3.  `2d` plans are deeper (8 stages: 37
    # layers between input and logits, every one storing fp16, over a 4 x 4 bottleneck whose InstanceNorm runs over 16 voxels): the
    # 512^2 plan measures 5.7e-3 / 4.9e-3 on the plane kernels and 4.7e-3 / 4.3e-3 on the linear-tap kernels they replace
    # (FNN_NO_ZP=1, profiles/r06_plans_parity.txt) - a property of the fp16 chain, not of a kernel: 1.5 x the gate, like the deep
    # random topologies of test_gpu_predictor.py
    # (the 7-stage thick-slice r = 2 student likewise: 3.8e-3 / 3.8e-3 on the tree, 4.0e-3 / 3.7e-3 with FNN_NO_ZP=1)
    k = 1.5 if nd == 2 or n >= 7 else 1.0
    assert mr <= k * MAX_REL and rr <= k * RMSE_REL
    assert not any('generic' in name for name in p._engine.kernel_log()), p._engine.kernel_log()
