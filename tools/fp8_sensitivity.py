"""Where does the e4m3 conv path lose its accuracy?  ResEnc r=2 student (BASELINE config 5's topology) at 64^3, random
weights: fp32 CPU oracle vs the engine with e4m3 operands in the 3x3x3 stride-1 convs of ONE resolution level at a time
(FNN_FP8_LEVELS), of the deep levels only, and of all levels; relative RMSE of the logits and label agreement.
usage (GPU box): python tools/fp8_sensitivity.py > profiles/r03_fp8_sensitivity.txt"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden'))
os.environ['FNN_KNOBS'] = '1'
os.environ['FNN_ZR_MIN_WGS'] = '1'        # every 3x3x3 stride-1 conv on the depth-shift kernels (small grids would otherwise keep the fp16 linear-tap kernels)
import torch
from fast_nnunet_amd import nnUNetPredictor
from fast_nnunet_amd.plans import PlansManager
from oracle.topology import UNetSpec
from oracle.unet import build as build_oracle, synthetic_state_dict

RESENC = UNetSpec('resenc', 1, 3, [16, 32, 64, 128, 160, 160], [(3, 3, 3)] * 6, [(1, 1, 1)] + [(2, 2, 2)] * 5, [1, 3, 4, 6, 6, 6], [1] * 5)
PLAIN = UNetSpec('plain', 1, 3, [16, 32, 64, 128, 160, 160], [(3, 3, 3)] * 6, [(1, 1, 1)] + [(2, 2, 2)] * 5, [2] * 6, [2] * 5)
patch = (64, 64, 64)


def predictor(spec, sd, dtype):
    pm = PlansManager({'dataset_name': 'fp8', 'plans_name': 'nnUNetPlans', 'configurations': {'3d_fullres': {
        'patch_size': list(patch), 'architecture': {'network_class_name': 'PlainConvUNet', 'arch_kwargs': {}, '_kw_requires_import': []}}}})
    dj = {'labels': {('background' if i == 0 else f'c{i}'): i for i in range(spec.num_heads)}, 'channel_names': {'0': 'CT'}, 'file_ending': '.nii.gz'}
    p = nnUNetPredictor(device=torch.device('cuda', 0), allow_tqdm=False, patches_per_forward=2, compute_dtype=dtype)
    p.manual_initialization(None, pm, pm.get_configuration('3d_fullres'), [sd], dj, 'nnUNetTrainer', None)
    return p


for name, spec in (('ResEnc r=2 (1,3,4,6,6,6)', RESENC), ('PlainConv r=2', PLAIN)):
    sd = synthetic_state_dict(spec, 808)
    net = build_oracle(spec, sd)
    x = torch.randn(2, 1, *patch, generator=torch.Generator().manual_seed(8))
    torch.set_num_threads(8)
    with torch.inference_mode():
        ref = net(x)
    key = {'network.' + k: v for k, v in sd.items()} if spec.kind == 'resenc' else sd
    print(f'== {name}, patch 64^3, random weights; e4m3 operands in the 3x3x3 stride-1 convs of ...')
    print(f'{"levels":28s} {"rel. RMSE":>10s} {"label agreement":>16s}')
    arms = [('none (f16)', None)] + [(f'level {l} only ({64 >> l}^3)', 1 << l) for l in range(6)] + \
           [('levels >= 1', 0b111110), ('levels >= 2', 0b111100), ('levels >= 3', 0b111000), ('all', 0b111111)]
    for label, mask in arms:
        if mask is None:
            os.environ.pop('FNN_FP8_LEVELS', None)
            p = predictor(spec, key, 'f16')
        else:
            os.environ['FNN_FP8_LEVELS'] = str(mask)
            p = predictor(spec, key, 'f8')
        got = p.forward_patches(x).cpu()
        rr = float((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
        agree = float((got.argmax(1) == ref.argmax(1)).float().mean())
        print(f'{label:28s} {rr:10.4f} {agree:16.4f}')
        del p
