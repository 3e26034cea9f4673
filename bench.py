#!/usr/bin/env python3
"""Benchmark of the sliding-window inference hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], SURVEY.md 8d "C2"): a distilled r=2
PlainConvUNet student with the fast_nnunet_bone_turbo geometry - patch
160x96x96, 61 classes, anisotropic kernels/strides planned from the .ini's
target spacing - predicting one synthetic 512^3 CT with half-overlap tiles and
Gaussian blending: 600 patches per volume.  Synthetic data, random-init weights
(seed 1234).  One step = one whole volume through
``nnUNetPredictor.predict_sliding_window_return_logits`` with the volume already
resident in HBM.  N > 1 shards the patches of the SAME volume over the ranks
(strong scaling) with an exchange of patch activations over RCCL and ends with every rank holding
the assembled label map (``--gather``); without a launcher ``--gpus N`` starts
the N ranks itself.

Prints ONE JSON line (rank 0) with the driver's contract plus ``roofline`` and
``cpu_baseline`` (N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0          # dense fp16/bf16, MI355X_MICROARCH.md "Chip-level parameters"
MFMA_PEAK_CLOCK_GHZ = 2.4          # the clock that peak is quoted at (same table: max clock 2400 MHz)
TRAFFIC_FILE = 'r06_traffic.json'  # per-launch HBM bytes of the conv family from separate rocprofv3 --pmc passes, per workload, keyed by csrc_sha256()

WORKLOADS = {
    # name: (spacing, patch, heads, reduction)
    'bone_turbo_r2': ((2.0, 0.9765625, 0.9765625), (160, 96, 96), 61, 2),
    'iso128_r2': ((1.0, 1.0, 1.0), (128, 128, 128), 2, 2),
    'iso128_teacher': ((1.0, 1.0, 1.0), (128, 128, 128), 2, 1),
    # BASELINE config 5: ResidualEncoderUNet student, blocks (1, 3, 4, 6, 6, 6), decoder n_conv 1
    # (nnUNetDistillationTrainer.py:248-266, residual_encoder_unet_planners.py:30-31); --dtype f8 for its fp8 conv path
    'resenc160_r2': ((1.0, 1.0, 1.0), (160, 160, 160), 3, 2),
}
RESENC_BLOCKS = (1, 3, 4, 6, 6, 6)
# the bench line is BASELINE configs[1] (bone_turbo_r2); the other workloads are for DESIGN.md's tables.


def plan_topology(spacing, patch, min_edge=4):
    """Strides / kernels per stage from spacing + patch (same planning rule the reference uses,
    experiment_planning/experiment_planners/network_topology.py:30-108; restated in the product because
    the benchmark must not depend on the oracle)."""
    dim = len(spacing)
    sp, size = [float(s) for s in spacing], [float(p) for p in patch]
    strides, kernels, k, pooled = [[1] * dim], [], [1] * dim, [0] * dim
    while True:
        ok = [i for i in range(dim) if size[i] >= 2 * min_edge]
        if not ok:
            break
        finest = min(sp[i] for i in ok)
        ok = [i for i in ok if sp[i] / finest < 2]
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
    return strides, kernels


def synthetic_checkpoint(features, kernels, strides, in_ch, heads, seed=1234):
    """Random-init state dict in the checkpoint key schema: He-normal(a=0.01) conv weights like the
    reference's InitWeights_He (utilities/network_initialization.py:4-12), small biases,
    InstanceNorm gamma~U(0.5,1.5), beta~N(0,0.1)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    gain = (2.0 / (1 + 0.01 ** 2)) ** 0.5

    def conv(prefix, cout, cin, k):
        fan_in = cin * int(np.prod(k))
        sd[prefix + '.conv.weight'] = torch.randn(cout, cin, *k, generator=g) * (gain / fan_in ** 0.5)
        sd[prefix + '.conv.bias'] = torch.randn(cout, generator=g) * 0.05
        sd[prefix + '.norm.weight'] = torch.rand(cout, generator=g) + 0.5
        sd[prefix + '.norm.bias'] = torch.randn(cout, generator=g) * 0.1

    n = len(features)
    ones = [1] * len(kernels[0])                       # (Conv2d weights are 4-D: `2d` configurations)
    cin = in_ch
    for s in range(n):
        for i in range(2):
            conv(f'encoder.stages.{s}.0.convs.{i}', features[s], cin, kernels[s])
            cin = features[s]
    for d in range(n - 1):
        below, skip, st = features[-(d + 1)], features[-(d + 2)], strides[-(d + 1)]
        fan_in = below * int(np.prod(st))
        sd[f'decoder.transpconvs.{d}.weight'] = torch.randn(below, skip, *st, generator=g) * (gain / fan_in ** 0.5)
        sd[f'decoder.transpconvs.{d}.bias'] = torch.randn(skip, generator=g) * 0.05
        conv(f'decoder.stages.{d}.convs.0', skip, 2 * skip, kernels[-(d + 2)])
        conv(f'decoder.stages.{d}.convs.1', skip, skip, kernels[-(d + 2)])
        sd[f'decoder.seg_layers.{d}.weight'] = torch.randn(heads, skip, *ones, generator=g) * (gain / skip ** 0.5)
        sd[f'decoder.seg_layers.{d}.bias'] = torch.randn(heads, generator=g) * 0.05
    return sd


def synthetic_resenc_checkpoint(features, kernels, strides, blocks, in_ch, heads, seed=1234):
    """ResidualEncoderUNet state dict in the checkpoint key schema: stem, BasicBlockD stages
    (conv1, conv2, projected skip), UNetDecoder with one conv per stage; same initialisation as synthetic_checkpoint."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    gain = (2.0 / (1 + 0.01 ** 2)) ** 0.5

    def conv(prefix, cout, cin, k, bias=True):
        fan_in = cin * int(np.prod(k))
        sd[prefix + '.conv.weight'] = torch.randn(cout, cin, *k, generator=g) * (gain / fan_in ** 0.5)
        if bias:
            sd[prefix + '.conv.bias'] = torch.randn(cout, generator=g) * 0.05
        sd[prefix + '.norm.weight'] = torch.rand(cout, generator=g) + 0.5
        sd[prefix + '.norm.bias'] = torch.randn(cout, generator=g) * 0.1

    n = len(features)
    ones = tuple([1] * len(kernels[0]))
    conv('encoder.stem.convs.0', features[0], in_ch, kernels[0])
    cin = features[0]
    for s in range(n):
        for b in range(blocks[s]):
            pre = f'encoder.stages.{s}.blocks.{b}'
            conv(pre + '.conv1', features[s], cin, kernels[s])
            conv(pre + '.conv2', features[s], features[s], kernels[s])
            strided = b == 0 and any(v != 1 for v in strides[s])
            if cin != features[s]:
                conv(f'{pre}.skip.{1 if strided else 0}', features[s], cin, ones, bias=False)
            cin = features[s]
    for d in range(n - 1):
        below, skip, st = features[-(d + 1)], features[-(d + 2)], strides[-(d + 1)]
        fan_in = below * int(np.prod(st))
        sd[f'decoder.transpconvs.{d}.weight'] = torch.randn(below, skip, *st, generator=g) * (gain / fan_in ** 0.5)
        sd[f'decoder.transpconvs.{d}.bias'] = torch.randn(skip, generator=g) * 0.05
        conv(f'decoder.stages.{d}.convs.0', skip, 2 * skip, kernels[-(d + 2)])
        sd[f'decoder.seg_layers.{d}.weight'] = torch.randn(heads, skip, *ones, generator=g) * (gain / skip ** 0.5)
        sd[f'decoder.seg_layers.{d}.bias'] = torch.randn(heads, generator=g) * 0.05
    return sd


def resolve_workload(args):
    """The workload as one dict: a named BASELINE workload (cubic --volume), or a plan file (--plan: a JSON object with
    name, patch, spacing, in_channels, heads, r, kind 'plain' | 'resenc', volume [X, Y, Z] - a realistic nnU-Net
    configuration off the BASELINE shapes, tools/plans/*.json)."""
    if args.plan:
        j = json.load(open(args.plan))
        patch = tuple(int(v) for v in j['patch'])
        w = dict(name=j.get('name', os.path.splitext(os.path.basename(args.plan))[0]), spacing=tuple(float(v) for v in j['spacing']),
                 patch=patch, heads=int(j['heads']), r=int(j.get('r', 1)), in_channels=int(j.get('in_channels', 1)),
                 resenc=j.get('kind', 'plain') == 'resenc', volume=tuple(int(v) for v in j['volume']),
                 max_features=int(j.get('max_features', 320 if len(patch) == 3 else 512)), plan=True, note=j.get('note', ''))
        if 'batch' in j and not args.batch_given:
            args.batch = int(j['batch'])
        return w
    return named_workload(args.workload, args.volume)


def named_workload(name, volume=512):
    spacing, patch, heads, r = WORKLOADS[name]
    return dict(name=name, spacing=spacing, patch=patch, heads=heads, r=r, in_channels=1,
                resenc=name.startswith('resenc'), volume=(volume,) * 3, max_features=320, plan=False, note='')


def build_predictor(w, device, batch, accumulate_in, compute_dtype='f16', mirror=False, folds=1):
    from fast_nnunet_amd import nnUNetPredictor
    from fast_nnunet_amd.plans import PlansManager
    if isinstance(w, str):                                      # a BASELINE workload by name
        w = named_workload(w)
    spacing, patch, heads, r, in_ch = w['spacing'], w['patch'], w['heads'], w['r'], w['in_channels']
    strides, kernels = plan_topology(spacing, patch)
    n = len(strides)
    plan_features = [min(w['max_features'], 32 * 2 ** i) for i in range(n)]
    features = [max(f // r, 8) for f in plan_features]
    resenc = w['resenc']
    nd = len(patch)
    conv_op = 'torch.nn.modules.conv.Conv%dd' % nd
    if resenc:
        blocks = (list(RESENC_BLOCKS) + [RESENC_BLOCKS[-1]] * n)[:n]
        sds = [synthetic_resenc_checkpoint(features, kernels, strides, blocks, in_ch, heads, seed=1234 + f) for f in range(folds)]
        arch = {'network_class_name': 'dynamic_network_architectures.architectures.unet.ResidualEncoderUNet',
                'arch_kwargs': {'n_stages': n, 'features_per_stage': plan_features,
                                'conv_op': conv_op, 'kernel_sizes': kernels, 'strides': strides, 'n_blocks_per_stage': blocks,
                                'n_conv_per_stage_decoder': [1] * (n - 1), 'conv_bias': True,
                                'norm_op_kwargs': {'eps': 1e-5, 'affine': True}},
                '_kw_requires_import': []}
    else:
        sds = [synthetic_checkpoint(features, kernels, strides, in_ch, heads, seed=1234 + f) for f in range(folds)]
        arch = {'network_class_name': 'dynamic_network_architectures.architectures.unet.PlainConvUNet',
                'arch_kwargs': {'n_stages': n, 'features_per_stage': plan_features,
                                'conv_op': conv_op, 'kernel_sizes': kernels, 'strides': strides, 'n_conv_per_stage': [2] * n,
                                'n_conv_per_stage_decoder': [2] * (n - 1), 'conv_bias': True,
                                'norm_op_kwargs': {'eps': 1e-5, 'affine': True}},
                '_kw_requires_import': []}
    cfg = '3d_fullres' if nd == 3 else '2d'
    pm = PlansManager({'dataset_name': 'Dataset000_Synthetic', 'plans_name': 'nnUNetPlans',
                       'configurations': {cfg: {'patch_size': list(patch), 'spacing': list(spacing),
                                                'architecture': arch}}})
    dj = {'labels': {('background' if i == 0 else f'class_{i}'): i for i in range(heads)},
          'channel_names': {str(c): 'CT' for c in range(in_ch)}, 'file_ending': '.nii.gz'}
    p = nnUNetPredictor(tile_step_size=0.5, use_gaussian=True, use_mirroring=mirror, perform_everything_on_device=True,
                        device=device, allow_tqdm=False, accumulate_in=accumulate_in, patches_per_forward=batch,
                        compute_dtype=compute_dtype)
    p._reduction = None
    p.manual_initialization(None, pm, pm.get_configuration(cfg), sds, dj, 'nnUNetDistillationTrainer',
                            tuple(range(nd)) if mirror else None)
    return p, sds[0], dict(features=features, kernels=kernels, strides=strides, patch=patch, heads=heads, r=r, resenc=resenc,
                           in_channels=in_ch, blocks=(blocks if resenc else None))


def synthetic_volume(size, device, channels=1):
    """Preprocessed CT-like tensor: HU ~ N(418.68, 412.19) clipped to [-60, 3068], z-scored with the
    fast_nnunet_bone_turbo constants (engine/config/fast_nnunet_bone_turbo.ini:15-18).  `size`: an edge (cube) or (X, Y, Z)."""
    shape = (size,) * 3 if isinstance(size, int) else tuple(size)
    g = torch.Generator(device='cpu').manual_seed(0)
    out = torch.empty((channels, *shape), dtype=torch.float32)
    for x in range(0, shape[0], 64):                       # chunked: keeps the host working set small
        hu = torch.randn((min(64, shape[0] - x), shape[1], shape[2]), generator=g) * 412.1883239746094 + 418.6798400878906
        out[0, x:x + hu.shape[0]] = (hu.clamp_(-60.0, 3068.0) - 418.6798400878906) / 412.1883239746094
    for c in range(1, channels):                           # further channels: the first one shifted by c voxels along z (cheap, distinct)
        out[c] = out[0].roll(c, dims=2)
    return out.to(device)


def cpu_baseline(sd, info, seconds_budget=60.0):
    """The oracle (CPU restatement of the reference path: fp32 network, fp16 Gaussian accumulate) timed on
    the host cores on a bounded sample of the same workload, at the two thread settings the reference uses:
    all cores (its CLI, predict_from_raw_data.py:954-958) and 8 threads (default_num_processes inside
    predict_logits_from_preprocessed_data, :479-480)."""
    from oracle import sliding_window as osw
    from oracle.topology import UNetSpec
    from oracle.unet import build as build_oracle
    n = len(info['features'])
    if info['resenc']:
        spec = UNetSpec('resenc', info['in_channels'], info['heads'], info['features'], [tuple(k) for k in info['kernels']],
                        [tuple(s) for s in info['strides']], list(info['blocks']), [1] * (n - 1))
    else:
        spec = UNetSpec('plain', info['in_channels'], info['heads'], info['features'], [tuple(k) for k in info['kernels']],
                        [tuple(s) for s in info['strides']], [2] * n, [2] * (n - 1))
    net = build_oracle(spec, sd)
    patch = info['patch']
    all_threads = torch.get_num_threads()
    # sub-volume that holds exactly 2 x 2 x 2 = 8 patches at step 0.5
    shape8 = tuple(int(p * 1.5) for p in patch)
    image8 = synthetic_volume(max(shape8), torch.device('cpu'), info['in_channels'])[:, :shape8[0], :shape8[1], :shape8[2]].contiguous()

    def run(threads):
        torch.set_num_threads(threads)
        with torch.inference_mode():
            t0 = time.perf_counter()
            net(image8[:, :patch[0], :patch[1], :patch[2]][None])        # warm-up + per-patch estimate
            one = time.perf_counter() - t0
        shape, image, n_patches = shape8, image8, 8
        if one * 9 > seconds_budget:                                      # slow setting: 2 patches instead of 8
            shape = (patch[0], patch[1], int(patch[2] * 1.5))
            image = image8[:, :shape[0], :shape[1], :shape[2]].contiguous()
            n_patches = 2
        t0 = time.perf_counter()
        osw.sliding_window_logits(net, image, patch, info['heads'], step=0.5, use_gaussian=True, accum='fp16')
        dt = time.perf_counter() - t0
        return {'value': round(n_patches / dt, 4), 'cores': threads, 's_per_patch': round(dt / n_patches, 4),
                'sample': f'{n_patches} patches ({shape[0]}x{shape[1]}x{shape[2]} sub-volume of the same synthetic CT)'}

    eight = run(min(8, all_threads))
    full = run(all_threads) if all_threads > 8 else dict(eight)
    torch.set_num_threads(all_threads)
    # `value` is the reference's own behaviour: predict_logits_from_preprocessed_data caps torch at default_num_processes
    # = 8 threads (predict_from_raw_data.py:479-480, configuration.py:5) whatever its CLI set before (:954-958); the
    # all-cores figure (slower: torch's CPU convs lose beyond a few dozen threads) is kept next to it
    return {'value': eight['value'], 'unit': 'patches/s', 'cores': eight['cores'], 'kind': 'port',
            'sample': eight['sample'] + f', fp32 network + fp16 accumulators, torch {torch.__version__} CPU, '
                                        f'{os.cpu_count()} host cpus visible',
            's_per_patch': eight['s_per_patch'],
            'all_cores': {'value': full['value'], 'unit': 'patches/s', 'cores': full['cores'],
                          'sample': full['sample'], 's_per_patch': full['s_per_patch']}}


def csrc_sha256():
    """Hash of the HIP sources the library is built from: the key that ties committed counter traffic to the code."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'fast-nnunet_amd', 'csrc')
    for f in sorted(glob.glob(os.path.join(d, '*.hip')) + glob.glob(os.path.join(d, '*.h')) + [os.path.join(d, 'Makefile')]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()


def traffic_key(args):
    return f'{args.workload}|{args.dtype}|mirror={int(args.mirror)}|{args.accum}|vol={args.volume}|batch={args.batch}'


def conv_family_names(kernel_counts):
    """The kernel templates of the conv family among a profiled step's launches (their names without template arguments)."""
    fam = sorted({k.split('<')[0].split(' ')[0] for k in kernel_counts if k.startswith('conv')})
    return fam or ['(none)']


def layer_table(engine, batch):
    """Per layer of the network: the kernel the launch rules picked and what it achieves - median microseconds per forward
    over the profiled step's full batches (HIP events around every launch, one stream), algorithmic TFLOP/s (2*MACs) and
    TB/s (every input read once + the output written once, fp16)."""
    layers = {L['index']: L for L in engine.layer_table()}
    per = {}
    for layer, fam, ms, flops, by, kern in engine.profile_launches():
        if layer < 0:
            continue
        per.setdefault(layer, []).append((flops, ms, by, kern, fam))
    rows = []
    for li in sorted(per):
        full = max(f for f, *_ in per[li])
        runs = [r for r in per[li] if r[0] == full] or per[li]          # the batches of `batch` patches (the tail batch is smaller)
        ms = float(np.median([r[1] for r in runs]))
        L = layers[li]
        flops, by = runs[0][0], runs[0][2]
        if L['type'] == 'tconv':                                        # (the engine prices transposed convs' bytes as 0: priced here)
            vi, vo = np.prod([int(v) for v in L['in_dims'].split('x')]), np.prod([int(v) for v in L['out_dims'].split('x')])
            by = 2.0 * (L['cin'] * vi + L['cout'] * vo) * (flops / L['flops'] if L['flops'] else batch)
        row = {'layer': li, 'type': L['type'] + ('+producer' if L['fused'] == 2 else ''), 'cin': L['cin'], 'cout': L['cout'],
               'kernel': L['kernel'], 'stride': L['stride'], 'out': L['out_dims'], 'picked': runs[0][3], 'us': round(ms * 1e3, 1),
               'tflops': round(flops / (ms * 1e-3) / 1e12, 1) if ms > 0 else None,
               'tbps': round(by / (ms * 1e-3) / 1e12, 2) if ms > 0 and by else None}
        # tensors are stored with their channels padded to 16: what a layer with fewer channels really moves (conv / tconv: fp16 in and out;
        # the input layer: fp32 in, padded fp16 out) - next to the algorithmic figure, not instead of it
        if ms > 0 and L['type'] in ('conv', 'tconv', 'input') and (L['cin'] % 16 or L['cout'] % 16):
            pad = lambda c: (c + 15) // 16 * 16
            vi, vo = np.prod([int(v) for v in L['in_dims'].split('x')]), np.prod([int(v) for v in L['out_dims'].split('x')])
            n_items = flops / L['flops'] if L['flops'] else batch
            cin_stored = 2 * pad(L['cout']) if (L['type'] == 'conv' and L['cin'] == 2 * L['cout']) else pad(L['cin'])   # (a decoder conv's two sources are padded one by one)
            real = (4.0 * L['cin'] * vi + 2.0 * pad(L['cout']) * vo) if L['type'] == 'input' else 2.0 * (cin_stored * vi + pad(L['cout']) * vo)
            row['tbps_padded'] = round(real * n_items / (ms * 1e-3) / 1e12, 2)
        rows.append(row)
    tot = sum(r['us'] for r in rows) or 1.0
    for r in rows:
        r['share'] = round(r['us'] / tot, 4)
    return rows


ALSO = [('C1 iso128_r2', ['--workload', 'iso128_r2']),
        ('C4 iso128_teacher, 5-fold ensemble', ['--workload', 'iso128_teacher', '--folds', '5']),
        ('C5 resenc160_r2 f16', ['--workload', 'resenc160_r2']),
        ('C5 resenc160_r2 f8', ['--workload', 'resenc160_r2', '--dtype', 'f8'])]


def also_block(device):
    """The other single-GPU BASELINE configurations, 3 timed steps each (1 warm-up), every one in a process of its own
    started AFTER this process has finished its GPU work (its memory is released first) - so that the driver's record
    carries C1 / C4 / C5 next to the headline.  Child lines are trimmed to the fields a reader compares."""
    import gc
    import subprocess
    gc.collect()
    torch.cuda.empty_cache()
    out = {}
    for name, flags in ALSO:
        cmd = [sys.executable, os.path.abspath(__file__), '--gpus', '1', '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-also',
               '--no-from-host'] + flags
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
            if r.returncode != 0 or not line:
                out[name] = {'error': (r.stderr or r.stdout)[-300:]}
                continue
            j = json.loads(line[-1])
            rf = j.get('roofline', {})
            out[name] = {'value': j['value'], 'unit': j['unit'], 'ms_per_step': j['ms_per_step'], 'steps': j['steps'], 'dtype': j['dtype'],
                         'workload': j['config']['workload'], 'folds': j['config']['folds'],
                         'frac': rf.get('frac'), 'achieved_tflops': rf.get('achieved'), 'clock_ghz': rf.get('clock_ghz'),
                         'frac_at_clock': rf.get('frac_at_clock'), 'traffic_over_algorithmic': rf.get('traffic_over_algorithmic'),
                         'hidden_by_batches_in_flight': rf.get('schedules', {}).get('hidden_by_batches_in_flight'),
                         'wall_s': round(time.perf_counter() - t0, 1)}
            if 'sec_per_volume_per_fold' in j:
                out[name]['sec_per_volume'] = j['sec_per_volume']
        except subprocess.TimeoutExpired:
            out[name] = {'error': 'timeout'}
    return out


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one per GPU) BEFORE anything in
    this process touches the GPU, wait for them, and leave with the worst exit code.  Rank 0 prints the JSON line."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    sys.exit(rc)


def main():
    # The clock probe (one extra untimed step, below) is a kernel that runs for most of a step on a stream of its own: with
    # HIP's default of four hardware queues it shares one with a stream of the engine and everything behind it waits (2490
    # instead of 3440 patches/s next to it, measured); with eight the step is the same with and without it (3440 / 3424).
    # The timed steps themselves do not care (3440 / 3440, 3347 / 3357 on two boxes).  Read at HIP's initialisation, i.e.
    # before the first torch.cuda call below.
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default='bone_turbo_r2', choices=list(WORKLOADS))
    ap.add_argument('--plan', default=None,
                    help='a plan file (tools/plans/*.json: patch, spacing, input channels, classes, r, kind, volume) instead of a named '
                         'workload: the topology comes from the planning rule (plan_topology), the line carries a per-layer table')
    ap.add_argument('--volume', type=int, default=512)
    ap.add_argument('--batch', type=int, default=None, help='patches per forward (default 32, or the plan file\'s)')
    ap.add_argument('--accum', default='fp16', choices=['fp16', 'fp32', 'fp16_autocast'],
                    help="accumulation arithmetic: 'fp16' = the reference without autocast (its CPU path; the oracle's "
                         "default and the bench line), 'fp16_autocast' = the reference on a GPU (fp16 network output)")
    ap.add_argument('--dtype', default='f16', choices=['f16', 'f8'],
                    help='operand format of the 3x3x3 stride-1 convolutions (f8: OCP e4m3, BASELINE config 5)')
    ap.add_argument('--mirror', action='store_true',
                    help='test-time mirroring over all three axes (8 evaluations per patch; the reference default, off in the '
                         'bone_turbo .ini and in the bench line)')
    ap.add_argument('--layers', action='store_true', help='add the per-layer table of the profiled step (always on with --plan)')
    ap.add_argument('--no-from-host', action='store_true', help='skip the steps that start from a host tensor (ms_per_step_from_host)')
    ap.add_argument('--no-also', action='store_true',
                    help='skip the `also` block (N = 1, default workload only: 3 timed steps each of the other single-GPU BASELINE '
                         'configurations, each in a process of its own)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-clock-probe', action='store_true', help='skip the extra step that samples the shader clock (fnn_clock_probe_*)')
    ap.add_argument('--force-sharded', action='store_true', help='run the multi-GPU code path even with one rank')
    ap.add_argument('--gather', default='none', choices=['labels', 'logits', 'none'],
                    help='multi-GPU: what the timed step ends with.  none (default): the N = 1 step sharded - every rank holds '
                         'the fp16 logits of the box it owns in its HBM, nothing is assembled; labels: argmax on the owner, '
                         'then all_gather of the uint8 slabs; logits: all_gather of the fp16 logits.  The assembled-labels '
                         'step is always timed too and reported as ms_per_step_labels_assembled')
    ap.add_argument('--folds', type=int, default=1,
                    help='fold ensemble (BASELINE configs[3]: 5 folds of the teacher): the folds stay resident, their logits are '
                         'averaged on the device (fnn_predict_volume_ensemble); one step = one volume through ALL folds')
    args = ap.parse_args()
    args.batch_given = args.batch is not None
    if args.batch is None:
        args.batch = 32

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus)                                  # does not return
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch one rank per GPU '
                         f'(python -m torch.distributed.run --nproc-per-node {args.gpus} ...) or drop WORLD_SIZE')
    distributed = world > 1 or args.force_sharded
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29544')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X; the engine has no CPU path')
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)

    accumulate_in = args.accum                                  # halo sums travel in the accumulator dtype
    w = resolve_workload(args)
    if len(w['patch']) == 2 and distributed:
        raise SystemExit('bench.py: the sharded path is 3-D only')
    predictor, sd, info = build_predictor(w, device, args.batch, accumulate_in, args.dtype, args.mirror, args.folds)
    vol = synthetic_volume(w['volume'], device, w['in_channels'])
    from fast_nnunet_amd import capi
    n_patches = capi.plan_volume(info['patch'], vol.shape[1:], 0.5)[2].shape[0]

    clock = {}

    def timed(step_fn, barrier, steps):
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step_fn()
            del out
        barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if distributed:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    dt_labels = None
    fold_list = list(range(args.folds)) if args.folds > 1 else None
    if distributed:
        from fast_nnunet_amd.dist import ShardedPredictor
        runner = ShardedPredictor(predictor, dist.group.WORLD)
        if args.gather == 'labels':
            step_fn = lambda: runner.predict_segmentation_from_preprocessed_data(vol)
        else:
            step_fn = lambda: runner.predict_sliding_window_return_logits(vol, gather=args.gather == 'logits', folds=fold_list)
        labels_fn = lambda: runner.predict_segmentation_from_preprocessed_data(vol)
        barrier = lambda: dist.barrier()
    elif args.folds > 1:
        step_fn = lambda: predictor.predict_logits_from_preprocessed_data(vol, on_device=True)
        barrier = lambda: None
    else:
        step_fn = lambda: predictor.predict_sliding_window_return_logits(vol)
        barrier = lambda: None

    step_est = None
    for _ in range(args.warmup):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = step_fn()
        del out
        torch.cuda.synchronize()
        step_est = time.perf_counter() - t0
    dt = timed(step_fn, barrier, args.steps)
    if step_est and not args.no_clock_probe and not args.no_roofline:
        # one extra, identical, UNTIMED step beside the clock probe (one sleeping wave, fnn_clock_probe_*): the shader clock the
        # device holds under this step.  Not in the timed region (it costs ~0.5 % there) and not in the profiled step below
        # (it stretches the per-launch durations by ~4 %, measured: 734 -> 700 TFLOP/s)
        step_timed = dt / args.steps
        for attempt in range(3):                                 # (a short step's duration jitters by more than the gate: sampled again)
            barrier()
            torch.cuda.synchronize()
            probe = capi.clock_probe_start(local_rank, min(30.0, 4.0 * step_est))
            out = step_fn()                                      # (returns when the volume is done: the engine's entry points are synchronous)
            clock['ghz'], clock['seconds'] = capi.clock_probe_stop(probe)   # raises the probe's flag and waits for ITS stream only
            del out
            torch.cuda.synchronize()
            # the sample only counts when the probed step ran like the timed ones: with fewer hardware queues than streams (HIP
            # initialised before main() set GPU_MAX_HW_QUEUES - a profiler's preload, an embedding process) the probe shares a queue
            # with an engine stream and the step waits behind it; then the clock is that of a stalled device
            tol = 0.15 if args.steps >= 3 else 0.35               # (one or two timed steps are themselves a noisy yardstick)
            if (1 - tol) * step_timed <= clock['seconds'] <= (1 + tol) * step_timed:
                break
            if attempt == 2:
                clock = {'rejected': f"probed step took {clock['seconds']:.3f} s against {step_timed:.3f} s timed: the probe disturbed the step "
                                     f"(GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}), sample dropped"}
    if distributed and args.gather != 'labels':
        out = labels_fn()                                        # (one untimed call: first-use allocations of this entry point)
        del out
        dt_labels = timed(labels_fn, barrier, max(1, min(args.steps, 5)))   # the step that ends with the label map on every rank

    from_host = None
    if not distributed and not args.no_from_host and not w['plan'] and args.folds == 1:
        # The same step as a reference caller sees it: the preprocessed volume is a CPU tensor (what the preprocessing iterator
        # yields, data_iterators.py:116-117; `data.to(results_device)` at predict_from_raw_data.py:579).  The engine uploads it
        # in tiles on a copy stream under the first batches (engine.hip, stage_volume / upload_box).  Untimed extras.
        def raw_copy(t):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            d = t.to(device)
            torch.cuda.synchronize()
            del d
            return (time.perf_counter() - t0) * 1e3
        k = max(1, min(args.steps, 3))
        vol_cpu = vol.cpu()
        raw_pageable = min(raw_copy(vol_cpu) for _ in range(2))
        fn = lambda: predictor.predict_sliding_window_return_logits(vol_cpu)
        out = fn(); del out
        dt_pageable = timed(fn, barrier, k) / k
        vol_pin = vol_cpu.pin_memory()
        raw_pinned = min(raw_copy(vol_pin) for _ in range(2))
        fn = lambda: predictor.predict_sliding_window_return_logits(vol_pin)
        out = fn(); del out
        dt_pinned = timed(fn, barrier, k) / k
        del vol_pin, vol_cpu
        from_host = {'ms_per_step_pinned': round(dt_pinned * 1e3, 3), 'ms_per_step_pageable': round(dt_pageable * 1e3, 3),
                     'raw_copy_ms_pinned': round(raw_pinned, 3), 'raw_copy_ms_pageable': round(raw_pageable, 3), 'steps': k,
                     'volume_mib': round(vol.numel() * 4 / 2 ** 20, 1),
                     'note': 'the same step with the volume handed over as a CPU tensor: uploaded in tiles (planes x rows) on a copy stream, a batch '
                             'starts when the tiles under its patches have landed; raw_copy_ms = one torch .to(device) of the same tensor'}

    flops_patch, act_bytes_patch = predictor._engine.patch_work()
    assembly = {'labels': 'labels on the owner of each box, all_gather of the uint8 slabs: the label map on every rank',
                'logits': 'all_gather of the fp16 logits of the owned boxes: the logits on every rank',
                'none': 'fp16 logits in HBM, every rank holds the box of the volume it owns (the N = 1 step sharded; nothing assembled)'}[args.gather] if distributed else \
        'fp16 logits [heads, X, Y, Z] in HBM (predict_sliding_window_return_logits)'
    out_kind = {'labels': 'label map assembled on every rank', 'logits': 'fp16 logits assembled on every rank',
                'none': 'fp16 logits resident in HBM'}[args.gather] if distributed else 'fp16 logits resident in HBM'
    vshape = 'x'.join(map(str, w['volume']))
    vol_desc = f'{args.volume}^3 volume' if not w['plan'] else f'{vshape} volume'
    what = 'distilled r=2 student, sliding window, one 512^3 CT' if not w['plan'] else f'plan {w["name"]}, sliding window, one {vshape} volume'
    result = {
        'metric': f'3d_fullres patches/sec ({what}; volume resident in HBM at step start; step ends with: {out_kind})',
        'value': round(n_patches * args.folds * args.steps / dt, 3),
        'unit': 'patches/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(dt / args.steps * 1e3, 3),
        'sec_per_volume': round(dt / args.steps, 4),
        'higher_is_better': True,
        'scaling': 'strong',
        'vs_baseline': None,
        'dtype': args.dtype,
        'data': 'synthetic',
        'config': {'workload': f'{w["name"]}: {"ResidualEncoderUNet" if info["resenc"] else "PlainConvUNet"} '
                               f'{"student" if info["r"] > 1 else "teacher"} r={info["r"]}, '
                               f'features {info["features"]}, {info["in_channels"]} input channel{"s" if info["in_channels"] > 1 else ""}, '
                               f'patch {"x".join(map(str, info["patch"]))}, {info["heads"]} classes, '
                               f'{vol_desc}, tile_step_size 0.5, Gaussian on, mirroring {"on (all axes)" if args.mirror else "off"}, '
                               f'{n_patches} patches/volume',
                   'patches_per_forward': args.batch,
                   'folds': args.folds,
                   'accumulators': accumulate_in,
                   'gflop_per_patch': round(flops_patch / 1e9, 2),
                   'step_output': assembly,
                   'gpu_max_hw_queues': os.environ.get('GPU_MAX_HW_QUEUES'),
                   'parallelism': f'patch-sharded x{world}, patch-activation exchange + slab gather over RCCL' if distributed else 'single GPU'},
    }
    if w['plan']:
        result['config']['plan'] = {'file': os.path.relpath(os.path.abspath(args.plan), ROOT), 'spacing': list(w['spacing']),
                                    'kernels': info['kernels'], 'strides': info['strides'], 'note': w['note']}
    if from_host is not None:
        result['ms_per_step_from_host'] = from_host['ms_per_step_pinned']
        from_host['minus_resident_ms'] = {'pinned': round(from_host['ms_per_step_pinned'] - result['ms_per_step'], 3),
                                          'pageable': round(from_host['ms_per_step_pageable'] - result['ms_per_step'], 3)}
        result['from_host'] = from_host
    if args.folds > 1:
        result['sec_per_volume_per_fold'] = round(dt / args.steps / args.folds, 4)
        result['config']['ensemble'] = (f'{args.folds} resident folds, logits averaged on the device; value counts one patch forward per fold '
                                        f'({n_patches} x {args.folds} per volume), sec_per_volume is the whole ensemble')
    if dt_labels is not None:
        result['ms_per_step_labels_assembled'] = round(dt_labels / max(1, min(args.steps, 5)) * 1e3, 3)
        result['value_labels_assembled'] = round(n_patches * args.folds / (dt_labels / max(1, min(args.steps, 5))), 3)
        result['series_note'] = ('rounds 1-3 timed the N > 1 step with the label map assembled on every rank (--gather labels); since round 4 '
                                 '`value` is the N = 1 step sharded (--gather none: each rank ends with the fp16 logits of its box) and the '
                                 'assembled step is `value_labels_assembled` / `ms_per_step_labels_assembled`: compare like with like')
    if distributed:
        # one extra, profiled step (the device is synchronised at every phase boundary, so it is slower than the timed
        # ones): where a rank's time goes and what it exchanges - per phase the MAX over ranks, bytes per rank as a list
        ph = runner.start_phases()
        out = step_fn()
        del out
        runner.phases = None
        keys = sorted(ph)
        mine = torch.tensor([float(ph[k]) for k in keys], device=device, dtype=torch.float64)
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        allv = torch.stack(allv).cpu()
        result['phases_profiled_step'] = {
            'note': 'wall ms per phase of one extra step with a device synchronisation at every phase boundary; max over ranks',
            **{k: round(float(allv[:, i].max()), 3) for i, k in enumerate(keys) if k.endswith('_ms')},
            'per_rank': {k: [int(v) for v in allv[:, i]] for i, k in enumerate(keys) if not k.endswith('_ms')},
            'mode': runner.last_mode,
            'rccl_ranks': dist.get_world_size()}
        assert dist.get_world_size() == args.gpus, 'the process group does not have one rank per requested GPU'

    if not args.no_roofline:
        # one extra, identical step with HIP events around every launch (recorded by the engine on the
        # launch stream) -> duration of the dominant kernel family (the MFMA convs).  N > 1: every rank runs the step (it is
        # collective), rank 0's profile is quoted - of its last fnn_patch_features call, i.e. its interior patches
        predictor._engine.set_profiling(True)
        out = step_fn()
        del out
        torch.cuda.synchronize()
        pr = predictor._engine.profile()
        import collections
        kernel_counts = dict(collections.Counter(predictor._engine.kernel_log()))   # which variant served every launch of that volume (N > 1: of the last engine call)
        layer_rows = layer_table(predictor._engine, args.batch) if (w['plan'] or args.layers) else None
        predictor._engine.set_profiling(False)
    if rank == 0 and not args.no_roofline:
        achieved = pr.conv_flops / (pr.conv_ms * 1e-3) / 1e12 if pr.conv_ms > 0 else 0.0
        launches = max(1, pr.conv_launches)
        algo_bytes = pr.conv_bytes / launches
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, 'profiles', TRAFFIC_FILE)
        if distributed:
            traffic_src = 'counter traffic is captured for the single-GPU step only'
        elif os.path.isfile(tpath):
            try:                                  # HBM bytes per launch of the same kernel family from separate PMC passes
                tj = json.load(open(tpath))
                if tj.get('csrc_sha256') != csrc_sha256():
                    traffic_src = f'profiles/{TRAFFIC_FILE} was captured on other kernel sources (csrc_sha256 differs): stale, not quoted'
                elif traffic_key(args) not in tj.get('workloads', {}):
                    traffic_src = f'profiles/{TRAFFIC_FILE} holds no capture of {traffic_key(args)}'
                else:
                    fam = tj['workloads'][traffic_key(args)]['families']['conv3d_mfma']
                    traffic = int(fam['bytes_per_launch'])
                    traffic_src = (f'profiles/{TRAFFIC_FILE}[{traffic_key(args)}]: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this '
                                   f'command on these kernel sources (reads x2 as MI355X_MICROARCH.md prescribes); not measured in this run')
            except Exception as ex:
                traffic_src = f'profiles/{TRAFFIC_FILE} unreadable: {ex}'
        result['roofline'] = {
            'kernel': 'MFMA conv family, as launched in the profiled step: ' + ', '.join(conv_family_names(kernel_counts)), 'bound': 'mfma',
            'achieved': round(achieved, 2), 'peak': MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
            'frac': round(achieved / MFMA_PEAK_TFLOPS, 4),
            # the shader clock the device held during most of one extra step identical to the timed ones (fnn_clock_probe_*:
            # s_memtime over s_memrealtime on one sleeping wave); `peak` assumes 2.4 GHz, `frac_at_clock` prices the same work
            # at the clock held (the profiled step that `achieved` comes from runs one stream and may hold a slightly other one)
            'clock_ghz': round(clock['ghz'], 3) if clock.get('ghz') else None,
            'clock_note': clock.get('rejected'),
            'clock_sampled_s': round(clock['seconds'], 3) if clock.get('seconds') else None,
            'frac_at_clock': round(achieved / (MFMA_PEAK_TFLOPS * clock['ghz'] / MFMA_PEAK_CLOCK_GHZ), 4) if clock.get('ghz') else None,
            'traffic': traffic, 'traffic_source': traffic_src,
            'algorithmic_bytes': int(algo_bytes),
            'traffic_over_algorithmic': round(traffic / algo_bytes, 3) if traffic and algo_bytes else None,
            'hbm_gbps_algorithmic': round(pr.conv_bytes / (pr.conv_ms * 1e-3) / 1e9, 1) if pr.conv_ms > 0 else None,
            'algorithmic_act_bytes_per_patch': int(act_bytes_patch),
            'launches': int(pr.conv_launches),
            'avg_launch_us': round(pr.conv_ms * 1e3 / launches, 2),
            'flop_per_launch': round(pr.conv_flops / launches / 1e9, 3),
            'flop_unit': 'GFLOP (2*MACs of the conv layers, SURVEY.md App. D method)',
            'time_share_ms': {'conv3d_mfma': round(pr.conv_ms, 2), 'stem': round(pr.stem_ms, 2),
                              'tconv': round(pr.tconv_ms, 2), 'seg_head_accumulate': round(pr.head_ms, 2),
                              'finalize': round(pr.finalize_ms, 2)},
            'whole_net_tflops': round(flops_patch * n_patches * args.folds / (dt / args.steps) / 1e12, 2),
            # `achieved` / `frac` describe the PROFILED step (one stream, events around every launch); `value` the timed steps
            # (four batches in flight on the engine's streams).  Both schedules side by side:
            'schedules': {'profiled_step_kernel_ms_sum': round(pr.total_ms, 2), 'profiled_step_family_ms': round(pr.conv_ms, 2),
                          'timed_step_ms': round(dt / args.steps * 1e3, 2),
                          'hidden_by_batches_in_flight': round(1.0 - (dt / args.steps * 1e3) / pr.total_ms, 4) if pr.total_ms > 0 else None,
                          'family_tflops_if_scaled_to_timed_step': round(achieved * pr.total_ms / (dt / args.steps * 1e3), 2) if pr.total_ms > 0 else None},
            'profiled': 'rank 0, its interior patches (the last fnn_patch_features call of one extra step)' if distributed else 'one extra volume on one stream',
            'launches_by_kernel': kernel_counts,
        }
        if layer_rows is not None:
            result['layers'] = layer_rows
    if rank == 0 and not distributed and not args.no_cpu_baseline:
        result['cpu_baseline'] = cpu_baseline(sd, info)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and not distributed and not args.no_also and not w['plan'] and args.workload == 'bone_turbo_r2' and args.folds == 1 \
            and args.dtype == 'f16' and not args.mirror and args.accum == 'fp16' and args.volume == 512:
        predictor._engine.close()                               # (every figure of this process is in `result` by now: its HBM goes back first)
        predictor._engine = None
        del vol
        result['also'] = also_block(device)
    if rank == 0:                                              # last, so that the JSON is the last line on stdout
        sys.stderr.flush()
        print(json.dumps(result), flush=True)


if __name__ == '__main__':
    main()
