"""CPU-only checks of the product's host side: the C-ABI library loads and exports
every declared symbol, its integer logic matches the reference's golden vectors,
the plans / checkpoint readers agree with the reference, and the engine refuses to
run without a GPU (no silent fallback)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from golden_cases import DATASET_JSONS, PLANS_NEW, PLANS_OLD, toy_unet_spec
from oracle import sliding_window as osw
from oracle.unet import synthetic_state_dict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def capi():
    from fast_nnunet_amd import capi as c
    c.load_library()
    return c


def test_library_exports_every_symbol_in_header(capi):
    header = open(os.path.join(ROOT, 'include', 'fnn.h')).read()
    header = re.sub(r'/\*.*?\*/', '', header, flags=re.S)
    declared = set(re.findall(r'\b(fnn_[a-z0-9_]+)\s*\(', header))
    assert len(declared) >= 18
    lib = capi.load_library()
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in include/fnn.h but not exported'
    assert declared == set(capi.EXPORTS)
    assert lib.fnn_abi_version() == 4


def test_opts_carry_the_step_size_as_a_double(capi):
    """The patch grid is ceil((image - patch) / (patch * step)) with the reference's Python float
    (sliding_window_prediction.py:38-41); a float32 step gives another count on e.g. step 0.7, patch 16, image 72."""
    assert capi.Opts.tile_step_size.size == 8
    header = open(os.path.join(ROOT, 'include', 'fnn.h')).read()
    assert re.search(r'double\s+tile_step_size', header)
    import math
    assert len(capi.compute_steps(72, 16, 0.7)) == math.ceil((72 - 16) / (16 * 0.7)) + 1 == 6
    assert capi.plan_volume((16, 16, 32), (72, 20, 40), 0.7)[2].shape[0] == 6 * 2 * 2


def test_compute_steps_matches_reference_golden(capi, golden_dir):
    from fast_nnunet_amd.sliding_window import compute_steps_for_sliding_window
    for c in json.load(open(os.path.join(golden_dir, 'steps.json'))):
        assert compute_steps_for_sliding_window(c['image'], c['patch'], c['step']) == c['steps'], c


def test_compute_steps_rejects_bad_arguments(capi):
    with pytest.raises(AssertionError):
        capi.compute_steps(10, 16, 0.5)          # image smaller than patch
    with pytest.raises(AssertionError):
        capi.compute_steps(32, 16, 0.0)
    with pytest.raises(AssertionError):
        capi.compute_steps(32, 16, 1.5)


@pytest.mark.parametrize('shape,patch,step', [((40, 36, 44), (16, 16, 16), 0.5), ((11, 30, 9), (16, 16, 16), 0.5),
                                              ((512, 512, 512), (160, 96, 96), 0.5), ((33, 47, 21), (20, 28, 20), 0.3),
                                              ((16, 16, 16), (16, 16, 16), 1.0)])
def test_plan_volume_matches_oracle_slicers(capi, shape, patch, step):
    padded, lo, origins = capi.plan_volume(patch, shape, step)
    pads, undo = osw.pad_to_patch(shape, patch)
    assert lo == [p[0] for p in pads]
    assert padded == [s + p[0] + p[1] for s, p in zip(shape, pads)]
    sl = osw.patch_slicers(padded, patch, step)
    assert origins.shape[0] == len(sl)
    assert origins.tolist() == [[s[1].start, s[2].start, s[3].start] for s in sl]


@pytest.mark.parametrize('shape,patch,step', [((5, 36, 44), (16, 16), 0.5), ((3, 11, 30), (16, 16), 0.5), ((1, 16, 40), (16, 32), 1.0)])
def test_plan_volume_2d_configuration_matches_oracle_slicers(capi, shape, patch, step):
    padded, lo, origins = capi.plan_volume(patch, shape, step)
    pads, _ = osw.pad_to_patch(shape, patch)
    assert lo == [p[0] for p in pads] and lo[0] == 0
    assert padded == [s + p[0] + p[1] for s, p in zip(shape, pads)]
    sl = osw.patch_slicers(padded, patch, step)
    assert origins.tolist() == [[s[1], s[2].start, s[3].start] for s in sl]


def test_spec_from_2d_checkpoint():
    from fast_nnunet_amd.arch import spec_from_state_dict, weight_blob
    from golden_cases import toy_unet_spec_2d
    ospec = toy_unet_spec_2d(2, 3)
    sd = synthetic_state_dict(ospec, 5)
    spec = spec_from_state_dict(sd, (16, 32))
    assert spec.spatial_dims == 2 and tuple(spec.patch) == (1, 16, 32)
    assert [tuple(k) for k in spec.kernels] == [(1, *k) for k in ospec.kernels]
    assert [tuple(s) for s in spec.strides] == [(1, *s) for s in ospec.strides]
    assert weight_blob(spec, sd).size == sum(v.numel() for k, v in sd.items() if 'seg_layers.0' not in k)
    with pytest.raises(RuntimeError, match='does not match'):
        spec_from_state_dict(sd, (16, 16, 32))


def test_patch_counts_of_the_benchmark_configs(capi):
    # 343 (128^3), 216 (160^3), 600 (160x96x96) patches on a 512^3 volume
    for patch, n in (((128,) * 3, 343), ((160,) * 3, 216), ((160, 96, 96), 600)):
        assert capi.plan_volume(patch, (512,) * 3, 0.5)[2].shape[0] == n


def test_gaussian_matches_reference_golden(golden_dir):
    from fast_nnunet_amd.sliding_window import compute_gaussian
    z = np.load(os.path.join(golden_dir, 'gaussian.npz'))
    for key in z.files:
        patch = tuple(int(i) for i in key.split('_')[1:])
        g = compute_gaussian(patch, sigma_scale=1. / 8, value_scaling_factor=10, device=torch.device('cpu'))
        assert np.array_equal(g.view(torch.int16).numpy().view(np.uint16), z[key])


def test_engine_fails_loudly_without_gpu(capi):
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    from fast_nnunet_amd.arch import spec_from_state_dict
    spec = spec_from_state_dict(synthetic_state_dict(toy_unet_spec(1, 3)), (16, 16, 32))
    with pytest.raises(RuntimeError, match='no HIP device|no CPU fallback'):
        capi.Engine(spec.to_desc(), 0, 2)
    from fast_nnunet_amd import nnUNetPredictor
    with pytest.raises(RuntimeError, match='no CPU path'):
        nnUNetPredictor(device=torch.device('cpu'))


def test_preprocess_entry_points_fail_loudly_on_host_memory(capi):
    raw = np.ones((1, 4, 4, 4), np.float32)
    with pytest.raises(AssertionError, match='device pointer'):
        capi.nonzero_bbox(raw.ctypes.data, raw.shape, (0, 1, 2))
    from fast_nnunet_amd.preprocess import DevicePreprocessor, compute_new_shape
    with pytest.raises(RuntimeError, match='no CPU path'):
        DevicePreprocessor(torch.device('cpu'))
    # compute_new_shape (default_resampling.py:25-31): python round, half to even
    assert compute_new_shape((100, 51, 7), (1.0, 1.0, 2.5), (2.0, 1.0, 1.0)) == [50, 51, 18]
    assert compute_new_shape((5,), (1.0,), (2.0,)) == [2]


def test_spec_from_state_dict_roundtrip_and_aliases():
    from fast_nnunet_amd.arch import spec_from_state_dict, weight_blob
    ospec = toy_unet_spec(2, 4)
    sd = synthetic_state_dict(ospec, 5)
    # add what real checkpoints carry: wrapper prefixes and aliased duplicates
    noisy = {}
    for k, v in sd.items():
        noisy['module._orig_mod.' + k] = v
        if '.conv.' in k:
            noisy['module._orig_mod.' + k.replace('.conv.', '.all_modules.0.')] = v
        if k.startswith('encoder.'):
            noisy['module._orig_mod.decoder.' + k] = v
    spec = spec_from_state_dict(noisy, (16, 16, 32))
    assert spec.features == ospec.features and spec.in_channels == 2 and spec.num_heads == 4
    assert [tuple(k) for k in spec.kernels] == [tuple(k) for k in ospec.kernels]
    assert [tuple(s) for s in spec.strides] == [tuple(s) for s in ospec.strides]
    assert spec.n_conv_enc == ospec.n_conv_enc and spec.n_conv_dec == ospec.n_conv_dec
    a, b = weight_blob(spec, noisy), weight_blob(spec, sd)
    assert np.array_equal(a, b)
    n_params = sum(v.numel() for k, v in sd.items() if 'seg_layers.0' not in k)
    assert a.size == n_params


def test_checkpoints_with_other_operators_are_refused():
    """A BatchNorm / ReLU network must not load silently: the engine hard-wires affine InstanceNorm + LeakyReLU and
    takes eps and the slope from the plans (the reference builds whatever the plans name, get_network_from_plans.py:9-43)."""
    from fast_nnunet_amd.arch import ops_from_plans, spec_from_state_dict, weight_blob
    ok = {'norm_op': 'torch.nn.modules.instancenorm.InstanceNorm3d', 'norm_op_kwargs': {'eps': 1e-4, 'affine': True},
          'nonlin': 'torch.nn.LeakyReLU', 'nonlin_kwargs': {'inplace': True, 'negative_slope': 0.2},
          'dropout_op': None, 'dropout_op_kwargs': None}
    assert ops_from_plans(ok) == (1e-4, 0.2)
    assert ops_from_plans({}) == (1e-5, 0.01)
    assert ops_from_plans({'norm_op': torch.nn.InstanceNorm2d, 'nonlin': torch.nn.LeakyReLU}) == (1e-5, 0.01)
    for bad in ({'norm_op': 'torch.nn.modules.batchnorm.BatchNorm3d'}, {'nonlin': 'torch.nn.ReLU'},
                {'norm_op': 'torch.nn.InstanceNorm3d', 'norm_op_kwargs': {'affine': False}},
                {'nonlin': 'torch.nn.LeakyReLU', 'nonlin_kwargs': {'negative_slope': 1.5}}):
        with pytest.raises(NotImplementedError):
            ops_from_plans({**ok, **bad})
    ospec = toy_unet_spec(1, 2)
    sd = dict(synthetic_state_dict(ospec, 5))
    spec = spec_from_state_dict(sd, (16, 16, 32))
    weight_blob(spec, sd)
    sd['encoder.stages.0.0.convs.0.norm.running_mean'] = torch.zeros(ospec.features[0])
    with pytest.raises(NotImplementedError, match='running_mean'):
        weight_blob(spec, sd)


def test_resenc_checkpoint_topology_and_blob():
    """ResidualEncoderUNet students carry a `network.` prefix (nnUNetDistillationTrainer.py:248) and a skip
    Sequential whose projection index depends on the presence of the AvgPool."""
    from fast_nnunet_amd import capi
    from fast_nnunet_amd.arch import spec_from_state_dict, weight_blob
    from oracle.topology import UNetSpec
    ospec = UNetSpec('resenc', 2, 3, [8, 16, 16, 24], [(1, 3, 3), (3, 3, 3), (3, 3, 3), (3, 3, 3)],
                     [(1, 1, 1), (1, 2, 2), (2, 2, 2), (2, 1, 1)], [1, 3, 2, 2], [1, 1, 1])
    sd = {'network.' + k: v for k, v in synthetic_state_dict(ospec, 2).items()}
    spec = spec_from_state_dict(sd, (16, 16, 16))
    assert spec.kind == capi.FNN_NET_RESENC
    assert spec.features == ospec.features and spec.n_conv_enc == ospec.n_conv_enc and spec.n_conv_dec == ospec.n_conv_dec
    assert [tuple(s) for s in spec.strides] == [tuple(s) for s in ospec.strides]
    assert [tuple(k) for k in spec.kernels] == [tuple(k) for k in ospec.kernels]
    n_params = sum(v.numel() for k, v in sd.items() if not any(f'seg_layers.{i}.' in k for i in range(2)))
    assert weight_blob(spec, sd).size == n_params


def test_plans_reader_matches_reference(golden_dir):
    import copy
    import warnings
    from fast_nnunet_amd.plans import PlansManager, determine_num_input_channels
    expected = json.load(open(os.path.join(golden_dir, 'plans_expected.json')))
    for tag, plans in (('new', PLANS_NEW), ('old', PLANS_OLD)):
        pm = PlansManager(copy.deepcopy(plans))
        assert set(pm.available_configurations) == set(expected[tag])
        for cfg, exp in expected[tag].items():
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                cm = pm.get_configuration(cfg)
            assert list(cm.patch_size) == exp['patch_size']
            assert cm.network_arch_class_name == exp['network_class_name']
            assert json.loads(json.dumps(cm.network_arch_init_kwargs)) == exp['arch_kwargs']
            assert json.loads(json.dumps(cm.pool_op_kernel_sizes)) == exp['pool_op_kernel_sizes']
            assert cm.previous_stage_name == exp['previous_stage']
            for name, dj in DATASET_JSONS.items():
                assert pm.get_label_manager(dj).num_segmentation_heads == exp[f'heads__{name}']
                assert determine_num_input_channels(pm, cm, dj) == exp[f'cin__{name}']


def test_plans_errors():
    from fast_nnunet_amd.plans import PlansManager, LabelManager
    pm = PlansManager({'configurations': {'a': {'inherits_from': 'b'}, 'b': {'inherits_from': 'a'}}})
    with pytest.raises(RuntimeError):
        pm.get_configuration('a')
    with pytest.raises(RuntimeError):
        pm.get_configuration('zzz')
    with pytest.raises(RuntimeError):
        LabelManager({'a': 1}, None)


def test_student_reduction_rule():
    from fast_nnunet_amd.arch import spec_from_plans
    kw = PLANS_NEW['configurations']['3d_fullres']['architecture']['arch_kwargs']
    assert spec_from_plans('PlainConvUNet', kw, 1, 61, (160, 96, 96), reduction=2).features == [16, 32, 64, 128, 160, 160]
    assert spec_from_plans('PlainConvUNet', kw, 1, 61, (160, 96, 96), reduction=8).features == [8, 8, 16, 32, 40, 40]
    assert spec_from_plans('PlainConvUNet', kw, 1, 61, (160, 96, 96), reduction=1).features == [32, 64, 128, 256, 320, 320]


# ---- engine .ini front-end (keys of the reference's engine/config/fast_nnunet_bone_turbo.ini)
ENGINE_INI = """[model]
file_name = fast_nnunet_bone_turbo.trt
input_name = input
output_name = output
num_class = 61

[input]
depth = 160
height = 96
width = 96
patch_size = 160, 96, 96
target_spacing = 2.0, 0.9765625, 0.9765625

[preprocessing]
mean = 418.6798400878906
std_dev = 412.1883239746094
lower_bound = -60.0
upper_bound = 3068.0

[inference]
use_mirroring = false
step_size = 0.5
use_gaussian = true
"""


def test_engine_ini_is_read_like_the_reference_config(tmp_path):
    from fast_nnunet_amd.engine_config import load_engine_config, plans_with_engine_config
    from fast_nnunet_amd.plans import PlansManager
    path = tmp_path / 'fast_nnunet_bone_turbo.ini'
    path.write_text(ENGINE_INI)
    cfg = load_engine_config(str(path))
    assert cfg.num_class == 61 and cfg.patch_size == [160, 96, 96]
    assert cfg.target_spacing == [2.0, 0.9765625, 0.9765625]
    assert (cfg.mean, cfg.std_dev, cfg.lower_bound, cfg.upper_bound) == (418.6798400878906, 412.1883239746094, -60.0, 3068.0)
    assert cfg.use_mirroring is False and cfg.use_gaussian is True and cfg.step_size == 0.5
    # the ini overrides what the plans say about spacing and normalisation, and nothing else
    plans = {'configurations': {'3d_fullres': {'patch_size': [160, 96, 96], 'spacing': [1.0, 1.0, 1.0],
                                               'normalization_schemes': ['ZScoreNormalization'], 'batch_size': 2,
                                               'architecture': {'network_class_name': 'PlainConvUNet', 'arch_kwargs': {},
                                                                '_kw_requires_import': []}}},
             'foreground_intensity_properties_per_channel': {'0': {'mean': 1.0, 'std': 2.0, 'median': 7.0}}}
    pm = PlansManager(plans_with_engine_config(plans, '3d_fullres', cfg))
    cm = pm.get_configuration('3d_fullres')
    assert cm.spacing == cfg.target_spacing and cm.normalization_schemes == ['CTNormalization'] and cm.batch_size == 2
    ip = pm.foreground_intensity_properties_per_channel['0']
    assert (ip['mean'], ip['std'], ip['percentile_00_5'], ip['percentile_99_5'], ip['median']) == \
        (cfg.mean, cfg.std_dev, cfg.lower_bound, cfg.upper_bound, 7.0)
    assert plans['configurations']['3d_fullres']['spacing'] == [1.0, 1.0, 1.0]          # the input is not mutated


@pytest.mark.parametrize('old, new, err', [
    ('depth = 160', 'depth = 128', 'disagrees with patch_size'),
    ('step_size = 0.5', 'step_size = 0', 'step_size'),
    ('use_gaussian = true', 'use_gaussian = maybe', 'use_gaussian'),
    ('[preprocessing]', '[preprocessing_x]', r'section \[preprocessing\] is missing'),
    ('target_spacing = 2.0, 0.9765625, 0.9765625', 'target_spacing = 2.0, 1.0', 'target_spacing'),
    ('std_dev = 412.1883239746094', 'std_dev = 0', 'std_dev'),
])
def test_engine_ini_errors(tmp_path, old, new, err):
    from fast_nnunet_amd.engine_config import load_engine_config
    assert old in ENGINE_INI
    path = tmp_path / 'bad.ini'
    path.write_text(ENGINE_INI.replace(old, new))
    with pytest.raises(ValueError, match=err):
        load_engine_config(str(path))
    with pytest.raises(FileNotFoundError):
        load_engine_config(str(tmp_path / 'absent.ini'))


def test_engine_front_end_call_order():
    from fast_nnunet_amd.engine_config import Engine
    e = Engine(device=torch.device('cuda'))
    with pytest.raises(RuntimeError, match='set_config'):
        e.set_workspace('/nonexistent')
    with pytest.raises(RuntimeError, match='set_workspace'):
        e.infer(np.zeros((4, 4, 4), np.float32), (1, 1, 1))


def test_bench_gpus_n_spawns_one_rank_per_gpu_and_propagates_failure():
    """`python bench.py --gpus 2` without a launcher starts two ranks before touching a GPU; on this CPU-only box both
    fail (no device), and the parent must report that instead of printing a 1-GPU line."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                         capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip('two GPUs are visible: this is the CPU-box check')
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith('{')]
    # with WORLD_SIZE set by a launcher the script must not spawn again, and must refuse a contradicting --gpus
    env2 = dict(env, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    out2 = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1'],
                          capture_output=True, text=True, timeout=300, env=env2, cwd=ROOT)
    assert out2.returncode != 0 and 'WORLD_SIZE=1' in (out2.stderr + out2.stdout)


def test_fp8_weight_quantiser_matches_torch_e4m3fn(capi):
    """FNN_PREC_F8 packs weights as OCP e4m3 on the host: the encoder must agree with torch's float8_e4m3fn (round to
    nearest even, subnormals down to 2^-9, saturation at +-448) on every value class."""
    g = torch.Generator().manual_seed(3)
    x = torch.cat([torch.randn(200000, generator=g) * 100, torch.randn(200000, generator=g), torch.randn(100000, generator=g) * 0.01,
                   torch.tensor([0.0, -0.0, 448.0, -448.0, 447.9, 464.0, 1e9, -1e9, 2.0 ** -6, 2.0 ** -9, 2.0 ** -10, 1.5 * 2.0 ** -10,
                                 0.0175, 0.0146484375, 240.0, 232.0, 1.0625, 1.1875])])
    # every e4m3 value and every midpoint between neighbours (ties go to the even mantissa)
    allv = torch.arange(0, 127, dtype=torch.uint8).view(torch.float8_e4m3fn).float()
    x = torch.cat([x, allv, -allv, (allv[:-1] + allv[1:]) / 2, -(allv[:-1] + allv[1:]) / 2])
    want = x.clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    got = capi.fp8_e4m3_encode(x.numpy())
    assert np.array_equal(got, want), np.flatnonzero(got != want)[:10]


def test_no_kernel_of_the_library_keeps_scratch():
    """The build leaves hipcc's per-kernel resource report next to every object (csrc/Makefile): every kernel must report
    ScratchSize 0 - a spilling variant is a slow path that only small-shape tests reach."""
    import glob
    import re
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'fast-nnunet_amd', 'csrc')
    reports = sorted(glob.glob(os.path.join(csrc, '*.res')))
    if not reports:
        pytest.skip('no resource reports: the library was not built by csrc/Makefile here')
    n, bad = 0, []
    for path in reports:
        name = None
        for line in open(path, errors='replace'):
            m = re.search(r'Function Name: (\S+)', line)
            if m:
                name = m.group(1)
            m = re.search(r'ScratchSize \[bytes/lane\]: (\d+)', line)
            if m and name:
                n += 1
                if int(m.group(1)) != 0:
                    bad.append((os.path.basename(path), name, int(m.group(1))))
    assert n >= 150, f'only {n} kernels found in the resource reports'
    assert not bad, f'kernels with scratch: {bad}'


def _gfx950_code_objects(so_path):
    """The device ELFs of a HIP shared library: .hip_fatbin holds one clang offload bundle per translation unit
    (magic, entry count, then per entry offset / size / triple)."""
    import struct
    data = open(so_path, 'rb').read()
    magic, out, pos = b'__CLANG_OFFLOAD_BUNDLE__', [], 0
    while True:
        start = data.find(magic, pos)
        if start < 0:
            return out
        n, = struct.unpack_from('<Q', data, start + len(magic))
        q = start + len(magic) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from('<QQQ', data, q)
            triple = data[q + 24:q + 24 + tlen].decode()
            q += 24 + tlen
            if 'gfx950' in triple and size:
                out.append(data[start + off:start + off + size])
        pos = start + len(magic)


def test_no_wide_buffer_store_of_the_library_has_a_register_soffset(tmp_path):
    """A buffer store of more than 8 bytes whose soffset is an SGPR is exempt from LLVM's "VALU write of store data" wait
    state; on gfx950 the store then picks up a data register that the next instruction overwrites (round 3:
    stem_mfma1_kernel stored the next column block's channel pair in some lanes - conv3d_thin.hip, tools/stem_check.cpp).
    Every 12 / 16-byte buffer store in the library's gfx950 code must therefore carry a literal soffset."""
    import re
    import shutil
    import subprocess
    objdump = shutil.which('llvm-objdump') or '/opt/rocm/lib/llvm/bin/llvm-objdump'
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'fast-nnunet_amd', 'csrc', 'libfnn_hip.so')
    if not os.path.exists(objdump) or not os.path.exists(so):
        pytest.skip('llvm-objdump or the built library is missing')
    objs = _gfx950_code_objects(so)
    assert objs, 'no gfx950 code object found in the library'
    n, bad = 0, []
    for i, blob in enumerate(objs):
        path = tmp_path / f'co{i}.elf'
        path.write_bytes(blob)
        text = subprocess.run([objdump, '-d', '--mcpu=gfx950', str(path)], capture_output=True, text=True, check=True).stdout
        for line in text.splitlines():
            m = re.search(r'\bbuffer_store_dwordx[34]\s+(.*?)(?://|$)', line)
            if not m:
                continue
            n += 1
            ops = [o.strip() for o in m.group(1).split(',')]            # vdata, vaddr, srsrc, "soffset [modifiers]"
            soffset = ops[-1].split()[0]
            if re.fullmatch(r's\d+|m0|vcc_lo|vcc_hi|ttmp\d+', soffset):
                bad.append(line.strip())
    assert n >= 100, f'only {n} wide buffer stores found'
    assert not bad, bad[:5]


def test_committed_counter_traffic_belongs_to_the_kernel_sources_in_the_tree():
    """bench.py quotes `roofline.traffic` from profiles/<round>_traffic.json only while its csrc_sha256 matches the HIP
    sources; a later kernel edit silently turns the figure into `traffic: null`.  This fails instead: re-capture
    (tools/capture.sh) after the last kernel change of a round, or delete the file."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    path = os.path.join(ROOT, 'profiles', bench.TRAFFIC_FILE)
    if not os.path.isfile(path):
        pytest.skip(f'no profiles/{bench.TRAFFIC_FILE} yet')
    doc = json.load(open(path))
    assert doc.get('csrc_sha256') == bench.csrc_sha256(), \
        f'profiles/{bench.TRAFFIC_FILE} was captured on other kernel sources: re-run tools/capture.sh (FNN_ROUND) and commit it'
    assert any(k.startswith('bone_turbo_r2|f16|mirror=0|fp16|') for k in doc.get('workloads', {}))


def test_the_library_reads_its_environment_only_through_the_knob_switch():
    """a production libfnn_hip.so must not change kernels because an unrelated process exported FNN_PIPES.
    Every FNN_* variable of the HIP sources is read through fnn_knob() (csrc/misc.hip), which answers nullptr unless
    FNN_KNOBS is set to something other than 0; the only direct getenv calls are that switch itself and the timing-only
    FNN_ZR_TMODE inside `#ifdef FNN_TMODE` (a diagnostic build that is never shipped).  The Python side reads FNN_LIB behind the
    same switch.  Scans the sources."""
    src_dir = os.path.join(ROOT, 'fast-nnunet_amd', 'csrc')
    direct, knobs = [], set()
    for name in sorted(os.listdir(src_dir)):
        if not name.endswith(('.hip', '.h')):
            continue
        text = open(os.path.join(src_dir, name)).read()
        depth_tmode = 0
        for ln, line in enumerate(text.splitlines(), 1):
            s = line.strip()
            if s.startswith('#ifdef FNN_TMODE'):
                depth_tmode += 1
            elif s.startswith('#endif') and depth_tmode:
                depth_tmode -= 1
            code = line.split('//')[0]
            for m in re.finditer(r'\bgetenv\s*\(\s*("?)([A-Za-z_0-9]*)', code):
                direct.append((name, ln, m.group(2), depth_tmode > 0))
            knobs.update(re.findall(r'fnn_knob\("([A-Z0-9_]+)"\)', code))
    allowed = {('misc.hip', 'FNN_KNOBS'), ('misc.hip', 'name')}
    for name, ln, var, in_tmode in direct:
        assert in_tmode or (name, var) in allowed, f'{name}:{ln} reads {var or "the environment"} without the FNN_KNOBS switch'
    body = open(os.path.join(src_dir, 'misc.hip')).read()
    m = re.search(r'const char \*fnn_knob\(const char \*name\) \{(.*?)\n\}', body, re.S)
    assert m and 'getenv("FNN_KNOBS")' in m.group(1) and 'on ? getenv(name) : nullptr' in m.group(1)
    assert len(knobs) >= 40 and all(k.startswith('FNN_') for k in knobs)
    # the Python binding: FNN_LIB only next to the switch; nothing else of the package reads FNN_* variables
    pkg = os.path.join(ROOT, 'fast-nnunet_amd')
    for name in sorted(os.listdir(pkg)):
        if name.endswith('.py'):
            for var in re.findall(r"environ(?:\.get)?[\(\[]\s*'(FNN_[A-Z0-9_]+)'", open(os.path.join(pkg, name)).read()):
                assert (name, var) in {('capi.py', 'FNN_KNOBS'), ('capi.py', 'FNN_LIB')}, (name, var)
    capi_src = open(os.path.join(pkg, 'capi.py')).read()
    assert "os.environ.get('FNN_LIB') if _KNOBS else None" in capi_src


def test_sweep_plans_get_the_reference_planners_topology(golden_dir):
    """tools/plans/*.json (the plan sweep): bench.py's restatement of get_pool_and_conv_props agrees with the
    oracle's on every plan of the sweep, and both with the reference-made tests/golden/topology.json where a plan's
    (spacing, patch) is one of its cases."""
    import glob
    import sys
    sys.path.insert(0, ROOT)
    import bench
    from oracle.topology import plan_pool_and_kernels as oracle_plan
    golden = {(tuple(c['spacing']), tuple(c['patch'])): c for c in json.load(open(os.path.join(golden_dir, 'topology.json')))}
    plans = sorted(glob.glob(os.path.join(ROOT, 'tools', 'plans', '*.json')))
    assert len(plans) >= 10
    hit = 0
    for pf in plans:
        j = json.load(open(pf))
        strides, kernels = bench.plan_topology(j['spacing'], j['patch'])
        o = oracle_plan(tuple(float(v) for v in j['spacing']), tuple(int(v) for v in j['patch']))
        assert [list(s) for s in o[0]] == strides and [list(k) for k in o[1]] == kernels, pf
        g = golden.get((tuple(float(v) for v in j['spacing']), tuple(int(v) for v in j['patch'])))
        if g is not None:
            hit += 1
            assert g['strides'] == strides and g['kernels'] == kernels, pf
        # the patch is divisible by the total stride (the planner pads it so: network_topology.py:96-108)
        for a in range(len(j['patch'])):
            total = int(np.prod([s[a] for s in strides]))
            assert j['patch'][a] % total == 0, (pf, a)
    assert hit >= 1


def test_patches_per_forward_bound_follows_the_patch_size(capi):
    """fnn_create: 1 .. 64 patches per forward, or - small patches - as many as give a forward 2^27 voxels, at most 512 (round 6: the deep layers
    of a 40 x 56 x 40 plan or a 2-D slice have too few voxels per item).  The bound is checked before any device is touched."""
    from fast_nnunet_amd.arch import spec_from_state_dict
    small = spec_from_state_dict(synthetic_state_dict(toy_unet_spec(1, 3)), (16, 16, 32))        # 8192 voxels: 512 allowed
    with pytest.raises(AssertionError, match=r"max_batch must be 1\.\.512"):
        capi.Engine(small.to_desc(), 0, 513)
    with pytest.raises(AssertionError, match=r"max_batch must be 1\.\.512"):
        capi.Engine(small.to_desc(), 0, 0)
    big = spec_from_state_dict(synthetic_state_dict(toy_unet_spec(1, 3)), (160, 160, 160))      # 4.1 M voxels: 64
    with pytest.raises(AssertionError, match=r"max_batch must be 1\.\.64"):
        capi.Engine(big.to_desc(), 0, 65)
    mid = spec_from_state_dict(synthetic_state_dict(toy_unet_spec(1, 3)), (64, 128, 128))       # 1 M voxels: 2^27 / 2^20 = 128
    with pytest.raises(AssertionError, match=r"max_batch must be 1\.\.128"):
        capi.Engine(mid.to_desc(), 0, 129)
