"""The step as a reference caller sees it: the preprocessed volume is a CPU tensor (the reference's
``predict_sliding_window_return_logits`` receives what the preprocessing iterator yields, data_iterators.py:116-117, and moves
it with ``data.to(results_device)``, predict_from_raw_data.py:579).  The engine uploads a host volume in tiles (planes x rows) on a copy
stream and starts a batch when the slabs under its patches have landed (engine.hip: stage_volume / upload_until); the result
must be the resident volume's, bit for bit - pageable and pinned, one and many channels, many slabs per volume, with
mirroring, folds, the label entry point, and a volume smaller than the patch (padded after a whole upload)."""
import os

import numpy as np
import pytest
import torch

from golden_cases import toy_unet_spec
from oracle.topology import UNetSpec
from oracle.unet import synthetic_state_dict
from test_gpu_predictor import _bits, _predictor

pytestmark = pytest.mark.gpu


@pytest.fixture()
def small_slabs():
    """Slabs of 64 KiB: a 40 x 48 x 80 volume travels in ~10 of them (the default, 16 MiB, would make it one)."""
    old = os.environ.get('FNN_UPLOAD_SLAB_BYTES')
    os.environ['FNN_UPLOAD_SLAB_BYTES'] = str(64 << 10)
    yield
    if old is None:
        os.environ.pop('FNN_UPLOAD_SLAB_BYTES', None)
    else:
        os.environ['FNN_UPLOAD_SLAB_BYTES'] = old


CASES = [
    ('toy3', toy_unet_spec(1, 3), (16, 16, 32), (40, 48, 80), None, 1),
    ('toy3_mirror', toy_unet_spec(1, 3), (16, 16, 32), (40, 33, 70), (0, 1, 2), 1),
    ('toy_2ch_folds', toy_unet_spec(2, 2), (16, 32, 16), (56, 40, 40), None, 2),
    ('in11', UNetSpec('plain', 11, 3, [16, 32], [(3, 3, 3)] * 2, [(1, 1, 1), (2, 2, 2)], [2, 2], [2]), (16, 16, 32), (48, 24, 40), None, 1),
    ('smaller_than_patch', toy_unet_spec(1, 3), (16, 16, 32), (12, 20, 20), None, 1),
]


@pytest.mark.parametrize('name,spec,patch,shape,mirror,folds', CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize('pinned', [False, True], ids=['pageable', 'pinned'])
def test_host_volume_gives_the_resident_volumes_logits_bit_for_bit(small_slabs, name, spec, patch, shape, mirror, folds, pinned):
    sds = [synthetic_state_dict(spec, 900 + f) for f in range(folds)]
    p = _predictor(spec, patch, sds, mirror=mirror, batch=4)
    vol = torch.randn((spec.in_channels, *shape), generator=torch.Generator().manual_seed(5))
    host = vol.pin_memory() if pinned else vol
    assert host.device.type == 'cpu' and host.is_pinned() == pinned
    if folds > 1:
        ref = p.predict_logits_from_preprocessed_data(vol.cuda(), on_device=True)
        got = p.predict_logits_from_preprocessed_data(host, on_device=True)
    else:
        ref = p.predict_sliding_window_return_logits(vol.cuda())
        got = p.predict_sliding_window_return_logits(host)
    assert got.is_cuda and got.shape == ref.shape
    assert np.array_equal(_bits(got), _bits(ref))
    # and again: the staging ring and the slab events are reused across calls
    got2 = p.predict_sliding_window_return_logits(host) if folds == 1 else p.predict_logits_from_preprocessed_data(host, on_device=True)
    assert np.array_equal(_bits(got2), _bits(ref))
    lab_ref = p.predict_segmentation_from_preprocessed_data(vol.cuda())
    lab = p.predict_segmentation_from_preprocessed_data(host)
    assert torch.equal(lab, lab_ref)


def test_host_volume_with_default_slab_size_and_a_volume_of_several_slabs():
    """No knob: 16 MiB slabs; 1 x 160 x 192 x 192 floats = 22.5 MiB travel in two."""
    spec, patch = toy_unet_spec(1, 3), (16, 16, 32)
    sd = synthetic_state_dict(spec, 11)
    p = _predictor(spec, patch, [sd], batch=32)
    vol = torch.randn((1, 160, 192, 192), generator=torch.Generator().manual_seed(6))
    ref = p.predict_sliding_window_return_logits(vol.cuda())
    for host in (vol, vol.pin_memory()):
        got = p.predict_sliding_window_return_logits(host)
        assert np.array_equal(_bits(got), _bits(ref))


def test_host_volume_through_the_accumulate_path_and_fp32_input_dtypes(small_slabs):
    """FNN_NO_GATHER-style accumulate path (fp32 accumulators select it) and a float64 CPU tensor (converted on the host)."""
    spec, patch = toy_unet_spec(1, 3), (16, 16, 32)
    sd = synthetic_state_dict(spec, 12)
    p = _predictor(spec, patch, [sd], batch=4, accumulate_in='fp32')
    vol = torch.randn((1, 40, 40, 72), generator=torch.Generator().manual_seed(7))
    ref = p.predict_sliding_window_return_logits(vol.cuda())
    got = p.predict_sliding_window_return_logits(vol.double())
    assert np.array_equal(_bits(got), _bits(ref))


def test_host_volume_with_the_gather_ring_and_with_the_gather_off(small_slabs, monkeypatch):
    """The ring form of the gather path (one run_patches call per x layer, a gather launch per output slab: FNN_GATHER_RING) and the
    accumulate path (FNN_NO_GATHER, read when the engine is created) take their patches from a volume that is still arriving."""
    spec, patch = toy_unet_spec(1, 3), (16, 16, 32)
    sd = synthetic_state_dict(spec, 13)
    vol = torch.randn((1, 72, 40, 72), generator=torch.Generator().manual_seed(8))
    p = _predictor(spec, patch, [sd], batch=4)
    ref = p.predict_sliding_window_return_logits(vol.cuda())
    monkeypatch.setenv('FNN_GATHER_RING', '2')
    got = p.predict_sliding_window_return_logits(vol)
    assert np.array_equal(_bits(got), _bits(ref))
    got = p.predict_sliding_window_return_logits(vol.pin_memory())
    assert np.array_equal(_bits(got), _bits(ref))
    monkeypatch.delenv('FNN_GATHER_RING')
    monkeypatch.setenv('FNN_NO_GATHER', '1')
    q = _predictor(spec, patch, [sd], batch=4)
    assert np.array_equal(_bits(q.predict_sliding_window_return_logits(vol)), _bits(q.predict_sliding_window_return_logits(vol.cuda())))
    assert np.array_equal(_bits(q.predict_sliding_window_return_logits(vol)), _bits(ref))
