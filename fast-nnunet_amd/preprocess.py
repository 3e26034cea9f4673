"""Whole-volume steps on either side of the sliding window, on the device.

Mirrors, with the reference's argument names and property keys,

* ``DefaultPreprocessor.run_case_npy`` (preprocessing/preprocessors/default_preprocessor.py:45-118) up to - and
  excluding - the resampling call: float32 copy, ``transpose_forward``, ``crop_to_nonzero``, per-channel intensity
  normalisation, then ``resample_data_or_seg_to_shape`` to the configuration's spacing
  (preprocessing/resampling/default_resampling.py:88-196: per-channel order-3 ``resize``, or per-slice resize + an
  order-0 pass along the anisotropic axis);
* ``convert_predicted_logits_to_segmentation_with_correct_shape`` (inference/export_prediction.py:16-53): logits
  resampled back to ``shape_after_cropping_and_before_resampling`` (order 1 by default), label rule, dtype rule,
  revert cropping, ``transpose_backward``.

Every numerical step is a HIP kernel behind ``include/fnn.h`` (``fnn_nonzero_bbox``, ``fnn_preprocess``,
``fnn_resample``, ``fnn_argmax_labels``, ``fnn_revert_labels``); torch only owns the device buffers.  No CPU path.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import capi

_SCHEMES = {'NoNormalization': capi.FNN_NORM_NONE, 'ZScoreNormalization': capi.FNN_NORM_ZSCORE,
            'CTNormalization': capi.FNN_NORM_CT, 'RescaleTo01Normalization': capi.FNN_NORM_RESCALE01,
            'RGBTo01Normalization': capi.FNN_NORM_RGB01}


ANISO_THRESHOLD = 3           # nnunetv2/configuration.py


def determine_do_sep_z_and_axis(force_separate_z, current_spacing, new_spacing,
                                separate_z_anisotropy_threshold: float = ANISO_THRESHOLD):
    """preprocessing/resampling/default_resampling.py:14-71."""
    def aniso(sp):
        return (np.max(sp) / np.min(sp)) > separate_z_anisotropy_threshold

    def lowres_axis(sp):
        return np.where(max(sp) / np.array(sp) == 1)[0]

    if force_separate_z is not None:
        do_separate_z = force_separate_z
        axis = lowres_axis(current_spacing) if force_separate_z else None
    elif aniso(current_spacing):
        do_separate_z, axis = True, lowres_axis(current_spacing)
    elif aniso(new_spacing):
        do_separate_z, axis = True, lowres_axis(new_spacing)
    else:
        do_separate_z, axis = False, None
    if axis is not None:
        if len(axis) in (2, 3):
            do_separate_z, axis = False, None
        else:
            axis = int(axis[0])
    return do_separate_z, axis


def compute_new_shape(old_shape: Sequence[int], old_spacing: Sequence[float], new_spacing: Sequence[float]):
    """preprocessing/resampling/default_resampling.py:25-31 (python ``round``: half to even)."""
    assert len(old_spacing) == len(old_shape) and len(old_shape) == len(new_spacing)
    return [int(round(i / j * k)) for i, j, k in zip(old_spacing, new_spacing, old_shape)]


class DevicePreprocessor:
    def __init__(self, device: torch.device = torch.device('cuda'), verbose: bool = False):
        if device.type != 'cuda':
            raise RuntimeError('DevicePreprocessor has no CPU path: pass a GPU device')
        self.device = device
        self.verbose = verbose

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    @torch.inference_mode()
    def run_case_npy(self, data: Union[np.ndarray, torch.Tensor], seg, properties: dict, plans_manager,
                     configuration_manager, dataset_json=None) -> Tuple[torch.Tensor, Optional[torch.Tensor], dict]:
        """-> (float32 ``[C, x, y, z]`` on the device, seg, properties) - the ``data`` the predictor consumes.
        ``seg`` (the previous stage's segmentation of a cascade, ``[1, s0, s1, s2]``) travels with the image: transposed,
        cropped to the same box and resampled with ``resampling_fn_seg`` (int16 on the device), so that it stays aligned
        (inference/data_iterators.py:195-201); the -1 the reference writes outside the non-zero mask (cropping.py:36) is
        not produced - it only ever meets ``convert_labelmap_to_one_hot`` over the foreground labels."""
        seg_in = seg
        with torch.cuda.device(self.device):
            raw = torch.as_tensor(data).to(device=self.device, dtype=torch.float32).contiguous()   # :49 astype(float32)
            assert raw.ndim == 4, 'data must have shape (C, X, Y, Z)'
            tf = [int(i) for i in plans_manager.transpose_forward]
            original_spacing = [properties['spacing'][i] for i in tf]
            shape_t = [int(raw.shape[1 + i]) for i in tf]
            properties['shape_before_cropping'] = tuple(shape_t)
            bbox = capi.nonzero_bbox(raw.data_ptr(), raw.shape, tf, self._stream())
            properties['bbox_used_for_cropping'] = bbox
            cropped = [hi - lo for lo, hi in bbox]
            properties['shape_after_cropping_and_before_resampling'] = tuple(cropped)
            target_spacing = list(configuration_manager.spacing)
            if len(target_spacing) < 3:                     # 2d configurations keep the slice spacing (:74-77)
                target_spacing = [original_spacing[0]] + target_spacing
            new_shape = compute_new_shape(cropped, original_spacing, target_spacing)
            schemes = configuration_manager.normalization_schemes
            masks = configuration_manager.use_mask_for_norm
            props = plans_manager.foreground_intensity_properties_per_channel
            norms = []
            for c in range(raw.shape[0]):
                if schemes[c] not in _SCHEMES:
                    raise RuntimeError(f"Unable to locate class '{schemes[c]}' for normalization")
                ip = props.get(str(c), {}) if props else {}
                if schemes[c] == 'CTNormalization':
                    assert ip, 'CTNormalization requires intensity properties'
                    norms.append((_SCHEMES[schemes[c]], ip['mean'], ip['std'], ip['percentile_00_5'], ip['percentile_99_5']))
                else:
                    norms.append((_SCHEMES[schemes[c]], 0., 1., 0., 0., int(bool(masks[c]))))
            out = torch.empty((raw.shape[0], *cropped), dtype=torch.float32, device=self.device)
            capi.preprocess(raw.data_ptr(), raw.shape, tf, bbox, norms, out.data_ptr(), self._stream())
            # normalisation happens before resampling, like the reference (:83-91)
            out = self.resample(out, new_shape, original_spacing, target_spacing,
                                getattr(configuration_manager, 'resampling_fn_data_kwargs', None) or
                                {'is_seg': False, 'order': 3, 'order_z': 0, 'force_separate_z': None})
            seg_out = None
            if seg_in is not None:
                sg = torch.as_tensor(seg_in).to(self.device)
                assert sg.ndim == 4 and tuple(sg.shape[1:]) == tuple(raw.shape[1:]), 'seg must have shape (1, X, Y, Z) of the image'
                sg = sg.permute(0, *[1 + i for i in tf])
                sg = sg[(slice(None), *[slice(lo, hi) for lo, hi in bbox])].to(torch.int16).contiguous()
                seg_out = self.resample_seg(sg, new_shape, original_spacing, target_spacing,
                                            getattr(configuration_manager, 'resampling_fn_seg_kwargs', None) or
                                            {'is_seg': True, 'order': 1, 'order_z': 0, 'force_separate_z': None})
        return out, seg_out, properties

    @torch.inference_mode()
    def resample_seg(self, seg: torch.Tensor, new_shape, current_spacing, new_spacing, kwargs: dict) -> torch.Tensor:
        """``resample_data_or_seg_to_shape(seg, ..., is_seg=True, order, order_z=0)`` (default_resampling.py:113-196)
        with ``resize_segmentation`` (batchgenerators): order 0 resizes the label image; otherwise every label's mask
        is resized (``fnn_resample``, the images' kernel) and the voxels where it reaches 0.5 take the label, labels
        ascending.  Thresholding commutes with the nearest-neighbour pass along an anisotropic axis, so the per-slice
        path is the same call with ``separate_axis``.  ``[C, x, y, z]`` integer tensor -> int16 on the device."""
        if int(kwargs.get('order_z', 0)) != 0:
            raise NotImplementedError('order_z != 0 for segmentations (the reference\'s plans use 0)')
        new_shape = [int(i) for i in new_shape]
        with torch.cuda.device(self.device):
            sg = seg.to(self.device)
            if [int(i) for i in sg.shape[1:]] == new_shape:
                return sg.to(torch.int16)
            order = int(kwargs.get('order', 1))
            kw = dict(kwargs, is_seg=False)
            if order == 0:
                return self.resample(sg.float(), new_shape, current_spacing, new_spacing, kw).round().to(torch.int16)
            out = torch.zeros((sg.shape[0], *new_shape), dtype=torch.int16, device=self.device)
            for c in torch.unique(sg).tolist():                              # ascending, like np.unique
                m = self.resample((sg == c).float(), new_shape, current_spacing, new_spacing, kw)
                out[m >= 0.5] = int(c)
        return out

    @torch.inference_mode()
    def resample(self, data: torch.Tensor, new_shape, current_spacing, new_spacing, kwargs: dict) -> torch.Tensor:
        """``resample_data_or_seg_to_shape(data, new_shape, current_spacing, new_spacing, **kwargs)`` for images /
        logits (``is_seg`` False): fp32 or fp16 ``[C, x, y, z]`` on the device."""
        if kwargs.get('is_seg', False):
            raise ValueError('is_seg=True: call resample_seg')
        do_sep, axis = determine_do_sep_z_and_axis(kwargs.get('force_separate_z', None), current_spacing, new_spacing,
                                                   kwargs.get('separate_z_anisotropy_threshold', ANISO_THRESHOLD))
        with torch.cuda.device(self.device):
            x = data.to(self.device)
            if x.dtype not in (torch.float32, torch.half):
                x = x.float()
            x = x.contiguous()
            out = torch.empty((x.shape[0], *[int(i) for i in new_shape]), dtype=x.dtype, device=self.device)
            capi.resample(x.data_ptr(), x.shape, new_shape, kwargs.get('order', 3), axis if do_sep else None,
                          x.dtype == torch.half, out.data_ptr(), self._stream(), order_z=kwargs.get('order_z', 0))
        return out

    @torch.inference_mode()
    def convert_predicted_logits_to_segmentation_with_correct_shape(self, predicted_logits: torch.Tensor, predictor,
                                                                    plans_manager, configuration_manager,
                                                                    properties_dict: dict) -> torch.Tensor:
        """inference/export_prediction.py:16-53 on the device: logits ``[heads, x, y, z]`` of the network grid ->
        label map on the original image grid.  ``predictor`` supplies the label rule (its ``label_manager``)."""
        spacing_transposed = [properties_dict['spacing'][i] for i in plans_manager.transpose_forward]
        target = list(configuration_manager.spacing)
        current_spacing = target if len(target) == len(properties_dict['shape_after_cropping_and_before_resampling']) \
            else [spacing_transposed[0], *target]
        kw = getattr(configuration_manager, 'resampling_fn_probabilities_kwargs', None) or \
            {'is_seg': False, 'order': 1, 'order_z': 0, 'force_separate_z': None}
        logits = self.resample(predicted_logits, properties_dict['shape_after_cropping_and_before_resampling'],
                               current_spacing, spacing_transposed, kw)
        seg = predictor.convert_logits_to_segmentation(logits)
        return self.revert_labels(seg, properties_dict, plans_manager, predictor.label_manager)

    @torch.inference_mode()
    def convert_predicted_logits_to_segmentation_and_probabilities(self, predicted_logits: torch.Tensor, predictor,
                                                                   plans_manager, configuration_manager,
                                                                   properties_dict: dict) -> Tuple[torch.Tensor, torch.Tensor]:
        """export_prediction.py:16-70 with ``return_probabilities=True``: -> (label map, float32 probabilities
        ``[heads, s0, s1, s2]``), both on the original image grid, both on the device."""
        spacing_transposed = [properties_dict['spacing'][i] for i in plans_manager.transpose_forward]
        target = list(configuration_manager.spacing)
        cropped = [int(i) for i in properties_dict['shape_after_cropping_and_before_resampling']]
        current_spacing = target if len(target) == len(cropped) else [spacing_transposed[0], *target]
        kw = getattr(configuration_manager, 'resampling_fn_probabilities_kwargs', None) or \
            {'is_seg': False, 'order': 1, 'order_z': 0, 'force_separate_z': None}
        logits = predicted_logits
        if [int(i) for i in logits.shape[1:]] != cropped:
            logits = self.resample(logits, cropped, current_spacing, spacing_transposed, kw)
        order, u16 = predictor._label_rule()
        with torch.cuda.device(self.device):
            lg = logits.to(self.device)
            if lg.dtype not in (torch.half, torch.float32):
                lg = lg.float()
            lg = lg.contiguous()
            before = [int(i) for i in properties_dict['shape_before_cropping']]
            tb = [int(i) for i in plans_manager.transpose_backward]
            grid = [before[j] for j in tb]
            probs = torch.empty((lg.shape[0], *grid), dtype=torch.float32, device=self.device)
            labels = torch.empty(grid, dtype=torch.int16 if u16 else torch.uint8, device=self.device)
            capi.export_probabilities(lg.data_ptr(), lg.dtype == torch.half, lg.shape[0], order,
                                      properties_dict['bbox_used_for_cropping'], before, tb, probs.data_ptr(),
                                      labels.data_ptr(), u16, self._stream())
            if u16:
                labels = labels.to(torch.int32) & 0xffff
        return labels, probs

    @torch.inference_mode()
    def revert_labels(self, segmentation: torch.Tensor, properties: dict, plans_manager, label_manager) -> torch.Tensor:
        """Cropped label map (uint8, or the int32-carried uint16 of the predictor) -> the original image grid."""
        u16 = len(label_manager.foreground_labels) >= 255                      # export_prediction.py:45-46
        with torch.cuda.device(self.device):
            seg = segmentation.to(self.device)
            seg = (seg.to(torch.int16) if u16 else seg.to(torch.uint8)).contiguous()
            before = [int(i) for i in properties['shape_before_cropping']]
            tb = [int(i) for i in plans_manager.transpose_backward]
            out = torch.empty([before[j] for j in tb], dtype=seg.dtype, device=self.device)
            capi.revert_labels(seg.data_ptr(), u16, properties['bbox_used_for_cropping'], before, tb, out.data_ptr(),
                               self._stream())
            if u16:
                out = out.to(torch.int32) & 0xffff
        return out
