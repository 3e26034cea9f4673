"""Drop-in ``nnUNetPredictor`` backed by the MI355X HIP engine.

Mirrors the public surface of the reference's
``nnunetv2.inference.predict_from_raw_data.nnUNetPredictor``
(distillation/nnunetv2/inference/predict_from_raw_data.py:39-680) for the hot
path: constructor knobs, ``initialize_from_trained_model_folder``,
``manual_initialization``, ``predict_sliding_window_return_logits``,
``predict_logits_from_preprocessed_data`` and the attributes callers read
(``network``, ``plans_manager``, ``configuration_manager``, ``label_manager``,
``dataset_json``, ``list_of_parameters``, ``allowed_mirroring_axes``,
``device``, ``verbose``).  Everything numerical happens in
``csrc/libfnn_hip.so``; torch is only used for device memory and streams.

There is no CPU path: the predictor raises if the HIP library or a GPU is
missing.
"""
from __future__ import annotations

import itertools
import os
from typing import List, Optional, Tuple, Union

import numpy as np
import torch

from . import capi
from .arch import ArchSpec, check_against_plans, ops_from_plans, spec_from_state_dict, weight_blob
from .plans import ConfigurationManager, PlansManager, determine_num_input_channels
from .sliding_window import compute_gaussian, compute_steps_for_sliding_window


def _load_json(path):
    import json
    with open(path) as f:
        return json.load(f)


class nnUNetPredictor(object):
    def __init__(self,
                 tile_step_size: float = 0.5,
                 use_gaussian: bool = True,
                 use_mirroring: bool = True,
                 perform_everything_on_device: bool = True,
                 device: torch.device = torch.device('cuda'),
                 verbose: bool = False,
                 verbose_preprocessing: bool = False,
                 allow_tqdm: bool = True,
                 accumulate_in: str = 'fp16',
                 patches_per_forward: int = 4,
                 compute_dtype: str = 'f16'):
        """Same knobs as the reference (:40-65) plus two engine choices:

        accumulate_in  'fp16' reproduces the reference's half accumulators and their rounding per patch visit as the
                       reference computes them WITHOUT autocast (its CPU path: fp32 logits and products, one rounding
                       per visit); 'fp16_autocast' as it computes them on a GPU (:591-593 - the network returns fp16,
                       so the logit, the mirror sums, the Gaussian product and the sum are each rounded to fp16);
                       'fp32' is the exact blend.
        patches_per_forward  how many patches one network forward batches.
        compute_dtype  'f16' (default: the mode every parity statement is for) or 'f8': OCP e4m3 operands in
                       the 3x3x3 stride-1 convolutions (BASELINE config 5; budget in DESIGN.md).
        """
        self.verbose = verbose
        self.verbose_preprocessing = verbose_preprocessing
        self.allow_tqdm = allow_tqdm
        self.plans_manager, self.configuration_manager, self.list_of_parameters, self.network, self.dataset_json, \
            self.trainer_name, self.allowed_mirroring_axes, self.label_manager = (None,) * 8
        self.tile_step_size = tile_step_size
        self.use_gaussian = use_gaussian
        self.use_mirroring = use_mirroring
        device = torch.device(device)
        if device.type != 'cuda':
            raise RuntimeError('this predictor runs on an AMD GPU through the HIP engine; there is no CPU path '
                               f'(got device={device}). Use the reference predictor for CPU inference.')
        self.device = device
        self.perform_everything_on_device = perform_everything_on_device
        if accumulate_in not in ('fp16', 'fp32', 'fp16_autocast'):
            raise ValueError("accumulate_in must be 'fp16', 'fp32' or 'fp16_autocast'")
        self.accumulate_in = accumulate_in
        self.patches_per_forward = int(patches_per_forward)
        if compute_dtype not in ('f16', 'f8'):
            raise ValueError("compute_dtype must be 'f16' or 'f8'")
        self.compute_dtype = compute_dtype
        self._engine: Optional[capi.Engine] = None
        self._spec: Optional[ArchSpec] = None
        self._active_fold = 0

    # ------------------------------------------------------------------ init
    def initialize_from_trained_model_folder(self, model_training_output_dir: str,
                                             use_folds: Union[Tuple[Union[int, str]], None],
                                             checkpoint_name: str = 'checkpoint_final.pth'):
        """Model folder -> plans, dataset.json, per-fold weights (:67-129)."""
        if use_folds is None:
            use_folds = nnUNetPredictor.auto_detect_available_folds(model_training_output_dir, checkpoint_name)
        dataset_json = _load_json(os.path.join(model_training_output_dir, 'dataset.json'))
        plans_manager = PlansManager(_load_json(os.path.join(model_training_output_dir, 'plans.json')))
        if isinstance(use_folds, (str, int)):
            use_folds = [use_folds]
        parameters, trainer_name, configuration_name, mirror_axes, init_args = [], None, None, None, {}
        for i, f in enumerate(use_folds):
            f = int(f) if f != 'all' else f
            checkpoint = torch.load(os.path.join(model_training_output_dir, f'fold_{f}', checkpoint_name),
                                    map_location=torch.device('cpu'), weights_only=False)
            if i == 0:
                trainer_name = checkpoint['trainer_name']
                init_args = checkpoint.get('init_args', {})
                configuration_name = init_args['configuration']
                mirror_axes = checkpoint.get('inference_allowed_mirroring_axes')
            parameters.append(checkpoint['network_weights'])
        configuration_manager = plans_manager.get_configuration(configuration_name)
        self.plans_manager = plans_manager
        self.configuration_manager = configuration_manager
        self.list_of_parameters = parameters
        self.dataset_json = dataset_json
        self.trainer_name = trainer_name
        self.allowed_mirroring_axes = mirror_axes
        self.label_manager = plans_manager.get_label_manager(dataset_json)
        self.network = None
        self._reduction = init_args.get('feature_reduction_factor')
        self._configuration_name = configuration_name
        self._build_engine()

    def manual_initialization(self, network, plans_manager: PlansManager,
                              configuration_manager: ConfigurationManager, parameters: Optional[List[dict]],
                              dataset_json: dict, trainer_name: str,
                              inference_allowed_mirroring_axes: Optional[Tuple[int, ...]]):
        """In-process initialisation (:131-154).  ``network`` may be the torch module the caller built
        (its state dict is read when ``parameters`` is None) or None when ``parameters`` are given."""
        self.plans_manager = plans_manager
        self.configuration_manager = configuration_manager
        self.list_of_parameters = parameters
        self.network = network
        self.dataset_json = dataset_json
        self.trainer_name = trainer_name
        self.allowed_mirroring_axes = inference_allowed_mirroring_axes
        self.label_manager = plans_manager.get_label_manager(dataset_json)
        self._reduction = None
        self._build_engine()

    @staticmethod
    def auto_detect_available_folds(model_training_output_dir, checkpoint_name):
        print('use_folds is None, attempting to auto detect available folds')
        found = []
        for name in sorted(os.listdir(model_training_output_dir)):
            full = os.path.join(model_training_output_dir, name)
            if os.path.isdir(full) and name.startswith('fold_') and name != 'fold_all' \
                    and os.path.isfile(os.path.join(full, checkpoint_name)):
                found.append(int(name.split('_')[-1]))
        print(f'found the following folds: {found}')
        return found

    def _state_dicts(self) -> List[dict]:
        if self.list_of_parameters:
            return list(self.list_of_parameters)
        if self.network is not None and hasattr(self.network, 'state_dict'):
            return [self.network.state_dict()]
        raise RuntimeError('no parameters: pass `parameters` or a network with a state_dict')

    def _build_engine(self):
        sds = self._state_dicts()
        patch = tuple(self.configuration_manager.patch_size)
        if len(patch) not in (2, 3):
            raise NotImplementedError('patch_size must have two (2d) or three (3d_fullres / 3d_lowres) entries')
        kw = {}
        try:
            kw = self.configuration_manager.network_arch_init_kwargs or {}
        except (KeyError, TypeError):
            kw = {}
        eps, slope = ops_from_plans(kw) if kw else (1e-5, 0.01)
        spec = spec_from_state_dict(sds[0], patch, eps=eps, slope=slope)
        if kw and 'n_stages' in kw and 'strides' in kw:
            check_against_plans(spec, kw, self._reduction)
        spec.precision = capi.FNN_PREC_F8 if self.compute_dtype == 'f8' else capi.FNN_PREC_F16
        heads = self.label_manager.num_segmentation_heads
        if heads != spec.num_heads:
            raise RuntimeError(f'checkpoint has {spec.num_heads} segmentation heads, dataset.json implies {heads}')
        if self._engine is not None:
            self._engine.close()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._engine = capi.Engine(spec.to_desc(), device=idx, max_batch=max(1, self.patches_per_forward))
        for f, sd in enumerate(sds):
            blob = weight_blob(spec, sd)
            if blob.size != self._engine.weight_count:
                raise RuntimeError('internal error: weight blob size mismatch')
            self._engine.load_weights(f, blob)
        g = compute_gaussian(patch, sigma_scale=1. / 8, value_scaling_factor=10, device=torch.device('cpu'))
        self._engine.set_gaussian(g.contiguous().view(torch.int16).numpy().view(np.uint16))
        self._spec = spec
        self._n_folds = len(sds)
        self._active_fold = 0

    # --------------------------------------------------------------- predict
    def _opts(self) -> capi.Opts:
        o = capi.Opts()
        o.tile_step_size = float(self.tile_step_size)
        o.use_gaussian = int(bool(self.use_gaussian))
        axes = self.allowed_mirroring_axes if self.use_mirroring else None
        if axes is not None:
            nd = self._spec.spatial_dims
            assert max(axes) <= nd - 1, 'mirror_axes does not match the dimension of the input!'
            o.n_mirror_axes = len(axes)
            for i, a in enumerate(axes):
                o.mirror_axes[i] = int(a) + (3 - nd)           # 2-D network axes (y, z) are engine axes 1, 2
        o.accum = {'fp16': capi.FNN_ACC_FP16_REFERENCE, 'fp32': capi.FNN_ACC_FP32, 'fp16_autocast': capi.FNN_ACC_FP16_AUTOCAST}[self.accumulate_in]
        o.out_dtype = capi.FNN_OUT_F16
        o.batch = self.patches_per_forward
        o.stream = torch.cuda.current_stream(self.device).cuda_stream
        return o

    def _internal_get_sliding_window_slicers(self, image_size: Tuple[int, ...]):
        """Patch windows in visit order (:506-538, both branches)."""
        patch = self.configuration_manager.patch_size
        if len(patch) < len(image_size):
            assert len(patch) == len(image_size) - 1, 'if tile_size has less entries than image_size, len(tile_size) ' \
                                                      'must be one shorter than len(image_size)'
            steps = compute_steps_for_sliding_window(image_size[1:], patch, self.tile_step_size)
            return [tuple([slice(None), d, *[slice(s, s + p) for s, p in zip(st, patch)]])
                    for d in range(image_size[0]) for st in itertools.product(*steps)]
        steps = compute_steps_for_sliding_window(image_size, patch, self.tile_step_size)
        return [tuple([slice(None), *[slice(s, s + p) for s, p in zip(st, patch)]])
                for st in itertools.product(*steps)]

    def _resident_or_host(self, t: torch.Tensor) -> torch.Tensor:
        """float32, contiguous.  A CPU tensor (what the reference's callers hold: the preprocessing iterator's output,
        data_iterators.py:116-117; moved with `data.to(results_device)` at :579) STAYS on the CPU: the engine uploads it by
        tiles (planes x rows) on a copy stream and starts a batch when the tiles under its patches have landed (pin the tensor
        for a DMA straight from it; a pageable one goes through the engine's pinned staging ring)."""
        if t.device.type == 'cpu':
            return t.to(dtype=torch.float32).contiguous()
        return t.to(device=self.device, dtype=torch.float32).contiguous()

    def _check_input(self, input_image):
        assert isinstance(input_image, torch.Tensor)
        assert input_image.ndim == 4, 'input_image must be a 4D np.ndarray or torch.Tensor (c, x, y, z)'
        if self._engine is None:
            raise RuntimeError('predictor is not initialised')

    @torch.inference_mode()
    def predict_sliding_window_return_logits(self, input_image: torch.Tensor) -> torch.Tensor:
        """[C,X,Y,Z] preprocessed image -> fp16 logits [heads,X,Y,Z] (:634-680)."""
        self._check_input(input_image)
        with torch.cuda.device(self.device):
            x = self._resident_or_host(input_image)
            out = torch.empty((self._spec.num_heads, *x.shape[1:]), dtype=torch.half, device=self.device)
            self._engine.predict_volume(x.data_ptr(), x.shape, self._opts(), out.data_ptr(), fold=self._active_fold)
        return out if self.perform_everything_on_device else out.cpu()

    @torch.inference_mode()
    def predict_logits_from_preprocessed_data(self, data: torch.Tensor, on_device: bool = False) -> torch.Tensor:
        """Mean over the folds; returned on the CPU like the reference (:471-504) unless ``on_device`` (not in the
        reference: skips its 15 GiB device-to-host copy of a 61-class 512^3 volume)."""
        self._check_input(data)
        with torch.cuda.device(self.device):
            x = self._resident_or_host(data)
            out = torch.empty((self._spec.num_heads, *x.shape[1:]), dtype=torch.half, device=self.device)
            self._engine.predict_volume(x.data_ptr(), x.shape, self._opts(), out.data_ptr(), n_folds=self._n_folds)
        if self.verbose:
            print('Prediction done')
        return out if on_device else out.to('cpu')

    @torch.inference_mode()
    def predict_single_npy_array(self, input_image: np.ndarray, image_properties: dict,
                                 segmentation_previous_stage: np.ndarray = None,
                                 output_file_truncated: str = None,
                                 save_or_return_probabilities: bool = False):
        """Raw image ``[C, s0, s1, s2]`` + ``{'spacing': ...}`` -> label map on the raw grid (numpy, uint8 / uint16) -
        predict_from_raw_data.py:423-468 with every step on the device: ``DevicePreprocessor.run_case_npy``
        (transpose, crop, normalise, resample), the sliding window, then
        ``convert_predicted_logits_to_segmentation_with_correct_shape`` (export_prediction.py:16-53).  When the case
        needs no resampling the labels are taken straight from the accumulators (no logits are materialised).
        ``save_or_return_probabilities=True`` returns ``(labels, float32 probabilities [heads, s0, s1, s2])`` like the
        reference (softmax / sigmoid, background probability 1 outside the crop box)."""
        from .preprocess import DevicePreprocessor
        if output_file_truncated is not None:
            raise NotImplementedError('image file export is the caller\'s side (SURVEY.md 8: image I/O out of scope)')
        pp = DevicePreprocessor(self.device, verbose=self.verbose)
        props = dict(image_properties)
        if self.verbose:
            print('preprocessing')
        data, seg, props = pp.run_case_npy(input_image, segmentation_previous_stage, props, self.plans_manager,
                                           self.configuration_manager, self.dataset_json)
        if segmentation_previous_stage is not None:
            # cascade: the previous stage's labels as one-hot channels behind the image (convert_labelmap_to_one_hot,
            # label_handling.py:259-292; data_iterators.py:202-204) - the network was built with that many inputs
            # (determine_num_input_channels, label_handling.py:305-310)
            fg = torch.as_tensor(list(self.label_manager.foreground_labels), device=seg.device, dtype=seg.dtype)
            onehot = (seg[0][None] == fg.view(-1, 1, 1, 1)).to(data.dtype)
            data = torch.cat((data, onehot), 0).contiguous()
            if data.shape[0] != self._spec.in_channels:
                raise RuntimeError(f'cascade input has {data.shape[0]} channels (image + {len(fg)} foreground labels), '
                                   f'the network expects {self._spec.in_channels}')
        if self.verbose:
            print('predicting')
        u16 = len(self.label_manager.foreground_labels) >= 255
        same_grid = tuple(data.shape[1:]) == tuple(props['shape_after_cropping_and_before_resampling'])
        if same_grid and not save_or_return_probabilities:
            seg = self.predict_segmentation_from_preprocessed_data(data)
            out = pp.revert_labels(seg, props, self.plans_manager, self.label_manager)
            return out.cpu().numpy().astype(np.uint16 if u16 else np.uint8)
        self._check_input(data)
        with torch.cuda.device(self.device):
            logits = torch.empty((self._spec.num_heads, *data.shape[1:]), dtype=torch.half, device=self.device)
            self._engine.predict_volume(data.data_ptr(), data.shape, self._opts(), logits.data_ptr(), n_folds=self._n_folds)
        if self.verbose:
            print('resampling to original shape')
        if save_or_return_probabilities:
            out, probs = pp.convert_predicted_logits_to_segmentation_and_probabilities(
                logits, self, self.plans_manager, self.configuration_manager, props)
            return out.cpu().numpy().astype(np.uint16 if u16 else np.uint8), probs.cpu().numpy()
        out = pp.convert_predicted_logits_to_segmentation_with_correct_shape(logits, self, self.plans_manager,
                                                                             self.configuration_manager, props)
        return out.cpu().numpy().astype(np.uint16 if u16 else np.uint8)

    def _label_rule(self):
        """(regions_class_order or None, uint16?) - LabelManager.convert_logits_to_segmentation
        (label_handling.py:163-181) and the dtype rule of export_prediction.py:45-46."""
        lm = self.label_manager
        order = None
        if lm.has_regions:
            assert lm.regions_class_order is not None, \
                'if region-based training is requested then you need to define regions_class_order!'
            order = [int(c) for c in lm.regions_class_order]
        return order, len(lm.foreground_labels) >= 255

    def predict_segmentation_from_preprocessed_data(self, data: torch.Tensor) -> torch.Tensor:
        """Label map on the device, skipping the full-logit D2H copy the reference pays at :386 before
        ``convert_logits_to_segmentation`` (label_handling.py:144-195): argmax for plain labels, sigmoid > 0.5
        painted in ``regions_class_order`` for region-based training; uint8, or uint16 (returned as int32 - torch
        has no uint16 arithmetic) when the dataset has >= 255 foreground labels (export_prediction.py:45-46).
        With one fold the labels are taken straight from the accumulators; the logits are never written."""
        self._check_input(data)
        order, u16 = self._label_rule()
        with torch.cuda.device(self.device):
            x = self._resident_or_host(data)
            self._engine.set_label_rule(order, uint16=u16)
            labels = torch.empty(x.shape[1:], dtype=torch.int16 if u16 else torch.uint8, device=self.device)
            self._engine.predict_labels(x.data_ptr(), x.shape, self._opts(), labels.data_ptr(), n_folds=self._n_folds)
            if u16:
                labels = labels.to(torch.int32) & 0xffff
        return labels

    def convert_logits_to_segmentation(self, predicted_logits: torch.Tensor) -> torch.Tensor:
        """``LabelManager.convert_logits_to_segmentation`` (label_handling.py:183-195) on resident logits
        ``[heads, X, Y, Z]`` (fp16 or fp32, on the device)."""
        from . import capi
        assert predicted_logits.ndim == 4 and predicted_logits.shape[0] == self._spec.num_heads
        order, u16 = self._label_rule()
        with torch.cuda.device(self.device):
            lg = predicted_logits.to(self.device)
            if lg.dtype not in (torch.half, torch.float32):
                lg = lg.float()
            lg = lg.contiguous()
            self._engine.set_label_rule(order, uint16=u16)
            labels = torch.empty(lg.shape[1:], dtype=torch.int16 if u16 else torch.uint8, device=self.device)
            self._engine.argmax_labels(lg.data_ptr(), capi.FNN_OUT_F32 if lg.dtype == torch.float32 else capi.FNN_OUT_F16,
                                       lg.shape[0], lg[0].numel(), labels.data_ptr(),
                                       torch.cuda.current_stream(self.device).cuda_stream)
            if u16:
                labels = labels.to(torch.int32) & 0xffff
        return labels

    @torch.inference_mode()
    def forward_patches(self, x: torch.Tensor) -> torch.Tensor:
        """``self.network(x)`` for a batch of patches: [n,C,px,py,pz] -> fp32 logits [n,heads,px,py,pz]
        ([n,C,py,pz] -> [n,heads,py,pz] for a 2-D configuration)."""
        two_d = self._spec.spatial_dims == 2
        sp = tuple(self._spec.patch[1:]) if two_d else tuple(self._spec.patch)
        assert x.ndim == len(sp) + 2 and tuple(x.shape[2:]) == sp
        with torch.cuda.device(self.device):
            xd = x.to(device=self.device, dtype=torch.float32).contiguous()
            out = torch.empty((x.shape[0], self._spec.num_heads, *x.shape[2:]), dtype=torch.float32, device=self.device)
            self._engine.forward_patches(xd.data_ptr(), x.shape[0], out.data_ptr(), fold=self._active_fold,
                                         stream=torch.cuda.current_stream(self.device).cuda_stream)
        return out
