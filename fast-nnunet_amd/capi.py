"""ctypes binding of the C ABI in ``include/fnn.h`` (``csrc/libfnn_hip.so``).

This is the stub a maintainer of the reference would add to call the engine
from ``nnUNetPredictor``.  There is no CPU fallback: if the
HIP library is missing or no GPU is visible the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

FNN_MAX_STAGES = 8
FNN_OK, FNN_E_INVALID, FNN_E_HIP, FNN_E_INF, FNN_E_UNSUPPORTED, FNN_E_STATE = 0, -1, -2, -3, -4, -5
FNN_NET_PLAIN, FNN_NET_RESENC = 0, 1
FNN_PREC_F16, FNN_PREC_F8 = 0, 1
FNN_ACC_FP16_REFERENCE, FNN_ACC_FP32, FNN_ACC_FP16_AUTOCAST = 0, 1, 2
FNN_OUT_F16, FNN_OUT_F32 = 0, 1
FNN_LABELS_ARGMAX, FNN_LABELS_REGIONS = 0, 1
FNN_LABEL_U8, FNN_LABEL_U16 = 0, 1
FNN_NORM_NONE, FNN_NORM_ZSCORE, FNN_NORM_CT, FNN_NORM_RESCALE01, FNN_NORM_RGB01 = 0, 1, 2, 3, 4

_HERE = os.path.dirname(os.path.abspath(__file__))
# FNN_LIB: another build of the same library (A-B comparisons of two builds inside one GPU session) - like every FNN_* switch
# of the library itself (csrc/misc.hip, fnn_knob) honoured only next to FNN_KNOBS=1: an embedding process's environment
# does not choose the code that runs
_KNOBS = os.environ.get('FNN_KNOBS', '0') not in ('', '0')
LIB_PATH = (os.environ.get('FNN_LIB') if _KNOBS else None) or os.path.join(_HERE, 'csrc', 'libfnn_hip.so')


class ArchDesc(C.Structure):
    _fields_ = [('kind', C.c_int32), ('n_stages', C.c_int32), ('in_channels', C.c_int32), ('num_heads', C.c_int32),
                ('features', C.c_int32 * FNN_MAX_STAGES),
                ('kernels', (C.c_int32 * 3) * FNN_MAX_STAGES),
                ('strides', (C.c_int32 * 3) * FNN_MAX_STAGES),
                ('n_conv_enc', C.c_int32 * FNN_MAX_STAGES),
                ('n_conv_dec', C.c_int32 * FNN_MAX_STAGES),
                ('patch', C.c_int32 * 3),
                ('eps', C.c_float), ('slope', C.c_float), ('spatial_dims', C.c_int32), ('precision', C.c_int32)]


class NormDesc(C.Structure):
    _fields_ = [('scheme', C.c_int32), ('mean', C.c_float), ('std', C.c_float), ('lower', C.c_float), ('upper', C.c_float),
                ('use_mask', C.c_int32)]


class ResampleDesc(C.Structure):
    _fields_ = [('order', C.c_int32), ('separate_axis', C.c_int32), ('order_z', C.c_int32), ('dtype', C.c_int32)]


class Opts(C.Structure):
    _fields_ = [('tile_step_size', C.c_double), ('use_gaussian', C.c_int32), ('n_mirror_axes', C.c_int32),
                ('mirror_axes', C.c_int32 * 3), ('accum', C.c_int32), ('out_dtype', C.c_int32),
                ('batch', C.c_int32), ('stream', C.c_void_p)]


class Profile(C.Structure):
    _fields_ = [('total_ms', C.c_double), ('conv_ms', C.c_double), ('stem_ms', C.c_double),
                ('tconv_ms', C.c_double), ('head_ms', C.c_double), ('finalize_ms', C.c_double),
                ('conv_launches', C.c_int64), ('conv_flops', C.c_double), ('n_patches', C.c_int64),
                ('conv_bytes', C.c_double)]


EXPORTS = ['fnn_abi_version', 'fnn_last_error', 'fnn_create', 'fnn_destroy', 'fnn_weight_count', 'fnn_load_weights',
           'fnn_set_gaussian', 'fnn_predict_volume', 'fnn_predict_volume_ensemble', 'fnn_predict_labels',
           'fnn_set_label_rule', 'fnn_accumulator_channels', 'fnn_accumulate_patches', 'fnn_normalize_box', 'fnn_labels_box', 'fnn_feature_channels', 'fnn_patch_features', 'fnn_gather_box', 'fnn_pack_regions', 'fnn_unpack_regions', 'fnn_forward_patches', 'fnn_argmax_labels', 'fnn_nonzero_bbox', 'fnn_preprocess', 'fnn_revert_labels', 'fnn_export_probabilities', 'fnn_resample', 'fnn_compute_steps', 'fnn_plan_volume', 'fnn_fp8_e4m3_encode',
           'fnn_set_profiling', 'fnn_get_profile', 'fnn_kernel_log', 'fnn_profile_launches', 'fnn_layer_table', 'fnn_patch_work', 'fnn_op_conv3d', 'fnn_op_conv_transpose3d', 'fnn_op_quotient_check', 'fnn_op_last_kernels', 'fnn_clock_probe_start', 'fnn_clock_probe_stop']

_lib = None


class EngineError(RuntimeError):
    pass


def load_library() -> C.CDLL:
    """Load ``libfnn_hip.so``; raises if it was not built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise EngineError(f'{LIB_PATH} is missing: build it with `make -C {os.path.dirname(LIB_PATH)}` '
                          f'(or `python -c "import __graft_entry__ as g; g.build()"`). '
                          f'There is no CPU fallback for the inference engine.')
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, f32p = C.c_void_p, C.c_int, C.c_int64, C.POINTER(C.c_float)
    lib.fnn_abi_version.restype = i32
    lib.fnn_last_error.restype = C.c_char_p
    lib.fnn_last_error.argtypes = [vp]
    lib.fnn_create.argtypes = [C.POINTER(ArchDesc), i32, i32, C.POINTER(vp)]
    lib.fnn_destroy.argtypes = [vp]
    lib.fnn_destroy.restype = None
    lib.fnn_weight_count.argtypes = [vp]
    lib.fnn_weight_count.restype = i64
    lib.fnn_load_weights.argtypes = [vp, i32, vp, i64]
    lib.fnn_set_gaussian.argtypes = [vp, vp, i64]
    lib.fnn_predict_volume.argtypes = [vp, i32, vp, C.POINTER(i64), C.POINTER(Opts), vp]
    lib.fnn_predict_volume_ensemble.argtypes = [vp, i32, vp, C.POINTER(i64), C.POINTER(Opts), vp]
    lib.fnn_predict_labels.argtypes = [vp, i32, vp, C.POINTER(i64), C.POINTER(Opts), vp]
    lib.fnn_set_label_rule.argtypes = [vp, i32, C.POINTER(i32), i32, i32]
    lib.fnn_accumulator_channels.argtypes = [vp]
    lib.fnn_accumulator_channels.restype = i64
    P64 = C.POINTER(i64)
    lib.fnn_accumulate_patches.argtypes = [vp, i32, vp, P64, C.POINTER(Opts), P64, i64, P64, P64, vp]
    lib.fnn_normalize_box.argtypes = [vp, vp, P64, C.POINTER(Opts), P64, P64, P64, P64, vp]
    lib.fnn_labels_box.argtypes = [vp, vp, P64, C.POINTER(Opts), P64, P64, P64, P64, vp]
    lib.fnn_feature_channels.argtypes = [vp]
    lib.fnn_feature_channels.restype = i64
    lib.fnn_patch_features.argtypes = [vp, i32, vp, P64, C.POINTER(Opts), P64, i64, vp, vp, i64, i64]
    lib.fnn_gather_box.argtypes = [vp, i32, vp, vp, C.POINTER(C.c_int32), i64, P64, C.POINTER(Opts), P64, P64, vp, vp]
    lib.fnn_pack_regions.argtypes = [vp, vp, i64, vp, i64, vp, vp]
    lib.fnn_unpack_regions.argtypes = [vp, vp, i64, vp, i64, vp, vp]
    lib.fnn_forward_patches.argtypes = [vp, i32, vp, i32, vp, vp]
    lib.fnn_argmax_labels.argtypes = [vp, vp, i32, i32, i64, vp, vp]
    lib.fnn_nonzero_bbox.argtypes = [vp, C.POINTER(i64), C.POINTER(i32), C.POINTER(i64), vp]
    lib.fnn_preprocess.argtypes = [vp, C.POINTER(i64), C.POINTER(i32), C.POINTER(i64), C.POINTER(NormDesc), vp, vp]
    lib.fnn_revert_labels.argtypes = [vp, i32, C.POINTER(i64), C.POINTER(i64), C.POINTER(i32), vp, vp]
    lib.fnn_export_probabilities.argtypes = [vp, i32, i32, C.POINTER(C.c_int32), C.POINTER(i64), C.POINTER(i64),
                                             C.POINTER(i32), vp, vp, i32, vp]
    lib.fnn_resample.argtypes = [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(ResampleDesc), vp, vp]
    lib.fnn_compute_steps.argtypes = [i64, i64, C.c_double, C.POINTER(i64), i32]
    lib.fnn_plan_volume.argtypes = [C.POINTER(C.c_int32), C.POINTER(i64), C.c_double, C.POINTER(i64), C.POINTER(i64),
                                    C.POINTER(i64), C.POINTER(C.c_int32), i64]
    lib.fnn_fp8_e4m3_encode.argtypes = [vp, i64, vp]
    lib.fnn_set_profiling.argtypes = [vp, i32]
    lib.fnn_get_profile.argtypes = [vp, C.POINTER(Profile)]
    lib.fnn_kernel_log.argtypes = [vp, C.c_char_p, i64]
    lib.fnn_kernel_log.restype = i64
    for name in ('fnn_profile_launches', 'fnn_layer_table'):       # (absent from an older build picked with FNN_LIB for an A-B)
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.argtypes = [vp, C.c_char_p, i64]
            fn.restype = i64
    lib.fnn_patch_work.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    I3 = C.POINTER(C.c_int)
    lib.fnn_op_conv3d.argtypes = [i32, i32, I3, f32p, i32, f32p, f32p, C.c_float, f32p, i32, f32p, f32p, C.c_float,
                                  f32p, f32p, i32, I3, I3, f32p, C.POINTER(C.c_double)]
    lib.fnn_op_conv_transpose3d.argtypes = [i32, i32, I3, f32p, i32, f32p, f32p, C.c_float, f32p, f32p, i32, I3, f32p]
    lib.fnn_op_quotient_check.argtypes = [i32, C.POINTER(C.c_uint64)]
    lib.fnn_op_last_kernels.argtypes = [C.c_char_p, i32]
    lib.fnn_clock_probe_start.argtypes = [i32, C.c_double, C.POINTER(C.c_void_p)]
    lib.fnn_clock_probe_stop.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    if lib.fnn_abi_version() != 4:
        raise EngineError('libfnn_hip.so has an unexpected ABI version')
    _lib = lib
    return lib


def _err(lib, handle) -> str:
    msg = lib.fnn_last_error(handle)
    return msg.decode() if msg else ''


def check(rc: int, lib, handle=None):
    """Map C status codes onto the exception types the reference raises."""
    if rc >= 0:
        return rc
    msg = _err(lib, handle)
    if rc == FNN_E_INVALID:
        raise AssertionError(msg)
    if rc == FNN_E_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise RuntimeError(msg)            # FNN_E_INF, FNN_E_HIP, FNN_E_STATE


def _f32p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


def fp8_e4m3_encode(x: np.ndarray) -> np.ndarray:
    """float32 -> OCP e4m3 bytes with the library's host quantiser (FNN_PREC_F8 weight packing)."""
    lib = load_library()
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty(x.shape, np.uint8)
    check(lib.fnn_fp8_e4m3_encode(x.ctypes.data, x.size, out.ctypes.data), lib)
    return out


def compute_steps(image_size: int, patch_size: int, step: float):
    lib = load_library()
    buf = (C.c_int64 * 4096)()
    n = check(lib.fnn_compute_steps(int(image_size), int(patch_size), float(step), buf, 4096), lib)
    return [int(buf[i]) for i in range(n)]


def plan_volume(patch: Sequence[int], shape_sp: Sequence[int], step: float):
    """-> (padded shape, low pads, origins [n,3]) exactly as the engine will visit them.  A patch with two
    entries is a `2d` configuration: every slice of the first axis, tiles over the other two."""
    lib = load_library()
    patch = [0, *patch] if len(patch) == 2 else list(patch)
    p = (C.c_int32 * 3)(*[int(i) for i in patch])
    s = (C.c_int64 * 3)(*[int(i) for i in shape_sp])
    padded, lo, n = (C.c_int64 * 3)(), (C.c_int64 * 3)(), C.c_int64(0)
    check(lib.fnn_plan_volume(p, s, float(step), padded, lo, C.byref(n), None, 0), lib)
    org = np.zeros((n.value, 3), dtype=np.int32)
    check(lib.fnn_plan_volume(p, s, float(step), padded, lo, C.byref(n),
                              org.ctypes.data_as(C.POINTER(C.c_int32)), n.value), lib)
    return list(padded), list(lo), org


def nonzero_bbox(raw_ptr: int, shape, transpose_forward, stream: int = 0):
    """-> [[lo, hi], ...] per TRANSPOSED axis (properties['bbox_used_for_cropping'])."""
    lib = load_library()
    bbox = (C.c_int64 * 6)()
    check(lib.fnn_nonzero_bbox(raw_ptr, (C.c_int64 * 4)(*[int(i) for i in shape]),
                               (C.c_int32 * 3)(*[int(i) for i in transpose_forward]), bbox, stream), lib)
    return [[int(bbox[2 * a]), int(bbox[2 * a + 1])] for a in range(3)]


def preprocess(raw_ptr: int, shape, transpose_forward, bbox, norms, out_ptr: int, stream: int = 0):
    """norms: one (scheme, mean, std, lower, upper[, use_mask]) per channel."""
    lib = load_library()
    nd = (NormDesc * len(norms))(*[NormDesc(int(n[0]), float(n[1]), float(n[2]), float(n[3]), float(n[4]),
                                            int(n[5]) if len(n) > 5 else 0) for n in norms])
    flat = (C.c_int64 * 6)(*[int(v) for ab in bbox for v in ab])
    check(lib.fnn_preprocess(raw_ptr, (C.c_int64 * 4)(*[int(i) for i in shape]),
                             (C.c_int32 * 3)(*[int(i) for i in transpose_forward]), flat, nd, out_ptr, stream), lib)


def revert_labels(seg_ptr: int, uint16: bool, bbox, shape_before_cropping, transpose_backward, out_ptr: int, stream: int = 0):
    lib = load_library()
    flat = (C.c_int64 * 6)(*[int(v) for ab in bbox for v in ab])
    check(lib.fnn_revert_labels(seg_ptr, FNN_LABEL_U16 if uint16 else FNN_LABEL_U8, flat,
                                (C.c_int64 * 3)(*[int(i) for i in shape_before_cropping]),
                                (C.c_int32 * 3)(*[int(i) for i in transpose_backward]), out_ptr, stream), lib)


def export_probabilities(logits_ptr: int, half: bool, heads: int, regions_class_order, bbox, shape_before_cropping,
                         transpose_backward, probs_ptr: int, labels_ptr: int, uint16: bool, stream: int = 0):
    """Probabilities + labels on the original grid from logits of the cropped grid (export_prediction.py:36-70)."""
    lib = load_library()
    order = None
    if regions_class_order is not None:
        assert len(regions_class_order) == heads
        order = (C.c_int32 * heads)(*[int(c) for c in regions_class_order])
    flat = (C.c_int64 * 6)(*[int(v) for ab in bbox for v in ab])
    check(lib.fnn_export_probabilities(logits_ptr, FNN_OUT_F16 if half else FNN_OUT_F32, int(heads), order, flat,
                                       (C.c_int64 * 3)(*[int(i) for i in shape_before_cropping]),
                                       (C.c_int32 * 3)(*[int(i) for i in transpose_backward]), probs_ptr, labels_ptr,
                                       FNN_LABEL_U16 if uint16 else FNN_LABEL_U8, stream), lib)


def resample(in_ptr: int, shape, new_shape, order: int, separate_axis, half: bool, out_ptr: int, stream: int = 0, order_z: int = 0):
    """resample_data_or_seg(is_seg=False) of a [C, ...] tensor (fp32, or fp16 when `half`)."""
    lib = load_library()
    d = ResampleDesc(int(order), -1 if separate_axis is None else int(separate_axis), int(order_z),
                     FNN_OUT_F16 if half else FNN_OUT_F32)
    check(lib.fnn_resample(in_ptr, (C.c_int64 * 4)(*[int(i) for i in shape]), (C.c_int64 * 3)(*[int(i) for i in new_shape]),
                           C.byref(d), out_ptr, stream), lib)


def op_conv3d(x, w, bias, k, stride, gamma=None, beta=None, slope=1.0, x2=None, gamma2=None, beta2=None, slope2=1.0,
              device=0, want_stats=False):
    """Single conv through the HIP kernel.  x [n,cin,D,H,W] float32 (host)."""
    lib = load_library()
    x = np.ascontiguousarray(x, np.float32)
    n, cin = x.shape[:2]
    dims = (C.c_int * 3)(*x.shape[2:])
    w = np.ascontiguousarray(w, np.float32)
    cout = w.shape[0]
    kk, ss = (C.c_int * 3)(*k), (C.c_int * 3)(*stride)
    od = [(x.shape[2 + i] + 2 * ((k[i] - 1) // 2) - k[i]) // stride[i] + 1 for i in range(3)]
    y = np.zeros((n, cout, *od), np.float32)
    stats = np.zeros((n, cout, 2), np.float64) if want_stats else None
    f = lambda a: None if a is None else np.ascontiguousarray(a, np.float32)
    x2, gamma, beta, gamma2, beta2, bias = f(x2), f(gamma), f(beta), f(gamma2), f(beta2), f(bias)
    rc = lib.fnn_op_conv3d(device, n, dims, _f32p(x), cin, _f32p(gamma), _f32p(beta), slope,
                           _f32p(x2), 0 if x2 is None else x2.shape[1], _f32p(gamma2), _f32p(beta2), slope2,
                           _f32p(w), _f32p(bias), cout, kk, ss, _f32p(y),
                           None if stats is None else stats.ctypes.data_as(C.POINTER(C.c_double)))
    check(rc, lib)
    return (y, stats) if want_stats else y


def op_conv_transpose3d(x, w, bias, stride, gamma=None, beta=None, slope=1.0, device=0):
    lib = load_library()
    x = np.ascontiguousarray(x, np.float32)
    n, cin = x.shape[:2]
    dims = (C.c_int * 3)(*x.shape[2:])
    w = np.ascontiguousarray(w, np.float32)
    cout = w.shape[1]
    ss = (C.c_int * 3)(*stride)
    y = np.zeros((n, cout, *[x.shape[2 + i] * stride[i] for i in range(3)]), np.float32)
    f = lambda a: None if a is None else np.ascontiguousarray(a, np.float32)
    gamma, beta, bias = f(gamma), f(beta), f(bias)
    rc = lib.fnn_op_conv_transpose3d(device, n, dims, _f32p(x), cin, _f32p(gamma), _f32p(beta), slope,
                                     _f32p(w), _f32p(bias), cout, ss, _f32p(y))
    check(rc, lib)
    return y


def op_last_kernels():
    """Kernel variants launched by this thread's last op_conv3d / op_conv_transpose3d call."""
    lib = load_library()
    buf = C.create_string_buffer(4096)
    lib.fnn_op_last_kernels(buf, 4096)
    return [k for k in buf.value.decode().split('\n') if k]


def clock_probe_start(device=0, max_seconds=1.0):
    """One sleeping wave that reads the shader-clock and the 100 MHz counters for max_seconds or until clock_probe_stop (include/fnn.h)."""
    lib = load_library()
    h = C.c_void_p()
    check(lib.fnn_clock_probe_start(device, float(max_seconds), C.byref(h)), lib)
    return h


def clock_probe_stop(handle):
    """(GHz the device held since clock_probe_start, seconds covered)."""
    lib = load_library()
    ghz, sec = C.c_double(), C.c_double()
    check(lib.fnn_clock_probe_stop(handle, C.byref(ghz), C.byref(sec)), lib)
    return ghz.value, sec.value


def op_quotient_check(device=0):
    """(differing pairs, pairs on the fast route, (a bits, b bits) of one differing pair) of the gather epilogue's quotient over all fp16 a, b >= +0."""
    lib = load_library()
    counts = (C.c_uint64 * 3)()
    check(lib.fnn_op_quotient_check(device, counts), lib)
    return int(counts[0]), int(counts[1]), (int(counts[2]) & 0xFFFF, int(counts[2]) >> 16)


class Engine:
    """Thin RAII wrapper around an ``fnn_engine*``."""

    def __init__(self, desc: ArchDesc, device: int = 0, max_batch: int = 4):
        self.lib = load_library()
        self.handle = C.c_void_p()
        self.desc = desc
        check(self.lib.fnn_create(C.byref(desc), int(device), int(max_batch), C.byref(self.handle)), self.lib, None)
        self.max_batch = max_batch
        self.device = device

    def close(self):
        if getattr(self, 'handle', None) is not None and self.handle:
            self.lib.fnn_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def weight_count(self) -> int:
        return int(self.lib.fnn_weight_count(self.handle))

    def load_weights(self, fold: int, blob: np.ndarray):
        blob = np.ascontiguousarray(blob, np.float32)
        check(self.lib.fnn_load_weights(self.handle, fold, blob.ctypes.data, blob.size), self.lib, self.handle)

    def set_gaussian(self, half_bits: np.ndarray):
        hb = np.ascontiguousarray(half_bits, np.uint16)
        check(self.lib.fnn_set_gaussian(self.handle, hb.ctypes.data, hb.size), self.lib, self.handle)

    def set_profiling(self, on: bool):
        check(self.lib.fnn_set_profiling(self.handle, int(on)), self.lib, self.handle)

    def profile(self) -> Profile:
        p = Profile()
        check(self.lib.fnn_get_profile(self.handle, C.byref(p)), self.lib, self.handle)
        return p

    def kernel_log(self):
        """Kernel variants of the last call made with profiling on, one per launch."""
        n = self.lib.fnn_kernel_log(self.handle, None, 0)
        buf = C.create_string_buffer(int(n))
        self.lib.fnn_kernel_log(self.handle, buf, n)
        return [k for k in buf.value.decode().split('\n') if k]

    def _text(self, fn):
        n = fn(self.handle, None, 0)
        buf = C.create_string_buffer(int(n))
        fn(self.handle, buf, n)
        return [r.split('\t') for r in buf.value.decode().split('\n') if r]

    def profile_launches(self):
        """Rows (layer, family, ms, flops, bytes, kernels) of every timed launch of the last profiled call."""
        return [(int(r[0]), r[1], float(r[2]), float(r[3]), float(r[4]), r[5] if len(r) > 5 else '') for r in self._text(self.lib.fnn_profile_launches)]

    def layer_table(self):
        """The engine's layer plan: dicts with index, type, cin, cout, kernel, stride, in_dims, out_dims, flops, bytes, fused."""
        keys = ('index', 'type', 'cin', 'cout', 'kernel', 'stride', 'in_dims', 'out_dims', 'flops', 'bytes', 'fused')
        conv = (int, str, int, int, str, str, str, str, float, float, int)
        return [dict((k, c(v)) for k, c, v in zip(keys, conv, r)) for r in self._text(self.lib.fnn_layer_table)]

    def patch_work(self):
        fl, by = C.c_double(), C.c_double()
        check(self.lib.fnn_patch_work(self.handle, C.byref(fl), C.byref(by)), self.lib, self.handle)
        return fl.value, by.value

    def predict_volume(self, vol_ptr: int, shape, opts: Opts, out_ptr: int, fold: int = 0, n_folds: int = 0):
        shp = (C.c_int64 * 4)(*[int(i) for i in shape])
        if n_folds > 0:
            rc = self.lib.fnn_predict_volume_ensemble(self.handle, n_folds, vol_ptr, shp, C.byref(opts), out_ptr)
        else:
            rc = self.lib.fnn_predict_volume(self.handle, fold, vol_ptr, shp, C.byref(opts), out_ptr)
        check(rc, self.lib, self.handle)

    def predict_labels(self, vol_ptr: int, shape, opts: Opts, labels_ptr: int, n_folds: int = 1):
        shp = (C.c_int64 * 4)(*[int(i) for i in shape])
        check(self.lib.fnn_predict_labels(self.handle, n_folds, vol_ptr, shp, C.byref(opts), labels_ptr),
              self.lib, self.handle)

    @property
    def accumulator_channels(self) -> int:
        return int(self.lib.fnn_accumulator_channels(self.handle))

    def accumulate_patches(self, vol_ptr, shape, opts, patch_ids, box_lo, box_hi, acc_ptr, fold=0):
        shp = (C.c_int64 * 4)(*[int(i) for i in shape])
        ids = (C.c_int64 * max(1, len(patch_ids)))(*[int(i) for i in patch_ids])
        lo, hi = (C.c_int64 * 3)(*[int(i) for i in box_lo]), (C.c_int64 * 3)(*[int(i) for i in box_hi])
        check(self.lib.fnn_accumulate_patches(self.handle, fold, vol_ptr, shp, C.byref(opts), ids, len(patch_ids),
                                              lo, hi, acc_ptr), self.lib, self.handle)

    def normalize_box(self, acc_ptr, shape, opts, box_lo, box_hi, out_lo, out_hi, out_ptr):
        shp = (C.c_int64 * 4)(*[int(i) for i in shape])
        a = [(C.c_int64 * 3)(*[int(i) for i in v]) for v in (box_lo, box_hi, out_lo, out_hi)]
        check(self.lib.fnn_normalize_box(self.handle, acc_ptr, shp, C.byref(opts), a[0], a[1], a[2], a[3], out_ptr),
              self.lib, self.handle)

    @property
    def feature_channels(self) -> int:
        return int(self.lib.fnn_feature_channels(self.handle))

    def patch_features(self, vol_ptr, shape, opts, patch_ids, feat_ptr, fss_ptr, fold=0, slot0=0, n_slots=None):
        """feat [n_eval][n_slots][P][C] / fss [n_eval][n_slots][2][C] (base pointers): patch_ids[i] -> slot slot0 + i."""
        shp = (C.c_int64 * 4)(*[int(i) for i in shape])
        ids = (C.c_int64 * max(1, len(patch_ids)))(*[int(i) for i in patch_ids])
        n_slots = slot0 + len(patch_ids) if n_slots is None else n_slots
        check(self.lib.fnn_patch_features(self.handle, fold, vol_ptr, shp, C.byref(opts), ids, len(patch_ids), feat_ptr, fss_ptr,
                                          C.c_int64(slot0), C.c_int64(n_slots)), self.lib, self.handle)

    def gather_box(self, feat_ptr, fss_ptr, slot_of_patch, shape, opts, out_lo, out_hi, logits_ptr=None, labels_ptr=None, fold=0,
                   n_slots=None):
        shp = (C.c_int64 * 4)(*[int(i) for i in shape])
        tab = np.ascontiguousarray(slot_of_patch, np.int32)
        n_slots = int(tab.max()) + 1 if n_slots is None else n_slots
        lo, hi = (C.c_int64 * 3)(*[int(i) for i in out_lo]), (C.c_int64 * 3)(*[int(i) for i in out_hi])
        check(self.lib.fnn_gather_box(self.handle, fold, feat_ptr, fss_ptr, tab.ctypes.data_as(C.POINTER(C.c_int32)),
                                      C.c_int64(max(1, n_slots)), shp, C.byref(opts), lo, hi, logits_ptr, labels_ptr),
              self.lib, self.handle)

    def pack_regions(self, feat_ptr, n_slots, regions_ptr, n, message_ptr, stream=0):
        """`n` sub-blocks of the kept activations (device table of fnn_region records, include/fnn.h) -> one message buffer."""
        check(self.lib.fnn_pack_regions(self.handle, feat_ptr, C.c_int64(n_slots), regions_ptr, C.c_int64(n), message_ptr, stream),
              self.lib, self.handle)

    def unpack_regions(self, feat_ptr, n_slots, regions_ptr, n, message_ptr, stream=0):
        check(self.lib.fnn_unpack_regions(self.handle, feat_ptr, C.c_int64(n_slots), regions_ptr, C.c_int64(n), message_ptr, stream),
              self.lib, self.handle)

    def labels_box(self, acc_ptr, shape, opts, box_lo, box_hi, out_lo, out_hi, labels_ptr):
        shp = (C.c_int64 * 4)(*[int(i) for i in shape])
        a = [(C.c_int64 * 3)(*[int(i) for i in v]) for v in (box_lo, box_hi, out_lo, out_hi)]
        check(self.lib.fnn_labels_box(self.handle, acc_ptr, shp, C.byref(opts), a[0], a[1], a[2], a[3], labels_ptr),
              self.lib, self.handle)

    def forward_patches(self, x_ptr: int, n: int, out_ptr: int, fold: int = 0, stream: int = 0):
        check(self.lib.fnn_forward_patches(self.handle, fold, x_ptr, n, out_ptr, stream), self.lib, self.handle)

    def set_label_rule(self, regions_class_order=None, uint16: bool = False):
        """``None`` = argmax over heads; a sequence = region-based training (sigmoid > 0.5 painted in that order)."""
        dt = FNN_LABEL_U16 if uint16 else FNN_LABEL_U8
        if regions_class_order is None:
            rc = self.lib.fnn_set_label_rule(self.handle, FNN_LABELS_ARGMAX, None, 0, dt)
        else:
            order = (C.c_int32 * len(regions_class_order))(*[int(i) for i in regions_class_order])
            rc = self.lib.fnn_set_label_rule(self.handle, FNN_LABELS_REGIONS, order, len(regions_class_order), dt)
        check(rc, self.lib, self.handle)

    def argmax_labels(self, logits_ptr, dtype, heads, n_vox, labels_ptr, stream=0):
        check(self.lib.fnn_argmax_labels(self.handle, logits_ptr, dtype, heads, n_vox, labels_ptr, stream),
              self.lib, self.handle)
