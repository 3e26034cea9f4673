/*
 * fnn.h - C ABI of the MI355X-native sliding-window 3D-U-Net inference engine.
 *
 * This is the drop-in boundary for the hot path of 77even/Fast-nnUNet.  The
 * reference has no native plugin interface for this path: the
 * boundary in the reference is the Python method
 *   nnUNetPredictor.predict_sliding_window_return_logits
 *     (distillation/nnunetv2/inference/predict_from_raw_data.py:634-680)
 * and the functions it drives.  Each entry point below names the reference
 * code it replaces.  Plain pointers and sizes only; no torch types.
 *
 * Conventions
 *   - every function returns 0 on success, a negative FNN_E_* code otherwise;
 *     fnn_last_error() gives the message (thread-local for fnn_create failures,
 *     per engine afterwards);
 *   - volumes / logits are channel-first contiguous [C, X, Y, Z], exactly the
 *     tensors the reference passes; pointers may be device (HIP) or host
 *     pointers, the engine detects which;
 *   - the engine never frees or mutates caller memory; outputs are
 *     caller-allocated;
 *   - one engine = one GPU; not re-entrant per engine (like the reference's
 *     predictor, which swaps fold weights in place, :486-489).
 */
#ifndef FNN_H
#define FNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FNN_MAX_STAGES 8
#define FNN_ABI_VERSION 4

enum {
    FNN_OK = 0,
    FNN_E_INVALID = -1,      /* bad argument (maps to AssertionError / ValueError)            */
    FNN_E_HIP = -2,          /* HIP runtime failure (maps to RuntimeError)                    */
    FNN_E_INF = -3,          /* 'Encountered inf in predicted array' (:622-625)               */
    FNN_E_UNSUPPORTED = -4,  /* topology the engine does not implement                        */
    FNN_E_STATE = -5         /* weights / gaussian not loaded                                 */
};

enum { FNN_NET_PLAIN = 0, FNN_NET_RESENC = 1 };
/* fnn_arch_desc.precision - the operand format of the matrix cores in the 3x3x3 stride-1 convolutions
 * (BASELINE config 5 asks for an fp8 conv path; the reference has no fp8 semantics):
 *   FNN_PREC_F16  fp16 operands everywhere (default; every parity statement in DESIGN.md is for this mode);
 *   FNN_PREC_F8   OCP e4m3 operands in those convolutions: weights quantised per output channel when they are
 *                 loaded, activations while they are staged (after InstanceNorm + LeakyReLU, x 8); fp32
 *                 accumulation; everything else (storage, statistics, strided / 1x1x1 / transposed convs, seg
 *                 head, accumulation) as in fp16.  Budget against the fp32 oracle: DESIGN.md, "fp8". */
enum { FNN_PREC_F16 = 0, FNN_PREC_F8 = 1 };
/* fnn_opts.accum - the arithmetic of the Gaussian-weighted accumulation (predict_from_raw_data.py:602-621):
 *   FNN_ACC_FP16_REFERENCE  the reference WITHOUT autocast (its CPU path, :591-593 `dummy_context`): the network's
 *                           logits are fp32, `prediction *= gaussian` is an fp32 product, `predicted_logits += prediction`
 *                           an fp32 sum rounded once to the fp16 accumulator.  Pinned by tests/golden/sliding_window.npz.
 *   FNN_ACC_FP32            fp32 accumulators (what the reference's error message recommends when fp16 overflows).
 *   FNN_ACC_FP16_AUTOCAST   the reference ON A GPU (`torch.autocast`: the network's output is fp16): the logit, every
 *                           mirror sum, the division by the number of evaluations, the product with the Gaussian and
 *                           the sum are each rounded to fp16.  Pinned by tests/golden/sliding_window_half.npz (made by
 *                           the reference's own predictor through networks that return fp16).  Served by the gather
 *                           path only (any number of classes: passes of 63 heads; FNN_E_UNSUPPORTED where that path
 *                           cannot run - a last stage of more than 32 channels, fp32 output - and for the
 *                           accumulate_* entry points). */
enum { FNN_ACC_FP16_REFERENCE = 0, FNN_ACC_FP32 = 1, FNN_ACC_FP16_AUTOCAST = 2 };
enum { FNN_OUT_F16 = 0, FNN_OUT_F32 = 1 };

/* Network topology: what the reference builds from plans.json + checkpoint
 * (PlainConvUNet / LiteNNUNetStudent: nnUNetDistillationTrainer.py:74-177,
 *  get_network_from_plans.py:9-43).  Features are already reduced for a student. */
typedef struct fnn_arch_desc {
    int32_t kind;                               /* FNN_NET_*                                  */
    int32_t n_stages;
    int32_t in_channels;
    int32_t num_heads;                          /* LabelManager.num_segmentation_heads        */
    int32_t features[FNN_MAX_STAGES];
    int32_t kernels[FNN_MAX_STAGES][3];         /* per-axis 1 or 3                            */
    int32_t strides[FNN_MAX_STAGES][3];         /* per-axis 1 or 2; stage 0 must be 1,1,1     */
    int32_t n_conv_enc[FNN_MAX_STAGES];         /* convs (plain) / residual blocks (resenc)   */
    int32_t n_conv_dec[FNN_MAX_STAGES];
    int32_t patch[3];                           /* ConfigurationManager.patch_size            */
    float eps;                                  /* InstanceNorm eps (1e-5)                    */
    float slope;                                /* LeakyReLU negative slope (0.01)            */
    int32_t spatial_dims;                       /* 0 or 3: 3-D configuration.  2: a `2d` configuration
                                                 * (Conv2d network, patch_size with two entries): patch[0],
                                                 * kernels[s][0] and strides[s][0] are 1 and EVERY slice of
                                                 * the first image axis is a tile position
                                                 * (predict_from_raw_data.py:508-524)               */
    int32_t precision;                          /* FNN_PREC_*                                  */
} fnn_arch_desc;

/* Knobs of nnUNetPredictor.__init__ (:40-65) + engine-side choices. */
typedef struct fnn_opts {
    double tile_step_size;                      /* 0 < s <= 1; a double like the reference's Python float:
                                                 * ceil((image - patch) / (patch * step)) of a float-rounded
                                                 * 0.3 or 0.7 gives another patch count (sliding_window_prediction.py:38-41) */
    int32_t use_gaussian;
    int32_t n_mirror_axes;                      /* 0 = no test-time mirroring                 */
    int32_t mirror_axes[3];                     /* spatial axes 0..2                          */
    int32_t accum;                              /* FNN_ACC_*                                  */
    int32_t out_dtype;                          /* FNN_OUT_*                                  */
    int32_t batch;                              /* patches per forward, <= max_batch          */
    void *stream;                               /* hipStream_t; NULL = the null stream        */
} fnn_opts;

typedef struct fnn_engine fnn_engine;

/* ---- lifecycle ----------------------------------------------------------- */
int fnn_abi_version(void);
const char *fnn_last_error(const fnn_engine *e);             /* e may be NULL */

/* Replaces network construction in initialize_from_trained_model_folder
 * (:104-118) / manual_initialization (:131-154).  max_batch: patches per forward the engine is planned for (the kernel
 * variants are chosen for it): 1 .. 64, or - small patches - as many as give a forward 2^27 voxels, at most 512. */
int fnn_create(const fnn_arch_desc *arch, int device, int max_batch, fnn_engine **out);
void fnn_destroy(fnn_engine *e);

/* Number of float32 values fnn_load_weights expects and the canonical order.
 * FNN_NET_PLAIN: for every encoder stage s, conv i: weight[F,Cin,kd,kh,kw], bias[F], gamma[F],
 * beta[F]; then for every decoder level d (deepest first): transpconv
 * weight[Cbelow,F,sd,sh,sw], bias[F]; its convs as above; finally the last
 * seg layer weight[heads,F0], bias[heads].
 * FNN_NET_RESENC (ResidualEncoderUNet, n_conv_enc = residual blocks per stage): the
 * stem conv (weight, bias, gamma, beta); then per stage, per block: conv1
 * (weight, bias, gamma, beta), conv2 (same), and - only when the block changes
 * the channel count - the 1x1x1 skip projection (weight[F,Cin], gamma, beta; it
 * has no bias); decoder and seg layer as above. */
int64_t fnn_weight_count(const fnn_engine *e);

/* Replaces network.load_state_dict(params) per fold (:486-489).  `blob` is
 * host memory in the order above.  Folds are kept resident on the device. */
int fnn_load_weights(fnn_engine *e, int fold, const float *blob, int64_t count);

/* The fp16 importance map of compute_gaussian (sliding_window_prediction.py:10-27),
 * patch[0]*patch[1]*patch[2] IEEE-half bit patterns, computed by the host
 * mirror with scipy exactly as the reference does. */
int fnn_set_gaussian(fnn_engine *e, const uint16_t *half_bits, int64_t count);

/* ---- the hot path -------------------------------------------------------- */
/* Replaces predict_sliding_window_return_logits (:634-680) for fold `fold`:
 * pad -> slicers -> per patch network (+mirroring) -> weighted accumulate ->
 * normalise -> un-pad.  vol: float32 [C,X,Y,Z]; out: [heads,X,Y,Z] of
 * opts->out_dtype.  FNN_E_INF if the normalised logits contain inf.
 * vol may be DEVICE memory (read in place) or HOST memory - what the reference's callers hold
 * (`data = data.to(results_device)`, :579): a host volume is uploaded in tiles (planes x rows) on a copy stream of the
 * engine's and a batch of patches starts when the tiles under its patches have landed (the patch order is
 * x-major, :532-537); pinned host memory is read by DMA directly, pageable memory through pinned staging. */
int fnn_predict_volume(fnn_engine *e, int fold, const float *vol, const int64_t shape[4],
                       const fnn_opts *opts, void *out_logits);

/* Replaces predict_logits_from_preprocessed_data (:471-504): mean over the
 * loaded folds [0, n_folds) - summed on the device instead of the reference's
 * per-fold device->host hop (:494-497). */
int fnn_predict_volume_ensemble(fnn_engine *e, int n_folds, const float *vol, const int64_t shape[4],
                                const fnn_opts *opts, void *out_logits);

/* One forward of the network on `n` patches (the `self.network(x)` call at
 * :545): x float32 [n,C,px,py,pz] -> logits float32 [n,heads,px,py,pz].
 * Used by parity tests and by callers that bring their own tiling. */
int fnn_forward_patches(fnn_engine *e, int fold, const float *x, int n, float *logits, void *stream);

/* How the label-map entry points turn logits into labels (engine state, default
 * ARGMAX / U8) - LabelManager.convert_logits_to_segmentation, label_handling.py:144-195:
 *   FNN_LABELS_ARGMAX  plain labels: argmax over heads, first maximum wins (:176-180);
 *   FNN_LABELS_REGIONS region-based training: label = 0, then for i in order
 *                      `if sigmoid(float(logit_i)) > 0.5: label = regions_class_order[i]`
 *                      (:163-170; n_regions must equal num_heads).
 * label_dtype follows export_prediction.py:45-46: uint8 when the dataset has
 * < 255 foreground labels, else uint16. */
enum { FNN_LABELS_ARGMAX = 0, FNN_LABELS_REGIONS = 1 };
enum { FNN_LABEL_U8 = 0, FNN_LABEL_U16 = 1 };
int fnn_set_label_rule(fnn_engine *e, int mode, const int32_t *regions_class_order, int n_regions, int label_dtype);

/* Label map without materialising the logits: for one fold the labels are taken
 * straight from the accumulators (divide, round to fp16, then the label rule -
 * exactly what convert_logits_to_segmentation would see); for several folds the
 * ensemble logits are formed first.  Replaces the reference's full-logit D2H
 * copy + numpy argmax (:386, label_handling.py:173-180).  labels: uint8 or
 * uint16 [X,Y,Z] per fnn_set_label_rule. */
int fnn_predict_labels(fnn_engine *e, int n_folds, const float *vol, const int64_t shape[4],
                       const fnn_opts *opts, void *labels);

/* Multi-GPU building blocks (not in the reference, whose only
 * inference parallelism is case-level -num_parts/-part_id, :918-925).
 * Accumulators are channels-last DEVICE buffers - fp32 when opts->accum is
 * FNN_ACC_FP32, fp16 otherwise - that cover a box of the
 * PADDED volume:  acc[bx][by][bz][HP],  HP = fnn_accumulator_channels(e)
 * (= num_heads + 1 rounded up to 8); channel h < num_heads holds sum(w * logit_h),
 * channel num_heads holds sum(w).  The caller zeroes them, exchanges / adds the
 * overlap regions between ranks, then normalises the part it owns.
 * fnn_accumulate_patches runs the listed patches (indices into the x-major patch
 * list of fnn_plan_volume; each must lie inside the box) and ADDS into acc.
 * fnn_normalize_box divides, un-pads and writes the un-padded box
 * [out_lo, out_hi) into `out`, the full [heads][X][Y][Z] tensor of opts->out_dtype. */
int64_t fnn_accumulator_channels(const fnn_engine *e);
int fnn_accumulate_patches(fnn_engine *e, int fold, const float *vol, const int64_t shape[4],
                           const fnn_opts *opts, const int64_t *patch_ids, int64_t n_ids,
                           const int64_t box_lo[3], const int64_t box_hi[3], void *acc);
int fnn_normalize_box(fnn_engine *e, const void *acc, const int64_t shape[4], const fnn_opts *opts,
                      const int64_t box_lo[3], const int64_t box_hi[3],
                      const int64_t out_lo[3], const int64_t out_hi[3], void *out_logits);

/* fnn_normalize_box's label-map twin: divide, round to fp16, apply the engine's label rule
 * (fnn_set_label_rule) and write the un-padded box [out_lo, out_hi) into `labels`, the full
 * [X][Y][Z] uint8 / uint16 map - what convert_logits_to_segmentation (label_handling.py:144-195)
 * would produce from that box's logits.  A rank of the sharded predictor labels the box it owns
 * and only the label slabs (not the logits) travel between the GPUs. */
int fnn_labels_box(fnn_engine *e, const void *acc, const int64_t shape[4], const fnn_opts *opts,
                   const int64_t box_lo[3], const int64_t box_hi[3],
                   const int64_t out_lo[3], const int64_t out_hi[3], void *labels);

/* The same sharding on the gather path (csrc/gather.hip; no accumulators, bit-identical to one GPU): a rank keeps the
 * last activation of the patches it ran - fnn_patch_features writes them, [n_ids][P][C] fp16 with P = patch voxels and
 * C = fnn_feature_channels(e), and their InstanceNorm rows [n_ids][2][C] float32, into DEVICE buffers of the caller -
 * receives from its neighbours the parts of THEIR patches that reach into the box it owns (caller's job: plain copies
 * of [dx][dy][dz][C] sub-blocks), and fnn_gather_box then forms every voxel of the un-padded box [out_lo, out_hi) from
 * all patches that cover it, in the reference's visiting order: slot_of_patch[pid] (HOST array over the x-major patch
 * list of fnn_plan_volume) = where patch pid's activation sits in feat / fss, or -1 where the caller knows the patch
 * does not reach the box.  Writes fp16 logits [heads][X][Y][Z] and / or labels [X][Y][Z] (either may be NULL) of the
 * full-size tensors.
 * Slots and mirroring (ABI 3): the caller's buffers hold `n_slots` patch slots per evaluation - feat
 * [n_eval][n_slots][P][C], fss [n_eval][n_slots][2][C], n_eval = 2^(opts->n_mirror_axes) in the order of
 * _internal_maybe_mirror_and_predict (predict_from_raw_data.py:541-557; 1 without mirroring); fnn_patch_features
 * writes patch patch_ids[i] into slot slot0 + i of every evaluation.  The activation of a mirrored evaluation is
 * stored in the network's (flipped) coordinates: voxel v of the patch sits at P - 1 - v along every flipped axis, which
 * is where a caller that moves sub-blocks between ranks has to put them (fast-nnunet_amd/dist.py, FeatureExchange). */
int64_t fnn_feature_channels(const fnn_engine *e);
int fnn_patch_features(fnn_engine *e, int fold, const float *vol, const int64_t shape[4], const fnn_opts *opts,
                       const int64_t *patch_ids, int64_t n_ids, void *feat, float *fss, int64_t slot0, int64_t n_slots);
int fnn_gather_box(fnn_engine *e, int fold, const void *feat, const float *fss, const int32_t *slot_of_patch,
                   int64_t n_slots, const int64_t shape[4], const fnn_opts *opts, const int64_t out_lo[3],
                   const int64_t out_hi[3], void *out_logits, void *labels);

/* The exchange between fnn_patch_features and fnn_gather_box as ONE launch per peer and direction (ABI 4; not in the
 * reference, whose only inference parallelism is case level: predict_from_raw_data.py:918-925).  A rank
 * sends each neighbour the sub-blocks of its kept activations that reach into the neighbour's owned box:
 * fnn_pack_regions copies `n` sub-blocks of feat ([n_eval][n_slots][PD][PH][PW][C] fp16, C = fnn_feature_channels) into
 * one contiguous message buffer, fnn_unpack_regions lands a received message in the slots of the foreign patches.
 * `regions`: a DEVICE table (built once per decomposition, it does not change from volume to volume of one shape) of
 * fnn_region records - evaluation, slot, the sub-block [lo, hi) in the slot's own (stored, i.e. flipped for a mirrored
 * evaluation) voxel coordinates, and where the block sits in the message in units of 16 bytes; blocks are [d][h][w][C]
 * contiguous.  Both run on `stream` without synchronising it; every pointer is device memory. */
typedef struct fnn_region {
    int32_t eval, slot;
    int32_t lo[3], hi[3];
    int32_t off16;                  /* block start in the message buffer, 16-byte units */
    int32_t reserved;
} fnn_region;
int fnn_pack_regions(fnn_engine *e, const void *feat, int64_t n_slots, const fnn_region *regions, int64_t n,
                     void *message, void *stream);
int fnn_unpack_regions(fnn_engine *e, void *feat, int64_t n_slots, const fnn_region *regions, int64_t n,
                       const void *message, void *stream);

/* LabelManager.convert_logits_to_segmentation on resident logits with the
 * engine's label rule: logits [heads, n_vox] f16/f32 -> labels uint8/uint16. */
int fnn_argmax_labels(fnn_engine *e, const void *logits, int dtype, int heads, int64_t n_vox,
                      void *labels, void *stream);

/* ---- whole-volume steps around the sliding window ----
 * DefaultPreprocessor.run_case_npy WITHOUT its resampling call
 * (preprocessing/preprocessors/default_preprocessor.py:45-93): float32 image
 * [C][s0][s1][s2] -> transpose_forward -> crop to the bounding box of the non-zero
 * region (preprocessing/cropping/cropping.py:7-39) -> per-channel intensity
 * normalisation (preprocessing/normalization/default_normalization_schemes.py).
 * Device pointers only.  Resampling to the target spacing is fnn_resample. */
enum { FNN_NORM_NONE = 0,        /* NoNormalization                                   :70-74 */
       FNN_NORM_ZSCORE = 1,      /* ZScoreNormalization, whole-image branch            :45-49 */
       FNN_NORM_CT = 2,          /* CTNormalization: clip, - mean, / max(std, 1e-8)    :53-67 */
       FNN_NORM_RESCALE01 = 3,   /* RescaleTo01Normalization                           :77-84 */
       FNN_NORM_RGB01 = 4 };     /* RGBTo01Normalization                               :87-98 */
typedef struct fnn_norm_desc {
    int32_t scheme;              /* FNN_NORM_*                                                 */
    float mean, std;             /* CT: intensityproperties['mean'], ['std']                   */
    float lower, upper;          /* CT: ['percentile_00_5'], ['percentile_99_5']               */
    int32_t use_mask;            /* ZScore: use_mask_for_norm - statistics and normalisation only inside
                                  * binary_fill_holes(non-zero mask) (cropping.py:7-39, schemes :36-44)   */
} fnn_norm_desc;

/* bbox[2a], bbox[2a+1] = [lo, hi) along TRANSPOSED axis a of the voxels where any
 * channel is non-zero (the full extent for an all-zero image): properties
 * ['bbox_used_for_cropping'] of the reference. */
int fnn_nonzero_bbox(const float *raw, const int64_t shape[4], const int32_t transpose_forward[3],
                     int64_t bbox[6], void *stream);
/* out: float32 [C][bbox extents] = normalise(crop(transpose(raw))); norm[C]. */
int fnn_preprocess(const float *raw, const int64_t shape[4], const int32_t transpose_forward[3],
                   const int64_t bbox[6], const fnn_norm_desc *norm, float *out, void *stream);
/* The label half of convert_predicted_logits_to_segmentation_with_correct_shape
 * (inference/export_prediction.py:43-53): seg [bbox extents] (FNN_LABEL_U8 / U16)
 * -> zeros of shape_before_cropping with seg inserted at bbox -> transpose_backward.
 * out: [shape_before_cropping[transpose_backward[j]] for j in 0..2]. */
int fnn_revert_labels(const void *seg, int label_dtype, const int64_t bbox[6],
                      const int64_t shape_before_cropping[3], const int32_t transpose_backward[3],
                      void *out, void *stream);

/* convert_predicted_logits_to_segmentation_with_correct_shape(..., return_probabilities=True)
 * after its resampling step (inference/export_prediction.py:36-70): logits [heads][bbox
 * extents] (FNN_OUT_F16 / FNN_OUT_F32) -> apply_inference_nonlin (fp32 softmax over the heads;
 * sigmoid when regions_class_order != NULL, label_handling.py:125-139), the label rule on the
 * probabilities (:163-181), revert_cropping_on_probabilities (:197-221: background probability
 * 1 outside the box for plain labels, 0 for regions), both transposes back.
 * probs: float32 [heads][shape_before_cropping[transpose_backward[j]]]; labels: the same grid. */
int fnn_export_probabilities(const void *logits, int logits_dtype, int heads,
                             const int32_t *regions_class_order, const int64_t bbox[6],
                             const int64_t shape_before_cropping[3], const int32_t transpose_backward[3],
                             float *probs, void *labels, int label_dtype, void *stream);

/* resample_data_or_seg(..., is_seg=False) of the reference
 * (preprocessing/resampling/default_resampling.py:113-196): per channel
 * skimage.transform.resize(order, mode='edge', anti_aliasing=False) - evaluated as
 * scipy.ndimage.zoom(float64, out/in, order, mode='nearest', grid_mode=True) + clip
 * to the input range - or, with separate_axis >= 0, that resize per 2-D slice and
 * an order-0 pass along the axis (:147-188).  in / out: [C][...] of `dtype`
 * (FNN_OUT_F32 for images, FNN_OUT_F16 for fp16 logits), device pointers.
 * The caller decides separate_axis (determine_do_sep_z_and_axis, :34-71). */
typedef struct fnn_resample_desc {
    int32_t order;               /* 0, 1 or 3 (resampling_fn_*_kwargs['order'])               */
    int32_t separate_axis;       /* -1, or the anisotropic axis                                */
    int32_t order_z;             /* 0 (the only value the reference's plans use)               */
    int32_t dtype;               /* FNN_OUT_F16 / FNN_OUT_F32                                  */
} fnn_resample_desc;
int fnn_resample(const void *in, const int64_t shape[4], const int64_t new_shape[3],
                 const fnn_resample_desc *desc, void *out, void *stream);

/* ---- host-side integer logic (no GPU needed) ------------------------------ */
/* compute_steps_for_sliding_window (sliding_window_prediction.py:30-54) for one
 * axis; returns the number of steps written (<= cap) or a negative error. */
int fnn_compute_steps(int64_t image_size, int64_t patch_size, double step, int64_t *steps, int cap);

/* Padded shape, low-side pad, number of patches and (optionally) the patch
 * origins [n][3] in the reference's visit order - x slowest, z fastest - for a
 * volume (pad_nd_image use at :657-659 and the slicer loop :525-537). */
/* patch[0] == 0 denotes a 2-D configuration (patch = {0, py, pz}): every slice of the first axis. */
int fnn_plan_volume(const int32_t patch[3], const int64_t shape_sp[3], double step, int64_t padded[3],
                    int64_t pad_lo[3], int64_t *n_patches, int32_t *origins, int64_t origins_cap);

/* The weight quantiser of FNN_PREC_F8: float32 -> OCP e4m3 ("fn": +-448 saturating, 0x7f = NaN), round to nearest
 * even - the encoding v_mfma_f32_16x16x32_fp8_fp8 reads on gfx950 (not MI300's fnuz).  Exposed so that the host-side
 * packing can be checked against another implementation without a GPU. */
int fnn_fp8_e4m3_encode(const float *in, int64_t n, uint8_t *out);

/* ---- timing / introspection ----------------------------------------------- */
/* Per-kernel-family device time of the last fnn_predict_volume call, measured
 * with HIP events on the launch stream when profiling is enabled. */
typedef struct fnn_profile {
    double total_ms;
    double conv_ms, stem_ms, tconv_ms, head_ms, finalize_ms;
    int64_t conv_launches;
    double conv_flops;                           /* algorithmic 2*MACs of the timed convs     */
    int64_t n_patches;
    double conv_bytes;                           /* algorithmic HBM bytes of the timed convs: every input
                                                  * element read once, every output element written once */
} fnn_profile;
int fnn_set_profiling(fnn_engine *e, int enabled);
int fnn_get_profile(const fnn_engine *e, fnn_profile *out);
/* Which kernel variant served every launch of the last call made while profiling was on (one line per launch, in launch
 * order: "conv3d_zr_kernel<2,8>", "conv_row_kernel<6,1,0>", ...): the launchers choose variants from the layer shape and
 * the planned batch, and a silent fall-back to a generic kernel is otherwise invisible.  Returns the bytes needed
 * (with the terminating 0); writes at most `cap`.  No counterpart in the reference. */
int64_t fnn_kernel_log(const fnn_engine *e, char *buf, int64_t cap);

/* Per-launch rows of the same profiled call (measurement aid of bench.py --plan / tools/plan_sweep.py; additive in ABI 4,
 * no counterpart in the reference): one line per timed launch, tab-separated -
 *   layer index (-1: seg head / gather / finalize), family (conv | stem | tconv | head | finalize), milliseconds between
 *   the launch's two HIP events, algorithmic 2*MACs, algorithmic HBM bytes, kernel variant(s) as in fnn_kernel_log.
 * fnn_layer_table describes the layers those indices name: index, type, input channels, output channels, kernel, stride,
 * input dims, output dims, 2*MACs per patch, algorithmic bytes per patch, 1 = computed inside its consumer / 2 = a
 * consumer that recomputes its producer / 0.  Both return the bytes needed and write at most `cap`. */
int64_t fnn_profile_launches(const fnn_engine *e, char *buf, int64_t cap);
int64_t fnn_layer_table(const fnn_engine *e, char *buf, int64_t cap);

/* Algorithmic work of one patch forward: 2*MACs of convs, transposed convs and
 * the seg head; ideal fp16 activation bytes (each activation written once and
 * read once).  SURVEY.md 8d. */
int fnn_patch_work(const fnn_engine *e, double *flops, double *act_bytes);

/* ---- single-op entry points (parity tests call the kernels through these) -- */
/* Host float32 NCDHW in / out; the library converts to its device layout
 * (fp16, channels-last), runs the HIP kernel, converts back.
 * Input transform fused into the kernel's load path: when gammaK != NULL the
 * input K is treated as a RAW conv output: InstanceNorm3d(eps=1e-5, affine)
 * with these gamma/beta is applied (statistics of the fp16-rounded input,
 * computed by the wrapper the way a producer kernel's epilogue would), then
 * LeakyReLU(slopeK).  gammaK == NULL = identity.  A second input (x2 != NULL)
 * is the concat-free replacement of torch.cat((x, x2), 1).
 * stats_out (may be NULL): per (n, cout) sum and sum of squares of the
 * fp16-rounded outputs, doubles [n][cout][2]. */
int fnn_op_conv3d(int device, int n, const int dims[3],
                  const float *x, int cin, const float *gamma1, const float *beta1, float slope1,
                  const float *x2, int cin2, const float *gamma2, const float *beta2, float slope2,
                  const float *w, const float *bias, int cout, const int k[3], const int stride[3],
                  float *y, double *stats_out);
int fnn_op_conv_transpose3d(int device, int n, const int dims[3],
                            const float *x, int cin, const float *gamma1, const float *beta1, float slope1,
                            const float *w, const float *bias, int cout, const int stride[3], float *y);

/* The kernel variants the last fnn_op_conv3d / fnn_op_conv_transpose3d call of this thread launched, one per line (the
 * launchers pick a variant from the layer's shape; the op tests pin which one a case exercises).  Returns the size needed. */
int fnn_op_last_kernels(char *buf, int cap);

/* The shader clock the device holds while other work runs (measurement aid, no counterpart in the reference; additive in
 * ABI 4): fnn_clock_probe_start launches ONE wave on a stream of its own that sleeps between reads of the shader-clock
 * counter (s_memtime) and the constant 100 MHz counter (s_memrealtime) until max_seconds have passed on the latter or
 * fnn_clock_probe_stop raises a flag in mapped host memory; stop waits for it and returns elapsed shader cycles / elapsed
 * time in GHz.  (A device-wide synchronisation waits for the probe like for any kernel: give it less time than the
 * region it samples.)  bench.py runs one extra, untimed step beside it: under sustained MFMA load the chip lowers its
 * clock, and a roofline fraction priced at 2.4 GHz does not say how much of what the clock allows a
 * kernel uses.  The probe is not free: next to it per-launch durations stretch by ~4 %, and with HIP's default of four
 * hardware queues it can share a queue with a caller's stream, whose work then waits for it (GPU_MAX_HW_QUEUES=8). */
int fnn_clock_probe_start(int device, double max_seconds, void **probe);
int fnn_clock_probe_stop(void *probe, double *ghz, double *seconds);

/* Self-check of the closing division of the seg-head gather (predicted_logits /= n_predictions,
 * predict_from_raw_data.py:619): runs the kernel's shared-reciprocal quotient over every fp16 value a and every fp16
 * b with a clear sign bit and counts the pairs whose fp16 result differs from IEEE fp32 division rounded to fp16
 * (counts[0], must be 0), the pairs that took the fast route at all (counts[1]) and one differing pair as
 * a_bits | b_bits << 16 (counts[2], 0 when there is none). */
int fnn_op_quotient_check(int device, unsigned long long counts[3]);

#ifdef __cplusplus
}
#endif
#endif /* FNN_H */
