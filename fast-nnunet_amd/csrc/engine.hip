// engine.hip - host side of the C ABI in include/fnn.h.
//
// Owns: the layer plan derived from fnn_arch_desc (what the reference builds
// in PlainConvEncoder / UNetDecoder, nnUNetDistillationTrainer.py:141-173),
// fp16 weight packing into MFMA fragment order, the activation / statistics /
// accumulator arenas in HBM, and the sliding-window driver that replaces
// predict_from_raw_data.py:560-680 (x-major patch order, Gaussian weighting,
// accumulate, normalise, un-pad; mirroring; fold ensembling).
#include "fnn_device.h"
#include "../../include/fnn.h"

#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <algorithm>
#include <climits>
#include <vector>
#include <thread>

namespace {

thread_local std::string g_err;

struct Layer {
    enum Type { STEM, CONV, TCONV, POOL, COMBINE, GATHER } type;   // GATHER: the patches' input windows as an fp16 tensor (stem on the conv kernels)
    int n_src = 1;
    int cin_real[2] = {0, 0}, cin_pad[2] = {0, 0};
    int cout_real = 0, cout_pad = 0;
    int k[3] = {1, 1, 1}, s[3] = {1, 1, 1};
    int in_dims[3], out_dims[3];
    int src_layer[2] = {-1, -1};      // producing layer (-1 = the volume)
    bool has_norm = true;             // a conv's output is normalised by its consumers; a tconv's / pool's / block's is not
    bool act = true;                  // LeakyReLU after the norm (false: conv2 and the skip projection of a residual block)
    bool has_bias = true;             // the skip projection of a residual block has no bias
    // offsets
    size_t w_off = 0;                 // halves, packed weights (STEM: floats in fparam)
    size_t bias_off = 0, gamma_off = 0, beta_off = 0;   // floats
    size_t stats_off = 0;             // doubles
    size_t ss_off = 0;                // float2 (scale, shift) per channel
    size_t out_off = 0;               // halves, activation arena (for batch = 1)
    int64_t blob_w = 0, blob_b = 0, blob_g = 0, blob_beta = 0;
    int chunks = 0, ksteps = 0, packing = 0;
    int stats_slots = FNN_STAT_REPL;  // rows per item in the stats buffer: atomics' replicas, or one row per tile
    // conv3d_thin.hip: the stem as one MFMA per 16 voxels (w_off2 = its weight fragment in wpk); a CONV that recomputes
    // its producer while staging (fuse = FUSE_STEM / FUSE_TCONV); a producer whose output is never written (virtual)
    bool mfma_stem = false, virtual_out = false;
    int pool_layer = -1;              // COMBINE: the POOL layer (the next stage's skip path) whose output this launch writes too; that POOL has pool_fused
    bool pool_fused = false;
    bool chunk_major = false;         // output stored [C / 16][voxels][16] (fnn_device.h, SrcDesc): convs / transposed convs whose consumers are convs
    bool fp8 = false;                 // conv3d_zr8_kernel: e4m3 operands; oscale_off = per-cout output scales (floats)
    size_t oscale_off = 0;
    int fuse = 0;
    size_t w_off2 = 0;
    double flops = 0;                 // 2*MACs per patch
    double bytes = 0;                 // algorithmic HBM bytes per patch: every input read once + the output written once (fp16)
};

struct FoldWeights {
    f16 *wpk = nullptr;
    float *fparam = nullptr;
    bool loaded = false;
};

inline int pad16(int c) { return (c + 15) / 16 * 16; }
// z pitch of the volume accumulators: rows start 16-byte aligned for the vectorised read-modify-write
inline long long zpitch(long long z) { return (z + 7) / 8 * 8; }

}  // namespace

struct fnn_engine {
    fnn_arch_desc arch;
    int device = 0;
    int max_batch = 1;
    std::string err;
    bool fuse_enabled = true;               // FNN_NO_FUSE (read when the engine is created) keeps every layer a kernel of its own
    // FNN_FUSE_STEM / FNN_FUSE_TCONV = 0 | 1.  The transposed-conv fusion is on (+1.2 % on the benchmark).  The stem
    // fusion is on where the row-streaming kernels take both halves (conv_row_stem_kernel + stem_row_kernel's statistics
    // pass, conv3d_row.hip); in tile form (FNN_FUSE_STEM=1 forces it) it is built and tested but slower than two kernels:
    // its consumer is instruction-bound (one MFMA per 16 halo voxels needs ~45 instructions around it), 1020 + 340 us
    // per batch against 575 + 756 us unfused.
    int fuse_stem = -1;                     // -1: where the row kernels run it
    bool fuse_tconv = true;
    std::vector<Layer> layers;
    int head_src = -1;                      // layer feeding the seg head
    int hblocks = 0, head_ksteps = 0;
    size_t head_w_off = 0, head_bias_off = 0;
    int n_gpass = 1;                        // gather passes of <= 63 heads (GatherParams::n_pass); > 1: their own packs below
    size_t gpass_w_off = 0, gpass_bias_off = 0;
    int64_t blob_head_w = 0, blob_head_b = 0;
    int64_t blob_count = 0;
    size_t wpk_halves = 0, fparam_floats = 0, stats_doubles = 0, act_halves = 0, ss_count = 0;
    std::vector<FoldWeights> folds;
    f16 *act = nullptr;
    double *stats = nullptr;
    float *ss = nullptr;                    // [layer][max_batch][2][C]
    // Batches in flight (fnn_accumulate / predict): one activation arena and one internal stream per batch in flight
    // (four by default since round 6, FNN_PIPES = 2..8; measured 2 -> 3: +1.5 %, 3 -> 4: +0.3 ... +0.9 %, beyond: nothing).  The network alternates between
    // HBM-bound (thin full-resolution convs, seg head) and MFMA-bound kernels; with the following batches' forwards on
    // the other streams they overlap.  The heads stay ordered (events), so the accumulation order - and with it every
    // rounding - is the reference's.  288 GB of HBM make the extra arenas free.
    static constexpr int MAXP = 8;
    int n_pipe = 0;                         // arenas / streams allocated (0 until the first multi-batch run)
    f16 *actp[MAXP] = {}; double *statsp[MAXP] = {}; float *ssp[MAXP] = {};      // [0] aliases act / stats / ss
    hipStream_t pipe[MAXP] = {};
    hipEvent_t ev_start = nullptr, ev_head[MAXP] = {}, ev_done[MAXP] = {};
    f16 *gauss = nullptr;
    f16 *ones = nullptr;                    // weight map of use_gaussian = 0 (the kernels load the map unconditionally)
    int *inf_flag = nullptr;
    // label rule of the label-map entry points (fnn_set_label_rule)
    int label_mode = FNN_LABELS_ARGMAX, label_u16 = 0;
    int *label_order = nullptr;             // device: regions_class_order[num_heads]
    int *origins = nullptr; size_t origins_cap = 0;
    int *origins_host = nullptr; size_t origins_host_cap = 0; hipEvent_t origins_ev = nullptr;   // pinned staging, as for steps_dev below
    void *acc = nullptr; size_t acc_bytes = 0;
    float *vol_tmp = nullptr; size_t vol_tmp_bytes = 0;
    // A volume that arrives in HOST memory (the reference's callers hand over the CPU tensor of the preprocessing iterator,
    // predict_from_raw_data.py:579 `data = data.to(results_device)`): it is uploaded in tiles (planes x rows) on a copy stream of
    // the engine's, and a batch of patches starts as soon as the tiles under its patches have landed - the patch order is
    // x-major, y next (:532-537), so the tiles are needed roughly in the order they travel.  Pinned source: DMA straight from it; pageable
    // source: through a ring of pinned staging buffers filled by a few host threads.
    struct Upload {
        bool active = false, pinned_src = false;
        const float *host = nullptr; float *dev = nullptr;
        int C = 0; int64_t X = 0, Y = 0, Z = 0;
        // the volume travels in tiles of slab_x planes x slab_y rows x the whole z extent (x-major patch order: the first batch's
        // patches cover the first x layer and a part of y - it starts when THOSE tiles have landed, not the whole layer)
        int64_t slab_x = 1, slab_y = 1, nxs = 0, nyb = 0;
        std::vector<char> issued;              // [nxs][nyb]
        size_t n_issued = 0;
        std::vector<hipEvent_t> landed;        // event pool (kept between calls): one per upload_box call that issued something
        size_t ev_next = 0;
        hipEvent_t last = nullptr;             // the youngest recorded event: everything issued so far lies in front of it
        hipStream_t st = nullptr;
        static constexpr int RING = 3;
        float *stage[RING] = {}; size_t stage_bytes = 0; hipEvent_t stage_free[RING] = {}; bool stage_used[RING] = {};
        int next_stage = 0;
        hipEvent_t go = nullptr;
    } up;
    float *vol_pad = nullptr; size_t vol_pad_bytes = 0;
    void *out_tmp = nullptr; size_t out_tmp_bytes = 0;
    float *patch_buf = nullptr; size_t patch_buf_bytes = 0;
    // gather path (gather.hip): the last conv's raw output and InstanceNorm of every patch of the volume
    void *feat = nullptr; size_t feat_bytes = 0;
    void *featss = nullptr; size_t featss_bytes = 0;
    void *featssh = nullptr; size_t featssh_bytes = 0;   // the same rows in fp16, staging layout (GatherParams::fssh)
    int *steps_dev = nullptr; size_t steps_cap = 0;
    int *steps_host = nullptr; size_t steps_host_cap = 0;   // pinned staging of the tile-start / slot tables (no stream sync for a host temporary)
    hipEvent_t steps_ev = nullptr;                          // the previous upload from steps_host has been read
    std::vector<std::string> klog;          // kernel variants of the last profiled call, one entry per launch (fnn_kernel_log)
    bool gather_enabled = true;             // FNN_NO_GATHER (read when the engine is created): always accumulate in HBM
    double head_flops = 0, patch_flops = 0, patch_act_bytes = 0;
    // profiling
    bool profiling = false;
    struct Ev { hipEvent_t a, b; int family; double flops, bytes; int layer; size_t klog_lo, klog_hi; };
    std::vector<Ev> evs; size_t ev_used = 0;
    int cur_layer = -1;                     // layer index of the launches being issued (-1: head / gather / finalize)
    std::vector<std::string> launch_rows;   // fnn_profile_launches: one row per timed launch of the last profiled call
    fnn_profile prof{};
};

// message of an entry point that has no engine handle (prep.hip)
void fnn_set_global_error(const char *msg) { g_err = msg ? msg : ""; }

namespace {

int fail(fnn_engine *e, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    if (e) e->err = buf;
    return code;
}

#define HIPCHK(e, call)                                                                              \
    do {                                                                                             \
        hipError_t _r = (call);                                                                      \
        if (_r != hipSuccess) return fail(e, FNN_E_HIP, "%s failed: %s", #call, hipGetErrorString(_r)); \
    } while (0)

bool is_device_ptr(const void *p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

int ensure(fnn_engine *e, void **p, size_t *have, size_t need) {
    if (*have >= need && *p) return 0;
    if (*p) (void)hipFree(*p);
    *p = nullptr; *have = 0;
    HIPCHK(e, hipMalloc(p, need));
    *have = need;
    return 0;
}

// Small integer tables (patch origins, tile starts, a sharded caller's slot table) -> device without a stream
// synchronisation: the host copy lives in a pinned buffer the engine owns; the only wait is for the PREVIOUS upload from
// that buffer to have been read (an event that has long fired by the next call).
int upload_ints(fnn_engine *e, const int *src, size_t n, int **dev, size_t *dev_cap, int **host, size_t *host_cap,
                hipEvent_t *ev, hipStream_t st) {
    const size_t bytes = n * sizeof(int) + 64;
    {
        void *t = *dev;
        if (int rc = ensure(e, &t, dev_cap, bytes)) return rc;
        *dev = (int *)t;
    }
    if (!*ev) HIPCHK(e, hipEventCreateWithFlags(ev, hipEventDisableTiming));
    else HIPCHK(e, hipEventSynchronize(*ev));
    if (*host_cap < bytes) {
        if (*host) (void)hipHostFree(*host);
        *host = nullptr; *host_cap = 0;
        HIPCHK(e, hipHostMalloc((void **)host, bytes, hipHostMallocDefault));
        *host_cap = bytes;
    }
    memcpy(*host, src, n * sizeof(int));
    HIPCHK(e, hipMemcpyAsync(*dev, *host, n * sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHK(e, hipEventRecord(*ev, st));
    return 0;
}

// ---------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------
int build_plan(fnn_engine *e) {
    const fnn_arch_desc &a = e->arch;
    if (a.kind != FNN_NET_PLAIN && a.kind != FNN_NET_RESENC) return fail(e, FNN_E_UNSUPPORTED, "unknown network kind %d", a.kind);
    if (a.n_stages < 2 || a.n_stages > FNN_MAX_STAGES) return fail(e, FNN_E_INVALID, "n_stages out of range");
    if (a.in_channels < 1 || a.in_channels > 4096) return fail(e, FNN_E_UNSUPPORTED, "in_channels must be 1..4096");
    if (a.num_heads < 1 || a.num_heads > 256) return fail(e, FNN_E_UNSUPPORTED, "num_heads must be 1..256");
    for (int s = 0; s < a.n_stages; ++s)
        for (int d = 0; d < 3; ++d) {
            if (a.kernels[s][d] != 1 && a.kernels[s][d] != 3) return fail(e, FNN_E_UNSUPPORTED, "kernel sizes must be 1 or 3");
            if (a.strides[s][d] != 1 && a.strides[s][d] != 2) return fail(e, FNN_E_UNSUPPORTED, "strides must be 1 or 2");
            if (s == 0 && a.strides[s][d] != 1) return fail(e, FNN_E_UNSUPPORTED, "stage 0 must have stride 1");
        }
    if (a.spatial_dims != 0 && a.spatial_dims != 2 && a.spatial_dims != 3) return fail(e, FNN_E_INVALID, "spatial_dims must be 2 or 3");
    if (a.spatial_dims == 2) {
        if (a.patch[0] != 1) return fail(e, FNN_E_INVALID, "a 2-D configuration has patch[0] == 1");
        for (int s = 0; s < a.n_stages; ++s)
            if (a.kernels[s][0] != 1 || a.strides[s][0] != 1)
                return fail(e, FNN_E_INVALID, "a 2-D configuration has kernel and stride 1 along the first axis");
    }
    int dims[FNN_MAX_STAGES][3];
    for (int d = 0; d < 3; ++d) dims[0][d] = a.patch[d];
    for (int s = 1; s < a.n_stages; ++s)
        for (int d = 0; d < 3; ++d) {
            const int k = a.kernels[s][d], st = a.strides[s][d], pad = (k - 1) / 2;
            dims[s][d] = (dims[s - 1][d] + 2 * pad - k) / st + 1;
            if (dims[s][d] * st != dims[s - 1][d])
                return fail(e, FNN_E_INVALID, "patch size %d is not divisible by the pooling of axis %d", a.patch[d], d);
        }
    int64_t blob = 0;
    bool next_has_bias = true, next_act = true;
    auto add_conv = [&](Layer::Type type, int nsrc, const int cin[2], const int src[2], int cout, const int32_t *k,
                        const int32_t *s, const int *in_d, const int *out_d) {
        Layer L;
        L.type = type; L.n_src = nsrc; L.has_bias = next_has_bias; L.act = next_act;
        int cin_tot = 0;
        for (int i = 0; i < nsrc; ++i) {
            L.cin_real[i] = cin[i]; L.cin_pad[i] = (type == Layer::STEM) ? cin[i] : pad16(cin[i]);
            L.src_layer[i] = src[i]; cin_tot += cin[i];
        }
        L.cout_real = cout; L.cout_pad = pad16(cout);
        for (int d = 0; d < 3; ++d) { L.k[d] = k[d]; L.s[d] = s[d]; L.in_dims[d] = in_d[d]; L.out_dims[d] = out_d[d]; }
        const int T = k[0] * k[1] * k[2];
        L.blob_w = blob; blob += (int64_t)cout * cin_tot * T;
        L.blob_b = blob; if (L.has_bias) blob += cout;
        L.blob_g = blob; blob += cout;
        L.blob_beta = blob; blob += cout;
        L.flops = 2.0 * cout * cin_tot * T * out_d[0] * out_d[1] * out_d[2];
        L.bytes = 2.0 * ((double)cin_tot * in_d[0] * in_d[1] * in_d[2] * (type == Layer::STEM ? 2 : 1)     // fp32 volume
                         + (double)cout * out_d[0] * out_d[1] * out_d[2]);
        e->layers.push_back(L);
        return (int)e->layers.size() - 1;
    };
    const int32_t one[3] = {1, 1, 1};
    int prev = -1, prev_c = a.in_channels;
    int enc_last[FNN_MAX_STAGES];
    auto add_aux = [&](Layer::Type type, int src0, int src1, int channels, const int32_t *st, const int *in_d, const int *out_d) {
        Layer L;
        L.type = type; L.n_src = src1 >= 0 ? 2 : 1; L.has_norm = false;
        L.src_layer[0] = src0; L.src_layer[1] = src1;
        L.cin_real[0] = channels; L.cin_pad[0] = pad16(channels);
        L.cout_real = channels; L.cout_pad = pad16(channels);
        for (int d = 0; d < 3; ++d) { L.s[d] = st[d]; L.in_dims[d] = in_d[d]; L.out_dims[d] = out_d[d]; }
        e->layers.push_back(L);
        return (int)e->layers.size() - 1;
    };
    // The first conv of the network.  One input channel in 3-D: the stem kernels (conv3d_thin.hip / conv3d_row.hip: the window
    // of the fp32 volume in LDS, one MFMA per 16 voxels).  More channels (multi-modal MR, a cascade's one-hot channels) or a
    // `2d` configuration (depth-1 tiles: a sixteenth of that kernel's tile): the patches' windows are first written as an fp16
    // tensor with the channels padded to 16 (patch_input_kernel, misc.hip) and the stem is an ordinary conv layer on the MFMA
    // conv kernels - plan sweep, round 6: the generic multi-channel stem took 17 % (4 channels) to 55 % (14) of a forward.
    auto add_stem = [&](int cin, int cout, const int32_t *k, const int *d) {
        const bool via_conv = (cin >= 2 || a.spatial_dims == 2) && fnn_knob("FNN_STEM_DIRECT") == nullptr;   // (knob: A-B aid / the tests of the generic stem kernel)
        const int c[2] = {cin, 0};
        if (!via_conv) {
            const int src[2] = {-1, -1};
            return add_conv(Layer::STEM, 1, c, src, cout, k, one, d, d);
        }
        Layer G;
        G.type = Layer::GATHER; G.n_src = 1; G.has_norm = false;
        G.cin_real[0] = cin; G.cin_pad[0] = pad16(cin); G.cout_real = cin; G.cout_pad = pad16(cin);
        for (int q = 0; q < 3; ++q) { G.in_dims[q] = d[q]; G.out_dims[q] = d[q]; }
        e->layers.push_back(G);
        const int src[2] = {(int)e->layers.size() - 1, -1};
        const int li = add_conv(Layer::CONV, 1, c, src, cout, k, one, d, d);
        e->layers[li].bytes += 2.0 * cin * (double)d[0] * d[1] * d[2];                  // the network input is fp32 in HBM
        return li;
    };
    if (a.kind == FNN_NET_PLAIN) {
        for (int s = 0; s < a.n_stages; ++s) {
            if (a.n_conv_enc[s] < 1) return fail(e, FNN_E_INVALID, "n_conv_per_stage must be >= 1");
            for (int i = 0; i < a.n_conv_enc[s]; ++i) {
                const int cin[2] = {prev_c, 0}, src[2] = {prev, -1};
                const bool first = (s == 0 && i == 0);
                const int *in_d = (i == 0 && s > 0) ? dims[s - 1] : dims[s];
                prev = first ? add_stem(prev_c, a.features[s], a.kernels[s], dims[s])
                             : add_conv(Layer::CONV, 1, cin, src, a.features[s], a.kernels[s], i == 0 ? a.strides[s] : one, in_d, dims[s]);
                prev_c = a.features[s];
            }
            enc_last[s] = prev;
        }
    } else {
        // ResidualEncoderUNet: stem conv, then per stage n_conv_enc[s] BasicBlockD blocks
        //   y = LeakyReLU(norm(conv2(act(norm(conv1(x))))) + skip(x)),  skip = [AvgPool(stride)] [1x1x1 conv (no bias) + norm]
        {
            prev = add_stem(a.in_channels, a.features[0], a.kernels[0], dims[0]);
            prev_c = a.features[0];
        }
        for (int s = 0; s < a.n_stages; ++s) {
            if (a.n_conv_enc[s] < 1) return fail(e, FNN_E_INVALID, "n_blocks_per_stage must be >= 1");
            for (int b = 0; b < a.n_conv_enc[s]; ++b) {
                const int32_t *st = b == 0 ? a.strides[s] : one;
                const int *in_d = (b == 0 && s > 0) ? dims[s - 1] : dims[s];
                const bool strided = st[0] != 1 || st[1] != 1 || st[2] != 1;
                const int F = a.features[s];
                const int cin1[2] = {prev_c, 0}, src1[2] = {prev, -1};
                const int c1 = add_conv(Layer::CONV, 1, cin1, src1, F, a.kernels[s], st, in_d, dims[s]);
                const int cin2[2] = {F, 0}, src2[2] = {c1, -1};
                next_act = false;
                const int c2 = add_conv(Layer::CONV, 1, cin2, src2, F, a.kernels[s], one, dims[s], dims[s]);
                next_act = true;
                int skip = prev;
                if (strided) {
                    skip = add_aux(Layer::POOL, prev, -1, prev_c, st, in_d, dims[s]);
                    // the pooled tensor is written by the launch that forms the block output it pools (combine_pool_kernel)
                    Layer &Cb = e->layers[prev];
                    if (e->fuse_enabled && fnn_knob("FNN_NO_POOL_FUSE") == nullptr && Cb.type == Layer::COMBINE &&
                        combine_pool_ok(in_d[0], in_d[1], in_d[2], st[0], st[1], st[2])) {
                        Cb.pool_layer = skip; e->layers[skip].pool_fused = true;
                    }
                }
                if (prev_c != F) {
                    const int cinp[2] = {prev_c, 0}, srcp[2] = {skip, -1};
                    next_has_bias = false; next_act = false;
                    skip = add_conv(Layer::CONV, 1, cinp, srcp, F, one, one, dims[s], dims[s]);
                    next_has_bias = true; next_act = true;
                }
                prev = add_aux(Layer::COMBINE, c2, skip, F, one, dims[s], dims[s]);
                prev_c = F;
            }
            enc_last[s] = prev;
        }
    }
    for (int d = 0; d < a.n_stages - 1; ++d) {
        const int lvl = a.n_stages - 2 - d;           // encoder stage whose skip is consumed
        const int below = prev_c, skip = a.features[lvl];
        const int32_t *st = a.strides[lvl + 1];
        Layer T;
        T.type = Layer::TCONV; T.n_src = 1; T.has_norm = false;
        T.cin_real[0] = below; T.cin_pad[0] = pad16(below); T.src_layer[0] = prev;
        T.cout_real = skip; T.cout_pad = pad16(skip);
        for (int q = 0; q < 3; ++q) { T.k[q] = st[q]; T.s[q] = st[q]; T.in_dims[q] = dims[lvl + 1][q]; T.out_dims[q] = dims[lvl][q]; }
        const int taps = st[0] * st[1] * st[2];
        T.blob_w = blob; blob += (int64_t)below * skip * taps;
        T.blob_b = blob; blob += skip;
        T.flops = 2.0 * below * skip * (double)dims[lvl][0] * dims[lvl][1] * dims[lvl][2];
        e->layers.push_back(T);
        const int tl = (int)e->layers.size() - 1;
        if (a.n_conv_dec[d] < 1) return fail(e, FNN_E_INVALID, "n_conv_per_stage_decoder must be >= 1");
        for (int i = 0; i < a.n_conv_dec[d]; ++i) {
            if (i == 0) {
                const int cin[2] = {skip, skip}, src[2] = {tl, enc_last[lvl]};
                prev = add_conv(Layer::CONV, 2, cin, src, skip, a.kernels[lvl], one, dims[lvl], dims[lvl]);
            } else {
                const int cin[2] = {skip, 0}, src[2] = {prev, -1};
                prev = add_conv(Layer::CONV, 1, cin, src, skip, a.kernels[lvl], one, dims[lvl], dims[lvl]);
            }
        }
        prev_c = skip;
    }
    e->head_src = prev;
    e->blob_head_w = blob; blob += (int64_t)a.num_heads * a.features[0];
    e->blob_head_b = blob; blob += a.num_heads;
    e->blob_count = blob;

    // ---- producers recomputed inside their consumer's staging (conv3d_thin.hip)
    auto thin_probe = [&](const Layer &L, int fuse, const Layer *T) {
        ThinParams tp{};
        ConvParams &q = tp.c;
        q.n_src = L.n_src; q.chunks = (L.cin_pad[0] + (L.n_src > 1 ? L.cin_pad[1] : 0)) / 16;
        q.src[0].C = L.cin_pad[0]; q.src[1].C = L.n_src > 1 ? L.cin_pad[1] : 0;
        q.Di = L.in_dims[0]; q.Hi = L.in_dims[1]; q.Wi = L.in_dims[2];
        q.Do = L.out_dims[0]; q.Ho = L.out_dims[1]; q.Wo = L.out_dims[2];
        q.Cout = L.cout_pad; q.kd = L.k[0]; q.kh = L.k[1]; q.kw = L.k[2]; q.sd = L.s[0]; q.sh = L.s[1]; q.sw = L.s[2];
        tp.fuse = fuse;
        if (T) {
            tp.low.C = T->cin_pad[0];
            tp.Dl = T->in_dims[0]; tp.Hl = T->in_dims[1]; tp.Wl = T->in_dims[2];
            tp.tsd = T->s[0]; tp.tsh = T->s[1]; tp.tsw = T->s[2];
        } else { tp.tsd = tp.tsh = tp.tsw = 1; }
        return conv_thin_ok(tp);
    };
    auto row_stem_probe = [&](const Layer &L, const Layer &P) {          // conv_row_stem_kernel + stem_row_kernel (statistics)
        ThinParams tp{};
        ConvParams &q = tp.c;
        q.n_src = 1; q.chunks = L.cin_pad[0] / 16; q.src[0].C = L.cin_pad[0];
        q.Di = L.in_dims[0]; q.Hi = L.in_dims[1]; q.Wi = L.in_dims[2];
        q.Do = L.out_dims[0]; q.Ho = L.out_dims[1]; q.Wo = L.out_dims[2];
        q.Cout = L.cout_pad; q.kd = L.k[0]; q.kh = L.k[1]; q.kw = L.k[2]; q.sd = L.s[0]; q.sh = L.s[1]; q.sw = L.s[2];
        q.packing = FNN_PACK_LINEAR; q.ksteps = conv3d_ksteps(FNN_PACK_LINEAR, L.k[0] * L.k[1] * L.k[2]);
        tp.fuse = FUSE_STEM;
        StemParams sp{};
        sp.C = P.cin_real[0]; sp.kd = P.k[0]; sp.kh = P.k[1]; sp.kw = P.k[2]; sp.Cout = P.cout_pad;
        sp.PD = P.out_dims[0]; sp.PH = P.out_dims[1]; sp.PW = P.out_dims[2];
        return conv_row_ok(tp) && stem_row_ok(sp);
    };
    for (Layer &L : e->layers)
        if (L.type == Layer::STEM) {
            if (!stem_mfma_ok(L.cin_real[0], L.k[0], L.k[1], L.k[2], L.cout_pad)) return fail(e, FNN_E_UNSUPPORTED, "stem conv shape");
            L.mfma_stem = true;
        }
    if (a.spatial_dims != 2) {
        std::vector<int> consumers(e->layers.size(), 0);
        for (const Layer &L : e->layers)
            for (int i = 0; i < L.n_src; ++i) if (L.src_layer[i] >= 0) consumers[L.src_layer[i]]++;
        for (size_t li = 0; li < e->layers.size(); ++li) {
            Layer &L = e->layers[li];
            if (!e->fuse_enabled || L.type != Layer::CONV) continue;
            const int s0 = L.src_layer[0];
            if (s0 < 0) continue;
            Layer &P = e->layers[s0];
            if (e->fuse_stem != 0 && L.n_src == 1 && P.type == Layer::STEM && P.mfma_stem && P.cin_real[0] == 1 && P.cout_pad == 16 &&
                consumers[s0] == 1 && P.k[0] == L.k[0] &&
                thin_probe(L, FUSE_STEM, nullptr) && (e->fuse_stem == 1 || row_stem_probe(L, P))) {
                L.fuse = FUSE_STEM; P.virtual_out = true;
            } else if (e->fuse_tconv && L.n_src == 2 && P.type == Layer::TCONV && consumers[s0] == 1 && P.cout_pad == 16 && thin_probe(L, FUSE_TCONV, &P)) {
                L.fuse = FUSE_TCONV; P.virtual_out = true;
            }
        }
    }

    // activation layouts: a tensor of more than 16 channels that only conv / transposed-conv kernels read is stored
    // chunk-major, so that a consumer's 16-channel chunk is whole cache lines instead of 32 bytes of every record
    if (fnn_knob("FNN_NO_CHUNK_MAJOR") == nullptr) {
        const int cm_min = fnn_knob("FNN_CM_MIN") ? atoi(fnn_knob("FNN_CM_MIN")) : 16;       // A-B aid: chunk-major above this many channels
        std::vector<int> ok(e->layers.size(), 1);
        for (const Layer &L : e->layers)
            for (int i = 0; i < L.n_src; ++i)
                if (L.src_layer[i] >= 0 && L.type != Layer::CONV && L.type != Layer::TCONV && L.type != Layer::POOL &&
                    L.type != Layer::COMBINE) ok[L.src_layer[i]] = 0;
        for (size_t li = 0; li < e->layers.size(); ++li) {
            Layer &L = e->layers[li];
            L.chunk_major = ok[li] && (int)li != e->head_src &&
                            (L.type == Layer::CONV || L.type == Layer::TCONV || L.type == Layer::POOL || L.type == Layer::COMBINE) &&
                            L.cout_pad > cm_min && a.spatial_dims != 2;
        }
    }

    // device offsets
    size_t wpk = 0, fp = 0, st = 0, act = 0, ssn = 0;
    double flops = 0, bytes = 0;
    for (Layer &L : e->layers) {
        const size_t ovox = (size_t)L.out_dims[0] * L.out_dims[1] * L.out_dims[2];
        if (L.type == Layer::STEM) {
            const int T = L.k[0] * L.k[1] * L.k[2];
            L.w_off = fp; fp += (size_t)L.cin_real[0] * T * L.cout_pad;
            if (L.mfma_stem) { L.w_off2 = wpk; wpk += (size_t)(L.cout_pad / 16) * stem_mfma_ksteps(L.cin_real[0], L.k[0] * L.k[1] * L.k[2]) * 512; }
        } else if (L.type == Layer::CONV) {
            const int T = L.k[0] * L.k[1] * L.k[2];
            L.chunks = (L.cin_pad[0] + (L.n_src > 1 ? L.cin_pad[1] : 0)) / 16;
            {
                ConvParams q{};                               // the shape facts the launcher's variant choice looks at
                q.plan_N = e->max_batch; q.N = e->max_batch; q.Cout = L.cout_pad; q.chunks = L.chunks;
                q.Do = L.out_dims[0]; q.Ho = L.out_dims[1]; q.Wo = L.out_dims[2];
                q.kd = L.k[0]; q.kh = L.k[1]; q.kw = L.k[2]; q.sd = L.s[0]; q.sh = L.s[1]; q.sw = L.s[2];
                q.n_src = L.n_src; q.src[0].C = L.cin_pad[0]; q.src[1].C = L.n_src > 1 ? L.cin_pad[1] : 0;
                q.Di = L.in_dims[0]; q.Hi = L.in_dims[1]; q.Wi = L.in_dims[2];
                // e4m3 operands: the stride-1 3x3x3 layers the fp8 ZR kernel takes (the strided depth-shift kernel is fp16 only).
                // The probes see the SAME fp8 flag the launch will carry (the variant choice depends on it: the fp16-only
                // six-row tiles) - packing, tile depth and statistics rows are then the launch's; a layer the fp8 pick refuses
                // stays fp16 with whatever that pick gives it.
                q.fp8 = a.precision == FNN_PREC_F8 && !L.fuse && T == 27 && L.s[0] == 1 && L.s[1] == 1 && L.s[2] == 1 &&
                        !(L.src_layer[0] >= 0 && e->layers[L.src_layer[0]].type == Layer::GATHER);   // (the network input keeps fp16: the stem was never an fp8 layer)
                if (q.fp8 && fnn_knob("FNN_FP8_LEVELS")) {
                    // sensitivity studies (tools/fp8_sensitivity.py): e4m3 operands only at the resolution levels of the bit mask
                    // (level = how many times the patch's voxel count was divided by ~8 on the way to this layer's output)
                    const double P = (double)a.patch[0] * a.patch[1] * a.patch[2];
                    const int level = (int)std::lround(std::log2(P / (double)ovox) / 3.0);
                    q.fp8 = ((atoi(fnn_knob("FNN_FP8_LEVELS")) >> level) & 1) != 0;
                }
                L.packing = L.fuse ? FNN_PACK_LINEAR : conv3d_packing(q);
                if (q.fp8 && L.packing != FNN_PACK_ZR) { q.fp8 = 0; L.packing = conv3d_packing(q); }
                L.fp8 = q.fp8 != 0;
            }
            if (L.packing == FNN_PACK_ZP) L.chunks = conv_zp_chunks(L.cin_pad[0], L.n_src > 1 ? L.cin_pad[1] : 0);   // 32-channel chunks
            L.ksteps = conv3d_ksteps(L.packing, T);
            L.w_off = wpk; wpk += (size_t)(L.cout_pad / 16) * L.chunks * L.ksteps * 512;
        } else if (L.type == Layer::TCONV) {
            const int taps = L.s[0] * L.s[1] * L.s[2];
            L.ksteps = (L.cin_pad[0] + 31) / 32;
            L.w_off = wpk; wpk += (size_t)taps * (L.cout_pad / 16) * L.ksteps * 512;
        }
        L.bias_off = fp; fp += L.cout_pad;
        if (L.fp8) { L.oscale_off = fp; fp += L.cout_pad; }
        if (L.has_norm) { L.gamma_off = fp; fp += L.cout_pad; L.beta_off = fp; fp += L.cout_pad; }
        L.stats_slots = FNN_STAT_REPL;
        if (L.type == Layer::STEM) L.stats_slots = stem_mfma_stats_slots(L.out_dims[0], L.out_dims[1], L.out_dims[2]);
        else if (L.type == Layer::CONV && L.fuse) L.stats_slots = FNN_STAT_REPL;
        else if (L.type == Layer::CONV) {
            ConvParams q{};
            q.plan_N = e->max_batch; q.N = e->max_batch; q.Cout = L.cout_pad; q.chunks = L.chunks;
            q.Do = L.out_dims[0]; q.Ho = L.out_dims[1]; q.Wo = L.out_dims[2];
            q.kd = L.k[0]; q.kh = L.k[1]; q.kw = L.k[2]; q.sd = L.s[0]; q.sh = L.s[1]; q.sw = L.s[2];
            q.fp8 = L.fp8;
            q.n_src = L.n_src; q.src[0].C = L.cin_pad[0]; q.src[1].C = L.n_src > 1 ? L.cin_pad[1] : 0;
            q.Di = L.in_dims[0]; q.Hi = L.in_dims[1]; q.Wi = L.in_dims[2];
            if (L.packing == FNN_PACK_ZP) q.chunks = (L.cin_pad[0] + (L.n_src > 1 ? L.cin_pad[1] : 0)) / 16;
            L.stats_slots = conv3d_stats_slots(q);
        }
        L.stats_off = st; if (L.has_norm) st += (size_t)L.stats_slots * L.cout_pad * 2;
        L.ss_off = ssn; if (L.has_norm) ssn += L.cout_pad;
        L.out_off = act; act += ovox * L.cout_pad;
        flops += L.flops;
        bytes += 2.0 * ovox * L.cout_real * 2.0;            // written once + read once, fp16
    }
    e->hblocks = (a.num_heads + 1 + 15) / 16;               // + the weight-sum channel (zero weights, bias 1: its "logit" is 1)
    e->head_ksteps = (pad16(a.features[0]) + 31) / 32;
    e->head_w_off = wpk; wpk += (size_t)e->hblocks * e->head_ksteps * 512;
    e->head_bias_off = fp; fp += (size_t)e->hblocks * 16;
    e->n_gpass = (a.num_heads + 62) / 63;
    if (e->n_gpass > 1) {
        e->gpass_w_off = wpk; wpk += (size_t)e->n_gpass * 4 * 512;
        e->gpass_bias_off = fp; fp += (size_t)e->n_gpass * 64;
    }
    const double P = (double)a.patch[0] * a.patch[1] * a.patch[2];
    e->head_flops = 2.0 * a.num_heads * a.features[0] * P;
    e->patch_flops = flops + e->head_flops;
    // skips are read twice (next encoder stage and decoder); network input read once
    for (int s = 0; s < a.n_stages - 1; ++s) {
        const Layer &L = e->layers[enc_last[s]];
        bytes += (double)L.out_dims[0] * L.out_dims[1] * L.out_dims[2] * L.cout_real * 2.0;
    }
    e->patch_act_bytes = bytes;
    e->wpk_halves = wpk; e->fparam_floats = fp; e->stats_doubles = st; e->act_halves = act; e->ss_count = ssn;
    return 0;
}

// ---------------------------------------------------------------------------
// weight packing (host)
// ---------------------------------------------------------------------------
inline uint16_t f2h_bits(float f) { f16 h = (f16)f; uint16_t b; memcpy(&b, &h, 2); return b; }

void pack_conv(const Layer &L, const float *W, uint16_t *dst) {
    const int T = L.k[0] * L.k[1] * L.k[2];
    const int cin_tot = L.cin_real[0] + (L.n_src > 1 ? L.cin_real[1] : 0);
    const int nblk = L.cout_pad / 16;
    for (int cb = 0; cb < nblk; ++cb)
        for (int ch = 0; ch < L.chunks; ++ch)
            for (int ks = 0; ks < L.ksteps; ++ks)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int k = 8 * (lane >> 4) + j;
                        const int tap = conv3d_kstep_tap(L.packing, ks, k >> 4, T), c = ch * 16 + (k & 15);
                        const int co = conv3d_pack_cout(L.packing, nblk, cb, lane & 15);
                        int src = 0, cl = c;
                        if (c >= L.cin_pad[0]) { src = 1; cl = c - L.cin_pad[0]; }
                        float v = 0.f;
                        if (tap >= 0 && co < L.cout_real && cl < L.cin_real[src]) {
                            const int ci = (src ? L.cin_real[0] : 0) + cl;
                            v = W[((size_t)co * cin_tot + ci) * T + tap];
                        }
                        dst[((((size_t)cb * L.chunks + ch) * L.ksteps + ks) * 64 + lane) * 8 + j] = f2h_bits(v);
                    }
}

// OCP e4m3 ("fn": no infinities, 0x7f = NaN), round to nearest even, saturating at +-448
uint8_t f2e4m3(float f) {
    const uint8_t sign = std::signbit(f) ? 0x80 : 0;
    float a = std::fabs(f);
    if (!(a == a)) return sign | 0x7f;
    if (a >= 448.f) return sign | 0x7e;
    if (a < 0x1p-6f) {                                          // subnormal: multiples of 2^-9
        const int q = (int)std::nearbyint(a * 512.f);           // 0 .. 8 (8 = the smallest normal)
        return sign | (uint8_t)q;                               // q = 8 -> exponent field 1, mantissa 0 = 0x08
    }
    int e;
    const float m = std::frexp(a, &e);                          // a = m * 2^e, m in [0.5, 1)
    int q = (int)std::nearbyint(m * 16.f);                      // 8 .. 16
    int E = e - 1;                                              // a = (q / 8) * 2^E
    if (q == 16) { q = 8; ++E; }
    if (E > 8 || (E == 8 && q > 14)) return sign | 0x7e;
    return sign | (uint8_t)(((E + 7) << 3) | (q - 8));
}

// fp8 weights of a ZR layer: same fragment order as pack_conv at one byte per element, one scale per output channel
// (max |w| of the channel -> 448); scales[co] = w_scale / FNN_FP8_ACT_MULT is what the kernel's epilogue multiplies by.
#define FNN_FP8_ACT_MULT 8.0f
void pack_conv_fp8(const Layer &L, const float *W, uint8_t *dst, float *scales) {
    const int T = L.k[0] * L.k[1] * L.k[2];
    const int cin_tot = L.cin_real[0] + (L.n_src > 1 ? L.cin_real[1] : 0);
    const int nblk = L.cout_pad / 16;
    std::vector<float> inv(L.cout_pad, 0.f);
    for (int co = 0; co < L.cout_pad; ++co) {
        float mx = 0.f;
        if (co < L.cout_real)
            for (size_t i = 0; i < (size_t)cin_tot * T; ++i) mx = std::max(mx, std::fabs(W[(size_t)co * cin_tot * T + i]));
        const float ws = mx > 0.f ? mx / 448.f : 1.f;
        inv[co] = 1.f / ws;
        scales[co] = ws / FNN_FP8_ACT_MULT;
    }
    for (int cb = 0; cb < nblk; ++cb)
        for (int ch = 0; ch < L.chunks; ++ch)
            for (int ks = 0; ks < L.ksteps; ++ks)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int k = 8 * (lane >> 4) + j;
                        const int tap = conv3d_kstep_tap(L.packing, ks, k >> 4, T), c = ch * 16 + (k & 15);
                        const int co = conv3d_pack_cout(L.packing, nblk, cb, lane & 15);
                        int src = 0, cl = c;
                        if (c >= L.cin_pad[0]) { src = 1; cl = c - L.cin_pad[0]; }
                        float v = 0.f;
                        if (tap >= 0 && co < L.cout_real && cl < L.cin_real[src]) {
                            const int ci = (src ? L.cin_real[0] : 0) + cl;
                            v = W[((size_t)co * cin_tot + ci) * T + tap] * inv[co];
                        }
                        dst[((((size_t)cb * L.chunks + ch) * L.ksteps + ks) * 64 + lane) * 8 + j] = f2e4m3(v);
                    }
}

void pack_tconv(const Layer &L, const float *W, uint16_t *dst) {
    const int taps = L.s[0] * L.s[1] * L.s[2];
    const int nblk = L.cout_pad / 16;
    for (int tap = 0; tap < taps; ++tap)
        for (int cb = 0; cb < nblk; ++cb)
            for (int ks = 0; ks < L.ksteps; ++ks)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int ci = ks * 32 + 8 * (lane >> 4) + j, co = cb * 16 + (lane & 15);
                        float v = 0.f;
                        if (ci < L.cin_real[0] && co < L.cout_real) v = W[((size_t)ci * L.cout_real + co) * taps + tap];
                        dst[((((size_t)tap * nblk + cb) * L.ksteps + ks) * 64 + lane) * 8 + j] = f2h_bits(v);
                    }
}

void pack_head(int heads, int cin, int hblocks, int ksteps, const float *W, uint16_t *dst) {
    for (int hb = 0; hb < hblocks; ++hb)
        for (int ks = 0; ks < ksteps; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int ci = ks * 32 + 8 * (lane >> 4) + j, h = hb * 16 + (lane & 15);
                    float v = 0.f;
                    if (ci < cin && h < heads) v = W[(size_t)h * cin + ci];
                    dst[(((size_t)hb * ksteps + ks) * 64 + lane) * 8 + j] = f2h_bits(v);
                }
}

// ---------------------------------------------------------------------------
// profiling helpers
// ---------------------------------------------------------------------------
enum { FAM_CONV = 0, FAM_STEM, FAM_TCONV, FAM_HEAD, FAM_FINAL };

struct Scope {
    fnn_engine *e; hipStream_t st; int idx = -1;
    Scope(fnn_engine *e_, hipStream_t st_, int family, double flops, double bytes = 0) : e(e_), st(st_) {
        fnn_klog_target(e->profiling ? &e->klog : nullptr);  // the launchers inside this scope note their kernel variant
        if (!e->profiling) return;
        if (e->ev_used == e->evs.size()) {
            fnn_engine::Ev ev{};
            if (hipEventCreate(&ev.a) != hipSuccess || hipEventCreate(&ev.b) != hipSuccess) return;
            e->evs.push_back(ev);
        }
        idx = (int)e->ev_used++;
        e->evs[idx].family = family; e->evs[idx].flops = flops; e->evs[idx].bytes = bytes;
        e->evs[idx].layer = e->cur_layer; e->evs[idx].klog_lo = e->evs[idx].klog_hi = e->klog.size();
        (void)hipEventRecord(e->evs[idx].a, st);
    }
    ~Scope() { if (idx >= 0) { (void)hipEventRecord(e->evs[idx].b, st); e->evs[idx].klog_hi = e->klog.size(); } }
};

void collect_profile(fnn_engine *e, int64_t n_patches) {
    fnn_profile &p = e->prof;
    p = fnn_profile{};
    p.n_patches = n_patches;
    e->launch_rows.clear();
    for (size_t i = 0; i < e->ev_used; ++i) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, e->evs[i].a, e->evs[i].b) != hipSuccess) continue;
        p.total_ms += ms;
        {
            static const char *const fam[] = {"conv", "stem", "tconv", "head", "finalize"};
            const fnn_engine::Ev &v = e->evs[i];
            char row[320];
            std::string names;
            for (size_t k = v.klog_lo; k < v.klog_hi && k < e->klog.size(); ++k) { if (!names.empty()) names += " + "; names += e->klog[k]; }
            snprintf(row, sizeof row, "%d\t%s\t%.6f\t%.6g\t%.6g\t%s", v.layer, fam[v.family >= 0 && v.family <= 4 ? v.family : 4],
                     (double)ms, v.flops, v.bytes, names.c_str());
            e->launch_rows.push_back(row);
        }
        switch (e->evs[i].family) {
            case FAM_CONV: p.conv_ms += ms; p.conv_launches++; p.conv_flops += e->evs[i].flops; p.conv_bytes += e->evs[i].bytes; break;
            case FAM_STEM: p.stem_ms += ms; break;
            case FAM_TCONV: p.tconv_ms += ms; break;
            case FAM_HEAD: p.head_ms += ms; break;
            default: p.finalize_ms += ms; break;
        }
    }
    e->ev_used = 0;
}

// ---------------------------------------------------------------------------
// one network forward for `nb` patches (activations only; the head is separate)
// ---------------------------------------------------------------------------
// the fp16 scale / shift rows (SrcDesc::ssh) live behind the fp32 rows of the same arena: [ss_count * max_batch * 2 + 4] floats,
// then ss_count * max_batch * 2 halves
static unsigned short *ssh_rows(fnn_engine *e) { return (unsigned short *)(e->ss + e->ss_count * e->max_batch * 2 + 4); }

SrcDesc make_src(fnn_engine *e, const FoldWeights &fw, int layer, int nb) {
    const Layer &L = e->layers[layer];
    SrcDesc s{};
    s.ptr = e->act + L.out_off * e->max_batch;
    s.C = L.cout_pad;
    if (L.chunk_major) { s.vs = 16; s.cs = 16LL * L.out_dims[0] * L.out_dims[1] * L.out_dims[2]; }
    else { s.vs = L.cout_pad; s.cs = 16; }
    if (L.has_norm) {
        s.ss = e->ss + L.ss_off * e->max_batch * 2;
        s.ssh = ssh_rows(e) + L.ss_off * e->max_batch * 2;           // halves: [N][C / 8][16] = 2 C per item
        s.slope = L.act ? e->arch.slope : 1.f;
    } else {
        s.ss = nullptr; s.ssh = nullptr; s.slope = 1.f;
    }
    (void)fw;
    (void)nb;
    return s;
}

// head_out / head_ss: where the network's last layer writes its raw output and its (scale, shift) rows instead of the
// arena (the gather path keeps them per patch)
int forward_batch(fnn_engine *e, int fold, const float *vol, long long vol_batch_stride, const long long vdim[3],
                  const int *origins_dev, int nb, const int flip[3], hipStream_t st, f16 *head_out = nullptr,
                  float *head_ss = nullptr, unsigned short *head_ssh = nullptr) {
    const FoldWeights &fw = e->folds[fold];
    HIPCHK(e, hipMemsetAsync(e->stats, 0, e->stats_doubles * e->max_batch * sizeof(double), st));
    struct LayerMark { fnn_engine *e; ~LayerMark() { e->cur_layer = -1; } } mark{e};
    for (size_t li = 0; li < e->layers.size(); ++li) {
        const Layer &L = e->layers[li];
        e->cur_layer = (int)li;
        f16 *out = e->act + L.out_off * e->max_batch;
        if (head_out && (int)li == e->head_src) out = head_out;
        double *stats_out = L.has_norm ? e->stats + L.stats_off * e->max_batch : nullptr;
        int rc = 0;
        if (L.type == Layer::STEM) {
            StemParams p{};
            p.vol = vol; p.vol_batch_stride = vol_batch_stride; p.C = L.cin_real[0];
            p.X = vdim[0]; p.Y = vdim[1]; p.Z = vdim[2];
            p.origins = origins_dev;
            p.flip_d = flip[0]; p.flip_h = flip[1]; p.flip_w = flip[2];
            p.PD = L.out_dims[0]; p.PH = L.out_dims[1]; p.PW = L.out_dims[2];
            p.kd = L.k[0]; p.kh = L.k[1]; p.kw = L.k[2];
            p.Cout = L.cout_pad;
            p.w = fw.fparam + L.w_off; p.bias = fw.fparam + L.bias_off;
            p.out = out; p.stats_out = stats_out;
            p.tiles_d = (p.PD + FNN_TILE_D - 1) / FNN_TILE_D;
            p.tiles_h = (p.PH + FNN_TILE_H - 1) / FNN_TILE_H;
            p.tiles_w = (p.PW + FNN_TILE_W - 1) / FNN_TILE_W;
            Scope sc(e, st, FAM_STEM, L.flops * nb, L.bytes * nb);
            if (L.virtual_out) p.out = nullptr;                       // statistics only: the consumer recomputes the values
            rc = launch_stem_mfma(p, fw.wpk + L.w_off2, nb, st);
        } else if (L.type == Layer::CONV) {
            ConvParams p{};
            p.n_src = L.n_src;
            for (int i = 0; i < L.n_src; ++i) p.src[i] = make_src(e, fw, L.src_layer[i], nb);
            if (L.n_src == 1) { p.src[1] = p.src[0]; p.src[1].C = 0; }
            p.N = nb; p.plan_N = e->max_batch; p.Di = L.in_dims[0]; p.Hi = L.in_dims[1]; p.Wi = L.in_dims[2];
            p.Do = L.out_dims[0]; p.Ho = L.out_dims[1]; p.Wo = L.out_dims[2];
            p.Cout = L.cout_pad;
            p.kd = L.k[0]; p.kh = L.k[1]; p.kw = L.k[2];
            p.sd = L.s[0]; p.sh = L.s[1]; p.sw = L.s[2];
            p.pd = (L.k[0] - 1) / 2; p.ph = (L.k[1] - 1) / 2; p.pw = (L.k[2] - 1) / 2;
            p.wpk = fw.wpk + L.w_off; p.bias = fw.fparam + L.bias_off;
            p.out = out; p.stats_out = stats_out; p.stats_slots = L.stats_slots;
            if (L.chunk_major) { p.out_vs = 16; p.out_cs = 16LL * p.Do * p.Ho * p.Wo; }
            p.tiles_d = (p.Do + FNN_TILE_D - 1) / FNN_TILE_D;
            p.tiles_h = (p.Ho + FNN_TILE_H - 1) / FNN_TILE_H;
            p.tiles_w = (p.Wo + FNN_TILE_W - 1) / FNN_TILE_W;
            p.chunks = L.chunks; p.ksteps = L.ksteps; p.packing = L.packing;
            p.fp8 = L.fp8; p.oscale = L.fp8 ? fw.fparam + L.oscale_off : nullptr; p.act_mult = FNN_FP8_ACT_MULT;
            p.tile_d = FNN_TILE_D;
            Scope sc(e, st, FAM_CONV, L.flops * nb, L.bytes * nb);
            if (L.fuse) {
                ThinParams tp{};
                tp.c = p; tp.fuse = L.fuse;
                const Layer &P = e->layers[L.src_layer[0]];
                tp.fbias = fw.fparam + P.bias_off;
                if (L.fuse == FUSE_STEM) {
                    tp.fw = fw.wpk + P.w_off2;
                    tp.vol = vol; tp.vol_batch_stride = vol_batch_stride; tp.Y = vdim[1]; tp.Z = vdim[2];
                    tp.origins = origins_dev; tp.flip_d = flip[0]; tp.flip_h = flip[1]; tp.flip_w = flip[2];
                    tp.fss = e->ss + P.ss_off * e->max_batch * 2; tp.fslope = e->arch.slope;
                } else {
                    tp.fw = fw.wpk + P.w_off;
                    tp.low = make_src(e, fw, P.src_layer[0], nb);
                    tp.Dl = P.in_dims[0]; tp.Hl = P.in_dims[1]; tp.Wl = P.in_dims[2];
                    tp.tsd = P.s[0]; tp.tsh = P.s[1]; tp.tsw = P.s[2];
                }
                rc = launch_conv_thin(tp, st);
            } else
            rc = launch_conv3d(p, st);
        } else if (L.type == Layer::GATHER) {
            PatchInputParams p{};
            p.vol = vol; p.vol_batch_stride = vol_batch_stride; p.C = L.cin_real[0]; p.Cpad = L.cout_pad;
            p.X = vdim[0]; p.Y = vdim[1]; p.Z = vdim[2];
            p.origins = origins_dev;
            p.flip_d = flip[0]; p.flip_h = flip[1]; p.flip_w = flip[2];
            p.PD = L.out_dims[0]; p.PH = L.out_dims[1]; p.PW = L.out_dims[2]; p.N = nb;
            p.out = out;
            if (L.chunk_major) { p.out_vs = 16; p.out_cs = 16LL * L.out_dims[0] * L.out_dims[1] * L.out_dims[2]; }
            Scope sc(e, st, FAM_STEM, 0, 2.0 * nb * (2.0 * L.cin_real[0] + L.cout_pad) * L.out_dims[0] * L.out_dims[1] * L.out_dims[2]);
            rc = launch_patch_input(p, st);
        } else if (L.type == Layer::POOL) {
            if (L.pool_fused) continue;                               // written by the COMBINE launch of its source
            PoolParams p{};
            p.src = make_src(e, fw, L.src_layer[0], nb);
            p.N = nb; p.Di = L.in_dims[0]; p.Hi = L.in_dims[1]; p.Wi = L.in_dims[2];
            p.sd = L.s[0]; p.sh = L.s[1]; p.sw = L.s[2];
            p.out = out;
            if (L.chunk_major) { p.out_vs = 16; p.out_cs = 16LL * L.out_dims[0] * L.out_dims[1] * L.out_dims[2]; }
            Scope sc(e, st, FAM_TCONV, 0);
            rc = launch_avgpool(p, st);
        } else if (L.type == Layer::COMBINE) {
            CombineParams p{};
            p.a = make_src(e, fw, L.src_layer[0], nb);
            p.b = make_src(e, fw, L.src_layer[1], nb);
            p.vox = (long long)L.out_dims[0] * L.out_dims[1] * L.out_dims[2];
            p.N = nb; p.slope = e->arch.slope; p.out = out;
            if (L.chunk_major) { p.out_vs = 16; p.out_cs = 16LL * p.vox; }
            if (L.pool_layer >= 0) {
                const Layer &P = e->layers[L.pool_layer];
                p.pool_out = e->act + P.out_off * e->max_batch;
                p.D = L.out_dims[0]; p.H = L.out_dims[1]; p.W = L.out_dims[2];
                p.psd = P.s[0]; p.psh = P.s[1]; p.psw = P.s[2];
                if (P.chunk_major) { p.pool_vs = 16; p.pool_cs = 16LL * P.out_dims[0] * P.out_dims[1] * P.out_dims[2]; }
            }
            Scope sc(e, st, FAM_TCONV, 0);
            rc = launch_combine(p, st);
        } else {
            TconvParams p{};
            p.src = make_src(e, fw, L.src_layer[0], nb);
            p.N = nb; p.Di = L.in_dims[0]; p.Hi = L.in_dims[1]; p.Wi = L.in_dims[2];
            p.sd = L.s[0]; p.sh = L.s[1]; p.sw = L.s[2];
            p.Cout = L.cout_pad; p.wpk = fw.wpk + L.w_off; p.bias = fw.fparam + L.bias_off;
            p.out = out; p.ksteps = L.ksteps; p.nblk = L.cout_pad / 16;
            if (L.chunk_major) { p.out_vs = 16; p.out_cs = 16LL * L.out_dims[0] * L.out_dims[1] * L.out_dims[2]; }
            if (L.virtual_out) continue;                              // computed inside its consumer (conv3d_thin.hip)
            Scope sc(e, st, FAM_TCONV, L.flops * nb);
            rc = launch_tconv(p, st);
        }
        if (rc != 0) return fail(e, rc == -1 ? FNN_E_UNSUPPORTED : FNN_E_HIP, "kernel launch failed at layer %zu (rc=%d)", li, rc);
        if (L.has_norm) {
            StatsFinalizeParams q{};
            q.stats = stats_out; q.gamma = fw.fparam + L.gamma_off; q.beta = fw.fparam + L.beta_off;
            q.ss = e->ss + L.ss_off * e->max_batch * 2; q.C = L.cout_pad; q.nrep = L.stats_slots;
            q.ssh = ssh_rows(e) + L.ss_off * e->max_batch * 2;
            if (head_ss && (int)li == e->head_src) q.ss = head_ss;
            if (head_ssh && (int)li == e->head_src) q.ssh = head_ssh;
            q.inv_count = 1.f / ((float)L.out_dims[0] * L.out_dims[1] * L.out_dims[2]); q.eps = e->arch.eps;
            if (launch_stats_finalize(q, nb, st) != 0) return fail(e, FNN_E_HIP, "stats finalize launch failed");
        }
    }
    return 0;
}

HeadParams make_head(fnn_engine *e, int fold, int b) {
    const fnn_arch_desc &a = e->arch;
    const FoldWeights &fw = e->folds[fold];
    HeadParams h{};
    h.src = make_src(e, fw, e->head_src, 0);
    h.b = b; h.PD = a.patch[0]; h.PH = a.patch[1]; h.PW = a.patch[2];
    h.heads = a.num_heads; h.hblocks = e->hblocks; h.ksteps = e->head_ksteps;
    h.wpk = fw.wpk + e->head_w_off; h.bias = fw.fparam + e->head_bias_off;
    h.fx = h.fy = h.fz = INT_MAX;                           // every voxel is read unless run_patches knows better
    return h;
}

int steps_1d(int64_t image, int64_t patch, double step, std::vector<int64_t> &out) {
    // compute_steps_for_sliding_window (sliding_window_prediction.py:30-54), one axis
    if (!(step > 0 && step <= 1) || patch <= 0 || image < patch) return -1;
    const double target = (double)patch * step;
    const int64_t n = (int64_t)std::ceil((double)(image - patch) / target) + 1;
    out.clear();
    if (n > 1) {
        const double actual = (double)(image - patch) / (double)(n - 1);
        for (int64_t i = 0; i < n; ++i) out.push_back((int64_t)std::nearbyint(actual * (double)i));   // half-to-even
    } else {
        out.push_back(0);
    }
    return 0;
}

struct VolPlan {
    int64_t padded[3], lo[3];
    std::vector<int64_t> steps[3];
    std::vector<int> origins;          // [n][3], x-major (predict_from_raw_data.py:532-537)
    int64_t n_patches = 0;
};

int plan_volume_p(const int32_t patch[3], const int64_t sp[3], double step, bool two_d, VolPlan &vp);
int plan_volume(const fnn_arch_desc &a, const int64_t sp[3], double step, VolPlan &vp) {
    return plan_volume_p(a.patch, sp, step, a.spatial_dims == 2, vp);
}
// _internal_get_sliding_window_slicers (:506-538): padding to >= patch, tile starts per axis, x-major order.
// two_d: the `2d` branch (:508-524) - the first axis is not tiled, every slice is visited once.
int plan_volume_p(const int32_t patch[3], const int64_t sp[3], double step, bool two_d, VolPlan &vp) {
    for (int d = 0; d < 3; ++d) {
        if (two_d && d == 0) {
            if (sp[0] < 1) return -1;
            vp.padded[0] = sp[0]; vp.lo[0] = 0;
            vp.steps[0].clear();
            for (int64_t i = 0; i < sp[0]; ++i) vp.steps[0].push_back(i);
            continue;
        }
        if (sp[d] < 1 || patch[d] < 1) return -1;
        const int64_t target = sp[d] > patch[d] ? sp[d] : patch[d];
        const int64_t diff = target - sp[d];
        vp.padded[d] = target; vp.lo[d] = diff / 2;        // the odd voxel goes to the high side
        if (steps_1d(target, patch[d], step, vp.steps[d]) != 0) return -1;
    }
    vp.origins.clear();
    for (int64_t x : vp.steps[0])
        for (int64_t y : vp.steps[1])
            for (int64_t z : vp.steps[2]) { vp.origins.push_back((int)x); vp.origins.push_back((int)y); vp.origins.push_back((int)z); }
    vp.n_patches = (int64_t)vp.origins.size() / 3;
    return 0;
}

// mirror-axis subsets in the reference's order: by size, then lexicographic (:551-553)
std::vector<std::vector<int>> mirror_combos(const fnn_opts &o) {
    std::vector<std::vector<int>> out;
    const int n = o.n_mirror_axes;
    // enumerate combinations of positions in lexicographic order
    for (int size = 1; size <= n; ++size) {
        std::vector<int> idx(size);
        for (int i = 0; i < size; ++i) idx[i] = i;
        while (true) {
            std::vector<int> c;
            for (int i : idx) c.push_back(o.mirror_axes[i]);
            out.push_back(c);
            int i = size - 1;
            while (i >= 0 && idx[i] == n - size + i) --i;
            if (i < 0) break;
            ++idx[i];
            for (int j = i + 1; j < size; ++j) idx[j] = idx[j - 1] + 1;
        }
    }
    return out;
}

int check_ready(fnn_engine *e, int fold, const fnn_opts *o) {
    if (!e) return FNN_E_INVALID;
    if (!o) return fail(e, FNN_E_INVALID, "opts is NULL");
    if (fold < 0 || fold >= (int)e->folds.size() || !e->folds[fold].loaded) return fail(e, FNN_E_STATE, "weights of fold %d are not loaded", fold);
    if (!(o->tile_step_size > 0 && o->tile_step_size <= 1)) return fail(e, FNN_E_INVALID, "step_size must be larger than 0 and smaller or equal to 1");
    if (o->use_gaussian && !e->gauss) return fail(e, FNN_E_STATE, "use_gaussian is set but fnn_set_gaussian was not called");
    if (o->n_mirror_axes < 0 || o->n_mirror_axes > 3) return fail(e, FNN_E_INVALID, "n_mirror_axes out of range");
    for (int i = 0; i < o->n_mirror_axes; ++i)
        if (o->mirror_axes[i] < 0 || o->mirror_axes[i] > 2) return fail(e, FNN_E_INVALID, "mirror_axes does not match the dimension of the input!");
    if (o->accum < FNN_ACC_FP16_REFERENCE || o->accum > FNN_ACC_FP16_AUTOCAST) return fail(e, FNN_E_INVALID, "accum is not a FNN_ACC_* value");
    return 0;
}

// the entry points on accumulator buffers (multi-GPU accumulate mode, > 63 classes) only know the two buffer arithmetics
inline int no_autocast(fnn_engine *e, const fnn_opts *o, const char *who) {
    return o->accum == FNN_ACC_FP16_AUTOCAST ? fail(e, FNN_E_UNSUPPORTED, "%s: FNN_ACC_FP16_AUTOCAST is served by the gather path only", who) : 0;
}

struct Box { int64_t lo[3], hi[3]; };

int upload_box(fnn_engine *e, int64_t x_lo, int64_t x_hi, int64_t y_lo, int64_t y_hi, hipStream_t st);     // (below, next to stage_volume)

inline int acc_hp(const fnn_arch_desc &a) { return (a.num_heads + 1 + 7) / 8 * 8; }

// Runs the listed patches and accumulates into `acc`, which covers `box` of the padded volume
// ([bx][by][bz][HP], channels-last, channel num_heads = weight sum).
// `fresh`: `ids` is the volume's whole patch list in visiting order and `acc` holds nothing yet (it need not even be
// zeroed): voxels no earlier patch has touched are then written without being read (HeadParams::fx).
// keep_features: no head, no accumulation - patch ids[i] leaves its last activation in e->feat[i] (gather path).
int run_patches(fnn_engine *e, int fold, const float *vol_dev, const VolPlan &vp, const fnn_opts &o,
                const std::vector<int64_t> &ids, const int *ids_origins_dev, const Box &box, void *acc, int acc_fp32,
                hipStream_t st, bool fresh = false, bool keep_features = false, int64_t slot0 = 0, int64_t n_slots = 0,
                void *feat_ext = nullptr, float *fss_ext = nullptr) {
    const fnn_arch_desc &a = e->arch;
    const long long vdim[3] = {(long long)vp.padded[0], (long long)vp.padded[1], (long long)vp.padded[2]};
    int B = o.batch > 0 ? o.batch : e->max_batch;
    if (B > e->max_batch) B = e->max_batch;
    const auto combos = mirror_combos(o);
    const bool tta = !combos.empty();
    const size_t P = (size_t)a.patch[0] * a.patch[1] * a.patch[2];
    if (tta && !keep_features) {
        void *pbuf = e->patch_buf;
        if (int rc = ensure(e, &pbuf, &e->patch_buf_bytes, (size_t)B * a.num_heads * P * sizeof(float))) return rc;
        e->patch_buf = (float *)pbuf;
    }
    const int64_t np = (int64_t)ids.size();
    if (np > B) {                                           // balanced batches: 75 patches run as 19+19+19+18, not 24+24+24+3
        const int64_t nbat = (np + B - 1) / B;
        B = (int)((np + nbat - 1) / nbat);
    }
    for (int64_t i = 0; i < np && !keep_features; ++i) {
        const int *oo = &vp.origins[ids[i] * 3];
        for (int d = 0; d < 3; ++d)
            if (oo[d] < box.lo[d] || oo[d] + a.patch[d] > box.hi[d])
                return fail(e, FNN_E_INVALID, "patch %lld lies outside the accumulator box", (long long)ids[i]);
    }
    const size_t featC = keep_features ? (size_t)e->layers[e->head_src].cout_pad : 0;
    // ---- several batches in flight (see fnn_engine::pipe)
    static const bool no_pipe = fnn_knob("FNN_NO_PIPELINE") != nullptr;                // A-B aid
    const bool pipelined = !no_pipe && (!tta || keep_features) && !e->profiling && np > B;
    f16 *const act0 = e->act; double *const stats0 = e->stats; float *const ss0 = e->ss;
    static const int want_pipes = fnn_knob("FNN_PIPES") ? atoi(fnn_knob("FNN_PIPES")) : 4;   // round 6: four beat three by 0.3-0.9 % on C1 / C2 / C4 / C5 (profiles/r06_pipes_sweep.txt); beyond four: nothing
    int NP = want_pipes < 2 ? 2 : (want_pipes > fnn_engine::MAXP ? fnn_engine::MAXP : want_pipes);
    if (pipelined) {
        if (e->n_pipe < NP) {
            // the arenas of the batches in flight take at most half of the device's memory (a full-width teacher's is 29 GB at
            // batch 32, the 160^3 ResEnc student's 45 GB) and never what is not free: fewer batches in flight then, not a failed call
            const size_t arena = e->act_halves * e->max_batch * sizeof(f16) + e->stats_doubles * e->max_batch * sizeof(double) +
                                 (e->ss_count * e->max_batch * 3 + 8) * sizeof(float);
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && arena > 0) {
                const long long by_total = (long long)(total_b / 2 / arena), by_free = (long long)(free_b * 8 / 10 / arena) + e->n_pipe;
                const long long cap = std::max<long long>(2, std::min(by_total, by_free));
                if (NP > cap) NP = (int)std::max<long long>(cap, e->n_pipe);
            }
        }
        if (e->n_pipe < NP) {
            if (!e->ev_start) HIPCHK(e, hipEventCreateWithFlags(&e->ev_start, hipEventDisableTiming));
            for (int k = e->n_pipe; k < NP; ++k) {
                if (k > 0) {
                    HIPCHK(e, hipMalloc((void **)&e->actp[k], e->act_halves * e->max_batch * sizeof(f16)));
                    HIPCHK(e, hipMalloc((void **)&e->statsp[k], e->stats_doubles * e->max_batch * sizeof(double)));
                    HIPCHK(e, hipMalloc((void **)&e->ssp[k], (e->ss_count * e->max_batch * 3 + 8) * sizeof(float)));   // fp32 rows + fp16 rows (ssh_rows)
                }
                HIPCHK(e, hipStreamCreateWithFlags(&e->pipe[k], hipStreamNonBlocking));
                HIPCHK(e, hipEventCreateWithFlags(&e->ev_head[k], hipEventDisableTiming));
                HIPCHK(e, hipEventCreateWithFlags(&e->ev_done[k], hipEventDisableTiming));
                e->n_pipe = k + 1;
            }
        }
        e->actp[0] = act0; e->statsp[0] = stats0; e->ssp[0] = ss0;
        HIPCHK(e, hipEventRecord(e->ev_start, st));
        for (int k = 0; k < NP; ++k) HIPCHK(e, hipStreamWaitEvent(e->pipe[k], e->ev_start, 0));
    }
    std::vector<int> uniq[3];                             // distinct patch positions per axis (fresh: overlap with the previous one)
    if (fresh) {
        for (int d = 0; d < 3; ++d) {
            for (int64_t i = 0; i < vp.n_patches; ++i) uniq[d].push_back(vp.origins[i * 3 + d]);
            std::sort(uniq[d].begin(), uniq[d].end());
            uniq[d].erase(std::unique(uniq[d].begin(), uniq[d].end()), uniq[d].end());
        }
    }
    auto first_visit = [&](const int *oo, int d) -> int {
        if (!fresh) return INT_MAX;
        const auto it = std::lower_bound(uniq[d].begin(), uniq[d].end(), oo[d]);
        if (it == uniq[d].begin()) return 0;
        const int ov = *(it - 1) + a.patch[d] - oo[d];
        return ov > 0 ? ov : 0;
    };
    struct Restore {                                      // whatever happens, the engine ends on its first arena
        fnn_engine *e; f16 *a; double *s; float *ss;
        ~Restore() { e->act = a; e->stats = s; e->ss = ss; }
    } restore{e, act0, stats0, ss0};
    hipStream_t user_st = st;
    int64_t bi = 0;
    for (int64_t p0 = 0; p0 < np; p0 += B, ++bi) {
        const int nb = (int)((np - p0 < B) ? np - p0 : B);
        const int *org = ids_origins_dev + p0 * 3;
        const int k = (int)(bi % NP);
        if (pipelined) {
            st = e->pipe[k];
            e->act = e->actp[k]; e->stats = e->statsp[k]; e->ss = e->ssp[k];
        }
        if (e->up.active) {                                   // a volume still arriving from the host: the box this batch's patches read
            int64_t lo[2] = {INT64_MAX, INT64_MAX}, hi[2] = {0, 0};   // (un-padded volume: padded == shape along every axis here)
            for (int b = 0; b < nb; ++b)
                for (int d = 0; d < 2; ++d) {
                    const int64_t o0 = vp.origins[ids[p0 + b] * 3 + d];
                    lo[d] = std::min(lo[d], o0); hi[d] = std::max(hi[d], o0 + a.patch[d]);
                }
            if (int rc = upload_box(e, lo[0], hi[0], lo[1], hi[1], st)) return rc;
        }
        for (size_t ci = 0; ci <= (tta ? combos.size() : 0); ++ci) {
            int flip[3] = {0, 0, 0};
            if (ci > 0) for (int ax : combos[ci - 1]) flip[ax] = 1;
            if (keep_features) {                              // [evaluation][slot]: the batch's items stay contiguous
                const size_t item = (size_t)ci * n_slots + slot0 + p0;
                if (int rc = forward_batch(e, fold, vol_dev, 0, vdim, org, nb, flip, st,
                                           (f16 *)(feat_ext ? feat_ext : e->feat) + item * P * featC,
                                           (fss_ext ? fss_ext : (float *)e->featss) + item * 2 * featC,
                                           feat_ext ? nullptr : (unsigned short *)e->featssh + item * 2 * featC)) return rc;
                continue;
            }
            if (int rc = forward_batch(e, fold, vol_dev, 0, vdim, org, nb, flip, st)) return rc;
            if (pipelined && bi > 0) HIPCHK(e, hipStreamWaitEvent(st, e->ev_head[(bi - 1) % NP], 0));   // heads in patch order
            for (int b = 0; b < nb; ++b) {
                HeadParams h = make_head(e, fold, b);
                const int *oo = &vp.origins[ids[p0 + b] * 3];
                h.gauss = o.use_gaussian ? e->gauss : e->ones;
                h.acc = acc; h.AX = box.hi[0] - box.lo[0]; h.Y = box.hi[1] - box.lo[1]; h.Z = box.hi[2] - box.lo[2];
                h.HP = acc_hp(a);
                h.ox = oo[0] - (int)box.lo[0]; h.oy = oo[1] - (int)box.lo[1]; h.oz = oo[2] - (int)box.lo[2];
                h.flip_d = flip[0]; h.flip_h = flip[1]; h.flip_w = flip[2];
                h.acc_fp32 = acc_fp32;
                h.fx = first_visit(oo, 0); h.fy = first_visit(oo, 1); h.fz = first_visit(oo, 2);
                if (tta) { h.mode = ci == 0 ? 1 : 2; h.patch_buf = e->patch_buf + (size_t)b * a.num_heads * P; }
                Scope sc(e, st, FAM_HEAD, e->head_flops);
                if (launch_head(h, st) != 0) return fail(e, FNN_E_HIP, "seg head launch failed");
            }
            if (pipelined) HIPCHK(e, hipEventRecord(e->ev_head[k], st));
        }
        if (tta && !keep_features) {
            for (int b = 0; b < nb; ++b) {
                const int *oo = &vp.origins[ids[p0 + b] * 3];
                PatchAccParams q{};
                q.patch_buf = e->patch_buf + (size_t)b * a.num_heads * P;
                q.n_div = (int)combos.size() + 1;
                q.PD = a.patch[0]; q.PH = a.patch[1]; q.PW = a.patch[2]; q.heads = a.num_heads;
                q.gauss = o.use_gaussian ? e->gauss : e->ones;
                q.acc = acc; q.AX = box.hi[0] - box.lo[0]; q.Y = box.hi[1] - box.lo[1]; q.Z = box.hi[2] - box.lo[2];
                q.HP = acc_hp(a);
                q.ox = oo[0] - (int)box.lo[0]; q.oy = oo[1] - (int)box.lo[1]; q.oz = oo[2] - (int)box.lo[2];
                q.acc_fp32 = acc_fp32;
                Scope sc(e, st, FAM_HEAD, 0);
                if (launch_patch_acc(q, st) != 0) return fail(e, FNN_E_HIP, "patch accumulate launch failed");
            }
        }
    }
    if (pipelined) {
        for (int k = 0; k < NP; ++k) {
            HIPCHK(e, hipEventRecord(e->ev_done[k], e->pipe[k]));
            HIPCHK(e, hipStreamWaitEvent(user_st, e->ev_done[k], 0));
        }
    }
    return 0;
}

bool is_pinned_host_ptr(const void *p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

// host -> pinned staging with a few threads: one core's memcpy (~10 GB/s) is slower than the link
void parallel_copy(float *dst, const float *src, size_t n) {
    const size_t min_part = 1u << 20;                                          // floats
    int parts = (int)std::min<size_t>(4, (n + min_part - 1) / min_part);
    if (parts <= 1) { memcpy(dst, src, n * sizeof(float)); return; }
    std::thread th[3];
    const size_t per = (n + parts - 1) / parts;
    for (int t = 1; t < parts; ++t) {
        const size_t lo = per * t, hi = std::min(n, lo + per);
        th[t - 1] = std::thread([=] { if (hi > lo) memcpy(dst + lo, src + lo, (hi - lo) * sizeof(float)); });
    }
    memcpy(dst, src, std::min(n, per) * sizeof(float));
    for (int t = 1; t < parts; ++t) th[t - 1].join();
}

// Issues the upload of every tile of [x_lo, x_hi) x [y_lo, y_hi) (planes x rows, whole z extent) that has not left yet, then makes `st` wait
// for everything issued so far (one copy stream: in order).
int upload_box(fnn_engine *e, int64_t x_lo, int64_t x_hi, int64_t y_lo, int64_t y_hi, hipStream_t st) {
    fnn_engine::Upload &u = e->up;
    if (!u.active) return 0;
    x_lo = std::max<int64_t>(0, x_lo); y_lo = std::max<int64_t>(0, y_lo);
    x_hi = std::min(u.X, x_hi); y_hi = std::min(u.Y, y_hi);
    if (x_hi <= x_lo || y_hi <= y_lo) { x_lo = 0; x_hi = std::min<int64_t>(u.X, 1); y_lo = 0; y_hi = std::min<int64_t>(u.Y, 1); }
    const size_t row = (size_t)u.Z, pitch = (size_t)u.Y * u.Z * sizeof(float);
    bool any = false;
    for (int64_t xs = x_lo / u.slab_x; xs <= (x_hi - 1) / u.slab_x && u.n_issued < u.issued.size(); ++xs)
        for (int64_t yb = y_lo / u.slab_y; yb <= (y_hi - 1) / u.slab_y; ++yb) {
            char &done = u.issued[(size_t)(xs * u.nyb + yb)];
            if (done) continue;
            done = 1; ++u.n_issued; any = true;
            const int64_t x0 = xs * u.slab_x, x1 = std::min(u.X, x0 + u.slab_x), y0 = yb * u.slab_y, y1 = std::min(u.Y, y0 + u.slab_y);
            const size_t width = (size_t)(y1 - y0) * row * sizeof(float), height = (size_t)(x1 - x0);
            const bool whole_rows = y0 == 0 && y1 == u.Y;     // the tile is one contiguous run
            if (u.pinned_src) {
                for (int c = 0; c < u.C; ++c) {
                    const size_t off = (((size_t)c * u.X + x0) * u.Y + y0) * row;
                    if (whole_rows) HIPCHK(e, hipMemcpyAsync(u.dev + off, u.host + off, width * height, hipMemcpyHostToDevice, u.st));
                    else HIPCHK(e, hipMemcpy2DAsync(u.dev + off, pitch, u.host + off, pitch, width, height, hipMemcpyHostToDevice, u.st));
                }
            } else {
                const int k = u.next_stage;
                u.next_stage = (k + 1) % fnn_engine::Upload::RING;
                if (u.stage_used[k]) HIPCHK(e, hipEventSynchronize(u.stage_free[k]));          // its previous tile has left
                const size_t tile = width / sizeof(float) * height;                            // floats per channel, packed
                for (int c = 0; c < u.C; ++c) {
                    const float *src0 = u.host + (((size_t)c * u.X + x0) * u.Y + y0) * row;
                    float *dst0 = u.stage[k] + (size_t)c * tile;
                    if (whole_rows) { parallel_copy(dst0, src0, tile); continue; }               // one contiguous run
                    // rows y0 .. y1 of every plane of the tile: the planes shared out over a few host threads
                    const size_t wf = width / sizeof(float), pf = pitch / sizeof(float);
                    const int parts = (int)std::min<size_t>(4, std::max<size_t>(1, tile / (1u << 18)));
                    auto planes = [=](size_t h0, size_t h1) { for (size_t hx = h0; hx < h1; ++hx) memcpy(dst0 + hx * wf, src0 + hx * pf, wf * sizeof(float)); };
                    std::thread th[3];
                    const size_t per = (height + parts - 1) / parts;
                    for (int t = 1; t < parts; ++t) th[t - 1] = std::thread(planes, std::min(height, per * t), std::min(height, per * (t + 1)));
                    planes(0, std::min(height, per));
                    for (int t = 1; t < parts; ++t) th[t - 1].join();
                }
                for (int c = 0; c < u.C; ++c) {
                    const size_t off = (((size_t)c * u.X + x0) * u.Y + y0) * row;
                    HIPCHK(e, hipMemcpy2DAsync(u.dev + off, pitch, u.stage[k] + (size_t)c * tile, width, width, height, hipMemcpyHostToDevice, u.st));
                }
                HIPCHK(e, hipEventRecord(u.stage_free[k], u.st));
                u.stage_used[k] = true;
            }
        }
    if (any) {
        if (u.ev_next == u.landed.size()) {
            hipEvent_t ev;
            HIPCHK(e, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            u.landed.push_back(ev);
        }
        u.last = u.landed[u.ev_next++];
        HIPCHK(e, hipEventRecord(u.last, u.st));
    }
    if (u.last) HIPCHK(e, hipStreamWaitEvent(st, u.last, 0));
    return 0;
}

// Brings the input volume onto the device and pads it when smaller than the patch.  `slabbed`: the caller's batches call
// upload_box() for the box their patches read (run_patches); otherwise the whole volume is waited for here.
int stage_volume(fnn_engine *e, const float *vol, const int64_t shape[4], const VolPlan &vp, hipStream_t st,
                 const float **vol_dev, bool slabbed = false) {
    const size_t nin = (size_t)shape[0] * shape[1] * shape[2] * shape[3];
    const float *src = vol;
    const bool need_pad = vp.padded[0] != shape[1] || vp.padded[1] != shape[2] || vp.padded[2] != shape[3];
    fnn_engine::Upload &u = e->up;
    u.active = false;
    if (!is_device_ptr(vol)) {
        void *t = e->vol_tmp;
        if (int rc = ensure(e, &t, &e->vol_tmp_bytes, nin * sizeof(float))) return rc;
        e->vol_tmp = (float *)t;
        src = e->vol_tmp;
        u.host = vol; u.dev = e->vol_tmp; u.C = (int)shape[0]; u.X = shape[1]; u.Y = shape[2]; u.Z = shape[3];
        u.pinned_src = is_pinned_host_ptr(vol);
        // tiles of ~4 MiB per channel: an eighth of the rows (at least 16) x as many planes as fill the tile
        const size_t slab_bytes = fnn_knob("FNN_UPLOAD_SLAB_BYTES") ? (size_t)atoll(fnn_knob("FNN_UPLOAD_SLAB_BYTES")) : (4u << 20);   // tests: many tiles in a small volume
        u.slab_y = fnn_knob("FNN_UPLOAD_WHOLE_ROWS") ? u.Y : std::min<int64_t>(u.Y, std::max<int64_t>(16, (u.Y + 7) / 8));          // (knob: x slabs only - the first form)
        const size_t rows_bytes = (size_t)u.slab_y * u.Z * sizeof(float);
        u.slab_x = std::max<int64_t>(1, (int64_t)(slab_bytes / std::max<size_t>(1, rows_bytes)));
        u.nxs = (u.X + u.slab_x - 1) / u.slab_x; u.nyb = (u.Y + u.slab_y - 1) / u.slab_y;
        u.issued.assign((size_t)(u.nxs * u.nyb), 0);
        u.n_issued = 0; u.ev_next = 0; u.last = nullptr;
        if (!u.st) HIPCHK(e, hipStreamCreateWithFlags(&u.st, hipStreamNonBlocking));
        if (!u.go) HIPCHK(e, hipEventCreateWithFlags(&u.go, hipEventDisableTiming));
        if (!u.pinned_src) {
            const size_t need = (size_t)u.C * u.slab_x * rows_bytes;
            if (u.stage_bytes < need) {
                for (int k = 0; k < fnn_engine::Upload::RING; ++k) {
                    if (u.stage[k]) { if (u.stage_used[k]) (void)hipEventSynchronize(u.stage_free[k]); (void)hipHostFree(u.stage[k]); u.stage[k] = nullptr; }
                    u.stage_used[k] = false;
                }
                u.stage_bytes = 0;
                for (int k = 0; k < fnn_engine::Upload::RING; ++k) {
                    HIPCHK(e, hipHostMalloc((void **)&u.stage[k], need, hipHostMallocDefault));
                    if (!u.stage_free[k]) HIPCHK(e, hipEventCreateWithFlags(&u.stage_free[k], hipEventDisableTiming));
                }
                u.stage_bytes = need;
            }
        }
        // the copy stream starts behind whatever the caller's stream still does with the staging area
        HIPCHK(e, hipEventRecord(u.go, st));
        HIPCHK(e, hipStreamWaitEvent(u.st, u.go, 0));
        u.active = true;
        if (!slabbed || need_pad) { if (int rc = upload_box(e, 0, u.X, 0, u.Y, st)) return rc; u.active = false; }
    }
    if (need_pad) {
        const size_t npad = (size_t)shape[0] * vp.padded[0] * vp.padded[1] * vp.padded[2];
        void *t = e->vol_pad;
        if (int rc = ensure(e, &t, &e->vol_pad_bytes, npad * sizeof(float))) return rc;
        e->vol_pad = (float *)t;
        const long long s[3] = {(long long)shape[1], (long long)shape[2], (long long)shape[3]};
        const long long d[3] = {(long long)vp.padded[0], (long long)vp.padded[1], (long long)vp.padded[2]};
        const long long lo[3] = {(long long)vp.lo[0], (long long)vp.lo[1], (long long)vp.lo[2]};
        if (launch_pad_volume(src, e->vol_pad, (int)shape[0], s, d, lo, st) != 0) return fail(e, FNN_E_HIP, "pad launch failed");
        src = e->vol_pad;
    }
    *vol_dev = src;
    return 0;
}

// Origins of the listed patches -> device (the stem conv reads them).
int upload_origins(fnn_engine *e, const VolPlan &vp, const std::vector<int64_t> &ids, hipStream_t st) {
    std::vector<int> host(ids.size() * 3);
    for (size_t i = 0; i < ids.size(); ++i)
        for (int d = 0; d < 3; ++d) host[i * 3 + d] = vp.origins[ids[i] * 3 + d];
    return upload_ints(e, host.data(), host.size(), &e->origins, &e->origins_cap, &e->origins_host, &e->origins_host_cap, &e->origins_ev, st);
}

FinalizeParams make_finalize(fnn_engine *e, const void *acc, const Box &box, const int64_t out_lo[3],
                             const int64_t out_hi[3], const VolPlan &vp, const int64_t shape[4], const fnn_opts &o,
                             int acc_fp32, int mode, void *out) {
    // out_lo / out_hi: box of the UN-PADDED volume that is written
    FinalizeParams f{};
    f.acc = acc;
    f.AX = box.hi[0] - box.lo[0]; f.Y = box.hi[1] - box.lo[1]; f.Z = box.hi[2] - box.lo[2];
    f.HP = acc_hp(e->arch);
    f.lo_x = (int)(out_lo[0] + vp.lo[0] - box.lo[0]); f.lo_y = (int)(out_lo[1] + vp.lo[1] - box.lo[1]);
    f.lo_z = (int)(out_lo[2] + vp.lo[2] - box.lo[2]);
    f.OX = out_hi[0] - out_lo[0]; f.OY = out_hi[1] - out_lo[1]; f.OZ = out_hi[2] - out_lo[2];
    f.out_X = shape[1]; f.out_Y = shape[2]; f.out_Z = shape[3];
    f.out_x = out_lo[0]; f.out_y = out_lo[1]; f.out_z = out_lo[2];
    f.heads = e->arch.num_heads; f.acc_fp32 = acc_fp32; f.out_fp32 = o.out_dtype == FNN_OUT_F32;
    f.mode = mode; f.out = out; f.inf_flag = e->inf_flag;
    return f;
}

int accumulate_whole_volume(fnn_engine *e, int fold, const float *vol_dev, const VolPlan &vp, const fnn_opts &o,
                            Box &box, hipStream_t st) {
    const fnn_arch_desc &a = e->arch;
    const int acc_fp32 = o.accum == FNN_ACC_FP32;
    const size_t esz = acc_fp32 ? 4 : 2;
    const size_t nvox = (size_t)vp.padded[0] * vp.padded[1] * vp.padded[2];
    const size_t bytes = nvox * acc_hp(a) * esz;
    if (int rc = ensure(e, &e->acc, &e->acc_bytes, bytes)) return rc;
    // The whole list in visiting order: the seg head writes every voxel's first visit without reading it, so the
    // 17 GB zero fill (and an eighth of the accumulator reads) is skipped.  Mirroring accumulates through the patch
    // buffer and the generic head kernel: those keep the zero fill.
    static const bool no_fv = fnn_knob("FNN_NO_FIRST_VISIT") != nullptr;            // A-B aid
    const bool fresh = !no_fv && o.n_mirror_axes == 0 && launch_head_first_visit_ok(make_head(e, fold, 0));
    if (!fresh) HIPCHK(e, hipMemsetAsync(e->acc, 0, bytes, st));
    for (int d = 0; d < 3; ++d) { box.lo[d] = 0; box.hi[d] = vp.padded[d]; }
    std::vector<int64_t> ids(vp.n_patches);
    for (int64_t i = 0; i < vp.n_patches; ++i) ids[i] = i;
    return run_patches(e, fold, vol_dev, vp, o, ids, e->origins, box, e->acc, acc_fp32, st, fresh);
}

// The gather kernel's integer tables for a volume: tile starts (+ window bases of axes with more than 64 positions,
// gather.hip gather_tile_windows).  false: an axis has more than 64 tiles over one voxel.
bool gather_tables(const fnn_arch_desc &a, const VolPlan &vp, std::vector<int> *tab, int off[3]) {
    const long long *st[3] = {(const long long *)vp.steps[0].data(), (const long long *)vp.steps[1].data(), (const long long *)vp.steps[2].data()};
    const int n[3] = {(int)vp.steps[0].size(), (int)vp.steps[1].size(), (int)vp.steps[2].size()};
    const int ext[3] = {a.patch[0], a.patch[1], a.patch[2]};
    const long long pad[3] = {(long long)vp.padded[0], (long long)vp.padded[1], (long long)vp.padded[2]};
    size_t count = 0;
    if (!gather_tile_windows(st, n, ext, pad, nullptr, off, &count)) return false;
    if (tab) { tab->assign(count, 0); (void)gather_tile_windows(st, n, ext, pad, tab->data(), off, &count); }
    return true;
}
void gather_set_tables(GatherParams &g, const int *dev, const int off[3]) {
    g.steps = dev;
    g.base_x = off[0] >= 0 ? dev + off[0] : nullptr; g.base_y = off[1] >= 0 ? dev + off[1] : nullptr; g.base_z = off[2] >= 0 ? dev + off[2] : nullptr;
    g.windowed = 1;
}

// Plan of the gather path (gather.hip) for a volume: how many x layers of patches are kept at a time.
struct GatherPlan { bool ok = false; int n_eval = 1, ring = 0, cover = 1; size_t layer_items = 0, feat_bytes = 0; const char *why = ""; };

// No gather when the head does not fit the kernel's registers or when not even the layers that cover one output slab fit
// next to what is already allocated; otherwise the whole volume's patches when they fit (one launch at the end), else
// a ring of `cover` layers with one launch per output slab.
GatherPlan gather_plan(fnn_engine *e, const VolPlan &vp, const fnn_opts &o, size_t pending_bytes = 0) {
    GatherPlan gp;
    gp.why = "FNN_NO_GATHER is set or the output is not fp16";
    if (!e->gather_enabled || o.out_dtype != FNN_OUT_F16) return gp;
    const Layer &H = e->layers[e->head_src];
    GatherParams g{};
    g.heads = e->arch.num_heads; g.C = H.cout_pad; g.PD = e->arch.patch[0]; g.PH = e->arch.patch[1]; g.PW = e->arch.patch[2];
    g.nx = (int)vp.steps[0].size(); g.ny = (int)vp.steps[1].size(); g.nz = (int)vp.steps[2].size();
    gp.n_eval = 1 + (int)mirror_combos(o).size();
    g.n_eval = gp.n_eval; g.n_pass = e->n_gpass;
    { int off[3]; g.windowed = gather_tables(e->arch, vp, nullptr, off) ? 1 : 0; }
    gp.why = "the network's head does not fit the gather kernel (a normalised last layer of <= 32 channels, <= 8 evaluations per patch, <= 64 tiles of one axis over a voxel)";
    if (!H.has_norm || e->head_ksteps != 1 || !gather_ok(g)) return gp;
    gp.why = "not enough free HBM for the patch activations that cover one output slab";
    const auto &sx = vp.steps[0];
    const int nx = (int)sx.size();
    gp.cover = 1;
    for (int i = 0; i < nx; ++i) {                            // layers still needed when layer i has just been produced
        int c = 0;
        for (int j = 0; j <= i; ++j) c += sx[j] + g.PD > sx[i];
        gp.cover = std::max(gp.cover, c);
    }
    const size_t P = (size_t)g.PD * g.PH * g.PW;
    gp.layer_items = (size_t)vp.steps[1].size() * vp.steps[2].size();
    const size_t layer_bytes = gp.layer_items * gp.n_eval * P * H.cout_pad * sizeof(f16);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return gp;
    const double frac = fnn_knob("FNN_GATHER_MEM_FRACTION") ? atof(fnn_knob("FNN_GATHER_MEM_FRACTION")) : 0.6;
    const int force_ring = fnn_knob("FNN_GATHER_RING") ? atoi(fnn_knob("FNN_GATHER_RING")) : 0;     // tests: a ring although all fits
    // (the accumulators are not needed then; `pending_bytes`: what the caller still allocates after this plan - output / label staging)
    const double budget = frac * ((double)(free_b + e->feat_bytes + e->acc_bytes) - (double)pending_bytes);
    if (!force_ring && (double)layer_bytes * nx <= budget) gp.ring = nx;
    else if ((double)layer_bytes * gp.cover <= budget) gp.ring = std::min(nx, std::max(gp.cover, force_ring));
    else return gp;
    gp.feat_bytes = layer_bytes * gp.ring;
    gp.ok = true;
    return gp;
}

int gather_whole_volume(fnn_engine *e, int fold, const float *vol_dev, const VolPlan &vp, const int64_t shape[4],
                        const fnn_opts &o, const GatherPlan &gp, int mode, void *out, void *labels, const int *lab_order,
                        hipStream_t st) {
    const fnn_arch_desc &a = e->arch;
    const Layer &H = e->layers[e->head_src];
    if (gp.feat_bytes > e->feat_bytes && e->acc) { (void)hipFree(e->acc); e->acc = nullptr; e->acc_bytes = 0; }
    if (int rc = ensure(e, &e->feat, &e->feat_bytes, gp.feat_bytes)) return rc;
    const int64_t n_slots = (int64_t)gp.layer_items * gp.ring;
    if (int rc = ensure(e, &e->featss, &e->featss_bytes, (size_t)n_slots * gp.n_eval * 2 * H.cout_pad * sizeof(float))) return rc;
    if (int rc = ensure(e, &e->featssh, &e->featssh_bytes, (size_t)n_slots * gp.n_eval * 2 * H.cout_pad * sizeof(f16))) return rc;
    std::vector<int> steps;
    int win_off[3];
    if (!gather_tables(a, vp, &steps, win_off)) return fail(e, FNN_E_UNSUPPORTED, "more than 64 tiles of one axis over a voxel");
    if (int rc = upload_ints(e, steps.data(), steps.size(), &e->steps_dev, &e->steps_cap, &e->steps_host, &e->steps_host_cap, &e->steps_ev, st)) return rc;
    Box box;
    for (int d = 0; d < 3; ++d) { box.lo[d] = 0; box.hi[d] = vp.padded[d]; }
    const FoldWeights &fw = e->folds[fold];
    GatherParams g{};
    g.feat = (const f16 *)e->feat; g.fss = (const float *)e->featss; g.fssh = (const unsigned short *)e->featssh; g.C = H.cout_pad;
    g.n_eval = gp.n_eval; g.n_slots = (int)n_slots; g.ring = gp.ring;
    {
        const auto combos = mirror_combos(o);
        g.flipmask[0] = 0;
        for (size_t ci = 0; ci < combos.size() && ci + 1 < 8; ++ci) {
            int m = 0;
            for (int ax : combos[ci]) m |= 1 << ax;
            g.flipmask[ci + 1] = m;
        }
    }
    g.slope = H.act ? a.slope : 1.f;
    gather_set_tables(g, e->steps_dev, win_off);
    g.nx = (int)vp.steps[0].size(); g.ny = (int)vp.steps[1].size(); g.nz = (int)vp.steps[2].size();
    g.PD = a.patch[0]; g.PH = a.patch[1]; g.PW = a.patch[2];
    g.wpk = fw.wpk + e->head_w_off; g.bias = fw.fparam + e->head_bias_off; g.heads = a.num_heads; g.hblocks = e->hblocks;
    g.n_pass = e->n_gpass; g.pass_wpk = fw.wpk + e->gpass_w_off; g.pass_bias = fw.fparam + e->gpass_bias_off;
    g.gauss = o.use_gaussian ? e->gauss : e->ones;
    g.lo_x = (int)vp.lo[0]; g.lo_y = (int)vp.lo[1]; g.lo_z = (int)vp.lo[2];
    g.OX = shape[1]; g.OY = shape[2]; g.OZ = shape[3];
    g.y_lo = 0; g.y_hi = (int)shape[2]; g.z_lo = 0; g.z_hi = (int)shape[3]; g.slot_tab = nullptr;
    g.acc_mode = o.accum; g.out_fp32 = 0;
    g.out_vec = out && shape[3] % 8 == 0 && ((size_t)out % 16) == 0;
    g.mode = mode; g.out = out; g.labels = labels; g.label_u16 = e->label_u16; g.order = lab_order; g.inf_flag = e->inf_flag;
    auto launch = [&](int x_lo, int x_hi, double n_patches) {
        g.x_lo = x_lo; g.x_hi = x_hi;
        Scope sc(e, st, FAM_HEAD, e->head_flops * n_patches * gp.n_eval);
        return launch_gather(g, st) == 0 ? 0 : fail(e, FNN_E_HIP, "gather launch failed");
    };
    if (gp.ring == g.nx) {                                     // every patch of the volume is kept: one pass at the end
        std::vector<int64_t> ids(vp.n_patches);
        for (int64_t i = 0; i < vp.n_patches; ++i) ids[i] = i;
        if (int rc = run_patches(e, fold, vol_dev, vp, o, ids, e->origins, box, nullptr, 0, st, false, true, 0, n_slots)) return rc;
        return launch(0, (int)shape[1], (double)vp.n_patches);
    }
    // ring: after x layer ix the output slab up to the next layer's first voxel is complete (padded coordinates
    // [steps[ix], steps[ix + 1]); the un-padded range is clamped); the next layer then overwrites the oldest slot
    const int64_t L = (int64_t)gp.layer_items;
    for (int ix = 0; ix < g.nx; ++ix) {
        std::vector<int64_t> ids(L);
        for (int64_t i = 0; i < L; ++i) ids[i] = ix * L + i;
        if (int rc = run_patches(e, fold, vol_dev, vp, o, ids, e->origins + ix * L * 3, box, nullptr, 0, st, false, true,
                                 (ix % gp.ring) * L, n_slots)) return rc;
        const int64_t p_lo = ix == 0 ? 0 : vp.steps[0][ix], p_hi = ix + 1 < g.nx ? vp.steps[0][ix + 1] : vp.padded[0];
        const int x_lo = (int)std::max<int64_t>(0, p_lo - vp.lo[0]), x_hi = (int)std::min<int64_t>(shape[1], p_hi - vp.lo[0]);
        if (x_hi > x_lo) if (int rc = launch(x_lo, x_hi, (double)L)) return rc;
    }
    return 0;
}

int predict_impl(fnn_engine *e, int fold0, int n_folds, const float *vol, const int64_t shape[4], const fnn_opts *o,
                 void *out, void *labels) {
    for (int f = fold0; f < fold0 + n_folds; ++f)
        if (int rc = check_ready(e, f, o)) return rc;
    if (!vol || (!out && !labels) || !shape) return fail(e, FNN_E_INVALID, "NULL argument");
    const fnn_arch_desc &a = e->arch;
    if (shape[0] != a.in_channels) return fail(e, FNN_E_INVALID, "input has %lld channels, the network expects %d", (long long)shape[0], a.in_channels);
    HIPCHK(e, hipSetDevice(e->device));
    hipStream_t st = (hipStream_t)o->stream;
    VolPlan vp;
    if (plan_volume(a, shape + 1, o->tile_step_size, vp) != 0) return fail(e, FNN_E_INVALID, "invalid volume shape / step size");
    const float *vol_dev = nullptr;
    if (int rc = stage_volume(e, vol, shape, vp, st, &vol_dev, true)) return rc;
    struct UploadDone { fnn_engine *e; ~UploadDone() { e->up.active = false; } } upload_done{e};
    {
        std::vector<int64_t> all(vp.n_patches);
        for (int64_t i = 0; i < vp.n_patches; ++i) all[i] = i;
        if (int rc = upload_origins(e, vp, all, st)) return rc;
    }
    const size_t nvox_out = (size_t)shape[1] * shape[2] * shape[3];
    const size_t nout = (size_t)a.num_heads * nvox_out;
    const size_t osz = o->out_dtype == FNN_OUT_F32 ? 4 : 2;
    const bool want_logits = out != nullptr;
    const bool out_on_dev = out && is_device_ptr(out);
    const bool lab_on_dev = labels && is_device_ptr(labels);
    const size_t lab_bytes = nvox_out * (e->label_u16 ? 2 : 1);
    // (the plan first: nothing is allocated yet when it refuses the request, and it decides how labels are formed; the
    // staging buffers that ARE allocated afterwards - fp16 / fp32 logits when the caller's are on the host or the labels
    // come from materialised logits, the label map of a host caller - come out of its budget)
    size_t pending = 0;
    {
        const bool maybe_direct = labels && !want_logits && n_folds == 1 && e->n_gpass == 1;
        if (!maybe_direct && !out_on_dev && nout * osz > e->out_tmp_bytes) pending += nout * osz - e->out_tmp_bytes;
        if (labels && !lab_on_dev) pending += lab_bytes;
    }
    const GatherPlan gp = gather_plan(e, vp, *o, pending);
    if (!gp.ok && o->accum == FNN_ACC_FP16_AUTOCAST)
        return fail(e, FNN_E_UNSUPPORTED, "FNN_ACC_FP16_AUTOCAST needs the gather path: %s", gp.why);
    // argmax straight from the accumulators / the gather kernel's registers; with more than 63 classes the gather kernel
    // runs in passes over the heads, so the labels come from its logits
    // (accumulate path: labels_from_acc_coop_kernel covers up to 32 lanes of 8 channels per voxel - 254 classes; beyond, the
    // labels come from the logits like an ensemble's)
    const bool labels_direct = labels && !want_logits && n_folds == 1 && !(gp.ok && e->n_gpass > 1) && (gp.ok || acc_hp(a) <= 256);
    void *out_dev = out;
    if (!labels_direct && !out_on_dev) {
        if (int rc = ensure(e, &e->out_tmp, &e->out_tmp_bytes, nout * osz)) return rc;
        out_dev = e->out_tmp;
    }
    void *lab_dev = labels;
    void *lab_tmp = nullptr;
    const int *lab_order = e->label_mode == FNN_LABELS_REGIONS ? e->label_order : nullptr;
    if (labels && !e->label_u16 && e->label_mode == FNN_LABELS_ARGMAX && a.num_heads > 256)
        return fail(e, FNN_E_INVALID, "%d classes do not fit uint8 labels: fnn_set_label_rule(..., FNN_LABEL_U16)", a.num_heads);
    if (labels && !lab_on_dev) { HIPCHK(e, hipMalloc(&lab_tmp, lab_bytes)); lab_dev = lab_tmp; }
    HIPCHK(e, hipMemsetAsync(e->inf_flag, 0, sizeof(int), st));
    e->ev_used = 0; e->klog.clear();
    const int acc_fp32 = o->accum == FNN_ACC_FP32;
    const int64_t zero3[3] = {0, 0, 0}, full3[3] = {shape[1], shape[2], shape[3]};
    int rc = 0;
    for (int f = 0; f < n_folds && rc == 0; ++f) {
        if (gp.ok) {
            rc = gather_whole_volume(e, fold0 + f, vol_dev, vp, shape, *o, gp, f > 0 ? 1 : 0, labels_direct ? nullptr : out_dev,
                                     labels_direct ? lab_dev : nullptr, lab_order, st);
            continue;
        }
        Box box;
        rc = accumulate_whole_volume(e, fold0 + f, vol_dev, vp, *o, box, st);
        if (rc) break;
        FinalizeParams fp = make_finalize(e, e->acc, box, zero3, full3, vp, shape, *o, acc_fp32, f > 0 ? 1 : 0, out_dev);
        Scope sc(e, st, FAM_FINAL, 0);
        if (labels_direct) { if (launch_labels_from_acc(fp, lab_dev, e->label_u16, lab_order, st) != 0) rc = fail(e, FNN_E_HIP, "labels launch failed"); }
        else if (launch_finalize(fp, st) != 0) rc = fail(e, FNN_E_HIP, "finalize launch failed");
    }
    if (rc == 0 && !labels_direct && n_folds > 1)
        if (launch_scale_output(out_dev, o->out_dtype == FNN_OUT_F32, (long long)nout, n_folds, e->inf_flag, st) != 0)
            rc = fail(e, FNN_E_HIP, "scale launch failed");
    if (rc == 0 && labels && !labels_direct)
        if (launch_argmax(out_dev, o->out_dtype == FNN_OUT_F32, a.num_heads, (long long)nvox_out, lab_dev, e->label_u16, lab_order, st) != 0)
            rc = fail(e, FNN_E_HIP, "argmax launch failed");
    int flag = 0;
    if (rc == 0) {
        hipError_t r1 = hipMemcpyAsync(&flag, e->inf_flag, sizeof(int), hipMemcpyDeviceToHost, st);
        if (r1 == hipSuccess && want_logits && !out_on_dev) r1 = hipMemcpyAsync(out, out_dev, nout * osz, hipMemcpyDeviceToHost, st);
        if (r1 == hipSuccess && labels && !lab_on_dev) r1 = hipMemcpyAsync(labels, lab_dev, lab_bytes, hipMemcpyDeviceToHost, st);
        if (r1 == hipSuccess) r1 = hipStreamSynchronize(st);
        if (r1 != hipSuccess) rc = fail(e, FNN_E_HIP, "copy back failed: %s", hipGetErrorString(r1));
    }
    if (lab_tmp) (void)hipFree(lab_tmp);
    if (rc) return rc;
    if (e->profiling) collect_profile(e, vp.n_patches * n_folds);
    if (flag)
        return fail(e, FNN_E_INF, "Encountered inf in predicted array. Aborting... If this problem persists, reduce "
                                  "value_scaling_factor in compute_gaussian or increase the dtype of predicted_logits to fp32");
    return 0;
}

}  // namespace

// ===========================================================================
// C ABI
// ===========================================================================
extern "C" {

int fnn_abi_version(void) { return FNN_ABI_VERSION; }

const char *fnn_last_error(const fnn_engine *e) { return e ? e->err.c_str() : g_err.c_str(); }

int fnn_create(const fnn_arch_desc *arch, int device, int max_batch, fnn_engine **out) {
    if (!arch || !out) return fail(nullptr, FNN_E_INVALID, "NULL argument");
    // up to 64 patches per forward, or - small patches (a 40 x 56 x 40 plan, a 2-D slice) - as many as give a forward 2^27 voxels
    // (64 patches of 128^3), at most 512: the deep layers of a small patch have too few voxels per item to fill the chip
    {
        const long long pv = (long long)arch->patch[0] * arch->patch[1] * arch->patch[2];
        const long long cap = pv > 0 ? std::max<long long>(64, std::min<long long>(512, (1ll << 27) / pv)) : 64;
        if (max_batch < 1 || max_batch > cap) return fail(nullptr, FNN_E_INVALID, "max_batch must be 1..%lld for this patch size", cap);
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, FNN_E_HIP, "no HIP device is available: the MI355X engine has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(nullptr, FNN_E_INVALID, "device %d out of range (%d visible)", device, ndev);
    fnn_engine *e = new fnn_engine();
    e->arch = *arch; e->device = device; e->max_batch = max_batch;
    e->fuse_enabled = fnn_knob("FNN_NO_FUSE") == nullptr;
    if (const char *v = fnn_knob("FNN_FUSE_STEM")) e->fuse_stem = atoi(v) != 0 ? 1 : 0;
    if (const char *v = fnn_knob("FNN_FUSE_TCONV")) e->fuse_tconv = atoi(v) != 0;
    e->gather_enabled = fnn_knob("FNN_NO_GATHER") == nullptr;
    if (e->arch.eps <= 0) e->arch.eps = 1e-5f;
    int rc = build_plan(e);
    if (rc != 0) { g_err = e->err; delete e; return rc; }
    auto bail = [&](const char *what, hipError_t r) {
        fail(nullptr, FNN_E_HIP, "%s failed: %s", what, hipGetErrorString(r));
        fnn_destroy(e);
        return FNN_E_HIP;
    };
    hipError_t r;
    if ((r = hipSetDevice(device)) != hipSuccess) return bail("hipSetDevice", r);
    if ((r = hipMalloc((void **)&e->act, e->act_halves * max_batch * sizeof(f16))) != hipSuccess) return bail("hipMalloc(activations)", r);
    if ((r = hipMalloc((void **)&e->stats, e->stats_doubles * max_batch * sizeof(double))) != hipSuccess) return bail("hipMalloc(stats)", r);
    if ((r = hipMalloc((void **)&e->ss, (e->ss_count * max_batch * 3 + 8) * sizeof(float))) != hipSuccess) return bail("hipMalloc(scale/shift)", r);
    if ((r = hipMalloc((void **)&e->inf_flag, sizeof(int))) != hipSuccess) return bail("hipMalloc(flag)", r);
    {
        const size_t P = (size_t)arch->patch[0] * arch->patch[1] * arch->patch[2];
        std::vector<uint16_t> one(P, 0x3c00);                                       // fp16 1.0
        if ((r = hipMalloc((void **)&e->ones, P * 2)) != hipSuccess) return bail("hipMalloc(ones)", r);
        if ((r = hipMemcpy(e->ones, one.data(), P * 2, hipMemcpyHostToDevice)) != hipSuccess) return bail("hipMemcpy(ones)", r);
    }
    *out = e;
    return 0;
}

void fnn_destroy(fnn_engine *e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    for (auto &f : e->folds) { if (f.wpk) (void)hipFree(f.wpk); if (f.fparam) (void)hipFree(f.fparam); }
    for (int k = 0; k < fnn_engine::MAXP; ++k) {
        if (e->pipe[k]) (void)hipStreamDestroy(e->pipe[k]);
        if (e->ev_head[k]) (void)hipEventDestroy(e->ev_head[k]);
        if (e->ev_done[k]) (void)hipEventDestroy(e->ev_done[k]);
        if (k > 0) { (void)hipFree(e->actp[k]); (void)hipFree(e->statsp[k]); (void)hipFree(e->ssp[k]); }
    }
    if (e->ev_start) (void)hipEventDestroy(e->ev_start);
    for (hipEvent_t ev : e->up.landed) (void)hipEventDestroy(ev);
    for (int k = 0; k < fnn_engine::Upload::RING; ++k) {
        if (e->up.stage[k]) (void)hipHostFree(e->up.stage[k]);
        if (e->up.stage_free[k]) (void)hipEventDestroy(e->up.stage_free[k]);
    }
    if (e->up.go) (void)hipEventDestroy(e->up.go);
    if (e->up.st) (void)hipStreamDestroy(e->up.st);
    if (e->steps_ev) (void)hipEventDestroy(e->steps_ev);
    if (e->origins_ev) (void)hipEventDestroy(e->origins_ev);
    if (e->steps_host) (void)hipHostFree(e->steps_host);
    if (e->origins_host) (void)hipHostFree(e->origins_host);
    void *ptrs[] = {e->feat, e->featss, e->featssh, e->steps_dev, e->ones, e->label_order, e->act, e->stats, e->ss, e->gauss, e->inf_flag, e->origins, e->acc, e->vol_tmp, e->vol_pad, e->out_tmp, e->patch_buf};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (auto &ev : e->evs) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    delete e;
}

int64_t fnn_weight_count(const fnn_engine *e) { return e ? e->blob_count : -1; }

int fnn_load_weights(fnn_engine *e, int fold, const float *blob, int64_t count) {
    if (!e) return FNN_E_INVALID;
    if (!blob) return fail(e, FNN_E_INVALID, "NULL blob");
    if (count != e->blob_count) return fail(e, FNN_E_INVALID, "weight blob has %lld values, expected %lld", (long long)count, (long long)e->blob_count);
    if (fold < 0 || fold >= 64) return fail(e, FNN_E_INVALID, "fold index out of range");
    HIPCHK(e, hipSetDevice(e->device));
    if ((int)e->folds.size() <= fold) e->folds.resize(fold + 1);
    FoldWeights &fw = e->folds[fold];
    std::vector<uint16_t> wpk(e->wpk_halves, 0);
    std::vector<float> fp(e->fparam_floats, 0.f);
    for (const Layer &L : e->layers) {
        if (L.type == Layer::POOL || L.type == Layer::COMBINE || L.type == Layer::GATHER) continue;
        const float *W = blob + L.blob_w;
        if (L.type == Layer::STEM) {
            const int T = L.k[0] * L.k[1] * L.k[2], C = L.cin_real[0];
            for (int c = 0; c < C; ++c)
                for (int t = 0; t < T; ++t)
                    for (int co = 0; co < L.cout_real; ++co)
                        fp[L.w_off + ((size_t)c * T + t) * L.cout_pad + co] = W[((size_t)co * C + c) * T + t];
            if (L.mfma_stem) {                                        // MFMA "A" fragments [cout block][k-step]: lane (cout, k-group), k = c * T + tap
                const int KST = stem_mfma_ksteps(C, T);
                for (int cb = 0; cb < L.cout_pad / 16; ++cb)
                    for (int ks = 0; ks < KST; ++ks)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 8; ++j) {
                                const int co = cb * 16 + (lane & 15);
                                int c = 0, tap = 0;
                                const bool live = stem_mfma_kmap(C, T, ks, 8 * (lane >> 4) + j, &c, &tap) && co < L.cout_real;
                                wpk[L.w_off2 + ((size_t)(cb * KST + ks) * 64 + lane) * 8 + j] =
                                    f2h_bits(live ? W[((size_t)co * C + c) * T + tap] : 0.f);
                            }
            }
        } else if (L.type == Layer::CONV) {
            if (L.packing == FNN_PACK_ZP)
                conv_zp_pack(W, L.cout_real, L.cout_pad, L.cin_real[0], L.cin_pad[0], L.n_src > 1 ? L.cin_real[1] : 0,
                             L.n_src > 1 ? L.cin_pad[1] : 0, wpk.data() + L.w_off);
            else if (L.fp8) pack_conv_fp8(L, W, (uint8_t *)(wpk.data() + L.w_off), fp.data() + L.oscale_off);
            else pack_conv(L, W, wpk.data() + L.w_off);
        } else {
            pack_tconv(L, W, wpk.data() + L.w_off);
        }
        for (int c = 0; c < L.cout_real; ++c) {
            // A conv bias in front of an InstanceNorm cancels exactly (the norm removes the channel mean), so it is
            // dropped: the raw conv outputs are stored in fp16, and a large bias would only cost them resolution
            // (|bias| = 10 in front of a unit-variance channel: 2^-7 instead of 2^-11 of sigma).
            fp[L.bias_off + c] = (L.has_bias && !L.has_norm) ? blob[L.blob_b + c] : 0.f;
            if (L.has_norm) { fp[L.gamma_off + c] = blob[L.blob_g + c]; fp[L.beta_off + c] = blob[L.blob_beta + c]; }
        }
    }
    pack_head(e->arch.num_heads, e->arch.features[0], e->hblocks, e->head_ksteps, blob + e->blob_head_w, wpk.data() + e->head_w_off);
    for (int h = 0; h < e->arch.num_heads; ++h) fp[e->head_bias_off + h] = blob[e->blob_head_b + h];
    fp[e->head_bias_off + e->arch.num_heads] = 1.f;         // accumulator channel `heads`: 1 * gaussian = the weight itself
    for (int k = 0; k < (e->n_gpass > 1 ? e->n_gpass : 0); ++k) {      // gather passes: heads 63 k .. + cnt - 1, then the weight-sum row
        const int h0 = 63 * k, cnt = std::min(63, e->arch.num_heads - h0), cin = e->arch.features[0];
        pack_head(cnt, cin, 4, 1, blob + e->blob_head_w + (size_t)h0 * cin, wpk.data() + e->gpass_w_off + (size_t)k * 4 * 512);
        for (int h = 0; h < cnt; ++h) fp[e->gpass_bias_off + (size_t)k * 64 + h] = blob[e->blob_head_b + h0 + h];
        fp[e->gpass_bias_off + (size_t)k * 64 + cnt] = 1.f;
    }
    if (!fw.wpk) HIPCHK(e, hipMalloc((void **)&fw.wpk, wpk.size() * 2 + 1024));   // (+ 1 KB: see fnn_op_conv3d)
    if (!fw.fparam) HIPCHK(e, hipMalloc((void **)&fw.fparam, fp.size() * 4));
    HIPCHK(e, hipMemcpy(fw.wpk, wpk.data(), wpk.size() * 2, hipMemcpyHostToDevice));
    HIPCHK(e, hipMemcpy(fw.fparam, fp.data(), fp.size() * 4, hipMemcpyHostToDevice));
    fw.loaded = true;
    return 0;
}

int fnn_set_gaussian(fnn_engine *e, const uint16_t *half_bits, int64_t count) {
    if (!e) return FNN_E_INVALID;
    const int64_t P = (int64_t)e->arch.patch[0] * e->arch.patch[1] * e->arch.patch[2];
    if (!half_bits || count != P) return fail(e, FNN_E_INVALID, "gaussian must have %lld values", (long long)P);
    HIPCHK(e, hipSetDevice(e->device));
    if (!e->gauss) HIPCHK(e, hipMalloc((void **)&e->gauss, P * 2));
    HIPCHK(e, hipMemcpy(e->gauss, half_bits, P * 2, hipMemcpyHostToDevice));
    return 0;
}

int fnn_predict_volume(fnn_engine *e, int fold, const float *vol, const int64_t shape[4], const fnn_opts *opts, void *out) {
    if (!e) return FNN_E_INVALID;
    return predict_impl(e, fold, 1, vol, shape, opts, out, nullptr);
}

int fnn_predict_volume_ensemble(fnn_engine *e, int n_folds, const float *vol, const int64_t shape[4], const fnn_opts *opts, void *out) {
    if (!e) return FNN_E_INVALID;
    if (n_folds < 1) return fail(e, FNN_E_INVALID, "n_folds must be >= 1");
    return predict_impl(e, 0, n_folds, vol, shape, opts, out, nullptr);
}

int fnn_predict_labels(fnn_engine *e, int n_folds, const float *vol, const int64_t shape[4], const fnn_opts *opts, void *labels) {
    if (!e) return FNN_E_INVALID;
    if (n_folds < 1) return fail(e, FNN_E_INVALID, "n_folds must be >= 1");
    if (!labels) return fail(e, FNN_E_INVALID, "NULL labels");
    return predict_impl(e, 0, n_folds, vol, shape, opts, nullptr, labels);
}

int64_t fnn_accumulator_channels(const fnn_engine *e) { return e ? acc_hp(e->arch) : -1; }

int fnn_accumulate_patches(fnn_engine *e, int fold, const float *vol, const int64_t shape[4], const fnn_opts *opts,
                           const int64_t *patch_ids, int64_t n_ids, const int64_t box_lo[3], const int64_t box_hi[3],
                           void *acc) {
    if (int rc = check_ready(e, fold, opts)) return rc;
    if (int rc = no_autocast(e, opts, "fnn_accumulate_patches")) return rc;
    if (!vol || !acc || !box_lo || !box_hi || (n_ids > 0 && !patch_ids)) return fail(e, FNN_E_INVALID, "NULL argument");
    if (!is_device_ptr(acc)) return fail(e, FNN_E_INVALID, "accumulators must be device memory");
    HIPCHK(e, hipSetDevice(e->device));
    hipStream_t st = (hipStream_t)opts->stream;
    VolPlan vp;
    if (plan_volume(e->arch, shape + 1, opts->tile_step_size, vp) != 0) return fail(e, FNN_E_INVALID, "invalid volume shape / step size");
    Box box;
    for (int d = 0; d < 3; ++d) {
        box.lo[d] = box_lo[d]; box.hi[d] = box_hi[d];
        if (box.lo[d] < 0 || box.hi[d] > vp.padded[d] || box.lo[d] >= box.hi[d]) return fail(e, FNN_E_INVALID, "box out of bounds");
    }
    std::vector<int64_t> ids(patch_ids, patch_ids + n_ids);
    for (int64_t id : ids) if (id < 0 || id >= vp.n_patches) return fail(e, FNN_E_INVALID, "patch id out of range");
    if (ids.empty()) return 0;
    const float *vol_dev = nullptr;
    if (int rc = stage_volume(e, vol, shape, vp, st, &vol_dev)) return rc;
    if (int rc = upload_origins(e, vp, ids, st)) return rc;
    e->ev_used = 0; e->klog.clear();
    if (int rc = run_patches(e, fold, vol_dev, vp, *opts, ids, e->origins, box, acc, opts->accum == FNN_ACC_FP32, st)) return rc;
    if (e->profiling) { HIPCHK(e, hipStreamSynchronize(st)); collect_profile(e, n_ids); }
    return 0;
}

int fnn_normalize_box(fnn_engine *e, const void *acc, const int64_t shape[4], const fnn_opts *opts,
                      const int64_t box_lo[3], const int64_t box_hi[3], const int64_t out_lo[3], const int64_t out_hi[3],
                      void *out) {
    if (!e || !opts) return FNN_E_INVALID;
    if (int rc = no_autocast(e, opts, "fnn_normalize_box")) return rc;
    if (!acc || !out || !box_lo || !box_hi || !out_lo || !out_hi) return fail(e, FNN_E_INVALID, "NULL argument");
    if (!is_device_ptr(acc) || !is_device_ptr(out)) return fail(e, FNN_E_INVALID, "fnn_normalize_box needs device pointers");
    HIPCHK(e, hipSetDevice(e->device));
    hipStream_t st = (hipStream_t)opts->stream;
    VolPlan vp;
    if (plan_volume(e->arch, shape + 1, opts->tile_step_size, vp) != 0) return fail(e, FNN_E_INVALID, "invalid volume shape / step size");
    Box box;
    for (int d = 0; d < 3; ++d) {
        box.lo[d] = box_lo[d]; box.hi[d] = box_hi[d];
        if (out_lo[d] < 0 || out_hi[d] > shape[1 + d] || out_lo[d] >= out_hi[d]) return fail(e, FNN_E_INVALID, "output box out of bounds");
        if (out_lo[d] + vp.lo[d] < box.lo[d] || out_hi[d] + vp.lo[d] > box.hi[d])
            return fail(e, FNN_E_INVALID, "output box is not covered by the accumulator box");
    }
    HIPCHK(e, hipMemsetAsync(e->inf_flag, 0, sizeof(int), st));
    FinalizeParams f = make_finalize(e, acc, box, out_lo, out_hi, vp, shape, *opts, opts->accum == FNN_ACC_FP32, 0, out);
    if (launch_finalize(f, st) != 0) return fail(e, FNN_E_HIP, "finalize launch failed");
    int flag = 0;
    HIPCHK(e, hipMemcpyAsync(&flag, e->inf_flag, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(e, hipStreamSynchronize(st));
    if (flag) return fail(e, FNN_E_INF, "Encountered inf in predicted array.");
    return 0;
}

int fnn_labels_box(fnn_engine *e, const void *acc, const int64_t shape[4], const fnn_opts *opts,
                   const int64_t box_lo[3], const int64_t box_hi[3], const int64_t out_lo[3], const int64_t out_hi[3],
                   void *labels) {
    if (!e || !opts) return FNN_E_INVALID;
    if (int rc = no_autocast(e, opts, "fnn_labels_box")) return rc;
    if (!acc || !labels || !box_lo || !box_hi || !out_lo || !out_hi) return fail(e, FNN_E_INVALID, "NULL argument");
    if (!is_device_ptr(acc) || !is_device_ptr(labels)) return fail(e, FNN_E_INVALID, "fnn_labels_box needs device pointers");
    if (acc_hp(e->arch) > 256)
        return fail(e, FNN_E_UNSUPPORTED, "fnn_labels_box serves up to 254 classes (%d here): take the logits (fnn_normalize_box) and fnn_argmax_labels", e->arch.num_heads);
    if (!e->label_u16 && e->label_mode == FNN_LABELS_ARGMAX && e->arch.num_heads > 256)
        return fail(e, FNN_E_INVALID, "%d classes do not fit uint8 labels: fnn_set_label_rule(..., FNN_LABEL_U16)", e->arch.num_heads);
    HIPCHK(e, hipSetDevice(e->device));
    hipStream_t st = (hipStream_t)opts->stream;
    VolPlan vp;
    if (plan_volume(e->arch, shape + 1, opts->tile_step_size, vp) != 0) return fail(e, FNN_E_INVALID, "invalid volume shape / step size");
    Box box;
    for (int d = 0; d < 3; ++d) {
        box.lo[d] = box_lo[d]; box.hi[d] = box_hi[d];
        if (out_lo[d] < 0 || out_hi[d] > shape[1 + d] || out_lo[d] >= out_hi[d]) return fail(e, FNN_E_INVALID, "output box out of bounds");
        if (out_lo[d] + vp.lo[d] < box.lo[d] || out_hi[d] + vp.lo[d] > box.hi[d])
            return fail(e, FNN_E_INVALID, "output box is not covered by the accumulator box");
    }
    HIPCHK(e, hipMemsetAsync(e->inf_flag, 0, sizeof(int), st));
    FinalizeParams f = make_finalize(e, acc, box, out_lo, out_hi, vp, shape, *opts, opts->accum == FNN_ACC_FP32, 0, nullptr);
    const int *order = e->label_mode == FNN_LABELS_REGIONS ? e->label_order : nullptr;
    if (launch_labels_from_acc(f, labels, e->label_u16, order, st) != 0) return fail(e, FNN_E_HIP, "labels launch failed");
    int flag = 0;
    HIPCHK(e, hipMemcpyAsync(&flag, e->inf_flag, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(e, hipStreamSynchronize(st));
    if (flag) return fail(e, FNN_E_INF, "Encountered inf in predicted array.");
    return 0;
}

int64_t fnn_feature_channels(const fnn_engine *e) { return e ? e->layers[e->head_src].cout_pad : -1; }

int fnn_patch_features(fnn_engine *e, int fold, const float *vol, const int64_t shape[4], const fnn_opts *opts,
                       const int64_t *patch_ids, int64_t n_ids, void *feat, float *fss, int64_t slot0, int64_t n_slots) {
    if (int rc = check_ready(e, fold, opts)) return rc;
    if (!vol || !feat || !fss || (n_ids > 0 && !patch_ids)) return fail(e, FNN_E_INVALID, "NULL argument");
    if (!is_device_ptr(feat) || !is_device_ptr(fss)) return fail(e, FNN_E_INVALID, "feature buffers must be device memory");
    if (slot0 < 0 || n_slots < slot0 + n_ids) return fail(e, FNN_E_INVALID, "slots [%lld, %lld) do not fit %lld slots", (long long)slot0, (long long)(slot0 + n_ids), (long long)n_slots);
    if (1 + (int)mirror_combos(*opts).size() > 8) return fail(e, FNN_E_UNSUPPORTED, "more than 8 evaluations per patch");
    if (!e->layers[e->head_src].has_norm) return fail(e, FNN_E_UNSUPPORTED, "the network's last layer has no InstanceNorm");
    HIPCHK(e, hipSetDevice(e->device));
    hipStream_t st = (hipStream_t)opts->stream;
    VolPlan vp;
    if (plan_volume(e->arch, shape + 1, opts->tile_step_size, vp) != 0) return fail(e, FNN_E_INVALID, "invalid volume shape / step size");
    std::vector<int64_t> ids(patch_ids, patch_ids + n_ids);
    for (int64_t id : ids) if (id < 0 || id >= vp.n_patches) return fail(e, FNN_E_INVALID, "patch id out of range");
    if (ids.empty()) return 0;
    const float *vol_dev = nullptr;
    if (int rc = stage_volume(e, vol, shape, vp, st, &vol_dev)) return rc;
    if (int rc = upload_origins(e, vp, ids, st)) return rc;
    Box box;
    for (int d = 0; d < 3; ++d) { box.lo[d] = 0; box.hi[d] = vp.padded[d]; }
    e->ev_used = 0; e->klog.clear();
    if (int rc = run_patches(e, fold, vol_dev, vp, *opts, ids, e->origins, box, nullptr, 0, st, false, true, slot0, n_slots, feat, fss)) return rc;
    if (e->profiling) { HIPCHK(e, hipStreamSynchronize(st)); collect_profile(e, n_ids); }
    return 0;
}

int fnn_gather_box(fnn_engine *e, int fold, const void *feat, const float *fss, const int32_t *slot_of_patch,
                   int64_t n_slots, const int64_t shape[4], const fnn_opts *opts, const int64_t out_lo[3], const int64_t out_hi[3],
                   void *out_logits, void *labels) {
    if (int rc = check_ready(e, fold, opts)) return rc;
    if (!feat || !fss || !slot_of_patch || !out_lo || !out_hi || (!out_logits && !labels)) return fail(e, FNN_E_INVALID, "NULL argument");
    if (!is_device_ptr(feat) || !is_device_ptr(fss) || (out_logits && !is_device_ptr(out_logits)) || (labels && !is_device_ptr(labels)))
        return fail(e, FNN_E_INVALID, "fnn_gather_box needs device pointers");
    if (opts->out_dtype != FNN_OUT_F16) return fail(e, FNN_E_UNSUPPORTED, "fnn_gather_box: fp16 logits");
    const auto combos = mirror_combos(*opts);
    HIPCHK(e, hipSetDevice(e->device));
    hipStream_t st = (hipStream_t)opts->stream;
    VolPlan vp;
    if (plan_volume(e->arch, shape + 1, opts->tile_step_size, vp) != 0) return fail(e, FNN_E_INVALID, "invalid volume shape / step size");
    for (int d = 0; d < 3; ++d)
        if (out_lo[d] < 0 || out_hi[d] > shape[1 + d] || out_lo[d] >= out_hi[d]) return fail(e, FNN_E_INVALID, "output box out of bounds");
    const fnn_arch_desc &a = e->arch;
    const Layer &H = e->layers[e->head_src];
    GatherParams g{};
    g.heads = a.num_heads; g.C = H.cout_pad; g.PD = a.patch[0]; g.PH = a.patch[1]; g.PW = a.patch[2]; g.n_eval = 1 + (int)combos.size();
    g.n_pass = e->n_gpass;
    g.nx = (int)vp.steps[0].size(); g.ny = (int)vp.steps[1].size(); g.nz = (int)vp.steps[2].size();
    std::vector<int> tab;
    int win_off[3];
    g.windowed = gather_tables(a, vp, &tab, win_off) ? 1 : 0;
    if (labels && e->n_gpass > 1)
        return fail(e, FNN_E_UNSUPPORTED, "fnn_gather_box writes labels for <= 63 classes; with %d take the logits and fnn_argmax_labels", a.num_heads);
    if (!H.has_norm || e->head_ksteps != 1 || !gather_ok(g)) return fail(e, FNN_E_UNSUPPORTED, "this network's head does not fit the gather kernel");
    for (int64_t i = 0; i < vp.n_patches; ++i)
        if (slot_of_patch[i] >= n_slots) return fail(e, FNN_E_INVALID, "slot %d of patch %lld is beyond the %lld slots", slot_of_patch[i], (long long)i, (long long)n_slots);
    // tile starts (+ window bases), then the slot table, in one device buffer
    const size_t n_steps = tab.size();
    for (int64_t i = 0; i < vp.n_patches; ++i) tab.push_back(slot_of_patch[i]);
    if (int rc = upload_ints(e, tab.data(), tab.size(), &e->steps_dev, &e->steps_cap, &e->steps_host, &e->steps_host_cap, &e->steps_ev, st)) return rc;
    HIPCHK(e, hipMemsetAsync(e->inf_flag, 0, sizeof(int), st));
    const FoldWeights &fw = e->folds[fold];
    {   // the kernel reads the InstanceNorm rows in the conv kernels' fp16 staging layout: converted from the caller's fp32 rows
        const size_t items = (size_t)n_slots * g.n_eval;
        if (int rc = ensure(e, &e->featssh, &e->featssh_bytes, items * 2 * H.cout_pad * sizeof(f16))) return rc;
        if (launch_fss_to_ssh(fss, (unsigned short *)e->featssh, (long long)items, H.cout_pad, st) != 0) return fail(e, FNN_E_HIP, "row conversion launch failed");
    }
    g.feat = (const f16 *)feat; g.fss = fss; g.fssh = (const unsigned short *)e->featssh;
    g.n_slots = (int)n_slots; g.ring = 1; g.flipmask[0] = 0;       // evaluation f of slot s: item f * n_slots + s (fnn_patch_features)
    for (size_t ci = 0; ci < combos.size() && ci + 1 < 8; ++ci) {
        int m = 0;
        for (int ax : combos[ci]) m |= 1 << ax;
        g.flipmask[ci + 1] = m;
    }
    g.slope = H.act ? a.slope : 1.f;
    gather_set_tables(g, e->steps_dev, win_off);
    g.slot_tab = e->steps_dev + n_steps;
    g.wpk = fw.wpk + e->head_w_off; g.bias = fw.fparam + e->head_bias_off; g.hblocks = e->hblocks;
    g.pass_wpk = fw.wpk + e->gpass_w_off; g.pass_bias = fw.fparam + e->gpass_bias_off;
    g.gauss = opts->use_gaussian ? e->gauss : e->ones;
    g.lo_x = (int)vp.lo[0]; g.lo_y = (int)vp.lo[1]; g.lo_z = (int)vp.lo[2];
    g.OX = shape[1]; g.OY = shape[2]; g.OZ = shape[3];
    g.x_lo = (int)out_lo[0]; g.x_hi = (int)out_hi[0]; g.y_lo = (int)out_lo[1]; g.y_hi = (int)out_hi[1];
    g.z_lo = (int)out_lo[2]; g.z_hi = (int)out_hi[2];
    g.acc_mode = opts->accum; g.out_fp32 = 0; g.mode = 0; g.inf_flag = e->inf_flag;
    g.label_u16 = e->label_u16; g.order = e->label_mode == FNN_LABELS_REGIONS ? e->label_order : nullptr;
    if (labels && !e->label_u16 && e->label_mode == FNN_LABELS_ARGMAX && a.num_heads > 256)
        return fail(e, FNN_E_INVALID, "%d classes do not fit uint8 labels", a.num_heads);
    e->ev_used = 0; e->klog.clear();
    if (out_logits) {
        g.out = out_logits; g.labels = nullptr;
        g.out_vec = shape[3] % 8 == 0 && ((size_t)out_logits % 16) == 0;
        Scope sc(e, st, FAM_HEAD, 0);
        if (launch_gather(g, st) != 0) return fail(e, FNN_E_HIP, "gather launch failed");
    }
    if (labels) {
        g.out = nullptr; g.labels = labels; g.out_vec = 0;
        Scope sc(e, st, FAM_HEAD, 0);
        if (launch_gather(g, st) != 0) return fail(e, FNN_E_HIP, "gather launch failed");
    }
    int flag = 0;
    HIPCHK(e, hipMemcpyAsync(&flag, e->inf_flag, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(e, hipStreamSynchronize(st));
    if (flag) return fail(e, FNN_E_INF, "Encountered inf in predicted array.");
    return 0;
}

static int region_copy(fnn_engine *e, void *feat, int64_t n_slots, const fnn_region *regions, int64_t n, void *message,
                       void *stream, bool pack) {
    if (!e) return FNN_E_INVALID;
    if (n == 0) return 0;
    if (!feat || !regions || !message || n < 0 || n_slots < 1) return fail(e, FNN_E_INVALID, "bad argument");
    if (!is_device_ptr(feat) || !is_device_ptr(regions) || !is_device_ptr(message))
        return fail(e, FNN_E_INVALID, "fnn_pack_regions / fnn_unpack_regions need device pointers (the region table too)");
    if (n > 65535) return fail(e, FNN_E_INVALID, "more than 65535 regions in one message");
    static_assert(sizeof(fnn_region) == 40, "fnn_region is ten 32-bit words");
    HIPCHK(e, hipSetDevice(e->device));
    const fnn_arch_desc &a = e->arch;
    if (launch_region_copy(feat, n_slots, (const int *)regions, (int)n, message, a.patch[0], a.patch[1], a.patch[2],
                           e->layers[e->head_src].cout_pad, pack, (hipStream_t)stream) != 0)
        return fail(e, FNN_E_HIP, "region copy launch failed");
    return 0;
}

int fnn_pack_regions(fnn_engine *e, const void *feat, int64_t n_slots, const fnn_region *regions, int64_t n, void *message, void *stream) {
    return region_copy(e, const_cast<void *>(feat), n_slots, regions, n, message, stream, true);
}

int fnn_unpack_regions(fnn_engine *e, void *feat, int64_t n_slots, const fnn_region *regions, int64_t n, const void *message, void *stream) {
    return region_copy(e, feat, n_slots, regions, n, const_cast<void *>(message), stream, false);
}

int fnn_forward_patches(fnn_engine *e, int fold, const float *x, int n, float *logits, void *stream) {
    if (!e) return FNN_E_INVALID;
    if (fold < 0 || fold >= (int)e->folds.size() || !e->folds[fold].loaded) return fail(e, FNN_E_STATE, "weights of fold %d are not loaded", fold);
    if (!x || !logits || n < 1) return fail(e, FNN_E_INVALID, "bad argument");
    HIPCHK(e, hipSetDevice(e->device));
    hipStream_t st = (hipStream_t)stream;
    const fnn_arch_desc &a = e->arch;
    const size_t P = (size_t)a.patch[0] * a.patch[1] * a.patch[2];
    const size_t nin = (size_t)n * a.in_channels * P, nout = (size_t)n * a.num_heads * P;
    const float *xd = x;
    if (!is_device_ptr(x)) {
        void *t = e->vol_tmp;
        if (int rc = ensure(e, &t, &e->vol_tmp_bytes, nin * 4)) return rc;
        e->vol_tmp = (float *)t;
        HIPCHK(e, hipMemcpyAsync(e->vol_tmp, x, nin * 4, hipMemcpyHostToDevice, st));
        xd = e->vol_tmp;
    }
    float *od = logits;
    const bool out_dev = is_device_ptr(logits);
    if (!out_dev) {
        if (int rc = ensure(e, &e->out_tmp, &e->out_tmp_bytes, nout * 4)) return rc;
        od = (float *)e->out_tmp;
    }
    std::vector<int> zeros((size_t)e->max_batch * 3, 0);
    void *t = e->origins;
    if (int rc = ensure(e, &t, &e->origins_cap, zeros.size() * sizeof(int))) return rc;
    e->origins = (int *)t;
    HIPCHK(e, hipMemcpyAsync(e->origins, zeros.data(), zeros.size() * sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHK(e, hipStreamSynchronize(st));
    const long long vdim[3] = {a.patch[0], a.patch[1], a.patch[2]};
    const int flip[3] = {0, 0, 0};
    e->ev_used = 0; e->klog.clear();
    for (int p0 = 0; p0 < n; p0 += e->max_batch) {
        const int nb = (n - p0 < e->max_batch) ? n - p0 : e->max_batch;
        if (int rc = forward_batch(e, fold, xd + (size_t)p0 * a.in_channels * P, (long long)(a.in_channels * P), vdim,
                                   e->origins, nb, flip, st)) return rc;
        for (int b = 0; b < nb; ++b) {
            HeadParams h = make_head(e, fold, b);
            h.mode = 1; h.patch_buf = od + (size_t)(p0 + b) * a.num_heads * P;
            h.acc_fp32 = 1;
            Scope sc(e, st, FAM_HEAD, e->head_flops);
            if (launch_head(h, st) != 0) return fail(e, FNN_E_HIP, "seg head launch failed");
        }
    }
    if (!out_dev) HIPCHK(e, hipMemcpyAsync(logits, od, nout * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(e, hipStreamSynchronize(st));
    if (e->profiling) collect_profile(e, n);
    return 0;
}

int fnn_set_label_rule(fnn_engine *e, int mode, const int32_t *regions_class_order, int n_regions, int label_dtype) {
    if (!e) return FNN_E_INVALID;
    if ((mode != FNN_LABELS_ARGMAX && mode != FNN_LABELS_REGIONS) || (label_dtype != FNN_LABEL_U8 && label_dtype != FNN_LABEL_U16))
        return fail(e, FNN_E_INVALID, "unknown label mode / dtype");
    if (mode == FNN_LABELS_REGIONS) {
        if (!regions_class_order || n_regions != e->arch.num_heads)
            return fail(e, FNN_E_INVALID, "regions_class_order needs one entry per segmentation head (%d), got %d", e->arch.num_heads, n_regions);
        const int limit = label_dtype == FNN_LABEL_U16 ? 65535 : 255;
        for (int i = 0; i < n_regions; ++i)
            if (regions_class_order[i] < 0 || regions_class_order[i] > limit)
                return fail(e, FNN_E_INVALID, "regions_class_order[%d] = %d does not fit the label dtype", i, regions_class_order[i]);
        HIPCHK(e, hipSetDevice(e->device));
        if (!e->label_order) HIPCHK(e, hipMalloc((void **)&e->label_order, sizeof(int) * (size_t)e->arch.num_heads));
        HIPCHK(e, hipMemcpy(e->label_order, regions_class_order, sizeof(int) * (size_t)n_regions, hipMemcpyHostToDevice));
    }
    e->label_mode = mode;
    e->label_u16 = label_dtype == FNN_LABEL_U16;
    return FNN_OK;
}

int fnn_argmax_labels(fnn_engine *e, const void *logits, int dtype, int heads, int64_t n_vox, void *labels, void *stream) {
    if (!e) return FNN_E_INVALID;
    if (!logits || !labels || heads < 1) return fail(e, FNN_E_INVALID, "bad argument");
    const bool regions = e->label_mode == FNN_LABELS_REGIONS;
    if (regions && heads != e->arch.num_heads) return fail(e, FNN_E_INVALID, "the region rule was set for %d heads, got %d", e->arch.num_heads, heads);
    if (!regions && !e->label_u16 && heads > 256) return fail(e, FNN_E_INVALID, "%d classes do not fit uint8 labels", heads);
    if (!is_device_ptr(logits) || !is_device_ptr(labels)) return fail(e, FNN_E_INVALID, "fnn_argmax_labels needs device pointers");
    HIPCHK(e, hipSetDevice(e->device));
    if (launch_argmax(logits, dtype == FNN_OUT_F32, heads, n_vox, labels, e->label_u16, regions ? e->label_order : nullptr,
                      (hipStream_t)stream) != 0)
        return fail(e, FNN_E_HIP, "argmax launch failed");
    return 0;
}

int fnn_compute_steps(int64_t image_size, int64_t patch_size, double step, int64_t *steps, int cap) {
    std::vector<int64_t> s;
    if (steps_1d(image_size, patch_size, step, s) != 0) return fail(nullptr, FNN_E_INVALID, "image size must be as large or larger than patch_size, 0 < step <= 1");
    if ((int)s.size() > cap) return fail(nullptr, FNN_E_INVALID, "steps buffer too small (%zu needed)", s.size());
    for (size_t i = 0; i < s.size(); ++i) steps[i] = s[i];
    return (int)s.size();
}

int fnn_plan_volume(const int32_t patch[3], const int64_t shape_sp[3], double step, int64_t padded[3], int64_t pad_lo[3],
                    int64_t *n_patches, int32_t *origins, int64_t origins_cap) {
    if (!patch || !shape_sp) return fail(nullptr, FNN_E_INVALID, "NULL argument");
    VolPlan vp;
    const bool two_d = patch[0] == 0;
    const int32_t pp[3] = {two_d ? 1 : patch[0], patch[1], patch[2]};
    if (plan_volume_p(pp, shape_sp, step, two_d, vp) != 0) return fail(nullptr, FNN_E_INVALID, "invalid volume shape / patch / step size");
    for (int d = 0; d < 3; ++d) { if (padded) padded[d] = vp.padded[d]; if (pad_lo) pad_lo[d] = vp.lo[d]; }
    if (n_patches) *n_patches = vp.n_patches;
    if (origins) {
        if (origins_cap < vp.n_patches) return fail(nullptr, FNN_E_INVALID, "origins buffer too small");
        for (size_t i = 0; i < vp.origins.size(); ++i) origins[i] = vp.origins[i];
    }
    return 0;
}

int fnn_fp8_e4m3_encode(const float *in, int64_t n, uint8_t *out) {
    if (!in || !out || n < 0) return fail(nullptr, FNN_E_INVALID, "NULL argument");
    for (int64_t i = 0; i < n; ++i) out[i] = f2e4m3(in[i]);
    return 0;
}

int fnn_set_profiling(fnn_engine *e, int enabled) { if (!e) return FNN_E_INVALID; e->profiling = enabled != 0; return 0; }

int fnn_get_profile(const fnn_engine *e, fnn_profile *out) { if (!e || !out) return FNN_E_INVALID; *out = e->prof; return 0; }

int64_t fnn_kernel_log(const fnn_engine *e, char *buf, int64_t cap) {
    if (!e) return FNN_E_INVALID;
    std::string all;
    for (const std::string &k : e->klog) { all += k; all += '\n'; }
    if (buf && cap > 0) {
        const size_t n = std::min((size_t)cap - 1, all.size());
        memcpy(buf, all.data(), n);
        buf[n] = 0;
    }
    return (int64_t)all.size() + 1;
}

int64_t fnn_profile_launches(const fnn_engine *e, char *buf, int64_t cap) {
    if (!e) return FNN_E_INVALID;
    std::string all;
    for (const std::string &k : e->launch_rows) { all += k; all += '\n'; }
    if (buf && cap > 0) {
        const size_t n = std::min((size_t)cap - 1, all.size());
        memcpy(buf, all.data(), n);
        buf[n] = 0;
    }
    return (int64_t)all.size() + 1;
}

int64_t fnn_layer_table(const fnn_engine *e, char *buf, int64_t cap) {
    if (!e) return FNN_E_INVALID;
    static const char *const ty[] = {"stem", "conv", "tconv", "pool", "combine", "input"};
    std::string all;
    for (size_t li = 0; li < e->layers.size(); ++li) {
        const Layer &L = e->layers[li];
        char row[256];
        snprintf(row, sizeof row, "%zu\t%s\t%d\t%d\t%dx%dx%d\t%dx%dx%d\t%dx%dx%d\t%dx%dx%d\t%.6g\t%.6g\t%d\n", li, ty[L.type],
                 L.cin_real[0] + (L.n_src > 1 ? L.cin_real[1] : 0), L.cout_real, L.k[0], L.k[1], L.k[2], L.s[0], L.s[1], L.s[2],
                 L.in_dims[0], L.in_dims[1], L.in_dims[2], L.out_dims[0], L.out_dims[1], L.out_dims[2], L.flops, L.bytes,
                 L.virtual_out ? 1 : (L.fuse ? 2 : 0));
        all += row;
    }
    if (buf && cap > 0) {
        const size_t n = std::min((size_t)cap - 1, all.size());
        memcpy(buf, all.data(), n);
        buf[n] = 0;
    }
    return (int64_t)all.size() + 1;
}

int fnn_patch_work(const fnn_engine *e, double *flops, double *act_bytes) {
    if (!e) return FNN_E_INVALID;
    if (flops) *flops = e->patch_flops;
    if (act_bytes) *act_bytes = e->patch_act_bytes;
    return 0;
}

}  // extern "C"
