// Shared device-side definitions for the gfx950 (MI355X) kernels.
//
// Data layout in HBM:
//   activations   fp16, NDHWC ("channels-last"), channel count padded to a
//                 multiple of 16 so that one MFMA k-group (8 halves = 16 B) is
//                 one aligned vector load;
//   raw conv out  stored BEFORE InstanceNorm; the per-(n, channel) sum and sum
//                 of squares are accumulated by the producing kernel's epilogue
//                 into `stats` (double, FNN_STAT_REPL replicas to spread the
//                 atomics), and the CONSUMER applies
//                 gamma*(x-mean)*rstd+beta and LeakyReLU while it stages its
//                 input tile;
//   weights       fp16, pre-packed on the host into MFMA fragment order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define FNN_STAT_REPL 8          // replicas of every stats row (one per blockIdx & 7)
#define FNN_TILE_D 4             // conv output tile: 4 x 8 x 8 voxels per 256-thread workgroup
#define FNN_TILE_H 8
#define FNN_TILE_W 8

// One input tensor of a conv / transposed conv / seg head, with the fused
// "normalise + LeakyReLU on load" description.
struct SrcDesc {
    const f16 *ptr;              // [N][D][H][W][C]
    int C;                       // padded channel count (multiple of 16)
    const float *ss;             // [N][2][C]: scale row then shift row of the producer's InstanceNorm, written
                                 // by stats_finalize_kernel; nullptr = identity
    float slope;                 // LeakyReLU slope applied after the affine (1.0 = none)
    // Element (voxel v, channel c) of a batch item sits at v * vs + (c >> 4) * cs + (c & 15): channels-last is vs = C,
    // cs = 16 (also what vs = 0 means); chunk-major [C / 16][voxels][16] is vs = 16, cs = 16 * voxels - a consumer's
    // 16-channel chunk is then contiguous per voxel run instead of 32 bytes of every 2 C-byte record
    int vs;
    long long cs;
    // The same scale / shift rows rounded to fp16 and packed the way a staging thread wants them (round 3): per batch
    // item and group of 8 channels 16 halves - 8 scales, then 8 shifts - i.e. [N][C / 8][16]: one 32-byte load instead of
    // 16 scalars + select + convert per chunk.  Written by stats_finalize_kernel next to `ss`; nullptr with ss = nullptr
    // (identity: conv3d_identity_ssh()).
    const unsigned short *ssh;
};
#define FNN_VS(S) ((S).vs ? (S).vs : (S).C)
#define FNN_CS(S) ((S).vs ? (S).cs : 16LL)
#define FNN_OVS(P) ((P).out_vs ? (P).out_vs : (P).Cout)
#define FNN_OCS(P) ((P).out_vs ? (P).out_cs : 16LL)

struct ConvParams {
    SrcDesc src[2];
    int n_src;
    int N, Di, Hi, Wi;           // input spatial size (both sources)
    int plan_N;                  // batch size the engine was planned for (kernel-variant choice); 0 = N
    int Do, Ho, Wo;              // output spatial size
    int Cout;                    // padded
    int kd, kh, kw, sd, sh, sw, pd, ph, pw;
    const f16 *wpk;              // [cout_blk][chunk][kstep][64 lanes][8]
    const float *bias;           // [Cout]
    f16 *out;                    // [N][Do][Ho][Wo][Cout], or chunk-major (out_vs, out_cs as SrcDesc::vs, cs; 0 = channels-last)
    int out_vs;
    long long out_cs;
    double *stats_out;           // [N][REPL][Cout][2] or nullptr
    int tiles_d, tiles_h, tiles_w;
    int tile_d;                  // output tile depth: 4, or 8 for the pipelined kernel with 8 column blocks per wave
    unsigned long long *dbg;     // diagnostic builds only (-DFNN_STAMPS): per-workgroup s_memtime stamps
    int tmode;                   // diagnostic builds only (-DFNN_TMODE): timing-only switches of the ZR kernel (wrong results)
    int chunks;                  // 16-channel chunks over all sources
    int ksteps;                  // MFMA k-steps per chunk: conv3d_ksteps(packing, taps)
    const float *ident_ss;       // conv3d_identity_ss(): ones[512] then zeros[512] (set by the launchers that need it)
    const unsigned short *ident_ssh;   // conv3d_identity_ssh(): the same as SrcDesc::ssh rows for 512 channels
    int packing;                 // FNN_PACK_*: which taps share a k-step (fixed per layer when the weights are packed)
    int fp8;                     // conv3d_zr8_kernel: e4m3 operands (weights packed at 8 B per lane)
    const float *oscale;         // fp8: [Cout] w_scale[cout] / act_mult, applied to the accumulators before the bias
    float act_mult;              // fp8: activations are quantised as e4m3(value * act_mult)
    int stats_slots;             // rows per batch item in stats_out (conv3d_stats_slots): FNN_STAT_REPL replicas filled by
                                 // atomics, or one row per tile written with plain stores (ZR kernel)
};

// Weight packings of a conv layer.  One k-step (K = 32) always holds 2 taps x 16 input channels;
//   FNN_PACK_LINEAR : k-step ks = taps (2 ks, 2 ks + 1) in d-major linear order, ceil(T / 2) k-steps;
//   FNN_PACK_ZR     : 3x3x3 only, 15 k-steps: ks = 3 * pr + dz holds the in-plane taps (2 pr, 2 pr + 1) of depth
//                     offset dz (in-plane tap 9 = zero padding) - the order conv3d_zr_kernel walks.
#define FNN_PACK_LINEAR 0
#define FNN_PACK_ZR 1
//   FNN_PACK_ZP     : (1, 3, 3) stride 1, conv2d_zp.hip: chunks of 32 input channels (never across the two sources), 9 k-steps
//                     per chunk, k-step dx * 3 + dy = the 32 channels of tap (dy, dx); ConvParams::chunks counts THESE chunks.
#define FNN_PACK_ZP 2
int conv_zp_chunks(int cin_pad0, int cin_pad1);
void conv_zp_pack(const float *W, int cout_real, int cout_pad, int cin_real0, int cin_pad0, int cin_real1, int cin_pad1, unsigned short *dst);
struct ConvParams;
bool conv2d_zp_ok(const ConvParams &p);
int conv2d_zp_stats_slots(const ConvParams &p);
int launch_conv2d_zp(const ConvParams &p, hipStream_t st);

struct StemParams {
    const float *vol;            // [C][X][Y][Z] fp32 (the padded volume)
    long long vol_batch_stride;  // elements between the volumes of consecutive batch items (0 = one volume)
    int C;
    long long X, Y, Z;
    const int *origins;          // [N][3] patch origin in the volume
    int flip_d, flip_h, flip_w;  // test-time mirroring of the network input
    int PD, PH, PW;              // patch = conv input = conv output size (stride 1)
    int kd, kh, kw;
    int Cout;                    // padded
    const float *w;              // [C][taps][Cout] fp32
    const float *bias;           // [Cout]
    f16 *out;                    // [N][PD][PH][PW][Cout]
    double *stats_out;
    int tiles_d, tiles_h, tiles_w;
};

// Thin full-resolution conv with a fused producer (conv3d_thin.hip)
#define FUSE_STEM 1
#define FUSE_TCONV 2
struct ThinParams {
    ConvParams c;                // the consumer conv; FUSE_TCONV: src[1] = the skip, src[0] is never read
    int fuse;                    // FUSE_*
    const f16 *fw;               // producer weights as MFMA "A" fragments: stem [64][8]; tconv [tap][64][8]
    const float *fbias;          // producer bias [16]
    // FUSE_STEM
    const float *vol;            // [1][X][Y][Z] fp32 (the padded volume)
    long long vol_batch_stride;
    long long Y, Z;
    const int *origins;          // [N][3]
    int flip_d, flip_h, flip_w;
    const float *fss;            // [N][2][16] scale / shift of the stem's InstanceNorm
    float fslope;
    // FUSE_TCONV
    SrcDesc low;                 // input of the transposed conv with its on-load transform
    int Dl, Hl, Wl;
    int tsd, tsh, tsw;           // its strides (= kernel)
};

struct TconvParams {
    SrcDesc src;
    int N, Di, Hi, Wi;
    int sd, sh, sw;
    int Cout;                    // padded
    const f16 *wpk;              // [tap][cout_blk][kstep][64][8]
    const float *bias;
    f16 *out;                    // [N][Di*sd][Hi*sh][Wi*sw][Cout], or chunk-major (out_vs, out_cs; 0 = channels-last)
    int out_vs;
    long long out_cs;
    int ksteps;                  // ceil(Cin / 32)
    int nblk;                    // Cout / 16
    int row_store;               // set by launch_tconv: pairs of w-phase taps stored as contiguous rows
    int lds_w;                   // set by launch_tconv: >= 4 k-steps - the weight fragments of a k-step once per workgroup through LDS
};

struct HeadParams {
    SrcDesc src;                 // [N][PD][PH][PW][C]
    int b;                       // batch item handled by this launch
    int PD, PH, PW;
    int heads;
    int hblocks;                 // ceil(heads / 16)
    int ksteps;                  // ceil(C / 32)
    const f16 *wpk;              // [hblock][kstep][64][8]
    const float *bias;           // [hblocks*16]
    const f16 *gauss;            // [PD][PH][PW] or nullptr (= weight 1)
    void *acc;                   // [AX][Y][Z][HP] fp16 or fp32, channel `heads` = weight sum
    long long AX, Y, Z;
    int HP;                      // round_up(heads + 1, 8)
    int ox, oy, oz;              // patch origin relative to the accumulator box
    int flip_d, flip_h, flip_w;
    int mode;                    // 0 fused accumulate, 1 patch buffer '=', 2 patch buffer '+='
    float *patch_buf;            // [heads][PD*PH*PW] fp32 (modes 1, 2)
    int acc_fp32;
    // first-visit thresholds: a voxel (d, h, w) of this patch has not been touched by an earlier patch of the volume
    // when d >= fx && h >= fy && w >= fz (fx = overlap with the previous patch position along x, ...).  Such voxels
    // are written as 0 + contribution without reading the accumulator, which then needs no zero fill.
    // INT_MAX = every voxel is read (accumulators that were zeroed or hold other patches' sums).
    int fx, fy, fz;
};

struct PatchAccParams {          // patch buffer -> volume accumulators (mirroring path)
    const float *patch_buf;      // [heads][P]
    int n_div;                   // number of mirrored evaluations (the reference divides)
    int PD, PH, PW, heads;
    const f16 *gauss;
    void *acc;
    long long AX, Y, Z;
    int HP;
    int ox, oy, oz;
    int acc_fp32;
};

// Head + accumulate + normalise from kept patch features (gather.hip)
struct GatherParams {
    const f16 *feat;             // [n_eval][n_slots][PD][PH][PW][C]: raw output of the network's last conv per (mirrored
                                 // evaluation, patch slot); patch (ix, iy, iz) sits in slot ((ix % ring) * ny + iy) * nz + iz
    const float *fss;            // [n_eval][n_slots][2][C]: scale row, shift row of its InstanceNorm
    const unsigned short *fssh;  // the same rows as fp16 in the staging layout of SrcDesc::ssh: [n_eval][n_slots][C / 8][16] (8 scales, 8 shifts)
    int n_eval;                  // 1, or 1 + the number of mirror-axis subsets (test-time mirroring)
    int flipmask[8];             // per evaluation: bit 0 / 1 / 2 = the network input was flipped along d / h / w
    int n_slots, ring;           // ring = x layers of patches kept (nx: the whole volume)
    int x_lo, x_hi;              // un-padded x range this launch writes
    int y_lo, y_hi, z_lo, z_hi;  // and y / z ranges (z_lo is where the 64-voxel runs start)
    const int *slot_tab;         // patch id -> slot (or -1: not held), instead of the ring rule; nullptr = ring rule
    int C;                       // padded channels (16 or 32)
    float slope;
    const int *steps;            // device: tile starts per axis, x then y then z (ascending)
    int nx, ny, nz;
    // an axis with more than 64 tile positions (a 2-D configuration's slices, a small patch in a long volume): per padded
    // coordinate c the index of the first tile that reaches it (start + extent > c); the tiles that can cover c - or meet
    // the 64-voxel run that starts at c, along z - are the 64 from there (gather_tile_windows checks that on the host).
    // nullptr: the axis has <= 64 positions, window base 0
    const int *base_x, *base_y, *base_z;
    int windowed;                // host check passed: axes with more than 64 positions come with their base tables
    int PD, PH, PW;
    const f16 *wpk;              // seg head [hblock][64][8] (one k-step)
    const float *bias;           // [hblocks * 16], bias[heads] = 1 (the weight-sum channel)
    int heads, hblocks;
    // more than 63 classes (heads + the weight-sum row > 4 blocks of 16): passes of <= 63 heads, each with its own packed
    // fragments [4][64][8] and biases [64] (row `heads of the pass` = the weight-sum channel) - launch_gather runs them one
    // after the other over the same kept activations, every pass writing its own planes of `out` (logits only)
    int n_pass;                  // 0 / 1: the single pack above
    const f16 *pass_wpk;         // [n_pass][4][64][8]
    const float *pass_bias;      // [n_pass][64]
    const f16 *gauss;            // [PD][PH][PW] (all ones without Gaussian weighting)
    int lo_x, lo_y, lo_z;        // un-padded voxel (0, 0, 0) in the padded volume
    long long OX, OY, OZ;        // un-padded size = output size
    int acc_mode, out_fp32, out_vec; // acc_mode = FNN_ACC_* (include/fnn.h)
    int mode;                    // 0 write, 1 add to the existing output (fold ensembling)
    void *out;                   // [heads][OX][OY][OZ] fp16 / fp32, or
    void *labels;                // [OX][OY][OZ] uint8 / uint16 (then `out` is unused)
    int label_u16;
    const int *order;            // regions_class_order or nullptr (argmax)
    int *inf_flag;
    int ieee_div;                    // A-B aid (FNN_GATHER_IEEE): IEEE division per value in the epilogue
};

// tile starts of the three axes + (for an axis with more than 64 positions) its window-base table, ready for upload:
// tab = [steps x | steps y | steps z | base tables ...]; off[d] = offset of axis d's base table in tab or -1.
// false: some coordinate is covered by more than 64 tiles of one axis (the accumulate path serves that)
bool gather_tile_windows(const long long *const steps[3], const int n[3], const int extent[3], const long long padded[3],
                         int *tab, int off[3], size_t *count);
struct FinalizeParams {
    const void *acc;             // [AX][Y][Z][HP]
    long long AX, Y, Z;
    int HP;
    int lo_x, lo_y, lo_z;        // first voxel of the box inside the accumulator
    long long OX, OY, OZ;        // box size
    long long out_X, out_Y, out_Z;   // full output size
    long long out_x, out_y, out_z;   // where the box goes in the output
    int heads;
    int acc_fp32;
    int out_fp32;
    int mode;                    // 0 write, 1 add to existing output (fold ensembling)
    void *out;                   // [heads][out_X][out_Y][out_Z]
    int *inf_flag;
};

#ifdef FNN_STAMPS
#define FNN_STAMP_DECL unsigned long long _st[12]; int _si = 0;
#define FNN_STAMP() do { if (_si < 12) { __builtin_amdgcn_sched_barrier(0); _st[_si++] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); } } while (0)
#define FNN_STAMP_FLUSH(dbg) do { if ((dbg) && threadIdx.x == 0) { for (int _i = 0; _i < 12; ++_i) (dbg)[(size_t)blockIdx.x * 12 + _i] = _i < _si ? _st[_i] : 0ull; } } while (0)
#else
#define FNN_STAMP_DECL
#define FNN_STAMP() do {} while (0)
#define FNN_STAMP_FLUSH(dbg) do {} while (0)
#endif

// Tuning / A-B switches (FNN_NO_GATHER, FNN_PIPES, FNN_ZR_TD, ...; each is documented where it is read): environment
// variables that are honoured ONLY when FNN_KNOBS=1 is set as well - a production process does not change behaviour
// because of a stray variable; the tests and tools/ set it.
const char *fnn_knob(const char *name);
// Which kernel variant a launcher picked: recorded per launch while the engine profiles (fnn_kernel_log), a no-op otherwise.
void fnn_note_kernel(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
void fnn_klog_target(void *vector_of_strings);          // where this thread's notes go (nullptr: nowhere)

static __device__ __forceinline__ float leaky(float x, float slope) { return x > 0.f ? x : x * slope; }

// Two MFMA results of the same 16 voxels or of two column blocks (lane = voxel r, channels 4 q .. 4 q + 3 of each) -> one
// 16-byte store per lane:
// v_permlane16_swap exchanges the odd 16-lane rows of block b with the even rows of block b + 1, after which lane
// (r, q) holds channels 8 (q >> 1) .. + 7 of voxel r of block b + (q & 1) - half the store instructions of the
// 8-byte form.  (Measured neutral on the benchmark: these kernels are not bound by store issue.)
typedef int fnn_u32x4r __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ fnn_u32x4r pair_to_b128(const f16x4 &a, const f16x4 &b) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 ua = __builtin_bit_cast(u32x2, a), ub = __builtin_bit_cast(u32x2, b);
    const auto lo = __builtin_amdgcn_permlane16_swap(ua[0], ub[0], false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(ua[1], ub[1], false, false);
    return (fnn_u32x4r){(int)lo[0], (int)hi[0], (int)lo[1], (int)hi[1]};
}
                                      // (batch item, plane, strip start, row group)

// x * scale + shift of a fragment of 8 channels, the engine's normalise-on-load: scale and shift rounded to fp16, one
// v_pk_fma_f16 per channel pair (make NORM_FP32=1: fp32 fma, then one rounding).  Every kernel that stages or consumes a
// raw conv output goes through this form, so that paths that must agree bit for bit (fused / unfused transposed conv,
// gather / accumulate seg head) do.
static __device__ __forceinline__ f16x8 fnn_norm8(const f16x8 &x, const float (&sc)[8], const float (&sh)[8]) {
#ifdef FNN_NORM_FP32
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (f16)fmaf((float)x[j], sc[j], sh[j]);
    return o;
#else
    f16x8 sc_h, sh_h;
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc_h[j] = (f16)sc[j]; sh_h[j] = (f16)sh[j]; }
    return x * sc_h + sh_h;
#endif
}

// The network input of a batch of patches as an fp16 tensor: patch windows of the fp32 volume (mirroring applied), the
// channel count padded to a multiple of 16 with zeros - what the stem conv of a multi-channel (or `2d`) configuration reads
// through the regular MFMA conv kernels (misc.hip, patch_input_kernel).
struct PatchInputParams {
    const float *vol;            // [C][X][Y][Z] fp32 (the padded volume)
    long long vol_batch_stride;  // elements between the volumes of consecutive batch items (0 = one volume)
    int C, Cpad;
    long long X, Y, Z;
    const int *origins;          // [N][3] patch origin in the volume
    int flip_d, flip_h, flip_w;  // test-time mirroring of the network input
    int PD, PH, PW, N;
    f16 *out;                    // [N][PD][PH][PW][Cpad], or chunk-major (out_vs, out_cs; 0 = channels-last)
    int out_vs;
    long long out_cs;
};
int launch_patch_input(const PatchInputParams &p, hipStream_t st);

// Residual-encoder blocks (BasicBlockD): skip-path average pooling and the block's closing
//   y = LeakyReLU( norm2(conv2) + skip )
struct PoolParams {
    SrcDesc src;                 // [N][Di][Hi][Wi][C] with its on-load transform
    int N, Di, Hi, Wi;
    int sd, sh, sw;              // AvgPool3d(kernel = stride)
    f16 *out;                    // [N][Di/sd][Hi/sh][Wi/sw][C], final values; or chunk-major (out_vs, out_cs; 0 = channels-last)
    int out_vs;
    long long out_cs;
};

struct CombineParams {
    SrcDesc a;                   // conv2 raw output (+ its InstanceNorm, no activation)
    SrcDesc b;                   // skip: projection conv raw output (+ norm) or an already final tensor
    long long vox;               // voxels per batch item
    int N;
    float slope;                 // LeakyReLU slope of the block output
    f16 *out;                    // [N][vox][C], final values; or chunk-major (out_vs, out_cs; 0 = channels-last)
    int out_vs;
    long long out_cs;
    // Round 5: the block output that closes a stage is pooled by the next stage's skip path (AvgPool3d(stride) of exactly
    // these values): combine_pool_kernel writes that tensor in the same pass instead of avgpool_kernel re-reading the one
    // just written.  pool_out != nullptr: D, H, W = the block output's size (each a multiple of its stride), psd / psh / psw
    // the pooling strides, pool_vs / pool_cs the pooled tensor's layout (0 = channels-last)
    f16 *pool_out;
    int D, H, W, psd, psh, psw;
    int pool_vs;
    long long pool_cs;
};

struct StatsFinalizeParams {
    const double *stats;         // [N][nrep][C][2]
    int nrep;                    // rows per batch item (ConvParams::stats_slots of the producer)
    const float *gamma, *beta;   // [C]
    float *ss;                   // [N][2][C]
    unsigned short *ssh;         // [N][C / 8][16] halves (8 scales, 8 shifts), or nullptr (SrcDesc::ssh)
    int C;
    float inv_count, eps;
};

// launchers (implemented in the .hip files)
int launch_stats_finalize(const StatsFinalizeParams &p, int N, hipStream_t st);
int launch_region_copy(void *feat, long long n_slots, const int *regions, int n, void *message, int PD, int PH, int PW, int C, bool pack, hipStream_t st);   // fnn_pack_regions / fnn_unpack_regions
int launch_fss_to_ssh(const float *fss, unsigned short *ssh, long long items, int C, hipStream_t st);   // [items][2][C] fp32 rows -> [items][C / 8][16] fp16 (SrcDesc::ssh)
int launch_avgpool(const PoolParams &p, hipStream_t st);
int launch_combine(const CombineParams &p, hipStream_t st);
bool combine_pool_ok(int D, int H, int W, int sd, int sh, int sw);     // can the combine launch write the pooled tensor of these strides too
int launch_conv3d(const ConvParams &p, hipStream_t st);
size_t conv3d_lds_bytes(const ConvParams &p, int nb);
int conv3d_pick_nb(int nblk);
const float *conv3d_identity_ss();
// packing the launcher will expect for a layer of this shape (decided from the PLANNED batch size)
int conv3d_packing(const ConvParams &p);
const unsigned short *conv3d_identity_ssh();
int conv3d_ksteps(int packing, int taps);
int conv3d_kstep_tap(int packing, int ks, int half, int taps);     // linear tap index, or -1 = zero padding
int conv3d_pack_cout(int packing, int nblk, int cb, int m);        // output channel in row m of cout block cb of the packed weights
int launch_conv3d_zr(const ConvParams &p, hipStream_t st);
bool conv3d_zq12_ok(const ConvParams &p);                               // conv3d_zq.hip: 3x3x3 stride 1 over planes of 9 .. 12 voxels per axis
int launch_conv3d_zq12(ConvParams p, hipStream_t st);                  // -1 = not this kernel's layer
bool conv3d_s2_ok(const ConvParams &p);                                  // conv3d_s2.hip: 3x3x3 stride (2,2,2), Cout % 64 == 0
int launch_conv3d_s2(ConvParams p, hipStream_t st);                     // -1 = not this kernel's layer
int conv3d_stats_slots(const ConvParams &p);                           // rows per item the layer's kernel writes into stats_out
int launch_tconv(const TconvParams &p, hipStream_t st);
bool stem_mfma_ok(int C, int kd, int kh, int kw, int cout_pad);
int stem_mfma_stats_slots(int PD, int PH, int PW);
int stem_mfma_ksteps(int C, int taps);
bool stem_mfma_kmap(int C, int taps, int ks, int k, int *c, int *tap);   // (channel, tap) of element k of k-step ks; false = padding
int launch_stem_mfma(const StemParams &p, const f16 *wfrag, int N, hipStream_t st);    // p.out == nullptr: statistics only
bool gather_ok(const GatherParams &p);
int launch_gather(const GatherParams &p, hipStream_t st);
int launch_quotient_check(unsigned long long *counts, hipStream_t st);   // counts[0] = differing pairs, [1] = pairs on the fast route, [2] = an example
bool conv_thin_ok(const ThinParams &tp);
int launch_conv_thin(const ThinParams &tp, hipStream_t st);
bool conv_row_ok(const ThinParams &tp);                                 // conv3d_row.hip: tp.fuse = 0 or FUSE_TCONV
int launch_conv_row(const ThinParams &tp, hipStream_t st);             // -1 = not this kernel's layer
bool stem_row_ok(const StemParams &p);
int launch_stem_row(const StemParams &p, int N, hipStream_t st);       // -1 = not this kernel's stem
int launch_head(const HeadParams &p, hipStream_t st);
bool launch_head_first_visit_ok(const HeadParams &p);   // does launch_head() honour fx / fy / fz for these parameters?
int launch_patch_acc(const PatchAccParams &p, hipStream_t st);
int launch_finalize(const FinalizeParams &p, hipStream_t st);
int launch_labels_from_acc(const FinalizeParams &p, void *labels, int label_u16, const int *order, hipStream_t st);
int launch_scale_output(void *out, int out_fp32, long long n, int divisor, int *inf_flag, hipStream_t st);
int launch_argmax(const void *logits, int fp32, int heads, long long nvox, void *labels, int label_u16, const int *order,
                  hipStream_t st);
int launch_pad_volume(const float *src, float *dst, int C, const long long s[3], const long long d[3],
                      const long long lo[3], hipStream_t st);
