// gather.hip - seg head + Gaussian-weighted accumulation + normalisation WITHOUT volume accumulators (gfx950).
//
// The reference accumulates patch by patch into whole-volume buffers (predict_from_raw_data.py:602-621):
//     pred *= gaussian;  predicted_logits[sl] += pred;  n_predictions[sl] += gaussian;  ...  predicted_logits /= n
// Done literally on the GPU that is a read-modify-write of every accumulator line per patch visit: for 61 classes
// 2 x 128 B per voxel and visit next to 32 B of network output - 425 MB per patch, 18 % of the benchmark's time, plus a
// 34 GB normalisation pass.  But a voxel's value depends only on the <= 8 patches that cover it, in visiting order:
//     acc = 0;  for p in covering patches (ascending = the reference's x-major order):  acc = fp16(acc + logit_p * g_p)
// So the network's last activation (16 channels, 32 B per voxel) of EVERY patch of the volume is kept in HBM
// (28 GB for 600 patches of 160 x 96 x 96 - the 288 GB make that free) and one pass over the volume does the rest:
// a wave owns 64 consecutive z voxels; for every covering patch it loads their 32-byte feature vectors, applies the
// producer's InstanceNorm + LeakyReLU, runs the 1x1x1 seg head on the matrix cores, multiplies by the patch's Gaussian
// weight and adds into accumulators that live in REGISTERS, rounding to fp16 after every visit exactly like the
// reference's half-precision buffers (or keeping fp32: FNN_ACC_FP32); then divides by the weight sum, checks for inf
// and writes the un-padded fp16 logits (or the label map) - every byte of the result is written once, every feature
// byte read once: 44 GB instead of ~290 GB per 512^3 volume.  Results are bit-identical to the accumulate path.
// Test-time mirroring (:541-557): the activations of all 2^k mirrored evaluations are kept and their logits summed in
// fp32 per patch visit (read at the flipped voxel), divided by 2^k, then weighted - again the reference's order.
// Memory: patches are produced x layer by x layer; a layer is needed until the output slab behind it is written, so a
// RING of layers bounds the footprint (the whole volume when it fits: one launch at the end).
//
// Replaces _internal_predict_sliding_window_return_logits' accumulation and normalisation (:602-625) and, for the
// label entry points, LabelManager.convert_logits_to_segmentation (label_handling.py:144-195).
#include "fnn_device.h"
#include <cstdlib>

namespace {

static __device__ __forceinline__ float acc_add_product_1(float a, float t, float g) {
#pragma clang fp contract(off)
    const float c = t * g;                                     // the reference rounds the product before the add: no fma
    return a + c;
}

// FNN_ACC_FP16_AUTOCAST: the product of two fp16 numbers rounded to fp16, then an fp16 + fp16 add rounded to fp16 -
// torch's half arithmetic (computed in fp32, rounded once: the fp32 product of two halves is exact, and an fp32 sum of
// two halves can only be inexact when the smaller one is below a quarter ulp of the result, where both roundings agree).
// No contraction: v_pk_fma_f16 would skip the product's rounding.
static __device__ __forceinline__ f16x2 acc_add_product_h2(f16x2 a, f16x2 t, f16x2 g) {
#pragma clang fp contract(off)
    const f16x2 c = t * g;
    return a + c;
}
static __device__ __forceinline__ f16x2 add_h2(f16x2 a, f16x2 b) {
#pragma clang fp contract(off)
    return a + b;
}
static __device__ __forceinline__ f16x2 round_h2(float a, float b) { f16x2 r; r[0] = (f16)a; r[1] = (f16)b; return r; }

// The closing division (predict_from_raw_data.py:619: predicted_logits /= n_predictions, half tensors: torch divides in
// fp32 and rounds the quotient to fp16).  IEEE division costs 11 instructions and a quarter-rate v_rcp_f32 per value,
// and the 64 values of a lane share their divisor: the reciprocal is taken ONCE per 16 voxels (v_rcp_f32 + one Newton
// step = the correctly rounded reciprocal) and a value costs q0 = a y and two rounds of r = a - b q, q += r y
// (Markstein's correction; ONE round is a unit of fp32 off often enough to flip the fp16 rounding of 3900 pairs where
// the quotient sits on an fp16 tie).  For fp16-valued a and b > 0 rounding q to fp16 gives the bits of the IEEE route - checked over ALL
// 2^16 x 2^15 pairs by quotient_check_kernel below (tests/test_gpu_ops.py) - apart from the sign of a zero, copied
// from a afterwards, and non-finite a or b = 0 / inf / NaN, which leave q NaN: any q that is not |q| < 65520 (fp16
// infinity's rounding boundary) sends the group down the IEEE route, which then also raises the reference's inf flag.
static __device__ __forceinline__ float quot_rcp(float b) {
#pragma clang fp contract(off)
    const float y0 = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, y0, 1.f);
    return __builtin_fmaf(e, y0, y0);
}
static __device__ __forceinline__ float quot_fast(float a, float b, float y) {
#pragma clang fp contract(off)
    const float q0 = a * y;
    const float r0 = __builtin_fmaf(-b, q0, a);
    const float q1 = __builtin_fmaf(r0, y, q0);
    const float r1 = __builtin_fmaf(-b, q1, a);
    return __builtin_fmaf(r1, y, q1);
}
static __device__ __forceinline__ bool quot_odd(float q) { return !(__builtin_fabsf(q) < 65520.f); }
static __device__ __forceinline__ f16x2 quot_sign(f16x2 q, f16x2 a) {          // magnitude of q, sign of a
    const unsigned r = (__builtin_bit_cast(unsigned, q) & 0x7FFF7FFFu) | (__builtin_bit_cast(unsigned, a) & 0x80008000u);
    return __builtin_bit_cast(f16x2, r);
}

struct Pick {                                                  // LabelPick of misc.hip over this lane's heads, mergeable
    float best; int arg; int nan; int hit;
};

}  // namespace

// fp16-valued running sum (one half of a packed pair) + fp32 product, rounded to fp32 exactly like v_add_f32:
// v_fma_mix_f32 reads the half in place (a * 1.0 + c, the product is exact), so the up-convert and the add are ONE
// instruction (tools/hw_probe.cpp compares the bits with (float)a + c for every fp16 a, subnormals and NaNs included).
// The rounding to fp16 stays a separate v_cvt_pk_f16_f32: the reference rounds twice (fp32 sum, then the fp16 store).
static __device__ __forceinline__ float add_half_lo(unsigned a2, float c) {
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(a2), "v"(c));
    return r;
}
static __device__ __forceinline__ float add_half_hi(unsigned a2, float c) {
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(a2), "v"(c));
    return r;
}
static __device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}

// What one visit (a patch that covers part of the wave's 64-voxel run, one mirrored evaluation of it) brings in: the raw
// feature vectors of the four 16-voxel groups, the evaluation's InstanceNorm rows, the patch's Gaussian weights
// K16: a network whose last layer has 16 (padded) channels runs the head as v_mfma_f32_16x16x16_f16 - a lane holds 4
// channels of its voxel (8-byte loads, every lane live, half the normalisation work and 20 registers less) instead of 8
// channels of a K = 32 operand whose upper half is zero.  The results are the K = 32 form's bit for bit
// (tools/hw_probe.cpp Q2: 0 of 16.8 M values differ), so the seg-head kernels of the accumulate path still agree.
typedef int gather_i32x2 __attribute__((ext_vector_type(2)));
template <bool K16> struct GatherK { typedef fnn_u32x4r XV; typedef f16x8 FV; static constexpr int N = 8; };
template <> struct GatherK<true> { typedef gather_i32x2 XV; typedef f16x4 FV; static constexpr int N = 4; };
static __device__ __forceinline__ f32x4 head_mfma(const f16x8 &a, const f16x8 &b, const f32x4 &c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
static __device__ __forceinline__ f32x4 head_mfma(const f16x4 &a, const f16x4 &b, const f32x4 &c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }

template <int G, bool K16>
struct GatherVisit {
    typename GatherK<K16>::XV x[G];
#ifdef FNN_NORM_FP32
    float sc32[GatherK<K16>::N], sh32[GatherK<K16>::N];
#else
    typename GatherK<K16>::XV sc, sh;                           // the scales and shifts of this lane's channels (fp16: SrcDesc::ssh rows)
#endif
    unsigned short gall;                                        // the patch's Gaussian weight at z voxel `lane` of the run
};
struct GatherGeo { int slot, dx, dy, oz; };                     // wave-uniform: where the patch keeps its activation, the run's place inside it

// HB = head blocks of 16 (heads + the weight-sum channel <= 16 HB); LABELS: write the label map instead of the logits.
// ACCM = the accumulate arithmetic (include/fnn.h): 0 FNN_ACC_FP16_REFERENCE - fp32 logits, fp32 product and sum, one
// rounding to fp16 per visit (the reference without autocast: its CPU path); 1 FNN_ACC_FP32; 2 FNN_ACC_FP16_AUTOCAST -
// the reference on a GPU (predict_from_raw_data.py:591-593: the network's output is fp16): logit, mirror sums, product
// and sum each rounded to fp16 - packed fp16 arithmetic, half the instructions of mode 0.
//
// Round 4: the visit loop.  A wave used to find its covering patches by walking the three tile-start tables with one
// scalar load + wait per entry (~66 dependent round trips per wave) and hipcc sank the first group's feature load into
// the branch that uses it (a second exposed round trip per visit).  Now (1) lane i of the wave holds tile start i of each
// axis (three vector loads, one round trip), the covering patches are three ballots, a patch's start a v_readlane - no
// memory access per visit; (2) a visit's loads are buffer loads whose addresses are a lane constant + a scalar (out-of-
// patch lanes fall outside the slot's num_records and read zeros: no selects), issued together and pinned in front of
// the arithmetic; (3) the InstanceNorm rows arrive as the fp16 rows the staging threads of the conv kernels use
// (SrcDesc::ssh: two 16-byte loads instead of four + 8 converts + 10 shuffles); (4) the fp16 -> fp32 up-convert of a
// running sum and its add are one v_fma_mix_f32.
// Kept from rounds 2 / 3: 64-voxel runs, sums as fp16 pairs at four waves per SIMD, untouched 16-voxel groups skipped,
// the shared-reciprocal quotient.
// Measured on the 512^3 x 61 benchmark volume (tools/gather_ab.py, bits equal in every arm): round 3's kernel 18.3 ms;
// tables + pinned buffer loads + v_fma_mix 17.1; + the next visit prefetched 16.8 (K = 32, three waves per SIMD);
// + the K = 16 head 14.7 (four waves per SIMD WITH the prefetch: 119 registers; without it, five waves: 17.2);
// autocast arithmetic 17.3 -> 13.5.  Dropped: a head's four 128-byte row pieces of a workgroup stored as one 512-byte
// segment behind a workgroup barrier (14.73 vs 14.75 ms: the scattered stores are not what it waits for).
template <int HB, int ACCM, bool LABELS, bool TTA, bool K16>
__global__ __launch_bounds__(256, (!TTA && ACCM != 1) ? (K16 ? 4 : 3) : (TTA && K16 ? 3 : 1)) void gather_head_kernel(const GatherParams p) {
    constexpr bool PF = !TTA && ACCM != 1;                     // the next visit's loads in flight during this visit's arithmetic
    typedef typename GatherK<K16>::XV XV;
    typedef typename GatherK<K16>::FV FV;
    constexpr int KN = GatherK<K16>::N;                        // channels per lane
    constexpr int G = 4, ZW = 16 * G, TP = ZW + 8;             // 16-voxel groups and z voxels per wave; row pitch of the LDS transpose
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, q = lane >> 4;   // (a scalar: so are the wave's x, y, z0 and what a visit derives from them)
    f16 *sT = (f16 *)smem + wave * (HB * 16 * TP);             // per wave: [HB * 16 heads][ZW z (+8 pad)] fp16

    // wave -> (x, y, run of 64 z) of the UN-PADDED output
    const int ny_box = p.y_hi - p.y_lo, nz_box = p.z_hi - p.z_lo;
    const long long zruns = (nz_box + ZW - 1) / ZW;
    long long wid = (long long)blockIdx.x * 4 + wave;
    const long long total = (long long)(p.x_hi - p.x_lo) * ny_box * zruns;
    if (wid >= total) return;
    const int zr = (int)(wid % zruns); wid /= zruns;
    const int y = p.y_lo + (int)(wid % ny_box);
    const int x = p.x_lo + (int)(wid / ny_box);
    const int z0 = p.z_lo + zr * ZW;
    const int xp = x + p.lo_x, yp = y + p.lo_y, zp0 = z0 + p.lo_z;            // padded-volume coordinates

    // seg head fragments: A operand per head block, bias of this lane's 4 heads per block
    FV wf[HB];
    f32x4 bv[HB];
#pragma unroll
    for (int hb = 0; hb < HB; ++hb) {
        const int hbc = hb < p.hblocks ? hb : p.hblocks - 1;     // HB = 4 with 3 blocks: the copy's rows are >= heads, ignored
        // the packed K = 32 fragments: lane (r, q') holds k = 8 q' .. 8 q' + 7 of head r; a K = 16 lane (r, q) wants k = 4 q .. 4 q + 3
        if (K16) wf[hb] = *(const FV *)(p.wpk + ((size_t)hbc * 64 + r + 16 * (q >> 1)) * 8 + 4 * (q & 1));
        else wf[hb] = *(const FV *)(p.wpk + ((size_t)hbc * 64 + lane) * 8);
        bv[hb] = *(const f32x4 *)(p.bias + hbc * 16 + q * 4);
    }
    constexpr bool ACC32 = ACCM == 1, ACH = ACCM == 2, PKS = ACCM != 1;   // PKS: the sums are fp16 values - kept as fp16 pairs (half the registers)
    f32x4 acc[G][HB];                                          // [16-voxel group][head block] x 4 heads: fp16-valued unless ACC32
    f16x2 ah[PKS ? G : 1][HB][2];                              // PKS: the sums as fp16 pairs (heads 4q + {0,1}, {2,3})
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int hb = 0; hb < HB; ++hb) {
            acc[g][hb] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (PKS) { ah[PKS ? g : 0][hb][0] = (f16x2){0, 0}; ah[PKS ? g : 0][hb][1] = (f16x2){0, 0}; }
        }

    const int P = p.PD * p.PH * p.PW;
    const bool live = K16 || q * 8 < p.C;                      // (K = 32 on 16 channels: the upper lanes' operand is zero)
    const int c0 = K16 ? q * 4 : (live ? q * 8 : 0);           // this lane's first channel
    const f16 slope_h = (f16)p.slope;
    const int *sx = p.steps, *sy = p.steps + p.nx, *sz = p.steps + p.nx + p.ny;

    // ---- lane constants of a visit's loads (byte offsets; a visit adds scalars to them)
    const int zl = zp0 + r;                                                 // this lane's z in group 0 (padded coordinates)
    const unsigned c2 = (unsigned)p.C * 2;                                  // bytes per voxel record
    const unsigned fl = (unsigned)zl * c2 + (unsigned)c0 * 2;               // features: z ascending ...
    const unsigned fln = (unsigned)c0 * 2 - (unsigned)zl * c2;              // ... and of an evaluation flipped along w (unsigned wrap)
    const unsigned gl = (unsigned)(zp0 + lane) * 2;                         // Gaussian weights: lane l takes z voxel l of the run
    const unsigned slot_bytes = (unsigned)P * c2;                           // < 2^31 (gather_ok)
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void *)p.gauss, 0, P * 2, 0x00020000);
#ifndef FNN_NORM_FP32
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)p.fssh, 0, p.n_eval * p.n_slots * (int)c2 * 2, 0x00020000);
#endif

    // one evaluation's loads: the four groups' feature vectors + its InstanceNorm rows (+ the patch's Gaussian weights:
    // lane l takes the weight of z voxel l of the run, a group's 16 come back through ds_bpermute).  Lanes outside the
    // patch along z read the neighbouring row of the slot (valid memory, masked later) or fall outside num_records (zeros).
    // (`g0`: the run's first group that goes into v.x[0] - the mirrored path takes the run two groups at a time)
    const auto issue = [&](auto &v, const GatherGeo &e, int f, int fm, int g0 = 0) {
        constexpr int GV = (int)(sizeof(v.x) / sizeof(v.x[0]));
        const int fx = (fm & 1) ? p.PD - 1 - e.dx : e.dx, fy = (fm & 2) ? p.PH - 1 - e.dy : e.dy;   // output voxel of an evaluation whose input was flipped
        const size_t ev = (size_t)f * p.n_slots + e.slot;
        const int rowbase = (fx * p.PH + fy) * p.PW;
        const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc((void *)(p.feat + ev * P * p.C), 0, slot_bytes, 0x00020000);
        // A 16-voxel group of the run that the patch does not reach is not fetched (round 5): its lanes' offset becomes one the
        // buffer's range check refuses - zeros, no memory traffic.  A patch spans 96 z voxels and meets two or three 64-voxel
        // runs, every visit used to fetch all four groups: FETCH_SIZE (x2: tools/fetch_calib.cpp finds the counter at half
        // the bytes for 4 / 8 / 16-byte loads alike) had the kernel read 46.6 GB for 28.3 GB of kept activations, 1.64x.
        const int zlo = zp0 - e.oz;                            // z of the run's first voxel inside the patch (wave-uniform)
        if (TTA && (fm & 4)) {
            const unsigned b = fln + (unsigned)(rowbase + p.PW - 1 + e.oz) * c2;
#pragma unroll
            for (int gv = 0; gv < GV; ++gv) {
                const int g = g0 + gv;
#ifdef FNN_GATHER_NOREACH
                const bool reach = true;
#else
                const bool reach = zlo + 16 * g + 15 >= 0 && zlo + 16 * g < p.PW;
#endif
                const unsigned vo = reach ? b - (unsigned)g * 16u * c2 : 0x80000000u;
                if constexpr (K16) v.x[gv] = __builtin_bit_cast(XV, __builtin_amdgcn_raw_buffer_load_b64(rf, vo, 0, 0));
                else v.x[gv] = __builtin_bit_cast(XV, __builtin_amdgcn_raw_buffer_load_b128(rf, vo, 0, 0));
            }
        } else {
            const unsigned b = fl + (unsigned)(rowbase - e.oz) * c2;
#pragma unroll
            for (int gv = 0; gv < GV; ++gv) {
                const int g = g0 + gv;
#ifdef FNN_GATHER_NOREACH
                const bool reach = true;
#else
                const bool reach = zlo + 16 * g + 15 >= 0 && zlo + 16 * g < p.PW;
#endif
                const unsigned vo = reach ? b + (unsigned)g * 16u * c2 : 0x80000000u;
                if constexpr (K16) v.x[gv] = __builtin_bit_cast(XV, __builtin_amdgcn_raw_buffer_load_b64(rf, vo, 0, 0));
                else v.x[gv] = __builtin_bit_cast(XV, __builtin_amdgcn_raw_buffer_load_b128(rf, vo, 0, 0));
            }
        }
#ifdef FNN_NORM_FP32
        const float *qs = p.fss + ev * 2 * p.C + c0;
#pragma unroll
        for (int j = 0; j < KN; ++j) { v.sc32[j] = qs[j]; v.sh32[j] = qs[p.C + j]; }
#else
        const unsigned so = (unsigned)(c0 >> 3) * 32 + (unsigned)(c0 & 7) * 2;   // rows of [C / 8][8 scales, 8 shifts]
        if constexpr (K16) {
            v.sc = __builtin_bit_cast(XV, __builtin_amdgcn_raw_buffer_load_b64(rs, so, (unsigned)ev * c2 * 2, 0));
            v.sh = __builtin_bit_cast(XV, __builtin_amdgcn_raw_buffer_load_b64(rs, so + 16, (unsigned)ev * c2 * 2, 0));
        } else {
            v.sc = __builtin_bit_cast(XV, __builtin_amdgcn_raw_buffer_load_b128(rs, so, (unsigned)ev * c2 * 2, 0));
            v.sh = __builtin_bit_cast(XV, __builtin_amdgcn_raw_buffer_load_b128(rs, so + 16, (unsigned)ev * c2 * 2, 0));
        }
#endif
        if (f == 0) v.gall = __builtin_amdgcn_raw_buffer_load_b16(rg, gl + (unsigned)((e.dx * p.PH + e.dy) * p.PW - e.oz) * 2u, 0, 0);
    };
    // the loads stay in front of the arithmetic: hipcc sinks a pure load into the branch that uses it (a group's skip
    // test), where its round trip is exposed once per group
    const auto pin = [&](auto &v) {
        if constexpr (sizeof(v.x) / sizeof(v.x[0]) == 2) asm volatile("" : "+v"(v.x[0]), "+v"(v.x[1]), "+v"(v.gall));
        else asm volatile("" : "+v"(v.x[0]), "+v"(v.x[1]), "+v"(v.x[2]), "+v"(v.x[3]), "+v"(v.gall));
    };
    const auto normed = [&](const auto &v, int g) -> FV {
        const FV xr = __builtin_bit_cast(FV, v.x[g]);
#ifdef FNN_NORM_FP32
        FV o;
#pragma unroll
        for (int j = 0; j < KN; ++j) o[j] = (f16)fmaf((float)xr[j], v.sc32[j], v.sh32[j]);   // fnn_norm8's fp32 form
#else
        FV o = xr * __builtin_bit_cast(FV, v.sc) + __builtin_bit_cast(FV, v.sh);   // fnn_norm8's arithmetic on the rows' own fp16 values
#endif
        o = __builtin_elementwise_max(o, o * slope_h);
        if (!K16 && !live) {
#pragma unroll
            for (int j = 0; j < KN; ++j) o[j] = (f16)0.f;
        }
        return o;
    };
    const auto weight_of = [&](const auto &v, int g) -> f16 {     // group g's Gaussian weight of this lane's voxel
        const int w = __builtin_amdgcn_ds_bpermute((16 * g + r) << 2, (int)v.gall);
        return __builtin_bit_cast(f16, (unsigned short)w);
    };

    // ---- the visits: the patches whose box holds (xp, yp) and meets the run [zp0, zp0 + 64) along z, in x-major order
    // (the reference's).  Lane i holds tile start i of each axis (<= 64 per axis: gather_ok); the covering sets are three
    // ballots, a patch's start a v_readlane: no memory access between visits apart from a sharded caller's slot table.
    constexpr int NEVER = 0x3fffffff;                          // a tile start no coordinate reaches
    // (an axis with more than 64 positions: the 64 from the first tile that reaches this coordinate - GatherParams::base_x)
#ifdef FNN_GATHER_NOWIN
    constexpr int bx = 0, by = 0, bz = 0;
#else
    // (scalars: x, y, z0 come from the wave's index, which hipcc does not know to be wave-uniform - as vector values the bases
    // made every visit's tile indices, the ring rule's modulo and the slot vector arithmetic: +2 ms on the benchmark volume)
    const int bx = __builtin_amdgcn_readfirstlane(p.base_x ? p.base_x[__builtin_amdgcn_readfirstlane(xp)] : 0);
    const int by = __builtin_amdgcn_readfirstlane(p.base_y ? p.base_y[__builtin_amdgcn_readfirstlane(yp)] : 0);
    const int bz = __builtin_amdgcn_readfirstlane(p.base_z ? p.base_z[__builtin_amdgcn_readfirstlane(zp0)] : 0);
#endif
    const int tx = bx + lane < p.nx ? sx[bx + lane] : NEVER, ty = by + lane < p.ny ? sy[by + lane] : NEVER, tz = bz + lane < p.nz ? sz[bz + lane] : NEVER;
    const unsigned long long MX = __builtin_amdgcn_ballot_w64(tx <= xp && xp - tx < p.PD);
    const unsigned long long MY = __builtin_amdgcn_ballot_w64(ty <= yp && yp - ty < p.PH);
    const unsigned long long MZ = __builtin_amdgcn_ballot_w64(tz < zp0 + ZW && zp0 - tz < p.PW);
    struct Cur { unsigned long long rx, ry, rz; };
    const auto geo = [&](const Cur &c) -> GatherGeo {
        const int jx = __builtin_ctzll(c.rx), jy = __builtin_ctzll(c.ry), jz = __builtin_ctzll(c.rz);
        GatherGeo e;
        e.dx = xp - __builtin_amdgcn_readlane(tx, jx); e.dy = yp - __builtin_amdgcn_readlane(ty, jy); e.oz = __builtin_amdgcn_readlane(tz, jz);
        const int gx = jx + bx, gy = jy + by, gz = jz + bz;    // tile indices on the whole axes
        const int pid = (gx * p.ny + gy) * p.nz + gz;
        e.slot = p.slot_tab ? p.slot_tab[pid] : ((gx % p.ring) * p.ny + gy) * p.nz + gz;   // -1: not held here (a sharded caller's table)
        return e;
    };
    const auto advance = [&](Cur &c) -> bool {                 // z fastest; false = past the last visit
        c.rz &= c.rz - 1; if (c.rz) return true;
        c.rz = MZ; c.ry &= c.ry - 1; if (c.ry) return true;
        c.ry = MY; c.rx &= c.rx - 1; return c.rx != 0;
    };

    // one visit's arithmetic on the registers `issue` filled (not the mirrored form: below)
    const auto consume = [&](GatherVisit<G, K16> &v, const GatherGeo &e) {
        pin(v);
        const int dz0 = zl - e.oz;                             // this lane's z inside the patch, group 0
#ifndef FNN_GATHER_WEIGHT_PER_GROUP
        // the four groups' Gaussian weights (one ds_bpermute each) requested up front: per group the request sat right in front of
        // its wait, and the wait in front of the group's MFMAs
        f16 ghs[G];
#pragma unroll
        for (int g = 0; g < G; ++g) ghs[g] = weight_of(v, g);
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int g = 0; g < G; ++g) {
            // a patch that starts or ends inside the run leaves whole 16-voxel groups untouched: skip them
            // (wave-uniform; 3.5 patches intersect a 64-voxel run along z, 2 cover each voxel)
            const bool in = (unsigned)(dz0 + 16 * g) < (unsigned)p.PW;
            const unsigned long long inm = __builtin_amdgcn_ballot_w64(in);
            if (inm == 0) continue;
            const FV o = normed(v, g);
#ifndef FNN_GATHER_WEIGHT_PER_GROUP
            const f16 gh = ghs[g];
#else
            const f16 gh = weight_of(v, g);
#endif
            const float gw = (float)gh;
            // channel `heads` has zero weights and bias 1: its product is the weight itself; the product is rounded
            // before the sum (no fma), one rounding to fp16 per visit; lanes outside the patch keep their sums (and
            // signed zeros)
#if !defined(FNN_GATHER_NOMIX) && !defined(FNN_GATHER_SELECT) && !defined(FNN_GATHER_EXEC_PER_BLOCK)
            if constexpr (!ACH && PKS && HB == 4) {
                // all four head blocks of the group under ONE EXEC window (two writes of EXEC per group visit instead of eight):
                // the four MFMAs and the sixteen products first (all lanes), then the sixteen add + up-convert and the eight
                // roundings of the per-block form below, in one asm statement
                float t[4][4];
                unsigned a[4][2];
                f32x4 d[4];
                // the four MFMAs back to back into their own registers (written per block with its products, hipcc puts every
                // block's result into the same four registers: MFMA, eight wait states, four products - four times in a row)
#pragma unroll
                for (int hb = 0; hb < 4; ++hb) d[hb] = head_mfma(wf[hb], o, bv[hb]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int hb = 0; hb < 4; ++hb) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) t[hb][j] = mul_rn(d[hb][j], gw);
                    a[hb][0] = __builtin_bit_cast(unsigned, ah[g][hb][0]);
                    a[hb][1] = __builtin_bit_cast(unsigned, ah[g][hb][1]);
                }
                unsigned long long sv;
#define FNN_MIX4(A, B, T0, T1, T2, T3)                                                             \
                "v_fma_mix_f32 %[" #T0 "], %[" #A "], 1.0, %[" #T0 "] op_sel_hi:[1,0,0]\n\t"                  \
                "v_fma_mix_f32 %[" #T1 "], %[" #A "], 1.0, %[" #T1 "] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"  \
                "v_fma_mix_f32 %[" #T2 "], %[" #B "], 1.0, %[" #T2 "] op_sel_hi:[1,0,0]\n\t"                  \
                "v_fma_mix_f32 %[" #T3 "], %[" #B "], 1.0, %[" #T3 "] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
#define FNN_CVT2(A, B, T0, T1, T2, T3)                                                             \
                "v_cvt_pk_f16_f32 %[" #A "], %[" #T0 "], %[" #T1 "]\n\t"                                      \
                "v_cvt_pk_f16_f32 %[" #B "], %[" #T2 "], %[" #T3 "]\n\t"
                asm volatile("s_mov_b64 %[sv], exec\n\t"
                             "s_and_b64 exec, exec, %[m]\n\t"
                             FNN_MIX4(a00, a01, t00, t01, t02, t03) FNN_MIX4(a10, a11, t10, t11, t12, t13)
                             FNN_MIX4(a20, a21, t20, t21, t22, t23) FNN_MIX4(a30, a31, t30, t31, t32, t33)
                             FNN_CVT2(a00, a01, t00, t01, t02, t03) FNN_CVT2(a10, a11, t10, t11, t12, t13)
                             FNN_CVT2(a20, a21, t20, t21, t22, t23) FNN_CVT2(a30, a31, t30, t31, t32, t33)
                             "s_mov_b64 exec, %[sv]"
                             : [a00] "+v"(a[0][0]), [a01] "+v"(a[0][1]), [a10] "+v"(a[1][0]), [a11] "+v"(a[1][1]),
                               [a20] "+v"(a[2][0]), [a21] "+v"(a[2][1]), [a30] "+v"(a[3][0]), [a31] "+v"(a[3][1]),
                               [t00] "+v"(t[0][0]), [t01] "+v"(t[0][1]), [t02] "+v"(t[0][2]), [t03] "+v"(t[0][3]),
                               [t10] "+v"(t[1][0]), [t11] "+v"(t[1][1]), [t12] "+v"(t[1][2]), [t13] "+v"(t[1][3]),
                               [t20] "+v"(t[2][0]), [t21] "+v"(t[2][1]), [t22] "+v"(t[2][2]), [t23] "+v"(t[2][3]),
                               [t30] "+v"(t[3][0]), [t31] "+v"(t[3][1]), [t32] "+v"(t[3][2]), [t33] "+v"(t[3][3]), [sv] "=&s"(sv)
                             : [m] "s"(inm));
#undef FNN_MIX4
#undef FNN_CVT2
#pragma unroll
                for (int hb = 0; hb < 4; ++hb) {
                    ah[g][hb][0] = __builtin_bit_cast(f16x2, a[hb][0]);
                    ah[g][hb][1] = __builtin_bit_cast(f16x2, a[hb][1]);
                }
                continue;
            }
#endif
#if !defined(FNN_GATHER_SELECT) && !defined(FNN_GATHER_EXEC_PER_BLOCK)
            if constexpr (ACH && HB == 4) {
                // the autocast arithmetic the same way: the rounded logits' products and the sums' adds (v_pk_mul_f16, v_pk_add_f16: the
                // instructions hipcc emits for acc_add_product_h2) of all four head blocks under one EXEC window - no selects
                f32x4 d[4];
#pragma unroll
                for (int hb = 0; hb < 4; ++hb) d[hb] = head_mfma(wf[hb], o, bv[hb]);
                __builtin_amdgcn_sched_barrier(0);
                unsigned t[4][2], a[4][2];
#pragma unroll
                for (int hb = 0; hb < 4; ++hb) {
                    t[hb][0] = __builtin_bit_cast(unsigned, round_h2(d[hb][0], d[hb][1]));      // the network's fp16 output
                    t[hb][1] = __builtin_bit_cast(unsigned, round_h2(d[hb][2], d[hb][3]));
                    a[hb][0] = __builtin_bit_cast(unsigned, ah[g][hb][0]);
                    a[hb][1] = __builtin_bit_cast(unsigned, ah[g][hb][1]);
                }
                const unsigned g2 = __builtin_bit_cast(unsigned, (f16x2){gh, gh});
                unsigned long long sv;
#define FNN_PKMUL2(T0, T1) "v_pk_mul_f16 %[" #T0 "], %[" #T0 "], %[g2]\n\tv_pk_mul_f16 %[" #T1 "], %[" #T1 "], %[g2]\n\t"
#define FNN_PKADD2(A0, A1, T0, T1) "v_pk_add_f16 %[" #A0 "], %[" #A0 "], %[" #T0 "]\n\tv_pk_add_f16 %[" #A1 "], %[" #A1 "], %[" #T1 "]\n\t"
                asm volatile("s_mov_b64 %[sv], exec\n\t"
                             "s_and_b64 exec, exec, %[m]\n\t"
                             FNN_PKMUL2(t00, t01) FNN_PKMUL2(t10, t11) FNN_PKMUL2(t20, t21) FNN_PKMUL2(t30, t31)
                             FNN_PKADD2(a00, a01, t00, t01) FNN_PKADD2(a10, a11, t10, t11)
                             FNN_PKADD2(a20, a21, t20, t21) FNN_PKADD2(a30, a31, t30, t31)
                             "s_mov_b64 exec, %[sv]"
                             : [a00] "+v"(a[0][0]), [a01] "+v"(a[0][1]), [a10] "+v"(a[1][0]), [a11] "+v"(a[1][1]),
                               [a20] "+v"(a[2][0]), [a21] "+v"(a[2][1]), [a30] "+v"(a[3][0]), [a31] "+v"(a[3][1]),
                               [t00] "+v"(t[0][0]), [t01] "+v"(t[0][1]), [t10] "+v"(t[1][0]), [t11] "+v"(t[1][1]),
                               [t20] "+v"(t[2][0]), [t21] "+v"(t[2][1]), [t30] "+v"(t[3][0]), [t31] "+v"(t[3][1]), [sv] "=&s"(sv)
                             : [m] "s"(inm), [g2] "v"(g2));
#undef FNN_PKMUL2
#undef FNN_PKADD2
#pragma unroll
                for (int hb = 0; hb < 4; ++hb) {
                    ah[g][hb][0] = __builtin_bit_cast(f16x2, a[hb][0]);
                    ah[g][hb][1] = __builtin_bit_cast(f16x2, a[hb][1]);
                }
                continue;
            }
#endif
            // (fewer head blocks, fp32 sums: per block; the MFMAs still back to back into their own registers)
            f32x4 dd[HB];
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) dd[hb] = head_mfma(wf[hb], o, bv[hb]);   // logit: the bias is the C operand, as in the seg-head kernels
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                const f32x4 d = dd[hb];
                if (ACH) {
                    const f16x2 t01 = round_h2(d[0], d[1]);      // the network's fp16 output
                    const f16x2 t23 = round_h2(d[2], d[3]);
                    const f16x2 gw2 = {gh, gh};
                    f16x2 (&a2)[2] = ah[PKS ? g : 0][hb];
                    const f16x2 n01 = acc_add_product_h2(a2[0], t01, gw2), n23 = acc_add_product_h2(a2[1], t23, gw2);
                    a2[0] = in ? n01 : a2[0];
                    a2[1] = in ? n23 : a2[1];
                } else if (PKS) {
#if !defined(FNN_GATHER_NOMIX) && !defined(FNN_GATHER_SELECT)
                    // Round 5: the four sums of a head block are updated under an EXEC mask of the lanes inside the patch instead
                    // of through one select per pair (`if (in) a = v` comes back from hipcc as v_cndmask): the up-convert + add
                    // (v_fma_mix_f32, as add_half_lo / _hi) and the rounding to fp16 pairs of all four values in ONE asm
                    // statement, so that nothing else is scheduled between the two writes of EXEC.  The products are formed
                    // outside (all lanes): hipcc places the wait states between the MFMA and its first reader there.
                    {
                        unsigned a01 = __builtin_bit_cast(unsigned, ah[PKS ? g : 0][hb][0]), a23 = __builtin_bit_cast(unsigned, ah[PKS ? g : 0][hb][1]);
                        float t0 = mul_rn(d[0], gw), t1 = mul_rn(d[1], gw), t2 = mul_rn(d[2], gw), t3 = mul_rn(d[3], gw);
                        unsigned long long sv;
                        asm volatile("s_mov_b64 %[sv], exec\n\t"
                                     "s_and_b64 exec, exec, %[m]\n\t"
                                     "v_fma_mix_f32 %[t0], %[a01], 1.0, %[t0] op_sel_hi:[1,0,0]\n\t"
                                     "v_fma_mix_f32 %[t1], %[a01], 1.0, %[t1] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                                     "v_fma_mix_f32 %[t2], %[a23], 1.0, %[t2] op_sel_hi:[1,0,0]\n\t"
                                     "v_fma_mix_f32 %[t3], %[a23], 1.0, %[t3] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                                     "s_nop 0\n\t"
                                     "v_cvt_pk_f16_f32 %[a01], %[t0], %[t1]\n\t"
                                     "v_cvt_pk_f16_f32 %[a23], %[t2], %[t3]\n\t"
                                     "s_mov_b64 exec, %[sv]"
                                     : [a01] "+v"(a01), [a23] "+v"(a23), [t0] "+v"(t0), [t1] "+v"(t1), [t2] "+v"(t2), [t3] "+v"(t3), [sv] "=&s"(sv)
                                     : [m] "s"(inm));
                        ah[PKS ? g : 0][hb][0] = __builtin_bit_cast(f16x2, a01);
                        ah[PKS ? g : 0][hb][1] = __builtin_bit_cast(f16x2, a23);
                    }
#else
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        f16x2 &a2 = ah[PKS ? g : 0][hb][k];
#ifdef FNN_GATHER_NOMIX
                        const f16x2 nv = round_h2(acc_add_product_1((float)a2[0], d[2 * k], gw), acc_add_product_1((float)a2[1], d[2 * k + 1], gw));
#else
                        const unsigned au = __builtin_bit_cast(unsigned, a2);
                        const f16x2 nv = round_h2(add_half_lo(au, mul_rn(d[2 * k], gw)), add_half_hi(au, mul_rn(d[2 * k + 1], gw)));
#endif
                        a2 = in ? nv : a2;
                    }
#endif
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float sv = acc_add_product_1(acc[g][hb][j], d[j], gw);
                        acc[g][hb][j] = in ? sv : acc[g][hb][j];
                    }
                }
            }
        }
    };

    if (MX != 0 && MY != 0 && MZ != 0) {
        Cur c0 = {MX, MY, MZ};
        if (TTA) {
            // predict_from_raw_data.py:541-557: the 2^k evaluations of a patch, their logits summed in the reference's order
            // (fp32, or fp16 under autocast), divided by 2^k, then weighted
            bool more = true;
            while (more) {
                const GatherGeo e = geo(c0);
                more = advance(c0);
                if (e.slot < 0) continue;
                const int dz0 = zl - e.oz;
                bool in[G];
#pragma unroll
                for (int g = 0; g < G; ++g) in[g] = (unsigned)(dz0 + 16 * g) < (unsigned)p.PW;
                // The run two groups at a time (round 5): the evaluations' running sums of two groups are half the registers of all
                // four (HB = 4: 183 -> 3 waves per SIMD), a half no group of which the patch reaches is skipped; within a half the
                // evaluations two at a time (their count is 2, 4 or 8): the loads of evaluation f + 1 leave before the arithmetic
                // of evaluation f (two register sets; unconditional loads - behind the last one the last evaluation is re-read).
                const float nf = (float)p.n_eval;
                // 2^k evaluations (every subset of k mirror axes): x / 2^k = x * 2^-k to the bit (both are the correctly rounded
                // value of the same real number, subnormals included) - one multiply instead of the ~10 instructions of an IEEE
                // division per value (gather_ok refuses any other count)
                const float rnf = 1.0f / nf;
                const auto div_n = [&](float x) -> float { return mul_rn(x, rnf); };
#pragma unroll
                for (int gh = 0; gh < G; gh += 2) {
                    if ((__builtin_amdgcn_ballot_w64(in[gh]) | __builtin_amdgcn_ballot_w64(in[gh + 1])) == 0) continue;
                    f32x4 tsum[2][HB];                         // running fp32 sum of the logits
                    f16x2 tsh[ACH ? 2 : 1][HB][2];             // ... or the running fp16 sum (autocast arithmetic)
                    GatherVisit<2, K16> v, vb;
                    const auto eval_body = [&](GatherVisit<2, K16> &w, int f) {
                        pin(w);
#pragma unroll
                        for (int g2 = 0; g2 < 2; ++g2) {
                            if (__builtin_amdgcn_ballot_w64(in[gh + g2]) == 0) continue;
                            const FV o = normed(w, g2);
                            // (back to back, as in consume(), where the registers are there: the forms with fp16 running sums and
                            // the K = 32 head would lose a wave per SIMD to them)
                            constexpr bool B2B = K16 && !ACH;
                            f32x4 dd[B2B ? HB : 1];
                            if (B2B) {
#pragma unroll
                                for (int hb = 0; hb < HB; ++hb) dd[B2B ? hb : 0] = head_mfma(wf[hb], o, bv[hb]);
                                __builtin_amdgcn_sched_barrier(0);
                            }
#pragma unroll
                            for (int hb = 0; hb < HB; ++hb) {
                                const f32x4 d = B2B ? dd[B2B ? hb : 0] : head_mfma(wf[hb], o, bv[hb]);
                                if (ACH) {
                                    const f16x2 t01 = round_h2(d[0], d[1]), t23 = round_h2(d[2], d[3]);
                                    f16x2 (&ts)[2] = tsh[ACH ? g2 : 0][hb];
                                    ts[0] = f == 0 ? t01 : add_h2(ts[0], t01);
                                    ts[1] = f == 0 ? t23 : add_h2(ts[1], t23);
                                } else {
                                    tsum[g2][hb] = f == 0 ? d : tsum[g2][hb] + d;
                                }
                            }
                        }
                    };
                    issue(v, e, 0, p.flipmask[0], gh);
                    for (int f = 0; f < p.n_eval; f += 2) {
                        issue(vb, e, f + 1, p.flipmask[f + 1], gh);
                        eval_body(v, f);
                        const int f2 = f + 2 < p.n_eval ? f + 2 : p.n_eval - 1;
                        issue(v, e, f2 ? f2 : 1, p.flipmask[f2 ? f2 : 1], gh);    // (never evaluation 0 again: that one also loads the Gaussian weights)
                        eval_body(vb, f + 1);
                    }
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2) {
                        const int g = gh + g2;
                        if (__builtin_amdgcn_ballot_w64(in[g]) == 0) continue;
                        const f16 gh16 = weight_of(v, g);
                        const float gw = (float)gh16;
                        if (ACH) {
                            const f16x2 gw2 = {gh16, gh16};
#pragma unroll
                            for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                                for (int k = 0; k < 2; ++k) {
                                    const f16x2 ts = tsh[ACH ? g2 : 0][hb][k];
                                    const f16x2 t = round_h2(div_n((float)ts[0]), div_n((float)ts[1]));   // half /= int
                                    f16x2 &a2 = ah[PKS ? g : 0][hb][k];
                                    const f16x2 nv = acc_add_product_h2(a2, t, gw2);
                                    a2 = in[g] ? nv : a2;
                                }
                            continue;
                        }
#pragma unroll
                        for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                            for (int k = 0; k < 2; ++k) {
                                const float t0 = div_n(tsum[g2][hb][2 * k]);         // prediction /= (len(axes_combinations) + 1)
                                const float t1 = div_n(tsum[g2][hb][2 * k + 1]);
                                if (PKS) {
                                    f16x2 &a2 = ah[PKS ? g : 0][hb][k];
                                    const f16x2 nv = round_h2(acc_add_product_1((float)a2[0], t0, gw), acc_add_product_1((float)a2[1], t1, gw));
                                    a2 = in[g] ? nv : a2;
                                } else {
                                    const float s0 = acc_add_product_1(acc[g][hb][2 * k], t0, gw), s1 = acc_add_product_1(acc[g][hb][2 * k + 1], t1, gw);
                                    acc[g][hb][2 * k] = in[g] ? s0 : acc[g][hb][2 * k];
                                    acc[g][hb][2 * k + 1] = in[g] ? s1 : acc[g][hb][2 * k + 1];
                                }
                            }
                    }
                }
            }
        } else if (PF) {
            // software pipeline over the visits: the loads of visit v + 1 leave before the arithmetic of visit v (two
            // register sets, the loop unrolled by two so that both are named); a sharded caller's slot of visit v + 2 is
            // requested one step earlier still.  The loads are unconditional - behind the last visit they re-read the
            // current one's lines - because a load in a conditional block makes hipcc's next wait a vmcnt(0).
            GatherVisit<G, K16> va, vb;
            const auto held = [](GatherGeo e) { e.slot = e.slot < 0 ? 0 : e.slot; return e; };   // (a patch a sharded caller does not hold: loaded from slot 0, not consumed)
            GatherGeo e0 = geo(c0);
            Cur c1 = c0;
            bool ok1 = advance(c1);
            GatherGeo e1 = ok1 ? geo(c1) : e0;
            issue(va, held(e0), 0, 0);
            while (true) {
                issue(vb, held(e1), 0, 0);
                Cur c2 = c1;
                const bool ok2 = ok1 && advance(c2);
                const GatherGeo e2 = ok2 ? geo(c2) : e1;
                if (e0.slot >= 0) consume(va, e0);
                if (!ok1) break;
                issue(va, held(e2), 0, 0);
                Cur c3 = c2;
                const bool ok3 = ok2 && advance(c3);
                const GatherGeo e3 = ok3 ? geo(c3) : e2;
                if (e1.slot >= 0) consume(vb, e1);
                if (!ok2) break;
                e0 = e2; e1 = e3; c1 = c3; ok1 = ok3;
            }
        } else {
            GatherVisit<G, K16> v;
            bool more = true;
            while (more) {
                const GatherGeo e = geo(c0);
                more = advance(c0);
                if (e.slot < 0) continue;
                issue(v, e, 0, 0);
                consume(v, e);
            }
        }
    }

    // ---- per 16-voxel group (one at a time: 16 live values instead of 64): normalise - logits = sum / weight sum,
    // rounded to fp16 (:619), the weight sum sits in row `heads` - then the label pick or the LDS transpose
    const int wrow = p.heads, whb = wrow >> 4, wq = (wrow >> 2) & 3, wj = wrow & 3;
    bool bad = false;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const auto sum_of = [&](int hb, int j) -> float { return PKS ? (float)ah[PKS ? g : 0][hb][j >> 1][j & 1] : acc[g][hb][j]; };
        float wsum = 0.f;
#pragma unroll
        for (int hb = 0; hb < HB; ++hb)
#pragma unroll
            for (int j = 0; j < 4; ++j) if (hb == whb && j == wj) wsum = sum_of(hb, j);
        wsum = __shfl(wsum, wq * 16 + r, 64);
        const bool zok = z0 + 16 * g + r < p.z_hi;
        const int lrem = wrow & 15;                            // classes in block whb (the row behind them holds the weight sum)
        if constexpr (LABELS && PKS) {
            // ---- argmax labels without the 64 quotients (round 5).  The logit of a head is Q(a) = fp16(fl32(a / w)) of its sum a
            // (both fp16 values, w > 0): monotone in a.  So the winners are the heads with Q(a) = Q(A), A = the largest sum, and
            // Q(a) >= Q(A)  <=>  fl32(a / w) > m, or = m with the tie going up (Q(A) even), m = the midpoint below Q(A)
            // <=>  a > T or (a = T and Q(A) even), T = m w EXACTLY in fp32 (12 x 11 significant bits) - and fl32(a / w) = m only
            // when a = T: an fp16 a != T is at least 2^-22 |T| away from the 23-bit T, a / w then 2^-22 m from m.  One quotient
            // (the shared-reciprocal form the logits take, checked against IEEE division over all pairs), one threshold, one
            // compare per head; the lowest winning head = torch's first maximum.  Anything unusual - a non-finite sum, w not a
            // positive finite number, a quotient near the fp16 range's ends, regions instead of argmax - takes the chain below.
            if (!p.order && !p.ieee_div) {
                const f16 HINF = __builtin_bit_cast(f16, (unsigned short)0x7c00), HNINF = __builtin_bit_cast(f16, (unsigned short)0xfc00);
                // block whb ends with the weight-sum row: vmk[k] = 0xffff per half of its pair k that is a class
                unsigned vmk[2], xw[2] = {0u, 0u};
#pragma unroll
                for (int k = 0; k < 2; ++k) vmk[k] = (q * 4 + 2 * k < lrem ? 0xffffu : 0u) | (q * 4 + 2 * k + 1 < lrem ? 0xffff0000u : 0u);
                // (v_pk_max / min_f16 through asm: the builtin quiets its operands first - three instructions per maximum)
                const auto pkmax = [](unsigned a2, unsigned b2) { unsigned r2; asm("v_pk_max_f16 %0, %1, %2" : "=v"(r2) : "v"(a2), "v"(b2)); return r2; };
                const auto pkmin = [](unsigned a2, unsigned b2) { unsigned r2; asm("v_pk_min_f16 %0, %1, %2" : "=v"(r2) : "v"(a2), "v"(b2)); return r2; };
                unsigned mxu = 0xfc00fc00u, mnu = 0x7c007c00u;
                f16x2 sm = {(f16)0.f, (f16)0.f};
#pragma unroll
                for (int hb = 0; hb < HB; ++hb) {
                    if (hb > whb) continue;                    // (wave-uniform)
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const unsigned x2 = __builtin_bit_cast(unsigned, ah[PKS ? g : 0][hb][k]);
                        unsigned xa = x2, xi = x2, xs = x2;
                        if (hb == whb) {                       // (wave-uniform)
                            xa = (x2 & vmk[k]) | (0xfc00fc00u & ~vmk[k]);
                            xi = (x2 & vmk[k]) | (0x7c007c00u & ~vmk[k]);
                            xs = x2 & vmk[k];
                            xw[k] = xa;
                        }
                        mxu = pkmax(mxu, xa);
                        mnu = pkmin(mnu, xi);
                        sm = add_h2(sm, __builtin_bit_cast(f16x2, xs));   // NaN / inf among the sums end up here (the packed max ignores a NaN)
                    }
                }
                const f16x2 mx = __builtin_bit_cast(f16x2, mxu), mn = __builtin_bit_cast(f16x2, mnu);
                f16x2 P;                                       // (largest sum, largest negated sum) of the voxel's heads
                P[0] = mx[0] > mx[1] ? mx[0] : mx[1];
                P[1] = mn[0] < mn[1] ? -mn[0] : -mn[1];
                if (!(__builtin_fabsf((float)sm[0]) < 65520.f) || !(__builtin_fabsf((float)sm[1]) < 65520.f)) P[0] = HINF;
#pragma unroll
                for (int m = 16; m < 64; m <<= 1) {
                    const unsigned o2 = (unsigned)__shfl_xor((int)__builtin_bit_cast(unsigned, P), m, 64);
                    P = __builtin_bit_cast(f16x2, pkmax(__builtin_bit_cast(unsigned, P), o2));
                }
                const float A = (float)P[0], Bm = (float)P[1];
                const float yr = quot_rcp(wsum);
                const float qA = quot_fast(A, wsum, yr);
                const f16 qh = (f16)qA;
                const unsigned qb = __builtin_bit_cast(unsigned short, qh), qmag = qb & 0x7fffu;
                bool slow = !(A < 65520.f) || !(wsum > 0.f) || !(wsum < 65520.f) || !(__builtin_fmaxf(__builtin_fabsf(A), Bm) * yr < 32768.f) ||
                            qmag < 0x0400u || qmag >= 0x7bffu;
                const unsigned pb = (qb & 0x8000u) ? qb + 1u : qb - 1u;       // the fp16 value below Q(A)
                const float mid = 0.5f * ((float)__builtin_bit_cast(f16, (unsigned short)pb) + (float)qh);
                const float T = mul_rn(mid, wsum);
                const f16 Tn = (f16)T;
                const float tf = (float)Tn;
                const unsigned tb = __builtin_bit_cast(unsigned short, Tn);
                const bool down = tf > T || (tf == T && !(qb & 1u));         // the largest fp16 value that does NOT win: Tn itself, or the one below it
                const unsigned tb2 = (tb & 0x8000u) ? tb + 1u : (tb == 0u ? 0x8001u : tb - 1u);
                const unsigned tdb = down ? tb2 : tb;
                slow |= (tdb & 0x7fffu) == 0u || (tdb & 0x7fffu) >= 0x7c00u;
                if (__builtin_amdgcn_ballot_w64(slow) == 0) {
                    // this lane's lowest winning head as the index hb * 4 + j of its 16: per pair two compares (the high half
                    // through SDWA) and two selects, the pairs descending so that the lowest index is written last; two wait
                    // states between a compare and the select that reads its mask (gfx950)
                    int win = 16;
#pragma unroll
                    for (int hb = HB - 1; hb >= 0; --hb) {
                        if (hb > whb) continue;
#pragma unroll
                        for (int k = 1; k >= 0; --k) {
                            const unsigned xv = hb == whb ? xw[k] : __builtin_bit_cast(unsigned, ah[PKS ? g : 0][hb][k]);
                            unsigned long long mh;
                            asm("v_cmp_gt_f16_sdwa %1, %2, %3 src0_sel:WORD_1 src1_sel:WORD_0\n\t"
                                "v_cmp_gt_f16_e32 vcc, %2, %3\n\t"
                                "s_nop 0\n\t"
                                "v_cndmask_b32_e64 %0, %0, %4, %1\n\t"
                                "v_cndmask_b32_e64 %0, %0, %5, vcc"
                                : "+v"(win), "=&s"(mh) : "v"(xv), "v"(tdb), "n"(hb * 4 + 2 * k + 1), "n"(hb * 4 + 2 * k) : "vcc");
                        }
                    }
                    int lab = win < 16 ? (win >> 2) * 16 + q * 4 + (win & 3) : 0x7fff;
#pragma unroll
                    for (int m = 16; m < 64; m <<= 1) { const int ol = __shfl_xor(lab, m, 64); lab = ol < lab ? ol : lab; }
                    const int z = z0 + 16 * g + r;
                    if (q == 0 && z < p.z_hi) {
                        const size_t o = ((size_t)x * p.OY + y) * p.OZ + z;
                        if (p.label_u16) ((uint16_t *)p.labels)[o] = (uint16_t)lab; else ((uint8_t *)p.labels)[o] = (uint8_t)lab;
                    }
                    continue;
                }
            }
        }
        bool odd = true;
        f16x2 qh[HB][2];                                       // the group's logits: heads 4q + {0,1}, {2,3} per block
        if (PKS) {                                             // fp16-valued sums: the shared-reciprocal quotient (above)
            const float yr = quot_rcp(wsum);
            unsigned long long om = 0;                         // lanes with an odd quotient: compares into scalar masks
#pragma unroll
            for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float q0 = quot_fast(sum_of(hb, 2 * e), wsum, yr), q1 = quot_fast(sum_of(hb, 2 * e + 1), wsum, yr);
                    om |= __builtin_amdgcn_ballot_w64(quot_odd(q0)) | __builtin_amdgcn_ballot_w64(quot_odd(q1));
                    qh[hb][e] = quot_sign(round_h2(q0, q1), ah[PKS ? g : 0][hb][e]);
                }
            odd = om != 0 || p.ieee_div;                       // wave-uniform; never taken on finite data
        }
        if (odd) {
#pragma unroll
            for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f16 rr = (f16)__fdiv_rn(sum_of(hb, j), wsum);
                    const int head = hb * 16 + q * 4 + j;
                    bad |= zok && head < p.heads && isinf((float)rr);
                    qh[hb][j >> 1][j & 1] = rr;
                }
        }
        if (LABELS) {
            // LabelPick (misc.hip): argmax with torch's rules - the first NaN wins, else the largest value, the lowest
            // head among equals - is a maximum under a total order, so the lane picks over ITS 16 heads (ascending: a
            // strict compare keeps the first) and the four lanes of a voxel merge once, comparing head indices on ties:
            // 4 cross-lane moves per 16 voxels and lane instead of 32.  Measured and dropped: maximum by max3 + two
            // cross-lane steps, then the lowest head that equals it, with this chain kept for groups that hold a NaN -
            // 7x fewer instructions on the usual path, 35 instead of 29 ms.
            // Regions: the highest head above the threshold - a plain maximum of indices.
            float b = 0.f; int a = -1; bool n = false; int h = -1;
#pragma unroll
            for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int head = hb * 16 + q * 4 + j;
                    const float v = (float)qh[hb][j >> 1][j & 1];
                    if (head < p.heads) {
                        if (v > 0x1.8p-24f) h = head;
                        if (a < 0) { b = v; a = head; n = v != v; }
                        else if (!n && (v > b || v != v)) { b = v; a = head; n = v != v; }
                    }
                }
#pragma unroll
            for (int m = 16; m < 64; m <<= 1) {
                const float ob = __shfl_xor(b, m, 64);
                const int oa = __shfl_xor(a, m, 64);
                const bool on = ob != ob;
                bool take;                                     // is the other lane's pick the better one?
                if (a < 0) take = true;
                else if (oa < 0) take = false;
                else if (n || on) take = on && (!n || oa < a);
                else take = ob > b || (ob == b && oa < a);
                if (take) { b = ob; a = oa; n = on; }
                if (p.order) { const int oh = __shfl_xor(h, m, 64); h = h > oh ? h : oh; }
            }
            const int z = z0 + 16 * g + r;
            if (q == 0 && z < p.z_hi) {
                const int lab = p.order ? (h >= 0 ? p.order[h] : 0) : a;
                const size_t o = ((size_t)x * p.OY + y) * p.OZ + z;
                if (p.label_u16) ((uint16_t *)p.labels)[o] = (uint16_t)lab; else ((uint8_t *)p.labels)[o] = (uint8_t)lab;
            }
        } else {
            // logits: transpose through LDS so that a head's 64 z values leave as one 128-byte row
#pragma unroll
            for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                for (int j = 0; j < 4; ++j) sT[(hb * 16 + q * 4 + j) * TP + 16 * g + r] = qh[hb][j >> 1][j & 1];
        }
    }
    if (bad) atomicOr(p.inf_flag, 1);
    if (LABELS) return;

    const size_t plane = (size_t)p.OX * p.OY * p.OZ;
    const size_t rowoff = ((size_t)x * p.OY + y) * p.OZ + z0;
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    const int nz = (int)(p.z_hi - z0 < ZW ? p.z_hi - z0 : ZW);
    const bool vec = p.out_vec && nz == ZW && (p.z_lo & 7) == 0;   // 16-byte aligned rows
    if (!p.out_fp32 && vec) {
        constexpr int PCS = ZW / 8, HPP = 64 / PCS;             // 16-byte pieces per row, heads per pass
        for (int h8 = 0; h8 < p.heads; h8 += HPP) {
            const int head = h8 + lane / PCS, piece = lane % PCS;
            if (head < p.heads) {
                f16x8 v = *(const f16x8 *)(sT + head * TP + piece * 8);
                f16 *o = (f16 *)p.out + (size_t)head * plane + rowoff + piece * 8;
                if (p.mode) {
                    const f16x8 old = *(const f16x8 *)o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (f16)((float)old[e] + (float)v[e]);
                }
                *(f16x8 *)o = v;
            }
        }
    } else {
        for (int head = 0; head < p.heads; ++head) {
            if (lane < nz) {
                const f16 v = sT[head * TP + lane];
                const size_t o = (size_t)head * plane + rowoff + lane;
                if (p.out_fp32) {
                    float *op = (float *)p.out + o;
                    *op = p.mode ? *op + (float)v : (float)v;
                } else {
                    f16 *op = (f16 *)p.out + o;
                    *op = p.mode ? (f16)((float)*op + (float)v) : v;
                }
            }
        }
    }
}

// Exhaustive check of the epilogue's quotient: every fp16 bit pattern a against every b with a clear sign bit (the
// weight sum is a sum of non-negative products).  Counts the pairs where the shared-reciprocal route, taken the way
// the kernel takes it (quot_odd -> IEEE), and the IEEE route differ in the fp16 bits of the result.
__global__ __launch_bounds__(256) void quotient_check_kernel(unsigned long long *n_diff, unsigned long long *n_fast) {
    const unsigned id = blockIdx.x * 256u + threadIdx.x;
    const unsigned short ab = (unsigned short)(id & 0xFFFFu), bb = (unsigned short)(id >> 16);
    const f16 ah = __builtin_bit_cast(f16, ab), bh = __builtin_bit_cast(f16, bb);
    const float a = (float)ah, b = (float)bh;
    const f16 ref = (f16)__fdiv_rn(a, b);
    const float q = quot_fast(a, b, quot_rcp(b));
    const bool odd = quot_odd(q);
    const f16x2 fast2 = quot_sign(round_h2(q, q), (f16x2){ah, ah});
    const unsigned short rb = __builtin_bit_cast(unsigned short, ref), fb = __builtin_bit_cast(unsigned short, (f16)fast2[0]);
    const bool diff = !odd && rb != fb;
    const unsigned long long md = __builtin_amdgcn_ballot_w64(diff), mf = __builtin_amdgcn_ballot_w64(!odd);
    if (diff) atomicMax(n_diff + 2, (unsigned long long)id);           // an example for the failure message: a | b << 16
    if ((threadIdx.x & 63) == 0) {
        if (md) atomicAdd(n_diff, (unsigned long long)__builtin_popcountll(md));
        if (mf) atomicAdd(n_fast, (unsigned long long)__builtin_popcountll(mf));
    }
}

int launch_quotient_check(unsigned long long *counts, hipStream_t st) {          // counts[3], zeroed by the caller
    hipLaunchKernelGGL(quotient_check_kernel, dim3(1u << 23), dim3(256), 0, st, counts, counts + 1);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

#define FNN_GATHER_PASS_HEADS 63                                   // heads per pass: 63 + the weight-sum row = 4 blocks of 16
bool gather_ok(const GatherParams &p) {
    const int hblocks = (p.heads + 1 + 15) / 16;
    return (hblocks <= 4 || p.n_pass > 1) && p.C <= 32 && p.C % 8 == 0 && (long long)p.PD * p.PH * p.PW < (1LL << 31) / 32 &&
           (long long)p.PD * p.PH * p.PW * p.C * 2 < (1LL << 31) &&   // a slot's bytes stay below the staging's 0x80000000 "not fetched" offset
           p.n_eval <= 8 &&
           (p.n_eval & (p.n_eval - 1)) == 0 &&                 // 1, 2, 4 or 8 evaluations (the mean over them is a multiply by 2^-k)
           ((p.nx <= 64 && p.ny <= 64 && p.nz <= 64) || p.windowed);   // a wave holds 64 of an axis' tile starts one per lane
}

bool gather_tile_windows(const long long *const steps[3], const int n[3], const int extent[3], const long long padded[3],
                         int *tab, int off[3], size_t *count) {
    size_t k = 0;
    for (int d = 0; d < 3; ++d)
        for (int i = 0; i < n[d]; ++i, ++k) if (tab) tab[k] = (int)steps[d][i];
    for (int d = 0; d < 3; ++d) {
        off[d] = -1;
        if (n[d] <= 64) continue;
        off[d] = (int)k;
        const int run = d == 2 ? 64 : 1;                       // a wave owns one (x, y) and 64 consecutive z
        int b = 0;
        for (long long c = 0; c < padded[d]; ++c, ++k) {
            while (b < n[d] && steps[d][b] + extent[d] <= c) ++b;          // first tile that reaches c (starts ascend)
            if (b + 64 < n[d] && steps[d][b + 64] < c + run) return false; // a 65th tile meets [c, c + run)
            if (tab) tab[k] = b;
        }
    }
    *count = k;
    return true;
}

template <int HB, bool TTA, bool K16>
static int launch_gather_hb(const GatherParams &p, hipStream_t st) {
    const long long waves = (long long)(p.x_hi - p.x_lo) * (p.y_hi - p.y_lo) * ((p.z_hi - p.z_lo + 63) / 64);
    if (waves <= 0) return 0;
    const dim3 grid((unsigned)((waves + 3) / 4));
    const size_t lds = (size_t)4 * HB * 16 * 72 * 2;
    const bool labels = p.labels != nullptr;
    fnn_note_kernel("gather_head_kernel<%d,%d,%d,%d,%d>", HB, p.acc_mode, (int)labels, (int)TTA, (int)K16);
    if (p.acc_mode == 1) {
        if constexpr (!K16) {
            if (labels) hipLaunchKernelGGL((gather_head_kernel<HB, 1, true, TTA, false>), grid, dim3(256), lds, st, p);
            else hipLaunchKernelGGL((gather_head_kernel<HB, 1, false, TTA, false>), grid, dim3(256), lds, st, p);
        } else return -1;
    } else if (p.acc_mode == 2) {
        if (labels) hipLaunchKernelGGL((gather_head_kernel<HB, 2, true, TTA, K16>), grid, dim3(256), lds, st, p);
        else hipLaunchKernelGGL((gather_head_kernel<HB, 2, false, TTA, K16>), grid, dim3(256), lds, st, p);
    } else {
        if (labels) hipLaunchKernelGGL((gather_head_kernel<HB, 0, true, TTA, K16>), grid, dim3(256), lds, st, p);
        else hipLaunchKernelGGL((gather_head_kernel<HB, 0, false, TTA, K16>), grid, dim3(256), lds, st, p);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

static int launch_gather_one(const GatherParams &p, hipStream_t st);

int launch_gather(const GatherParams &p0, hipStream_t st) {
    if (!gather_ok(p0)) return -1;
    GatherParams p = p0;
    p.ieee_div = fnn_knob("FNN_GATHER_IEEE") != nullptr;
    if ((p.heads + 1 + 15) / 16 <= 4) return launch_gather_one(p, st);
    // more than 63 classes: one pass per 63 heads over the same activations (the reference has no such limit,
    // predict_from_raw_data.py:587-590); labels need all heads at once - the caller takes the argmax of the logits
    if (p.labels || !p.out || !p.pass_wpk || !p.pass_bias) return -1;
    const size_t plane = (size_t)p.OX * p.OY * p.OZ * (p.out_fp32 ? 4 : 2);
    const int total = p.heads;
    for (int k = 0; k < p.n_pass; ++k) {
        GatherParams q = p;
        const int h0 = k * FNN_GATHER_PASS_HEADS;
        q.heads = total - h0 < FNN_GATHER_PASS_HEADS ? total - h0 : FNN_GATHER_PASS_HEADS;
        q.hblocks = (q.heads + 1 + 15) / 16;
        q.wpk = p.pass_wpk + (size_t)k * 4 * 512;
        q.bias = p.pass_bias + (size_t)k * 64;
        q.out = (char *)p.out + (size_t)h0 * plane;
        if (int rc = launch_gather_one(q, st)) return rc;
    }
    return 0;
}

static int launch_gather_one(const GatherParams &p, hipStream_t st) {
    const int hblocks = (p.heads + 1 + 15) / 16;
    if (p.n_eval > 1) {                                             // mirrored evaluations
        // (round 5: 16 channels take the K = 16 head here too - half the lanes of the K = 32 form load and normalise zeros)
        if (p.C == 16 && p.acc_mode != 1 && fnn_knob("FNN_GATHER_K32") == nullptr) {
            if (hblocks == 1) return launch_gather_hb<1, true, true>(p, st);
            if (hblocks == 2) return launch_gather_hb<2, true, true>(p, st);
            return launch_gather_hb<4, true, true>(p, st);
        }
        if (hblocks == 1) return launch_gather_hb<1, true, false>(p, st);
        if (hblocks == 2) return launch_gather_hb<2, true, false>(p, st);
        return launch_gather_hb<4, true, false>(p, st);
    }
    // 16 channels: the K = 16 head (FNN_GATHER_K32: tests run the K = 32 kernels on the same networks - the same bits)
    if (p.C == 16 && p.acc_mode != 1 && fnn_knob("FNN_GATHER_K32") == nullptr) {
        if (hblocks == 1) return launch_gather_hb<1, false, true>(p, st);
        if (hblocks == 2) return launch_gather_hb<2, false, true>(p, st);
        return launch_gather_hb<4, false, true>(p, st);
    }
    if (hblocks == 1) return launch_gather_hb<1, false, false>(p, st);
    if (hblocks == 2) return launch_gather_hb<2, false, false>(p, st);
    return launch_gather_hb<4, false, false>(p, st);
}
