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

// HB = head blocks of 16 (heads + the weight-sum channel <= 16 HB); LABELS: write the label map instead of the logits.
// ACCM = the accumulate arithmetic (include/fnn.h): 0 FNN_ACC_FP16_REFERENCE - fp32 logits, fp32 product and sum, one
// rounding to fp16 per visit (the reference without autocast: its CPU path); 1 FNN_ACC_FP32; 2 FNN_ACC_FP16_AUTOCAST -
// the reference on a GPU (predict_from_raw_data.py:591-593: the network's output is fp16): logit, mirror sums, product
// and sum each rounded to fp16 - packed fp16 arithmetic, half the instructions of mode 0.
// The kernel is latency bound, not instruction bound: the packed fp16 arithmetic of ACCM = 2 (half the accumulate
// instructions) changed nothing at equal occupancy, a fourth wave per SIMD (its 128 registers instead of 140, asked for
// below) took 18 % off: 23.5 -> 19.3 ms per 512^3 x 61 volume.  ACCM = 0 holds 162 registers (fp32-valued sums): forced
// to 128 it spills and takes 53 ms; with 32-voxel runs per wave (half the accumulators, 4 waves per SIMD without
// scratch) it takes 26-27 ms - the per-wave set-up doubles - so it keeps 64-voxel runs and three waves per SIMD.
// Also measured and dropped: the loads of visit v + 1 issued before the arithmetic of visit v (a second set of
// feature / weight / scale registers: 168 registers, three waves per SIMD): 20.2 -> 24.1 ms - the fourth wave hides more
// than the prefetch does; packed-fp16 normalisation and the head bias as the MFMA's C operand (-25 % VALU work per
// visit) changed nothing: neither instruction issue nor a single wave's round trips bound it, the number of waves does.
// A FIFTH wave per SIMD (96 registers: the head's fragments and biases read from LDS at every use instead of sitting in 32
// registers, the logits leaving through a half-size transpose buffer in two passes) took 24.1 ms instead of 19.0, a
// sixth (80 registers, spills) 31.6: the LDS round trip in front of every MFMA costs more than the wave brings.
// The closing division through one reciprocal per 16 voxels (quot_fast below: ~280 fewer instructions per group)
// took 20.0 -> 19.0 ms.
template <int HB, int ACCM, bool LABELS, bool TTA>
__global__ __launch_bounds__(256, (!TTA && ACCM != 1) ? 4 : 1) void gather_head_kernel(const GatherParams p) {
    constexpr int G = 4, ZW = 16 * G, TP = ZW + 8;             // 16-voxel groups and z voxels per wave; row pitch of the LDS transpose
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
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
    f16x8 wf[HB];
    f32x4 bv[HB];
#pragma unroll
    for (int hb = 0; hb < HB; ++hb) {
        const int hbc = hb < p.hblocks ? hb : p.hblocks - 1;     // HB = 4 with 3 blocks: the copy's rows are >= heads, ignored
        wf[hb] = *(const f16x8 *)(p.wpk + ((size_t)hbc * 64 + lane) * 8);
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
    const int c0 = q * 8 < p.C ? q * 8 : 0;
    const bool live = q * 8 < p.C;
    const f16 slope_h = (f16)p.slope;
    const int *sx = p.steps, *sy = p.steps + p.nx, *sz = p.steps + p.nx + p.ny;

    for (int ix = 0; ix < p.nx; ++ix) {
        const int ox = sx[ix];
        if (xp < ox || xp >= ox + p.PD) continue;              // wave-uniform
        for (int iy = 0; iy < p.ny; ++iy) {
            const int oy = sy[iy];
            if (yp < oy || yp >= oy + p.PH) continue;
            for (int iz = 0; iz < p.nz; ++iz) {
                const int oz = sz[iz];
                if (zp0 + ZW <= oz || zp0 >= oz + p.PW) continue;
                const int pid = (ix * p.ny + iy) * p.nz + iz;
                const int slot = p.slot_tab ? p.slot_tab[pid] : ((ix % p.ring) * p.ny + iy) * p.nz + iz;
                if (slot < 0) continue;                        // not held here (a sharded caller's table): nothing to add
                const int dx = xp - ox, dy = yp - oy;
                bool in[G];
                f16 graw[G];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int zp = zp0 + 16 * g + r;
                    in[g] = zp >= oz && zp < oz + p.PW;
                    graw[g] = p.gauss[in[g] ? (dx * p.PH + dy) * p.PW + zp - oz : 0];
                }
                f32x4 tsum[TTA ? G : 1][HB];                  // mirrored evaluations: running fp32 sum of the logits
                f16x2 tsh[TTA && ACH ? G : 1][HB][2];         // ... or the running fp16 sum (autocast arithmetic)
                for (int f = 0; f < (TTA ? p.n_eval : 1); ++f) {
                    const int fm = TTA ? p.flipmask[f] : 0;
                    // the patch-space voxel (dx, dy, dz) is output voxel (PD-1-dx, ...) of an evaluation whose input was flipped
                    const int fx = (fm & 1) ? p.PD - 1 - dx : dx, fy = (fm & 2) ? p.PH - 1 - dy : dy;
                    const size_t ev = (size_t)f * p.n_slots + slot;
                    // the evaluation's InstanceNorm of this lane's 8 channels
                    const float *qs = p.fss + ev * 2 * p.C + c0;
                    const float4 s0 = *(const float4 *)qs, s1 = *(const float4 *)(qs + 4);
                    const float4 h0 = *(const float4 *)(qs + p.C), h1 = *(const float4 *)(qs + p.C + 4);
                    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
                    const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                    const int rowbase = (fx * p.PH + fy) * p.PW;
                    const f16 *fp = p.feat + ev * P * p.C + c0;
                    f16x8 xraw[G];
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const int dz = zp0 + 16 * g + r - oz;
                        const int v = in[g] ? rowbase + ((fm & 4) ? p.PW - 1 - dz : dz) : 0;
                        xraw[g] = *(const f16x8 *)(fp + (size_t)v * p.C);
                    }
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        // a patch that starts or ends inside the run leaves whole 16-voxel groups untouched: skip them
                        // (wave-uniform; 3.5 patches intersect a 64-voxel run along z, 2 cover each voxel)
                        if (__builtin_amdgcn_ballot_w64(in[g]) == 0) continue;
                        f16x8 o = fnn_norm8(xraw[g], sc, sh);  // norm_act_frag's arithmetic (misc.hip)
                        o = __builtin_elementwise_max(o, o * slope_h);
                        if (!live) o = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
                        const float gw = (float)graw[g];
                        // channel `heads` has zero weights and bias 1: its product is the weight itself; the product is
                        // rounded before the sum (no fma), one rounding to fp16 per visit; lanes outside the patch keep their
                        // sums (and signed zeros).  Measured and dropped: packed fp32 (v_pk_add_f32 / v_pk_mul_f32 issue at well
                        // under half the scalar rate on gfx950) and an exec-masked block per group behind all four MFMAs -
                        // both 1.6x slower than this select-per-value form, whose MFMAs hide behind the previous block's
                        // arithmetic.
#pragma unroll
                        for (int hb = 0; hb < HB; ++hb) {
                            const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[hb], o, bv[hb], 0, 0, 0);   // logit: the bias is the C operand, as in the seg-head kernels
                            if (ACH) {
                                const f16x2 t01 = round_h2(d[0], d[1]);      // the network's fp16 output
                                const f16x2 t23 = round_h2(d[2], d[3]);
                                if (TTA) {
                                    f16x2 (&ts)[2] = tsh[TTA && ACH ? g : 0][hb];
                                    ts[0] = f == 0 ? t01 : add_h2(ts[0], t01);
                                    ts[1] = f == 0 ? t23 : add_h2(ts[1], t23);
                                } else {
                                    const f16x2 gw2 = {graw[g], graw[g]};
                                    f16x2 (&a2)[2] = ah[PKS ? g : 0][hb];
                                    const f16x2 n01 = acc_add_product_h2(a2[0], t01, gw2), n23 = acc_add_product_h2(a2[1], t23, gw2);
                                    a2[0] = in[g] ? n01 : a2[0];
                                    a2[1] = in[g] ? n23 : a2[1];
                                }
                            } else if (TTA) {
                                // predict_from_raw_data.py:541-557: net(x) + sum over the mirror subsets, in their order (fp32)
                                const f32x4 t = d;
                                tsum[TTA ? g : 0][hb] = f == 0 ? t : tsum[TTA ? g : 0][hb] + t;
                            } else if (PKS) {
#pragma unroll
                                for (int e = 0; e < 2; ++e) {
                                    f16x2 &a2 = ah[PKS ? g : 0][hb][e];
                                    const f16x2 nv = round_h2(acc_add_product_1((float)a2[0], d[2 * e], gw),
                                                              acc_add_product_1((float)a2[1], d[2 * e + 1], gw));
                                    a2 = in[g] ? nv : a2;
                                }
                            } else {
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    const float sv = acc_add_product_1(acc[g][hb][j], d[j], gw);
                                    acc[g][hb][j] = in[g] ? sv : acc[g][hb][j];
                                }
                            }
                        }
                    }
                }
                if (TTA) {
                    const float nf = (float)p.n_eval;
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        if (__builtin_amdgcn_ballot_w64(in[g]) == 0) continue;
                        const float gw = (float)graw[g];
                        if (ACH) {
                            const f16x2 gw2 = {graw[g], graw[g]};
#pragma unroll
                            for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                                for (int e = 0; e < 2; ++e) {
                                    const f16x2 ts = tsh[TTA && ACH ? g : 0][hb][e];
                                    const f16x2 t = round_h2(__fdiv_rn((float)ts[0], nf), __fdiv_rn((float)ts[1], nf));   // half /= int
                                    f16x2 &a2 = ah[PKS ? g : 0][hb][e];
                                    const f16x2 nv = acc_add_product_h2(a2, t, gw2);
                                    a2 = in[g] ? nv : a2;
                                }
                            continue;
                        }
#pragma unroll
                        for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                const float t0 = __fdiv_rn(tsum[TTA ? g : 0][hb][2 * e], nf);         // prediction /= (len(axes_combinations) + 1)
                                const float t1 = __fdiv_rn(tsum[TTA ? g : 0][hb][2 * e + 1], nf);
                                if (PKS) {
                                    f16x2 &a2 = ah[PKS ? g : 0][hb][e];
                                    const f16x2 nv = round_h2(acc_add_product_1((float)a2[0], t0, gw), acc_add_product_1((float)a2[1], t1, gw));
                                    a2 = in[g] ? nv : a2;
                                } else {
                                    const float s0 = acc_add_product_1(acc[g][hb][2 * e], t0, gw), s1 = acc_add_product_1(acc[g][hb][2 * e + 1], t1, gw);
                                    acc[g][hb][2 * e] = in[g] ? s0 : acc[g][hb][2 * e];
                                    acc[g][hb][2 * e + 1] = in[g] ? s1 : acc[g][hb][2 * e + 1];
                                }
                            }
                    }
                }
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

    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    const size_t plane = (size_t)p.OX * p.OY * p.OZ;
    const size_t rowoff = ((size_t)x * p.OY + y) * p.OZ + z0;
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
    return (hblocks <= 4 || p.n_pass > 1) && p.C <= 32 && p.C % 8 == 0 && (long long)p.PD * p.PH * p.PW < (1LL << 31) / 32 && p.n_eval <= 8;
}

template <int HB, bool TTA>
static int launch_gather_hb(const GatherParams &p, hipStream_t st) {
    const long long waves = (long long)(p.x_hi - p.x_lo) * (p.y_hi - p.y_lo) * ((p.z_hi - p.z_lo + 63) / 64);
    if (waves <= 0) return 0;
    const dim3 grid((unsigned)((waves + 3) / 4));
    const size_t lds = (size_t)4 * HB * 16 * 72 * 2;
    const bool labels = p.labels != nullptr;
    if (p.acc_mode == 1) {
        if (labels) hipLaunchKernelGGL((gather_head_kernel<HB, 1, true, TTA>), grid, dim3(256), lds, st, p);
        else hipLaunchKernelGGL((gather_head_kernel<HB, 1, false, TTA>), grid, dim3(256), lds, st, p);
    } else if (p.acc_mode == 2) {
        if (labels) hipLaunchKernelGGL((gather_head_kernel<HB, 2, true, TTA>), grid, dim3(256), lds, st, p);
        else hipLaunchKernelGGL((gather_head_kernel<HB, 2, false, TTA>), grid, dim3(256), lds, st, p);
    } else {
        if (labels) hipLaunchKernelGGL((gather_head_kernel<HB, 0, true, TTA>), grid, dim3(256), lds, st, p);
        else hipLaunchKernelGGL((gather_head_kernel<HB, 0, false, TTA>), grid, dim3(256), lds, st, p);
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
    if (p.n_eval > 1) {
        if (hblocks == 1) return launch_gather_hb<1, true>(p, st);
        if (hblocks == 2) return launch_gather_hb<2, true>(p, st);
        return launch_gather_hb<4, true>(p, st);
    }
    if (hblocks == 1) return launch_gather_hb<1, false>(p, st);
    if (hblocks == 2) return launch_gather_hb<2, false>(p, st);
    return launch_gather_hb<4, false>(p, st);
}
