// conv3d_thin.hip - the full-resolution end of the network without its two largest HBM round trips (gfx950).
//
// At full resolution the U-Net has 16 channels: its layers are bound by HBM, not by the matrix cores (47 MB per
// tensor and patch of the benchmark network).  Two of those tensors exist only to be read once by the next kernel:
//   * the stem's output (first conv, 1 input channel), read by the second conv of stage 0;
//   * the last ConvTranspose3d's output ("up"), read - next to the skip - by the first conv of the last decoder stage.
// Both producers are tiny GEMMs per voxel (stem: K = taps <= 32; transposed conv with kernel = stride: one input voxel
// per output voxel, K = 32 input channels), so the consumer recomputes them while it stages its halo tile:
//   conv_thin_kernel<KD, 1, FUSE_STEM>   image of the 16-channel chunk = LeakyReLU(norm(stem(x))) computed with one
//                                         MFMA per 16 halo voxels from a raw fp32 window of the volume kept in LDS;
//   conv_thin_kernel<KD, 2, FUSE_TCONV>  chunk 0 ("up") = transposed conv of the normalised low-resolution voxels
//                                         (one load of 16 B per lane = a ready MFMA operand, one MFMA per stride
//                                         phase), chunk 1 (skip) staged as in conv3d_persist_kernel.
// The stem's InstanceNorm statistics need the whole patch before any voxel can be normalised: stem_mfma_kernel
// computes them in a pass of its own over the 1-channel volume (6 MB per patch instead of writing 47 MB); with
// `out` set it is also the stand-alone stem (same arithmetic, so fused and unfused engines agree bit for bit).
//
// Arithmetic of the stem: x and the weights rounded to fp16, products exact, fp32 accumulation in the MFMA's order,
// one rounding to fp16 - the same contract as every other conv of the engine.
//
// Measured and dropped (round 2): the single-channel (1, 3, 3) stem as ONE v_mfma_f32_32x32x16_f16 per 32 voxels (K = 16
// holds the 9 taps, raw window kept in fp16, eight ds_read_u16 at constant offsets, A rows permuted so that a lane ends
// up with 8 consecutive channels = one 16-byte store): 2.25x fewer instructions on paper, SLOWER on the GPU - stem pass
// 568 us against 337 us per batch, fused consumer 1417 us against 1020 us (the d16 loads chain through their destination
// registers and the 32x32 MFMA's result arrives 64 cycles late).  The 16x16x32 form below stays.
//
// Replaces (with conv3d.hip / misc.hip) the ConvDropoutNormReLU stacks and the transpconvs of the reference's
// PlainConvUNet decoder, nnUNetDistillationTrainer.py:141-173; patch slicing predict_from_raw_data.py:560-566.
#include "fnn_device.h"
#include "conv_common.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int TH_PW = 12;                                    // LDS row pitch of the halo image (voxels), 4 mod 8

// fp16 affine + LeakyReLU of 4 values exactly as conv3d_persist_kernel::commit() does it for 8
static __device__ __forceinline__ f16x4 norm_act4(f16x4 h, const float (&sc)[4], const float (&sh)[4], f16 slope_h) {
#ifdef FNN_NORM_FP32
    f16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (f16)fmaf((float)h[j], sc[j], sh[j]);
#else
    f16x4 sc_h, sh_h;
#pragma unroll
    for (int j = 0; j < 4; ++j) { sc_h[j] = (f16)sc[j]; sh_h[j] = (f16)sh[j]; }
    f16x4 o = h * sc_h + sh_h;
#endif
    return __builtin_elementwise_max(o, o * slope_h);
}

// One stem GEMM: D[16 couts, 16 voxels] from the raw fp32 window in LDS.  Lane (r = voxel, q = k-group): element j
// of the B operand is tap k = 8 q + j of voxel r (k >= taps: any finite value, its weights are zero).
static __device__ __forceinline__ f16x4 stem_block(const float *sRaw, int rawbase, const int (&tapoff)[8], f16x8 wf,
                                                   const float4 &bias) {
    f16x8 xb;
#pragma unroll
    for (int j = 0; j < 8; ++j) xb[j] = (f16)sRaw[rawbase + tapoff[j]];
    const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xb, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    f16x4 h;
    h[0] = (f16)(d[0] + bias.x); h[1] = (f16)(d[1] + bias.y); h[2] = (f16)(d[2] + bias.z); h[3] = (f16)(d[3] + bias.w);
    return h;
}

}  // namespace

// ----------------------------------------------------------------------------
// stem: first conv of the network (+ statistics; output optional)
// ----------------------------------------------------------------------------
// Workgroup = 16 x 8 x 8 output voxels (64 column blocks, 16 per wave), 16 output channels (blockIdx.y picks the block);
// raw window of the patch in LDS (C x (15 + kd) x (7 + kh) x (7 + kw) floats); zero padding at the PATCH border; mirroring
// flips the window read.  GEMM K = C * taps in k-steps of 32 (one for a single-channel CT): element k of the im2col
// column is input channel k / taps, tap k % taps; the per-k LDS offsets come from a table.
#define STEMM_TD 16
#define STEMM_CG 8                                                   // input channels staged at a time (57.6 KB of raw window at 3x3x3)
// More than STEMM_CG input channels (a cascade stage with many foreground labels, label_handling.py:294-311: image + one
// one-hot channel per label) run as groups of STEMM_CG channels: a group's window is staged, its k-steps (K = 8 channels x
// taps, padded to a multiple of 32) are accumulated into the 16 column blocks' accumulators, the next group follows.
// (A single k-step with 16 or 32 output channels - every single-channel CT stem - runs stem_mfma1_kernel below.)
__global__ __launch_bounds__(256) void stem_mfma_kernel(const StemParams p, const f16 *wfrag, const int ksteps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
    int t = blockIdx.x;
    const int tw = t % p.tiles_w; t /= p.tiles_w;
    const int th = t % p.tiles_h; t /= p.tiles_h;
    const int td = t % p.tiles_d;
    const int n = t / p.tiles_d;
    const int cb = blockIdx.y;
    const int pd = (p.kd - 1) / 2, ph = (p.kh - 1) / 2, pw = (p.kw - 1) / 2;
    const int RD = STEMM_TD - 1 + p.kd, RH = 7 + p.kh, RW = 7 + p.kw, RVOX = RD * RH * RW, T = p.kd * p.kh * p.kw;
    const int CGn = p.C < STEMM_CG ? p.C : STEMM_CG;                     // channels per group
    const int ngroups = (p.C + STEMM_CG - 1) / STEMM_CG, ksg = ksteps / ngroups;   // k-steps per group
    float *sRaw = (float *)smem;                                         // [CGn][RD][RH][RW]
    int *sTab = (int *)(sRaw + ((CGn * RVOX + 3) & ~3));                 // [ksg * 32] LDS offset of im2col element k of a group
    float *sRed = (float *)(sTab + ksg * 32);                            // [4 waves][16][2]

    const int ox = p.origins[n * 3 + 0], oy = p.origins[n * 3 + 1], oz = p.origins[n * 3 + 2];
    const int d0 = td * STEMM_TD - pd, h0 = th * 8 - ph, w0 = tw * 8 - pw;
    const float *voln = p.vol + (size_t)n * p.vol_batch_stride;
    auto stage = [&](int c0) {                                           // channels c0 .. c0 + CGn - 1 (beyond C: zeros, their weights are zero too)
        for (int cl = 0; cl < CGn; ++cl)
            for (int v = tid; v < RVOX; v += 256) {
                const int zd = v / (RH * RW), rem = v - zd * (RH * RW), zh = rem / RW, zw = rem - zh * RW;
                int d = d0 + zd, h = h0 + zh, w = w0 + zw;
                const bool ok = d >= 0 && d < p.PD && h >= 0 && h < p.PH && w >= 0 && w < p.PW && c0 + cl < p.C;
                if (p.flip_d) d = p.PD - 1 - d;
                if (p.flip_h) h = p.PH - 1 - h;
                if (p.flip_w) w = p.PW - 1 - w;
                const int c = c0 + cl < p.C ? c0 + cl : 0;
                const float val = voln[(((size_t)c * p.X + (ox + (ok ? d : 0))) * p.Y + (oy + (ok ? h : 0))) * p.Z + (oz + (ok ? w : 0))];
                sRaw[cl * RVOX + v] = ok ? val : 0.f;
            }
    };
    stage(0);
    for (int k = tid; k < ksg * 32; k += 256) {
        const int kk = k < CGn * T ? k : 0;                              // padding elements: any finite value, their weights are zero
        const int c = kk / T, tap = kk - c * T;
        sTab[k] = c * RVOX + ((tap / (p.kh * p.kw)) * RH + (tap / p.kw) % p.kh) * RW + tap % p.kw;
    }
    const float4 bias = *(const float4 *)(p.bias + cb * 16 + q * 4);
    __syncthreads();

    float t1[4] = {0.f, 0.f, 0.f, 0.f}, t2[4] = {0.f, 0.f, 0.f, 0.f};
    const f16x2 ones = {(f16)1.f, (f16)1.f};
    const int oh = th * 8 + (r >> 3), ow = tw * 8 + (r & 7);
    auto finish = [&](int j, const f32x4 &d) {                           // bias, store, statistics of column block j
        const int dl = wave * 4 + (j >> 2), hp = j & 3;
        const int od = td * STEMM_TD + dl, ohh = oh + 2 * hp;
        f16x4 h;
        h[0] = (f16)(d[0] + bias.x); h[1] = (f16)(d[1] + bias.y); h[2] = (f16)(d[2] + bias.z); h[3] = (f16)(d[3] + bias.w);
        const bool ok = od < p.PD && ohh < p.PH && ow < p.PW;
        if (ok && p.out) *(f16x4 *)(p.out + ((((size_t)n * p.PD + od) * p.PH + ohh) * p.PW + ow) * p.Cout + cb * 16 + q * 4) = h;
        if (!ok) h = (f16x4){0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f16x2 pr = {h[c], (f16)0.f};
            t1[c] = __builtin_amdgcn_fdot2(pr, ones, t1[c], false);
            t2[c] = __builtin_amdgcn_fdot2(pr, pr, t2[c], false);
        }
    };
    if (ngroups == 1) {
#pragma unroll 4
        for (int j = 0; j < 16; ++j) {                                   // column block = (depth slice, pair of h rows)
            const int dl = wave * 4 + (j >> 2), hp = j & 3;
            const int rawbase = (dl * RH + 2 * hp + (r >> 3)) * RW + (r & 7);
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
            for (int ks = 0; ks < ksteps; ++ks) {
                const f16x8 wf = *(const f16x8 *)(wfrag + ((size_t)(cb * ksteps + ks) * 64 + lane) * 8);
                f16x8 xb;
#pragma unroll
                for (int e = 0; e < 8; ++e) xb[e] = (f16)sRaw[rawbase + sTab[ks * 32 + 8 * q + e]];
                d = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xb, d, 0, 0, 0);
            }
            finish(j, d);
        }
    } else {
        f32x4 dacc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) dacc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int g = 0; g < ngroups; ++g) {
            if (g > 0) {
                __syncthreads();                                         // every wave is done with the previous group's window
                stage(g * STEMM_CG);
                __syncthreads();
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int dl = wave * 4 + (j >> 2), hp = j & 3;
                const int rawbase = (dl * RH + 2 * hp + (r >> 3)) * RW + (r & 7);
                for (int ks = 0; ks < ksg; ++ks) {
                    const f16x8 wf = *(const f16x8 *)(wfrag + ((size_t)(cb * ksteps + g * ksg + ks) * 64 + lane) * 8);
                    f16x8 xb;
#pragma unroll
                    for (int e = 0; e < 8; ++e) xb[e] = (f16)sRaw[rawbase + sTab[ks * 32 + 8 * q + e]];
                    dacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xb, dacc[j], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) finish(j, dacc[j]);
    }
    if (p.stats_out) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float a = row16_sum(t1[c]), b = row16_sum(t2[c]);
            if (r == 0) { sRed[(wave * 16 + q * 4 + c) * 2] = a; sRed[(wave * 16 + q * 4 + c) * 2 + 1] = b; }
        }
        __syncthreads();
        if (tid < 32) {
            const int c = tid >> 1, which = tid & 1;
            double v = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += (double)sRed[(w * 16 + c) * 2 + which];
            const int slot = (td * p.tiles_h + th) * p.tiles_w + tw;
            p.stats_out[(((size_t)n * (p.tiles_d * p.tiles_h * p.tiles_w) + slot) * p.Cout + cb * 16 + c) * 2 + which] = v;
        }
    }
}

// Single-channel stems with (KD, 3, 3) taps and 16 or 32 output channels: every CT stem of the isotropic baseline
// workloads (3 x 3 x 3; the (1, 3, 3) stems with full rows run stem_row_kernel).  Same tile, window, k order, arithmetic
// (fp32 accumulation from zero, + bias, one rounding) and statistics row per tile as stem_mfma_kernel, so the values are
// its values bit for bit; what differs is the instruction count per output byte (round 3: the generic kernel ran the
// teacher's 32-channel stem at 1.05 TB/s of output, 10 % of that workload's time; 55 vector instructions per 16 voxels):
//   * ALL cout blocks of a tile in one workgroup: the window is staged once and a column block's operand (8 LDS reads,
//     4 packed converts) feeds NCB MFMAs;
//   * the window's dimensions are compile-time: a lane's 8 tap addresses are computed once, the column block is the
//     ds_read's immediate offset; staging walks (zd, zh, zw) incrementally instead of dividing per element;
//   * 16-byte buffer stores: the two cout blocks of a voxel (NCB = 2) or two column blocks (NCB = 1) exchanged with
//     v_permlane16_swap (pair_to_b128), voxels beyond the patch sent past num_records.  The block part of the address is
//     ADDED to the lane part, not passed as the scalar offset: with an SGPR soffset hipcc places the next block's first
//     write of a data register directly behind the store (LLVM's hazard recogniser exempts stores with a register
//     soffset from the "VALU write of >8-byte store data" wait state), and on gfx950 the store then picks up the new
//     value in some lanes - seen as sporadic wrong channel pairs 4-5 of lanes 12-15 (tools/stem_check.cpp);
//   * statistics two column blocks at a time (v_dot2 on pairs).
template <int NCB, int KD>
__global__ __launch_bounds__(256, 4) void stem_mfma1_kernel(const StemParams p, const f16 *wfrag) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int t = blockIdx.x;
    const int tw = t % p.tiles_w; t /= p.tiles_w;
    const int th = t % p.tiles_h; t /= p.tiles_h;
    const int td = t % p.tiles_d;
    const int n = t / p.tiles_d;
    constexpr int PDK = (KD - 1) / 2, RD = STEMM_TD - 1 + KD, RH = 10, RW = 10, RHW = RH * RW, RVOX = RD * RHW, T = KD * 9;
    float *sRaw = (float *)smem;                                         // [RD][RH][RW]
    float *sRed = sRaw + ((RVOX + 3) & ~3);                              // [4 waves][16 NCB][2]

    const int ox = p.origins[n * 3 + 0], oy = p.origins[n * 3 + 1], oz = p.origins[n * 3 + 2];
    const int d0 = td * STEMM_TD - PDK, h0 = th * 8 - 1, w0 = tw * 8 - 1;
    {
        // window element e = (zd, zh, zw), e + 256 by increments; its address = patch origin + one 32-bit offset built with
        // 24-bit multiplies (the launcher checks Y Z < 2^24); mirroring = coordinate a + s x with (a, s) = (P - 1, -1);
        // elements outside the patch read the origin and are zeroed (the conv's padding)
        const float *org = p.vol + (size_t)n * p.vol_batch_stride + ((size_t)ox * p.Y + oy) * p.Z + oz;
        const unsigned YZ = (unsigned)(p.Y * p.Z), Zs = (unsigned)p.Z;
        const int ad = p.flip_d ? p.PD - 1 : 0, sd = p.flip_d ? -1 : 1, ah = p.flip_h ? p.PH - 1 : 0, sh = p.flip_h ? -1 : 1;
        const int aw = p.flip_w ? p.PW - 1 : 0, sw = p.flip_w ? -1 : 1;
        constexpr int s_d = 256 / RHW, s_rem = 256 - s_d * RHW, s_h = s_rem / RW, s_w = s_rem - s_h * RW;
        int zd = tid / RHW, zh = (tid - zd * RHW) / RW, zw = tid - zd * RHW - zh * RW;
        constexpr int NE = (RVOX + 255) / 256;
        float xr[NE];
        bool okr[NE];
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            const int d = d0 + zd, h = h0 + zh, w = w0 + zw;
            okr[u] = (unsigned)d < (unsigned)p.PD && (unsigned)h < (unsigned)p.PH && (unsigned)w < (unsigned)p.PW &&
                     (u + 1 < NE || tid + u * 256 < RVOX);
            const unsigned off = __umul24((unsigned)(ad + sd * d), YZ) + __umul24((unsigned)(ah + sh * h), Zs) + (unsigned)(aw + sw * w);
            xr[u] = *(const float *)((const char *)org + (okr[u] ? off * 4u : 0u));   // 32-bit BYTE offset: scalar base + VGPR offset load
            zw += s_w;
            const int cw = zw >= RW;
            zw -= cw ? RW : 0;
            zh += s_h + cw;
            const int ch = zh >= RH;
            zh -= ch ? RH : 0;
            zd += s_d + ch;
        }
#pragma unroll
        for (int u = 0; u < NE; ++u)
            if (u + 1 < NE || tid + u * 256 < RVOX) sRaw[tid + u * 256] = okr[u] ? xr[u] : 0.f;
    }
    f16x8 wf[NCB];
    float4 bias[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        wf[cb] = *(const f16x8 *)(wfrag + ((size_t)cb * 64 + lane) * 8);
        bias[cb] = *(const float4 *)(p.bias + cb * 16 + q * 4);
    }
    // byte address of tap k = 8 q + e of this lane's voxel of column block (0, 0); k >= taps: tap 0 (its weights are zero)
    int tapaddr[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 8 * q + e, tap = k < T ? k : 0;
        tapaddr[e] = (((tap / 9) * RH + (tap / 3) % 3 + (r >> 3)) * RW + tap % 3 + (r & 7)) * 4;
    }
    __syncthreads();

    float t1[NCB][4], t2[NCB][4];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int c = 0; c < 4; ++c) { t1[cb][c] = 0.f; t2[cb][c] = 0.f; }
    const f16x2 ones = {(f16)1.f, (f16)1.f};
    const unsigned item_bytes = (unsigned)p.PD * p.PH * p.PW * (NCB * 32);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out ? p.out + (size_t)n * (item_bytes >> 1) : p.out, 0,
                                                                           p.out ? item_bytes : 0u, 0x00020000);
    // NCB = 2: lane (r, q) stores channels 16 (q & 1) + 8 (q >> 1) .. + 7 of voxel r of its block; NCB = 1: channels
    // 8 (q >> 1) .. + 7 of voxel r of block j + (q & 1) (two h rows further down)
    const int oh_l = th * 8 + (r >> 3) + (NCB == 1 ? 2 * (q & 1) : 0), ow_l = tw * 8 + (r & 7);
    const unsigned lane_off = (unsigned)(oh_l * p.PW + ow_l) * (NCB * 32) + (NCB == 2 ? (q & 1) * 32 + (q >> 1) * 16 : (q >> 1) * 16);
    const bool full_tile = td * STEMM_TD + STEMM_TD <= p.PD && th * 8 + 8 <= p.PH && tw * 8 + 8 <= p.PW;   // the tile lies inside the patch
    // (launch bounds of >= 2 waves per SIMD: hipcc then keeps the MFMA results in VGPRs instead of copying them out of AGPRs)
    const auto blocks = [&](auto FULL) {
    constexpr bool full = decltype(FULL)::value;
#pragma unroll
    for (int j = 0; j < 16; j += 2) {                                    // column blocks j, j + 1 = (depth slice, two pairs of h rows)
        const int dl = wave * 4 + (j >> 2), od = td * STEMM_TD + dl;
        f16x2 h[2][NCB][2];                                              // [block][cout block][channel pair]
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int hp = (j & 3) + s;
            const char *bp = (const char *)sRaw + wave * (4 * RHW * 4);   // the wave's depth slices; the rest is the immediate offset
            f32x2 xv[4];
#pragma unroll
            for (int e = 0; e < 8; ++e) xv[e >> 1][e & 1] = *(const float *)(bp + tapaddr[e] + (((j >> 2) * RH + 2 * hp) * RW) * 4);
            f16x8 xb;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f16x2 c2 = __builtin_convertvector(xv[e], f16x2);
                xb[2 * e] = c2[0]; xb[2 * e + 1] = c2[1];
            }
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[cb], xb, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                const f32x2 s01 = {d[0] + bias[cb].x, d[1] + bias[cb].y}, s23 = {d[2] + bias[cb].z, d[3] + bias[cb].w};
                h[s][cb][0] = __builtin_convertvector(s01, f16x2);
                h[s][cb][1] = __builtin_convertvector(s23, f16x2);
            }
        }
        bool ok[2] = {true, true};
        if (!full) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                ok[s] = od < p.PD && th * 8 + 2 * ((j & 3) + s) + (r >> 3) < p.PH && tw * 8 + (r & 7) < p.PW;
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    if (!ok[s]) { h[s][cb][0] = (f16x2){0, 0}; h[s][cb][1] = (f16x2){0, 0}; }
                }
            }
        }
        if (p.out) {                                                     // uniform
            const unsigned blk = (unsigned)((od * p.PH + 2 * (j & 3)) * p.PW) * (NCB * 32);      // uniform part of the address
            if (NCB == 2) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const f16x4 a = {h[s][0][0][0], h[s][0][0][1], h[s][0][1][0], h[s][0][1][1]};
                    const f16x4 b = {h[s][NCB - 1][0][0], h[s][NCB - 1][0][1], h[s][NCB - 1][1][0], h[s][NCB - 1][1][1]};
                    const unsigned vo = full || ok[s] ? lane_off : 0x80000000u;
                    __builtin_amdgcn_raw_buffer_store_b128(pair_to_b128(a, b), rsrc, vo + blk + (unsigned)(s * 2 * p.PW) * (NCB * 32), 0, 0);
                }
            } else {
                const f16x4 a = {h[0][0][0][0], h[0][0][0][1], h[0][0][1][0], h[0][0][1][1]};
                const f16x4 b = {h[1][0][0][0], h[1][0][0][1], h[1][0][1][0], h[1][0][1][1]};
                const unsigned vo = full || ((q & 1) ? ok[1] : ok[0]) ? lane_off : 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b128(pair_to_b128(a, b), rsrc, vo + blk, 0, 0);
            }
        }
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f16x2 pr = {h[0][cb][c >> 1][c & 1], h[1][cb][c >> 1][c & 1]};
                t1[cb][c] = __builtin_amdgcn_fdot2(pr, ones, t1[cb][c], false);
                t2[cb][c] = __builtin_amdgcn_fdot2(pr, pr, t2[cb][c], false);
            }
    }
    };
    if (full_tile) blocks(std::true_type{}); else blocks(std::false_type{});
    if (p.stats_out) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float a = row16_sum(t1[cb][c]), b = row16_sum(t2[cb][c]);
                if (r == 0) { sRed[((wave * NCB + cb) * 16 + q * 4 + c) * 2] = a; sRed[((wave * NCB + cb) * 16 + q * 4 + c) * 2 + 1] = b; }
            }
        __syncthreads();
        if (tid < 32 * NCB) {
            const int c = tid >> 1, which = tid & 1;                     // c = channel = cb * 16 + (q * 4 + c)
            double v = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += (double)sRed[(w * NCB * 16 + c) * 2 + which];
            const int slot = (td * p.tiles_h + th) * p.tiles_w + tw;
            p.stats_out[(((size_t)n * (p.tiles_d * p.tiles_h * p.tiles_w) + slot) * p.Cout + c) * 2 + which] = v;
        }
    }
}

// every stem the engine accepts: any number of input channels (groups of STEMM_CG beyond that many), per-axis kernel 1 | 3
bool stem_mfma_ok(int C, int kd, int kh, int kw, int cout_pad) {
    auto k13 = [](int k) { return k == 1 || k == 3; };
    return C >= 1 && C <= 4096 && k13(kd) && k13(kh) && k13(kw) && cout_pad % 16 == 0;
}
// k-steps of the packed stem weights: C <= STEMM_CG: K = C * taps; more: per group of STEMM_CG channels K = STEMM_CG * taps
// padded to a multiple of 32 (element k of group g: channel g * STEMM_CG + k / taps, tap k % taps)
int stem_mfma_ksteps(int C, int taps) {
    if (C <= STEMM_CG) return (C * taps + 31) / 32;
    return ((C + STEMM_CG - 1) / STEMM_CG) * ((STEMM_CG * taps + 31) / 32);
}
// (channel, tap) of element k of k-step ks of the packed stem weights, or false = zero padding
bool stem_mfma_kmap(int C, int taps, int ks, int k, int *c, int *tap) {
    if (C <= STEMM_CG) {
        const int kk = ks * 32 + k;
        if (kk >= C * taps) return false;
        *c = kk / taps; *tap = kk % taps;
        return true;
    }
    const int ksg = (STEMM_CG * taps + 31) / 32, g = ks / ksg, kk = (ks % ksg) * 32 + k;
    const int cc = g * STEMM_CG + kk / taps;
    if (kk >= STEMM_CG * taps || cc >= C) return false;
    *c = cc; *tap = kk % taps;
    return true;
}

int stem_mfma_stats_slots(int PD, int PH, int PW) { return ((PD + STEMM_TD - 1) / STEMM_TD) * ((PH + 7) / 8) * ((PW + 7) / 8); }

// `wfrag`: the stem weights as MFMA "A" fragments [cout block][k-step][64][8]; p.out == nullptr: statistics only.
int launch_stem_mfma(const StemParams &p_in, const f16 *wfrag, int N, hipStream_t st) {
    StemParams p = p_in;
    if (!stem_mfma_ok(p.C, p.kd, p.kh, p.kw, p.Cout)) return -1;
    {
        const int rc = launch_stem_row(p, N, st);                        // one channel, (1, 3, 3), full rows: conv3d_row.hip
        if (rc != -1) return rc;
    }
    p.tiles_d = (p.PD + STEMM_TD - 1) / STEMM_TD;
    p.tiles_h = (p.PH + 7) / 8;
    p.tiles_w = (p.PW + 7) / 8;
    const int ks = stem_mfma_ksteps(p.C, p.kd * p.kh * p.kw);
    const int RVOX = (STEMM_TD - 1 + p.kd) * (7 + p.kh) * (7 + p.kw);
    const int cgn = p.C < STEMM_CG ? p.C : STEMM_CG, ngroups = (p.C + STEMM_CG - 1) / STEMM_CG;
    const size_t lds = (size_t)((cgn * RVOX + 3) & ~3) * 4 + (size_t)(ks / ngroups) * 32 * 4 + 4 * 16 * 2 * 4;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)stem_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const bool no_one = fnn_knob("FNN_NO_STEM1") != nullptr;    // A-B aid, read per launch: a test compares the two kernels in one process
    if (p.C == 1 && p.kh == 3 && p.kw == 3 && (p.Cout == 16 || p.Cout == 32) && !no_one &&
        (size_t)p.PD * p.PH * p.PW * p.Cout * 2 < (1ull << 31) && p.Y * p.Z < (1 << 24) && (size_t)p.PD * p.Y * p.Z < (1ull << 30)) {
        const int RV = (STEMM_TD - 1 + p.kd) * 100;
        const size_t lds1 = (size_t)((RV + 3) & ~3) * 4 + (size_t)4 * (p.Cout / 16) * 16 * 2 * 4;
        const dim3 grid1(N * p.tiles_d * p.tiles_h * p.tiles_w);
        fnn_note_kernel("stem_mfma1_kernel<%d,%d>", p.Cout / 16, p.kd);
        if (p.Cout == 16 && p.kd == 3) hipLaunchKernelGGL((stem_mfma1_kernel<1, 3>), grid1, dim3(256), lds1, st, p, wfrag);
        else if (p.Cout == 16) hipLaunchKernelGGL((stem_mfma1_kernel<1, 1>), grid1, dim3(256), lds1, st, p, wfrag);
        else if (p.kd == 3) hipLaunchKernelGGL((stem_mfma1_kernel<2, 3>), grid1, dim3(256), lds1, st, p, wfrag);
        else hipLaunchKernelGGL((stem_mfma1_kernel<2, 1>), grid1, dim3(256), lds1, st, p, wfrag);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    const dim3 grid(N * p.tiles_d * p.tiles_h * p.tiles_w, p.Cout / 16);
    fnn_note_kernel("stem_mfma_kernel");
    hipLaunchKernelGGL(stem_mfma_kernel, grid, dim3(256), lds, st, p, wfrag, ks);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ----------------------------------------------------------------------------
// thin full-resolution conv with a fused producer
// ----------------------------------------------------------------------------
// Persistent like conv3d_persist_kernel<1, 4, true, KS, CH, PF>: 16 output channels, tile 4 x 8 x 8, (KD, 3, 3)
// taps, all weight fragments resident in LDS, halo image double buffered, one barrier per work item
// (tile, 16-channel chunk), the next item's global loads in flight during the MFMAs of the current one.
template <int KD, int CH, int FUSE, int NCLS, int WPS>
__global__ __launch_bounds__(256, WPS) void conv_thin_kernel(const ThinParams tp, const int total_tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ConvParams &p = tp.c;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
    constexpr int TD = 4, ID = TD + KD - 1, IH = 10, IW = 10, IVOX = ID * IH * IW, T = 9 * KD;
    constexpr int KS = (T + 1) / 2, TS = CH * KS;
    constexpr int PF = (IVOX * 2 + 255) / 256;
    constexpr int ABYTES = (ID * IH * TH_PW * 32 + 1023) & ~1023;
    constexpr int NCB = (IVOX + 15) / 16, NCBW = (NCB + 3) / 4;          // stem: halo column blocks (per wave)
    constexpr int RD = ID + KD - 1, RVOX = RD * 144, RPF = (RVOX + 255) / 256;   // stem: raw window [RD][12][12]
    constexpr int NLBW = 3;                                              // tconv: low-resolution column blocks per wave

    char *sA0 = smem;
    char *sW = smem + 2 * ABYTES;                                        // [TS][64][16 B]
    char *sF = sW + TS * 1024;                                           // fused producer's weight fragments
    constexpr int nfw = FUSE == FUSE_STEM ? 1 : NCLS;
    float *sRaw = (float *)(sF + nfw * 1024);                            // stem: 2 x [RD][12][12]
    int *sTap = (int *)(sRaw + (FUSE == FUSE_STEM ? 2 * RVOX : 0));
    double *sRed = (double *)(sTap + 64);                                // [4 waves][16][2]

    int t_begin, t_end;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int qd = nwg >> 3, rm = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        const int g = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
        t_begin = (int)((long long)total_tiles * g / nwg);
        t_end = (int)((long long)total_tiles * (g + 1) / nwg);
    }
    if (t_begin >= t_end) return;

    // ---- one-time set-up
    for (int idx = tid; idx < TS * 64; idx += 256) ((uint4 *)sW)[idx] = ((const uint4 *)p.wpk)[idx];
    for (int idx = tid; idx < nfw * 64; idx += 256) ((uint4 *)sF)[idx] = ((const uint4 *)tp.fw)[idx];
    if (tid < 2 * KS) {
        int off = 0, par = 0;
        if (tid < T) {
            const int a = tid / 9, b = (tid / 3) % 3, c = tid % 3;
            off = ((a * IH + b) * TH_PW + c) * 32;
            par = b & 1;
        }
        sTap[tid * 2] = off + 16 * par;
        sTap[tid * 2 + 1] = off + 16 * (1 - par);
    }
    for (int i = tid; i < 4 * 16 * 2; i += 256) sRed[i] = 0.0;
    const int cg = tid & 1;
    int rel[PF];                                                         // regular staging: packed halo coords
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int idx = tid + u * 256, v = idx >> 1;
        const int zw = v % IW, zh = (v / IW) % IH, zd = v / (IW * IH);
        rel[u] = idx < IVOX * 2 ? (zd << 16) | (zh << 8) | zw : -1;
    }
    int base[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        int od_l, oh_l, ow_l;
        mb_coords<4>(wave, mb, r, od_l, oh_l, ow_l);
        base[mb] = ((od_l * IH + oh_l) * TH_PW + ow_l) * 32;
    }
    const int kgp = ((lane >> 4) & 1) ^ ((lane & 15) >> 3);
    const float4 bv4 = *(const float4 *)(p.bias + q * 4);
    float4 bv[1] = {bv4};
    double dsum[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) { dsum[j][0] = 0.0; dsum[j][1] = 0.0; }
    f32x4 acc[4][1];

    // ---- fused producer: per-lane constants
    // stem: column blocks cb = wave + 4 j of the halo's IVOX voxels
    int s_rawbase[NCBW], s_img[NCBW], s_zp[NCBW];
    int tapoff[8];
    f16x8 fwf = {0, 0, 0, 0, 0, 0, 0, 0};
    float4 fbias = make_float4(0.f, 0.f, 0.f, 0.f);
    int rrel[RPF];
    if (FUSE == FUSE_STEM) {
#pragma unroll
        for (int j = 0; j < NCBW; ++j) {
            const int v = (wave + 4 * j) * 16 + r;
            const int vc = v < IVOX ? v : IVOX - 1;
            const int zd = vc / 100, rem = vc - zd * 100, zh = rem / 10, zw = rem - zh * 10;
            s_rawbase[j] = (zd * 12 + zh) * 12 + zw;
            s_img[j] = ((zd * IH + zh) * TH_PW + zw) * 32 + (((q >> 1) ^ (zh & 1)) * 16) + (q & 1) * 8;
            s_zp[j] = v < IVOX ? (zd << 16) | (zh << 8) | zw : -1;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * q + j, tap = k < T ? k : 0;
            tapoff[j] = ((tap / 9) * 12 + (tap / 3) % 3) * 12 + tap % 3;
        }
#pragma unroll
        for (int u = 0; u < RPF; ++u) {
            const int e = tid + u * 256;
            const int rd = e / 144, rem = e - rd * 144, rh = rem / 12, rw = rem - rh * 12;
            rrel[u] = e < RVOX ? (rd << 16) | (rh << 8) | rw : -1;
        }
        fbias = *(const float4 *)(tp.fbias + q * 4);
    } else {
        fbias = *(const float4 *)(tp.fbias + q * 4);
    }
    // tconv: low-resolution halo voxels, LD x LH x LW per tile
    const int LH = tp.tsh == 2 ? 6 : IH, LW = tp.tsw == 2 ? 6 : IW;
    const int LD = tp.tsd == 2 ? (ID + 2) / 2 : ID;                      // KD = 3: ID = 6 -> 4; KD = 1: ID = 4 -> 3 (od0 is a multiple of 4)
    const int LV = LD * LH * LW;
    // The halo origin is odd in-plane and od0 is a multiple of 4, so where a (low voxel, stride phase) pair lands in
    // the halo image does not depend on the tile: per block j the image offset of phase 0 and a mask of the phases that
    // fall inside the halo, per phase a lane-constant offset (with the row swizzle: the parity of zh is the phase's).
    int l_zp[NLBW], l_img[NLBW], l_in[NLBW], c_off[NCLS];
    if (FUSE == FUSE_TCONV) {
        const float rlw = 1.0f / (float)LW, rlh = 1.0f / (float)LH;
        const int dd = tp.tsd == 2 ? ((KD - 1) / 2) & 1 : 0, dh = tp.tsh == 2 ? 1 : 0, dw = tp.tsw == 2 ? 1 : 0;   // id0 - s * lo
#pragma unroll
        for (int j = 0; j < NLBW; ++j) {
            const int l = (wave + 4 * j) * 16 + r;
            const int lc = l < LV ? l : LV - 1;
            const int row = small_div(lc, LW, rlw), lw = lc - row * LW;
            const int ld = small_div(row, LH, rlh), lh = row - ld * LH;
            l_zp[j] = l < LV ? (ld << 16) | (lh << 8) | lw : -1;
            const int zd0 = ld * tp.tsd - dd, zh0 = lh * tp.tsh - dh, zw0 = lw * tp.tsw - dw;
            l_img[j] = ((zd0 * IH + zh0) * TH_PW + zw0) * 32 + (q & 1) * 8;
            int m = 0;
#pragma unroll
            for (int cls = 0; cls < NCLS; ++cls) {
                const int jd = cls / (tp.tsh * tp.tsw), jh = (cls / tp.tsw) % tp.tsh, jw = cls % tp.tsw;
                const unsigned zd = (unsigned)(zd0 + jd), zh = (unsigned)(zh0 + jh), zw = (unsigned)(zw0 + jw);
                m |= (l < LV && zd < (unsigned)ID && zh < (unsigned)IH && zw < (unsigned)IW) ? 1 << cls : 0;
            }
            l_in[j] = m;
        }
#pragma unroll
        for (int cls = 0; cls < NCLS; ++cls) {
            const int jd = cls / (tp.tsh * tp.tsw), jh = (cls / tp.tsw) % tp.tsh, jw = cls % tp.tsw;
            c_off[cls] = ((jd * IH + jh) * TH_PW + jw) * 32 + (((q >> 1) ^ ((jh - dh) & 1)) * 16);
        }
    }
    __syncthreads();
    if (FUSE == FUSE_STEM) fwf = *(const f16x8 *)(sF + lane * 16);
    int toffs[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) toffs[ks] = sTap[(2 * ks + (lane >> 5)) * 2 + kgp];

    auto tile_coords = [&](int t, int &n, int &od0, int &oh0, int &ow0) {
        const int tw = t % p.tiles_w; t /= p.tiles_w;
        const int th = t % p.tiles_h; t /= p.tiles_h;
        const int td = t % p.tiles_d;
        n = t / p.tiles_d;
        od0 = td * TD; oh0 = th * 8; ow0 = tw * 8;
    };
    auto next_tile = [&](int &n, int &od0, int &oh0, int &ow0) {
        ow0 += 8;
        if (ow0 >= p.tiles_w * 8) {
            ow0 = 0; oh0 += 8;
            if (oh0 >= p.tiles_h * 8) {
                oh0 = 0; od0 += TD;
                if (od0 >= p.tiles_d * TD) { od0 = 0; ++n; }
            }
        }
    };

    // ---- regular staging (chunk CH - 1 = the consumer's last source: loads + normalise + LDS image)
    int offv[PF];
    f16x8 xr[PF];
    float4 scr[2], shr[2];
    float slope_next = 1.f;
    auto set_offsets = [&](int od0, int oh0, int ow0) {
        const int id0 = od0 - (KD - 1) / 2, ih0 = oh0 - 1, iw0 = ow0 - 1;
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const unsigned gd = (unsigned)(id0 + (rel[u] >> 16)), gh = (unsigned)(ih0 + ((rel[u] >> 8) & 255)),
                           gw = (unsigned)(iw0 + (rel[u] & 255));
            const bool ok = rel[u] >= 0 && gd < (unsigned)p.Di && gh < (unsigned)p.Hi && gw < (unsigned)p.Wi;
            offv[u] = ok ? (int)(__umul24(__umul24(gd, (unsigned)p.Hi) + gh, (unsigned)p.Wi) + gw) : -1;
        }
    };
    auto issue_reg = [&](int n) {
        const SrcDesc &S = p.src[1];
        const int c_loc = cg * 8, sC = S.C;
        const int vs = FNN_VS(S);                                        // activation layout: fnn_device.h, SrcDesc
        const char *sp = (const char *)(S.ptr + (size_t)n * p.Di * p.Hi * p.Wi * sC + (c_loc >> 4) * FNN_CS(S) + (c_loc & 15));
        slope_next = S.slope;
        const float *qs = S.ss ? S.ss + (size_t)(2 * n) * sC + c_loc : p.ident_ss + c_loc;
        const float *qh = S.ss ? qs + sC : p.ident_ss + 512 + c_loc;
        scr[0] = *(const float4 *)qs; scr[1] = *(const float4 *)(qs + 4);
        shr[0] = *(const float4 *)qh; shr[1] = *(const float4 *)(qh + 4);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < PF; ++u) xr[u] = *(const f16x8 *)(sp + (unsigned)((offv[u] >= 0 ? offv[u] : 0) * vs * 2));
    };
    auto commit_reg = [&](char *dst) {
        const f16 slope_h = (f16)slope_next;
        const float sc[8] = {scr[0].x, scr[0].y, scr[0].z, scr[0].w, scr[1].x, scr[1].y, scr[1].z, scr[1].w};
        const float sh[8] = {shr[0].x, shr[0].y, shr[0].z, shr[0].w, shr[1].x, shr[1].y, shr[1].z, shr[1].w};
#ifndef FNN_NORM_FP32
        f16x8 sc_h, sh_h;
#pragma unroll
        for (int j = 0; j < 8; ++j) { sc_h[j] = (f16)sc[j]; sh_h[j] = (f16)sh[j]; }
#endif
#pragma unroll
        for (int u = 0; u < PF; ++u) {
#ifdef FNN_NORM_FP32
            f16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (f16)fmaf((float)xr[u][j], sc[j], sh[j]);
#else
            f16x8 o = xr[u] * sc_h + sh_h;
#endif
            o = __builtin_elementwise_max(o, o * slope_h);
            if (offv[u] < 0) o = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
            const int zd = rel[u] >> 16, zh = (rel[u] >> 8) & 255, zw = rel[u] & 255;
            if (rel[u] >= 0) *(f16x8 *)(dst + ((zd * IH + zh) * TH_PW + zw) * 32 + ((cg ^ (zh & 1)) * 16)) = o;
        }
    };

    // ---- fused stem: raw window of tile (n, od0, oh0, ow0) -> registers -> sRaw[slot]; sRaw[slot] -> image
    float rawv[RPF];
    float ssc[4], ssh[4];                                                // stem InstanceNorm of this lane's 4 channels
    auto issue_raw = [&](int n, int od0, int oh0, int ow0) {
        const int ox = tp.origins[n * 3 + 0], oy = tp.origins[n * 3 + 1], oz = tp.origins[n * 3 + 2];
        const float *voln = tp.vol + (size_t)n * tp.vol_batch_stride;
        const int d0 = od0 - (KD - 1), h0 = oh0 - 2, w0 = ow0 - 2;
#pragma unroll
        for (int u = 0; u < RPF; ++u) {
            int d = d0 + (rrel[u] >> 16), h = h0 + ((rrel[u] >> 8) & 255), w = w0 + (rrel[u] & 255);
            const bool ok = rrel[u] >= 0 && (unsigned)d < (unsigned)p.Di && (unsigned)h < (unsigned)p.Hi && (unsigned)w < (unsigned)p.Wi;
            if (tp.flip_d) d = p.Di - 1 - d;
            if (tp.flip_h) h = p.Hi - 1 - h;
            if (tp.flip_w) w = p.Wi - 1 - w;
            const float val = voln[((size_t)(ox + (ok ? d : 0)) * tp.Y + (oy + (ok ? h : 0))) * tp.Z + (oz + (ok ? w : 0))];
            rawv[u] = ok ? val : 0.f;
        }
    };
    auto store_raw = [&](int slot) {
#pragma unroll
        for (int u = 0; u < RPF; ++u)
            if (rrel[u] >= 0) sRaw[slot * RVOX + tid + u * 256] = rawv[u];
    };
    auto load_stem_ss = [&](int n) {
        const float4 a = *(const float4 *)(tp.fss + (size_t)(2 * n) * 16 + q * 4);
        const float4 b = *(const float4 *)(tp.fss + (size_t)(2 * n + 1) * 16 + q * 4);
        ssc[0] = a.x; ssc[1] = a.y; ssc[2] = a.z; ssc[3] = a.w;
        ssh[0] = b.x; ssh[1] = b.y; ssh[2] = b.z; ssh[3] = b.w;
    };
    auto stem_to_image = [&](int slot, char *dst, int od0, int oh0, int ow0) {
        const int id0 = od0 - (KD - 1) / 2, ih0 = oh0 - 1, iw0 = ow0 - 1;
        const f16 slope_h = (f16)tp.fslope;
        const float *raw = sRaw + slot * RVOX;
#pragma unroll
        for (int j = 0; j < NCBW; ++j) {                                 // no branch per block: the chains interleave; a block
            const f16x4 h = stem_block(raw, s_rawbase[j], tapoff, fwf, fbias);   // past the halo recomputes its last voxel and stores nothing
            f16x4 o = norm_act4(h, ssc, ssh, slope_h);
            const unsigned gd = (unsigned)(id0 + (s_zp[j] >> 16)), gh = (unsigned)(ih0 + ((s_zp[j] >> 8) & 255)),
                           gw = (unsigned)(iw0 + (s_zp[j] & 255));
            if (!(gd < (unsigned)p.Di && gh < (unsigned)p.Hi && gw < (unsigned)p.Wi)) o = (f16x4){0, 0, 0, 0};   // conv zero padding
            if (s_zp[j] >= 0) *(f16x4 *)(dst + s_img[j]) = o;
        }
    };

    // ---- fused transposed conv: low-resolution voxels of the tile's halo -> xr[0 .. NLBW) -> image of chunk 0
    int l_ok[NLBW];
    int lo_d = 0, lo_h = 0, lo_w = 0;                                    // first low-resolution index of the halo per axis
    auto issue_low = [&](int n, int od0, int oh0, int ow0) {
        const SrcDesc &S = tp.low;
        const int id0 = od0 - (KD - 1) / 2, ih0 = oh0 - 1, iw0 = ow0 - 1;
        lo_d = tp.tsd == 2 ? (id0 >> 1) : id0;                           // arithmetic shift = floor (id0 may be -1)
        lo_h = tp.tsh == 2 ? (ih0 >> 1) : ih0;
        lo_w = tp.tsw == 2 ? (iw0 >> 1) : iw0;
        const int c0 = q * 8, sC = S.C;
        const int cc = c0 < sC ? c0 : 0;
        const int vs = FNN_VS(S);
        const char *sp = (const char *)(S.ptr + (size_t)n * tp.Dl * tp.Hl * tp.Wl * sC + (cc >> 4) * FNN_CS(S) + (cc & 15));
        const float *qs = S.ss ? S.ss + (size_t)(2 * n) * sC + cc : p.ident_ss + cc;
        const float *qh = S.ss ? qs + sC : p.ident_ss + 512 + cc;
        scr[0] = *(const float4 *)qs; scr[1] = *(const float4 *)(qs + 4);
        shr[0] = *(const float4 *)qh; shr[1] = *(const float4 *)(qh + 4);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NLBW; ++j) {
            const unsigned gd = (unsigned)(lo_d + (l_zp[j] >> 16)), gh = (unsigned)(lo_h + ((l_zp[j] >> 8) & 255)),
                           gw = (unsigned)(lo_w + (l_zp[j] & 255));
            const bool ok = l_zp[j] >= 0 && gd < (unsigned)tp.Dl && gh < (unsigned)tp.Hl && gw < (unsigned)tp.Wl;
            l_ok[j] = (ok && c0 < sC ? 1 : 0) | (ok ? 2 : 0);
            const unsigned off = ok ? __umul24(__umul24(gd, (unsigned)tp.Hl) + gh, (unsigned)tp.Wl) + gw : 0u;
            xr[j] = *(const f16x8 *)(sp + off * (unsigned)(vs * 2));
        }
    };
    auto low_to_image = [&](char *dst) {
        const f16 slope_h = (f16)tp.low.slope;
        const float sc[8] = {scr[0].x, scr[0].y, scr[0].z, scr[0].w, scr[1].x, scr[1].y, scr[1].z, scr[1].w};
        const float sh[8] = {shr[0].x, shr[0].y, shr[0].z, shr[0].w, shr[1].x, shr[1].y, shr[1].z, shr[1].w};
#pragma unroll
        for (int j = 0; j < NLBW; ++j) {                                 // no branch per block (see stem_to_image)
            f16x8 o = fnn_norm8(xr[j], sc, sh);                          // load_act_frag's arithmetic (misc.hip)
            o = __builtin_elementwise_max(o, o * slope_h);
            if (!(l_ok[j] & 1)) o = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int cls = 0; cls < NCLS; ++cls) {
                const f16x8 wf = *(const f16x8 *)(sF + (cls * 64 + lane) * 16);
                const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, o, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                f16x4 h;
                h[0] = (f16)(d[0] + fbias.x); h[1] = (f16)(d[1] + fbias.y); h[2] = (f16)(d[2] + fbias.z); h[3] = (f16)(d[3] + fbias.w);
                // a low-resolution voxel outside its tensor <=> its full-resolution voxels outside the patch: zero padding
                if (!(l_ok[j] & 2)) h = (f16x4){0, 0, 0, 0};
                if ((l_in[j] >> cls) & 1) *(f16x4 *)(dst + l_img[j] + c_off[cls]) = h;
            }
        }
    };

    auto kloop = [&](const char *sA, int ch) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            f16x8 xf[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) xf[mb] = *(const f16x8 *)(sA + base[mb] + toffs[ks]);
            const f16x8 wf = *(const f16x8 *)(sW + ((size_t)(ch * KS + ks) * 64 + lane) * 16);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) acc[mb][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf[mb], acc[mb][0], 0, 0, 0);
        }
    };
    auto epilogue = [&](int n, int od0, int oh0, int ow0) {
        float t1[1][4], t2[1][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { t1[0][j] = 0.f; t2[0][j] = 0.f; }
        tile_epilogue<1, 4>(p, acc, bv, n, od0, oh0, ow0, 0, wave, lane, t1, t2);
        if (p.stats_out) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { dsum[j][0] += (double)t1[0][j]; dsum[j][1] += (double)t2[0][j]; }
        }
    };
    auto flush_stats = [&](int n) {
        if (!p.stats_out) return;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double a = row16_sum_f64(dsum[j][0]), b = row16_sum_f64(dsum[j][1]);
            if (r == 0) { double *slot = sRed + (wave * 16 + q * 4 + j) * 2; slot[0] = a; slot[1] = b; }
            dsum[j][0] = 0.0; dsum[j][1] = 0.0;
        }
        __syncthreads();
        if (tid < 32) {
            const int c = tid >> 1, which = tid & 1;
            double v = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { v += sRed[(w * 16 + c) * 2 + which]; sRed[(w * 16 + c) * 2 + which] = 0.0; }
            unsafeAtomicAdd(p.stats_out + (((size_t)n * FNN_STAT_REPL + (blockIdx.x & (FNN_STAT_REPL - 1))) * p.Cout + c) * 2 + which, v);
        }
        __syncthreads();
    };
    auto zero_acc = [&]() {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[mb][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };

    int n_cur, od0, oh0, ow0;
    tile_coords(t_begin, n_cur, od0, oh0, ow0);
    int buf = 0;

    if (FUSE == FUSE_STEM) {
        // raw windows run two tiles ahead (registers, then sRaw[t & 1]); the image one tile ahead
        int n1 = n_cur, d1 = od0, h1 = oh0, w1 = ow0;                   // tile t + 1
        issue_raw(n_cur, od0, oh0, ow0);
        load_stem_ss(n_cur);
        store_raw(0);
        if (t_begin + 1 < t_end) next_tile(n1, d1, h1, w1);
        issue_raw(n1, d1, h1, w1);
        __syncthreads();
        stem_to_image(0, sA0, od0, oh0, ow0);
        store_raw(1);
        __syncthreads();
        for (int t = t_begin; t < t_end; ++t) {
            int n2 = n1, d2 = d1, h2 = h1, w2 = w1;                      // tile t + 2
            if (t + 2 < t_end) next_tile(n2, d2, h2, w2);
            issue_raw(n2, d2, h2, w2);
            load_stem_ss(n1);                                            // the next tile's batch item (unconditional: see conv3d.hip on waits)
            zero_acc();
            kloop(sA0 + buf * ABYTES, 0);
            epilogue(n_cur, od0, oh0, ow0);
            stem_to_image((t + 1 - t_begin) & 1, sA0 + (buf ^ 1) * ABYTES, d1, h1, w1);
            store_raw((t - t_begin) & 1);
            __syncthreads();
            buf ^= 1;
            if (n1 != n_cur || t + 1 == t_end) flush_stats(n_cur);
            n_cur = n1; od0 = d1; oh0 = h1; ow0 = w1;
            n1 = n2; d1 = d2; h1 = h2; w1 = w2;
        }
    } else {
        issue_low(n_cur, od0, oh0, ow0);
        low_to_image(sA0);
        __syncthreads();
        for (int t = t_begin; t < t_end; ++t) {
            // item (t, 0): "up" chunk; prefetch the skip chunk of the same tile
            set_offsets(od0, oh0, ow0);
            issue_reg(n_cur);
            zero_acc();
            kloop(sA0 + buf * ABYTES, 0);
            commit_reg(sA0 + (buf ^ 1) * ABYTES);
            __syncthreads();
            buf ^= 1;
            // item (t, 1): skip chunk; prefetch the next tile's low-resolution voxels
            int n1 = n_cur, d1 = od0, h1 = oh0, w1 = ow0;
            if (t + 1 < t_end) next_tile(n1, d1, h1, w1);
            issue_low(n1, d1, h1, w1);
            kloop(sA0 + buf * ABYTES, 1);
            epilogue(n_cur, od0, oh0, ow0);
            low_to_image(sA0 + (buf ^ 1) * ABYTES);
            __syncthreads();
            buf ^= 1;
            if (n1 != n_cur || t + 1 == t_end) flush_stats(n_cur);
            n_cur = n1; od0 = d1; oh0 = h1; ow0 = w1;
        }
    }
}

static size_t thin_lds_bytes(int kd, int ch, int fuse, int ncls) {
    const int ID = 4 + kd - 1, T = 9 * kd, KS = (T + 1) / 2;
    const size_t ab = (size_t)((ID * 10 * TH_PW * 32 + 1023) & ~1023);
    const int RVOX = (ID + kd - 1) * 144;
    return 2 * ab + (size_t)ch * KS * 1024 + (size_t)(fuse == FUSE_STEM ? 1 : ncls) * 1024 +
           (fuse == FUSE_STEM ? (size_t)2 * RVOX * 4 : 0) + 256 + 4 * 16 * 2 * 8;
}

// Can launch_conv_thin run this layer?  (engine.hip asks before it plans the fusion)
bool conv_thin_ok(const ThinParams &tp) {
    const ConvParams &p = tp.c;
    if (p.Cout != 16 || p.kh != 3 || p.kw != 3 || (p.kd != 1 && p.kd != 3) || p.sd != 1 || p.sh != 1 || p.sw != 1) return false;
    if (p.Di != p.Do || p.Hi != p.Ho || p.Wi != p.Wo) return false;
    const unsigned long long item = 2ull * p.Do * p.Ho * p.Wo * 16;
    if (item >= (1ull << 31) || (long long)p.Di * p.Hi >= (1 << 24)) return false;
    const int ID = 4 + p.kd - 1;
    if (tp.fuse == FUSE_STEM) {
        if (p.n_src != 1 || p.chunks != 1) return false;
    } else if (tp.fuse == FUSE_TCONV) {
        if (p.n_src != 2 || p.chunks != 2 || p.src[1].C != 16 || tp.low.C > 32 || tp.low.C % 16) return false;
        const int LD = tp.tsd == 2 ? (ID + 2) / 2 : ID, LH = tp.tsh == 2 ? 6 : 10, LW = tp.tsw == 2 ? 6 : 10;
        if (LD * LH * LW > 3 * 4 * 16) return false;
        // two stride phases never pass the window bound above, and a 3 x 3 x 3 consumer's two double-buffered images + weights
        // never fit twice per CU: those four instantiations could not be reached and are gone (round 3)
        const int ncls = tp.tsd * tp.tsh * tp.tsw;
        if (p.kd != 1 || (ncls != 4 && ncls != 8)) return false;
        if (tp.Dl * tp.tsd != p.Di || tp.Hl * tp.tsh != p.Hi || tp.Wl * tp.tsw != p.Wi) return false;
        if ((long long)tp.Dl * tp.Hl >= (1 << 24)) return false;
    } else {
        return false;
    }
    return thin_lds_bytes(p.kd, p.chunks, tp.fuse, tp.tsd * tp.tsh * tp.tsw) * 2 <= 160 * 1024;
}

template <int KD, int CH, int FUSE, int NCLS, int WPS>
static int launch_thin_w(ThinParams tp, hipStream_t st) {
    ConvParams &p = tp.c;
    p.tile_d = 4;
    p.tiles_d = (p.Do + 3) / 4; p.tiles_h = (p.Ho + 7) / 8; p.tiles_w = (p.Wo + 7) / 8;
    const int total = p.N * p.tiles_d * p.tiles_h * p.tiles_w;
    p.ident_ss = conv3d_identity_ss();
    if (!p.ident_ss) return -2;
    const size_t lds = thin_lds_bytes(KD, CH, FUSE, tp.tsd * tp.tsh * tp.tsw);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)conv_thin_kernel<KD, CH, FUSE, NCLS, WPS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > WPS) per_cu = WPS;
    int gx = 256 * per_cu;
    if (gx > total) gx = total;
    fnn_note_kernel("conv_thin_kernel<%d,%d,%d,%d,%d>", KD, CH, FUSE, NCLS, WPS);
    hipLaunchKernelGGL((conv_thin_kernel<KD, CH, FUSE, NCLS, WPS>), dim3(gx), dim3(256), lds, st, tp, total);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <int KD, int CH, int FUSE, int NCLS>
static int launch_thin_t(const ThinParams &tp, hipStream_t st) {
    // two workgroups per CU.  (Three - 168 registers, 24 B of scratch per lane - were 1.2 % faster end to end while these
    // kernels ran the benchmark's full-resolution layers; the row-streaming kernels took those over, and no kernel of the
    // library keeps scratch: round 3.)
    return launch_thin_w<KD, CH, FUSE, NCLS, 2>(tp, st);
}

int launch_conv_thin(const ThinParams &tp, hipStream_t st) {
    if (!conv_thin_ok(tp)) return -1;
    {                                                                    // full rows (and stride (1, 2, 2) for the transposed conv): the row-streaming form
        const int rc = launch_conv_row(tp, st);
        if (rc != -1) return rc;
    }
    const int kd = tp.c.kd;
    if (tp.fuse == FUSE_STEM) return kd == 1 ? launch_thin_t<1, 1, FUSE_STEM, 1>(tp, st) : launch_thin_t<3, 1, FUSE_STEM, 1>(tp, st);
    // FUSE_TCONV: (1, 3, 3) consumers behind a transposed conv of stride (1, 2, 2) or (2, 2, 2) (conv_thin_ok)
    switch (tp.tsd * tp.tsh * tp.tsw) {
        case 4: return launch_thin_t<1, 2, FUSE_TCONV, 4>(tp, st);
        case 8: return launch_thin_t<1, 2, FUSE_TCONV, 8>(tp, st);
    }
    return -1;
}
