// conv3d_row.hip - the 16-channel full-resolution (1, 3, 3) convs as row streams (gfx950).
//
// At full resolution the U-Net of an anisotropic plan has 16 channels and (1, 3, 3) kernels: 5 MFMAs per 16 voxels and
// chunk against 1 KB of HBM traffic - these layers are bound by HBM only if everything AROUND the MFMAs is cheap.  The
// tile kernels (conv3d_persist_kernel, conv_thin_kernel) spend 500 - 1000 instructions per wave and 256-voxel tile
// on halo coordinates, bounds, per-element LDS addresses, store addresses and barriers: they are bound by VALU issue at
// 2.4 - 4 TB/s.  Here a workgroup walks DOWN a strip of one depth plane, full rows of W = 16 NBLK voxels at a time
// (W = 64 ... 192; beyond 128 the k-loop takes a row's column blocks in two halves and one workgroup fills a CU):
//   * no halo along w (the row is complete; two zero columns in LDS are the conv's padding), none along h inside a
//     strip (a ring of 12 LDS rows, each row staged exactly once), none along d (kd = 1);
//   * a group of 4 input rows is ONE contiguous run of 8 W x 16 B in HBM: the loads of a thread are base + constant,
//     the base is a scalar that moves with the step;
//   * wave j computes output row 4 s + j of step s: NBLK column blocks of 16 consecutive voxels; every LDS operand read
//     is lane constant + scalar row offset + immediate block offset, every output store scalar row base + lane
//     constant + immediate;
//   * one barrier per step; groups s + 2 (LDS write) and s + 3 (global loads) are in flight during step s (a second
//     group of loads in flight - two register sets, branch-free step - measured on the fused kernel: -1.3 % end to end).
//     Also measured and dropped here (kept in stem_row_kernel, where it pays): the step's stores outside the
//     tail-group branch as buffer stores that the hardware drops - 584 -> 617 us and 902 -> 915 us.
// LDS image: row pitch (W + 2) voxels x 32 B; the two 16-byte channel halves of a voxel are swapped where bit 2 of its
// column is set, which makes the 16 lanes of an operand read (16 consecutive voxels, one half) hit every bank group
// exactly twice = the full ds_read_b128 rate.
// TCONV: chunk 0 is the last ConvTranspose3d's output, computed by this kernel from the low-resolution tensor while it
// stages (stride (1, 2, 2), kernel = stride: two low rows give four "up" rows; one 16-byte load per lane is the MFMA
// operand, one MFMA per stride phase) - the "up" tensor is never written or read (as conv_thin_kernel<.., FUSE_TCONV>).
//
// Arithmetic = the other conv kernels': packed-fp16 normalise-on-load, MFMA k-steps in the FNN_PACK_LINEAR order,
// fp32 accumulation, bias, one rounding to fp16, statistics of the rounded values (fp32 per step, double across).
//
// Replaces (with conv3d.hip / conv3d_thin.hip) the ConvDropoutNormReLU stacks and the transpconv of the reference's
// PlainConvUNet at full resolution, nnUNetDistillationTrainer.py:141-173.
#include "fnn_device.h"
#include "conv_common.h"
#include <cstdlib>

namespace {

struct RowCur { int n, d, h0, g; };

template <int NBLK, int CH, bool TCONV>
__global__ __launch_bounds__(256, CH == 1 ? (NBLK <= 8 ? 3 : 2) : (NBLK <= 8 ? 2 : 1)) void conv_row_kernel(const ThinParams tp, const int total_units,
                                                                         const int strips, const int SH) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ConvParams &p = tp.c;
    constexpr int W = 16 * NBLK, PB = (W + 2) * 32, RINGB = 12 * PB;     // row pitch, bytes per chunk ring (3 groups x 4 rows)
    constexpr int PF = NBLK / 2;                                         // 16-byte elements per thread and 4-row group
    constexpr int NPL = TCONV ? 1 : CH;                                  // chunks staged from a plain source
    constexpr int TB = TCONV ? (NBLK + 3) / 4 : 1;                       // low-resolution blocks per wave and group
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4, hl = lane >> 5, kh = q & 1;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.Hi, D = p.Di, G = SH >> 2;
    double *sRed = (double *)(smem + CH * RINGB);                        // [4 waves][16][2]

    // ---- this workgroup's units (contiguous, XCD aware) and its group stream
    int u_begin, u_end;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int qd = nwg >> 3, rm = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        const int g = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
        u_begin = (int)((long long)total_units * g / nwg);
        u_end = (int)((long long)total_units * (g + 1) / nwg);
    }
    if (u_begin >= u_end) return;
    const int n_groups = (u_end - u_begin) * (G + 1);
    auto advance = [&](RowCur &c) {
        if (++c.g > G) {
            c.g = 0; c.h0 += SH;
            if (c.h0 >= H) { c.h0 = 0; if (++c.d >= D) { c.d = 0; ++c.n; } }
        }
    };

    // ---- one-time set-up
    // zero columns 0 and W + 1 of every ring row: the conv's padding along w, never overwritten
    if (tid < CH * 12 * 4) {
        const int row = tid >> 2, which = (tid >> 1) & 1, half = tid & 1;
        *(f16x8 *)(smem + row * PB + (which ? (W + 1) * 32 : 0) + half * 16) = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
    }
    for (int i = tid; i < 4 * 16 * 2; i += 256) sRed[i] = 0.0;
    // plain staging: element e = tid + 256 u of a group = (row i, column, 8-channel half); i is wave uniform
    const int cg = tid & 1;
    int e_row[PF], e_goff[PF], e_lds[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int e = tid + 256 * u, i = e / (2 * W), c = e - i * (2 * W), col = (c >> 1) + 1;
        e_row[u] = __builtin_amdgcn_readfirstlane(i);
        e_goff[u] = c * 16;                                              // bytes inside the row (C = 16: 32 B per voxel)
        e_lds[u] = col * 32 + ((cg ^ ((col >> 2) & 1)) * 16);
    }
    // operand reads: k-step ks = taps (2 ks, 2 ks + 1); lane (voxel r, tap hl, half kh); column of tap (tr, tc) = r + tc
    int lanec[5];
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
        const int t = 2 * ks + hl < 9 ? 2 * ks + hl : 8;                 // padded slot: any finite data (its weights are 0)
        const int col = r + t % 3;
        lanec[ks] = col * 32 + ((kh ^ ((col >> 2) & 1)) * 16);
    }
    // weights: all k-steps of all chunks stay in registers (FNN_PACK_LINEAR, one cout block)
    f16x8 wf[CH][5];
#pragma unroll
    for (int ch = 0; ch < CH; ++ch)
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) wf[ch][ks] = *(const f16x8 *)(p.wpk + ((size_t)(ch * 5 + ks) * 64 + lane) * 8);
    const float4 bv = *(const float4 *)(p.bias + q * 4);
    // fused transposed conv: its four phase fragments, bias and image offsets
    f16x8 fwf[TCONV ? 4 : 1];
    f32x4 fb = {0.f, 0.f, 0.f, 0.f};
    int up_lds[2] = {0, 0};
    if (TCONV) {
#pragma unroll
        for (int cls = 0; cls < 4; ++cls) fwf[cls] = *(const f16x8 *)(tp.fw + ((size_t)cls * 64 + lane) * 8);
        const float4 t = *(const float4 *)(tp.fbias + q * 4);
        fb = (f32x4){t.x, t.y, t.z, t.w};
#pragma unroll
        for (int jw = 0; jw < 2; ++jw) {
            const int col = 2 * r + jw + 1;                              // + 32 per low block: bit 2 unchanged
            up_lds[jw] = col * 32 + (((q >> 1) ^ ((col >> 2) & 1)) * 16) + (q & 1) * 8;
        }
    }
    const size_t plane_rows = (size_t)H;                                 // rows per (n, d) plane
    const int Hl = tp.Hl, Wl = tp.Wl;

    // ---- staging state
    f16x8 xr[NPL][PF];
    f16x8 xl[TB];
    f16x8 sc_h[NPL], sh_h[NPL];
    f16 slope_h[NPL];
    float lsc[8], lsh[8];
#ifndef FNN_NORM_FP32
    f16x8 lsc_h = {0, 0, 0, 0, 0, 0, 0, 0}, lsh_h = lsc_h;              // the low tensor's rows rounded to fp16 once per batch item (round 5: fnn_norm8 converted them per low block)
#endif
    int n_ss = -1;
    auto load_ss = [&](int n) {                                          // wave-uniform addresses: scalar loads
#pragma unroll
        for (int pc = 0; pc < NPL; ++pc) {
            const SrcDesc &S = p.src[TCONV ? 1 : pc];
            const float *qs = S.ss ? S.ss + (size_t)(2 * n) * 16 : p.ident_ss;
            const float *qh = S.ss ? qs + 16 : p.ident_ss + 512;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                sc_h[pc][j] = (f16)(cg ? qs[8 + j] : qs[j]);
                sh_h[pc][j] = (f16)(cg ? qh[8 + j] : qh[j]);
            }
            slope_h[pc] = (f16)S.slope;
        }
        if (TCONV) {
            const SrcDesc &S = tp.low;
            const float *qs = S.ss ? S.ss + (size_t)(2 * n) * 32 : p.ident_ss;
            const float *qh = S.ss ? qs + 32 : p.ident_ss + 512;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                lsc[j] = q == 0 ? qs[j] : q == 1 ? qs[8 + j] : q == 2 ? qs[16 + j] : qs[24 + j];
                lsh[j] = q == 0 ? qh[j] : q == 1 ? qh[8 + j] : q == 2 ? qh[16 + j] : qh[24 + j];
            }
#ifndef FNN_NORM_FP32
#pragma unroll
            for (int j = 0; j < 8; ++j) { lsc_h[j] = (f16)lsc[j]; lsh_h[j] = (f16)lsh[j]; }
#endif
        }
    };
    auto issue = [&](const RowCur &c) {
        const int rbase = c.h0 - 1 + 4 * c.g;
#pragma unroll
        for (int pc = 0; pc < NPL; ++pc) {
            const SrcDesc &S = p.src[TCONV ? 1 : pc];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                int row = rbase + e_row[u];
                row = row < 0 ? 0 : (row >= H ? H - 1 : row);            // invalid rows: any valid address (zeroed in commit)
                const char *base = (const char *)(S.ptr + (((size_t)c.n * D + c.d) * plane_rows + row) * (W * 16));
                xr[pc][u] = *(const f16x8 *)(base + (unsigned)e_goff[u]);
            }
        }
        if (TCONV) {
            const int lbase = (c.h0 >> 1) - 1 + 2 * c.g;
#pragma unroll
            for (int t = 0; t < TB; ++t) {
                int bb = wave + 4 * t;
                bb = bb < NBLK ? bb : wave;                              // idle slot: a cached re-read, ignored in commit
                const int lr = bb / (NBLK / 2), bl = bb - lr * (NBLK / 2);
                int lrow = lbase + lr;
                lrow = lrow < 0 ? 0 : (lrow >= Hl ? Hl - 1 : lrow);
                // (voxel v, channels 8 q ..) of the 32-channel low tensor: v * vs + (q >> 1) * cs + (q & 1) * 8 (fnn_device.h, SrcDesc)
                const char *base = (const char *)(tp.low.ptr + (size_t)c.n * tp.Dl * Hl * Wl * 32 + (q >> 1) * FNN_CS(tp.low) +
                                                  (((size_t)c.d * Hl + lrow) * Wl) * FNN_VS(tp.low));
                xl[t] = *(const f16x8 *)(base + (unsigned)((16 * bl + r) * (FNN_VS(tp.low) * 2) + (q & 1) * 16));
            }
        }
    };
    auto commit = [&](const RowCur &c, int slot) {
        if (c.n != n_ss) { load_ss(c.n); n_ss = c.n; }                   // uniform; scalar loads only
        const int rbase = c.h0 - 1 + 4 * c.g;
#pragma unroll
        for (int pc = 0; pc < NPL; ++pc) {
            char *ring = smem + (TCONV ? 1 : pc) * RINGB;
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int row = rbase + e_row[u];
                f16x8 o = xr[pc][u] * sc_h[pc] + sh_h[pc];
                o = __builtin_elementwise_max(o, o * slope_h[pc]);
                if (row < 0 || row >= H) o = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};      // uniform: the conv's zero padding along h
                *(f16x8 *)(ring + (slot * 4 + e_row[u]) * PB + e_lds[u]) = o;
            }
        }
        if (TCONV) {
            const int lbase = (c.h0 >> 1) - 1 + 2 * c.g;
            const f16 lslope = (f16)tp.low.slope;
#pragma unroll
            for (int t = 0; t < TB; ++t) {
                const int bb = wave + 4 * t;
                if (bb < NBLK) {                                         // uniform
                    const int lr = bb / (NBLK / 2), bl = bb - lr * (NBLK / 2);
                    const int lrow = lbase + lr;
                    const bool ok = lrow >= 0 && lrow < Hl;
#ifdef FNN_NORM_FP32
                    f16x8 o = fnn_norm8(xl[t], lsc, lsh);                // load_act_frag's arithmetic (misc.hip)
#else
                    f16x8 o = xl[t] * lsc_h + lsh_h;                     // fnn_norm8 = load_act_frag's arithmetic (misc.hip) on the pre-rounded rows
#endif
                    o = __builtin_elementwise_max(o, o * lslope);
                    char *dst = smem + (slot * 4 + 2 * lr) * PB + bl * 1024;
                    // the four phases' MFMAs back to back into their own registers (as conv_row_stem_kernel's stem rows), then
                    // bias added behind the MFMA like tconv_mfma_kernel: the unfused engine's bits
                    f32x4 dd[4];
#pragma unroll
                    for (int cls = 0; cls < 4; ++cls)
                        dd[cls] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fwf[cls], o, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#ifndef FNN_STEM_SERIAL
                    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
                    for (int cls = 0; cls < 4; ++cls) {
                        f16x4 h;
                        h[0] = (f16)(dd[cls][0] + fb[0]); h[1] = (f16)(dd[cls][1] + fb[1]); h[2] = (f16)(dd[cls][2] + fb[2]); h[3] = (f16)(dd[cls][3] + fb[3]);
                        if (!ok) h = (f16x4){0, 0, 0, 0};                // rows outside the patch: the conv's zero padding
                        *(f16x4 *)(dst + (cls >> 1) * PB + up_lds[cls & 1]) = h;
                    }
                }
            }
        }
    };

    // ---- per-lane statistics (fp32 inside a step, double across)
    double dsum[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) { dsum[j][0] = 0.0; dsum[j][1] = 0.0; }
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

    // ---- one step: output row 4 s + wave of the unit, from ring groups `slot` and `slot + 1`
    const unsigned out_lane = (unsigned)((r + 16 * (q & 1)) * 32 + (q >> 1) * 16);     // after pair_to_b128
    const f16x2 ones = {(f16)1.f, (f16)1.f};
    auto step = [&](const RowCur &c, int slot) {
        const int slot1 = slot == 2 ? 0 : slot + 1;
        f32x4 acc[NBLK];
#pragma unroll
        for (int b = 0; b < NBLK; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) {
            // window row wi = wave + tap row of this chunk's ring: the plain rings start one row above the step's first
            // output row, the "up" ring two rows above
            const int sh0 = (TCONV && ch == 0) ? 1 : 0;
            int ro[3];
#pragma unroll
            for (int tr = 0; tr < 3; ++tr) {
                const int wi = wave + tr + sh0;
                ro[tr] = (wi < 4 ? slot * 4 + wi : slot1 * 4 + wi - 4) * PB;
            }
            const char *ring = smem + ch * RINGB;
#pragma unroll
            for (int ks = 0; ks < 5; ++ks) {
                // taps (2 ks, 2 ks + 1): rows 0 0, 0 1, 1 1, 2 2, 2 (2): only k-step 1 mixes rows
                const int vo = lanec[ks] + (ks == 1 ? (hl ? ro[1] : ro[0]) : ro[ks == 0 ? 0 : ks == 2 ? 1 : 2]);
                constexpr int BG = NBLK <= 8 ? NBLK : NBLK / 2;          // wide rows: two halves (operand registers)
#pragma unroll
                for (int b0 = 0; b0 < NBLK; b0 += BG) {
                    f16x8 xf[BG];
#pragma unroll
                    for (int b = 0; b < BG; ++b) xf[b] = *(const f16x8 *)(ring + vo + (b0 + b) * 512);
#pragma unroll
                    for (int b = 0; b < BG; ++b) acc[b0 + b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ch][ks], xf[b], acc[b0 + b], 0, 0, 0);
                }
            }
        }
        // epilogue: bias, fp16, one 8-byte store per lane and block (512 contiguous bytes per wave instruction), statistics
        const int orow = c.h0 + 4 * c.g + wave;
        char *obase = (char *)(p.out + (((size_t)c.n * D + c.d) * plane_rows + orow) * (W * 16));
        float t1[4] = {0.f, 0.f, 0.f, 0.f}, t2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < NBLK; b += 2) {
            f16x4 o[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                o[h][0] = (f16)(acc[b + h][0] + bv.x);
                o[h][1] = (f16)(acc[b + h][1] + bv.y);
                o[h][2] = (f16)(acc[b + h][2] + bv.z);
                o[h][3] = (f16)(acc[b + h][3] + bv.w);
            }
            *(fnn_u32x4r *)(obase + out_lane + b * 512) = pair_to_b128(o[0], o[1]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f16x2 pr = {o[0][j], o[1][j]};
                t1[j] = __builtin_amdgcn_fdot2(pr, ones, t1[j], false);
                t2[j] = __builtin_amdgcn_fdot2(pr, pr, t2[j], false);
            }
        }
        if (p.stats_out) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { dsum[j][0] += (double)t1[j]; dsum[j][1] += (double)t2[j]; }
        }
    };

    // ---- the stream
    RowCur cc;                                                           // group gi: computed
    {
        const int u = u_begin, strip = u % strips, pd = u / strips;
        cc.h0 = strip * SH; cc.d = pd % D; cc.n = pd / D; cc.g = 0;
    }
    RowCur cw = cc, ci = cc;                                             // groups gi + 2 (LDS write), gi + 3 (global loads)
    int gw = 0, gl = 0;                                                  // their stream indices (clamped to the last group)
    __syncthreads();                                                     // zero columns, sRed
    issue(ci);
    commit(cw, 0);
    if (gl < n_groups - 1) { advance(ci); ++gl; }
    if (gw < n_groups - 1) { advance(cw); ++gw; }
    issue(ci);
    commit(cw, 1);
    if (gl < n_groups - 1) { advance(ci); ++gl; }
    if (gw < n_groups - 1) { advance(cw); ++gw; }
    issue(ci);
    __syncthreads();
    int slot = 0;
    for (int gi = 0; gi < n_groups; ++gi) {
        if (cc.g < G) step(cc, slot);
        commit(cw, slot == 0 ? 2 : slot - 1);                            // group gi + 2 -> slot (gi + 2) % 3
        if (gl < n_groups - 1) { advance(ci); ++gl; }
        if (gw < n_groups - 1) { advance(cw); ++gw; }
        issue(ci);
        __syncthreads();
        const int n_prev = cc.n;
        advance(cc);
        slot = slot == 2 ? 0 : slot + 1;
        if (cc.n != n_prev || gi + 1 == n_groups) flush_stats(n_prev);
    }
}

// ----------------------------------------------------------------------------
// the stem (first conv: one input channel read from the fp32 volume, (1, 3, 3)) as a row stream
// ----------------------------------------------------------------------------
// Same walk as conv_row_kernel.  The ring holds, per raw row and output column w, the 8-byte entry
// (x[w - 1], x[w], x[w + 1], 0) in fp16: the im2col column of a voxel is then two aligned ds_read_b64 - k = 0 .. 7 =
// entries of rows h - 1 and h (lane quarter 0), k = 8 .. 11 = the entry of row h + 1 (quarter 1), the remaining k
// carry zero weights - and ONE MFMA per 16 voxels.  A raw value is written into the three entries it belongs to
// (ds_write_b16); x[-1], x[W] and the fourth slot stay zero from the start: the conv's zero padding at the PATCH
// border.  Arithmetic: x and the weights rounded to fp16, fp32 accumulation on top of the bias (the MFMA's C operand),
// one rounding - as stem_mfma_kernel up to the order of the fp32 sums (their last bit can differ).  Statistics: one row
// per (plane, strip).  STORE = false: the statistics pass in front of conv_row_stem_kernel (p.out == nullptr).
template <int NBLK, bool STORE>
__global__ __launch_bounds__(256, NBLK <= 8 ? 8 : 6) void stem_row_kernel(const StemParams p, const int total_units, const int strips,
                                                           const int SH, const int slots) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int W = 16 * NBLK, PB = W * 8, RINGB = 12 * PB;
    constexpr int PF = (4 * W + 255) / 256;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.PH, D = p.PD, G = SH >> 2;
    float *sRed = (float *)(smem + RINGB);                               // [4 waves][16][2]

    int u_begin, u_end;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int qd = nwg >> 3, rm = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        const int g = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
        u_begin = (int)((long long)total_units * g / nwg);
        u_end = (int)((long long)total_units * (g + 1) / nwg);
    }
    if (u_begin >= u_end) return;
    const int n_groups = (u_end - u_begin) * (G + 1);
    auto advance = [&](RowCur &c) {
        if (++c.g > G) {
            c.g = 0; c.h0 += SH;
            if (c.h0 >= H) { c.h0 = 0; if (++c.d >= D) { c.d = 0; ++c.n; } }
        }
    };
    for (int i = tid; i < RINGB / 16; i += 256) ((uint4 *)smem)[i] = make_uint4(0, 0, 0, 0);

    // weights -> A fragment in this kernel's k order; bias of this lane's 4 channels
    f16x8 wf = {0, 0, 0, 0, 0, 0, 0, 0};
    {
        const int m = lane & 15;
        if (q == 0) {
#pragma unroll
            for (int t = 0; t < 6; ++t) wf[t + t / 3] = (f16)p.w[(size_t)t * p.Cout + m];
        } else if (q == 1) {
#pragma unroll
            for (int t = 6; t < 9; ++t) wf[t - 6] = (f16)p.w[(size_t)t * p.Cout + m];
        }
    }
    const f32x4 bv = *(const f32x4 *)(p.bias + q * 4);
    // staging: element e = tid + 256 u = (row i of the group, column w)
    int e_row[PF], e_col[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int e = tid + 256 * u, i = e / W;
        e_row[u] = i < 4 ? i : -1;
        e_col[u] = e - i * W;
    }
    float xr[PF];
    const float *voln0 = p.vol;
    // the patch origin is re-read only when the batch item changes: the load is a VECTOR memory load (the kernel stores
    // to global memory, so hipcc will not use the scalar cache for it) whose wait would drain the loads and stores in
    // flight once per step
    int n_org = -1, ox = 0, oy = 0, oz = 0;
    auto issue = [&](const RowCur &c) {
        const int rbase = c.h0 - 1 + 4 * c.g;
        if (c.n != n_org) { ox = p.origins[c.n * 3 + 0]; oy = p.origins[c.n * 3 + 1]; oz = p.origins[c.n * 3 + 2]; n_org = c.n; }
        const int dd = p.flip_d ? D - 1 - c.d : c.d;
        // scalar base of the patch's plane + one 32-bit offset per element (stem_row_ok: Y Z < 2^30, so the byte offset fits too): as 64-bit products
        // of the long long strides the addresses were a third of this kernel's vector instructions
        const float *plane = voln0 + (size_t)c.n * p.vol_batch_stride + ((size_t)(ox + dd) * p.Y + oy) * p.Z + oz;
        const unsigned Zu = (unsigned)p.Z;
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            int row = rbase + (e_row[u] < 0 ? 0 : e_row[u]);
            row = row < 0 ? 0 : (row >= H ? H - 1 : row);                // invalid rows: any valid address (zeroed in commit)
            const int hh = p.flip_h ? H - 1 - row : row, ww = p.flip_w ? W - 1 - e_col[u] : e_col[u];
            xr[u] = *(const float *)((const char *)plane + ((unsigned)hh * Zu + (unsigned)ww) * 4u);   // 32-bit BYTE offset: scalar base + VGPR offset load
        }
    };
    // (Measured and dropped: a thread loading the three values of its entry and writing it whole - one ds_write_b64
    // instead of three ds_write_b16 per value, three loads instead of one: 180 -> 190 us for the statistics pass.)
    auto commit = [&](const RowCur &c, int slot) {
        const int rbase = c.h0 - 1 + 4 * c.g;
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            if (e_row[u] < 0) continue;
            const int row = rbase + e_row[u];
            const f16 v = (row < 0 || row >= H) ? (f16)0.f : (f16)xr[u];
            char *rowp = smem + (slot * 4 + e_row[u]) * PB;
            const int w = e_col[u];
            *(f16 *)(rowp + w * 8 + 2) = v;                              // x[w] of entry w
            if (w + 1 < W) *(f16 *)(rowp + (w + 1) * 8) = v;             // x[w' - 1] of entry w' = w + 1
            if (w > 0) *(f16 *)(rowp + (w - 1) * 8 + 4) = v;             // x[w' + 1] of entry w' = w - 1
        }
    };

    const unsigned out_lane = (unsigned)((r + 16 * (q & 1)) * 32 + (q >> 1) * 16);     // after pair_to_b128
    const f16x2 ones = {(f16)1.f, (f16)1.f};
    float t1[4] = {0.f, 0.f, 0.f, 0.f}, t2[4] = {0.f, 0.f, 0.f, 0.f};   // per-lane sums of the unit (<= 16 rows x NBLK values: fp32)
    // Branch-free: a tail group (no output rows) runs the step too, with its stores sent beyond the buffer's num_records,
    // which the hardware drops (p.out == nullptr, the statistics-only pass: num_records 0).  With `if (has rows) step()`
    // the number of stores in flight is unknown to hipcc's waitcnt pass and commit(), whose loads are OLDER than the
    // step's stores, waits for vmcnt(0) - the store acknowledgements - once per step: -5 % on this kernel.  (The conv
    // kernels above lose more to the tail group's wasted k-loop than they gain: measured, left as they are.)
    const unsigned item_bytes = (unsigned)D * H * W * 32;
    auto step = [&](const RowCur &c, int slot, bool valid) {
        const int slot1 = slot == 2 ? 0 : slot + 1;
        int ro[3];
#pragma unroll
        for (int tr = 0; tr < 3; ++tr) {
            const int wi = wave + tr;
            ro[tr] = (wi < 4 ? slot * 4 + wi : slot1 * 4 + wi - 4) * PB;
        }
        const int va = r * 8 + (q == 1 ? ro[2] : ro[0]), vb = r * 8 + (q == 1 ? ro[2] : ro[1]);
        const int orow = c.h0 + 4 * c.g + wave;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out ? p.out + (size_t)c.n * (item_bytes >> 1) : p.out, 0,
                                                                               p.out ? item_bytes : 0u, 0x00020000);
        const unsigned vbase = valid ? out_lane + (unsigned)((c.d * H + orow) * (W * 32)) : 0x80000000u;
#pragma unroll
        for (int b = 0; b < NBLK; b += 2) {
            f16x4 o[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f16x4 ea = *(const f16x4 *)(smem + va + (b + h) * 128), eb = *(const f16x4 *)(smem + vb + (b + h) * 128);
                const f16x8 xb = {ea[0], ea[1], ea[2], ea[3], eb[0], eb[1], eb[2], eb[3]};
                const f32x4 dd = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xb, bv, 0, 0, 0);          // the bias is the C operand
                o[h][0] = (f16)dd[0]; o[h][1] = (f16)dd[1]; o[h][2] = (f16)dd[2]; o[h][3] = (f16)dd[3];
            }
            if (STORE) __builtin_amdgcn_raw_buffer_store_b128(pair_to_b128(o[0], o[1]), rsrc, vbase + b * 512, 0, 0);
            if (valid) {                                                 // uniform; no memory operation inside
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f16x2 pr = {o[0][j], o[1][j]};
                    t1[j] = __builtin_amdgcn_fdot2(pr, ones, t1[j], false);
                    t2[j] = __builtin_amdgcn_fdot2(pr, pr, t2[j], false);
                }
            }
        }
    };
    auto unit_stats = [&](const RowCur &c) {                             // called by every thread at the end of a unit
        if (p.stats_out) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = row16_sum(t1[j]), b = row16_sum(t2[j]);
                if (r == 0) { sRed[(wave * 16 + q * 4 + j) * 2] = a; sRed[(wave * 16 + q * 4 + j) * 2 + 1] = b; }
            }
            __syncthreads();
            if (tid < 32) {
                const int ch = tid >> 1, which = tid & 1;
                double v = 0;
#pragma unroll
                for (int w = 0; w < 4; ++w) v += (double)sRed[(w * 16 + ch) * 2 + which];
                const int slot_id = c.d * strips + c.h0 / SH;
                p.stats_out[(((size_t)c.n * slots + slot_id) * p.Cout + ch) * 2 + which] = v;
            }
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { t1[j] = 0.f; t2[j] = 0.f; }
    };

    RowCur cc;
    {
        const int u = u_begin, strip = u % strips, pd = u / strips;
        cc.h0 = strip * SH; cc.d = pd % D; cc.n = pd / D; cc.g = 0;
    }
    RowCur cw = cc, ci = cc;
    int gw = 0, gl = 0;
    __syncthreads();                                                     // the zeroed ring
    issue(ci);
    commit(cw, 0);
    if (gl < n_groups - 1) { advance(ci); ++gl; }
    if (gw < n_groups - 1) { advance(cw); ++gw; }
    issue(ci);
    commit(cw, 1);
    if (gl < n_groups - 1) { advance(ci); ++gl; }
    if (gw < n_groups - 1) { advance(cw); ++gw; }
    issue(ci);
    __syncthreads();
    int slot = 0;
    for (int gi = 0; gi < n_groups; ++gi) {
        step(cc, slot, cc.g < G);
        commit(cw, slot == 0 ? 2 : slot - 1);
        if (gl < n_groups - 1) { advance(ci); ++gl; }
        if (gw < n_groups - 1) { advance(cw); ++gw; }
        issue(ci);
        __syncthreads();
        if (cc.g == G) unit_stats(cc);                                   // the unit's tail group: its steps are done
        advance(cc);
        slot = slot == 2 ? 0 : slot + 1;
    }
}

// ----------------------------------------------------------------------------
// stem + first conv: conv_row_kernel<NBLK, 1, false> whose 16-channel rows are COMPUTED, not loaded
// ----------------------------------------------------------------------------
// FUSE_STEM: the conv's source LeakyReLU(norm(stem(x))) is recomputed from the fp32 volume while staging - the stem's
// raw output (32 B per voxel written by stem_row_kernel, read back here: 3 GB per 32 patches of 160 x 96 x 96) never
// exists.  The stem's InstanceNorm statistics come from stem_row_kernel run with out == nullptr (same entry ring, same
// k order, same MFMA: the values recomputed here are the ones it counted, bit for bit).
// Per step: the conv's k-loop of group gi | the stem rows of group gi + 2 (wave j: row j of the group, one MFMA per
// 16 voxels from a ring of x entries, bias, rounding, normalise, LeakyReLU, ds_write_b64 into the conv's ring) | the x
// entries of group gi + 3 (six raw rows: the group's four and one either side, re-read per group - 4 B per voxel) |
// the raw loads of group gi + 4.  One barrier per step.
template <int NBLK>
__global__ __launch_bounds__(256, NBLK <= 8 ? 3 : 2) void conv_row_stem_kernel(const ThinParams tp, const int total_units,
                                                                                const int strips, const int SH) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ConvParams &p = tp.c;
    constexpr int W = 16 * NBLK, PB = (W + 2) * 32, RINGB = 12 * PB;     // the conv's ring: 3 groups x 4 rows
    constexpr int XPB = W * 8, XSLOT = 6 * XPB;                          // x entries: (x[w-1], x[w], x[w+1], 0) fp16; 2 slots x 6 rows
    constexpr int PF = (6 * W + 255) / 256;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4, hl = lane >> 5, kh = q & 1;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.Hi, D = p.Di, G = SH >> 2;
    char *xring = smem + RINGB;
    double *sRed = (double *)(smem + RINGB + 2 * XSLOT);                 // [4 waves][16][2]

    int u_begin, u_end;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int qd = nwg >> 3, rm = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        const int g = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
        u_begin = (int)((long long)total_units * g / nwg);
        u_end = (int)((long long)total_units * (g + 1) / nwg);
    }
    if (u_begin >= u_end) return;
    const int n_groups = (u_end - u_begin) * (G + 1);
    auto advance = [&](RowCur &c) {
        if (++c.g > G) {
            c.g = 0; c.h0 += SH;
            if (c.h0 >= H) { c.h0 = 0; if (++c.d >= D) { c.d = 0; ++c.n; } }
        }
    };

    // ---- one-time set-up: zero padding columns of the conv's ring, the whole x ring (its border slots stay zero)
    if (tid < 12 * 4) {
        const int row = tid >> 2, which = (tid >> 1) & 1, half = tid & 1;
        *(f16x8 *)(smem + row * PB + (which ? (W + 1) * 32 : 0) + half * 16) = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
    }
    for (int i = tid; i < 2 * XSLOT / 16; i += 256) ((uint4 *)xring)[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < 4 * 16 * 2; i += 256) sRed[i] = 0.0;
    // conv operand reads and weights (conv_row_kernel)
    int lanec[5];
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
        const int t = 2 * ks + hl < 9 ? 2 * ks + hl : 8;
        const int col = r + t % 3;
        lanec[ks] = col * 32 + ((kh ^ ((col >> 2) & 1)) * 16);
    }
    f16x8 wf[5];
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) wf[ks] = *(const f16x8 *)(p.wpk + ((size_t)ks * 64 + lane) * 8);
    const float4 bv = *(const float4 *)(p.bias + q * 4);
    // the stem's A fragment in stem_row_kernel's k order, rebuilt from tp.fw (k = tap: lane (cout, k >> 3), element k & 7)
    f16x8 swf = {0, 0, 0, 0, 0, 0, 0, 0};
    if (q == 0) {
#pragma unroll
        for (int t = 0; t < 6; ++t) swf[t + t / 3] = tp.fw[(size_t)r * 8 + t];
    } else if (q == 1) {
        swf[0] = tp.fw[(size_t)r * 8 + 6]; swf[1] = tp.fw[(size_t)r * 8 + 7]; swf[2] = tp.fw[(size_t)(16 + r) * 8];
    }
    const f32x4 sbv = *(const f32x4 *)(tp.fbias + q * 4);
    const f16 sslope = (f16)tp.fslope;
    const int st_lds = (r + 1) * 32 + (((q >> 1) ^ (((r + 1) >> 2) & 1)) * 16) + (q & 1) * 8;   // + 512 per column block: bit 2 unchanged
    // raw staging: element e = tid + 256 u = (row i of the six, column w)
    int e_row[PF], e_col[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int e = tid + 256 * u, i = e / W;
        e_row[u] = i < 6 ? i : -1;
        e_col[u] = e - i * W;
    }
    float xr[PF];
    int n_org = -1, ox = 0, oy = 0, oz = 0;                              // re-read when the batch item changes (stem_row_kernel)
    auto issue = [&](const RowCur &c) {
        const int rbase = c.h0 - 2 + 4 * c.g;                            // raw row of element row 0
        if (c.n != n_org) { ox = tp.origins[c.n * 3 + 0]; oy = tp.origins[c.n * 3 + 1]; oz = tp.origins[c.n * 3 + 2]; n_org = c.n; }
        const int dd = tp.flip_d ? D - 1 - c.d : c.d;
        const float *plane = tp.vol + (size_t)c.n * tp.vol_batch_stride + ((size_t)(ox + dd) * tp.Y + oy) * tp.Z + oz;   // (stem_row_kernel)
        const unsigned Zu = (unsigned)tp.Z;
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            int row = rbase + (e_row[u] < 0 ? 0 : e_row[u]);
            row = row < 0 ? 0 : (row >= H ? H - 1 : row);                // invalid rows: any valid address (zeroed in xcommit)
            const int hh = tp.flip_h ? H - 1 - row : row, ww = tp.flip_w ? W - 1 - e_col[u] : e_col[u];
            xr[u] = *(const float *)((const char *)plane + ((unsigned)hh * Zu + (unsigned)ww) * 4u);   // 32-bit BYTE offset: scalar base + VGPR offset load
        }
    };
    auto xcommit = [&](const RowCur &c, int xs) {
        const int rbase = c.h0 - 2 + 4 * c.g;
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            if (e_row[u] < 0) continue;
            const int row = rbase + e_row[u];
            const f16 v = (row < 0 || row >= H) ? (f16)0.f : (f16)xr[u];  // the stem's zero padding along h
            char *rowp = xring + xs * XSLOT + e_row[u] * XPB;
            const int w = e_col[u];
            *(f16 *)(rowp + w * 8 + 2) = v;
            if (w + 1 < W) *(f16 *)(rowp + (w + 1) * 8) = v;
            if (w > 0) *(f16 *)(rowp + (w - 1) * 8 + 4) = v;
        }
    };
    // the stem's InstanceNorm of this lane's four channels (tp.fss: [N][2][16])
    f16x4 ssc = {0, 0, 0, 0}, ssh = {0, 0, 0, 0};
    int n_ss = -1;
    auto stem_rows = [&](const RowCur &c, int slot, int xs) {
        if (c.n != n_ss) {
            const float *qs = tp.fss + (size_t)(2 * c.n) * 16 + q * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) { ssc[j] = (f16)qs[j]; ssh[j] = (f16)qs[16 + j]; }
            n_ss = c.n;
        }
        const int row = c.h0 - 1 + 4 * c.g + wave;                       // this wave's stem row = row `wave` of the conv's group
        const bool ok = row >= 0 && row < H;                             // uniform: outside the patch the CONV pads with zeros
        const char *xb0 = xring + xs * XSLOT + wave * XPB + r * 8;
        const char *pa = xb0 + (q == 1 ? 2 * XPB : 0), *pb = xb0 + (q == 1 ? 2 * XPB : XPB);
        char *dst = smem + (slot * 4 + wave) * PB + st_lds;
        if (!ok) {
#pragma unroll
            for (int b = 0; b < NBLK; ++b) *(f16x4 *)(dst + b * 512) = (f16x4){0, 0, 0, 0};
            return;
        }
#ifndef FNN_STEM_SERIAL
        // the row's MFMAs back to back into their own registers, then their commits (written per block, hipcc puts every block's
        // result into the same four registers: MFMA, eight wait states, the commit - NBLK times in a row per step)
        f32x4 dd[NBLK];
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
            const f16x4 ea = *(const f16x4 *)(pa + b * 128), eb = *(const f16x4 *)(pb + b * 128);
            const f16x8 xb = {ea[0], ea[1], ea[2], ea[3], eb[0], eb[1], eb[2], eb[3]};
            dd[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(swf, xb, sbv, 0, 0, 0);                // bias = the C operand: stem_row_kernel's value
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
            f16x4 o = {(f16)dd[b][0], (f16)dd[b][1], (f16)dd[b][2], (f16)dd[b][3]};
            o = o * ssc + ssh;                                           // conv_row_kernel's commit on it
            o = __builtin_elementwise_max(o, o * sslope);
            *(f16x4 *)(dst + b * 512) = o;
        }
#else
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
            const f16x4 ea = *(const f16x4 *)(pa + b * 128), eb = *(const f16x4 *)(pb + b * 128);
            const f16x8 xb = {ea[0], ea[1], ea[2], ea[3], eb[0], eb[1], eb[2], eb[3]};
            const f32x4 dd = __builtin_amdgcn_mfma_f32_16x16x32_f16(swf, xb, sbv, 0, 0, 0);      // bias = the C operand: stem_row_kernel's value
            f16x4 o = {(f16)dd[0], (f16)dd[1], (f16)dd[2], (f16)dd[3]};
            o = o * ssc + ssh;                                           // conv_row_kernel's commit on it
            o = __builtin_elementwise_max(o, o * sslope);
            *(f16x4 *)(dst + b * 512) = o;
        }
#endif
    };

    // ---- statistics and the conv step: conv_row_kernel<NBLK, 1, false>
    double dsum[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) { dsum[j][0] = 0.0; dsum[j][1] = 0.0; }
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
    const unsigned out_lane = (unsigned)((r + 16 * (q & 1)) * 32 + (q >> 1) * 16);
    const f16x2 ones = {(f16)1.f, (f16)1.f};
    // A unit's tail group has no output rows: k-loop and stores skipped.  (Measured, both slightly slower: the stores
    // outside the branch, dropped by the hardware beyond num_records, so that xcommit's wait for its older loads is
    // vmcnt(5) instead of vmcnt(0) = the store acknowledgements - what pays in stem_row_kernel costs 30 us per launch here;
    // the k-loop run for the tail group too - one basic block, the scheduler free to mix the phases: +0.3 % on the family.)
    const unsigned item_bytes = (unsigned)D * H * W * 32;
    auto step = [&](const RowCur &c, int slot, const bool valid) {
        const int slot1 = slot == 2 ? 0 : slot + 1;
        f16x4 o[NBLK];
#pragma unroll
        for (int b = 0; b < NBLK; ++b) o[b] = (f16x4){0, 0, 0, 0};
      if (valid) {
        f32x4 acc[NBLK];
#pragma unroll
        for (int b = 0; b < NBLK; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        int ro[3];
#pragma unroll
        for (int tr = 0; tr < 3; ++tr) {
            const int wi = wave + tr;
            ro[tr] = (wi < 4 ? slot * 4 + wi : slot1 * 4 + wi - 4) * PB;
        }
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) {
            const int vo = lanec[ks] + (ks == 1 ? (hl ? ro[1] : ro[0]) : ro[ks == 0 ? 0 : ks == 2 ? 1 : 2]);
            constexpr int BG = NBLK <= 8 ? NBLK : NBLK / 2;
#pragma unroll
            for (int b0 = 0; b0 < NBLK; b0 += BG) {
                f16x8 xf[BG];
#pragma unroll
                for (int b = 0; b < BG; ++b) xf[b] = *(const f16x8 *)(smem + vo + (b0 + b) * 512);
#pragma unroll
                for (int b = 0; b < BG; ++b) acc[b0 + b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks], xf[b], acc[b0 + b], 0, 0, 0);
            }
        }
        float t1[4] = {0.f, 0.f, 0.f, 0.f}, t2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < NBLK; b += 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                o[b + h][0] = (f16)(acc[b + h][0] + bv.x);
                o[b + h][1] = (f16)(acc[b + h][1] + bv.y);
                o[b + h][2] = (f16)(acc[b + h][2] + bv.z);
                o[b + h][3] = (f16)(acc[b + h][3] + bv.w);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f16x2 pr = {o[b][j], o[b + 1][j]};
                t1[j] = __builtin_amdgcn_fdot2(pr, ones, t1[j], false);
                t2[j] = __builtin_amdgcn_fdot2(pr, pr, t2[j], false);
            }
        }
        if (p.stats_out) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { dsum[j][0] += (double)t1[j]; dsum[j][1] += (double)t2[j]; }
        }
      }
        const int orow = c.h0 + 4 * c.g + wave;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)c.n * (item_bytes >> 1), 0, item_bytes, 0x00020000);
        const unsigned vbase = valid ? out_lane + (unsigned)((c.d * H + orow) * (W * 32)) : 0x80000000u;
        if (valid)
#pragma unroll
        for (int b = 0; b < NBLK; b += 2) __builtin_amdgcn_raw_buffer_store_b128(pair_to_b128(o[b], o[b + 1]), rsrc, vbase + b * 512, 0, 0);
    };

    // ---- the stream: cc = group gi (conv), cs = gi + 2 (stem rows), cx = gi + 3 (x entries), ci = gi + 4 (raw loads)
    RowCur cc;
    {
        const int u = u_begin, strip = u % strips, pd = u / strips;
        cc.h0 = strip * SH; cc.d = pd % D; cc.n = pd / D; cc.g = 0;
    }
    RowCur cs = cc, cx = cc, ci = cc;
    int gs = 0, gx = 0, gl = 0;
    auto next = [&](RowCur &c, int &g) { if (g < n_groups - 1) { advance(c); ++g; } };
    __syncthreads();                                                     // zeroed rings
    issue(ci);
    xcommit(cx, 0); next(cx, gx); next(ci, gl);
    issue(ci);
    __syncthreads();
    stem_rows(cs, 0, 0); next(cs, gs);
    xcommit(cx, 1); next(cx, gx); next(ci, gl);
    issue(ci);
    __syncthreads();
    stem_rows(cs, 1, 1); next(cs, gs);
    xcommit(cx, 0); next(cx, gx); next(ci, gl);
    issue(ci);
    __syncthreads();
    int slot = 0;
    for (int gi = 0; gi < n_groups; ++gi) {
        step(cc, slot, cc.g < G);
        stem_rows(cs, slot == 0 ? 2 : slot - 1, gi & 1); next(cs, gs);   // group gi + 2 -> ring slot (gi + 2) % 3, x slot (gi + 2) & 1
        xcommit(cx, (gi + 1) & 1); next(cx, gx); next(ci, gl);           // group gi + 3
        issue(ci);
        __syncthreads();
        const int n_prev = cc.n;
        advance(cc);
        slot = slot == 2 ? 0 : slot + 1;
        if (cc.n != n_prev || gi + 1 == n_groups) flush_stats(n_prev);
    }
}

int pick_strips(int H, int &SH) {
    for (int k = 1; k <= 8; ++k)
        if (H % k == 0 && (H / k) % 4 == 0 && H / k <= 64) { SH = H / k; return k; }
    SH = H;
    return 1;
}

template <int NBLK, int CH, bool TCONV>
int launch_row_t(ThinParams tp, hipStream_t st) {
    ConvParams &p = tp.c;
    p.ident_ss = conv3d_identity_ss();
    if (!p.ident_ss) return -2;
    int SH;
    const int strips = pick_strips(p.Hi, SH);
    const int total = p.N * p.Di * strips;
    constexpr int PB = (16 * NBLK + 2) * 32;
    const size_t lds = (size_t)CH * 12 * PB + 4 * 16 * 2 * 8;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)conv_row_kernel<NBLK, CH, TCONV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    int per_cu = (int)((160 * 1024) / lds);
    // persistent workgroups per CU.  One chunk, rows of <= 128 voxels: LDS would take three, TWO are faster (round 6, profiles/r06_row_wpc.txt,
    // r06_row1_wpc_ab.txt: the 16 -> 16 layer at 160 x 96 x 96 609-620 us at three, 564-595 at two, 712 at one; the step does not notice; the fused
    // stem kernel below is the other way round: 591 us at three, 665 at two)
    static const bool row1_three = fnn_knob("FNN_ROW1_WPC3") != nullptr;                       // A-B aid: the cap of rounds 2-5
    int cap = CH == 1 ? (NBLK <= 8 ? (row1_three ? 3 : 2) : 2) : (NBLK <= 8 ? 2 : 1);
    static const int wpc_knob = fnn_knob("FNN_ROW_WPC") ? atoi(fnn_knob("FNN_ROW_WPC")) : 0;      // A-B aid: fewer persistent workgroups per CU (room for another stream's kernels)
    if (wpc_knob > 0 && wpc_knob < cap) cap = wpc_knob;
    if (per_cu > cap) per_cu = cap;
    int gx = 256 * per_cu;
    if (gx > total) gx = total;
    fnn_note_kernel("conv_row_kernel<%d,%d,%d>", NBLK, CH, (int)TCONV);
    hipLaunchKernelGGL((conv_row_kernel<NBLK, CH, TCONV>), dim3(gx), dim3(256), lds, st, tp, total, strips, SH);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <int CH, bool TCONV>
int launch_row_n(const ThinParams &tp, hipStream_t st) {
    switch (tp.c.Wi) {
        case 64: return launch_row_t<4, CH, TCONV>(tp, st);
        case 96: return launch_row_t<6, CH, TCONV>(tp, st);
        case 128: return launch_row_t<8, CH, TCONV>(tp, st);
        case 160: return launch_row_t<10, CH, TCONV>(tp, st);
        case 192: return launch_row_t<12, CH, TCONV>(tp, st);
    }
    return -1;
}

template <int NBLK>
int launch_row_stem_t(ThinParams tp, hipStream_t st) {
    ConvParams &p = tp.c;
    int SH;
    const int strips = pick_strips(p.Hi, SH);
    const int total = p.N * p.Di * strips;
    constexpr int W = 16 * NBLK;
    const size_t lds = (size_t)12 * (W + 2) * 32 + (size_t)2 * 6 * W * 8 + 4 * 16 * 2 * 8;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)conv_row_stem_kernel<NBLK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    int per_cu = (int)((160 * 1024) / lds);
    int cap = NBLK <= 8 ? 3 : 2;
    static const int wpc_knob = fnn_knob("FNN_ROW_WPC") ? atoi(fnn_knob("FNN_ROW_WPC")) : 0;      // A-B aid (launch_row_t)
    if (wpc_knob > 0 && wpc_knob < cap) cap = wpc_knob;
    if (per_cu > cap) per_cu = cap;
    int gx = 256 * per_cu;
    if (gx > total) gx = total;
    fnn_note_kernel("conv_row_stem_kernel<%d>", NBLK);
    hipLaunchKernelGGL((conv_row_stem_kernel<NBLK>), dim3(gx), dim3(256), lds, st, tp, total, strips, SH);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_row_stem(const ThinParams &tp, hipStream_t st) {
    switch (tp.c.Wi) {
        case 64: return launch_row_stem_t<4>(tp, st);
        case 96: return launch_row_stem_t<6>(tp, st);
        case 128: return launch_row_stem_t<8>(tp, st);
        case 160: return launch_row_stem_t<10>(tp, st);
        case 192: return launch_row_stem_t<12>(tp, st);
    }
    return -1;
}

}  // namespace

// Can the row kernel run this layer?  tp.fuse = 0 (plain sources), FUSE_TCONV or FUSE_STEM (the producer then is a
// one-channel (1, 3, 3) stem whose statistics pass is stem_row_kernel: engine.hip asks stem_row_ok too).
bool conv_row_ok(const ThinParams &tp) {
    const bool off = fnn_knob("FNN_NO_ROW") != nullptr;                              // A-B aid (read per call: tests toggle it)
    const ConvParams &p = tp.c;
    if (off || p.Cout != 16 || p.kd != 1 || p.kh != 3 || p.kw != 3 || p.sd != 1 || p.sh != 1 || p.sw != 1 || p.fp8) return false;
    if (p.Di != p.Do || p.Hi != p.Ho || p.Wi != p.Wo || p.packing != FNN_PACK_LINEAR || p.ksteps != 5) return false;
    if ((p.Wi != 64 && p.Wi != 96 && p.Wi != 128 && p.Wi != 160 && p.Wi != 192) || p.Hi % 4 != 0 || p.Hi < 8) return false;
    if (p.stats_out && p.stats_slots != FNN_STAT_REPL) return false;
    if (tp.fuse == FUSE_TCONV) {
        if (p.n_src != 2 || p.chunks != 2 || p.src[1].C != 16 || tp.low.C != 32) return false;
        if (tp.tsd != 1 || tp.tsh != 2 || tp.tsw != 2 || tp.Dl != p.Di || tp.Hl * 2 != p.Hi || tp.Wl * 2 != p.Wi) return false;
        return true;
    }
    if (tp.fuse == FUSE_STEM)
        return fnn_knob("FNN_NO_STEM_ROW") == nullptr && p.n_src == 1 && p.chunks == 1 && p.src[0].C == 16 && tp.Y * tp.Z < (1ll << 30);
    if (tp.fuse != 0) return false;
    if (p.chunks != p.n_src || p.chunks < 1 || p.chunks > 2) return false;
    for (int i = 0; i < p.n_src; ++i) if (p.src[i].C != 16) return false;
    return true;
}

int launch_conv_row(const ThinParams &tp, hipStream_t st) {
    if (!conv_row_ok(tp)) return -1;
    if (tp.fuse == FUSE_TCONV) return launch_row_n<2, true>(tp, st);
    if (tp.fuse == FUSE_STEM) return launch_row_stem(tp, st);
    return tp.c.chunks == 1 ? launch_row_n<1, false>(tp, st) : launch_row_n<2, false>(tp, st);
}

// stem in row form: one input channel, (1, 3, 3), 16 output channels, rows of 64 / 96 / 128 voxels
bool stem_row_ok(const StemParams &p) {
    const bool off = fnn_knob("FNN_NO_ROW") != nullptr || fnn_knob("FNN_NO_STEM_ROW") != nullptr;            // A-B aids
    if (off || p.C != 1 || p.kd != 1 || p.kh != 3 || p.kw != 3 || p.Cout != 16) return false;
    if ((p.PW != 64 && p.PW != 96 && p.PW != 128 && p.PW != 160 && p.PW != 192) || p.PH % 4 != 0 || p.PH < 8) return false;
    if (2ull * p.PD * p.PH * p.PW * 16 >= (1ull << 31) || p.Y * p.Z >= (1ll << 30)) return false;     // 32-bit output and in-plane input offsets
    int SH;
    const int strips = pick_strips(p.PH, SH);
    return p.PD * strips <= stem_mfma_stats_slots(p.PD, p.PH, p.PW);     // one statistics row per (plane, strip)
}

int launch_stem_row(const StemParams &p, int N, hipStream_t st) {
    if (!stem_row_ok(p)) return -1;
    int SH;
    const int strips = pick_strips(p.PH, SH);
    const int total = N * p.PD * strips, slots = stem_mfma_stats_slots(p.PD, p.PH, p.PW);
    const size_t lds = (size_t)12 * p.PW * 8 + 4 * 16 * 2 * 4;
    static const int wpc = fnn_knob("FNN_STEM_WPC") ? atoi(fnn_knob("FNN_STEM_WPC")) : 8;          // A-B aid: workgroups per CU
    int gx = 256 * wpc;
    if (gx > total) gx = total;
    // p.out == nullptr: the statistics pass of a fused consumer (conv_row_stem_kernel) - no stores, no 16-byte shuffles
#define STEM_ROW_LAUNCH(NB)                                                                                            \
    do {                                                                                                               \
        fnn_note_kernel("stem_row_kernel<%d,%d>", NB, p.out ? 1 : 0);                                                  \
        if (p.out) hipLaunchKernelGGL((stem_row_kernel<NB, true>), dim3(gx), dim3(256), lds, st, p, total, strips, SH, slots);    \
        else hipLaunchKernelGGL((stem_row_kernel<NB, false>), dim3(gx), dim3(256), lds, st, p, total, strips, SH, slots);         \
    } while (0)
    switch (p.PW) {
        case 64: STEM_ROW_LAUNCH(4); break;
        case 96: STEM_ROW_LAUNCH(6); break;
        case 128: STEM_ROW_LAUNCH(8); break;
        case 160: STEM_ROW_LAUNCH(10); break;
        default: STEM_ROW_LAUNCH(12); break;
    }
#undef STEM_ROW_LAUNCH
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
