// conv3d_s2.hip - 3x3x3 stride-(2,2,2) Conv3d with all 64 output channels of a group per staged halo (gfx950).
//
// A stride-2 conv reads 8+ input voxels per output voxel: per 16-channel chunk the halo of a 4 x 8 x 8 output tile is
// 9 x 17 x 17 voxels (83 KB) - staging, not the matrix cores, is what such a layer costs.  The linear-tap kernels
// (conv3d_persist_kernel<2, 2, ...>) stage a 5 x 17 x 17 halo per 2 x 8 x 8 tile and 32 output channels: once per cout
// group (2x / 4x for 64 / 128 channels), every operand fragment feeds two MFMAs, every weight fragment two - the
// k-loop is bound by LDS reads (1.0 per MFMA, the activation reads 2-way bank conflicted) and the layer by staging:
// 7 - 12 % of the MFMA peak.  Here:
//   * 512 threads, one workgroup per CU, tile 4 x 8 x 8, FOUR cout blocks (64 channels) per staged halo chunk: the halo
//     of a 64-channel layer is staged once, of a 128-channel layer twice; 1.16x fewer halo voxels per output than the
//     2-deep tile;
//   * wave = (depth slice of the tile, pair of cout blocks): 4 operand + 2 weight fragment reads per 8 MFMAs (0.75);
//   * LDS image with the columns de-interleaved (even ones first): the 8 voxels of an operand row are contiguous for
//     every tap.  ds_read_b128 serves a wave in four groups of 16 lanes - {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and
//     the same + 32 - over 64 banks (256 B): a group holds voxels 0-3 of one row and 4-7 of the other in one channel
//     half and the complementary quarters in the other half.  With the two rows of a column block (two input rows
//     apart) 128 B (mod 256) apart - row pitch 18 voxels - and the channel halves swapped where bit 1 of the input row
//     is set, the 16 lanes of every group hit 16 different 16-byte slots: conflict free (pitch 17: 2-way);
//   * persistent: a workgroup walks a contiguous range of (tile, cout group) units; the global loads of the next
//     (unit, chunk) item - halo elements, weight fragments, scale / shift - are in flight during the MFMAs of the
//     current one; statistics in LDS doubles, flushed when the batch item changes.
// Same arithmetic, weight packing (FNN_PACK_LINEAR, 14 k-steps) and epilogue as the other conv kernels.
//
// Replaces the strided ConvDropoutNormReLU of the reference's PlainConvEncoder stages
// (nnUNetDistillationTrainer.py:141-173).
#include "fnn_device.h"
#include "conv_common.h"
#include <cstdlib>

namespace {

constexpr int S2_ID = 9, S2_IH = 17, S2_IW = 17;
constexpr int S2_PW = 18;                                               // LDS row pitch (voxels): see the bank note below
constexpr int S2_ABYTES = S2_ID * S2_IH * S2_PW * 32;                   // 88128
constexpr int S2_KS = 14, S2_NB = 4;
constexpr int S2_WBYTES = S2_NB * S2_KS * 1024;                         // 57344
constexpr int S2_WPF = (S2_NB * S2_KS * 64) / 512;                      // 7 weight elements

static __device__ __forceinline__ int s2_pos(int zw) { return (zw & 1) ? 9 + (zw >> 1) : (zw >> 1); }

// SH: rows / columns of a halo plane that are STAGED (17: a full 8 x 8 output tile; 13: a layer whose output plane is at most
// 6 x 6 - one tile per plane whose rows and columns 13 .. 16 nobody needs); MBV: the column blocks (pairs of output rows) of a
// depth slice that hold output rows of the layer (4, or 3 for planes of at most 6 rows).  The LDS image keeps its geometry.
template <int SH, int MBV>
__global__ __launch_bounds__(512, 1) void conv3d_s2_kernel(const ConvParams p, const int total_units, const int groups) {
    constexpr int S2_IVOX = S2_ID * SH * SH;                             // staged halo voxels per chunk (2601 / 1521)
    constexpr int S2_PF = (S2_IVOX * 2 + 511) / 512;                     // 11 / 6 halo elements per thread and chunk
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4, hl = lane >> 5, kh = q & 1;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bg = wave >> 1, cp = wave & 1;                            // depth slice of the tile, pair of cout blocks
    FNN_STAMP_DECL
    char *sA = smem;
    char *sW = smem + S2_ABYTES;                                        // [4 cout blocks][14][64 lanes][16 B]
    double *sAcc = (double *)(sW + S2_WBYTES);                          // [64 channels][2]: statistics of the current batch item
    float *sRed = (float *)(sAcc + 128);                                // [8 waves][32][2]

    int u_begin, u_end;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int qd = nwg >> 3, rm = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        const int g = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
        u_begin = (int)((long long)total_units * g / nwg);
        u_end = (int)((long long)total_units * (g + 1) / nwg);
    }
    if (u_begin >= u_end) return;

    // ---- one-time set-up
    if (tid < 128) sAcc[tid] = 0.0;
    const int cg = tid & 1;
    // this thread's halo elements: element u = (zd, zh, zw, half) of index tid + 512 u (the last one exists only for
    // idx < 2 IVOX).  Kept per element: its place in the LDS image in 16-byte units, two per register (round 5: commit() used
    // to rebuild it per item from a packed coordinate triple - 15 of the ~35 vector instructions an element cost there; the
    // coordinates themselves are only needed once per unit, in set_offsets(), and are re-derived there from the index)
    unsigned ldp[(S2_PF + 1) / 2];
    const auto elem_coords = [&](int u, int &zd, int &zh, int &zw) {
        const int idx = tid + u * 512, v = (idx < S2_IVOX * 2 ? idx : S2_IVOX * 2 - 1) >> 1;
        zd = v / (SH * SH);
        const int rem = v - zd * (SH * SH);
        zh = rem / SH; zw = rem - zh * SH;
    };
#pragma unroll
    for (int u = 0; u < S2_PF; ++u) {
        int zd, zh, zw;
        elem_coords(u, zd, zh, zw);
        const unsigned l = (unsigned)((__mul24(__mul24(zd, S2_IH) + zh, S2_PW) + s2_pos(zw)) * 2 + (cg ^ ((zh >> 1) & 1)));
        if (u & 1) ldp[u >> 1] |= l << 16; else ldp[u >> 1] = l;
    }
    const bool has_last = tid + (S2_PF - 1) * 512 < S2_IVOX * 2;
    // operand reads: k-step ks = taps (2 ks, 2 ks + 1), d-major; lane (voxel r of the block's two rows, tap hl, half kh);
    // block mb = rows 2 mb, 2 mb + 1 of depth slice bg: + mb * 4 input rows
    int lanec[S2_KS];
#pragma unroll
    for (int ks = 0; ks < S2_KS; ++ks) {
        const int t = 2 * ks + hl < 27 ? 2 * ks + hl : 26;               // padded slot: any finite data (its weights are 0)
        const int td = t / 9, th = (t / 3) % 3, tw = t % 3;
        const int zd = 2 * bg + td, zh = 2 * (r >> 3) + th, zw = 2 * (r & 7) + tw;
        lanec[ks] = ((zd * S2_IH + zh) * S2_PW + s2_pos(zw)) * 32 + ((kh ^ ((zh >> 1) & 1)) * 16);
    }
    constexpr int MB_STEP = 4 * S2_PW * 32;                              // block mb + 1: four input rows further

    const int per_cb = S2_KS * 64;                                       // 16-byte weight elements per (cout block, chunk)
    // weight element tid + 512 u of a chunk = (cout block cbl, rem): its global element (cbl * chunks) * per_cb + rem
    // = idx + cbl * (chunks - 1) * per_cb, computed where it is used (an array would cost 7 registers)
    const int wskip = (p.chunks - 1) * per_cb;

    auto unit_coords = [&](int u, int &n, int &od0, int &oh0, int &ow0, int &grp) {
        grp = u % groups; int t = u / groups;
        const int tw = t % p.tiles_w; t /= p.tiles_w;
        const int th = t % p.tiles_h; t /= p.tiles_h;
        const int td = t % p.tiles_d;
        n = t / p.tiles_d;
        od0 = td * 4; oh0 = th * 8; ow0 = tw * 8;
    };

    int offv[S2_PF];
    f16x8 xr[S2_PF], wr[S2_WPF];
    float4 scr[2], shr[2];
    float slope_next = 1.f;
    auto set_offsets = [&](int od0, int oh0, int ow0) {
        const int id0 = 2 * od0 - 1, ih0 = 2 * oh0 - 1, iw0 = 2 * ow0 - 1;
#pragma unroll
        for (int u = 0; u < S2_PF; ++u) {
            int zd, zh, zw;
            int tv = tid;
            asm volatile("" : "+v"(tv));                                 // keep the coordinates out of loop-invariant registers (they spill)
            {
                const int idx = tv + u * 512, v = (idx < S2_IVOX * 2 ? idx : S2_IVOX * 2 - 1) >> 1;
                zd = v / (SH * SH);
                const int rem = v - zd * (SH * SH);
                zh = rem / SH; zw = rem - zh * SH;
            }
            const unsigned gd = (unsigned)(id0 + zd), gh = (unsigned)(ih0 + zh), gw = (unsigned)(iw0 + zw);
            const bool ok = gd < (unsigned)p.Di && gh < (unsigned)p.Hi && gw < (unsigned)p.Wi;
            offv[u] = ok ? (int)(__umul24(__umul24(gd, (unsigned)p.Hi) + gh, (unsigned)p.Wi) + gw) : -1;
        }
    };
    // The next item's global loads are issued INSIDE the k-loop, one or two per k-step: 22 wave-wide 16-byte loads per wave
    // and item keep the CU's texture-address path busy for ~2.8 k cycles (stamps), during which - issued in one block in
    // front of the k-loop - no wave reached its MFMAs.
    const char *i_sp = nullptr;
    const f16x8 *i_wp = nullptr;
    const float *i_qs = nullptr, *i_qh = nullptr;
    int i_sc2 = 0, i_nbv = S2_NB;
    const int nblk_all = p.Cout >> 4;
    auto issue_setup = [&](int n, int grp, int ch) {
        i_nbv = nblk_all - grp * S2_NB < S2_NB ? nblk_all - grp * S2_NB : S2_NB;
        const int c_glob = ch * 16;
        const int s = (c_glob < p.src[0].C) ? 0 : 1;
        const int c_loc = c_glob - (s ? p.src[0].C : 0) + cg * 8;
        const int sC = p.src[s].C;
        i_sp = (const char *)(p.src[s].ptr + (size_t)n * p.Di * p.Hi * p.Wi * sC + (c_loc >> 4) * FNN_CS(p.src[s]) + (c_loc & 15));
        i_sc2 = FNN_VS(p.src[s]) * 2;                                    // activation layout: fnn_device.h, SrcDesc
        slope_next = p.src[s].slope;
        i_qs = p.src[s].ss ? p.src[s].ss + (size_t)(2 * n) * sC + c_loc : p.ident_ss + c_loc;
        i_qh = p.src[s].ss ? i_qs + sC : p.ident_ss + 512 + c_loc;
        i_wp = (const f16x8 *)p.wpk + (size_t)(grp * S2_NB * p.chunks + ch) * per_cb;
    };
    auto issue_part = [&](int ks) {                                      // k-step 0: scale / shift; 1 ..: halo element ks - 1 and weight element ks - 1
        if (ks == 0) {
            scr[0] = *(const float4 *)i_qs; scr[1] = *(const float4 *)(i_qs + 4);
            shr[0] = *(const float4 *)i_qh; shr[1] = *(const float4 *)(i_qh + 4);
        } else {
            const int u = ks - 1;                                        // unconditional: branches around loads make hipcc drain vmcnt
            if (u < S2_PF) xr[u] = *(const f16x8 *)(i_sp + __umul24((unsigned)(offv[u] >= 0 ? offv[u] : 0), (unsigned)i_sc2));   // voxels < 2^24 (launcher)
            if (u < S2_WPF) {
                const int idx = tid + u * 512, cbl = (idx * 74899) >> 26;  // idx / 896 for idx < 3584
                // (a last group of fewer than four cout blocks: the missing blocks re-read block 0 - finite values nobody stores)
                wr[u] = i_wp[cbl < i_nbv ? idx + cbl * wskip : idx - cbl * per_cb];
            }
        }
    };
    auto issue = [&](int n, int grp, int ch) {                          // all at once (prologue)
        issue_setup(n, grp, ch);
#pragma unroll
        for (int ks = 0; ks <= (S2_PF > S2_WPF ? S2_PF : S2_WPF); ++ks) issue_part(ks);
    };
    auto commit = [&]() {
        const f16 slope_h = (f16)slope_next;
        const float sc[8] = {scr[0].x, scr[0].y, scr[0].z, scr[0].w, scr[1].x, scr[1].y, scr[1].z, scr[1].w};
        const float sh[8] = {shr[0].x, shr[0].y, shr[0].z, shr[0].w, shr[1].x, shr[1].y, shr[1].z, shr[1].w};
#ifndef FNN_NORM_FP32
        f16x8 sc_h, sh_h;
#pragma unroll
        for (int j = 0; j < 8; ++j) { sc_h[j] = (f16)sc[j]; sh_h[j] = (f16)sh[j]; }
#endif
#pragma unroll
        for (int u = 0; u < S2_PF; ++u) {
#ifdef FNN_NORM_FP32
            f16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (f16)fmaf((float)xr[u][j], sc[j], sh[j]);
#else
            f16x8 o = xr[u] * sc_h + sh_h;
#endif
            o = __builtin_elementwise_max(o, o * slope_h);
            if (offv[u] < 0) o = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};       // the conv's zero padding
            const unsigned ldso = ((u & 1) ? ldp[u >> 1] >> 16 : ldp[u >> 1] & 0xffffu) << 4;
            if (u + 1 < S2_PF || has_last) *(f16x8 *)(sA + ldso) = o;
        }
#pragma unroll
        for (int u = 0; u < S2_WPF; ++u) ((f16x8 *)sW)[tid + u * 512] = wr[u];
    };

    f32x4 acc[4][2];
    // k-loop: the fragments of k-step ks + 1 are requested before the MFMAs of ks (two register sets): with the reads of
    // a k-step issued and awaited in front of its MFMAs a wave spent ~330 cycles per k-step for 128 cycles of MFMA, and
    // two waves per SIMD did not cover each other
    auto read_frags = [&](int ks, f16x8 (&xf)[4], f16x8 (&wf)[2]) {
#pragma unroll
        for (int mb = 0; mb < MBV; ++mb) xf[mb] = *(const f16x8 *)(sA + lanec[ks] + mb * MB_STEP);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) wf[nb] = *(const f16x8 *)(sW + (((2 * cp + nb) * S2_KS + ks) * 64 + lane) * 16);
    };
    auto mfmas = [&](const f16x8 (&xf)[4], const f16x8 (&wf)[2]) {
#pragma unroll
        for (int mb = 0; mb < MBV; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nb], xf[mb], acc[mb][nb], 0, 0, 0);
    };
    auto kloop = [&]() {
        f16x8 xa[4], wa[2], xb[4], wb[2];
        read_frags(0, xa, wa);
#pragma unroll
        for (int ks = 0; ks < S2_KS; ks += 2) {
            read_frags(ks + 1, xb, wb);
            __builtin_amdgcn_sched_barrier(0);                           // all six reads leave before the first MFMA (hipcc sinks them otherwise)
            issue_part(ks);
            mfmas(xa, wa);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 2 < S2_KS) read_frags(ks + 2, xa, wa);
            __builtin_amdgcn_sched_barrier(0);
            issue_part(ks + 1);
            mfmas(xb, wb);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // statistics of one unit: fp32 inside the tile (as the other kernels), double in LDS across the units of a batch item
    auto unit_stats = [&](float (&t1)[2][4], float (&t2)[2][4]) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = row16_sum(t1[nb][j]), b = row16_sum(t2[nb][j]);
                if (r == 0) {
                    const int c = nb * 16 + q * 4 + j;
                    sRed[(wave * 32 + c) * 2] = a;
                    sRed[(wave * 32 + c) * 2 + 1] = b;
                }
            }
    };
    auto fold_stats = [&]() {                                            // after a barrier: 8 waves -> 64 channels x 2 doubles
        if (tid < 128) {
            const int c = tid >> 1, which = tid & 1, cpx = c >> 5, cl = c & 31;
            double v = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) v += (double)sRed[((2 * b + cpx) * 32 + cl) * 2 + which];
            sAcc[tid] += v;
        }
    };
    auto flush_stats = [&](int n, int grp) {                            // the tid < 128 threads own sAcc[tid]: no barrier needed
        if (p.stats_out && tid < 128) {
            const int c = tid >> 1, which = tid & 1;
            if (grp * 64 + c < p.Cout)
            unsafeAtomicAdd(p.stats_out + (((size_t)n * FNN_STAT_REPL + (blockIdx.x & (FNN_STAT_REPL - 1))) * p.Cout + grp * 64 + c) * 2 + which, sAcc[tid]);
            sAcc[tid] = 0.0;
        }
    };

    // ---- the item stream
    int n_cur, od0, oh0, ow0, grp;
    unit_coords(u_begin, n_cur, od0, oh0, ow0, grp);
    set_offsets(od0, oh0, ow0);
    issue(n_cur, grp, 0);
    __syncthreads();                                                     // sAcc
    commit();
    __syncthreads();
    for (int u = u_begin; u < u_end; ++u) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        int n_nx = n_cur, d_nx = od0, h_nx = oh0, w_nx = ow0, g_nx = grp;
        for (int ch = 0; ch < p.chunks; ++ch) {
            const bool last = ch + 1 == p.chunks;
#ifdef FNN_STAMPS
            const bool stamp_it = u == u_begin + 1;
            if (stamp_it) FNN_STAMP();                                   // item start
#endif
            if (last) {
                // the very last item prefetches itself again: issue / commit stay on an unconditional path
                if (u + 1 < u_end) unit_coords(u + 1, n_nx, d_nx, h_nx, w_nx, g_nx);
                set_offsets(d_nx, h_nx, w_nx);
                issue_setup(n_nx, g_nx, u + 1 < u_end ? 0 : ch);
            } else {
                issue_setup(n_cur, grp, ch + 1);
            }
#ifdef FNN_STAMPS
            if (stamp_it) FNN_STAMP();                                   // next item's addresses ready
#endif
            kloop();
#ifdef FNN_STAMPS
            if (stamp_it) FNN_STAMP();                                   // k-loop done
#endif
            if (last) {
                float t1[2][4], t2[2][4];
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int j = 0; j < 4; ++j) { t1[nb][j] = 0.f; t2[nb][j] = 0.f; }
                if (grp * 4 + 2 * cp < nblk_all) {                       // (a short last group: this wave's pair of cout blocks may not exist)
                    float4 bv[2];
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) bv[nb] = *(const float4 *)(p.bias + (grp * 4 + 2 * cp + nb) * 16 + q * 4);
                    tile_epilogue<2, 4>(p, acc, bv, n_cur, od0, oh0, ow0, grp * 4 + 2 * cp, bg, lane, t1, t2);
                }
                unit_stats(t1, t2);
            }
            __syncthreads();                                             // every wave is done reading this item; sRed complete
#ifdef FNN_STAMPS
            if (stamp_it) FNN_STAMP();                                   // (epilogue +) barrier
#endif
            if (last) {
                fold_stats();
                if (n_nx != n_cur || g_nx != grp || u + 1 == u_end) flush_stats(n_cur, grp);
            }
            commit();
#ifdef FNN_STAMPS
            if (stamp_it) FNN_STAMP();                                   // next item staged
#endif
            __syncthreads();
#ifdef FNN_STAMPS
            if (stamp_it) FNN_STAMP();                                   // barrier
#endif
        }
        n_cur = n_nx; od0 = d_nx; oh0 = h_nx; ow0 = w_nx; grp = g_nx;
    }
    FNN_STAMP_FLUSH(p.dbg);
}

}  // namespace

bool conv3d_s2_ok(const ConvParams &p) {
    static const bool off = fnn_knob("FNN_NO_S2") != nullptr;                        // A-B aid
    if (off || p.kd != 3 || p.kh != 3 || p.kw != 3 || p.sd != 2 || p.sh != 2 || p.sw != 2 || p.fp8) return false;
    // whole groups of 64 output channels, or (round 5) a last group of 32: 160 = 64 + 64 + 32
    if (p.packing != FNN_PACK_LINEAR || p.ksteps != S2_KS || p.Cout % 32 != 0 || p.Cout < 64) return false;
    if (p.stats_out && p.stats_slots != FNN_STAT_REPL) return false;
    if ((long long)p.Di * p.Hi * p.Wi >= (1 << 24)) return false;                  // 24-bit voxel index arithmetic in the kernel
    for (int i = 0; i < p.n_src; ++i)
        if (2ull * p.Di * p.Hi * p.Wi * p.src[i].C >= (1ull << 32)) return false;
    if (2ull * p.Do * p.Ho * p.Wo * p.Cout >= (1ull << 31)) return false;
    // enough units to give every CU a few: below that the 2 x 8 x 8 kernels' many small workgroups win
    const int plan_n = p.plan_N > 0 ? p.plan_N : p.N;
    const long long units = (long long)plan_n * ((p.Do + 3) / 4) * ((p.Ho + 7) / 8) * ((p.Wo + 7) / 8) * ((p.Cout + 63) / 64);
    // (round 5: 384 instead of 768 - the 128 -> 160 layer at 20 x 6 x 6, 480 units, 170 -> 96 us against the 2 x 8 x 8 kernel)
    static const int min_units = fnn_knob("FNN_S2_MIN_UNITS") ? atoi(fnn_knob("FNN_S2_MIN_UNITS")) : 384;       // A-B aid
    return units >= min_units;
}

int launch_conv3d_s2(ConvParams p, hipStream_t st) {
    if (!conv3d_s2_ok(p)) return -1;
    p.tile_d = 4;
    p.tiles_d = (p.Do + 3) / 4; p.tiles_h = (p.Ho + 7) / 8; p.tiles_w = (p.Wo + 7) / 8;
    p.ident_ss = conv3d_identity_ss();
    if (!p.ident_ss) return -2;
    const int groups = (p.Cout + 63) / 64;                              // the last group may hold two cout blocks instead of four
    const int total = p.N * p.tiles_d * p.tiles_h * p.tiles_w * groups;
    const size_t lds = (size_t)S2_ABYTES + S2_WBYTES + 128 * 8 + 8 * 32 * 2 * 4;
    static const int wgs_knob = fnn_knob("FNN_S2_WGS") ? atoi(fnn_knob("FNN_S2_WGS")) : 256;        // A-B aid: fewer persistent workgroups (CUs left to another stream's kernels)
    const int gx = total < wgs_knob ? total : wgs_knob;
    const bool small = p.Ho <= 6 && p.Wo <= 6;                         // one tile per plane, input planes of at most 13 x 13 (with the padding)
    static bool attr_set[2] = {false, false};
    if (!attr_set[small]) {
        if (small) (void)hipFuncSetAttribute((const void *)conv3d_s2_kernel<13, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        else (void)hipFuncSetAttribute((const void *)conv3d_s2_kernel<17, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set[small] = true;
    }
    fnn_note_kernel(small ? "conv3d_s2_kernel<13,3>" : "conv3d_s2_kernel");
    if (small) hipLaunchKernelGGL((conv3d_s2_kernel<13, 3>), dim3(gx), dim3(512), lds, st, p, total, groups);
    else hipLaunchKernelGGL((conv3d_s2_kernel<17, 4>), dim3(gx), dim3(512), lds, st, p, total, groups);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
