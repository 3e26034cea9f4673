// conv3d_zr.hip - 3x3x3 stride-1 Conv3d with depth-shift operand reuse (gfx950).
//
// Same GEMM view and fusions as conv3d_lds_kernel (conv3d.hip): D[cout, voxel] = W[cout, k] X[k, voxel]
// on v_mfma_f32_16x16x32_f16, halo tile of one 16-channel chunk + that chunk's weight fragments in LDS,
// producer InstanceNorm + LeakyReLU applied while staging, own statistics in the epilogue.  What changes
// is who owns which voxels: a wave owns two h rows (16 voxels = one MFMA column block) in EVERY depth
// slice of the TD x 8 x 8 output tile.  The "B" fragment of halo plane p for an in-plane tap pair is then
// the operand of depth tap dz for output slice p - dz, for all three dz: it is read from LDS once and
// feeds 3 MFMAs per cout block.  Per (chunk, tap pair) a wave issues TD + 2 activation reads and 3 NB weight
// reads for 3 TD NB MFMAs - 0.33 LDS reads per MFMA at NB = 2, TD = 8 instead of 0.63 in the linear-tap
// kernels, which at 32 output channels were bound by LDS bandwidth, not by the matrix cores.
// Cost: the 9 in-plane taps pair up into 5 k-steps (one half padded) -> 30 tap slots instead of 28.
//
// Replaces the same reference code as conv3d.hip (ConvDropoutNormReLU stacks,
// nnUNetDistillationTrainer.py:141-173).
#include "fnn_device.h"
#include "conv_common.h"
#include <cstdlib>

int conv3d_ksteps(int packing, int taps) { return packing == FNN_PACK_ZR ? 15 : (taps + 1) / 2; }

int conv3d_kstep_tap(int packing, int ks, int half, int taps) {
    if (packing == FNN_PACK_ZR) {
        const int pr = ks / 3, dz = ks % 3, t = 2 * pr + half;
        return t < 9 ? dz * 9 + t : -1;
    }
    const int t = 2 * ks + half;
    return t < taps ? t : -1;
}

// (cout blocks per workgroup, tile depth) the ZR kernel would run with, or false when the layer keeps the
// linear-tap kernels: not 3x3x3 / stride 1, or too few workgroups to fill the chip.
static bool zr_pick(const ConvParams &p, int &nb, int &td) {
    static const bool off = getenv("FNN_CONV_NO_ZR") != nullptr;                  // A-B aid
    static const int max_cout = getenv("FNN_ZR_MAX_COUT") ? atoi(getenv("FNN_ZR_MAX_COUT")) : 1 << 30;
    static const int min_cout = getenv("FNN_ZR_MIN_COUT") ? atoi(getenv("FNN_ZR_MIN_COUT")) : 0;
    if (off || p.kd != 3 || p.kh != 3 || p.kw != 3 || p.sd != 1 || p.sh != 1 || p.sw != 1) return false;
    if (p.Cout > max_cout || p.Cout < min_cout) return false;
    const int nblk = p.Cout / 16;
    nb = nblk % 2 == 0 ? 2 : 1;
    const int plan_n = p.plan_N > 0 ? p.plan_N : p.N;
    const long long th = (p.Ho + 7) / 8, tw = (p.Wo + 7) / 8;
    for (td = 8; td >= 4; td -= 4) {
        if (p.Do < td) continue;
        if ((long long)plan_n * ((p.Do + td - 1) / td) * th * tw * (nblk / nb) >= 768) return true;
    }
    return false;
}

int conv3d_packing(const ConvParams &p) {
    int nb, td;
    return zr_pick(p, nb, td) ? FNN_PACK_ZR : FNN_PACK_LINEAR;
}

template <int NB, int TD>
__global__ __launch_bounds__(256, 2) void conv3d_zr_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    FNN_STAMP_DECL
    FNN_STAMP();                                              // 0: entry
    constexpr int IH = 10, IW = 10, PW = 12, ID = TD + 2;    // halo tile, row pitch 12 = 4 (mod 8) voxels
    constexpr int PS = IH * PW * 32;                          // bytes per halo plane
    constexpr int ABYTES = (ID * PS + 1023) & ~1023;
    constexpr int KS = 15;
    constexpr int IELEM = ID * IH * IW * 2;                   // 16-byte halo elements per chunk
    constexpr int PF = (IELEM + 255) / 256;
    constexpr int WTOT = NB * KS * 64;                        // 16-byte weight elements per chunk
    constexpr int WPF = (WTOT + 255) / 256;

    // XCD-aware, bijective remap (blocks b and b + 8 share an XCD)
    int t;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int qd = nwg >> 3, rm = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        t = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
    }
    const int tw = t % p.tiles_w; t /= p.tiles_w;
    const int th = t % p.tiles_h; t /= p.tiles_h;
    const int td = t % p.tiles_d;
    const int n = t / p.tiles_d;
    const int cb0 = blockIdx.y * NB;
    const int od0 = td * TD, oh0 = th * 8, ow0 = tw * 8;

    char *sA = smem;                                          // halo image: [ID][IH][PW] voxels x 32 B, halves swapped on odd rows
    char *sW = smem + ABYTES;                                 // [NB][15][64 lanes][16 B]
    float *sBias = (float *)(sW + NB * KS * 1024);

    int toff[5];                                              // filled in after the first loads have left
    f32x4 acc[TD][NB];
    // this thread's share of the prefetch: PF halo elements (voxel, 8-channel half) + WPF weight elements
    const int cg = tid & 1;
    int offv[PF];                                             // global voxel index; -1 = zero padding, -2 = no element
    int ldso[PF];
    {
        const int id0 = od0 - 1, ih0 = oh0 - 1, iw0 = ow0 - 1;
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int idx = tid + u * 256;
            const int v = idx >> 1;
            const int zd = v / (IH * IW), rem = v - zd * (IH * IW), zh = rem / IW, zw = rem - zh * IW;
            const int gd = id0 + zd, gh = ih0 + zh, gw = iw0 + zw;
            const bool ok = gd >= 0 && gd < p.Di && gh >= 0 && gh < p.Hi && gw >= 0 && gw < p.Wi;
            offv[u] = idx < IELEM ? (ok ? ((n * p.Di + gd) * p.Hi + gh) * p.Wi + gw : -1) : -2;
            ldso[u] = zd * PS + (zh * PW + zw) * 32 + ((cg ^ (zh & 1)) * 16);
        }
    }
    int wofs[WPF];
#pragma unroll
    for (int u = 0; u < WPF; ++u) {
        const int idx = tid + u * 256;
        const int idc = idx < WTOT ? idx : WTOT - 1;
        const int nb = idc >= KS * 64 ? 1 : 0;                // NB <= 2
        wofs[u] = (cb0 + nb) * p.chunks * (KS * 64) + idc - nb * (KS * 64);
    }
    f16x8 xr[PF], wr[WPF];
    float4 scr[2], shr[2];
    float slope_next = 1.f;

    auto issue = [&](int ch) {
        const int c_glob = ch * 16;
        const int s = (c_glob < p.src[0].C) ? 0 : 1;
        const int c_uni = c_glob - (s ? p.src[0].C : 0);
        const int c_loc = c_uni + cg * 8;
        const int sC = p.src[s].C;
        // uniform 64-bit base (SGPRs) + per-lane 32-bit byte offset: one address VGPR per load (tensors < 4 GiB)
        const char *sp = (const char *)(p.src[s].ptr + c_uni);
#pragma unroll
        for (int u = 0; u < PF; ++u)                          // unconditional: branches around loads make hipcc drain vmcnt
            xr[u] = *(const f16x8 *)(sp + (unsigned)(((offv[u] >= 0 ? offv[u] : 0) * sC + cg * 8) * 2));
#pragma unroll
        for (int u = 0; u < WPF; ++u) wr[u] = *(const f16x8 *)((const char *)p.wpk + (unsigned)((wofs[u] + ch * (KS * 64)) * 16));
        slope_next = p.src[s].slope;
        if (p.src[s].ss) {
            const float *q4 = p.src[s].ss + (size_t)(2 * n) * sC + c_loc;
            scr[0] = *(const float4 *)q4; scr[1] = *(const float4 *)(q4 + 4);
            shr[0] = *(const float4 *)(q4 + sC); shr[1] = *(const float4 *)(q4 + sC + 4);
        } else {
            scr[0] = scr[1] = make_float4(1.f, 1.f, 1.f, 1.f);
            shr[0] = shr[1] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto commit = [&]() {
        const f16 slope_h = (f16)slope_next;
        const float sc[8] = {scr[0].x, scr[0].y, scr[0].z, scr[0].w, scr[1].x, scr[1].y, scr[1].z, scr[1].w};
        const float sh[8] = {shr[0].x, shr[0].y, shr[0].z, shr[0].w, shr[1].x, shr[1].y, shr[1].z, shr[1].w};
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            if ((u + 1) * 256 > IELEM && offv[u] == -2) continue;
            f16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (f16)fmaf((float)xr[u][j], sc[j], sh[j]);
            o = __builtin_elementwise_max(o, o * slope_h);
            if (offv[u] < 0) o = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};      // the conv's zero padding
            *(f16x8 *)(sA + ldso[u]) = o;
        }
#pragma unroll
        for (int u = 0; u < WPF; ++u) {
            const int idx = tid + u * 256;
            if ((u + 1) * 256 <= WTOT || idx < WTOT) ((f16x8 *)sW)[idx] = wr[u];
        }
    };
    auto kloop = [&]() {
#pragma unroll
        for (int pr = 0; pr < 5; ++pr) {
            const char *bp = sA + toff[pr];
            f16x8 xf[ID];
#pragma unroll
            for (int pl = 0; pl < ID; ++pl) xf[pl] = *(const f16x8 *)(bp + pl * PS);
#pragma unroll
            for (int dz = 0; dz < 3; ++dz) {
                f16x8 wf[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) wf[nb] = *(const f16x8 *)(sW + ((nb * KS + pr * 3 + dz) * 64 + lane) * 16);
#pragma unroll
                for (int j = 0; j < TD; ++j)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[j][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nb], xf[j + dz], acc[j][nb], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);                // keep the next pair's reads from being hoisted: registers
        }
    };

    FNN_STAMP();                                              // 1: prefetch coordinates done
    issue(0);
    __builtin_amdgcn_sched_barrier(0);                        // the loads leave first; the rest of the set-up runs under them
    FNN_STAMP();                                              // 2: first loads issued
    if (tid < NB * 16) sBias[tid] = p.bias[cb0 * 16 + tid];
    // MFMA "B" operand: lane = (voxel r of the wave's two rows, k-group): k-group bit 1 picks the tap of the pair,
    // bit 0 the 8-channel half
    {
        const int r = lane & 15, hl = lane >> 5, kh = (lane >> 4) & 1;
#pragma unroll
        for (int pr = 0; pr < 5; ++pr) {
            const int tp = 2 * pr + hl < 9 ? 2 * pr + hl : 8;   // padded slot: any finite data (its weights are 0)
            const int row = 2 * wave + (r >> 3) + tp / 3, col = (r & 7) + tp % 3;
            toff[pr] = (row * PW + col) * 32 + ((kh ^ (row & 1)) * 16);
        }
    }

#pragma unroll
    for (int j = 0; j < TD; ++j)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[j][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    commit();
    __syncthreads();
    FNN_STAMP();                                              // 3: first chunk staged
    // the last chunk is peeled off so that the wait for the prefetch sits on an unconditional path (see conv3d_lds_kernel)
    for (int ch = 0; ch + 1 < p.chunks; ++ch) {
        issue(ch + 1);                                        // global loads stay in flight during the MFMAs
        kloop();
        FNN_STAMP();                                          // k-loop done
        __syncthreads();                                      // every wave is done reading this chunk
        commit();
        __syncthreads();
        FNN_STAMP();                                          // next chunk staged
    }
    kloop();
    FNN_STAMP();
    __syncthreads();

    // ---- epilogue: bias, fp16 store, statistics
    {
        float4 bv[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) bv[nb] = *(const float4 *)(sBias + nb * 16 + (lane >> 4) * 4);
        float t1[NB][4], t2[NB][4];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int j = 0; j < 4; ++j) { t1[nb][j] = 0.f; t2[nb][j] = 0.f; }
        tile_epilogue<NB, TD, true>(p, acc, bv, n, od0, oh0, ow0, cb0, wave, lane, t1, t2);
        if (p.stats_out) stats_to_global<NB>(p, t1, t2, (float *)smem, n, cb0, wave, lane, tid);
    }
    FNN_STAMP();                                              // epilogue done
    FNN_STAMP_FLUSH(p.dbg);
}

// ----------------------------------------------------------------------------
// warp-specialised variant: 4 MFMA waves + 4 staging waves per workgroup
// ----------------------------------------------------------------------------
// conv3d_zr_kernel's k-loops run at the matrix-core rate but are ~40 % of a workgroup's life: the same four
// waves also wait for the halo, normalise it, write LDS and run the epilogue, and with two workgroups per CU the
// phases overlap only by chance.  Here a 512-thread workgroup owns a CU: waves 0-3 ("consumers") do nothing but
// ds_read + MFMA + the tile epilogue, waves 4-7 ("producers") prefetch, normalise and stage the NEXT work item
// (tile, 16-channel chunk) into the other half of a double-buffered LDS image (2 x (halo + weights) = 136 KB of the
// 160 KB).  One s_barrier per item hands a buffer over.  The workgroup walks a contiguous tile range, so the
// pipeline never drains between tiles.  Statistics: per consumer wave, fp32 within a tile, double across tiles
// (wave-private LDS slots), atomics per (wave, batch item).
template <int NB>
__global__ __launch_bounds__(512, 1) void conv3d_zrs_kernel(const ConvParams p, const int total_tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TD = 8;
    constexpr int IH = 10, IW = 10, PW = 12, ID = TD + 2;
    constexpr int PS = IH * PW * 32;
    constexpr int ABYTES = (ID * PS + 1023) & ~1023;
    constexpr int KS = 15;
    constexpr int WBYTES = NB * KS * 1024;
    constexpr int IELEM = ID * IH * IW * 2;
    constexpr int PF = (IELEM + 255) / 256;
    constexpr int WTOT = NB * KS * 64;
    constexpr int WPF = (WTOT + 255) / 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto bufp = [&](int b) -> char * { return smem + (b & 1) * (ABYTES + WBYTES); };   // buffer b: [halo image][weights]
    float *sBias = (float *)(smem + 2 * (ABYTES + WBYTES));
    double *sRed = (double *)(sBias + NB * 16);               // [4 consumer waves][NB * 16][2]
    const int cb0 = blockIdx.y * NB;

    int t_begin, t_end;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int qd = nwg >> 3, rm = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        const int g = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
        t_begin = (int)((long long)total_tiles * g / nwg);
        t_end = (int)((long long)total_tiles * (g + 1) / nwg);
    }
    if (t_begin >= t_end) return;
    const int n_items = (t_end - t_begin) * p.chunks;

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

    if (tid < NB * 16) sBias[tid] = p.bias[cb0 * 16 + tid];
    for (int i = tid; i < 4 * NB * 16 * 2; i += 512) sRed[i] = 0.0;
    __syncthreads();

    if (wave >= 4) {
        // =============================== producers ===============================
        const int ptid = tid - 256;
        const int cg = ptid & 1;
        const unsigned cg_b = (unsigned)cg * 16u;
        int rel[PF];                                          // LDS byte offset | zw << 16 | zh << 20 | zd << 24, -1 = none
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int idx = ptid + u * 256;
            const int v = idx >> 1;
            const int zd = v / (IH * IW), rem = v - zd * (IH * IW), zh = rem / IW, zw = rem - zh * IW;
            const int lo = zd * PS + (zh * PW + zw) * 32 + ((cg ^ (zh & 1)) * 16);
            rel[u] = idx < IELEM ? lo | (zw << 16) | (zh << 20) | (zd << 24) : -1;
        }
        const int wbase0 = cb0 * p.chunks * (KS * 64), wbase1 = (cb0 + 1) * p.chunks * (KS * 64) - KS * 64;
        // Two items in flight per producer thread: an item period (~4 k cycles of MFMAs) is shorter than a loaded-HBM
        // round trip, so the data of item i + 1 AND i + 2 is requested while item i multiplies.
        struct Set {
            int offv[PF];
            f16x8 xr[PF], wr[WPF];
            float4 scr[2], shr[2];
            float slope;
        } X0, X1;
        auto request = [&](Set &S, int n, int ch, int od0, int oh0, int ow0) {
            const int id0 = od0 - 1, ih0 = oh0 - 1, iw0 = ow0 - 1;
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int gd = id0 + (rel[u] >> 24), gh = ih0 + ((rel[u] >> 20) & 15), gw = iw0 + ((rel[u] >> 16) & 15);
                const bool ok = rel[u] >= 0 && (unsigned)gd < (unsigned)p.Di && (unsigned)gh < (unsigned)p.Hi &&
                                (unsigned)gw < (unsigned)p.Wi;
                S.offv[u] = ok ? (gd * p.Hi + gh) * p.Wi + gw : -1;
            }
            const int c_glob = ch * 16;
            const int s = (c_glob < p.src[0].C) ? 0 : 1;
            const int c_uni = c_glob - (s ? p.src[0].C : 0);
            const int sC = p.src[s].C;
            const char *sp = (const char *)(p.src[s].ptr + (size_t)n * p.Di * p.Hi * p.Wi * sC + c_uni);
#pragma unroll
            for (int u = 0; u < PF; ++u)
                S.xr[u] = *(const f16x8 *)(sp + ((unsigned)((S.offv[u] >= 0 ? S.offv[u] : 0) * sC * 2) + cg_b));
#pragma unroll
            for (int u = 0; u < WPF; ++u) {
                const int idx = ptid + u * 256;
                const int idc = idx < WTOT ? idx : WTOT - 1;
                const int wo = (idc >= KS * 64 ? wbase1 : wbase0) + idc + ch * (KS * 64);
                S.wr[u] = *(const f16x8 *)((const char *)p.wpk + (unsigned)(wo * 16));
            }
            S.slope = p.src[s].slope;
            const float *qs = p.src[s].ss ? p.src[s].ss + (size_t)(2 * n) * sC + c_uni + cg * 8 : p.ident_ss + c_uni + cg * 8;
            const float *qh = p.src[s].ss ? qs + sC : p.ident_ss + 512 + c_uni + cg * 8;
            S.scr[0] = *(const float4 *)qs; S.scr[1] = *(const float4 *)(qs + 4);
            S.shr[0] = *(const float4 *)qh; S.shr[1] = *(const float4 *)(qh + 4);
        };
        auto commit = [&](Set &S, char *buf) {
            const f16 slope_h = (f16)S.slope;
            const float sc[8] = {S.scr[0].x, S.scr[0].y, S.scr[0].z, S.scr[0].w, S.scr[1].x, S.scr[1].y, S.scr[1].z, S.scr[1].w};
            const float sh[8] = {S.shr[0].x, S.shr[0].y, S.shr[0].z, S.shr[0].w, S.shr[1].x, S.shr[1].y, S.shr[1].z, S.shr[1].w};
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                f16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (f16)fmaf((float)S.xr[u][j], sc[j], sh[j]);
                o = __builtin_elementwise_max(o, o * slope_h);
                if (S.offv[u] < 0) o = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
                if ((u + 1) * 256 <= IELEM || rel[u] >= 0) *(f16x8 *)(buf + (rel[u] & 0xffff)) = o;
            }
#pragma unroll
            for (int u = 0; u < WPF; ++u) {
                const int idx = ptid + u * 256;
                if ((u + 1) * 256 <= WTOT || idx < WTOT) ((f16x8 *)(buf + ABYTES))[idx] = S.wr[u];
            }
        };
        int n, od0, oh0, ow0, ch = 0;
        tile_coords(t_begin, n, od0, oh0, ow0);
        auto advance = [&]() {                                // the next item in (tile, chunk) order
            if (++ch == p.chunks) { ch = 0; next_tile(n, od0, oh0, ow0); }
        };
        request(X0, n, ch, od0, oh0, ow0);                    // item 0
        commit(X0, bufp(0));
        if (n_items > 1) { advance(); request(X1, n, ch, od0, oh0, ow0); }      // item 1
        if (n_items > 2) { advance(); request(X0, n, ch, od0, oh0, ow0); }      // item 2
#pragma unroll 1
        for (int it = 0; it < n_items; it += 2) {
            __syncthreads();                                  // item `it` is visible; buffer (it + 1) & 1 is free
            if (it + 1 < n_items) {
                commit(X1, bufp(1));                          // item it + 1
                if (it + 3 < n_items) { advance(); request(X1, n, ch, od0, oh0, ow0); }
            } else break;
            __syncthreads();                                  // item it + 1 visible; buffer 0 free
            if (it + 2 < n_items) {
                commit(X0, bufp(0));                          // item it + 2
                if (it + 4 < n_items) { advance(); request(X0, n, ch, od0, oh0, ow0); }
            }
        }
        return;
    }

    // ================================= consumers =================================
    FNN_STAMP_DECL
    int toff[5];
    {
        const int r = lane & 15, hl = lane >> 5, kh = (lane >> 4) & 1;
#pragma unroll
        for (int pr = 0; pr < 5; ++pr) {
            const int tp = 2 * pr + hl < 9 ? 2 * pr + hl : 8;
            const int row = 2 * wave + (r >> 3) + tp / 3, col = (r & 7) + tp % 3;
            toff[pr] = (row * PW + col) * 32 + ((kh ^ (row & 1)) * 16);
        }
    }
    f32x4 acc[TD][NB];
#pragma unroll
    for (int j = 0; j < TD; ++j)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[j][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 bv[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) bv[nb] = *(const float4 *)(sBias + nb * 16 + (lane >> 4) * 4);
    double *myRed = sRed + wave * NB * 16 * 2;
    auto flush_stats = [&](int n) {                           // wave-private slots: no workgroup barrier needed
        if (!p.stats_out) return;
        if (lane < NB * 16 * 2) {
            const int c = lane >> 1, which = lane & 1;
            const double v = myRed[c * 2 + which];
            myRed[c * 2 + which] = 0.0;
            unsafeAtomicAdd(p.stats_out + (((size_t)n * FNN_STAT_REPL + ((blockIdx.x + wave) & (FNN_STAT_REPL - 1))) * p.Cout
                                           + cb0 * 16 + c) * 2 + which, v);
        }
    };
    int n_cur, od0, oh0, ow0;
    tile_coords(t_begin, n_cur, od0, oh0, ow0);
    int ch = 0;
#pragma unroll 1
    for (int it = 0; it < n_items; ++it) {
#ifdef FNN_STAMPS
        const bool stamp_it = it >= 4 && it < 8;
        if (stamp_it) FNN_STAMP();                            // before the barrier
#endif
        __syncthreads();
#ifdef FNN_STAMPS
        if (stamp_it) FNN_STAMP();                            // item visible
#endif
        const char *sA = bufp(it);
        const char *sW = sA + ABYTES;
        // software pipeline over the 5 in-plane tap pairs: the fragments of pair p + 1 are read while pair p multiplies
        f16x8 xf[2][ID];
#pragma unroll
        for (int pl = 0; pl < ID; ++pl) xf[0][pl] = *(const f16x8 *)(sA + toff[0] + pl * PS);
#pragma unroll
        for (int pr = 0; pr < 5; ++pr) {
            if (pr + 1 < 5) {
#pragma unroll
                for (int pl = 0; pl < ID; ++pl) xf[(pr + 1) & 1][pl] = *(const f16x8 *)(sA + toff[pr + 1] + pl * PS);
            }
#pragma unroll
            for (int dz = 0; dz < 3; ++dz) {
                f16x8 wf[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) wf[nb] = *(const f16x8 *)(sW + ((nb * KS + pr * 3 + dz) * 64 + lane) * 16);
#pragma unroll
                for (int j = 0; j < TD; ++j)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[j][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nb], xf[pr & 1][j + dz], acc[j][nb], 0, 0, 0);
            }
        }
#ifdef FNN_STAMPS
        if (stamp_it) FNN_STAMP();                            // k-loop done
#endif
        if (++ch == p.chunks) {
            ch = 0;
            float t1[NB][4], t2[NB][4];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int j = 0; j < 4; ++j) { t1[nb][j] = 0.f; t2[nb][j] = 0.f; }
            tile_epilogue<NB, TD, true>(p, acc, bv, n_cur, od0, oh0, ow0, cb0, wave, lane, t1, t2);
            if (p.stats_out) {
                const int q = lane >> 4;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float sa = row16_sum(t1[nb][j]), sb = row16_sum(t2[nb][j]);
                        if ((lane & 15) == 0) {
                            double *slot = myRed + (nb * 16 + q * 4 + j) * 2;
                            slot[0] += (double)sa;
                            slot[1] += (double)sb;
                        }
                    }
            }
#pragma unroll
            for (int j = 0; j < TD; ++j)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[j][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int n_prev = n_cur;
            next_tile(n_cur, od0, oh0, ow0);
            if (n_cur != n_prev || it + 1 == n_items) flush_stats(n_prev);
        }
    }
    FNN_STAMP_FLUSH(p.dbg);
}

template <int NB>
static int launch_zrs(ConvParams p, hipStream_t st) {
    p.tile_d = 8;
    p.tiles_d = (p.Do + 7) / 8;
    p.tiles_h = (p.Ho + 7) / 8;
    p.tiles_w = (p.Wo + 7) / 8;
    p.ident_ss = conv3d_identity_ss();
    if (!p.ident_ss) return -2;
    const int total = p.N * p.tiles_d * p.tiles_h * p.tiles_w;
    const int groups = (p.Cout / 16) / NB;
    const size_t lds = 2 * ((size_t)((10 * 10 * 12 * 32 + 1023) & ~1023) + (size_t)NB * 15 * 1024) + (size_t)NB * 64 +
                       (size_t)4 * NB * 16 * 2 * 8;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)conv3d_zrs_kernel<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    int gx = 256 / groups;                                    // one workgroup per CU over all cout groups
    if (gx < 1) gx = 1;
    if (gx > total) gx = total;
    dim3 grid(gx, groups);
    hipLaunchKernelGGL((conv3d_zrs_kernel<NB>), grid, dim3(512), lds, st, p, total);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <int NB, int TD>
static int launch_zr(ConvParams p, hipStream_t st) {
    p.tile_d = TD;
    p.tiles_d = (p.Do + TD - 1) / TD;
    p.tiles_h = (p.Ho + 7) / 8;
    p.tiles_w = (p.Wo + 7) / 8;
    const size_t lds = (size_t)(((TD + 2) * 10 * 12 * 32 + 1023) & ~1023) + (size_t)NB * 15 * 1024 + (size_t)NB * 64;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)conv3d_zr_kernel<NB, TD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    dim3 grid(p.N * p.tiles_d * p.tiles_h * p.tiles_w, (p.Cout / 16) / NB);
    hipLaunchKernelGGL((conv3d_zr_kernel<NB, TD>), grid, dim3(256), lds, st, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// A persistent form of this kernel (tile ranges per workgroup, cross-tile prefetch, like conv3d_persist_kernel) was
// built and measured: 5-20 % SLOWER on every layer of the benchmark net - <2, 8> does not fit 256 VGPRs next to the
// prefetch registers, <2, 4> loses the operand reuse - so one tile per workgroup it stays.
// Runs the layer on the ZR kernel; the weights must have been packed as FNN_PACK_ZR (p.packing).
int launch_conv3d_zr(const ConvParams &p, hipStream_t st) {
    int nb, td;
    if (p.packing != FNN_PACK_ZR || p.ksteps != 15 || !zr_pick(p, nb, td)) return -1;
    {
        // warp-specialised persistent variant when every CU gets a few tiles per cout group
        static const bool no_zrs = getenv("FNN_NO_ZRS") != nullptr;                 // A-B aids
        static const int zrs_min = getenv("FNN_ZRS_MIN_TILES") ? atoi(getenv("FNN_ZRS_MIN_TILES")) : 4;
        const int plan_n = p.plan_N > 0 ? p.plan_N : p.N;
        const int groups = (p.Cout / 16) / nb;
        const long long tiles = (long long)plan_n * ((p.Do + 7) / 8) * ((p.Ho + 7) / 8) * ((p.Wo + 7) / 8);
        if (!no_zrs && td == 8 && groups <= 256 && tiles * groups >= (long long)zrs_min * 256)
            return nb == 2 ? launch_zrs<2>(p, st) : launch_zrs<1>(p, st);
    }
    if (nb == 2) return td == 8 ? launch_zr<2, 8>(p, st) : launch_zr<2, 4>(p, st);
    return td == 8 ? launch_zr<1, 8>(p, st) : launch_zr<1, 4>(p, st);
}
