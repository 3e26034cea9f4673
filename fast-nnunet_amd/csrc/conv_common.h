// conv_common.h - pieces shared by the MFMA conv kernels (conv3d.hip, conv3d_zr.hip)
#pragma once
#include "fnn_device.h"

// ----------------------------------------------------------------------------
// shared pieces of the stride-1 MFMA conv kernels
// ----------------------------------------------------------------------------
// Sum over the 16 lanes of a DPP row = the 16 voxels of one MFMA column block (4 VALU ops, no LDS).
static __device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));  // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));  // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xF, 0xF, true));  // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xF, 0xF, true));  // row_ror:1
    return v;
}

// the same for a double (two 32-bit DPP moves per step)
static __device__ __forceinline__ double row16_sum_f64(double v) {
#define FNN_ROR64(ctrl)                                                                                              \
    {                                                                                                                \
        const unsigned long long b = __builtin_bit_cast(unsigned long long, v);                                      \
        const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, ctrl, 0xF, 0xF, true);                       \
        const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), ctrl, 0xF, 0xF, true);               \
        v += __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);                    \
    }
    FNN_ROR64(0x128) FNN_ROR64(0x124) FNN_ROR64(0x122) FNN_ROR64(0x121)
#undef FNN_ROR64
    return v;
}

// small-integer division by a workgroup-uniform divisor (0 <= v < 2^16): float reciprocal + correction
static __device__ __forceinline__ int small_div(int v, int d, float rcp) {
    int q = (int)((float)v * rcp);
    q -= (q * d > v);
    q += ((q + 1) * d <= v);
    return q;
}

// Where a wave's column block `mb` sits in the output tile.  MB = 4 or 8: tile MB x 8 x 8, wave = depth
// slice (+4), block = two h rows.  MB = 2 (strided convs): tile 2 x 8 x 8, wave = (depth slice, h half).
// ZR (conv3d_zr_kernel): tile MB x 8 x 8, wave = two h rows, block = depth slice.
template <int MB, bool ZR = false>
static __device__ __forceinline__ void mb_coords(int wave, int mb, int r, int &od_l, int &oh_l, int &ow_l) {
    if (ZR) { od_l = mb; oh_l = 2 * wave + (r >> 3); }
    else if (MB == 2) { od_l = wave >> 1; oh_l = 4 * (wave & 1) + 2 * mb + (r >> 3); }
    else { od_l = wave + 4 * (mb >> 2); oh_l = 2 * (mb & 3) + (r >> 3); }
    ow_l = r & 7;
}

// Epilogue of one output tile: bias, round to fp16, channels-last store (4 consecutive channels per
// lane), and this lane's partial sums of the rounded values (fp32 within the tile).
// Branch-free: voxels outside the tensor get a buffer offset beyond num_records, which the hardware
// drops.  With `if (ok)` around global stores the number of stores in flight was unknown to hipcc's
// waitcnt pass, and a persistent kernel that prefetches its next tile across the epilogue had to wait
// for vmcnt(0) - the store acknowledgements - before it could touch the prefetched data.
typedef int fnn_i32x2 __attribute__((ext_vector_type(2)));
template <int NB, int MB, bool ZR = false, bool BIAS = true>
static __device__ __forceinline__ void tile_epilogue(const ConvParams &p, const f32x4 (&acc)[MB][NB], const float4 (&bv)[NB],
                                                     int n, int od0, int oh0, int ow0, int cb0, int wave, int lane,
                                                     float (&t1)[NB][4], float (&t2)[NB][4]) {
    const int q = lane >> 4, r = lane & 15;
    const unsigned item_bytes = (unsigned)p.Do * p.Ho * p.Wo * p.Cout * 2;          // < 2^31 (checked by the launchers)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)n * (item_bytes >> 1), 0,
                                                                           item_bytes, 0x00020000);
    const unsigned ovs2 = (unsigned)FNN_OVS(p) * 2, ocs2 = (unsigned)(FNN_OCS(p) * 2);     // output layout (fnn_device.h, SrcDesc)
    const unsigned coff = (unsigned)cb0 * ocs2 + (unsigned)q * 8;
    // Statistics of the fp16-rounded outputs with packed dot products: the values of one channel at two voxels (m-blocks
    // mb, mb + 1) are packed into one register; v_dot2_f32_f16(pair, (1, 1), t1) adds both to the sum and
    // v_dot2_f32_f16(pair, pair, t2) both squares to the sum of squares - 1.5 instructions per value instead of 4
    // (convert back, select, add, fma); out-of-range voxels contribute a zeroed pair.
    static_assert(MB % 2 == 0, "m-blocks are processed in pairs");
    const f16x2 ones = {(f16)1.f, (f16)1.f};
#pragma unroll
    for (int mb = 0; mb < MB; mb += 2) {
        bool okv[2];
        unsigned voff[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int od_l, oh_l, ow_l;
            mb_coords<MB, ZR>(wave, mb + h, r, od_l, oh_l, ow_l);
            const int od = od0 + od_l, oh = oh0 + oh_l, ow = ow0 + ow_l;
            okv[h] = od < p.Do && oh < p.Ho && ow < p.Wo;
            voff[h] = okv[h] ? (unsigned)((od * p.Ho + oh) * p.Wo + ow) * ovs2 + coff : 0x80000000u;
#ifdef FNN_TMODE
            if (p.tmode & 4) voff[h] = 0x80000000u;
#endif
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            f16x4 o[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                o[h][0] = (f16)(BIAS ? acc[mb + h][nb][0] + bv[nb].x : acc[mb + h][nb][0]);
                o[h][1] = (f16)(BIAS ? acc[mb + h][nb][1] + bv[nb].y : acc[mb + h][nb][1]);
                o[h][2] = (f16)(BIAS ? acc[mb + h][nb][2] + bv[nb].z : acc[mb + h][nb][2]);
                o[h][3] = (f16)(BIAS ? acc[mb + h][nb][3] + bv[nb].w : acc[mb + h][nb][3]);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(fnn_i32x2, o[h]), rsrc, voff[h], nb * ocs2, 0);
                if (!okv[h]) o[h] = (f16x4){0, 0, 0, 0};
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f16x2 pr = {o[0][j], o[1][j]};
                t1[nb][j] = __builtin_amdgcn_fdot2(pr, ones, t1[nb][j], false);
                t2[nb][j] = __builtin_amdgcn_fdot2(pr, pr, t2[nb][j], false);
            }
        }
    }
}

// Workgroup reduction of the statistics and the double atomics into replica (blockIdx.x & 7).
// `sRed` = 4 * NB * 32 floats of LDS that nobody else uses between the two barriers.
// SLOT: the workgroup owns row `slot` of the item's stats rows and stores its sums there (no atomics: the double
// atomics of one-tile-per-workgroup kernels cost 5 % of the whole benchmark - 1.5 M of them per launch).
// PERM: the ZR kernels' channel order at NB = 2 (conv3d_pack_cout): lane quarter q holds channels q * 8 + nb * 4 + j.
// `at(i)`: where float i of the 4 * NB * 32 reduction floats lives in LDS (a plain array, or slots that a kernel has free)
template <int NB, bool SLOT = false, bool PERM = false, int NW = 4, typename At>
static __device__ __forceinline__ void stats_to_global_at(const ConvParams &p, float (&t1)[NB][4], float (&t2)[NB][4], At at,
                                                          int n, int cb0, int wave, int lane, int tid, int slot = 0) {
    const int q = lane >> 4, r = lane & 15;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = row16_sum(t1[nb][j]), b = row16_sum(t2[nb][j]);
            if (r == 0) {
                const int c = PERM ? q * 8 + nb * 4 + j : nb * 16 + q * 4 + j;
                *at((wave * NB * 16 + c) * 2) = a;
                *at((wave * NB * 16 + c) * 2 + 1) = b;
            }
        }
    __syncthreads();
    if (tid < NB * 16 * 2) {
        const int c = tid >> 1, which = tid & 1;
        double v = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += (double)*at((w * NB * 16 + c) * 2 + which);   // NW = the workgroup's waves
        if (SLOT) p.stats_out[(((size_t)n * p.stats_slots + slot) * p.Cout + cb0 * 16 + c) * 2 + which] = v;
        else unsafeAtomicAdd(p.stats_out + (((size_t)n * FNN_STAT_REPL + (blockIdx.x & (FNN_STAT_REPL - 1))) * p.Cout + cb0 * 16 + c) * 2 + which, v);
    }
}

template <int NB, bool SLOT = false, bool PERM = false, int NW = 4>
static __device__ __forceinline__ void stats_to_global(const ConvParams &p, float (&t1)[NB][4], float (&t2)[NB][4], float *sRed,
                                                       int n, int cb0, int wave, int lane, int tid, int slot = 0) {
    stats_to_global_at<NB, SLOT, PERM, NW>(p, t1, t2, [sRed](int i) { return sRed + i; }, n, cb0, wave, lane, tid, slot);
}

