// conv3d_zq.hip - 3x3x3 stride-1 Conv3d over whole 12 x 12 planes, eight depth slices per tile, eight waves that load the
// four SIMDs evenly (gfx950; round 5).
//
// conv3d_zr12_kernel (conv3d_zr.hip) gave the 128-channel stage of a 96 x 96 in-plane patch (planes of 12 x 12 voxels)
// tiles of 4 x 12 x 12 with NINE waves - one 4 x 4 column block each: three of them share a SIMD, the other SIMDs hold two,
// and the workgroup takes as long as the crowded SIMD (matrix pipes busy 0.40, 32 % of the MFMA peak; a deeper tile
// spilled: nine waves leave 168 registers each).  Here the same planes with
//   * tile 8 x 12 x 12 (halo 10 x 14 x 14: 1.25 instead of 1.5 staged planes per output plane, a chunk's 30 KB of weight
//     fragments serve 1152 outputs instead of 576);
//   * 8 waves = (depth quarter dq: output slices 2 dq, 2 dq + 1) x (block group: blocks 0 .. 4 | 5 .. 8 of the plane's nine
//     4 x 4 column blocks); the groups alternate so that the two waves of every SIMD own 5 + 4 blocks: all four SIMDs
//     carry 9 blocks x 2 slices - two waves per SIMD, 256 registers each;
//   * the depth-shift reuse of the ZR kernels inside a wave's two slices: the fragment of halo plane pl serves depth tap dz
//     of slice pl - dz: per (chunk, in-plane tap pair, block) 4 activation reads + (once per pair) 6 weight reads for 12
//     MFMAs: 0.43 LDS reads per MFMA (nine one-block waves at four slices: 0.5);
//   * staging as in conv3d_zr_kernel since round 3: a thread owns one (row, column, 8-channel half) of the halo and walks
//     the ten planes - buffer loads with a per-thread constant + a scalar per plane, columns outside the tensor read zeros
//     through the range check, fp16 scale / shift rows, the chunk's loads issued in five slices along the k-loop.
// LDS image as conv3d_zr12_kernel: row pitch 20 voxels, channel halves swapped on odd rows (rows 0 / 3 of a block in one
// half, 1 / 2 in the other: the 16 lanes of a ds_read_b128 group hit 16 different slots).  Same weights (FNN_PACK_ZR,
// interleaved cout order), same arithmetic per output value (bias added to the fp32 sum in the epilogue): the OUTPUT BITS
// are conv3d_zr12_kernel's; the statistics row per tile sums other voxel groups (fp32 inside a tile, double across).
//
// Replaces the stride-1 ConvDropoutNormReLU blocks of the reference's PlainConvEncoder / UNetDecoder at that stage
// (nnUNetDistillationTrainer.py:141-173).
#include "fnn_device.h"
#include "conv_common.h"
#include <cstdlib>

namespace {

typedef unsigned zq_u32x4 __attribute__((ext_vector_type(4)));
typedef int zq_i32x4 __attribute__((ext_vector_type(4)));

constexpr int ZQ_TD = 8, ZQ_ID = ZQ_TD + 2, ZQ_IH = 14, ZQ_IW = 14, ZQ_PW = 20;
constexpr int ZQ_PS = ZQ_IH * ZQ_PW * 32;                       // bytes per halo plane (8960)
constexpr int ZQ_ABYTES = ZQ_ID * ZQ_PS;                        // 89600
constexpr int ZQ_KS = 15, ZQ_WB = ZQ_KS * 64;                   // 16-byte weight elements per cout block and chunk (960)
constexpr int ZQ_NT = 512, ZQ_NB = 2;
constexpr int ZQ_WBYTES = ZQ_NB * ZQ_KS * 1024;                 // 30720
constexpr int ZQ_LDS = ZQ_ABYTES + ZQ_WBYTES + 8 * 32 * 2 * 4;  // + the statistics' reduction floats

__global__ __launch_bounds__(512, 2) void conv3d_zq12_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dq = wave >> 1, grp = (wave ^ (wave >> 2)) & 1;   // waves w and w + 4 share a SIMD: one of each group
    const int nblk = grp ? 4 : 5, b0 = grp ? 5 : 0;
    const int td = blockIdx.x % p.tiles_d, n = blockIdx.x / p.tiles_d;
    const int cb0 = blockIdx.y * ZQ_NB;
    const int od0 = td * ZQ_TD;

    char *sA = smem;                                           // [10][14][20] voxels x 32 B, halves swapped on odd rows
    char *sW = smem + ZQ_ABYTES;                               // [2][15][64 lanes][16 B]
    float *sRed = (float *)(sW + ZQ_WBYTES);                   // [8 waves][32][2]

    // ---- this thread's halo column (392 of the 512 threads)
    const int col = tid >> 1, cg = tid & 1;
    const int zh = (col * 2341) >> 15, zw = col - zh * ZQ_IW;  // col / 14 for col < 256
    const bool has_col = tid < 2 * ZQ_IH * ZQ_IW;
    const int gh = zh - 1, gw = zw - 1;
    const bool ok_hw = has_col & ((unsigned)gh < (unsigned)p.Hi) & ((unsigned)gw < (unsigned)p.Wi);
    const int hw_lin = __mul24(gh, p.Wi) + gw;
    const int ldso0 = (zh * ZQ_PW + zw) * 32 + ((cg ^ (zh & 1)) * 16);
    const int wlds = ZQ_ABYTES + tid * 16;
    unsigned pmask = (1u << ZQ_ID) - 1;                        // halo planes that lie inside the tensor
    if (od0 == 0) pmask &= ~1u;
    {
        const int over = od0 + ZQ_TD + 1 - p.Di;
        if (over > 0) pmask &= (1u << (ZQ_ID - over)) - 1;
    }
    pmask = __builtin_amdgcn_readfirstlane(pmask);

    f32x4 acc[5][2][ZQ_NB];
    zq_u32x4 xr[ZQ_ID], wr[ZQ_NB][2];
    zq_u32x4 ssv[2];
    float slope_next = 1.f;
    __amdgpu_buffer_rsrc_t rx, rw[ZQ_NB];
    unsigned voff = 0x80000000u, plane_bytes = 0;
    auto prep = [&](int ch) {
        const int c_glob = ch * 16;
        const int s = (c_glob < p.src[0].C) ? 0 : 1;
        const int c_uni = c_glob - (s ? p.src[0].C : 0);
        const int sC = p.src[s].C;
        const int vs = FNN_VS(p.src[s]);                       // activation layout: fnn_device.h, SrcDesc
        const unsigned item_bytes = (unsigned)p.Di * p.Hi * p.Wi * sC * 2;
        const f16 *sp = p.src[s].ptr + (size_t)n * (item_bytes >> 1) + (c_uni >> 4) * FNN_CS(p.src[s]);
        rx = __builtin_amdgcn_make_buffer_rsrc((void *)sp, 0, item_bytes, 0x00020000);
        slope_next = p.src[s].slope;
        {
            const unsigned short *q = p.src[s].ssh ? p.src[s].ssh + ((size_t)n * sC + c_uni) * 2 : p.ident_ssh + c_uni * 2;
            const zq_u32x4 *qv = (const zq_u32x4 *)(q + cg * 16);
            ssv[0] = qv[0]; ssv[1] = qv[1];
        }
        voff = ok_hw ? (unsigned)hw_lin * (unsigned)(vs * 2) + cg * 16 : 0x80000000u;
        plane_bytes = (unsigned)p.Hi * p.Wi * vs * 2;
#pragma unroll
        for (int nb = 0; nb < ZQ_NB; ++nb) {
            const f16 *wp = p.wpk + ((size_t)((cb0 + nb) * p.chunks + ch) * ZQ_WB) * 8;
            rw[nb] = __builtin_amdgcn_make_buffer_rsrc((void *)wp, 0, ZQ_WB * 16, 0x00020000);
        }
    };
    // the chunk's 14 loads per thread leave in five slices, one per tap pair of the k-loop (issued in one block they queue up
    // in the texture-address path and the MFMAs behind them cannot issue)
    auto load_part = [&](int part) {
#pragma unroll
        for (int u = 0; u < ZQ_ID; ++u) {
            if (u * 5 / ZQ_ID != part) continue;
            int gd = od0 - 1 + u;
            gd = gd < 0 ? 0 : (gd >= p.Di ? p.Di - 1 : gd);    // scalar; a clamped plane's image is zeroed in commit()
            xr[u] = __builtin_bit_cast(zq_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff, (unsigned)gd * plane_bytes, 0));
        }
#pragma unroll
        for (int e = 0; e < ZQ_NB * 2; ++e) {                  // element tid + 512 u of block nb; beyond the block: range check, zeros, no traffic
            if (e * 5 / (ZQ_NB * 2) != part) continue;
            const int nb = e >> 1, u = e & 1;
            wr[nb][u] = __builtin_bit_cast(zq_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw[nb], tid * 16, u * 8192, 0));
        }
    };
    auto commit = [&]() {
        const f16 slope_h = (f16)slope_next;
        // x * scale + shift with scale and shift in fp16 (v_pk_fma_f16: fnn_norm8's arithmetic); a column outside the tensor:
        // 0 * 0 + 0 = the conv's zero padding
        const zq_u32x4 zero4 = {0u, 0u, 0u, 0u};
        const f16x8 sc_h = __builtin_bit_cast(f16x8, ok_hw ? ssv[0] : zero4), sh_h = __builtin_bit_cast(f16x8, ok_hw ? ssv[1] : zero4);
        if (has_col) {
#pragma unroll
            for (int u = 0; u < ZQ_ID; ++u) {
                const f16x8 x = __builtin_bit_cast(f16x8, xr[u]);
#ifdef FNN_NORM_FP32
                f16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (f16)fmaf((float)x[j], (float)sc_h[j], (float)sh_h[j]);
#else
                f16x8 o = x * sc_h + sh_h;
#endif
                o = __builtin_elementwise_max(o, o * slope_h);
                *(f16x8 *)(sA + ldso0 + u * ZQ_PS) = o;
            }
            if (pmask != (1u << ZQ_ID) - 1) {                  // border tiles along d: planes outside the tensor
                unsigned pm = pmask;
                asm volatile("" : "+s"(pm));
#pragma unroll
                for (int u = 0; u < ZQ_ID; ++u)
                    if (!((pm >> u) & 1)) *(f16x8 *)(sA + ldso0 + u * ZQ_PS) = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
#pragma unroll
        for (int nb = 0; nb < ZQ_NB; ++nb) {
            *(zq_u32x4 *)(smem + wlds + nb * ZQ_WB * 16) = wr[nb][0];
            if (tid < ZQ_WB - 512) *(zq_u32x4 *)(smem + wlds + (nb * ZQ_WB + 512) * 16) = wr[nb][1];
        }
    };

    // MFMA "B" operand of block 0 of the plane: lane = (voxel r of the 4 x 4 block, k-group): bit 1 of the k-group picks the
    // tap of the pair, bit 0 the 8-channel half; + this wave's first halo plane.  A block's own offset is a scalar.
    int toff[5];
    {
        const int r = lane & 15, hl = lane >> 5, kh = (lane >> 4) & 1;
#pragma unroll
        for (int pr = 0; pr < 5; ++pr) {
            const int tp = 2 * pr + hl < 9 ? 2 * pr + hl : 8;   // padded slot: any finite data (its weights are 0)
            const int row = (r >> 2) + tp / 3, cl = (r & 3) + tp % 3;
            toff[pr] = (row * ZQ_PW + cl) * 32 + ((kh ^ (row & 1)) * 16) + dq * 2 * ZQ_PS;
        }
    }
    int boff[5];                                               // (4 bh rows: an even number, the swap parity is the lane's)
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int b = b0 + i, bh = (b * 11) >> 5, bw = b - bh * 3;   // b / 3 for b < 9
        boff[i] = __builtin_amdgcn_readfirstlane((4 * bh * ZQ_PW + 4 * bw) * 32);
    }

    auto read_x = [&](f16x8 (&xf)[4], int pr, int i) {
        const char *bp = sA + toff[pr] + boff[i];
#pragma unroll
        for (int pl = 0; pl < 4; ++pl) xf[pl] = *(const f16x8 *)(bp + pl * ZQ_PS);
    };
    auto kloop = [&](bool prefetch) {
#pragma unroll
        for (int pr = 0; pr < 5; ++pr) {
            if (prefetch) load_part(pr);
            f16x8 wf[3][ZQ_NB];
#pragma unroll
            for (int dz = 0; dz < 3; ++dz)
#pragma unroll
                for (int nb = 0; nb < ZQ_NB; ++nb) wf[dz][nb] = *(const f16x8 *)(sW + ((nb * ZQ_KS + pr * 3 + dz) * 64 + lane) * 16);
            f16x8 xa[4], xb[4];
            read_x(xa, pr, 0);
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                f16x8 (&cur)[4] = (i & 1) ? xb : xa;
                f16x8 (&nxt)[4] = (i & 1) ? xa : xb;
                if (i + 1 < 5 && (i + 1 < 4 || nblk == 5)) read_x(nxt, pr, i + 1);   // the next block's fragments leave before this block's MFMAs
                __builtin_amdgcn_sched_barrier(0);
                if (i < 4 || nblk == 5) {
#pragma unroll
                    for (int dz = 0; dz < 3; ++dz)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int nb = 0; nb < ZQ_NB; ++nb)
                                acc[i][j][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[dz][nb], cur[j + dz], acc[i][j][nb], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    prep(0);
#pragma unroll
    for (int part = 0; part < 5; ++part) load_part(part);
    __builtin_amdgcn_sched_barrier(0);                         // the loads leave first; the set-up above runs under them
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int nb = 0; nb < ZQ_NB; ++nb) acc[i][j][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    commit();
    __syncthreads();
    // the last chunk is peeled off so that the wait for the prefetch sits on an unconditional path
    for (int ch = 0; ch + 1 < p.chunks; ++ch) {
        prep(ch + 1);
        kloop(true);
        __syncthreads();                                       // every wave is done reading this chunk
        commit();
        __syncthreads();
    }
    kloop(false);

    // ---- epilogue: bias, 16-byte stores (the interleaved cout order: lane quarter q holds channels q * 8 .. + 7), statistics
    {
        const int q = lane >> 4, r = lane & 15;
        float4 bv[ZQ_NB];
#pragma unroll
        for (int nb = 0; nb < ZQ_NB; ++nb) bv[nb] = *(const float4 *)(p.bias + cb0 * 16 + q * 8 + nb * 4);
        float t1[ZQ_NB][4], t2[ZQ_NB][4];
#pragma unroll
        for (int nb = 0; nb < ZQ_NB; ++nb)
#pragma unroll
            for (int j = 0; j < 4; ++j) { t1[nb][j] = 0.f; t2[nb][j] = 0.f; }
        const unsigned item_bytes = (unsigned)p.Do * p.Ho * p.Wo * p.Cout * 2;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)n * (item_bytes >> 1), 0, item_bytes, 0x00020000);
        const unsigned ovs2 = (unsigned)FNN_OVS(p) * 2;
        const unsigned coff = (unsigned)(cb0 + (q >> 1)) * (unsigned)(FNN_OCS(p) * 2) + (unsigned)(q & 1) * 16;   // output layout: fnn_device.h
        const f16x2 ones = {(f16)1.f, (f16)1.f};
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            if (i == 4 && nblk == 4) break;
            const int b = b0 + i, bh = (b * 11) >> 5, bw = b - bh * 3;
            const int oh = 4 * bh + (r >> 2), ow = 4 * bw + (r & 3);
            const bool ok_o = oh < p.Ho && ow < p.Wo;
            f16x8 o[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int od = od0 + 2 * dq + h;
                const bool ok = ok_o && od < p.Do;
                const unsigned vo = ok ? (unsigned)((od * p.Ho + oh) * p.Wo + ow) * ovs2 + coff : 0x80000000u;
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    o[h][nb * 4 + 0] = (f16)(acc[i][h][nb][0] + bv[nb].x);
                    o[h][nb * 4 + 1] = (f16)(acc[i][h][nb][1] + bv[nb].y);
                    o[h][nb * 4 + 2] = (f16)(acc[i][h][nb][2] + bv[nb].z);
                    o[h][nb * 4 + 3] = (f16)(acc[i][h][nb][3] + bv[nb].w);
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(zq_i32x4, o[h]), rsrc, vo, 0, 0);
                if (!ok) o[h] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
            }
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f16x2 pr2 = {o[0][nb * 4 + j], o[1][nb * 4 + j]};
                    t1[nb][j] = __builtin_amdgcn_fdot2(pr2, ones, t1[nb][j], false);
                    t2[nb][j] = __builtin_amdgcn_fdot2(pr2, pr2, t2[nb][j], false);
                }
        }
        if (p.stats_out) {
#pragma unroll
            for (int nb = 0; nb < ZQ_NB; ++nb)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = row16_sum(t1[nb][j]), b = row16_sum(t2[nb][j]);
                    if (r == 0) {
                        const int c = q * 8 + nb * 4 + j;
                        sRed[(wave * 32 + c) * 2] = a;
                        sRed[(wave * 32 + c) * 2 + 1] = b;
                    }
                }
            __syncthreads();
            if (tid < 64) {
                const int c = tid >> 1, which = tid & 1;
                double v = 0;
#pragma unroll
                for (int w = 0; w < 8; ++w) v += (double)sRed[(w * 32 + c) * 2 + which];
                p.stats_out[(((size_t)n * p.stats_slots + td) * p.Cout + cb0 * 16 + c) * 2 + which] = v;
            }
        }
    }
}

}  // namespace

// planes of 9 .. 12 x 9 .. 12 voxels, at least 8 deep, an even number of cout blocks; the layer's statistics buffer must have
// a row per tile (conv3d_stats_slots)
bool conv3d_zq12_ok(const ConvParams &p) {
    const bool off = fnn_knob("FNN_NO_ZQ12") != nullptr;                             // A-B aid, read per call (the tests compare the two kernels' bits): conv3d_zr12_kernel instead
    if (off || p.fp8 || p.packing != FNN_PACK_ZR || p.ksteps != ZQ_KS) return false;
    if (p.kd != 3 || p.kh != 3 || p.kw != 3 || p.sd != 1 || p.sh != 1 || p.sw != 1) return false;
    if ((p.Cout / 16) % 2 != 0 || p.Ho <= 8 || p.Ho > 12 || p.Wo <= 8 || p.Wo > 12 || p.Do < ZQ_TD) return false;
    // depth-8 tiles must not waste much more of the layer's depth than conv3d_zr12_kernel's depth-4 tiles do (Do = 10: 16
    // computed slices against 12, and 2 instead of 3 tiles per item to spread over the CUs)
    if (((p.Do + 7) / 8) * 8 * 10 > ((p.Do + 3) / 4) * 4 * 11) return false;
    return !p.stats_out || p.stats_slots >= (p.Do + ZQ_TD - 1) / ZQ_TD;
}

int launch_conv3d_zq12(ConvParams p, hipStream_t st) {
    if (!conv3d_zq12_ok(p)) return -1;
    p.tile_d = ZQ_TD;
    p.tiles_d = (p.Do + ZQ_TD - 1) / ZQ_TD; p.tiles_h = 1; p.tiles_w = 1;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)conv3d_zq12_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    p.ident_ss = conv3d_identity_ss();
    p.ident_ssh = conv3d_identity_ssh();
    if (!p.ident_ss || !p.ident_ssh) return -2;
    dim3 grid(p.N * p.tiles_d, (p.Cout / 16) / ZQ_NB);
    fnn_note_kernel("conv3d_zq12_kernel");
    hipLaunchKernelGGL(conv3d_zq12_kernel, grid, dim3(ZQ_NT), ZQ_LDS, st, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
