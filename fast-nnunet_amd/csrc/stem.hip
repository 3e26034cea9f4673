// Stem conv of the network: reads the fp32 volume window at the patch origin directly (no patch copy,
// SURVEY.md K11), fp32 VALU, raw fp16 output + instance-norm statistics.
//
// Replaces the first torch.nn.Conv3d of the encoder as composed by dynamic_network_architectures'
// ConvDropoutNormReLU (instantiated at nnUNetDistillationTrainer.py:141-173), fed by the patch slicing of
// predict_from_raw_data.py:560-566.
//
// This file is compiled with -fno-slp-vectorize (see the Makefile).
#include "fnn_device.h"
#include "conv_common.h"
#include <cstdlib>

// ----------------------------------------------------------------------------
// stem conv: fp32 volume window -> raw fp16 + statistics
// ----------------------------------------------------------------------------
// fp32 FMA (1..8 input channels: no MFMA shape fits).  Workgroup = 16 x 8 x 8 output voxels x 16 output
// channels; a thread owns one (h, w) column and walks 4 depth slices, so the statistics' cross-lane
// reduction is paid once per 4 voxels (it dominated the one-voxel-per-thread version: 192 ds_bpermute per
// voxel).  The conv's zero padding is at the PATCH border (each patch is an independent network input),
// not at the volume border; mirroring flips the window read.
#define STEM_TD 16
__global__ __launch_bounds__(256) void stem_conv_kernel(const StemParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int t = blockIdx.x;
    const int tw = t % p.tiles_w; t /= p.tiles_w;
    const int th = t % p.tiles_h; t /= p.tiles_h;
    const int td = t % p.tiles_d;
    const int n = t / p.tiles_d;
    const int cb = blockIdx.y;                       // block of 16 output channels

    const int pd = (p.kd - 1) / 2, ph = (p.kh - 1) / 2, pw = (p.kw - 1) / 2;
    const int ID = STEM_TD - 1 + p.kd, IH = FNN_TILE_H - 1 + p.kh, IW = FNN_TILE_W - 1 + p.kw;
    const int IVOX = ID * IH * IW;
    const int T = p.kd * p.kh * p.kw;

    float *sIn = (float *)smem;                      // [C][IVOX]
    float *sW = sIn + ((p.C * IVOX + 3) & ~3);       // [C][T][16]
    float *sRed = sW + p.C * T * 16;                 // [16 rows][16][2]

    const int ox = p.origins[n * 3 + 0], oy = p.origins[n * 3 + 1], oz = p.origins[n * 3 + 2];
    const int d0 = td * STEM_TD - pd, h0 = th * FNN_TILE_H - ph, w0 = tw * FNN_TILE_W - pw;
    const float rcp_iw = 1.0f / (float)IW, rcp_ih = 1.0f / (float)IH;
    const float *voln = p.vol + (size_t)n * p.vol_batch_stride;
    for (int c = 0; c < p.C; ++c)
        for (int v = tid; v < IVOX; v += 256) {
            const int row = small_div(v, IW, rcp_iw), zw = v - row * IW;
            const int zd = small_div(row, IH, rcp_ih), zh = row - zd * IH;
            int d = d0 + zd, h = h0 + zh, w = w0 + zw;
            const bool ok = d >= 0 && d < p.PD && h >= 0 && h < p.PH && w >= 0 && w < p.PW;
            if (p.flip_d) d = p.PD - 1 - d;
            if (p.flip_h) h = p.PH - 1 - h;
            if (p.flip_w) w = p.PW - 1 - w;
            const float val = voln[(((size_t)c * p.X + (ox + (ok ? d : 0))) * p.Y + (oy + (ok ? h : 0))) * p.Z + (oz + (ok ? w : 0))];
            sIn[c * IVOX + v] = ok ? val : 0.f;
        }
    for (int idx = tid; idx < p.C * T * 16; idx += 256) sW[idx] = p.w[(size_t)(idx >> 4) * p.Cout + cb * 16 + (idx & 15)];
    float bias[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) bias[j] = p.bias[cb * 16 + j];
    __syncthreads();

    const int ow_l = lane & 7, oh_l = lane >> 3;
    const int oh = th * FNN_TILE_H + oh_l, ow = tw * FNN_TILE_W + ow_l;
    float t1[16], t2[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { t1[j] = 0.f; t2[j] = 0.f; }
    // taps outermost: a tap's 16 weights are read from LDS once and serve the thread's 4 depth slices (they were
    // re-read per slice: 144 ds_read_b128 per thread against 576 FMAs)
    float acc[4][16];
#pragma unroll
    for (int dd = 0; dd < 4; ++dd)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[dd][j] = bias[j];
    for (int c = 0; c < p.C; ++c)
        for (int a = 0; a < p.kd; ++a)
            for (int b = 0; b < p.kh; ++b)
                for (int e = 0; e < p.kw; ++e) {
                    const float4 *wv = (const float4 *)(sW + ((c * T) + (a * p.kh + b) * p.kw + e) * 16);
                    const float4 w0 = wv[0], w1 = wv[1], w2 = wv[2], w3 = wv[3];
#pragma unroll
                    for (int dd = 0; dd < 4; ++dd) {
                        const float x = sIn[c * IVOX + ((wave * 4 + dd + a) * IH + (oh_l + b)) * IW + (ow_l + e)];
                        acc[dd][0] = fmaf(x, w0.x, acc[dd][0]); acc[dd][1] = fmaf(x, w0.y, acc[dd][1]);
                        acc[dd][2] = fmaf(x, w0.z, acc[dd][2]); acc[dd][3] = fmaf(x, w0.w, acc[dd][3]);
                        acc[dd][4] = fmaf(x, w1.x, acc[dd][4]); acc[dd][5] = fmaf(x, w1.y, acc[dd][5]);
                        acc[dd][6] = fmaf(x, w1.z, acc[dd][6]); acc[dd][7] = fmaf(x, w1.w, acc[dd][7]);
                        acc[dd][8] = fmaf(x, w2.x, acc[dd][8]); acc[dd][9] = fmaf(x, w2.y, acc[dd][9]);
                        acc[dd][10] = fmaf(x, w2.z, acc[dd][10]); acc[dd][11] = fmaf(x, w2.w, acc[dd][11]);
                        acc[dd][12] = fmaf(x, w3.x, acc[dd][12]); acc[dd][13] = fmaf(x, w3.y, acc[dd][13]);
                        acc[dd][14] = fmaf(x, w3.z, acc[dd][14]); acc[dd][15] = fmaf(x, w3.w, acc[dd][15]);
                    }
                }
#pragma unroll
    for (int dd = 0; dd < 4; ++dd) {
        const int od = td * STEM_TD + wave * 4 + dd;
        const bool ok = od < p.PD && oh < p.PH && ow < p.PW;
        f16x8 o0, o1;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const f16 hv = (f16)acc[dd][j];
            if (j < 8) o0[j] = hv; else o1[j - 8] = hv;
            const float f = ok ? (float)hv : 0.f;
            t1[j] += f;
            t2[j] = fmaf(f, f, t2[j]);
        }
        if (ok) {
            f16 *dst = p.out + ((((size_t)n * p.PD + od) * p.PH + oh) * p.PW + ow) * p.Cout + cb * 16;
            *(f16x8 *)dst = o0;
            *(f16x8 *)(dst + 8) = o1;
        }
    }
    if (p.stats_out) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float a = row16_sum(t1[j]), b = row16_sum(t2[j]);
            if ((lane & 15) == 0) {
                sRed[((wave * 4 + (lane >> 4)) * 16 + j) * 2] = a;
                sRed[((wave * 4 + (lane >> 4)) * 16 + j) * 2 + 1] = b;
            }
        }
        __syncthreads();
        if (tid < 32) {
            const int c = tid >> 1, which = tid & 1;
            double v = 0;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) v += (double)sRed[(rr * 16 + c) * 2 + which];
            // one stats row per tile, plain store (see stats_to_global in conv_common.h)
            const int slot = (td * p.tiles_h + th) * p.tiles_w + tw;
            p.stats_out[(((size_t)n * (p.tiles_d * p.tiles_h * p.tiles_w) + slot) * p.Cout + cb * 16 + c) * 2 + which] = v;
        }
    }
}

int stem_stats_slots(int PD, int PH, int PW) {
    return ((PD + STEM_TD - 1) / STEM_TD) * ((PH + FNN_TILE_H - 1) / FNN_TILE_H) * ((PW + FNN_TILE_W - 1) / FNN_TILE_W);
}

int launch_stem(const StemParams &p_in, int N, hipStream_t st) {
    StemParams p = p_in;
    p.tiles_d = (p.PD + STEM_TD - 1) / STEM_TD;
    p.tiles_h = (p.PH + FNN_TILE_H - 1) / FNN_TILE_H;
    p.tiles_w = (p.PW + FNN_TILE_W - 1) / FNN_TILE_W;
    const int ID = STEM_TD - 1 + p.kd, IH = FNN_TILE_H - 1 + p.kh, IW = FNN_TILE_W - 1 + p.kw;
    const int IVOX = ID * IH * IW, T = p.kd * p.kh * p.kw;
    const size_t lds = (size_t)((p.C * IVOX + 3) & ~3) * 4 + (size_t)p.C * T * 16 * 4 + 16 * 16 * 2 * 4;
    if (lds > 160 * 1024) return -1;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)stem_conv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    dim3 grid(N * p.tiles_d * p.tiles_h * p.tiles_w, p.Cout / 16);
    hipLaunchKernelGGL(stem_conv_kernel, grid, dim3(256), lds, st, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
