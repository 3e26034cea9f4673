// misc.hip - the remaining gfx950 kernels of the sliding-window path:
//   tconv_mfma_kernel   ConvTranspose3d(kernel = stride)  
//   seg_head_kernel     1x1x1 seg head + Gaussian weighting + accumulate into
//                       the HBM-resident volume accumulators (K6 + K7)
//   patch_acc_kernel    mirrored-evaluation mean -> accumulators (K9 path)
//   finalize_kernel     acc / weight-sum, inf check, un-pad (K8)
//   argmax_kernel       logits -> labels (K10)
#include "fnn_device.h"
#include <cstdlib>

static __device__ __forceinline__ void load_scale_shift(const SrcDesc &s, int n, float2 *sSS, int tid, int nthreads) {
    for (int c = tid; c < s.C; c += nthreads) sSS[c] = s.ss ? s.ss[(size_t)n * s.C + c] : make_float2(1.f, 0.f);
}

// InstanceNorm statistics -> per (n, channel) (scale, shift):  y = x * scale + shift
//   mean = sum / count, var = sumsq / count - mean^2 (biased, like torch), scale = gamma / sqrt(var + eps)
// One thread per (n, channel); the 8 replicas were filled by the producer's epilogue atomics.
__global__ void stats_finalize_kernel(const StatsFinalizeParams p) {
    const int c = threadIdx.x + blockIdx.y * blockDim.x, n = blockIdx.x;
    if (c >= p.C) return;
    const double *st = p.stats + ((size_t)n * FNN_STAT_REPL * p.C + c) * 2;
    double s1 = 0, s2 = 0;
#pragma unroll
    for (int r = 0; r < FNN_STAT_REPL; ++r) {
        s1 += st[(size_t)r * p.C * 2];
        s2 += st[(size_t)r * p.C * 2 + 1];
    }
    const double mean = s1 * (double)p.inv_count;
    double var = s2 * (double)p.inv_count - mean * mean;
    var = var > 0 ? var : 0;
    const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
    const float sc = p.gamma[c] * rstd;
    p.ss[(size_t)n * p.C + c] = make_float2(sc, p.beta[c] - (float)mean * sc);
}

int launch_stats_finalize(const StatsFinalizeParams &p, int N, hipStream_t st) {
    const int bs = p.C < 256 ? ((p.C + 63) / 64) * 64 : 256;
    hipLaunchKernelGGL(stats_finalize_kernel, dim3(N, (p.C + bs - 1) / bs), dim3(bs), 0, st, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// Activation fragment (MFMA B operand) of 16 voxels x 32 channels, read from a
// channels-last tensor with the producer's norm + LeakyReLU applied.
static __device__ __forceinline__ f16x8 load_act_frag(const SrcDesc &s, size_t vox, bool vox_ok, int c0,
                                                      const float2 *sSS) {
    f16x8 o;
    if (vox_ok && c0 < s.C) {
        const f16x8 x = *(const f16x8 *)(s.ptr + vox * s.C + c0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float2 ss = sSS[c0 + j];
            o[j] = (f16)leaky((float)x[j] * ss.x + ss.y, s.slope);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)0.f;
    }
    return o;
}

// ----------------------------------------------------------------------------
// transposed conv, kernel = stride: one GEMM per kernel tap
//   D[cout, voxel] = sum_cin W_tap[cout, cin] * X[cin, voxel]
// grid.x = N * ceil(vox / 256), grid.y = taps * (nblk / NBT); wave = 64 voxels.
// ----------------------------------------------------------------------------
template <int NBT>
__global__ __launch_bounds__(256) void tconv_mfma_kernel(const TconvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *sSS = (float2 *)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int vox_in = p.Di * p.Hi * p.Wi;
    const int wg_per_n = (vox_in + 255) / 256;
    const int n = blockIdx.x / wg_per_n;
    const int v0 = (blockIdx.x - n * wg_per_n) * 256 + wave * 64;
    const int groups = p.nblk / NBT;
    const int tap = blockIdx.y / groups;
    const int cb0 = (blockIdx.y - tap * groups) * NBT;

    load_scale_shift(p.src, n, sSS, tid, 256);
    __syncthreads();

    f32x4 acc[4][NBT];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBT; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int r = lane & 15, q = lane >> 4;
    for (int ks = 0; ks < p.ksteps; ++ks) {
        f16x8 xf[4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const int v = v0 + mb * 16 + r;
            xf[mb] = load_act_frag(p.src, (size_t)n * vox_in + v, v < vox_in, ks * 32 + q * 8, sSS);
        }
#pragma unroll
        for (int nb = 0; nb < NBT; ++nb) {
            const f16x8 wf = *(const f16x8 *)(p.wpk + ((((size_t)tap * p.nblk + cb0 + nb) * p.ksteps + ks) * 64 + lane) * 8);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf[mb], acc[mb][nb], 0, 0, 0);
        }
    }

    const int jd = tap / (p.sh * p.sw), jh = (tap / p.sw) % p.sh, jw = tap % p.sw;
    const int Do = p.Di * p.sd, Ho = p.Hi * p.sh, Wo = p.Wi * p.sw;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        const int v = v0 + mb * 16 + r;
        if (v >= vox_in) continue;
        const int iw = v % p.Wi, ih = (v / p.Wi) % p.Hi, id = v / (p.Wi * p.Hi);
        const size_t ov = (((size_t)n * Do + id * p.sd + jd) * Ho + ih * p.sh + jh) * Wo + iw * p.sw + jw;
#pragma unroll
        for (int nb = 0; nb < NBT; ++nb) {
            const int co = (cb0 + nb) * 16 + q * 4;
            const float4 bv = *(const float4 *)(p.bias + co);
            f16x4 o;
            o[0] = (f16)(acc[mb][nb][0] + bv.x);
            o[1] = (f16)(acc[mb][nb][1] + bv.y);
            o[2] = (f16)(acc[mb][nb][2] + bv.z);
            o[3] = (f16)(acc[mb][nb][3] + bv.w);
            *(f16x4 *)(p.out + ov * p.Cout + co) = o;
        }
    }
}

int launch_tconv(const TconvParams &p, hipStream_t st) {
    const int vox_in = p.Di * p.Hi * p.Wi;
    const int taps = p.sd * p.sh * p.sw;
    const size_t lds = (size_t)p.src.C * 8;
    const int nbt = (p.nblk % 4 == 0) ? 4 : (p.nblk % 2 == 0) ? 2 : 1;
    dim3 grid(p.N * ((vox_in + 255) / 256), taps * (p.nblk / nbt));
    if (nbt == 4) hipLaunchKernelGGL(tconv_mfma_kernel<4>, grid, dim3(256), lds, st, p);
    else if (nbt == 2) hipLaunchKernelGGL(tconv_mfma_kernel<2>, grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL(tconv_mfma_kernel<1>, grid, dim3(256), lds, st, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ----------------------------------------------------------------------------
// accumulate one weighted logit into the volume accumulators
// ----------------------------------------------------------------------------
// Reference rounding (predict_from_raw_data.py:611-614):
//   pred (fp32) *= gaussian (fp16)      -> fp32 product
//   acc (fp16)[sl] += pred              -> fp32 add, ONE round-to-nearest-even to fp16
//   n   (fp16)[sl] += gaussian          -> fp16 + fp16
// fp16 subnormals must survive (5.96e-8 weights): no flush-to-zero is used.
static __device__ __forceinline__ void acc_add(void *acc, size_t idx, float v, int fp32) {
    // __fadd_rn: never contracted with the producing multiply (the reference rounds the product first)
    if (fp32) {
        ((float *)acc)[idx] = __fadd_rn(((float *)acc)[idx], v);
    } else {
        f16 *a = (f16 *)acc;
        a[idx] = (f16)__fadd_rn((float)a[idx], v);
    }
}

// ----------------------------------------------------------------------------
// seg head: D[head, voxel] = Wseg[head, c] * act[c, voxel]  (+ bias), then
//   mode 0: acc[head, origin + voxel] += D * gauss[voxel]; wsum += gauss
//   mode 1/2: patch_buf[head, unflip(voxel)] (=, +=) D        (mirroring path)
// One wave = 64 consecutive patch voxels; the MFMA result is transposed through
// LDS so that the read-modify-write of every head is a contiguous run.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void seg_head_kernel(const HeadParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float2 *sSS = (float2 *)smem;
    float *sT = (float *)(smem + ((p.src.C * 8 + 255) & ~255)) + wave * (64 * 65);   // [64 heads][64(+1) voxels]
    const int P = p.PD * p.PH * p.PW;

    load_scale_shift(p.src, p.b, sSS, tid, 256);
    __syncthreads();

    const int v0 = (blockIdx.x * 4 + wave) * 64;
    if (v0 >= P) return;
    const int r = lane & 15, q = lane >> 4;

    // this lane's voxel for the read-back / RMW phase
    const int v = v0 + lane;
    const bool vok = v < P;
    int w = v % p.PW, h = (v / p.PW) % p.PH, d = v / (p.PW * p.PH);
    if (p.flip_d) d = p.PD - 1 - d;
    if (p.flip_h) h = p.PH - 1 - h;
    if (p.flip_w) w = p.PW - 1 - w;
    const int pv = (d * p.PH + h) * p.PW + w;                 // voxel index in patch space
    float g = 1.f;
    if (vok && p.gauss && p.mode == 0) g = (float)p.gauss[pv];
    const size_t aidx = ((size_t)(p.ox + d) * p.Y + (p.oy + h)) * p.Z + (p.oz + w);
    const size_t plane = (size_t)p.AX * p.Y * p.Z;

    for (int hb0 = 0; hb0 < p.hblocks; hb0 += 4) {
        const int nhb = min(4, p.hblocks - hb0);
        f32x4 acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < p.ksteps; ++ks) {
            f16x8 xf[4];
#pragma unroll
            for (int vb = 0; vb < 4; ++vb) {
                const int vv = v0 + vb * 16 + r;
                xf[vb] = load_act_frag(p.src, (size_t)p.b * P + vv, vv < P, ks * 32 + q * 8, sSS);
            }
#pragma unroll
            for (int hb = 0; hb < 4; ++hb) {
                if (hb < nhb) {
                    const f16x8 wf = *(const f16x8 *)(p.wpk + (((size_t)(hb0 + hb) * p.ksteps + ks) * 64 + lane) * 8);
#pragma unroll
                    for (int vb = 0; vb < 4; ++vb)
                        acc[hb][vb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf[vb], acc[hb][vb], 0, 0, 0);
                }
            }
        }
        // transpose through LDS: sT[head_local][voxel_local]
#pragma unroll
        for (int hb = 0; hb < 4; ++hb)
#pragma unroll
            for (int vb = 0; vb < 4; ++vb)
#pragma unroll
                for (int j = 0; j < 4; ++j) sT[(hb * 16 + q * 4 + j) * 65 + vb * 16 + r] = acc[hb][vb][j];
        __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): this wave's LDS writes landed
        __builtin_amdgcn_wave_barrier();
        if (vok) {
            const int nh = min(64, p.heads - hb0 * 16);
            for (int hl = 0; hl < nh; ++hl) {
                const int head = hb0 * 16 + hl;
                const float val = sT[hl * 65 + lane] + p.bias[head];
                if (p.mode == 0) {
                    acc_add(p.acc, (size_t)head * plane + aidx, __fmul_rn(val, g), p.acc_fp32);
                } else {
                    float *pb = p.patch_buf + (size_t)head * P + pv;
                    *pb = (p.mode == 1) ? val : (*pb + val);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (vok && p.mode == 0) {
        if (p.acc_fp32) ((float *)p.wsum)[aidx] += g;
        else { f16 *wsp = (f16 *)p.wsum; wsp[aidx] = (f16)((float)wsp[aidx] + g); }
    }
}

// ----------------------------------------------------------------------------
// fused seg head + accumulate (mode 0), vectorised read-modify-write
// ----------------------------------------------------------------------------
// The accumulators have a z pitch that is a multiple of 8 and the patch rows
// are walked in accumulator-aligned groups of 8 voxels: a row of PW voxels that
// starts at z = oz becomes GP = ceil(((oz & 7) + PW) / 8) groups, elements
// outside the patch contribute exactly 0.  One wave = 8 groups (64 padded
// voxels); after the MFMA the [head][voxel] tile is transposed through LDS so
// that every lane owns (one head, one group): a 16-byte (fp16) or 2 x 16-byte
// (fp32) read-modify-write, 8 heads x 128 contiguous bytes per wave instruction.
#define HEAD_HBP 2                       // head blocks (of 16) per pass
#define HEAD_LDT 68                      // LDS row stride in floats (64 + pad, keeps 16-B alignment)
__global__ __launch_bounds__(256) void seg_head_acc_kernel(const HeadParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float2 *sSS = (float2 *)smem;
    float *sT = (float *)(smem + ((p.src.C * 8 + 255) & ~255)) + wave * (HEAD_HBP * 16 * HEAD_LDT);
    const int P = p.PD * p.PH * p.PW;
    const int zoff = p.oz & 7;
    const int RW = ((zoff + p.PW + 7) >> 3) << 3;         // padded row length
    const int PV = p.PD * p.PH * RW;

    load_scale_shift(p.src, p.b, sSS, tid, 256);
    __syncthreads();

    const int v0 = (blockIdx.x * 4 + wave) * 64;
    if (v0 >= PV) return;
    const int r = lane & 15, q = lane >> 4;

    // MFMA operand voxels of this lane (4 column blocks)
    size_t src_vox[4];
    bool src_ok[4];
#pragma unroll
    for (int vb = 0; vb < 4; ++vb) {
        const int v = v0 + vb * 16 + r;
        const int wz = v % RW, hh = (v / RW) % p.PH, dd = v / (RW * p.PH);
        const int w = wz - zoff;
        src_ok[vb] = v < PV && w >= 0 && w < p.PW;
        src_vox[vb] = (size_t)p.b * P + ((size_t)dd * p.PH + hh) * p.PW + w;
    }
    // read-modify-write role of this lane: group gl of the tile, head sub-index hs
    const int gl = lane & 7, hs = lane >> 3;
    const int vg = v0 + 8 * gl;
    const bool g_ok = vg < PV;
    const int wz0 = vg % RW, gh = (vg / RW) % p.PH, gd = vg / (RW * p.PH);
    float g[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int w = wz0 + e - zoff;
        const bool ok = g_ok && w >= 0 && w < p.PW;
        g[e] = ok ? (p.gauss ? (float)p.gauss[((size_t)gd * p.PH + gh) * p.PW + w] : 1.f) : 0.f;
    }
    const size_t aidx = ((size_t)(p.ox + gd) * p.Y + (p.oy + gh)) * p.Z + (p.oz - zoff) + wz0;   // multiple of 8
    const size_t plane = (size_t)p.AX * p.Y * p.Z;

    for (int hb0 = 0; hb0 < p.hblocks; hb0 += HEAD_HBP) {
        f32x4 acc[HEAD_HBP][4];
#pragma unroll
        for (int a = 0; a < HEAD_HBP; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < p.ksteps; ++ks) {
            f16x8 xf[4];
#pragma unroll
            for (int vb = 0; vb < 4; ++vb) xf[vb] = load_act_frag(p.src, src_vox[vb], src_ok[vb], ks * 32 + q * 8, sSS);
#pragma unroll
            for (int hb = 0; hb < HEAD_HBP; ++hb) {
                if (hb0 + hb < p.hblocks) {
                    const f16x8 wf = *(const f16x8 *)(p.wpk + (((size_t)(hb0 + hb) * p.ksteps + ks) * 64 + lane) * 8);
#pragma unroll
                    for (int vb = 0; vb < 4; ++vb)
                        acc[hb][vb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf[vb], acc[hb][vb], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int hb = 0; hb < HEAD_HBP; ++hb)
#pragma unroll
            for (int vb = 0; vb < 4; ++vb)
#pragma unroll
                for (int j = 0; j < 4; ++j) sT[(hb * 16 + q * 4 + j) * HEAD_LDT + vb * 16 + r] = acc[hb][vb][j];
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        if (g_ok) {
#pragma unroll
            for (int i = 0; i < HEAD_HBP * 2; ++i) {
                const int hl = 8 * i + hs;
                const int head = hb0 * 16 + hl;
                if (head < p.heads) {
                    const float bias = p.bias[head];
                    const f32x4 t0 = *(const f32x4 *)(sT + hl * HEAD_LDT + 8 * gl);
                    const f32x4 t1 = *(const f32x4 *)(sT + hl * HEAD_LDT + 8 * gl + 4);
                    float c[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { c[e] = __fmul_rn(t0[e] + bias, g[e]); c[4 + e] = __fmul_rn(t1[e] + bias, g[4 + e]); }
                    if (p.acc_fp32) {
                        f32x4 *ap = (f32x4 *)((float *)p.acc + (size_t)head * plane + aidx);
                        f32x4 a0 = ap[0], a1 = ap[1];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {          // elements outside the patch keep their bits (-0 stays -0)
                            a0[e] = g[e] != 0.f ? __fadd_rn(a0[e], c[e]) : a0[e];
                            a1[e] = g[4 + e] != 0.f ? __fadd_rn(a1[e], c[4 + e]) : a1[e];
                        }
                        ap[0] = a0; ap[1] = a1;
                    } else {
                        f16x8 *ap = (f16x8 *)((f16 *)p.acc + (size_t)head * plane + aidx);
                        f16x8 a = *ap;
#pragma unroll
                        for (int e = 0; e < 8; ++e) a[e] = g[e] != 0.f ? (f16)__fadd_rn((float)a[e], c[e]) : a[e];
                        *ap = a;
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (g_ok && hs == 0) {
        if (p.acc_fp32) {
            f32x4 *wp = (f32x4 *)((float *)p.wsum + aidx);
            f32x4 a0 = wp[0], a1 = wp[1];
#pragma unroll
            for (int e = 0; e < 4; ++e) { a0[e] += g[e]; a1[e] += g[4 + e]; }
            wp[0] = a0; wp[1] = a1;
        } else {
            f16x8 *wp = (f16x8 *)((f16 *)p.wsum + aidx);
            f16x8 a = *wp;
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = (f16)((float)a[e] + g[e]);
            *wp = a;
        }
    }
}

int launch_head(const HeadParams &p, hipStream_t st) {
    const int P = p.PD * p.PH * p.PW;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)seg_head_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)seg_head_acc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    static const bool force_scalar = getenv("FNN_HEAD_SCALAR") != nullptr;      // debugging aid
    if (!force_scalar && p.mode == 0 && (p.Z & 7) == 0 && p.oz >= 0) {
        const int RW = (((p.oz & 7) + p.PW + 7) >> 3) << 3;
        const long long PV = (long long)p.PD * p.PH * RW;
        const size_t lds = (size_t)((p.src.C * 8 + 255) & ~255) + (size_t)4 * HEAD_HBP * 16 * HEAD_LDT * 4;
        hipLaunchKernelGGL(seg_head_acc_kernel, dim3((unsigned)((PV + 255) / 256)), dim3(256), lds, st, p);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    const size_t lds = (size_t)((p.src.C * 8 + 255) & ~255) + (size_t)4 * 64 * 65 * 4;
    dim3 grid((P + 255) / 256);
    hipLaunchKernelGGL(seg_head_kernel, grid, dim3(256), lds, st, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ----------------------------------------------------------------------------
// mirrored evaluations: mean of the patch buffer -> accumulators
// (predict_from_raw_data.py:556 `prediction /= n`, then :611-614)
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void patch_acc_kernel(const PatchAccParams p) {
    const int P = p.PD * p.PH * p.PW;
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= P) return;
    const int w = v % p.PW, h = (v / p.PW) % p.PH, d = v / (p.PW * p.PH);
    const float g = p.gauss ? (float)p.gauss[v] : 1.f;
    const size_t aidx = ((size_t)(p.ox + d) * p.Y + (p.oy + h)) * p.Z + (p.oz + w);
    const size_t plane = (size_t)p.AX * p.Y * p.Z;
    const float div = (float)p.n_div;
    for (int head = 0; head < p.heads; ++head) {
        const float val = p.patch_buf[(size_t)head * P + v] / div;
        acc_add(p.acc, (size_t)head * plane + aidx, __fmul_rn(val, g), p.acc_fp32);
    }
    if (p.acc_fp32) ((float *)p.wsum)[aidx] += g;
    else { f16 *wsp = (f16 *)p.wsum; wsp[aidx] = (f16)((float)wsp[aidx] + g); }
}

int launch_patch_acc(const PatchAccParams &p, hipStream_t st) {
    const int P = p.PD * p.PH * p.PW;
    hipLaunchKernelGGL(patch_acc_kernel, dim3((P + 255) / 256), dim3(256), 0, st, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ----------------------------------------------------------------------------
// normalise + un-pad (+ fold ensembling)
// (predict_from_raw_data.py:620-625, :679, :494-500)
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void finalize_kernel(const FinalizeParams p) {
    const long long nout = p.OX * p.OY * p.OZ;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nout) return;
    const long long z = i % p.OZ, y = (i / p.OZ) % p.OY, x = i / (p.OZ * p.OY);
    const size_t aidx = ((size_t)(x + p.lo_x) * p.Y + (y + p.lo_y)) * p.Z + (z + p.lo_z);
    const size_t plane = (size_t)p.AX * p.Y * p.Z;
    const float wsum = p.acc_fp32 ? ((const float *)p.wsum)[aidx] : (float)((const f16 *)p.wsum)[aidx];
    bool bad = false;
    for (int head = 0; head < p.heads; ++head) {
        const float a = p.acc_fp32 ? ((const float *)p.acc)[(size_t)head * plane + aidx]
                                   : (float)((const f16 *)p.acc)[(size_t)head * plane + aidx];
        const float qf = a / wsum;
        const size_t oidx = (size_t)head * nout + i;
        if (p.out_fp32) {
            float *o = (float *)p.out;
            const float r = p.acc_fp32 ? qf : (float)(f16)qf;      // reference-rounding mode rounds to half first
            o[oidx] = p.mode ? o[oidx] + r : r;
            bad |= isinf(o[oidx]);
        } else {
            f16 *o = (f16 *)p.out;
            const f16 r = (f16)qf;
            bad |= isinf((float)r);
            o[oidx] = p.mode ? (f16)((float)o[oidx] + (float)r) : r;
        }
    }
    if (bad) atomicOr(p.inf_flag, 1);
}

int launch_finalize(const FinalizeParams &p, hipStream_t st) {
    const long long nout = p.OX * p.OY * p.OZ;
    hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, st, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

__global__ __launch_bounds__(256) void scale_output_kernel(void *out, int out_fp32, long long n, int divisor) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (out_fp32) ((float *)out)[i] /= (float)divisor;
    else ((f16 *)out)[i] = (f16)((float)((f16 *)out)[i] / (float)divisor);
}

int launch_scale_output(void *out, int out_fp32, long long n, int divisor, int *, hipStream_t st) {
    hipLaunchKernelGGL(scale_output_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out, out_fp32, n, divisor);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ----------------------------------------------------------------------------
// argmax over heads, first maximum wins (numpy argmax, label_handling.py:177)
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void argmax_kernel(const void *logits, int fp32, int heads, long long nvox,
                                                     uint8_t *labels) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nvox) return;
    float best = fp32 ? ((const float *)logits)[i] : (float)((const f16 *)logits)[i];
    int arg = 0;
    bool best_nan = best != best;
    for (int h = 1; h < heads; ++h) {
        const float v = fp32 ? ((const float *)logits)[(size_t)h * nvox + i] : (float)((const f16 *)logits)[(size_t)h * nvox + i];
        // numpy: the first NaN wins; otherwise strictly greater replaces
        if (!best_nan && (v > best || v != v)) { best = v; arg = h; best_nan = v != v; }
    }
    labels[i] = (uint8_t)arg;
}

int launch_argmax(const void *logits, int fp32, int heads, long long nvox, uint8_t *labels, hipStream_t st) {
    hipLaunchKernelGGL(argmax_kernel, dim3((unsigned)((nvox + 255) / 256)), dim3(256), 0, st, logits, fp32, heads, nvox, labels);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ----------------------------------------------------------------------------
// zero-pad a volume that is smaller than the patch (pad_nd_image use at :657)
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pad_volume_kernel(const float *src, float *dst, int C, long long sx, long long sy,
                                                         long long sz, long long dx, long long dy, long long dz,
                                                         long long lx, long long ly, long long lz) {
    const long long n = (long long)C * dx * dy * dz;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long z = i % dz, y = (i / dz) % dy, x = (i / (dz * dy)) % dx, c = i / (dz * dy * dx);
    const long long ux = x - lx, uy = y - ly, uz = z - lz;
    float v = 0.f;
    if (ux >= 0 && ux < sx && uy >= 0 && uy < sy && uz >= 0 && uz < sz) v = src[((c * sx + ux) * sy + uy) * sz + uz];
    dst[i] = v;
}

int launch_pad_volume(const float *src, float *dst, int C, const long long s[3], const long long d[3],
                      const long long lo[3], hipStream_t st) {
    const long long n = (long long)C * d[0] * d[1] * d[2];
    hipLaunchKernelGGL(pad_volume_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, C, s[0], s[1], s[2],
                       d[0], d[1], d[2], lo[0], lo[1], lo[2]);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
