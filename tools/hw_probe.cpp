// tools/hw_probe.cpp - questions about gfx950 arithmetic that decide kernel designs (diagnostic, not product).
//   hipcc --offload-arch=gfx950 -O2 tools/hw_probe.cpp -o gpurun_out/hw_probe && gpurun_out/hw_probe
// Q1  v_mfma_f32_16x16x32_f16 with the bias as the C operand  vs  the bias split into three fp16 pieces in unused k slots
//     (B = 1.0 there) with C = 0: the same bits?   (would free the 16 bias registers of gather_head_kernel)
// Q2  v_mfma_f32_16x16x32_f16 on 16 real channels (k >= 16 zero)  vs  v_mfma_f32_16x16x16_f16: the same bits?
// Q3  v_fma_mix_f32(a.f16, 1.0, c.f32)  vs  (float)a + c: the same bits for every fp16 a (subnormals, inf, NaN included)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t r_ = (x); if (r_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(r_)); exit(2); } } while (0)

// one wave per tile: A [16][16] halves (row-major head x k), B [16][16] halves (k x voxel), bias [16] floats
__global__ void mfma_forms(const f16 *A, const f16 *B, const float *bias, unsigned *diff12, unsigned *diff13, int tiles) {
    const int lane = threadIdx.x & 63, tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (tile >= tiles) return;
    const f16 *a = A + (size_t)tile * 256, *b = B + (size_t)tile * 256;
    const float *bs = bias + (size_t)tile * 16;
    const int r = lane & 15, q = lane >> 4;
    f16x8 a32 = {0, 0, 0, 0, 0, 0, 0, 0}, b32 = a32, a32b = a32, b32b = a32;
    if (q < 2) for (int j = 0; j < 8; ++j) { a32[j] = a[r * 16 + 8 * q + j]; b32[j] = b[(8 * q + j) * 16 + r]; }
    a32b = a32; b32b = b32;
    if (q == 2) {                                             // k = 16, 17, 18: the bias of head r in three fp16 pieces x 1.0
        const float bv = bs[r];
        const f16 h0 = (f16)bv; const float r1 = bv - (float)h0;
        const f16 h1 = (f16)r1; const float r2 = r1 - (float)h1;
        const f16 h2 = (f16)r2;
        a32b[0] = h0; a32b[1] = h1; a32b[2] = h2;
        b32b[0] = (f16)1.f; b32b[1] = (f16)1.f; b32b[2] = (f16)1.f;
    }
    f16x4 a16, b16;
    for (int j = 0; j < 4; ++j) { a16[j] = a[r * 16 + 4 * q + j]; b16[j] = b[(4 * q + j) * 16 + r]; }
    f32x4 c;
    for (int j = 0; j < 4; ++j) c[j] = bs[4 * q + j];         // C/D: row (head) = 4 q + j, col = r
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a32, b32, c, 0, 0, 0);
    const f32x4 d2 = __builtin_amdgcn_mfma_f32_16x16x16f16(a16, b16, c, 0, 0, 0);
    const f32x4 d3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a32b, b32b, zero, 0, 0, 0);
    unsigned n12 = 0, n13 = 0;
    for (int j = 0; j < 4; ++j) {
        n12 += __builtin_bit_cast(unsigned, d1[j]) != __builtin_bit_cast(unsigned, d2[j]);
        n13 += __builtin_bit_cast(unsigned, d1[j]) != __builtin_bit_cast(unsigned, d3[j]);
    }
    if (n12) atomicAdd(diff12, n12);
    if (n13) atomicAdd(diff13, n13);
}

__global__ void mix_add(const float *cs, int ncs, unsigned *diff_lo, unsigned *diff_hi, unsigned *example) {
    const unsigned id = blockIdx.x * blockDim.x + threadIdx.x;       // a = id & 0xffff, c index = id >> 16
    const unsigned short ab = (unsigned short)(id & 0xffffu);
    const int ci = (int)(id >> 16);
    if (ci >= ncs) return;
    const float c = cs[ci];
    const f16 ah = __builtin_bit_cast(f16, ab);
    float ref;
    {
#pragma clang fp contract(off)
        ref = (float)ah + c;
    }
    const unsigned packed_lo = ab, packed_hi = (unsigned)ab << 16;
    float lo, hi;
    asm volatile("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(packed_lo), "v"(c));
    asm volatile("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(hi) : "v"(packed_hi), "v"(c));
    const unsigned rb = __builtin_bit_cast(unsigned, ref), lb = __builtin_bit_cast(unsigned, lo), hb = __builtin_bit_cast(unsigned, hi);
    const bool rn = ref != ref;
    const bool dl = rn ? !(lo != lo) : rb != lb, dh = rn ? !(hi != hi) : rb != hb;     // NaNs: any NaN is fine here (payloads are compared below for the f16 result)
    // what the kernel stores is the fp16 rounding of the sum: compare those bits too, NaN payload included
    const unsigned short r16 = __builtin_bit_cast(unsigned short, (f16)ref), l16 = __builtin_bit_cast(unsigned short, (f16)lo);
    if (dl || r16 != l16) { atomicAdd(diff_lo, 1u); example[0] = ab; example[1] = __builtin_bit_cast(unsigned, c); example[2] = rb; example[3] = lb; }
    if (dh) atomicAdd(diff_hi, 1u);
}

static f16 rnd_h(double scale) { return (f16)(float)(scale * ((double)rand() / RAND_MAX * 2.0 - 1.0)); }

int main() {
    srand(1234);
    const int tiles = 1 << 16;
    std::vector<f16> A((size_t)tiles * 256), B((size_t)tiles * 256);
    std::vector<float> bias((size_t)tiles * 16);
    for (int t = 0; t < tiles; ++t) {
        const double sa = t % 3 == 0 ? 0.35 : (t % 3 == 1 ? 4.0 : 0.01), sb = t % 5 == 0 ? 30.0 : 1.5;
        for (int i = 0; i < 256; ++i) { A[(size_t)t * 256 + i] = rnd_h(sa); B[(size_t)t * 256 + i] = rnd_h(sb); }
        for (int i = 0; i < 16; ++i) bias[(size_t)t * 16 + i] = (float)(0.05 * ((double)rand() / RAND_MAX * 2.0 - 1.0) * (t % 7 == 0 ? 100.0 : 1.0));
    }
    f16 *dA, *dB; float *dbias; unsigned *dd;
    CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dbias, bias.size() * 4)); CK(hipMalloc(&dd, 64));
    CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbias, bias.data(), bias.size() * 4, hipMemcpyHostToDevice)); CK(hipMemset(dd, 0, 64));
    hipLaunchKernelGGL(mfma_forms, dim3(tiles / 4), dim3(256), 0, 0, dA, dB, dbias, dd, dd + 1, tiles);
    CK(hipDeviceSynchronize());
    unsigned h[16];
    CK(hipMemcpy(h, dd, 64, hipMemcpyDeviceToHost));
    printf("Q1 bias as C vs bias in three k slots: %u of %d values differ\n", h[1], tiles * 256);
    printf("Q2 16x16x32 (k >= 16 zero) vs 16x16x16:  %u of %d values differ\n", h[0], tiles * 256);

    std::vector<float> cs;
    const float special[] = {0.f, -0.f, 1.f, -1.f, 5.9604645e-8f, 1e-10f, 6.1e-5f, 65504.f, 1e6f, INFINITY, -INFINITY, NAN, 0.00048828125f, 3.0517578e-5f, 1.1754944e-38f, 1e-40f};
    for (float v : special) cs.push_back(v);
    for (int i = 0; i < 240; ++i) {
        const double m = (double)rand() / RAND_MAX * 2.0 - 1.0;
        cs.push_back((float)(m * std::pow(2.0, (double)(rand() % 40) - 28.0)));
    }
    float *dcs; CK(hipMalloc(&dcs, cs.size() * 4)); CK(hipMemcpy(dcs, cs.data(), cs.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dd, 0, 64));
    const unsigned total = (unsigned)cs.size() << 16;
    hipLaunchKernelGGL(mix_add, dim3(total / 256), dim3(256), 0, 0, dcs, (int)cs.size(), dd, dd + 1, dd + 4);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, dd, 64, hipMemcpyDeviceToHost));
    printf("Q3 v_fma_mix_f32(a.h, 1.0, c) vs (float)a + c over %u pairs: lo form %u differ, hi form %u differ", total, h[0], h[1]);
    if (h[0]) printf("  (example a=%04x c=%08x ref=%08x mix=%08x)", h[4], h[5], h[6], h[7]);
    printf("\n");
    return 0;
}
