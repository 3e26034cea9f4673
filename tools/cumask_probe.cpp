// tools/cumask_probe.cpp - what a HIP stream's CU mask selects on MI355X, and what the selected CUs can pull from HBM
// (diagnostic, not product).
//   hipcc --offload-arch=gfx950 -O2 tools/cumask_probe.cpp -o gpurun_out/cumask_probe && gpurun_out/cumask_probe
// Q1  bits [0, n) of hipExtStreamCreateWithCUMask: which (XCD, shader engine, CU) do the workgroups land on?  (decides whether
//     "the first 64 bits" is 8 CUs in each of the 8 XCDs - every L2 stays in use - or two whole XCDs)
// Q2  a streaming copy (16 B per lane, grid-strided, 1 GB) on n CUs: GB/s.  The HBM-bound full-resolution kernels of the
//     network can keep at most this share of their rate on a masked stream.
// Q3  the same copy on a masked stream WHILE an MFMA loop runs on the complementary CUs: do the two slow each other down?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <map>

#define CK(x) do { hipError_t r_ = (x); if (r_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(r_)); exit(2); } } while (0)

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void where_kernel(unsigned *out) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    // spin a little so that the workgroups spread over every CU the queue may use
    unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < 20000) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}

__global__ void copy_kernel(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

__global__ void mfma_kernel(float *out, int iters) {
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (f16)(threadIdx.x * 0.001f + j); b[j] = (f16)(j * 0.5f); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
    }
    if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.f) out[0] = 1.f;
}

static hipStream_t masked(int lo, int hi) {
    uint32_t m[8] = {};
    for (int c = lo; c < hi; ++c) m[c >> 5] |= 1u << (c & 31);
    hipStream_t s;
    CK(hipExtStreamCreateWithCUMask(&s, 8, m));
    return s;
}

int main() {
    const int NWG = 4096;
    unsigned *d_out; CK(hipMalloc(&d_out, NWG * 2 * 4));
    std::vector<unsigned> h(NWG * 2);
    for (int n : {32, 64, 96, 128, 256}) {
        hipStream_t s = masked(0, n);
        hipLaunchKernelGGL(where_kernel, dim3(NWG), dim3(64), 0, s, d_out);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), d_out, NWG * 8, hipMemcpyDeviceToHost));
        std::map<unsigned, std::map<unsigned, int>> per_xcc;      // xcc -> (se, sh, cu) -> workgroups
        for (int i = 0; i < NWG; ++i) {
            const unsigned xcc = h[2 * i] & 0xf, hw = h[2 * i + 1];
            const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            per_xcc[xcc][(se << 8) | (sh << 4) | cu]++;
        }
        printf("Q1  mask bits [0,%d): ", n);
        int total = 0;
        for (auto &x : per_xcc) { printf("XCD%u:%zu CUs  ", x.first, x.second.size()); total += (int)x.second.size(); }
        printf("= %d distinct CUs\n", total);
        CK(hipStreamDestroy(s));
    }
    const size_t bytes = 1ull << 30, nvec = bytes / 16;
    uint4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    float *mo; CK(hipMalloc(&mo, 64));
    hipEvent_t e0, e1, m0, m1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&m0)); CK(hipEventCreate(&m1));
    for (int n : {32, 48, 64, 96, 128, 192, 256}) {
        hipStream_t s = masked(0, n);
        const int grid = n * 8;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, s, a, b, nvec);
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("Q2  copy 1 GiB on %3d CUs alone: %7.1f us  %6.0f GB/s (read + write)\n", n, ms * 100.f, 2.0 * bytes * 10 / (ms * 1e-3) / 1e9);
        if (n < 256) {
            hipStream_t h2 = masked(n, 256);
            const int mgrid = (256 - n) * 8;                                 // 8 waves per CU of the complement
            // calibrate the MFMA loop alone, then both together
            const int iters = 200000;
            CK(hipEventRecord(m0, h2));
            hipLaunchKernelGGL(mfma_kernel, dim3(mgrid), dim3(64), 0, h2, mo, iters);
            CK(hipEventRecord(m1, h2));
            CK(hipStreamSynchronize(h2));
            float mms_alone; CK(hipEventElapsedTime(&mms_alone, m0, m1));
            CK(hipEventRecord(m0, h2));
            hipLaunchKernelGGL(mfma_kernel, dim3(mgrid), dim3(64), 0, h2, mo, iters);
            CK(hipEventRecord(m1, h2));
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, s, a, b, nvec);
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(h2));
            float mms; CK(hipEventElapsedTime(&mms, m0, m1)); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("Q3  ... beside an MFMA loop on the other %3d CUs: copy %7.1f us %6.0f GB/s; MFMA loop %.2f ms alone, %.2f ms beside the copy\n",
                   256 - n, ms * 100.f, 2.0 * bytes * 10 / (ms * 1e-3) / 1e9, mms_alone, mms);
            CK(hipStreamDestroy(h2));
        }
        CK(hipStreamDestroy(s));
    }
    return 0;
}
