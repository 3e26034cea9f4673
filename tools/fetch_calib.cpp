// tools/fetch_calib.cpp - what FETCH_SIZE counts per byte for loads of 4 / 8 / 16 bytes per lane on gfx950 (diagnostic).
//   hipcc --offload-arch=gfx950 -O2 tools/fetch_calib.cpp -o /tmp/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/fc -- /tmp/fetch_calib      (one pass; then the same with WRITE_SIZE)
// Each kernel reads 1 GiB once (sum into a register, one store per wave) with consecutive lanes on consecutive elements:
// read4_kernel 4 B per lane (256 B per wave instruction), read8_kernel 8 B (512 B: the gather kernel's feature loads at 16
// channels), read16_kernel 16 B (1 KiB: every conv kernel's loads).  tools/traffic.py doubles FETCH_SIZE ("wide coalesced reads
// report half"): this says for which widths that is right.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t r_ = (x); if (r_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(r_)); exit(2); } } while (0)

template <typename T>
__device__ unsigned fold(const T &v);
template <> __device__ unsigned fold<unsigned>(const unsigned &v) { return v; }
template <> __device__ unsigned fold<uint2>(const uint2 &v) { return v.x ^ v.y; }
template <> __device__ unsigned fold<uint4>(const uint4 &v) { return v.x ^ v.y ^ v.z ^ v.w; }

template <typename T>
__global__ void read_kernel(const T *__restrict__ a, unsigned *out, size_t n) {
    unsigned s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s ^= fold<T>(a[i]);
    if (s == 0x12345678u) out[0] = s;
}

int main() {
    const size_t bytes = 1ull << 30;
    void *a; unsigned *o;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&o, 64));
    CK(hipMemset(a, 1, bytes));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(read_kernel<unsigned>, dim3(4096), dim3(256), 0, 0, (const unsigned *)a, o, bytes / 4);
        hipLaunchKernelGGL(read_kernel<uint2>, dim3(4096), dim3(256), 0, 0, (const uint2 *)a, o, bytes / 8);
        hipLaunchKernelGGL(read_kernel<uint4>, dim3(4096), dim3(256), 0, 0, (const uint4 *)a, o, bytes / 16);
    }
    CK(hipDeviceSynchronize());
    printf("three rounds of read_kernel<unsigned / uint2 / uint4> over %zu bytes each\n", bytes);
    return 0;
}
