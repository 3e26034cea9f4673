// resample.hip - image / logit resampling between spacings, gfx950.
//
// Replaces resample_data_or_seg(..., is_seg=False) (preprocessing/resampling/default_resampling.py:113-196),
// i.e. skimage.transform.resize(order, mode='edge', anti_aliasing=False) per channel (or per slice + an order-0
// pass along the anisotropic axis).  skimage (>= 0.19) evaluates resize as
// scipy.ndimage.zoom(float64 image, out/in, order, mode='nearest', grid_mode=True) and clips to the input range;
// for order 3 zoom (a) pads 12 edge samples, (b) turns samples into cubic B-spline coefficients with the
// recursive pole-(sqrt(3)-2) filter, (c) evaluates 4 taps per axis at x = (o + 0.5) * in/out - 0.5.
//
// On the GPU the recursive filter becomes its impulse response: c[i] = sum_k h[k] x[i+k], h[k] = -6z/(1-z^2) z^|k|,
// truncated at |k| = 30 (z^31 ~ 2e-18: below fp64 resolution) with mirror indexing at the padded ends - fully
// parallel and coalesced instead of one serial recursion per line.  All arithmetic is fp64 like the reference's
// (`data.astype(float)`); agreement with scipy's zoom is ~1e-14 relative, i.e. results are the same fp32 / fp16
// numbers except where a value sits on a rounding boundary.
#include "fnn_device.h"
#include "../../include/fnn.h"
#include <cfloat>
#include <cmath>

void fnn_set_global_error(const char *msg);      // engine.hip

namespace {

constexpr int NPAD = 12;       // scipy _prepad_for_spline_filter, mode 'nearest'
constexpr int KT = 30;         // FIR half length

static int fail_msg(int code, const char *msg) { fnn_set_global_error(msg); return code; }

static bool dev_ptr(const void *p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged;
}

struct Geo {
    long long in[3], out[3];   // spatial sizes
    int pad[3];                // NPAD on resampled axes with order 3, else 0
    long long pd[3];           // padded sizes
    int sep;                   // separate axis or -1
    int order;
};

template <typename T> static __device__ __forceinline__ double ld(const T *p, long long i) { return (double)p[i]; }

// ---- channel -> fp64, edge padded
template <typename T>
__global__ __launch_bounds__(256) void pad_kernel(const T *in, Geo g, double *P) {
    const long long n = g.pd[0] * g.pd[1] * g.pd[2];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    long long c[3] = {i / (g.pd[1] * g.pd[2]), (i / g.pd[2]) % g.pd[1], i % g.pd[2]};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        c[a] -= g.pad[a];
        c[a] = c[a] < 0 ? 0 : (c[a] >= g.in[a] ? g.in[a] - 1 : c[a]);
    }
    P[i] = ld(in, (c[0] * g.in[1] + c[1]) * g.in[2] + c[2]);
}

// ---- B-spline prefilter along one axis: Q = h * P, mirror at the ends of the padded axis
__global__ __launch_bounds__(256) void fir_kernel(const double *P, Geo g, int axis, double *Q) {
    const long long n = g.pd[0] * g.pd[1] * g.pd[2];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long stride = axis == 2 ? 1 : (axis == 1 ? g.pd[2] : g.pd[1] * g.pd[2]);
    const long long len = g.pd[axis];
    const long long pos = (i / stride) % len;
    const long long base = i - pos * stride;
    const double z = -0.26794919243112270647;                 // sqrt(3) - 2
    const double h0 = -6.0 * z / (1.0 - z * z);
    double acc = h0 * P[i], zk = z;
    const long long period = 2 * len - 2;
    for (int k = 1; k <= KT; ++k) {
        long long a = pos - k, b = pos + k;
        if (len == 1) { a = 0; b = 0; }
        else {
            a = a < 0 ? -a : a; a = a % period; a = a >= len ? period - a : a;
            b = b % period; b = b >= len ? period - b : b;
        }
        acc += h0 * zk * (P[base + a * stride] + P[base + b * stride]);
        zk *= z;
    }
    Q[i] = acc;
}

// ---- per channel (sep < 0) or per slice of the separate axis: min / max of the input (resize's clip range)
template <typename T>
__global__ __launch_bounds__(256) void minmax_kernel(const T *in, Geo g, unsigned *mm /* [slices][2] ordered */) {
    // one block row per slice (blockIdx.y); the whole channel is one "slice" when sep < 0
    const int s = blockIdx.y;
    const long long n = g.in[0] * g.in[1] * g.in[2];
    const long long per = g.sep < 0 ? n : n / g.in[g.sep];
    float mn = FLT_MAX, mx = -FLT_MAX;
    bool any = false;
    for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < per; j += (long long)gridDim.x * 256) {
        long long idx = j;
        if (g.sep == 0) idx = (long long)s * g.in[1] * g.in[2] + j;
        else if (g.sep == 1) idx = ((j / g.in[2]) * g.in[1] + s) * g.in[2] + j % g.in[2];
        else if (g.sep == 2) idx = j * g.in[2] + s;
        const float v = (float)in[idx];
        mn = v < mn ? v : mn; mx = v > mx ? v : mx;
        any = true;
    }
    auto ord = [](float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
    if (any) { atomicMin(&mm[2 * s], ord(mn)); atomicMax(&mm[2 * s + 1], ord(mx)); }
}

static __device__ __forceinline__ float unord(unsigned u) {
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __uint_as_float(u);
}

// ---- evaluation at the output grid (zoom_shift with grid_mode), clip, cast
template <typename T>
__global__ __launch_bounds__(256) void interp_kernel(const double *C, Geo g, const unsigned *mm, T *out) {
    const long long n = g.out[0] * g.out[1] * g.out[2];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long o[3] = {i / (g.out[1] * g.out[2]), (i / g.out[2]) % g.out[1], i % g.out[2]};
    long long i0[3];
    double w[3][4];
    int nt[3];
    long long slice = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double zoom = (double)g.in[a] / (double)g.out[a];
        if (a == g.sep) {
            // order-0 pass of map_coordinates(mode='nearest') along the separate axis (default_resampling.py:176-188)
            double x = zoom * ((double)o[a] + 0.5) - 0.5;
            long long s = (long long)floor(x + 0.5);
            s = s < 0 ? 0 : (s >= g.in[a] ? g.in[a] - 1 : s);
            if (g.in[a] == g.out[a]) s = o[a];
            i0[a] = s; nt[a] = 1; w[a][0] = 1.0; slice = s;
            continue;
        }
        if (g.in[a] == g.out[a] && g.order != 3) { i0[a] = o[a]; nt[a] = 1; w[a][0] = 1.0; continue; }
        double x = (double)o[a] * zoom + (0.5 * zoom - 0.5);
        if (g.order == 3) {
            x += g.pad[a];
            const double f = floor(x), t = x - f;
            i0[a] = (long long)f - 1; nt[a] = 4;
            const double u = 1.0 - t;                            // same expressions as scipy's spline weights
            w[a][1] = (t * t * (t - 2.0) * 3.0 + 4.0) / 6.0;
            w[a][2] = (u * u * (u - 2.0) * 3.0 + 4.0) / 6.0;
            w[a][0] = u * u * u / 6.0;
            w[a][3] = 1.0 - w[a][0] - w[a][1] - w[a][2];
        } else if (g.order == 1) {
            x = x < 0 ? 0 : (x > (double)(g.in[a] - 1) ? (double)(g.in[a] - 1) : x);     // mode 'nearest'
            const double f = floor(x), t = x - f;
            i0[a] = (long long)f; nt[a] = 2; w[a][0] = 1.0 - t; w[a][1] = t;
        } else {
            long long s = (long long)floor(x + 0.5);
            s = s < 0 ? 0 : (s >= g.in[a] ? g.in[a] - 1 : s);
            i0[a] = s; nt[a] = 1; w[a][0] = 1.0;
        }
    }
    double acc = 0;
    for (int a0 = 0; a0 < nt[0]; ++a0) {
        long long c0 = i0[0] + a0; c0 = c0 < 0 ? 0 : (c0 >= g.pd[0] ? g.pd[0] - 1 : c0);
        for (int a1 = 0; a1 < nt[1]; ++a1) {
            long long c1 = i0[1] + a1; c1 = c1 < 0 ? 0 : (c1 >= g.pd[1] ? g.pd[1] - 1 : c1);
            for (int a2 = 0; a2 < nt[2]; ++a2) {                // taps in C order, ((c * w0) * w1) * w2 like zoom_shift
                long long c2 = i0[2] + a2; c2 = c2 < 0 ? 0 : (c2 >= g.pd[2] ? g.pd[2] - 1 : c2);
                acc += ((C[(c0 * g.pd[1] + c1) * g.pd[2] + c2] * w[0][a0]) * w[1][a1]) * w[2][a2];
            }
        }
    }
    const double lo = (double)unord(mm[2 * (g.sep < 0 ? 0 : slice)]), hi = (double)unord(mm[2 * (g.sep < 0 ? 0 : slice) + 1]);
    acc = acc < lo ? lo : (acc > hi ? hi : acc);               // resize(clip=True)
    out[i] = (T)acc;
}

template <typename T>
static int run(const T *in, const int64_t shape[4], const int64_t new_shape[3], const fnn_resample_desc &d, T *out, hipStream_t st) {
    Geo g{};
    g.sep = d.separate_axis; g.order = d.order;
    bool same = true;
    for (int a = 0; a < 3; ++a) {
        g.in[a] = shape[1 + a]; g.out[a] = new_shape[a];
        same &= g.in[a] == g.out[a];
        g.pad[a] = (d.order == 3 && a != g.sep) ? NPAD : 0;
        g.pd[a] = g.in[a] + 2 * g.pad[a];
    }
    const long long nin = g.in[0] * g.in[1] * g.in[2], nout = g.out[0] * g.out[1] * g.out[2];
    if (same) {                                               // "no resampling necessary" (:193-195)
        if (hipMemcpyAsync(out, in, (size_t)shape[0] * nin * sizeof(T), hipMemcpyDeviceToDevice, st) != hipSuccess) return -2;
        return hipStreamSynchronize(st) == hipSuccess ? 0 : -2;
    }
    const long long npd = g.pd[0] * g.pd[1] * g.pd[2];
    const int slices = g.sep < 0 ? 1 : (int)g.in[g.sep];
    double *P = nullptr, *Q = nullptr;
    unsigned *mm = nullptr;
    if (hipMalloc((void **)&P, (size_t)npd * 8) != hipSuccess) return -2;
    if (d.order == 3 && hipMalloc((void **)&Q, (size_t)npd * 8) != hipSuccess) { (void)hipFree(P); return -2; }
    if (hipMalloc((void **)&mm, (size_t)slices * 8) != hipSuccess) { (void)hipFree(P); (void)hipFree(Q); return -2; }
    std::vector<unsigned> init((size_t)slices * 2);
    for (int s = 0; s < slices; ++s) { init[2 * s] = 0xffffffffu; init[2 * s + 1] = 0u; }
    hipError_t r = hipSuccess;
    for (int64_t c = 0; c < shape[0] && r == hipSuccess; ++c) {
        const T *inc = in + c * nin;
        r = hipMemcpyAsync(mm, init.data(), init.size() * 4, hipMemcpyHostToDevice, st);
        if (r != hipSuccess) break;
        const long long per = g.sep < 0 ? nin : nin / g.in[g.sep];
        long long bx = (per + 255) / 256; if (bx > 2048) bx = 2048;
        hipLaunchKernelGGL(minmax_kernel<T>, dim3((unsigned)bx, (unsigned)slices), dim3(256), 0, st, inc, g, mm);
        hipLaunchKernelGGL(pad_kernel<T>, dim3((unsigned)((npd + 255) / 256)), dim3(256), 0, st, inc, g, P);
        const double *coef = P;
        if (d.order == 3) {
            double *src = P, *dst = Q;
            for (int a = 0; a < 3; ++a) {
                if (a == g.sep) continue;
                hipLaunchKernelGGL(fir_kernel, dim3((unsigned)((npd + 255) / 256)), dim3(256), 0, st, src, g, a, dst);
                double *t = src; src = dst; dst = t;
            }
            coef = src;
        }
        hipLaunchKernelGGL(interp_kernel<T>, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, st, coef, g, mm, out + c * nout);
        r = hipGetLastError();
    }
    if (r == hipSuccess) r = hipStreamSynchronize(st);
    (void)hipFree(P); (void)hipFree(Q); (void)hipFree(mm);
    return r == hipSuccess ? 0 : -2;
}

}  // namespace

extern "C" int fnn_resample(const void *in, const int64_t shape[4], const int64_t new_shape[3], const fnn_resample_desc *d,
                            void *out, void *stream) {
    if (!in || !shape || !new_shape || !d || !out) return fail_msg(FNN_E_INVALID, "NULL argument");
    if (d->order != 0 && d->order != 1 && d->order != 3) return fail_msg(FNN_E_UNSUPPORTED, "interpolation order must be 0, 1 or 3");
    if (d->separate_axis < -1 || d->separate_axis > 2) return fail_msg(FNN_E_INVALID, "separate_axis must be -1 .. 2");
    if (d->separate_axis >= 0 && d->order_z != 0) return fail_msg(FNN_E_UNSUPPORTED, "order_z other than 0 is not implemented");
    if (d->dtype != FNN_OUT_F16 && d->dtype != FNN_OUT_F32) return fail_msg(FNN_E_INVALID, "unknown dtype");
    for (int a = 0; a < 4; ++a) if (shape[a] < 1) return fail_msg(FNN_E_INVALID, "bad shape");
    for (int a = 0; a < 3; ++a) if (new_shape[a] < 1) return fail_msg(FNN_E_INVALID, "bad new_shape");
    if (!dev_ptr(in) || !dev_ptr(out)) return fail_msg(FNN_E_INVALID, "fnn_resample needs device pointers (no CPU path)");
    const int rc = d->dtype == FNN_OUT_F32 ? run<float>((const float *)in, shape, new_shape, *d, (float *)out, (hipStream_t)stream)
                                           : run<f16>((const f16 *)in, shape, new_shape, *d, (f16 *)out, (hipStream_t)stream);
    if (rc != 0) return fail_msg(FNN_E_HIP, "resample kernels failed");
    return FNN_OK;
}
