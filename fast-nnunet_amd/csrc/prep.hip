// prep.hip - whole-volume steps on either side of the sliding window, gfx950.
//
//   fnn_nonzero_bbox    crop_to_nonzero's bounding box (preprocessing/cropping/cropping.py:7-39)
//   fnn_preprocess      transpose_forward + crop + per-channel intensity normalisation
//                       (preprocessing/preprocessors/default_preprocessor.py:45-93 without the resampling call;
//                       preprocessing/normalization/default_normalization_schemes.py:27-109)
//   fnn_revert_labels   label map back to the uncropped, untransposed grid (inference/export_prediction.py:43-53)
//
// All three are HBM-bound element-wise / reduction kernels: one pass over the raw volume for the box, one pass
// for the statistics of the schemes that need them, one read + one write for the result.
#include "fnn_device.h"
#include "../../include/fnn.h"
#include <cfloat>
#include <climits>
#include <cstdio>
#include <cmath>

extern "C" const char *fnn_last_error(const fnn_engine *e);
void fnn_set_global_error(const char *msg);      // engine.hip

namespace {

struct PrepGeom {
    long long s[3];              // raw spatial shape
    int tf[3];                   // transposed axis a = raw axis tf[a]
    int C;
};

static bool dev_ptr(const void *p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged;
}

static int fail_msg(int code, const char *msg) { fnn_set_global_error(msg); return code; }

// ---- bounding box of the voxels where any channel is non-zero, in transposed coordinates
__global__ __launch_bounds__(256) void bbox_kernel(const float *raw, PrepGeom g, int *box /* lo[3], hi[3] (inclusive) */) {
    __shared__ int sbox[6];
    if (threadIdx.x < 3) sbox[threadIdx.x] = INT_MAX;
    else if (threadIdx.x < 6) sbox[threadIdx.x] = -1;
    __syncthreads();
    const long long n = g.s[0] * g.s[1] * g.s[2];
    int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {-1, -1, -1};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        bool nz = false;
        for (int c = 0; c < g.C; ++c) nz |= raw[c * n + i] != 0.f;           // NaN != 0 is true, like numpy
        if (nz) {
            const int r[3] = {(int)(i / (g.s[1] * g.s[2])), (int)((i / g.s[2]) % g.s[1]), (int)(i % g.s[2])};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const int t = r[g.tf[a]];
                lo[a] = t < lo[a] ? t : lo[a];
                hi[a] = t > hi[a] ? t : hi[a];
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (hi[a] >= 0) { atomicMin(&sbox[a], lo[a]); atomicMax(&sbox[3 + a], hi[a]); }
    }
    __syncthreads();
    if (threadIdx.x < 3) { if (sbox[threadIdx.x] != INT_MAX) atomicMin(&box[threadIdx.x], sbox[threadIdx.x]); }
    else if (threadIdx.x < 6) { if (sbox[threadIdx.x] >= 0) atomicMax(&box[threadIdx.x], sbox[threadIdx.x]); }
}

struct PrepParams {
    PrepGeom g;
    long long lo[3], ext[3];     // crop box in transposed coordinates: origin and extent
    int scheme[8];
    float a[8], b[8], lower[8], upper[8];     // per channel: out = (clip(x) - a) / b
    int use_mask[8];             // ZScore inside the filled non-zero mask only (use_mask_for_norm)
    const uint8_t *mask;         // [ext0][ext1][ext2]: 2 = outside, anything else = inside (or nullptr)
};

// raw offset of transposed-cropped voxel (t0, t1, t2)
static __device__ __forceinline__ long long raw_index(const PrepParams &p, long long t0, long long t1, long long t2) {
    long long r[3];
    r[p.g.tf[0]] = t0 + p.lo[0]; r[p.g.tf[1]] = t1 + p.lo[1]; r[p.g.tf[2]] = t2 + p.lo[2];
    return (r[0] * p.g.s[1] + r[1]) * p.g.s[2] + r[2];
}

// ---- per-channel statistics over the crop box: sum, sum of squares (double), min, max
__global__ __launch_bounds__(256) void stats_kernel(const float *raw, PrepParams p, int c, double *sums /*[3]: sum, sumsq, count*/, unsigned *mm /*[2] ordered*/) {
    __shared__ double ssum[3][4];
    __shared__ unsigned smm[2];
    if (threadIdx.x == 0) { smm[0] = 0xffffffffu; smm[1] = 0u; }
    __syncthreads();
    const long long n = p.ext[0] * p.ext[1] * p.ext[2], nraw = p.g.s[0] * p.g.s[1] * p.g.s[2];
    double s1 = 0, s2 = 0, cnt = 0;
    float mn = FLT_MAX, mx = -FLT_MAX;
    bool any = false;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long t2 = i % p.ext[2], t1 = (i / p.ext[2]) % p.ext[1], t0 = i / (p.ext[2] * p.ext[1]);
        const float v = raw[(long long)c * nraw + raw_index(p, t0, t1, t2)];
        if (p.use_mask[c] && p.mask[i] == 2) continue;          // outside the filled non-zero mask
        s1 += (double)v; s2 += (double)v * (double)v; cnt += 1.0;
        mn = v < mn ? v : mn; mx = v > mx ? v : mx;
        any = true;
    }
    // order-preserving map float -> unsigned for the atomics
    auto ord = [](float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
    if (any) { atomicMin(&smm[0], ord(mn)); atomicMax(&smm[1], ord(mx)); }
    for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_down(s1, off); s2 += __shfl_down(s2, off); cnt += __shfl_down(cnt, off); }
    if ((threadIdx.x & 63) == 0) { ssum[0][threadIdx.x >> 6] = s1; ssum[1][threadIdx.x >> 6] = s2; ssum[2][threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsafeAtomicAdd(&sums[0], ssum[0][0] + ssum[0][1] + ssum[0][2] + ssum[0][3]);
        unsafeAtomicAdd(&sums[1], ssum[1][0] + ssum[1][1] + ssum[1][2] + ssum[1][3]);
        unsafeAtomicAdd(&sums[2], ssum[2][0] + ssum[2][1] + ssum[2][2] + ssum[2][3]);
        atomicMin(&mm[0], smm[0]); atomicMax(&mm[1], smm[1]);
    }
}

// ---- create_nonzero_mask (cropping.py:7-17) inside the crop box: 0 = some channel non-zero, 1 = background, and the
// background voxels on the box faces start as 2 = outside.  binary_fill_holes == "background that cannot reach the
// border through 6-connected background stays inside"; restricting it to the box is exact because everything beyond
// a box face that is not the image border is background connected to the image border.
__global__ __launch_bounds__(256) void mask_init_kernel(const float *raw, PrepParams p, uint8_t *m) {
    const long long n = p.ext[0] * p.ext[1] * p.ext[2], nraw = p.g.s[0] * p.g.s[1] * p.g.s[2];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long t2 = i % p.ext[2], t1 = (i / p.ext[2]) % p.ext[1], t0 = i / (p.ext[2] * p.ext[1]);
    const long long r = raw_index(p, t0, t1, t2);
    bool nz = false;
    for (int c = 0; c < p.g.C; ++c) nz |= raw[(long long)c * nraw + r] != 0.f;
    const bool face = t0 == 0 || t1 == 0 || t2 == 0 || t0 == p.ext[0] - 1 || t1 == p.ext[1] - 1 || t2 == p.ext[2] - 1;
    m[i] = nz ? 0 : (face ? 2 : 1);
}

// One sweep: every thread owns a line along `axis` and carries "outside" forwards and backwards through runs of
// background; a voxel also turns outside when a neighbour on another line already is (checked along the way).
__global__ __launch_bounds__(256) void mask_sweep_kernel(PrepParams p, int axis, uint8_t *m, int *changed) {
    const long long e[3] = {p.ext[0], p.ext[1], p.ext[2]};
    const int a1 = axis == 0 ? 1 : 0, a2 = axis == 2 ? 1 : 2;
    const long long lines = e[a1] * e[a2];
    const long long l = (long long)blockIdx.x * 256 + threadIdx.x;
    if (l >= lines) return;
    const long long c1 = l / e[a2], c2 = l % e[a2];
    const long long st[3] = {e[1] * e[2], e[2], 1};
    const long long base = c1 * st[a1] + c2 * st[a2];
    bool any = false;
    for (int dir = 0; dir < 2; ++dir) {
        bool carry = false;
        for (long long k = 0; k < e[axis]; ++k) {
            const long long pos = dir ? e[axis] - 1 - k : k;
            const long long i = base + pos * st[axis];
            uint8_t v = m[i];
            if (v == 1) {
                bool out = carry;
                if (!out) {                                   // side neighbours on the two other axes
                    if (c1 > 0) out |= m[i - st[a1]] == 2;
                    if (c1 < e[a1] - 1) out |= m[i + st[a1]] == 2;
                    if (c2 > 0) out |= m[i - st[a2]] == 2;
                    if (c2 < e[a2] - 1) out |= m[i + st[a2]] == 2;
                }
                if (out) { m[i] = 2; v = 2; any = true; }
            }
            carry = v == 2;
        }
    }
    if (any) *changed = 1;
}

// ---- out[c][t0][t1][t2] = normalise_c(raw[c][transposed, cropped])
__global__ __launch_bounds__(256) void apply_kernel(const float *raw, PrepParams p, float *out) {
    const long long n = p.ext[0] * p.ext[1] * p.ext[2], nraw = p.g.s[0] * p.g.s[1] * p.g.s[2];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * p.g.C) return;
    const int c = (int)(i / n);
    const long long j = i - (long long)c * n;
    const long long t2 = j % p.ext[2], t1 = (j / p.ext[2]) % p.ext[1], t0 = j / (p.ext[2] * p.ext[1]);
    float v = raw[(long long)c * nraw + raw_index(p, t0, t1, t2)];
    switch (p.scheme[c]) {
    case FNN_NORM_CT:                       // np.clip, -= mean, /= max(std, 1e-8): three fp32 roundings
        v = v < p.lower[c] ? p.lower[c] : (v > p.upper[c] ? p.upper[c] : v);      // NaN stays NaN like np.clip
        v = __fdiv_rn(__fsub_rn(v, p.a[c]), p.b[c]);
        break;
    case FNN_NORM_ZSCORE:                   // image[mask] = (image[mask] - mean) / max(std, 1e-8): outside stays as it is
        if (!(p.use_mask[c] && p.mask[j] == 2)) v = __fdiv_rn(__fsub_rn(v, p.a[c]), p.b[c]);
        break;
    case FNN_NORM_RESCALE01:
        v = __fdiv_rn(__fsub_rn(v, p.a[c]), p.b[c]);
        break;
    case FNN_NORM_RGB01:
        v = __fdiv_rn(v, 255.f);
        break;
    default: break;
    }
    out[i] = v;
}

// ---- labels [b0][b1][b2] (transposed, cropped) -> out [s0][s1][s2] in the original axis order
template <typename LT>
__global__ __launch_bounds__(256) void revert_kernel(const LT *seg, long long lo0, long long lo1, long long lo2, long long e0,
                                                     long long e1, long long e2, long long o0, long long o1, long long o2,
                                                     int tb0, int tb1, int tb2, LT *out) {
    const long long n = o0 * o1 * o2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long oc[3] = {i / (o1 * o2), (i / o2) % o1, i % o2};
    long long t[3];
    t[tb0] = oc[0]; t[tb1] = oc[1]; t[tb2] = oc[2];                     // out axis j = transposed axis tb[j]
    const long long d0 = t[0] - lo0, d1 = t[1] - lo1, d2 = t[2] - lo2;
    LT v = 0;
    if (d0 >= 0 && d0 < e0 && d1 >= 0 && d1 < e1 && d2 >= 0 && d2 < e2) v = seg[(d0 * e1 + d1) * e2 + d2];
    out[i] = v;
}

// ---- logits [H][b0][b1][b2] (transposed, cropped grid) -> probabilities [H][s0][s1][s2] + labels [s0][s1][s2] on the
// original grid: apply_inference_nonlin (fp32 softmax over the heads / sigmoid for regions), the label rule on the
// probabilities, both reverted croppings (background probability 1 outside the box for plain labels), transposes back.
// expf / the division follow the device's fp32 library: probabilities agree with torch's CPU softmax to ~1e-7.
template <typename IT, typename LT>
__global__ __launch_bounds__(256) void export_prob_kernel(const IT *logits, int H, const int *order, long long lo0, long long lo1,
                                                          long long lo2, long long e0, long long e1, long long e2, long long o0,
                                                          long long o1, long long o2, int tb0, int tb1, int tb2, float *probs,
                                                          LT *labels) {
    const long long n = o0 * o1 * o2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long oc[3] = {i / (o1 * o2), (i / o2) % o1, i % o2};
    long long t[3];
    t[tb0] = oc[0]; t[tb1] = oc[1]; t[tb2] = oc[2];
    const long long d0 = t[0] - lo0, d1 = t[1] - lo1, d2 = t[2] - lo2;
    if (!(d0 >= 0 && d0 < e0 && d1 >= 0 && d1 < e1 && d2 >= 0 && d2 < e2)) {
        for (int h = 0; h < H; ++h) probs[(size_t)h * n + i] = (!order && h == 0) ? 1.f : 0.f;
        labels[i] = 0;
        return;
    }
    const size_t plane = (size_t)e0 * e1 * e2, v = ((size_t)d0 * e1 + d1) * e2 + d2;
    if (order) {
        int seg = 0;
        for (int h = 0; h < H; ++h) {
            const float x = (float)logits[h * plane + v];
            const float pr = 1.f / (1.f + expf(-x));
            probs[(size_t)h * n + i] = pr;
            if (pr > 0.5f) seg = order[h];
        }
        labels[i] = (LT)seg;
        return;
    }
    float mx = (float)logits[v];
    for (int h = 1; h < H; ++h) mx = fmaxf(mx, (float)logits[h * plane + v]);
    float sum = 0.f;
    for (int h = 0; h < H; ++h) sum += expf((float)logits[h * plane + v] - mx);
    float best = -1.f; int arg = 0;
    for (int h = 0; h < H; ++h) {
        const float pr = expf((float)logits[h * plane + v] - mx) / sum;
        probs[(size_t)h * n + i] = pr;
        if (pr > best) { best = pr; arg = h; }                   // first maximum wins (numpy / torch argmax)
    }
    labels[i] = (LT)arg;
}

static int check_perm(const int32_t t[3]) {
    int seen = 0;
    for (int i = 0; i < 3; ++i) { if (t[i] < 0 || t[i] > 2) return -1; seen |= 1 << t[i]; }
    return seen == 7 ? 0 : -1;
}

}  // namespace

extern "C" {

int fnn_nonzero_bbox(const float *raw, const int64_t shape[4], const int32_t transpose_forward[3], int64_t bbox[6], void *stream) {
    if (!raw || !shape || !transpose_forward || !bbox) return fail_msg(FNN_E_INVALID, "NULL argument");
    if (check_perm(transpose_forward) != 0) return fail_msg(FNN_E_INVALID, "transpose_forward is not a permutation of (0, 1, 2)");
    if (shape[0] < 1 || shape[0] > 8 || shape[1] < 1 || shape[2] < 1 || shape[3] < 1) return fail_msg(FNN_E_INVALID, "bad shape (1..8 channels)");
    if (!dev_ptr(raw)) return fail_msg(FNN_E_INVALID, "fnn_nonzero_bbox needs a device pointer (no CPU path)");
    hipStream_t st = (hipStream_t)stream;
    PrepGeom g{};
    for (int d = 0; d < 3; ++d) { g.s[d] = shape[1 + d]; g.tf[d] = transpose_forward[d]; }
    g.C = (int)shape[0];
    int *box = nullptr;
    if (hipMalloc((void **)&box, 6 * sizeof(int)) != hipSuccess) return fail_msg(FNN_E_HIP, "hipMalloc failed");
    const int init[6] = {INT_MAX, INT_MAX, INT_MAX, -1, -1, -1};
    int h[6];
    hipError_t r = hipMemcpyAsync(box, init, sizeof(init), hipMemcpyHostToDevice, st);
    const long long n = g.s[0] * g.s[1] * g.s[2];
    long long blocks = (n + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (r == hipSuccess) { hipLaunchKernelGGL(bbox_kernel, dim3((unsigned)blocks), dim3(256), 0, st, raw, g, box); r = hipGetLastError(); }
    if (r == hipSuccess) r = hipMemcpyAsync(h, box, sizeof(h), hipMemcpyDeviceToHost, st);
    if (r == hipSuccess) r = hipStreamSynchronize(st);
    (void)hipFree(box);
    if (r != hipSuccess) return fail_msg(FNN_E_HIP, hipGetErrorString(r));
    for (int a = 0; a < 3; ++a) {
        if (h[3 + a] < 0) { bbox[2 * a] = 0; bbox[2 * a + 1] = shape[1 + transpose_forward[a]]; }   // empty mask: the full extent
        else { bbox[2 * a] = h[a]; bbox[2 * a + 1] = h[3 + a] + 1; }
    }
    return FNN_OK;
}

int fnn_preprocess(const float *raw, const int64_t shape[4], const int32_t transpose_forward[3], const int64_t bbox[6],
                   const fnn_norm_desc *norm, float *out, void *stream) {
    if (!raw || !shape || !transpose_forward || !bbox || !norm || !out) return fail_msg(FNN_E_INVALID, "NULL argument");
    if (check_perm(transpose_forward) != 0) return fail_msg(FNN_E_INVALID, "transpose_forward is not a permutation of (0, 1, 2)");
    if (shape[0] < 1 || shape[0] > 8) return fail_msg(FNN_E_INVALID, "1..8 channels");
    if (!dev_ptr(raw) || !dev_ptr(out)) return fail_msg(FNN_E_INVALID, "fnn_preprocess needs device pointers (no CPU path)");
    hipStream_t st = (hipStream_t)stream;
    PrepParams p{};
    for (int d = 0; d < 3; ++d) { p.g.s[d] = shape[1 + d]; p.g.tf[d] = transpose_forward[d]; }
    p.g.C = (int)shape[0];
    for (int a = 0; a < 3; ++a) {
        p.lo[a] = bbox[2 * a]; p.ext[a] = bbox[2 * a + 1] - bbox[2 * a];
        if (p.lo[a] < 0 || p.ext[a] < 1 || bbox[2 * a + 1] > shape[1 + transpose_forward[a]]) return fail_msg(FNN_E_INVALID, "bbox outside the (transposed) image");
    }
    const long long n = p.ext[0] * p.ext[1] * p.ext[2];
    double *sums = nullptr;
    if (hipMalloc((void **)&sums, 8 * 5 * sizeof(double)) != hipSuccess) return fail_msg(FNN_E_HIP, "hipMalloc failed");
    hipError_t r = hipSuccess;
    bool want_mask = false;
    for (int c = 0; c < p.g.C; ++c) {
        p.use_mask[c] = norm[c].scheme == FNN_NORM_ZSCORE && norm[c].use_mask;
        want_mask |= p.use_mask[c] != 0;
    }
    uint8_t *mask = nullptr;
    if (want_mask) {
        // seg = where(binary_fill_holes(nonzero_mask), 0, -1) of crop_to_nonzero (cropping.py:19-39), as a byte map
        int *changed = nullptr;
        if (hipMalloc((void **)&mask, (size_t)n) != hipSuccess || hipMalloc((void **)&changed, sizeof(int)) != hipSuccess) {
            (void)hipFree(sums); (void)hipFree(mask);
            return fail_msg(FNN_E_HIP, "hipMalloc failed");
        }
        hipLaunchKernelGGL(mask_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, raw, p, mask);
        for (int iter = 0; iter < 4096 && r == hipSuccess; ++iter) {
            int h = 0;
            r = hipMemcpyAsync(changed, &h, sizeof(int), hipMemcpyHostToDevice, st);
            for (int axis = 0; axis < 3 && r == hipSuccess; ++axis) {
                const long long lines = n / p.ext[axis];
                hipLaunchKernelGGL(mask_sweep_kernel, dim3((unsigned)((lines + 255) / 256)), dim3(256), 0, st, p, axis, mask, changed);
                r = hipGetLastError();
            }
            if (r == hipSuccess) r = hipMemcpyAsync(&h, changed, sizeof(int), hipMemcpyDeviceToHost, st);
            if (r == hipSuccess) r = hipStreamSynchronize(st);
            if (r != hipSuccess || !h) break;
        }
        (void)hipFree(changed);
        p.mask = mask;
    }
    for (int c = 0; c < p.g.C && r == hipSuccess; ++c) {
        const fnn_norm_desc &d = norm[c];
        p.scheme[c] = d.scheme;
        p.a[c] = 0.f; p.b[c] = 1.f; p.lower[c] = d.lower; p.upper[c] = d.upper;
        if (d.scheme == FNN_NORM_CT) {
            p.a[c] = d.mean; p.b[c] = d.std > 1e-8f ? d.std : 1e-8f;           // max(std_intensity, 1e-8)
        } else if (d.scheme == FNN_NORM_ZSCORE || d.scheme == FNN_NORM_RESCALE01) {
            // statistics of this channel over the crop box
            double *sc = sums + c * 5;
            unsigned *mm = (unsigned *)(sc + 3);
            const double z[3] = {0, 0, 0};
            const unsigned mi[2] = {0xffffffffu, 0u};
            double hs[3]; unsigned hm[2];
            r = hipMemcpyAsync(sc, z, sizeof(z), hipMemcpyHostToDevice, st);
            if (r == hipSuccess) r = hipMemcpyAsync(mm, mi, sizeof(mi), hipMemcpyHostToDevice, st);
            long long blocks = (n + 255) / 256;
            if (blocks > 256 * 16) blocks = 256 * 16;
            if (r == hipSuccess) { hipLaunchKernelGGL(stats_kernel, dim3((unsigned)blocks), dim3(256), 0, st, raw, p, c, sc, mm); r = hipGetLastError(); }
            if (r == hipSuccess) r = hipMemcpyAsync(hs, sc, sizeof(hs), hipMemcpyDeviceToHost, st);
            if (r == hipSuccess) r = hipMemcpyAsync(hm, mm, sizeof(hm), hipMemcpyDeviceToHost, st);
            if (r == hipSuccess) r = hipStreamSynchronize(st);
            if (r != hipSuccess) break;
            auto unord = [](unsigned u) { u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u; return __builtin_bit_cast(float, u); };
            if (d.scheme == FNN_NORM_ZSCORE) {
                const double cntd = hs[2] > 0 ? hs[2] : 1.0;             // voxels inside the mask (all of them without one)
                const double mean = hs[0] / cntd;
                double var = hs[1] / cntd - mean * mean;
                var = var > 0 ? var : 0;
                const float stdf = (float)sqrt(var);
                p.a[c] = (float)mean; p.b[c] = stdf > 1e-8f ? stdf : 1e-8f;
            } else {
                const float mn = unord(hm[0]), mx = unord(hm[1]);
                const float range = mx - mn;                                   // = (image - image.min()).max() in fp32
                p.a[c] = mn; p.b[c] = range > 1e-8f ? range : 1e-8f;
            }
        } else if (d.scheme != FNN_NORM_NONE && d.scheme != FNN_NORM_RGB01) {
            (void)hipFree(sums); (void)hipFree(mask);
            return fail_msg(FNN_E_UNSUPPORTED, "unknown normalisation scheme");
        }
    }
    if (r == hipSuccess) {
        const long long total = n * p.g.C;
        hipLaunchKernelGGL(apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, raw, p, out);
        r = hipGetLastError();
    }
    if (r == hipSuccess) r = hipStreamSynchronize(st);
    (void)hipFree(sums);
    (void)hipFree(mask);
    if (r != hipSuccess) return fail_msg(FNN_E_HIP, hipGetErrorString(r));
    return FNN_OK;
}

int fnn_revert_labels(const void *seg, int label_dtype, const int64_t bbox[6], const int64_t shape_before_cropping[3],
                      const int32_t transpose_backward[3], void *out, void *stream) {
    if (!seg || !bbox || !shape_before_cropping || !transpose_backward || !out) return fail_msg(FNN_E_INVALID, "NULL argument");
    if (check_perm(transpose_backward) != 0) return fail_msg(FNN_E_INVALID, "transpose_backward is not a permutation of (0, 1, 2)");
    if (label_dtype != FNN_LABEL_U8 && label_dtype != FNN_LABEL_U16) return fail_msg(FNN_E_INVALID, "unknown label dtype");
    if (!dev_ptr(seg) || !dev_ptr(out)) return fail_msg(FNN_E_INVALID, "fnn_revert_labels needs device pointers (no CPU path)");
    long long lo[3], ext[3], o[3];
    for (int a = 0; a < 3; ++a) {
        lo[a] = bbox[2 * a]; ext[a] = bbox[2 * a + 1] - bbox[2 * a];
        if (lo[a] < 0 || ext[a] < 1 || bbox[2 * a + 1] > shape_before_cropping[a]) return fail_msg(FNN_E_INVALID, "bbox outside shape_before_cropping");
    }
    for (int j = 0; j < 3; ++j) o[j] = shape_before_cropping[transpose_backward[j]];
    const long long n = o[0] * o[1] * o[2];
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (label_dtype == FNN_LABEL_U16)
        hipLaunchKernelGGL(revert_kernel<uint16_t>, grid, dim3(256), 0, st, (const uint16_t *)seg, lo[0], lo[1], lo[2], ext[0], ext[1], ext[2],
                           o[0], o[1], o[2], transpose_backward[0], transpose_backward[1], transpose_backward[2], (uint16_t *)out);
    else
        hipLaunchKernelGGL(revert_kernel<uint8_t>, grid, dim3(256), 0, st, (const uint8_t *)seg, lo[0], lo[1], lo[2], ext[0], ext[1], ext[2],
                           o[0], o[1], o[2], transpose_backward[0], transpose_backward[1], transpose_backward[2], (uint8_t *)out);
    hipError_t r = hipGetLastError();
    if (r == hipSuccess) r = hipStreamSynchronize(st);
    if (r != hipSuccess) return fail_msg(FNN_E_HIP, hipGetErrorString(r));
    return FNN_OK;
}

int fnn_export_probabilities(const void *logits, int logits_dtype, int heads, const int32_t *regions_class_order,
                             const int64_t bbox[6], const int64_t shape_before_cropping[3],
                             const int32_t transpose_backward[3], float *probs, void *labels, int label_dtype, void *stream) {
    if (!logits || !bbox || !shape_before_cropping || !transpose_backward || !probs || !labels) return fail_msg(FNN_E_INVALID, "NULL argument");
    if (check_perm(transpose_backward) != 0) return fail_msg(FNN_E_INVALID, "transpose_backward is not a permutation of (0, 1, 2)");
    if (label_dtype != FNN_LABEL_U8 && label_dtype != FNN_LABEL_U16) return fail_msg(FNN_E_INVALID, "unknown label dtype");
    if (logits_dtype != FNN_OUT_F16 && logits_dtype != FNN_OUT_F32) return fail_msg(FNN_E_INVALID, "unknown logits dtype");
    if (heads < 1 || heads > 4096) return fail_msg(FNN_E_INVALID, "bad number of heads");
    if (!dev_ptr(logits) || !dev_ptr(probs) || !dev_ptr(labels)) return fail_msg(FNN_E_INVALID, "fnn_export_probabilities needs device pointers (no CPU path)");
    long long lo[3], ext[3], o[3];
    for (int a = 0; a < 3; ++a) {
        lo[a] = bbox[2 * a]; ext[a] = bbox[2 * a + 1] - bbox[2 * a];
        if (lo[a] < 0 || ext[a] < 1 || bbox[2 * a + 1] > shape_before_cropping[a]) return fail_msg(FNN_E_INVALID, "bbox outside shape_before_cropping");
    }
    for (int j = 0; j < 3; ++j) o[j] = shape_before_cropping[transpose_backward[j]];
    const long long n = o[0] * o[1] * o[2];
    hipStream_t st = (hipStream_t)stream;
    int *order = nullptr;
    hipError_t r = hipSuccess;
    if (regions_class_order) {
        if (hipMalloc((void **)&order, heads * sizeof(int)) != hipSuccess) return fail_msg(FNN_E_HIP, "hipMalloc failed");
        r = hipMemcpyAsync(order, regions_class_order, heads * sizeof(int), hipMemcpyHostToDevice, st);
    }
    const dim3 grid((unsigned)((n + 255) / 256));
    const int tb0 = transpose_backward[0], tb1 = transpose_backward[1], tb2 = transpose_backward[2];
#define FNN_EXPORT_LAUNCH(IT, LT)                                                                                              \
    hipLaunchKernelGGL((export_prob_kernel<IT, LT>), grid, dim3(256), 0, st, (const IT *)logits, heads, order, lo[0], lo[1],   \
                       lo[2], ext[0], ext[1], ext[2], o[0], o[1], o[2], tb0, tb1, tb2, probs, (LT *)labels)
    if (r == hipSuccess) {
        if (logits_dtype == FNN_OUT_F16) {
            if (label_dtype == FNN_LABEL_U16) FNN_EXPORT_LAUNCH(_Float16, uint16_t); else FNN_EXPORT_LAUNCH(_Float16, uint8_t);
        } else {
            if (label_dtype == FNN_LABEL_U16) FNN_EXPORT_LAUNCH(float, uint16_t); else FNN_EXPORT_LAUNCH(float, uint8_t);
        }
        r = hipGetLastError();
    }
#undef FNN_EXPORT_LAUNCH
    if (r == hipSuccess) r = hipStreamSynchronize(st);
    if (order) (void)hipFree(order);
    if (r != hipSuccess) return fail_msg(FNN_E_HIP, hipGetErrorString(r));
    return FNN_OK;
}

}  // extern "C"
