// ops_api.hip - single-op entry points of include/fnn.h (fnn_op_*).
//
// Parity tests drive the HIP kernels one at a time through these: host float32
// NCDHW tensors in, host float32 out.  The wrapper does what the engine does
// around a kernel: convert to fp16 channels-last with padded channels, pack the
// weights into MFMA fragment order, provide the producer-side InstanceNorm
// statistics, launch, convert back.
#include "fnn_device.h"
#include "../../include/fnn.h"

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

inline int pad16(int c) { return (c + 15) / 16 * 16; }
inline uint16_t f2h_bits(float f) { f16 h = (f16)f; uint16_t b; memcpy(&b, &h, 2); return b; }
inline float h2f_bits(uint16_t b) { f16 h; memcpy(&h, &b, 2); return (float)h; }

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    bool alloc(size_t n) { return hipMalloc(&p, n ? n : 16) == hipSuccess; }
    template <class T> T *as() { return (T *)p; }
};

// NCDHW fp32 -> NDHWC fp16 with channel padding; also the per-(n,c) statistics of the rounded values
// FNN_OP_CHUNK_MAJOR=1 (next to FNN_KNOBS=1; read per call): tensors of more than 16 channels travel chunk-major
// ([C / 16][voxels][16]; fnn_device.h, SrcDesc) through the conv / transposed-conv ops, as in the engine
bool op_chunk_major(int cp) { const char *v = fnn_knob("FNN_OP_CHUNK_MAJOR"); return v && v[0] != '0' && cp > 16; }

void to_ndhwc(const float *x, int n, int c, int cp, size_t vox, std::vector<uint16_t> &out, std::vector<double> *stats, bool cm = false) {
    out.assign((size_t)n * vox * cp, 0);
    if (stats) stats->assign((size_t)n * FNN_STAT_REPL * cp * 2, 0.0);
    for (int b = 0; b < n; ++b)
        for (int ch = 0; ch < c; ++ch) {
            double s1 = 0, s2 = 0;
            const float *src = x + ((size_t)b * c + ch) * vox;
            for (size_t v = 0; v < vox; ++v) {
                const uint16_t hb = f2h_bits(src[v]);
                out[cm ? ((size_t)b * cp / 16 + ch / 16) * vox * 16 + v * 16 + ch % 16 : ((size_t)b * vox + v) * cp + ch] = hb;
                const double f = h2f_bits(hb);
                s1 += f; s2 += f * f;
            }
            if (stats) {
                // spread over two replicas to exercise the replica sum
                double *st = stats->data() + ((size_t)b * FNN_STAT_REPL * cp + ch) * 2;
                st[0] = s1 * 0.25; st[1] = s2 * 0.25;
                st[(size_t)3 * cp * 2] = s1 * 0.75; st[(size_t)3 * cp * 2 + 1] = s2 * 0.75;
            }
        }
}

struct SrcHolder {
    DevBuf act, stats, gamma, beta, ss, ssh;
    SrcDesc d{};
};

bool make_src(SrcHolder &h, const float *x, int n, int c, size_t vox, const float *gamma, const float *beta, float slope,
              bool layout_aware = false) {
    const int cp = pad16(c);
    std::vector<uint16_t> a;
    std::vector<double> st;
    const bool cm = layout_aware && op_chunk_major(cp);
    to_ndhwc(x, n, c, cp, vox, a, gamma ? &st : nullptr, cm);
    if (!h.act.alloc(a.size() * 2)) return false;
    if (hipMemcpy(h.act.p, a.data(), a.size() * 2, hipMemcpyHostToDevice) != hipSuccess) return false;
    h.d.ptr = h.act.as<f16>(); h.d.C = cp; h.d.slope = 1.f;
    if (cm) { h.d.vs = 16; h.d.cs = 16LL * (long long)vox; }
    if (gamma) {
        std::vector<float> g(cp, 0.f), b(cp, 0.f);
        for (int i = 0; i < c; ++i) { g[i] = gamma[i]; b[i] = beta ? beta[i] : 0.f; }
        if (!h.stats.alloc(st.size() * 8) || !h.gamma.alloc(cp * 4) || !h.beta.alloc(cp * 4)) return false;
        (void)hipMemcpy(h.stats.p, st.data(), st.size() * 8, hipMemcpyHostToDevice);
        (void)hipMemcpy(h.gamma.p, g.data(), cp * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(h.beta.p, b.data(), cp * 4, hipMemcpyHostToDevice);
        // what the engine does between producer and consumer: statistics -> (scale, shift)
        if (!h.ss.alloc((size_t)n * cp * 8) || !h.ssh.alloc((size_t)n * cp * 4)) return false;
        StatsFinalizeParams q{};
        q.stats = h.stats.as<double>(); q.gamma = h.gamma.as<float>(); q.beta = h.beta.as<float>();
        q.ss = h.ss.as<float>(); q.ssh = h.ssh.as<unsigned short>(); q.C = cp; q.nrep = FNN_STAT_REPL; q.inv_count = 1.f / (float)vox; q.eps = 1e-5f;
        if (launch_stats_finalize(q, n, 0) != 0) return false;
        h.d.ss = h.ss.as<float>();
        h.d.ssh = h.ssh.as<unsigned short>();
        h.d.slope = slope;
    }
    return true;
}

}  // namespace

// kernel variants of the last fnn_op_* call of this thread (fnn_note_kernel at the launch sites): fnn_op_last_kernels
static thread_local std::vector<std::string> g_op_kernels;
struct OpKlog {
    OpKlog() { g_op_kernels.clear(); fnn_klog_target(&g_op_kernels); }
    ~OpKlog() { fnn_klog_target(nullptr); }
};

extern "C" {

int fnn_op_conv3d(int device, int n, const int dims[3],
                  const float *x, int cin, const float *gamma1, const float *beta1, float slope1,
                  const float *x2, int cin2, const float *gamma2, const float *beta2, float slope2,
                  const float *w, const float *bias, int cout, const int k[3], const int stride[3],
                  float *y, double *stats_out) {
    if (!x || !w || !y || !dims || !k || !stride || n < 1 || cin < 1 || cout < 1) return FNN_E_INVALID;
    if (hipSetDevice(device) != hipSuccess) return FNN_E_HIP;
    OpKlog klog;
    const size_t vox = (size_t)dims[0] * dims[1] * dims[2];
    const int nsrc = x2 ? 2 : 1;
    SrcHolder s1, s2;
    if (!make_src(s1, x, n, cin, vox, gamma1, beta1, slope1, true)) return FNN_E_HIP;
    if (x2 && !make_src(s2, x2, n, cin2, vox, gamma2, beta2, slope2, true)) return FNN_E_HIP;
    const int cp1 = pad16(cin), cp2 = x2 ? pad16(cin2) : 0, cop = pad16(cout);
    const int T = k[0] * k[1] * k[2], cin_tot = cin + (x2 ? cin2 : 0);
    ConvParams p{};
    p.n_src = nsrc; p.src[0] = s1.d;
    if (x2) p.src[1] = s2.d; else { p.src[1] = s1.d; p.src[1].C = 0; }
    p.N = n; p.Di = dims[0]; p.Hi = dims[1]; p.Wi = dims[2];
    p.kd = k[0]; p.kh = k[1]; p.kw = k[2]; p.sd = stride[0]; p.sh = stride[1]; p.sw = stride[2];
    p.pd = (k[0] - 1) / 2; p.ph = (k[1] - 1) / 2; p.pw = (k[2] - 1) / 2;
    p.Do = (p.Di + 2 * p.pd - p.kd) / p.sd + 1; p.Ho = (p.Hi + 2 * p.ph - p.kh) / p.sh + 1; p.Wo = (p.Wi + 2 * p.pw - p.kw) / p.sw + 1;
    p.Cout = cop; p.chunks = (cp1 + cp2) / 16;
    p.packing = conv3d_packing(p); p.ksteps = conv3d_ksteps(p.packing, T);
    p.stats_slots = conv3d_stats_slots(p);
    const size_t slots = (size_t)p.stats_slots;
    p.tiles_d = (p.Do + FNN_TILE_D - 1) / FNN_TILE_D; p.tiles_h = (p.Ho + FNN_TILE_H - 1) / FNN_TILE_H;
    p.tiles_w = (p.Wo + FNN_TILE_W - 1) / FNN_TILE_W;
    p.tile_d = FNN_TILE_D;
    // pack weights [cout][cin_tot][T] -> [cb][chunk][ks][lane][8]
    if (p.packing == FNN_PACK_ZP) p.chunks = conv_zp_chunks(cp1, cp2);
    std::vector<uint16_t> wp((size_t)(cop / 16) * p.chunks * p.ksteps * 512, 0);
    if (p.packing == FNN_PACK_ZP) conv_zp_pack(w, cout, cop, cin, cp1, x2 ? cin2 : 0, cp2, wp.data());
    else
    for (int cb = 0; cb < cop / 16; ++cb)
        for (int ch = 0; ch < p.chunks; ++ch)
            for (int ks = 0; ks < p.ksteps; ++ks)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int kk = 8 * (lane >> 4) + j, tap = conv3d_kstep_tap(p.packing, ks, kk >> 4, T), c = ch * 16 + (kk & 15);
                        const int co = conv3d_pack_cout(p.packing, cop / 16, cb, lane & 15);
                        int src = 0, cl = c;
                        if (c >= cp1) { src = 1; cl = c - cp1; }
                        const int creal = src ? cin2 : cin;
                        float v = 0.f;
                        if (tap >= 0 && co < cout && cl < creal) v = w[((size_t)co * cin_tot + (src ? cin : 0) + cl) * T + tap];
                        wp[((((size_t)cb * p.chunks + ch) * p.ksteps + ks) * 64 + lane) * 8 + j] = f2h_bits(v);
                    }
    std::vector<float> bp(cop, 0.f);
    if (bias) for (int i = 0; i < cout; ++i) bp[i] = bias[i];
    const size_t ovox = (size_t)p.Do * p.Ho * p.Wo;
    DevBuf dw, db, dout, dst;
    // (+ 1 KB: the ZR kernels' last weight load of a block reads one wave past it - the bytes are dropped, the address must exist)
    if (!dw.alloc(wp.size() * 2 + 1024) || !db.alloc(cop * 4) || !dout.alloc((size_t)n * ovox * cop * 2) ||
        !dst.alloc((size_t)n * slots * cop * 16)) return FNN_E_HIP;
    (void)hipMemcpy(dw.p, wp.data(), wp.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(db.p, bp.data(), cop * 4, hipMemcpyHostToDevice);
    (void)hipMemset(dst.p, 0, (size_t)n * slots * cop * 16);
    (void)hipMemset(dout.p, 0, (size_t)n * ovox * cop * 2);
    p.wpk = dw.as<f16>(); p.bias = db.as<float>(); p.out = dout.as<f16>(); p.stats_out = dst.as<double>();
    const bool ocm = op_chunk_major(cop);
    if (ocm) { p.out_vs = 16; p.out_cs = 16LL * (long long)ovox; }
#ifdef FNN_STAMPS
    DevBuf ddbg;
    const size_t dbg_n = (size_t)1 << 20;
    if (!ddbg.alloc(dbg_n * 8)) return FNN_E_HIP;
    (void)hipMemset(ddbg.p, 0, dbg_n * 8);
    p.dbg = ddbg.as<unsigned long long>();
    (void)launch_conv3d(p, 0);                      // warm-up
    (void)hipDeviceSynchronize();
    (void)hipMemset(dst.p, 0, (size_t)n * slots * cop * 16);
#endif
    if (fnn_knob("FNN_OP_TIME")) {                    // diagnostic: mean duration of 30 launches of this layer (after 2 warm-ups)
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int i = 0; i < 2; ++i) (void)launch_conv3d(p, 0);
        (void)hipEventRecord(e0, 0);
        for (int i = 0; i < 30; ++i) (void)launch_conv3d(p, 0);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        fprintf(stderr, "[op time] conv3d %d+%d -> %d at %dx%dx%d, N = %d: %.1f us per launch\n", cin, cin2, cout, p.Di, p.Hi, p.Wi, n, ms * (1000.f / 30.f));
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        (void)hipMemset(dst.p, 0, (size_t)n * slots * cop * 16);
    }
#ifdef FNN_STAMPS
    hipEvent_t ev0, ev1;
    (void)hipEventCreate(&ev0); (void)hipEventCreate(&ev1);
    (void)hipEventRecord(ev0, 0);
#endif
    const int rc = launch_conv3d(p, 0);
    if (rc != 0) return rc == -1 ? FNN_E_UNSUPPORTED : FNN_E_HIP;
#ifdef FNN_STAMPS
    (void)hipEventRecord(ev1, 0);
#endif
    if (hipDeviceSynchronize() != hipSuccess) return FNN_E_HIP;
#ifdef FNN_STAMPS
    {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, ev0, ev1);
        fprintf(stderr, "[stamps] launch %.1f us; ", ms * 1000.f);
        std::vector<unsigned long long> h(dbg_n);
        (void)hipMemcpy(h.data(), ddbg.p, dbg_n * 8, hipMemcpyDeviceToHost);
        double sum[12] = {0}; long cnt = 0;
        unsigned long long tmin = ~0ull, tmax = 0;
        for (size_t w = 0; w * 12 + 11 < dbg_n; ++w) {
            if (h[w * 12] == 0) continue;
            ++cnt;
            int last = 0;
            for (int i = 1; i < 12; ++i) if (h[w * 12 + i]) { sum[i] += (double)(h[w * 12 + i] - h[w * 12 + i - 1]); last = i; }
            if (h[w * 12] < tmin) tmin = h[w * 12];
            if (h[w * 12 + last] > tmax) tmax = h[w * 12 + last];
        }
        fprintf(stderr, "[stamps] %ld workgroups, kernel span %.0f ticks; mean ticks per segment:", cnt, (double)(tmax - tmin));
        for (int i = 1; i < 12; ++i) fprintf(stderr, " %d:%.0f", i, cnt ? sum[i] / cnt : 0.0);
        fprintf(stderr, "\n");
    }
#endif
    std::vector<uint16_t> ho((size_t)n * ovox * cop);
    std::vector<double> hs((size_t)n * slots * cop * 2);
    (void)hipMemcpy(ho.data(), dout.p, ho.size() * 2, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hs.data(), dst.p, hs.size() * 8, hipMemcpyDeviceToHost);
    for (int b = 0; b < n; ++b)
        for (int co = 0; co < cout; ++co) {
            for (size_t v = 0; v < ovox; ++v)
                y[((size_t)b * cout + co) * ovox + v] =
                    h2f_bits(ho[ocm ? ((size_t)b * cop / 16 + co / 16) * ovox * 16 + v * 16 + co % 16 : ((size_t)b * ovox + v) * cop + co]);
            if (stats_out) {
                double a = 0, q = 0;
                for (size_t r = 0; r < slots; ++r) {
                    a += hs[(((size_t)b * slots + r) * cop + co) * 2];
                    q += hs[(((size_t)b * slots + r) * cop + co) * 2 + 1];
                }
                stats_out[((size_t)b * cout + co) * 2] = a;
                stats_out[((size_t)b * cout + co) * 2 + 1] = q;
            }
        }
    return 0;
}

int fnn_op_conv_transpose3d(int device, int n, const int dims[3],
                            const float *x, int cin, const float *gamma1, const float *beta1, float slope1,
                            const float *w, const float *bias, int cout, const int stride[3], float *y) {
    if (!x || !w || !y || !dims || !stride || n < 1 || cin < 1 || cout < 1) return FNN_E_INVALID;
    if (hipSetDevice(device) != hipSuccess) return FNN_E_HIP;
    OpKlog klog;
    const size_t vox = (size_t)dims[0] * dims[1] * dims[2];
    SrcHolder s1;
    if (!make_src(s1, x, n, cin, vox, gamma1, beta1, slope1, true)) return FNN_E_HIP;
    const int cp = pad16(cin), cop = pad16(cout);
    const int taps = stride[0] * stride[1] * stride[2];
    TconvParams p{};
    p.src = s1.d; p.N = n; p.Di = dims[0]; p.Hi = dims[1]; p.Wi = dims[2];
    p.sd = stride[0]; p.sh = stride[1]; p.sw = stride[2];
    p.Cout = cop; p.nblk = cop / 16; p.ksteps = (cp + 31) / 32;
    std::vector<uint16_t> wp((size_t)taps * p.nblk * p.ksteps * 512, 0);
    for (int tap = 0; tap < taps; ++tap)
        for (int cb = 0; cb < p.nblk; ++cb)
            for (int ks = 0; ks < p.ksteps; ++ks)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int ci = ks * 32 + 8 * (lane >> 4) + j, co = cb * 16 + (lane & 15);
                        float v = 0.f;
                        if (ci < cin && co < cout) v = w[((size_t)ci * cout + co) * taps + tap];
                        wp[((((size_t)tap * p.nblk + cb) * p.ksteps + ks) * 64 + lane) * 8 + j] = f2h_bits(v);
                    }
    std::vector<float> bp(cop, 0.f);
    if (bias) for (int i = 0; i < cout; ++i) bp[i] = bias[i];
    const size_t ovox = vox * taps;
    DevBuf dw, db, dout;
    if (!dw.alloc(wp.size() * 2) || !db.alloc(cop * 4) || !dout.alloc((size_t)n * ovox * cop * 2)) return FNN_E_HIP;
    (void)hipMemcpy(dw.p, wp.data(), wp.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(db.p, bp.data(), cop * 4, hipMemcpyHostToDevice);
    (void)hipMemset(dout.p, 0, (size_t)n * ovox * cop * 2);
    p.wpk = dw.as<f16>(); p.bias = db.as<float>(); p.out = dout.as<f16>();
    const bool ocm = op_chunk_major(cop);
    if (ocm) { p.out_vs = 16; p.out_cs = 16LL * (long long)ovox; }
    if (launch_tconv(p, 0) != 0) return FNN_E_HIP;
    if (hipDeviceSynchronize() != hipSuccess) return FNN_E_HIP;
    std::vector<uint16_t> ho((size_t)n * ovox * cop);
    (void)hipMemcpy(ho.data(), dout.p, ho.size() * 2, hipMemcpyDeviceToHost);
    for (int b = 0; b < n; ++b)
        for (int co = 0; co < cout; ++co)
            for (size_t v = 0; v < ovox; ++v)
                y[((size_t)b * cout + co) * ovox + v] =
                    h2f_bits(ho[ocm ? ((size_t)b * cop / 16 + co / 16) * ovox * 16 + v * 16 + co % 16 : ((size_t)b * ovox + v) * cop + co]);
    return 0;
}

int fnn_op_last_kernels(char *buf, int cap) {
    std::string all;
    for (const std::string &k : g_op_kernels) { all += k; all += '\n'; }
    if (buf && cap > 0) { strncpy(buf, all.c_str(), (size_t)cap - 1); buf[cap - 1] = 0; }
    return (int)all.size() + 1;
}

int fnn_op_quotient_check(int device, unsigned long long counts[3]) {
    if (!counts) return FNN_E_INVALID;
    if (hipSetDevice(device) != hipSuccess) return FNN_E_HIP;
    DevBuf d;
    if (!d.alloc(24)) return FNN_E_HIP;
    (void)hipMemset(d.p, 0, 24);
    if (launch_quotient_check(d.as<unsigned long long>(), 0) != 0) return FNN_E_HIP;
    if (hipDeviceSynchronize() != hipSuccess) return FNN_E_HIP;
    (void)hipMemcpy(counts, d.p, 24, hipMemcpyDeviceToHost);
    return 0;
}

}  // extern "C"

// ----------------------------------------------------------------------------
// fnn_clock_probe_*: the shader clock under load (include/fnn.h)
// ----------------------------------------------------------------------------
namespace {
__global__ void clock_probe_kernel(unsigned long long *out, const int *flag, unsigned long long max_ticks) {
    if (threadIdx.x) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long i = 0;
    for (;; ++i) {                                            // ~16 us per turn
        __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127);
        if (__builtin_amdgcn_s_memrealtime() - r0 >= max_ticks) break;
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) break;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[0] = t1 - t0; out[1] = r1 - r0; out[2] = i;
}
struct ClockProbe {
    int device = 0;
    hipStream_t st = nullptr;
    unsigned long long *out = nullptr;
    int *flag = nullptr;                                      // mapped host memory
};
}  // namespace

int fnn_clock_probe_start(int device, double max_seconds, void **probe) {
    if (!probe || !(max_seconds > 0) || max_seconds > 60) return FNN_E_INVALID;
    *probe = nullptr;
    if (hipSetDevice(device) != hipSuccess) return FNN_E_HIP;
    ClockProbe *c = new ClockProbe;
    c->device = device;
    int *dflag = nullptr;
    if (hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking) != hipSuccess || hipMalloc((void **)&c->out, 32) != hipSuccess ||
        hipHostMalloc((void **)&c->flag, sizeof(int), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&dflag, c->flag, 0) != hipSuccess) {
        if (c->flag) (void)hipHostFree(c->flag);
        if (c->out) (void)hipFree(c->out);
        if (c->st) (void)hipStreamDestroy(c->st);
        delete c;
        return FNN_E_HIP;
    }
    *c->flag = 0;
    (void)hipMemsetAsync(c->out, 0, 32, c->st);
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, c->st, c->out, dflag, (unsigned long long)(max_seconds * 1e8));
    if (hipGetLastError() != hipSuccess) { (void)hipHostFree(c->flag); (void)hipFree(c->out); (void)hipStreamDestroy(c->st); delete c; return FNN_E_HIP; }
    *probe = c;
    return FNN_OK;
}

int fnn_clock_probe_stop(void *probe, double *ghz, double *seconds) {
    ClockProbe *c = (ClockProbe *)probe;
    if (!c) return FNN_E_INVALID;
    (void)hipSetDevice(c->device);
    __atomic_store_n(c->flag, 1, __ATOMIC_RELEASE);
    unsigned long long h[4] = {0, 0, 0, 0};
    const bool ok = hipStreamSynchronize(c->st) == hipSuccess && hipMemcpy(h, c->out, 32, hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipHostFree(c->flag); (void)hipFree(c->out); (void)hipStreamDestroy(c->st);
    delete c;
    if (!ok || h[1] == 0) return FNN_E_HIP;
    if (ghz) *ghz = (double)h[0] / (double)h[1] * 0.1;        // s_memrealtime: 100 MHz
    if (seconds) *seconds = (double)h[1] * 1e-8;
    return FNN_OK;
}
