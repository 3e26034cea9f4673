// conv2d_zp.hip - (1, 3, 3) stride-1 convolutions of 32 and more channels: whole-plane tiles with row-shift operand
// reuse (gfx950).  Round 6.
//
// Who runs here: every conv of a `2d` configuration (Conv2d 3 x 3, run as depth-1 3-D: the reference builds it from the same
// plans file, nnUNetDistillationTrainer.py:141-173 / get_network_from_plans) and the (1, 3, 3) stages of an anisotropic
// 3-D configuration once they carry 32 or more channels (experiment_planners/network_topology.py:30-108: the first
// stages of a thick-slice patch keep kernel 1 along the slice axis).  Before this kernel those layers ran the linear-tap
// kernel in (4 | 8) x 8 x 8 tiles - for a depth-1 tensor a quarter or an eighth of every MFMA's columns were voxels -
// at 3-19 % of the MFMA peak (profiles/r06_plan_sweep_v0.txt: 2-D networks 5 % of peak end to end).
//
// GEMM view: D[cout, voxel] = W[cout, k] X[k, voxel] on v_mfma_f32_16x16x32_f16 with K = 32 input channels of ONE tap
// (9 k-steps per 32-channel chunk, no padded tap slot; conv3d_zr_kernel pairs two taps of 16 channels).  A wave owns 16
// consecutive voxels of a row (one MFMA column block) in EVERY row of its TH-row strip: the "B" fragment of halo row p for
// the in-row tap dx is the operand of vertical tap dy for output row p - dy, for all three dy - read from LDS once, it
// feeds 3 NB MFMAs (the depth-shift reuse of conv3d_zr_kernel turned by 90 degrees).  Per (chunk, dx) a wave issues TH + 2
// activation reads and 3 NB weight reads for 3 TH NB MFMAs: 0.33 LDS reads per MFMA at TH = 8, NB = 2.
//
// LDS image of a chunk: [halo row][k-group line kq = 0 .. 3][column] x 16 B; line kq starts at kq QP + (kq >> 1) 64 with QP a
// multiple of 256.  A wave's ds_read_b128 of a fragment - lane (r, kq): column c0 + r + dx of line kq - touches, per hardware
// lane group of 16 (MI355X_MICROARCH.md, LDS: {0-3, 12-15, 20-27}, ...: eight lanes of one k-group and eight of its pair
// partner), sixteen consecutive 16-byte slots of two lines whose starts are congruent mod 256: conflict free for every dx.
// The staging's ds_write_b128 groups (eight lanes = four columns x the two 8-channel halves of one 16-channel record) hit
// lines kq and kq + 2, 64 B apart mod 128: eight different 16-byte slots of the stores' 128-byte bank period.  (The first
// form kept the halves in neighbouring lines 0 mod 128 apart: every staging store a 2-way conflict, SQ_LDS_BANK_CONFLICT 21 % of
// the LDS-active cycles - profiles/r06_pmc_zp.txt.)
//
// Everything else is the ZR kernel's: producer InstanceNorm + LeakyReLU applied while staging (packed fp16), buffer loads
// with hardware range checks (a column outside the tensor or a channel beyond the source: offset 0x80000000, zeros, no
// traffic), the next chunk's global loads in flight during the k-loop, bias as the accumulators' start, 16-byte
// channels-last (or chunk-major) stores in the interleaved channel order of conv3d_pack_cout, statistics of the rounded
// outputs per tile in a row of their own (no atomics).
#include "fnn_device.h"
#include "conv_common.h"
#include <cstdlib>
#include <cstring>

// ---------------------------------------------------------------------------
// host side: chunking and weight packing (FNN_PACK_ZP)
// ---------------------------------------------------------------------------
// 32-channel chunks, never across the two sources of a decoder conv: ceil(C0 / 32) + ceil(C1 / 32)
int conv_zp_chunks(int cin_pad0, int cin_pad1) { return (cin_pad0 + 31) / 32 + (cin_pad1 + 31) / 32; }

// [cout block][chunk][k-step = dx * 3 + dy][64 lanes][8]: lane (m = cout row, kq), element j = input channel
// (kq & 1) * 16 + (kq >> 1) * 8 + j of the chunk at tap (dy, dx) - k-groups 0 / 1 are the LOWER 8-channel halves of the chunk's
// two 16-channel records, 2 / 3 the upper halves (the order of the kernels' LDS lines); channels beyond the source's (the upper half of a last 16-channel chunk) and padded output
// channels are zero.  W: [cout][cin0 + cin1][3][3] (kd = 1).
void conv_zp_pack(const float *W, int cout_real, int cout_pad, int cin_real0, int cin_pad0, int cin_real1, int cin_pad1,
                  unsigned short *dst) {
    const int nblk = cout_pad / 16, n0 = (cin_pad0 + 31) / 32, nch = conv_zp_chunks(cin_pad0, cin_pad1);
    const int cin_tot = cin_real0 + cin_real1;
    for (int cb = 0; cb < nblk; ++cb)
        for (int ch = 0; ch < nch; ++ch)
            for (int ks = 0; ks < 9; ++ks)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int dx = ks / 3, dy = ks % 3, tap = dy * 3 + dx;
                        const int kq = lane >> 4;                           // k-group kq = channels (kq & 1) * 16 + (kq >> 1) * 8 .. + 7 (below: the LDS lines)
                        const int src = ch < n0 ? 0 : 1, cl = (src ? ch - n0 : ch) * 32 + (kq & 1) * 16 + (kq >> 1) * 8 + j;
                        const int creal = src ? cin_real1 : cin_real0;
                        const int co = conv3d_pack_cout(FNN_PACK_ZP, nblk, cb, lane & 15);
                        float v = 0.f;
                        if (co < cout_real && cl < creal) v = W[((size_t)co * cin_tot + (src ? cin_real0 : 0) + cl) * 9 + tap];
                        const f16 h = (f16)v;
                        unsigned short b;
                        memcpy(&b, &h, 2);
                        dst[((((size_t)cb * nch + ch) * 9 + ks) * 64 + lane) * 8 + j] = b;
                    }
}

// (waves along the columns, rows per wave) for a layer's plane, or false when the layer keeps the other kernels
static bool zp_pick(const ConvParams &p, int &wc, int &th) {
#ifdef FNN_NORM_FP32
    return false;                                             // (the A-B build with fp32 normalise-on-load keeps the linear-tap kernels)
#endif
    static const bool off = fnn_knob("FNN_NO_ZP") != nullptr;                       // A-B aid
    if (fnn_knob("FNN_CONV_V1") != nullptr) return false;                           // (read per call: the test of the generic kernel)
    if (off || p.kd != 1 || p.kh != 3 || p.kw != 3 || p.sd != 1 || p.fp8) return false;
    const bool strided = p.sh == 2 && p.sw == 2;
    if (!strided && !(p.sh == 1 && p.sw == 1)) return false;
    if (strided && fnn_knob("FNN_NO_ZPS") != nullptr) return false;                 // A-B aid
    // cout blocks in pairs; an odd number of blocks (16 output channels: the full-resolution level of an r = 2 student) one at a
    // time, only where no other kernel fits - depth < 4, i.e. `2d` configurations (at depth >= 4 the persistent 4 x 8 x 8 kernels
    // reach 4 TB/s on those layers) - and not strided (a down-sampling conv doubles its channels)
    if (p.Cout % 32 != 0 && (strided || p.Do >= 4 || p.Cout % 16 != 0)) return false;
    const long long vox = (long long)p.Do * p.Ho * p.Wo;
    const int cmax = p.src[0].C > p.src[1].C ? p.src[0].C : p.src[1].C;
    // 32-bit byte offsets inside a batch item, and the 0x80000000 "not fetched" offset must lie beyond every tensor
    if (vox * 2 * (cmax > 0 ? cmax : 16) >= (1ll << 31) || vox * 2 * p.Cout >= (1ll << 31) || vox >= (1 << 24)) return false;
    if (strided) {                                                                  // conv2d_zps_kernel: 8 x 32 or 16 x 16 output tiles
        const long long ivox = (long long)p.Di * p.Hi * p.Wi;
        if (ivox * 2 * (cmax > 0 ? cmax : 16) >= (1ll << 31) || ivox >= (1 << 24)) return false;
        wc = p.Wo > 16 ? 2 : 1; th = 4;
        return true;
    }
    if (p.Wo > 32) { wc = 4; th = 8; }
    else if (p.Wo > 16) { wc = 2; th = 8; }
    else { wc = 1; th = 4; }
    return true;
}

bool conv2d_zp_ok(const ConvParams &p) { int wc, th; return zp_pick(p, wc, th); }

int conv2d_zp_stats_slots(const ConvParams &p) {
    int wc, th;
    if (!zp_pick(p, wc, th)) return FNN_STAT_REPL;
    const int rows = (4 / wc) * th, cols = wc * 16;             // (the strided kernel: th = 4 -> 8 x 32 or 16 x 16)
    return p.Do * ((p.Ho + rows - 1) / rows) * ((p.Wo + cols - 1) / cols);
}

namespace {

typedef unsigned zp_u32x4 __attribute__((ext_vector_type(4)));
typedef int zp_i32x4 __attribute__((ext_vector_type(4)));

// HALF: every source has 16 channels (the full-resolution level of an r = 2 student, a stem on the conv kernels): only the two lines of a chunk's
// lower record are kept (conv2d_zps_kernel below: k-groups 1 and 3 read the lines of 0 and 2 - finite values times zero weights), the image is half as
// large and, with one cout block, three workgroups fit a CU.
template <int TH, int WC, int NB = 2, bool HALF = false>
__global__ __launch_bounds__(256, HALF && NB == 1 ? 3 : 2) void conv2d_zp_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int WR = 4 / WC;                                // waves along the rows
    constexpr int ROWS = WR * TH, COLS = WC * 16;             // output tile
    constexpr int IH = ROWS + 2, IWC = COLS + 2;              // halo image
    constexpr int QP = ((IWC * 16 + 64 + 255) / 256) * 256;   // bytes of one (row, k-group) line (+ the 64-byte shift of lines 2, 3)
    constexpr int ROWB = (HALF ? 2 : 4) * QP;
    constexpr int ABYTES = IH * ROWB;
    constexpr int RPP = 4 / WC;                               // halo rows staged per pass (one per group of 64 WC threads)
    constexpr int NP = (IH + RPP - 1) / RPP;                  // passes
    constexpr int KS = 9, WB = KS * 64;                       // 16-byte weight elements per cout block and chunk
    constexpr int WPB = (WB + 255) / 256;                     // 3: the last one by wave 0 only (576 = 2 x 256 + 64)
    static_assert(IH * 2 <= 64, "the two extra halo columns of every row fit the 64 threads of a channel group");

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware, bijective remap (blocks b and b + 8 share an XCD): neighbouring tiles share halo rows in one L2
    int t;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int qd = nwg >> 3, rm = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        t = __builtin_amdgcn_readfirstlane((xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx);
    }
    const int tile_in_item = __builtin_amdgcn_readfirstlane(t % (p.Do * p.tiles_h * p.tiles_w));
    const int tw = __builtin_amdgcn_readfirstlane(t % p.tiles_w); t = __builtin_amdgcn_readfirstlane(t / p.tiles_w);
    const int th = __builtin_amdgcn_readfirstlane(t % p.tiles_h); t = __builtin_amdgcn_readfirstlane(t / p.tiles_h);
    const int d = __builtin_amdgcn_readfirstlane(t % p.Do);
    const int n = __builtin_amdgcn_readfirstlane(t / p.Do);
    const int cb0 = blockIdx.y * NB;
    const int oh0 = th * ROWS, ow0 = tw * COLS;
    const int wr = wave / WC, wcol = wave % WC;               // scalar: the wave's strip of rows and its column block

    char *sA = smem;                                          // halo image
    char *sW = smem + ABYTES;                                 // [NB][9][64 lanes][16 B]

    // ---- staging: this thread's column and 8-channel group; the halo rows come by passes (row = pass * RPP + rp, scalar)
    const int rp = __builtin_amdgcn_readfirstlane(tid / (64 * WC));
    const int tt = tid % (64 * WC);
    const int cg4 = tt / (32 * WC), col = (tt % (32 * WC)) >> 1, half = tt & 1;
    const int q_st = cg4 * 2 + half;                          // channels 8 q_st .. + 7 of the 32-channel chunk
    const int gw = ow0 - 1 + col;
    const bool ok_w = (unsigned)gw < (unsigned)p.Wi;
    const int line_st = HALF ? half * QP + half * 64 : (half * 2 + cg4) * QP + half * 64;  // this thread's k-group line (file header)
    const bool stager = !(HALF && cg4);                      // HALF: the threads of the upper record stage nothing
    const int lds_main = line_st + col * 16;
    // the two extra columns (COLS, COLS + 1) of every halo row: 2 IH items per channel group, spread over its 64 threads
    const int i64 = rp * (16 * WC) + col;
    const bool has_x = i64 < IH * 2 && stager;
    const int xu = i64 >> 1, xcol = COLS + (i64 & 1);
    const int gh_x = oh0 - 1 + xu, gw_x = ow0 - 1 + xcol;
    const bool ok_x = has_x & ((unsigned)gh_x < (unsigned)p.Hi) & ((unsigned)gw_x < (unsigned)p.Wi);
    const int lds_x = xu * ROWB + line_st + xcol * 16;

    f32x4 acc[TH][NB];
    zp_u32x4 xr[NP], xx, wrg[NB][WPB], ssv[2];
    float slope_next = 1.f;
    __amdgpu_buffer_rsrc_t rx, rw[NB];
    unsigned voff = 0x80000000u, voff_x = 0x80000000u, row_bytes = 0, plane_off = 0;
    bool ch_ok = true;

    const int n0 = (p.src[0].C + 31) >> 5;                    // chunks of the first source
    auto prep = [&](int ch) {
        const int s = ch < n0 ? 0 : 1;
        const int c_uni = (s ? ch - n0 : ch) * 32;
        const int sC = p.src[s].C;
        const int vs = FNN_VS(p.src[s]);
        const long long cs = FNN_CS(p.src[s]);
        const unsigned item_bytes = (unsigned)p.Di * p.Hi * p.Wi * sC * 2;
        const f16 *sp = p.src[s].ptr + (size_t)n * (item_bytes >> 1) + (c_uni >> 4) * cs;
        rx = __builtin_amdgcn_make_buffer_rsrc((void *)sp, 0, item_bytes - (unsigned)((c_uni >> 4) * cs * 2), 0x00020000);
        slope_next = p.src[s].slope;
        ch_ok = c_uni + 8 * q_st < sC && stager;                        // (the upper half of a source's last 16-channel chunk does not exist)
        {
            const int cq = ch_ok ? c_uni + 8 * q_st : 0;
            const unsigned short *q = p.src[s].ssh ? p.src[s].ssh + ((size_t)n * sC + cq) * 2 : p.ident_ssh + cq * 2;
            const zp_u32x4 *qv = (const zp_u32x4 *)q;
            ssv[0] = qv[0]; ssv[1] = qv[1];
        }
        const unsigned piece = (unsigned)cg4 * (unsigned)(cs * 2) + (unsigned)half * 16u;
        voff = (ok_w & ch_ok) ? (unsigned)gw * (unsigned)(vs * 2) + piece : 0x80000000u;
        voff_x = (ok_x & ch_ok) ? (unsigned)(gh_x * p.Wi + gw_x) * (unsigned)(vs * 2) + piece : 0x80000000u;
        row_bytes = (unsigned)p.Wi * vs * 2;
        plane_off = (unsigned)d * (unsigned)p.Hi * row_bytes;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const f16 *wp = p.wpk + ((size_t)((cb0 + nb) * p.chunks + ch) * WB) * 8;
            rw[nb] = __builtin_amdgcn_make_buffer_rsrc((void *)wp, 0, WB * 16, 0x00020000);
        }
    };
    // the chunk's loads in three slices, one per dx of the k-loop (issued in one block the wave-wide loads of every wave
    // of the CU queue up in the texture-address path and the MFMAs behind them cannot issue)
    auto load_part = [&](int part) {
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            if (u * 3 / NP != part) continue;
            int gh = oh0 - 1 + u * RPP + rp;
            gh = gh < 0 ? 0 : (gh >= p.Hi ? p.Hi - 1 : gh);   // scalar; a clamped row's image is zeroed in commit()
            xr[u] = __builtin_bit_cast(zp_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff, plane_off + (unsigned)gh * row_bytes, 0));
        }
        if (part == 2) xx = __builtin_bit_cast(zp_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff_x, plane_off, 0));
#pragma unroll
        for (int e = 0; e < NB * WPB; ++e) {
            if (e * 3 / (NB * WPB) != part) continue;
            const int nb = e / WPB, u = e % WPB;
            wrg[nb][u] = __builtin_bit_cast(zp_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw[nb], tid * 16, u * (256 * 16), 0));
        }
    };
    auto commit = [&]() {
        const f16 slope_h = (f16)slope_next;
        const zp_u32x4 zero4 = {0u, 0u, 0u, 0u};
        // a column outside the tensor / a channel beyond the source: 0 * 0 + 0 = the conv's zero padding
        const bool okm = ok_w & ch_ok, okx = ok_x & ch_ok;
        {
            const f16x8 sc_h = __builtin_bit_cast(f16x8, okm ? ssv[0] : zero4), sh_h = __builtin_bit_cast(f16x8, okm ? ssv[1] : zero4);
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                const int row = u * RPP + rp;                 // scalar
                if (NP * RPP > IH && row >= IH) continue;     // (WC = 1: the last pass is half empty)
                const int gh = oh0 - 1 + row;
                f16x8 o = __builtin_bit_cast(f16x8, xr[u]) * sc_h + sh_h;
                o = __builtin_elementwise_max(o, o * slope_h);
                if ((unsigned)gh >= (unsigned)p.Hi) o = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};      // scalar condition: a row above / below the tensor
                if (stager) *(f16x8 *)(sA + row * ROWB + lds_main) = o;
            }
        }
        if (has_x) {
            const f16x8 sc_h = __builtin_bit_cast(f16x8, okx ? ssv[0] : zero4), sh_h = __builtin_bit_cast(f16x8, okx ? ssv[1] : zero4);
            f16x8 o = __builtin_bit_cast(f16x8, xx) * sc_h + sh_h;
            o = __builtin_elementwise_max(o, o * slope_h);
            *(f16x8 *)(sA + lds_x) = o;
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int u = 0; u < WPB; ++u)
                if (u + 1 < WPB || wave == 0) *(zp_u32x4 *)(sW + ((nb * WB + u * 256) + tid) * 16) = wrg[nb][u];
    };
    // MFMA "B" operand: lane (r = column of the wave's block, q = 8-channel group)
    const int boff = (wr * TH) * ROWB + (HALF ? (lane >> 5) : (lane >> 4)) * QP + (lane >> 5) * 64 + (wcol * 16 + (lane & 15)) * 16;
    auto kloop = [&](bool prefetch) {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            if (prefetch) load_part(dx);
            const char *bp = sA + boff + dx * 16;
            f16x8 xf[TH + 2];
#pragma unroll
            for (int pl = 0; pl < TH + 2; ++pl) xf[pl] = *(const f16x8 *)(bp + pl * ROWB);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                f16x8 wf[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) wf[nb] = *(const f16x8 *)(sW + ((nb * KS + dx * 3 + dy) * 64 + lane) * 16);
#pragma unroll
                for (int j = 0; j < TH; ++j)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[j][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nb], xf[j + dy], acc[j][nb], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);                // keep the next dx's reads from being hoisted: registers
        }
    };

    prep(0);
#pragma unroll
    for (int part = 0; part < 3; ++part) load_part(part);
    __builtin_amdgcn_sched_barrier(0);                        // the loads leave first; the rest of the set-up runs under them
    {
        // the bias is where the accumulators start: lane quarter q holds channels q * 8 + nb * 4 .. + 3 of the block pair
        f32x4 b0[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
            b0[nb] = *(const f32x4 *)(p.bias + cb0 * 16 + (NB == 2 ? (lane >> 4) * 8 + nb * 4 : (lane >> 4) * 4));   // (NB = 1: the plain channel order)
#pragma unroll
        for (int j = 0; j < TH; ++j)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[j][nb] = b0[nb];
    }
    commit();
    __syncthreads();
    for (int ch = 0; ch + 1 < p.chunks; ++ch) {               // (the last chunk is peeled off: its wait for a prefetch would be conditional)
        prep(ch + 1);
        kloop(true);
        __syncthreads();                                      // every wave is done reading this chunk
        commit();
        __syncthreads();
    }
    kloop(false);
    __syncthreads();

    // ---- epilogue: round to fp16, one 16-byte store per (voxel, lane), statistics of the rounded values
    {
        const int q = lane >> 4, r = lane & 15;
        const unsigned item_bytes = (unsigned)p.Do * p.Ho * p.Wo * p.Cout * 2;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)n * (item_bytes >> 1), 0, item_bytes, 0x00020000);
        const unsigned ovs2 = (unsigned)FNN_OVS(p) * 2;
        const unsigned coff = (unsigned)(cb0 + (q >> 1)) * (unsigned)(FNN_OCS(p) * 2) + (unsigned)(q & 1) * 16;
        const int ow = ow0 + wcol * 16 + r;
        const bool ok_c = ow < p.Wo;
        const f16x2 ones = {(f16)1.f, (f16)1.f};
        float t1[NB][4], t2[NB][4];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int j = 0; j < 4; ++j) { t1[nb][j] = 0.f; t2[nb][j] = 0.f; }
        if constexpr (NB == 1) {
            // one cout block: lane (r, q) holds channels 4 q .. + 3 of its voxel in rows jr and jr + 1; v_permlane16_swap (pair_to_b128) turns the
            // two rows' 8-byte pieces into ONE 16-byte store per lane - channels 8 (q >> 1) .. + 7 of voxel r of row jr + (q & 1)
            const unsigned coff1 = (unsigned)cb0 * (unsigned)(FNN_OCS(p) * 2) + (unsigned)(q >> 1) * 16;
#pragma unroll
            for (int jr = 0; jr < TH; jr += 2) {
                f16x4 o[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int oh = oh0 + wr * TH + jr + h;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[h][j] = (f16)acc[jr + h][0][j];
                    if (!(ok_c && oh < p.Ho)) o[h] = (f16x4){0, 0, 0, 0};
                }
                const int ohs = oh0 + wr * TH + jr + (q & 1);
                const unsigned vo = (ok_c && ohs < p.Ho) ? (unsigned)((d * p.Ho + ohs) * p.Wo + ow) * ovs2 + coff1 : 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(zp_i32x4, pair_to_b128(o[0], o[1])), rsrc, vo, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f16x2 pr = {o[0][j], o[1][j]};
                    t1[0][j] = __builtin_amdgcn_fdot2(pr, ones, t1[0][j], false);
                    t2[0][j] = __builtin_amdgcn_fdot2(pr, pr, t2[0][j], false);
                }
            }
            if (p.stats_out) stats_to_global<1, true, false, 4>(p, t1, t2, (float *)smem, n, cb0, wave, lane, tid, tile_in_item);
            return;
        }
#pragma unroll
        for (int jr = 0; jr < TH; jr += 2) {
            f16x8 o[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int oh = oh0 + wr * TH + jr + h;
                const bool ok = ok_c && oh < p.Ho;
                const unsigned vo = ok ? (unsigned)((d * p.Ho + oh) * p.Wo + ow) * ovs2 + coff : 0x80000000u;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[h][nb * 4 + j] = (f16)acc[jr + h][nb][j];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(zp_i32x4, o[h]), rsrc, vo, 0, 0);
                if (!ok) o[h] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f16x2 pr = {o[0][nb * 4 + j], o[1][nb * 4 + j]};
                    t1[nb][j] = __builtin_amdgcn_fdot2(pr, ones, t1[nb][j], false);
                    t2[nb][j] = __builtin_amdgcn_fdot2(pr, pr, t2[nb][j], false);
                }
        }
        if (p.stats_out) stats_to_global<NB, true, true, 4>(p, t1, t2, (float *)smem, n, cb0, wave, lane, tid, tile_in_item);
    }
}

template <int TH, int WC, int NB = 2, bool HALF = false>
int launch_zp(ConvParams p, hipStream_t st) {
    constexpr int WR = 4 / WC, ROWS = WR * TH, COLS = WC * 16, IH = ROWS + 2, IWC = COLS + 2;
    constexpr int QP = ((IWC * 16 + 64 + 255) / 256) * 256;
    const size_t lds = (size_t)IH * (HALF ? 2 : 4) * QP + NB * 9 * 1024;
    p.tiles_d = p.Do;
    p.tiles_h = (p.Ho + ROWS - 1) / ROWS;
    p.tiles_w = (p.Wo + COLS - 1) / COLS;
    if (p.stats_out && p.stats_slots < p.Do * p.tiles_h * p.tiles_w) return -1;   // (a plan sized for another tiling)
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)conv2d_zp_kernel<TH, WC, NB, HALF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    p.ident_ss = conv3d_identity_ss();
    p.ident_ssh = conv3d_identity_ssh();
    if (!p.ident_ss || !p.ident_ssh) return -2;
    const long long tiles = (long long)p.N * p.Do * p.tiles_h * p.tiles_w;
    if (tiles >= (1ll << 31)) return -1;
    dim3 grid((unsigned)tiles, (p.Cout / 16) / NB);
    if (NB == 2 && !HALF) fnn_note_kernel("conv2d_zp_kernel<%d,%d>", TH, WC);
    else fnn_note_kernel(HALF ? "conv2d_zp_kernel<%d,%d,%d,half>" : "conv2d_zp_kernel<%d,%d,%d>", TH, WC, NB);
    hipLaunchKernelGGL((conv2d_zp_kernel<TH, WC, NB, HALF>), grid, dim3(256), lds, st, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ----------------------------------------------------------------------------
// stride (1, 2, 2): the down-sampling conv of a (1, 3, 3) stage
// ----------------------------------------------------------------------------
// out[j][x] = sum w[dy][dx] in[2 j + dy - 1][2 x + dx - 1].  The halo image keeps the EVEN and the ODD input columns of a
// row in lines of their own: tap dx = 0 reads odd column x - 1, dx = 1 even column x, dx = 2 odd column x - a wave's
// fragment is again 16 consecutive 16-byte slots of four congruent lines (conflict free), not a stride-2 walk.  A workgroup
// stages (2 ROWS + 1) x (2 COLS + 1) input voxels for ROWS x COLS outputs - four times the bytes per MFMA of the stride-1
// kernel - so it keeps NB = 4 cout blocks (64 output channels) per staged image where the layer has them (a down-sampling
// conv has twice its input's channels: every cout group re-reads the whole input), and one workgroup per CU (124 KB).
// Same packing (FNN_PACK_ZP), staging arithmetic, epilogue and statistics rows as conv2d_zp_kernel.
// Measured and dropped (round 6): 4 x 32 tiles with two cout blocks for layers of one or two chunks - 64 KB, two workgroups per CU, so that a
// single-chunk tile has a neighbour to hide its load latency behind: SLOWER, 32 -> 64 at 256^2 575 -> 736 us, 64 -> 128 527 -> 600 us, the thick-slice
// plan's 32 -> 64 1677 -> 2174 us (profiles/r06_zps_th2_ab.txt): half the MFMAs per staged byte and a 9 / 4 instead of 17 / 8 halo cost more than the overlap buys.
// HALF: one source of 16 channels (the first down-sampling conv of an r = 2 student) - the chunk's upper 16 channels do not exist (their
// weights are zero): only the two lines of the lower record are kept (k-groups 1 and 3 read the lines of 0 and 2: finite values times zero), the
// image is half as large - 62 KB with two cout blocks - and TWO workgroups fit a CU: a single-chunk tile has a neighbour to hide its loads behind.
template <int NB, int WC, bool HALF = false>
__global__ __launch_bounds__(256, HALF ? 2 : 1) void conv2d_zps_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TH = 4;
    constexpr int WR = 4 / WC;
    constexpr int ROWS = WR * TH, COLS = WC * 16;             // output tile: 8 x 32 or 16 x 16
    constexpr int IH = 2 * ROWS + 1, ICOLS = 2 * COLS + 1;    // input halo
    constexpr int EOFF = 0, OOFF = COLS * 16 + 32;            // even-column slots, then odd-column slots 32 B off the stores' 128-byte bank period:
                                                              // a store group's four columns (odd, even, odd, even) x two halves (64 B apart) = eight slots
    constexpr int QP = ((OOFF + (COLS + 1) * 16 + 64 + 255) / 256) * 256;
    constexpr int ROWB = (HALF ? 2 : 4) * QP;
    constexpr int ABYTES = IH * ROWB;
    constexpr int RPP = 2 / WC;                               // input rows staged per pass: 128 WC threads per row
    constexpr int NP = (IH + RPP - 1) / RPP;
    constexpr int KS = 9, WB = KS * 64, WPB = (WB + 255) / 256;
    static_assert(IH <= 64, "the extra input column of every row fits the 64 threads of a channel group");

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int t;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int qd = nwg >> 3, rm = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        t = __builtin_amdgcn_readfirstlane((xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx);
    }
    const int tile_in_item = __builtin_amdgcn_readfirstlane(t % (p.Do * p.tiles_h * p.tiles_w));
    const int tw = __builtin_amdgcn_readfirstlane(t % p.tiles_w); t = __builtin_amdgcn_readfirstlane(t / p.tiles_w);
    const int th = __builtin_amdgcn_readfirstlane(t % p.tiles_h); t = __builtin_amdgcn_readfirstlane(t / p.tiles_h);
    const int d = __builtin_amdgcn_readfirstlane(t % p.Do);
    const int n = __builtin_amdgcn_readfirstlane(t / p.Do);
    const int cb0 = blockIdx.y * NB;
    const int oh0 = th * ROWS, ow0 = tw * COLS;
    const int wr = wave / WC, wcol = wave % WC;

    char *sA = smem;
    char *sW = smem + ABYTES;                                 // [NB][9][64 lanes][16 B]

    // ---- staging: thread = (input column ci of the halo row, 8-channel group); input column ci is image column 2 ow0 - 1 + ci:
    // ci even -> an ODD column, slot ci / 2 of the odd line; ci odd -> an even column, slot (ci - 1) / 2 of the even line
    const int rp = __builtin_amdgcn_readfirstlane(tid / (128 * WC));
    const int tt = tid % (128 * WC);
    const int cg4 = tt / (64 * WC), ci = (tt % (64 * WC)) >> 1, half = tt & 1;
    const int q_st = cg4 * 2 + half;
    const int gw = 2 * ow0 - 1 + ci;
    const bool ok_w = (unsigned)gw < (unsigned)p.Wi;
    const int line_st = HALF ? half * QP + half * 64 : (half * 2 + cg4) * QP + half * 64;
    const bool stager = !(HALF && cg4);                      // HALF: the threads of the upper record stage nothing
    const int lds_main = line_st + ((ci & 1) ? EOFF + (ci >> 1) * 16 : OOFF + (ci >> 1) * 16);
    // the last input column (ci = 2 COLS, odd line slot COLS) of every halo row: IH items per channel group
    const int i64 = rp * (32 * WC) + ci;                      // 0 .. 63 within the channel group
    const bool has_x = i64 < IH && stager;
    const int gh_x = 2 * oh0 - 1 + i64, gw_x = 2 * ow0 - 1 + 2 * COLS;
    const bool ok_x = has_x & ((unsigned)gh_x < (unsigned)p.Hi) & ((unsigned)gw_x < (unsigned)p.Wi);
    const int lds_x = i64 * ROWB + line_st + OOFF + COLS * 16;

    f32x4 acc[TH][NB];
    zp_u32x4 xr[NP], xx, wrg[NB][WPB], ssv[2];
    float slope_next = 1.f;
    __amdgpu_buffer_rsrc_t rx, rw[NB];
    unsigned voff = 0x80000000u, voff_x = 0x80000000u, row_bytes = 0, plane_off = 0;
    bool ch_ok = true;

    const int n0 = (p.src[0].C + 31) >> 5;
    auto prep = [&](int ch) {
        const int s = ch < n0 ? 0 : 1;
        const int c_uni = (s ? ch - n0 : ch) * 32;
        const int sC = p.src[s].C;
        const int vs = FNN_VS(p.src[s]);
        const long long cs = FNN_CS(p.src[s]);
        const unsigned item_bytes = (unsigned)p.Di * p.Hi * p.Wi * sC * 2;
        const f16 *sp = p.src[s].ptr + (size_t)n * (item_bytes >> 1) + (c_uni >> 4) * cs;
        rx = __builtin_amdgcn_make_buffer_rsrc((void *)sp, 0, item_bytes - (unsigned)((c_uni >> 4) * cs * 2), 0x00020000);
        slope_next = p.src[s].slope;
        ch_ok = c_uni + 8 * q_st < sC && stager;
        {
            const int cq = ch_ok ? c_uni + 8 * q_st : 0;
            const unsigned short *q = p.src[s].ssh ? p.src[s].ssh + ((size_t)n * sC + cq) * 2 : p.ident_ssh + cq * 2;
            const zp_u32x4 *qv = (const zp_u32x4 *)q;
            ssv[0] = qv[0]; ssv[1] = qv[1];
        }
        const unsigned piece = (unsigned)cg4 * (unsigned)(cs * 2) + (unsigned)half * 16u;
        voff = (ok_w & ch_ok) ? (unsigned)gw * (unsigned)(vs * 2) + piece : 0x80000000u;
        voff_x = (ok_x & ch_ok) ? (unsigned)(gh_x * p.Wi + gw_x) * (unsigned)(vs * 2) + piece : 0x80000000u;
        row_bytes = (unsigned)p.Wi * vs * 2;
        plane_off = (unsigned)d * (unsigned)p.Hi * row_bytes;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const f16 *wp = p.wpk + ((size_t)((cb0 + nb) * p.chunks + ch) * WB) * 8;
            rw[nb] = __builtin_amdgcn_make_buffer_rsrc((void *)wp, 0, WB * 16, 0x00020000);
        }
    };
    auto load_part = [&](int part) {
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            if (u * 3 / NP != part) continue;
            int gh = 2 * oh0 - 1 + u * RPP + rp;
            gh = gh < 0 ? 0 : (gh >= p.Hi ? p.Hi - 1 : gh);
            xr[u] = __builtin_bit_cast(zp_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff, plane_off + (unsigned)gh * row_bytes, 0));
        }
        if (part == 2) xx = __builtin_bit_cast(zp_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff_x, plane_off, 0));
#pragma unroll
        for (int e = 0; e < NB * WPB; ++e) {
            if (e * 3 / (NB * WPB) != part) continue;
            const int nb = e / WPB, u = e % WPB;
            wrg[nb][u] = __builtin_bit_cast(zp_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw[nb], tid * 16, u * (256 * 16), 0));
        }
    };
    auto commit = [&]() {
        const f16 slope_h = (f16)slope_next;
        const zp_u32x4 zero4 = {0u, 0u, 0u, 0u};
        const bool okm = ok_w & ch_ok, okx = ok_x & ch_ok;
        {
            const f16x8 sc_h = __builtin_bit_cast(f16x8, okm ? ssv[0] : zero4), sh_h = __builtin_bit_cast(f16x8, okm ? ssv[1] : zero4);
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                const int row = u * RPP + rp;
                if (NP * RPP > IH && row >= IH) continue;
                const int gh = 2 * oh0 - 1 + row;
                f16x8 o = __builtin_bit_cast(f16x8, xr[u]) * sc_h + sh_h;
                o = __builtin_elementwise_max(o, o * slope_h);
                if ((unsigned)gh >= (unsigned)p.Hi) o = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
                if (stager) *(f16x8 *)(sA + row * ROWB + lds_main) = o;
            }
        }
        if (has_x) {
            const f16x8 sc_h = __builtin_bit_cast(f16x8, okx ? ssv[0] : zero4), sh_h = __builtin_bit_cast(f16x8, okx ? ssv[1] : zero4);
            f16x8 o = __builtin_bit_cast(f16x8, xx) * sc_h + sh_h;
            o = __builtin_elementwise_max(o, o * slope_h);
            *(f16x8 *)(sA + lds_x) = o;
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int u = 0; u < WPB; ++u)
                if (u + 1 < WPB || wave == 0) *(zp_u32x4 *)(sW + ((nb * WB + u * 256) + tid) * 16) = wrg[nb][u];
    };
    // MFMA "B" operand of output column x = wcol 16 + r, tap dx: dx = 0 -> odd slot x, dx = 1 -> even slot x, dx = 2 -> odd slot x + 1
    const int bbase = (2 * wr * TH) * ROWB + (HALF ? (lane >> 5) : (lane >> 4)) * QP + (lane >> 5) * 64 + (wcol * 16 + (lane & 15)) * 16;
    auto kloop = [&](bool prefetch) {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            if (prefetch) load_part(dx);
            const char *bp = sA + bbase + (dx == 1 ? EOFF : OOFF + (dx == 2 ? 16 : 0));
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                f16x8 wf[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) wf[nb] = *(const f16x8 *)(sW + ((nb * KS + dx * 3 + dy) * 64 + lane) * 16);
#pragma unroll
                for (int j = 0; j < TH; ++j) {
                    const f16x8 xf = *(const f16x8 *)(bp + (2 * j + dy) * ROWB);
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[j][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nb], xf, acc[j][nb], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    prep(0);
#pragma unroll
    for (int part = 0; part < 3; ++part) load_part(part);
    __builtin_amdgcn_sched_barrier(0);
    {
        f32x4 b0[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) b0[nb] = *(const f32x4 *)(p.bias + (cb0 + (nb & ~1)) * 16 + (lane >> 4) * 8 + (nb & 1) * 4);
#pragma unroll
        for (int j = 0; j < TH; ++j)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[j][nb] = b0[nb];
    }
    commit();
    __syncthreads();
    for (int ch = 0; ch + 1 < p.chunks; ++ch) {
        prep(ch + 1);
        kloop(true);
        __syncthreads();
        commit();
        __syncthreads();
    }
    kloop(false);
    __syncthreads();

    // ---- epilogue: per pair of cout blocks one 16-byte store per (voxel, lane); statistics per pair
    {
        const int q = lane >> 4, r = lane & 15;
        const unsigned item_bytes = (unsigned)p.Do * p.Ho * p.Wo * p.Cout * 2;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)n * (item_bytes >> 1), 0, item_bytes, 0x00020000);
        const unsigned ovs2 = (unsigned)FNN_OVS(p) * 2;
        const int ow = ow0 + wcol * 16 + r;
        const bool ok_c = ow < p.Wo;
        const f16x2 ones = {(f16)1.f, (f16)1.f};
#pragma unroll
        for (int pr2 = 0; pr2 < NB / 2; ++pr2) {
            const unsigned coff = (unsigned)(cb0 + 2 * pr2 + (q >> 1)) * (unsigned)(FNN_OCS(p) * 2) + (unsigned)(q & 1) * 16;
            float t1[2][4], t2[2][4];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int j = 0; j < 4; ++j) { t1[nb][j] = 0.f; t2[nb][j] = 0.f; }
#pragma unroll
            for (int jr = 0; jr < TH; jr += 2) {
                f16x8 o[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int oh = oh0 + wr * TH + jr + h;
                    const bool ok = ok_c && oh < p.Ho;
                    const unsigned vo = ok ? (unsigned)((d * p.Ho + oh) * p.Wo + ow) * ovs2 + coff : 0x80000000u;
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[h][nb * 4 + j] = (f16)acc[jr + h][2 * pr2 + nb][j];
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(zp_i32x4, o[h]), rsrc, vo, 0, 0);
                    if (!ok) o[h] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
                }
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f16x2 pr = {o[0][nb * 4 + j], o[1][nb * 4 + j]};
                        t1[nb][j] = __builtin_amdgcn_fdot2(pr, ones, t1[nb][j], false);
                        t2[nb][j] = __builtin_amdgcn_fdot2(pr, pr, t2[nb][j], false);
                    }
            }
            if (p.stats_out) {
                if (pr2 > 0) __syncthreads();                 // the previous pair's sums have been read out of the scratch
                stats_to_global<2, true, true, 4>(p, t1, t2, (float *)smem, n, cb0 + 2 * pr2, wave, lane, tid, tile_in_item);
            }
        }
    }
}

template <int NB, int WC, bool HALF = false>
int launch_zps(ConvParams p, hipStream_t st) {
    constexpr int WR = 4 / WC, ROWS = WR * 4, COLS = WC * 16, IH = 2 * ROWS + 1;
    constexpr int OOFF = COLS * 16 + 32, QP = ((OOFF + (COLS + 1) * 16 + 64 + 255) / 256) * 256;
    const size_t lds = (size_t)IH * (HALF ? 2 : 4) * QP + (size_t)NB * 9 * 1024;
    p.tiles_d = p.Do;
    p.tiles_h = (p.Ho + ROWS - 1) / ROWS;
    p.tiles_w = (p.Wo + COLS - 1) / COLS;
    if (p.stats_out && p.stats_slots < p.Do * p.tiles_h * p.tiles_w) return -1;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)conv2d_zps_kernel<NB, WC, HALF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    p.ident_ss = conv3d_identity_ss();
    p.ident_ssh = conv3d_identity_ssh();
    if (!p.ident_ss || !p.ident_ssh) return -2;
    const long long tiles = (long long)p.N * p.Do * p.tiles_h * p.tiles_w;
    if (tiles >= (1ll << 31)) return -1;
    dim3 grid((unsigned)tiles, (p.Cout / 16) / NB);
    fnn_note_kernel(HALF ? "conv2d_zps_kernel<%d,%d,half>" : "conv2d_zps_kernel<%d,%d>", NB, WC);
    hipLaunchKernelGGL((conv2d_zps_kernel<NB, WC, HALF>), grid, dim3(256), lds, st, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

// Runs the layer; the weights must have been packed as FNN_PACK_ZP and p.chunks = conv_zp_chunks(...).
int launch_conv2d_zp(const ConvParams &p, hipStream_t st) {
    int wc, th;
    if (p.packing != FNN_PACK_ZP || p.ksteps != 9 || !zp_pick(p, wc, th)) return -1;
    if (p.chunks != conv_zp_chunks(p.src[0].C, p.n_src > 1 ? p.src[1].C : 0)) return -1;
    if (p.sh == 2) {
        const bool four = (p.Cout / 16) % 4 == 0;
        const bool half16 = p.n_src == 1 && p.src[0].C == 16 && !four && fnn_knob("FNN_ZPS_NO_HALF") == nullptr;   // (knob: A-B aid)
        if (half16) return wc == 2 ? launch_zps<2, 2, true>(p, st) : launch_zps<2, 1, true>(p, st);
        if (wc == 2) return four ? launch_zps<4, 2>(p, st) : launch_zps<2, 2>(p, st);
        return four ? launch_zps<4, 1>(p, st) : launch_zps<2, 1>(p, st);
    }
    const bool half16 = p.src[0].C == 16 && (p.n_src < 2 || p.src[1].C == 16) && fnn_knob("FNN_ZP_NO_HALF") == nullptr;   // (knob: A-B aid)
    if ((p.Cout / 16) % 2 != 0) {                             // an odd number of cout blocks: one per workgroup
        if (half16) {
            if (wc == 4) return launch_zp<8, 4, 1, true>(p, st);
            if (wc == 2) return launch_zp<8, 2, 1, true>(p, st);
            return launch_zp<4, 1, 1, true>(p, st);
        }
        if (wc == 4) return launch_zp<8, 4, 1>(p, st);
        if (wc == 2) return launch_zp<8, 2, 1>(p, st);
        return launch_zp<4, 1, 1>(p, st);
    }
    if (half16) {
        if (wc == 4) return launch_zp<8, 4, 2, true>(p, st);
        if (wc == 2) return launch_zp<8, 2, 2, true>(p, st);
        return launch_zp<4, 1, 2, true>(p, st);
    }
    if (wc == 4) return launch_zp<8, 4>(p, st);
    if (wc == 2) return launch_zp<8, 2>(p, st);
    return launch_zp<4, 1>(p, st);
}
