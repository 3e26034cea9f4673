// Debug harness (GPU box): runs launch_stem_mfma on a random one-channel volume and dumps output + statistics rows.
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 -I fast-nnunet_amd/csrc tools/stem_check.cpp -L fast-nnunet_amd/csrc -lfnn_hip -o /tmp/stem_check
//   /tmp/stem_check <Cout> <kd> <P> <out file> [N]        (FNN_KNOBS=1 FNN_NO_STEM1=1 for the generic kernel)
#include "fnn_device.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
int stem_mfma_ksteps(int C, int taps);
bool stem_mfma_kmap(int C, int taps, int ks, int k, int *c, int *tap);
int stem_mfma_stats_slots(int PD, int PH, int PW);
static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t b; __builtin_memcpy(&b, &h, 2); return b; }
int main(int argc, char **argv) {
    const int Cout = atoi(argv[1]), kd = atoi(argv[2]), P = atoi(argv[3]);
    const int N = argc > 5 ? atoi(argv[5]) : 2, X = P + 8, T = kd * 9;
    std::vector<float> vol((size_t)X * X * X), bias(Cout);
    srand(7);
    for (auto &v : vol) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
    std::vector<float> W((size_t)Cout * T);
    for (auto &v : W) v = (rand() / (float)RAND_MAX - 0.5f);
    for (auto &v : bias) v = (rand() / (float)RAND_MAX - 0.5f);
    const int KST = stem_mfma_ksteps(1, T);
    std::vector<uint16_t> wfrag((size_t)(Cout / 16) * KST * 64 * 8);
    for (int cb = 0; cb < Cout / 16; ++cb) for (int ks = 0; ks < KST; ++ks) for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j) {
        int c = 0, tap = 0;
        const int co = cb * 16 + (lane & 15);
        const bool live = stem_mfma_kmap(1, T, ks, 8 * (lane >> 4) + j, &c, &tap);
        wfrag[((size_t)(cb * KST + ks) * 64 + lane) * 8 + j] = f2h(live ? W[(size_t)co * T + tap] : 0.f);
    }
    std::vector<int> org(N * 3);
    for (int i = 0; i < N; ++i) { org[i * 3] = (i * 3) % 9; org[i * 3 + 1] = (i * 5) % 9; org[i * 3 + 2] = (i * 7) % 9; }
    float *dvol, *dbias; uint16_t *dw; int *dorg; f16 *dout; double *dstats;
    const int slots = stem_mfma_stats_slots(P, P, P);
    const size_t nout = (size_t)N * P * P * P * Cout, nst = (size_t)N * slots * Cout * 2;
    hipMalloc(&dvol, vol.size() * 4); hipMalloc(&dbias, Cout * 4); hipMalloc(&dw, wfrag.size() * 2); hipMalloc(&dorg, N * 12);
    hipMalloc(&dout, nout * 2); hipMalloc(&dstats, nst * 8);
    hipMemcpy(dvol, vol.data(), vol.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dbias, bias.data(), Cout * 4, hipMemcpyHostToDevice);
    hipMemcpy(dw, wfrag.data(), wfrag.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dorg, org.data(), N * 12, hipMemcpyHostToDevice);
    hipMemset(dout, 0xff, nout * 2); hipMemset(dstats, 0, nst * 8);
    StemParams p{};
    p.vol = dvol; p.vol_batch_stride = 0; p.C = 1; p.X = X; p.Y = X; p.Z = X; p.origins = dorg;
    p.PD = P; p.PH = P; p.PW = P; p.kd = kd; p.kh = 3; p.kw = 3; p.Cout = Cout; p.w = nullptr; p.bias = dbias; p.out = dout; p.stats_out = dstats;
    const int rc = launch_stem_mfma(p, (const f16 *)dw, N, 0);
    hipDeviceSynchronize();
    if (argc > 6) {                                                    // timing: "t" = as is, "s" = statistics only (no stores)
        StemParams pt = p;
        if (argv[6][0] == 's') pt.out = nullptr;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 3; ++i) launch_stem_mfma(pt, (const f16 *)dw, N, 0);
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i) launch_stem_mfma(pt, (const f16 *)dw, N, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.1f us per launch (%d items of %d^3, %d channels: %.2f GB out)\n", argv[6], ms * 50.f, N, P, Cout, nout * 2 / 1e9);
    }
    printf("rc %d err %s\n", rc, hipGetErrorString(hipGetLastError()));
    std::vector<uint16_t> out(nout); std::vector<double> st(nst);
    hipMemcpy(out.data(), dout, nout * 2, hipMemcpyDeviceToHost); hipMemcpy(st.data(), dstats, nst * 8, hipMemcpyDeviceToHost);
    FILE *f = fopen(argv[4], "wb");
    fwrite(out.data(), 2, nout, f); fwrite(st.data(), 8, nst, f); fclose(f);
    return 0;
}
