// Stand-alone check + timing of the weight-stationary linear2 kernel (k_lin2.hip.h) against the 256 x 256-tile kernel it replaces
// (k_gemm_glds<..., EpiPieces<EpiLinear2>>, k_gemm.hip.h): the two must leave the residual stream h BIT FOR BIT equal.
//   build: hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/lin2_harness.hip -o tools/_exp/lin2_harness
//   run:   tools/_exp/lin2_harness [tokens] [D] [K] [tokens_per_traj] [shared_gate 0/1] [iters] [rpx]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

#include "../lam_slide_amd/csrc/k_gemm.hip.h"
#ifdef LIN2_PROBED  // tools/build_harness.sh lin2: the product kernel + tools/experiments/lin2_probes.patch (timing arms, cycle stamps: results wrong when set)
#include "_exp/k_lin2_probed.hip.h"
#else
#include "../lam_slide_amd/csrc/k_lin2.hip.h"
#define LIN2_PROBE 0
#endif

#ifndef LIN2_HB2
#define LIN2_HB2 1
#endif

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

static unsigned rng_state = 12345u;
static unsigned rnd() {
    rng_state = rng_state * 1664525u + 1013904223u;
    return rng_state >> 8;
}
static float rndf() { return (float)(rnd() & 0xFFFF) / 32768.0f - 1.0f; }  // [-1, 1)
static u16 f2bf_host(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    u += 0x7FFF + ((u >> 16) & 1);
    return (u16)(u >> 16);
}

template <int K, int NCH_, int NS_>
void run_case(int N, int F, int tpt, int shared, int iters, int rpx_arg) {
    const int Npad = (N + 255) / 256 * 256, Fpad = (F + 255) / 256 * 256;
    const int ntraj = (N + tpt - 1) / tpt, MODW = 6 * F + 64, mod_stride = shared ? 0 : MODW;
    printf("case N=%d F=%d K=%d tokens/traj=%d shared_gate=%d  (chunks per block %d, ring slots %d, residual images %d)\n", N, F, K, tpt, shared, NCH_, NS_, LIN2_HB2 ? 2 : 1);
    std::vector<u16> hW((size_t)Fpad * K), hZ((size_t)Npad * K);
    std::vector<float> hb(Fpad), hg((size_t)ntraj * MODW), hh((size_t)N * F);
    for (auto &v : hW) v = f2bf_host(rndf() * 0.05f);
    for (auto &v : hZ) v = f2bf_host(rndf() * 1.5f);
    for (auto &v : hb) v = rndf() * 0.3f;
    for (auto &v : hg) v = rndf();
    for (auto &v : hh) v = rndf() * 2.0f;
    u16 *W, *Wp, *Z;
    float *b, *gt, *h0, *h1;
    CK(hipMalloc(&W, hW.size() * 2)); CK(hipMalloc(&Wp, hW.size() * 2)); CK(hipMalloc(&Z, hZ.size() * 2));
    CK(hipMalloc(&b, hb.size() * 4)); CK(hipMalloc(&gt, hg.size() * 4));
    CK(hipMalloc(&h0, hh.size() * 4)); CK(hipMalloc(&h1, hh.size() * 4));
    CK(hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(Z, hZ.data(), hZ.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(gt, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(h0, hh.data(), hh.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(h1, hh.data(), hh.size() * 4, hipMemcpyHostToDevice));
    auto magic_of = [](int dv) { return dv == 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)dv + 1); };
    const float *gate = gt + 2 * F;  // (shift, scale, gate of a sub-block: the gate is the third F-vector of the row)

    // ---- old kernel ----
    EpiLinear2 e2{b, gate, h0, F, mod_stride, tpt, 32, magic_of(tpt), nullptr, nullptr, nullptr};
    using Epi = EpiPieces<EpiLinear2>;
    GemmArgs ga{W, Z, F, N, K, 0, 0, 0};
    auto kold = k_gemm_glds<256, 256, 2, 4, 64, 2, true, Epi>;
    const size_t lds_old = GemmCfg<256, 256, 2, 4, 64, 2, true, Epi>::lds_bytes;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kold), hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    const int tiles = ((N + 255) / 256) * (Fpad / 256);
    const int grid_old = tiles < 256 ? tiles : 256;
    // ---- new kernel ----
    using C2 = Lin2Cfg<K, NCH_, NS_, LIN2_HB2 != 0>;
    hipLaunchKernelGGL(k_lin2_pack, dim3(256), dim3(256), 0, 0, Wp, W, F, K);
    CK(hipDeviceSynchronize());
    const int slices = F / 128;
    int rpx = rpx_arg > 0 ? rpx_arg : 32 / slices;
    const int NBLK = (N + 31) / 32;
    while (rpx > 1 && 8 * rpx > NBLK) --rpx;
    const int ranges = 8 * rpx;
    // trajectories one range can span
    const int max_blocks = (NBLK + ranges - 1) / ranges + 1;
    int gate_rows = shared ? 1 : (max_blocks * 32 + tpt - 1) / tpt + 1;
    if (gate_rows > C2::max_gate_rows) { printf("  gate table too large (%d rows > %d)\n", gate_rows, C2::max_gate_rows); return; }
    unsigned long long *dbg;
    const size_t dbg_bytes = (LIN2_PROBE & 1024) ? (size_t)N * F * 2 + 4096 : (size_t)256 * 8 * 16 * 8;  // (probe 1024: the bf16 rows a fused LayerNorm would write)
    CK(hipMalloc(&dbg, dbg_bytes));
    CK(hipMemset(dbg, 0, dbg_bytes));
#ifdef LIN2_PROBED
    Lin2Args la{Wp, Z, b, gate, h1, F, N, mod_stride, tpt, magic_of(tpt), slices, rpx, gate_rows, dbg};
#else
    Lin2Args la{Wp, Z, b, gate, h1, F, N, mod_stride, tpt, magic_of(tpt), slices, rpx, gate_rows};
#endif
    auto knew = k_linear2_ws<K, NCH_, NS_, LIN2_HB2 != 0>;
    const size_t lds_new = C2::lds_bytes(gate_rows);
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(knew), hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    const int gnew = 8 * slices * rpx;
    printf("  lds old %zu new %zu, grid old %d new %d (slices %d, ranges %d, gate rows %d)\n", lds_old, lds_new, grid_old, gnew, slices, ranges, gate_rows);

    hipLaunchKernelGGL(kold, dim3(grid_old), dim3(512), lds_old, 0, ga, Epi(e2));
    CK(hipDeviceSynchronize());
    printf("  old kernel ran\n"); fflush(stdout);
    hipLaunchKernelGGL(knew, dim3(gnew), dim3(512), lds_new, 0, la);
    CK(hipDeviceSynchronize());
    printf("  new kernel ran\n"); fflush(stdout);

    std::vector<float> r0(hh.size()), r1(hh.size());
    CK(hipMemcpy(r0.data(), h0, hh.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r1.data(), h1, hh.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0, first = 0, unchanged = 0;
    for (size_t i = 0; i < hh.size(); ++i) {
        if (memcmp(&r0[i], &r1[i], 4) != 0) { if (!bad) first = i; ++bad; }
        if (memcmp(&r0[i], &hh[i], 4) == 0) ++unchanged;
    }
    printf("  mismatches: %zu of %zu (first n=%zu f=%zu old %.6g new %.6g start %.6g); elements the old kernel left unchanged: %zu\n", bad, hh.size(),
           bad ? first / F : 0, bad ? first % F : 0, bad ? r0[first] : 0.f, bad ? r1[first] : 0.f, bad ? hh[first] : 0.f, unchanged);
    if (bad) {  // where: per feature block of 32 and per token-block position
        size_t by_fb[16] = {0};
        for (size_t i = 0; i < hh.size(); ++i)
            if (memcmp(&r0[i], &r1[i], 4) != 0) by_fb[((i % F) / 32) & 15]++;
        printf("  by feature block:");
        for (int k = 0; k < F / 32 && k < 16; ++k) printf(" %zu", by_fb[k]);
        printf("\n");
    }
    printf("  %s\n", bad == 0 ? "BITS EQUAL" : "DIFFERENT");

#if (LIN2_PROBE & 128)
    {
        std::vector<unsigned long long> hd(256 * 8 * 8);
        CK(hipMemcpy(hd.data(), dbg, hd.size() * 8, hipMemcpyDeviceToHost));
        for (int wg : {0, 1, 100, 255})
            for (int w : {0, 3, 4, 7}) {
                const unsigned long long *d = &hd[((size_t)wg * 8 + w) * 8];
                if (d[1]) printf("  wg %3d wave %d: kernel %.0f cycles in %.1f us -> %.0f MHz; chunk-steps %llu: counted wait %.0f  barrier %.0f  work %.0f cycles per chunk-step\n", wg, w,
                                 (double)d[0], d[1] / 100.0, (double)d[0] / (d[1] / 100.0), d[5], (double)d[2] / d[5], (double)d[3] / d[5], (double)d[4] / d[5]);
            }
    }
#endif
    // timing, interleaved rounds (the residual stream keeps accumulating: values grow, timing does not care)
    hipEvent_t ev0, ev1;
    CK(hipEventCreate(&ev0)); CK(hipEventCreate(&ev1));
    const double flop = 2.0 * N * (double)K * F;
    for (int round = 0; round < 3; ++round) {
        float ms_old = 0, ms_new = 0;
        CK(hipEventRecord(ev0, 0));
        for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(kold, dim3(grid_old), dim3(512), lds_old, 0, ga, Epi(e2));
        CK(hipEventRecord(ev1, 0)); CK(hipEventSynchronize(ev1)); CK(hipEventElapsedTime(&ms_old, ev0, ev1));
        CK(hipEventRecord(ev0, 0));
        for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(knew, dim3(gnew), dim3(512), lds_new, 0, la);
        CK(hipEventRecord(ev1, 0)); CK(hipEventSynchronize(ev1)); CK(hipEventElapsedTime(&ms_new, ev0, ev1));
        printf("  round %d: old %.4f ms/launch (%.0f TF/s)   new %.4f ms/launch (%.0f TF/s)\n", round, ms_old / iters,
               flop / (ms_old / iters * 1e-3) * 1e-12, ms_new / iters, flop / (ms_new / iters * 1e-3) * 1e-12);
    }
    hipFree(W); hipFree(Wp); hipFree(Z); hipFree(b); hipFree(gt); hipFree(h0); hipFree(h1); hipFree(dbg);
}

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 245760, F = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 1536;
    const int tpt = argc > 4 ? atoi(argv[4]) : 7680, shared = argc > 5 ? atoi(argv[5]) : 1, iters = argc > 6 ? atoi(argv[6]) : 20;
    const int rpx = argc > 7 ? atoi(argv[7]) : 0;
#ifndef LIN2_NCH
#define LIN2_NCH 0
#define LIN2_NS 0
#endif
#ifndef LIN2_FOR_K
#define LIN2_FOR_K 1536  // the width the LIN2_NCH / LIN2_NS override applies to; the others run their product instances
#endif
    if (K == 1536) run_case<1536, (LIN2_NCH && LIN2_FOR_K == 1536) ? LIN2_NCH : 3, (LIN2_NCH && LIN2_FOR_K == 1536) ? LIN2_NS : 3>(N, F, tpt, shared, iters, rpx);
    else if (K == 768) run_case<768, (LIN2_NCH && LIN2_FOR_K == 768) ? LIN2_NCH : 3, (LIN2_NCH && LIN2_FOR_K == 768) ? LIN2_NS : 3>(N, F, tpt, shared, iters, rpx);
    else if (K == 1280) run_case<1280, (LIN2_NCH && LIN2_FOR_K == 1280) ? LIN2_NCH : 4, (LIN2_NCH && LIN2_FOR_K == 1280) ? LIN2_NS : 4>(N, F, tpt, shared, iters, rpx);
    else if (K == 384) run_case<384, 3, 3>(N, F, tpt, shared, iters, rpx);
    else printf("unsupported shape\n");
    return 0;
}
