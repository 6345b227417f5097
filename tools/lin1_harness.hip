// Stand-alone check + timing of the token-stationary linear1 kernel (k_lin1.hip.h) against the 256 x 256-tile kernel it replaces
// (k_gemm_glds<..., EpiLinear1<HDP>>, k_gemm.hip.h): the two must agree BIT FOR BIT on qkv and z.
//   build: hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/lin1_harness.hip -o tools/_exp/lin1_harness
//   run:   tools/_exp/lin1_harness [tokens] [D] [heads] [mlp_ratio] [iters] [grid]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

#include "../lam_slide_amd/csrc/k_gemm.hip.h"
#ifdef LIN1_PROBED  // tools/build_harness.sh lin1: the product kernel + tools/experiments/lin1_probes.patch (timing arms, cycle stamps: results wrong when set)
#include "_exp/k_lin1_probed.hip.h"
#else
#include "../lam_slide_amd/csrc/k_lin1.hip.h"
#endif

#ifndef LIN1_NW
#define LIN1_NW 8  // waves per workgroup of the token-stationary kernel (4: 128-token tiles, one wave per SIMD)
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

template <int HDP, int K>
void run_case(int N, int D, int H, int mlp_ratio, int iters, int grid_new, int pos_mode) {
    const int HHD = H * HDP, M = D * mlp_ratio, F = 3 * HHD + M;
    const int Npad = (N + 255) / 256 * 256, Fpad = (F + 255) / 256 * 256;
    const int n_pos = 256;
    printf("case N=%d D=%d H=%d hdp=%d M=%d F=%d pos_mode=%d\n", N, D, H, HDP, M, F, pos_mode);
    std::vector<u16> hW((size_t)Fpad * K), hX((size_t)Npad * K);
    std::vector<float> hb(Fpad), hq((size_t)n_pos * (HDP / 2) * 4), hk((size_t)n_pos * (HDP / 2) * 4);
    for (auto &v : hW) v = f2bf_host(rndf() * 0.08f);
    for (auto &v : hX) v = f2bf_host(rndf() * 1.5f);
    for (auto &v : hb) v = rndf() * 0.3f;
    for (auto &v : hq) v = rndf();
    for (auto &v : hk) v = rndf();
    u16 *W, *X, *qkv0, *z0, *qkv1, *z1;
    float *b;
    float4 *rq, *rk;
    const size_t qkv_bytes = (size_t)Npad * 3 * HHD * 2, z_bytes = (size_t)Npad * (HHD + M) * 2;
    CK(hipMalloc(&W, hW.size() * 2)); CK(hipMalloc(&X, hX.size() * 2)); CK(hipMalloc(&b, hb.size() * 4));
    CK(hipMalloc(&rq, hq.size() * 4)); CK(hipMalloc(&rk, hk.size() * 4));
    CK(hipMalloc(&qkv0, qkv_bytes)); CK(hipMalloc(&z0, z_bytes)); CK(hipMalloc(&qkv1, qkv_bytes)); CK(hipMalloc(&z1, z_bytes));
    CK(hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(X, hX.data(), hX.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(rq, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(rk, hk.data(), hk.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(qkv0, 0xEE, qkv_bytes)); CK(hipMemset(z0, 0xEE, z_bytes));
    CK(hipMemset(qkv1, 0xEE, qkv_bytes)); CK(hipMemset(z1, 0xEE, z_bytes));

    auto magic_of = [](int dv) { return dv == 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)dv + 1); };
    const int pdiv = pos_mode ? 256 : 1, pmod = pos_mode ? 30 : 256;
    const float inv_hd = 1.0f / (HDP == 32 ? (D / H) : (D / H)), premul = 1.4426950408889634f / sqrtf((float)(D / H));

    // ---- old kernel ----
    using Epi = EpiLinear1<HDP>;
    Epi e{b, nullptr, nullptr, nullptr, rq, rk, qkv0, z0, HHD, M, pdiv, pmod, magic_of(pdiv), magic_of(pmod), inv_hd, premul, 32};
    GemmArgs ga{W, X, F, N, K, 0, 0, 0};
    auto kold = k_gemm_glds<256, 256, 2, 4, 64, 2, true, Epi>;
    const size_t lds_old = GemmCfg<256, 256, 2, 4, 64, 2, true, Epi>::lds_bytes + (size_t)Fpad * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kold), hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    const int tiles = ((N + 255) / 256) * (Fpad / 256);
    const int grid_old = tiles < 256 ? tiles : 256;
    // ---- new kernel ----
    unsigned long long *dbg;
    CK(hipMalloc(&dbg, 256 * 8 * 14 * 8));
    CK(hipMemset(dbg, 0, 256 * 8 * 14 * 8));
#ifdef LIN1_PROBED
    Lin1Args la{W, X, b, rq, rk, qkv1, z1, F, N, HHD, M, pdiv, pmod, magic_of(pdiv), magic_of(pmod), inv_hd, premul, 1, dbg};
#else
    Lin1Args la{W, X, b, rq, rk, qkv1, z1, F, N, HHD, M, pdiv, pmod, magic_of(pdiv), magic_of(pmod), inv_hd, premul, 1};
#endif
    auto knew = k_linear1_ts<HDP, K, LIN1_NW>;
    const size_t lds_new = Lin1Cfg<HDP, K, LIN1_NW>::lds_bytes(F);
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(knew), hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    constexpr int TT = 32 * LIN1_NW;
    const long units = (long)((N + TT - 1) / TT) * (F / 32);
    int gnew = grid_new;
    if (gnew > units / 2) gnew = (int)(units / 2);
    if (getenv("LIN1_WPT")) {  // tile-aligned split for small launches, as launch_linear1_ts_t chooses it
        const int ntile = (N + TT - 1) / TT, wpt = std::min(256 / ntile, F / 64);
        if (wpt >= 2) { la.wpt = wpt; gnew = wpt * ntile; }
    }
    printf("  lds old %zu new %zu, grid old %d new %d (wpt %d)\n", lds_old, lds_new, grid_old, gnew, la.wpt);

    hipLaunchKernelGGL(kold, dim3(grid_old), dim3(512), lds_old, 0, ga, e);
    CK(hipDeviceSynchronize());
    printf("  old kernel ran\n"); fflush(stdout);
    hipLaunchKernelGGL(knew, dim3(gnew), dim3(64 * LIN1_NW), lds_new, 0, la);
    CK(hipDeviceSynchronize());
    printf("  new kernel ran\n"); fflush(stdout);

    // compare the valid region (N rows)
    std::vector<u16> a0(qkv_bytes / 2), a1(qkv_bytes / 2), c0(z_bytes / 2), c1(z_bytes / 2);
    CK(hipMemcpy(a0.data(), qkv0, qkv_bytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(a1.data(), qkv1, qkv_bytes, hipMemcpyDeviceToHost));
    CK(hipMemcpy(c0.data(), z0, z_bytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(c1.data(), z1, z_bytes, hipMemcpyDeviceToHost));
    size_t bad_q = 0, bad_z = 0, first_q = (size_t)-1, first_z = (size_t)-1;
    for (size_t n = 0; n < (size_t)N; ++n) {
        for (int f = 0; f < 3 * HHD; ++f) {
            const size_t i = n * 3 * HHD + f;
            if (a0[i] != a1[i]) { if (!bad_q) first_q = i; ++bad_q; }
        }
        for (int f = HHD; f < HHD + M; ++f) {  // linear1 writes the mlp part of z; columns below HHD belong to the attention kernel
            const size_t i = n * (HHD + M) + f;
            if (c0[i] != c1[i]) { if (!bad_z) first_z = i; ++bad_z; }
        }
    }
    printf("  mismatches: qkv %zu (first n=%zu f=%zu old %04x new %04x), z %zu (first n=%zu f=%zu old %04x new %04x)\n", bad_q,
           bad_q ? first_q / (3 * HHD) : 0, bad_q ? first_q % (3 * HHD) : 0, bad_q ? a0[first_q] : 0, bad_q ? a1[first_q] : 0, bad_z,
           bad_z ? first_z / (HHD + M) : 0, bad_z ? first_z % (HHD + M) : 0, bad_z ? c0[first_z] : 0, bad_z ? c1[first_z] : 0);
    printf("  %s\n", bad_q + bad_z == 0 ? "BITS EQUAL" : "DIFFERENT");

#if defined(LIN1_PROBE) && (LIN1_PROBE & 128)
    {
        std::vector<unsigned long long> hd(256 * 8 * 4);
        CK(hipMemcpy(hd.data(), dbg, hd.size() * 8, hipMemcpyDeviceToHost));
        std::vector<unsigned long long> hc(256 * 8 * 10);
        CK(hipMemcpy(hc.data(), dbg + (size_t)gnew * 8 * 4, hc.size() * 8, hipMemcpyDeviceToHost));
        for (int wg : {0, 100, 255})
            for (int w : {0, 4}) {
                const unsigned long long *c = &hc[((size_t)wg * 8 + w) * 10];
                if (c[1]) printf("  wg %3d wave %d: kernel %.0f core cycles in %.1f us -> %.0f MHz; %llu segments; later ones: older ops done %.0f, activation loads done +%.0f, barrier +%.0f cycles\n", wg, w,
                                 (double)c[0], c[1] / 100.0, (double)c[0] / (c[1] / 100.0), c[5], c[5] > 1 ? (double)c[2] / (c[5] - 1) : 0.0, c[5] > 1 ? (double)c[3] / (c[5] - 1) : 0.0, c[5] > 1 ? (double)c[4] / (c[5] - 1) : 0.0);
                if (c[5] > 1) printf("       per later segment: compute-only step %.0f, fused steps %.0f, next segment's requests %.0f, drain + flush %.0f cycles\n", (double)c[6] / (c[5] - 1), (double)c[7] / (c[5] - 1), (double)c[8] / (c[5] - 1), (double)c[9] / (c[5] - 1));
            }
        for (int wg : {0, 1, 100, 255})
            for (int w : {0, 3, 4, 7}) {
                const unsigned long long *d = &hd[((size_t)wg * 8 + w) * 4];
                if (d[3]) printf("  wg %3d wave %d: steps %llu  head %.0f  issue+flush+init %.0f  body %.0f cycles/step\n", wg, w, d[3], (double)d[0] / d[3], (double)d[1] / d[3], (double)d[2] / d[3]);
            }
    }
#endif
    // timing, interleaved rounds
    hipEvent_t ev0, ev1;
    CK(hipEventCreate(&ev0)); CK(hipEventCreate(&ev1));
    const double flop = 2.0 * N * (double)K * F;
    for (int round = 0; round < 3; ++round) {
        float ms_old = 0, ms_new = 0;
        CK(hipEventRecord(ev0, 0));
        for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(kold, dim3(grid_old), dim3(512), lds_old, 0, ga, e);
        CK(hipEventRecord(ev1, 0)); CK(hipEventSynchronize(ev1)); CK(hipEventElapsedTime(&ms_old, ev0, ev1));
        CK(hipEventRecord(ev0, 0));
        for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(knew, dim3(gnew), dim3(64 * LIN1_NW), lds_new, 0, la);
        CK(hipEventRecord(ev1, 0)); CK(hipEventSynchronize(ev1)); CK(hipEventElapsedTime(&ms_new, ev0, ev1));
        printf("  round %d: old %.4f ms/launch (%.0f TF/s)   new %.4f ms/launch (%.0f TF/s)\n", round, ms_old / iters,
               flop / (ms_old / iters * 1e-3) * 1e-12, ms_new / iters, flop / (ms_new / iters * 1e-3) * 1e-12);
    }
    hipFree(W); hipFree(X); hipFree(b); hipFree(rq); hipFree(rk); hipFree(qkv0); hipFree(z0); hipFree(qkv1); hipFree(z1);
}

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 245760, D = argc > 2 ? atoi(argv[2]) : 512, H = argc > 3 ? atoi(argv[3]) : 16;
    const int mr = argc > 4 ? atoi(argv[4]) : 2, iters = argc > 5 ? atoi(argv[5]) : 20, grid = argc > 6 ? atoi(argv[6]) : 256;
    const int pos_mode = argc > 7 ? atoi(argv[7]) : 0;
    const int hd = D / H, hdp = hd <= 16 ? 16 : 32;
    if (D == 512 && hdp == 32) run_case<32, 512>(N, D, H, mr, iters, grid, pos_mode);
    else if (D == 256 && hdp == 16) run_case<16, 256>(N, D, H, mr, iters, grid, pos_mode);
    else if (D == 256 && hdp == 32) run_case<32, 256>(N, D, H, mr, iters, grid, pos_mode);
    else if (D == 128 && hdp == 32) run_case<32, 128>(N, D, H, mr, iters, grid, pos_mode);
    else if (D == 384 && hdp == 32) run_case<32, 384>(N, D, H, mr, iters, grid, pos_mode);
    else printf("unsupported shape\n");
    return 0;
}
