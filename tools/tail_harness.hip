// Stand-alone check + timing of the row-owning tail kernel (k_tail.hip.h): up-projection -> GELU in registers -> down-projection + attention
// out-projection + gated residual + next LayerNorm, against a double-precision host restatement on sampled token rows.
//   build: hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/tail_harness.hip -o tools/_exp/tail_harness
//   run:   tools/_exp/tail_harness [tokens] [D] [M] [tokens_per_traj] [shared_mods 0/1] [iters] [HHD] [grid]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

#ifdef TAIL_STAMPED  // tools/build_harness.sh tail <name> -DTAIL_STAMP: the product kernel + tools/experiments/tail_stamps.patch (s_memtime sums per phase)
#include "_exp/k_tail_stamped.hip.h"
#else
#include "../lam_slide_amd/csrc/k_tail.hip.h"
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
static float bf2f_host(u16 v) {
    unsigned u = (unsigned)v << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

template <int D, int HHD>
void run_case(int N, int M, int tpt, int shared, int iters, int grid_arg) {
    using C = TailCfg<D, HHD>;
    constexpr int NW = C::NW;
    const int Npad = (N + 255) / 256 * 256, F1 = 3 * HHD + M, F1pad = (F1 + 255) / 256 * 256, K2 = HHD + M, Dpad = (D + 255) / 256 * 256;
    const int ntraj = (N + tpt - 1) / tpt, MODW = 14 * D, mod_stride = shared ? 0 : MODW;
    printf("case N=%d D=%d HHD=%d M=%d tokens/traj=%d shared_mods=%d waves/wg=%d  (chunk %d KiB, ring slots %d, DMA per wave+chunk %d)\n", N, D, HHD, M, tpt, shared, NW,
           C::CH / 1024, C::NS, C::PPW);
    std::vector<u16> hW1((size_t)F1pad * D), hW2((size_t)Dpad * K2), hA((size_t)Npad * D), hZ((size_t)Npad * K2);
    std::vector<float> hb1(F1pad), hb2(D), hmods((size_t)(shared ? 1 : ntraj) * MODW), hh((size_t)N * D);
    for (auto &v : hW1) v = f2bf_host(rndf() * 0.08f);
    for (auto &v : hW2) v = f2bf_host(rndf() * 0.05f);
    for (auto &v : hA) v = f2bf_host(rndf() * 1.5f);
    for (auto &v : hZ) v = f2bf_host(rndf() * 1.0f);
    for (auto &v : hb1) v = rndf() * 0.3f;
    for (auto &v : hb2) v = rndf() * 0.3f;
    for (auto &v : hmods) v = rndf();
    for (auto &v : hh) v = rndf() * 2.0f;
    u16 *W1, *W2, *A, *A2, *Z, *wt;
    float *b1, *b2, *mods, *h;
    CK(hipMalloc(&W1, hW1.size() * 2)); CK(hipMalloc(&W2, hW2.size() * 2)); CK(hipMalloc(&A, hA.size() * 2)); CK(hipMalloc(&A2, hA.size() * 2));
    CK(hipMalloc(&Z, hZ.size() * 2)); CK(hipMalloc(&wt, C::stream_bytes(M)));
    CK(hipMalloc(&b1, hb1.size() * 4)); CK(hipMalloc(&b2, hb2.size() * 4)); CK(hipMalloc(&mods, hmods.size() * 4)); CK(hipMalloc(&h, (size_t)Npad * D * 4));
    CK(hipMemcpy(W1, hW1.data(), hW1.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(W2, hW2.data(), hW2.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(A2, 0, hA.size() * 2));
    CK(hipMemcpy(Z, hZ.data(), hZ.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(b1, hb1.data(), hb1.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b2, hb2.data(), hb2.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(mods, hmods.data(), hmods.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(h, hh.data(), hh.size() * 4, hipMemcpyHostToDevice));
    auto magic_of = [](int dv) { return dv == 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)dv + 1); };
    // modulation row of a trajectory: [shift | scale | gate] of this sub-block, then [shift' | scale'] of the next
    const float *gate = mods + 2 * D, *nshift = mods + 3 * D, *nscale = mods + 4 * D;

    hipLaunchKernelGGL(k_tail_pack, dim3(256), dim3(256), 0, 0, wt, W1, W2, D, HHD, M);
    CK(hipDeviceSynchronize());
    auto kern = k_tail<D, HHD>;
    const size_t lds = C::lds_bytes(M) + (getenv("TAIL_LDS_PAD") ? atoi(getenv("TAIL_LDS_PAD")) : 0);
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    const int ntile = (N + 255) / 256;
    int wgs = 256;
    const int grid = grid_arg > 0 ? grid_arg : std::min((N + 31) / 32, wgs);
#ifdef TAIL_STAMPED
    unsigned long long *stamps;
    CK(hipMalloc(&stamps, 256 * 8 * 8 * 8));
    CK(hipMemset(stamps, 0, 256 * 8 * 8 * 8));
#endif
    TailArgs ta{wt, A, Z, b1 + 3 * HHD, b2, gate, h, A2, nshift, nscale, N, M, K2, mod_stride, tpt, magic_of(tpt)};
#ifdef TAIL_STAMPED
    ta.stamps = stamps;
#endif
    printf("  lds %zu bytes, grid %d x %d threads, %d tiles of %d tokens, stream %.2f MB\n", lds, grid, NW * 64, ntile, 256, C::stream_bytes(M) / 1e6);
    const int reps = getenv("TAIL_REPS") ? atoi(getenv("TAIL_REPS")) : 1;
    for (int rep = 0; rep < reps; ++rep) {
    CK(hipMemcpy(h, hh.data(), hh.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(A2, 0, hA.size() * 2));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, 0, ta);
    CK(hipDeviceSynchronize());

    std::vector<float> r1(hh.size());
    std::vector<u16> ra(hA.size());
    CK(hipMemcpy(r1.data(), h, hh.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ra.data(), A2, hA.size() * 2, hipMemcpyDeviceToHost));
    // host restatement on sampled rows
    std::vector<int> rows;
    for (int k = 0; k < 48; ++k) rows.push_back((int)((long)k * (N - 1) / 47));
    for (int k = 0; k < 16; ++k) rows.push_back(std::min(N - 1, 255 + k * 7));
    for (int k = 0; k < 8; ++k) rows.push_back(std::max(0, N - 1 - k));
    double worst_h = 0, worst_a = 0, rms_upd = 0, cnt = 0;
    int n_bad = 0;
    size_t a_flips = 0, a_total = 0;
    for (int n : rows) {
        const size_t mo = (size_t)(shared ? 0 : n / tpt) * MODW;
        std::vector<double> g(M), hn(D);
        for (int m = 0; m < M; ++m) {
            double u = hb1[3 * HHD + m];
            for (int k = 0; k < D; ++k) u += (double)bf2f_host(hW1[(size_t)(3 * HHD + m) * D + k]) * bf2f_host(hA[(size_t)n * D + k]);
            const double ge = 0.5 * u * (1.0 + erf(u / sqrt(2.0)));
            g[m] = bf2f_host(f2bf_host((float)ge));
        }
        double mean = 0;
        for (int f = 0; f < D; ++f) {
            double o = hb2[f];
            for (int k = 0; k < HHD; ++k) o += (double)bf2f_host(hW2[(size_t)f * K2 + k]) * bf2f_host(hZ[(size_t)n * K2 + k]);
            for (int m = 0; m < M; ++m) o += (double)bf2f_host(hW2[(size_t)f * K2 + HHD + m]) * g[m];
            const double upd = (double)hmods[mo + 2 * D + f] * o;
            hn[f] = (double)hh[(size_t)n * D + f] + upd;
            mean += hn[f];
            rms_upd += upd * upd;
            cnt += 1;
            worst_h = std::max(worst_h, fabs(hn[f] - (double)r1[(size_t)n * D + f]));
            if (fabs(hn[f] - (double)r1[(size_t)n * D + f]) > 5e-3 && n_bad++ < 12) printf("    bad h: n=%d (wave tile %d, row %d) f=%d got %.5f want %.5f start %.5f\n", n, n / 32, n % 32, f, r1[(size_t)n * D + f], hn[f], hh[(size_t)n * D + f]);
        }
        mean /= D;
        double var = 0;
        for (int f = 0; f < D; ++f) var += (hn[f] - mean) * (hn[f] - mean);
        const double rstd = 1.0 / sqrt(var / D + 1e-6);
        for (int f = 0; f < D; ++f) {
            const double y = (hn[f] - mean) * rstd * (1.0 + hmods[mo + 4 * D + f]) + hmods[mo + 3 * D + f];
            const double got = bf2f_host(ra[(size_t)n * D + f]);
            const double e = fabs(y - got) / std::max(1.0, fabs(y));
            worst_a = std::max(worst_a, e);
            ++a_total;
            if (e > 0.0045) ++a_flips;
        }
    }
    rms_upd = sqrt(rms_upd / cnt);
    // untouched rows beyond N?  (h has exactly N rows; a_next rows >= N must stay zero)
    size_t pad_dirty = 0;
    for (size_t i = (size_t)N * D; i < ra.size(); ++i) pad_dirty += ra[i] != 0;
    printf("  rows checked %zu: max |h' - ref| = %.3e (rms of the update %.3e -> %.2e relative), a_next worst rel %.3e, beyond one bf16 ulp: %zu of %zu, dirty pad elements %zu\n",
           rows.size(), worst_h, rms_upd, worst_h / rms_upd, worst_a, a_flips, a_total, pad_dirty);
    printf("  %s\n", (worst_h / rms_upd < 5e-3 && worst_a < 1.2e-2 && pad_dirty == 0) ? "RESULT OK" : "RESULT WRONG");
    }

    hipEvent_t ev0, ev1;
    CK(hipEventCreate(&ev0)); CK(hipEventCreate(&ev1));
    const double flop = 2.0 * N * ((double)D * M * 2 + (double)HHD * D);
    for (int round = 0; round < 3; ++round) {
        float ms = 0;
        CK(hipEventRecord(ev0, 0));
        for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, 0, ta);
        CK(hipEventRecord(ev1, 0)); CK(hipEventSynchronize(ev1)); CK(hipEventElapsedTime(&ms, ev0, ev1));
        printf("  round %d: %.4f ms/launch = %.1f us (%.0f TF/s, %.3f of 2.5 PF)\n", round, ms / iters, 1e3 * ms / iters, flop / (ms / iters * 1e-3) * 1e-12,
               flop / (ms / iters * 1e-3) * 1e-12 / 2500.0);
    }
#ifdef TAIL_STAMPED
    {   // one more launch, then the stamps of the last launch: cycles per round and phase
        CK(hipMemset(stamps, 0, 256 * 8 * 8 * 8));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, 0, ta);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> hs(256 * 8 * 8);
        CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
        for (int wg : {0, 77, 200})
            for (int w : {0, 4, 7}) {
                const unsigned long long *c = &hs[((size_t)wg * 8 + w) * 8];
                if (c[5]) printf("  wg %3d wave %d: %llu rounds; per round (shader cycles by s_memtime): z rows %.0f, z + O phase %.0f, a rows + transposition %.0f, mlp %.0f, epilogue %.0f\n", wg, w, c[5],
                                 (double)c[0] / c[5], (double)c[1] / c[5], (double)c[2] / c[5], (double)c[3] / c[5], (double)c[4] / c[5]);
            }
    }
#endif
    hipFree(W1); hipFree(W2); hipFree(A); hipFree(A2); hipFree(Z); hipFree(wt); hipFree(b1); hipFree(b2); hipFree(mods); hipFree(h);
}

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 163840, D = argc > 2 ? atoi(argv[2]) : 256, M = argc > 3 ? atoi(argv[3]) : 1024;
    const int tpt = argc > 4 ? atoi(argv[4]) : 160, shared = argc > 5 ? atoi(argv[5]) : 0, iters = argc > 6 ? atoi(argv[6]) : 20;
    const int HHD = argc > 7 ? atoi(argv[7]) : D, grid = argc > 8 ? atoi(argv[8]) : 0;
    if (D == 256 && HHD == 256) run_case<256, 256>(N, M, tpt, shared, iters, grid);
    else if (D == 128 && HHD == 128) run_case<128, 128>(N, M, tpt, shared, iters, grid);
    else printf("unsupported shape\n");
    return 0;
}
