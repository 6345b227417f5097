// Host-side launchers of the measured-and-rejected structures and A/B arms: included by lam_slide_amd/csrc/host_launch.hip.h ONLY in
// -DLSL_EXPERIMENTS builds (tools/build_experiments.sh puts tools/experiments/ on the include path); the product library has empty hooks
// in their place.  Inside the anonymous namespace of host_common.hip.h, behind launch_gemm_glds / device_cus / tune_int.
#pragma once

template <int BK, int NS, int NB, class Epi>
void launch_gemm_pp_t(const GemmArgs &g, const Epi &epi, hipStream_t st) {
    auto kern = k_gemm_pp<BK, NS, NB, Epi>;
    constexpr size_t lds = GemmPPCfg<BK, NS, Epi>::lds_bytes;
    LSL_ALLOW_LDS(kern, lds);
    const int tiles = ((g.N + 255) / 256) * ((g.F + 127) / 128);
    int grid = device_cus();
    grid -= grid % 8;
    if (grid > tiles) grid = tiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, g, epi);
}

// epilogue of tile i inside the main loop of tile i+1 (k_gemm_drain.hip.h); false when the shape is outside what it covers
template <class Epi>
bool launch_gemm_drain(const GemmArgs &g, const Epi &epi, hipStream_t st) {
    if (g.F % 256 != 0 || g.N % 128 != 0 || g.K % 64 != 0 || g.K / 64 < 2) return false;
    auto kern = k_gemm_drain<Epi>;
    constexpr size_t lds = GemmDrainCfg<Epi>::lds_bytes;
    LSL_ALLOW_LDS(kern, lds);
    const int tiles = (g.N / 128) * (g.F / 256);
    int grid = device_cus();
    grid -= grid % 8;
    if (grid > tiles) grid = tiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, g, epi);
    return true;
}

// ping-pong halves (k_gemm_pp.hip.h); false when the shape is outside what the schedule covers
template <int BK, int NS, class Epi>
bool launch_gemm_pp(const GemmArgs &g, const Epi &epi, hipStream_t st) {
    if (g.K % BK != 0 || g.F % 32 != 0) return false;
    const int E = g.K / BK - (NS - 1);  // intervals that carry epilogue pieces
    if (E < 1 || E > 64) return false;
    if (E <= 16) launch_gemm_pp_t<BK, NS, 1>(g, epi, st);
    else if (E <= 32) launch_gemm_pp_t<BK, NS, 2>(g, epi, st);
    else launch_gemm_pp_t<BK, NS, 4>(g, epi, st);
    return true;
}


// variants of launch_gemm that the product never takes
template <class Epi>
bool launch_gemm_experiment(int variant, const GemmArgs &g, const Epi &epi, hipStream_t st, bool pp_ok) {
    if (variant == 30 && launch_gemm_drain(g, epi, st)) return true;
    if (variant == 20 && pp_ok && launch_gemm_pp<32, 4>(g, epi, st)) return true;
    if (variant == 21 && pp_ok && launch_gemm_pp<64, 2>(g, epi, st)) return true;
    if (variant == 22 && pp_ok && launch_gemm_pp<64, 3>(g, epi, st)) return true;
    if (variant == 8 && g.F % 32 == 0 && pp_ok) { launch_gemm_glds<256, 256, 2, 4, 32, 3, true>(g, EpiPieces<Epi>(epi), st); return true; }  // variant 6 with the piece epilogue
    switch (variant) {
        case 26: if constexpr (std::is_same<Epi, EpiLinear2>::value) { launch_gemm_glds<192, 128, 2, 4, 64, 2, false>(g, epi, st); return true; } else break;  // 192 features x 128 tokens, 8 waves of 96 x 32
        case 29: if constexpr (std::is_same<Epi, EpiLinear2>::value) { launch_gemm_glds<192, 256, 2, 4, 64, 2, false>(g, epi, st); return true; } else break;  // 8 waves of 96 x 64
        case 23: launch_gemm_glds<512, 128, 4, 2, 32, 3, false>(g, epi, st); return true;  // whole residual rows per workgroup (F = 512): 120 KiB ring
        case 24: launch_gemm_glds<512, 128, 4, 2, 32, 2, false>(g, epi, st); return true;
        case 25: launch_gemm_glds<512, 128, 4, 2, 32, 3, true>(g, EpiPieces<Epi>(epi), st); return true;
        case 16: launch_gemm_glds<128, 256, 2, 4, 64, 2, false>(g, epi, st); return true;  // 128 features x 256 tokens, 8 waves of 64 x 64
        case 17: launch_gemm_glds<128, 256, 2, 4, 32, 3, false>(g, epi, st); return true;
        case 18: launch_gemm_glds<128, 256, 1, 8, 64, 2, false>(g, epi, st); return true;  // 8 waves of 128 x 32
        case 13: launch_gemm_glds<256, 128, 2, 2, 32, 2, true>(g, epi, st); return true;
        case 14: launch_gemm_glds<256, 128, 2, 2, 64, 2, true>(g, epi, st); return true;  // 4 waves, one per SIMD, 64-deep k-tiles, one workgroup per CU
        case 6: launch_gemm_glds<256, 256, 2, 4, 32, 3, true>(g, epi, st); return true;
        default: break;
    }
    return false;
}

// LSL_HEAD_MFMA=0: the scalar-FMA output head (A/B measurements)
template <int NE, int VEC>
bool launch_head_experiment(float *x, float *out, const float *h, const float *shift, const float *scale, int stride, const float *Wo, const float *bo, int n,
                            int C, int tpt, int do_step, float ax, float am, float aw, const float *noise, unsigned long long seed, unsigned step,
                            unsigned long long eo, float *trace, float as, const float *saved, float *save_out, hipStream_t st) {
    static const int mfma = tune_int("LSL_HEAD_MFMA", 1);
    if (mfma) return false;
    auto kern = k_head_step<NE, VEC>;
    constexpr size_t lds = head_lds_bytes<NE>();
    LSL_ALLOW_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3(std::min((n + HEAD_TOK - 1) / HEAD_TOK, 256)), dim3(256), lds, st, x, out, h, shift, scale, stride, Wo, bo, n, C, tpt,
                       do_step, ax, am, aw, noise, seed, step, eo, trace, as, saved, save_out);
    return true;
}
