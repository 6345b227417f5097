// Host side, part 2 of 4: one launch helper per kernel family - grid / LDS arithmetic and the choice between instances (which depends on
// the model and on T, L only where results could differ, never on the batch).  Inside the anonymous namespace opened by host_common.hip.h.
#pragma once

// ---- launch helpers -------------------------------------------------------------------------------

// Rejected structures and A/B arms (ping-pong / drain GEMMs, tilings 6 / 8 / 13-18 / 23-26 / 29, the scalar output head, every LSL_* tuning knob)
// live in tools/experiments/host_launch_experiments.hip.h, on the include path of tools/build_experiments.sh only; the product has empty hooks.
#ifdef LSL_EXPERIMENTS
constexpr bool lsl_experiments = true;
template <int NE, int VEC>
bool launch_head_experiment(float *, float *, const float *, const float *, const float *, int, const float *, const float *, int, int, int, int, float, float,
                            float, const float *, unsigned long long, unsigned, unsigned long long, float *, float, const float *, float *, hipStream_t);
template <class Epi>
bool launch_gemm_experiment(int, const GemmArgs &, const Epi &, hipStream_t, bool);
#else
constexpr bool lsl_experiments = false;
template <int NE, int VEC, class... A>
bool launch_head_experiment(A...) { return false; }
template <class Epi>
bool launch_gemm_experiment(int, const GemmArgs &, const Epi &, hipStream_t, bool) { return false; }
#endif

int device_cus();
int env_int(const char *name, int dflt);

template <int NE, int VEC>
void launch_ln_mod_t(u16 *a, const float *h, const float *shift, const float *scale, int stride, int n, int tpt, hipStream_t st) {
    if constexpr (NE % 4 == 0) {
        static const int persist = tune_int("LSL_LN_PERSIST", 16);  // workgroups per CU of the persistent form; 0 = one wave per token
        if (persist > 0) {
            const int grid = std::min((n + 3) / 4, device_cus() * persist);
            static const int nt = (tune_int("LSL_NT", 3) >> 3) & 1;
            hipLaunchKernelGGL((k_ln_modulate_v4<NE>), dim3(grid), dim3(256), 0, st, a, h, shift, scale, stride, n, tpt, nt);
            return;
        }
    }
    hipLaunchKernelGGL((k_ln_modulate<NE, VEC>), dim3((n + 3) / 4), dim3(256), 0, st, a, h, shift, scale, stride, n, tpt);
}
template <int NE, int VEC>
void launch_ln_inplace_t(float *h, int n, float eps, hipStream_t st) {
    hipLaunchKernelGGL((k_ln_inplace<NE, VEC>), dim3((n + 3) / 4), dim3(256), 0, st, h, n, eps);
}
int device_cus();
int env_int(const char *name, int dflt);

template <int NE, int VEC>
void launch_head_mfma(float *x, float *out, const float *h, const float *shift, const float *scale, int stride, const float *Wo,
                      const float *bo, int n, int C, int tpt, int do_step, float ax, float am, float aw, const float *noise,
                      unsigned long long seed, unsigned step, unsigned long long eo, float *trace, float as, const float *saved, float *save_out,
                      hipStream_t st) {
    auto kern = k_head_step_mfma<NE, VEC>;
    const size_t lds = head_mfma_lds_bytes<NE>(C <= 32);
    LSL_ALLOW_LDS(kern, head_mfma_lds_bytes<NE>(false));
    // two workgroups per CU where the LDS image allows it (any hidden size with <= 32 channels): one workgroup's LayerNorm / weight-load latencies under the other's MFMAs
    static const int per_cu = tune_int("LSL_HEAD_PER_CU", 2);
    const int wgs = device_cus() * (per_cu >= 2 && 2 * lds <= (size_t)160 * 1024 ? 2 : 1);
    hipLaunchKernelGGL(kern, dim3(std::min((n + HEAD_TOK - 1) / HEAD_TOK, wgs)), dim3(256), lds, st, x, out, h, shift, scale, stride,
                       Wo, bo, n, C, tpt, do_step, ax, am, aw, noise, seed, step, eo, trace, as, saved, save_out);
}

template <int NE, int VEC>
void launch_head_t(float *x, float *out, const float *h, const float *shift, const float *scale, int stride, const float *Wo,
                   const float *bo, int n, int C, int tpt, int do_step, float ax, float am, float aw, const float *noise,
                   unsigned long long seed, unsigned step, unsigned long long eo, float *trace, float as, const float *saved, float *save_out,
                   hipStream_t st) {
    if (launch_head_experiment<NE, VEC>(x, out, h, shift, scale, stride, Wo, bo, n, C, tpt, do_step, ax, am, aw, noise, seed, step, eo, trace, as, saved, save_out, st)) return;
    launch_head_mfma<NE, VEC>(x, out, h, shift, scale, stride, Wo, bo, n, C, tpt, do_step, ax, am, aw, noise, seed, step, eo, trace, as, saved, save_out, st);
}

#define DISPATCH_D(D, FN, ...)                          \
    switch ((D) / 64) {                                 \
        case 1: FN<1, 1>(__VA_ARGS__); break;           \
        case 2: FN<2, 2>(__VA_ARGS__); break;           \
        case 3: FN<3, 1>(__VA_ARGS__); break;           \
        case 4: FN<4, 2>(__VA_ARGS__); break;           \
        case 5: FN<5, 1>(__VA_ARGS__); break;           \
        case 6: FN<6, 2>(__VA_ARGS__); break;           \
        case 7: FN<7, 1>(__VA_ARGS__); break;           \
        default: FN<8, 2>(__VA_ARGS__); break;          \
    }

int device_cus();

// (stats / npad: ln_fuse handles, launch_embed<1> only - the rows' statistics for the first sub-block's fused LayerNorm; embed_stats_ok says where)
bool embed_stats_ok(int C, int D) { return C <= 32 && D % 256 == 0 && D <= 512; }
template <int MODE>
int launch_embed(float *out, const float *in, const float *W, const float *b, const float *b2, const float *me,
                 const int64_t *mask, const float *base, int n, int C, int D, hipStream_t st, float2 *stats = nullptr, int npad = 0) {
    if (C > 32 && C % 8 == 0 && C <= 128 && D % 32 == 0) {  // wide inputs: fp32 MFMA form (k_embed_mfma)
        const int tpw = 3, ngrp = (D / 32 + tpw - 1) / tpw;
        const long units = (long)((n + 31) / 32) * ngrp;
        const dim3 g((unsigned)((units + 3) / 4));
        switch (C / 8) {
#define LSL_EMB_CASE(CK) case CK: hipLaunchKernelGGL((k_embed_mfma<MODE, CK>), g, dim3(256), 0, st, out, in, W, b, b2, me, mask, base, n, C, D, tpw); return 0;
            LSL_EMB_CASE(5) LSL_EMB_CASE(6) LSL_EMB_CASE(7) LSL_EMB_CASE(8) LSL_EMB_CASE(9) LSL_EMB_CASE(10) LSL_EMB_CASE(11) LSL_EMB_CASE(12)
            LSL_EMB_CASE(13) LSL_EMB_CASE(14) LSL_EMB_CASE(15) LSL_EMB_CASE(16)
#undef LSL_EMB_CASE
        }
    }
    // persistent workgroups (weights fetched once each) over tiles of 64 tokens
    int tok = EMB_TOK;
    // 32 inputs, hidden <= 512 (every shipped narrow-input model): the weight rows reach the registers through LDS (k_embed), which makes
    // a workgroup's prologue cheap enough for 32- or 16-token tiles when the launch has fewer than two 64-token tiles per CU
    static const int stage = tune_int("LSL_EMBED_LDS", 1);
    const bool w_lds = stage && C == 32 && D % 4 == 0 && D <= 512;
    if (w_lds)
        while (tok > 16 && (n + tok - 1) / tok < 2 * device_cus()) tok /= 2;
    const dim3 grid(std::min((n + tok - 1) / tok, 2 * device_cus())), blk(256);
    if (C <= 32) {
        const size_t lds = w_lds ? embed_w_lds_bytes<32, 4>(D) : 0;
        if (stats && MODE == 1 && embed_stats_ok(C, D)) {
            auto kern = k_embed<32, 1, 4, true>;
            LSL_ALLOW_LDS(kern, (embed_w_lds_bytes<32, 4>(512)));
            hipLaunchKernelGGL(kern, grid, blk, lds, st, out, in, W, b, b2, me, mask, base, n, C, D, w_lds ? 1 : 0, tok, stats, npad);
            return 0;
        }
        auto kern = k_embed<32, MODE, 4>;
        LSL_ALLOW_LDS(kern, (embed_w_lds_bytes<32, 4>(512)));
        hipLaunchKernelGGL(kern, grid, blk, lds, st, out, in, W, b, b2, me, mask, base, n, C, D, w_lds ? 1 : 0, tok, (float2 *)nullptr, 0);
    } else if (C <= 64) hipLaunchKernelGGL((k_embed<64, MODE, 2>), grid, blk, 0, st, out, in, W, b, b2, me, mask, base, n, C, D, 0, tok, (float2 *)nullptr, 0);
    else if (C <= 96) hipLaunchKernelGGL((k_embed<96, MODE, 2>), grid, blk, 0, st, out, in, W, b, b2, me, mask, base, n, C, D, 0, tok, (float2 *)nullptr, 0);
    else hipLaunchKernelGGL((k_embed<128, MODE, 1>), grid, blk, 0, st, out, in, W, b, b2, me, mask, base, n, C, D, 0, tok, (float2 *)nullptr, 0);
    return 0;
}

int env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}
// Kernel-selection / timing knobs (LSL_GEMM*, LSL_NT, LSL_STAGGER, LSL_PROBE, ...): read from the environment only in
// -DLSL_EXPERIMENTS builds; the product library always runs its measured defaults.
int tune_int(const char *name, int dflt) { return lsl_experiments ? env_int(name, dflt) : dflt; }

int device_cus() {  // of the current device (entry points switch to the stream's device first)
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    int n = cache[dev & 63].load(std::memory_order_relaxed);
    if (n > 0) return n;
    n = 256;
    (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    if (n <= 0) n = 256;
    cache[dev & 63].store(n, std::memory_order_relaxed);
    return n;
}

template <int BF, int BT, int NWF, int NWT, int BK, int NS, bool PERSIST, class Epi>
void launch_gemm_glds(const GemmArgs &g, const Epi &epi, hipStream_t st) {
    auto kern = k_gemm_glds<BF, BT, NWF, NWT, BK, NS, PERSIST, Epi>;
    // + the bias vector of the whole GEMM, kept in LDS by epilogues that start the accumulators from it (k_gemm.hip.h)
    const size_t lds = GemmCfg<BF, BT, NWF, NWT, BK, NS, PERSIST, Epi>::lds_bytes + (Epi::lds_bias ? (size_t)((g.F + BF - 1) / BF) * BF * 4 : 0);
    LSL_ALLOW_LDS(kern, (size_t)163840);
    const int ntt = (g.N + BT - 1) / BT, tiles = ntt * ((g.F + BF - 1) / BF);
    int grid = tiles;
    GemmArgs ga = g;
    ga.rows = 0;
    if (PERSIST) {  // as many workgroups as fit at once (LDS-limited), a multiple of 8 so the XCD mapping stays regular
        const int per_cu = (int)(163840 / lds) < 1 ? 1 : (int)(163840 / lds);
        grid = device_cus() * per_cu;
        grid -= grid % 8;
        if (grid > tiles) grid = tiles;
        // row-owner walk (the epilogue finishes whole token rows: fused LayerNorm of linear2) only when there are at least as many
        // token tiles as workgroups; smaller launches keep the flat list, which spreads the feature tiles over more CUs
        if (Epi::row_owner && g.rows && ntt >= grid) ga.rows = 1;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NWF * NWT * 64), lds, st, ga, epi);
}
// whether launch_gemm_glds would take the row-owner walk for this launch (the caller then lets the epilogue write the next LayerNorm)
template <int BF, int BT, int NWF, int NWT, int BK, int NS, class Epi>
bool gemm_rows_walk(int F, int N) {
    const size_t lds = GemmCfg<BF, BT, NWF, NWT, BK, NS, true, Epi>::lds_bytes;
    const int per_cu = (int)(163840 / lds) < 1 ? 1 : (int)(163840 / lds);
    int grid = device_cus() * per_cu;
    grid -= grid % 8;
    const int ntt = (N + BT - 1) / BT, tiles = ntt * ((F + BF - 1) / BF);
    if (grid > tiles) grid = tiles;
    return ntt >= grid;
}


#ifdef LSL_EXPERIMENTS
#include "host_launch_experiments.hip.h"  // (needs launch_gemm_glds / device_cus above)
#endif

// linear1 on the token-stationary kernel (k_lin1.hip.h): hidden sizes 128 / 256 / 384 / 512, sections (q | k | v | mlp) on multiples of 64
// features.  Same bits as the tile kernels below (tools/lin1_harness.hip), so the choice between them may depend on the launch size.
template <int HDP, int K, int NW = 8, bool LNF = false>
void launch_linear1_ts_t(const Lin1Args &a, hipStream_t st) {
    using C = Lin1Cfg<HDP, K, NW>;
    auto kern = k_linear1_ts<HDP, K, NW, LNF>;
    LSL_ALLOW_LDS(kern, (size_t)163840);
    const int ntile = (a.N + C::TT - 1) / C::TT, nb = a.F / 32;
    const long units = (long)ntile * nb;
    int grid = (int)std::min<long>(device_cus(), units / 2);
    Lin1Args b = a;
    // fewer tiles than workgroups: whole workgroups per tile, one segment each (k_lin1.hip.h "Work split"); same bits either way
    static const int align = tune_int("LSL_LIN1_ALIGN", 1);
    const int wpt = std::min(device_cus() / ntile, nb / 2);
    b.wpt = align && wpt >= 2 ? wpt : 0;
    if (b.wpt) grid = b.wpt * ntile;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), LNF ? C::lds_bytes_lnf(a.F) : C::lds_bytes(a.F), st, b);
}
// K = 512, few tokens (md17_bench B = 1: 7 680): 4-wave workgroups on 128-token tiles - twice the tiles, half the activation prologue per
// workgroup, one wave per SIMD (512 registers: no scratch).  Same bits (tools/lin1_harness.hip -DLIN1_NW=4); 3 840 tokens 24.0 -> 19.9 us,
// 7 680 34.4 -> 30.4, 15 360 50.4 -> 53.0 (so: up to 10 240); slower at every size for K <= 384 and at every large launch
// (profiles/r06_experiments.txt section 6).
int linear1_ts_waves(int D, int N) {
    static const int nw4_max = tune_int("LSL_LIN1_NW4_MAX", 10240);
    return D == 512 && N <= nw4_max ? 4 : 8;
}
template <int HDP>
void launch_linear1_ts_512(const Lin1Args &a, hipStream_t st) {
    if (linear1_ts_waves(512, a.N) == 4) return launch_linear1_ts_t<HDP, 512, 4>(a, st);
    return launch_linear1_ts_t<HDP, 512, 8>(a, st);
}
bool linear1_ts_ok(int hdp, int D, int F1, int HHD, int N) {
    static const int on = tune_int("LSL_LIN1_TS", 1);
    if (!on || (hdp != 16 && hdp != 32) || (D != 128 && D != 256 && D != 384 && D != 512) || F1 % 64 != 0 || HHD % 64 != 0 || N < 1) return false;
    return (size_t)(D <= 256 ? 4 : 3) * 32 * (2 * D + 16) + 8 * 4096 + (size_t)F1 * 4 <= (size_t)163840;  // weight ring (Lin1Cfg::NS slots) + staging + bias vector (Lin1Cfg::lds_bytes)
}
// LayerNorm + modulate inside linear1's activation load (k_lin1.hip.h, LNF instances): no LayerNorm launch, no bf16 `a` buffer - the kernel reads
// the fp32 residual stream.  Per HANDLE (lsl_model_set_ln_fuse; LSL_LN_FUSE=1 makes it the default of new handles, 0 disables it), never per
// batch: its rounding differs from the standalone kernel's.  Shapes: the token-stationary kernel's, with the (1 + scale | shift) rows of every
// trajectory a 256-token tile can touch in LDS - one shared row, or tokens per trajectory >= 128 (hidden <= 256) / >= 256 (wider).
int ln_fuse_env() {
    static const int v = env_int("LSL_LN_FUSE", -1);
    return v;
}
bool linear1_lnf_ok(int hdp, int D, int F1, int HHD, int N, int tpt, int mod_stride) {
    if (ln_fuse_env() == 0 || !linear1_ts_ok(hdp, D, F1, HHD, N) || D % 128 != 0) return false;
    if (mod_stride != 0 && tpt < (D <= 256 ? 128 : 256)) return false;
    return (size_t)(D <= 256 ? 4 : 3) * 32 * (2 * D + 16) + 8 * 4096 + (size_t)F1 * 4 + (size_t)(D <= 256 ? 3 : 2) * 2 * D * 4 <= (size_t)163840;
}
void launch_linear1_lnf(int hdp, int D, const Lin1Args &a, hipStream_t st) {
    switch ((hdp == 32 ? 0 : 4) + D / 128 - 1) {
        case 0: return launch_linear1_ts_t<32, 128, 8, true>(a, st);
        case 1: return launch_linear1_ts_t<32, 256, 8, true>(a, st);
        case 2: return launch_linear1_ts_t<32, 384, 8, true>(a, st);
        case 3: return launch_linear1_ts_t<32, 512, 8, true>(a, st);
        case 4: return launch_linear1_ts_t<16, 128, 8, true>(a, st);
        case 5: return launch_linear1_ts_t<16, 256, 8, true>(a, st);
        case 6: return launch_linear1_ts_t<16, 384, 8, true>(a, st);
        default: return launch_linear1_ts_t<16, 512, 8, true>(a, st);
    }
}
void launch_linear1_ts(int hdp, int D, const Lin1Args &a, hipStream_t st) {
    switch ((hdp == 32 ? 0 : 4) + D / 128 - 1) {
        case 0: return launch_linear1_ts_t<32, 128>(a, st);
        case 1: return launch_linear1_ts_t<32, 256>(a, st);
        case 2: return launch_linear1_ts_t<32, 384>(a, st);
        case 3: return launch_linear1_ts_512<32>(a, st);
        case 4: return launch_linear1_ts_t<16, 128>(a, st);
        case 5: return launch_linear1_ts_t<16, 256>(a, st);
        case 6: return launch_linear1_ts_t<16, 384>(a, st);
        default: return launch_linear1_ts_512<16>(a, st);
    }
}

// linear2 + gated residual update on the weight-stationary kernel (k_lin2.hip.h): F a multiple of 128, K2 one of the instantiated widths
// (K2 / 8 stationary registers per wave: 2 048, peptide, does not fit).  Same bits as the tile kernels (tools/lin2_harness.hip), so the
// choice may depend on the launch.  LSL_LIN2_WS=0 (read in the product too: the GPU suite compares the two paths bit for bit) turns it off.
bool linear2_ws_shape_ok(int D, int K2) {
    static const int on = env_int("LSL_LIN2_WS", 1);
    return on && D % 128 == 0 && D <= 512 && (K2 == 1536 || K2 == 1280 || K2 == 768 || K2 == 384);
}
template <int K, int NCH, int NS, bool LNS = false>
bool launch_linear2_ws_t(Lin2Args a, int shared, hipStream_t st) {
    using C = Lin2Cfg<K, NCH, NS, true>;
    auto kern = k_linear2_ws<K, NCH, NS, true, LNS>;
    // grid = 8 x slices x rpx workgroups, at most one per CU; fewer token ranges than 32-token blocks
    const int slices = a.F / 128, cus = device_cus(), NBLK = (a.N + 31) / 32;
    int rpx = std::max(1, cus / (8 * slices));
    while (rpx > 1 && 8 * rpx > NBLK) --rpx;
    const int ranges = 8 * rpx, max_blocks = (NBLK + ranges - 1) / ranges + 1;
    const int gate_rows = shared ? 1 : (max_blocks * 32 + a.tpt - 1) / a.tpt + 1;  // trajectories one range can span
    if (gate_rows > C::max_gate_rows) return false;
    a.slices = slices;
    a.rpx = rpx;
    a.gate_rows = gate_rows;
    LSL_ALLOW_LDS(kern, (size_t)163840);
    hipLaunchKernelGGL(kern, dim3(8 * slices * rpx), dim3(512), C::lds_bytes(gate_rows), st, a);
    return true;
}
bool launch_linear2_ws(int K2, const Lin2Args &a, int shared, hipStream_t st) {
    if (a.stats) {  // LNS instances: per-wave row statistics of the updated rows beside h (the next sub-block's LayerNorm runs inside linear1)
        switch (K2) {
            case 1536: return launch_linear2_ws_t<1536, 3, 3, true>(a, shared, st);
            case 1280: return launch_linear2_ws_t<1280, 4, 4, true>(a, shared, st);
            case 768: return launch_linear2_ws_t<768, 3, 3, true>(a, shared, st);
            case 384: return launch_linear2_ws_t<384, 3, 3, true>(a, shared, st);
            default: return false;
        }
    }
    switch (K2) {
        case 1536: return launch_linear2_ws_t<1536, 3, 3>(a, shared, st);
        case 1280: return launch_linear2_ws_t<1280, 4, 4>(a, shared, st);  // (4 chunks of 160 columns: 40 of 64 lanes per LDS-DMA instruction instead of 32; round 5: 0.168-0.171 -> 0.165 ms at 163 840 tokens, 18.6 -> 17.1 us at 10 240)
        case 768: return launch_linear2_ws_t<768, 3, 3>(a, shared, st);
        case 384: return launch_linear2_ws_t<384, 3, 3>(a, shared, st);
        default: return false;
    }
}

// The back half of a sub-block on the row-owning tail kernel (k_tail.hip.h): up-projection -> GELU in registers -> down-projection + attention
// out-projection + gated residual + the next sub-block's LayerNorm.  Instances: hidden 256 with heads * head_dim_pad = 256 (the output tile
// of a wave is hidden / 2 accumulator registers: 512 does not fit two waves per SIMD, and the one-wave-per-SIMD form measured slower than
// what it replaces - profiles/r06_tail_experiments.txt).  Chosen per HANDLE (lsl_model_set_tail; LSL_TAIL=1 makes it the default of new
// handles, LSL_TAIL=0 disables it - both read in the product: A/B runs and the GPU suite compare the two decompositions), never per batch.
int tail_env() {
    static const int v = env_int("LSL_TAIL", -1);
    return v;
}
bool tail_shape_ok(int D, int HHD, int M) {
    return tail_env() != 0 && D == 256 && HHD == 256 && M % 64 == 0 && M >= 64 && TailCfg<256, 256>::lds_bytes(M) <= (size_t)163840;  // (an even number of 32-feature mlp blocks: the kernel's pipeline has no parity branches)
}
size_t tail_stream_bytes(const lsl_model *m) { return TailCfg<256, 256>::stream_bytes(m->d.mlp_dim); }
void launch_tail(const TailArgs &a, hipStream_t st) {
    using C = TailCfg<256, 256>;
    auto kern = k_tail<256, 256>;
    LSL_ALLOW_LDS(kern, (size_t)163840);
    const int nwt = (a.N + 31) / 32;  // wave tiles of 32 tokens: every workgroup gets at least one
    hipLaunchKernelGGL(kern, dim3(std::min(nwt, device_cus())), dim3(512), C::lds_bytes(a.M), st, a);
}

// GEMM tiling (tuning knob LSL_GEMM; every variant sums k in the same order, so results are identical).
//   (features x tokens, waves, BK x ring stages):
//   5  256x256  8 waves 64x2, one tile per workgroup
//   6  256x256  8 waves 32x3, persistent workgroups + next-tile prefetch during the epilogue
//   10 128x128  4 waves 32x3 (used when F is not a multiple of 256: D = 128 / 384 models)
//   7  256x256  8 waves 64x2, persistent, piece-form epilogue (4 KiB staging per wave, next piece prefetched): linear2 default
//   8  256x256  8 waves 32x3, persistent, piece-form epilogue
//   11 128x128  4 waves 64x2
//   12 256x256  8 waves 64x2, persistent, two-phase epilogue staged in ring slot 1 (needs an even number of k-tiles)
//   13 256x128  4 waves 32x2, persistent, two workgroups per CU
//   15 256x256 16 waves 64x2 (64x64 per wave, 4 waves/SIMD: the light linear2 epilogue fits the 128-VGPR budget and the
//      extra occupancy hides load / store latency)
// 20-22: ping-pong halves (k_gemm_pp.hip.h).  Default (-1): 12 for linear1 (5 when K < 512 or not a multiple of 128, 6 when not a multiple of 64), 7 for linear2: the fastest pair measured on MI355X (profiles/r01_gemm_variants.txt lists
// every variant that was tried, including the ones no longer compiled in).
template <class Epi>
int gemm_variant(int F, int K, int N = 1 << 30) {
    static const int forced_all = tune_int("LSL_GEMM", -1);
    static const int forced_1 = tune_int("LSL_GEMM1", -1), forced_2 = tune_int("LSL_GEMM2", -1);  // per GEMM: linear1 / linear2
    const int forced_one = std::is_same<Epi, EpiLinear2>::value ? forced_2 : forced_1;
    const int forced = forced_one >= 0 ? forced_one : forced_all;
    // 256-wide feature tiles waste MFMA work when F is not a multiple of 256 (D = 128 / 384 models): use 128 x 128 there
    const bool ragged = F % 256 != 0 && (F % 256 <= 128);
    const int ragged_variant = std::is_same<Epi, EpiLinear2>::value && K % 64 == 0 ? 11 : 10;  // measured on the D = 384 / 128 models
    if (forced >= 0) return forced;
    // linear2 of the 384-wide models (peptide: F = 384, K = 2 048, which no weight-stationary instance holds): the launch is bound by the
    // operand bytes each CU pulls through its L1 miss path (measured 49 GB/s per CU with two 128 x 128 workgroups per CU: the per-CU limit);
    // 192 x 128 tiles move 17 % fewer operand bytes per FLOP with as many workgroups as CUs, and a third ring slot covers the latency one
    // workgroup per CU leaves exposed: 63.5 -> 54.3 ms per 100 evaluations at 16 000 tokens (profiles/r05_experiments.txt).  Same bits as
    // every other tiling (same k order per element).  Only when the tiles fill at least half of the CUs.
    if (std::is_same<Epi, EpiLinear2>::value && F % 192 == 0 && F % 256 != 0 && K % 64 == 0 && (long)((N + 127) / 128) * (F / 192) * 2 >= (long)device_cus())
        return 28;
    if (ragged) return ragged_variant;
    // Small launches (one or two trajectories of the MD17 models, the reference's own B = 4 case): 256 x 256 tiles leave most of the chip
    // idle or run two rounds for 1.2 rounds of work; 128 x 128 tiles (two workgroups per CU) fill it.  Measured (profiles/
    // r02_experiments.txt): md17_bench B = 1 49.1 -> 38.0 ms per call, B = 2 63.0 -> 59.3, md17_ref B = 4 8.00 -> 7.65; from B = 4 of
    // md17_bench on the large tiles win again.  The tile shape does not change any output bit (every element is the same k-ascending
    // chain of 16-deep MFMA steps and the same epilogue arithmetic - checked by the batch-32-vs-batch-1 test at the headline shape),
    // so this may depend on the launch size.
    static const int small_rule = tune_int("LSL_SMALL_TILES", 1);
    const long tiles256 = (long)((N + 255) / 256) * ((F + 255) / 256);
    const int cus = device_cus();
    if (small_rule && K % 64 == 0 && tiles256 * (std::is_same<Epi, EpiLinear2>::value ? 2 : 4) <= (long)cus * (std::is_same<Epi, EpiLinear2>::value ? 1 : 5))
        return 11;  // linear2: tiles <= CUs / 2; linear1: tiles <= 1.25 CUs
    return std::is_same<Epi, EpiLinear2>::value ? (K % 128 == 0 ? 7 : 15) : (K % 128 == 0 ? 12 : 5);
}

// linear2 can also write the next sub-block's LayerNorm + modulate (EpiLinear2::finish_rows) when it runs as the persistent
// row-owner kernel over whole rows of D = 256 or 512 features and the launch has at least as many token tiles as workgroups
bool linear2_can_fuse_ln(int D, int N, int K2) {
    static const int off = tune_int("LSL_LN_FUSE", 0) == 0;  // measured and rejected (k_gemm.hip.h: EpiLinear2): experiments builds only
    if (off || D % 256 != 0 || gemm_variant<EpiLinear2>(D, K2) != 7 || K2 % 128 != 0) return false;
    return gemm_rows_walk<256, 256, 2, 4, 64, 2, EpiPieces<EpiLinear2>>(D, N);
}

template <class Epi>
void launch_gemm(const u16 *W, const u16 *X, int F, int N, int K, const Epi &epi_in, hipStream_t st, int hhd = 32, bool rows = false) {
    const int variant = gemm_variant<Epi>(F, K, N);
    static const int probe = tune_int("LSL_PROBE", 0);
    static const int stagger = tune_int("LSL_STAGGER", 0);
    GemmArgs g{W, X, F, N, K, rows ? 1 : 0, stagger, probe};
    // LSL_NT bit 0: linear1 output, bit 1: linear2 residual update, bit 2: attention output, bit 3: LayerNorm+modulate output
    static const int nt = tune_int("LSL_NT", 3);
    Epi epi = epi_in;
    epi.probe = probe | ((nt >> (std::is_same<Epi, EpiLinear2>::value ? 1 : 0)) & 1 ? 32 : 0);
    const bool pp_ok = !std::is_same<Epi, EpiLinear2>::value ? hhd % 32 == 0 : true;  // linear1 sections start on 32-feature tiles
    if (launch_gemm_experiment(variant, g, epi, st, pp_ok)) return;  // (rejected structures: -DLSL_EXPERIMENTS builds only)
    if (variant == 12 && K % 128 == 0) return launch_gemm_glds<256, 256, 2, 4, 64, 2, true>(g, epi, st);  // 5 made persistent (staging in ring slot 1)
    if constexpr (lsl_experiments || std::is_same<Epi, EpiLinear2>::value) {  // (linear1's piece epilogue exists in the experiments build only)
        if (variant == 7 && F % 32 == 0 && pp_ok && K % 128 == 0) return launch_gemm_glds<256, 256, 2, 4, 64, 2, true>(g, EpiPieces<Epi>(epi), st);  // persistent, 64-deep k-tiles, piece epilogue
    }
    switch (variant) {
        case 5: return launch_gemm_glds<256, 256, 2, 4, 64, 2, false>(g, epi, st);
        case 10: return launch_gemm_glds<128, 128, 2, 2, 32, 3, false>(g, epi, st);
        case 11: return launch_gemm_glds<128, 128, 2, 2, 64, 2, false>(g, epi, st);
        case 28:  // 192 features x 128 tokens, 8 waves of 96 x 32, three 64-deep ring slots (linear2 of the 384-wide models)
            if constexpr (std::is_same<Epi, EpiLinear2>::value) return launch_gemm_glds<192, 128, 2, 4, 64, 3, false>(g, epi, st);
            else break;
        case 15: return launch_gemm_glds<256, 256, 4, 4, 64, 2, false>(g, epi, st);
        default: return launch_gemm_glds<256, 256, 2, 4, 64, 2, false>(g, epi, st);  // (K is a multiple of 64: hidden sizes are)
    }
}

template <int HDP, int NW, int ITEMS, int NKT>
void launch_attention_rows(const AttnArgs &a, hipStream_t st) {
    auto kern = k_attention_rows<HDP, NW, ITEMS, NKT>;
    const size_t lds = (size_t)ITEMS * 2 * (NKT > 0 ? NKT * 32 : (a.S + 31) & ~31) * HDP * 2 + NW * sizeof(float);  // K, V, key-norm slots
    LSL_ALLOW_LDS(kern, NKT > 0 ? lds : (size_t)160 * 1024);
    const long items = (long)a.n_seq * a.H;
    hipLaunchKernelGGL(kern, dim3((unsigned)((items + ITEMS - 1) / ITEMS)), dim3(NW * 64), lds, st, a);
}

// persistent, double-buffered form (k_attention_stream): axes of more than 128 positions (unit = (sequence, head, group of 256 queries), keys in
// chunks of 256: peptide's T = 1000 is 4 groups x 4 chunks) and of 9 .. 32 positions
// with a multiple of 8 heads (8 heads of a sequence per unit).  The choice depends on the model and on T, L only - never on the batch - so a
// trajectory's bits are the same in any batch.  LSL_ATTN_STREAM=0 (read in the product too: A/B runs) keeps k_attention_rows.
int attention_stream_mode(int S, int H) {  // 0: k_attention_rows / tiny / online, 1: stream SHORT, 2: stream LONG
    static const int on = env_int("LSL_ATTN_STREAM", 1);
    if (!on) return 0;
    static const int long_min = tune_int("LSL_ATTN_LONG_MIN", 129);  // shortest axis on the LONG form
    if (S >= long_min) return 2;  // (round 5: any length - keys in chunks of 256 through the two images, queries in groups of 8 tiles)
    if (S > 8 && S <= 32 && H % 8 == 0) return 1;
    return 0;
}
// q / k / v as head-major planes (k_lin1.hip.h, Lin1Args::planes): spatial sub-blocks (positions = consecutive tokens) whose attention
// runs the LONG stream kernel, token-stationary linear1.  LSL_QKV_PLANES=0 keeps token-major rows (A/B runs).
bool qkv_planes_ok(int hdp, int hidden, int heads, int S, bool temporal, bool lin1_ts) {
    static const int on = env_int("LSL_QKV_PLANES", 1);
    (void)hidden;
    return on && !temporal && lin1_ts && heads % (64 / hdp) == 0 && S <= 256 && attention_stream_mode(S, heads) == 2;
}
// tiny SPATIAL axes (L = 2, 4, 8: positions and sequences are consecutive tokens) on the SHORT stream kernel, 32 / L sequences to a tile with
// the scores outside the block diagonal masked (AttnArgs::blk): replaces k_attention_tiny.  LSL_ATTN_GROUP=0 keeps the lane-per-query kernel.
bool attention_grouped_ok(const AttnArgs &a) {
    static const int on = env_int("LSL_ATTN_GROUP", 1), stream_on = env_int("LSL_ATTN_STREAM", 1);
    return on && stream_on && a.S >= 2 && a.S <= 8 && (a.S & (a.S - 1)) == 0 && a.inner == 1 && a.pos_stride == 1 && a.outer_stride == a.S && a.H % 8 == 0 &&
           a.kmax2 != nullptr;
}
template <int HDP>
bool launch_attention_stream(const AttnArgs &a_in, hipStream_t st) {
    AttnArgs a = a_in;
    a.blk = 0;
    a.n_tok = 0;
    if (attention_grouped_ok(a)) {  // present the tokens as sequences of 32 rows
        a.blk = a.S;
        a.n_tok = a.n_seq * a.S;
        a.n_seq = (a.n_tok + 31) / 32;
        a.S = 32;
        a.outer_stride = 32;
    }
    const int mode = attention_stream_mode(a.S, a.H);
    const bool is_long = mode == 2;
    if (!mode || !a.kmax2) return false;
    const size_t lds = (size_t)2 * 2 * 256 * HDP * 2 + (size_t)8 * 32 * HDP * 2;  // two images of K | V, 256 rows each; a 32-row query image per wave
    const long n_units = is_long ? (long)a.n_seq * a.H * (((a.S + 31) / 32 + 7) / 8) : (long)a.n_seq * (a.H / 8);  // LONG: (sequence, head, group of 8 query tiles)
    if (n_units + 2L * device_cus() >= (1L << 31)) return false;  // (the kernel counts units in 32 bits)
    const int grid = (int)std::min<long>(2L * device_cus(), n_units);  // two workgroups per CU (2 x 80 KiB of LDS at 32-wide heads)
    // plain stores: behind streaming stores the in-order vector-memory queue reports the next unit's requests late (measured: 0.78 vs 0.27 ms)
    AttnArgs b = a;
    b.nt = 0;
    auto go2 = [&](auto kern) {
        LSL_ALLOW_LDS(kern, lds);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, b);
    };
    if (!is_long && a.blk > 0) go2(k_attention_stream<HDP, false, false, false, true>);
    else if (!is_long) go2(k_attention_stream<HDP, false>);
    else if (a.S <= 256) go2(k_attention_stream<HDP, true>);
    else if constexpr (HDP == 32) {  // keys in chunks of 256, queries in groups of 8 tiles
        if (a.hd == 24) go2(k_attention_stream<HDP, true, true, true>);  // (peptide: the padded head's spare V column carries the softmax denominator)
        else go2(k_attention_stream<HDP, true, true>);
    } else go2(k_attention_stream<HDP, true, true>);
    return true;
}

template <int HDP>
void launch_attention_linear_t(const AttnArgs &a, hipStream_t st) {
    const long units = (long)a.n_seq * a.H;
    hipLaunchKernelGGL((k_attention_linear<HDP>), dim3((unsigned)std::min<long>(units, 8L * device_cus())), dim3(256), 0, st, a);
}

template <int HDP>
void launch_attention_t(const AttnArgs &a, hipStream_t st) {
    if (launch_attention_stream<HDP>(a, st)) return;
    const int Sp = (a.S + 31) & ~31;
    static const int online = tune_int("LSL_ATTN_ONLINE", 0);  // 1: force the online-softmax kernel (A/B measurements)
    if (!online && a.S <= 8) {  // one lane per (query, head), no MFMA padding
        const long lanes = (long)a.n_seq * a.S * a.H;
        hipLaunchKernelGGL((k_attention_tiny<HDP>), dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, st, a);
        return;
    }
    if (!online && (size_t)2 * Sp * HDP * 2 + 64 <= (size_t)160 * 1024) {  // two-pass softmax, K/V of one (sequence, head) in LDS
        // long axes (peptide T = 1000): 16 waves - with the max pass gone (AttnArgs::bound) the kernel is a chain of MFMA -> exp2 -> MFMA per
        // tile, and four waves per SIMD hide it better than two (attention 320.6 -> 303.8 ms per 1000-step call; with the max pass
        // 8 waves were as fast, profiles/r02_experiments.txt)
        static const int nw16 = tune_int("LSL_ATTN_NW16", 1);
        if (Sp > 256 && nw16) return launch_attention_rows<HDP, 16, 1, 0>(a, st);
        if (Sp > 256) return launch_attention_rows<HDP, 8, 1, 0>(a, st);
        if (Sp <= 32) return launch_attention_rows<HDP, 4, 4, 1>(a, st);
        if (Sp <= 64) return launch_attention_rows<HDP, 4, 2, 2>(a, st);
        if (Sp <= 128) return launch_attention_rows<HDP, 4, 1, 4>(a, st);
        if (Sp <= 192) return launch_attention_rows<HDP, 4, 1, 6>(a, st);
        return launch_attention_rows<HDP, 4, 1, 8>(a, st);
    }
    const long items = (long)a.n_seq * a.H;
    const size_t per_item = (size_t)2 * Sp * HDP * 2;
    if (Sp <= 32) {
        auto kern = k_attention<HDP, 4, 4>;
        hipLaunchKernelGGL(kern, dim3((unsigned)((items + 3) / 4)), dim3(256), 4 * per_item, st, a);
    } else if (Sp <= 64) {
        auto kern = k_attention<HDP, 4, 2>;
        hipLaunchKernelGGL(kern, dim3((unsigned)((items + 1) / 2)), dim3(256), 2 * per_item, st, a);
    } else if (Sp <= 512) {
        auto kern = k_attention<HDP, 4, 1>;
        LSL_ALLOW_LDS(kern, 65536);
        hipLaunchKernelGGL(kern, dim3((unsigned)items), dim3(256), per_item, st, a);
    } else {
        auto kern = k_attention<HDP, 8, 1>;
        LSL_ALLOW_LDS(kern, 160 * 1024);
        hipLaunchKernelGGL(kern, dim3((unsigned)items), dim3(512), per_item, st, a);
    }
}

template <bool PRE, bool POST>
void launch_dense(float *out, const float *in, const float *W, const float *bias, const float *add, int rows, int I, int O,
                  int add_stride, hipStream_t st, bool single = false, int add_mod = 0) {
    // The choice must not depend on the BATCH: the two kernels sum k in different orders, and a trajectory's result has to be the
    // same bits whatever batch it is sampled in (K-sample batching, sharding, pass size).  `single` marks the calls that have one
    // row by construction (the sampler's shared time without class conditioning: one conditioning vector for any batch); they
    // take the wave-per-output kernel (a coalesced GEMV, 8x faster at one row than the 64-row tile kernel).
    if (single && rows == 1 && I <= 512)
        hipLaunchKernelGGL((k_dense_rows<PRE, POST>), dim3((O + 3) / 4), dim3(256), 0, st, out, in, W, bias, add, rows, I, O, add_stride, add_mod);
    else if ((I == 128 || I == 256) && tune_int("LSL_DENSE_MFMA", 1)) {  // many rows, usual widths: the fp32 matrix pipe (k_dense_mfma)
        const dim3 grid((O + 31) / 32, (rows + 31) / 32);
        if (I == 128) hipLaunchKernelGGL((k_dense_mfma<PRE, POST, 32>), grid, dim3(256), 0, st, out, in, W, bias, add, rows, I, O, add_stride, add_mod);
        else hipLaunchKernelGGL((k_dense_mfma<PRE, POST, 64>), grid, dim3(256), 0, st, out, in, W, bias, add, rows, I, O, add_stride, add_mod);
    } else if (I % 4 == 0)
        hipLaunchKernelGGL((k_dense_tiled<PRE, POST>), dim3((O + 63) / 64, (rows + 63) / 64), dim3(256), 0, st, out, in, W, bias, add, rows,
                           I, O, add_stride, add_mod);
    else
        hipLaunchKernelGGL((k_dense_rows<PRE, POST>), dim3((O + 3) / 4), dim3(256), 0, st, out, in, W, bias, add, rows, I, O, add_stride, add_mod);
}

// the short GEMM chains in front of the trajectory-resident kernel (k_dense_mfma: latency-optimised; one kernel for any row count)
template <bool PRE, bool POST>
void launch_dense_small(float *out, const float *in, const float *W, const float *bias, const float *add, int rows, int I, int O,
                        int add_stride, hipStream_t st, int add_mod = 0) {
    const dim3 grid((O + 31) / 32, (rows + 31) / 32);
    if (I == 128)
        hipLaunchKernelGGL((k_dense_mfma<PRE, POST, 32>), grid, dim3(256), 0, st, out, in, W, bias, add, rows, I, O, add_stride, add_mod);
    else if (I == 256)
        hipLaunchKernelGGL((k_dense_mfma<PRE, POST, 64>), grid, dim3(256), 0, st, out, in, W, bias, add, rows, I, O, add_stride, add_mod);
    else
        launch_dense<PRE, POST>(out, in, W, bias, add, rows, I, O, add_stride, st, false, add_mod);
}

