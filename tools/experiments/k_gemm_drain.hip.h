// "Drain" form of the bf16 MFMA GEMM (same product, operand layouts, k order and epilogue arithmetic as k_gemm.hip.h -> identical
// results): the epilogue of tile i runs INSIDE the main loop of tile i+1, so the two phases that add up in the one-tile-at-a-time
// kernels (MFMA loop + operand waits 214 ms, epilogue 159 ms per step for linear1) share the same wall time.
//
// What the ping-pong experiment (k_gemm_pp.hip.h) taught: a role that runs on ONE wave per SIMD is latency- and issue-bound.  Here
// every wave does both jobs and the two waves of a SIMD stay symmetric: a 256-feature x 128-token tile, 8 waves of 64 x 64
// (64 accumulator VGPRs), and a second register set `prev` that holds the finished sums of the previous tile.  Each k-tile
// interval consists of the MFMA block of the current tile (16 MFMAs per wave at BK = 64) and, in 8 of the intervals, one
// epilogue piece of the previous tile (Epi::piece<C>: a 32 x 16 slice, software-prefetched).  The only asymmetry is the ORDER
// inside an interval: waves 0-3 run [MFMA block, piece], waves 4-7 (their SIMD partners) run [piece, MFMA block], so that on every
// SIMD one wave feeds the matrix pipe while the other issues VALU / LDS / stores - without it both waves want the pipe at the same
// time and then both do vector work with the pipe idle.  One s_barrier per k-tile, as in the plain kernel.
//
// Shapes: full tiles only (F % 256 == 0, N % 128 == 0: the epilogue stores are unconditional, which keeps the number of vector
// memory operations per piece fixed); K % 64 == 0.  Other shapes use the plain kernels.
#pragma once
#include "k_gemm.hip.h"  // lam_slide_amd/csrc (the including translation unit's directory)

template <class Epi>
struct GemmDrainCfg {
    static constexpr int BF = 256, BT = 128, BK = 64, NS = 2;
    static constexpr size_t stage_bytes = (size_t)(BF + BT) * BK * 2;
    static constexpr size_t lds_bytes = NS * stage_bytes + (size_t)8 * Epi::pp_stage_bytes;
    static_assert(lds_bytes <= 163840, "LDS budget (160 KiB per workgroup)");
};

template <class Epi>
__global__ void __launch_bounds__(512, 2) k_gemm_drain(GemmArgs g, Epi epi) {
    using Cfg = GemmDrainCfg<Epi>;
    constexpr int BF = Cfg::BF, BT = Cfg::BT, BK = Cfg::BK, NS = Cfg::NS, MI = 2, NJ = 2, NP = 4 * MI;  // NP pieces per wave tile
    constexpr int ROWB = BK * 2, STAGE = (int)Cfg::stage_bytes, RPP = 1024 / ROWB, CPR = ROWB / 16;
    constexpr int WP = BF / RPP / 8, XP = BT / RPP / 8, KSUB = BK / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave >> 2;  // waves w and w + 4 share a SIMD
    const int wf = (wave & 3), wt = half;  // 4 x 2 waves over the 256 x 128 tile: partners work on the two token halves
    const int r = lane & 31, hf = lane >> 5;
    const int ntt = g.N / BT, nft = g.F / BF, ntiles = ntt * nft;
    const int nk = g.K / BK;
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;

    const int lrow = wave * RPP + lane / CPR;
    const int lchunk = (swz_bk<BK>(lrow, lane % CPR) - lrow * ROWB) >> 4;
    const size_t piece_step = (size_t)8 * RPP * g.K;
    const u16 *srcW = nullptr, *srcX = nullptr;
    int f_base = 0, n_base = 0;
    auto set_tile = [&](int j) {
        const int tile = xcd_remap((int)blockIdx.x + j * (int)gridDim.x, ntiles);
        f_base = (tile % nft) * BF;
        n_base = (tile / nft) * BT;
        srcW = g.W + (size_t)(f_base + lrow) * g.K + lchunk * 8;
        srcX = g.X + (size_t)(n_base + lrow) * g.K + lchunk * 8;
    };
    auto issue = [&](int kt, int slot) {
        if (LSL_PROBE(g.probe, 1)) return;
        char *dst = smem + slot * STAGE;
#pragma unroll
        for (int i = 0; i < WP; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(srcW + i * piece_step + kt * BK),
                                             (LDS_PTR(void))(dst + (wave + 8 * i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < XP; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(srcX + i * piece_step + kt * BK),
                                             (LDS_PTR(void))(dst + (BF / RPP + wave + 8 * i) * 1024), 16, 0, 0);
    };
    int offA[KSUB], offB[KSUB];
#pragma unroll
    for (int ks = 0; ks < KSUB; ++ks) {
        offA[ks] = swz_bk<BK>(r, 2 * ks + hf) + wf * 64 * ROWB;
        offB[ks] = swz_bk<BK>(r, 2 * ks + hf) + (BF + wt * 64) * ROWB;
    }
    char *stage = smem + NS * STAGE + (size_t)wave * Epi::pp_stage_bytes;

    f32x16 cur[MI][NJ], prev[MI][NJ];
    int fe = 0, ne = 0;  // wave tile whose finished sums `prev` holds
    int gk = 0;          // k-tiles consumed so far (ring slot = gk % NS)
    int pend = 0;  // vector-memory instructions this wave has issued since its last LDS-DMA loads (a lower bound)
    auto wait_dma = [&]() {  // the LDS-DMA loads of the k-tile about to be used have landed; the piece's own traffic may still fly
        switch (pend) {
            case 6: wait_vmcnt<6>(); break;
            case 5: wait_vmcnt<5>(); break;
            case 3: wait_vmcnt<3>(); break;
            case 2: wait_vmcnt<2>(); break;
            default: wait_vmcnt<0>(); break;
        }
    };
    auto mfma_block = [&](int slot) {
        if (LSL_PROBE(g.probe, 2)) return;
        const char *sb = smem + slot * STAGE;
        auto rd = [&](int ks, bf16x8(&a)[MI], bf16x8(&b)[NJ]) {
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = as_bf16x8(*reinterpret_cast<const u32x4 *>(sb + offA[ks] + i * 32 * ROWB));
#pragma unroll
            for (int j = 0; j < NJ; ++j) b[j] = as_bf16x8(*reinterpret_cast<const u32x4 *>(sb + offB[ks] + j * 32 * ROWB));
        };
        auto mm = [&](const bf16x8(&a)[MI], const bf16x8(&b)[NJ]) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) cur[i][j] = mfma32(a[i], b[j], cur[i][j]);
        };
        bf16x8 a0[MI], b0[NJ], a1[MI], b1[NJ];  // fragments of sub-step ks+1 are read before the MFMAs of sub-step ks
        rd(0, a0, b0);
#pragma unroll
        for (int ks = 0; ks < KSUB; ks += 2) {
            rd(ks + 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            mm(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 2 < KSUB) rd(ks + 2, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            mm(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if (my_tiles > 0) {
        set_tile(0);
        issue(0, 0);
    }
    for (int s = 0; s <= my_tiles; ++s) {  // step s: MAIN loop of tile s (if any) + epilogue of tile s-1 (if any)
        const bool main = s < my_tiles, have_prev = s > 0 && !(LSL_PROBE(g.probe, 4));
        if (main) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) cur[i][j][e] = 0.0f;
        }
        typename Epi::Pipe kc;
        if (have_prev) epi.template fetch<0>(prev, stage, fe, ne, lane, g.F, g.N, kc);
        int kt = 0;
        const int tile_f = f_base, tile_n = n_base;  // (set_tile for the NEXT tile happens inside the last interval)
        // one interval: publish k-tile kt, request the next one, then the MFMA block and (in NP of the intervals) one piece, in the
        // order of this wave's half
#define LSL_INTERVAL(PIECE_STMT, OPS)                                                                         \
    {                                                                                                         \
        wait_dma();                                                                                           \
        __builtin_amdgcn_s_barrier();                                                                         \
        asm volatile("" ::: "memory");                                                                        \
        if (kt + 1 < nk) issue(kt + 1, (gk + 1) % NS);                                                        \
        else if (s + 1 < my_tiles) {                                                                          \
            set_tile(s + 1);                                                                                  \
            issue(0, (gk + 1) % NS);                                                                          \
        }                                                                                                     \
        _Pragma("nounroll") for (int ph = 0; ph < 2; ++ph) {                                                  \
            if (ph == half) mfma_block(gk % NS);                                                              \
            else { PIECE_STMT }                                                                               \
        }                                                                                                     \
        pend = (OPS);                                                                                         \
        ++kt;                                                                                                 \
        ++gk;                                                                                                 \
    }
#define LSL_STEP(C)                                                                                           \
    if (main) {                                                                                               \
        const int hi = ((C + 1) * nk) / NP;                                                                   \
        if (kt < hi) {                                                                                        \
            while (kt + 1 < hi) LSL_INTERVAL(;, 0)                                                            \
            LSL_INTERVAL(if (have_prev) epi.template piece<C>(prev, stage, fe, ne, lane, g.F, g.N, kc);,      \
                         have_prev ? Epi::piece_ops(C == NP - 1) : 0)                                         \
        } else if (have_prev) epi.template piece<C>(prev, stage, fe, ne, lane, g.F, g.N, kc);                 \
    } else if (have_prev) epi.template piece<C>(prev, stage, fe, ne, lane, g.F, g.N, kc);
        LSL_STEP(0) LSL_STEP(1) LSL_STEP(2) LSL_STEP(3) LSL_STEP(4) LSL_STEP(5) LSL_STEP(6) LSL_STEP(7)
#undef LSL_STEP
#undef LSL_INTERVAL
        if (main) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) prev[i][j] = cur[i][j];
            fe = tile_f + wf * 64;
            ne = tile_n + wt * 64;
        }
    }
}
