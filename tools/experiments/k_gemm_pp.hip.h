// Ping-pong form of the bf16 MFMA GEMM of k_gemm.hip.h (same product, same operand layouts, same k order, same epilogue
// arithmetic -> identical results), built to overlap the three phases that the one-tile-at-a-time kernels run back to back
// (measured on MI355X, profiles/r01_gemm_variants.txt: MFMA loop 171 ms + epilogue 177 ms + barriers 34 ms = the 385 ms of
// linear1; the phases do not overlap because all 8 waves of a workgroup are in the same phase at any time).
//
// One 512-thread workgroup per CU = two HALVES of 4 waves (waves 0-3 / 4-7: one wave of each half on every SIMD).  A half
// owns a 128-feature x 256-token tile (4 waves side by side along the tokens, 128 x 64 per wave = 8 MFMA tiles, 128
// accumulator VGPRs) and alternates between two roles, always opposite to the other half:
//   MAIN      nk intervals of  [wait own LDS-DMA | s_barrier | issue k-tile kt+NS-1 | BK/16 x (6 ds_read_b128, 8 MFMA)], the
//             fragments read one k16 sub-step ahead of their MFMAs.  An s_barrier round trip costs ~300 cycles on MI355X
//             (measured: an empty barrier loop runs at 140-155 ns per iteration), so BK = 64 (32 MFMAs per wave per barrier)
//   EPILOGUE  the 16 pieces of Epi::piece<C> (norm / RoPE / GELU / stores of the tile it has just accumulated) spread over
//             the first E = nk-(NS-1) intervals, one s_barrier per interval; in the last NS-1 intervals it only issues the
//             first k-tiles of ITS next tile into the ring slots the other half is vacating
// so on every SIMD one wave feeds the matrix pipe while its partner runs VALU / LDS / global stores, and the operand ring
// (NS slots of [128 W rows | 256 X rows] x BK) carries one continuous stream of k-tiles, alternately owned by the halves.
// s_barrier is workgroup-wide, which is what keeps the halves exactly one role apart: both execute nk barriers per phase.
//
// vmcnt discipline: a wave's LDS-DMA loads are only waited on by the wave itself (counted s_waitcnt), visibility to the other 3
// waves of the half comes from the barrier.  The epilogue pieces (whose stores and table loads share the vmcnt queue) all
// come BEFORE the first prefetch of the next tile in program order, so at the start of a MAIN phase the queue ends with
// exactly the NS-1 prefetched k-tiles and the usual counted wait applies.
#pragma once
#include "k_gemm.hip.h"  // lam_slide_amd/csrc (the including translation unit's directory)

template <int BK, int NS, class Epi>
struct GemmPPCfg {
    static constexpr int BF = 128, BT = 256;
    static constexpr size_t stage_bytes = (size_t)(BF + BT) * BK * 2;                    // one k-tile of both operands
    static constexpr size_t out_bytes = (size_t)4 * Epi::pp_stage_bytes;  // the half in the epilogue role
    static constexpr size_t lds_bytes = NS * stage_bytes + out_bytes;
    static_assert(lds_bytes <= 163840, "LDS budget (160 KiB per workgroup)");
};

template <int BK, int NS, int NB, class Epi>  // NB: upper bound of the barriers between two pieces, E <= 16 NB
__global__ void __launch_bounds__(512, 2) k_gemm_pp(GemmArgs g, Epi epi) {
    using Cfg = GemmPPCfg<BK, NS, Epi>;
    constexpr int BF = Cfg::BF, BT = Cfg::BT, MI = 4, NJ = 2;
    constexpr int ROWB = BK * 2, STAGE = (int)Cfg::stage_bytes;
    constexpr int CPR = ROWB / 16, RPP = 1024 / ROWB;               // 16-byte chunks per row, rows per 1 KiB LDS-DMA piece
    constexpr int WP = BF / RPP / 4, XP = BT / RPP / 4, LPS = WP + XP;  // pieces per wave per k-tile
    constexpr int KSUB = BK / 16;
    static_assert(Epi::pieces == 16 && NS >= 2 && NS <= 5 && (BK == 32 || BK == 64), "ping-pong schedule");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave >> 2, w4 = wave & 3;
    const int r = lane & 31, hf = lane >> 5;
    const int ntt = (g.N + BT - 1) / BT, nft = (g.F + BF - 1) / BF, ntiles = ntt * nft;
    const int nk = g.K / BK, E = nk - (NS - 1);  // host guarantees nk >= NS
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;

    // LDS-DMA sources of this wave: pieces w4, w4+4, ... (RPP rows each) of the W rows and of the X rows of its half's tile
    const int lrow = w4 * RPP + lane / CPR;
    const int lchunk = (swz_bk<BK>(lrow, lane % CPR) - lrow * ROWB) >> 4;
    const size_t piece_step = (size_t)4 * RPP * g.K;
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
                                             (LDS_PTR(void))(dst + (w4 + 4 * i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < XP; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(srcX + i * piece_step + kt * BK),
                                             (LDS_PTR(void))(dst + (BF / RPP + w4 + 4 * i) * 1024), 16, 0, 0);
    };
    int offA[KSUB], offB[KSUB];
#pragma unroll
    for (int ks = 0; ks < KSUB; ++ks) {
        offA[ks] = swz_bk<BK>(r, 2 * ks + hf);
        offB[ks] = swz_bk<BK>(r, 2 * ks + hf) + (BF + w4 * 64) * ROWB;
    }
    char *stage = smem + NS * STAGE + (size_t)w4 * Epi::pp_stage_bytes;

    f32x16 acc[MI][NJ];
    int fe = 0, ne = 0;  // wave tile whose sums `acc` holds
    int base = 0;        // ring slot of k-tile 0 of the tile in its MAIN phase
    auto read_frags = [&](const char *sb, int ks, bf16x8(&a)[MI], bf16x8(&b)[NJ]) {
#pragma unroll
        for (int i = 0; i < MI; ++i) a[i] = as_bf16x8(*reinterpret_cast<const u32x4 *>(sb + offA[ks] + i * 32 * ROWB));
#pragma unroll
        for (int j = 0; j < NJ; ++j) b[j] = as_bf16x8(*reinterpret_cast<const u32x4 *>(sb + offB[ks] + j * 32 * ROWB));
    };
    auto mfma_all = [&](const bf16x8(&a)[MI], const bf16x8(&b)[NJ]) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
    };
    if (half == 0 && my_tiles > 0) {
        set_tile(0);
#pragma unroll
        for (int s = 0; s < NS - 1; ++s) issue(s, s);
    }

    for (int p = 0; p <= my_tiles; ++p) {  // phase p: MAIN of tile p (half p & 1), EPILOGUE of tile p-1 + prefetch of tile p+1 (other half)
        if ((p & 1) == half) {
            const bool active = p < my_tiles;
            if (g.stagger & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
            for (int kt = 0; kt < nk; ++kt) {
                if (active) {
                    switch (min(nk - 1 - kt, NS - 2)) {  // own k-tiles issued after k-tile kt
                        case 0: wait_vmcnt<0>(); break;
                        case 1: wait_vmcnt<LPS>(); break;
                        case 2: wait_vmcnt<2 * LPS>(); break;
                        default: wait_vmcnt<3 * LPS>(); break;
                    }
                }
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (!active) continue;
                if (kt + NS - 1 < nk) issue(kt + NS - 1, (base + kt + NS - 1) % NS);
                if (LSL_PROBE(g.probe, 2)) continue;
                const char *sb = smem + ((base + kt) % NS) * STAGE;
                // fragments of sub-step ks+1 are read before the MFMAs of sub-step ks: a role has ONE wave per SIMD
                bf16x8 a0[MI], b0[NJ], a1[MI], b1[NJ];
                read_frags(sb, 0, a0, b0);
#pragma unroll
                for (int ks = 0; ks < KSUB; ks += 2) {
                    read_frags(sb, ks + 1, a1, b1);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_all(a0, b0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (ks + 2 < KSUB) read_frags(sb, ks + 2, a0, b0);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_all(a1, b1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            fe = f_base;
            ne = n_base + w4 * 64;
            if (g.stagger & 1) __builtin_amdgcn_s_setprio(0);
        } else {
            const bool has_prev = p >= 1 && !(LSL_PROBE(g.probe, 4)), has_next = p + 1 < my_tiles;
            if (has_next) set_tile(p + 1);
            const int nbase = (base + nk) % NS;  // ring slot of the next tile's k-tile 0
            int lane_e = lane;                   // opaque copy: keeps the pieces' per-lane address arithmetic inside this branch
            asm volatile("" : "+v"(lane_e));     // (hoisted out of the phase loop it would sit in VGPRs through the MAIN role)
            int done = 0;                        // barriers of this phase executed so far
            // run this phase's barriers up to and including number `target` (< E).  Straight-line, not a loop: in front of a loop
            // hipcc drains vmcnt (its loop-preheader flush), which would wait for the piece's own store and prefetches.
            auto bars_to = [&](int target) {
#pragma unroll
                for (int u = 0; u < NB; ++u)
                    if (done <= target) {
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                        ++done;
                    }
            };
            if (has_prev) {  // one branch around ALL pieces: a join after each piece would copy the prefetched registers (and wait)
                typename Epi::Pipe kc;  // what the next piece needs from memory, requested one piece ahead
                epi.template fetch<0>(acc, stage, fe, ne, lane_e, g.F, g.N, kc);
#define LSL_PIECE(C)                                                                     \
    bars_to(((C) * E) >> 4);                                                             \
    epi.template piece<C>(acc, stage, fe, ne, lane_e, g.F, g.N, kc);
                LSL_PIECE(0) LSL_PIECE(1) LSL_PIECE(2) LSL_PIECE(3) LSL_PIECE(4) LSL_PIECE(5) LSL_PIECE(6) LSL_PIECE(7)
                LSL_PIECE(8) LSL_PIECE(9) LSL_PIECE(10) LSL_PIECE(11) LSL_PIECE(12) LSL_PIECE(13) LSL_PIECE(14) LSL_PIECE(15)
            }
#undef LSL_PIECE
            asm volatile("" ::: "memory");  // every store of the pieces stays ahead of the prefetches in program order
            while (done < E) {
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                ++done;
            }
            for (int i = 0; i < NS - 1; ++i) {  // intervals E .. nk-1: the other half vacates one ring slot per interval
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (has_next) issue(i, (nbase + i) % NS);
            }
        }
        base = (base + nk) % NS;
    }
}
