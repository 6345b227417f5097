// linear2 of a ParallelMLPAttentionV2 block + the gated residual update (mmdit.py:248, latent_si_v31.py:53,60) as a WEIGHT-STATIONARY,
// token-streaming bf16 MFMA kernel: round-4 form.  Same contract and the same bits as k_gemm_glds<..., EpiPieces<EpiLinear2>> (k_gemm.hip.h).
//
// Why: in the 256 x 256-tile kernel a workgroup's main loop (L2 -> LDS feed of both operands) and its epilogue (4 KB per token of fp32
// residual read-modify-write, HBM-bound by itself) run one after the other: 0.30 + 0.19 ms per 245 760-token launch, and no tile-level
// variant changed that (profiles/r03_experiments.txt).  The token-stationary form of linear1 does not carry over: K = D + M = 1 536 bf16
// per token is 384 VGPRs for 32 tokens, and re-filling 384 KB of stationary tokens per workgroup and tile comes in bursts that nothing hides.
//
// Structure:
//   * the WEIGHTS are the stationary operand, for the whole lifetime of a workgroup: a workgroup owns 128 of the F output features and a
//     contiguous range of the tokens; its 8 waves are 4 PAIRS, pair p owns features 32 p .. 32 p + 31 of the slice.  The two waves of a pair
//     split K: the "lo" wave keeps W[32 rows][0, K/2) as the MFMA A fragments of its K / 32 k-steps (K / 8 VGPRs: 192 at K = 1 536), the
//     "hi" wave keeps W[32 rows][K/2, K).  Loaded once (k_lin2_pack writes them in fragment order: every load is 1 KiB contiguous);
//   * the TOKENS stream through LDS in blocks of 32: every wave multiplies the same token block with its own weights, so a block is read
//     from L2 once per workgroup and from LDS by 8 waves;
//   * one output tile (32 features x 32 tokens) is ONE accumulation chain in ascending k, exactly the tile kernel's: the lo wave runs the
//     first half from zero and hands its 16 accumulator registers to the hi wave through 4 KiB of LDS, the hi wave continues the chain one
//     block later (the MFMA's C input) and finishes the tile.  The pair is a two-stage pipeline, skewed by one token block;
//   * a chunk of the LDS ring therefore holds, per token row, [KC columns of the lo half of block m | KC columns of the hi half of block
//     m - 1]; NCH chunks per block (3 at K = 1 536: KC = 256), NS ring slots, one workgroup barrier per chunk, filled by LDS-DMA NS - 1
//     chunks ahead;
//   * the hi wave's epilogue (+ bias, gate, + residual) of tile m - 1 runs inside its MFMA chain of tile m: the residual rows arrive by
//     LDS-DMA (no registers) more than a block earlier, the arithmetic runs in the accumulator layout against that LDS image IN PLACE, and
//     the image leaves row-wise as whole 128-byte segments (streaming stores);
//   * a wave's vector-memory operations complete in issue order, so the roles are split by what may miss to HBM: only the lo waves request
//     (and wait for) token chunks - their queue holds nothing else -, the hi waves own the residual requests and the stores.
// Work split: grid = 8 x slices x ranges-per-XCD; the F / 128 slice workgroups of one token range sit on one XCD (block b and b + 8 share
// one: speed only) and share the range's token rows through its L2.
//
// Bits: every output element = the k-ascending chain of 16-deep MFMA steps from zero, + bias, fma with the gate onto the residual:
// identical to the tile kernels for any launch size (tools/lin2_harness.hip compares the two bit for bit).
#pragma once
#include <type_traits>

#include "common.hip.h"

struct Lin2Args {
    const u16 *Wp;      // packed by k_lin2_pack: [F / 32][2][K / 32][64 lanes] x 16 B
    const u16 *Z;       // [N rounded up to 256][K] bf16
    const float *bias;  // [F]
    const float *gate;  // mods + gate offset, row stride mod_stride (0: one row shared by every trajectory)
    float *h;           // [N][F] fp32 residual stream, updated in place
    int F, N;
    int mod_stride, tpt;  // tokens per trajectory
    unsigned tpt_magic;   // floor(2^32 / tpt) + 1 (0 when tpt == 1)
    int slices, rpx;      // feature slices of 128 (F / 128) and token ranges per XCD: grid = 8 * slices * rpx
    int gate_rows;        // rows of the LDS gate table (>= the trajectories one token range spans; host-checked)
    // LNS instances (round 6: the next sub-block's LayerNorm inside linear1, k_lin1.hip.h LNF): beside h, every hi wave leaves
    //   stats [F / 32][npad] float2 = (mean, sum of squared deviations) of each UPDATED token row over the wave's 32 features,
    // which k_ln_finalize combines into the row's (rstd, -mean rstd).  h itself is bit-identical to the plain instance's.
    float2 *stats;
    int npad;
};

#ifndef LIN2_PD
#define LIN2_PD 2  // token fragments requested this many k-steps ahead of their MFMA
#endif

template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F &f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>());
        static_for<N, I + 1>(f);
    }
}

template <int K, int NCH, int NS, bool HB2>
struct Lin2Cfg {
    static_assert(K % (32 * NCH) == 0, "K / 2 splits into NCH chunks of whole 16-deep k-steps");
    static_assert(NCH >= 3 && NS >= 3, "epilogue schedule: arithmetic, stores, residual request in chunks 0, 1, 2");
    static constexpr int KH = K / 2;          // columns per wave of a pair
    static constexpr int KSH = KH / 16;       // k-steps = stationary A fragments per wave
    static constexpr int KC = KH / NCH;       // columns per wave and chunk
    static constexpr int MPC = KC / 16;       // MFMAs per wave and chunk
    static constexpr int ROWB = 4 * KC;       // bytes per token row of a chunk: [lo KC | hi KC] bf16
    static constexpr int LPR = ROWB / 16;     // active lanes of a row's DMA instruction
    static_assert(LPR <= 64, "one DMA instruction per token row");
    // row pitch = ROWB + 16: (pitch / 4) mod 64 = 4 x odd, so the ds_read_b128 of a B fragment (16 lanes = 16 consecutive token rows,
    // same column) is conflict-free with one per-lane base + an immediate per k-step
    static constexpr int PITCH = ROWB + 16;
    static constexpr int CHUNK = 32 * PITCH;
    static constexpr int RING = NS * CHUNK;
    static constexpr int EXCH = RING;                 // 4 pairs x 4 KiB: lo -> hi accumulator hand-off (lane-private slots)
    static constexpr int HBUF = EXCH + 4 * 4096;      // 4 hi waves x (HB2 ? 2 : 1) x 4 KiB: residual rows of one or two blocks
    static constexpr int HB_PER_WAVE = HB2 ? 8192 : 4096;
    static constexpr int BIAS = HBUF + 4 * HB_PER_WAVE;  // 128 floats
    static constexpr int GATE = BIAS + 512;           // gate_rows x 128 floats
    static constexpr size_t lds_bytes(int gate_rows) { return (size_t)GATE + (size_t)gate_rows * 512; }
    static constexpr int max_gate_rows = (163840 - GATE) / 512;
    static_assert(max_gate_rows >= 1, "LDS budget (160 KiB per workgroup)");
};

// W [F][K] row-major -> fragment order of k_linear2_ws: [F / 32][2 halves][K / 32 k-steps][64 lanes] x 8 bf16, lane (r = lane & 31, hf =
// lane >> 5) of k-step ks of half o = W[32 fb + r][o K/2 + 16 ks + 8 hf .. + 7]: the A operand of mfma32 as it is loaded (common.hip.h)
__global__ void __launch_bounds__(256) k_lin2_pack(u16 *out, const u16 *W, int F, int K) {
    const int ksh = K / 32;
    const long total = (long)(F / 32) * 2 * ksh * 64;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int lane = (int)(i & 63);
        long t = i >> 6;
        const int ks = (int)(t % ksh);
        t /= ksh;
        const int o = (int)(t & 1), fb = (int)(t >> 1);
        const u16 *src = W + (size_t)(32 * fb + (lane & 31)) * K + o * (K / 2) + 16 * ks + 8 * (lane >> 5);
        *reinterpret_cast<u32x4 *>(out + i * 8) = *reinterpret_cast<const u32x4 *>(src);
    }
}

template <int K, int NCH, int NS, bool HB2, bool LNS = false>
__global__ void __launch_bounds__(512, 2) k_linear2_ws(Lin2Args g) {
    using C = Lin2Cfg<K, NCH, NS, HB2>;
    constexpr int KSH = C::KSH, KC = C::KC, MPC = C::MPC, PITCH = C::PITCH, CHUNK = C::CHUNK, PD = LIN2_PD;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hf = lane >> 5;
    const int hi = wave >> 2;  // 0: first half of every chain, 1: second half + epilogue
    const int p = wave & 3;                                        // pair = 32-feature group of the slice

    // workgroup -> (feature slice, token range): the slices of one range share an XCD (blocks b and b + 8 do: speed only)
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int slice = idx % g.slices, range = xcd * g.rpx + idx / g.slices, ranges = 8 * g.rpx;
    const int NBLK = (g.N + 31) >> 5;
    const int blk0 = (int)((long)NBLK * range / ranges), blk1 = (int)((long)NBLK * (range + 1) / ranges);
    const int nblk = blk1 - blk0;
    if (nblk <= 0) return;  // (uniform)
    const int f0 = slice * 128 + 32 * p;  // first feature of the pair

    const unsigned lds0 = (unsigned)(size_t)(LDS_PTR(char))(smem);
    float *const bias_lds = reinterpret_cast<float *>(smem + C::BIAS);
    float *const gate_lds = reinterpret_cast<float *>(smem + C::GATE);

    // ---- stationary weights: this wave's 32 rows x K / 2 columns as A fragments ----
    bf16x8 wreg[KSH];
    {
        const u32x4 *wp = reinterpret_cast<const u32x4 *>(g.Wp) + ((size_t)((slice * 4 + p) * 2 + hi) * KSH) * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < KSH; ++ks) wreg[ks] = as_bf16x8(wp[(size_t)ks * 64]);
    }
    // (waited for below with a builtin that hipcc's wait-count pass sees: left pending, it would put s_waitcnt vmcnt(0) in front of the first
    // use of each register INSIDE the loops, which at run time also drains the LDS-DMA requests and touches it cannot see)
    // ---- bias of the slice and the gate rows of the trajectories this range spans ----
    const int n_lo = blk0 * 32, n_hi = min(blk1 * 32, g.N) - 1;
    const unsigned traj_lo = g.mod_stride ? (g.tpt_magic ? __umulhi((unsigned)n_lo, g.tpt_magic) : (unsigned)n_lo) : 0u;
    {
        const unsigned traj_hi = g.mod_stride ? (g.tpt_magic ? __umulhi((unsigned)n_hi, g.tpt_magic) : (unsigned)n_hi) : 0u;
        const int rows = min((int)(traj_hi - traj_lo) + 1, g.gate_rows);
        if (tid < 32) *reinterpret_cast<float4 *>(bias_lds + 4 * tid) = *reinterpret_cast<const float4 *>(g.bias + slice * 128 + 4 * tid);
        for (int i = tid; i < rows * 32; i += 512) {
            const int row = i >> 5, c4 = i & 31;
            *reinterpret_cast<float4 *>(gate_lds + row * 128 + 4 * c4) =
                *reinterpret_cast<const float4 *>(g.gate + (size_t)(traj_lo + row) * g.mod_stride + slice * 128 + 4 * c4);
        }
    }

    // ---- token chunks.  Chunk c = NCH m + j (block-step m = 0 .. nblk, j = 0 .. NCH - 1) lives in ring slot c % NS and holds, per token row,
    // [lo half of block m, columns j KC .. | hi half of block m - 1, columns K/2 + j KC ..].  Requested by the lo waves only: wave p rows 8 p ..
    // 8 p + 7, one LDS-DMA instruction per row.  Lanes below LPR / 2 carry the lo part, the rest the hi part; both parts from ONE uniform
    // base (the row of block m - 1, column j KC) plus a per-lane offset: + K bytes for the hi part's columns, + 32 rows for the lo part's
    // block (0 rows when both parts come from the same block: the clamped first and last block-steps, whose other half is never used).
    const unsigned half_lanes = C::LPR / 2;
    const unsigned voff_same = lane < half_lanes ? 16u * lane : (unsigned)K + 16u * (lane - half_lanes);
    const unsigned voff_next = lane < half_lanes ? 64u * K + 16u * lane : (unsigned)K + 16u * (lane - half_lanes);
    // (both addresses are wave-uniform by construction; the readfirstlanes make that provable where a ring-slot counter lives in a VGPR)
    auto uni_ptr = [](const char *q) __attribute__((always_inline)) {
        const unsigned long long v = (unsigned long long)q;
        const unsigned lo32 = __builtin_amdgcn_readfirstlane((unsigned)v), hi32 = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<const char *>(((unsigned long long)hi32 << 32) | lo32);
    };
    auto dma_row = [&](const char *src_, unsigned dst_, unsigned voff) __attribute__((always_inline)) {
        const char *src = uni_ptr(src_);
        const unsigned dst = __builtin_amdgcn_readfirstlane(dst_);
        if (C::LPR == 64 || lane < C::LPR)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(dst) : "memory");
    };
    // source row 8 p of chunk (m, j), its per-lane offsets; past the last block-step the requests still go out, clamped: the counted waits
    // rely on 8 per chunk-step
    auto chunk_src = [&](int m, int j, unsigned &voff) __attribute__((always_inline)) {
        int mh = min(max(m - 1, 0), nblk - 1), ml = min(m, nblk - 1);
        voff = ml > mh ? voff_next : voff_same;
        return reinterpret_cast<const char *>(g.Z) + ((size_t)(blk0 + mh) * 32 + 8 * p) * (2 * K) + (size_t)j * (2 * KC);
    };

    // B fragment of the wave's k-step i of a chunk: token row r, columns (hi ? KC : 0) + 16 i + 8 hf
    const int boff = r * PITCH + (hi * KC + 8 * hf) * 2;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;

    // one chunk's MFMAs (chunk j of the block, in ring slot `slot`) with side jobs in their shadows: side(i) runs behind MFMA i
    auto chain = [&](auto jc, int slot, bool zero_start, auto side) __attribute__((always_inline)) {
        constexpr int J = decltype(jc)::value;
        const char *sb = smem + slot * CHUNK + boff;
        bf16x8 fr[PD];
#pragma unroll
        for (int i = 0; i < PD; ++i) fr[i] = as_bf16x8(*reinterpret_cast<const u32x4 *>(sb + 32 * i));
#pragma unroll
        for (int i = 0; i < MPC; ++i) {
            if (J == 0 && i == 0 && zero_start) {
                f32x16 z;
#pragma unroll
                for (int e = 0; e < 16; ++e) z[e] = 0.0f;
                acc = mfma32(wreg[J * MPC + i], fr[i % PD], z);
            } else
                acc = mfma32(wreg[J * MPC + i], fr[i % PD], acc);
            if (i + PD < MPC) fr[i % PD] = as_bf16x8(*reinterpret_cast<const u32x4 *>(sb + 32 * (i + PD)));
            side(i);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    char *const exch = smem + C::EXCH + p * 4096;
    auto next_slot = [](int s) __attribute__((always_inline)) { return s + 1 == NS ? 0 : s + 1; };

    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the stationary weights
    __syncthreads();                     // bias / gate tables

    if (!hi) {
        // =================== lo waves: first half of every chain, all token-chunk requests ===================
        // prologue: chunks 0 .. NS - 2 into slots 0 .. NS - 2
        int rq_m = 0, rq_j = 0, rq_slot = 0;  // next chunk to request
        auto rq_advance = [&]() __attribute__((always_inline)) {
            if (++rq_j == NCH) { rq_j = 0; ++rq_m; }
            rq_slot = next_slot(rq_slot);
        };
#pragma unroll
        for (int c = 0; c < NS - 1; ++c) {
            unsigned voff;
            const char *src = chunk_src(rq_m, rq_j, voff);
            const unsigned dst = lds0 + rq_slot * CHUNK + 8 * p * PITCH;
#pragma unroll
            for (int i = 0; i < 8; ++i) dma_row(src + (size_t)i * (2 * K), dst + i * PITCH, voff);
            rq_advance();
        }
        int slot = 0;
        for (int m = 0; m <= nblk; ++m) {  // block-step m
            const bool work = m < nblk;     // (the last block-step only feeds the hi waves)
            auto step = [&](auto jc) __attribute__((always_inline)) {
                // the chunk to compute was requested NS - 1 chunk-steps ago; younger: the 8 requests of each of the NS - 2 chunks behind it
                wait_vmcnt<8 * (NS - 2)>();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                // request the chunk NS - 1 ahead (its slot held the previous chunk, which every wave has left)
                unsigned voff;
                const char *src = chunk_src(rq_m, rq_j, voff);
                const unsigned dst = lds0 + rq_slot * CHUNK + 8 * p * PITCH;
                rq_advance();
                if (work) {
                    // the four lo waves run in lockstep behind the barrier: requests issued at the same point of the chain arrive at the
                    // vector-memory unit together and queue (measured: ~110 cycles per instruction, 8 per chunk-step in each wave's in-order
                    // stream beside its 16 MFMAs).  Issuing them wave by wave behind the MFMAs i = p (mod 4) was measured and dropped (0.431 vs
                    // 0.418 ms per cfg-2 launch: two back-to-back requests cost more than they save): one row behind every second MFMA
                    chain(jc, slot, true, [&](int i) __attribute__((always_inline)) {
                        if (i % 2 == 1 && i / 2 < 8) dma_row(src + (size_t)(i / 2) * (2 * K), dst + (i / 2) * PITCH, voff);
                    });
                    if (MPC < 16) {
#pragma unroll
                        for (int i = MPC / 2; i < 8; ++i) dma_row(src + (size_t)i * (2 * K), dst + i * PITCH, voff);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) dma_row(src + (size_t)i * (2 * K), dst + i * PITCH, voff);
                }
                slot = next_slot(slot);
            };
            static_for<NCH>(step);
            if (work) {  // hand the half-finished tile to the hi wave (lane-private slots: lane l, register group q at q 1024 + 16 l)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4 *>(exch + q * 1024 + lane * 16) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // written before this wave reaches the next barrier
            }
        }
        wait_vmcnt<0>();  // the run-ahead requests must not land in LDS after the workgroup has gone
        return;
    }

    // =================== hi waves: second half of every chain, epilogue, residual rows, touches ===================
    // Every per-lane address below is derived from an OPAQUE copy of the lane id made inside the loop (hipcc otherwise hoists them all out
    // of the block loop and keeps them in registers it does not have beside 192 stationary ones: scratch spills, whose reloads are
    // s_waitcnt vmcnt(0) - at run time a wait for every touch and request in flight; measured 0.26 ms per launch).
    char *const hbuf = smem + C::HBUF + p * C::HB_PER_WAVE;
    const unsigned hbuf_lds = lds0 + C::HBUF + p * C::HB_PER_WAVE;
    const unsigned row_bytes = 4u * g.F;
    auto hb_off = [](int b) __attribute__((always_inline)) { return HB2 ? (b & 1) * 4096 : 0; };
    auto opaque_lane = [&]() __attribute__((always_inline)) {
        int l = lane;
        asm volatile("" : "+v"(l));
        return l;
    };
    // residual rows of block b of the range -> the wave's LDS image: 4 LDS-DMA instructions of 8 rows x 128 B, slot (row, s) <- chunk s ^ (row & 7)
    auto h_request = [&](int b) __attribute__((always_inline)) {
        const int l = opaque_lane(), tr = l >> 3, ch = (l & 7) ^ (tr & 7);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned n = (unsigned)min((blk0 + b) * 32 + 8 * i + tr, g.N - 1);
            const unsigned voff = n * row_bytes + 4u * f0 + 16u * ch, dst = __builtin_amdgcn_readfirstlane(hbuf_lds + hb_off(b) + i * 1024);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(g.h), "s"(dst) : "memory");
        }
    };
    // (LNS) statistics of block b's UPDATED row over this wave's 32 features, from the values epi_math has just computed (in registers: read back
    // from the LDS image they cost a write -> read turn-around on the hi waves' critical path): the lane's 16 values, two-pass, then the row's 32
    // (the other half sits on lane ^ 32) by the pairwise update of Chan et al. - both lanes compute the same symmetric expression, the exchange
    // is two v_permlane32_swap.  One 8-byte store per row and wave.
    auto epi_stats_regs = [&](int b, const float4 (&w)[4]) __attribute__((always_inline)) {
        const int l = opaque_lane(), rr = l & 31, hh = l >> 5;
        const int n_row = (blk0 + b) * 32 + rr;
        float s1 = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) s1 += (w[q].x + w[q].y) + (w[q].z + w[q].w);
        const float m16 = s1 * (1.0f / 16.0f);
        float q16 = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float dx = w[q].x - m16, dy = w[q].y - m16, dz = w[q].z - m16, dw = w[q].w - m16;
            q16 += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
        const auto pm = __builtin_amdgcn_permlane32_swap(__float_as_uint(m16), __float_as_uint(m16), false, false);
        const auto pq = __builtin_amdgcn_permlane32_swap(__float_as_uint(q16), __float_as_uint(q16), false, false);
        const float m_lo = __uint_as_float((unsigned)pm[0]), m_hi = __uint_as_float((unsigned)pm[1]), dm = m_lo - m_hi;
        const u32x2 st = {__float_as_uint(0.5f * (m_lo + m_hi)),
                          __float_as_uint((__uint_as_float((unsigned)pq[0]) + __uint_as_float((unsigned)pq[1])) + 8.0f * dm * dm)};
        if (hh == 0 && n_row < g.N) {
            const unsigned voff = (unsigned)(((f0 >> 5) * g.npad + n_row) * 8);
            asm volatile("s_nop 4\n\tglobal_store_dwordx2 %0, %1, %2\n\ts_nop 1" ::"v"(voff), "v"(st), "s"(g.stats) : "memory");
        }
    };
    auto epi_math = [&](int b) __attribute__((always_inline)) {  // block b: register group q = features 8 q + 4 hf .. + 3 of token r
        const int l = opaque_lane(), rr = l & 31, hh = l >> 5;
        const int n_row = (blk0 + b) * 32 + rr;
        const unsigned n = (unsigned)min(n_row, g.N - 1);
        const unsigned traj = g.mod_stride ? (g.tpt_magic ? __umulhi(n, g.tpt_magic) : n) - traj_lo : 0u;
        const float *bp = bias_lds + 32 * p + 4 * hh, *gp = gate_lds + traj * 128 + 32 * p + 4 * hh;
        char *hrow = hbuf + hb_off(b) + rr * 128;
        float4 w[4];  // (LNS) the updated values, for the row statistics below: they take over the registers of the accumulators they consume
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bs = *reinterpret_cast<const float4 *>(bp + 8 * q);
            const float4 gt = *reinterpret_cast<const float4 *>(gp + 8 * q);
            float4 *hp = reinterpret_cast<float4 *>(hrow + 16 * ((2 * q + hh) ^ (rr & 7)));
            float4 hv = *hp;
            hv.x = fmaf(gt.x, acc[4 * q] + bs.x, hv.x);
            hv.y = fmaf(gt.y, acc[4 * q + 1] + bs.y, hv.y);
            hv.z = fmaf(gt.z, acc[4 * q + 2] + bs.z, hv.z);
            hv.w = fmaf(gt.w, acc[4 * q + 3] + bs.w, hv.w);
            *hp = hv;
            if constexpr (LNS) w[q] = hv;
        }
        if constexpr (LNS) epi_stats_regs(b, w);
    };
    auto epi_store = [&](int b, int i, bool ragged) __attribute__((always_inline)) {  // rows 8 i .. 8 i + 7 of block b, whole 128-byte segments
        const int l = opaque_lane(), tr = l >> 3, ch = (l & 7) ^ (tr & 7);
        const u32x4 v = *reinterpret_cast<const u32x4 *>(hbuf + hb_off(b) + i * 1024 + l * 16);
        const int n = (blk0 + b) * 32 + 8 * i + tr;
        const unsigned voff = (unsigned)min(n, g.N - 1) * row_bytes + 4u * f0 + 16u * ch;
        if (ragged) {
            if (n < g.N) store16(reinterpret_cast<char *>(g.h) + voff, v, true);
        } else
            asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(g.h) : "memory");
    };

    // hi block-step m (1 .. nblk): chain of block B = m - 1 from the lo wave's hand-off, epilogue of block B - 1.  Vector-memory order of a
    // block-step: 4 stores (block B - 1: chunk 1), 4 requests (chunk 2) of the residual rows of block B + 1 (HB2: two images, the one the
    // stores have just read is free) or B (one image).  So at the arithmetic of block B (chunk 0 of the next block-step) its rows are older
    // than: HB2: 4 stores (B - 1, if any) + 4 requests (B + 1); one image: nothing.
    // Measured and dropped (profiles/r04_experiments.txt): TOUCH loads (one dword per 128-byte line, never waited for) that pull the token
    // rows / residual rows of the blocks ahead into the L2 - every touched line costs the CU's vector-memory path as much as the line's real
    // load, and behind a touch the in-order queue returns the residual rows later, not earlier.
    int slot = 0;
    if (HB2) h_request(0);
    for (int c = 0; c < NCH; ++c) {
        __builtin_amdgcn_s_barrier();
        slot = next_slot(slot);
    }
    for (int m = 1; m <= nblk; ++m) {
        const int B = m - 1;
        auto step = [&](auto jc) __attribute__((always_inline)) {
            constexpr int J = decltype(jc)::value;
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (J == 0) {
                // the finished tile's arithmetic (on the accumulators, against the LDS image of its residual rows), then take over the next
                // chain from the lo wave
                if (B >= 1) {
                    if (!HB2) wait_vmcnt<0>();
                    else if (B >= 2) wait_vmcnt<8 + (LNS ? 1 : 0)>();  // (LNS: + the statistics store of the block before)
                    else wait_vmcnt<4>();
                    asm volatile("" ::: "memory");
                    epi_math(B - 1);
                }
                const int l = opaque_lane();
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4 *>(exch + q * 1024 + l * 16);
                    acc[4 * q] = v.x; acc[4 * q + 1] = v.y; acc[4 * q + 2] = v.z; acc[4 * q + 3] = v.w;
                }
            }
            constexpr int GAP = MPC >= 13 ? 3 : MPC >= 9 ? 2 : 1, IN_CHAIN = (MPC - 2) / GAP + 1 < 4 ? (MPC - 2) / GAP + 1 : 4;  // stores inside the chain
            chain(jc, slot, false, [&](int i) __attribute__((always_inline)) {
                if (J == 1 && B >= 1 && i >= 1 && (i - 1) % GAP == 0 && (i - 1) / GAP < IN_CHAIN) epi_store(B - 1, (i - 1) / GAP, false);
                if (J == 2 && i == 1) h_request(HB2 ? min(B + 1, nblk - 1) : B);
            });
            if (J == 1 && B >= 1) {
#pragma unroll
                for (int k = IN_CHAIN; k < 4; ++k) epi_store(B - 1, k, false);
            }
            slot = next_slot(slot);
        };
        static_for<NCH>(step);
    }
    // the last tile: its residual rows were requested during the last block-step
    wait_vmcnt<0>();
    asm volatile("" ::: "memory");
    {
        const int b = nblk - 1;
        const bool ragged = (blk0 + b) * 32 + 32 > g.N;
        epi_math(b);
#pragma unroll
        for (int i = 0; i < 4; ++i) epi_store(b, i, ragged);
    }
    wait_vmcnt<0>();
}
