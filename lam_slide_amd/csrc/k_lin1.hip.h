// linear1 of a ParallelMLPAttentionV2 block (mmdit.py:240-249) as a TOKEN-STATIONARY bf16 MFMA kernel: round-3 form of the dominant
// kernel of the sampling loop.  Same contract and the same bits as k_gemm_glds<..., EpiLinear1<HDP>> (k_gemm.hip.h), different structure.
//
// Why: in the 256 x 256-tile kernel a workgroup's three phases add up (profiles/r02_experiments.txt): L2->LDS operand feed (as long as
// the MFMAs at K = 512: 1 KiB of operands per 131 072 FLOP), MFMAs, and an epilogue (bias, QK-RMSNorm, RoPE, erf-GELU, transposition,
// stores) that runs with the matrix pipe idle and is amortised over only K / 16 = 32 MFMAs per accumulator tile.
//
// Structure:
//   * a workgroup = 8 waves = 256 tokens; wave w KEEPS its 32 tokens x K activations in registers as the MFMA B fragments of all K / 16
//     k-steps (K / 4 VGPRs: 128 at K = 512), loaded once per token tile;
//   * the weights stream through LDS in blocks of 32 features x K (one head of 32, or two of 16): a ring of 3 blocks filled by LDS-DMA,
//     two blocks ahead, ONE workgroup barrier per block; every wave reads the whole block (A fragments, ds_read_b128, XOR-swizzled on the
//     DMA source side).  Operand feed per FLOP is HALF that of a 256 x 256 tile (the activations never re-stream) and every byte of it
//     is an L2 hit (the weight matrix is 2.6 MB);
//   * per block a wave issues K / 16 MFMAs into ONE 32 x 32 accumulator tile (a single dependent chain runs at the full MFMA rate) while
//     the epilogue of the PREVIOUS block (second accumulator tile) is computed by the same wave in the MFMA shadows: two accumulator
//     sets of 16 VGPRs instead of 128 accumulator VGPRs that all wait for one epilogue;
//   * the epilogue arithmetic is EpiLinear1's (accumulator layout: a lane owns one token and 16 of the block's 32 features; the bias is
//     the initial accumulator; RoPE pairs lane-local; one v_permlane32_swap per head norm), specialised per section (q|k, v, mlp) OUTSIDE the
//     block loop: the section of a weight block is the same for all waves; two blocks are gathered in 4 KiB of wave-private LDS and leave as
//     whole 128-byte row segments;
//   * work = (token tile, block) pairs in one linear order, cut into equal contiguous ranges: every workgroup does the same number of
//     blocks whatever the tile count (no fractional last round).
//
// Bits: every output element is bias + the k-ascending chain of 16-deep MFMA steps, then EpiLinear1's arithmetic in the same order:
// identical to k_gemm_glds for any launch size (tools/lin1_harness.hip compares the two kernels bit for bit).
#pragma once
#include <type_traits>

#include "common.hip.h"

struct Lin1Args {
    const u16 *W;       // [F][K] bf16 (rows padded to 256 by packing.py)
    const u16 *X;       // [N rounded up to 256][K] bf16
    const float *bias;  // [F rounded up to 256]
    const float4 *rope_q, *rope_k;  // [n_pos][HDP/2] (c s0, sn s1, sn s0, c s1): RoPE x QK-norm scale (k_rope_scaled)
    u16 *qkv;           // [N rounded up to 256][3 HHD]
    u16 *z;             // [N rounded up to 256][HHD + M]
    int F, N, HHD, M;
    int pos_div, pos_mod;           // position of token n in its sequence: (n / pos_div) % pos_mod
    unsigned div_magic, mod_magic;  // floor(2^32 / d) + 1 (0 when d == 1)
    float inv_hd, q_premul;
    int nt;                         // streaming stores
};

template <int HDP, int K>
struct Lin1Cfg {
    static_assert(K % 128 == 0 && K <= 512, "hidden sizes 128 / 256 / 384 / 512");
    static constexpr int KS = K / 16;                   // k-steps = B fragments a wave keeps
    static constexpr int ROWB = 2 * K;                  // bytes per weight row
    // LDS image of a weight block: 32 rows at a pitch of ROWB + 16 bytes.  One DMA instruction carries ONE row (K / 8 active lanes, lane-linear
    // in LDS, source linear too), so rows can be padded: the pitch is 16 (mod 256), consecutive rows start one 16-byte bank slot apart and the
    // ds_read_b128 of an A fragment (16 lanes = 16 different rows mod 16, same column) is conflict-free with NO swizzle: the address of
    // k-step ks is one per-lane base + the immediate 32 ks.
    static constexpr int PITCH = ROWB + 16;
    static constexpr int BLK = 32 * PITCH;              // one weight block
    static constexpr int NS = 3;                        // ring slots
    static constexpr int RING = NS * BLK;
    static constexpr int STAGE = 8 * 4096;              // wave-private output staging
    static constexpr int LPR = ROWB / 16;               // active lanes of a DMA instruction: 64 / 48 / 32 / 16
    static constexpr int PPW = 4;                       // DMA instructions (rows) per wave per block
    static constexpr size_t lds_bytes(int F) { return (size_t)RING + STAGE + (size_t)F * 4; }
};

enum { LIN1_QK = 0, LIN1_V = 1, LIN1_MLP = 2 };

template <int HDP, int K>
__global__ void __launch_bounds__(512, 2) k_linear1_ts(Lin1Args g) {
    using C = Lin1Cfg<HDP, K>;
    constexpr int KS = C::KS, BLK = C::BLK, PPW = C::PPW;
    constexpr int NCO = HDP == 32 ? 8 : 4;  // rotation pairs a lane owns per head
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hf = lane >> 5;
    char *const stage = smem + C::RING + wave * 4096;
    float *const bias_lds = reinterpret_cast<float *>(smem + C::RING + C::STAGE);

    const int NB = g.F >> 5;  // weight blocks (F is a multiple of 64: sections start on multiples of 64)
    const int ntile = (g.N + 255) >> 8;
    const long U = (long)ntile * NB;
    const long i0 = (U * blockIdx.x / gridDim.x) & ~1L, i1 = blockIdx.x + 1 == gridDim.x ? U : ((U * (blockIdx.x + 1) / gridDim.x) & ~1L);
    if (i0 >= i1) return;  // (uniform)

    for (int i = tid * 4; i < g.F; i += 512 * 4) *reinterpret_cast<float4 *>(bias_lds + i) = *reinterpret_cast<const float4 *>(g.bias + i);

    // ---- weight ring: block `blk` -> slot; wave w requests rows w, w + 8, w + 16, w + 24 of a block, one LDS-DMA instruction each ----
    // The DMA instruction is inline asm ON PURPOSE: behind the builtin, hipcc's waitcnt pass puts s_waitcnt vmcnt(0) in front of the next LDS
    // access of any kind (it cannot tell the staging image and the bias vector from the ring slot being filled), i.e. it waits for the
    // block it has just requested.  Hidden, the requests stay in flight; their completion is ordered by the counted waits of step_head().
    // (m0 = LDS destination base; written in the same statement that uses it, the compiler's value restored.  s_nop 4: the base SGPRs
    // may come straight from a scalar ALU instruction, and nothing pads the 5 wait states of "SALU writes SGPR -> VMEM reads it" inside asm.)
    const unsigned lds0 = (unsigned)(size_t)(LDS_PTR(char))(smem);
    const unsigned lane_src = lane * 16;  // byte offset of the lane's 16-byte chunk in a weight row
    auto issue = [&](int blk, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int row = wave + 8 * i;
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + slot * BLK + row * C::PITCH);
            const unsigned long long sa = (unsigned long long)(g.W + ((size_t)blk * 32 + row) * K);  // uniform; provably so for the "s" operand:
            const unsigned long long src = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(sa >> 32)) << 32) |
                                           (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)sa);
            unsigned keep_m0;
            if (C::LPR == 64 || lane < C::LPR)
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep_m0)
                             : "v"(lane_src), "s"(src), "s"(dst)
                             : "memory");
        }
    };
    const int aoff = r * C::PITCH + 16 * hf;  // A fragment of k-step ks: + 32 ks

    // ---- state ----
    bf16x8 xreg[KS];   // this wave's 32 tokens: B fragments of every k-step
    float4 co[NCO];    // rotation coefficients of the lane's token for the current q / k section
    f32x16 acc0, acc1; // block b accumulates into acc[b & 1]
    int slot_c = 0;    // ring slot of the next block to compute
    int dma_blk, dma_slot;  // next block to request and its slot
    int n_wave = 0;    // first token of this wave in the current tile
    float post = 1.0f;

    auto advance = [&](int &b) __attribute__((always_inline)) { b = b + 1 == NB ? 0 : b + 1; };
    auto next_slot = [&](int s) __attribute__((always_inline)) { return s == C::NS - 1 ? 0 : s + 1; };

    // ring prologue: the first two blocks of the range
    {
        int b = (int)(i0 % NB);
        issue(b, 0);
        advance(b);
        issue(b, 1);
        advance(b);
        dma_blk = b;
        dma_slot = 2;
    }

    auto init_acc = [&](f32x16 &a, int blk) __attribute__((always_inline)) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const float4 b = *reinterpret_cast<const float4 *>(bias_lds + blk * 32 + 8 * q4 + 4 * hf);
            a[4 * q4] = b.x; a[4 * q4 + 1] = b.y; a[4 * q4 + 2] = b.z; a[4 * q4 + 3] = b.w;
        }
    };
    // step head: the block to compute has landed (all but the youngest PPW vector-memory operations of every wave are done: the youngest
    // are always the DMA of the block after it), every wave has left the previous block -> its slot is free for the block two ahead
    auto step_head = [&]() __attribute__((always_inline)) {
        wait_vmcnt<PPW>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    auto request_next = [&]() __attribute__((always_inline)) {
        issue(dma_blk, dma_slot);  // (past the end of the range: a harmless re-read of the following blocks into a free slot)
        advance(dma_blk);
        dma_slot = next_slot(dma_slot);
    };
    // wave-private staging: [32 tokens][64 features] bf16 = 128-byte rows of eight 16-byte chunks, chunk index XOR (row & 7).
    //   write (accumulator layout): lane (token r, half hf), block half ii, feature group q: 8 bytes at r 128 + 16 ((4 ii + q) ^ (r & 7)) + 8 hf
    //                               = wr0 ^ (64 ii + 16 q): one per-lane register, the XOR constant is compile-time
    //   read (row-wise): lane (tr = lane >> 3, c = lane & 7), rows row0 + tr, row0 = 0, 8, 16, 24: rd0 + 128 row0 (immediates)
    const unsigned wr0 = (unsigned)(size_t)(LDS_PTR(char))(stage) + r * 128 + 8 * hf + ((r & 7) << 4);
    const unsigned rd0 = (unsigned)(size_t)(LDS_PTR(char))(stage) + (lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) << 4);
    // the 64-feature slab that starts at block `b_even` is complete in the staging image: whole 128-byte row segments, 8 token rows per store
    // (32-bit per-lane offsets against a uniform base: both output buffers are far below 4 GiB per pass)
    auto flush_slab = [&](int b_even) __attribute__((always_inline)) {
        const int fs = b_even * 32;
        const bool to_qkv = fs < 3 * g.HHD;  // (uniform: sections start on multiples of 64 features)
        const unsigned stride_b = 2u * (to_qkv ? 3 * g.HHD : g.HHD + g.M);
        const char *base = reinterpret_cast<const char *>(to_qkv ? g.qkv + fs : g.z + (fs - 2 * g.HHD)) + (size_t)n_wave * stride_b;
        const unsigned voff = (lane >> 3) * stride_b + 16 * (lane & 7);
        u32x4 pk[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) pk[i] = *reinterpret_cast<const LDS_PTR(u32x4)>(rd0 + 1024 * i);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(voff + 8 * i * stride_b), "v"(pk[i]), "s"(base) : "memory");
    };
    auto put_group = [&](int ii, int q, float v0, float v1, float v2, float v3) __attribute__((always_inline)) {
        const u32x2 pk = {pack2(v0, v1), pack2(v2, v3)};
        *reinterpret_cast<LDS_PTR(u32x2)>(wr0 ^ (unsigned)(64 * ii + 16 * q)) = pk;
    };
    // EpiLinear1's arithmetic on one accumulator tile (k_gemm.hip.h: EpiLinear1::run; same operations in the same order), cut into 8 slices
    // so that each can sit behind 1/8 of the next block's MFMAs; c0 / c1 carry values between slices; result into half `ii` of the staging
    // image, one group of 4 features (8 bytes) at a time
    auto epi_slice = [&](auto sec_c, int s, const f32x16 &a, int ii, float &c0, float &c1) __attribute__((always_inline)) {
        constexpr int SEC = decltype(sec_c)::value;
        if constexpr (SEC == LIN1_QK && HDP == 32) {
            if (s == 0) c0 = 0.0f;
            if (s < 2) {
#pragma unroll
                for (int e = 0; e < 8; ++e) c0 = fmaf(a[8 * s + e], a[8 * s + e], c0);
            }
            if (s == 1) c0 = rsqrtf(fmaf(half_pair_sum(c0), g.inv_hd, 1e-6f)) * post;
            if (s >= 2 && s < 6) {
                const int q = s - 2;
                float o[4];
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int k = 2 * q + kk;
                    const float x0 = a[2 * k], x1 = a[2 * k + 1];
                    o[2 * kk] = c0 * fmaf(co[k].x, x0, -co[k].y * x1);
                    o[2 * kk + 1] = c0 * fmaf(co[k].z, x0, co[k].w * x1);
                }
                put_group(ii, q, o[0], o[1], o[2], o[3]);
            }
        } else if constexpr (SEC == LIN1_QK) {  // two 16-wide heads per block: accumulator rows 0-15 and 16-31
            if (s == 0) {
                c0 = 0.0f;
#pragma unroll
                for (int e = 0; e < 8; ++e) c0 = fmaf(a[e], a[e], c0);
            }
            if (s == 1) {
                c1 = 0.0f;
#pragma unroll
                for (int e = 0; e < 8; ++e) c1 = fmaf(a[8 + e], a[8 + e], c1);
                c0 = rsqrtf(fmaf(half_pair_sum(c0), g.inv_hd, 1e-6f)) * post;
                c1 = rsqrtf(fmaf(half_pair_sum(c1), g.inv_hd, 1e-6f)) * post;
            }
            if (s >= 2 && s < 6) {
                const int q = s - 2;
                const float rr = q < 2 ? c0 : c1;
                float o[4];
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int k = 2 * q + kk;
                    const float x0 = a[2 * k], x1 = a[2 * k + 1];
                    const float4 cf = co[k & (NCO - 1)];
                    o[2 * kk] = rr * fmaf(cf.x, x0, -cf.y * x1);
                    o[2 * kk + 1] = rr * fmaf(cf.z, x0, cf.w * x1);
                }
                put_group(ii, q, o[0], o[1], o[2], o[3]);
            }
        } else if constexpr (SEC == LIN1_MLP) {
            const float g0 = gelu_fast(a[2 * s]), g1 = gelu_fast(a[2 * s + 1]);
            if (s & 1) put_group(ii, s >> 1, c0, c1, g0, g1);
            else { c0 = g0; c1 = g1; }
        } else {
            if (s < 4) put_group(ii, s, a[4 * s], a[4 * s + 1], a[4 * s + 2], a[4 * s + 3]);
        }
    };
    // One step = [head] + 8 slices, each = K / 128 MFMAs of the block being computed (DO_MFMA, into acc[1 - PAR]) + one slice of the epilogue
    // of the previous block (DO_EPI, from acc[PAR], section SEC), with a scheduling fence between slices: the epilogue's vector instructions
    // issue in the shadows of the MFMAs of the same wave (and of its SIMD partner), and no more than one slice's temporaries are live.
    // A fragments are requested PD k-steps ahead of the MFMA that takes them.
    auto step = [&](auto sec_c, auto par_c, auto mfma_c, auto epi_c) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_c)::value;
        constexpr bool DO_MFMA = decltype(mfma_c)::value != 0, DO_EPI = decltype(epi_c)::value != 0;
        constexpr int MPS = KS / 8, PD = 3;
        f32x16 &ac = PAR ? acc0 : acc1;
        const f32x16 &ae = PAR ? acc1 : acc0;
        const char *sb = smem + slot_c * BLK + aoff;
        bf16x8 fr[PD];
        float c0 = 0.0f, c1 = 0.0f;
        if (DO_MFMA) {
#pragma unroll
            for (int ks = 0; ks < PD; ++ks) fr[ks] = as_bf16x8(*reinterpret_cast<const u32x4 *>(sb + 32 * ks));
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (DO_MFMA) {
#pragma unroll
                for (int m = 0; m < MPS; ++m) {
                    const int ks = s * MPS + m;
                    ac = mfma32(fr[ks % PD], xreg[ks], ac);
                    if (ks + PD < KS) fr[ks % PD] = as_bf16x8(*reinterpret_cast<const u32x4 *>(sb + 32 * (ks + PD)));
                }
            }
            if (DO_EPI) epi_slice(sec_c, s, ae, PAR, c0, c1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (DO_MFMA) slot_c = next_slot(slot_c);
    };
    std::integral_constant<int, 0> I0;
    std::integral_constant<int, 1> I1;
    // fused step: MFMAs of block e + 1 beside the epilogue of block e (parity PAR = e & 1, section SEC)
    auto fused = [&](auto sec_c, auto par_c, int e, bool pending) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_c)::value;
        step_head();
        if (PAR == 0 && pending) flush_slab(e - 2);  // stores BEFORE the DMA request: the youngest operations at the next head are the DMA
        request_next();
        init_acc(PAR ? acc0 : acc1, e + 1);
        step(sec_c, par_c, I1, I1);
    };
    auto run = [&](auto sec_c, int ea, int eb, int b0) __attribute__((always_inline)) {  // fused steps for epilogue blocks [ea, eb) of one section; ea is even
        int e = ea;
        for (; e + 1 < eb; e += 2) {
            fused(sec_c, I0, e, e > b0);
            fused(sec_c, I1, e + 1, false);
        }
        if (e < eb) fused(sec_c, I0, e, e > b0);
    };
    auto load_co = [&](const float4 *tab, unsigned pos) __attribute__((always_inline)) {
        const float4 *t = tab + (size_t)pos * (HDP / 2) + 2 * hf;
#pragma unroll
        for (int k = 0; k < NCO; ++k) co[k] = t[4 * (k >> 1) + (k & 1)];
    };

    const int qb = g.HHD >> 5;  // blocks per q / k / v section
    long i = i0;
    __syncthreads();  // bias vector in LDS
    while (i < i1) {  // one segment = blocks [b0, b1) of one token tile; b0, b1 even
        const int tile = (int)(i / NB), b0 = (int)(i % NB);
        const int b1 = (int)((long)NB - b0 < i1 - i ? NB : b0 + (i1 - i));
        n_wave = tile * 256 + wave * 32;
        // the wave's tokens: B fragments of all k-steps (X is padded to whole tiles)
        {
            const u16 *xr = g.X + (size_t)(n_wave + r) * K + 8 * hf;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) xreg[ks] = as_bf16x8(*reinterpret_cast<const u32x4 *>(xr + 16 * ks));
        }
        const unsigned nn = (unsigned)min(n_wave + r, g.N - 1);
        const unsigned n1 = g.div_magic ? __umulhi(nn, g.div_magic) : nn;
        const unsigned pos = g.mod_magic ? n1 - __umulhi(n1, g.mod_magic) * (unsigned)g.pos_mod : 0u;

        // first block of the segment: MFMAs only.  The vector-memory queue holds this segment's register loads behind the ring's DMA, so
        // the counted wait does not apply: drain everything BEFORE the barrier (a wave may read a block only once every wave's pieces
        // of it have landed)
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        request_next();
        init_acc(acc0, b0);
        step(std::integral_constant<int, LIN1_V>(), I1, I1, I0);
        // fused steps e = b0 .. b1 - 2, split by the section of e
        const int e_end = b1 - 1;
        {
            const int lo = max(b0, 0), hi = min(e_end, qb);
            if (lo < hi) {
                load_co(g.rope_q, pos);
                post = g.q_premul;
                run(std::integral_constant<int, LIN1_QK>(), lo, hi, b0);
            }
        }
        {
            const int lo = max(b0, qb), hi = min(e_end, 2 * qb);
            if (lo < hi) {
                load_co(g.rope_k, pos);
                post = 1.0f;
                run(std::integral_constant<int, LIN1_QK>(), lo, hi, b0);
            }
        }
        {
            const int lo = max(b0, 2 * qb), hi = min(e_end, 3 * qb);
            if (lo < hi) run(std::integral_constant<int, LIN1_V>(), lo, hi, b0);
        }
        {
            const int lo = max(b0, 3 * qb), hi = min(e_end, NB);
            if (lo < hi) run(std::integral_constant<int, LIN1_MLP>(), lo, hi, b0);
        }
        // last block of the segment (odd): epilogue only, then its slab
        {
            const int e = b1 - 1;
            if (e < 2 * qb) {
                load_co(e < qb ? g.rope_q : g.rope_k, pos);
                post = e < qb ? g.q_premul : 1.0f;
                step(std::integral_constant<int, LIN1_QK>(), I1, I0, I1);
            } else if (e < 3 * qb) {
                step(std::integral_constant<int, LIN1_V>(), I1, I0, I1);
            } else {
                step(std::integral_constant<int, LIN1_MLP>(), I1, I0, I1);
            }
            flush_slab(e - 1);
        }
        i += b1 - b0;
    }
    wait_vmcnt<0>();  // the ring's run-ahead requests must not land in LDS after the workgroup has gone
}
