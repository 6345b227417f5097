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
//     two blocks ahead, ONE workgroup barrier per block (K <= 256: 4 blocks, one barrier per two - LIN1_B2 below); every wave reads the whole block (A fragments, ds_read_b128, XOR-swizzled on the
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
//   * LNF = true (round 6; lsl_model_set_ln_fuse): X is the fp32 residual stream - the wave's rows arrive as 128-byte lines, are normalised and
//     modulated (LayerNorm + modulate of the sub-block, statistics from k_linear2_ws<LNS> + k_ln_finalize, modulation rows in an LDS table),
//     rounded once and turned into the same fragments through the staging image: no LayerNorm launch, no bf16 operand buffer.  Costs the launch
//     ~ 36 us at 245 760 x 512 (the rows arrive behind the previous segment's store acknowledgements), saves the 130 us LayerNorm launch.
//   * NW = 4 (round 6): the same kernel on 4-wave workgroups and 128-token tiles - one wave per SIMD (512 registers: no scratch at K = 512), every
//     wave requests 8 rows of a block.  Same bits.  Slower wherever the launch is large (a lone wave's latencies have no partner to fill them:
//     709.8 vs 610.9 us at 245 760 x 512), faster where the activation prologue dominates: K = 512 up to ~10 000 tokens (host_launch.hip.h).
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
    int wpt;                        // 0: the (tile, block) sequence is cut evenly over the grid; > 0: wpt workgroups per token tile, grid = wpt x tiles
    int planes;                     // 1 = q / k / v leave as head-major planes qkv[section][head][npad tokens][HDP] (a (sequence, head)'s rows are
                                    // then contiguous: k_attention_stream's spatial units), 0 = token-major rows qkv[token][3 HHD]
    int npad;                       // tokens rounded up to 256 (the plane pitch)
    // LNF instances (k_linear1_ts<.., true>): X is the fp32 residual stream h [N rounded up to 256][K]; LayerNorm (eps 1e-6) + modulate of this
    // sub-block (latent_si_v31.py:50-51,57-58) are applied while the rows become MFMA fragments: (x rstd - mean rstd)(1 + scale_k) + shift_k
    const float2 *ln_stat;             // [N rounded up to 256] (rstd, -mean rstd) per token (k_linear2_ws<LNS> + k_ln_finalize)
    const float *ln_shift, *ln_scale;  // modulation rows, row stride ln_mod_stride (0: one row shared by every trajectory)
    int ln_mod_stride, ln_tpt;         // tokens per trajectory
    unsigned ln_tpt_magic;             // floor(2^32 / tpt) + 1 (0 when tpt == 1)
};

// K <= 256 (a block is 16 or 8 MFMAs per wave against the same per-step head): 4 ring slots and ONE wait + barrier per PAIR of blocks - both
// blocks of the next pair are requested in the pair's first step (8 instructions behind its 8 slices), confirmed at the head of the next pair.
// Bit-identical (tools/gpu_lin1b2.sh: 22 harness runs); 163 840 x 256: 0.206 -> 0.198 ms, 10 240 x 256: 19.3 -> 18.8 us.  K = 384 / 512 have no LDS
// for a fourth slot, and the barrier is not what costs there: the same pairing with the two halves of the workgroup swapping priority every
// step (2) gains nothing - the halves' skew is the SIMD's issue arbitration, not the barrier count (profiles/r04_experiments.txt).
#ifndef LIN1_B2
#define LIN1_B2 1
#endif
#ifndef LIN1_B2_KMAX
#define LIN1_B2_KMAX 256  // widest K with the paired form
#endif

template <int HDP, int K, int NW = 8>
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
    static constexpr int NS = (LIN1_B2 && K <= LIN1_B2_KMAX) ? 4 : 3;  // ring slots
    static constexpr int RING = NS * BLK;
    static_assert(NW == 8 || NW == 4, "two waves per SIMD (256-token tiles) or one (128-token tiles)");
    static constexpr int TT = 32 * NW;                  // tokens per tile
    static constexpr int STAGE = NW * 4096;             // wave-private output staging
    static constexpr int LPR = ROWB / 16;               // active lanes of a DMA instruction: 64 / 48 / 32 / 16
    static constexpr int PPW = 32 / NW;                 // DMA instructions (rows) per wave per block
    static constexpr size_t lds_bytes(int F) { return (size_t)RING + STAGE + (size_t)F * 4; }
    // LNF: (1 + scale | shift) rows of the trajectories a token tile can touch: 3 (tokens per trajectory >= TT / 2) up to K = 256, 2 (>= TT) above
    static constexpr int LN_SLOTS = K <= 256 ? 3 : 2;
    static constexpr size_t lds_bytes_lnf(int F) { return lds_bytes(F) + (size_t)LN_SLOTS * 2 * K * 4; }
};

enum { LIN1_QK = 0, LIN1_V = 1, LIN1_MLP = 2 };

#ifndef LIN1_XLOAD
#define LIN1_XLOAD 1  // 1: activations loaded as whole cache lines and transposed through the staging image; 0: fragment-shaped loads
#endif
#ifndef LIN1_PD
#define LIN1_PD 3  // A fragments requested this many k-steps ahead of their MFMA
#endif

// erf-GELU of an accumulator value: gelu_fast() (common.hip.h) with max(x, 0) as ONE v_max_f32.  Through the builtins hipcc emits two (a
// canonicalising v_max x, x first), and the epilogue of an mlp block is bound by its vector-instruction count.  Inline asm is invisible to
// the hazard recognizer, so this form may only read values that the matrix pipe finished writing long ago: here the accumulator tile of the
// PREVIOUS block (hundreds of cycles).  Same value as gelu_fast for every non-NaN input except the SIGN of a zero result (x = -0.0: v_max
// gives +0, v_med3 keeps -0, and the bf16 packing keeps the sign bit): the two linear1 paths are bit-identical up to that.
__device__ __forceinline__ float lin1_gelu(float x) {
    const float ax = fabsf(x);
    const float h = __builtin_amdgcn_exp2f(LSL_GELU_Q(ax));
    float relu;
    asm("v_max_f32 %0, 0, %1" : "=v"(relu) : "v"(x));
    return fmaf(-ax, h, relu);
}

template <int HDP, int K, int NW = 8, bool LNF = false>
__global__ void __launch_bounds__(NW * 64, NW / 4) k_linear1_ts(Lin1Args g) {
    using C = Lin1Cfg<HDP, K, NW>;
    constexpr int KS = C::KS, BLK = C::BLK, PPW = C::PPW, ROWB = C::ROWB;
    constexpr int NCO = HDP == 32 ? 8 : 4;  // rotation pairs a lane owns per head
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hf = lane >> 5;
    char *const stage = smem + C::RING + wave * 4096;
    float *const bias_lds = reinterpret_cast<float *>(smem + C::RING + C::STAGE);

    const int NB = g.F >> 5;  // weight blocks (F is a multiple of 64: sections start on multiples of 64)
    const int ntile = (g.N + C::TT - 1) / C::TT;
    const long U = (long)ntile * NB;
    // Work split.  Large launches: the (tile, block) sequence cut evenly (on even blocks: sections start on multiples of 64 features and
    // blocks alternate accumulators) over the persistent workgroups; a range that crosses a tile boundary pays a second segment
    // (activation load, pipeline fill, drain).  Small launches (fewer tiles than workgroups): wpt workgroups share a tile, every range is
    // ONE segment - the longest workgroup, which sets the launch time, no longer has two.
    long i0, i1;
    if (g.wpt > 0) {
        const int tile = blockIdx.x / g.wpt, part = blockIdx.x - tile * g.wpt;
        i0 = (long)tile * NB + ((NB * part / g.wpt) & ~1);
        i1 = (long)tile * NB + (part + 1 == g.wpt ? NB : ((NB * (part + 1) / g.wpt) & ~1));
    } else {
        i0 = (U * blockIdx.x / gridDim.x) & ~1L;
        i1 = blockIdx.x + 1 == gridDim.x ? U : ((U * (blockIdx.x + 1) / gridDim.x) & ~1L);
    }
    if (i0 >= i1) return;  // (uniform)

    // In an 8-wave workgroup the second-dispatched half loses the issue arbitration on every SIMD (priority, then age): measured, waves 4-7
    // take 30 % longer over a block than their partners, which then wait for them at the barrier.  A static raise of that half only swaps
    // the roles (measured: waves 0-3 then take 30 % longer), alternating the priority slice by slice slows both (+10 %): off.  A 4-wave
    // form of this kernel (one wave per SIMD, 64 tokens and the whole 512-register file per wave, each weight fragment read once for two
    // MFMAs) was built and is bit-identical, but as scheduled by hipcc it is 11 % slower at K = 256 and spills at K = 512
    // (profiles/r03_experiments.txt).
    for (int i = tid * 4; i < g.F; i += NW * 64 * 4) *reinterpret_cast<float4 *>(bias_lds + i) = *reinterpret_cast<const float4 *>(g.bias + i);

    // ---- weight ring: block -> slot; wave w requests rows 4 w .. 4 w + 3 of a block, one LDS-DMA instruction per row ----
    // The DMA instruction is inline asm ON PURPOSE: behind the builtin, hipcc's waitcnt pass puts s_waitcnt vmcnt(0) in front of the next LDS
    // access of any kind (it cannot tell the staging image and the bias vector from the ring slot being filled), i.e. it waits for the
    // block it has just requested.  Hidden, the requests stay in flight; their completion is ordered by the counted waits of step_head().
    // One uniform base per wave and block for both sides; the instruction's immediate offset (i ROWB: added to the global AND the LDS
    // address) walks the rows, m0 = LDS base + 16 i supplies the pitch padding.  m0 is written in the statement that uses it; s_add_u32
    // writes SCC, which the statement declares (hipcc keeps ring-slot compares live in SCC across it otherwise: wrong slots, now and then).
    const unsigned lds0 = (unsigned)(size_t)(LDS_PTR(char))(smem);
    const unsigned lane_src = lane * 16;  // byte offset of the lane's 16-byte chunk in a weight row
    const char *const w_rows = reinterpret_cast<const char *>(g.W) + (size_t)(PPW * wave) * ROWB;  // row PPW w of block 0
    auto issue_piece = [&](const char *src, unsigned dst, auto ic) __attribute__((always_inline)) {
        constexpr int I = decltype(ic)::value & 3;
        if (decltype(ic)::value >= 4) {  // (4-wave form: rows 4 .. 7 of the wave's share; the immediate offset reaches 3 rows at K = 512)
            src += 4 * ROWB;
            dst += 4 * C::PITCH;
        }
        const unsigned ls = lane_src;  // (an odr-use: a generic lambda does not capture a variable that only appears as an asm operand)
        if (C::LPR == 64 || lane < C::LPR) {
            // (s_nop 4 on the first piece: src / dst may come straight from a scalar ALU instruction and nothing pads the 5 wait states of
            // "SALU writes SGPR -> VMEM reads it" inside asm; the later pieces read the same registers, long since written)
            if (I == 0 || decltype(ic)::value >= 4)
                asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1 offset:%4" ::"v"(ls), "s"(src), "s"(dst), "n"(16 * I), "n"(ROWB * I) : "memory", "scc");
            else
                asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%4" ::"v"(ls), "s"(src), "s"(dst), "n"(16 * I), "n"(ROWB * I) : "memory", "scc");
        }
    };
    auto req_src = [&](int blk) __attribute__((always_inline)) { return w_rows + (size_t)blk * (32 * ROWB); };
    auto req_dst = [&](int slot) __attribute__((always_inline)) { return lds0 + slot * BLK + PPW * wave * C::PITCH; };
    auto issue = [&](int blk, int slot) __attribute__((always_inline)) {
        const char *src = req_src(blk);
        const unsigned dst = req_dst(slot);
        issue_piece(src, dst, std::integral_constant<int, 0>());
        issue_piece(src, dst, std::integral_constant<int, 1>());
        issue_piece(src, dst, std::integral_constant<int, 2>());
        issue_piece(src, dst, std::integral_constant<int, 3>());
        if (PPW == 8) {
            issue_piece(src, dst, std::integral_constant<int, 4>());
            issue_piece(src, dst, std::integral_constant<int, 5>());
            issue_piece(src, dst, std::integral_constant<int, 6>());
            issue_piece(src, dst, std::integral_constant<int, 7>());
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
        int b = (int)(i0 - (i0 / NB) * NB);
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
    // Vector-memory bookkeeping.  A fused step issues, in this order: [the 4 stores of a finished slab: every other step] and the 4 DMA
    // instructions of the block two ahead, all inside its MFMA chain.  At the head of a step the block to compute must have landed: it
    // was requested two steps ago, and everything the wave issued after that request is younger: the previous step's stores (if it
    // flushed a slab) and its 4 DMA instructions.  So s_waitcnt vmcnt(8) (previous step flushed) or vmcnt(4) leaves exactly those in
    // flight: no head waits for the acknowledgement of a store issued less than a step ago.  Extra younger operations only make a
    // counted wait conservative, never wrong.  Then the workgroup barrier: every wave's pieces have landed and every wave has left the
    // previous block, whose slot is free for the block two ahead.
    constexpr bool B2 = LIN1_B2 != 0 && K <= LIN1_B2_KMAX;
    auto step_head = [&](auto flushed_c) __attribute__((always_inline)) {
        if (B2) {  // (head of a pair of blocks: both were requested two steps ago, behind them only the previous step's slab stores)
            if (decltype(flushed_c)::value) wait_vmcnt<PPW>();
            else wait_vmcnt<0>();
        } else {
            if (decltype(flushed_c)::value) wait_vmcnt<2 * PPW>();
            else wait_vmcnt<PPW>();
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // wave-private staging: [32 tokens][64 features] bf16 = 128-byte rows of eight 16-byte chunks, chunk index XOR (row & 7).
    //   write (accumulator layout): lane (token r, half hf), block half ii, feature group q: 8 bytes at r 128 + 16 ((4 ii + q) ^ (r & 7)) + 8 hf
    //                               = wr0 ^ (64 ii + 16 q): one per-lane register, the XOR constant is compile-time
    //   read (row-wise): lane (tr = lane >> 3, c = lane & 7), rows row0 + tr, row0 = 0, 8, 16, 24: rd0 + 128 row0 (immediates)
    const unsigned wr0 = (unsigned)(size_t)(LDS_PTR(char))(stage) + r * 128 + 8 * hf + ((r & 7) << 4);
    const unsigned rd0 = (unsigned)(size_t)(LDS_PTR(char))(stage) + (lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) << 4);
    // A finished 64-feature slab (blocks b_even, b_even + 1) leaves the staging image as whole 128-byte row segments, 8 token rows per store
    // instruction (32-bit per-lane offsets against a uniform base: both output buffers are far below 4 GiB per pass); two halves of 2
    // instructions each, so that a fused step can place them inside its MFMA chain.  Per segment: the wave's first row in both buffers.
    const char *row_q = nullptr, *row_z = nullptr;
    // q / k / v as head-major planes (Lin1Args::planes, 32-wide heads): block b IS plane b (section x head); a token row of a plane is 64
    // bytes, and of the slab's row-wise lanes (row lane >> 3, chunk c = lane & 7) those with c >= 4 belong to the next plane: the same
    // flush with other constants - 8 rows x 64 B contiguous per plane and instruction
    // (16-wide heads: a block is two heads, the slab four; a plane row is 32 bytes = two chunks)
    const bool planes = g.planes != 0;  // (uniform)
    constexpr unsigned PROW = 2u * HDP, CPH = HDP / 8;  // bytes of a plane's token row, 16-byte chunks per head
    const unsigned plane_bytes = PROW * (unsigned)g.npad;
    const unsigned stride_q = planes ? PROW : 2u * 3 * g.HHD, stride_z = 2u * (g.HHD + g.M);
    const unsigned blk_q = planes ? (32 / HDP) * plane_bytes : 64u;  // bytes between the q / k / v destinations of consecutive blocks
    const unsigned voff_q = planes ? ((lane & 7) / CPH) * plane_bytes + (lane >> 3) * PROW + 16 * ((lane & 7) % CPH) : (lane >> 3) * stride_q + 16 * (lane & 7);
    const unsigned voff_z = (lane >> 3) * stride_z + 16 * (lane & 7);
    const int slab_z = 3 * (g.HHD >> 5);  // first block that goes to z (uniform; sections start on multiples of 64 features)
    const char *fl_base = nullptr;
    unsigned fl_stride8 = 0, fl_voff = 0;
    auto flush_setup = [&](int b_even) __attribute__((always_inline)) {
        const bool to_qkv = b_even < slab_z;
        fl_base = to_qkv ? row_q : row_z;
        fl_stride8 = 8 * (to_qkv ? stride_q : stride_z);
        fl_voff = to_qkv ? voff_q + blk_q * (unsigned)b_even : voff_z + 64u * (unsigned)b_even;  // (32-bit: a pass's q / k / v and z stay below 4 GiB)
    };
    auto flush_read = [&](int half, u32x4 (&pk)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) pk[i] = *reinterpret_cast<const LDS_PTR(u32x4)>(rd0 + 1024 * (2 * half + i));
    };
    auto flush_store = [&](int half, const u32x4 (&pk)[2]) __attribute__((always_inline)) {
        asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(fl_voff + (2 * half) * fl_stride8), "v"(pk[0]), "s"(fl_base) : "memory");
        asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(fl_voff + (2 * half + 1) * fl_stride8), "v"(pk[1]), "s"(fl_base) : "memory");
    };
    auto flush_store_visible = [&](int half, const u32x4 (&pk)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_nontemporal_store(pk[i], reinterpret_cast<u32x4 *>(const_cast<char *>(fl_base) + (size_t)(fl_voff + (2 * half + i) * fl_stride8)));
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
            const float g0 = lin1_gelu(a[2 * s]), g1 = lin1_gelu(a[2 * s + 1]);
            if (s & 1) put_group(ii, s >> 1, c0, c1, g0, g1);
            else { c0 = g0; c1 = g1; }
        } else {
            if (s < 4) put_group(ii, s, a[4 * s], a[4 * s + 1], a[4 * s + 2], a[4 * s + 3]);
        }
    };
    // One step = 8 slices, each = K / 128 MFMAs of the block being computed (DO_MFMA, into acc[1 - PAR]) + one slice of the epilogue of the
    // previous block (DO_EPI, from acc[PAR], section SEC) + one side job (a pair of slab stores, one DMA instruction), with a scheduling
    // fence between slices: everything that is not an MFMA issues in the shadows of the MFMAs of the same wave (and of its SIMD partner),
    // and no more than one slice's temporaries are live.  A fragments are requested PD k-steps ahead of the MFMA that takes them.
    auto step = [&](auto sec_c, auto par_c, auto mfma_c, auto epi_c, auto side) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_c)::value;
        constexpr bool DO_MFMA = decltype(mfma_c)::value != 0, DO_EPI = decltype(epi_c)::value != 0;
        constexpr int MPS = KS / 8, PD = LIN1_PD, MH = (MPS + 1) / 2;
        f32x16 &ac = PAR ? acc0 : acc1;
        const f32x16 &ae = PAR ? acc1 : acc0;
        const char *sb = smem + slot_c * BLK + aoff;
        bf16x8 fr[PD];
        float c0 = 0.0f, c1 = 0.0f;
        if (DO_MFMA) {
#pragma unroll
            for (int ks = 0; ks < PD; ++ks) fr[ks] = as_bf16x8(*reinterpret_cast<const u32x4 *>(sb + 32 * ks));
        }
        side(std::integral_constant<int, -1>());
#define LIN1_SIDE(S) if (s == S) side(std::integral_constant<int, S>());
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#pragma unroll
            for (int m = 0; m < MPS; ++m) {
                if (m == MH) { LIN1_SIDE(0) LIN1_SIDE(1) LIN1_SIDE(2) LIN1_SIDE(3) LIN1_SIDE(4) LIN1_SIDE(5) LIN1_SIDE(6) LIN1_SIDE(7) }
                if (DO_MFMA) {
                    const int ks = s * MPS + m;
                    ac = mfma32(fr[ks % PD], xreg[ks], ac);
                    if (ks + PD < KS) fr[ks % PD] = as_bf16x8(*reinterpret_cast<const u32x4 *>(sb + 32 * (ks + PD)));
                }
            }
            if (MH == MPS) { LIN1_SIDE(0) LIN1_SIDE(1) LIN1_SIDE(2) LIN1_SIDE(3) LIN1_SIDE(4) LIN1_SIDE(5) LIN1_SIDE(6) LIN1_SIDE(7) }  // (one MFMA per slice)
            if (DO_EPI) epi_slice(sec_c, s, ae, PAR, c0, c1);
            __builtin_amdgcn_sched_barrier(0);
        }
#undef LIN1_SIDE
        if (decltype(mfma_c)::value != 0) slot_c = next_slot(slot_c);
    };
    std::integral_constant<int, 0> I0;
    std::integral_constant<int, 1> I1;
    auto no_side = [](auto) __attribute__((always_inline)) {};
    // fused step: MFMAs of block e + 1 beside the epilogue of block e (parity PAR = e & 1, section SEC).  FLUSH (PAR == 0 steps except the first
    // of a segment): the slab (e - 2, e - 1) is complete in the staging image and leaves during this step; PREV_FLUSHED: the previous step did
    auto fused = [&](auto sec_c, auto par_c, auto flush_c, auto prev_c, int e) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_c)::value;
        constexpr bool FLUSH = decltype(flush_c)::value != 0;
        if (!B2 || PAR == 1) step_head(prev_c);  // (B2: the pair (e + 1, e + 2) starts with the odd e)
        u32x4 pk[2];
        if (FLUSH) flush_setup(e - 2);
        const char *src = req_src(dma_blk);
        const unsigned dst = req_dst(dma_slot);
        int blk2 = dma_blk;
        advance(blk2);
        const char *src2 = req_src(blk2);
        const unsigned dst2 = req_dst(next_slot(dma_slot));
        init_acc(PAR ? acc0 : acc1, e + 1);
        step(sec_c, par_c, I1, I1, [&](auto sl) __attribute__((always_inline)) {
            constexpr int SL = decltype(sl)::value;
            if (FLUSH) {  // slab stores: rows read just ahead of their stores, all of it ahead of this step's first staging write
                if (SL == -1) flush_read(0, pk);
                if (SL == 0) {
                    flush_store(0, pk);
                    flush_read(1, pk);
                }
                if (SL == 1) flush_store(1, pk);
            }
            constexpr int PS = PPW / 4;  // pieces per slice: 1 (8 waves), 2 (4 waves)
            if (!B2) {
                if (SL >= 2 && SL < 6) {
                    issue_piece(src, dst, std::integral_constant<int, (SL >= 2 && SL < 6) ? PS * (SL - 2) : 0>());
                    if (PS == 2) issue_piece(src, dst, std::integral_constant<int, (SL >= 2 && SL < 6) ? PS * (SL - 2) + 1 : 0>());
                }
            } else if (PAR == 1) {  // both blocks of the next pair: their pieces behind the 8 slices (PAR == 1 steps never flush)
                if (SL >= 0 && SL < 4) {
                    issue_piece(src, dst, std::integral_constant<int, (SL >= 0 && SL < 4) ? PS * SL : 0>());
                    if (PS == 2) issue_piece(src, dst, std::integral_constant<int, (SL >= 0 && SL < 4) ? PS * SL + 1 : 0>());
                }
                if (SL >= 4 && SL < 8) {
                    issue_piece(src2, dst2, std::integral_constant<int, (SL >= 4 && SL < 8) ? PS * (SL - 4) : 0>());
                    if (PS == 2) issue_piece(src2, dst2, std::integral_constant<int, (SL >= 4 && SL < 8) ? PS * (SL - 4) + 1 : 0>());
                }
            }
        });
        if (!B2) {
            advance(dma_blk);
            dma_slot = next_slot(dma_slot);
        } else if (PAR == 1) {
            advance(dma_blk);
            advance(dma_blk);
            dma_slot = next_slot(next_slot(dma_slot));
        }
    };
    // fused steps for the epilogue blocks [ea, eb) of one section; every even step flushes (the segment's first fused step is not run here)
    auto run = [&](auto sec_c, int ea, int eb) __attribute__((always_inline)) {
        int e = ea;
        if (e < eb && (e & 1)) {
            fused(sec_c, I1, I0, I0, e);  // (odd start: the step before it was the segment's first fused step, which does not flush)
            ++e;
        }
        for (; e + 1 < eb; e += 2) {
            fused(sec_c, I0, I1, I0, e);
            fused(sec_c, I1, I0, I1, e + 1);
        }
        if (e < eb) fused(sec_c, I0, I1, I0, e);
    };
    auto load_co = [&](const float4 *tab, unsigned pos) __attribute__((always_inline)) {
        const float4 *t = tab + (size_t)pos * (HDP / 2) + 2 * hf;
#pragma unroll
        for (int k = 0; k < NCO; ++k) co[k] = t[4 * (k >> 1) + (k & 1)];
    };
    std::integral_constant<int, LIN1_QK> SQK;
    std::integral_constant<int, LIN1_V> SV;
    std::integral_constant<int, LIN1_MLP> SMLP;

    // The wave's tokens as B fragments of all k-steps (X is padded to whole tiles): whole 128-byte lines per 8 lanes (8 rows x 128 B per
    // instruction: 8 cache lines, where a fragment-shaped load of one k-step touches 32 - the texture-address unit then needs ~ 58 cycles
    // per instruction, 15 000 cycles for a workgroup's 256: measured), then - finish_x - line by line through the wave's staging image into
    // fragment order, in place: line j of every row holds the k-steps 4 j .. 4 j + 3; chunk c of row t sits at t 128 + 16 (c ^ ((t >> 1) & 7)),
    // conflict-free for both accesses.
    constexpr bool XLINES = LIN1_XLOAD != 0 && (K <= 384 || LIN1_XLOAD >= 2);  // (K = 512: 7 more spilled registers for - 1.7 %: not taken)
    auto load_x = [&](int nw) __attribute__((always_inline)) {
        if (!XLINES) {
            const u16 *xr = g.X + (size_t)(nw + r) * K + 8 * hf;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) xreg[ks] = as_bf16x8(*reinterpret_cast<const u32x4 *>(xr + 16 * ks));
        } else {
            const u16 *xr = g.X + (size_t)(nw + (lane >> 3)) * K + 8 * (lane & 7);
#pragma unroll
            for (int j = 0; j < KS / 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) xreg[4 * j + q] = as_bf16x8(*reinterpret_cast<const u32x4 *>(xr + (size_t)(8 * q) * K + 64 * j));
        }
    };
    auto finish_x = [&]() __attribute__((always_inline)) {
        if (!XLINES) return;
        const unsigned t0 = lane >> 3;  // row of instruction q: t0 + 8 q, (row >> 1) & 7 = ((t0 >> 1) + 4 q) & 7
        const unsigned xw = (unsigned)(size_t)(LDS_PTR(char))(stage) + t0 * 128;
        const unsigned xr0 = (unsigned)(size_t)(LDS_PTR(char))(stage) + r * 128;
#pragma unroll
        for (int j = 0; j < KS / 4; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<LDS_PTR(u32x4)>(xw + 1024 * q + ((((lane & 7) ^ ((t0 >> 1) + 4 * q)) & 7) << 4)) = as_u32x4(xreg[4 * j + q]);
#pragma unroll
            for (int m = 0; m < 4; ++m)
                xreg[4 * j + m] = as_bf16x8(*reinterpret_cast<const LDS_PTR(u32x4)>(xr0 + ((((2 * m + hf) ^ (r >> 1)) & 7) << 4)));
        }
    };

    // ---- LNF: the wave's 32 rows of the fp32 residual stream -> LayerNorm + modulate -> bf16 B fragments, in one pass: K / 64 passes of one
    // bf16 line (64 columns = 4 k-steps), 8 line-shaped loads (8 rows x 128 B each) and 32 registers of rows at a time; the rows' statistics
    // come from the producer of h (Lin1Args::ln_stat), the modulation rows of the tile's trajectories from an LDS table.  The same arithmetic
    // per element as k_ln_modulate (one rounding to bf16); the statistics are combined in another order, so the operand can differ from the
    // default path's by an ulp of bf16 here and there: the form is a property of the model handle, never of the launch.
    float *const ln_tab = bias_lds + ((g.F + 3) & ~3);  // [LN_SLOTS][2][K]: 1 + scale, shift
    auto traj_of = [&](unsigned n) __attribute__((always_inline)) { return g.ln_tpt_magic ? __umulhi(n, g.ln_tpt_magic) : n; };
    auto load_x_lnf = [&](int tile_, int nw) __attribute__((always_inline)) {
        typedef __attribute__((ext_vector_type(4))) float f4;
        const float *xf = reinterpret_cast<const float *>(g.X);
        // the modulation rows of the tile's trajectories (uniform: first trajectory of the tile, slots clamped to the pass's last trajectory)
        const unsigned t_first = g.ln_mod_stride ? traj_of((unsigned)(tile_ * C::TT)) : 0u;
        const unsigned t_last = g.ln_mod_stride ? traj_of((unsigned)(g.N - 1)) : 0u;
        for (int i = tid * 4; i < C::LN_SLOTS * K; i += NW * 64 * 4) {
            const int slot = i / K, col = i - slot * K;
            const size_t mo = (size_t)min(t_first + (unsigned)slot, t_last) * g.ln_mod_stride + col;
            const f4 sc = *reinterpret_cast<const f4 *>(g.ln_scale + mo), sh = *reinterpret_cast<const f4 *>(g.ln_shift + mo);
            *reinterpret_cast<f4 *>(ln_tab + (size_t)slot * 2 * K + col) = f4{1.0f + sc[0], 1.0f + sc[1], 1.0f + sc[2], 1.0f + sc[3]};
            *reinterpret_cast<f4 *>(ln_tab + (size_t)slot * 2 * K + K + col) = sh;
        }
        const unsigned rowi = lane >> 3, chunk = lane & 7;
        const unsigned sbase = (unsigned)(size_t)(LDS_PTR(char))(stage);
        const unsigned xr0 = sbase + r * 128;
        const float *xr = xf + (size_t)(nw + rowi) * K + 4 * chunk;
        float2 st[4];        // (rstd, -mean rstd) of rows rowi + 8 q
        const float *tb[4];  // their trajectory's table row, at the lane's first column
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned nn = (unsigned)min(nw + (int)rowi + 8 * q, g.N - 1);
            st[q] = g.ln_stat[nn];
            const unsigned slot = g.ln_mod_stride ? min(traj_of(nn) - t_first, (unsigned)(C::LN_SLOTS - 1)) : 0u;
            tb[q] = ln_tab + (size_t)slot * 2 * K + 4 * chunk;
        }
        __syncthreads();  // the table is complete (every wave left the previous segment's long ago: a segment has at least one block barrier)
#pragma unroll
        for (int p = 0; p < K / 64; ++p) {
            f4 v[2][4];
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int q = 0; q < 4; ++q) v[e][q] = *reinterpret_cast<const f4 *>(xr + (size_t)(8 * q) * K + 64 * p + 32 * e);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned row = rowi + 8 * q, sw = (row >> 1) & 7;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const f4 sc = *reinterpret_cast<const f4 *>(tb[q] + 64 * p + 32 * e), sh = *reinterpret_cast<const f4 *>(tb[q] + K + 64 * p + 32 * e);
                    f4 x = v[e][q];
#pragma unroll
                    for (int c = 0; c < 4; ++c) x[c] = fmaf(x[c], st[q].x, st[q].y) * sc[c] + sh[c];
                    const u32x2 pk = {pack2(x[0], x[1]), pack2(x[2], x[3])};
                    *reinterpret_cast<LDS_PTR(u32x2)>(sbase + row * 128 + ((((4 * e + (chunk >> 1)) ^ sw) & 7) << 4) + 8 * (chunk & 1)) = pk;
                }
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
                xreg[4 * p + m] = as_bf16x8(*reinterpret_cast<const LDS_PTR(u32x4)>(xr0 + ((((2 * m + hf) ^ (r >> 1)) & 7) << 4)));
            __builtin_amdgcn_sched_barrier(0);  // (one pass's rows at a time)
        }
    };
    const int qb = g.HHD >> 5;  // blocks per q / k / v section
    // (tile, first block, blocks left in the range): no division inside the loop - a later segment always starts a tile at block 0
    int tile = (int)(i0 / NB), b0 = (int)(i0 - (long)tile * NB), left = (int)(i1 - i0);
    bool first_seg = true;
    bool co_ready = false;  // the rotation table of this segment's first section was requested at the end of the previous segment
    __syncthreads();  // bias vector in LDS
    while (left > 0) {  // one segment = blocks [b0, b1) of one token tile; b0, b1 even
        const int b1 = NB - b0 < left ? NB : b0 + left;
        n_wave = tile * C::TT + wave * 32;
        row_q = reinterpret_cast<const char *>(g.qkv) + (size_t)n_wave * stride_q;
        row_z = reinterpret_cast<const char *>(g.z) + (size_t)n_wave * stride_z - 4 * (size_t)g.HHD;  // (z column of feature f: f - 2 HHD)
        // the wave's tokens: B fragments of all k-steps (X is padded to whole tiles)
        if constexpr (LNF) load_x_lnf(tile, n_wave);  // (a prefetch of the next tile's rows at the end of a segment - 32 to 128 registers - was tried:
        else {                                          // hipcc spills it, 256-640 B of scratch at K = 512)
            if (first_seg) load_x(n_wave);  // (later segments: requested at the end of the previous one)
            finish_x();
        }
        const unsigned nn = (unsigned)min(n_wave + r, g.N - 1);
        const unsigned n1 = g.div_magic ? __umulhi(nn, g.div_magic) : nn;
        const unsigned pos = g.mod_magic ? n1 - __umulhi(n1, g.mod_magic) * (unsigned)g.pos_mod : 0u;

        // first block of the segment: MFMAs only.  The activation loads sit in the vector-memory queue behind the ring's DMA requests, so the
        // per-step counted wait does not apply here.  A later segment's loads were issued BEFORE the previous segment's last four slab stores
        // (see the end of the loop): vmcnt(4) covers the loads and every ring request, and leaves exactly those stores in flight - their
        // acknowledgements take ~ 10 000 cycles under this kernel's write stream and used to be waited for at every tile boundary
        // (profiles/r03_experiments.txt).  Then the barrier: a wave may read a block only once every wave's pieces of it have landed.
        if (first_seg) wait_vmcnt<0>();
        else wait_vmcnt<4>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue(dma_blk, dma_slot);
        advance(dma_blk);
        dma_slot = next_slot(dma_slot);
        if (B2) {  // (the pair's second block too)
            issue(dma_blk, dma_slot);
            advance(dma_blk);
            dma_slot = next_slot(dma_slot);
        }
        init_acc(acc0, b0);
        step(SV, I1, I1, I0, no_side);
        // the segment's first fused step (e = b0, even): nothing to flush yet
        const int e_end = b1 - 1;
        if (b0 < 2 * qb) {
            if (!co_ready) load_co(b0 < qb ? g.rope_q : g.rope_k, pos);
            post = b0 < qb ? g.q_premul : 1.0f;
            fused(SQK, I0, I0, I0, b0);
        } else if (b0 < 3 * qb) {
            fused(SV, I0, I0, I0, b0);
        } else {
            fused(SMLP, I0, I0, I0, b0);
        }
        // fused steps e = b0 + 1 .. b1 - 2, split by the section of e
        {
            const int lo = max(b0 + 1, 0), hi = min(e_end, qb);
            if (lo < hi) run(SQK, lo, hi);  // (co holds the q table: the segment started inside the q section)
        }
        {
            const int lo = max(b0 + 1, qb), hi = min(e_end, 2 * qb);
            if (lo < hi) {
                if (lo == qb) {  // entering the k section (otherwise the segment started inside it and co holds the k table)
                    load_co(g.rope_k, pos);
                    post = 1.0f;
                }
                run(SQK, lo, hi);
            }
        }
        {
            const int lo = max(b0 + 1, 2 * qb), hi = min(e_end, 3 * qb);
            if (lo < hi) run(SV, lo, hi);
        }
        {
            const int lo = max(b0 + 1, 3 * qb), hi = min(e_end, NB);
            if (lo < hi) run(SMLP, lo, hi);
        }
        // last block of the segment (odd): epilogue only, then its slab.  The activation registers are free from here on: the NEXT segment's
        // rows are requested first, so that they are older than the slab's stores (which the next segment then does not wait for).  These four
        // stores are ordinary (compiler-visible) stores: hipcc then knows they are younger than the loads and waits with vmcnt(4), not vmcnt(0),
        // where it consumes the loaded registers.
        const int left_next = left - (b1 - b0);
        co_ready = false;
        if (left_next > 0) {  // (uniform) the next segment: tile + 1 from block 0, i.e. the q section
            asm volatile("" ::: "memory");
            const int nw = (tile + 1) * C::TT + wave * 32;
            if constexpr (!LNF) load_x(nw);
            if (b1 - 1 >= 2 * qb) {  // the drain below is not a q / k block: co is free for the next segment's q table
                const unsigned nn2 = (unsigned)min(nw + r, g.N - 1);
                const unsigned m1 = g.div_magic ? __umulhi(nn2, g.div_magic) : nn2;
                load_co(g.rope_q, g.mod_magic ? m1 - __umulhi(m1, g.mod_magic) * (unsigned)g.pos_mod : 0u);
                co_ready = true;
            }
            asm volatile("" ::: "memory");
        }
        {
            const int e = b1 - 1;
            if (e < 2 * qb) {
                step(SQK, I1, I0, I1, no_side);  // (co is current: e is odd, so its section was entered by an earlier step of this segment)
            } else if (e < 3 * qb) {
                step(SV, I1, I0, I1, no_side);
            } else {
                step(SMLP, I1, I0, I1, no_side);
            }
            flush_setup(e - 1);
            u32x4 pk[2];
            flush_read(0, pk);
            flush_store_visible(0, pk);
            flush_read(1, pk);
            flush_store_visible(1, pk);
        }
        left = left_next;
        tile += 1;
        b0 = 0;
        first_seg = false;
    }
    wait_vmcnt<0>();  // the ring's run-ahead requests must not land in LDS after the workgroup has gone
}
