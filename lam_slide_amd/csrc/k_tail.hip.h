// The back half of a ParallelMLPAttentionV2 sub-block (mmdit.py:240-249: gelu(mlp), linear2 over [attention | gelu(mlp)]) together with the
// gated residual update and the NEXT sub-block's LayerNorm + modulate (latent_si_v31.py:45-63) as ONE token-stationary, row-owning kernel:
//
//     u      = a W1m^T + b1m                      (the mlp rows of linear1: M features)
//     out    = z_attn Wo^T + gelu(u) W2m^T + b2   (linear2 = [Wo | W2m] over [attention | gelu(mlp)])
//     h     += gate * out
//     a_next = bf16(LN_1e-6(h) (1 + scale') + shift')
//
// Why (round 6): with linear1 -> z -> linear2 -> LayerNorm as separate kernels a token and sub-block moves 20 KB through HBM at cfg 2 (12.3 KB at
// NBA); the GELU'd mlp half of z (2 M bytes per token) is written only to be read once, linear2 is weight-stationary (a row of h is split
// over four CUs, nothing row-local can be fused into it) and LayerNorm re-reads the rows linear2 has just written.  Here a wave OWNS 32 token
// rows for a whole tile: the mlp activations never leave the registers, h is read and written once, and the row statistics are wave-local.
//
// Structure (the token-stationary form of k_lin1.hip.h, with a second GEMM fed from the accumulators of the first):
//   * a workgroup = 8 waves x 32 tokens (two waves per SIMD, hidden <= 256); a wave keeps the whole output tile out^T[D features][32 tokens]
//     (D / 2 accumulator VGPRs) and the first half of its tokens' activations `a` as MFMA B fragments in registers (D / 8 VGPRs); the other
//     half lives as lane-linear 1 KiB fragments in 8 KiB of wave-private LDS (whose first 4 KiB double as the staging image outside the
//     mlp phase): the registers that frees hold a second accumulator tile and a second packed fragment - the software pipeline below;
//   * the WEIGHTS stream through LDS as ONE linear sequence of 1 KiB MFMA A fragments, packed once per call in exactly the order the kernel
//     consumes them (k_tail_pack): chunks of D / 16 fragments = [Wo k-steps 2c, 2c+1 x all D/32 output tiles] for the attention half,
//     then per mlp block j of 32 features [W1m block j: D/16 k-steps] and [W2m columns of block j: D/32 output tiles x 2 k-steps], in the
//     order U(0) U(1) D(0) U(2) D(1) ... U(MB-1) D(MB-2) D(MB-1).  A ring of 4 chunks = two PAIRS filled by LDS-DMA one pair ahead, one wait
//     + one workgroup barrier per pair (at one per chunk the barrier's drain left the matrix pipe idle for a third of a step); the
//     sequence is cyclic, so the ring runs through tile boundaries;
//   * up-projection U(j): D/16 MFMAs into one 32 x 32 accumulator tile (bias as the initial value; blocks alternate between two tiles); its
//     16 values per lane, erf-GELU'd and rounded to bf16, ARE the B fragments of the down-projection D(j) - accumulator registers 8 s ..
//     8 s + 7 of a lane are k-step s - with W2m's columns permuted to the accumulator's row order by the packer (the P-from-accumulator
//     form of k_attn.hip.h).  The GELU of block j runs in the MFMA shadows of THIS wave's D(j - 1) and U(j + 1), one pair of values behind
//     every fourth MFMA: a wave's vector instructions overlap only with its own MFMAs - beside its SIMD partner's MFMA chain they advance
//     at 9 % of their rate (tools/microbench_coissue.hip, profiles/r06_experiments.txt section 5);
//   * epilogue per tile and wave, through the 4 KiB staging image: the h rows arrive row-wise (whole 128-byte segments, 8 rows per
//     instruction), are re-read in the accumulator layout, updated, summed; the updated rows leave the same way; then the second pass over
//     the registers writes a_next.  Amortised over >= 1 000 MFMAs per wave.
//
// Numerics: out = the k-ascending chain over [attention | mlp] (mlp features inside a block in the accumulator's row order), + bias, fma with the
// gate onto the residual like EpiLinear2 (k_gemm.hip.h); row statistics two-pass in fp32 like k_ln_modulate_v4 with another summation tree.
// Not bit-identical to the linear2 / LayerNorm kernels it replaces (same numerics class: fp32 accumulation of bf16 products).
#pragma once
#include <type_traits>

#include "common.hip.h"

struct TailArgs {
    const u16 *wt;       // packed weight stream of this sub-block (k_tail_pack)
    const u16 *A;        // [N rounded up to 256][D] bf16: this sub-block's LayerNorm + modulate (linear1's input)
    const u16 *Z;        // [N rounded up to 256][zw] bf16: attention output in columns [0, HHD)
    const float *b1;     // [M]: linear1 bias, mlp section
    const float *b2;     // [D]
    const float *gate;   // mods + gate offset, row stride mod_stride (0: one row shared by every trajectory)
    float *h;            // [N rounded up to 256][D] fp32 residual stream, rows [0, N) updated in place
    u16 *a_next;         // [N rounded up to 256][D] bf16 or NULL (last sub-block: the head normalises itself); may alias A
    const float *ln_shift, *ln_scale;  // next sub-block's modulation rows (row stride mod_stride)
    int N, M, zw;
    int mod_stride, tpt;  // tokens per trajectory
    unsigned tpt_magic;   // floor(2^32 / tpt) + 1 (0 when tpt == 1)
};

template <int D, int HHD>
struct TailCfg {
    static_assert(D % 64 == 0 && D <= 256 && HHD % 64 == 0, "hidden sizes 64 .. 256 (the output tile of a wave is D / 2 accumulator registers)");
    static constexpr int NW = 8;            // waves per workgroup: two per SIMD
    static constexpr int NT = D / 32;       // output tiles of 32 features
    static constexpr int KS = D / 16;       // k-steps of the up-projection
    static constexpr int KR = KS / 2;       // ... of which the first KR stay in registers as B fragments, the rest in the wave's LDS image
    static constexpr int KZ = HHD / 16;     // k-steps of the attention half
    static constexpr int CHF = D / 16;      // fragments per chunk
    static constexpr int CH = CHF * 1024;   // bytes per chunk
    static constexpr int CO = HHD / 32;     // chunks of the attention half
    static constexpr int NS = 4;            // ring slots = two PAIRS of chunks: one pair in use, the next in flight
    static constexpr int PPW = CHF / NW;    // DMA instructions per wave and chunk
    static_assert(CHF % NW == 0, "whole DMA instructions per wave");
    static constexpr int AIMG = (KS - KR) * 1024 < 4096 ? 4096 : (KS - KR) * 1024;  // per wave: activation fragments KR .. KS - 1; its first 4 KiB double as the staging image
    static constexpr int RING = NS * CH, WAVE = NW * AIMG;
    static constexpr size_t lds_bytes(int M) { return (size_t)RING + WAVE + (size_t)M * 4; }
    static size_t stream_bytes(int M) { return (size_t)(CO + 2 * (M / 32)) * CH; }
};

// Weight stream of one sub-block.  16-byte piece i = lane (r = lane & 31, hf = lane >> 5) of fragment f of chunk c:
//   c < CO                      Wo:  fragment f = 2 ft + s -> W2[32 ft + r][32 c + 16 s + 8 hf + 0..7]
//   then e = c - CO: e = 0 U(0); odd e < 2 MB - 1: U((e + 1) / 2); even e: D(e / 2 - 1); e = 2 MB - 1: D(MB - 1)
//   U(j): fragment ks -> W1[3 HHD + 32 j + r][16 ks + 8 hf + 0..7]
//   D(j): fragment f = 2 ft + s -> W2[32 ft + r][HHD + 32 j + phi(s, hf, 0..7)],  phi = 16 s + 8 (i >> 2) + 4 hf + (i & 3):
//         the mlp feature whose GELU sits in accumulator register 8 s + i of a lane of half hf (common.hip.h: mfma32 C/D map)
__global__ void __launch_bounds__(256) k_tail_pack(u16 *out, const u16 *W1, const u16 *W2, int D, int HHD, int M) {
    const int CHF = D / 16, CO = HHD / 32, MB = M / 32, K2 = HHD + M;
    const long total = (long)(CO + 2 * MB) * CHF * 64;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int lane = (int)(i & 63), r = lane & 31, hf = lane >> 5;
        long t = i >> 6;
        const int f = (int)(t % CHF), c = (int)(t / CHF);
        u32x4 v;
        if (c < CO) {
            const int ft = f >> 1, s = f & 1;
            v = *reinterpret_cast<const u32x4 *>(W2 + (size_t)(32 * ft + r) * K2 + 32 * c + 16 * s + 8 * hf);
        } else {
            const int e = c - CO;
            const bool up = e == 0 || ((e & 1) && e < 2 * MB - 1);
            if (up) {
                const int j = e == 0 ? 0 : (e + 1) >> 1;
                v = *reinterpret_cast<const u32x4 *>(W1 + (size_t)(3 * HHD + 32 * j + r) * D + 16 * f + 8 * hf);
            } else {
                const int j = e == 2 * MB - 1 ? MB - 1 : (e >> 1) - 1;
                const int ft = f >> 1, s = f & 1;
                const u16 *src = W2 + (size_t)(32 * ft + r) * K2 + HHD + 32 * j + 16 * s + 4 * hf;
                const u32x2 lo = *reinterpret_cast<const u32x2 *>(src), hi = *reinterpret_cast<const u32x2 *>(src + 8);
                v = u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
        }
        *reinterpret_cast<u32x4 *>(out + i * 8) = v;
    }
}

template <int D, int HHD>
__global__ void __launch_bounds__(512, 2) k_tail(TailArgs g) {
    using C = TailCfg<D, HHD>;
    constexpr int NW = C::NW, NT = C::NT, KS = C::KS, KR = C::KR, KZ = C::KZ, CHF = C::CHF, CH = C::CH, CO = C::CO, NS = C::NS, PPW = C::PPW;
    constexpr int PD = 2;  // A fragments requested this many MFMAs ahead
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char *const aimg = smem + C::RING + wave * C::AIMG;  // this wave's activation fragments KR .. KS - 1; [0, 4 KiB) = its staging image outside the mlp phase
    float *const b1_lds = reinterpret_cast<float *>(smem + C::RING + C::WAVE);

    // Work = wave tiles of 32 tokens, cut evenly over the workgroups; a workgroup walks its range in rounds of NW wave tiles (one per wave).  In
    // a last, partial round the waves without a tile only keep the ring going.
    const int MB = g.M >> 5, NPAIR = (CO + 2 * MB) >> 1;  // (M is a multiple of 64: MB even)
    const int nwt = (g.N + 31) >> 5;
    const int w0 = (int)((long)nwt * blockIdx.x / gridDim.x), w1 = (int)((long)nwt * (blockIdx.x + 1) / gridDim.x);
    if (w0 >= w1) return;  // (uniform)
    const int rounds = (w1 - w0 + NW - 1) / NW;

    for (int i = tid * 4; i < g.M; i += NW * 64 * 4) *reinterpret_cast<float4 *>(b1_lds + i) = *reinterpret_cast<const float4 *>(g.b1 + i);

    // Everything a phase needs per lane (row / chunk indices, staging addresses, row pointers, fragment bases) is derived from a lane id that
    // is laundered INSIDE that phase: hipcc cannot hoist those values out of the round loop, where - with the output tile, the activations and
    // two accumulator tiles resident - they would live in scratch, and a scratch reload inside the chunk loop is a vector-memory operation
    // whose wait also waits for the ring requests just issued.
    auto fresh_lane = [&]() __attribute__((always_inline)) {
        int l = lane;
        asm volatile("" : "+v"(l));
        return l;
    };
    // ---- weight ring: the stream's chunk k -> slot k mod 4; wave w requests fragments PPW w .. PPW w + PPW - 1 of a chunk, one LDS-DMA
    // instruction each (inline asm on purpose, k_lin1.hip.h: behind the builtin hipcc waits for the request in front of the next LDS access)
    const unsigned lds0 = (unsigned)(size_t)(LDS_PTR(char))(smem);
    const char *const w_base = reinterpret_cast<const char *>(g.wt) + (size_t)wave * PPW * 1024;
    auto issue_piece = [&](const char *src, unsigned dst, auto ic) __attribute__((always_inline)) {
        constexpr int I = decltype(ic)::value;
        const unsigned ls = (unsigned)fresh_lane() * 16u;
        asm volatile("s_add_u32 m0, %2, 0\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" ::"v"(ls), "s"(src), "s"(dst), "n"(1024 * I) : "memory", "scc");
    };
    int p_src = 0, slot_d = 0;  // next PAIR of the stream to request (index in [0, NPAIR)), and its first slot (0 or 2)
    const char *d_src = nullptr;  // the pair being requested: its 2 PPW pieces are issued one at a time INSIDE the MFMA chain of the pair's
    unsigned d_dst = 0;           // first chunk (an LDS-DMA instruction costs the wave ~ 100 cycles of issue: k_lin1.hip.h "side jobs")
    auto pair_begin = [&]() __attribute__((always_inline)) {
        d_src = w_base + (size_t)p_src * (2 * CH);
        d_dst = lds0 + slot_d * CH + wave * PPW * 1024;
        p_src = p_src + 1 == NPAIR ? 0 : p_src + 1;
        slot_d ^= 2;
    };
    static_assert(PPW >= 1 && PPW <= 2 && CHF >= 8, "2 PPW pieces per wave and pair, behind MFMAs 1, 3, 5, 7 of a chunk");
    auto issue_q = [&](auto qc) __attribute__((always_inline)) {  // piece q of the pair: chunk q / PPW, fragment q % PPW of the wave's PPW
        constexpr int Q = decltype(qc)::value;
        if constexpr (Q < 2 * PPW) issue_piece(d_src + (Q / PPW) * CH, d_dst + (Q / PPW) * CH, std::integral_constant<int, Q % PPW>());
    };
    auto issue_pair = [&]() __attribute__((always_inline)) {
        pair_begin();
        issue_q(std::integral_constant<int, 0>());
        issue_q(std::integral_constant<int, 1>());
        issue_q(std::integral_constant<int, 2>());
        issue_q(std::integral_constant<int, 3>());
    };
    // behind MFMA f of the pair's first chunk (DMA = 1): one piece behind MFMAs 1, 3, 5, 7
    auto dma_behind = [&](int f, auto dma_c) __attribute__((always_inline)) {
        if constexpr (decltype(dma_c)::value != 0) {
            if (f == 1) issue_q(std::integral_constant<int, 0>());
            if (f == 3) issue_q(std::integral_constant<int, 1>());
            if (f == 5) issue_q(std::integral_constant<int, 2>());
            if (f == 7) issue_q(std::integral_constant<int, 3>());
        }
    };
    int slot_c = 0;  // slot of the next chunk this wave computes
    // Head of a PAIR of chunks: one wait + one workgroup barrier per 2 CHF MFMAs of a wave.  Both chunks were requested a whole pair ago (EXTRA =
    // vector-memory operations the wave has issued since, which may stay in flight); every wave has left the previous pair, whose two slots
    // take the next pair's requests.
    auto pair_head = [&](auto extra_c) __attribute__((always_inline)) {  // (the requests follow inside the first chunk: dma_behind)
        wait_vmcnt<decltype(extra_c)::value>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        pair_begin();
    };
    auto next_slot = [&]() __attribute__((always_inline)) { slot_c = (slot_c + 1) & (NS - 1); };
    auto frag = [&](const char *sb, int f) __attribute__((always_inline)) { return as_bf16x8(*reinterpret_cast<const u32x4 *>(sb + f * 1024)); };

    std::integral_constant<int, 0> E0;
    issue_pair();
    __syncthreads();  // bias table

    const unsigned st0 = (unsigned)(size_t)(LDS_PTR(char))(aimg);
    const unsigned b1_base = (unsigned)(size_t)(LDS_PTR(char))(smem + C::RING + C::WAVE);  // (uniform)
    // rows of a [tokens][K] bf16 matrix as MFMA B fragments (k_lin1.hip.h load_x / finish_x): whole 128-byte lines per 8 lanes (8 rows per
    // instruction), then line by line through the wave's staging image into fragment order: line j of every row holds the k-steps
    // 4 j .. 4 j + 3; chunk c of row t sits at t 128 + 16 (c ^ ((t >> 1) & 7)), conflict-free for both accesses
    auto load_rows = [&](auto &xreg, auto ks_c, const u16 *X, int n0, int stride) __attribute__((always_inline)) {
        constexpr int NK = decltype(ks_c)::value;
        const int l = fresh_lane(), chunk = l & 7, rowi = l >> 3;
        const u16 *xr = X + (size_t)(n0 + rowi) * stride + 8 * chunk;
#pragma unroll
        for (int j = 0; j < NK / 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) xreg[4 * j + q] = as_bf16x8(*reinterpret_cast<const u32x4 *>(xr + (size_t)(8 * q) * stride + 64 * j));
    };
    auto finish_line = [&](bf16x8 (&x4)[4]) __attribute__((always_inline)) {  // one line (4 row-wise registers) -> its 4 fragments, in place
        const int l = fresh_lane(), chunk = l & 7, rowi = l >> 3, r = l & 31, hf = l >> 5;
        const unsigned xw = st0 + rowi * 128, xr0 = st0 + r * 128;
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<LDS_PTR(u32x4)>(xw + 1024 * q + (((chunk ^ ((rowi >> 1) + 4 * q)) & 7) << 4)) = as_u32x4(x4[q]);
#pragma unroll
        for (int m = 0; m < 4; ++m) x4[m] = as_bf16x8(*reinterpret_cast<const LDS_PTR(u32x4)>(xr0 + ((((2 * m + hf) ^ (r >> 1)) & 7) << 4)));
    };
    constexpr int NPF = NT < 4 ? NT : 4;  // feature tiles of h in flight per wave
    auto load_h = [&](f32x4_t (&v)[4], unsigned hoff, int ft) __attribute__((always_inline)) {  // rows (lane >> 3) + 8 i, 128 bytes of feature tile ft
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4_t *>(reinterpret_cast<const char *>(g.h) + (size_t)(hoff + (unsigned)(8 * i * 4 * D + 128 * ft)));
    };

    for (int rd = 0; rd < rounds; ++rd) {
        const int wt = w0 + rd * NW + wave;
        if (wt >= w1) {  // (uniform) no tile for this wave in the last round: pass the round's barriers, keep requesting
            for (int c = 0; c < NPAIR; ++c) {
                pair_head(E0);
                issue_q(std::integral_constant<int, 0>());
                issue_q(std::integral_constant<int, 1>());
                issue_q(std::integral_constant<int, 2>());
                issue_q(std::integral_constant<int, 3>());
            }
            continue;
        }
        const int n_wave = wt * 32;
        f32x16 out[NT];
#pragma unroll
        for (int ft = 0; ft < NT; ++ft)
#pragma unroll
            for (int i = 0; i < 16; ++i) out[ft][i] = 0.0f;

        // ---- attention half: out^T += Wo z^T ----
        {
            bf16x8 zreg[KZ];
            load_rows(zreg, std::integral_constant<int, KZ>(), g.Z, n_wave, g.zw);
#pragma unroll
            for (int j = 0; j < KZ / 4; ++j) finish_line(*reinterpret_cast<bf16x8(*)[4]>(&zreg[4 * j]));
#pragma unroll
            for (int c = 0; c < CO; ++c) {
                if ((c & 1) == 0) pair_head(E0);
                const char *sb = smem + slot_c * CH + fresh_lane() * 16;
                bf16x8 fr[PD];
#pragma unroll
                for (int f = 0; f < PD; ++f) fr[f] = frag(sb, f);
#pragma unroll
                for (int f = 0; f < CHF; ++f) {
                    out[f >> 1] = mfma32(fr[f % PD], zreg[2 * c + (f & 1)], out[f >> 1]);
                    if (f + PD < CHF) fr[f % PD] = frag(sb, f + PD);
                    if ((c & 1) == 0) dma_behind(f, std::integral_constant<int, 1>());
                }
                next_slot();
                __builtin_amdgcn_sched_barrier(0);  // (one chunk's fragments and temporaries live at a time)
            }
        }

        // ---- activations: k-steps 0 .. KR - 1 as B fragments in registers, KR .. KS - 1 as lane-linear 1 KiB fragments in the wave's LDS image.
        // The staging image IS the first 4 KiB of that image: the register lines go first, then the upper LDS line(s), the line that lands on
        // the staging image itself last (its fragments are in registers when they overwrite it).
        bf16x8 areg[KR];
        {
            bf16x8 raw[KS];
            load_rows(raw, std::integral_constant<int, KS>(), g.A, n_wave, D);
#pragma unroll
            for (int j = 0; j < KR / 4; ++j) {
                finish_line(*reinterpret_cast<bf16x8(*)[4]>(&raw[4 * j]));
#pragma unroll
                for (int m = 0; m < 4; ++m) areg[4 * j + m] = raw[4 * j + m];
            }
            const unsigned fw = st0 + (unsigned)fresh_lane() * 16u;
#pragma unroll
            for (int j = KS / 4 - 1; j >= KR / 4; --j) {
                finish_line(*reinterpret_cast<bf16x8(*)[4]>(&raw[4 * j]));
#pragma unroll
                for (int m = 0; m < 4; ++m) *reinterpret_cast<LDS_PTR(u32x4)>(fw + 1024 * (4 * j + m - KR)) = as_u32x4(raw[4 * j + m]);
            }
        }

        // ---- mlp, software-pipelined inside the wave: block j accumulates in up0 (even j) / up1 (odd j); its GELU runs in the MFMA shadows of
        // D(j - 1) and U(j + 1) - a wave's vector instructions overlap only with its OWN MFMAs (its SIMD partner's do not: profiles/
        // r06_experiments.txt section 5) - one pair of values behind every STRIDE-th MFMA, into gwE (even j) / gwO (odd j).
        f32x16 up0, up1;
        u32x4 gwE[2], gwO[2];
        auto gelu_behind = [&](int i, int stride, int first_word, const f32x16 &ue, u32x4 (&gd)[2]) __attribute__((always_inline)) {
            if (i % stride == stride - 1) {
                const int s = first_word + i / stride;  // packed word s = accumulator registers 2 s, 2 s + 1
                gd[s >> 2][s & 3] = gelu_pair_bf16(ue[2 * s], ue[2 * s + 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // U(j): the chain of block j from its bias into `uc`; behind it the GELU words [first_word, first_word + CHF / stride) of `ue` -> gd
        auto step_up = [&](f32x16 &uc, int j, int stride, int first_word, const f32x16 &ue, u32x4 (&gd)[2], auto gelu_c, auto dma_c) __attribute__((always_inline)) {
            constexpr bool DO_GELU = decltype(gelu_c)::value != 0;
            {
                unsigned ba = b1_base + 16u * (unsigned)(fresh_lane() >> 5) + 128u * (unsigned)j;
                asm volatile("" : "+v"(ba));  // (one per-lane base + immediates; no strength-reduced running pointer)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const f32x4_t b = *reinterpret_cast<const LDS_PTR(f32x4_t)>(ba + 32 * q4);
                    uc[4 * q4] = b[0]; uc[4 * q4 + 1] = b[1]; uc[4 * q4 + 2] = b[2]; uc[4 * q4 + 3] = b[3];
                }
            }
            const int l16 = fresh_lane() * 16;
            const char *sb = smem + slot_c * CH + l16;
            const char *ab = aimg + l16;
            bf16x8 fr[PD], bfr[PD];
#pragma unroll
            for (int f = 0; f < PD; ++f) fr[f] = frag(sb, f);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + PD == KR || (ks == 0 && KR < PD)) {  // the first LDS-resident activation fragments: requested PD MFMAs ahead
#pragma unroll
                    for (int f = 0; f < PD; ++f) bfr[f] = frag(ab, f);
                }
                if (ks < KR) uc = mfma32(fr[ks % PD], areg[ks], uc);
                else {
                    uc = mfma32(fr[ks % PD], bfr[(ks - KR) % PD], uc);
                    if (ks + PD < KS) bfr[(ks - KR) % PD] = frag(ab, ks - KR + PD);
                }
                if (ks + PD < KS) fr[ks % PD] = frag(sb, ks + PD);
                dma_behind(ks, dma_c);
                if (DO_GELU) gelu_behind(ks, stride, first_word, ue, gd);
            }
            next_slot();
            __builtin_amdgcn_sched_barrier(0);
        };
        // D(j): out^T += W2m(block j) gelu(u_j)^T with the packed words of `gw` as B fragments; behind it GELU words of `ue` -> gd
        auto step_down = [&](const u32x4 (&gw)[2], int stride, int first_word, const f32x16 &ue, u32x4 (&gd)[2], auto gelu_c, auto dma_c) __attribute__((always_inline)) {
            constexpr bool DO_GELU = decltype(gelu_c)::value != 0;
            const char *sb = smem + slot_c * CH + fresh_lane() * 16;
            bf16x8 fr[PD];
#pragma unroll
            for (int f = 0; f < PD; ++f) fr[f] = frag(sb, f);
#pragma unroll
            for (int f = 0; f < CHF; ++f) {
                out[f >> 1] = mfma32(fr[f % PD], as_bf16x8(gw[f & 1]), out[f >> 1]);
                if (f + PD < CHF) fr[f % PD] = frag(sb, f + PD);
                dma_behind(f, dma_c);
                if (DO_GELU) gelu_behind(f, stride, first_word, ue, gd);
            }
            next_slot();
            __builtin_amdgcn_sched_barrier(0);
        };
        std::integral_constant<int, 0> I0;
        std::integral_constant<int, 1> I1;
        constexpr int S8 = CHF / 8 > 0 ? CHF / 8 : 1, S4 = CHF / 4;  // one pair behind every S8-th MFMA: 8 words per chunk; every S4-th: 4 words
        // pair (U0, U1): GELU(0) whole behind U(1)
        pair_head(E0);
        step_up(up0, 0, 1, 0, up1, gwE, I0, I1);
        step_up(up1, 1, S8, 0, up0, gwE, I1, I0);
        // pairs (D(j - 1), U(j + 1)), j = 1 .. MB - 2: GELU(j) behind both, half each
        for (int j = 1; j + 1 <= MB - 2; j += 2) {  // (MB is even: host-checked - no parity branches, whose merged register state hipcc spills)
            pair_head(E0);
            step_down(gwE, S4, 0, up1, gwO, I1, I1);     // D(j - 1) | GELU(j) words 0-3      (j odd)
            step_up(up0, j + 1, S4, 4, up1, gwO, I1, I0);  // U(j + 1) | GELU(j) words 4-7
            pair_head(E0);
            step_down(gwO, S4, 0, up0, gwE, I1, I1);     // D(j)     | GELU(j + 1) words 0-3
            step_up(up1, j + 2, S4, 4, up0, gwE, I1, I0);  // U(j + 2) | GELU(j + 1) words 4-7
        }
        // pair (D(MB - 2), D(MB - 1)): GELU(MB - 1) (odd block: up1) whole behind D(MB - 2).  Between the two the h rows of the first NPF feature
        // tiles are requested, behind the registers of the accumulator tiles and one fragment pair (dead now), so that they arrive under the last
        // chain.  Rows beyond N are read (h is padded to whole 256-row tiles), never written.
        f32x4_t hv[NPF][4];
        pair_head(E0);
        step_down(gwE, S8, 0, up1, gwO, I1, I1);
        {
            const int l = fresh_lane();
            const unsigned hoff0 = (unsigned)(n_wave + (l >> 3)) * (unsigned)(4 * D) + 16u * (l & 7);
#pragma unroll
            for (int k = 0; k < NPF; ++k) load_h(hv[k], hoff0, k);
        }
        step_down(gwO, 1, 0, up1, gwE, I0, I0);

        // ---- epilogue: h += gate (out + b2); LayerNorm + modulate of the next sub-block ----
        const int le = fresh_lane(), chunk = le & 7, rowi = le >> 3, r = le & 31, hf = le >> 5;
        const int n_r = min(n_wave + r, g.N - 1);
        const unsigned traj = g.tpt_magic ? __umulhi((unsigned)n_r, g.tpt_magic) : (unsigned)n_r;
        const size_t mo = (size_t)traj * g.mod_stride;
        const unsigned wr_row = st0 + rowi * 128 + (((chunk ^ rowi) & 7) << 4);  // row-wise access: rows rowi + 8 i (same swizzle: (row & 7) = rowi)
        const unsigned hoff = (unsigned)(n_wave + rowi) * (unsigned)(4 * D) + 16u * chunk;  // (a pass's h stays below 4 GiB: host-checked)
        float sum = 0.0f;
#pragma unroll
        for (int ft = 0; ft < NT; ++ft) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<LDS_PTR(f32x4_t)>(wr_row + 1024 * i) = hv[ft % NPF][i];
            if (ft + NPF < NT) load_h(hv[ft % NPF], hoff, ft + NPF);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4_t hq = *reinterpret_cast<const LDS_PTR(f32x4_t)>(st0 + r * 128 + ((((2 * q + hf) ^ r) & 7) << 4));
                const int f = 32 * ft + 8 * q + 4 * hf;
                const float4 gt = *reinterpret_cast<const float4 *>(g.gate + mo + f);
                const float4 bb = *reinterpret_cast<const float4 *>(g.b2 + f);
                f32x16 &o = out[ft];
                o[4 * q] = fmaf(gt.x, o[4 * q] + bb.x, hq[0]);
                o[4 * q + 1] = fmaf(gt.y, o[4 * q + 1] + bb.y, hq[1]);
                o[4 * q + 2] = fmaf(gt.z, o[4 * q + 2] + bb.z, hq[2]);
                o[4 * q + 3] = fmaf(gt.w, o[4 * q + 3] + bb.w, hq[3]);
                sum += (o[4 * q] + o[4 * q + 1]) + (o[4 * q + 2] + o[4 * q + 3]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<LDS_PTR(f32x4_t)>(st0 + r * 128 + ((((2 * q + hf) ^ r) & 7) << 4)) = f32x4_t{out[ft][4 * q], out[ft][4 * q + 1], out[ft][4 * q + 2], out[ft][4 * q + 3]};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = n_wave + rowi + 8 * i;
                const f32x4_t v = *reinterpret_cast<const LDS_PTR(f32x4_t)>(wr_row + 1024 * i);
                if (n < g.N) *reinterpret_cast<f32x4_t *>(reinterpret_cast<char *>(g.h) + (size_t)(hoff + (unsigned)(8 * i * 4 * D + 128 * ft))) = v;
            }
        }
        if (g.a_next) {  // (uniform)
            constexpr float invD = 1.0f / (float)D;
            const float mean = half_pair_sum(sum) * invD;
            float qs = 0.0f;
#pragma unroll
            for (int ft = 0; ft < NT; ++ft)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float d = out[ft][i] - mean;
                    qs = fmaf(d, d, qs);
                }
            const float rstd = rsqrtf(half_pair_sum(qs) * invD + 1e-6f);
#pragma unroll
            for (int fp = 0; fp < NT / 2; ++fp) {  // 64 features = one 128-byte bf16 row segment per token
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    const int ft = 2 * fp + ii;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int f = 32 * ft + 8 * q + 4 * hf;
                        const float4 sc = *reinterpret_cast<const float4 *>(g.ln_scale + mo + f);
                        const float4 sf = *reinterpret_cast<const float4 *>(g.ln_shift + mo + f);
                        const f32x16 &o = out[ft];
                        const u32x2 pk = {pack2((o[4 * q] - mean) * rstd * (1.0f + sc.x) + sf.x, (o[4 * q + 1] - mean) * rstd * (1.0f + sc.y) + sf.y),
                                          pack2((o[4 * q + 2] - mean) * rstd * (1.0f + sc.z) + sf.z, (o[4 * q + 3] - mean) * rstd * (1.0f + sc.w) + sf.w)};
                        *reinterpret_cast<LDS_PTR(u32x2)>(st0 + r * 128 + ((((4 * ii + q) ^ r) & 7) << 4) + 8 * hf) = pk;
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int n = n_wave + rowi + 8 * i;
                    const u32x4 v = *reinterpret_cast<const LDS_PTR(u32x4)>(wr_row + 1024 * i);
                    if (n < g.N) *reinterpret_cast<u32x4 *>(g.a_next + (size_t)n * D + 64 * fp + 8 * chunk) = v;
                }
            }
        }
    }
    wait_vmcnt<0>();  // the ring's run-ahead requests must not land in LDS after the workgroup has gone
}
