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
//   * a workgroup = NW waves x 32 tokens; wave w keeps its tokens' activations `a` (D / 4 VGPRs) as the MFMA B fragments of every k-step and
//     the whole output tile out^T[D features][32 tokens] (D / 2 accumulator VGPRs): NW = 8 (two waves per SIMD) for D <= 256, NW = 4 (one wave
//     per SIMD, the 512-register file) above;
//   * the WEIGHTS stream through LDS as ONE linear sequence of 1 KiB MFMA A fragments, packed once per call in exactly the order the kernel
//     consumes them (k_tail_pack): chunks of D / 16 fragments = [Wo k-steps 2c, 2c+1 x all D/32 output tiles] for the attention half,
//     then per mlp block j of 32 features [W1m block j: D/16 k-steps] and [W2m columns of block j: D/32 output tiles x 2 k-steps], in the
//     order U(0) U(1) D(0) U(2) D(1) ... U(MB-1) D(MB-2) D(MB-1).  A ring of NS chunks filled by LDS-DMA NS - 1 chunks ahead, one counted
//     wait + one workgroup barrier per chunk; the sequence is cyclic, so the ring runs through tile boundaries;
//   * up-projection U(j): D/16 MFMAs into one 32 x 32 accumulator tile (bias as the initial value) while the erf-GELU of block j - 1 (the
//     other accumulator tile) is computed in their shadows, 2 values per eighth of the chain; its 16 values per lane, rounded to bf16, ARE
//     the B fragments of the down-projection D(j - 1) - accumulator registers 8 s .. 8 s + 7 of a lane are k-step s - with W2m's columns
//     permuted to the accumulator's row order by the packer (the P-from-accumulator form of k_attn.hip.h);
//   * epilogue per tile and wave, through 4 KiB of wave-private LDS: the h rows arrive row-wise (whole 128-byte segments, 8 rows per
//     instruction), are re-read in the accumulator layout, updated, summed; the updated rows leave the same way; then the second pass over
//     the registers writes a_next.  Amortised over >= 1 000 MFMAs per wave.
//
// Numerics: out = the k-ascending chain over [attention | mlp] (mlp features inside a block in the accumulator's row order), + bias, fma with the
// gate onto the residual like EpiLinear2 (k_gemm.hip.h); row statistics two-pass in fp32 like k_ln_modulate_v4 with another summation tree.
// Not bit-identical to the linear2 / LayerNorm kernels it replaces (same numerics class: fp32 accumulation of bf16 products).
#pragma once
#include <type_traits>

#include "common.hip.h"
#include "k_lin1.hip.h"  // lin1_gelu

struct TailArgs {
    const u16 *wt;       // packed weight stream of this sub-block (k_tail_pack)
    const u16 *A;        // [N rounded up to 256][D] bf16: this sub-block's LayerNorm + modulate (linear1's input)
    const u16 *Z;        // [N rounded up to 256][zw] bf16: attention output in columns [0, HHD)
    const float *b1;     // [M]: linear1 bias, mlp section
    const float *b2;     // [D]
    const float *gate;   // mods + gate offset, row stride mod_stride (0: one row shared by every trajectory)
    float *h;            // [N][D] fp32 residual stream, updated in place
    u16 *a_next;         // [N rounded up to 256][D] bf16 or NULL (last sub-block: the head normalises itself); may alias A
    const float *ln_shift, *ln_scale;  // next sub-block's modulation rows (row stride mod_stride)
    int N, M, zw;
    int mod_stride, tpt;  // tokens per trajectory
    unsigned tpt_magic;   // floor(2^32 / tpt) + 1 (0 when tpt == 1)
#ifdef TAIL_STAMPS
    unsigned long long *dbg;  // tools/tail_harness.hip only: cycle sums per workgroup and wave
#endif
};

template <int D, int HHD, int NW>
struct TailCfg {
    static_assert(D % 64 == 0 && D <= 512 && HHD % 64 == 0, "hidden sizes 128 .. 512");
    static constexpr int NT = D / 32;       // output tiles of 32 features
    static constexpr int KS = D / 16;       // k-steps of the up-projection = B fragments a wave keeps
    static constexpr int KZ = HHD / 16;     // k-steps of the attention half
    static constexpr int CHF = D / 16;      // fragments per chunk
    static constexpr int CH = CHF * 1024;   // bytes per chunk
    static constexpr int CO = HHD / 32;     // chunks of the attention half
    static constexpr int NS = NW == 8 ? 4 : 3;  // two chunks in flight, one (two with the skewed half) in use
    static constexpr int PPW = CHF / NW;    // DMA instructions per wave and chunk
    static_assert(CHF % NW == 0, "whole DMA instructions per wave");
    static constexpr int RING = NS * CH, STAGE = NW * 4096;
    static constexpr int TT = NW * 32;      // tokens per tile
    static constexpr size_t lds_bytes(int M) { return (size_t)RING + STAGE + (size_t)M * 4; }
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

#ifndef TAIL_PD
#define TAIL_PD 0  // A fragments requested this many MFMAs ahead (0: 2 at two waves per SIMD, 6 at one)
#endif
#ifndef TAIL_SKEW
#define TAIL_SKEW 1  // waves 4-7 of an 8-wave workgroup run one chunk behind waves 0-3 (their SIMD partners)
#endif

template <int D, int HHD, int NW>
__global__ void __launch_bounds__(NW * 64, NW / 4) k_tail(TailArgs g) {
    using C = TailCfg<D, HHD, NW>;
    constexpr int NT = C::NT, KS = C::KS, KZ = C::KZ, CHF = C::CHF, CH = C::CH, CO = C::CO, NS = C::NS, PPW = C::PPW;
    constexpr int PD = TAIL_PD ? TAIL_PD : (NW == 8 ? 2 : 6);
    constexpr bool SKEW = TAIL_SKEW && NW == 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hf = lane >> 5;
    char *const stage = smem + C::RING + wave * 4096;
    float *const b1_lds = reinterpret_cast<float *>(smem + C::RING + C::STAGE);

    // Work = wave tiles of 32 tokens, cut evenly over the workgroups; a workgroup walks its range in rounds of NW wave tiles (one per wave).  In
    // a last, partial round the waves without a tile only keep the ring going: with one active wave per SIMD the round is bound by half the
    // MFMA work, so 2.5 rounds of work take about 2.6 round times, not 3.
    const int MB = g.M >> 5, NCH = CO + 2 * MB;
    const int nwt = (g.N + 31) >> 5;
    const int w0 = (int)((long)nwt * blockIdx.x / gridDim.x), w1 = (int)((long)nwt * (blockIdx.x + 1) / gridDim.x);
    if (w0 >= w1) return;  // (uniform)
    const int rounds = (w1 - w0 + NW - 1) / NW;

    for (int i = tid * 4; i < g.M; i += NW * 64 * 4) *reinterpret_cast<float4 *>(b1_lds + i) = *reinterpret_cast<const float4 *>(g.b1 + i);

    // ---- weight ring: the stream's chunk -> slot; wave w requests fragments PPW w .. PPW w + PPW - 1 of a chunk, one LDS-DMA instruction each
    // (inline asm on purpose, k_lin1.hip.h: behind the builtin hipcc waits for the request in front of the next LDS access of any kind)
    const unsigned lds0 = (unsigned)(size_t)(LDS_PTR(char))(smem);
    const unsigned lane_src = lane * 16;
    const char *const w_base = reinterpret_cast<const char *>(g.wt) + (size_t)wave * PPW * 1024;
    auto issue_piece = [&](const char *src, unsigned dst, auto ic) __attribute__((always_inline)) {
        constexpr int I = decltype(ic)::value;
        const unsigned ls = lane_src;
        asm volatile("s_add_u32 m0, %2, 0\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" ::"v"(ls), "s"(src), "s"(dst), "n"(1024 * (I & 3)) : "memory", "scc");
    };
    int c_src = 0, slot_d = 0;  // next chunk of the stream to request (index in [0, NCH)), and its slot
    auto issue = [&]() __attribute__((always_inline)) {
        const char *src = w_base + (size_t)c_src * CH;
        const unsigned dst = lds0 + slot_d * CH + wave * PPW * 1024;
        issue_piece(src, dst, std::integral_constant<int, 0>());
        if constexpr (PPW > 1) issue_piece(src, dst, std::integral_constant<int, 1>());
        if constexpr (PPW > 2) issue_piece(src, dst, std::integral_constant<int, 2>());
        if constexpr (PPW > 3) issue_piece(src, dst, std::integral_constant<int, 3>());
        if constexpr (PPW > 4) {
            issue_piece(src + 4096, dst + 4096, std::integral_constant<int, 0>());
            if constexpr (PPW > 5) issue_piece(src + 4096, dst + 4096, std::integral_constant<int, 1>());
            if constexpr (PPW > 6) issue_piece(src + 4096, dst + 4096, std::integral_constant<int, 2>());
            if constexpr (PPW > 7) issue_piece(src + 4096, dst + 4096, std::integral_constant<int, 3>());
        }
        c_src = c_src + 1 == NCH ? 0 : c_src + 1;
        slot_d = slot_d + 1 == NS ? 0 : slot_d + 1;
    };
    int slot_c = 0;  // slot of the next chunk this wave computes
    // Barrier b of the workgroup: chunk b has landed (requested two barriers ago; younger than it are only the requests of chunk b + 1 - extra
    // younger operations make the counted wait conservative, never wrong); the leading waves (all waves without the skew) compute chunk b
    // behind it, the skewed half chunk b - 1; nobody reads chunk b - 2 (b - 1 without the skew) any more, and chunk b + 2 is requested into
    // its slot.  Every wave passes the same barriers and issues at each of them.
#ifdef TAIL_STAMPS
    unsigned long long st_sum[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto stamp = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        __builtin_amdgcn_sched_barrier(0);
        return t;
    };
    unsigned long long st_t = stamp();
    const unsigned long long st_begin = st_t;
#define TAIL_ST(k) { const unsigned long long t_ = stamp(); st_sum[k] += t_ - st_t; st_t = t_; }
#else
#define TAIL_ST(k)
#endif
    auto step_head = [&]() __attribute__((always_inline)) {
        TAIL_ST(9)
        wait_vmcnt<PPW>();
        TAIL_ST(0)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        TAIL_ST(1)
        issue();
        TAIL_ST(2)
    };
    auto next_slot = [&]() __attribute__((always_inline)) { slot_c = slot_c + 1 == NS ? 0 : slot_c + 1; };
    auto frag = [&](const char *sb, int f) __attribute__((always_inline)) { return as_bf16x8(*reinterpret_cast<const u32x4 *>(sb + f * 1024)); };

    issue();
    issue();
    __syncthreads();  // bias table
    const bool lag = SKEW && wave >= 4;  // (uniform)
    if (lag) step_head();  // the skewed half starts one barrier late ...

    const int chunk = lane & 7, rowi = lane >> 3;
    const unsigned st0 = (unsigned)(size_t)(LDS_PTR(char))(stage);
    // rows of a [tokens][K] bf16 matrix as MFMA B fragments (k_lin1.hip.h load_x / finish_x): whole 128-byte lines per 8 lanes (8 rows per
    // instruction), then line by line through the wave's staging image into fragment order, in place: line j of every row holds the k-steps
    // 4 j .. 4 j + 3; chunk c of row t sits at t 128 + 16 (c ^ ((t >> 1) & 7)), conflict-free for both accesses
    auto load_rows = [&](auto &xreg, auto ks_c, const u16 *X, int n0, int stride) __attribute__((always_inline)) {
        constexpr int NK = decltype(ks_c)::value;
        const u16 *xr = X + (size_t)(n0 + rowi) * stride + 8 * chunk;
#pragma unroll
        for (int j = 0; j < NK / 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) xreg[4 * j + q] = as_bf16x8(*reinterpret_cast<const u32x4 *>(xr + (size_t)(8 * q) * stride + 64 * j));
    };
    auto finish_rows = [&](auto &xreg, auto ks_c) __attribute__((always_inline)) {
        constexpr int NK = decltype(ks_c)::value;
        const unsigned xw = st0 + rowi * 128, xr0 = st0 + r * 128;
#pragma unroll
        for (int j = 0; j < NK / 4; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<LDS_PTR(u32x4)>(xw + 1024 * q + (((chunk ^ ((rowi >> 1) + 4 * q)) & 7) << 4)) = as_u32x4(xreg[4 * j + q]);
#pragma unroll
            for (int m = 0; m < 4; ++m)
                xreg[4 * j + m] = as_bf16x8(*reinterpret_cast<const LDS_PTR(u32x4)>(xr0 + ((((2 * m + hf) ^ (r >> 1)) & 7) << 4)));
        }
    };

    for (int rd = 0; rd < rounds; ++rd) {
        const int wt = w0 + rd * NW + wave;
        if (wt >= w1) {  // (uniform) no tile for this wave in the last round: pass the round's barriers, keep requesting
            for (int c = 0; c < NCH; ++c) {
                step_head();
                next_slot();
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
            finish_rows(zreg, std::integral_constant<int, KZ>());
            TAIL_ST(6)
#pragma unroll
            for (int c = 0; c < CO; ++c) {
                step_head();
                const char *sb = smem + slot_c * CH + lane * 16;
                bf16x8 fr[PD];
#pragma unroll
                for (int f = 0; f < PD; ++f) fr[f] = frag(sb, f);
#pragma unroll
                for (int f = 0; f < CHF; ++f) {
                    out[f >> 1] = mfma32(fr[f % PD], zreg[2 * c + (f & 1)], out[f >> 1]);
                    if (f + PD < CHF) fr[f % PD] = frag(sb, f + PD);
                }
                next_slot();
                TAIL_ST(3)
            }
        }

        // ---- mlp: up-projection -> GELU in registers -> down-projection ----
        bf16x8 areg[KS];
        load_rows(areg, std::integral_constant<int, KS>(), g.A, n_wave, D);
        finish_rows(areg, std::integral_constant<int, KS>());
        TAIL_ST(6)
        f32x16 up0, up1;
        auto init_up = [&](f32x16 &a, int j) __attribute__((always_inline)) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 b = *reinterpret_cast<const float4 *>(b1_lds + j * 32 + 8 * q4 + 4 * hf);
                a[4 * q4] = b.x; a[4 * q4 + 1] = b.y; a[4 * q4 + 2] = b.z; a[4 * q4 + 3] = b.w;
            }
        };
        // U step: the chain of block j into `uc`; GELU of `ue` (block j - 1) in 8 slices behind eighths of the chain -> gw (8 packed words =
        // the two B fragments of D(j - 1))
        auto step_up = [&](f32x16 &uc, const f32x16 &ue, u32x4 (&gw)[2], auto mfma_c, auto gelu_c) __attribute__((always_inline)) {
            constexpr bool DO_MFMA = decltype(mfma_c)::value != 0, DO_GELU = decltype(gelu_c)::value != 0;
            constexpr int MPS = KS / 8 > 0 ? KS / 8 : 1;
            const char *sb = smem + slot_c * CH + lane * 16;
            bf16x8 fr[PD];
            if (DO_MFMA) {
#pragma unroll
                for (int f = 0; f < PD; ++f) fr[f] = frag(sb, f);
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if (DO_MFMA) {
#pragma unroll
                    for (int m = 0; m < MPS; ++m) {
                        const int ks = s * MPS + m;
                        if (ks < KS) {
                            uc = mfma32(fr[ks % PD], areg[ks], uc);
                            if (ks + PD < KS) fr[ks % PD] = frag(sb, ks + PD);
                        }
                    }
                }
                if (DO_GELU) {
                    // (the chain that wrote `ue` ended a whole step ago when DO_MFMA; the builtin form is used when it has only just ended)
                    const float g0 = DO_MFMA ? lin1_gelu(ue[2 * s]) : gelu_fast(ue[2 * s]);
                    const float g1 = DO_MFMA ? lin1_gelu(ue[2 * s + 1]) : gelu_fast(ue[2 * s + 1]);
                    gw[s >> 2][s & 3] = pack2(g0, g1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (DO_MFMA) next_slot();
            TAIL_ST(4)
        };
        auto step_down = [&](const u32x4 (&gw)[2]) __attribute__((always_inline)) {
            const char *sb = smem + slot_c * CH + lane * 16;
            bf16x8 fr[PD];
#pragma unroll
            for (int f = 0; f < PD; ++f) fr[f] = frag(sb, f);
#pragma unroll
            for (int f = 0; f < CHF; ++f) {
                out[f >> 1] = mfma32(fr[f % PD], as_bf16x8(gw[f & 1]), out[f >> 1]);
                if (f + PD < CHF) fr[f % PD] = frag(sb, f + PD);
            }
            next_slot();
            TAIL_ST(5)
        };
        std::integral_constant<int, 0> I0;
        std::integral_constant<int, 1> I1;
        u32x4 gw[2];
        step_head();
        init_up(up0, 0);
        step_up(up0, up1, gw, I1, I0);
        for (int j = 1; j + 1 < MB; j += 2) {  // blocks j (odd, into up1) and j + 1 (even, into up0)
            step_head();
            init_up(up1, j);
            step_up(up1, up0, gw, I1, I1);
            step_head();
            step_down(gw);
            step_head();
            init_up(up0, j + 1);
            step_up(up0, up1, gw, I1, I1);
            step_head();
            step_down(gw);
        }
        if ((MB & 1) == 0) {  // MB even: the last block is odd
            step_head();
            init_up(up1, MB - 1);
            step_up(up1, up0, gw, I1, I1);
            step_head();
            step_down(gw);
            step_up(up0, up1, gw, I0, I1);
        } else {
            step_up(up1, up0, gw, I0, I1);
        }
        step_head();
        step_down(gw);

        // ---- epilogue: h += gate (out + b2); LayerNorm + modulate of the next sub-block ----
        const int n_r = min(n_wave + r, g.N - 1);
        const unsigned traj = g.tpt_magic ? __umulhi((unsigned)n_r, g.tpt_magic) : (unsigned)n_r;
        const size_t mo = (size_t)traj * g.mod_stride;
        const unsigned wr_row = st0 + rowi * 128 + (((chunk ^ rowi) & 7) << 4);  // row-wise access: rows rowi + 8 i (same swizzle: (row & 7) = rowi)
        float sum = 0.0f;
#pragma unroll
        for (int ft = 0; ft < NT; ++ft) {
            float4 hv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = min(n_wave + rowi + 8 * i, g.N - 1);
                hv[i] = *reinterpret_cast<const float4 *>(g.h + (size_t)n * D + 32 * ft + 4 * chunk);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<LDS_PTR(f32x4_t)>(wr_row + 1024 * i) = f32x4_t{hv[i].x, hv[i].y, hv[i].z, hv[i].w};
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
                if (n < g.N) *reinterpret_cast<f32x4_t *>(g.h + (size_t)n * D + 32 * ft + 4 * chunk) = v;
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
        TAIL_ST(7)
    }
    if (SKEW && !lag) {  // ... and the leading half passes one more at the end: the same number of barriers for every wave
        wait_vmcnt<PPW>();
        __builtin_amdgcn_s_barrier();
    }
    wait_vmcnt<0>();  // the ring's run-ahead requests must not land in LDS after the workgroup has gone
#ifdef TAIL_STAMPS
    if (lane == 0 && g.dbg) {
        st_sum[8] = stamp() - st_begin;
        for (int k = 0; k < 10; ++k) g.dbg[((size_t)blockIdx.x * NW + wave) * 10 + k] = st_sum[k];
    }
#endif
}
