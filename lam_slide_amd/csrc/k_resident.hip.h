// Trajectory-resident sampler for small trajectories (BASELINE config 3: pedestrian, T*L = 40 tokens of D = 128).
//
// Every operation of LatentSIV3 is independent per trajectory (attention stays inside (b,t) or (b,l) groups, latent_si_v31.py:51-61;
// conditioning is per sample), so when one trajectory's tokens fit one workgroup's LDS the WHOLE sampling loop of that trajectory -
// every state update, each with all 2*depth sub-blocks, the output head and the affine step - runs in ONE launch with one workgroup
// per trajectory and no inter-workgroup communication at all:
//   * activations never leave the CU: residual stream h (fp32), LayerNorm output a, q/k/v and [attention | GELU(mlp)] z (bf16) live in
//     LDS (~141 KiB for 40 tokens with the output-projection weights), the state x too; HBM sees the conditioning embedding and the modulation rows only;
//   * the weights (3 MB bf16 for the pedestrian model: they fit the XCD's 4 MiB L2) are streamed L2 -> VGPR as MFMA A fragments, each
//     fragment used by exactly one wave (the GEMV regime: no LDS staging, no barriers, deep prefetch: cdna_hip_programming.md 5,
//     "GEMV / M <= 16" row); the token operand (B fragments) of a GEMM is read from LDS ONCE and kept in registers for all feature
//     tiles (eight waves, two per SIMD, 256 VGPRs each: the epilogues, LayerNorms and the attention are latency-bound chains, and a second
//     wave per SIMD is what fills their stalls - measured 2.3 -> see profiles/r02_resident.txt);
//   * 16x16x32 bf16 MFMA, transposed product Ct[f][n] as in k_gemm.hip.h: features on accumulator rows (4 consecutive per lane, so
//     RoPE pairs are lane-local), tokens on lanes; 40 tokens pad to 48 (3 tiles), not to 64.
// The general path needs ~55 dependent launches of ~5 us per state update for such a batch (launch-bound: 3.2-3.4 ms for a
// 10-update call of 20 trajectories, hipGraph replay included); this kernel is one launch per group of <= 48 updates.
//
// Numerics: the same operand roundings as the general path (bf16 a, q, k, v, GELU output and attention output; fp32 everything else),
// except that softmax probabilities stay fp32 (they never become an MFMA operand here).  Results are within the same parity bars but
// not bit-identical to the general path; a model takes ONE of the two paths for every batch size (the choice depends on the model
// and on T*L only), so batch / shard / K-folding invariance holds bit for bit.
//
// Reference lines: latent_si_v31.py:45-63,168-188; mmdit.py:11-22,85-148,184-249; integrators.py:29-37,103-120.
#pragma once
#include "common.hip.h"
#include "k_small.hip.h"

typedef __attribute__((ext_vector_type(4))) float f32x4v;

constexpr int RES_D = 128, RES_H = 4, RES_HD = 32, RES_HHD = 128, RES_M = 256, RES_F1 = 640, RES_K2 = 384;
constexpr int RES_MAX_STEPS = 48, RES_MAX_BLOCKS = 16, RES_MAX_C = 32;
constexpr int RES_HS = 132;        // h row stride in floats (528 B: the 16 token rows of a tile fall on distinct 16-byte LDS slots)
constexpr int RES_QS = 392;        // qkv row stride in bf16 (784 B, same reason for the per-lane row reads of the attention)
// Small parameters of one stage (= one sub-block, or the output head), staged in LDS one stage ahead so that no phase waits on an L2
// round trip for them: [b1 640 | qs 32 | ks 32 | b2 128 | shift 128 | scale 128 | gate 128]; the head stage holds [shift 128 | scale 128]
constexpr int RES_PAR = 1280, RES_P_QS = 640, RES_P_KS = 672, RES_P_B2 = 704, RES_P_SHIFT = 832, RES_P_SCALE = 960, RES_P_GATE = 1088;

struct ResBlock {
    const u16 *w1;  // [640 (padded)][128] bf16
    const float *b1, *qs, *ks;
    const u16 *w2;  // [128 (padded)][384] bf16
    const float *b2;
};

struct ResArgs {
    float *x;                 // [B][n_t][C] state, updated in place
    const float *cond_emb;    // [B][n_t][D] fp32: cond_to_emb(x_cond) + biases + mask embedding (k_embed MODE 0)
    const float *mods;        // [n_steps][rows][MODW] modulation tables of every step of this launch (rows = B, or 1 when shared)
    long mods_step_stride;    // floats between steps
    int mods_traj_stride;     // floats between trajectories (0: one shared row)
    const float *x_in_w;      // [D][C]
    const float *out_w, *out_b;  // [C][D], [C]
    const float *noise;       // [.. steps][B*n_t*C] or NULL (device Philox)
    long noise_step_stride;
    unsigned long long seed, elem_offset;
    unsigned step0;           // index of this launch's first step in the sampler's step table (noise slice / Philox counter)
    float *trace;             // optional [.. steps][B*n_t*C]
    long trace_step_stride;
    int n_t, T, L, C, depth, normalize, n_steps;
    int skip;                 // -DLSL_EXPERIMENTS builds only (LSL_RES_SKIP, results WRONG): 1 attention, 2 linear1, 4 linear2, 8 LayerNorm
    float theta, q_premul;
    float4 step[RES_MAX_STEPS];  // (t, ax, am, aw) of lsl_step
    ResBlock blk[RES_MAX_BLOCKS];
};

template <int NNT>
struct ResLds {
    static constexpr int NP = 16 * NNT;
    static constexpr size_t h = 0;                                        // fp32 [NP][RES_HS]
    static constexpr size_t a = h + (size_t)NP * RES_HS * 4;              // bf16 [NP][128], 16-byte chunks XOR-swizzled by the row
    static constexpr size_t qkv = a + (size_t)NP * RES_D * 2;             // bf16 [NP][RES_QS]
    static constexpr size_t z = qkv + (size_t)NP * RES_QS * 2;            // bf16 [NP][384], chunks swizzled inside each 256-byte group
    static constexpr size_t x = z + (size_t)NP * RES_K2 * 2;              // fp32 [NP][32]
    static constexpr size_t wo = x + (size_t)NP * RES_MAX_C * 4;          // fp32 [32][RES_HS]: output projection weights (resident for the launch)
    static constexpr size_t par = wo + (size_t)RES_MAX_C * RES_HS * 4;    // fp32 [2][RES_PAR]: small parameters of the current / next stage
    static constexpr size_t rope = par + (size_t)2 * RES_PAR * 4;         // float2 [T + L][16]
    static constexpr size_t bytes(int T, int L) { return rope + (size_t)(T + L) * 16 * 8; }
};

__device__ __forceinline__ f32x4v mfma16(bf16x8 a, bf16x8 b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// sum over the 4 lanes {l, l^16, l^32, l^48} (the four 4-feature row groups of a 16x16 accumulator column): two VALU lane exchanges
// (gfx950 v_permlane16_swap / v_permlane32_swap; with both operands = v the two results are the values of the even and of the odd
// 16- / 32-lane rows, summed in the same order in every lane) instead of ds_bpermute round trips: one wave per SIMD has nothing to
// switch to while an LDS-queue shuffle is in flight
__device__ __forceinline__ float col_sum4(float v) {
    unsigned u = __builtin_bit_cast(unsigned, v);
    auto p = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = __builtin_bit_cast(float, (unsigned)p[0]) + __builtin_bit_cast(float, (unsigned)p[1]);
    u = __builtin_bit_cast(unsigned, v);
    p = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)p[0]) + __builtin_bit_cast(float, (unsigned)p[1]);
}
// sum over a group of 16 consecutive lanes (one DPP row): quad xor 1, quad xor 2, row_half_mirror, row_mirror - VALU only
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}

// byte offset of 16-byte chunk c of row r in a bf16 image with 256-byte chunk groups (a: one group per row; z: three)
__device__ __forceinline__ int res_swz(int row, int chunk) { return ((chunk & ~15) | ((chunk ^ row) & 15)) << 4; }

constexpr int RES_NW = 8, RES_NTHR = RES_NW * 64;

// max over the 4 lanes {l, l^16, l^32, l^48} (see col_sum4)
__device__ __forceinline__ float col_max4(float v) {
    unsigned u = __builtin_bit_cast(unsigned, v);
    auto p = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = fmaxf(__builtin_bit_cast(float, (unsigned)p[0]), __builtin_bit_cast(float, (unsigned)p[1]));
    u = __builtin_bit_cast(unsigned, v);
    p = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)p[0]), __builtin_bit_cast(float, (unsigned)p[1]));
}

// Workgroup barrier for LDS hand-offs WITHOUT draining the vector-memory queue: __syncthreads() emits s_waitcnt vmcnt(0), which would make
// every weight / parameter prefetch (global -> VGPR, issued a phase ahead on purpose) land before the next phase may start - measured:
// 4 barriers per sub-block each exposing an L2 round trip, ~45 us of a 141 us evaluation.  LDS operations of this wave are complete
// (lgkmcnt(0)) before it arrives; loads to registers stay in flight and are waited for by hipcc at their first use.
__device__ __forceinline__ void res_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

#ifdef LSL_EXPERIMENTS
// phase clock of workgroup 0 / wave 0 (tools/resident_probe.py --stamps): cycles spent up to the barrier that ends each phase
__device__ unsigned long long g_res_stamps[8];
#define RES_STAMP(PHASE)                                                                  \
    do {                                                                                  \
        if (blockIdx.x == 0 && tid == 0) {                                                \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();                 \
            atomicAdd(&g_res_stamps[PHASE], now_ - stamp_t_);                             \
            stamp_t_ = now_;                                                              \
        }                                                                                 \
    } while (0)
#else
#define RES_STAMP(PHASE) do {} while (0)
#endif

template <int NNT>
__global__ void __launch_bounds__(RES_NTHR, 2) k_resident(ResArgs A) {
    using LO = ResLds<NNT>;
    constexpr int NP = LO::NP, D = RES_D, NT = RES_NTHR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *hs = reinterpret_cast<float *>(smem + LO::h);
    char *as = smem + LO::a;
    u16 *qs_ = reinterpret_cast<u16 *>(smem + LO::qkv);
    char *zs = smem + LO::z;
    float *xs = reinterpret_cast<float *>(smem + LO::x);
    float *wos = reinterpret_cast<float *>(smem + LO::wo);
    float *pars = reinterpret_cast<float *>(smem + LO::par);
    float2 *rope_l = reinterpret_cast<float2 *>(smem + LO::rope), *rope_t = rope_l + A.L * 16;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g4 = lane >> 4;
    const int b = blockIdx.x, n_t = A.n_t, C = A.C, T = A.T, L = A.L;
    const size_t xoff = (size_t)b * n_t * C;
    const float *cond = A.cond_emb + (size_t)b * n_t * D;

    // ---- once per launch: state, RoPE tables, zero padding rows of the MFMA token operands ---------------------------------------
    for (int i = tid; i < n_t * RES_MAX_C; i += NT) {  // columns >= C are zero (the embedding walks all RES_MAX_C of them)
        const int n = i / RES_MAX_C, c = i % RES_MAX_C;
        xs[i] = c < C ? A.x[xoff + (size_t)n * C + c] : 0.0f;
    }
    for (int i = tid; i < (T + L) * 16; i += NT) {
        const int p = i < L * 16 ? i / 16 : (i - L * 16) / 16, j = i & 15;
        const double omega = 1.0 / pow((double)A.theta, (double)(2 * j) / (double)RES_HD);
        const double ang = (double)p * omega;
        rope_l[i] = make_float2((float)cos(ang), (float)sin(ang));  // (rope_t follows rope_l in memory)
    }
    for (int i = tid; i < (NP - n_t) * (D * 2 / 16); i += NT) *reinterpret_cast<u32x4 *>(as + (size_t)n_t * 256 + i * 16) = u32x4{0, 0, 0, 0};
    for (int i = tid; i < (NP - n_t) * (RES_K2 * 2 / 16); i += NT) *reinterpret_cast<u32x4 *>(zs + (size_t)n_t * 768 + i * 16) = u32x4{0, 0, 0, 0};
    const int e_d = tid & 127, e_par = tid >> 7;  // embedding: this thread's output column, token residue mod 4
    // the conditioning embedding of this thread's (column, tokens) does not change between state updates: registers
    float cnd[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) cnd[k] = (e_par + 4 * k) < n_t ? cond[(size_t)(e_par + 4 * k) * D + e_d] : 0.0f;
    // parameter staging: stage q = s * (nb + 1) + bi (bi == nb: the output head); values travel global -> registers at the start of the
    // previous stage and registers -> LDS buffer (q & 1) at its end
    const int nb = 2 * A.depth;
    float pr[3];
    auto par_issue = [&](int s2, int bi2) {
        const float *mods2 = A.mods + (size_t)s2 * A.mods_step_stride + (size_t)b * A.mods_traj_stride;
        // every lane loads unconditionally from a valid (clamped) address chosen by pointer arithmetic: a load inside a branch gets its
        // s_waitcnt at the end of that branch, i.e. the "prefetch" would wait for L2 right here (measured: ~3 us per stage)
        const ResBlock &B2 = A.blk[min(bi2, nb - 1)];
        const float *mb2 = mods2 + (size_t)(bi2 >> 1) * 6 * D + ((bi2 & 1) ? 3 * D : 0);  // (bi2 == nb: depth * 6 D = the adaLN rows)
        const bool head = bi2 >= nb;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int i = tid + NT * k;
            const float *src = i < RES_P_QS ? B2.b1 + i : i < RES_P_KS ? B2.qs + (i - RES_P_QS) : i < RES_P_B2 ? B2.ks + (i - RES_P_KS)
                               : i < RES_P_SHIFT ? B2.b2 + (i - RES_P_B2) : mb2 + min(i - RES_P_SHIFT, 3 * D - 1);
            if (head) src = mb2 + min(i, 2 * D - 1);
            pr[k] = *src;
        }
    };
    auto par_commit = [&](int q) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (tid + NT * k < RES_PAR) pars[(q & 1) * RES_PAR + tid + NT * k] = pr[k];
    };
    par_issue(0, 0);
    par_commit(0);
    // head: this thread's output channel and token group; the projection weights live in LDS (rows padded like h)
    const int h_c = tid & 31, h_grp = tid >> 5;
    for (int i = tid; i < RES_MAX_C * (D / 4); i += NT) {
        const int c = i / (D / 4), d = (i % (D / 4)) * 4;
        *reinterpret_cast<float4 *>(wos + (size_t)c * RES_HS + d) =
            c < C ? *reinterpret_cast<const float4 *>(A.out_w + (size_t)c * D + d) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float bo = h_c < C ? A.out_b[h_c] : 0.0f;
    res_barrier();

    // positions of this lane's NNT tokens along the two attended axes (no integer division in the epilogues)
    int pos_l[NNT], pos_t[NNT];
#pragma unroll
    for (int nt = 0; nt < NNT; ++nt) {
        const int nn = min(16 * nt + r16, n_t - 1);
        pos_l[nt] = nn % L;
        pos_t[nt] = nn / L;
    }
    // LayerNorm (+ optional modulate) of the rows of h, 16 lanes per row (8 values each), 4 rows per wave pass.
    // MODE 0: h <- LN_eps(h) in place; MODE 1: a (bf16, swizzled) <- LN(h)(1+scale)+shift; MODE 2: fp32 rows into `dst` (stride RES_HS)
    auto layer_norm = [&](int mode, float eps, const float *shift, const float *scale, float *dst) {
        const int sub = lane >> 4, col = (lane & 15) * 8;
        for (int row = wave * 4 + sub; row < ((n_t + 15) & ~15); row += 4 * RES_NW) {
            const bool ok = row < n_t;
            const float *hp = hs + (size_t)(ok ? row : 0) * RES_HS + col;
            const float4 v0 = *reinterpret_cast<const float4 *>(hp), v1 = *reinterpret_cast<const float4 *>(hp + 4);
            float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            float s = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            const float mean = row16_sum(s) * (1.0f / D);
            float q = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] -= mean;
                q = fmaf(v[e], v[e], q);
            }
            const float rstd = rsqrtf(row16_sum(q) * (1.0f / D) + eps);
            if (mode != 0) {
                const float4 s0 = *reinterpret_cast<const float4 *>(scale + col), s1 = *reinterpret_cast<const float4 *>(scale + col + 4);
                const float4 f0 = *reinterpret_cast<const float4 *>(shift + col), f1 = *reinterpret_cast<const float4 *>(shift + col + 4);
                const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, sf[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e] * rstd, 1.0f + sc[e], sf[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= rstd;
            }
            if (!ok) continue;
            if (mode == 1) {
                const u32x4 pk = {pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
                *reinterpret_cast<u32x4 *>(as + (size_t)row * 256 + res_swz(row, lane & 15)) = pk;
            } else {
                float *o = (mode == 0 ? hs : dst) + (size_t)row * RES_HS + col;
                *reinterpret_cast<float4 *>(o) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4 *>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
        }
    };

    // W fragments of one 16-feature tile: lane (r16, g4) holds W[f0 + r16][32 ks + 8 g4 .. + 7] for every k-step (row-contiguous weights:
    // the four 1 KiB loads of a K = 128 tile cover one contiguous 4 KiB block)
    auto load_w = [&](const u16 *W, int K, int f0, int nks, bf16x8 *dst) {
        const u16 *p = W + (size_t)(f0 + r16) * K + 8 * g4;
#pragma unroll 12
        for (int ks = 0; ks < nks; ++ks) dst[ks] = as_bf16x8(*reinterpret_cast<const u32x4 *>(p + 32 * ks));
    };

#ifdef LSL_EXPERIMENTS
    unsigned long long stamp_t_ = __builtin_amdgcn_s_memtime();
#endif
    for (int s = 0; s < A.n_steps; ++s) {
        // ---- embedding: h = x Wx^T + cond_emb (latent_si_v31.py:172), optional LayerNorm eps 1e-5 (:173-174) -------------------
        float wx[RES_MAX_C];  // this thread's column of the input projection (L1-resident; live only here: 256-VGPR budget)
#pragma unroll
        for (int c = 0; c < RES_MAX_C; c += 4) {
            const float4 w4 = c < C ? *reinterpret_cast<const float4 *>(A.x_in_w + (size_t)e_d * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            wx[c] = w4.x; wx[c + 1] = w4.y; wx[c + 2] = w4.z; wx[c + 3] = w4.w;
        }
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const int n = e_par + 4 * k;
            if (n >= n_t) continue;  // (a full unroll with static register indices: cnd[] must not be indexed dynamically)
            float acc = cnd[k];
            const float *xr = xs + n * RES_MAX_C;
#pragma unroll
            for (int c = 0; c < RES_MAX_C; c += 4) {
                const float4 xv = *reinterpret_cast<const float4 *>(xr + c);
                acc = fmaf(xv.x, wx[c], acc);
                acc = fmaf(xv.y, wx[c + 1], acc);
                acc = fmaf(xv.z, wx[c + 2], acc);
                acc = fmaf(xv.w, wx[c + 3], acc);
            }
            hs[(size_t)n * RES_HS + e_d] = acc;
        }
        res_barrier();
        if (A.normalize) {
            layer_norm(0, 1e-5f, nullptr, nullptr, nullptr);
            res_barrier();
        }
        RES_STAMP(4);

        for (int bi = 0; bi < nb; ++bi) {
            const ResBlock &B = A.blk[bi];
            const int temporal = bi & 1;
            const int q = s * (nb + 1) + bi;
            const float *pb = pars + (q & 1) * RES_PAR;  // this stage's small parameters (LDS)
            par_issue(s, bi + 1);                        // the next stage's (bi + 1 == nb: the head) travel to registers meanwhile
            // first weight fragments of linear1 are requested before the LayerNorm: they do not depend on it
            bf16x8 wA[2][4];
            load_w(B.w1, D, wave * 32, 4, wA[0]);
            load_w(B.w1, D, wave * 32 + 16, 4, wA[1]);
            if (!LSL_PROBE(A.skip, 8)) layer_norm(1, 1e-6f, pb + RES_P_SHIFT, pb + RES_P_SCALE, nullptr);
            res_barrier();
            RES_STAMP(0);
            // ---- linear1 (+ bias, QK-RMSNorm, RoPE, GELU): a[NP][128] x W1[640][128]^T -> qkv, z ---------------------------------
            // 20 pairs of 16-feature tiles (one 32-wide head, or 32 mlp features), pairs wave, wave + 8, wave + 16
            if (!LSL_PROBE(A.skip, 2)) {
                bf16x8 xf[NNT][4];
#pragma unroll
                for (int nt = 0; nt < NNT; ++nt)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
                        xf[nt][ks] = as_bf16x8(*reinterpret_cast<const u32x4 *>(as + (size_t)(16 * nt + r16) * 256 + res_swz(16 * nt + r16, 4 * ks + g4)));
                const float2 *rtab = temporal ? rope_t : rope_l;
                // One weight buffer, refilled for the NEXT pair right after the current pair's MFMAs have been issued: the loads then have the
                // whole epilogue to land, and at the top of the loop the only vector-memory operations in flight are exactly the fragments
                // the MFMAs need.  (With a second buffer prefetched BEFORE the MFMAs, hipcc's s_waitcnt bookkeeping merges the loop paths
                // and waits vmcnt(0) in front of the first MFMA, i.e. for the prefetch itself - seen in the .s as vmcnt(7) ... vmcnt(0).)
#pragma unroll 1
                for (int pair = wave; pair < 20; pair += RES_NW) {
                    const int f0 = pair * 32;
                    f32x4v acc[2][NNT];
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii) {
                        const float4 bq = *reinterpret_cast<const float4 *>(pb + f0 + 16 * ii + 4 * g4);
#pragma unroll
                        for (int nt = 0; nt < NNT; ++nt) acc[ii][nt] = f32x4v{bq.x, bq.y, bq.z, bq.w};
                    }
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                            for (int nt = 0; nt < NNT; ++nt) acc[ii][nt] = mfma16(wA[ii][ks], xf[nt][ks], acc[ii][nt]);
                    if (pair + RES_NW < 20) {
                        load_w(B.w1, D, f0 + RES_NW * 32, 4, wA[0]);
                        load_w(B.w1, D, f0 + RES_NW * 32 + 16, 4, wA[1]);
                    }
                    const int sec = pair >> 2;  // 0 q, 1 k, 2 v, 3-4 mlp (wave-uniform)
#pragma unroll
                    for (int nt = 0; nt < NNT; ++nt) {
                        const int n = 16 * nt + r16;
                        float v[8] = {acc[0][nt][0], acc[0][nt][1], acc[0][nt][2], acc[0][nt][3], acc[1][nt][0], acc[1][nt][1], acc[1][nt][2], acc[1][nt][3]};
                        if (sec < 2) {
                            float ss = 0.0f;
#pragma unroll
                            for (int e = 0; e < 8; ++e) ss = fmaf(v[e], v[e], ss);
                            ss = col_sum4(ss);
                            const float rr = rsqrtf(fmaf(ss, 1.0f / RES_HD, 1e-6f)) * (sec == 0 ? A.q_premul : 1.0f);
                            const int pos = temporal ? pos_t[nt] : pos_l[nt];
                            const float *sc = pb + (sec == 0 ? RES_P_QS : RES_P_KS);
#pragma unroll
                            for (int ii = 0; ii < 2; ++ii) {
                                const int d0 = 16 * ii + 4 * g4;  // first of this lane's 4 consecutive channels inside the head
                                const float4 s4 = *reinterpret_cast<const float4 *>(sc + d0);
                                const float4 cs = *reinterpret_cast<const float4 *>(rtab + pos * 16 + (d0 >> 1));  // (c0, s0, c1, s1)
                                const float x0 = v[4 * ii] * rr * s4.x, x1 = v[4 * ii + 1] * rr * s4.y, x2 = v[4 * ii + 2] * rr * s4.z, x3 = v[4 * ii + 3] * rr * s4.w;
                                v[4 * ii] = cs.x * x0 - cs.y * x1;
                                v[4 * ii + 1] = cs.y * x0 + cs.x * x1;
                                v[4 * ii + 2] = cs.z * x2 - cs.w * x3;
                                v[4 * ii + 3] = cs.w * x2 + cs.z * x3;
                            }
                        } else if (sec >= 3) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = gelu_fast(v[e]);
                        }
                        if (n < n_t) {
#pragma unroll
                            for (int ii = 0; ii < 2; ++ii) {
                                const u32x2 pk = {pack2(v[4 * ii], v[4 * ii + 1]), pack2(v[4 * ii + 2], v[4 * ii + 3])};
                                const int f = f0 + 16 * ii + 4 * g4;
                                if (sec < 3) *reinterpret_cast<u32x2 *>(qs_ + (size_t)n * RES_QS + f) = pk;
                                else {  // z column = HHD + (f - 3 HHD): 16-byte chunk index, 8-byte half inside it
                                    const int zc = f - 2 * RES_HHD;
                                    *reinterpret_cast<u32x2 *>(zs + (size_t)n * 768 + res_swz(n, zc >> 3) + (zc & 4) * 2) = pk;
                                }
                            }
                        }
                    }
                }
            }
            // linear2's weight tile (one 16-feature tile per wave) is requested before the attention
            bf16x8 w2f[12];
            load_w(B.w2, RES_K2, wave * 16, 12, w2f);
            res_barrier();
            RES_STAMP(1);
            // ---- attention over the spatial (sequence (t), positions l) or temporal (sequence (l), positions t) axis ------------------
            const int S = temporal ? T : L, n_seq = temporal ? L : T, sstride = temporal ? L : 1, qbase_mul = temporal ? 1 : L;
            if (LSL_PROBE(A.skip, 1)) {
            } else if (S > 8 && S <= 32) {
                // MFMA form, one unit = (sequence, head, 16-query tile): St = K Q^T (keys on accumulator rows, queries on lanes), softmax over
                // this lane's 4 (x2 key tiles) scores and the three other row groups of the column, Ot = V^T P^T with P taken straight from
                // the accumulator registers as the B operand (k slot j of row group g: key 4g + j for j < 4, key 16 + 4g + j - 4 otherwise) and
                // V^T fragments in the same key order from transposed LDS reads (ds_read_b64_tr_b16).
                const int QT = (S + 15) >> 4;
                for (int unit = wave; unit < n_seq * RES_H * QT; unit += RES_NW) {
                    const int qt = unit % QT, hh = (unit / QT) & 3, seq = unit / (QT * RES_H);
                    const int n0 = seq * qbase_mul;  // token of position p: n0 + p * sstride
                    const int pq = 16 * qt + r16;
                    const u16 *qp = qs_ + (size_t)(n0 + min(pq, S - 1) * sstride) * RES_QS + hh * RES_HD + 8 * g4;
                    const bf16x8 qf = as_bf16x8(*reinterpret_cast<const u32x4 *>(qp));
                    f32x4v sc[2];
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt) {
                        const int pk = min(16 * kt + r16, S - 1);
                        const bf16x8 kf = as_bf16x8(*reinterpret_cast<const u32x4 *>(qs_ + (size_t)(n0 + pk * sstride) * RES_QS + RES_HHD + hh * RES_HD + 8 * g4));
                        sc[kt] = mfma16(kf, qf, f32x4v{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (16 * kt + 4 * g4 + e >= S) sc[kt][e] = -INFINITY;
                    }
                    float mx = fmaxf(fmaxf(fmaxf(sc[0][0], sc[0][1]), fmaxf(sc[0][2], sc[0][3])), fmaxf(fmaxf(sc[1][0], sc[1][1]), fmaxf(sc[1][2], sc[1][3])));
                    mx = col_max4(mx);
                    float pe[8], sum = 0.0f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        pe[e] = __builtin_amdgcn_exp2f(sc[e >> 2][e & 3] - mx);
                        sum += pe[e];
                    }
                    sum = col_sum4(sum);
                    const u32x4 pw = {pack2(pe[0], pe[1]), pack2(pe[2], pe[3]), pack2(pe[4], pe[5]), pack2(pe[6], pe[7])};
                    const float inv = 1.0f / sum;
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        // transposed read: lane 4 q' + p' of a 16-lane group supplies row q' (key 4 g + q'), columns 4 p' .. 4 p' + 3 of the block
                        typedef __attribute__((ext_vector_type(8))) short s16x8;
                        s16x4 vt[2];
#pragma unroll
                        for (int kt = 0; kt < 2; ++kt) {
                            const int pk = min(16 * kt + 4 * g4 + (r16 >> 2), S - 1);
                            const u16 *vp = qs_ + (size_t)(n0 + pk * sstride) * RES_QS + 2 * RES_HHD + hh * RES_HD + 16 * dt + 4 * (r16 & 3);
                            vt[kt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(vp));
                        }
                        const s16x8 vv = {vt[0][0], vt[0][1], vt[0][2], vt[0][3], vt[1][0], vt[1][1], vt[1][2], vt[1][3]};
                        const f32x4v o = mfma16(__builtin_bit_cast(bf16x8, vv), as_bf16x8(pw), f32x4v{0.f, 0.f, 0.f, 0.f});
                        if (pq < S) {
                            const int n = n0 + pq * sstride, zc = hh * RES_HD + 16 * dt + 4 * g4;
                            const u32x2 pk2 = {pack2(o[0] * inv, o[1] * inv), pack2(o[2] * inv, o[3] * inv)};
                            *reinterpret_cast<u32x2 *>(zs + (size_t)n * 768 + res_swz(n, zc >> 3) + (zc & 4) * 2) = pk2;
                        }
                    }
                }
            } else {
                // short (or long) sequences: one lane per (query token, head); keys walk the sequence with an online softmax (fp32 probabilities)
                for (int item = tid; item < n_t * RES_H; item += NT) {
                    const int n = item >> 2, hh = item & 3;
                    const int k0 = temporal ? n % L : (n / L) * L;
                    float q[RES_HD];
                    {
                        const u16 *qp = qs_ + (size_t)n * RES_QS + hh * RES_HD;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const u32x4 w = *reinterpret_cast<const u32x4 *>(qp + 8 * c);
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                q[8 * c + 2 * k] = __uint_as_float(w[k] << 16);
                                q[8 * c + 2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
                            }
                        }
                    }
                    float o[RES_HD];
#pragma unroll
                    for (int d = 0; d < RES_HD; ++d) o[d] = 0.0f;
                    float mx = -INFINITY, sum = 0.0f;
                    for (int j = 0; j < S; ++j) {
                        const u16 *kp = qs_ + (size_t)(k0 + j * sstride) * RES_QS + RES_HHD + hh * RES_HD;
                        float dot = 0.0f;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const u32x4 w = *reinterpret_cast<const u32x4 *>(kp + 8 * c);
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                dot = fmaf(q[8 * c + 2 * k], __uint_as_float(w[k] << 16), dot);
                                dot = fmaf(q[8 * c + 2 * k + 1], __uint_as_float(w[k] & 0xffff0000u), dot);
                            }
                        }
                        const float mnew = fmaxf(mx, dot);
                        const float alpha = __builtin_amdgcn_exp2f(mx - mnew), p = __builtin_amdgcn_exp2f(dot - mnew);
                        mx = mnew;
                        sum = fmaf(sum, alpha, p);
                        const u16 *vp = kp + RES_HHD;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const u32x4 w = *reinterpret_cast<const u32x4 *>(vp + 8 * c);
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                o[8 * c + 2 * k] = fmaf(o[8 * c + 2 * k], alpha, p * __uint_as_float(w[k] << 16));
                                o[8 * c + 2 * k + 1] = fmaf(o[8 * c + 2 * k + 1], alpha, p * __uint_as_float(w[k] & 0xffff0000u));
                            }
                        }
                    }
                    const float inv = 1.0f / sum;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const u32x4 w = {pack2(o[8 * c] * inv, o[8 * c + 1] * inv), pack2(o[8 * c + 2] * inv, o[8 * c + 3] * inv),
                                         pack2(o[8 * c + 4] * inv, o[8 * c + 5] * inv), pack2(o[8 * c + 6] * inv, o[8 * c + 7] * inv)};
                        *reinterpret_cast<u32x4 *>(zs + (size_t)n * 768 + res_swz(n, 4 * hh + c)) = w;
                    }
                }
            }
            res_barrier();
            RES_STAMP(2);
            // ---- linear2 + gate * (.) + residual: z[NP][384] x W2[128][384]^T, h += gate (acc + b2)   (latent_si_v31.py:53,60) ------
            // 8 feature tiles of 16, one per wave
            if (!LSL_PROBE(A.skip, 4)) {
                const int f0 = wave * 16;
                const float4 bq = *reinterpret_cast<const float4 *>(pb + RES_P_B2 + f0 + 4 * g4);
                const float4 gt = *reinterpret_cast<const float4 *>(pb + RES_P_GATE + f0 + 4 * g4);
                f32x4v acc[NNT];
#pragma unroll
                for (int nt = 0; nt < NNT; ++nt) acc[nt] = f32x4v{bq.x, bq.y, bq.z, bq.w};
#pragma unroll
                for (int ks = 0; ks < 12; ++ks)
#pragma unroll
                    for (int nt = 0; nt < NNT; ++nt) {
                        const bf16x8 zf = as_bf16x8(*reinterpret_cast<const u32x4 *>(zs + (size_t)(16 * nt + r16) * 768 + res_swz(16 * nt + r16, 4 * ks + g4)));
                        acc[nt] = mfma16(w2f[ks], zf, acc[nt]);
                    }
#pragma unroll
                for (int nt = 0; nt < NNT; ++nt) {
                    const int n = 16 * nt + r16;
                    if (n < n_t) {
                        float *hp = hs + (size_t)n * RES_HS + f0 + 4 * g4;
                        float4 hv = *reinterpret_cast<float4 *>(hp);
                        hv.x = fmaf(gt.x, acc[nt][0], hv.x);
                        hv.y = fmaf(gt.y, acc[nt][1], hv.y);
                        hv.z = fmaf(gt.z, acc[nt][2], hv.z);
                        hv.w = fmaf(gt.w, acc[nt][3], hv.w);
                        *reinterpret_cast<float4 *>(hp) = hv;
                    }
                }
            }
            par_commit(q + 1);
            res_barrier();
            RES_STAMP(3);
        }

        // ---- output head (latent_si_v31.py:185-187) fused with the sampler's affine step -------------------------------------------
        const int qh = s * (nb + 1) + nb;
        const float *ph = pars + (qh & 1) * RES_PAR;  // adaLN: shift [0, D), scale [D, 2 D)
        if (s + 1 < A.n_steps) par_issue(s + 1, 0);
        float *ln = reinterpret_cast<float *>(smem + LO::qkv);  // fp32 rows [n_t][RES_HS]: q/k/v are dead here (NP * 784 B >= n_t * 528 B)
        layer_norm(2, 1e-6f, ph, ph + D, ln);
        res_barrier();
        const float4 sp = A.step[s];  // (t, ax, am, aw)
        if (h_c < C) {
            for (int n = h_grp; n < n_t; n += NT / 32) {
                const float *lr = ln + (size_t)n * RES_HS, *wr = wos + (size_t)h_c * RES_HS;
                float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f, acc3 = 0.0f;
#pragma unroll 8
                for (int d = 0; d < D; d += 4) {
                    const float4 lv = *reinterpret_cast<const float4 *>(lr + d), wv = *reinterpret_cast<const float4 *>(wr + d);
                    acc0 = fmaf(lv.x, wv.x, acc0);
                    acc1 = fmaf(lv.y, wv.y, acc1);
                    acc2 = fmaf(lv.z, wv.z, acc2);
                    acc3 = fmaf(lv.w, wv.w, acc3);
                }
                const float mo = ((acc0 + acc1) + (acc2 + acc3)) + bo;
                const size_t e = xoff + (size_t)n * C + h_c;
                float xn = sp.y * xs[n * RES_MAX_C + h_c] + sp.z * mo;
                if (sp.w != 0.0f)
                    xn += sp.w * (A.noise ? A.noise[(size_t)(A.step0 + s) * A.noise_step_stride + e] : philox_normal(A.seed, A.step0 + s, A.elem_offset + e));
                xs[n * RES_MAX_C + h_c] = xn;
                if (A.trace) A.trace[(size_t)(A.step0 + s) * A.trace_step_stride + e] = xn;
            }
        }
        if (s + 1 < A.n_steps) par_commit(qh + 1);
        res_barrier();
        RES_STAMP(5);
    }
    for (int i = tid; i < n_t * C; i += NT) A.x[xoff + i] = xs[(i / C) * RES_MAX_C + (i % C)];
}
