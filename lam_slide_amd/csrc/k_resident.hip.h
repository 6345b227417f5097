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
#include <type_traits>

typedef __attribute__((ext_vector_type(4))) float f32x4v;

constexpr int RES_D = 128, RES_H = 4, RES_HD = 32, RES_HHD = 128, RES_M = 256, RES_F1 = 640, RES_K2 = 384;
constexpr int RES_MAX_STEPS = 48, RES_MAX_BLOCKS = 16, RES_MAX_C = 32;
constexpr int RES_HS = 132;        // h row stride in floats (528 B: the 16 token rows of a tile fall on distinct 16-byte LDS slots)
constexpr int RES_QS = 392;        // qkv row stride in bf16 (784 B, same reason for the per-lane row reads of the attention)
constexpr int RES_XS = 36;         // state row stride in floats (144 B: conflict-free 16-byte reads of the embedding's token operand)
// Small parameters of one stage (= one sub-block, or the output head), staged in LDS one stage ahead so that no phase waits on an L2
// round trip for them: [b1 640 | qs 32 | ks 32 | b2 128 | shift 128 | scale 128 | gate 128]; the head stage holds [shift 128 | scale 128]
constexpr int RES_PAR = 1280, RES_P_QS = 640, RES_P_KS = 672, RES_P_B2 = 704, RES_P_SHIFT = 832, RES_P_SCALE = 960, RES_P_GATE = 1088;

struct ResBlock {
    const u16 *w1;  // linear1 weights in FRAGMENT order (k_res_pack): [40 tiles][4 k-steps][64 lanes][8] bf16
    const u16 *w2;  // linear2 weights, same order: [8 tiles][12 k-steps][64 lanes][8]
};
constexpr size_t RES_W1_ELEMS = (size_t)640 * 128, RES_W2_ELEMS = (size_t)128 * 384;
// Once per call (one launch): (a) the GEMM weights of every block are re-ordered into the MFMA A-fragment order, so that each of the
// kernel's weight loads (one k-step of one 16-feature tile, 16 bytes per lane) reads 1 KiB CONTIGUOUS - measured 2.6x the load rate of
// the row-major form, whose 64 lanes touch 16 rows x 64 bytes (a CU takes ~55 cycles per such instruction, ~24 per contiguous one), and
// the weight stream is what bounds this kernel; (b) the small per-block parameters are gathered into one row per block
// [b1 640 | qs 32 | ks 32 | b2 128], so that the one-stage-ahead parameter staging is one base pointer + lane offset.
struct ResPack {
    const u16 *w1[RES_MAX_BLOCKS], *w2[RES_MAX_BLOCKS];  // row-major [F pad][K] (packing.py)
    const float *b1[RES_MAX_BLOCKS], *qs[RES_MAX_BLOCKS], *ks[RES_MAX_BLOCKS], *b2[RES_MAX_BLOCKS];
};
__global__ void __launch_bounds__(256) k_res_pack(u16 *wout, float *pout, ResPack P) {
    const int bi = blockIdx.x, job = blockIdx.y, tid = threadIdx.x;  // job 0..39: linear1 tile, 40..47: linear2 tile, 48: parameters
    u16 *wb = wout + (size_t)bi * (RES_W1_ELEMS + RES_W2_ELEMS);
    if (job < 48) {
        const bool second = job >= 40;
        const int tile = second ? job - 40 : job, K = second ? RES_K2 : RES_D, nks = K / 32;
        const u16 *src = second ? P.w2[bi] : P.w1[bi];
        u16 *dst = (second ? wb + RES_W1_ELEMS : wb) + (size_t)tile * 16 * K;
        for (int i = tid; i < nks * 64; i += 256) {
            const int ks = i >> 6, lane = i & 63;
            *reinterpret_cast<u32x4 *>(dst + (size_t)i * 8) = *reinterpret_cast<const u32x4 *>(src + (size_t)(tile * 16 + (lane & 15)) * K + 32 * ks + 8 * (lane >> 4));
        }
    } else {
        for (int i = tid; i < RES_P_SHIFT; i += 256)
            pout[(size_t)bi * RES_P_SHIFT + i] = i < RES_P_QS ? P.b1[bi][i] : i < RES_P_KS ? P.qs[bi][i - RES_P_QS] : i < RES_P_B2 ? P.ks[bi][i - RES_P_KS] : P.b2[bi][i - RES_P_B2];
    }
}

struct ResArgs {
    float *x;                 // [B][n_t][C] state, updated in place
    float *cond_emb;          // [B][n_t][D] fp32 scratch: cond_to_emb(x_cond) + biases + mask embedding, written by the kernel's prologue
    const float *x_cond;      // [B][n_t][C]
    const int64_t *mask;      // [B][n_t]
    const float *cond_w, *cond_b, *x_in_b, *mask_emb;  // [D][C], [D], [D], [2][D]   (latent_si_v31.py:172)
    const float *mods;        // [n_steps][rows][MODW] modulation tables of every step of this launch (rows = B, or 1 when shared)
    long mods_step_stride;    // floats between steps
    int mods_traj_stride;     // floats between trajectories (0: one shared row)
    const float *blkpar;      // [2 depth][RES_P_SHIFT] (k_res_pack)
    const float *x_in_w;      // [D][C]
    const float *out_w, *out_b;  // [C][D], [C]
    const float *noise;       // [.. steps][B*n_t*C] or NULL (device Philox)
    long noise_step_stride;
    unsigned long long seed, elem_offset;
    unsigned step0;           // index of this launch's first step in the sampler's step table (noise slice / Philox counter)
    float *trace;             // optional [.. steps][B*n_t*C]
    long trace_step_stride;
    int n_t, T, L, C, depth, normalize, n_steps;
    int skip;                 // -DLSL_EXPERIMENTS builds only (LSL_RES_SKIP, results WRONG): 1 attention, 4 linear2, 8 LayerNorm
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
    static constexpr size_t x = z + (size_t)NP * RES_K2 * 2;              // fp32 [NP][RES_XS]
    static constexpr size_t wo = x + (size_t)NP * RES_XS * 4;             // fp32 [32][RES_HS]: output projection weights (resident for the launch)
    static constexpr size_t par = wo + (size_t)RES_MAX_C * RES_HS * 4;    // fp32 [2][RES_PAR]: small parameters of the current / next stage
    static constexpr size_t rope = par + (size_t)2 * RES_PAR * 4;         // float2 [T + L][16]
    static constexpr size_t bytes(int T, int L) { return rope + (size_t)(T + L) * 16 * 8; }
};

__device__ __forceinline__ f32x4v mfma16(bf16x8 a, bf16x8 b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
// exact fp32 multiply-add (v_mfma_f32_16x16x4_f32): four k-steps from one float4 per operand.  Lane (r16, g4) supplies A[row r16][k] and
// B[k][col r16] for the k values {4 g4 + e}; any k order is fine as long as both operands use the same one.
__device__ __forceinline__ f32x4v mfma16_f32x4(float4 a, float4 b, f32x4v c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
}

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

    int tid = threadIdx.x, lane = tid & 63;  // (not const: RES_OPAQUE_LANE below)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int r16 = lane & 15, g4 = lane >> 4;
    // hipcc hoists every per-lane address of the phases below (dozens of 64-bit pairs) out of the step / block loops and then spills them;
    // declaring the lane coordinates "modified" at the top of a loop body keeps each address computation next to its use
#define RES_OPAQUE_LANE() asm volatile("" : "+v"(tid), "+v"(lane), "+v"(r16), "+v"(g4))
    const int b = blockIdx.x, n_t = A.n_t, C = A.C, T = A.T, L = A.L;
    const size_t xoff = (size_t)b * n_t * C;
    float *cond = A.cond_emb + (size_t)b * n_t * D;

#ifdef LSL_EXPERIMENTS
    if (A.skip) {  // the phase-skipping probes read buffers nobody wrote: make them zeros, not whatever the previous kernel left in LDS
        for (int i = threadIdx.x * 16; i < (int)LO::bytes(A.T, A.L); i += RES_NTHR * 16) *reinterpret_cast<u32x4 *>(smem + i) = u32x4{0, 0, 0, 0};
        __syncthreads();
    }
#endif
    // ---- once per launch: state, RoPE tables, zero padding rows of the MFMA token operands ---------------------------------------
    for (int i = tid; i < NP * RES_MAX_C; i += NT) {  // columns >= C and rows >= n_t are zero (the embedding's MFMA walks all of them)
        const int n = i / RES_MAX_C, c = i % RES_MAX_C;
        xs[n * RES_XS + c] = (c < C && n < n_t) ? A.x[xoff + (size_t)n * C + c] : 0.0f;
    }
    for (int i = tid; i < (T + L) * 16; i += NT) {
        const int p = i < L * 16 ? i / 16 : (i - L * 16) / 16, j = i & 15;
        const double omega = 1.0 / pow((double)A.theta, (double)(2 * j) / (double)RES_HD);
        const double ang = (double)p * omega;
        rope_l[i] = make_float2((float)cos(ang), (float)sin(ang));  // (rope_t follows rope_l in memory)
    }
    for (int i = tid; i < (NP - n_t) * (D * 2 / 16); i += NT) *reinterpret_cast<u32x4 *>(as + (size_t)n_t * 256 + i * 16) = u32x4{0, 0, 0, 0};
    for (int i = tid; i < (NP - n_t) * (RES_K2 * 2 / 16); i += NT) *reinterpret_cast<u32x4 *>(zs + (size_t)n_t * 768 + i * 16) = u32x4{0, 0, 0, 0};
    // embedding (fp32 MFMA, features on accumulator rows): wave w owns features 16 w .. + 15 of all tokens.  Its slice of the input
    // projection and of the conditioning embedding (the accumulators' initial value) is the same for every state update, but 20 registers
    // are not free across the block loop: they are re-requested (L2 hits) in the head phase of the previous update
    // the conditioning embedding itself is computed here, once per launch, on the same fp32 MFMA tiles (it used to be a launch of the
    // general path's persistent embedding kernel: 21 us for 800 tokens): every lane stores the float4 it will re-read
    {
        float4 cw[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = 16 * j + 4 * g4;
            cw[j] = *reinterpret_cast<const float4 *>(A.cond_w + (size_t)(16 * wave + r16) * C + min(c, C - 4));
            if (c >= C) cw[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float4 cb = *reinterpret_cast<const float4 *>(A.cond_b + 16 * wave + 4 * g4), xb = *reinterpret_cast<const float4 *>(A.x_in_b + 16 * wave + 4 * g4);
#pragma unroll
        for (int nt = 0; nt < NNT; ++nt) {
            const int n = 16 * nt + r16, nn = min(n, n_t - 1);
            const float *xr = A.x_cond + ((size_t)b * n_t + nn) * C;
            const float4 me = *reinterpret_cast<const float4 *>(A.mask_emb + (A.mask[(size_t)b * n_t + nn] != 0 ? D : 0) + 16 * wave + 4 * g4);
            f32x4v acc = {(cb.x + xb.x) + me.x, (cb.y + xb.y) + me.y, (cb.z + xb.z) + me.z, (cb.w + xb.w) + me.w};
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (16 * j >= C) break;  // (uniform)
                const int c = 16 * j + 4 * g4;
                float4 xv = *reinterpret_cast<const float4 *>(xr + min(c, C - 4));
                if (c >= C) xv = make_float4(0.f, 0.f, 0.f, 0.f);
                acc = mfma16_f32x4(cw[j], xv, acc);
            }
            if (n < n_t) *reinterpret_cast<float4 *>(cond + (size_t)n * D + 16 * wave + 4 * g4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
    }
    float4 wxf[2], cnd[NNT];
    auto load_embed_operands = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = 16 * j + 4 * g4;
            wxf[j] = *reinterpret_cast<const float4 *>(A.x_in_w + (size_t)(16 * wave + r16) * C + min(c, C - 4));
            if (c >= C) wxf[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int nt = 0; nt < NNT; ++nt) {
            cnd[nt] = *reinterpret_cast<const float4 *>(cond + (size_t)min(16 * nt + r16, n_t - 1) * D + 16 * wave + 4 * g4);
            if (16 * nt + r16 >= n_t) cnd[nt] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    load_embed_operands();
    // parameter staging: stage q = s * (nb + 1) + bi (bi == nb: the output head); values travel global -> registers at the start of the
    // previous stage and registers -> LDS buffer (q & 1) at its end
    const int nb = 2 * A.depth;
    float pr[3];
    auto par_issue = [&](int s2, int bi2) {
        const float *mods2 = A.mods + (size_t)s2 * A.mods_step_stride + (size_t)b * A.mods_traj_stride;
        // every lane loads unconditionally from a valid (clamped) address: a load inside a branch gets its s_waitcnt at the end of that
        // branch, i.e. the "prefetch" would wait for L2 right here (measured: ~3 us per stage)
        const float *bp = A.blkpar + (size_t)min(bi2, nb - 1) * RES_P_SHIFT;
        const float *mb2 = mods2 + (size_t)(bi2 >> 1) * 6 * D + ((bi2 & 1) ? 3 * D : 0);  // (bi2 == nb: depth * 6 D = the adaLN rows)
        const bool head = bi2 >= nb;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int i = tid + NT * k;
            const float *p0 = bp + min(i, RES_P_SHIFT - 1);
            const float *p1 = mb2 + (head ? min(i, 2 * D - 1) : min(max(i - RES_P_SHIFT, 0), 3 * D - 1));
            pr[k] = *((i < RES_P_SHIFT && !head) ? p0 : p1);
        }
    };
    auto par_commit = [&](int q) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (tid + NT * k < RES_PAR) pars[(q & 1) * RES_PAR + tid + NT * k] = pr[k];
    };
    par_issue(0, 0);
    par_commit(0);
    // head (fp32 MFMA): the projection weights live in LDS (rows padded like h); wave w < 2 NNT owns channels 16 (w / NNT) .. + 15 of
    // token tile w % NNT
    for (int i = tid; i < RES_MAX_C * (D / 4); i += NT) {
        const int c = i / (D / 4), d = (i % (D / 4)) * 4;
        *reinterpret_cast<float4 *>(wos + (size_t)c * RES_HS + d) =
            c < C ? *reinterpret_cast<const float4 *>(A.out_w + (size_t)c * D + d) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int h_nt = wave % NNT, h_ct = wave / NNT, h_c0 = 16 * h_ct + 4 * g4;
    const bool h_on = wave < 2 * NNT && 16 * h_ct < C;  // (wave-uniform)
    const float4 bo = (h_on && h_c0 < C) ? *reinterpret_cast<const float4 *>(A.out_b + h_c0) : make_float4(0.f, 0.f, 0.f, 0.f);
    res_barrier();

    // positions of this lane's NNT tokens along the two attended axes (no integer division in the epilogues)
    int pos_l[NNT], pos_t[NNT];
#pragma unroll
    for (int nt = 0; nt < NNT; ++nt) {
        const int nn = min(16 * nt + r16, n_t - 1);
        pos_l[nt] = nn % L;
        pos_t[nt] = nn / L;
    }
    // LayerNorm (+ optional modulate) of the rows of h: 8 lanes per row (16 values each), 8 rows per wave, every row in ONE pass (64 >= NP).
    // MODE 0: h <- LN_eps(h) in place; MODE 1: a (bf16, swizzled) <- LN(h)(1+scale)+shift; MODE 2: fp32 rows into `dst` (stride RES_HS)
    auto row8_sum = [&](float v) {  // sum over a group of 8 consecutive lanes: quad xor 1, quad xor 2, row_half_mirror
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
        return v;
    };
    auto layer_norm = [&](int mode, float eps, const float *shift, const float *scale, float *dst) {
        const int row = wave * 8 + (lane >> 3), l8 = lane & 7, col = l8 * 16;
        if (wave * 8 >= n_t) return;  // (wave-uniform)
        const bool ok = row < n_t;
        const float *hp = hs + (size_t)(ok ? row : 0) * RES_HS + col;
        float v[16];
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const float4 t = *reinterpret_cast<const float4 *>(hp + 4 * c4);
            v[4 * c4] = t.x; v[4 * c4 + 1] = t.y; v[4 * c4 + 2] = t.z; v[4 * c4 + 3] = t.w;
        }
        float4 sc[4], sf[4];  // modulation rows: requested with the row itself, used two reductions later
        if (mode != 0) {
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                sc[c4] = *reinterpret_cast<const float4 *>(scale + col + 4 * c4);
                sf[c4] = *reinterpret_cast<const float4 *>(shift + col + 4 * c4);
            }
        }
        float s = 0.0f;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) s += (v[4 * c4] + v[4 * c4 + 1]) + (v[4 * c4 + 2] + v[4 * c4 + 3]);
        const float mean = row8_sum(s) * (1.0f / D);
        float q = 0.0f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            v[e] -= mean;
            q = fmaf(v[e], v[e], q);
        }
        const float rstd = rsqrtf(row8_sum(q) * (1.0f / D) + eps);
        if (mode != 0) {
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                v[4 * c4] = fmaf(v[4 * c4] * rstd, 1.0f + sc[c4].x, sf[c4].x);
                v[4 * c4 + 1] = fmaf(v[4 * c4 + 1] * rstd, 1.0f + sc[c4].y, sf[c4].y);
                v[4 * c4 + 2] = fmaf(v[4 * c4 + 2] * rstd, 1.0f + sc[c4].z, sf[c4].z);
                v[4 * c4 + 3] = fmaf(v[4 * c4 + 3] * rstd, 1.0f + sc[c4].w, sf[c4].w);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] *= rstd;
        }
        if (!ok) return;
        if (mode == 1) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const u32x4 pk = {pack2(v[8 * h2], v[8 * h2 + 1]), pack2(v[8 * h2 + 2], v[8 * h2 + 3]), pack2(v[8 * h2 + 4], v[8 * h2 + 5]), pack2(v[8 * h2 + 6], v[8 * h2 + 7])};
                *reinterpret_cast<u32x4 *>(as + (size_t)row * 256 + res_swz(row, 2 * l8 + h2)) = pk;
            }
        } else {
            float *o = (mode == 0 ? hs : dst) + (size_t)row * RES_HS + col;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) *reinterpret_cast<float4 *>(o + 4 * c4) = make_float4(v[4 * c4], v[4 * c4 + 1], v[4 * c4 + 2], v[4 * c4 + 3]);
        }
    };

    // W fragments of one 16-feature tile (rows f0 .. f0 + 15, f0 a multiple of 16): lane (r16, g4) gets W[f0 + r16][32 ks + 8 g4 .. + 7] for
    // every k-step; in the packed order that is 16 bytes at (tile, ks, lane): every load instruction reads 1 KiB contiguous
    auto load_w = [&](const u16 *W, int K, int f0, int nks, bf16x8 *dst) {
        const u16 *p = W + (size_t)f0 * K + 8 * lane;
#pragma unroll 12
        for (int ks = 0; ks < nks; ++ks) dst[ks] = as_bf16x8(*reinterpret_cast<const u32x4 *>(p + 512 * ks));
    };

    // linear1's 40 feature tiles of 16, five per wave with the same mix in every wave (equal MFMA and VALU work, no loop, no branch):
    // one q (waves 0-3) or k (waves 4-7) head = 2 tiles, one v tile, two mlp tiles (one GELU pair)
    const int l1_sec = wave >> 2;
    const int l1_f[5] = {l1_sec * RES_HHD + (wave & 3) * RES_HD, l1_sec * RES_HHD + (wave & 3) * RES_HD + 16, 2 * RES_HHD + 16 * wave,
                         3 * RES_HHD + 32 * wave, 3 * RES_HHD + 32 * wave + 16};
    bf16x8 w1f[5][4];  // their weight fragments, requested one phase ahead of their use
    auto load_w1 = [&](const u16 *W1) {
#pragma unroll
        for (int t = 0; t < 5; ++t) load_w(W1, D, l1_f[t], 4, w1f[t]);
    };
    load_w1(A.blk[0].w1);
#ifdef LSL_EXPERIMENTS
    unsigned long long stamp_t_ = __builtin_amdgcn_s_memtime();
#endif
    for (int s = 0; s < A.n_steps; ++s) {
        RES_OPAQUE_LANE();
        // ---- embedding: h = x Wx^T + cond_emb (latent_si_v31.py:172), optional LayerNorm eps 1e-5 (:173-174) -------------------
        // features 16 w .. + 15 (accumulator rows 4 g4 .. + 3) of every token (lanes r16), k = the C <= 32 input channels, fp32 MFMA
#pragma unroll
        for (int nt = 0; nt < NNT; ++nt) {
            f32x4v acc = {cnd[nt].x, cnd[nt].y, cnd[nt].z, cnd[nt].w};
            const float *xr = xs + (16 * nt + r16) * RES_XS + 4 * g4;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (16 * j >= C) break;  // (uniform)
                acc = mfma16_f32x4(wxf[j], *reinterpret_cast<const float4 *>(xr + 16 * j), acc);
            }
            if (16 * nt + r16 < n_t) *reinterpret_cast<float4 *>(hs + (size_t)(16 * nt + r16) * RES_HS + 16 * wave + 4 * g4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
        res_barrier();
        if (A.normalize) {
            layer_norm(0, 1e-5f, nullptr, nullptr, nullptr);
            res_barrier();
        }
        RES_STAMP(4);

        for (int bi = 0; bi < nb; ++bi) {
            RES_OPAQUE_LANE();
            const ResBlock &B = A.blk[bi];
            const int temporal = bi & 1;
            const int q = s * (nb + 1) + bi;
            const float *pb = pars + (q & 1) * RES_PAR;  // this stage's small parameters (LDS)
            par_issue(s, bi + 1);                        // the next stage's (bi + 1 == nb: the head) travel to registers meanwhile
            if (!LSL_PROBE(A.skip, 8)) layer_norm(1, 1e-6f, pb + RES_P_SHIFT, pb + RES_P_SCALE, nullptr);
            res_barrier();
            RES_STAMP(0);
            // ---- linear1 (+ bias, QK-RMSNorm, RoPE, GELU): a[NP][128] x W1[640][128]^T -> qkv, z ---------------------------------
            // Straight-line code, the same five tiles for every wave (l1_f): the weights were requested a phase ago (w1f), the token
            // operand is read once, all 60 MFMAs are issued back to back - the mlp tiles first, so that their GELUs (the longest VALU
            // chains) run in the shadow of the other tiles' MFMAs - and linear2's weight tile is requested as soon as w1f is dead.
            bf16x8 w2f[12];
            {
                bf16x8 xf[NNT][4];
#pragma unroll
                for (int nt = 0; nt < NNT; ++nt)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
                        xf[nt][ks] = as_bf16x8(*reinterpret_cast<const u32x4 *>(as + (size_t)(16 * nt + r16) * 256 + res_swz(16 * nt + r16, 4 * ks + g4)));
                f32x4v acc[5][NNT];
#pragma unroll
                for (int t = 0; t < 5; ++t) {
                    const float4 bq = *reinterpret_cast<const float4 *>(pb + l1_f[t] + 4 * g4);
#pragma unroll
                    for (int nt = 0; nt < NNT; ++nt) acc[t][nt] = f32x4v{bq.x, bq.y, bq.z, bq.w};
                }
                // tile-major; once the NEXT tile's MFMAs are issued a tile's weight registers are refilled with the next stage's tile (the
                // next block's, or block 0's for the next state update): the weight stream (the CU takes ~24 cycles per 1 KiB load, 256
                // loads per sub-block) then runs under this phase's MFMAs and VALU-heavy epilogue instead of in front of linear2
                const u16 *w1n = A.blk[bi + 1 < nb ? bi + 1 : 0].w1;
                // (One tile behind, and the scheduler may not move anything across the fences.  With the loads issued directly behind their
                // own tile's MFMAs the two-tile instance of this kernel produced run-to-run differences of 1e-7 .. 1e-5 - see
                // profiles/r02_resident.txt; cause not established, every variant that keeps a tile of MFMAs or a VALU block between
                // an MFMA and the next load into a register it read is bit-stable over thousands of reruns.)
#pragma unroll
                for (int tt = 0; tt < 5; ++tt) {
                    const int t = (tt + 3) % 5;  // 3, 4 (mlp), 0, 1 (head), 2 (v)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                        for (int nt = 0; nt < NNT; ++nt) acc[t][nt] = mfma16(w1f[t][ks], xf[nt][ks], acc[t][nt]);
#ifdef RES_LOADS_BEHIND_MFMA  // (measured-and-withdrawn order, see the note at the linear2 tile below)
                    __builtin_amdgcn_sched_barrier(0);
                    if (tt >= 1) load_w(w1n, D, l1_f[(tt + 2) % 5], 4, w1f[(tt + 2) % 5]);
                    __builtin_amdgcn_sched_barrier(0);
#endif
                }
                const float2 *rtab = temporal ? rope_t : rope_l;
                // mlp: erf-GELU, z columns HHD + (f - 3 HHD)
#pragma unroll
                for (int nt = 0; nt < NNT; ++nt) {
                    const int n = 16 * nt + r16;
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = gelu_fast(acc[3 + ii][nt][e]);
                        if (n < n_t) {
                            const u32x2 pk = {pack2(v[0], v[1]), pack2(v[2], v[3])};
                            const int zc = l1_f[3 + ii] + 4 * g4 - 2 * RES_HHD;  // 16-byte chunk index zc >> 3, 8-byte half inside it
                            *reinterpret_cast<u32x2 *>(zs + (size_t)n * 768 + res_swz(n, zc >> 3) + (zc & 4) * 2) = pk;
                        }
                    }
                }
                // the last tile's refill and linear2's weight tile: behind the GELU block
#ifndef RES_LOADS_BEHIND_MFMA
                // Product order: EVERY weight load of the next stage sits behind the GELU block, i.e. a whole VALU block behind the last MFMA
                // that touched the registers it fills.  The faster order (-DRES_LOADS_BEHIND_MFMA: refill one tile behind the MFMAs, 6 % faster
                // on the 20-sample scene) gave run-to-run differences of 1e-7 .. 1e-5 when the refill loads came 2-3 instructions behind
                // those MFMAs and none at 4-9 (profiles/r02_resident.txt); the cause was never established, so the product does not depend
                // on that distance: tools/isa_scan.py (tests/test_isa_scan.py) fails the build if any vector-memory load lands in a
                // register that was the C/D operand of an MFMA fewer than 12 instructions earlier.
                __builtin_amdgcn_sched_barrier(0);
                load_w1(w1n);
                load_w(B.w2, RES_K2, wave * 16, 12, w2f);
                __builtin_amdgcn_sched_barrier(0);
#else
                __builtin_amdgcn_sched_barrier(0);
                load_w(w1n, D, l1_f[2], 4, w1f[2]);
                load_w(B.w2, RES_K2, wave * 16, 12, w2f);
                __builtin_amdgcn_sched_barrier(0);
#endif
                // this wave's q (waves 0-3) or k (4-7) head: RMS norm over the head's 32 channels, scale, RoPE (q: * softmax scale * log2 e)
                const float *sc = pb + (l1_sec == 0 ? RES_P_QS : RES_P_KS);
                const float post = l1_sec == 0 ? A.q_premul : 1.0f;
#pragma unroll
                for (int nt = 0; nt < NNT; ++nt) {
                    const int n = 16 * nt + r16;
                    float v[8] = {acc[0][nt][0], acc[0][nt][1], acc[0][nt][2], acc[0][nt][3], acc[1][nt][0], acc[1][nt][1], acc[1][nt][2], acc[1][nt][3]};
                    {
                        float ss = 0.0f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) ss = fmaf(v[e], v[e], ss);
                        ss = col_sum4(ss);
                        const float rr = rsqrtf(fmaf(ss, 1.0f / RES_HD, 1e-6f)) * post;
                        const int pos = temporal ? pos_t[nt] : pos_l[nt];
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
                    }
                    if (n < n_t) {
#pragma unroll
                        for (int ii = 0; ii < 2; ++ii) {
                            const u32x2 pk = {pack2(v[4 * ii], v[4 * ii + 1]), pack2(v[4 * ii + 2], v[4 * ii + 3])};
                            *reinterpret_cast<u32x2 *>(qs_ + (size_t)n * RES_QS + l1_f[ii] + 4 * g4) = pk;
                        }
                        const u32x2 pv = {pack2(acc[2][nt][0], acc[2][nt][1]), pack2(acc[2][nt][2], acc[2][nt][3])};  // v: as is
                        *reinterpret_cast<u32x2 *>(qs_ + (size_t)n * RES_QS + l1_f[2] + 4 * g4) = pv;
                    }
                }
            }
            res_barrier();
            RES_STAMP(1);
            // ---- attention over the spatial (sequence (t), positions l) or temporal (sequence (l), positions t) axis ------------------
            const int S = temporal ? T : L, n_seq = temporal ? L : T, sstride = temporal ? L : 1, qbase_mul = temporal ? 1 : L;
            if (LSL_PROBE(A.skip, 1)) {
            } else if (S > 8) {  // (S <= 32: resident_ok)
                // MFMA form, one unit = (sequence, head, 16-query tile): St = K Q^T (keys on accumulator rows, queries on lanes), softmax over
                // this lane's 4 (x2 key tiles) scores and the three other row groups of the column, Ot = V^T P^T with P taken straight from
                // the accumulator registers as the B operand (k slot j of row group g: key 4g + j for j < 4, key 16 + 4g + j - 4 otherwise) and
                // V^T fragments in the same key order from transposed LDS reads (ds_read_b64_tr_b16).
                // Two units per pass and every LDS read of both (q, k, V^T: none depends on the softmax) requested before the first MFMA: the
                // phase is a latency chain (LDS -> MFMA -> exchange -> exp -> MFMA -> store), two independent chains fill each other's gaps.
                const int QT = (S + 15) >> 4, n_units = n_seq * RES_H * QT;
                typedef __attribute__((ext_vector_type(8))) short s16x8;
                for (int unit0 = wave; unit0 < n_units; unit0 += 2 * RES_NW) {
                    int n0[2], pq[2], hh[2];
                    bool on[2];
                    bf16x8 qf[2], kf[2][2];
                    s16x4 vt[2][2][2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int unit = unit0 + u * RES_NW;
                        on[u] = unit < n_units;  // (wave-uniform; an absent second unit recomputes the first and stores nothing)
                        const int un = on[u] ? unit : unit0;
                        const int qt = un % QT, seq = un / (QT * RES_H);
                        hh[u] = (un / QT) & 3;
                        n0[u] = seq * qbase_mul;  // token of position p: n0 + p * sstride
                        pq[u] = 16 * qt + r16;
                        qf[u] = as_bf16x8(*reinterpret_cast<const u32x4 *>(qs_ + (size_t)(n0[u] + min(pq[u], S - 1) * sstride) * RES_QS + hh[u] * RES_HD + 8 * g4));
#pragma unroll
                        for (int kt = 0; kt < 2; ++kt) {
                            const int pk = min(16 * kt + r16, S - 1);
                            kf[u][kt] = as_bf16x8(*reinterpret_cast<const u32x4 *>(qs_ + (size_t)(n0[u] + pk * sstride) * RES_QS + RES_HHD + hh[u] * RES_HD + 8 * g4));
                        }
                        // transposed reads: lane 4 q' + p' of a 16-lane group supplies row q' (key 4 g + q'), columns 4 p' .. 4 p' + 3 of the block
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                            for (int kt = 0; kt < 2; ++kt) {
                                const int pk = min(16 * kt + 4 * g4 + (r16 >> 2), S - 1);
                                const u16 *vp = qs_ + (size_t)(n0[u] + pk * sstride) * RES_QS + 2 * RES_HHD + hh[u] * RES_HD + 16 * dt + 4 * (r16 & 3);
                                vt[u][dt][kt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(vp));
                            }
                    }
                    f32x4v sc[2][2];
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int kt = 0; kt < 2; ++kt) sc[u][kt] = mfma16(kf[u][kt], qf[u], f32x4v{0.f, 0.f, 0.f, 0.f});
                    u32x4 pw[2];
                    float inv[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
#pragma unroll
                        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (16 * kt + 4 * g4 + e >= S) sc[u][kt][e] = -INFINITY;
                        float mx = fmaxf(fmaxf(fmaxf(sc[u][0][0], sc[u][0][1]), fmaxf(sc[u][0][2], sc[u][0][3])),
                                         fmaxf(fmaxf(sc[u][1][0], sc[u][1][1]), fmaxf(sc[u][1][2], sc[u][1][3])));
                        mx = col_max4(mx);
                        float pe[8], sum = 0.0f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            pe[e] = __builtin_amdgcn_exp2f(sc[u][e >> 2][e & 3] - mx);
                            sum += pe[e];
                        }
                        sum = col_sum4(sum);
                        pw[u] = u32x4{pack2(pe[0], pe[1]), pack2(pe[2], pe[3]), pack2(pe[4], pe[5]), pack2(pe[6], pe[7])};
                        inv[u] = 1.0f / sum;
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt) {
                            const s16x8 vv = {vt[u][dt][0][0], vt[u][dt][0][1], vt[u][dt][0][2], vt[u][dt][0][3],
                                              vt[u][dt][1][0], vt[u][dt][1][1], vt[u][dt][1][2], vt[u][dt][1][3]};
                            const f32x4v o = mfma16(__builtin_bit_cast(bf16x8, vv), as_bf16x8(pw[u]), f32x4v{0.f, 0.f, 0.f, 0.f});
                            if (on[u] && pq[u] < S) {
                                const int n = n0[u] + pq[u] * sstride, zc = hh[u] * RES_HD + 16 * dt + 4 * g4;
                                const u32x2 pk2 = {pack2(o[0] * inv[u], o[1] * inv[u]), pack2(o[2] * inv[u], o[3] * inv[u])};
                                *reinterpret_cast<u32x2 *>(zs + (size_t)n * 768 + res_swz(n, zc >> 3) + (zc & 4) * 2) = pk2;
                            }
                        }
                }
            } else {
                // short sequences (S <= 8): FOUR lanes (one DPP quad) per (query token, head), each owning 8 of the head's 32 channels: the
                // partial dot products meet in two quad exchanges, the softmax is computed redundantly, each lane finishes and stores its own
                // 16-byte chunk.  (One lane per item left 160 of 512 lanes busy on a 4x longer dependent chain; fp32 probabilities.)
                // Compiled for SP = 2, 4 or 8 key slots (slots >= S re-read the last key and get probability 0) so that every LDS read of
                // an item - q, SP keys, SP values - is requested before the first use: one LDS latency instead of 1 + 2 S.
                auto quad_attention = [&](auto sp_tag) {
                    constexpr int SP = decltype(sp_tag)::value;
                    for (int it = tid; it < n_t * RES_H * 4; it += NT) {
                        const int c = it & 3, hh = (it >> 2) & 3, n = it >> 4;
                        const int k0 = temporal ? n % L : (n / L) * L;
                        const u16 *kp = qs_ + (size_t)k0 * RES_QS + RES_HHD + hh * RES_HD + 8 * c;  // key j: + j * sstride * RES_QS; value: + RES_HHD
                        const u32x4 qw = *reinterpret_cast<const u32x4 *>(qs_ + (size_t)n * RES_QS + hh * RES_HD + 8 * c);
                        u32x4 kw[SP], vw[SP];
#pragma unroll
                        for (int j = 0; j < SP; ++j) {
                            const u16 *kj = kp + (size_t)min(j, S - 1) * sstride * RES_QS;
                            kw[j] = *reinterpret_cast<const u32x4 *>(kj);
                            vw[j] = *reinterpret_cast<const u32x4 *>(kj + RES_HHD);
                        }
                        float dot[SP];
#pragma unroll
                        for (int j = 0; j < SP; ++j) {
                            float d = 0.0f;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                d = fmaf(__uint_as_float(qw[k] << 16), __uint_as_float(kw[j][k] << 16), d);
                                d = fmaf(__uint_as_float(qw[k] & 0xffff0000u), __uint_as_float(kw[j][k] & 0xffff0000u), d);
                            }
                            d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0xB1, 0xF, 0xF, true));
                            d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0x4E, 0xF, 0xF, true));
                            dot[j] = j < S ? d : -INFINITY;
                        }
                        float mx = dot[0];
#pragma unroll
                        for (int j = 1; j < SP; ++j) mx = fmaxf(mx, dot[j]);
                        float sum = 0.0f;
#pragma unroll
                        for (int j = 0; j < SP; ++j) {
                            dot[j] = __builtin_amdgcn_exp2f(dot[j] - mx);
                            sum += dot[j];
                        }
                        const float inv = 1.0f / sum;
                        float o[8];
#pragma unroll
                        for (int d = 0; d < 8; ++d) o[d] = 0.0f;
#pragma unroll
                        for (int j = 0; j < SP; ++j)
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                o[2 * k] = fmaf(dot[j], __uint_as_float(vw[j][k] << 16), o[2 * k]);
                                o[2 * k + 1] = fmaf(dot[j], __uint_as_float(vw[j][k] & 0xffff0000u), o[2 * k + 1]);
                            }
                        const u32x4 w = {pack2(o[0] * inv, o[1] * inv), pack2(o[2] * inv, o[3] * inv), pack2(o[4] * inv, o[5] * inv), pack2(o[6] * inv, o[7] * inv)};
                        *reinterpret_cast<u32x4 *>(zs + (size_t)n * 768 + res_swz(n, 4 * hh + c)) = w;
                    }
                };
                if (S <= 2) quad_attention(std::integral_constant<int, 2>{});
                else if (S <= 4) quad_attention(std::integral_constant<int, 4>{});
                else quad_attention(std::integral_constant<int, 8>{});
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
                // token operand through a 3-deep register ring, two k-steps ahead of the MFMAs (read where it is used, hipcc emits
                // read / lgkmcnt(0) / MFMA per k-step: twelve exposed LDS latencies); the residual rows are requested up front too
                auto zread = [&](int ks, bf16x8 *dst) {
#pragma unroll
                    for (int nt = 0; nt < NNT; ++nt)
                        dst[nt] = as_bf16x8(*reinterpret_cast<const u32x4 *>(zs + (size_t)(16 * nt + r16) * 768 + res_swz(16 * nt + r16, 4 * ks + g4)));
                };
                bf16x8 zf[3][NNT];
                zread(0, zf[0]);
                zread(1, zf[1]);
                float4 hv[NNT];
#pragma unroll
                for (int nt = 0; nt < NNT; ++nt) hv[nt] = *reinterpret_cast<const float4 *>(hs + (size_t)min(16 * nt + r16, n_t - 1) * RES_HS + f0 + 4 * g4);
#pragma unroll
                for (int ks = 0; ks < 12; ++ks) {
                    if (ks + 2 < 12) zread(ks + 2, zf[(ks + 2) % 3]);
#pragma unroll
                    for (int nt = 0; nt < NNT; ++nt) acc[nt] = mfma16(w2f[ks], zf[ks % 3][nt], acc[nt]);
                }
#pragma unroll
                for (int nt = 0; nt < NNT; ++nt) {
                    const int n = 16 * nt + r16;
                    if (n < n_t) {
                        float *hp = hs + (size_t)n * RES_HS + f0 + 4 * g4;
                        hv[nt].x = fmaf(gt.x, acc[nt][0], hv[nt].x);
                        hv[nt].y = fmaf(gt.y, acc[nt][1], hv[nt].y);
                        hv[nt].z = fmaf(gt.z, acc[nt][2], hv[nt].z);
                        hv[nt].w = fmaf(gt.w, acc[nt][3], hv[nt].w);
                        *reinterpret_cast<float4 *>(hp) = hv[nt];
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
        load_embed_operands();  // for the next update's embedding
        float *ln = reinterpret_cast<float *>(smem + LO::qkv);  // fp32 rows [n_t][RES_HS]: q/k/v are dead here (NP * 784 B >= n_t * 528 B)
        layer_norm(2, 1e-6f, ph, ph + D, ln);
        res_barrier();
        const float4 sp = A.step[s];  // (t, ax, am, aw)
        if (h_on) {
            // out[c][n] = sum_d Wo[c][d] ln[n][d]: channels on accumulator rows, tokens on lanes, exact fp32 MFMA (both operands from LDS,
            // 16 bytes per lane and read; rows >= n_t of `ln` are stale q/k/v bytes: they only reach token columns that are not stored)
            const float *wr = wos + (size_t)(16 * h_ct + r16) * RES_HS + 4 * g4, *lr = ln + (size_t)(16 * h_nt + r16) * RES_HS + 4 * g4;
            f32x4v acc = {bo.x, bo.y, bo.z, bo.w};
#pragma unroll
            for (int j = 0; j < D / 16; ++j)
                acc = mfma16_f32x4(*reinterpret_cast<const float4 *>(wr + 16 * j), *reinterpret_cast<const float4 *>(lr + 16 * j), acc);
            const int n = 16 * h_nt + r16;
            if (n < n_t && h_c0 < C) {
                const size_t e = xoff + (size_t)n * C + h_c0;
                float *xp = xs + n * RES_XS + h_c0;
                const float4 xo = *reinterpret_cast<const float4 *>(xp);
                float xn[4] = {sp.y * xo.x + sp.z * acc[0], sp.y * xo.y + sp.z * acc[1], sp.y * xo.z + sp.z * acc[2], sp.y * xo.w + sp.z * acc[3]};
                if (sp.w != 0.0f) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        xn[k] += sp.w * (A.noise ? A.noise[(size_t)(A.step0 + s) * A.noise_step_stride + e + k] : philox_normal(A.seed, A.step0 + s, A.elem_offset + e + k));
                }
                *reinterpret_cast<float4 *>(xp) = make_float4(xn[0], xn[1], xn[2], xn[3]);
                if (A.trace) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) A.trace[(size_t)(A.step0 + s) * A.trace_step_stride + e + k] = xn[k];
                }
            }
        }
        if (s + 1 < A.n_steps) par_commit(qh + 1);
        res_barrier();
        RES_STAMP(5);
    }
    for (int i = tid; i < n_t * C; i += NT) A.x[xoff + i] = xs[(i / C) * RES_XS + (i % C)];
}
