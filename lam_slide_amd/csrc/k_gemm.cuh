// bf16 MFMA GEMM for linear1 / linear2 of a ParallelMLPAttentionV2 block (mmdit.py:240-249), with the
// surrounding element-wise work fused into the epilogue.
//
// Orientation: the kernel computes the TRANSPOSED product  Ct[f][n] = sum_k W[f][k] * X[n][k]
// (W = nn.Linear weight [out,in], X = activations [tokens, in]; both are k-contiguous, so both MFMA
// operands are plain 16-byte row reads).  With features on the accumulator rows and tokens on the
// lanes, one lane holds 16 of the 32 features of a tile for ONE token: per-head RMS norm and the RoPE
// pair rotation are in-register (+ one exchange with lane^32), and every store is 4 consecutive
// features of one token.
//
// Tiling: workgroup tile BF x BT (features x tokens), BK = 64, NWF x NWT waves, each wave owns
// (BF/NWF) x (BT/NWT) as 32x32 MFMA tiles.  LDS double buffer, register-staged global loads
// (issue early / write late), XOR-swizzled 16-byte chunks so ds_read_b128 fragment reads are
// conflict-free:  physical chunk = chunk ^ ((row >> 1) & 7)  for 128-byte rows.
#pragma once
#include "common.cuh"

constexpr int GEMM_BK = 64;

struct GemmArgs {
    const u16 *W;  // [F][K] bf16
    const u16 *X;  // [N][K] bf16
    int F, N, K;
};

__device__ __forceinline__ int swz_off(int row, int chunk) {  // byte offset inside a [rows][64] bf16 tile
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// Epilogue protocol: `epi.token(n, valid)` precomputes everything that depends only on the lane's token (row
// pointers, position, modulation row) once per 32-token column; `epi.tile(acc, f0, tok, hf)` consumes one 32x32
// accumulator tile whose rows are features f0 + acc_row(reg, hf) of that token.
template <int MI, int NJ, class Epi>
__device__ __forceinline__ void run_epilogue(const Epi &epi, f32x16 (&acc)[MI][NJ], int f_wave, int n_lane, int hf, int F, int N) {
    typename Epi::Tok tok[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) tok[j] = epi.token(n_lane + j * 32, n_lane + j * 32 < N);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int f0 = f_wave + i * 32;
        if (f0 >= F) continue;  // wave-uniform
#pragma unroll
        for (int j = 0; j < NJ; ++j) epi.tile(acc[i][j], f0, tok[j], hf);
    }
}

template <int BF, int BT, int NWF, int NWT, class Epi>
__global__ void __launch_bounds__(NWF *NWT * 64) k_gemm_wx(GemmArgs g, Epi epi) {
    constexpr int NT = NWF * NWT * 64;
    constexpr int WF = BF / NWF, WT = BT / NWT;  // wave tile
    constexpr int MI = WF / 32, NJ = WT / 32;    // MFMA tiles per wave
    constexpr int W_LOADS = BF * 8 / NT, X_LOADS = BT * 8 / NT;  // 16-byte chunks per thread per k-tile
    static_assert(BF * 8 % NT == 0 && BT * 8 % NT == 0, "tile/threads mismatch");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *Ws = smem;                           // [2][BF][64] bf16
    char *Xs = smem + 2 * BF * 128;            // [2][BT][64] bf16

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wf = wave / NWT, wt = wave % NWT;
    const int r = lane & 31, hf = lane >> 5;

    const int ntt = (g.N + BT - 1) / BT, nft = (g.F + BF - 1) / BF;
    const int tile = xcd_remap(blockIdx.x, ntt * nft);
    const int f_base = (tile % nft) * BF, n_base = (tile / nft) * BT;

    // global -> register staging: thread covers row (tid/8 + i*NT/8), chunk tid%8
    const int lrow = tid >> 3, lchunk = tid & 7;
    u32x4 wreg[W_LOADS], xreg[X_LOADS];
    const u16 *wsrc[W_LOADS];
    const u16 *xsrc[X_LOADS];
#pragma unroll
    for (int i = 0; i < W_LOADS; ++i) {
        const int f = min(f_base + lrow + i * (NT / 8), g.F - 1);
        wsrc[i] = g.W + (size_t)f * g.K + lchunk * 8;
    }
#pragma unroll
    for (int i = 0; i < X_LOADS; ++i) {
        const int n = min(n_base + lrow + i * (NT / 8), g.N - 1);
        xsrc[i] = g.X + (size_t)n * g.K + lchunk * 8;
    }
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < W_LOADS; ++i) wreg[i] = *reinterpret_cast<const u32x4 *>(wsrc[i] + kt * GEMM_BK);
#pragma unroll
        for (int i = 0; i < X_LOADS; ++i) xreg[i] = *reinterpret_cast<const u32x4 *>(xsrc[i] + kt * GEMM_BK);
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < W_LOADS; ++i)
            *reinterpret_cast<u32x4 *>(Ws + buf * BF * 128 + swz_off(lrow + i * (NT / 8), lchunk)) = wreg[i];
#pragma unroll
        for (int i = 0; i < X_LOADS; ++i)
            *reinterpret_cast<u32x4 *>(Xs + buf * BT * 128 + swz_off(lrow + i * (NT / 8), lchunk)) = xreg[i];
    };

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int nk = g.K / GEMM_BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const char *wb = Ws + buf * BF * 128, *xb = Xs + buf * BT * 128;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 af[MI], bfr[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i)
                af[i] = as_bf16x8(*reinterpret_cast<const u32x4 *>(wb + swz_off(wf * WF + i * 32 + r, 2 * ks + hf)));
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                bfr[j] = as_bf16x8(*reinterpret_cast<const u32x4 *>(xb + swz_off(wt * WT + j * 32 + r, 2 * ks + hf)));
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = mfma32(af[i], bfr[j], acc[i][j]);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    run_epilogue<MI, NJ>(epi, acc, f_base + wf * WF, n_base + wt * WT + r, hf, g.F, g.N);
}

// ---------------------------------------------------------------------------------------------------
// Same product, operands streamed by LDS-DMA (global_load_lds_dwordx4) into an NS-deep ring of k-tiles, so
// NS-1 k-tiles of loads stay in flight across the per-k-tile barrier (counted s_waitcnt vmcnt, raw s_barrier)
// and no VGPRs or ds_write instructions are spent on staging.  One stage image = [BF rows of W | BT rows of X]
// x 64 bf16; a wave instruction lands 1 KiB = 8 rows x 128 B lane-linearly, so the read-side XOR swizzle is
// applied to the per-lane SOURCE address (both sides use the same involution chunk ^ ((row >> 1) & 7)).
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// byte offset of 16-byte chunk `chunk` of row `row` inside a [rows][BK] bf16 stage image (BK = 64 or 32)
template <int BK>
__device__ __forceinline__ int swz_bk(int row, int chunk) {
    if (BK == 64) return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
    return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4);
}

template <int BF, int BT, int NWF, int NWT, int BK, int NS, class Epi>
__global__ void __launch_bounds__(NWF *NWT * 64) k_gemm_glds(GemmArgs g, Epi epi) {
    constexpr int NW = NWF * NWT;
    constexpr int WF = BF / NWF, WT = BT / NWT;
    constexpr int MI = WF / 32, NJ = WT / 32;
    constexpr int ROWB = BK * 2;                 // bytes per stage row
    constexpr int CPR = ROWB / 16;               // 16-byte chunks per row (8 or 4)
    constexpr int RPP = 1024 / ROWB;             // rows per 1 KiB LDS-DMA piece (8 or 16)
    constexpr int STAGE = (BF + BT) * ROWB;      // bytes
    constexpr int PIECES = (BF + BT) / RPP;      // 1 KiB pieces per stage
    constexpr int LPS = PIECES / NW;             // LDS-DMA instructions per wave per stage
    static_assert(PIECES % NW == 0 && BF % 16 == 0 && BT % 16 == 0 && NS >= 2 && NS <= 6 && (BK == 32 || BK == 64), "bad tiling");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wf = wave / NWT, wt = wave % NWT;
    const int r = lane & 31, hf = lane >> 5;

    const int ntt = (g.N + BT - 1) / BT, nft = (g.F + BF - 1) / BF;
    const int tile = xcd_remap(blockIdx.x, ntt * nft);
    const int f_base = (tile % nft) * BF, n_base = (tile / nft) * BT;

    const u16 *src[LPS];
#pragma unroll
    for (int i = 0; i < LPS; ++i) {
        const int piece = wave + i * NW;
        const int row = piece * RPP + lane / CPR;
        const int chunk = (swz_bk<BK>(row, lane % CPR) - row * ROWB) >> 4;  // logical chunk stored at this physical slot
        src[i] = row < BF ? g.W + (size_t)min(f_base + row, g.F - 1) * g.K + chunk * 8
                          : g.X + (size_t)min(n_base + row - BF, g.N - 1) * g.K + chunk * 8;
    }
    auto issue = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < LPS; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src[i] + kt * BK),
                                             (LDS_PTR(void))(smem + buf * STAGE + (wave + i * NW) * 1024), 16, 0, 0);
    };

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int nk = g.K / BK;
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nk) issue(s, s);

    for (int kt = 0; kt < nk; ++kt) {
        // k-tiles after this one that are already in flight: min(nk - 1 - kt, NS - 2); wait for everything older
        const int ahead = min(nk - 1 - kt, NS - 2);
        switch (ahead) {
            case 0: wait_vmcnt<0>(); break;
            case 1: wait_vmcnt<LPS>(); break;
            case 2: wait_vmcnt<2 * LPS>(); break;
            case 3: wait_vmcnt<3 * LPS>(); break;
            default: wait_vmcnt<4 * LPS>(); break;
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + NS - 1 < nk) issue(kt + NS - 1, (kt + NS - 1) % NS);
        const char *sb = smem + (kt % NS) * STAGE;
        // Fragments are double-buffered across the 16-deep k sub-steps: the ds_read_b128 of sub-step ks+1 are
        // issued before the MFMAs of sub-step ks (hipcc otherwise emits read / lgkmcnt(0) / 2 MFMAs chains that
        // expose the LDS latency every two MFMAs); only the first sub-step of a k-tile waits on LDS.
        auto load_frags = [&](int ks, bf16x8(&a)[MI], bf16x8(&b)[NJ]) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
                a[i] = as_bf16x8(*reinterpret_cast<const u32x4 *>(sb + swz_bk<BK>(wf * WF + i * 32 + r, 2 * ks + hf)));
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                b[j] = as_bf16x8(*reinterpret_cast<const u32x4 *>(sb + swz_bk<BK>(BF + wt * WT + j * 32 + r, 2 * ks + hf)));
        };
        auto mfma_all = [&](const bf16x8(&a)[MI], const bf16x8(&b)[NJ]) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
        };
        constexpr int KS = BK / 16;
        bf16x8 a0[MI], b0[NJ], a1[MI], b1[NJ];
        load_frags(0, a0, b0);
#pragma unroll
        for (int ks = 0; ks < KS; ks += 2) {
            load_frags(ks + 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_all(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 2 < KS) load_frags(ks + 2, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_all(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    run_epilogue<MI, NJ>(epi, acc, f_base + wf * WF, n_base + wt * WT + r, hf, g.F, g.N);
}

// ---------------------------------------------------------------------------------------------------
// linear1 epilogue: + bias; q/k heads: RMS norm * scale, RoPE (q additionally * softmax scale * log2 e);
// v: as is; mlp: erf-GELU.  Output bf16:  qkv[n][0 .. 3*HHD)  and  z[n][HHD .. HHD+M).
// (mmdit.py:241-248, 129-148, 85-90, 11-18)
template <int HDP>
struct EpiLinear1 {
    const float *bias;     // [F1]
    const float *qs, *ks;  // [HDP]
    const float2 *rope;    // [n_pos][HDP/2] (cos, sin)
    u16 *qkv;              // [N][3*HHD]
    u16 *z;                // [N][HHD + M]
    int HHD, M;
    int pos_div, pos_mod;  // position of token n inside its sequence: (n / pos_div) % pos_mod
    float inv_hd;          // 1 / true head_dim
    float q_premul;        // head_dim^-0.5 * log2(e), folded into q for the exp2-based softmax

    struct Tok {
        u16 *qkv_row, *z_row;  // z_row is pre-offset so that feature f lands at z_row[f]
        const float2 *tab;
        bool valid;
    };
    __device__ __forceinline__ Tok token(int n, bool valid) const {
        Tok t;
        const unsigned nn = valid ? (unsigned)n : 0u;
        t.qkv_row = qkv + (size_t)nn * (3 * HHD);
        t.z_row = z + (size_t)nn * (HHD + M) - 2 * HHD;
        t.tab = rope + (size_t)((nn / (unsigned)pos_div) % (unsigned)pos_mod) * (HDP / 2);
        t.valid = valid;
        return t;
    }

    __device__ __forceinline__ void tile(const f32x16 &acc, int f0, const Tok &t, int hf) const {
        float v[16];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const float4 b = *reinterpret_cast<const float4 *>(bias + f0 + 8 * q4 + 4 * hf);
            v[4 * q4] = acc[4 * q4] + b.x;
            v[4 * q4 + 1] = acc[4 * q4 + 1] + b.y;
            v[4 * q4 + 2] = acc[4 * q4 + 2] + b.z;
            v[4 * q4 + 3] = acc[4 * q4 + 3] + b.w;
        }
        const int sec = f0 / HHD;  // 0 q, 1 k, 2 v, >= 3 mlp (wave-uniform: HHD is a multiple of 32)
        if (sec < 2) {
            const float *sc = sec == 0 ? qs : ks;
            const float post = sec == 0 ? q_premul : 1.0f;
            constexpr int GROUPS = 32 / HDP;  // heads per 32-feature tile
            constexpr int RPG = 16 / GROUPS;  // registers per head
#pragma unroll
            for (int gi = 0; gi < GROUPS; ++gi) {
                float ss = 0.0f;
#pragma unroll
                for (int e = 0; e < RPG; ++e) ss = fmaf(v[gi * RPG + e], v[gi * RPG + e], ss);
                ss += xhalf(ss);
                const float rr = rsqrtf(fmaf(ss, inv_hd, 1e-6f)) * post;
#pragma unroll
                for (int q4 = 0; q4 < RPG / 4; ++q4) {
                    const int e = gi * RPG + 4 * q4;
                    const int d = (8 * q4 + 4 * hf) & (HDP - 1);  // first of 4 consecutive channels inside the head
                    const float4 s4 = *reinterpret_cast<const float4 *>(sc + d);
                    const float4 cs = *reinterpret_cast<const float4 *>(t.tab + (d >> 1));  // (cos0, sin0, cos1, sin1)
                    const float x0 = v[e] * rr * s4.x, x1 = v[e + 1] * rr * s4.y;
                    const float x2 = v[e + 2] * rr * s4.z, x3 = v[e + 3] * rr * s4.w;
                    v[e] = cs.x * x0 - cs.y * x1;
                    v[e + 1] = cs.y * x0 + cs.x * x1;
                    v[e + 2] = cs.z * x2 - cs.w * x3;
                    v[e + 3] = cs.w * x2 + cs.z * x3;
                }
            }
        } else if (sec >= 3) {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = gelu_fast(v[e]);
        }
        if (!t.valid) return;
        u16 *dst = (sec < 3 ? t.qkv_row : t.z_row) + f0 + 4 * hf;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            u32x2 pk = {pack2(v[4 * q4], v[4 * q4 + 1]), pack2(v[4 * q4 + 2], v[4 * q4 + 3])};
            *reinterpret_cast<u32x2 *>(dst + 8 * q4) = pk;
        }
    }
};

// linear2 epilogue: h[n][f] += gate[b][f] * (acc + bias[f])   (latent_si_v31.py:53,60; fp32 residual)
struct EpiLinear2 {
    const float *bias;  // [D]
    const float *gate;  // mods + gate offset, row stride mod_stride
    float *h;           // [N][D]
    int D, mod_stride, tokens_per_traj;

    struct Tok {
        float *h_row;
        const float *gate_row;
        bool valid;
    };
    __device__ __forceinline__ Tok token(int n, bool valid) const {
        const unsigned nn = valid ? (unsigned)n : 0u;
        return Tok{h + (size_t)nn * D, gate + (size_t)(nn / (unsigned)tokens_per_traj) * mod_stride, valid};
    }
    __device__ __forceinline__ void tile(const f32x16 &acc, int f0, const Tok &t, int hf) const {
        if (!t.valid) return;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int f = f0 + 8 * q4 + 4 * hf;
            const float4 b = *reinterpret_cast<const float4 *>(bias + f);
            const float4 gt = *reinterpret_cast<const float4 *>(t.gate_row + f);
            float4 hv = *reinterpret_cast<float4 *>(t.h_row + f);
            hv.x = fmaf(gt.x, acc[4 * q4] + b.x, hv.x);
            hv.y = fmaf(gt.y, acc[4 * q4 + 1] + b.y, hv.y);
            hv.z = fmaf(gt.z, acc[4 * q4 + 2] + b.z, hv.z);
            hv.w = fmaf(gt.w, acc[4 * q4 + 3] + b.w, hv.w);
            *reinterpret_cast<float4 *>(t.h_row + f) = hv;
        }
    }
};
