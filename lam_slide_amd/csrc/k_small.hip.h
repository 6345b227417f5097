// fp32 kernels around the MFMA blocks: RoPE table, conditioning vector and modulation tables, input
// embedding, LayerNorm+modulate, output projection fused with the sampler's state update, device noise.
// These carry < 1 % of the FLOPs and stay in fp32 on purpose ("Accuracy target").
#pragma once
#include "common.hip.h"

// ---------------------------------------------------------------------------------------------------
// RoPE table: tab[p][j] = (cos, sin)(p * theta^(-2j/hd)), angle in fp64 then rounded (mmdit.py:75-82).
// j >= hd/2 (head padding) gets the identity rotation.
__global__ void k_rope_table(float2 *tab, int n_pos, int hd, int hdp, float theta) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = hdp / 2;
    if (i >= n_pos * half) return;
    const int p = i / half, j = i % half;
    float2 v = make_float2(1.0f, 0.0f);
    if (2 * j < hd) {
        const double omega = 1.0 / pow((double)theta, (double)(2 * j) / (double)hd);
        const double ang = (double)p * omega;
        v = make_float2((float)cos(ang), (float)sin(ang));
    }
    tab[i] = v;
}

// RoPE table with the QK-norm scales of one attention block folded in (linear1 epilogue, k_gemm.hip.h):
//   out[p][j] = (c s0, sn s1, sn s0, c s1),  (c, sn) = (cos, sin)(p * theta^(-2j/hd)),  s0 = scale[2j], s1 = scale[2j+1]
// so that the rotated, scaled pair is  (o0, o1) = rr * (c s0 x0 - sn s1 x1,  sn s0 x0 + c s1 x1)  for the RMS factor rr
// (mmdit.py:129-148 then 85-90).  One launch covers up to 16 (block, q|k) tables.
struct RopeScaledJobs {
    float4 *out[16];
    const float *scale[16];
    int n_pos[16];
    float *sq_bound[16];  // or NULL: receives head_dim * max_d scale_d^2, an upper bound of the squared norm of every vector this table scales
                          // (RMS-normalised over head_dim, times scale, rotated: k_attention_stream's softmax shift)
    int n_jobs;
};
__global__ void k_rope_scaled(RopeScaledJobs jobs, int hd, int hdp, float theta) {
    const int job = blockIdx.y;
    if (job >= jobs.n_jobs) return;
    if (blockIdx.x == 0 && threadIdx.x == 0 && jobs.sq_bound[job]) {
        float mx = 0.0f;
        for (int d = 0; d < hd; ++d) mx = fmaxf(mx, jobs.scale[job][d] * jobs.scale[job][d]);
        *jobs.sq_bound[job] = mx * (float)hd;
    }
    const int half = hdp / 2, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= jobs.n_pos[job] * half) return;
    const int p = i / half, j = i % half;
    float c = 1.0f, sn = 0.0f;
    if (2 * j < hd) {
        const double omega = 1.0 / pow((double)theta, (double)(2 * j) / (double)hd);
        const double ang = (double)p * omega;
        c = (float)cos(ang);
        sn = (float)sin(ang);
    }
    const float s0 = jobs.scale[job][2 * j], s1 = jobs.scale[job][2 * j + 1];
    jobs.out[job][i] = make_float4(c * s0, sn * s1, sn * s0, c * s1);
}

// q / k / v planes [3 H][npad][hdp] (Lin1Args::planes) -> token-major rows [n][3 H hdp]: lsl_debug_taps hands the taps out in one layout
__global__ void k_planes_to_rows(u16 *rows, const u16 *planes, int n, int npad, int heads3, int hdp) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // one 16-byte chunk each
    const int cph = hdp / 8;
    if (i >= (long)n * heads3 * cph) return;
    const int c = (int)(i % cph), p = (int)((i / cph) % heads3);
    const long tok = (i / cph) / heads3;
    *reinterpret_cast<u32x4 *>(rows + (tok * heads3 + p) * hdp + 8 * c) = *reinterpret_cast<const u32x4 *>(planes + ((long)p * npad + tok) * hdp + 8 * c);
}

// ---------------------------------------------------------------------------------------------------
// Sinusoidal time features (mmdit.py:93-115): args = (1000 t) * freqs in fp32, [cos | sin].
// t_ptr == nullptr: every row uses t_scalar.
__global__ void k_time_features(float *out, const float *t_ptr, float t_scalar, const float *freqs, int rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * 128) return;
    const int b = i >> 7, k = i & 127;
    const float t = 1000.0f * (t_ptr ? t_ptr[b] : t_scalar);
    const float a = t * freqs[k];
    out[b * 256 + k] = cosf(a);
    out[b * 256 + 128 + k] = sinf(a);
}

// The same features for the rows (step s, trajectory b) of several sampler steps at once (k_resident.hip.h computes the modulation
// tables of a whole group of state updates before its single launch): row = s * rows_per_step + b, time = t[s].
struct StepTimes {
    float t[48];
};
__global__ void k_time_features_steps(float *out, StepTimes st, int n_steps, int rows_per_step, const float *freqs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_steps * rows_per_step * 128) return;
    const int row = i >> 7, k = i & 127;
    const float a = 1000.0f * st.t[row / rows_per_step] * freqs[k];
    out[row * 256 + k] = cosf(a);
    out[row * 256 + 128 + k] = sinf(a);
}

// ---------------------------------------------------------------------------------------------------
// out[b][o] = post(sum_i W[o][i] * pre(in[b][i]) + bias[o] + add[b % add_mod][o]) (add_mod = 0: add[b]); one wave per output feature,
// the weight row stays in registers while the wave walks the batch (gridDim.y > 1: the rows in gridDim.y contiguous ranges; a row's sum
// does not depend on how the rows are cut).  I <= 512.
template <bool PRE_SILU, bool POST_SILU>
__global__ void __launch_bounds__(256) k_dense_rows(float *out, const float *in, const float *W, const float *bias,
                                                    const float *add, int rows, int I, int O, int add_stride, int add_mod) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= O) return;
    float w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = lane + 64 * k;
        w[k] = i < I ? W[(size_t)o * I + i] : 0.0f;
    }
    const float bo = bias[o];
    const int rpb = (rows + (int)gridDim.y - 1) / (int)gridDim.y, b_hi = min(rows, ((int)blockIdx.y + 1) * rpb);
    // four rows per trip: their inputs are requested together (a row's sum keeps its own order; one row per trip waited a full L2 round
    // trip per row - 50 rows of the grouped modulation tables: 257 us)
    for (int b0 = (int)blockIdx.y * rpb; b0 < b_hi; b0 += 4) {
        float v[4][8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = b0 + r;
            if (b >= b_hi) break;  // (uniform)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = lane + 64 * k;
                v[r][k] = i < I ? in[(size_t)b * I + i] : 0.0f;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = b0 + r;
            if (b >= b_hi) break;
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = lane + 64 * k;
                if (i < I) s = fmaf(w[k], PRE_SILU ? silu(v[r][k]) : v[r][k], s);
            }
            s = wave_sum(s);
            if (lane == 0) {
                s += bo;
                if (add) s += add[(size_t)(add_mod ? b % add_mod : b) * add_stride + o];
                if (POST_SILU) s = silu(s);
                out[(size_t)b * O + o] = s;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Same contract as k_dense_rows for MANY rows (class-conditioned batches: one conditioning row per trajectory, up to
// thousands): a 64 x 64 (rows x outputs) register-tiled fp32 product, K in chunks of 16 through LDS (stored
// k-major so the float4 reads of 4 rows / 4 outputs are conflict-free).  The wave-per-output kernel above walks the
// rows serially and took 40 of 60 ms per sampling call of the pedestrian model at 1280 rows.
template <bool PRE_SILU, bool POST_SILU>
__global__ void __launch_bounds__(256) k_dense_tiled(float *out, const float *in, const float *W, const float *bias,
                                                     const float *add, int rows, int I, int O, int add_stride, int add_mod) {
    __shared__ __attribute__((aligned(16))) float As[16][64 + 4];
    __shared__ __attribute__((aligned(16))) float Ws[16][64 + 4];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int r0 = blockIdx.y * 64, o0 = blockIdx.x * 64;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0f;
    const int lr = tid >> 2, lk = (tid & 3) * 4;  // loader: row (or output) lr, k offset lk..lk+3
    for (int k0 = 0; k0 < I; k0 += 16) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), w = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r0 + lr < rows && k0 + lk < I) a = *reinterpret_cast<const float4 *>(in + (size_t)(r0 + lr) * I + k0 + lk);
        if (o0 + lr < O && k0 + lk < I) w = *reinterpret_cast<const float4 *>(W + (size_t)(o0 + lr) * I + k0 + lk);
        if (PRE_SILU) a = make_float4(silu(a.x), silu(a.y), silu(a.z), silu(a.w));
        __syncthreads();
        As[lk][lr] = a.x; As[lk + 1][lr] = a.y; As[lk + 2][lr] = a.z; As[lk + 3][lr] = a.w;
        Ws[lk][lr] = w.x; Ws[lk + 1][lr] = w.y; Ws[lk + 2][lr] = w.z; Ws[lk + 3][lr] = w.w;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float4 av = *reinterpret_cast<const float4 *>(&As[k][4 * ty]);
            const float4 wv = *reinterpret_cast<const float4 *>(&Ws[k][4 * tx]);
            const float ar[4] = {av.x, av.y, av.z, av.w}, wr[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(ar[i], wr[j], acc[i][j]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int b = r0 + 4 * ty + i;
        if (b >= rows) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int o = o0 + 4 * tx + j;
            if (o >= O) continue;
            float v = acc[i][j] + bias[o];
            if (add) v += add[(size_t)(add_mod ? b % add_mod : b) * add_stride + o];
            if (POST_SILU) v = silu(v);
            out[(size_t)b * O + o] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// The same contract once more for the SHORT chains of tiny GEMMs in front of the trajectory-resident kernel (rows = state updates x
// conditioning rows <= a few hundred, I = 128 / 256): there each launch is pure latency, and the 64 x 64 FMA tiles above put a whole
// 256-deep k loop on two or eight workgroups (13-22 us per launch, five launches per sampling call).  Here a workgroup owns a 32 x 32
// output tile, its four waves each take a quarter of k on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32: exact fp32 multiply-add),
// every operand byte of a wave is requested before its first MFMA, and the four partial tiles meet in LDS in a fixed order.
// One kernel for every batch size (the choice depends on the model only), so a trajectory's tables do not depend on its batch.
// Also taken by the general path for class-conditioned batches of the usual widths (I = 128 / 256): NBA +0.9 % at 1 024 rows, +3.6 % at 64.
typedef __attribute__((ext_vector_type(16))) float f32x16_d;
template <bool PRE_SILU, bool POST_SILU, int KQ>  // KQ = I / 4 (k per wave): 32 or 64
__global__ void __launch_bounds__(256) k_dense_mfma(float *out, const float *in, const float *W, const float *bias, const float *add,
                                                    int rows, int I, int O, int add_stride, int add_mod) {
    __shared__ float Ps[4][32][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hf = lane >> 5;
    const int r0 = blockIdx.y * 32, o0 = blockIdx.x * 32;
    const float *ap = in + (size_t)min(r0 + r, rows - 1) * I + wave * KQ + 4 * hf;
    const float *wp = W + (size_t)min(o0 + r, O - 1) * I + wave * KQ + 4 * hf;
    float4 a[KQ / 8], w[KQ / 8];  // lane (r, hf): k = 8 j + 4 hf .. + 3 of the wave's slice, the same k order for both operands
#pragma unroll
    for (int j = 0; j < KQ / 8; ++j) {
        a[j] = *reinterpret_cast<const float4 *>(ap + 8 * j);
        w[j] = *reinterpret_cast<const float4 *>(wp + 8 * j);
    }
    f32x16_d acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
    for (int j = 0; j < KQ / 8; ++j) {
        float4 av = a[j];
        if (PRE_SILU) av = make_float4(silu(av.x), silu(av.y), silu(av.z), silu(av.w));
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, w[j].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, w[j].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, w[j].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, w[j].w, acc, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) Ps[wave][acc_row(e, hf)][r] = acc[e];  // (row of the tile = input row, column = output)
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int idx = tid + 256 * q, row = idx >> 5, col = idx & 31, b = r0 + row, o = o0 + col;
        if (b >= rows || o >= O) continue;
        float v = ((Ps[0][row][col] + Ps[1][row][col]) + Ps[2][row][col]) + Ps[3][row][col] + bias[o];
        if (add) v += add[(size_t)(add_mod ? b % add_mod : b) * add_stride + o];
        if (POST_SILU) v = silu(v);
        out[(size_t)b * O + o] = v;
    }
}

// ---------------------------------------------------------------------------------------------------
// Small-K projection C -> D (x_in / cond_to_emb, latent_si_v31.py:172), exact fp32 FMA chains over c.
//   MODE 0 (once per sample): out = in @ W^T + bias + bias2 + mask_emb[mask]      (cond_to_emb part)
//   MODE 1 (every evaluation): out = in @ W^T + base                              (x_in part + cached)
// Workgroup = EMB_TOK tokens.  A thread owns 4 consecutive output columns (their weight rows live in registers,
// all output traffic is 16 bytes per lane) and walks every (256 / (D/4))-th token; the token's inputs are read
// from LDS as wave-uniform 16-byte pieces.
constexpr int EMB_TOK = 64;
template <int CMAX, int ND>
constexpr size_t embed_w_lds_bytes(int D) { return (size_t)(D / ND) * (ND * CMAX / 4 + 1) * 16; }
// STATS (round 6, ln_fuse handles; D a multiple of 64 ND): beside `out`, every wave leaves (mean, sum of squared deviations) of each token row
// over its 64 ND features in stats[D / (64 ND)][npad] - what k_ln_finalize (group size 64 ND) turns into the first sub-block's LayerNorm
// statistics; `out` itself is bit-identical to the plain instance's.
template <int CMAX, int MODE, int ND, bool STATS = false>  // ND output columns per thread: 4 (C <= 32), 2 (C <= 96), 1
__global__ void __launch_bounds__(256) k_embed(float *out, const float *in, const float *W, const float *bias,
                                               const float *bias2, const float *mask_emb, const int64_t *mask,
                                               const float *base, int N, int C, int D, int w_lds, int tok, float2 *stats = nullptr, int npad = 0) {
    __shared__ __attribute__((aligned(16))) float xs[EMB_TOK][CMAX];
    extern __shared__ __attribute__((aligned(16))) float4 embed_w_lds[];  // w_lds: (D / ND) x (ND * CMAX / 4 + 1) 16-byte pieces
    const int n_tiles = (N + tok - 1) / tok;  // tok <= EMB_TOK tokens per tile (small launches: smaller tiles, so that every CU gets work)
    const int DQ = D / ND, groups = DQ >= 256 ? 1 : 256 / DQ;
    // persistent workgroups: a thread's weight rows are fetched once and stay in registers while it walks the token tiles
    for (int dq = threadIdx.x % (DQ < 256 ? DQ : 256); dq < DQ; dq += 256) {
        const int tg = DQ >= 256 ? 0 : threadIdx.x / DQ, d = ND * dq;
        const bool active = tg < groups;  // (threads beyond groups * DQ only help with the staging)
        float w[ND][CMAX];
        if (w_lds) {
            // C == CMAX: a thread's ND weight rows are ND * C * 4 contiguous bytes.  Fetched by the lanes directly, every load instruction
            // touches 64 different 128-byte lines (16 bytes of each): ~10 us of prologue per workgroup, two thirds of the launch at
            // 7 680 tokens.  Here the matrix is read in order (1 KiB per wave and instruction) into LDS, one (PIECES + 1) * 16-byte row per
            // owner, and every thread picks its pieces from there: the same values in the same registers.
            constexpr int PIECES = ND * CMAX / 4;
            const int n4 = D * CMAX / 4;
            const float4 *W4 = reinterpret_cast<const float4 *>(W);
#pragma unroll 8
            for (int i = threadIdx.x; i < n4; i += 256) embed_w_lds[(i / PIECES) * (PIECES + 1) + (i % PIECES)] = W4[i];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < ND; ++j)
#pragma unroll
                for (int c = 0; c < CMAX; c += 4) {
                    const float4 w4 = active ? embed_w_lds[dq * (PIECES + 1) + (j * CMAX + c) / 4] : make_float4(0.f, 0.f, 0.f, 0.f);
                    w[j][c] = w4.x; w[j][c + 1] = w4.y; w[j][c + 2] = w4.z; w[j][c + 3] = w4.w;
                }
        } else if ((C & 3) == 0) {  // 16-byte loads (every shipped model): a quarter of the load instructions of the per-element form, which made
                             // this prologue ~20 us of a 39 us launch at 23 000 tokens (md17 reference shape)
#pragma unroll
            for (int j = 0; j < ND; ++j)
#pragma unroll
                for (int c = 0; c < CMAX; c += 4) {
                    const float4 w4 = (active && c < C) ? *reinterpret_cast<const float4 *>(W + (size_t)(d + j) * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                    w[j][c] = w4.x; w[j][c + 1] = w4.y; w[j][c + 2] = w4.z; w[j][c + 3] = w4.w;
                }
        } else {
#pragma unroll
            for (int j = 0; j < ND; ++j)
#pragma unroll
                for (int c = 0; c < CMAX; ++c) w[j][c] = (active && c < C) ? W[(size_t)(d + j) * C + c] : 0.0f;
        }
        float b0[ND];
#pragma unroll
        for (int j = 0; j < ND; ++j) b0[j] = (MODE == 0 && active) ? bias[d + j] + bias2[d + j] : 0.0f;
        for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
            const int n0 = tile * tok;
            const int ntok = min(tok, N - n0);
            __syncthreads();  // the previous tile's products have finished reading xs
            for (int i = threadIdx.x; i < tok * CMAX; i += 256) {
                const int tkn = i / CMAX, c = i % CMAX;
                xs[tkn][c] = (tkn < ntok && c < C) ? in[(size_t)(n0 + tkn) * C + c] : 0.0f;
            }
            __syncthreads();
            if (!active) continue;
            // tokens in groups of PF: the group's `base` rows (the only HBM reads of the loop) are all requested before the first
            // product, so PF loads are in flight per thread instead of one or two
            constexpr int PF = 4;
            for (int t0 = tg; t0 < ntok; t0 += PF * groups) {
                float add[PF][ND];
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const int tkn = min(t0 + u * groups, ntok - 1);
                    const size_t row = (size_t)(n0 + tkn) * D + d;
                    if (MODE == 0) {
                        const float *me = mask_emb + (mask[n0 + tkn] != 0 ? D : 0) + d;
#pragma unroll
                        for (int j = 0; j < ND; ++j) add[u][j] = b0[j] + me[j];
                    } else {
#pragma unroll
                        for (int j = 0; j < ND; ++j) add[u][j] = base[row + j];
                    }
                }
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const int tkn = t0 + u * groups;
                    if (tkn >= ntok) break;
                    const size_t row = (size_t)(n0 + tkn) * D + d;
                    float acc[ND];
#pragma unroll
                    for (int j = 0; j < ND; ++j) acc[j] = 0.0f;
#pragma unroll
                    for (int c = 0; c < CMAX; c += 4) {
                        const float4 xv = *reinterpret_cast<const float4 *>(&xs[tkn][c]);
#pragma unroll
                        for (int j = 0; j < ND; ++j) acc[j] = fmaf(xv.x, w[j][c], acc[j]);
#pragma unroll
                        for (int j = 0; j < ND; ++j) acc[j] = fmaf(xv.y, w[j][c + 1], acc[j]);
#pragma unroll
                        for (int j = 0; j < ND; ++j) acc[j] = fmaf(xv.z, w[j][c + 2], acc[j]);
#pragma unroll
                        for (int j = 0; j < ND; ++j) acc[j] = fmaf(xv.w, w[j][c + 3], acc[j]);
                    }
#pragma unroll
                    for (int j = 0; j < ND; ++j) acc[j] = acc[j] + add[u][j];
#pragma unroll
                    for (int j = 0; j < ND; ++j) out[row + j] = acc[j];
                    if constexpr (STATS) {  // (the wave's 64 lanes hold 64 ND consecutive features of ONE token: DQ is a multiple of 64, host-checked)
                        float s = 0.0f;
#pragma unroll
                        for (int j = 0; j < ND; ++j) s += acc[j];
                        const float mean = wave_sum_dpp(s) * (1.0f / (64.0f * ND));
                        float q = 0.0f;
#pragma unroll
                        for (int j = 0; j < ND; ++j) q = fmaf(acc[j] - mean, acc[j] - mean, q);
                        const float m2 = wave_sum_dpp(q);
                        if ((threadIdx.x & 63) == 0) stats[(size_t)(dq >> 6) * npad + n0 + tkn] = make_float2(mean, m2);
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// The same projection for WIDE inputs (C > 32, a multiple of 8: the peptide model's 96 latent channels), where the kernel above
// keeps 2 x 96 weights per thread in registers and walks 24 LDS reads per token: 65 us per evaluation for 55 MB of traffic.  Here
// a wave owns 32 tokens (their C inputs stay in registers as MFMA A fragments) and three 32-feature tiles; the product runs on
// v_mfma_f32_32x32x2_f32 (exact fp32 multiply-add), the accumulator leaves as 128-byte row segments.  Used for C > 32 only: the
// bits of the narrow-input models (MD17, pedestrian, NBA) do not change.
template <int MODE, int CK>  // CK = C / 8 float4 pieces per lane and operand
__global__ void __launch_bounds__(256) k_embed_mfma(float *out, const float *in, const float *W, const float *bias, const float *bias2,
                                                    const float *mask_emb, const int64_t *mask, const float *base, int N, int C, int D,
                                                    int tiles_per_wave) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hf = lane >> 5;
    const int nft = D / 32, ngrp = (nft + tiles_per_wave - 1) / tiles_per_wave;
    const long unit = (long)blockIdx.x * 4 + wave;  // (token tile, feature group)
    const long n_units = (long)((N + 31) / 32) * ngrp;
    if (unit >= n_units) return;
    const int n0 = (int)(unit / ngrp) * 32, ft0 = (int)(unit % ngrp) * tiles_per_wave;
    float4 a[CK];  // lane (r, hf): in[n0 + r][8 j + 4 hf .. + 3]
    {
        const float *ap = in + (size_t)min(n0 + r, N - 1) * C + 4 * hf;
#pragma unroll
        for (int j = 0; j < CK; ++j) a[j] = *reinterpret_cast<const float4 *>(ap + 8 * j);
    }
    for (int ft = ft0; ft < min(ft0 + tiles_per_wave, nft); ++ft) {
        const int d0 = ft * 32;
        float4 w[CK];
        const float *wp = W + (size_t)(d0 + r) * C + 4 * hf;
#pragma unroll
        for (int j = 0; j < CK; ++j) w[j] = *reinterpret_cast<const float4 *>(wp + 8 * j);
        // what is added to the product, requested before the MFMAs: lane = feature d0 + r, registers = 16 token rows
        float addv[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int n = min(n0 + acc_row(e, hf), N - 1);
            if (MODE == 0) addv[e] = mask_emb[(mask[n] != 0 ? D : 0) + d0 + r];
            else addv[e] = base[(size_t)n * D + d0 + r];
        }
        const float b0 = MODE == 0 ? bias[d0 + r] + bias2[d0 + r] : 0.0f;
        f32x16_d acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
        for (int j = 0; j < CK; ++j) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].x, w[j].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].y, w[j].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].z, w[j].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].w, w[j].w, acc, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int n = n0 + acc_row(e, hf);
            if (n < N) out[(size_t)n * D + d0 + r] = MODE == 0 ? acc[e] + (b0 + addv[e]) : acc[e] + addv[e];
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Row LayerNorm helpers: one wave per row of D floats, D = 64 * NE (NE <= 8).  A lane owns NE values:
// VEC = 2 (D multiple of 128): float2 pieces at d = 2*lane + 128*k; VEC = 1: d = lane + 64*k.
// Two-pass statistics in registers.
template <int NE, int VEC>
__device__ __forceinline__ int row_col(int lane, int e) { return VEC == 2 ? 2 * lane + 128 * (e >> 1) + (e & 1) : lane + 64 * e; }

template <int NE, int VEC>
__device__ __forceinline__ void row_load(const float *row, int lane, float (&v)[NE]) {
    if (VEC == 2) {
#pragma unroll
        for (int k = 0; k < NE / 2; ++k) {
            const float2 t = *reinterpret_cast<const float2 *>(row + 2 * lane + 128 * k);
            v[2 * k] = t.x;
            v[2 * k + 1] = t.y;
        }
    } else {
#pragma unroll
        for (int k = 0; k < NE; ++k) v[k] = row[lane + 64 * k];
    }
}
template <int NE, bool DPP = false>
__device__ __forceinline__ void row_stats(const float (&v)[NE], float eps, float &mean, float &rstd) {
    constexpr float invD = 1.0f / (float)(NE * 64);
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < NE; ++k) s += v[k];
    mean = (DPP ? wave_sum_dpp(s) : wave_sum(s)) * invD;
    float q = 0.0f;
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        const float a = v[k] - mean;
        q += a * a;
    }
    rstd = rsqrtf((DPP ? wave_sum_dpp(q) : wave_sum(q)) * invD + eps);
}

// h <- LayerNorm_{eps}(h) in place (normalize=True models, latent_si_v31.py:173-174, eps 1e-5).
template <int NE, int VEC>
__global__ void __launch_bounds__(256) k_ln_inplace(float *h, int N, float eps) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    constexpr int D = NE * 64;
    float *row = h + (size_t)n * D;
    float v[NE];
    row_load<NE, VEC>(row, lane, v);
    float mean, rstd;
    row_stats<NE>(v, eps, mean, rstd);
#pragma unroll
    for (int k = 0; k < NE; ++k) row[row_col<NE, VEC>(lane, k)] = (v[k] - mean) * rstd;
}

// a = bf16(LayerNorm_{1e-6}(h) * (1 + scale_b) + shift_b): the A operand of linear1
// (latent_si_v31.py:50,57 with mmdit.py:21-22).  shift/scale rows are indexed by trajectory.
template <int NE, int VEC>
__global__ void __launch_bounds__(256) k_ln_modulate(u16 *a, const float *h, const float *shift, const float *scale,
                                                     int mod_stride, int N, int tokens_per_traj) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    constexpr int D = NE * 64;
    float v[NE];
    row_load<NE, VEC>(h + (size_t)n * D, lane, v);
    float mean, rstd;
    row_stats<NE>(v, 1e-6f, mean, rstd);
    const size_t mo = (size_t)(n / tokens_per_traj) * mod_stride;
    u16 *arow = a + (size_t)n * D;
    if (VEC == 2) {
#pragma unroll
        for (int k = 0; k < NE / 2; ++k) {
            const int d = 2 * lane + 128 * k;
            const float2 sc = *reinterpret_cast<const float2 *>(scale + mo + d);
            const float2 sf = *reinterpret_cast<const float2 *>(shift + mo + d);
            const float y0 = (v[2 * k] - mean) * rstd * (1.0f + sc.x) + sf.x;
            const float y1 = (v[2 * k + 1] - mean) * rstd * (1.0f + sc.y) + sf.y;
            *reinterpret_cast<unsigned *>(arow + d) = pack2(y0, y1);
        }
    } else {
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const int d = lane + 64 * k;
            arow[d] = f2bf((v[k] - mean) * rstd * (1.0f + scale[mo + d]) + shift[mo + d]);
        }
    }
}

// The same operation for D a multiple of 256, persistent: a wave walks tokens gw, gw + (waves in the grid), ... with the next
// row already requested (16-byte loads, 1 KiB per wave instruction) while the current one is reduced and written (8-byte
// bf16x4 stores).  HBM-bound: 2 KB read + 1 KB written per token.
template <int NE>
__global__ void __launch_bounds__(256) k_ln_modulate_v4(u16 *a, const float *h, const float *shift, const float *scale, int mod_stride,
                                                        int N, int tokens_per_traj, int nt) {
    static_assert(NE % 4 == 0, "D must be a multiple of 256");
    constexpr int D = NE * 64, Q = NE / 4;
    constexpr float invD = 1.0f / (float)D;
    const int lane = threadIdx.x & 63;
    const int stride = gridDim.x * 4;
    int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float4 v[Q], vn[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) v[k] = *reinterpret_cast<const float4 *>(h + (size_t)n * D + 4 * lane + 256 * k);
    for (; n < N; n += stride) {
        const int nn = n + stride < N ? n + stride : n;
#pragma unroll
        for (int k = 0; k < Q; ++k) vn[k] = *reinterpret_cast<const float4 *>(h + (size_t)nn * D + 4 * lane + 256 * k);
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < Q; ++k) s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
        const float mean = wave_sum(s) * invD;
        float q = 0.0f;
#pragma unroll
        for (int k = 0; k < Q; ++k) {
            const float dx = v[k].x - mean, dy = v[k].y - mean, dz = v[k].z - mean, dw = v[k].w - mean;
            q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
        const float rstd = rsqrtf(wave_sum(q) * invD + 1e-6f);
        const size_t mo = (size_t)(n / tokens_per_traj) * mod_stride;
#pragma unroll
        for (int k = 0; k < Q; ++k) {
            const int d = 4 * lane + 256 * k;
            const float4 sc = *reinterpret_cast<const float4 *>(scale + mo + d);
            const float4 sf = *reinterpret_cast<const float4 *>(shift + mo + d);
            const u32x2 pk = {pack2((v[k].x - mean) * rstd * (1.0f + sc.x) + sf.x, (v[k].y - mean) * rstd * (1.0f + sc.y) + sf.y),
                              pack2((v[k].z - mean) * rstd * (1.0f + sc.z) + sf.z, (v[k].w - mean) * rstd * (1.0f + sc.w) + sf.w)};
            store8(a + (size_t)n * D + d, pk, nt);
        }
#pragma unroll
        for (int k = 0; k < Q; ++k) v[k] = vn[k];
    }
}

// Per-token LayerNorm statistics from the per-wave partials k_linear2_ws<LNS> leaves beside the residual stream (round 6): parts = D / gsz pairs
// (mean, sum of squared deviations) over gsz features each (32: k_linear2_ws; 64 ND: k_embed<STATS>), plane-major [part][npad]; combined in part order by the pairwise update of Chan et
// al. (a fixed order: a token's statistics do not depend on the launch) into (rstd, -mean rstd), eps 1e-6 like k_ln_modulate.
__global__ void __launch_bounds__(256) k_ln_finalize(float2 *tok, const float2 *parts_in, int parts, int npad, int N, float gsz = 32.0f) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float2 st[16];  // (parts <= 16: hidden <= 512; one pass over the partials, all requests in flight together)
#pragma unroll
    for (int p = 0; p < 16; ++p) st[p] = p < parts ? parts_in[(size_t)p * npad + n] : make_float2(0.0f, 0.0f);
    float msum = 0.0f, q = 0.0f;
#pragma unroll
    for (int p = 0; p < 16; ++p) msum += st[p].x;
    const float mean = msum / (float)parts;
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const float d = st[p].x - mean;
        if (p < parts) q += st[p].y + gsz * d * d;
    }
    const float rstd = rsqrtf(q / (gsz * (float)parts) + 1e-6f);
    tok[n] = make_float2(rstd, -mean * rstd);
}

// ---------------------------------------------------------------------------------------------------
// Runge-Kutta arithmetic of the adaptive ODE sampler (lam_slide_amd/transport.py: dopri5_solve; the reference reaches torchdiffeq's
// rk_common.py through integrators.py:67-78): stage states, the solution, the dense-output coefficients and the error ratio of a step, fp32.
// Every term is a rounded product added to a rounded running sum in the order of the list (no FMA contraction), so the result does not
// depend on the launch shape; the host restatement used for CPU tensors follows the same order.
struct RkTerms {
    const float *x[8];
    float c[8];
    int n;
};
// a rounded product (hipcc builds device code with -ffp-contract=fast: a * b + c becomes an FMA in the backend whatever the spelling or the
// pragma; the empty asm makes the product a value of its own)
__device__ __forceinline__ float rk_mul(float a, float b) {
    float p = a * b;
    asm volatile("" : "+v"(p));
    return p;
}
__device__ __forceinline__ float rk_sum(const RkTerms &t, unsigned long long i) {
    float acc = rk_mul(t.c[0], t.x[0][i]);
#pragma unroll
    for (int j = 1; j < 8; ++j)
        if (j < t.n) acc = acc + rk_mul(t.c[j], t.x[j][i]);
    return acc;
}
// out[i] = sum_j c[j] x[j][i]
__global__ void __launch_bounds__(256) k_rk_lincomb(float *out, RkTerms t, unsigned long long n) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) out[i] = rk_sum(t, i);
}
// out[i] = e + x (d + x (c + x (b + x a))): the quartic dense output of an accepted step (rk_common.py: _interp_evaluate)
__global__ void __launch_bounds__(256) k_rk_poly4(float *out, const float *a, const float *b, const float *c, const float *d, const float *e, float x,
                                                  unsigned long long n) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) {
        float v = b[i] + rk_mul(x, a[i]);
        v = c[i] + rk_mul(x, v);
        v = d[i] + rk_mul(x, v);
        out[i] = e[i] + rk_mul(x, v);
    }
}
// partial[b] = sum over block b's elements of (err[i] / (atol + rtol max(|y0[i]|, |y1[i]|)))^2, err = sum_j c[j] k[j]: fixed grid-stride
// assignment, per-thread sums in element order, fixed tree over the block (deterministic)
__global__ void __launch_bounds__(256) k_rk_error_partial(float *partial, const float *y0, const float *y1, RkTerms k, float atol, float rtol,
                                                          unsigned long long n) {
    __shared__ float red[256];
    float s = 0.0f;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) {
        const float q = rk_sum(k, i) / (atol + rtol * fmaxf(fabsf(y0[i]), fabsf(y1[i])));
        s += q * q;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
// *ratio = sqrt(sum_b partial[b] / n)
__global__ void __launch_bounds__(256) k_rk_error_final(float *ratio, const float *partial, int n_partial, unsigned long long n) {
    __shared__ double red[256];
    double s = 0.0;
    for (int b = threadIdx.x; b < n_partial; b += 256) s += (double)partial[b];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *ratio = (float)sqrt(red[0] / (double)n);
}

// ---------------------------------------------------------------------------------------------------
// Philox4x32-10 + Box-Muller: standard normal for (seed, step, element).  Documented stream:
// counter = (elem/4 lo, elem/4 hi, step, 0), key = (seed lo, seed hi); element e takes output e % 4
// after Box-Muller on the pairs (r0,r1) -> (z0,z1), (r2,r3) -> (z2,z3).
__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0];
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c[2];
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0;
        const unsigned n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1;
        const unsigned n3 = (unsigned)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ float philox_normal(unsigned long long seed, unsigned step, unsigned long long elem) {
    const unsigned long long blk = elem >> 2;
    unsigned c[4] = {(unsigned)blk, (unsigned)(blk >> 32), step, 0u};
    philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
    const int pair = (int)(elem & 2);
    const float u1 = ((float)c[pair] + 1.0f) * 2.3283064365386963e-10f;  // (0, 1]
    const float u2 = (float)c[pair + 1] * 2.3283064365386963e-10f;
    const float rad = sqrtf(-2.0f * __logf(u1));
    float sn, cs;
    __sincosf(6.283185307179586f * u2, &sn, &cs);
    return (elem & 1) ? rad * sn : rad * cs;
}

// Initial state of a sampling call, x_0 ~ N(0, 1) (lightning_base.py:231 torch.randn_like(x_cond)): the same documented stream as the
// per-step noise, under the reserved step index LSL_INIT_STEP, so that a sharded run (elem_offset = global index of the rank's first
// element) draws exactly the slice of the unsharded run.
constexpr unsigned LSL_INIT_STEP = 0xFFFFFFFFu;
__global__ void __launch_bounds__(256) k_randn(float *x, unsigned long long n, unsigned long long seed, unsigned step, unsigned long long elem_offset) {
    for (unsigned long long e = (unsigned long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (unsigned long long)gridDim.x * 256)
        x[e] = philox_normal(seed, step, elem_offset + e);
}

// A record of lsl_sample_ex without a network evaluation: x <- ax x + aw w + as saved (the noise injection of a Heun step), optionally
// copied to the saved-state buffer and to the trace.  Same noise addressing as the output-head kernels.
__global__ void __launch_bounds__(256) k_state_affine(float *x, unsigned long long n, float ax, float aw, float as, const float *noise,
                                                      unsigned long long seed, unsigned step, unsigned long long elem_offset, const float *saved,
                                                      float *save_out, float *trace) {
    for (unsigned long long e = (unsigned long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (unsigned long long)gridDim.x * 256) {
        float xn = ax * x[e];
        if (aw != 0.0f) xn += aw * (noise ? noise[e] : philox_normal(seed, step, elem_offset + e));
        if (saved) xn += as * saved[e];
        x[e] = xn;
        if (trace) trace[e] = xn;
        if (save_out) save_out[e] = xn;
    }
}

// ---------------------------------------------------------------------------------------------------
// Output head fused with the sampler update (latent_si_v31.py:185-187 + the affine step of lsl_step):
//   m = Linear_{D->C}(LayerNorm_{1e-6}(h) * (1 + scale) + shift)          (fp32 FMA chain, exact fp32 weights)
//   STEP: x <- ax * x + am * m + aw * w        else: out <- m
// Persistent workgroups walk 32-token tiles.  Phase 1: each wave normalises + modulates 8 token rows into LDS (fp32).
// Phase 2: thread (token, cg) accumulates 4 consecutive output channels over D, reading the token row and the
// transposed weight slab W_s[cg][d] (float4 = 4 channels) from LDS; slabs of 32 channels are cycled for C > 32.
// LDS rows are padded (+4 floats / +1 float4) so the 8 tokens / 8 channel groups of a wave hit distinct banks.
constexpr int HEAD_TOK = 32;
template <int NE>
constexpr size_t head_lds_bytes() { return (size_t)8 * (NE * 64 + 1) * 16 + (size_t)HEAD_TOK * (NE * 64 + 4) * 4; }

template <int NE, int VEC>
__global__ void __launch_bounds__(256) k_head_step(float *x, float *out, const float *h, const float *shift,
                                                   const float *scale, int mod_stride, const float *Wo, const float *bo,
                                                   int N, int C, int tokens_per_traj, int do_step, float ax, float am,
                                                   float aw, const float *noise, unsigned long long seed, unsigned step,
                                                   unsigned long long elem_offset, float *trace, float as, const float *saved,
                                                   float *save_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int D = NE * 64, AS = D + 4, WS = D + 1, RPW = HEAD_TOK / 4;  // rows per wave
    float4 *Ws = reinterpret_cast<float4 *>(smem);                    // [8][WS]
    float *As = reinterpret_cast<float *>(smem + (size_t)8 * WS * 16);  // [HEAD_TOK][AS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tk = tid >> 3, cg = tid & 7;
    const int n_tiles = (N + HEAD_TOK - 1) / HEAD_TOK;
    const bool w_resident = C <= 32;  // one 32-channel slab: fill it once per workgroup, then walk token tiles

    auto fill_w = [&](int c0) {
#pragma unroll 8
        for (int i = tid; i < 32 * D; i += 256) {
            const int cl = i / D, d = i - cl * D, c = c0 + cl;
            reinterpret_cast<float *>(&Ws[(cl >> 2) * WS + d])[cl & 3] = c < C ? Wo[(size_t)c * D + d] : 0.0f;
        }
    };
    if (w_resident) fill_w(0);

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int n0 = tile * HEAD_TOK;
        // phase 1: all of this wave's rows are requested before any reduction, so their latencies overlap
        float v[RPW][NE];
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int n = min(n0 + wave + 4 * i, N - 1);
            row_load<NE, VEC>(h + (size_t)n * D, lane, v[i]);
        }
        __syncthreads();  // previous tile's phase 2 has finished reading As (and Ws when not resident)
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int n = min(n0 + wave + 4 * i, N - 1);
            float mean, rstd;
            row_stats<NE>(v[i], 1e-6f, mean, rstd);
            const size_t mo = (size_t)(n / tokens_per_traj) * mod_stride;
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                const int d = row_col<NE, VEC>(lane, k);
                As[(wave + 4 * i) * AS + d] = (v[i][k] - mean) * rstd * (1.0f + scale[mo + d]) + shift[mo + d];
            }
        }
        const int n = n0 + tk;
        for (int c0 = 0; c0 < C; c0 += 32) {
            if (!w_resident) {
                if (c0) __syncthreads();  // previous slab fully consumed
                fill_w(c0);
            }
            __syncthreads();  // As (and Ws) visible
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
            const float *ar = As + tk * AS;
            const float4 *wr = Ws + cg * WS;
#pragma unroll 2
            for (int d = 0; d < D; d += 4) {
                const float4 av = *reinterpret_cast<const float4 *>(ar + d);
                const float4 w0 = wr[d], w1 = wr[d + 1], w2 = wr[d + 2], w3 = wr[d + 3];
                a0 = fmaf(av.x, w0.x, a0); a1 = fmaf(av.x, w0.y, a1); a2 = fmaf(av.x, w0.z, a2); a3 = fmaf(av.x, w0.w, a3);
                a0 = fmaf(av.y, w1.x, a0); a1 = fmaf(av.y, w1.y, a1); a2 = fmaf(av.y, w1.z, a2); a3 = fmaf(av.y, w1.w, a3);
                a0 = fmaf(av.z, w2.x, a0); a1 = fmaf(av.z, w2.y, a1); a2 = fmaf(av.z, w2.z, a2); a3 = fmaf(av.z, w2.w, a3);
                a0 = fmaf(av.w, w3.x, a0); a1 = fmaf(av.w, w3.y, a1); a2 = fmaf(av.w, w3.z, a2); a3 = fmaf(av.w, w3.w, a3);
            }
            if (n < N) {
                const float acc[4] = {a0, a1, a2, a3};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = c0 + 4 * cg + j;
                    if (c < C) {
                        const float m = acc[j] + bo[c];
                        const size_t e = (size_t)n * C + c;
                        if (do_step) {
                            float xn = ax * x[e] + am * m;
                            if (aw != 0.0f) xn += aw * (noise ? noise[e] : philox_normal(seed, step, elem_offset + e));
                            if (saved) xn += as * saved[e];  // (lsl_step_ex: a state kept by an earlier record, e.g. Heun's x_hat)
                            x[e] = xn;
                            if (trace) trace[e] = xn;
                            if (save_out) save_out[e] = xn;
                        } else {
                            out[e] = m;
                        }
                    }
                }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// Output head + state update, fp32 MFMA form.  Same contract as k_head_step; the [32 tokens x D] x [D x 32 channels] product runs on
// v_mfma_f32_32x32x2_f32 (exact fp32 multiply-add, 1/16 of the bf16 rate - ample here) instead of LDS-fed scalar FMAs, which were
// LDS-bandwidth bound (5 ds_read_b128 per 16 FMAs, 82 KB of LDS reads per token).  The four waves split k; W^T fragments stay in
// registers across token tiles (C <= 32); the next tile's h rows are requested before the current tile's product, so the single
// workgroup a CU holds never waits on HBM with nothing else to do.
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
template <int NE>
constexpr size_t head_mfma_lds_bytes(bool one_slab) {  // one_slab (C <= 32): the partial products overlay the activation rows, dead by then
    constexpr size_t rows = (size_t)HEAD_TOK * (NE * 64 + 4) * 4, parts = (size_t)4 * 32 * 33 * 4;
    return one_slab ? (rows > parts ? rows : parts) : rows + parts;
}

template <int NE, int VEC>
__global__ void __launch_bounds__(256) k_head_step_mfma(float *x, float *out, const float *h, const float *shift, const float *scale,
                                                        int mod_stride, const float *Wo, const float *bo, int N, int C, int tokens_per_traj,
                                                        int do_step, float ax, float am, float aw, const float *noise, unsigned long long seed,
                                                        unsigned step, unsigned long long elem_offset, float *trace, float as,
                                                        const float *saved, float *save_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int D = NE * 64, AS = D + 4, RPW = HEAD_TOK / 4, KW = D / 4, JS = KW / 8;  // k range per wave, 8-deep steps per wave
    float *As = reinterpret_cast<float *>(smem);  // [HEAD_TOK][AS]  LayerNorm'ed + modulated rows
    const bool w_resident = C <= 32;
    float *Ps = w_resident ? As : As + HEAD_TOK * AS;  // [4][32][33] per-wave partial products (token, channel); one slab: over the dead rows
                                                        // (66 instead of 83 KB at hidden 512: two workgroups per CU)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hf = lane >> 5;
    const int tk = tid >> 3, cg = tid & 7;
    const int n_tiles = (N + HEAD_TOK - 1) / HEAD_TOK;

    float4 wf[JS];  // B operand: W^T[k][channel r] for this wave's k range, lane (r, hf) holds k = 8 j + 4 hf .. + 3
    auto load_w = [&](int c0) {
        const int c = c0 + r;
#pragma unroll
        for (int j = 0; j < JS; ++j)
            wf[j] = c < C ? *reinterpret_cast<const float4 *>(Wo + (size_t)c * D + wave * KW + 8 * j + 4 * hf) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    if (w_resident) load_w(0);

    float v[RPW][NE];
    auto load_rows = [&](int tile) {
#pragma unroll
        for (int i = 0; i < RPW; ++i) row_load<NE, VEC>(h + (size_t)min(tile * HEAD_TOK + wave + 4 * i, N - 1) * D, lane, v[i]);
    };
    if ((int)blockIdx.x < n_tiles) load_rows(blockIdx.x);

    // a tile of 32 tokens nearly always lies inside ONE trajectory (and with a shared modulation row every tile does): the lane's shift /
    // scale values then stay in registers across the tile's rows and across tiles, instead of 2 NE loads in front of every row's
    // stores (128 dependent load instructions per wave and tile: the LayerNorm phase waited on them, profiles/r04_experiments.txt)
    float sc1[NE], sh[NE];
    int cur_traj = -1;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int n0 = tile * HEAD_TOK;
        const int t_first = mod_stride ? n0 / tokens_per_traj : 0, t_last = mod_stride ? min(n0 + HEAD_TOK - 1, N - 1) / tokens_per_traj : 0;
        const bool one_traj = t_first == t_last;  // (uniform)
        if (one_traj && t_first != cur_traj) {
            const size_t mo = (size_t)t_first * mod_stride;
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                const int d = row_col<NE, VEC>(lane, k);
                sc1[k] = 1.0f + scale[mo + d];
                sh[k] = shift[mo + d];
            }
            cur_traj = t_first;
        }
        __syncthreads();  // the previous tile has finished with As and Ps
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int n = min(n0 + wave + 4 * i, N - 1);
            float mean, rstd;
            row_stats<NE, true>(v[i], 1e-6f, mean, rstd);  // (two waves per SIMD: the DPP form, the LDS-queue round trips of wave_sum were exposed)
            if (one_traj) {
#pragma unroll
                for (int k = 0; k < NE; ++k) As[(wave + 4 * i) * AS + row_col<NE, VEC>(lane, k)] = (v[i][k] - mean) * rstd * sc1[k] + sh[k];
            } else {
                const size_t mo = (size_t)(n / tokens_per_traj) * mod_stride;
#pragma unroll
                for (int k = 0; k < NE; ++k) {
                    const int d = row_col<NE, VEC>(lane, k);
                    As[(wave + 4 * i) * AS + d] = (v[i][k] - mean) * rstd * (1.0f + scale[mo + d]) + shift[mo + d];
                }
            }
        }
        if (tile + (int)gridDim.x < n_tiles) load_rows(tile + gridDim.x);  // in flight during the product below
        __syncthreads();
        const int n = n0 + tk;
        // (one slab) the state values this thread will update are requested now, in front of the product, instead of behind it
        float xs[4] = {0.f, 0.f, 0.f, 0.f};
        if (w_resident && do_step && n < N) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (4 * cg + j < C) xs[j] = x[(size_t)n * C + 4 * cg + j];
        }
        for (int c0 = 0; c0 < C; c0 += 32) {
            if (!w_resident) load_w(c0);
            f32x16_t acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
            const float *ar = As + r * AS + wave * KW + 4 * hf;
#pragma unroll
            for (int j = 0; j < JS; ++j) {
                const float4 av = *reinterpret_cast<const float4 *>(ar + 8 * j);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, wf[j].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, wf[j].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, wf[j].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, wf[j].w, acc, 0, 0, 0);
            }
            if (c0 || w_resident) __syncthreads();  // the previous slab's partials have been consumed / every wave has read its activation rows
#pragma unroll
            for (int e = 0; e < 16; ++e) Ps[(wave * 32 + acc_row(e, hf)) * 33 + r] = acc[e];
            __syncthreads();
            if (n < N) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int cl = 4 * cg + j, c = c0 + cl;
                    if (c < C) {
                        const float m = ((Ps[(0 * 32 + tk) * 33 + cl] + Ps[(1 * 32 + tk) * 33 + cl]) + Ps[(2 * 32 + tk) * 33 + cl]) +
                                        Ps[(3 * 32 + tk) * 33 + cl] + bo[c];
                        const size_t e = (size_t)n * C + c;
                        if (do_step) {
                            float xn = ax * (w_resident ? xs[j] : x[e]) + am * m;
                            if (aw != 0.0f) xn += aw * (noise ? noise[e] : philox_normal(seed, step, elem_offset + e));
                            if (saved) xn += as * saved[e];  // (lsl_step_ex: a state kept by an earlier record, e.g. Heun's x_hat)
                            x[e] = xn;
                            if (trace) trace[e] = xn;
                            if (save_out) save_out[e] = xn;
                        } else {
                            out[e] = m;
                        }
                    }
                }
            }
        }
    }
}
