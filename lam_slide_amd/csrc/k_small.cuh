// fp32 kernels around the MFMA blocks: RoPE table, conditioning vector and modulation tables, input
// embedding, LayerNorm+modulate, output projection fused with the sampler's state update, device noise.
// These carry < 1 % of the FLOPs and stay in fp32 on purpose ("Accuracy target").
#pragma once
#include "common.cuh"

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

// ---------------------------------------------------------------------------------------------------
// out[b][o] = post(sum_i W[o][i] * pre(in[b][i]) + bias[o] + add[b][o]); one wave per output feature,
// the weight row stays in registers while the wave walks the batch.  I <= 512.
template <bool PRE_SILU, bool POST_SILU>
__global__ void __launch_bounds__(256) k_dense_rows(float *out, const float *in, const float *W, const float *bias,
                                                    const float *add, int rows, int I, int O, int add_stride) {
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
    for (int b = 0; b < rows; ++b) {
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = lane + 64 * k;
            if (i < I) {
                float v = in[(size_t)b * I + i];
                if (PRE_SILU) v = silu(v);
                s = fmaf(w[k], v, s);
            }
        }
        s = wave_sum(s);
        if (lane == 0) {
            s += bo;
            if (add) s += add[(size_t)b * add_stride + o];
            if (POST_SILU) s = silu(s);
            out[(size_t)b * O + o] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Small-K projection C -> D (x_in / cond_to_emb, latent_si_v31.py:172).
//   MODE 0 (once per sample): out = in @ W^T + bias + bias2 + mask_emb[mask]      (cond_to_emb part)
//   MODE 1 (every evaluation): out = in @ W^T + base                              (x_in part + cached)
template <int CMAX, int MODE>
__global__ void __launch_bounds__(256) k_embed(float *out, const float *in, const float *W, const float *bias,
                                               const float *bias2, const float *mask_emb, const int64_t *mask,
                                               const float *base, int N, int C, int D) {
    constexpr int TOK = 32;
    __shared__ float xs[TOK][CMAX + 1];
    const int n0 = blockIdx.x * TOK;
    for (int i = threadIdx.x; i < TOK * C; i += blockDim.x) {
        const int tkn = i / C, c = i % C;
        xs[tkn][c] = (n0 + tkn < N) ? in[(size_t)(n0 + tkn) * C + c] : 0.0f;
    }
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float w[CMAX];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) w[c] = c < C ? W[(size_t)d * C + c] : 0.0f;
        float b0 = 0.0f;
        if (MODE == 0) b0 = bias[d] + bias2[d];
        for (int tkn = 0; tkn < TOK; ++tkn) {
            const int n = n0 + tkn;
            if (n >= N) break;
            float s = 0.0f;
#pragma unroll
            for (int c = 0; c < CMAX; ++c) s = fmaf(xs[tkn][c], w[c], s);
            if (MODE == 0) s += b0 + mask_emb[(mask[n] != 0 ? D : 0) + d];
            else s += base[(size_t)n * D + d];
            out[(size_t)n * D + d] = s;
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
template <int NE>
__device__ __forceinline__ void row_stats(const float (&v)[NE], float eps, float &mean, float &rstd) {
    constexpr float invD = 1.0f / (float)(NE * 64);
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < NE; ++k) s += v[k];
    mean = wave_sum(s) * invD;
    float q = 0.0f;
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        const float a = v[k] - mean;
        q += a * a;
    }
    rstd = rsqrtf(wave_sum(q) * invD + eps);
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

// ---------------------------------------------------------------------------------------------------
// Output head fused with the sampler update (latent_si_v31.py:185-187 + the affine step of lsl_step):
//   m = Linear_{D->C}(LayerNorm_{1e-6}(h) * (1 + scale) + shift)
//   STEP: x <- ax * x + am * m + aw * w        else: out <- m
// One wave per token; each lane forms partial dot products over its strided slice of the row for
// CT channels at a time, then a butterfly reduce-scatter leaves channel (lane >> 1) summed in lane pairs.
template <int NE, int VEC>
__global__ void __launch_bounds__(256) k_head_step(float *x, float *out, const float *h, const float *shift,
                                                   const float *scale, int mod_stride, const float *Wo, const float *bo,
                                                   int N, int C, int tokens_per_traj, int do_step, float ax, float am,
                                                   float aw, const float *noise, unsigned long long seed, unsigned step,
                                                   unsigned long long elem_offset, float *trace) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    constexpr int D = NE * 64;
    float v[NE];
    row_load<NE, VEC>(h + (size_t)n * D, lane, v);
    float mean, rstd;
    row_stats<NE>(v, 1e-6f, mean, rstd);
    const size_t mo = (size_t)(n / tokens_per_traj) * mod_stride;
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        const int d = row_col<NE, VEC>(lane, k);
        v[k] = (v[k] - mean) * rstd * (1.0f + scale[mo + d]) + shift[mo + d];
    }
    for (int c0 = 0; c0 < C; c0 += 32) {
        float part[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int c = c0 + j;
            float s = 0.0f;
            if (c < C) {
                const float *wr = Wo + (size_t)c * D;
#pragma unroll
                for (int k = 0; k < NE; ++k) s = fmaf(v[k], wr[row_col<NE, VEC>(lane, k)], s);
            }
            part[j] = s;
        }
        // reduce-scatter over the wave: after the 5 halving rounds lane l holds channel c0 + (l >> 1)
#pragma unroll
        for (int width = 32, cnt = 16; width >= 2; width >>= 1, cnt >>= 1) {
            const bool upper = (lane & width) != 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (j < cnt) {
                    const float keep = upper ? part[j + cnt] : part[j];
                    const float send = upper ? part[j] : part[j + cnt];
                    part[j] = keep + __shfl_xor(send, width, 64);
                }
            }
        }
        float m = part[0] + __shfl_xor(part[0], 1, 64);
        const int c = c0 + (lane >> 1);
        if ((lane & 1) == 0 && c < C) {
            m += bo[c];
            const size_t e = (size_t)n * C + c;
            if (do_step) {
                float xn = ax * x[e] + am * m;
                if (aw != 0.0f) xn += aw * (noise ? noise[e] : philox_normal(seed, step, elem_offset + e));
                x[e] = xn;
                if (trace) trace[e] = xn;
            } else {
                out[e] = m;
            }
        }
    }
}
