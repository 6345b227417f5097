// fp32 kernels of the frozen stage-1 decode that follows the sampler and of the encode that precedes it (8f.3:
// models/components/encoder.py:34-41,96-103 + quant, lightning_base.py:22-25,37-40): post_quant -> Decoder
// (models/composites/lightning_base.py:28-31,42-44, models/components/decoder.py:82-102, modules/torch_modules.py:104-264).
// The decode is < 0.1 % of the path's FLOPs (a few hundred MFLOP per frame against 13 TFLOP per trajectory), so these are
// plain fp32 kernels: no bf16 rounding enters the decoded coordinates, the quantity the parity metric is stated on.
#pragma once
#include "common.hip.h"

// LayerNorm over the last dimension, one wave per row (nn.LayerNorm: biased variance, eps inside the sqrt).  w == nullptr: no affine.
__global__ void __launch_bounds__(256) k_dec_ln(float *out, const float *in, const float *w, const float *b, int rows, int D, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float *x = in + (size_t)row * D;
    float s = 0.0f;
    for (int i = lane; i < D; i += 64) s += x[i];
    const float mean = wave_sum(s) / D;
    float v = 0.0f;
    for (int i = lane; i < D; i += 64) {
        const float d = x[i] - mean;
        v = fmaf(d, d, v);
    }
    const float rstd = rsqrtf(wave_sum(v) / D + eps);
    for (int i = lane; i < D; i += 64) {
        float y = (x[i] - mean) * rstd;
        if (w) y = fmaf(y, w[i], b[i]);
        out[(size_t)row * D + i] = y;
    }
}

// rows of an embedding table picked by index (entity_embeddings.py:25-33; the max_norm clipping of the looked-up rows is
// applied to the whole table once, when the weights are packed)
__global__ void __launch_bounds__(256) k_dec_gather(float *out, const float *table, const int64_t *idx, int rows, int E, int n_entities) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    long e = idx[row];
    e = e < 0 ? 0 : (e >= n_entities ? n_entities - 1 : e);
    for (int i = lane; i < E; i += 64) out[(size_t)row * E + i] = table[(size_t)e * E + i];
}

// encoder context rows: out[r] = [x[r] | table[entities[r]]]   (encoder.py:34-37)
__global__ void __launch_bounds__(256) k_enc_context(float *out, const float *x, const float *table, const int64_t *idx, int rows, int DX,
                                                     int E, int n_entities) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    long e = idx[row];
    e = e < 0 ? 0 : (e >= n_entities ? n_entities - 1 : e);
    float *o = out + (size_t)row * (DX + E);
    for (int i = lane; i < DX; i += 64) o[i] = x[(size_t)row * DX + i];
    for (int i = lane; i < E; i += 64) o[DX + i] = table[(size_t)e * E + i];
}

// the learned latent array repeated for every frame (encoder.py:39: repeat "N D -> B N D")
__global__ void __launch_bounds__(256) k_enc_broadcast(float *out, const float *latents, long total, int per_frame) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < total) out[i] = latents[i % per_frame];
}

template <int ACT>
__device__ __forceinline__ float dec_act(float x) {
    if (ACT == 1) return gelu_erf(x);  // src.modules.torch_modules.GELU (exact erf, torch_modules.py:20-33)
    if (ACT == 2) return 0.5f * x * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));  // nn.GELU(approximate="tanh")
    return x;
}

// out[r][o] = act(in[r] . W[o] + bias[o]) + res[r][o]     (nn.Linear; bias / res optional; res may alias out)
// 64 x 64 output tile per workgroup, 4 x 4 per thread, k in steps of 16 through LDS; I % 4 == 0.
template <int ACT>
__global__ void __launch_bounds__(256) k_dec_dense(float *out, const float *in, const float *W, const float *bias, const float *res,
                                                   int rows, int I, int O) {
    __shared__ __attribute__((aligned(16))) float As[16][64 + 4];
    __shared__ __attribute__((aligned(16))) float Ws[16][64 + 4];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int r0 = blockIdx.y * 64, o0 = blockIdx.x * 64;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0f;
    const int lr = tid >> 2, lk = (tid & 3) * 4;
    for (int k0 = 0; k0 < I; k0 += 16) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), w = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r0 + lr < rows && k0 + lk < I) a = *reinterpret_cast<const float4 *>(in + (size_t)(r0 + lr) * I + k0 + lk);
        if (o0 + lr < O && k0 + lk < I) w = *reinterpret_cast<const float4 *>(W + (size_t)(o0 + lr) * I + k0 + lk);
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
        const int r = r0 + 4 * ty + i;
        if (r >= rows) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int o = o0 + 4 * tx + j;
            if (o >= O) continue;
            float v = acc[i][j] + (bias ? bias[o] : 0.0f);
            v = dec_act<ACT>(v);
            if (res) v += res[(size_t)r * O + o];
            out[(size_t)r * O + o] = v;
        }
    }
}

// softmax(q k^T / sqrt(dh)) v for one (frame, head) per workgroup, fp32 (torch_modules.py:150-218: optional per-head RMS
// norm of q and k with a learned scale, F.scaled_dot_product_attention).  K and V of the head sit in LDS; a thread owns
// one query row and makes two passes over the keys (max, then exp / sum / weighted V).  DH <= 64.
struct DecAttnArgs {
    const float *q, *k, *v;  // row strides ldq, ldk, ldv; head h at column offset h * dh
    float *out;              // row stride ldo
    const float *q_scale, *k_scale;  // [dh] or nullptr (no QK norm)
    const unsigned char *key_mask;   // [frames, Sk], non-zero = attend (attn_mask of F.scaled_dot_product_attention), or nullptr
    int ldq, ldk, ldv, ldo;
    int Sq, Sk, dh, H;
};

template <int DH>
__global__ void __launch_bounds__(256) k_dec_attn(DecAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float kv[];  // K [Sk][DH], V [Sk][DH], additive key mask [Sk] (0 or -inf)
    const int f = blockIdx.x / a.H, h = blockIdx.x % a.H;
    float *Ks = kv, *Vs = kv + (size_t)a.Sk * DH, *Ms = kv + (size_t)2 * a.Sk * DH;
    const int dh = a.dh;
    for (int s = threadIdx.x; s < a.Sk; s += blockDim.x) {
        const float *kr = a.k + (size_t)(f * a.Sk + s) * a.ldk + h * dh;
        const float *vr = a.v + (size_t)(f * a.Sk + s) * a.ldv + h * dh;
        float ss = 0.0f;
        for (int d = 0; d < dh; ++d) ss = fmaf(kr[d], kr[d], ss);
        const float rr = a.k_scale ? rsqrtf(ss / dh + 1e-6f) : 1.0f;
        for (int d = 0; d < dh; ++d) {
            Ks[s * DH + d] = a.k_scale ? kr[d] * rr * a.k_scale[d] : kr[d];
            Vs[s * DH + d] = vr[d];
        }
        Ms[s] = (a.key_mask && !a.key_mask[(size_t)f * a.Sk + s]) ? -INFINITY : 0.0f;
    }
    __syncthreads();
    const float scale = rsqrtf((float)dh);
    for (int qi = threadIdx.x; qi < a.Sq; qi += blockDim.x) {
        const float *qr = a.q + (size_t)(f * a.Sq + qi) * a.ldq + h * dh;
        float q[DH];
        float ss = 0.0f;
#pragma unroll
        for (int d = 0; d < DH; ++d) {
            q[d] = d < dh ? qr[d] : 0.0f;
            ss = fmaf(q[d], q[d], ss);
        }
        const float rr = a.q_scale ? rsqrtf(ss / dh + 1e-6f) : 1.0f;
#pragma unroll
        for (int d = 0; d < DH; ++d)
            if (d < dh) q[d] = (a.q_scale ? q[d] * rr * a.q_scale[d] : q[d]) * scale;
        float m = -INFINITY;
        for (int s = 0; s < a.Sk; ++s) {
            float sc = 0.0f;
#pragma unroll
            for (int d = 0; d < DH; ++d)
                if (d < dh) sc = fmaf(q[d], Ks[s * DH + d], sc);
            m = fmaxf(m, sc + Ms[s]);
        }
        float o[DH], l = 0.0f;
#pragma unroll
        for (int d = 0; d < DH; ++d) o[d] = 0.0f;
        for (int s = 0; s < a.Sk; ++s) {
            float sc = 0.0f;
#pragma unroll
            for (int d = 0; d < DH; ++d)
                if (d < dh) sc = fmaf(q[d], Ks[s * DH + d], sc);
            const float p = __expf(sc + Ms[s] - m);
            l += p;
#pragma unroll
            for (int d = 0; d < DH; ++d)
                if (d < dh) o[d] = fmaf(p, Vs[s * DH + d], o[d]);
        }
        const float inv = 1.0f / l;
        float *orow = a.out + (size_t)(f * a.Sq + qi) * a.ldo + h * dh;
#pragma unroll
        for (int d = 0; d < DH; ++d)
            if (d < dh) orow[d] = o[d] * inv;
    }
}
