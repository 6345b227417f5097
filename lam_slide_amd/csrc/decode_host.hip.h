// Host side of lsl_decode / lsl_encode: the launch sequences of the frozen stage-1 decode and encode (kernels in k_decode.hip.h).
#pragma once
#include "k_decode.hip.h"

struct lsl_encoder {
    lsl_encoder_desc d;
    lsl_encoder_weights w;
    std::vector<lsl_dec_block> cross_blocks, self_blocks;
};

struct lsl_decoder {
    lsl_decoder_desc d;
    lsl_decoder_weights w;
    std::vector<lsl_dec_block> self_blocks, cross_blocks;
};

namespace {

struct DecWs {
    float *lat, *q, *xn, *cn, *qb, *kvb, *att, *hid, *ext;
};

inline size_t dec_align(size_t n) { return (n + 63) & ~(size_t)63; }

// every buffer in floats; returns the total in bytes
size_t dec_carve(const lsl_decoder_desc &d, int frames, int L, int A, char *base, DecWs *ws) {
    const int split = d.num_split > 1 ? d.num_split : 1;
    const size_t nl = (size_t)frames * L, na = (size_t)frames * A, nmax = std::max(nl * split, na);
    const int inner_l = d.heads_latent * d.dim_head_latent, inner_c = d.heads_cross * d.dim_head_cross;
    const int dmax = std::max(std::max(d.dim_latent, d.dim_query), std::max(d.in_dim, d.dim_emb));
    const int imax = std::max(3 * inner_l, 2 * inner_c);
    size_t off = 0;
    auto take = [&](size_t floats) {
        float *p = base ? reinterpret_cast<float *>(base + off) : nullptr;
        off += dec_align(floats * sizeof(float));
        return p;
    };
    DecWs w;
    w.lat = take(nl * d.dim_latent);
    w.q = take(na * d.dim_query);
    w.xn = take(nmax * dmax);
    w.cn = take(nmax * dmax);
    w.qb = take(nmax * imax);
    w.kvb = take(nmax * imax);
    w.att = take(nmax * std::max(inner_l, inner_c));
    w.hid = take(nmax * dmax);
    w.ext = split > 1 ? take(nl * split * d.dim_latent) : nullptr;
    if (ws) *ws = w;
    return off;
}

void dec_ln(float *out, const float *in, const float *w, const float *b, int rows, int D, hipStream_t st) {
    hipLaunchKernelGGL(k_dec_ln, dim3((rows + 3) / 4), dim3(256), 0, st, out, in, w, b, rows, D, 1e-5f);
}

void dec_dense(int act, float *out, const float *in, const float *W, const float *bias, const float *res, int rows, int I, int O,
               hipStream_t st) {
    const dim3 grid((O + 63) / 64, (rows + 63) / 64);
    if (act == 1) hipLaunchKernelGGL((k_dec_dense<1>), grid, dim3(256), 0, st, out, in, W, bias, res, rows, I, O);
    else if (act == 2) hipLaunchKernelGGL((k_dec_dense<2>), grid, dim3(256), 0, st, out, in, W, bias, res, rows, I, O);
    else hipLaunchKernelGGL((k_dec_dense<0>), grid, dim3(256), 0, st, out, in, W, bias, res, rows, I, O);
}

int dec_attn(const DecAttnArgs &a, int frames, hipStream_t st) {
    const size_t lds = ((size_t)2 * a.Sk * (a.dh <= 16 ? 16 : a.dh <= 32 ? 32 : 64) + a.Sk) * sizeof(float);
    if (a.dh > 64 || lds > 64 * 1024) return fail(-3, "decode attention: dim_head %d / %d keys exceed the LDS tile", a.dh, a.Sk);
    const dim3 grid(frames * a.H);
    if (a.dh <= 16) hipLaunchKernelGGL((k_dec_attn<16>), grid, dim3(256), lds, st, a);
    else if (a.dh <= 32) hipLaunchKernelGGL((k_dec_attn<32>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((k_dec_attn<64>), grid, dim3(256), lds, st, a);
    return 0;
}

// x <- x + to_out(attention(LN(x), LN_c(ctx)));  x <- x + FF(LN(x))     (torch_modules.py:221-264)
// x: [frames * Sx, dim]; ctx: [frames * Sc, cdim] or nullptr (self-attention)
int dec_block(const lsl_dec_block &b, float *x, int Sx, int dim, const float *ctx, int Sc, int cdim, int H, int dh, int act, int frames,
              const DecWs &ws, hipStream_t st, const unsigned char *key_mask = nullptr) {
    const int inner = H * dh, nx = frames * Sx;
    dec_ln(ws.xn, x, b.ln_w, b.ln_b, nx, dim, st);
    DecAttnArgs a{};
    a.q_scale = b.q_scale;
    a.k_scale = b.k_scale;
    a.key_mask = key_mask;
    a.dh = dh;
    a.H = H;
    a.Sq = Sx;
    a.out = ws.att;
    a.ldo = inner;
    if (!ctx) {
        dec_dense(0, ws.qb, ws.xn, b.w_q, nullptr, nullptr, nx, dim, 3 * inner, st);  // to_qkv, chunk(3) = column thirds
        a.q = ws.qb;
        a.k = ws.qb + inner;
        a.v = ws.qb + 2 * inner;
        a.ldq = a.ldk = a.ldv = 3 * inner;
        a.Sk = Sx;
    } else {
        const int nc = frames * Sc;
        dec_ln(ws.cn, ctx, b.lnc_w, b.lnc_b, nc, cdim, st);
        dec_dense(0, ws.qb, ws.xn, b.w_q, nullptr, nullptr, nx, dim, inner, st);
        dec_dense(0, ws.kvb, ws.cn, b.w_kv, nullptr, nullptr, nc, cdim, 2 * inner, st);  // to_kv, chunk(2)
        a.q = ws.qb;
        a.ldq = inner;
        a.k = ws.kvb;
        a.v = ws.kvb + inner;
        a.ldk = a.ldv = 2 * inner;
        a.Sk = Sc;
    }
    if (int rc = dec_attn(a, frames, st)) return rc;
    dec_dense(0, x, ws.att, b.w_out, b.b_out, x, nx, inner, dim, st);
    dec_ln(ws.xn, x, b.ff_ln_w, b.ff_ln_b, nx, dim, st);
    dec_dense(act, ws.hid, ws.xn, b.ff_w1, b.ff_b1, nullptr, nx, dim, dim, st);
    dec_dense(0, x, ws.hid, b.ff_w2, b.ff_b2, x, nx, dim, dim, st);
    return 0;
}

// encoder scratch: the same buffers as the decoder's, sized for latents [frames*N, dim_latent] and context [frames*A, dim_ctx]
size_t enc_carve(const lsl_encoder_desc &d, int frames, int A, char *base, DecWs *ws, float **ctx) {
    const size_t nl = (size_t)frames * d.num_latents, na = (size_t)frames * A, nmax = std::max(nl, na);
    const int dim_ctx = d.dim_input + d.dim_emb;
    const int inner_l = d.heads_latent * d.dim_head_latent, inner_c = d.heads_cross * d.dim_head_cross;
    const int dmax = std::max(dim_ctx, d.dim_latent), imax = std::max(3 * inner_l, 2 * inner_c);
    size_t off = 0;
    auto take = [&](size_t floats) {
        float *p = base ? reinterpret_cast<float *>(base + off) : nullptr;
        off += dec_align(floats * sizeof(float));
        return p;
    };
    DecWs w;
    w.lat = take(nl * d.dim_latent);
    w.q = nullptr;
    float *c = take(na * dim_ctx);
    w.xn = take(nmax * dmax);
    w.cn = take(nmax * dmax);
    w.qb = take(nmax * imax);
    w.kvb = take(nmax * imax);
    w.att = take(nmax * std::max(inner_l, inner_c));
    w.hid = take(nmax * dmax);
    w.ext = nullptr;
    if (ws) *ws = w;
    if (ctx) *ctx = c;
    return off;
}

}  // namespace
