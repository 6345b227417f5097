// Host side, part 4 of 4: C entry points of the frozen stage-1 decoder / encoder (decode_host.hip.h holds their launch helpers).  Inside
// the extern "C" block of lsl_api.hip.
#pragma once

int lsl_decoder_create(const lsl_decoder_desc *desc, const lsl_decoder_weights *w, lsl_decoder **out) try {
    if (!desc || !w || !out) return fail(-1, "null decoder argument");
    const lsl_decoder_desc &d = *desc;
    if (d.in_dim % 4 || d.dim_latent % 4 || d.dim_query % 4 || d.dim_emb % 4 || (d.heads_latent * d.dim_head_latent) % 4 ||
        (d.heads_cross * d.dim_head_cross) % 4)
        return fail(-3, "decoder widths must be multiples of 4");
    if (d.dim_head_latent > 64 || d.dim_head_cross > 64 || d.dim_head_latent < 1 || d.dim_head_cross < 1) return fail(-3, "decoder dim_head must be 1..64");
    if (d.act != 1 && d.act != 2) return fail(-3, "decoder activation must be 1 (erf GELU) or 2 (tanh GELU)");
    if (d.num_block_attn < 0 || d.num_block_cross < 0 || d.out_dim < 1 || d.n_entities < 1 || d.num_split < 0) return fail(-3, "bad decoder description");
    if (d.num_split > 1 && (!w->ext_w || !w->ext_b)) return fail(-2, "decoder with num_split > 1 needs the extender weights");
    lsl_decoder *dec = new (std::nothrow) lsl_decoder();
    if (!dec) return fail(-5, "out of host memory");
    dec->d = d;
    dec->w = *w;
    if (d.num_block_attn) dec->self_blocks.assign(w->self_blocks, w->self_blocks + d.num_block_attn);
    if (d.num_block_cross) dec->cross_blocks.assign(w->cross_blocks, w->cross_blocks + d.num_block_cross);
    dec->w.self_blocks = dec->self_blocks.data();
    dec->w.cross_blocks = dec->cross_blocks.data();
    *out = dec;
    return 0;
} catch (const std::bad_alloc &) {
    return fail(-5, "out of host memory");
} catch (...) {
    return fail(-11, "unexpected C++ exception");
}

void lsl_decoder_destroy(lsl_decoder *d) { delete d; }

size_t lsl_decode_workspace_bytes(const lsl_decoder *d, int32_t frames, int32_t L, int32_t A) {
    if (!d || frames <= 0 || L <= 0 || A <= 0) return 0;
    return dec_carve(d->d, frames, L, A, nullptr, nullptr);
}

// Decoder.forward (decoder.py:88-102) after post_quant (lightning_base.py:28-31,42-44)
int lsl_decode(lsl_decoder *dec, const float *z, const int64_t *entities, int32_t frames, int32_t L, int32_t A, float *out, void *workspace,
               size_t workspace_bytes, void *stream) try {
    DeviceGuard dev_guard_((hipStream_t)stream);
    if (!dec || !z || !entities || !out) return fail(-1, "null decode argument");
    if (frames <= 0 || L <= 0 || A <= 0) return fail(-3, "decode: empty input");
    const lsl_decoder_desc &d = dec->d;
    const lsl_decoder_weights &w = dec->w;
    if (workspace_bytes < dec_carve(d, frames, L, A, nullptr, nullptr) || !workspace) return fail(-4, "decode workspace too small");
    hipStream_t st = (hipStream_t)stream;
    DecWs ws;
    dec_carve(d, frames, L, A, (char *)workspace, &ws);
    const int nl = frames * L, na = frames * A;
    // post_quant: LayerNorm(C, elementwise_affine=False) then Linear(C, dim_latent)
    dec_ln(ws.xn, z, nullptr, nullptr, nl, d.in_dim, st);
    dec_dense(0, ws.lat, ws.xn, w.pq_w, w.pq_b, nullptr, nl, d.in_dim, d.dim_latent, st);
    // queries = query_mlp(entity_embedding(entities))   (dropout is identity in eval)
    hipLaunchKernelGGL(k_dec_gather, dim3((na + 3) / 4), dim3(256), 0, st, ws.xn, w.table, entities, na, d.dim_emb, d.n_entities);
    dec_dense(0, ws.q, ws.xn, w.qm_w, w.qm_b, nullptr, na, d.dim_emb, d.dim_query, st);
    for (int i = 0; i < d.num_block_attn; ++i)
        if (int rc = dec_block(w.self_blocks[i], ws.lat, L, d.dim_latent, nullptr, 0, 0, d.heads_latent, d.dim_head_latent, d.act, frames, ws, st)) return rc;
    for (int i = 0; i < d.num_block_cross; ++i)
        if (int rc = dec_block(w.cross_blocks[i], ws.lat, L, d.dim_latent, ws.q, A, d.dim_query, d.heads_cross, d.dim_head_cross, d.act, frames, ws, st)) return rc;
    const float *ctx = ws.lat;
    int Lc = L;
    if (d.num_split > 1) {  // extender: 1x1 conv D -> D*N per latent, "B (D N) L -> B (L N) D"; the host reordered the rows to (N, D)
        dec_dense(0, ws.ext, ws.lat, w.ext_w, w.ext_b, nullptr, nl, d.dim_latent, d.num_split * d.dim_latent, st);
        ctx = ws.ext;
        Lc = L * d.num_split;
    }
    if (int rc = dec_block(w.out_block, ws.q, A, d.dim_query, ctx, Lc, d.dim_latent, d.heads_cross, d.dim_head_cross, d.act, frames, ws, st)) return rc;
    dec_dense(d.act, ws.hid, ws.q, w.head_w1, w.head_b1, nullptr, na, d.dim_query, d.dim_query, st);
    dec_dense(0, out, ws.hid, w.head_w2, w.head_b2, nullptr, na, d.dim_query, d.out_dim, st);
    LSL_CHECK_LAUNCH("lsl_decode");
    return 0;
} catch (const std::bad_alloc &) {
    return fail(-5, "out of host memory");
} catch (...) {
    return fail(-11, "unexpected C++ exception");
}

int lsl_encoder_create(const lsl_encoder_desc *desc, const lsl_encoder_weights *w, lsl_encoder **out) try {
    if (!desc || !w || !out) return fail(-1, "null encoder argument");
    const lsl_encoder_desc &d = *desc;
    if (d.dim_input % 4 || d.dim_emb % 4 || d.dim_latent % 4 || (d.heads_latent * d.dim_head_latent) % 4 || (d.heads_cross * d.dim_head_cross) % 4)
        return fail(-3, "encoder widths must be multiples of 4");
    if (d.dim_head_latent > 64 || d.dim_head_cross > 64 || d.dim_head_latent < 1 || d.dim_head_cross < 1) return fail(-3, "encoder dim_head must be 1..64");
    if (d.act != 1 && d.act != 2) return fail(-3, "encoder activation must be 1 (erf GELU) or 2 (tanh GELU)");
    if (d.num_block_attn < 0 || d.num_block_cross < 0 || d.num_latents < 1 || d.n_entities < 1) return fail(-3, "bad encoder description");
    lsl_encoder *enc = new (std::nothrow) lsl_encoder();
    if (!enc) return fail(-5, "out of host memory");
    enc->d = d;
    enc->w = *w;
    if (d.num_block_cross) enc->cross_blocks.assign(w->cross_blocks, w->cross_blocks + d.num_block_cross);
    if (d.num_block_attn) enc->self_blocks.assign(w->self_blocks, w->self_blocks + d.num_block_attn);
    enc->w.cross_blocks = enc->cross_blocks.data();
    enc->w.self_blocks = enc->self_blocks.data();
    *out = enc;
    return 0;
} catch (const std::bad_alloc &) {
    return fail(-5, "out of host memory");
} catch (...) {
    return fail(-11, "unexpected C++ exception");
}

void lsl_encoder_destroy(lsl_encoder *e) { delete e; }

size_t lsl_encode_workspace_bytes(const lsl_encoder *e, int32_t frames, int32_t A) {
    if (!e || frames <= 0 || A <= 0) return 0;
    return enc_carve(e->d, frames, A, nullptr, nullptr, nullptr);
}

// quant(Encoder.forward(x, entities, mask))   (encoder.py:96-103, lightning_base.py:37-40)
int lsl_encode(lsl_encoder *enc, const float *x, const int64_t *entities, const unsigned char *mask, int32_t frames, int32_t A, float *out,
               void *workspace, size_t workspace_bytes, void *stream) try {
    DeviceGuard dev_guard_((hipStream_t)stream);
    if (!enc || !x || !entities || !out) return fail(-1, "null encode argument");
    if (frames <= 0 || A <= 0) return fail(-3, "encode: empty input");
    const lsl_encoder_desc &d = enc->d;
    const lsl_encoder_weights &w = enc->w;
    if (workspace_bytes < enc_carve(d, frames, A, nullptr, nullptr, nullptr) || !workspace) return fail(-4, "encode workspace too small");
    hipStream_t st = (hipStream_t)stream;
    DecWs ws;
    float *ctx;
    enc_carve(d, frames, A, (char *)workspace, &ws, &ctx);
    const int N = d.num_latents, nl = frames * N, na = frames * A, dim_ctx = d.dim_input + d.dim_emb;
    // prepare_inputs of EncoderBase: context = mlp(cat(x, entity_embedding(entities))), latents = the learned array per frame
    hipLaunchKernelGGL(k_enc_context, dim3((na + 3) / 4), dim3(256), 0, st, ws.xn, x, w.table, entities, na, d.dim_input, d.dim_emb, d.n_entities);
    dec_dense(d.act, ws.hid, ws.xn, w.mlp_w1, w.mlp_b1, nullptr, na, dim_ctx, d.dim_latent, st);
    dec_dense(0, ctx, ws.hid, w.mlp_w2, w.mlp_b2, nullptr, na, d.dim_latent, dim_ctx, st);
    const long total = (long)nl * d.dim_latent;
    hipLaunchKernelGGL(k_enc_broadcast, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ws.lat, w.latents, total, N * d.dim_latent);
    for (int i = 0; i < d.num_block_cross; ++i)
        if (int rc = dec_block(w.cross_blocks[i], ws.lat, N, d.dim_latent, ctx, A, dim_ctx, d.heads_cross, d.dim_head_cross, d.act, frames, ws, st, mask)) return rc;
    for (int i = 0; i < d.num_block_attn; ++i)
        if (int rc = dec_block(w.self_blocks[i], ws.lat, N, d.dim_latent, nullptr, 0, 0, d.heads_latent, d.dim_head_latent, d.act, frames, ws, st)) return rc;
    // quant: Linear(dim_latent, dim_latent) then LayerNorm(dim_latent, elementwise_affine=False)
    dec_dense(0, ws.hid, ws.lat, w.quant_w, w.quant_b, nullptr, nl, d.dim_latent, d.dim_latent, st);
    dec_ln(out, ws.hid, nullptr, nullptr, nl, d.dim_latent, st);
    LSL_CHECK_LAUNCH("lsl_encode");
    return 0;
} catch (const std::bad_alloc &) {
    return fail(-5, "out of host memory");
} catch (...) {
    return fail(-11, "unexpected C++ exception");
}
