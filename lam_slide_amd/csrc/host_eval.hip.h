// Host side, part 3 of 4: the kernel sequence of one network evaluation (conditioning tables, sub-blocks, embedding, head), of a pass, and
// of the trajectory-resident sampler; argument checks of a call.  Inside the anonymous namespace opened by host_common.hip.h.
#pragma once

// ---- pieces of one evaluation ----------------------------------------------------------------------

// conditioning vector -> all modulation tables for `rows` trajectories (latent_si_v31.py:176-178,
// mmdit.py:184-197).  t_dev == nullptr: scalar t.  yemb == nullptr: no class conditioning.
int run_mods(lsl_model *m, const Workspace &ws, const float *t_dev, float t_scalar, const float *yemb, int rows,
             float *vec_out, float *mods_out, hipStream_t st) {
    const lsl_weights &w = m->w;
    const int D = m->d.hidden;
    m->prof.begin(6, st);
    hipLaunchKernelGGL(k_time_features, dim3((rows * 128 + 255) / 256), dim3(256), 0, st, ws.tfeat, t_dev, t_scalar, w.time_freqs, rows);
    const bool single = !t_dev && !yemb;  // shared scalar time, no class vector: one row whatever the batch
    launch_dense<false, true>(ws.hid, ws.tfeat, w.time_w1, w.time_b1, nullptr, rows, 256, D, 0, st, single);
    launch_dense<false, false>(vec_out, ws.hid, w.time_w2, w.time_b2, yemb, rows, D, D, D, st, single);
    launch_dense<true, false>(mods_out, vec_out, w.mod_w, w.mod_b, nullptr, rows, D, m->MODW, 0, st, single);
    m->prof.end(6, st);
    LSL_CHECK_LAUNCH("modulation");
    return 0;
}

// The same tables for `count` sampler records at once (shared scalar time, no class vector: one row per record).  Every row goes
// through the kernels run_mods uses for its single row (k_dense_rows: a row's sum does not depend on the other rows of the launch),
// so a record's table has the same bits as the one run_mods computes in front of a single evaluation.
int run_mods_steps(lsl_model *m, const Workspace &ws, const float *times, int count, hipStream_t st) {
    const lsl_weights &w = m->w;
    const int D = m->d.hidden;
    m->prof.begin(6, st);
    for (int c0 = 0; c0 < count; c0 += 48) {
        StepTimes tt;
        const int nc = std::min(48, count - c0);
        for (int s = 0; s < nc; ++s) tt.t[s] = times[c0 + s];
        hipLaunchKernelGGL(k_time_features_steps, dim3((nc * 128 + 255) / 256), dim3(256), 0, st, ws.tf_all + (size_t)c0 * 256, tt, nc, 1, w.time_freqs);
    }
    const unsigned gy = (unsigned)((count + 7) / 8);  // 8 rows per workgroup
    hipLaunchKernelGGL((k_dense_rows<false, true>), dim3((D + 3) / 4, gy), dim3(256), 0, st, ws.hid_all, ws.tf_all, w.time_w1, w.time_b1, nullptr, count, 256, D, 0, 0);
    hipLaunchKernelGGL((k_dense_rows<false, false>), dim3((D + 3) / 4, gy), dim3(256), 0, st, ws.vec_all, ws.hid_all, w.time_w2, w.time_b2, nullptr, count, D, D, D, 0);
    // (the wide last layer: a workgroup's four weight rows are its HBM traffic, re-read once per row range - 16 rows per workgroup)
    hipLaunchKernelGGL((k_dense_rows<true, false>), dim3((m->MODW + 3) / 4, (unsigned)((count + 15) / 16)), dim3(256), 0, st, ws.mods_all, ws.vec_all, w.mod_w, w.mod_b, nullptr, count, D, m->MODW, 0, 0);
    m->prof.end(6, st);
    LSL_CHECK_LAUNCH("modulation (group of records)");
    return 0;
}

// vec_in(y) (mmdit.py:118-126), constant over a sample
int run_yemb(lsl_model *m, const Workspace &ws, const float *y, int rows, hipStream_t st) {
    const lsl_weights &w = m->w;
    const int D = m->d.hidden, V = m->d.vec_in_dim;
    launch_dense<false, true>(ws.hid, y, w.vec_w1, w.vec_b1, nullptr, rows, V, D, 0, st);
    launch_dense<false, false>(ws.yemb, ws.hid, w.vec_w2, w.vec_b2, nullptr, rows, D, D, 0, st);
    LSL_CHECK_LAUNCH("vec_in");
    return 0;
}

void run_tables(const lsl_model *m, const Workspace &ws, int T, int L, hipStream_t st) {
    const int half = m->d.head_dim_pad / 2;
    if (ws.w2p) {  // linear2 weights in MFMA-fragment order (k_linear2_ws keeps them in registers for a whole launch: every load 1 KiB contiguous)
        const size_t per = (size_t)m->d.hidden * m->K2;
        for (int bi = 0; bi < 2 * m->d.depth; ++bi)
            hipLaunchKernelGGL(k_lin2_pack, dim3(128), dim3(256), 0, st, ws.w2p + (size_t)bi * per, (const u16 *)m->blocks[bi].w2, m->d.hidden, m->K2);
    }
    if (ws.wtail) {  // tail models: the weight stream of every sub-block in the order k_tail consumes it
        for (int bi = 0; bi < 2 * m->d.depth; ++bi)
            hipLaunchKernelGGL(k_tail_pack, dim3(256), dim3(256), 0, st, ws.wtail + (size_t)bi * ws.wtail_stride, (const u16 *)m->blocks[bi].w1,
                               (const u16 *)m->blocks[bi].w2, m->d.hidden, m->HHD, m->d.mlp_dim);
    }
    hipLaunchKernelGGL(k_rope_table, dim3((L * half + 255) / 256), dim3(256), 0, st, ws.rope_l, L, m->d.head_dim, m->d.head_dim_pad, m->d.theta);
    hipLaunchKernelGGL(k_rope_table, dim3((T * half + 255) / 256), dim3(256), 0, st, ws.rope_t, T, m->d.head_dim, m->d.head_dim_pad, m->d.theta);
    // the same tables with each attention block's query / key norm scales folded in (spatial blocks: L positions, temporal: T)
    const int nb = 2 * m->d.depth;
    for (int b0 = 0; b0 < 2 * nb; b0 += 16) {
        RopeScaledJobs jobs{};
        jobs.n_jobs = std::min(16, 2 * nb - b0);
        int max_pos = 0;
        for (int k = 0; k < jobs.n_jobs; ++k) {
            const int t = b0 + k, bi = t >> 1;
            jobs.out[k] = ws.rope_qk + (size_t)t * ws.rope_qk_stride;
            jobs.scale[k] = (t & 1) ? m->blocks[bi].ks : m->blocks[bi].qs;
            jobs.n_pos[k] = (bi & 1) ? T : L;
            jobs.sq_bound[k] = ws.kmax2 + ((t & 1) ? 0 : nb) + bi;
            max_pos = std::max(max_pos, jobs.n_pos[k]);
        }
        hipLaunchKernelGGL(k_rope_scaled, dim3((max_pos * half + 255) / 256, jobs.n_jobs), dim3(256), 0, st, jobs, m->d.head_dim, m->d.head_dim_pad, m->d.theta);
    }
}

// one ParallelMLPAttentionV2 sub-block on h (in place): LN+modulate -> linear1 -> attention -> linear2
// a_ready: ws.a already holds this sub-block's LayerNorm + modulate (written by the previous sub-block's linear2); fuse_next: let this
// sub-block's linear2 write the next one's when the launch allows it (*a_written reports whether it did)
int run_block(lsl_model *m, const Workspace &ws, int bi, float *h, const float *mods, int mod_stride, int bc, int T, int L,
              hipStream_t st, bool a_ready = false, bool fuse_next = false, bool *a_written = nullptr, bool stop_before_linear2 = false,
              bool stats_ready = false, bool *stats_written = nullptr) {
    const lsl_model_desc &d = m->d;
    const lsl_block_weights &bw = m->blocks[bi];
    const int D = d.hidden, n = bc * T * L, layer = bi / 2, temporal = bi & 1;
    const float *mbase = mods + (size_t)layer * 6 * D + (temporal ? 3 * D : 0);  // shift, scale, gate
    // handles with lsl_model_set_ln_fuse: linear1 normalises the fp32 residual stream on load (no LayerNorm launch, ws.a unused); never for
    // the debug taps (which hand out `a`) and only on the workspace's own padded stream
    const bool tail_m = m->tail && ws.wtail && !stop_before_linear2;
    // (every condition depends on the model and on T, L only - never on the batch: a trajectory's bits must not)
    const bool lnf_shape = m->ln_fuse && !tail_m && ws.lnstat && ws.w2p && h == ws.h &&  // (k_tail reads `a` itself; the statistics come from k_linear2_ws)
                           linear1_lnf_ok(d.head_dim_pad, D, m->F1, m->HHD, n, T * L, mod_stride);
    const bool lnf = lnf_shape && stats_ready && !a_ready && !stop_before_linear2;  // ws.lnstat holds the statistics of h (the previous linear2's)
    if (stats_written) *stats_written = false;
    if (!a_ready && !lnf) {
        m->prof.begin(3, st);
        DISPATCH_D(D, launch_ln_mod_t, ws.a, h, mbase, mbase + D, mod_stride, n, T * L, st);
        m->prof.end(3, st);
    }
    m->prof.begin(0, st);

    // softmax attention: log2(e) / sqrt(head_dim) rides on q (the kernels use exp2); attention_linear takes the plain normalised, rotated q
    const float premul = m->attention_linear ? 1.0f : (float)(1.4426950408889634 / std::sqrt((double)d.head_dim));
    // position of token n along the attended axis = (n / pdiv) % pmod, done with multiply-high in the epilogue: exact while
    // n * d < 2^32, and n < 2^18 (pass size) with d <= T or L
    const int pdiv = temporal ? L : 1, pmod = temporal ? T : L;
    auto magic_of = [](int dv) { return dv == 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)dv + 1); };
    if ((unsigned long long)n * (unsigned)std::max(pdiv, pmod) >= (1ull << 32)) return fail(-3, "pass too large for the position arithmetic");
    // tail models (and never the debug taps, which hand out the GELU'd mlp half of z): linear1 computes q | k | v only
    const bool tail = tail_m;
    const int F1 = tail ? 3 * m->HHD : m->F1;
    const bool lin1_ts = linear1_ts_ok(d.head_dim_pad, D, F1, m->HHD, n);
    const int npad = (n + 255) & ~255;
    const bool planes = !m->attention_linear && qkv_planes_ok(d.head_dim_pad, D, d.heads, temporal ? T : L, temporal != 0, lin1_ts);
    // head-major planes are addressed with 32-bit per-lane byte offsets over the whole q | k | v buffer (k_lin1.hip.h flush, k_attn.hip.h
    // stream requests): a pass set larger than that through lsl_model_set_chunk / LSL_CHUNK_TRAJ is refused, never wrapped
    if (planes && (unsigned long long)npad * 3ull * (unsigned)m->HHD * 2ull >= (1ull << 32)) return fail(-3, "pass too large for the q/k/v plane offsets (%d tokens: at most %llu with this model)", n, (unsigned long long)((1ull << 32) / (6ull * (unsigned)m->HHD)) - 256);
    if (lin1_ts) {
        const Lin1Args la{(const u16 *)bw.w1, lnf ? (const u16 *)h : ws.a, bw.b1, ws.rope_qk + (size_t)(2 * bi) * ws.rope_qk_stride,
                          ws.rope_qk + (size_t)(2 * bi + 1) * ws.rope_qk_stride, ws.qkv, ws.z, F1, n, m->HHD, d.mlp_dim,
                          pdiv, pmod, magic_of(pdiv), magic_of(pmod), 1.0f / d.head_dim, premul, 1, 0, planes ? 1 : 0, npad,
                          ws.lnstat, mbase, mbase + D, mod_stride, T * L, magic_of(T * L)};
        if (lnf) {
            launch_linear1_lnf(d.head_dim_pad, D, la, st);
            m->prof.label(0, "k_linear1_ts<%d, %d, 8, true>%s", d.head_dim_pad, D, tail ? " (LayerNorm | q | k | v)" : " (LayerNorm fused)");
        } else {
            launch_linear1_ts(d.head_dim_pad, D, la, st);
            m->prof.label(0, "k_linear1_ts<%d, %d, %d>%s", d.head_dim_pad, D, linear1_ts_waves(D, n), tail ? " (q | k | v)" : "");
        }
    } else if (d.head_dim_pad == 32) {
        EpiLinear1<32> e{bw.b1, bw.qs, bw.ks, temporal ? ws.rope_t : ws.rope_l, ws.rope_qk + (size_t)(2 * bi) * ws.rope_qk_stride,
                         ws.rope_qk + (size_t)(2 * bi + 1) * ws.rope_qk_stride, ws.qkv, ws.z, m->HHD, d.mlp_dim,
                         pdiv, pmod, magic_of(pdiv), magic_of(pmod), 1.0f / d.head_dim, premul, 0};
        launch_gemm((const u16 *)bw.w1, ws.a, F1, n, D, e, st, m->HHD);
    } else {
        EpiLinear1<16> e{bw.b1, bw.qs, bw.ks, temporal ? ws.rope_t : ws.rope_l, ws.rope_qk + (size_t)(2 * bi) * ws.rope_qk_stride,
                         ws.rope_qk + (size_t)(2 * bi + 1) * ws.rope_qk_stride, ws.qkv, ws.z, m->HHD, d.mlp_dim,
                         pdiv, pmod, magic_of(pdiv), magic_of(pmod), 1.0f / d.head_dim, premul, 0};
        launch_gemm((const u16 *)bw.w1, ws.a, F1, n, D, e, st, m->HHD);
    }
    if (!lin1_ts) m->prof.label(0, "k_gemm_glds<EpiLinear1<%d>> (tiling %d)", d.head_dim_pad, gemm_variant<EpiLinear1<32>>(F1, D, n));
    m->prof.end(0, st);
    static const int nt_mask = tune_int("LSL_NT", 3);
    AttnArgs aa{};
    aa.nt = (nt_mask >> 2) & 1;
    aa.qkv = ws.qkv;
    aa.z = ws.z;
    aa.HHD = m->HHD;
    aa.zw = m->K2;
    aa.H = d.heads;
    aa.hd = d.head_dim;
    static const int attn_bound = tune_int("LSL_ATTN_BOUND", 1);
    aa.kmax2 = ws.kmax2 + bi;
    aa.qmax2 = ws.kmax2 + 2 * d.depth + bi;
    aa.premul = premul;
    aa.planes = planes ? 1 : 0;
    aa.npad = npad;
    aa.bound = attn_bound == 2 || (attn_bound == 1 && (temporal ? T : L) > 96);  // short axes: the max pass is one or two tiles, cheaper than the norms
    if (!temporal) {  // sequences (b,t), positions l
        aa.S = L; aa.n_seq = bc * T; aa.inner = 1; aa.outer_stride = L; aa.pos_stride = 1;
    } else {          // sequences (b,l), positions t
        aa.S = T; aa.n_seq = bc * L; aa.inner = L; aa.outer_stride = T * L; aa.pos_stride = L;
    }
    m->prof.begin(2, st);
    if (m->attention_linear) {
        if (d.head_dim_pad == 32) launch_attention_linear_t<32>(aa, st);
        else launch_attention_linear_t<16>(aa, st);
    } else if (d.head_dim_pad == 32) launch_attention_t<32>(aa, st);
    else launch_attention_t<16>(aa, st);
    m->prof.label(2, "%s", m->attention_linear ? "k_attention_linear" : attention_stream_mode(aa.S, aa.H) || attention_grouped_ok(aa) ? "k_attention_stream" : "k_attention_rows / k_attention_tiny / k_attention");
    m->prof.end(2, st);

    if (stop_before_linear2) {  // (lsl_debug_taps)
        LSL_CHECK_LAUNCH("block");
        return 0;
    }
    m->prof.begin(1, st);
    if ((unsigned long long)n * (unsigned)(T * L) >= (1ull << 32)) return fail(-3, "pass too large for the trajectory arithmetic");
    if (tail) {  // up-projection -> GELU -> down-projection + out-projection + gated residual + the next sub-block's LayerNorm + modulate
        if (h != ws.h) return fail(-3, "the tail kernel runs on the workspace's residual stream");
        if ((unsigned long long)npad * (unsigned)(4 * D) >= (1ull << 32)) return fail(-3, "pass too large for the residual-stream offsets");
        const bool next = bi + 1 < 2 * d.depth;
        const float *nb = mods + (size_t)((bi + 1) / 2) * 6 * D + (((bi + 1) & 1) ? 3 * D : 0);  // next sub-block: shift, scale
        const TailArgs ta{ws.wtail + (size_t)bi * ws.wtail_stride, ws.a, ws.z, bw.b1 + 3 * m->HHD, bw.b2, mbase + 2 * D, h, next ? ws.a : nullptr,
                          nb, nb + D, n, d.mlp_dim, m->K2, mod_stride, T * L, magic_of(T * L)};
        launch_tail(ta, st);
        m->prof.label(1, "k_tail<%d, %d>", D, m->HHD);
        if (a_written) *a_written = next;
        m->prof.end(1, st);
        LSL_CHECK_LAUNCH("block (tail)");
        return 0;
    }
    const bool fuse = fuse_next && !lnf_shape && bi + 1 < 2 * d.depth && linear2_can_fuse_ln(D, n, m->K2);  // (where the ln_fuse form applies, linear1 normalises)
    const float *nbase = mods + (size_t)((bi + 1) / 2) * 6 * D + (((bi + 1) & 1) ? 3 * D : 0);  // next sub-block: shift, scale
    bool on_ws = false;
    if (ws.w2p && !fuse && (unsigned long long)n * (unsigned)(4 * D) < (1ull << 32)) {  // (32-bit byte offsets into h)
        // ln_fuse handles: the rows' statistics for the NEXT sub-block's LayerNorm (inside its linear1) leave with the update
        const bool stats = lnf_shape && bi + 1 < 2 * d.depth && !stop_before_linear2;
        const Lin2Args l2{ws.w2p + (size_t)bi * D * m->K2, ws.z, bw.b2, mbase + 2 * D, h, D, n, mod_stride, T * L, magic_of(T * L), 0, 0, 0,
                          stats ? ws.lnparts : nullptr, npad};
        on_ws = launch_linear2_ws(m->K2, l2, mod_stride == 0, st);
        if (on_ws) m->prof.label(1, stats ? "k_linear2_ws<%d> (+ row statistics)" : "k_linear2_ws<%d>", m->K2);
        if (on_ws && stats) {
            hipLaunchKernelGGL(k_ln_finalize, dim3((n + 255) / 256), dim3(256), 0, st, ws.lnstat, ws.lnparts, D / 32, npad, n, 32.0f);
            if (stats_written) *stats_written = true;
        }
        // (which LayerNorm form the next sub-block runs must not depend on the launch: a pass the weight-stationary kernel cannot take - more
        // trajectories per token range than its gate table holds - is refused on ln_fuse handles, never served by the other form)
        if (stats && !on_ws) return fail(-3, "ln_fuse: a pass of %d tokens has too many trajectories per token range for k_linear2_ws; use smaller passes (lsl_model_set_chunk)", n);
    } else if (lnf_shape && bi + 1 < 2 * d.depth && !stop_before_linear2 && !fuse) {
        return fail(-3, "ln_fuse: pass too large for the residual-stream offsets (%d tokens)", n);
    }
    if (!on_ws) {
        EpiLinear2 e2{bw.b2, mbase + 2 * D, h, D, mod_stride, T * L, 0, magic_of(T * L), fuse ? ws.a : nullptr, nbase, nbase + D};
        launch_gemm((const u16 *)bw.w2, ws.z, D, n, m->K2, e2, st, 32, fuse);
        m->prof.label(1, "k_gemm_glds<EpiLinear2> (tiling %d)", gemm_variant<EpiLinear2>(D, m->K2, n));
    }
    if (a_written) *a_written = fuse && !on_ws;
    m->prof.end(1, st);
    LSL_CHECK_LAUNCH("block");
    return 0;
}

// One evaluation for a pass of bc trajectories; state already embedded?  No: embeds x first.
// do_step: fuse the affine update into the head; else write the network output to `out`.
int run_eval(lsl_model *m, const Workspace &ws, float *x, float *out, const float *t_dev, float t_scalar, bool have_y, int bc,
             int T, int L, int do_step, float ax, float am, float aw, const float *noise, uint64_t seed, unsigned step,
             uint64_t elem_off, float *trace, hipStream_t st, float as = 0.0f, const float *saved = nullptr, float *save_out = nullptr,
             const float *mods_ready = nullptr) {
    const lsl_model_desc &d = m->d;
    const int D = d.hidden, n = bc * T * L;
    // modulation rows: one per trajectory, or a single shared row when t is a scalar and there is no y
    const bool shared = (t_dev == nullptr) && !have_y;
    const int rows = shared ? 1 : bc;
    const int mod_stride = shared ? 0 : m->MODW;
    int rc = 0;
    const float *mods = ws.mods;
    if (mods_ready && shared) mods = mods_ready;  // this record's row of the group table (run_mods_steps)
    else rc = run_mods(m, ws, t_dev, t_scalar, have_y ? ws.yemb : nullptr, rows, ws.vec, ws.mods, st);
    if (rc) return rc;
    m->prof.begin(5, st);
    // ln_fuse handles (models without the in-place LayerNorm behind the embedding): the embedding leaves the rows' statistics, so that the
    // FIRST sub-block's LayerNorm runs inside its linear1 too (the same shape conditions as run_block's, which decide there again)
    const int npad_e = (n + 255) & ~255;
    const bool emb_stats = m->ln_fuse && !m->tail && !d.normalize && ws.lnstat && ws.w2p && embed_stats_ok(d.in_dim, D) &&
                           linear1_lnf_ok(d.head_dim_pad, D, m->F1, m->HHD, n, T * L, mod_stride);
    launch_embed<1>(ws.h, x, m->w.x_in_w, nullptr, nullptr, nullptr, nullptr, ws.cond_emb, n, d.in_dim, D, st, emb_stats ? ws.lnparts : nullptr, npad_e);
    if (emb_stats) hipLaunchKernelGGL(k_ln_finalize, dim3((n + 255) / 256), dim3(256), 0, st, ws.lnstat, ws.lnparts, D / 256, npad_e, n, 256.0f);
    if (d.normalize) { DISPATCH_D(D, launch_ln_inplace_t, ws.h, n, 1e-5f, st); }
    m->prof.end(5, st);
    LSL_CHECK_LAUNCH("embed");
    bool a_ready = false;  // the first sub-block of an evaluation runs the standalone LayerNorm; later ones get `a` from the previous linear2
    bool stats_ready = emb_stats;  // ... or (ln_fuse handles) the rows' statistics, and normalise inside their linear1
    for (int bi = 0; bi < 2 * d.depth; ++bi) {
        bool wrote = false, wrote_stats = false;
        rc = run_block(m, ws, bi, ws.h, mods, mod_stride, bc, T, L, st, a_ready, true, &wrote, false, stats_ready, &wrote_stats);
        if (rc) return rc;
        a_ready = wrote;
        stats_ready = wrote_stats;
    }
    const float *fm = mods + (size_t)d.depth * 6 * D;  // adaLN: shift, scale
    m->prof.begin(4, st);
    DISPATCH_D(D, launch_head_t, x, out, ws.h, fm, fm + D, mod_stride, m->w.out_w, m->w.out_b, n, d.in_dim, T * L, do_step, ax, am, aw,
               noise, (unsigned long long)seed, step, (unsigned long long)elem_off, trace, as, saved, save_out, st);
    m->prof.end(4, st);
    LSL_CHECK_LAUNCH("head");
    return 0;
}

int prepare_pass(lsl_model *m, const Workspace &ws, const float *x_cond, const int64_t *mask, const float *y, int bc, int T,
                 int L, hipStream_t st) {
    const lsl_model_desc &d = m->d;
    const int n = bc * T * L;
    launch_embed<0>(ws.cond_emb, x_cond, m->w.cond_w, m->w.cond_b, m->w.x_in_b, m->w.mask_emb, mask, nullptr, n, d.in_dim, d.hidden, st);
    LSL_CHECK_LAUNCH("cond_embed");
    if (y) return run_yemb(m, ws, y, bc, st);
    return 0;
}

// ---- trajectory-resident path (k_resident.hip.h): models whose whole trajectory fits one workgroup's LDS ----------------------
// The choice depends on the MODEL and on T*L only, never on the batch: a trajectory's bits are the same in any batch / shard / pass.
bool resident_ok(const lsl_model *m, int T, int L) {
    static const int off = env_int("LSL_RESIDENT", 1) == 0;  // documented runtime switch: 0 = always the general path
    const lsl_model_desc &d = m->d;
    return !off && !m->attention_linear && d.hidden == RES_D && d.heads == RES_H && d.head_dim == RES_HD && d.head_dim_pad == RES_HD && d.mlp_dim == RES_M &&
           d.in_dim <= RES_MAX_C && d.in_dim % 4 == 0 && 2 * d.depth <= RES_MAX_BLOCKS && (long)T * L <= 48 && T <= 32 && L <= 32;
}

struct ResWorkspace {
    float *cond_emb, *yemb, *tfeat, *hid, *vec, *mods, *blkpar;
    u16 *blkw;
    int steps_per_launch;
    size_t bytes;
};
ResWorkspace carve_resident(const lsl_model *m, char *base, int B, int T, int L, bool have_y) {
    const size_t n = (size_t)B * T * L, D = m->d.hidden;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char *p = base ? base + off : nullptr;
        off += align_up(bytes, 256);
        return p;
    };
    ResWorkspace ws;
    const size_t rows = have_y ? (size_t)B : 1;
    // modulation tables of a whole group of state updates are computed before the group's single launch: bound them to 256 MiB
    size_t spl = ((size_t)256 << 20) / (rows * m->MODW * 4);
    ws.steps_per_launch = (int)std::max<size_t>(1, std::min<size_t>(spl, RES_MAX_STEPS));
    const size_t rt = rows * ws.steps_per_launch;
    ws.cond_emb = (float *)take(n * D * 4);
    ws.yemb = (float *)take((size_t)B * D * 4);
    ws.tfeat = (float *)take(rt * 256 * 4);
    ws.hid = (float *)take(std::max(rt, (size_t)B) * D * 4);
    ws.vec = (float *)take(rt * D * 4);
    ws.mods = (float *)take(rt * m->MODW * 4);
    ws.blkpar = (float *)take((size_t)2 * m->d.depth * RES_P_SHIFT * 4);
    ws.blkw = (u16 *)take((size_t)2 * m->d.depth * (RES_W1_ELEMS + RES_W2_ELEMS) * 2);
    ws.bytes = off;
    return ws;
}

template <int NNT>
void launch_resident(const ResArgs &a, int B, int T, int L, hipStream_t st) {
    auto kern = k_resident<NNT>;
    const size_t lds = ResLds<NNT>::bytes(T, L);
    LSL_ALLOW_LDS(kern, (size_t)163840);
    hipLaunchKernelGGL(kern, dim3(B), dim3(RES_NTHR), lds, st, a);
}

int resident_sample(lsl_model *m, const lsl_io *io, const lsl_step *steps, int n_steps, const float *noise, uint64_t seed, uint64_t elem_offset,
                    float *trace, void *workspace, hipStream_t st) {
    const lsl_model_desc &d = m->d;
    const lsl_weights &w = m->w;
    const int B = io->B, T = io->T, L = io->L, n_t = T * L, D = d.hidden;
    const bool have_y = io->y != nullptr;
    const ResWorkspace ws = carve_resident(m, (char *)workspace, B, T, L, have_y);
    const int rows = have_y ? B : 1;
    if (have_y) {
        launch_dense_small<false, true>(ws.hid, io->y, w.vec_w1, w.vec_b1, nullptr, B, d.vec_in_dim, D, 0, st);
        launch_dense_small<false, false>(ws.yemb, ws.hid, w.vec_w2, w.vec_b2, nullptr, B, D, D, 0, st);
        LSL_CHECK_LAUNCH("vec_in");
    }
    ResArgs a;
    a.cond_emb = ws.cond_emb;
    a.x_cond = io->x_cond;
    a.mask = (const int64_t *)io->mask;
    a.cond_w = w.cond_w; a.cond_b = w.cond_b; a.x_in_b = w.x_in_b; a.mask_emb = w.mask_emb;
    a.x = io->x;
    a.mods = ws.mods;
    a.mods_step_stride = (long)rows * m->MODW;
    a.mods_traj_stride = have_y ? m->MODW : 0;
    a.x_in_w = w.x_in_w;
    a.out_w = w.out_w;
    a.out_b = w.out_b;
    a.noise = noise;
    a.noise_step_stride = (long)B * n_t * d.in_dim;
    a.seed = seed;
    a.elem_offset = elem_offset;
    a.trace = trace;
    a.trace_step_stride = (long)B * n_t * d.in_dim;
    a.n_t = n_t; a.T = T; a.L = L; a.C = d.in_dim; a.depth = d.depth; a.normalize = d.normalize;
    a.theta = d.theta;
    a.skip = tune_int("LSL_RES_SKIP", 0);
    a.q_premul = (float)(1.4426950408889634 / std::sqrt((double)d.head_dim));
    ResPack pack;
    for (int bi = 0; bi < 2 * d.depth; ++bi) {
        const lsl_block_weights &bw = m->blocks[bi];
        const u16 *wb = ws.blkw + (size_t)bi * (RES_W1_ELEMS + RES_W2_ELEMS);
        a.blk[bi] = ResBlock{wb, wb + RES_W1_ELEMS};
        pack.w1[bi] = (const u16 *)bw.w1; pack.w2[bi] = (const u16 *)bw.w2;
        pack.b1[bi] = bw.b1; pack.qs[bi] = bw.qs; pack.ks[bi] = bw.ks; pack.b2[bi] = bw.b2;
    }
    hipLaunchKernelGGL(k_res_pack, dim3(2 * d.depth, 49), dim3(256), 0, st, ws.blkw, ws.blkpar, pack);
    LSL_CHECK_LAUNCH("k_res_pack");
    a.blkpar = ws.blkpar;
    for (int s0 = 0; s0 < n_steps; s0 += ws.steps_per_launch) {
        const int ns = std::min(ws.steps_per_launch, n_steps - s0);
        StepTimes tt;
        for (int s = 0; s < ns; ++s) {
            tt.t[s] = steps[s0 + s].t;
            a.step[s] = make_float4(steps[s0 + s].t, steps[s0 + s].ax, steps[s0 + s].am, steps[s0 + s].aw);
        }
        const int rt = ns * rows;
        // conditioning vector -> modulation tables of the group's steps (latent_si_v31.py:176-178, mmdit.py:184-197); the tiled kernel
        // is used for any row count, so a trajectory's tables do not depend on the batch it is sampled in
        hipLaunchKernelGGL(k_time_features_steps, dim3((rt * 128 + 255) / 256), dim3(256), 0, st, ws.tfeat, tt, ns, rows, w.time_freqs);
        launch_dense_small<false, true>(ws.hid, ws.tfeat, w.time_w1, w.time_b1, nullptr, rt, 256, D, 0, st);
        launch_dense_small<false, false>(ws.vec, ws.hid, w.time_w2, w.time_b2, have_y ? ws.yemb : nullptr, rt, D, D, D, st, have_y ? B : 0);
        launch_dense_small<true, false>(ws.mods, ws.vec, w.mod_w, w.mod_b, nullptr, rt, D, m->MODW, 0, st);
        LSL_CHECK_LAUNCH("modulation");
        a.step0 = (unsigned)s0;
        a.n_steps = ns;
        if (n_t <= 32) launch_resident<2>(a, B, T, L, st);
        else launch_resident<3>(a, B, T, L, st);
        LSL_CHECK_LAUNCH("k_resident");
    }
    return 0;
}

int check_call(const lsl_model *m, const lsl_io *io, size_t ws_bytes, void *ws, int *chunk_out) {
    if (!m || !io) return fail(-1, "null model or io");
    if (!m->has_weights) return fail(-2, "weights not set");
    if (io->B <= 0 || io->T <= 0 || io->L <= 0) return fail(-3, "B, T, L must be positive");
    if (!io->x || !io->x_cond || !io->mask) return fail(-3, "x, x_cond and mask are required");
    if ((io->y != nullptr) != (m->d.vec_in_dim > 0) && io->y != nullptr) return fail(-3, "y given but the model has no vec_in");
    if ((size_t)io->T * io->L > (1u << 24)) return fail(-3, "T*L too large");
    const int chunk = default_chunk(m, io->B, io->T, io->L);
    size_t need = carve(m, nullptr, chunk, io->T, io->L).bytes;
    if (resident_ok(m, io->T, io->L)) need = std::max(need, carve_resident(m, nullptr, io->B, io->T, io->L, m->d.vec_in_dim > 0).bytes);
    if (!ws || ws_bytes < need) return fail(-4, "workspace too small: need %zu bytes, got %zu", need, ws_bytes);
    *chunk_out = chunk;
    return 0;
}

