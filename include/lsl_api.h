/*
 * lsl_api.h -- C ABI of liblamslide_hip.so: the MI355X (gfx950) implementation of LaM-SLidE's
 * second-stage latent SiT sampling path.
 *
 * The reference implements this path in Python only (no FFI exists there).  Each entry point below
 * names the reference interface it stands in for; the Python binding that a reference maintainer
 * would add is shown in INTEGRATION.md and implemented in lam_slide_amd/_lib.py.
 *
 * Conventions
 *   - Every pointer marked "device" is HBM memory owned by the caller (PyTorch).  The library allocates
 *     no device memory; scratch comes from the caller's workspace (lsl_workspace_bytes).
 *   - All calls are asynchronous: work is enqueued on `stream` (a hipStream_t passed as void*), no
 *     internal synchronisation, no host<->device copies.
 *   - Return value 0 = ok, negative = error; lsl_last_error() returns a thread-local message.
 *     Nothing aborts or throws across this boundary (allocation failures and stray C++ exceptions become error codes).
 *   - Launches go to the device that owns `stream`, whatever the calling thread's current device is.
 *   - Tensors are row-major contiguous.  State layout [B, T, L, C] fp32 exactly as the reference's
 *     LatentSIV3.forward takes it (latent_si_v31.py:168-170).
 */
#ifndef LSL_API_H
#define LSL_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default) /* the library is built with -fvisibility=hidden: these entry points are its whole export list */
#endif

/* 2: lsl_sample_ex takes n_trace; linear1 biases are read in whole 256-feature tiles (b1 zero-padded to a multiple of 256 floats);
 *    lsl_sample_ex, lsl_debug_taps, lsl_build_info exist.
 * 3: lsl_rk_lincomb / lsl_rk_dense / lsl_rk_error_ratio exist (state arithmetic of the adaptive and fixed-grid Runge-Kutta samplers);
 *    no signature of version 2 changed.
 * 4: lsl_model_set_attention_mode exists (attention_linear, mmdit.py:58-72); nothing else changed.
 * 5: lsl_model_set_tail, lsl_model_tail, lsl_profile_kernel_name exist; no signature of version 4 changed.
 * 6: lsl_model_set_ln_fuse, lsl_model_ln_fuse exist; no signature of version 5 changed. */
#define LSL_VERSION 6

typedef struct lsl_model lsl_model;

/* Hyper-parameters of LatentSIV3.__init__ (latent_si_v31.py:68-121) that fix the weight shapes.
 * B, T, L arrive per call. */
typedef struct lsl_model_desc {
    int32_t in_dim;       /* C: in_dim == out_dim                                         */
    int32_t hidden;       /* D: hidden_size, multiple of 64, <= 512                        */
    int32_t heads;        /* H: num_heads                                                  */
    int32_t head_dim;     /* D / H (16, 24 or 32 in the shipped configs)                   */
    int32_t head_dim_pad; /* packed head width: 16 if head_dim == 16, else 32 (zero padded) */
    int32_t mlp_dim;      /* M = int(D * mlp_ratio), multiple of 32                        */
    int32_t depth;        /* number of layers (share_weights is resolved by the packer)    */
    int32_t vec_in_dim;   /* V, or 0 when the model has no vec_in                          */
    int32_t normalize;    /* F.layer_norm after the input embedding (latent_si_v31.py:173) */
    float theta;          /* RoPE base (mmdit.py:75-82)                                    */
} lsl_model_desc;

/* One ParallelMLPAttentionV2 (mmdit.py:215-249), packed by lam_slide_amd/packing.py:
 *   w1  bf16 [F1 rounded up to 256][D]   rows = [q heads | k heads | v heads | mlp], each head padded to
 *                      head_dim_pad; zero rows up to a whole 256-row tile (read without clamping)
 *   b1  f32  [F1 rounded up to 256], zero padded; F1 = 3*H*head_dim_pad + M (the tile kernels copy whole 256-feature tiles of it
 *                      into LDS: a buffer of exactly F1 floats would be read out of bounds)
 *   qs, ks f32 [head_dim_pad]  QKNorm scales (zero in the padding)
 *   w2  bf16 [D rounded up to 256][K2]   columns = [attention heads (padded) | mlp],  K2 = H*head_dim_pad + M
 *   b2  f32  [D]                                                                         */
typedef struct lsl_block_weights {
    const void *w1;
    const float *b1;
    const float *qs;
    const float *ks;
    const void *w2;
    const float *b2;
} lsl_block_weights;

/* The weight contract of SURVEY.md a2 as device pointers (fp32 unless noted). */
typedef struct lsl_weights {
    const float *x_in_w, *x_in_b;       /* [D,C], [D]    x_in                                   */
    const float *cond_w, *cond_b;       /* [D,C], [D]    cond_to_emb                            */
    const float *mask_emb;              /* [2,D]         mask_to_emb                            */
    const float *time_freqs;            /* [128]         exp(-ln(1e4) k/128) (mmdit.py:103-105)  */
    const float *time_w1, *time_b1;     /* [D,256], [D]  time_in.in_layer                       */
    const float *time_w2, *time_b2;     /* [D,D], [D]    time_in.out_layer                      */
    const float *vec_w1, *vec_b1;       /* [D,V], [D]    vec_in.in_layer  (NULL if V == 0)      */
    const float *vec_w2, *vec_b2;       /* [D,D], [D]    vec_in.out_layer                       */
    const float *mod_w, *mod_b;         /* [(6*depth+2)*D, D], [(6*depth+2)*D]: blocks.i.modulation.lin
                                           stacked in layer order, then adaLN_modulation.1       */
    const float *out_w, *out_b;         /* [C,D], [C]    linear                                 */
    const lsl_block_weights *blocks;    /* HOST array [2*depth]: spatial_0, temporal_0, spatial_1, ... */
} lsl_weights;

/* Arguments of one LatentSIV3.forward(x, t, x_cond, x_cond_mask, y) (latent_si_v31.py:168-188). */
typedef struct lsl_io {
    float *x;              /* device [B,T,L,C]: network input; the samplers update it in place      */
    const float *x_cond;   /* device [B,T,L,C]                                                       */
    const int64_t *mask;   /* device [B,T,L] values 0/1 (x_cond_mask, lightning_base.py:246-247)     */
    const float *y;        /* device [B,V] or NULL                                                   */
    const float *t;        /* device [B] (lsl_forward only)                                          */
    float *out;            /* device [B,T,L,C] (lsl_forward only)                                    */
    int32_t B, T, L;
} lsl_io;

/* One state update of a sampler.  Every sampler the reference builds from
 * Transport.get_drift / get_score (transport.py:158-226) with a fixed time grid is affine in
 * (state, network output, noise) with coefficients that depend only on t:
 *      m = network(x, t);   x <- ax * x + am * m + aw * w
 * ODE Euler (integrators.py:103-120 + torchdiffeq fixed-grid euler), Euler-Maruyama
 * (integrators.py:29-37) and the "Mean"/"Euler"/"Tweedie" last step (transport.py:267-299) all have
 * this form; lam_slide_amd/transport.py derives (ax, am, aw) in float64 from the same formulas. */
typedef struct lsl_step {
    float t;    /* time handed to the network, as fp32 (integrators.py:107-114) */
    float ax, am, aw;
} lsl_step;

/* Extended step record (lsl_sample_ex): x <- ax * x + am * network(x, t) + aw * w + as * saved, where `saved` is a copy of the state
 * taken by an earlier record of the same call.  Enough for every fixed-grid sampler of the reference that is affine per stage, e.g. the
 * stochastic Heun step (integrators.py:39-51) = three records:
 *     noise in, keep a copy :  x <- x + sqrt(2 g dt) w                         (LSL_STEP_NO_NETWORK | LSL_STEP_SAVE)
 *     predictor             :  x <- (1 + dt a(t)) x + dt b(t) net(x, t)
 *     corrector             :  x <- (1/2 + dt a(t')/2) x + dt b(t')/2 net(x, t') + saved / 2,   t' = t + dt
 * (drift(x, t) = a(t) x + b(t) net(x, t); the predictor's dt K1 is x_p - x_hat, so no division is needed). */
#define LSL_STEP_NO_NETWORK 1 /* no network evaluation: x <- ax * x + aw * w + as * saved (am is ignored) */
#define LSL_STEP_SAVE 2       /* after the update, copy the state into the call's saved-state buffer */
typedef struct lsl_step_ex {
    float t;
    float ax, am, aw, as;
    int32_t flags;
    int32_t noise_index; /* slice of `noise` / device-stream step number used for w; ignored when aw == 0 */
    int32_t trace_index; /* slice of `trace` that receives the state after this record, or -1 */
} lsl_step_ex;

int lsl_version(void);
/* Compiler and target the library was built with ("clang <version> gfx950"): the kernels' register budgets and hand-placed waits are checked
 * against ONE compiler's code generation (tests/test_isa_scan.py); bench.py records the string next to its numbers. */
const char *lsl_build_info(void);
const char *lsl_last_error(void);

/* LatentSIV3.__init__ counterpart: validates the shape (-> ValueError in the Python wrapper). */
int lsl_model_create(const lsl_model_desc *desc, lsl_model **out);
/* load_state_dict counterpart; pointers must stay alive while the model is used. */
int lsl_model_set_weights(lsl_model *m, const lsl_weights *w);
void lsl_model_destroy(lsl_model *m);

/* Trajectories processed per pass (cache-residency knob); 0 = library default. */
int lsl_model_set_chunk(lsl_model *m, int32_t trajectories_per_pass);

/* ParallelMLPAttentionV2's attention_mode (mmdit.py:222-229, used at mmdit.py:247 -> attention(), mmdit.py:40-53): 0 = "scaled_dot_product"
 * (the default of a new model, every shipped config), 1 = any other string = attention_linear (mmdit.py:58-72): q softmax over the head
 * channels, k softmax over the positions, out = (q hd^-1/2) (k^T v).  Replaces the reference's constructor keyword; may be changed between
 * calls (cached graphs are dropped).  Models in linear mode always take the general path (lsl_sampler_path == 0). */
int lsl_model_set_attention_mode(lsl_model *m, int32_t mode);

/* Decomposition of a ParallelMLPAttentionV2 sub-block (mmdit.py:240-249) behind the attention.  0 (default): linear1 computes q | k | v | mlp,
 * linear2 and the next LayerNorm are kernels of their own.  1: linear1 computes q | k | v only and ONE row-owning kernel (k_tail) runs the mlp
 * up-projection, GELU, linear2 over [attention | gelu(mlp)], the gated residual update and the next sub-block's LayerNorm + modulate: 8.6
 * instead of 13.2 KB of measured HBM traffic per token and sub-block at hidden 256 / mlp 1024, faster from about 10^5 tokens per pass, slower below
 * (a workgroup streams the whole weight image per 256 tokens).  The two forms are not bit-identical (other summation order in linear2 and in
 * the row statistics; same error against the fp32 reference), so the choice belongs to the MODEL HANDLE - never to the batch: a trajectory's
 * bits stay the same in any batch, shard or pass.  Returns -21 if the model has no instance (hidden 256 with heads * head_dim_pad = 256 and mlp_dim a multiple of 64).
 * Environment LSL_TAIL=1 / 0 sets the default of new handles / disables the form (A/B runs, tests). */
int lsl_model_set_tail(lsl_model *m, int32_t on);
int32_t lsl_model_tail(const lsl_model *m); /* 1 if the handle runs the tail form */

/* LayerNorm + modulate of a sub-block (latent_si_v31.py:50-51,57-58) inside linear1's activation load.  0 (default): a LayerNorm kernel writes the
 * bf16 operand `a`, linear1 reads it.  1: no LayerNorm launch - linear1 reads the fp32 residual stream, normalises and modulates its rows while
 * it turns them into MFMA fragments (1.5 KB of HBM traffic per token and sub-block less at hidden 512, measured; faster from about 10^5 tokens per pass,
 * slower at small launches).  Wherever the token-stationary linear1 runs with its modulation rows in LDS (one shared row, or >= 128 / 256 tokens
 * per trajectory at hidden <= 256 / above; the first sub-block of an evaluation when the embedding kernel can leave the statistics: hidden 256 / 512,
 * <= 32 input channels, normalize = false); elsewhere, and on handles in the tail form, the standalone kernel stays.  One more bf16 rounding of a
 * deviation-sized value than the standalone kernel: results differ at the level of the bf16 operand (same error class against the fp32
 * reference), so the choice belongs to the MODEL HANDLE, never to the batch.  Environment LSL_LN_FUSE=1 / 0: default of new handles / disabled. */
int lsl_model_set_ln_fuse(lsl_model *m, int32_t on);
int32_t lsl_model_ln_fuse(const lsl_model *m);

/* Trajectories the library processes per pass for a call of this size (<= B). */
int32_t lsl_pass_size(const lsl_model *m, int32_t B, int32_t T, int32_t L);

/* Which kernel family lsl_sample uses for trajectories of T x L tokens of this model: 0 = the general path (one launch per kernel per
 * sub-block, any shape), 1 = the trajectory-resident path (csrc/k_resident.hip.h: models of the pedestrian family - hidden 128, 4 heads
 * of 32, mlp 256 - with T*L <= 48 and T, L <= 32: one workgroup per trajectory runs whole groups of state updates in a single launch).  The answer
 * depends on the model and on T, L only, never on the batch, so a trajectory's result does not depend on what it is batched with. */
int32_t lsl_sampler_path(const lsl_model *m, int32_t T, int32_t L);

/* Bytes of caller-provided device scratch needed for a call with these sizes. */
size_t lsl_workspace_bytes(const lsl_model *m, int32_t B, int32_t T, int32_t L);

/* LatentSIV3.forward: io->out = network(io->x, io->t, io->x_cond, io->mask, io->y). */
int lsl_forward(lsl_model *m, const lsl_io *io, void *workspace, size_t workspace_bytes, void *stream);

/* Sampler loop (Sampler.sample_ode / sample_sde inner loops): applies n_steps affine updates to io->x
 * in place.  noise: device [n_noise, B*T*L*C] standard-normal draws, slice s belongs to step s (the
 * reference draws one tensor per Euler-Maruyama step whether or not g(t) is zero, integrators.py:30);
 * steps >= n_noise must have aw == 0.  noise == NULL: steps with aw != 0 draw on the device
 * (Philox4x32-10 keyed by seed, counter = (step, global element index + elem_offset); elem_offset makes
 * sharded runs reproduce the unsharded stream).
 * trace: optional device [n_trace, B*T*L*C]: a record with trace_index k >= 0 writes the state after it to slice k (k < n_trace is
 * checked; -1 = not recorded; anything below -1 is rejected), or NULL (every trace_index is then ignored). */
int lsl_sample_ex(lsl_model *m, const lsl_io *io, const lsl_step_ex *steps, int32_t n_steps,
                  const float *noise, int32_t n_noise, uint64_t seed, uint64_t elem_offset, float *trace, int32_t n_trace,
                  void *workspace, size_t workspace_bytes, void *stream);
/* The same with plain records: record s uses noise slice / stream step s and writes trace slice s (trace: [n_steps, B*T*L*C] or NULL). */
int lsl_sample(lsl_model *m, const lsl_io *io, const lsl_step *steps, int32_t n_steps,
               const float *noise, int32_t n_noise, uint64_t seed, uint64_t elem_offset, float *trace,
               void *workspace, size_t workspace_bytes, void *stream);

/* Initial state of a sampling call: x[0..n) ~ N(0,1) from the same counter stream as the per-step noise (reserved step index
 * 0xFFFFFFFF), element e of the call = global element elem_offset + e.  Stands in for `torch.randn_like(x_cond)`
 * (models/composites/lightning_base.py:231); shard-invariant by construction. */
int lsl_randn(float *x, uint64_t n, uint64_t seed, uint64_t elem_offset, void *stream);

/* Runge-Kutta arithmetic of the adaptive ODE sampler on the device (the reference's default ODE method is torchdiffeq's dopri5,
 * modules/transport/transport.py:486-494 -> integrators.py:67-78; torchdiffeq's rk_common.py does these combinations with torch ops).  fp32;
 * every term is a rounded product added to the rounded running sum in list order (no FMA contraction), reductions in a fixed order: results
 * are deterministic and independent of the launch shape.  n_x, n_k <= 8; `out` may alias a term.
 *   lsl_rk_lincomb:     out[i] = sum_j c[j] x[j][i]                       (stage states, solution, mid-point, dense-output coefficients)
 *   lsl_rk_dense:       out[i] = e + x (d + x (c + x (b + x a)))           (dense output at the fraction x of an accepted step)
 *   lsl_rk_error_ratio: *ratio = sqrt(mean_i (err[i] / (atol + rtol max(|y0[i]|, |y1[i]|)))^2), err = sum_j c[j] k[j]
 *                       (the step controller's one scalar; `ratio` is a device pointer, `scratch` >= LSL_RK_SCRATCH_BYTES device bytes) */
#define LSL_RK_SCRATCH_BYTES 8192
int lsl_rk_lincomb(float *out, const float *const *x, const float *c, int32_t n_x, uint64_t n, void *stream);
int lsl_rk_dense(float *out, const float *a, const float *b, const float *c, const float *d, const float *e, float x, uint64_t n, void *stream);
int lsl_rk_error_ratio(float *ratio, const float *y0, const float *y1, const float *const *k, const float *c, int32_t n_k, float atol, float rtol,
                       uint64_t n, void *scratch, void *stream);

/* Test hooks: run a single kernel of the path on caller buffers (parity tests of intermediates). */
int lsl_debug_block(lsl_model *m, int32_t block_index /* 0..2*depth-1 */, const float *h_in, float *h_out,
                    const float *mods /* [B, (6*depth+2)*D] */, int32_t B, int32_t T, int32_t L,
                    void *workspace, size_t workspace_bytes, void *stream);
/* The same sub-block up to and including attention (LayerNorm + modulate, linear1 with its epilogue, attention), then the two
 * intermediate buffers as the kernels leave them:
 *   qkv_out bf16 [B*T*L][3 * H * head_dim_pad]  q (after QK-norm and RoPE, times head_dim^-1/2 * log2 e) | k (after QK-norm and RoPE) | v,
 *                                               head-major inside each third
 *   z_out   bf16 [B*T*L][H * head_dim_pad + M]  attention output (head-major) | GELU(mlp)
 * (reference taps: mmdit.py:241-248 q_norm / k_norm / apply_rope / attention / gelu). */
int lsl_debug_taps(lsl_model *m, int32_t block_index, const float *h_in, const float *mods, int32_t B, int32_t T, int32_t L,
                   void *qkv_out, void *z_out, void *workspace, size_t workspace_bytes, void *stream);
int lsl_debug_mods(lsl_model *m, const float *t, const float *y, int32_t B, float *vec_out, float *mods_out,
                   void *workspace, size_t workspace_bytes, void *stream);

/* Measurement hooks (bench.py roofline leg): bracket every launch of one kernel class with HIP events on
 * the stream it is launched on.  kernel: 0 linear1 GEMM, 1 linear2 GEMM, 2 attention, 3 LayerNorm+modulate,
 * 4 output head + state update, 5 input embedding, 6 modulation tables; -1 disables.
 * lsl_profile_read synchronises on the recorded events and returns their summed duration. */
int lsl_profile_enable(lsl_model *m, int32_t kernel, int32_t max_launches);
/* Name of the kernel the profiled class launched in the passes since lsl_profile_enable ("" if none): the label of a timing comes from the
 * library's own dispatch, not from a copy of its rule. */
const char *lsl_profile_kernel_name(const lsl_model *m);
int lsl_profile_read(lsl_model *m, double *total_ms, int32_t *launches);

/* ------------------------------------------------------------------------------------------------
 * Frozen stage-1 decode of the sampled latents: the step right after the sampler and the
 * second half of the parity metric ("decoded coordinates").  Stands in for
 *   first_stage.decode(latents, entities) = Decoder(post_quant(latents), entities)
 *   (models/composites/lightning_base.py:28-31,42-44; models/components/decoder.py:12-102;
 *    blocks: modules/torch_modules.py:104-264; entity table: modules/entity_embeddings.py:7-33).
 * fp32 throughout.  One block = PreNorm(Attention) + residual, PreNorm(FeedForward(dim)) + residual. */
typedef struct lsl_dec_block {
    const float *ln_w, *ln_b;        /* attn.norm                                  [dim]                  */
    const float *lnc_w, *lnc_b;      /* attn.norm_context (cross-attention blocks) [context_dim], else NULL */
    const float *w_q;                /* self: attn.fn.to_qkv.weight [3*inner, dim]; cross: attn.fn.to_q.weight [inner, dim] */
    const float *w_kv;               /* cross: attn.fn.to_kv.weight [2*inner, context_dim]; self: NULL    */
    const float *w_out, *b_out;      /* attn.fn.to_out [dim, inner], [dim]                                */
    const float *q_scale, *k_scale;  /* attn.fn.norm.{query,key}_norm.scale [dim_head], NULL without qk_norm */
    const float *ff_ln_w, *ff_ln_b;  /* ff.norm                                                           */
    const float *ff_w1, *ff_b1;      /* ff.fn.net.0.0 [dim, dim]                                          */
    const float *ff_w2, *ff_b2;      /* ff.fn.net.1   [dim, dim]                                          */
} lsl_dec_block;

typedef struct lsl_decoder_desc {    /* Decoder.__init__ arguments (decoder.py:14-29) + post_quant input width */
    int32_t in_dim;                  /* C of the sampled latents (post_quant = LayerNorm(C, no affine) + Linear(C, dim_latent)) */
    int32_t dim_latent, dim_query, dim_emb, n_entities;
    int32_t heads_latent, dim_head_latent, heads_cross, dim_head_cross;
    int32_t num_block_attn, num_block_cross;
    int32_t act;                     /* 1 = erf GELU (src.modules.torch_modules.GELU), 2 = nn.GELU(approximate="tanh") */
    int32_t out_dim;                 /* width of the decoded output head (3 for "pos")                    */
    int32_t num_split;               /* DecoderQuerySplitter (decoder.py:313-411, peptide): every latent becomes num_split context tokens
                                        for the output block; 0 or 1 = plain Decoder                           */
} lsl_decoder_desc;

typedef struct lsl_decoder_weights {
    const float *pq_w, *pq_b;        /* post_quant.1 [dim_latent, C], [dim_latent]                        */
    const float *table;              /* decoder.entity_embedding.embedding.weight [n_entities, dim_emb], rows ALREADY clipped to
                                        max_norm (nn.Embedding(max_norm=1) renormalises looked-up rows at forward time)      */
    const float *qm_w, *qm_b;        /* decoder.query_mlp.1 [dim_query, dim_emb]                          */
    const lsl_dec_block *self_blocks;   /* HOST array [num_block_attn]  decoder.self_attn_blocks.i       */
    const lsl_dec_block *cross_blocks;  /* HOST array [num_block_cross] decoder.cross_attn_blocks.i      */
    lsl_dec_block out_block;            /* decoder.output_block (queries attend to the latents)          */
    const float *ext_w, *ext_b;      /* decoder.extender.1 (1x1 Conv1d) as [num_split * dim_latent, dim_latent] with rows reordered to
                                        (split, feature): row n * dim_latent + d = conv channel d * num_split + n; NULL without split */
    const float *head_w1, *head_b1;  /* decoder.output_layers.<name>.0 [dim_query, dim_query]             */
    const float *head_w2, *head_b2;  /* decoder.output_layers.<name>.2 [out_dim, dim_query]               */
} lsl_decoder_weights;

typedef struct lsl_decoder lsl_decoder;
int lsl_decoder_create(const lsl_decoder_desc *desc, const lsl_decoder_weights *w, lsl_decoder **out);
void lsl_decoder_destroy(lsl_decoder *d);
size_t lsl_decode_workspace_bytes(const lsl_decoder *d, int32_t frames, int32_t L, int32_t A);
/* z: device [frames, L, C] latents; entities: device [frames, A] int64; out: device [frames, A, out_dim]. */
int lsl_decode(lsl_decoder *d, const float *z, const int64_t *entities, int32_t frames, int32_t L, int32_t A, float *out,
               void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Frozen stage-1 encode, the step before setup_conditioning: stands in for
 *   quant(Encoder(x, entities, mask))      (models/components/encoder.py:34-41,96-103; lightning_base.py:22-25,37-40)
 * where x is the output of the dataset-specific prepare_inputs (first_stage/<dataset>.py), which stays the caller's. */
typedef struct lsl_encoder_desc {    /* Encoder.__init__ arguments (encoder.py:46-61)                       */
    int32_t dim_input, dim_emb, n_entities, dim_latent, num_latents;
    int32_t heads_cross, dim_head_cross, heads_latent, dim_head_latent;
    int32_t num_block_cross, num_block_attn;
    int32_t act;                     /* 1 = erf GELU, 2 = tanh GELU                                       */
} lsl_encoder_desc;

typedef struct lsl_encoder_weights {
    const float *table;              /* encoder.entity_embedding.embedding.weight, rows already clipped to max_norm */
    const float *mlp_w1, *mlp_b1;    /* encoder.mlp.0 [dim_latent, dim_input + dim_emb]                    */
    const float *mlp_w2, *mlp_b2;    /* encoder.mlp.2 [dim_input + dim_emb, dim_latent]                    */
    const float *latents;            /* encoder.latents [num_latents, dim_latent]                          */
    const lsl_dec_block *cross_blocks;  /* HOST array [num_block_cross] encoder.cross_attn_blocks.i (latents attend to the context) */
    const lsl_dec_block *self_blocks;   /* HOST array [num_block_attn]  encoder.blocks_attn.i            */
    const float *quant_w, *quant_b;  /* quant.0 [dim_latent, dim_latent]; quant.1 = LayerNorm without affine */
} lsl_encoder_weights;

typedef struct lsl_encoder lsl_encoder;
int lsl_encoder_create(const lsl_encoder_desc *desc, const lsl_encoder_weights *w, lsl_encoder **out);
void lsl_encoder_destroy(lsl_encoder *e);
size_t lsl_encode_workspace_bytes(const lsl_encoder *e, int32_t frames, int32_t A);
/* x: device [frames, A, dim_input]; entities: device [frames, A] int64; mask: device [frames, A] bytes, non-zero = real entity,
 * or NULL (all real); out: device [frames, num_latents, dim_latent]. */
int lsl_encode(lsl_encoder *e, const float *x, const int64_t *entities, const unsigned char *mask, int32_t frames, int32_t A, float *out,
               void *workspace, size_t workspace_bytes, void *stream);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* LSL_API_H */
