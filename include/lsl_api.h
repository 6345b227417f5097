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
 *     Nothing aborts or throws across this boundary.
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

#define LSL_VERSION 1

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
 *   b1  f32  [F1]      F1 = 3*H*head_dim_pad + M
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

int lsl_version(void);
const char *lsl_last_error(void);

/* LatentSIV3.__init__ counterpart: validates the shape (-> ValueError in the Python wrapper). */
int lsl_model_create(const lsl_model_desc *desc, lsl_model **out);
/* load_state_dict counterpart; pointers must stay alive while the model is used. */
int lsl_model_set_weights(lsl_model *m, const lsl_weights *w);
void lsl_model_destroy(lsl_model *m);

/* Trajectories processed per pass (cache-residency knob); 0 = library default. */
int lsl_model_set_chunk(lsl_model *m, int32_t trajectories_per_pass);

/* Trajectories the library processes per pass for a call of this size (<= B). */
int32_t lsl_pass_size(const lsl_model *m, int32_t B, int32_t T, int32_t L);

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
 * trace: optional device [n_steps, B*T*L*C] receiving the state after every step, or NULL. */
int lsl_sample(lsl_model *m, const lsl_io *io, const lsl_step *steps, int32_t n_steps,
               const float *noise, int32_t n_noise, uint64_t seed, uint64_t elem_offset, float *trace,
               void *workspace, size_t workspace_bytes, void *stream);

/* Test hooks: run a single kernel of the path on caller buffers (parity tests of intermediates). */
int lsl_debug_block(lsl_model *m, int32_t block_index /* 0..2*depth-1 */, const float *h_in, float *h_out,
                    const float *mods /* [B, (6*depth+2)*D] */, int32_t B, int32_t T, int32_t L,
                    void *workspace, size_t workspace_bytes, void *stream);
int lsl_debug_mods(lsl_model *m, const float *t, const float *y, int32_t B, float *vec_out, float *mods_out,
                   void *workspace, size_t workspace_bytes, void *stream);

/* Measurement hooks (bench.py roofline leg): bracket every launch of one kernel class with HIP events on
 * the stream it is launched on.  kernel: 0 linear1 GEMM, 1 linear2 GEMM, 2 attention, 3 LayerNorm+modulate,
 * 4 output head + state update, 5 input embedding, 6 modulation tables; -1 disables.
 * lsl_profile_read synchronises on the recorded events and returns their summed duration. */
int lsl_profile_enable(lsl_model *m, int32_t kernel, int32_t max_launches);
int lsl_profile_read(lsl_model *m, double *total_ms, int32_t *launches);

#ifdef __cplusplus
}
#endif
#endif /* LSL_API_H */
