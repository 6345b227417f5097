// Host side of liblamslide_hip.so: the C ABI of include/lsl_api.h.  Enqueues the kernel sequence of one
// network evaluation (latent_si_v31.py:168-188) and of the sampler loops (integrators.py:67-78,103-120)
// on the caller's stream.  No allocation, no synchronisation, no host<->device copies.
#include "../../include/lsl_api.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <atomic>
#include <new>

#include "k_attn.hip.h"
#include "k_gemm.hip.h"
#include "k_lin1.hip.h"
#include "k_lin2.hip.h"
#include "k_small.hip.h"
#include "k_resident.hip.h"
#ifdef LSL_EXPERIMENTS  // measured-and-rejected GEMM structures, built only by tools/build_experiments.sh (never in the product library)
#include "k_gemm_pp.hip.h"        // tools/experiments/ (on the include path of tools/build_experiments.sh only)
#include "k_gemm_drain.hip.h"
#endif

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define LSL_CHECK_LAUNCH(name)                                                     \
    do {                                                                           \
        hipError_t e_ = hipGetLastError();                                         \
        if (e_ != hipSuccess) return fail(-10, "%s: %s", name, hipGetErrorString(e_)); \
    } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

struct Profiler {
    int kernel = -1, cap = 0, used = 0;
    std::vector<hipEvent_t> ev;  // 2 per launch
    void begin(int k, hipStream_t st) {
        if (k == kernel && used < cap) hipEventRecord(ev[2 * used], st);
    }
    void end(int k, hipStream_t st) {
        if (k == kernel && used < cap) hipEventRecord(ev[2 * used++ + 1], st);
    }
    void clear() {
        for (auto e : ev) hipEventDestroy(e);
        ev.clear();
        kernel = -1; cap = used = 0;
    }
};

struct lsl_model {
    Profiler prof;
    lsl_model_desc d;
    lsl_weights w;
    std::vector<lsl_block_weights> blocks;
    bool has_weights = false;
    int chunk = 0;
    int HHD, F1, K2, MODW;
    // second lane of lsl_sample (LSL_LANES=2): passes alternate between the caller's stream and this one, so that the memory-bound
    // kernels of one pass can share the chip with the GEMMs of the other (created on first use, fork / join by events)
    hipStream_t lane_stream = nullptr;
    hipEvent_t lane_fork = nullptr, lane_join = nullptr;
    // hipGraph cache of lsl_sample: a call whose arguments (pointers, sizes, step table) repeat is captured once and replayed; the
    // small-batch configs are launch-bound (~700 launches of a few microseconds per sampling call)
    struct GraphEntry {
        std::vector<unsigned char> key;
        hipGraphExec_t exec = nullptr;
        unsigned long long last_use = 0;
    };
    std::vector<GraphEntry> graphs;
    std::vector<std::vector<unsigned char>> seen;  // argument sets that ran eagerly once (capture happens on their second appearance)
    std::vector<std::vector<unsigned char>> uncapturable;  // argument sets whose capture failed: never tried again
    bool graph_stream_failed = false;                      // the internal capture stream could not be created: no further attempts
    unsigned long long graph_clock = 0;
    hipStream_t graph_stream = nullptr;  // capture happens on this internal stream (the caller's may be the legacy default stream, which
                                         // cannot be captured); the instantiated graph is launched on the caller's stream
};

namespace {

struct Workspace {
    float2 *rope_l, *rope_t;
    float4 *rope_qk;        // [2 * depth blocks][q, k][max(T, L) positions][head_dim_pad / 2]: RoPE x QK-norm scales (k_rope_scaled)
    size_t rope_qk_stride;  // float4 elements between consecutive (block, q|k) tables
    float *cond_emb, *h, *yemb, *tfeat, *hid, *vec, *mods;
    float *saved;  // [n][C] state kept by an LSL_STEP_SAVE record of lsl_sample_ex (Heun's x_hat)
    // models without class conditioning: the modulation tables of a GROUP of sampler records are computed before the records run
    // (one row per record: the time is shared by the batch), instead of four tiny dependent launches in front of every evaluation
    float *tf_all, *hid_all, *vec_all, *mods_all;
    int mods_group;  // records per group (0: class-conditioned model, tables per evaluation)
    u16 *a, *qkv, *z;
    float *kmax2;  // [2 * depth]: bound of |k|^2 per attention block (k_rope_scaled), for k_attention_stream's softmax shift
    u16 *w2p;  // linear2 weights of every sub-block in the fragment order of k_linear2_ws (k_lin2_pack, once per call), or NULL
    size_t bytes;
};

int env_int(const char *name, int dflt);
int tune_int(const char *name, int dflt);
bool linear2_ws_shape_ok(int D, int K2);

// Scratch layout for a pass over `bc` trajectories.
Workspace carve(const lsl_model *m, char *base, int bc, int T, int L) {
    const lsl_model_desc &d = m->d;
    const size_t n = (size_t)bc * T * L, D = d.hidden;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char *p = base ? base + off : nullptr;
        off += align_up(bytes, 256);
        return p;
    };
    Workspace ws;
    ws.rope_l = (float2 *)take((size_t)L * (d.head_dim_pad / 2) * sizeof(float2));
    ws.rope_t = (float2 *)take((size_t)T * (d.head_dim_pad / 2) * sizeof(float2));
    ws.rope_qk_stride = (size_t)std::max(T, L) * (d.head_dim_pad / 2);
    ws.rope_qk = (float4 *)take((size_t)4 * d.depth * ws.rope_qk_stride * sizeof(float4));
    ws.kmax2 = (float *)take((size_t)2 * d.depth * sizeof(float));
    ws.cond_emb = (float *)take(n * D * 4);
    ws.h = (float *)take(n * D * 4);
    ws.yemb = (float *)take((size_t)bc * D * 4);
    ws.tfeat = (float *)take((size_t)bc * 256 * 4);
    ws.hid = (float *)take((size_t)bc * D * 4);
    ws.vec = (float *)take((size_t)bc * D * 4);
    ws.mods = (float *)take((size_t)bc * m->MODW * 4);
    ws.saved = (float *)take(n * d.in_dim * 4);
    ws.mods_group = d.vec_in_dim > 0 ? 0 : (int)std::min<size_t>(1024, std::max<size_t>(1, ((size_t)16 << 20) / ((size_t)m->MODW * 4)));
    ws.tf_all = (float *)take((size_t)ws.mods_group * 256 * 4);
    ws.hid_all = (float *)take((size_t)ws.mods_group * D * 4);
    ws.vec_all = (float *)take((size_t)ws.mods_group * D * 4);
    ws.mods_all = (float *)take((size_t)ws.mods_group * m->MODW * 4);
    const size_t n_pad = align_up(n, 256);  // GEMM operand rows: whole 256-token tiles are read without clamping
    ws.a = (u16 *)take(n_pad * D * 2);
    ws.qkv = (u16 *)take(n_pad * 3 * m->HHD * 2);  // (padded like a / z: the token-stationary linear1 stores whole 256-token tiles)
    ws.z = (u16 *)take(n_pad * m->K2 * 2);
    ws.w2p = linear2_ws_shape_ok((int)D, m->K2) ? (u16 *)take((size_t)2 * d.depth * D * m->K2 * 2) : nullptr;
    ws.bytes = off;
    return ws;
}

// Two lanes (opt-in, LSL_LANES=2): the passes of a large batch alternate between the caller's stream and a second one, so that one
// half-batch's memory-bound kernels (LayerNorm, attention, output head: a quarter of the step) and the tails of its persistent GEMM launches
// share the chip with the other half's kernels: +1.4 % on the cfg-2 bench (profiles/r03_experiments.txt).  Only when each half still
// makes full-size launches (>= 64 Ki tokens); results do not depend on it (a trajectory's bits are independent of the batch it is sampled
// in).  Off by default: co-running kernels stretch each other's durations (k_linear1_ts 0.34 -> 0.46 ms in a rocprofv3 kernel trace), so a
// profile of the default configuration would no longer show per-kernel times that can be compared with a roofline.
int n_lanes() {
    static const int l = env_int("LSL_LANES", 1);
    return l >= 2 ? 2 : 1;
}
int lanes_for(int B, int T, int L) { return n_lanes() == 2 && B >= 2 && (size_t)B * T * L >= (size_t)131072 ? 2 : 1; }

int default_chunk(const lsl_model *m, int B, int T, int L) {
    if (m->chunk > 0) return m->chunk < B ? m->chunk : B;
    if (const char *e = getenv("LSL_CHUNK_TRAJ")) {
        const int v = atoi(e);
        if (v > 0) return v < B ? v : B;
    }
    // Measured on MI355X (profiles/r01_chunk_sweep.txt): the kernels are not helped by keeping a pass inside the
    // 256 MiB Infinity Cache; larger passes are faster (fewer, better filled launches).  Cap a pass at 256 Ki tokens so the
    // workspace stays at a few GiB (of 288).
    size_t c = (size_t)262144 / ((size_t)T * L ? (size_t)T * L : 1);
    if (c < 1) c = 1;
    if (c > (size_t)B) c = B;
    {  // equal passes: 1024 trajectories of 640 tokens are 342 + 342 + 340, not 409 + 409 + 206 (the short pass fills the chip worse)
        const size_t passes = ((size_t)B + c - 1) / c;
        c = ((size_t)B + passes - 1) / passes;
    }
    if (lanes_for(B, T, L) == 2 && c > (size_t)(B + 1) / 2) c = (B + 1) / 2;  // at least one pass per lane
    return (int)c;
}

template <typename K>
void allow_lds(K kernel, size_t bytes) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
// The dynamic-LDS attribute is a property of (kernel, device): one flag per device ordinal and per call site.
struct DevOnce {
    std::atomic<unsigned long long> bits{0};
    bool first() {
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned long long bit = 1ull << (dev & 63);
        if (bits.load(std::memory_order_relaxed) & bit) return false;
        bits.fetch_or(bit, std::memory_order_relaxed);
        return true;
    }
};
#define LSL_ALLOW_LDS(kern, bytes)                  \
    do {                                            \
        static DevOnce once_;                       \
        if (once_.first()) allow_lds(kern, bytes);  \
    } while (0)

// Calls may arrive with a current device other than the stream's (a model used on a second GPU of the process): launches,
// attributes and the CU count must follow the STREAM's device.
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(hipStream_t st) {
        int cur = 0, want = 0;
        if (hipGetDevice(&cur) != hipSuccess) return;
        if (hipStreamGetDevice(st, &want) != hipSuccess) { (void)hipGetLastError(); return; }
        if (want != cur && hipSetDevice(want) == hipSuccess) prev = cur;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// ---- launch helpers -------------------------------------------------------------------------------

int device_cus();
int env_int(const char *name, int dflt);

template <int NE, int VEC>
void launch_ln_mod_t(u16 *a, const float *h, const float *shift, const float *scale, int stride, int n, int tpt, hipStream_t st) {
    if constexpr (NE % 4 == 0) {
        static const int persist = tune_int("LSL_LN_PERSIST", 16);  // workgroups per CU of the persistent form; 0 = one wave per token
        if (persist > 0) {
            const int grid = std::min((n + 3) / 4, device_cus() * persist);
            static const int nt = (tune_int("LSL_NT", 3) >> 3) & 1;
            hipLaunchKernelGGL((k_ln_modulate_v4<NE>), dim3(grid), dim3(256), 0, st, a, h, shift, scale, stride, n, tpt, nt);
            return;
        }
    }
    hipLaunchKernelGGL((k_ln_modulate<NE, VEC>), dim3((n + 3) / 4), dim3(256), 0, st, a, h, shift, scale, stride, n, tpt);
}
template <int NE, int VEC>
void launch_ln_inplace_t(float *h, int n, float eps, hipStream_t st) {
    hipLaunchKernelGGL((k_ln_inplace<NE, VEC>), dim3((n + 3) / 4), dim3(256), 0, st, h, n, eps);
}
int device_cus();
int env_int(const char *name, int dflt);

template <int NE, int VEC>
void launch_head_mfma(float *x, float *out, const float *h, const float *shift, const float *scale, int stride, const float *Wo,
                      const float *bo, int n, int C, int tpt, int do_step, float ax, float am, float aw, const float *noise,
                      unsigned long long seed, unsigned step, unsigned long long eo, float *trace, float as, const float *saved, float *save_out,
                      hipStream_t st) {
    auto kern = k_head_step_mfma<NE, VEC>;
    const size_t lds = head_mfma_lds_bytes<NE>(C <= 32);
    LSL_ALLOW_LDS(kern, head_mfma_lds_bytes<NE>(false));
    // two workgroups per CU where the LDS image allows it (any hidden size with <= 32 channels): one workgroup's LayerNorm / weight-load latencies under the other's MFMAs
    static const int per_cu = tune_int("LSL_HEAD_PER_CU", 2);
    const int wgs = device_cus() * (per_cu >= 2 && 2 * lds <= (size_t)160 * 1024 ? 2 : 1);
    hipLaunchKernelGGL(kern, dim3(std::min((n + HEAD_TOK - 1) / HEAD_TOK, wgs)), dim3(256), lds, st, x, out, h, shift, scale, stride,
                       Wo, bo, n, C, tpt, do_step, ax, am, aw, noise, seed, step, eo, trace, as, saved, save_out);
}

template <int NE, int VEC>
void launch_head_t(float *x, float *out, const float *h, const float *shift, const float *scale, int stride, const float *Wo,
                   const float *bo, int n, int C, int tpt, int do_step, float ax, float am, float aw, const float *noise,
                   unsigned long long seed, unsigned step, unsigned long long eo, float *trace, float as, const float *saved, float *save_out,
                   hipStream_t st) {
#ifdef LSL_EXPERIMENTS
    static const int mfma = tune_int("LSL_HEAD_MFMA", 1);  // 0: the scalar-FMA kernel (A/B measurements)
    if (!mfma) {
        auto kern = k_head_step<NE, VEC>;
        constexpr size_t lds = head_lds_bytes<NE>();
        LSL_ALLOW_LDS(kern, lds);
        hipLaunchKernelGGL(kern, dim3(std::min((n + HEAD_TOK - 1) / HEAD_TOK, 256)), dim3(256), lds, st, x, out, h, shift, scale, stride, Wo, bo, n,
                           C, tpt, do_step, ax, am, aw, noise, seed, step, eo, trace, as, saved, save_out);
        return;
    }
#endif
    launch_head_mfma<NE, VEC>(x, out, h, shift, scale, stride, Wo, bo, n, C, tpt, do_step, ax, am, aw, noise, seed, step, eo, trace, as, saved, save_out, st);
}

#define DISPATCH_D(D, FN, ...)                          \
    switch ((D) / 64) {                                 \
        case 1: FN<1, 1>(__VA_ARGS__); break;           \
        case 2: FN<2, 2>(__VA_ARGS__); break;           \
        case 3: FN<3, 1>(__VA_ARGS__); break;           \
        case 4: FN<4, 2>(__VA_ARGS__); break;           \
        case 5: FN<5, 1>(__VA_ARGS__); break;           \
        case 6: FN<6, 2>(__VA_ARGS__); break;           \
        case 7: FN<7, 1>(__VA_ARGS__); break;           \
        default: FN<8, 2>(__VA_ARGS__); break;          \
    }

int device_cus();

template <int MODE>
int launch_embed(float *out, const float *in, const float *W, const float *b, const float *b2, const float *me,
                 const int64_t *mask, const float *base, int n, int C, int D, hipStream_t st) {
    if (C > 32 && C % 8 == 0 && C <= 128 && D % 32 == 0) {  // wide inputs: fp32 MFMA form (k_embed_mfma)
        const int tpw = 3, ngrp = (D / 32 + tpw - 1) / tpw;
        const long units = (long)((n + 31) / 32) * ngrp;
        const dim3 g((unsigned)((units + 3) / 4));
        switch (C / 8) {
#define LSL_EMB_CASE(CK) case CK: hipLaunchKernelGGL((k_embed_mfma<MODE, CK>), g, dim3(256), 0, st, out, in, W, b, b2, me, mask, base, n, C, D, tpw); return 0;
            LSL_EMB_CASE(5) LSL_EMB_CASE(6) LSL_EMB_CASE(7) LSL_EMB_CASE(8) LSL_EMB_CASE(9) LSL_EMB_CASE(10) LSL_EMB_CASE(11) LSL_EMB_CASE(12)
            LSL_EMB_CASE(13) LSL_EMB_CASE(14) LSL_EMB_CASE(15) LSL_EMB_CASE(16)
#undef LSL_EMB_CASE
        }
    }
    // persistent workgroups (weights fetched once each) over tiles of 64 tokens
    int tok = EMB_TOK;
    // 32 inputs, hidden <= 512 (every shipped narrow-input model): the weight rows reach the registers through LDS (k_embed), which makes
    // a workgroup's prologue cheap enough for 32- or 16-token tiles when the launch has fewer than two 64-token tiles per CU
    static const int stage = tune_int("LSL_EMBED_LDS", 1);
    const bool w_lds = stage && C == 32 && D % 4 == 0 && D <= 512;
    if (w_lds)
        while (tok > 16 && (n + tok - 1) / tok < 2 * device_cus()) tok /= 2;
    const dim3 grid(std::min((n + tok - 1) / tok, 2 * device_cus())), blk(256);
    if (C <= 32) {
        const size_t lds = w_lds ? embed_w_lds_bytes<32, 4>(D) : 0;
        auto kern = k_embed<32, MODE, 4>;
        LSL_ALLOW_LDS(kern, (embed_w_lds_bytes<32, 4>(512)));
        hipLaunchKernelGGL(kern, grid, blk, lds, st, out, in, W, b, b2, me, mask, base, n, C, D, w_lds ? 1 : 0, tok);
    } else if (C <= 64) hipLaunchKernelGGL((k_embed<64, MODE, 2>), grid, blk, 0, st, out, in, W, b, b2, me, mask, base, n, C, D, 0, tok);
    else if (C <= 96) hipLaunchKernelGGL((k_embed<96, MODE, 2>), grid, blk, 0, st, out, in, W, b, b2, me, mask, base, n, C, D, 0, tok);
    else hipLaunchKernelGGL((k_embed<128, MODE, 1>), grid, blk, 0, st, out, in, W, b, b2, me, mask, base, n, C, D, 0, tok);
    return 0;
}

int env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}
// Kernel-selection / timing knobs (LSL_GEMM*, LSL_NT, LSL_STAGGER, LSL_PROBE, ...): read from the environment only in
// -DLSL_EXPERIMENTS builds; the product library always runs its measured defaults.
int tune_int(const char *name, int dflt) {
#ifdef LSL_EXPERIMENTS
    return env_int(name, dflt);
#else
    (void)name;
    return dflt;
#endif
}

int device_cus() {  // of the current device (entry points switch to the stream's device first)
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    int n = cache[dev & 63].load(std::memory_order_relaxed);
    if (n > 0) return n;
    n = 256;
    (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    if (n <= 0) n = 256;
    cache[dev & 63].store(n, std::memory_order_relaxed);
    return n;
}

template <int BF, int BT, int NWF, int NWT, int BK, int NS, bool PERSIST, class Epi>
void launch_gemm_glds(const GemmArgs &g, const Epi &epi, hipStream_t st) {
    auto kern = k_gemm_glds<BF, BT, NWF, NWT, BK, NS, PERSIST, Epi>;
    // + the bias vector of the whole GEMM, kept in LDS by epilogues that start the accumulators from it (k_gemm.hip.h)
    const size_t lds = GemmCfg<BF, BT, NWF, NWT, BK, NS, PERSIST, Epi>::lds_bytes + (Epi::lds_bias ? (size_t)((g.F + BF - 1) / BF) * BF * 4 : 0);
    LSL_ALLOW_LDS(kern, (size_t)163840);
    const int ntt = (g.N + BT - 1) / BT, tiles = ntt * ((g.F + BF - 1) / BF);
    int grid = tiles;
    GemmArgs ga = g;
    ga.rows = 0;
    if (PERSIST) {  // as many workgroups as fit at once (LDS-limited), a multiple of 8 so the XCD mapping stays regular
        const int per_cu = (int)(163840 / lds) < 1 ? 1 : (int)(163840 / lds);
        grid = device_cus() * per_cu;
        grid -= grid % 8;
        if (grid > tiles) grid = tiles;
        // row-owner walk (the epilogue finishes whole token rows: fused LayerNorm of linear2) only when there are at least as many
        // token tiles as workgroups; smaller launches keep the flat list, which spreads the feature tiles over more CUs
        if (Epi::row_owner && g.rows && ntt >= grid) ga.rows = 1;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NWF * NWT * 64), lds, st, ga, epi);
}
// whether launch_gemm_glds would take the row-owner walk for this launch (the caller then lets the epilogue write the next LayerNorm)
template <int BF, int BT, int NWF, int NWT, int BK, int NS, class Epi>
bool gemm_rows_walk(int F, int N) {
    const size_t lds = GemmCfg<BF, BT, NWF, NWT, BK, NS, true, Epi>::lds_bytes;
    const int per_cu = (int)(163840 / lds) < 1 ? 1 : (int)(163840 / lds);
    int grid = device_cus() * per_cu;
    grid -= grid % 8;
    const int ntt = (N + BT - 1) / BT, tiles = ntt * ((F + BF - 1) / BF);
    if (grid > tiles) grid = tiles;
    return ntt >= grid;
}

#ifdef LSL_EXPERIMENTS
template <int BK, int NS, int NB, class Epi>
void launch_gemm_pp_t(const GemmArgs &g, const Epi &epi, hipStream_t st) {
    auto kern = k_gemm_pp<BK, NS, NB, Epi>;
    constexpr size_t lds = GemmPPCfg<BK, NS, Epi>::lds_bytes;
    LSL_ALLOW_LDS(kern, lds);
    const int tiles = ((g.N + 255) / 256) * ((g.F + 127) / 128);
    int grid = device_cus();
    grid -= grid % 8;
    if (grid > tiles) grid = tiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, g, epi);
}

// epilogue of tile i inside the main loop of tile i+1 (k_gemm_drain.hip.h); false when the shape is outside what it covers
template <class Epi>
bool launch_gemm_drain(const GemmArgs &g, const Epi &epi, hipStream_t st) {
    if (g.F % 256 != 0 || g.N % 128 != 0 || g.K % 64 != 0 || g.K / 64 < 2) return false;
    auto kern = k_gemm_drain<Epi>;
    constexpr size_t lds = GemmDrainCfg<Epi>::lds_bytes;
    LSL_ALLOW_LDS(kern, lds);
    const int tiles = (g.N / 128) * (g.F / 256);
    int grid = device_cus();
    grid -= grid % 8;
    if (grid > tiles) grid = tiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, g, epi);
    return true;
}

// ping-pong halves (k_gemm_pp.hip.h); false when the shape is outside what the schedule covers
template <int BK, int NS, class Epi>
bool launch_gemm_pp(const GemmArgs &g, const Epi &epi, hipStream_t st) {
    if (g.K % BK != 0 || g.F % 32 != 0) return false;
    const int E = g.K / BK - (NS - 1);  // intervals that carry epilogue pieces
    if (E < 1 || E > 64) return false;
    if (E <= 16) launch_gemm_pp_t<BK, NS, 1>(g, epi, st);
    else if (E <= 32) launch_gemm_pp_t<BK, NS, 2>(g, epi, st);
    else launch_gemm_pp_t<BK, NS, 4>(g, epi, st);
    return true;
}

#endif  // LSL_EXPERIMENTS

// linear1 on the token-stationary kernel (k_lin1.hip.h): hidden sizes 128 / 256 / 384 / 512, sections (q | k | v | mlp) on multiples of 64
// features.  Same bits as the tile kernels below (tools/lin1_harness.hip), so the choice between them may depend on the launch size.
template <int HDP, int K>
void launch_linear1_ts_t(const Lin1Args &a, hipStream_t st) {
    auto kern = k_linear1_ts<HDP, K>;
    LSL_ALLOW_LDS(kern, (size_t)163840);
    const int ntile = (a.N + 255) / 256, nb = a.F / 32;
    const long units = (long)ntile * nb;
    int grid = (int)std::min<long>(device_cus(), units / 2);
    Lin1Args b = a;
    // fewer tiles than workgroups: whole workgroups per tile, one segment each (k_lin1.hip.h "Work split"); same bits either way
    static const int align = tune_int("LSL_LIN1_ALIGN", 1);
    const int wpt = std::min(device_cus() / ntile, nb / 2);
    b.wpt = align && wpt >= 2 ? wpt : 0;
    if (b.wpt) grid = b.wpt * ntile;
    const size_t lds = Lin1Cfg<HDP, K>::lds_bytes(a.F);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, b);
}
bool linear1_ts_ok(int hdp, int D, int F1, int HHD, int N) {
    static const int on = tune_int("LSL_LIN1_TS", 1);
    if (!on || (hdp != 16 && hdp != 32) || (D != 128 && D != 256 && D != 384 && D != 512) || F1 % 64 != 0 || HHD % 64 != 0 || N < 1) return false;
    return (size_t)(D <= 256 ? 4 : 3) * 32 * (2 * D + 16) + 8 * 4096 + (size_t)F1 * 4 <= (size_t)163840;  // weight ring (Lin1Cfg::NS slots) + staging + bias vector (Lin1Cfg::lds_bytes)
}
void launch_linear1_ts(int hdp, int D, const Lin1Args &a, hipStream_t st) {
    switch ((hdp == 32 ? 0 : 4) + D / 128 - 1) {
        case 0: return launch_linear1_ts_t<32, 128>(a, st);
        case 1: return launch_linear1_ts_t<32, 256>(a, st);
        case 2: return launch_linear1_ts_t<32, 384>(a, st);
        case 3: return launch_linear1_ts_t<32, 512>(a, st);
        case 4: return launch_linear1_ts_t<16, 128>(a, st);
        case 5: return launch_linear1_ts_t<16, 256>(a, st);
        case 6: return launch_linear1_ts_t<16, 384>(a, st);
        default: return launch_linear1_ts_t<16, 512>(a, st);
    }
}

// linear2 + gated residual update on the weight-stationary kernel (k_lin2.hip.h): F a multiple of 128, K2 one of the instantiated widths
// (K2 / 8 stationary registers per wave: 2 048, peptide, does not fit).  Same bits as the tile kernels (tools/lin2_harness.hip), so the
// choice may depend on the launch.  LSL_LIN2_WS=0 (read in the product too: the GPU suite compares the two paths bit for bit) turns it off.
bool linear2_ws_shape_ok(int D, int K2) {
    static const int on = env_int("LSL_LIN2_WS", 1);
    return on && D % 128 == 0 && D <= 512 && (K2 == 1536 || K2 == 1280 || K2 == 768 || K2 == 384);
}
template <int K, int NCH, int NS>
bool launch_linear2_ws_t(Lin2Args a, int shared, hipStream_t st) {
    using C = Lin2Cfg<K, NCH, NS, true>;
    auto kern = k_linear2_ws<K, NCH, NS, true>;
    // grid = 8 x slices x rpx workgroups, at most one per CU; fewer token ranges than 32-token blocks
    const int slices = a.F / 128, cus = device_cus(), NBLK = (a.N + 31) / 32;
    int rpx = std::max(1, cus / (8 * slices));
    while (rpx > 1 && 8 * rpx > NBLK) --rpx;
    const int ranges = 8 * rpx, max_blocks = (NBLK + ranges - 1) / ranges + 1;
    const int gate_rows = shared ? 1 : (max_blocks * 32 + a.tpt - 1) / a.tpt + 1;  // trajectories one range can span
    if (gate_rows > C::max_gate_rows) return false;
    a.slices = slices;
    a.rpx = rpx;
    a.gate_rows = gate_rows;
    LSL_ALLOW_LDS(kern, (size_t)163840);
    hipLaunchKernelGGL(kern, dim3(8 * slices * rpx), dim3(512), C::lds_bytes(gate_rows), st, a);
    return true;
}
bool launch_linear2_ws(int K2, const Lin2Args &a, int shared, hipStream_t st) {
    switch (K2) {
        case 1536: return launch_linear2_ws_t<1536, 3, 3>(a, shared, st);
        case 1280: return launch_linear2_ws_t<1280, 5, 5>(a, shared, st);
        case 768: return launch_linear2_ws_t<768, 3, 3>(a, shared, st);
        case 384: return launch_linear2_ws_t<384, 3, 3>(a, shared, st);
        default: return false;
    }
}

// GEMM tiling (tuning knob LSL_GEMM; every variant sums k in the same order, so results are identical).
//   (features x tokens, waves, BK x ring stages):
//   5  256x256  8 waves 64x2, one tile per workgroup
//   6  256x256  8 waves 32x3, persistent workgroups + next-tile prefetch during the epilogue
//   10 128x128  4 waves 32x3 (used when F is not a multiple of 256: D = 128 / 384 models)
//   7  256x256  8 waves 64x2, persistent, piece-form epilogue (4 KiB staging per wave, next piece prefetched): linear2 default
//   8  256x256  8 waves 32x3, persistent, piece-form epilogue
//   11 128x128  4 waves 64x2
//   12 256x256  8 waves 64x2, persistent, two-phase epilogue staged in ring slot 1 (needs an even number of k-tiles)
//   13 256x128  4 waves 32x2, persistent, two workgroups per CU
//   15 256x256 16 waves 64x2 (64x64 per wave, 4 waves/SIMD: the light linear2 epilogue fits the 128-VGPR budget and the
//      extra occupancy hides load / store latency)
// 20-22: ping-pong halves (k_gemm_pp.hip.h).  Default (-1): 12 for linear1 (5 when K < 512 or not a multiple of 128, 6 when not a multiple of 64), 7 for linear2: the fastest pair measured on MI355X (profiles/r01_gemm_variants.txt lists
// every variant that was tried, including the ones no longer compiled in).
template <class Epi>
int gemm_variant(int F, int K, int N = 1 << 30) {
    static const int forced_all = tune_int("LSL_GEMM", -1);
    static const int forced_1 = tune_int("LSL_GEMM1", -1), forced_2 = tune_int("LSL_GEMM2", -1);  // per GEMM: linear1 / linear2
    const int forced_one = std::is_same<Epi, EpiLinear2>::value ? forced_2 : forced_1;
    const int forced = forced_one >= 0 ? forced_one : forced_all;
    // 256-wide feature tiles waste MFMA work when F is not a multiple of 256 (D = 128 / 384 models): use 128 x 128 there
    const bool ragged = F % 256 != 0 && (F % 256 <= 128);
    const int ragged_variant = std::is_same<Epi, EpiLinear2>::value && K % 64 == 0 ? 11 : 10;  // measured on the D = 384 / 128 models
    if (forced >= 0) return forced;
    if (ragged) return ragged_variant;
    // Small launches (one or two trajectories of the MD17 models, the reference's own B = 4 case): 256 x 256 tiles leave most of the chip
    // idle or run two rounds for 1.2 rounds of work; 128 x 128 tiles (two workgroups per CU) fill it.  Measured (profiles/
    // r02_experiments.txt): md17_bench B = 1 49.1 -> 38.0 ms per call, B = 2 63.0 -> 59.3, md17_ref B = 4 8.00 -> 7.65; from B = 4 of
    // md17_bench on the large tiles win again.  The tile shape does not change any output bit (every element is the same k-ascending
    // chain of 16-deep MFMA steps and the same epilogue arithmetic - checked by the batch-32-vs-batch-1 test at the headline shape),
    // so this may depend on the launch size.
    static const int small_rule = tune_int("LSL_SMALL_TILES", 1);
    const long tiles256 = (long)((N + 255) / 256) * ((F + 255) / 256);
    const int cus = device_cus();
    if (small_rule && K % 64 == 0 && tiles256 * (std::is_same<Epi, EpiLinear2>::value ? 2 : 4) <= (long)cus * (std::is_same<Epi, EpiLinear2>::value ? 1 : 5))
        return 11;  // linear2: tiles <= CUs / 2; linear1: tiles <= 1.25 CUs
    return std::is_same<Epi, EpiLinear2>::value ? (K % 128 == 0 ? 7 : 15) : (K % 128 == 0 ? 12 : 5);
}

// linear2 can also write the next sub-block's LayerNorm + modulate (EpiLinear2::finish_rows) when it runs as the persistent
// row-owner kernel over whole rows of D = 256 or 512 features and the launch has at least as many token tiles as workgroups
bool linear2_can_fuse_ln(int D, int N, int K2) {
    static const int off = tune_int("LSL_LN_FUSE", 0) == 0;  // measured and rejected (k_gemm.hip.h: EpiLinear2): experiments builds only
    if (off || D % 256 != 0 || gemm_variant<EpiLinear2>(D, K2) != 7 || K2 % 128 != 0) return false;
    return gemm_rows_walk<256, 256, 2, 4, 64, 2, EpiPieces<EpiLinear2>>(D, N);
}

template <class Epi>
void launch_gemm(const u16 *W, const u16 *X, int F, int N, int K, const Epi &epi_in, hipStream_t st, int hhd = 32, bool rows = false) {
    const int variant = gemm_variant<Epi>(F, K, N);
    static const int probe = tune_int("LSL_PROBE", 0);
    static const int stagger = tune_int("LSL_STAGGER", 0);
    GemmArgs g{W, X, F, N, K, rows ? 1 : 0, stagger, probe};
    // LSL_NT bit 0: linear1 output, bit 1: linear2 residual update, bit 2: attention output, bit 3: LayerNorm+modulate output
    static const int nt = tune_int("LSL_NT", 3);
    Epi epi = epi_in;
    epi.probe = probe | ((nt >> (std::is_same<Epi, EpiLinear2>::value ? 1 : 0)) & 1 ? 32 : 0);
    const bool pp_ok = !std::is_same<Epi, EpiLinear2>::value ? hhd % 32 == 0 : true;  // linear1 sections start on 32-feature tiles
#ifdef LSL_EXPERIMENTS
    if (variant == 30 && launch_gemm_drain(g, epi, st)) return;
    if (variant == 20 && pp_ok && launch_gemm_pp<32, 4>(g, epi, st)) return;
    if (variant == 21 && pp_ok && launch_gemm_pp<64, 2>(g, epi, st)) return;
    if (variant == 22 && pp_ok && launch_gemm_pp<64, 3>(g, epi, st)) return;
#endif
    if (variant == 12 && K % 128 == 0) return launch_gemm_glds<256, 256, 2, 4, 64, 2, true>(g, epi, st);  // 5 made persistent (staging in ring slot 1)
#ifdef LSL_EXPERIMENTS
    constexpr bool pieces_ok = true;
#else
    constexpr bool pieces_ok = std::is_same<Epi, EpiLinear2>::value;  // (linear1's piece epilogue exists in the experiments build only)
#endif
    if constexpr (pieces_ok) {
        if (variant == 7 && F % 32 == 0 && pp_ok && K % 128 == 0) return launch_gemm_glds<256, 256, 2, 4, 64, 2, true>(g, EpiPieces<Epi>(epi), st);  // persistent, 64-deep k-tiles, piece epilogue
#ifdef LSL_EXPERIMENTS
        if (variant == 8 && F % 32 == 0 && pp_ok) return launch_gemm_glds<256, 256, 2, 4, 32, 3, true>(g, EpiPieces<Epi>(epi), st);  // variant 6 with the piece epilogue
#endif
    }
    switch (variant) {
        case 5: return launch_gemm_glds<256, 256, 2, 4, 64, 2, false>(g, epi, st);
        case 10: return launch_gemm_glds<128, 128, 2, 2, 32, 3, false>(g, epi, st);
        case 11: return launch_gemm_glds<128, 128, 2, 2, 64, 2, false>(g, epi, st);
#ifdef LSL_EXPERIMENTS
        case 23: return launch_gemm_glds<512, 128, 4, 2, 32, 3, false>(g, epi, st);  // whole residual rows per workgroup (F = 512): 120 KiB ring
        case 24: return launch_gemm_glds<512, 128, 4, 2, 32, 2, false>(g, epi, st);
        case 25: return launch_gemm_glds<512, 128, 4, 2, 32, 3, true>(g, EpiPieces<Epi>(epi), st);
        case 16: return launch_gemm_glds<128, 256, 2, 4, 64, 2, false>(g, epi, st);  // 128 features x 256 tokens, 8 waves of 64 x 64
        case 17: return launch_gemm_glds<128, 256, 2, 4, 32, 3, false>(g, epi, st);
        case 18: return launch_gemm_glds<128, 256, 1, 8, 64, 2, false>(g, epi, st);  // 8 waves of 128 x 32
        case 13: return launch_gemm_glds<256, 128, 2, 2, 32, 2, true>(g, epi, st);
        case 14: return launch_gemm_glds<256, 128, 2, 2, 64, 2, true>(g, epi, st);  // 4 waves, one per SIMD, 64-deep k-tiles, one workgroup per CU
#endif
        case 15: return launch_gemm_glds<256, 256, 4, 4, 64, 2, false>(g, epi, st);
#ifdef LSL_EXPERIMENTS
        case 6: return launch_gemm_glds<256, 256, 2, 4, 32, 3, true>(g, epi, st);
#endif
        default: return launch_gemm_glds<256, 256, 2, 4, 64, 2, false>(g, epi, st);  // (K is a multiple of 64: hidden sizes are)
    }
}

template <int HDP, int NW, int ITEMS, int NKT>
void launch_attention_rows(const AttnArgs &a, hipStream_t st) {
    auto kern = k_attention_rows<HDP, NW, ITEMS, NKT>;
    const size_t lds = (size_t)ITEMS * 2 * (NKT > 0 ? NKT * 32 : (a.S + 31) & ~31) * HDP * 2 + NW * sizeof(float);  // K, V, key-norm slots
    LSL_ALLOW_LDS(kern, NKT > 0 ? lds : (size_t)160 * 1024);
    const long items = (long)a.n_seq * a.H;
    hipLaunchKernelGGL(kern, dim3((unsigned)((items + ITEMS - 1) / ITEMS)), dim3(NW * 64), lds, st, a);
}

// persistent, double-buffered form (k_attention_stream): axes of more than 128 positions (unit = (sequence, head, group of 256 queries), keys in
// chunks of 256: peptide's T = 1000 is 4 groups x 4 chunks) and of 9 .. 32 positions
// with a multiple of 8 heads (8 heads of a sequence per unit).  The choice depends on the model and on T, L only - never on the batch - so a
// trajectory's bits are the same in any batch.  LSL_ATTN_STREAM=0 (read in the product too: A/B runs) keeps k_attention_rows.
int attention_stream_mode(int S, int H) {  // 0: k_attention_rows / tiny / online, 1: stream SHORT, 2: stream LONG
    static const int on = env_int("LSL_ATTN_STREAM", 1);
    if (!on) return 0;
    if (S > 128) return 2;  // (round 5: any length - keys in chunks of 256 through the two images, queries in groups of 8 tiles)
    if (S > 8 && S <= 32 && H % 8 == 0) return 1;
    return 0;
}
// q / k / v as head-major planes (k_lin1.hip.h, Lin1Args::planes): spatial sub-blocks (positions = consecutive tokens) whose attention
// runs the LONG stream kernel, token-stationary linear1.  LSL_QKV_PLANES=0 keeps token-major rows (A/B runs).
bool qkv_planes_ok(int hdp, int hidden, int heads, int S, bool temporal, bool lin1_ts) {
    static const int on = env_int("LSL_QKV_PLANES", 1);
    (void)hidden;
    return on && !temporal && lin1_ts && heads % (64 / hdp) == 0 && S <= 256 && attention_stream_mode(S, heads) == 2;
}
// tiny SPATIAL axes (L = 2, 4, 8: positions and sequences are consecutive tokens) on the SHORT stream kernel, 32 / L sequences to a tile with
// the scores outside the block diagonal masked (AttnArgs::blk): replaces k_attention_tiny.  LSL_ATTN_GROUP=0 keeps the lane-per-query kernel.
bool attention_grouped_ok(const AttnArgs &a) {
    static const int on = env_int("LSL_ATTN_GROUP", 1), stream_on = env_int("LSL_ATTN_STREAM", 1);
    return on && stream_on && a.S >= 2 && a.S <= 8 && (a.S & (a.S - 1)) == 0 && a.inner == 1 && a.pos_stride == 1 && a.outer_stride == a.S && a.H % 8 == 0 &&
           a.kmax2 != nullptr;
}
template <int HDP>
bool launch_attention_stream(const AttnArgs &a_in, hipStream_t st) {
    AttnArgs a = a_in;
    a.blk = 0;
    a.n_tok = 0;
    if (attention_grouped_ok(a)) {  // present the tokens as sequences of 32 rows
        a.blk = a.S;
        a.n_tok = a.n_seq * a.S;
        a.n_seq = (a.n_tok + 31) / 32;
        a.S = 32;
        a.outer_stride = 32;
    }
    const int mode = attention_stream_mode(a.S, a.H);
    const bool is_long = mode == 2;
    if (!mode || !a.kmax2) return false;
    const size_t lds = (size_t)2 * 2 * 256 * HDP * 2 + (size_t)8 * 32 * HDP * 2;  // two images of K | V, 256 rows each; a 32-row query image per wave
    const long n_units = is_long ? (long)a.n_seq * a.H * (((a.S + 31) / 32 + 7) / 8) : (long)a.n_seq * (a.H / 8);  // LONG: (sequence, head, group of 8 query tiles)
    const int grid = (int)std::min<long>(2L * device_cus(), n_units);  // two workgroups per CU (2 x 80 KiB of LDS at 32-wide heads)
    // plain stores: behind streaming stores the in-order vector-memory queue reports the next unit's requests late (measured: 0.78 vs 0.27 ms)
    AttnArgs b = a;
    b.nt = 0;
    auto go2 = [&](auto kern) {
        LSL_ALLOW_LDS(kern, lds);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, b);
    };
    if (!is_long && a.blk > 0) go2(k_attention_stream<HDP, false, false, false, true>);
    else if (!is_long) go2(k_attention_stream<HDP, false>);
    else if (a.S <= 256) go2(k_attention_stream<HDP, true>);
    else if constexpr (HDP == 32) {  // keys in chunks of 256, queries in groups of 8 tiles
        if (a.hd == 24) go2(k_attention_stream<HDP, true, true, true>);  // (peptide: the padded head's spare V column carries the softmax denominator)
        else go2(k_attention_stream<HDP, true, true>);
    } else go2(k_attention_stream<HDP, true, true>);
    return true;
}

template <int HDP>
void launch_attention_t(const AttnArgs &a, hipStream_t st) {
    if (launch_attention_stream<HDP>(a, st)) return;
    const int Sp = (a.S + 31) & ~31;
    static const int online = tune_int("LSL_ATTN_ONLINE", 0);  // 1: force the online-softmax kernel (A/B measurements)
    if (!online && a.S <= 8) {  // one lane per (query, head), no MFMA padding
        const long lanes = (long)a.n_seq * a.S * a.H;
        hipLaunchKernelGGL((k_attention_tiny<HDP>), dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, st, a);
        return;
    }
    if (!online && (size_t)2 * Sp * HDP * 2 + 64 <= (size_t)160 * 1024) {  // two-pass softmax, K/V of one (sequence, head) in LDS
        // long axes (peptide T = 1000): 16 waves - with the max pass gone (AttnArgs::bound) the kernel is a chain of MFMA -> exp2 -> MFMA per
        // tile, and four waves per SIMD hide it better than two (attention 320.6 -> 303.8 ms per 1000-step call; with the max pass
        // 8 waves were as fast, profiles/r02_experiments.txt)
        static const int nw16 = tune_int("LSL_ATTN_NW16", 1);
        if (Sp > 256 && nw16) return launch_attention_rows<HDP, 16, 1, 0>(a, st);
        if (Sp > 256) return launch_attention_rows<HDP, 8, 1, 0>(a, st);
        if (Sp <= 32) return launch_attention_rows<HDP, 4, 4, 1>(a, st);
        if (Sp <= 64) return launch_attention_rows<HDP, 4, 2, 2>(a, st);
        if (Sp <= 128) return launch_attention_rows<HDP, 4, 1, 4>(a, st);
        if (Sp <= 192) return launch_attention_rows<HDP, 4, 1, 6>(a, st);
        return launch_attention_rows<HDP, 4, 1, 8>(a, st);
    }
    const long items = (long)a.n_seq * a.H;
    const size_t per_item = (size_t)2 * Sp * HDP * 2;
    if (Sp <= 32) {
        auto kern = k_attention<HDP, 4, 4>;
        hipLaunchKernelGGL(kern, dim3((unsigned)((items + 3) / 4)), dim3(256), 4 * per_item, st, a);
    } else if (Sp <= 64) {
        auto kern = k_attention<HDP, 4, 2>;
        hipLaunchKernelGGL(kern, dim3((unsigned)((items + 1) / 2)), dim3(256), 2 * per_item, st, a);
    } else if (Sp <= 512) {
        auto kern = k_attention<HDP, 4, 1>;
        LSL_ALLOW_LDS(kern, 65536);
        hipLaunchKernelGGL(kern, dim3((unsigned)items), dim3(256), per_item, st, a);
    } else {
        auto kern = k_attention<HDP, 8, 1>;
        LSL_ALLOW_LDS(kern, 160 * 1024);
        hipLaunchKernelGGL(kern, dim3((unsigned)items), dim3(512), per_item, st, a);
    }
}

template <bool PRE, bool POST>
void launch_dense(float *out, const float *in, const float *W, const float *bias, const float *add, int rows, int I, int O,
                  int add_stride, hipStream_t st, bool single = false, int add_mod = 0) {
    // The choice must not depend on the BATCH: the two kernels sum k in different orders, and a trajectory's result has to be the
    // same bits whatever batch it is sampled in (K-sample batching, sharding, pass size).  `single` marks the calls that have one
    // row by construction (the sampler's shared time without class conditioning: one conditioning vector for any batch); they
    // take the wave-per-output kernel (a coalesced GEMV, 8x faster at one row than the 64-row tile kernel).
    if (single && rows == 1 && I <= 512)
        hipLaunchKernelGGL((k_dense_rows<PRE, POST>), dim3((O + 3) / 4), dim3(256), 0, st, out, in, W, bias, add, rows, I, O, add_stride, add_mod);
    else if ((I == 128 || I == 256) && tune_int("LSL_DENSE_MFMA", 1)) {  // many rows, usual widths: the fp32 matrix pipe (k_dense_mfma)
        const dim3 grid((O + 31) / 32, (rows + 31) / 32);
        if (I == 128) hipLaunchKernelGGL((k_dense_mfma<PRE, POST, 32>), grid, dim3(256), 0, st, out, in, W, bias, add, rows, I, O, add_stride, add_mod);
        else hipLaunchKernelGGL((k_dense_mfma<PRE, POST, 64>), grid, dim3(256), 0, st, out, in, W, bias, add, rows, I, O, add_stride, add_mod);
    } else if (I % 4 == 0)
        hipLaunchKernelGGL((k_dense_tiled<PRE, POST>), dim3((O + 63) / 64, (rows + 63) / 64), dim3(256), 0, st, out, in, W, bias, add, rows,
                           I, O, add_stride, add_mod);
    else
        hipLaunchKernelGGL((k_dense_rows<PRE, POST>), dim3((O + 3) / 4), dim3(256), 0, st, out, in, W, bias, add, rows, I, O, add_stride, add_mod);
}

// the short GEMM chains in front of the trajectory-resident kernel (k_dense_mfma: latency-optimised; one kernel for any row count)
template <bool PRE, bool POST>
void launch_dense_small(float *out, const float *in, const float *W, const float *bias, const float *add, int rows, int I, int O,
                        int add_stride, hipStream_t st, int add_mod = 0) {
    const dim3 grid((O + 31) / 32, (rows + 31) / 32);
    if (I == 128)
        hipLaunchKernelGGL((k_dense_mfma<PRE, POST, 32>), grid, dim3(256), 0, st, out, in, W, bias, add, rows, I, O, add_stride, add_mod);
    else if (I == 256)
        hipLaunchKernelGGL((k_dense_mfma<PRE, POST, 64>), grid, dim3(256), 0, st, out, in, W, bias, add, rows, I, O, add_stride, add_mod);
    else
        launch_dense<PRE, POST>(out, in, W, bias, add, rows, I, O, add_stride, st, false, add_mod);
}

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
            jobs.sq_bound[k] = (t & 1) ? ws.kmax2 + bi : nullptr;
            max_pos = std::max(max_pos, jobs.n_pos[k]);
        }
        hipLaunchKernelGGL(k_rope_scaled, dim3((max_pos * half + 255) / 256, jobs.n_jobs), dim3(256), 0, st, jobs, m->d.head_dim, m->d.head_dim_pad, m->d.theta);
    }
}

// one ParallelMLPAttentionV2 sub-block on h (in place): LN+modulate -> linear1 -> attention -> linear2
// a_ready: ws.a already holds this sub-block's LayerNorm + modulate (written by the previous sub-block's linear2); fuse_next: let this
// sub-block's linear2 write the next one's when the launch allows it (*a_written reports whether it did)
int run_block(lsl_model *m, const Workspace &ws, int bi, float *h, const float *mods, int mod_stride, int bc, int T, int L,
              hipStream_t st, bool a_ready = false, bool fuse_next = false, bool *a_written = nullptr, bool stop_before_linear2 = false) {
    const lsl_model_desc &d = m->d;
    const lsl_block_weights &bw = m->blocks[bi];
    const int D = d.hidden, n = bc * T * L, layer = bi / 2, temporal = bi & 1;
    const float *mbase = mods + (size_t)layer * 6 * D + (temporal ? 3 * D : 0);  // shift, scale, gate
    if (!a_ready) {
        m->prof.begin(3, st);
        DISPATCH_D(D, launch_ln_mod_t, ws.a, h, mbase, mbase + D, mod_stride, n, T * L, st);
        m->prof.end(3, st);
    }
    m->prof.begin(0, st);

    const float premul = (float)(1.4426950408889634 / std::sqrt((double)d.head_dim));
    // position of token n along the attended axis = (n / pdiv) % pmod, done with multiply-high in the epilogue: exact while
    // n * d < 2^32, and n < 2^18 (pass size) with d <= T or L
    const int pdiv = temporal ? L : 1, pmod = temporal ? T : L;
    auto magic_of = [](int dv) { return dv == 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)dv + 1); };
    if ((unsigned long long)n * (unsigned)std::max(pdiv, pmod) >= (1ull << 32)) return fail(-3, "pass too large for the position arithmetic");
    const bool lin1_ts = linear1_ts_ok(d.head_dim_pad, D, m->F1, m->HHD, n);
    const int npad = (n + 255) & ~255;
    const bool planes = qkv_planes_ok(d.head_dim_pad, D, d.heads, temporal ? T : L, temporal != 0, lin1_ts);
    // head-major planes are addressed with 32-bit per-lane byte offsets over the whole q | k | v buffer (k_lin1.hip.h flush, k_attn.hip.h
    // stream requests): a pass set larger than that through lsl_model_set_chunk / LSL_CHUNK_TRAJ is refused, never wrapped
    if (planes && (unsigned long long)npad * 3ull * (unsigned)m->HHD * 2ull >= (1ull << 32)) return fail(-3, "pass too large for the q/k/v plane offsets (%d tokens: at most %llu with this model)", n, (unsigned long long)((1ull << 32) / (6ull * (unsigned)m->HHD)) - 256);
    if (lin1_ts) {
        const Lin1Args la{(const u16 *)bw.w1, ws.a, bw.b1, ws.rope_qk + (size_t)(2 * bi) * ws.rope_qk_stride,
                          ws.rope_qk + (size_t)(2 * bi + 1) * ws.rope_qk_stride, ws.qkv, ws.z, m->F1, n, m->HHD, d.mlp_dim,
                          pdiv, pmod, magic_of(pdiv), magic_of(pmod), 1.0f / d.head_dim, premul, 1, nullptr, 0, planes ? 1 : 0, npad};
        launch_linear1_ts(d.head_dim_pad, D, la, st);
    } else if (d.head_dim_pad == 32) {
        EpiLinear1<32> e{bw.b1, bw.qs, bw.ks, temporal ? ws.rope_t : ws.rope_l, ws.rope_qk + (size_t)(2 * bi) * ws.rope_qk_stride,
                         ws.rope_qk + (size_t)(2 * bi + 1) * ws.rope_qk_stride, ws.qkv, ws.z, m->HHD, d.mlp_dim,
                         pdiv, pmod, magic_of(pdiv), magic_of(pmod), 1.0f / d.head_dim, premul, 0};
        launch_gemm((const u16 *)bw.w1, ws.a, m->F1, n, D, e, st, m->HHD);
    } else {
        EpiLinear1<16> e{bw.b1, bw.qs, bw.ks, temporal ? ws.rope_t : ws.rope_l, ws.rope_qk + (size_t)(2 * bi) * ws.rope_qk_stride,
                         ws.rope_qk + (size_t)(2 * bi + 1) * ws.rope_qk_stride, ws.qkv, ws.z, m->HHD, d.mlp_dim,
                         pdiv, pmod, magic_of(pdiv), magic_of(pmod), 1.0f / d.head_dim, premul, 0};
        launch_gemm((const u16 *)bw.w1, ws.a, m->F1, n, D, e, st, m->HHD);
    }
    m->prof.end(0, st);
    static const int nt_mask = tune_int("LSL_NT", 3);
    AttnArgs aa;
    aa.nt = (nt_mask >> 2) & 1;
    aa.qkv = ws.qkv;
    aa.z = ws.z;
    aa.HHD = m->HHD;
    aa.zw = m->K2;
    aa.H = d.heads;
    aa.hd = d.head_dim;
    static const int attn_bound = tune_int("LSL_ATTN_BOUND", 1);
    aa.kmax2 = ws.kmax2 + bi;
    aa.planes = planes ? 1 : 0;
    aa.npad = npad;
    aa.bound = attn_bound == 2 || (attn_bound == 1 && (temporal ? T : L) > 96);  // short axes: the max pass is one or two tiles, cheaper than the norms
    if (!temporal) {  // sequences (b,t), positions l
        aa.S = L; aa.n_seq = bc * T; aa.inner = 1; aa.outer_stride = L; aa.pos_stride = 1;
    } else {          // sequences (b,l), positions t
        aa.S = T; aa.n_seq = bc * L; aa.inner = L; aa.outer_stride = T * L; aa.pos_stride = L;
    }
    m->prof.begin(2, st);
    if (d.head_dim_pad == 32) launch_attention_t<32>(aa, st);
    else launch_attention_t<16>(aa, st);
    m->prof.end(2, st);

    if (stop_before_linear2) {  // (lsl_debug_taps)
        LSL_CHECK_LAUNCH("block");
        return 0;
    }
    m->prof.begin(1, st);
    if ((unsigned long long)n * (unsigned)(T * L) >= (1ull << 32)) return fail(-3, "pass too large for the trajectory arithmetic");
    const bool fuse = fuse_next && bi + 1 < 2 * d.depth && linear2_can_fuse_ln(D, n, m->K2);
    const float *nbase = mods + (size_t)((bi + 1) / 2) * 6 * D + (((bi + 1) & 1) ? 3 * D : 0);  // next sub-block: shift, scale
    bool on_ws = false;
    if (ws.w2p && !fuse && (unsigned long long)n * (unsigned)(4 * D) < (1ull << 32)) {  // (32-bit byte offsets into h)
        const Lin2Args l2{ws.w2p + (size_t)bi * D * m->K2, ws.z, bw.b2, mbase + 2 * D, h, D, n, mod_stride, T * L, magic_of(T * L), 0, 0, 0, nullptr};
        on_ws = launch_linear2_ws(m->K2, l2, mod_stride == 0, st);
    }
    if (!on_ws) {
        EpiLinear2 e2{bw.b2, mbase + 2 * D, h, D, mod_stride, T * L, 0, magic_of(T * L), fuse ? ws.a : nullptr, nbase, nbase + D};
        launch_gemm((const u16 *)bw.w2, ws.z, D, n, m->K2, e2, st, 32, fuse);
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
    launch_embed<1>(ws.h, x, m->w.x_in_w, nullptr, nullptr, nullptr, nullptr, ws.cond_emb, n, d.in_dim, D, st);
    if (d.normalize) { DISPATCH_D(D, launch_ln_inplace_t, ws.h, n, 1e-5f, st); }
    m->prof.end(5, st);
    LSL_CHECK_LAUNCH("embed");
    bool a_ready = false;  // the first sub-block of an evaluation runs the standalone LayerNorm; later ones get `a` from the previous linear2
    for (int bi = 0; bi < 2 * d.depth; ++bi) {
        bool wrote = false;
        rc = run_block(m, ws, bi, ws.h, mods, mod_stride, bc, T, L, st, a_ready, true, &wrote);
        if (rc) return rc;
        a_ready = wrote;
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
    return !off && d.hidden == RES_D && d.heads == RES_H && d.head_dim == RES_HD && d.head_dim_pad == RES_HD && d.mlp_dim == RES_M &&
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
    size_t need = carve(m, nullptr, chunk, io->T, io->L).bytes * lanes_for(io->B, io->T, io->L);
    if (resident_ok(m, io->T, io->L)) need = std::max(need, carve_resident(m, nullptr, io->B, io->T, io->L, m->d.vec_in_dim > 0).bytes);
    if (!ws || ws_bytes < need) return fail(-4, "workspace too small: need %zu bytes, got %zu", need, ws_bytes);
    *chunk_out = chunk;
    return 0;
}

}  // namespace

#include "decode_host.hip.h"

extern "C" {

int lsl_version(void) { return LSL_VERSION; }
const char *lsl_build_info(void) { return "clang " __clang_version__ " gfx950"; }
const char *lsl_last_error(void) { return g_err; }

int lsl_model_create(const lsl_model_desc *desc, lsl_model **out) try {
    if (!desc || !out) return fail(-1, "null argument");
    const lsl_model_desc &d = *desc;
    if (d.heads <= 0 || d.hidden % d.heads != 0)
        return fail(-20, "Hidden size %d must be divisible by num_heads %d", d.hidden, d.heads);  // latent_si_v31.py:92-95
    if (d.head_dim != d.hidden / d.heads) return fail(-21, "head_dim must equal hidden / heads");
    if (d.hidden % 64 != 0 || d.hidden < 64 || d.hidden > 512) return fail(-21, "hidden_size %d unsupported (multiple of 64, 64..512)", d.hidden);
    if (d.head_dim % 2 != 0 || d.head_dim > 32) return fail(-21, "head_dim %d unsupported (even, <= 32)", d.head_dim);
    if (d.head_dim_pad != (d.head_dim <= 16 ? 16 : 32)) return fail(-21, "head_dim_pad must be 16 (head_dim <= 16) or 32");
    if ((d.heads * d.head_dim_pad) % 32 != 0) return fail(-21, "heads * head_dim_pad must be a multiple of 32");
    if (d.mlp_dim <= 0 || d.mlp_dim % 32 != 0) return fail(-21, "mlp_dim %d must be a positive multiple of 32", d.mlp_dim);
    if ((d.heads * d.head_dim_pad + d.mlp_dim) % 64 != 0) return fail(-21, "heads*head_dim_pad + mlp_dim must be a multiple of 64");
    if (3 * d.heads * d.head_dim_pad + d.mlp_dim > 7936) return fail(-21, "3 * heads * head_dim_pad + mlp_dim = %d too wide (the linear1 bias lives in LDS beside a 128 KiB operand ring: <= 7936)", 3 * d.heads * d.head_dim_pad + d.mlp_dim);
    if (d.in_dim <= 0 || d.in_dim > 128) return fail(-21, "in_dim %d unsupported (1..128)", d.in_dim);
    if (d.depth <= 0 || d.depth > 64) return fail(-21, "depth %d unsupported", d.depth);
    if (d.vec_in_dim < 0 || d.vec_in_dim > 512) return fail(-21, "vec_in_dim %d unsupported (<= 512)", d.vec_in_dim);
    lsl_model *m = new (std::nothrow) lsl_model();
    if (!m) return fail(-5, "out of host memory");
    m->d = d;
    m->HHD = d.heads * d.head_dim_pad;
    m->F1 = 3 * m->HHD + d.mlp_dim;
    m->K2 = m->HHD + d.mlp_dim;
    m->MODW = (6 * d.depth + 2) * d.hidden;
    *out = m;
    return 0;
} catch (const std::bad_alloc &) {
    return fail(-5, "out of host memory");
} catch (...) {
    return fail(-11, "unexpected C++ exception");
}

int lsl_model_set_weights(lsl_model *m, const lsl_weights *w) try {
    if (!m || !w || !w->blocks) return fail(-1, "null argument");
    const void *req[] = {w->x_in_w, w->x_in_b, w->cond_w, w->cond_b, w->mask_emb, w->time_freqs, w->time_w1, w->time_b1,
                         w->time_w2, w->time_b2, w->mod_w, w->mod_b, w->out_w, w->out_b};
    for (const void *p : req)
        if (!p) return fail(-2, "missing weight pointer");
    if (m->d.vec_in_dim > 0 && (!w->vec_w1 || !w->vec_b1 || !w->vec_w2 || !w->vec_b2)) return fail(-2, "missing vec_in weights");
    m->blocks.assign(w->blocks, w->blocks + 2 * m->d.depth);
    for (const auto &b : m->blocks)
        if (!b.w1 || !b.b1 || !b.qs || !b.ks || !b.w2 || !b.b2) return fail(-2, "missing block weight pointer");
    m->w = *w;
    m->w.blocks = m->blocks.data();
    m->has_weights = true;
    for (auto &g : m->graphs)  // captured launches hold the old weight pointers
        if (g.exec) hipGraphExecDestroy(g.exec);
    m->graphs.clear();
    m->seen.clear();
    m->uncapturable.clear();
    return 0;
} catch (const std::bad_alloc &) {
    return fail(-5, "out of host memory");
} catch (...) {
    return fail(-11, "unexpected C++ exception");
}

void lsl_model_destroy(lsl_model *m) {
    if (m) {
        m->prof.clear();
        if (m->lane_fork) hipEventDestroy(m->lane_fork);
        if (m->lane_join) hipEventDestroy(m->lane_join);
        if (m->lane_stream) hipStreamDestroy(m->lane_stream);
        for (auto &g : m->graphs)
            if (g.exec) hipGraphExecDestroy(g.exec);
        if (m->graph_stream) hipStreamDestroy(m->graph_stream);
    }
    delete m;
}

int lsl_profile_enable(lsl_model *m, int32_t kernel, int32_t max_launches) try {
    if (!m) return fail(-1, "null model");
    m->prof.clear();
    if (kernel < 0 || max_launches <= 0) return 0;
    m->prof.ev.resize(2 * (size_t)max_launches);
    for (auto &e : m->prof.ev)
        if (hipEventCreate(&e) != hipSuccess) return fail(-10, "hipEventCreate failed");
    m->prof.kernel = kernel;
    m->prof.cap = max_launches;
    return 0;
} catch (const std::bad_alloc &) {
    return fail(-5, "out of host memory");
} catch (...) {
    return fail(-11, "unexpected C++ exception");
}

int lsl_profile_read(lsl_model *m, double *total_ms, int32_t *launches) {
    if (!m || !total_ms || !launches) return fail(-1, "null argument");
    double tot = 0.0;
    for (int i = 0; i < m->prof.used; ++i) {
        hipEventSynchronize(m->prof.ev[2 * i + 1]);
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, m->prof.ev[2 * i], m->prof.ev[2 * i + 1]) != hipSuccess) return fail(-10, "hipEventElapsedTime failed");
        tot += ms;
    }
    *total_ms = tot;
    *launches = m->prof.used;
    m->prof.used = 0;
    return 0;
}

static void drop_graphs(lsl_model *m) {
    for (auto &g : m->graphs)
        if (g.exec) hipGraphExecDestroy(g.exec);
    m->graphs.clear();
    m->seen.clear();
    m->uncapturable.clear();
}

int lsl_model_set_chunk(lsl_model *m, int32_t c) try {
    if (!m || c < 0) return fail(-1, "bad argument");
    m->chunk = c;
    drop_graphs(m);
    return 0;
} catch (const std::bad_alloc &) {
    return fail(-5, "out of host memory");
} catch (...) {
    return fail(-11, "unexpected C++ exception");
}

int32_t lsl_pass_size(const lsl_model *m, int32_t B, int32_t T, int32_t L) {
    if (!m || B <= 0 || T <= 0 || L <= 0) return 0;
    return default_chunk(m, B, T, L);
}

#ifdef LSL_EXPERIMENTS
// tools only: read and clear the phase clock of k_resident (cycles of workgroup 0 / wave 0 per phase)
int lsl_debug_res_stamps(unsigned long long *out8) {
    hipDeviceSynchronize();
    unsigned long long z[8] = {0};
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_res_stamps), sizeof(z)) != hipSuccess) return fail(-10, "stamps");
    hipMemcpyToSymbol(HIP_SYMBOL(g_res_stamps), z, sizeof(z));
    return 0;
}
#endif

int32_t lsl_sampler_path(const lsl_model *m, int32_t T, int32_t L) {
    if (!m || T <= 0 || L <= 0) return -1;
    return resident_ok(m, T, L) && m->prof.kernel < 0 ? 1 : 0;  // (per-kernel profiling runs the general kernels: lsl_sample below)
}

size_t lsl_workspace_bytes(const lsl_model *m, int32_t B, int32_t T, int32_t L) {
    if (!m || B <= 0 || T <= 0 || L <= 0) return 0;
    size_t need = carve(m, nullptr, default_chunk(m, B, T, L), T, L).bytes * lanes_for(B, T, L);
    if (resident_ok(m, T, L)) need = std::max(need, carve_resident(m, nullptr, B, T, L, m->d.vec_in_dim > 0).bytes);
    return need;
}

int lsl_forward(lsl_model *m, const lsl_io *io, void *workspace, size_t workspace_bytes, void *stream) try {
    DeviceGuard dev_guard_((hipStream_t)stream);
    int chunk = 0;
    if (int rc = check_call(m, io, workspace_bytes, workspace, &chunk)) return rc;
    if (!io->t || !io->out) return fail(-3, "t and out are required");
    hipStream_t st = (hipStream_t)stream;
    const Workspace ws = carve(m, (char *)workspace, chunk, io->T, io->L);
    run_tables(m, ws, io->T, io->L, st);
    const size_t per = (size_t)io->T * io->L * m->d.in_dim;
    for (int b0 = 0; b0 < io->B; b0 += chunk) {
        const int bc = io->B - b0 < chunk ? io->B - b0 : chunk;
        const float *y = io->y ? io->y + (size_t)b0 * m->d.vec_in_dim : nullptr;
        if (int rc = prepare_pass(m, ws, io->x_cond + b0 * per, io->mask + (size_t)b0 * io->T * io->L, y, bc, io->T, io->L, st)) return rc;
        if (int rc = run_eval(m, ws, io->x + b0 * per, io->out + b0 * per, io->t + b0, 0.0f, y != nullptr, bc, io->T, io->L, 0, 0, 0, 0,
                              nullptr, 0, 0, 0, nullptr, st))
            return rc;
    }
    return 0;
} catch (const std::bad_alloc &) {
    return fail(-5, "out of host memory");
} catch (...) {
    return fail(-11, "unexpected C++ exception");
}

static int sample_enqueue(lsl_model *m, const lsl_io *io, const lsl_step_ex *steps, int32_t n_steps, const float *noise, uint64_t seed,
                          uint64_t elem_offset, float *trace, void *workspace, int chunk, hipStream_t st);

int lsl_sample(lsl_model *m, const lsl_io *io, const lsl_step *steps, int32_t n_steps, const float *noise, int32_t n_noise, uint64_t seed,
               uint64_t elem_offset, float *trace, void *workspace, size_t workspace_bytes, void *stream) try {
    if (!steps || n_steps <= 0) return fail(-3, "steps required");
    std::vector<lsl_step_ex> ex((size_t)n_steps);
    for (int s = 0; s < n_steps; ++s) ex[s] = lsl_step_ex{steps[s].t, steps[s].ax, steps[s].am, steps[s].aw, 0.0f, 0, s, s};
    return lsl_sample_ex(m, io, ex.data(), n_steps, noise, n_noise, seed, elem_offset, trace, n_steps, workspace, workspace_bytes, stream);
} catch (const std::bad_alloc &) {
    return fail(-5, "out of host memory");
} catch (...) {
    return fail(-11, "unexpected C++ exception");
}

int lsl_sample_ex(lsl_model *m, const lsl_io *io, const lsl_step_ex *steps, int32_t n_steps, const float *noise, int32_t n_noise, uint64_t seed,
                  uint64_t elem_offset, float *trace, int32_t n_trace, void *workspace, size_t workspace_bytes, void *stream) try {
    DeviceGuard dev_guard_((hipStream_t)stream);
    int chunk = 0;
    if (int rc = check_call(m, io, workspace_bytes, workspace, &chunk)) return rc;
    if (!steps || n_steps <= 0) return fail(-3, "steps required");
    bool plain = true, have_saved = false;
    for (int s = 0; s < n_steps; ++s) {
        const lsl_step_ex &sp = steps[s];
        if (sp.aw != 0.0f && (sp.noise_index < 0 || (noise && sp.noise_index >= n_noise)))
            return fail(-3, "step %d needs noise slice %d but only %d slices were given", s, sp.noise_index, n_noise);
        if (sp.trace_index < -1 || (trace && sp.trace_index >= n_trace))
            return fail(-3, "step %d: trace slice %d out of range (the trace buffer holds %d slices)", s, sp.trace_index, trace ? n_trace : 0);
        if (sp.as != 0.0f && !have_saved) return fail(-3, "step %d reads the saved state before any record saved one", s);
        have_saved |= (sp.flags & LSL_STEP_SAVE) != 0;
        plain &= sp.as == 0.0f && sp.flags == 0 && sp.noise_index == s && sp.trace_index == s;
    }
    hipStream_t st = (hipStream_t)stream;
    if (plain && resident_ok(m, io->T, io->L) && m->prof.kernel < 0) {  // small trajectories: the whole loop in one launch per group of updates
        std::vector<lsl_step> ps((size_t)n_steps);
        for (int s = 0; s < n_steps; ++s) ps[s] = lsl_step{steps[s].t, steps[s].ax, steps[s].am, steps[s].aw};
        return resident_sample(m, io, ps.data(), n_steps, noise, seed, elem_offset, trace, workspace, st);
    }
    // hipGraph replay (opt-in).  LSL_GRAPH: 0 off (default), 1 for launch-bound calls (at most 64 Ki tokens per pass and 4096 launches) whose
    // arguments repeat, 2 for every call of at most 4096 launches: the first appearance of an argument set runs eagerly (it also initialises
    // the per-kernel attributes), the second is captured, later ones are replayed.  Measured on MI355X (tools/latency_small_batch.py): a
    // 10-update pedestrian call (~600 launches) takes 3.38 ms eagerly and 3.23 ms replayed - the floor of the small-batch configs is the
    // GPU-side cost of ~600 dependent tiny kernels (5 us each), not the host launches, so replay buys 0-4 % and stays off by default.
    static const int use_graph = env_int("LSL_GRAPH", 0);
    const int passes = (io->B + chunk - 1) / chunk;
    const long est_launches = (long)passes * n_steps * (8L * m->d.depth + 6);
    const bool launch_bound = (size_t)chunk * io->T * io->L <= 65536;
    if (use_graph && (launch_bound || use_graph >= 2) && m->prof.kernel < 0 && lanes_for(io->B, io->T, io->L) == 1 && est_launches <= 4096) {
        std::vector<unsigned char> key;
        auto put = [&](const void *p, size_t n) { key.insert(key.end(), (const unsigned char *)p, (const unsigned char *)p + n); };
        put(io, sizeof(*io));
        put(steps, sizeof(lsl_step_ex) * n_steps);
        put(&noise, sizeof(noise));
        put(&seed, sizeof(seed));
        put(&elem_offset, sizeof(elem_offset));
        put(&trace, sizeof(trace));
        put(&workspace, sizeof(workspace));
        put(&chunk, sizeof(chunk));
        put(&st, sizeof(st));
        for (auto &g : m->graphs)
            if (g.key == key) {
                g.last_use = ++m->graph_clock;
                if (hipGraphLaunch(g.exec, st) != hipSuccess) return fail(-10, "hipGraphLaunch failed");
                return 0;
            }
        bool second = false, bad = m->graph_stream_failed;
        for (auto &k : m->uncapturable) bad |= (k == key);
        for (auto &k : m->seen) second |= (k == key);
        if (bad) {
            // capture failed before for this argument set (or no capture stream): eager from now on
        } else if (second) {
            hipGraph_t graph = nullptr;
            if (!m->graph_stream && hipStreamCreateWithFlags(&m->graph_stream, hipStreamNonBlocking) != hipSuccess) {
                m->graph_stream = nullptr;
                m->graph_stream_failed = true;
            }
            hipStream_t cs = m->graph_stream;
            if (cs && hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                const int rc = sample_enqueue(m, io, steps, n_steps, noise, seed, elem_offset, trace, workspace, chunk, cs);
                const hipError_t e = hipStreamEndCapture(cs, &graph);
                hipGraphExec_t exec = nullptr;
                if (rc == 0 && e == hipSuccess && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                    hipGraphDestroy(graph);
                    if (m->graphs.size() >= 8) {  // evict the least recently used
                        size_t lru = 0;
                        for (size_t i = 1; i < m->graphs.size(); ++i)
                            if (m->graphs[i].last_use < m->graphs[lru].last_use) lru = i;
                        hipGraphExecDestroy(m->graphs[lru].exec);
                        m->graphs.erase(m->graphs.begin() + lru);
                    }
                    lsl_model::GraphEntry ge;
                    ge.key = key;
                    ge.exec = exec;
                    ge.last_use = ++m->graph_clock;
                    m->graphs.push_back(std::move(ge));
                    if (hipGraphLaunch(exec, st) != hipSuccess) return fail(-10, "hipGraphLaunch failed");
                    return 0;
                }
                if (graph) hipGraphDestroy(graph);
            }
            (void)hipGetLastError();  // capture not possible: forget the error, run eagerly, and never try this argument set again
            for (size_t i = 0; i < m->seen.size(); ++i)
                if (m->seen[i] == key) {
                    m->seen.erase(m->seen.begin() + i);
                    break;
                }
            if (m->uncapturable.size() >= 16) m->uncapturable.erase(m->uncapturable.begin());
            m->uncapturable.push_back(key);
        } else {
            if (m->seen.size() >= 16) m->seen.erase(m->seen.begin());
            m->seen.push_back(key);
        }
    }
    return sample_enqueue(m, io, steps, n_steps, noise, seed, elem_offset, trace, workspace, chunk, st);
} catch (const std::bad_alloc &) {
    return fail(-5, "out of host memory");
} catch (...) {
    return fail(-11, "unexpected C++ exception");
}

static int sample_enqueue(lsl_model *m, const lsl_io *io, const lsl_step_ex *steps, int32_t n_steps, const float *noise, uint64_t seed,
                          uint64_t elem_offset, float *trace, void *workspace, int chunk, hipStream_t st) {
    // (per-kernel profiling brackets launches with events on ONE stream: un-overlapped, single lane)
    const int lanes = (lanes_for(io->B, io->T, io->L) == 2 && io->B > chunk && m->prof.kernel < 0) ? 2 : 1;
    hipStream_t lane_st[2] = {st, st};
    if (lanes == 2) {
        if (!m->lane_stream) {
            if (hipStreamCreateWithFlags(&m->lane_stream, hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&m->lane_fork, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&m->lane_join, hipEventDisableTiming) != hipSuccess)
                return fail(-10, "could not create the second lane's stream");
        }
        lane_st[1] = m->lane_stream;
        hipEventRecord(m->lane_fork, st);  // lane 1 starts after everything the caller enqueued before this call
        hipStreamWaitEvent(m->lane_stream, m->lane_fork, 0);
    }
    const size_t ws_lane = carve(m, nullptr, chunk, io->T, io->L).bytes;
    Workspace wss[2];
    for (int l = 0; l < lanes; ++l) {
        wss[l] = carve(m, (char *)workspace + l * ws_lane, chunk, io->T, io->L);
        run_tables(m, wss[l], io->T, io->L, lane_st[l]);
    }
    const size_t per = (size_t)io->T * io->L * m->d.in_dim;
    const size_t total = per * io->B;
    // no class conditioning: modulation tables per group of network records, one row each (run_mods_steps); a lane recomputes a group only
    // when its pass crosses into another one (calls of at most mods_group records: once per call).  LSL_MODS_GROUP=0: per evaluation.
    // (>= 2: at most that many records per group - the GPU suite crosses group boundaries with it)
    static const int group_on = env_int("LSL_MODS_GROUP", 1);
    const int G = (group_on && !io->y) ? (group_on >= 2 ? std::min(group_on, wss[0].mods_group) : wss[0].mods_group) : 0;
    std::vector<int> net_idx;
    std::vector<float> net_t;
    if (G) {
        net_idx.resize((size_t)n_steps);
        for (int s = 0; s < n_steps; ++s) {
            net_idx[s] = (int)net_t.size();
            if (!(steps[s].flags & LSL_STEP_NO_NETWORK)) net_t.push_back(steps[s].t);
        }
    }
    int cur_group[2] = {-1, -1};
    // passes in groups of `lanes`; within a group the launches of the lanes are interleaved step by step so that both queues fill together
    for (int g0 = 0; g0 < io->B; g0 += chunk * lanes) {
        int b0s[2], bcs[2], nl = 0;
        for (int l = 0; l < lanes; ++l) {
            const int b0 = g0 + l * chunk;
            if (b0 >= io->B) break;
            b0s[nl] = b0;
            bcs[nl] = io->B - b0 < chunk ? io->B - b0 : chunk;
            ++nl;
        }
        for (int l = 0; l < nl; ++l) {
            const float *y = io->y ? io->y + (size_t)b0s[l] * m->d.vec_in_dim : nullptr;
            if (int rc = prepare_pass(m, wss[l], io->x_cond + b0s[l] * per, io->mask + (size_t)b0s[l] * io->T * io->L, y, bcs[l], io->T, io->L, lane_st[l]))
                return rc;
        }
        for (int s = 0; s < n_steps; ++s) {
            const lsl_step_ex &sp = steps[s];
            for (int l = 0; l < nl; ++l) {
                const int b0 = b0s[l];
                const float *nz = nullptr;
                if (sp.aw != 0.0f && noise) nz = noise + (size_t)sp.noise_index * total + b0 * per;
                float *tr = trace && sp.trace_index >= 0 ? trace + (size_t)sp.trace_index * total + b0 * per : nullptr;
                const float *saved = sp.as != 0.0f ? wss[l].saved : nullptr;        // (the pass's own copy: passes run all records for their trajectories)
                float *save_out = (sp.flags & LSL_STEP_SAVE) ? wss[l].saved : nullptr;
                if (sp.flags & LSL_STEP_NO_NETWORK) {
                    const unsigned long long ne = (unsigned long long)bcs[l] * per;
                    hipLaunchKernelGGL(k_state_affine, dim3((unsigned)std::min<unsigned long long>((ne + 255) / 256, 2048)), dim3(256), 0, lane_st[l],
                                       io->x + b0 * per, ne, sp.ax, sp.aw, sp.as, nz, (unsigned long long)seed, (unsigned)sp.noise_index,
                                       (unsigned long long)(elem_offset + b0 * per), saved, save_out, tr);
                    LSL_CHECK_LAUNCH("state update");
                    continue;
                }
                const float *mods_ready = nullptr;
                if (G) {
                    const int k = net_idx[s], g = k / G;
                    if (cur_group[l] != g) {
                        if (int rc = run_mods_steps(m, wss[l], net_t.data() + (size_t)g * G, std::min(G, (int)net_t.size() - g * G), lane_st[l])) return rc;
                        cur_group[l] = g;
                    }
                    mods_ready = wss[l].mods_all + (size_t)(k - g * G) * m->MODW;
                }
                if (int rc = run_eval(m, wss[l], io->x + b0 * per, nullptr, nullptr, sp.t, io->y != nullptr, bcs[l], io->T, io->L, 1, sp.ax, sp.am, sp.aw,
                                      nz, seed, (unsigned)sp.noise_index, elem_offset + b0 * per, tr, lane_st[l], sp.as, saved, save_out, mods_ready))
                    return rc;
            }
        }
    }
    if (lanes == 2) {  // the caller's stream continues after lane 1 has finished
        hipEventRecord(m->lane_join, m->lane_stream);
        hipStreamWaitEvent(st, m->lane_join, 0);
    }
    return 0;
}

// x_0 ~ N(0, 1) from the documented counter stream (k_small.hip.h: k_randn); the reference draws torch.randn_like(x_cond)
// (lightning_base.py:231), whose generator stream cannot be reproduced off an NVIDIA/torch build anyway.
int lsl_randn(float *x, uint64_t n, uint64_t seed, uint64_t elem_offset, void *stream) try {
    DeviceGuard dev_guard_((hipStream_t)stream);
    if (!x && n) return fail(-1, "null argument");
    if (!n) return 0;
    const unsigned long long blocks = (n + 255) / 256;
    const unsigned grid = (unsigned)std::min<unsigned long long>(blocks, (unsigned long long)device_cus() * 16);
    hipLaunchKernelGGL(k_randn, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (unsigned long long)n, (unsigned long long)seed, LSL_INIT_STEP,
                       (unsigned long long)elem_offset);
    LSL_CHECK_LAUNCH("lsl_randn");
    return 0;
} catch (...) {
    return fail(-11, "unexpected C++ exception");
}

// ---- Runge-Kutta arithmetic of the adaptive sampler (k_small.hip.h: k_rk_*) ----
static int rk_terms(RkTerms &t, const float *const *x, const float *c, int32_t n_x) {
    if (!x || !c || n_x < 1 || n_x > 8) return fail(-3, "1 to 8 terms");
    t.n = n_x;
    for (int j = 0; j < 8; ++j) {
        t.x[j] = j < n_x ? x[j] : nullptr;
        t.c[j] = j < n_x ? c[j] : 0.0f;
        if (j < n_x && !x[j]) return fail(-1, "null term pointer");
    }
    return 0;
}
static unsigned rk_grid(uint64_t n) { return (unsigned)std::min<uint64_t>((n + 255) / 256, (uint64_t)device_cus() * 8); }

int lsl_rk_lincomb(float *out, const float *const *x, const float *c, int32_t n_x, uint64_t n, void *stream) try {
    DeviceGuard dev_guard_((hipStream_t)stream);
    if (!out && n) return fail(-1, "null argument");
    RkTerms t;
    if (int rc = rk_terms(t, x, c, n_x)) return rc;
    if (!n) return 0;
    hipLaunchKernelGGL(k_rk_lincomb, dim3(rk_grid(n)), dim3(256), 0, (hipStream_t)stream, out, t, (unsigned long long)n);
    LSL_CHECK_LAUNCH("lsl_rk_lincomb");
    return 0;
} catch (...) {
    return fail(-11, "unexpected C++ exception");
}

int lsl_rk_dense(float *out, const float *a, const float *b, const float *c, const float *d, const float *e, float x, uint64_t n, void *stream) try {
    DeviceGuard dev_guard_((hipStream_t)stream);
    if (n && (!out || !a || !b || !c || !d || !e)) return fail(-1, "null argument");
    if (!n) return 0;
    hipLaunchKernelGGL(k_rk_poly4, dim3(rk_grid(n)), dim3(256), 0, (hipStream_t)stream, out, a, b, c, d, e, x, (unsigned long long)n);
    LSL_CHECK_LAUNCH("lsl_rk_dense");
    return 0;
} catch (...) {
    return fail(-11, "unexpected C++ exception");
}

int lsl_rk_error_ratio(float *ratio, const float *y0, const float *y1, const float *const *k, const float *c, int32_t n_k, float atol, float rtol,
                       uint64_t n, void *scratch, void *stream) try {
    DeviceGuard dev_guard_((hipStream_t)stream);
    if (!ratio || !y0 || !y1 || !scratch) return fail(-1, "null argument");
    if (!n) return fail(-3, "empty state");
    RkTerms t;
    if (int rc = rk_terms(t, k, c, n_k)) return rc;
    const unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, LSL_RK_SCRATCH_BYTES / 4);  // (fixed for a given n: the sum's order is part of the contract)
    hipLaunchKernelGGL(k_rk_error_partial, dim3(grid), dim3(256), 0, (hipStream_t)stream, (float *)scratch, y0, y1, t, atol, rtol, (unsigned long long)n);
    hipLaunchKernelGGL(k_rk_error_final, dim3(1), dim3(256), 0, (hipStream_t)stream, ratio, (const float *)scratch, (int)grid, (unsigned long long)n);
    LSL_CHECK_LAUNCH("lsl_rk_error_ratio");
    return 0;
} catch (...) {
    return fail(-11, "unexpected C++ exception");
}

int lsl_debug_block(lsl_model *m, int32_t bi, const float *h_in, float *h_out, const float *mods, int32_t B, int32_t T, int32_t L,
                    void *workspace, size_t workspace_bytes, void *stream) try {
    DeviceGuard dev_guard_((hipStream_t)stream);
    if (!m || !m->has_weights) return fail(-2, "weights not set");
    if (bi < 0 || bi >= 2 * m->d.depth) return fail(-3, "block index out of range");
    const size_t need = carve(m, nullptr, B, T, L).bytes;
    if (!workspace || workspace_bytes < need) return fail(-4, "workspace too small: need %zu bytes", need);
    hipStream_t st = (hipStream_t)stream;
    const Workspace ws = carve(m, (char *)workspace, B, T, L);
    run_tables(m, ws, T, L, st);
    const size_t bytes = (size_t)B * T * L * m->d.hidden * 4;
    if (h_in != h_out) hipMemcpyAsync(h_out, h_in, bytes, hipMemcpyDeviceToDevice, st);
    return run_block(m, ws, bi, h_out, mods, m->MODW, B, T, L, st);
} catch (const std::bad_alloc &) {
    return fail(-5, "out of host memory");
} catch (...) {
    return fail(-11, "unexpected C++ exception");
}

int lsl_debug_taps(lsl_model *m, int32_t bi, const float *h_in, const float *mods, int32_t B, int32_t T, int32_t L, void *qkv_out,
                   void *z_out, void *workspace, size_t workspace_bytes, void *stream) try {
    DeviceGuard dev_guard_((hipStream_t)stream);
    if (!m || !m->has_weights) return fail(-2, "weights not set");
    if (bi < 0 || bi >= 2 * m->d.depth) return fail(-3, "block index out of range");
    if (!h_in || !mods || !qkv_out || !z_out || B <= 0 || T <= 0 || L <= 0) return fail(-3, "invalid arguments");
    const size_t need = carve(m, nullptr, B, T, L).bytes;
    if (!workspace || workspace_bytes < need) return fail(-4, "workspace too small: need %zu bytes", need);
    hipStream_t st = (hipStream_t)stream;
    const Workspace ws = carve(m, (char *)workspace, B, T, L);
    run_tables(m, ws, T, L, st);
    const size_t n = (size_t)B * T * L;
    hipMemcpyAsync(ws.h, h_in, n * m->d.hidden * 4, hipMemcpyDeviceToDevice, st);
    if (int rc = run_block(m, ws, bi, ws.h, mods, m->MODW, B, T, L, st, false, false, nullptr, true)) return rc;
    const bool temporal = bi & 1;
    if (qkv_planes_ok(m->d.head_dim_pad, m->d.hidden, m->d.heads, temporal ? T : L, temporal, linear1_ts_ok(m->d.head_dim_pad, m->d.hidden, m->F1, m->HHD, (int)n))) {
        const long chunks = (long)n * 3 * m->d.heads * (m->d.head_dim_pad / 8);  // the block left q / k / v as head-major planes: hand them out as token-major rows
        hipLaunchKernelGGL(k_planes_to_rows, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, st, (u16 *)qkv_out, ws.qkv, (int)n,
                           (int)((n + 255) & ~(size_t)255), 3 * m->d.heads, m->d.head_dim_pad);
    } else
        hipMemcpyAsync(qkv_out, ws.qkv, n * 3 * m->HHD * 2, hipMemcpyDeviceToDevice, st);
    hipMemcpyAsync(z_out, ws.z, n * m->K2 * 2, hipMemcpyDeviceToDevice, st);
    LSL_CHECK_LAUNCH("debug taps");
    return 0;
} catch (const std::bad_alloc &) {
    return fail(-5, "out of host memory");
} catch (...) {
    return fail(-11, "unexpected C++ exception");
}

int lsl_debug_mods(lsl_model *m, const float *t, const float *y, int32_t B, float *vec_out, float *mods_out, void *workspace,
                   size_t workspace_bytes, void *stream) try {
    DeviceGuard dev_guard_((hipStream_t)stream);
    if (!m || !m->has_weights) return fail(-2, "weights not set");
    const size_t need = carve(m, nullptr, B, 1, 1).bytes;
    if (!workspace || workspace_bytes < need) return fail(-4, "workspace too small: need %zu bytes", need);
    hipStream_t st = (hipStream_t)stream;
    const Workspace ws = carve(m, (char *)workspace, B, 1, 1);
    if (y) {
        if (int rc = run_yemb(m, ws, y, B, st)) return rc;
    }
    return run_mods(m, ws, t, 0.0f, y ? ws.yemb : nullptr, B, vec_out, mods_out, st);
} catch (const std::bad_alloc &) {
    return fail(-5, "out of host memory");
} catch (...) {
    return fail(-11, "unexpected C++ exception");
}

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

}  // extern "C"
