// Host side, part 1 of 4: error convention, the model handle, the scratch layout of a pass (carve), pass size, per-device one-time kernel
// attributes and the device guard.  Included by lsl_api.hip only (one translation unit; everything here has internal linkage).
#pragma once

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
    char name[96] = "";  // what the profiled class launched (lsl_profile_kernel_name)
    void label(int k, const char *fmt, ...) {
        if (k != kernel) return;
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(name, sizeof(name), fmt, ap);
        va_end(ap);
    }
    void clear() {
        for (auto e : ev) hipEventDestroy(e);
        ev.clear();
        kernel = -1; cap = used = 0;
        name[0] = 0;
    }
};

struct lsl_model {
    Profiler prof;
    lsl_model_desc d;
    lsl_weights w;
    std::vector<lsl_block_weights> blocks;
    bool has_weights = false;
    bool attention_linear = false;  // lsl_model_set_attention_mode: attention_linear (mmdit.py:58-72) instead of softmax attention
    int chunk = 0;
    int HHD, F1, K2, MODW;
    bool ln_fuse = false;  // LayerNorm + modulate inside linear1's activation load (k_lin1.hip.h LNF): lsl_model_set_ln_fuse, a property of the HANDLE
    bool tail = false;  // the back half of every sub-block runs k_tail (k_tail.hip.h): a property of the HANDLE (lsl_model_set_tail), never of the batch
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
    float *kmax2;  // [4 * depth]: bound of |k|^2 per attention block (k_rope_scaled), then of |q|^2 without the softmax pre-multiplier (same order): k_attention_stream's softmax shift
    u16 *w2p;  // linear2 weights of every sub-block in the fragment order of k_linear2_ws (k_lin2_pack, once per call), or NULL
    u16 *wtail;  // tail models: the weight stream of every sub-block (k_tail_pack, once per call); wtail_stride elements apart
    size_t wtail_stride;
    float2 *lnparts, *lnstat;  // ln_fuse handles: per-wave row statistics of k_linear2_ws<LNS> [D / 32][n rounded up to 256], per-token (rstd, -mean rstd)
    size_t bytes;
};

int env_int(const char *name, int dflt);
int tune_int(const char *name, int dflt);
bool linear2_ws_shape_ok(int D, int K2);
size_t tail_stream_bytes(const lsl_model *m);

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
    ws.kmax2 = (float *)take((size_t)4 * d.depth * sizeof(float));
    ws.cond_emb = (float *)take(n * D * 4);
    ws.h = (float *)take(align_up(n, 256) * D * 4);  // (whole 256-row tiles: k_tail reads the rows of its last wave tiles without clamping)
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
    ws.w2p = !m->tail && linear2_ws_shape_ok((int)D, m->K2) ? (u16 *)take((size_t)2 * d.depth * D * m->K2 * 2) : nullptr;
    ws.wtail_stride = m->tail ? tail_stream_bytes(m) / 2 : 0;
    ws.wtail = m->tail ? (u16 *)take((size_t)2 * d.depth * ws.wtail_stride * 2) : nullptr;
    ws.lnparts = m->ln_fuse && !m->tail ? (float2 *)take((size_t)(D / 32) * n_pad * sizeof(float2)) : nullptr;
    ws.lnstat = m->ln_fuse && !m->tail ? (float2 *)take(n_pad * sizeof(float2)) : nullptr;
    ws.bytes = off;
    return ws;
}

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

