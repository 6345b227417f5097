// Host side of liblamslide_hip.so: the C ABI of include/lsl_api.h.  Enqueues the kernel sequence of one
// network evaluation (latent_si_v31.py:168-188) and of the sampler loops (integrators.py:67-78,103-120)
// on the caller's stream.  No allocation, no synchronisation, no host<->device copies.
// One translation unit: this file = the entry points of the sampling path (model handle, forward, fused sampler + opt-in hipGraph replay,
// noise, Runge-Kutta state arithmetic, debug taps); host_common / host_launch / host_eval.hip.h = what they enqueue; decode_host.hip.h +
// stage1_api.hip.h = the frozen stage-1 encode / decode beside the path.
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
#include "k_tail.hip.h"
#include "k_small.hip.h"
#include "k_resident.hip.h"
#ifdef LSL_EXPERIMENTS  // measured-and-rejected GEMM structures, built only by tools/build_experiments.sh (never in the product library)
#include "k_gemm_pp.hip.h"        // tools/experiments/ (on the include path of tools/build_experiments.sh only)
#include "k_gemm_drain.hip.h"
#endif

#include "host_common.hip.h"
#include "host_launch.hip.h"
#include "host_eval.hip.h"

}  // namespace (opened in host_common.hip.h)

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
    m->tail = tail_env() == 1 && tail_shape_ok(d.hidden, m->HHD, d.mlp_dim);
    m->ln_fuse = ln_fuse_env() == 1;
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

int lsl_model_set_attention_mode(lsl_model *m, int32_t mode) try {
    if (!m || (mode != 0 && mode != 1)) return fail(-1, "attention mode must be 0 (scaled_dot_product) or 1 (linear)");
    m->attention_linear = mode == 1;
    drop_graphs(m);
    return 0;
} catch (const std::bad_alloc &) {
    return fail(-5, "out of host memory");
} catch (...) {
    return fail(-11, "unexpected C++ exception");
}

int lsl_model_set_tail(lsl_model *m, int32_t on) try {
    if (!m || (on != 0 && on != 1)) return fail(-1, "tail must be 0 or 1");
    if (on && !tail_shape_ok(m->d.hidden, m->HHD, m->d.mlp_dim))
        return fail(-21, "no tail kernel for this model (hidden 256 with heads * head_dim_pad = 256, mlp_dim a multiple of 64; LSL_TAIL=0 disables it)");
    if (m->tail != (on == 1)) drop_graphs(m);
    m->tail = on == 1;
    return 0;
} catch (const std::bad_alloc &) {
    return fail(-5, "out of host memory");
} catch (...) {
    return fail(-11, "unexpected C++ exception");
}
int32_t lsl_model_tail(const lsl_model *m) { return m && m->tail ? 1 : 0; }
int lsl_model_set_ln_fuse(lsl_model *m, int32_t on) try {
    if (!m || (on != 0 && on != 1)) return fail(-1, "ln_fuse must be 0 or 1");
    if (on && ln_fuse_env() == 0) return fail(-21, "LayerNorm fusion is disabled (LSL_LN_FUSE=0)");
    if (m->ln_fuse != (on == 1)) drop_graphs(m);
    m->ln_fuse = on == 1;
    return 0;
} catch (const std::bad_alloc &) {
    return fail(-5, "out of host memory");
} catch (...) {
    return fail(-11, "unexpected C++ exception");
}
int32_t lsl_model_ln_fuse(const lsl_model *m) { return m && m->ln_fuse ? 1 : 0; }
const char *lsl_profile_kernel_name(const lsl_model *m) { return m ? m->prof.name : ""; }

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
    size_t need = carve(m, nullptr, default_chunk(m, B, T, L), T, L).bytes;
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
    // hipGraph replay.  LSL_GRAPH: 1 (default since round 6) for launch-bound calls (at most 64 Ki tokens per pass and 4096 launches) whose
    // arguments repeat, 2 for every call of at most 4096 launches, 0 off: the first appearance of an argument set runs eagerly (it also
    // initialises the per-kernel attributes), the second is captured, later ones are replayed.  Bit-identical to the eager path
    // (test_graph_replay_matches_eager_bits).  Measured on MI355X (tools/latency_small_batch.py, profiles/r06_small_launches.txt): a
    // 10-update pedestrian call (~600 launches) 3.38 -> 3.23 ms, md17_bench B = 1 (50 updates) 30.0 -> 29.2 ms: the floor of the small-batch
    // configs is the GPU-side cost of their dependent tiny kernels, replay removes the host-side gaps between them (0-4 %).
    static const int use_graph = env_int("LSL_GRAPH", 1);
    const int passes = (io->B + chunk - 1) / chunk;
    const long est_launches = (long)passes * n_steps * (8L * m->d.depth + 6);
    const bool launch_bound = (size_t)chunk * io->T * io->L <= 65536;
    if (use_graph && (launch_bound || use_graph >= 2) && m->prof.kernel < 0 && est_launches <= 4096) {
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
    const Workspace ws = carve(m, (char *)workspace, chunk, io->T, io->L);
    run_tables(m, ws, io->T, io->L, st);
    const size_t per = (size_t)io->T * io->L * m->d.in_dim;
    const size_t total = per * io->B;
    // no class conditioning: modulation tables per group of network records, one row each (run_mods_steps); recomputed only when a pass
    // crosses into another group (calls of at most mods_group records: once per pass).  LSL_MODS_GROUP=0: per evaluation.
    // (>= 2: at most that many records per group - the GPU suite crosses group boundaries with it)
    static const int group_on = env_int("LSL_MODS_GROUP", 1);
    const int G = (group_on && !io->y) ? (group_on >= 2 ? std::min(group_on, ws.mods_group) : ws.mods_group) : 0;
    std::vector<int> net_idx;
    std::vector<float> net_t;
    if (G) {
        net_idx.resize((size_t)n_steps);
        for (int s = 0; s < n_steps; ++s) {
            net_idx[s] = (int)net_t.size();
            if (!(steps[s].flags & LSL_STEP_NO_NETWORK)) net_t.push_back(steps[s].t);
        }
    }
    int cur_group = -1;
    for (int b0 = 0; b0 < io->B; b0 += chunk) {  // one pass = all records for `bc` trajectories
        const int bc = io->B - b0 < chunk ? io->B - b0 : chunk;
        const float *y = io->y ? io->y + (size_t)b0 * m->d.vec_in_dim : nullptr;
        if (int rc = prepare_pass(m, ws, io->x_cond + b0 * per, io->mask + (size_t)b0 * io->T * io->L, y, bc, io->T, io->L, st)) return rc;
        for (int s = 0; s < n_steps; ++s) {
            const lsl_step_ex &sp = steps[s];
            const float *nz = nullptr;
            if (sp.aw != 0.0f && noise) nz = noise + (size_t)sp.noise_index * total + b0 * per;
            float *tr = trace && sp.trace_index >= 0 ? trace + (size_t)sp.trace_index * total + b0 * per : nullptr;
            const float *saved = sp.as != 0.0f ? ws.saved : nullptr;        // (the pass's own copy: a pass runs all records for its trajectories)
            float *save_out = (sp.flags & LSL_STEP_SAVE) ? ws.saved : nullptr;
            if (sp.flags & LSL_STEP_NO_NETWORK) {
                const unsigned long long ne = (unsigned long long)bc * per;
                hipLaunchKernelGGL(k_state_affine, dim3((unsigned)std::min<unsigned long long>((ne + 255) / 256, 2048)), dim3(256), 0, st,
                                   io->x + b0 * per, ne, sp.ax, sp.aw, sp.as, nz, (unsigned long long)seed, (unsigned)sp.noise_index,
                                   (unsigned long long)(elem_offset + b0 * per), saved, save_out, tr);
                LSL_CHECK_LAUNCH("state update");
                continue;
            }
            const float *mods_ready = nullptr;
            if (G) {
                const int k = net_idx[s], g = k / G;
                if (cur_group != g) {
                    if (int rc = run_mods_steps(m, ws, net_t.data() + (size_t)g * G, std::min(G, (int)net_t.size() - g * G), st)) return rc;
                    cur_group = g;
                }
                mods_ready = ws.mods_all + (size_t)(k - g * G) * m->MODW;
            }
            if (int rc = run_eval(m, ws, io->x + b0 * per, nullptr, nullptr, sp.t, io->y != nullptr, bc, io->T, io->L, 1, sp.ax, sp.am, sp.aw,
                                  nz, seed, (unsigned)sp.noise_index, elem_offset + b0 * per, tr, st, sp.as, saved, save_out, mods_ready))
                return rc;
        }
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
    // (on the workspace's residual stream, like an evaluation: its rows are padded to whole tiles)
    hipMemcpyAsync(ws.h, h_in, bytes, hipMemcpyDeviceToDevice, st);
    if (int rc = run_block(m, ws, bi, ws.h, mods, m->MODW, B, T, L, st)) return rc;
    hipMemcpyAsync(h_out, ws.h, bytes, hipMemcpyDeviceToDevice, st);
    return 0;
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

#include "stage1_api.hip.h"

}  // extern "C"
