// bf16 MFMA GEMM for linear1 / linear2 of a ParallelMLPAttentionV2 block (mmdit.py:240-249), with the
// surrounding element-wise work fused into the epilogue.
//
// Orientation: the kernel computes the TRANSPOSED product  Ct[f][n] = sum_k W[f][k] * X[n][k]
// (W = nn.Linear weight [out,in], X = activations [tokens, in]; both are k-contiguous, so both MFMA
// operands are plain 16-byte row reads).  With features on the accumulator rows and tokens on the
// lanes, one lane holds 16 of the 32 features of a tile for ONE token: per-head RMS norm and the RoPE
// pair rotation are in-register (+ one exchange with lane^32).
//
// Operand feed: LDS-DMA (global_load_lds_dwordx4) into an NS-deep ring of k-tiles; NS-1 k-tiles of loads
// stay in flight across the per-k-tile barrier (counted s_waitcnt vmcnt, raw s_barrier), no VGPRs or
// ds_write instructions are spent on staging.  One stage image = [BF rows of W | BT rows of X] x BK bf16; a
// wave instruction lands 1 KiB lane-linearly, so the read-side XOR swizzle of the 16-byte chunks (which makes
// the ds_read_b128 fragment reads conflict-free) is applied to the per-lane SOURCE address.
//
// Output: measured on MI355X (profiles/r01_*), per-lane stores straight from the accumulator layout touch 32
// cache lines per wave instruction (16-32 B each) and made the kernels store-transaction-bound, not MFMA- or
// load-bound.  The epilogue therefore transposes each wave's sub-tile through wave-private LDS (reusing the
// operand ring) and writes whole 128/256-byte row segments, 16 B per lane.
#pragma once
#include "common.hip.h"

struct GemmArgs {
    const u16 *W;  // [F][K] bf16
    const u16 *X;  // [N][K] bf16
    int F, N, K;
    int rows;     // persistent kernels whose epilogue is a row owner (linear2): 1 = walk token tiles and run all their feature tiles back to
                  // back (needed by the fused LayerNorm); 0 = flat tile list (small launches: more workgroups than token tiles)
    int stagger;  // persistent grids: initial delay of the second half of the workgroups, in units of 8128 cycles
    int probe;  // timing probes, honoured only in -DLSL_EXPERIMENTS builds (tools/), where they make results wrong: bit0 skip operand
                // loads, bit1 skip LDS reads + MFMAs, bit2 skip epilogue.  The product library ignores the field.
};

// byte offset of 16-byte chunk `chunk` of row `row` inside a [rows][BK] bf16 stage image (BK = 64 or 32)
template <int BK>
__device__ __forceinline__ int swz_bk(int row, int chunk) {
    if (BK == 64) return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
    return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4);
}

// wave-private output staging: rows of CH 16-byte chunks, chunk index XOR-swizzled by the row so that both the
// column-wise accumulator writes and the row-wise 16-byte reads spread over the banks
template <int CH>
__device__ __forceinline__ int stage_off(int row, int chunk) { return (row * CH + (chunk ^ (row & (CH - 1)))) << 4; }

template <int BF, int BT, int NWF, int NWT, int BK, int NS, bool PERSIST, class Epi>
struct GemmCfg {
    static constexpr int NW = NWF * NWT;
    static constexpr int WF = BF / NWF, WT = BT / NWT;
    static constexpr size_t ring_bytes = (size_t)NS * (BF + BT) * BK * 2;
    static constexpr size_t stage_bytes = (size_t)NW * Epi::template wave_stage_bytes<WF, WT>();
    // PERSIST: the next tile's first k-tiles stream into the ring while the epilogue of the current tile drains through the
    // staging area, so the two must not overlap: the staging sits BEHIND the ring - or, with a two-slot ring and an even number
    // of k-tiles (host check), in slot 1: the last k-tile of a tile lives there and is dead once the main loop has finished,
    // while the prefetch of the next tile's k-tile 0 goes to slot 0.  Not PERSIST: staging overlays the ring.
    static constexpr bool stage_in_slot1 = PERSIST && NS == 2 && stage_bytes <= ring_bytes / 2;
    static constexpr size_t stage_base = !PERSIST ? 0 : stage_in_slot1 ? ring_bytes / 2 : ring_bytes;
    static constexpr size_t lds_bytes =
        !PERSIST ? (ring_bytes > stage_bytes ? ring_bytes : stage_bytes) : stage_in_slot1 ? ring_bytes : ring_bytes + stage_bytes;
    static_assert(lds_bytes <= 163840, "LDS budget (160 KiB per workgroup)");
    // + the bias vector (Epi::lds_bias): nft * BF floats behind lds_bytes, sized by the launcher
};

template <int BF, int BT, int NWF, int NWT, int BK, int NS, bool PERSIST, class Epi>
__global__ void __launch_bounds__(NWF *NWT * 64, (NWF * NWT / 4 > 2 ? NWF * NWT / 4 : 2)) k_gemm_glds(GemmArgs g, Epi epi) {  // the whole workgroup resident, >= 2 waves/SIMD
    using Cfg = GemmCfg<BF, BT, NWF, NWT, BK, NS, PERSIST, Epi>;
    constexpr int NW = Cfg::NW, WF = Cfg::WF, WT = Cfg::WT;
    constexpr int MI = WF / 32, NJ = WT / 32;
    constexpr int ROWB = BK * 2;             // bytes per stage row
    constexpr int CPR = ROWB / 16;           // 16-byte chunks per row (8 or 4)
    constexpr int RPP = 1024 / ROWB;         // rows per 1 KiB LDS-DMA piece (8 or 16)
    constexpr int STAGE = (BF + BT) * ROWB;  // bytes
    constexpr int PIECES = (BF + BT) / RPP;  // 1 KiB pieces per stage
    constexpr int LPS = PIECES / NW;         // LDS-DMA instructions per wave per stage
    static_assert(PIECES % NW == 0 && BF % 16 == 0 && BT % 16 == 0 && NS >= 2 && NS <= 6 && (BK == 32 || BK == 64), "bad tiling");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wf = wave / NWT, wt = wave % NWT;
    const int r = lane & 31, hf = lane >> 5;

    // Persistent workgroups: virtual block v = blockIdx.x + s * gridDim.x walks the tile list; xcd_remap keeps the
    // feature tiles that share one token tile on one XCD and close in time (gridDim.x is a multiple of 8).
    const int ntt = (g.N + BT - 1) / BT, nft = (g.F + BF - 1) / BF, ntiles = ntt * nft;
    const int nk = g.K / BK;
    int f_base = 0, n_base = 0;
    // LDS-DMA sources: piece p = wave + i * NW covers stage rows p*RPP .. p*RPP+RPP-1; pieces below BF/RPP are W rows,
    // the rest X rows.  Both operand buffers are padded to whole tiles by the host (packing.py / carve()), so there is
    // no clamping and each operand needs one per-lane base pointer plus a uniform step between pieces.
    constexpr int WP = BF / RPP / NW, XP = BT / RPP / NW;  // pieces per wave per stage from W and from X
    static_assert(BF % (RPP * NW) == 0 && BT % (RPP * NW) == 0 && WP + XP == LPS, "stage split");
    const int lrow = wave * RPP + lane / CPR;
    const int lchunk = (swz_bk<BK>(lrow, lane % CPR) - lrow * ROWB) >> 4;  // logical chunk stored at this lane's slot
    const size_t piece_step = (size_t)NW * RPP * g.K;                       // elements between a wave's pieces
    const u16 *srcW = nullptr, *srcX = nullptr;
    // Row-owner walk (persistent epilogues that finish whole token rows, Epi::row_owner): a workgroup takes token tiles
    // blockIdx.x, blockIdx.x + gridDim.x, ... and runs ALL feature tiles of a token tile back to back, so that after the last one it
    // holds complete rows of the output (linear2: the updated residual rows, ready for the next sub-block's LayerNorm).
    const bool ROWS = PERSIST && Epi::row_owner && g.rows;
    const int n_walk = ROWS ? ((ntt > (int)blockIdx.x) ? (ntt - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0) * nft : ntiles;
    auto set_tile = [&](int v) {
        int tile_f, tile_n;
        if (ROWS) {
            tile_n = blockIdx.x + (v / nft) * gridDim.x;
            tile_f = v % nft;
        } else {
            const int tile = xcd_remap(v, ntiles);
            tile_f = tile % nft;
            tile_n = tile / nft;
        }
        f_base = tile_f * BF;
        n_base = tile_n * BT;
        srcW = g.W + (size_t)(f_base + lrow) * g.K + lchunk * 8;
        srcX = g.X + (size_t)(n_base + lrow) * g.K + lchunk * 8;
    };
    auto issue = [&](int kt, int buf) {
        if (LSL_PROBE(g.probe, 1)) return;
#pragma unroll
        for (int i = 0; i < WP; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(srcW + i * piece_step + kt * BK),
                                             (LDS_PTR(void))(smem + buf * STAGE + (wave + i * NW) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < XP; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(srcX + i * piece_step + kt * BK),
                                             (LDS_PTR(void))(smem + buf * STAGE + (wave + (WP + i) * NW) * 1024), 16, 0, 0);
    };
    // fragment read offsets: the swizzle term depends only on (lane, k sub-step) because every tile starts on a
    // multiple of 16 rows; per-tile row offsets are compile-time immediates
    constexpr int KSUB = BK / 16;
    int offA[KSUB], offB[KSUB];
#pragma unroll
    for (int ks = 0; ks < KSUB; ++ks) {
        offA[ks] = swz_bk<BK>(r, 2 * ks + hf) + wf * WF * ROWB;
        offB[ks] = swz_bk<BK>(r, 2 * ks + hf) + (BF + wt * WT) * ROWB;
    }
    auto prologue = [&]() {
#pragma unroll
        for (int s = 0; s < NS - 1; ++s)
            if (s < nk) issue(s, s);
    };
    char *stage = smem + Cfg::stage_base + (size_t)wave * Epi::template wave_stage_bytes<WF, WT>();

    // Co-resident workgroups run the same program on equal-sized tiles and would stay in lockstep (all in their main
    // loops together, all in their epilogues together).  g.stagger delays the second half of a persistent grid once, by
    // about half a tile, so one workgroup's MFMA phase lines up with its neighbour's epilogue (memory) phase.
    if (PERSIST && g.stagger > 0 && blockIdx.x >= gridDim.x / 2)
        for (int i = 0; i < g.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    if (PERSIST && g.stagger < 0) {  // same, keyed on the hardware wave slot of wave 0 (HW_REG_HW_ID[3:0]) instead of the block index
        if (tid == 0) *reinterpret_cast<volatile int *>(smem) = __builtin_amdgcn_s_getreg((3 << 11) | 4) & 1;
        __syncthreads();
        const int odd = *reinterpret_cast<volatile int *>(smem);
        __syncthreads();
        if (odd)
            for (int i = 0; i < -g.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    }
    // Epilogues with a per-feature bias (linear1) keep the whole bias vector in LDS behind the ring / staging for the lifetime of the
    // workgroup (F1 floats: 10 KiB for D = 512) and start every tile's accumulators FROM it: the bias costs no VALU instruction and no
    // global-load latency in the epilogue, and replaces the zero fill.
    const float *bias_lds = reinterpret_cast<const float *>(smem + Cfg::lds_bytes);
    if constexpr (Epi::lds_bias) {
        const int fpad = nft * BF;  // (the bias buffer is padded to whole 256-feature tiles by the host)
        for (int i = tid * 4; i < fpad; i += NW * 64 * 4)
            *reinterpret_cast<float4 *>(smem + Cfg::lds_bytes + (size_t)i * 4) = *reinterpret_cast<const float4 *>(epi.bias + i);
        __syncthreads();
    }
    const int v0 = ROWS ? 0 : (int)blockIdx.x, v_step = ROWS ? 1 : PERSIST ? (int)gridDim.x : ntiles;
    if (v0 >= n_walk) return;
    set_tile(v0);
    prologue();
    bool prev_full = false;  // the tile whose epilogue ran last was a full one (all its stores were issued)
    for (int v = v0; v < n_walk; v += v_step) {
    const bool first_tile = v == v0;
    f32x16 acc[MI][NJ];
    if constexpr (Epi::lds_bias) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 b = *reinterpret_cast<const float4 *>(bias_lds + f_base + wf * WF + i * 32 + 8 * q4 + 4 * hf);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    acc[i][j][4 * q4] = b.x;
                    acc[i][j][4 * q4 + 1] = b.y;
                    acc[i][j][4 * q4 + 2] = b.z;
                    acc[i][j][4 * q4 + 3] = b.w;
                }
            }
    } else {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    }

    for (int kt = 0; kt < nk; ++kt) {
        // k-tiles after this one that are already in flight: min(nk - 1 - kt, NS - 2); wait for everything older.
        // On later tiles the previous epilogue's stores are younger than this tile's first k-tiles in the vmcnt
        // queue, so the first wait drains everything (the k-tiles were issued a whole epilogue ago).
        const int ahead = (kt == 0 && !first_tile) ? 0 : min(nk - 1 - kt, NS - 2);
        if (PERSIST && kt == 0 && !first_tile && prev_full) {  // the previous tile's last stores may stay in flight
            constexpr int LB = Epi::template min_store_ops<MI, NJ>() < 48 ? Epi::template min_store_ops<MI, NJ>() : 48;
            wait_vmcnt<LB>();
        } else
        switch (ahead) {
            case 0: wait_vmcnt<0>(); break;
            case 1: wait_vmcnt<LPS>(); break;
            case 2: wait_vmcnt<2 * LPS>(); break;
            case 3: wait_vmcnt<3 * LPS>(); break;
            default: wait_vmcnt<4 * LPS>(); break;
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + NS - 1 < nk) issue(kt + NS - 1, (kt + NS - 1) % NS);
        const char *sb = smem + (kt % NS) * STAGE;
        // Fragments are double-buffered across the 16-deep k sub-steps: the ds_read_b128 of sub-step ks+1 are
        // issued before the MFMAs of sub-step ks (hipcc otherwise emits read / lgkmcnt(0) / 2 MFMAs chains).
        auto load_frags = [&](int ks, bf16x8(&a)[MI], bf16x8(&b)[NJ]) {
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = as_bf16x8(*reinterpret_cast<const u32x4 *>(sb + offA[ks] + i * 32 * ROWB));
#pragma unroll
            for (int j = 0; j < NJ; ++j) b[j] = as_bf16x8(*reinterpret_cast<const u32x4 *>(sb + offB[ks] + j * 32 * ROWB));
        };
        auto mfma_all = [&](const bf16x8(&a)[MI], const bf16x8(&b)[NJ]) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
        };
        constexpr int KS = BK / 16;
        if (LSL_PROBE(g.probe, 2)) continue;
        if (BK == 32) {
            bf16x8 a0[MI], b0[NJ], a1[MI], b1[NJ];
            load_frags(0, a0, b0);
            load_frags(1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_all(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_all(a1, b1);
        } else {  // BK = 64: one fragment set (the 256-VGPR budget of an 8-wave workgroup has no room for two)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 a0[MI], b0[NJ];
                load_frags(ks, a0, b0);
                mfma_all(a0, b0);
            }
        }
    }

    __syncthreads();  // every wave is done with the operand ring
    const int fw = f_base + wf * WF, nw = n_base + wt * WT;
    prev_full = f_base + BF <= g.F && n_base + BT <= g.N && !LSL_PROBE(g.probe, 4);
    const int n_tile = n_base;
    const bool rows_done = ROWS && (v % nft) == nft - 1;  // this was the last feature tile of the token tile
    if (PERSIST && v + v_step < n_walk) {  // stream the next tile's first k-tiles while this tile's epilogue runs
        set_tile(v + v_step);
        prologue();
    }
    if (!LSL_PROBE(g.probe, 4)) epi.template run<MI, NJ>(acc, stage, fw, nw, lane, g.F, g.N);
    if constexpr (PERSIST && Epi::row_owner) {
        if (rows_done && !LSL_PROBE(g.probe, 4)) epi.finish_rows(n_tile, BT, tid, NW, g.N);
    }
    }
}

// ---------------------------------------------------------------------------------------------------
// linear1 epilogue: + bias; q/k heads: RMS norm * scale, RoPE (q additionally * softmax scale * log2 e);
// v: as is; mlp: erf-GELU.  Output bf16:  qkv[n][0 .. 3*HHD)  and  z[n][HHD .. HHD+M).
// (mmdit.py:241-248, 129-148, 85-90, 11-18)
//
// Round-2 form.  All arithmetic runs in the ACCUMULATOR layout: a lane holds one token (column lane & 31) and, per 32 x 32 tile, 16 of
// the tile's 32 features in 4 groups of 4 consecutive ones (rows 8 g + 4 hf .. + 3), so
//   * a tile lies inside one section (q | k | v | mlp: sections start on multiples of 32) -> the branch is wave-uniform,
//   * RoPE pairs (2p, 2p+1) are lane-local; the QK-norm scales are folded into the rotation table once per call
//     (k_rope_scaled: (c s0, sn s1, sn s0, c s1) per pair), which a lane loads once per token column and keeps across head tiles,
//   * the head's sum of squares is 16 (8) FMAs plus ONE exchange with lane ^ 32 (v_permlane32_swap),
//   * GELU is element-wise.
// Only the finished bf16 values cross LDS (round 1 transposed the raw fp32 accumulators and did the arithmetic row-wise: twice the
// LDS bytes and redundant per-row table loads): per slab of 64 features x 32 tokens a lane writes its 4 + 4 packed 8-byte groups
// into a 4 KiB wave-private image [token][64 features] (16-byte chunks XOR-swizzled by the row) and the wave reads it back row-wise,
// 16 bytes per lane = whole 128-byte row segments, 8 token rows per store instruction.
template <int HDP>
struct EpiLinear1 {
    const float *bias;     // [F1 rounded up to 256]
    const float *qs, *ks;  // [HDP]                                   (piece form below: experiments only)
    const float2 *rope;    // [n_pos][HDP/2] (cos, sin)               (piece form below: experiments only)
    const float4 *rope_q, *rope_k;  // [n_pos][HDP/2] (c s0, sn s1, sn s0, c s1) with the query / key norm scales folded in
    u16 *qkv;              // [N][3*HHD]
    u16 *z;                // [N][HHD + M]
    int HHD, M;
    int pos_div, pos_mod;  // position of token n inside its sequence: (n / pos_div) % pos_mod
    unsigned div_magic, mod_magic;  // floor(2^32 / d) + 1 for d = pos_div, pos_mod (0 when d == 1): n / d == umulhi(n, magic) for n * d < 2^32
    float inv_hd;          // 1 / true head_dim
    float q_premul;        // head_dim^-0.5 * log2(e), folded into q for the exp2-based softmax
    int probe;             // bit5 (set by the launcher): streaming stores.  bits 3 / 4 (skip math / skip stores) are timing probes that
                           // exist only in -DLSL_EXPERIMENTS builds

    static constexpr bool lds_bias = true;  // the kernel starts the accumulators from the bias (kept in LDS); run() must not add it again
    static constexpr bool row_owner = false;
    template <int WF, int WT>
    static constexpr size_t wave_stage_bytes() { return (size_t)32 * 64 * 2; }
    // global stores a wave issues for a FULL tile (a lower bound of its vector-memory instructions in the epilogue): what a counted
    // s_waitcnt may leave in flight when a persistent workgroup only needs the LDS-DMA loads it issued BEFORE the epilogue
    template <int MI, int NJ>
    static constexpr int min_store_ops() { return MI * NJ * 2; }

    static __device__ __forceinline__ int soff(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

    template <int MI, int NJ>
    __device__ __forceinline__ void run(f32x16 (&acc)[MI][NJ], char *stage, int f_wave, int n_wave, int lane, int F, int N) const {
        static_assert(MI % 2 == 0, "feature slabs are 64 wide");
        constexpr int NCO = HDP == 32 ? 8 : 4;  // rotation pairs a lane owns per head
        // opaque copy of the lane id: without it hipcc hoists every per-lane 64-bit address of this epilogue out of the persistent tile
        // loop and keeps them alive across the main loop - as scratch spills (15 pointer pairs)
        asm volatile("" : "+v"(lane));
        const int r = lane & 31, hf = lane >> 5;
        const int tr = lane >> 3, c = lane & 7;  // read-back: token row within a group of 8, 8-feature chunk
#pragma unroll
        for (int i0 = 0; i0 < MI; i0 += 2) {
            const int fs = f_wave + i0 * 32;  // first feature of the slab
            if (fs >= F) continue;            // wave-uniform
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const unsigned nn = (unsigned)min(n_wave + j * 32 + r, N - 1);
                const unsigned n1 = div_magic ? __umulhi(nn, div_magic) : nn;
                const unsigned pos = mod_magic ? n1 - __umulhi(n1, mod_magic) * (unsigned)pos_mod : 0u;
                int csec = -1;
                float4 co[NCO];
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    const int ft = fs + ii * 32;
                    if (ft >= F) continue;  // wave-uniform (F is a multiple of 32)
                    const int sec_t = (ft >= HHD) + (ft >= 2 * HHD) + (ft >= 3 * HHD);  // 0 q, 1 k, 2 v, 3 mlp: wave-uniform
                    const int sec = LSL_PROBE(probe, 8) ? 2 : sec_t;
                    __builtin_amdgcn_sched_barrier(0);  // one tile's temporaries live at a time (256-VGPR budget beside 128 accumulators)
                    const f32x16 &a = acc[i0 + ii][j];  // (bias included: the kernel started the accumulators from it)
                    float x[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) x[e] = a[e];
                    if (sec < 2) {
                        if (sec != csec) {  // (a wave tile usually lies in one section: loaded once per token column)
                            const float4 *tab = (sec == 0 ? rope_q : rope_k) + (size_t)pos * (HDP / 2) + 2 * hf;
#pragma unroll
                            for (int k = 0; k < NCO; ++k) co[k] = tab[4 * (k >> 1) + (k & 1)];
                            csec = sec;
                        }
                        const float post = sec == 0 ? q_premul : 1.0f;
                        if (HDP == 32) {
                            float ss = 0.0f;
#pragma unroll
                            for (int e = 0; e < 16; ++e) ss = fmaf(x[e], x[e], ss);
                            ss = half_pair_sum(ss);
                            const float rr = rsqrtf(fmaf(ss, inv_hd, 1e-6f)) * post;
#pragma unroll
                            for (int k = 0; k < 8; ++k) {
                                const float x0 = x[2 * k], x1 = x[2 * k + 1];
                                x[2 * k] = rr * fmaf(co[k].x, x0, -co[k].y * x1);
                                x[2 * k + 1] = rr * fmaf(co[k].z, x0, co[k].w * x1);
                            }
                        } else {  // two 16-wide heads per tile: rows 0-15 and 16-31
                            float sa = 0.0f, sb = 0.0f;
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                sa = fmaf(x[e], x[e], sa);
                                sb = fmaf(x[8 + e], x[8 + e], sb);
                            }
                            sa = half_pair_sum(sa);
                            sb = half_pair_sum(sb);
                            const float ra = rsqrtf(fmaf(sa, inv_hd, 1e-6f)) * post, rb = rsqrtf(fmaf(sb, inv_hd, 1e-6f)) * post;
#pragma unroll
                            for (int k = 0; k < 8; ++k) {
                                const float x0 = x[2 * k], x1 = x[2 * k + 1], rr = k < 4 ? ra : rb;
                                const float4 cf = co[k & 3];
                                x[2 * k] = rr * fmaf(cf.x, x0, -cf.y * x1);
                                x[2 * k + 1] = rr * fmaf(cf.z, x0, cf.w * x1);
                            }
                        }
                    } else if (sec >= 3) {  // groups of 4: bounds the temporaries of the independent GELU chains (16 at once spill)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) x[4 * g + e] = gelu_fast(x[4 * g + e]);
                            const u32x2 pk = {pack2(x[4 * g], x[4 * g + 1]), pack2(x[4 * g + 2], x[4 * g + 3])};
                            *reinterpret_cast<u32x2 *>(stage + soff(r, 4 * ii + g) + 8 * hf) = pk;
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    if (sec < 3) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const u32x2 pk = {pack2(x[4 * g], x[4 * g + 1]), pack2(x[4 * g + 2], x[4 * g + 3])};
                            *reinterpret_cast<u32x2 *>(stage + soff(r, 4 * ii + g) + 8 * hf) = pk;
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                // read back row-wise: 8 lanes cover the 128-byte slab of one token row; 8 token rows per instruction
                const int f = fs + 8 * c;
                const bool f_ok = f < F;
                const bool to_qkv = f < 3 * HHD;
#pragma unroll
                for (int row0 = 0; row0 < 32; row0 += 8) {
                    const int row = row0 + tr, n = n_wave + j * 32 + row;
                    const u32x4 pk = *reinterpret_cast<const u32x4 *>(stage + soff(row, c));
                    if (f_ok && n < N && !LSL_PROBE(probe, 16)) {
                        u16 *dst = to_qkv ? qkv + (size_t)n * (3 * HHD) + f : z + (size_t)n * (HHD + M) + (f - 2 * HHD);
                        store16(dst, pk, probe & 32);
                    }
                }
            }
        }
    }
#ifdef LSL_EXPERIMENTS  // (the piece form of this epilogue is used by rejected GEMM structures only: tools/build_experiments.sh)
    // ---- the same epilogue as 16 software-pipelined pieces of a 128-feature x 64-token wave tile (k_gemm_pp.hip.h), where the
    // epilogue runs on ONE wave per SIMD and nothing but the wave's own instruction stream can hide latency.
    // Piece C = rows 16 (C & 1) .. +15 of the 32 x 32 accumulator tile [i = C >> 2][j = (C >> 1) & 1].  The tile is transposed
    // through 4 KiB of wave-private LDS (fp32, layout = swz_bk<64>: 32 token rows x 8 chunks of 4 features); in phase B a lane
    // owns 8 consecutive features of one token (4 lanes = one 32-wide head or two 16-wide ones: the sum of squares is 1-2
    // DPP quad exchanges), 16 token rows per instruction.  Everything piece C+1 needs from memory (its LDS rows, its RoPE
    // row, bias / scales when the feature tile changes) is requested while piece C computes: `Pipe` carries it.
    struct Pipe {
        float4 lo, hi;          // staged sums of the 8 features
        float4 c0, c1;          // RoPE (cos, sin) of the 4 pairs
    };
    static constexpr int pieces = 16;
    static constexpr size_t pp_stage_bytes = 4096;
    // lower bound of the vector-memory instructions piece<C> issues (bias 2 loads, [q/k scales 2,] next piece's RoPE row 2, store 1):
    // what a counted s_waitcnt may leave outstanding when it only wants the LDS-DMA loads issued before the piece
    static constexpr int piece_ops(bool last) { return last ? 3 : 5; }

    template <int C, int MI>
    __device__ __forceinline__ void fetch(f32x16 (&acc)[MI][2], char *stage, int f_wave, int n_wave, int lane, int F, int N, Pipe &k) const {
        constexpr int i = C >> 2, j = (C >> 1) & 1, hb = C & 1;
        const int r = lane & 31, hf = lane >> 5, tr = lane >> 2, c = lane & 3;
        const int f = f_wave + i * 32 + 8 * c;
        const int sec = LSL_PROBE(probe, 8) ? 2 : (f >= HHD) + (f >= 2 * HHD) + (f >= 3 * HHD);
        const int d = f & (HDP - 1);
        // every load below is unconditional (addresses clamped into the tables, unused values ignored by piece<C>): a load
        // inside a divergent branch gets its s_waitcnt at the end of that branch, i.e. right here instead of one piece later.
        // (bias and q/k scales are NOT carried in the pipe: 16 VGPRs that the drain kernel does not have; piece<C> reads them from
        // L1 where it uses them)
        {
            const unsigned nn = (unsigned)min(n_wave + j * 32 + 16 * hb + tr, N - 1);
            const unsigned n1 = div_magic ? __umulhi(nn, div_magic) : nn;
            const unsigned pos = mod_magic ? n1 - __umulhi(n1, mod_magic) * (unsigned)pos_mod : 0u;
            const float2 *tab = rope + (size_t)pos * (HDP / 2) + (d >> 1);
            k.c0 = *reinterpret_cast<const float4 *>(tab);
            k.c1 = *reinterpret_cast<const float4 *>(tab + 2);
        }
        if (hb == 0) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const f32x16 &a = acc[i][j];
                *reinterpret_cast<float4 *>(stage + swz_bk<64>(r, 2 * q4 + hf)) = make_float4(a[4 * q4], a[4 * q4 + 1], a[4 * q4 + 2], a[4 * q4 + 3]);
            }
        }
        const int row = 16 * hb + tr;
        k.lo = *reinterpret_cast<const float4 *>(stage + swz_bk<64>(row, 2 * c));
        k.hi = *reinterpret_cast<const float4 *>(stage + swz_bk<64>(row, 2 * c + 1));
    }

    template <int C, int MI>
    __device__ __forceinline__ void piece(f32x16 (&acc)[MI][2], char *stage, int f_wave, int n_wave, int lane, int F, int N, Pipe &k) const {
        constexpr int i = C >> 2, j = (C >> 1) & 1, hb = C & 1;
        const int tr = lane >> 2, c = lane & 3;
        const int f = f_wave + i * 32 + 8 * c;
        const bool f_ok = f < F;
        const int sec_out = (f >= HHD) + (f >= 2 * HHD) + (f >= 3 * HHD);  // 0 q, 1 k, 2 v, 3 mlp
        const int sec = LSL_PROBE(probe, 8) ? 2 : sec_out;
        const int n = n_wave + j * 32 + 16 * hb + tr;
        const int fc = min(f, F - 8), d = f & (HDP - 1);
        const float4 b0 = *reinterpret_cast<const float4 *>(bias + fc), b1 = *reinterpret_cast<const float4 *>(bias + fc + 4);
        float v[8] = {k.lo.x + b0.x, k.lo.y + b0.y, k.lo.z + b0.z, k.lo.w + b0.w, k.hi.x + b1.x, k.hi.y + b1.y, k.hi.z + b1.z, k.hi.w + b1.w};
        const float4 c0 = k.c0, c1 = k.c1;
        if (sec < 2) {
            const float *sc = (sec == 1 ? ks : qs) + d;
            const float4 s0 = *reinterpret_cast<const float4 *>(sc), s1 = *reinterpret_cast<const float4 *>(sc + 4);
            const float post = sec == 0 ? q_premul : 1.0f;
            float ss = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) ss = fmaf(v[e], v[e], ss);
            ss += quad_xor1(ss);
            if (HDP == 32) ss += quad_xor2(ss);
            const float rr = rsqrtf(fmaf(ss, inv_hd, 1e-6f)) * post;
            const float x0 = v[0] * rr * s0.x, x1 = v[1] * rr * s0.y, x2 = v[2] * rr * s0.z, x3 = v[3] * rr * s0.w;
            const float x4 = v[4] * rr * s1.x, x5 = v[5] * rr * s1.y, x6 = v[6] * rr * s1.z, x7 = v[7] * rr * s1.w;
            v[0] = c0.x * x0 - c0.y * x1; v[1] = c0.y * x0 + c0.x * x1;
            v[2] = c0.z * x2 - c0.w * x3; v[3] = c0.w * x2 + c0.z * x3;
            v[4] = c1.x * x4 - c1.y * x5; v[5] = c1.y * x4 + c1.x * x5;
            v[6] = c1.z * x6 - c1.w * x7; v[7] = c1.w * x6 + c1.z * x7;
        } else if (sec >= 3) {  // two groups of 4: bounds the temporaries (the role runs beside 128 accumulator VGPRs)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_fast(v[e]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 4; e < 8; ++e) v[e] = gelu_fast(v[e]);
        }
        // the next piece's rows / tables are requested once this piece's arithmetic no longer needs the current ones (no copies
        // of the pipe registers); they have the rest of the interval to arrive
        if (C + 1 < 4 * MI) fetch<(C + 1) % (4 * MI)>(acc, stage, f_wave, n_wave, lane, F, N, k);
        if (f_ok && n < N && !LSL_PROBE(probe, 16)) {
            const u32x4 pk = {pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
            u16 *dst = sec_out < 3 ? qkv + (size_t)n * (3 * HHD) + f : z + (size_t)n * (HHD + M) + (f - 2 * HHD);
            *reinterpret_cast<u32x4 *>(dst) = pk;  // 64 contiguous bytes per token row: half lines, no streaming stores (they need whole lines)
        }
    }
#endif
};

// linear2 epilogue: h[n][f] += gate[b][f] * (acc + bias[f])   (latent_si_v31.py:53,60; fp32 residual)
struct EpiLinear2 {
    const float *bias;  // [D]
    const float *gate;  // mods + gate offset, row stride mod_stride
    float *h;           // [N][D]
    int D, mod_stride, tokens_per_traj;
    int probe;  // bit5: streaming stores (set by the launcher)
    unsigned tpt_magic;  // floor(2^32 / tokens_per_traj) + 1 (0 when tokens_per_traj == 1): n / tokens_per_traj by multiply-high
    // Fused LayerNorm + modulate of the NEXT sub-block (persistent row-owner kernels, D a multiple of 256): once a workgroup has run
    // every feature tile of a token tile it owns the updated rows h[n][0..D) and writes a = bf16(LN_{1e-6}(h) (1 + scale) + shift), the
    // A operand of the next linear1, straight away (rows re-read from the XCD's L2).  MEASURED AND REJECTED (round 2, cfg 2): bit-identical
    // to the standalone kernel, but the standalone LayerNorm (55 ms per step, HBM-bound with every CU streaming) only moves into the
    // workgroup's own timeline, behind a full drain of its residual stores and at one workgroup's memory-level parallelism: linear2
    // 197 -> 286 ms for 55 -> 7.5 ms of LayerNorm.  Compiled only with -DLSL_EXPERIMENTS (LSL_LN_FUSE=1); the product never takes it.
    u16 *ln_out;                        // [N][D] bf16, or NULL
    const float *ln_shift, *ln_scale;   // next sub-block's modulation rows (row stride mod_stride)

    static constexpr bool lds_bias = false;
#ifdef LSL_EXPERIMENTS
    static constexpr bool row_owner = true;
#else
    static constexpr bool row_owner = false;
#endif

    // same arithmetic, in the same order, as k_ln_modulate_v4 (k_small.hip.h): the fused and the standalone form give identical bits
    __device__ __forceinline__ void finish_rows(int n_tile, int bt, int tid, int nwaves, int N) const {
#ifdef LSL_EXPERIMENTS
        if (!ln_out) return;  // uniform
        // every wave's residual stores have to be in L2 before any wave re-reads the rows: drain, then the workgroup barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int Q = D >> 8;  // 256-feature quarters per row: 1 (D = 256) or 2 (D = 512)
        const float invD = 1.0f / (float)D;
        constexpr int G = 8;   // rows in flight per wave (the accumulators are dead here: registers are plentiful, latency is not)
        const L2Reader hl2(h);  // (a pass holds at most 2^18 tokens x 2 KiB: offsets fit 32 bits)
        for (int r0 = wave * G; r0 < bt; r0 += nwaves * G) {
            float4 v[G][2];
#pragma unroll
            for (int u = 0; u < G; ++u) {
                const int n = min(n_tile + r0 + u, N - 1);
#pragma unroll
                for (int k = 0; k < 2; ++k)  // L1 still holds the rows as they were BEFORE the update: read past it (nt loads are L2-served)
                    v[u][k] = k < Q ? hl2.load16((unsigned)(((size_t)n * D + 4 * lane + 256 * k) * 4)) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < G; ++u) {
                const int n = n_tile + r0 + u;
                if (n >= N) break;  // uniform
                float s = 0.0f;
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (k < Q) s += (v[u][k].x + v[u][k].y) + (v[u][k].z + v[u][k].w);
                const float mean = wave_sum(s) * invD;
                float q = 0.0f;
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (k < Q) {
                        const float dx = v[u][k].x - mean, dy = v[u][k].y - mean, dz = v[u][k].z - mean, dw = v[u][k].w - mean;
                        q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
                    }
                const float rstd = rsqrtf(wave_sum(q) * invD + 1e-6f);
                const unsigned traj = tpt_magic ? __umulhi((unsigned)n, tpt_magic) : (unsigned)n;
                const size_t mo = (size_t)traj * mod_stride;
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (k < Q) {
                        const int d = 4 * lane + 256 * k;
                        const float4 sc = *reinterpret_cast<const float4 *>(ln_scale + mo + d);
                        const float4 sf = *reinterpret_cast<const float4 *>(ln_shift + mo + d);
                        const u32x2 pk = {pack2((v[u][k].x - mean) * rstd * (1.0f + sc.x) + sf.x, (v[u][k].y - mean) * rstd * (1.0f + sc.y) + sf.y),
                                          pack2((v[u][k].z - mean) * rstd * (1.0f + sc.z) + sf.z, (v[u][k].w - mean) * rstd * (1.0f + sc.w) + sf.w)};
                        store8(ln_out + (size_t)n * D + d, pk, probe & 32);
                    }
            }
        }
#endif
    }

    template <int WF, int WT>
    static constexpr size_t wave_stage_bytes() { return (size_t)32 * WT * 4; }  // one 32-feature slab of the wave tile
    template <int MI, int NJ>
    static constexpr int min_store_ops() { return MI * NJ * 4; }

    template <int MI, int NJ>
    __device__ __forceinline__ void run(f32x16 (&acc)[MI][NJ], char *stage, int f_wave, int n_wave, int lane, int F, int N) const {
        constexpr int WT = NJ * 32, CH = 8;  // 32 fp32 features = 8 chunks of 16 B per staged token row
        const int r = lane & 31, hf = lane >> 5;
        const int chunk = lane & 7;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int f0 = f_wave + i * 32;
            if (f0 >= F) continue;  // wave-uniform
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const float4 v = make_float4(acc[i][j][4 * q4], acc[i][j][4 * q4 + 1], acc[i][j][4 * q4 + 2], acc[i][j][4 * q4 + 3]);
                    *reinterpret_cast<float4 *>(stage + stage_off<CH>(j * 32 + r, 2 * q4 + hf)) = v;
                }
            __builtin_amdgcn_sched_barrier(0);  // one slab's temporaries live at a time (VGPR budget)
            // 8 lanes cover the 128-byte slab of one token row; 8 token rows per instruction
            const int f = f0 + 4 * chunk;
            const float4 b = *reinterpret_cast<const float4 *>(bias + f);
#pragma unroll 2
            for (int row0 = 0; row0 < WT; row0 += 8) {
                const int row = row0 + (lane >> 3), n = n_wave + row;
                const float4 a = *reinterpret_cast<const float4 *>(stage + stage_off<CH>(row, chunk));
                if (n < N) {
                    const float4 gt = *reinterpret_cast<const float4 *>(gate + (size_t)((unsigned)n / (unsigned)tokens_per_traj) * mod_stride + f);
                    float *hp = h + (size_t)n * D + f;
                    float4 hv = *reinterpret_cast<float4 *>(hp);
                    hv.x = fmaf(gt.x, a.x + b.x, hv.x);
                    hv.y = fmaf(gt.y, a.y + b.y, hv.y);
                    hv.z = fmaf(gt.z, a.z + b.z, hv.z);
                    hv.w = fmaf(gt.w, a.w + b.w, hv.w);
                    store16(hp, __builtin_bit_cast(u32x4, hv), probe & 32);
                }
            }
        }
    }
    // ---- 16 software-pipelined pieces of a 128-feature x 64-token wave tile (k_gemm_pp.hip.h; see EpiLinear1::Pipe): piece C =
    // rows 16 (C & 1) .. +15 of the accumulator tile [i = C >> 2][j = (C >> 1) & 1], staged through 4 KiB of LDS; a lane owns 4
    // features of one token, 8 token rows per instruction, two instructions per piece.  The h rows of piece C+1 are requested
    // while piece C computes.
    struct Pipe {
        float4 a[2], hv[2], gt[2], b;
    };
    static constexpr int pieces = 16;
    static constexpr size_t pp_stage_bytes = 4096;
    static constexpr int piece_ops(bool last) { return last ? 2 : 6; }  // next piece's gate + h rows 4 loads, 2 stores

    template <int C, int MI>
    __device__ __forceinline__ void fetch(f32x16 (&acc)[MI][2], char *stage, int f_wave, int n_wave, int lane, int F, int N, Pipe &k) const {
        constexpr int i = C >> 2, j = (C >> 1) & 1, hb = C & 1;
        const int r = lane & 31, hf = lane >> 5, chunk = lane & 7;
        const int f = f_wave + i * 32 + 4 * chunk;
        const int fc = min(f, F - 4);  // unconditional loads from clamped addresses (see EpiLinear1::fetch)
        if (C % 4 == 0) k.b = *reinterpret_cast<const float4 *>(bias + fc);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int n = min(n_wave + j * 32 + 16 * hb + 8 * it + (lane >> 3), N - 1);
            const unsigned traj = tpt_magic ? __umulhi((unsigned)n, tpt_magic) : (unsigned)n;
            k.gt[it] = *reinterpret_cast<const float4 *>(gate + (size_t)traj * mod_stride + fc);
            k.hv[it] = *reinterpret_cast<const float4 *>(h + (size_t)n * D + fc);
        }
        if (hb == 0) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const f32x16 &a = acc[i][j];
                *reinterpret_cast<float4 *>(stage + swz_bk<64>(r, 2 * q4 + hf)) = make_float4(a[4 * q4], a[4 * q4 + 1], a[4 * q4 + 2], a[4 * q4 + 3]);
            }
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) k.a[it] = *reinterpret_cast<const float4 *>(stage + swz_bk<64>(16 * hb + 8 * it + (lane >> 3), chunk));
    }

    template <int C, int MI>
    __device__ __forceinline__ void piece(f32x16 (&acc)[MI][2], char *stage, int f_wave, int n_wave, int lane, int F, int N, Pipe &k) const {
        constexpr int i = C >> 2, j = (C >> 1) & 1, hb = C & 1;
        const int chunk = lane & 7;
        const int f = f_wave + i * 32 + 4 * chunk;
        float4 out[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            out[it].x = fmaf(k.gt[it].x, k.a[it].x + k.b.x, k.hv[it].x);
            out[it].y = fmaf(k.gt[it].y, k.a[it].y + k.b.y, k.hv[it].y);
            out[it].z = fmaf(k.gt[it].z, k.a[it].z + k.b.z, k.hv[it].z);
            out[it].w = fmaf(k.gt[it].w, k.a[it].w + k.b.w, k.hv[it].w);
        }
        if (C + 1 < 4 * MI) fetch<(C + 1) % (4 * MI)>(acc, stage, f_wave, n_wave, lane, F, N, k);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int n = n_wave + j * 32 + 16 * hb + 8 * it + (lane >> 3);
            if (f < F && n < N) store16(h + (size_t)n * D + f, __builtin_bit_cast(u32x4, out[it]), probe & 32);
        }
    }
};


// The piece form of an epilogue (Epi::fetch / Epi::piece, written for k_gemm_pp.hip.h) run back to back as the epilogue of the
// one-tile-at-a-time kernel: 4 KiB of staging per wave instead of 8, which is what lets a PERSISTENT 256 x 256 kernel hold two
// 64-deep k-tiles (128 KiB) plus the staging of its 8 waves in the 160 KiB of LDS, and the software prefetch of the next
// piece's LDS rows / tables.  Wave tile must be 128 features x 64 tokens.
template <class E>
struct EpiPieces : E {
    __host__ __device__ EpiPieces(const E &e) : E(e) {}

    static constexpr bool lds_bias = false;  // the piece forms add the bias themselves
    static constexpr bool row_owner = E::row_owner;
    template <int WF, int WT>
    static constexpr size_t wave_stage_bytes() { return E::pp_stage_bytes; }
    template <int MI, int NJ>
    static constexpr int min_store_ops() { return 16 * (E::piece_ops(true) == 2 ? 2 : 1); }  // 16 pieces, 1 (linear1) or 2 (linear2) stores each

    template <int MI, int NJ>
    __device__ __forceinline__ void run(f32x16 (&acc)[MI][NJ], char *stage, int f_wave, int n_wave, int lane, int F, int N) const {
        static_assert(MI == 4 && NJ == 2, "piece form: 128 x 64 wave tiles");
        typename E::Pipe k;
        E::template fetch<0>(acc, stage, f_wave, n_wave, lane, F, N, k);
#define LSL_P(C) E::template piece<C>(acc, stage, f_wave, n_wave, lane, F, N, k);
        LSL_P(0) LSL_P(1) LSL_P(2) LSL_P(3) LSL_P(4) LSL_P(5) LSL_P(6) LSL_P(7)
        LSL_P(8) LSL_P(9) LSL_P(10) LSL_P(11) LSL_P(12) LSL_P(13) LSL_P(14) LSL_P(15)
#undef LSL_P
    }
};
