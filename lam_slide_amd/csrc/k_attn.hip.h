// Softmax attention over one axis of the [B, T, L] token grid (mmdit.py:42-55: SDPA, no mask, scale
// hd^-0.5), spatial (sequence = (b,t), positions l) or temporal (sequence = (b,l), positions t).  The
// reference materialises "(B T) L D" / "(B L) T D" copies (latent_si_v31.py:51-61); here the axis is a
// stride pattern over the token-major q/k/v buffer written by linear1's epilogue.
//
// One workgroup per ITEMS (sequence, head) pairs.  K and V of the pair are staged once in LDS
// (K XOR-swizzled for ds_read_b128 fragments, V plain for ds_read_b64_tr_b16 transposed fragments).
// Each wave owns 32-query tiles:
//     St = K Q^T     (A = K rows, B = Q rows)   -> lane (q = l&31, hf) holds 16 of the tile's 32 keys
//     online softmax in registers: running max / sum are per lane, one lane^32 exchange per tile
//     Ot = V^T P^T   (A = V via transposed LDS read, B = P straight from the accumulator registers)
// so every per-query quantity lives on the query's own lane.  q arrives pre-multiplied by
// hd^-0.5 * log2(e) (linear1 epilogue), so probabilities are exp2(s - max).
#pragma once
#include <type_traits>

#include "common.hip.h"

struct AttnArgs {
    const u16 *qkv;  // [N][3*HHD] bf16 (q | k | v), head-major inside each third
    u16 *z;          // [N][zw] bf16, attention output goes to columns [0, HHD)
    int HHD, zw, H;
    int S;           // sequence length (L or T)
    int n_seq;       // number of sequences
    // token of (seq, pos) = (seq / inner) * outer_stride + (seq % inner) + pos * pos_stride
    int inner, outer_stride, pos_stride;
    int nt;          // streaming stores for the output (common.hip.h: store8)
    int hd;          // true head_dim (<= HDP).  hd < HDP (peptide: 24 of 32): the padding channels of v are zero, and k_attention_rows
                     // turns channel hd of the staged V into ones, so that row hd of O^T = V^T P^T IS the softmax denominator
    int bound;       // k_attention_rows: 1 = softmax shifted by the Cauchy-Schwarz bound |q| max|k| instead of the row maximum when that is safe
    int planes, npad;    // k_attention_stream (spatial): 1 = q / k / v are head-major planes qkv[section][head][npad tokens][HDP] (k_lin1.hip.h)
    const float *kmax2;  // k_attention_stream: device scalar, an upper bound of |k_j|^2 for every key of this block (head_dim max_d ks_d^2: k_rope_scaled)
    const float *qmax2;  // k_attention_stream: the same bound for the queries BEFORE the softmax pre-multiplier `premul` (head_dim max_d qs_d^2)
    float premul;
    int blk, n_tok;      // k_attention_stream SHORT, tiny spatial axes (round 5): blk > 0 = the launch presents 32 / blk consecutive sequences of blk
                         // positions (a power of two <= 16) as ONE 32-row sequence (S = 32, n_seq = tiles of 32 tokens): scores outside the
                         // block diagonal are masked; n_tok = valid tokens (rows of the last tile past it are not stored)
};

template <int HDP>
__device__ __forceinline__ int k_swz(int row, int chunk) {
    // 16-byte chunk swizzle making the 16-lane groups of ds_read_b128 hit distinct banks
    if (HDP == 32) return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4);
    return row * 32 + ((chunk ^ ((row >> 3) & 1)) << 4);
}

template <int HDP, int NW, int ITEMS>
__global__ void __launch_bounds__(NW * 64, 2) k_attention(AttnArgs a) {  // (min waves/SIMD: MFMA results stay in VGPRs, see k_attention_rows)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWB = HDP * 2;         // bytes per K/V row
    constexpr int CPR = ROWB / 16;        // 16-byte chunks per row
    constexpr int KS = HDP / 16;          // k-steps of the QK^T contraction
    constexpr int WPI = NW / ITEMS;       // waves per (seq, head) item
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hf = lane >> 5;
    const int S = a.S, Sp = (S + 31) & ~31, nkt = Sp >> 5;
    const int item_local = wave / WPI, wsub = wave % WPI;
    const long item = (long)blockIdx.x * ITEMS + item_local;
    const bool item_ok = item < (long)a.n_seq * a.H;
    const int seq = item_ok ? (int)(item / a.H) : 0, head = item_ok ? (int)(item % a.H) : 0;
    const size_t tok0 = (size_t)(seq / a.inner) * a.outer_stride + (seq % a.inner);
    const size_t rs = (size_t)3 * a.HHD;  // qkv row stride (elements)
    char *Ks = smem + (size_t)item_local * 2 * Sp * ROWB;
    char *Vs = Ks + (size_t)Sp * ROWB;

    // stage K (swizzled) and V (plain); rows >= S are zero so padded keys contribute exactly 0 * 0
    {
        const int ltid = wsub * 64 + lane, lthreads = WPI * 64;
        const u16 *kbase = a.qkv + tok0 * rs + a.HHD + head * HDP;
        const u16 *vbase = kbase + a.HHD;
        for (int i = ltid; i < Sp * CPR; i += lthreads) {
            const int row = i / CPR, ch = i % CPR;
            u32x4 kv = {0, 0, 0, 0}, vv = {0, 0, 0, 0};
            if (row < S && item_ok) {
                const size_t off = (size_t)row * a.pos_stride * rs + ch * 8;
                kv = *reinterpret_cast<const u32x4 *>(kbase + off);
                vv = *reinterpret_cast<const u32x4 *>(vbase + off);
            }
            *reinterpret_cast<u32x4 *>(Ks + k_swz<HDP>(row, ch)) = kv;
            *reinterpret_cast<u32x4 *>(Vs + row * ROWB + ch * 16) = vv;
        }
    }
    __syncthreads();

    // transposed-read lane geometry (ds_read_b64_tr_b16, per 16-lane group: lane 4q+p supplies row q,
    // columns 4p..4p+3 of a 4x16 block; lane i receives column i of the 4 rows)
    const int gi = lane & 15, gq = gi >> 2, gp = gi & 3, grp = lane >> 4;
    const int v_col = (HDP == 32 ? (grp & 1) * 16 : 0) + 4 * gp;
    const int v_row0 = 4 * (grp >> 1) + gq;  // + 16*s + 8*half + 32*kt

    for (int qt = wsub; qt < nkt; qt += WPI) {
        // Q fragments straight from global memory (B operand: Q[q = r][hd = 16 s + 8 hf + j])
        const int qpos = min(qt * 32 + r, S - 1);
        const u16 *qrow = a.qkv + (tok0 + (size_t)qpos * a.pos_stride) * rs + head * HDP;
        bf16x8 qf[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) qf[s] = as_bf16x8(*reinterpret_cast<const u32x4 *>(qrow + 16 * s + 8 * hf));

        f32x16 o;
#pragma unroll
        for (int e = 0; e < 16; ++e) o[e] = 0.0f;
        float m_run = -INFINITY, l_run = 0.0f;

        for (int kt = 0; kt < nkt; ++kt) {
            f32x16 sacc;
#pragma unroll
            for (int e = 0; e < 16; ++e) sacc[e] = 0.0f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 kf = as_bf16x8(*reinterpret_cast<const u32x4 *>(Ks + k_swz<HDP>(kt * 32 + r, (KS == 2 ? 2 * s : 0) + hf)));
                sacc = mfma32(kf, qf[s], sacc);
            }
            if (kt * 32 + 32 > S) {  // wave-uniform: mask the zero-padded keys of the last tile
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (kt * 32 + acc_row(e, hf) >= S) sacc[e] = -INFINITY;
            }
            float mt = sacc[0];
#pragma unroll
            for (int e = 1; e < 16; ++e) mt = fmaxf(mt, sacc[e]);
            mt = fmaxf(mt, xhalf(mt));
            const float m_new = fmaxf(m_run, mt);
            const float alpha = exp2f(m_run - m_new);
            float psum = 0.0f;
            float p[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                p[e] = exp2f(sacc[e] - m_new);
                psum += p[e];
            }
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int e = 0; e < 16; ++e) o[e] *= alpha;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                // B operand: P[q][key], element j of k-step s = accumulator register 8 s + j
                u32x4 pw = {pack2(p[8 * s], p[8 * s + 1]), pack2(p[8 * s + 2], p[8 * s + 3]),
                            pack2(p[8 * s + 4], p[8 * s + 5]), pack2(p[8 * s + 6], p[8 * s + 7])};
                // A operand: V^T[hd = r][key], same key order: element j <-> key 16 s + 8 (j>>2) + 4 hf + (j&3)
                const int vr = kt * 32 + 16 * s + v_row0;
                const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(Vs + vr * ROWB + v_col * 2));
                const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(Vs + (vr + 8) * ROWB + v_col * 2));
                typedef __attribute__((ext_vector_type(8))) short s16x8;
                const s16x8 vv = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                o = mfma32(__builtin_bit_cast(bf16x8, vv), as_bf16x8(pw), o);
            }
        }
        const float inv_l = 1.0f / (l_run + xhalf(l_run));
        const int qglob = qt * 32 + r;
        if (qglob < S && item_ok) {
            u16 *dst = a.z + (tok0 + (size_t)qglob * a.pos_stride) * a.zw + head * HDP;
#pragma unroll
            for (int q4 = 0; q4 < HDP / 8; ++q4) {
                u32x2 pk = {pack2(o[4 * q4] * inv_l, o[4 * q4 + 1] * inv_l), pack2(o[4 * q4 + 2] * inv_l, o[4 * q4 + 3] * inv_l)};
                store8(dst + 8 * q4 + 4 * hf, pk, a.nt);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Short sequences (S <= 32 * NKT <= 256: every axis of the MD17 / pedestrian / NBA configs, both axes of peptide's
// spatial attention): softmax is the plain two-pass form (one max pass, one exp/sum/PV pass, no running rescale of
// O) -- per score element: max, sub, exp2, add, half a cvt; the scores are recomputed in the second pass.  Measured on
// MI355X the online form above spent 17 VALU instructions per score element (profiles/r01_rocprof_summary.txt:
// VALU : MFMA = 68 : 1), which is what bounds attention at head_dim 32, not the MFMAs.
template <int HDP, int NW, int ITEMS, int NKT>
__global__ void __launch_bounds__(NW * 64, 4) k_attention_rows(AttnArgs a) {  // >= 4 waves/SIMD: keeps the MFMA results in VGPRs (with the
    // whole 512-register budget hipcc parks them in AGPRs and spends a v_accvgpr_read per score element to get them back)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWB = HDP * 2, CPR = ROWB / 16, KS = HDP / 16, WPI = NW / ITEMS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hf = lane >> 5;
    const int S = a.S;
    const int nkt = NKT > 0 ? NKT : (S + 31) >> 5, Sp = nkt * 32;  // NKT = 0: key-tile count known at run time only
    // padded heads (hd = 24 of HDP = 32): the denominator comes out of the PV product (a column of ones in V's padding) instead of a
    // third MFMA against an all-ones operand: 6 instead of 8 MFMAs per (query tile, key tile) in a kernel whose MFMA and VALU time add
    const bool ones_col = HDP == 32 && a.hd == 24;  // (wave-uniform; acc_row(12, 0) == 24)
    const int item_local = wave / WPI, wsub = wave % WPI;
    // heads of one sequence share 128-byte lines of the token rows: keep neighbouring (sequence, head) items on one XCD
    const long item = (long)xcd_remap(blockIdx.x, gridDim.x) * ITEMS + item_local;
    const bool item_ok = item < (long)a.n_seq * a.H;
    const int seq = item_ok ? (int)(item / a.H) : 0, head = item_ok ? (int)(item % a.H) : 0;
    const size_t tok0 = (size_t)(seq / a.inner) * a.outer_stride + (seq % a.inner);
    const size_t rs = (size_t)3 * a.HHD;
    char *Ks = smem + (size_t)item_local * 2 * Sp * ROWB;
    char *Vs = Ks + (size_t)Sp * ROWB;
    // the wave's first query tile is requested together with K / V, ahead of the barrier: one exposed memory latency instead of two
    bf16x8 qf0[KS];
    {
        const int qpos = min(wsub * 32 + r, S - 1);
        const u16 *qrow = a.qkv + (tok0 + (size_t)qpos * a.pos_stride) * rs + head * HDP;
#pragma unroll
        for (int s = 0; s < KS; ++s) qf0[s] = as_bf16x8(*reinterpret_cast<const u32x4 *>(qrow + 16 * s + 8 * hf));
    }
    // Softmax shift without a max pass (a.bound): softmax is invariant under any per-query shift, and |s_ij| <= |q_i| |k_j| <= |q_i| max_j |k_j| =: m_i
    // costs O(S hd) instead of the S^2 score recomputation + one v_max per score of the max pass (this kernel is bound by its vector
    // instructions: 5.5 issue slots per score with the max pass, 4.5 without).  exp2(s - m_i) lies in [2^(-2 m_i), 1]: no underflow while
    // m_i <= 60 (q and k are RMS-normalised: m_i ~ 1.44 sqrt(hd) w_q w_k, about 8 for unit norm weights); a query tile with a larger bound
    // takes the max pass (wave-uniform).  The probabilities differ from the max-shifted ones by a per-query power-of-two-ish factor that
    // cancels in O / l; bf16 / fp32 relative precision does not depend on it.
    float kmax2 = 0.0f;
    float *const red = reinterpret_cast<float *>(smem + (size_t)ITEMS * 2 * Sp * ROWB);  // NW floats behind K / V (sized by the launcher)
    {
        const int ltid = wsub * 64 + lane, lthreads = WPI * 64;
        const u16 *kbase = a.qkv + tok0 * rs + a.HHD + head * HDP;
        const u16 *vbase = kbase + a.HHD;
        for (int i = ltid; i < Sp * CPR; i += lthreads) {
            const int row = i / CPR, ch = i % CPR;
            u32x4 kv = {0, 0, 0, 0}, vv = {0, 0, 0, 0};
            if (row < S && item_ok) {
                const size_t off = (size_t)row * a.pos_stride * rs + ch * 8;
                kv = *reinterpret_cast<const u32x4 *>(kbase + off);
                vv = *reinterpret_cast<const u32x4 *>(vbase + off);
                if (ones_col && ch == a.hd / 8) {  // channel hd (head dims are multiples of 8) of a REAL key row := 1.0 (bf16 0x3F80)
                    const int el = (a.hd & 7) >> 1;
                    vv[el] = (vv[el] & 0xffff0000u) | 0x3F80u;
                }
            }
            *reinterpret_cast<u32x4 *>(Ks + k_swz<HDP>(row, ch)) = kv;
            *reinterpret_cast<u32x4 *>(Vs + row * ROWB + ch * 16) = vv;
            if (a.bound) {  // squared norm of the key row: its CPR chunks sit on CPR neighbouring lanes
                float ss = 0.0f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float lo = __uint_as_float(kv[k] << 16), hi = __uint_as_float(kv[k] & 0xffff0000u);
                    ss = fmaf(lo, lo, fmaf(hi, hi, ss));
                }
                ss += __shfl_xor(ss, 1, 64);
                if (CPR == 4) ss += __shfl_xor(ss, 2, 64);
                kmax2 = fmaxf(kmax2, ss);
            }
        }
    }
    if (a.bound) {
#pragma unroll
        for (int m = 32; m >= CPR; m >>= 1) kmax2 = fmaxf(kmax2, __shfl_xor(kmax2, m, 64));
        if (lane == 0) red[wave] = kmax2;
    }
    __syncthreads();
    if (a.bound) {
        kmax2 = red[item_local * WPI];
#pragma unroll
        for (int w = 1; w < WPI; ++w) kmax2 = fmaxf(kmax2, red[item_local * WPI + w]);
    }

    const int gi = lane & 15, gq = gi >> 2, gp = gi & 3, grp = lane >> 4;
    const int v_off = (4 * (grp >> 1) + gq) * ROWB + ((HDP == 32 ? (grp & 1) * 16 : 0) + 4 * gp) * 2;
    f32x16 zero;
#pragma unroll
    for (int e = 0; e < 16; ++e) zero[e] = 0.0f;
    const int nqt = (S + 31) >> 5;

    for (int qt = wsub; qt < nqt; qt += WPI) {
        bf16x8 qf[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) qf[s] = qf0[s];
        if (qt + WPI < nqt) {  // the next query tile of this wave is requested now and consumed one iteration later
            const int qpos = min((qt + WPI) * 32 + r, S - 1);
            const u16 *qrow = a.qkv + (tok0 + (size_t)qpos * a.pos_stride) * rs + head * HDP;
#pragma unroll
            for (int s = 0; s < KS; ++s) qf0[s] = as_bf16x8(*reinterpret_cast<const u32x4 *>(qrow + 16 * s + 8 * hf));
        }

        // pass 1: row maximum only (scores are recomputed in pass 2: two extra MFMAs per tile are far cheaper than
        // keeping 16 * NKT score registers live, which costs the occupancy that hides the LDS / exp latencies)
        auto scores = [&](int kt, const f32x16 &init) {
            f32x16 t = mfma32(as_bf16x8(*reinterpret_cast<const u32x4 *>(Ks + k_swz<HDP>(kt * 32 + r, hf))), qf[0], init);
            if (KS == 2)
                t = mfma32(as_bf16x8(*reinterpret_cast<const u32x4 *>(Ks + k_swz<HDP>(kt * 32 + r, 2 + hf))), qf[KS - 1], t);
            if (kt * 32 + 32 > S) {  // wave-uniform: zero-padded keys of the last tile do not take part
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (kt * 32 + acc_row(e, hf) >= S) t[e] = -INFINITY;
            }
            return t;
        };
        float mx = -INFINITY;
        bool shifted = false;
        if (a.bound) {
            float qq = 0.0f;
#pragma unroll
            for (int s2 = 0; s2 < KS; ++s2) {
                const u32x4 w = __builtin_bit_cast(u32x4, qf[s2]);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float lo = __uint_as_float(w[k] << 16), hi = __uint_as_float(w[k] & 0xffff0000u);
                    qq = fmaf(lo, lo, fmaf(hi, hi, qq));
                }
            }
            qq += xhalf(qq);
            const float m = sqrtf(qq * kmax2) * 1.001f;
            shifted = __ballot(m > 60.0f) == 0;  // (wave-uniform)
            if (shifted) mx = m;
        }
        if (!shifted) {
#pragma unroll 2
            for (int kt = 0; kt < nkt; ++kt) {
                const f32x16 t = scores(kt, zero);
#pragma unroll
                for (int e = 0; e < 16; ++e) mx = fmaxf(mx, t[e]);
            }
            mx = fmaxf(mx, xhalf(mx));
        }
        // pass 2: the row maximum is subtracted by the MFMA itself (accumulator preset to -max), and the row sum comes out of a
        // third MFMA against an all-ones operand (sum over keys of the SAME bf16 probabilities that multiply V) - two VALU
        // operations less per score element, and this kernel is VALU-bound at head_dim 32
        f32x16 negmx;
#pragma unroll
        for (int e = 0; e < 16; ++e) negmx[e] = -mx;
        const u32x4 ones_w = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};  // eight bf16 1.0
        f32x16 o = zero, lsum = zero;
#pragma unroll 2
        for (int kt = 0; kt < nkt; ++kt) {
            const f32x16 t = scores(kt, negmx);
            float p[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) p[e] = __builtin_amdgcn_exp2f(t[e]);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                u32x4 pw = {pack2(p[8 * s], p[8 * s + 1]), pack2(p[8 * s + 2], p[8 * s + 3]),
                            pack2(p[8 * s + 4], p[8 * s + 5]), pack2(p[8 * s + 6], p[8 * s + 7])};
                const char *vb = Vs + (kt * 32 + 16 * s) * ROWB + v_off;
                const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(vb));
                const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(vb + 8 * ROWB));
                typedef __attribute__((ext_vector_type(8))) short s16x8;
                const s16x8 vv = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                o = mfma32(__builtin_bit_cast(bf16x8, vv), as_bf16x8(pw), o);
                if (!ones_col) lsum = mfma32(as_bf16x8(ones_w), as_bf16x8(pw), lsum);  // every row = sum over this tile's keys, column = query
            }
        }
        float l = lsum[0];
        if (ones_col) {  // row 24 of O^T: register 12 of the lanes with hf == 0; the other half gets it by one exchange
            const float mine = o[12], other = xhalf(mine);
            l = hf ? other : mine;
            if (!hf) o[12] = 0.0f;  // the padding channel itself stays zero in z
        }
        const float inv_l = 1.0f / l;
        const int qglob = qt * 32 + r;
        if (qglob < S && item_ok) {
            u16 *dst = a.z + (tok0 + (size_t)qglob * a.pos_stride) * a.zw + head * HDP;
#pragma unroll
            for (int q4 = 0; q4 < HDP / 8; ++q4) {
                u32x2 pk = {pack2(o[4 * q4] * inv_l, o[4 * q4 + 1] * inv_l), pack2(o[4 * q4 + 2] * inv_l, o[4 * q4 + 3] * inv_l)};
                store8(dst + 8 * q4 + 4 * hf, pk, a.nt);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Streaming form of k_attention_rows for the axes of the large configurations (round 4).  Measured on k_attention_rows at cfg 2
// (profiles/r03_rocprof_summary.txt): waves are parked at s_waitcnt / the staging barrier for 52 % of their lifetime - 15 360 (spatial) or
// 32 768 (temporal) short-lived workgroups each wait for their own K / V rows from HBM before they can start, and five resident
// workgroups per CU do not cover that.  Here a PERSISTENT 8-wave workgroup (two per CU) walks the units u = v, v + grid, ... and the rows
// of its next unit arrive by LDS-DMA (per-lane source addresses apply the read-side swizzle) in the second of two K | V images while the
// current unit is computed; each wave's next query tile arrives the same way in a wave-private 32-row image.  One barrier per unit.
//   unit (LONG: 128 < S <= 256)  = one (sequence, head): K / V images of 256 rows, wave w owns query tile w;
//   unit (SHORT: 8 < S <= 32)    = 8 consecutive heads of one sequence: 8 images of 32 rows, wave w owns head w of the group.
// Softmax without a maximum and without a shift: |s_ij| <= |q_i| |k_j| <= |q_i| sqrt(kmax2) =: m_i with kmax2 = head_dim max_d ks_d^2 >=
// |k_j|^2 for every key of the block (k = RMS-normalised x scale, rotated: AttnArgs::kmax2, written by k_rope_scaled) - a bound that needs
// neither a max pass nor the staged keys.  While m_i <= 60, exp2(s_ij) itself fits fp32 / bf16 and the common factor cancels in O / l; a
// query tile with a larger bound takes the exact max pass (k_attention_rows has the argument).  Row sums by the all-ones MFMA.
// Memory-pipe lessons built in (profiles/r04_experiments.txt): vector-memory operations complete in issue order, so whatever a wave
// issues in front of its next requests delays them - the output therefore leaves as 16-byte-per-lane row pieces through the (by then
// free) query image, with plain stores (8-byte stores from the accumulator layout: 32 partial writes per instruction; streaming stores
// 3 x slower); q / k / v of spatial sub-blocks come as head-major planes (AttnArgs::planes), where a unit's rows are one contiguous run -
// as token-major rows every 64-byte row piece drags a whole 128-byte line through the L1 miss path; units are dealt round-robin in
// XCD-contiguous order so that heads sharing lines meet in one L2.
#ifndef LSL_ATTN_XL_UNROLL
#define LSL_ATTN_XL_UNROLL 1
#endif
#ifndef LSL_ATTN_XL_PIPE
#define LSL_ATTN_XL_PIPE 1  // chunked-key form, 32-wide heads: exponentials of tile kt behind the MFMAs of tiles kt / kt + 1 (below)
#endif
template <int HDP, bool LONG, bool XL = false, bool ONES = false, bool PACK = false>
__global__ void __launch_bounds__(512, 4) k_attention_stream(AttnArgs a) {
    static_assert(LONG || !XL, "chunked keys / grouped queries exist for the LONG form only");
    static_assert(!LONG || !PACK, "packed tiny axes run the SHORT form");
    static_assert(!ONES || (XL && HDP == 32), "the denominator column needs a padded head (head_dim 24 of 32); instantiated for the chunked form");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWB = HDP * 2, CPR = ROWB / 16, KS = HDP / 16;
    constexpr int KVB = 256 * ROWB, BUF = 2 * KVB;  // K (or V) image of a stage; K | V
    constexpr int NI = BUF / 1024, IPW = NI / 8;    // LDS-DMA instructions per stage / per wave
    constexpr int RPI = 1024 / ROWB;                // image rows per instruction
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hf = lane >> 5;
    const int S = a.S;
    // LONG axes of any length (round 5): the keys of a (sequence, head) go through the two images in CHUNKS of 256 rows, the queries in GROUPS
    // of 8 tiles (one per wave); a unit = (sequence, head, query group) walks its NC key chunks as NC stages, the accumulators stay in
    // registers across them - with the unshifted softmax below there is nothing to rescale between chunks.  S <= 256: one chunk, one group,
    // i.e. exactly the round-4 kernel (same instruction order per tile, same bits).  SHORT: one stage per unit.
    // (XL = false: S <= 256, both counts are the compile-time constant 1 and the code below folds to the round-4 kernel)
    const int NC = XL ? (S + 255) >> 8 : 1;                         // key chunks
    const int QG = XL ? (((S + 31) >> 5) + 7) >> 3 : 1;             // query groups
    const int rows_last = S - 256 * (NC - 1);                       // key rows of the last chunk (1 .. 256)
    const int nkt_last = LONG ? (rows_last + 31) >> 5 : 1;          // key tiles of the last chunk
    const int qrows_last = S - 256 * (QG - 1);                      // query rows of the last group
    const int hgroups = a.H >> 3;
    // (32-bit unit arithmetic: the host keeps launches of 2^31 units and more off this kernel; 64-bit divisions by run-time values cost ~ 75
    // scalar instructions each in front of every unit's requests)
    const unsigned n_units = LONG ? (unsigned)a.n_seq * a.H * QG : (unsigned)a.n_seq * hgroups;
    // units are dealt round-robin over the workgroups in XCD-contiguous order: neighbouring units - the query groups of one (sequence, head),
    // which read the same keys, and neighbouring heads, whose 64-byte row pieces share 128-byte lines - run at the same time on one XCD and
    // meet in its L2 (with a contiguous range per workgroup the second head of a line came one unit-time later, after the line had left the
    // L2: FETCH_SIZE 1.38 GB per launch for 0.755 GB of rows)
    const unsigned ustep = gridDim.x, u0 = xcd_remap(blockIdx.x, gridDim.x), u1 = n_units;
    if (u0 >= u1) return;  // (uniform)
    const unsigned rs = 3u * a.HHD;
    const unsigned lds0 = (unsigned)(size_t)(LDS_PTR(char))(smem);
    const bool planes = LONG && a.planes;  // (uniform) plane layout: positions of a sequence are consecutive tokens (pos_stride 1)
    // padded heads (hd = 24 of HDP = 32, peptide): channel hd of every staged V row := 1.0, so that row hd of O^T = V^T P^T IS the softmax
    // denominator (k_attention_rows has the argument): 4 instead of 6 MFMAs per tile pair
    constexpr bool ones_col = ONES;  // (the host selects the instance: head_dim == 24; acc_row(12, 0) == 24)

    // per-lane parts of the K / V request addresses: the part that does not depend on the key row, and the row R (0 .. 255) of the image the
    // lane fills; the row's position term is added per request (rows past the end of the sequence - last chunk only - clamped)
    unsigned cst_kv[IPW];
    int row_kv[IPW];
    const unsigned pos_bytes = planes ? 2u * HDP : 2u * a.pos_stride * rs;  // bytes between consecutive positions of a sequence
#pragma unroll
    for (int k = 0; k < IPW; ++k) {
        const int i = wave * IPW + k, kv = i / (NI / 2), ii = i % (NI / 2);
        const int R = RPI * ii + lane / CPR, slot = lane % CPR;
        const int chunk = kv == 0 ? slot ^ (HDP == 32 ? (R >> 2) & 3 : (R >> 3) & 1) : slot;
        const int item_local = LONG ? 0 : R >> 5;
        row_kv[k] = LONG ? R : R & 31;
        cst_kv[k] = planes ? 2u * ((unsigned)(1 + kv) * a.H * a.npad * HDP + chunk * 8) : 2u * ((1 + kv) * a.HHD + item_local * HDP + chunk * 8);
        if (!XL && !PACK) cst_kv[k] += pos_bytes * (unsigned)min(row_kv[k], S - 1);  // (packed tiny axes: clamped per unit, below)
    }

    auto unit_tok0 = [&](unsigned u, int &head0, int &qg) __attribute__((always_inline)) {
        unsigned sh = u;
        qg = 0;
        if (LONG && QG > 1) {  // (uniform)
            sh = u / (unsigned)QG;
            qg = (int)(u - sh * (unsigned)QG);
        }
        const unsigned hdiv = LONG ? (unsigned)a.H : (unsigned)hgroups, seq = sh / hdiv, hidx = sh - seq * hdiv;
        head0 = LONG ? (int)hidx : 8 * (int)hidx;
        const unsigned so = seq / (unsigned)a.inner;
        return (size_t)so * a.outer_stride + (seq - so * (unsigned)a.inner);
    };
    // (wave-uniform by construction; the readfirstlanes make it provable for the "s" operands of the asm statements)
    auto uni_ptr = [](const char *q) __attribute__((always_inline)) {
        const unsigned long long v = (unsigned long long)q;
        const unsigned lo32 = __builtin_amdgcn_readfirstlane((unsigned)v), hi32 = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<const char *>(((unsigned long long)hi32 << 32) | lo32);
    };
    // The query tile of a wave's next unit arrives by LDS-DMA as well, in a WAVE-PRIVATE 32-row image (swizzled like K): only this wave
    // reads it, so it is single-buffered and needs no barrier - the wave requests the next tile once its own reads of the current one have
    // returned.  (Asm loads into registers were tried first: hipcc copied the destination registers of the in-flight loads at the loop's
    // back-edge and in front of the wait - NaNs; and once accumulator registers are named in asm it allocates the same ones itself.)
    constexpr int QIMG = 32 * ROWB, QPW = QIMG / 1024 > 0 ? QIMG / 1024 : 1;  // bytes and LDS-DMA instructions of a query tile
    char *const qimg = smem + 2 * BUF + wave * QIMG;
    unsigned cst_qt[QPW];  // (as for K / V: row-independent part; the tile row is RPI k + lane / CPR)
#pragma unroll
    for (int k = 0; k < QPW; ++k) {
        const int R = RPI * k + lane / CPR, slot = lane % CPR;  // tile row, 16-byte slot
        const int chunk = slot ^ (HDP == 32 ? (R >> 2) & 3 : (R >> 3) & 1);
        cst_qt[k] = 2u * ((LONG ? 0 : wave * HDP) + chunk * 8);
        if (!XL && !PACK) cst_qt[k] += pos_bytes * (unsigned)min((LONG ? 32 * wave : 0) + R, S - 1);
    }
    // K | V rows of key chunk c of unit u -> image SET
    // (a unit's first token, head and query group are computed ONCE - when its requests are issued, one unit ahead - and carried)
    auto request_kv = [&](size_t unit_tok, int head0, int c, int SET) __attribute__((always_inline)) {
        const size_t tok0 = unit_tok + (size_t)(256 * c) * a.pos_stride;
        const char *base = uni_ptr(reinterpret_cast<const char *>(a.qkv) +
                                   (planes ? 2 * (((size_t)head0 * a.npad + tok0) * HDP) : 2 * (tok0 * rs + (size_t)head0 * HDP)));
        const bool last = c == NC - 1;  // (uniform)
#pragma unroll
        for (int k = 0; k < IPW; ++k) {
            const int i = wave * IPW + k;
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + SET * BUF + (i / (NI / 2)) * KVB + (i % (NI / 2)) * 1024);
            // (!XL: one stage per unit, the clamped position was folded into cst_kv once)
            // packed tiny axes: the last tile of the launch may hold fewer than 32 tokens - rows past the end repeat the last one (finite values:
            // their probabilities are exactly 0, but 0 x an uninitialised V row could still be NaN)
            const unsigned vk = XL ? cst_kv[k] + pos_bytes * (unsigned)min(row_kv[k], (last ? rows_last : 256) - 1)
                                   : (PACK ? cst_kv[k] + pos_bytes * (unsigned)min(row_kv[k], min(32, a.n_tok - (int)tok0) - 1) : cst_kv[k]);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vk), "s"(base), "s"(dst) : "memory");
        }
    };
    // the wave's query tile of unit u -> its private image
    auto request_q = [&](size_t unit_tok, int head0, int qg) __attribute__((always_inline)) {
        const size_t tok0 = unit_tok + (size_t)(256 * qg) * a.pos_stride;
        const char *base = uni_ptr(reinterpret_cast<const char *>(a.qkv) +
                                   (planes ? 2 * (((size_t)head0 * a.npad + tok0) * HDP) : 2 * (tok0 * rs + (size_t)head0 * HDP)));
        const bool last = qg == QG - 1;  // (uniform)
#pragma unroll
        for (int k = 0; k < QPW; ++k) {
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + 2 * BUF + wave * QIMG + k * 1024);
            const int R = RPI * k + lane / CPR, lim = (LONG ? (last ? qrows_last : 256) : S) - 1;
            const unsigned vq = XL ? cst_qt[k] + pos_bytes * (unsigned)max(min(32 * wave + R, lim), 0)
                                   : (PACK ? cst_qt[k] + pos_bytes * (unsigned)min(R, min(32, a.n_tok - (int)tok0) - 1) : cst_qt[k]);
            if (QIMG >= 1024 || lane < QIMG / 16)
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vq), "s"(base), "s"(dst) : "memory");
        }
    };
    // ones_col: the V rows a wave requested itself (waves 4 - 7: rows 64 (w - 4) .. + 63 of the image), patched once they have landed and
    // before the stage's barrier
    auto patch_ones = [&](int SET) __attribute__((always_inline)) {
        if (ones_col && wave >= 4) {
            *reinterpret_cast<u16 *>(smem + SET * BUF + KVB + (64 * (wave - 4) + lane) * ROWB + 2 * a.hd) = (u16)0x3F80;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    };

    const int gi = lane & 15, gq = gi >> 2, gp = gi & 3, grp = lane >> 4;
    const int v_off = (4 * (grp >> 1) + gq) * ROWB + ((HDP == 32 ? (grp & 1) * 16 : 0) + 4 * gp) * 2;
    const int krow0 = LONG ? 0 : 32 * wave;  // first image row of this wave's keys
    const bool ragged = ((LONG ? rows_last : S) & 31) != 0;  // (uniform) the last key tile of the last chunk is partial
    const int last_rows = (LONG ? rows_last : S) - 32 * (nkt_last - 1);
    f32x16 zero;
#pragma unroll
    for (int e = 0; e < 16; ++e) zero[e] = 0.0f;
    const float kmax2 = a.kmax2 ? *a.kmax2 : 0.0f;
    // the bound for EVERY query of the block, from the norm scales alone: |q_i| <= premul sqrt(qmax2) (RMS-normalised over head_dim, times scale,
    // rotated).  At most 60 (the shipped models: 9 - 15): no per-query bound is computed at all (78 vector instructions per wave and unit)
    const bool all_shifted = a.kmax2 && a.qmax2 && sqrtf(*a.qmax2 * kmax2) * a.premul * 1.02f <= 60.0f;  // (uniform)
    // tiny axes packed 32 / blk sequences to a tile (AttnArgs::blk): bit e of the lane = accumulator register e holds a key of the lane's own
    // sequence (key row acc_row(e, hf), query row r: same block of blk rows)
    constexpr bool grouped = PACK;  // (the host selects the instance: AttnArgs::blk > 0)
    unsigned own_bits = 0xFFFFu;
    if (grouped) {
        own_bits = 0;
#pragma unroll
        for (int e = 0; e < 16; ++e)
            if (acc_row(e, hf) / a.blk == r / a.blk) own_bits |= 1u << e;
    }

    // Output: a wave's tile O^T (32 queries on the lanes, HDP channels on the accumulator rows) goes through the wave's query image - free
    // once the NEXT unit's query tile has been read into registers - and leaves as whole row pieces, 16 bytes per lane (HDP = 32: 16 rows x
    // 64 B per instruction).  Stored straight from the accumulator layout (8 bytes per lane, 32 partial 16-byte writes per instruction) the
    // stores clog the in-order vector-memory pipe in front of the next unit's requests: measured 0.78 ms per launch with streaming stores,
    // 0.27 with plain ones, 0.205 with no store at all (profiles/r04_experiments.txt).
    constexpr int OPW = QPW;  // output store instructions per wave and unit
    unsigned voff_zr[OPW];    // (relative to the unit's first query row)
#pragma unroll
    for (int k = 0; k < OPW; ++k) {
        const int R = RPI * k + lane / CPR, chunk = lane % CPR;
        const int pos = (LONG ? 32 * wave : 0) + R;
        voff_zr[k] = 2u * ((unsigned)pos * a.pos_stride * a.zw + (LONG ? 0 : wave * HDP) + chunk * 8);
    }
    auto read_q = [&](bf16x8 (&q)[KS]) __attribute__((always_inline)) {
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) q[s2] = as_bf16x8(*reinterpret_cast<const u32x4 *>(qimg + k_swz<HDP>(r, 2 * s2 + hf)));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // in registers before anything else is written to the image
    };
    int nx_head0, nx_qg;
    size_t nx_tok0 = unit_tok0(u0, nx_head0, nx_qg);
    request_kv(nx_tok0, nx_head0, 0, 0);
    request_q(nx_tok0, nx_head0, nx_qg);
    wait_vmcnt<0>();
    patch_ones(0);
    bf16x8 qn[KS];
    read_q(qn);
    int buf = 0;
    for (unsigned u = u0; u < u1; u += ustep) {
        bf16x8 qf[KS];
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) qf[s2] = qn[s2];
        const int head0 = nx_head0, qg = nx_qg;
        const size_t tok0 = nx_tok0;
        const bool last_g = qg == QG - 1;                                     // (uniform)
        const bool has_tile = !LONG || 256 * qg + 32 * wave < S;              // (uniform) LONG: wave w = query tile w of the group
        const bool more = u + ustep < u1;                                     // (uniform)
        float mx = -INFINITY;
        bool shifted = all_shifted;
        f32x16 o = zero, lsum = zero;
#pragma unroll 1
        for (int c = 0; c < NC; ++c) {  // stage (u, c)
            __builtin_amdgcn_s_barrier();  // every wave's rows of this stage have landed (each waited for its own); every wave has left the other image
            asm volatile("" ::: "memory");
            const bool last_c = c == NC - 1;  // (uniform)
            if (!last_c) request_kv(tok0, head0, c + 1, buf ^ 1);
            else if (more) {
                nx_tok0 = unit_tok0(u + ustep, nx_head0, nx_qg);
                request_kv(nx_tok0, nx_head0, 0, buf ^ 1);
                request_q(nx_tok0, nx_head0, nx_qg);
            }
            // softmax bound of this wave's queries (see above): decided once per unit, behind the barrier and the next stage's requests
            if (c == 0 && has_tile && a.kmax2 && !all_shifted) {
                float qq = 0.0f;
#pragma unroll
                for (int s2 = 0; s2 < KS; ++s2) {
                    const u32x4 w = __builtin_bit_cast(u32x4, qf[s2]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float lo = __uint_as_float(w[k] << 16), hi = __uint_as_float(w[k] & 0xffff0000u);
                        qq = fmaf(lo, lo, fmaf(hi, hi, qq));
                    }
                }
                qq += xhalf(qq);
                const float m = sqrtf(qq * kmax2) * 1.02f;
                shifted = __ballot(m > 60.0f) == 0;  // (wave-uniform)
                if (shifted) mx = m;
            }
            if (has_tile) {
                const char *Ks = smem + buf * BUF, *Vs = Ks + KVB;
                const int nkt = !LONG ? 1 : (last_c ? nkt_last : 8);
                auto scores = [&](int kt, const f32x16 &init) __attribute__((always_inline)) {
                    f32x16 t = mfma32(as_bf16x8(*reinterpret_cast<const u32x4 *>(Ks + k_swz<HDP>(krow0 + kt * 32 + r, hf))), qf[0], init);
                    if (KS == 2)
                        t = mfma32(as_bf16x8(*reinterpret_cast<const u32x4 *>(Ks + k_swz<HDP>(krow0 + kt * 32 + r, 2 + hf))), qf[KS - 1], t);
                    if (last_c && kt == nkt_last - 1 && ragged) {  // wave-uniform; only the last tile of the last chunk can hold clamped rows past the sequence: they do not take part
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                            if (acc_row(e, hf) >= last_rows) t[e] = -INFINITY;
                    }
                    if (grouped) {  // keys of the tile's other sequences do not take part
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                            if (!((own_bits >> e) & 1u)) t[e] = -INFINITY;
                    }
                    return t;
                };
                if (!shifted) {  // exact-maximum fallback, online over the chunks: this chunk's maximum, then the running sums are rescaled
                    float mc = -INFINITY;
#pragma unroll 1
                    for (int kt = 0; kt < nkt; ++kt) {
                        const f32x16 t = scores(kt, zero);
#pragma unroll
                        for (int e = 0; e < 16; ++e) mc = fmaxf(mc, t[e]);
                    }
                    mc = fmaxf(mc, xhalf(mc));
                    if (c > 0) {  // (NC == 1: mx = mc as in the round-4 kernel, nothing to rescale)
                        const float m_new = fmaxf(mx, mc), alpha = __builtin_amdgcn_exp2f(mx - m_new);
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            o[e] *= alpha;
                            lsum[e] *= alpha;
                        }
                        mx = m_new;
                    } else
                        mx = mc;
                }
                // Shifted by the bound (the usual case): NO shift is applied at all.  |s_ij| <= m_i <= 60, so exp2(s_ij) lies in [2^-60, 2^60]: it
                // fits fp32 and bf16 (8 exponent bits) as it is, the sums over the keys stay far below the fp32 range, and the common factor 2^m_i
                // cancels in O / l exactly as a shift would.  The score MFMA then starts from the constant 0 (an inline operand) instead of a
                // 16-register preset that hipcc copies into the accumulator in front of every tile (16 of the 60 vector instructions per tile
                // pair that the counters showed).  Exact-maximum fallback: the preset form.
                // (XL: the fallback subtracts the maximum with a vector instruction per score instead - a cold path, and the 16-register preset
                // beside two accumulator tiles that live across the chunk loop does not fit 128 registers)
                f32x16 negmx;
#pragma unroll
                for (int e = 0; e < 16; ++e) negmx[e] = (shifted || XL) ? 0.0f : -mx;
                const float sub = (XL && !shifted) ? mx : 0.0f;
                const u32x4 ones_w = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};  // eight bf16 1.0
                auto tile = [&](int kt, const f32x16 &t) __attribute__((always_inline)) {
                    float p[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) p[e] = __builtin_amdgcn_exp2f(XL && !shifted ? t[e] - sub : t[e]);
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        u32x4 pw = {pack2(p[8 * s2], p[8 * s2 + 1]), pack2(p[8 * s2 + 2], p[8 * s2 + 3]),
                                    pack2(p[8 * s2 + 4], p[8 * s2 + 5]), pack2(p[8 * s2 + 6], p[8 * s2 + 7])};
                        const char *vb = Vs + (krow0 + kt * 32 + 16 * s2) * ROWB + v_off;
                        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(vb));
                        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(vb + 8 * ROWB));
                        typedef __attribute__((ext_vector_type(8))) short s16x8;
                        const s16x8 vv = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                        o = mfma32(__builtin_bit_cast(bf16x8, vv), as_bf16x8(pw), o);
                        if (!ones_col) lsum = mfma32(as_bf16x8(ones_w), as_bf16x8(pw), lsum);  // every row = sum over this tile's keys, column = query
                    }
                };
                if (shifted) {
                    if (!LONG) tile(0, scores(0, zero));
                    else if (!XL) {
#pragma unroll 2
                        for (int kt = 0; kt < nkt; ++kt) tile(kt, scores(kt, zero));  // (fully unrolled, hipcc hoists every fragment address and spills)
                    } else if (!LSL_ATTN_XL_PIPE || KS != 2 || !ONES) {  // (the form without the denominator column has two more MFMAs and 16 more live registers per tile: it spills at 128)
#pragma unroll LSL_ATTN_XL_UNROLL
                        for (int kt = 0; kt < nkt; ++kt) tile(kt, scores(kt, zero));
                    } else {
                        // Chunked keys (T = 1000: 32 key tiles per query tile): the loop is compute-bound, and on this chip a wave's vector
                        // instructions overlap only with its OWN MFMAs (profiles/r06_experiments.txt section 5) - so the exponentials of key tile
                        // kt sit, a quarter at a time, behind the score MFMAs of tile kt + 1 and the P.V MFMAs of tile kt, each group fenced.
                        // Same operations on every element in the same order as the plain loop: same bits.
                        auto kfrag = [&](int kt, int c) __attribute__((always_inline)) {
                            return as_bf16x8(*reinterpret_cast<const u32x4 *>(Ks + k_swz<HDP>(krow0 + kt * 32 + r, c + hf)));
                        };
                        auto vfrag = [&](int kt, int s2) __attribute__((always_inline)) {
                            const char *vb = Vs + (krow0 + kt * 32 + 16 * s2) * ROWB + v_off;
                            const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(vb));
                            const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(vb + 8 * ROWB));
                            typedef __attribute__((ext_vector_type(8))) short s16x8;
                            const s16x8 vv = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                            return __builtin_bit_cast(bf16x8, vv);
                        };
                        auto exp4 = [&](const f32x16 &t, int q, float (&p)[4]) __attribute__((always_inline)) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) p[e] = __builtin_amdgcn_exp2f(t[4 * q + e]);
                            asm volatile("" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]));  // (computed HERE: hipcc otherwise sinks a quarter to its use, out of its MFMA shadow)
                        };
                        f32x16 t = scores(0, zero);
                        bf16x8 k0 = kfrag(nkt > 1 ? 1 : 0, 0), k1 = kfrag(nkt > 1 ? 1 : 0, 2);
                        // MFMA order per step: P.V second half of tile kt - 1 (carried: pwc, vbc; the first step's is 0 x 0 onto the accumulator: no
                        // bit changes), the two score MFMAs of tile kt + 1, P.V first half of tile kt - a quarter of tile kt's exponentials and
                        // two packs behind each: 40 vector cycles per 32-cycle MFMA, nothing of the loop outside an MFMA shadow
                        u32x4 pwc = {0u, 0u, 0u, 0u};
                        bf16x8 vbc = as_bf16x8(pwc);
                        auto step = [&](const f32x16 &ti, f32x16 &tn, int kt) __attribute__((always_inline)) {
                            const bf16x8 va = vfrag(kt, 0);
                            float pa[4], pb[4];
                            o = mfma32(vbc, as_bf16x8(pwc), o);
                            exp4(ti, 0, pa);
                            __builtin_amdgcn_sched_barrier(0);
                            vbc = vfrag(kt, 1);
                            tn = mfma32(k0, qf[0], zero);
                            exp4(ti, 1, pb);
                            const u32x4 pw0 = {pack2(pa[0], pa[1]), pack2(pa[2], pa[3]), pack2(pb[0], pb[1]), pack2(pb[2], pb[3])};
                            __builtin_amdgcn_sched_barrier(0);
                            tn = mfma32(k1, qf[1], tn);
                            exp4(ti, 2, pa);
                            const int kn = kt + 2 < nkt ? kt + 2 : kt + 1;  // (the fragments of the tile after the next: read one step ahead)
                            k0 = kfrag(kn, 0);
                            k1 = kfrag(kn, 2);
                            __builtin_amdgcn_sched_barrier(0);
                            o = mfma32(va, as_bf16x8(pw0), o);
                            exp4(ti, 3, pb);
                            pwc = u32x4{pack2(pa[0], pa[1]), pack2(pa[2], pa[3]), pack2(pb[0], pb[1]), pack2(pb[2], pb[3])};
                            if (last_c && kt + 2 == nkt_last && ragged) {  // (uniform) the chunk's last tile holds clamped rows past the sequence
#pragma unroll
                                for (int e = 0; e < 16; ++e)
                                    if (acc_row(e, hf) >= last_rows) tn[e] = -INFINITY;
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        };
                        f32x16 t2;  // (steps in pairs: the two score tiles swap roles, no copies)
                        int kt = 0;
#pragma unroll 1
                        for (; kt + 2 < nkt; kt += 2) {
                            step(t, t2, kt);
                            step(t2, t, kt + 1);
                        }
                        if (kt + 1 < nkt) {  // (uniform)
                            step(t, t2, kt);
                            t = t2;
                        }
                        o = mfma32(vbc, as_bf16x8(pwc), o);
                        tile(nkt - 1, t);
                    }
                } else {
#pragma unroll 1
                    for (int kt = 0; kt < nkt; ++kt) tile(kt, scores(kt, XL ? zero : negmx));
                }
            }
            wait_vmcnt<0>();  // the next stage's rows (and, behind the last chunk, the next unit's query tile): requested a whole stage ago
            asm volatile("" ::: "memory");
            if (!last_c || more) patch_ones(buf ^ 1);
            buf ^= 1;
        }
        if (more) read_q(qn);
        if (has_tile) {
            float l = lsum[0];
            if (ones_col) {  // row 24 of O^T: register 12 of the lanes with hf == 0; the other half gets it by one exchange
                const float mine = o[12], other = xhalf(mine);
                l = hf ? other : mine;
                if (!hf) o[12] = 0.0f;  // the padding channel itself stays zero in z
            }
            const float inv_l = 1.0f / l;
            // O^T -> the query image, rows = queries: lane (query r, half hf) writes channels 8 q4 + 4 hf .. + 3 (8 bytes) into the 16-byte slot of
            // chunk q4; then row-wise, 16 bytes per lane
#pragma unroll
            for (int q4 = 0; q4 < HDP / 8; ++q4) {
                const u32x2 pk = {pack2(o[4 * q4] * inv_l, o[4 * q4 + 1] * inv_l), pack2(o[4 * q4 + 2] * inv_l, o[4 * q4 + 3] * inv_l)};
                *reinterpret_cast<u32x2 *>(qimg + k_swz<HDP>(r, q4) + 8 * hf) = pk;
            }
            const char *zb = uni_ptr(reinterpret_cast<const char *>(a.z) + 2 * ((tok0 + (size_t)(256 * qg) * a.pos_stride) * a.zw + (size_t)head0 * HDP));
#pragma unroll
            for (int k = 0; k < OPW; ++k) {
                const int R = RPI * k + lane / CPR;
                const u32x4 v = *reinterpret_cast<const u32x4 *>(qimg + k_swz<HDP>(R, lane % CPR));
                const unsigned vz = voff_zr[k];
                if ((QIMG >= 1024 || lane < QIMG / 16) && (LONG ? 32 * wave : 0) + R < (LONG ? (last_g ? qrows_last : 256) : S) &&
                    (!grouped || (long)tok0 + R < (long)a.n_tok)) {
                    if (a.nt) asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(vz), "v"(v), "s"(zb) : "memory");
                    else asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(vz), "v"(v), "s"(zb) : "memory");
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the image has been read before the next query tile is requested into it
        }
    }
    wait_vmcnt<0>();
}

// ---------------------------------------------------------------------------------------------------
// Tiny sequences (S <= 8: spatial attention of the pedestrian (L = 2), NBA (L = 8) and peptide (L = 2) models).
// A 32 x 32 MFMA tile would be >= 94 % padding; here one lane owns one (query token, head): S dot products of
// head_dim, softmax over <= 8 scores in registers, S axpys.  K/V rows of a sequence are shared by its S lanes
// and come from L1/L2; lanes are ordered (token, head) so a wave reads whole token rows.
template <int HDP>
__global__ void __launch_bounds__(256) k_attention_tiny(AttnArgs a) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int S = a.S;
    if (idx >= (long)a.n_seq * S * a.H) return;
    const int head = (int)(idx % a.H);
    const long qi = idx / a.H;
    const int seq = (int)(qi / S), pos = (int)(qi % S);
    const size_t tok0 = (size_t)(seq / a.inner) * a.outer_stride + (seq % a.inner);
    const size_t rs = (size_t)3 * a.HHD;
    const u16 *base = a.qkv + tok0 * rs + head * HDP;
    float q[HDP];
    {
        const u16 *qrow = base + (size_t)pos * a.pos_stride * rs;
#pragma unroll
        for (int c = 0; c < HDP / 8; ++c) {
            const u32x4 w = *reinterpret_cast<const u32x4 *>(qrow + 8 * c);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                q[8 * c + 2 * k] = __uint_as_float(w[k] << 16);
                q[8 * c + 2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
            }
        }
    }
    float sc[8];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sc[j] = -INFINITY;
        if (j < S) {
            const u16 *krow = base + a.HHD + (size_t)j * a.pos_stride * rs;
            float d = 0.0f;
#pragma unroll
            for (int c = 0; c < HDP / 8; ++c) {
                const u32x4 w = *reinterpret_cast<const u32x4 *>(krow + 8 * c);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    d = fmaf(q[8 * c + 2 * k], __uint_as_float(w[k] << 16), d);
                    d = fmaf(q[8 * c + 2 * k + 1], __uint_as_float(w[k] & 0xffff0000u), d);
                }
            }
            sc[j] = d;
            mx = fmaxf(mx, d);
        }
    }
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sc[j] = j < S ? __builtin_amdgcn_exp2f(sc[j] - mx) : 0.0f;
        sum += sc[j];
    }
    // the reference rounds the probabilities to the value dtype only implicitly (fp32 SDPA); keep p in fp32 here
    float o[HDP];
#pragma unroll
    for (int d = 0; d < HDP; ++d) o[d] = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (j < S) {
            const u16 *vrow = base + 2 * a.HHD + (size_t)j * a.pos_stride * rs;
#pragma unroll
            for (int c = 0; c < HDP / 8; ++c) {
                const u32x4 w = *reinterpret_cast<const u32x4 *>(vrow + 8 * c);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    o[8 * c + 2 * k] = fmaf(sc[j], __uint_as_float(w[k] << 16), o[8 * c + 2 * k]);
                    o[8 * c + 2 * k + 1] = fmaf(sc[j], __uint_as_float(w[k] & 0xffff0000u), o[8 * c + 2 * k + 1]);
                }
            }
        }
    }
    const float inv = 1.0f / sum;
    u16 *dst = a.z + (tok0 + (size_t)pos * a.pos_stride) * a.zw + head * HDP;
#pragma unroll
    for (int c = 0; c < HDP / 8; ++c) {
        u32x4 w = {pack2(o[8 * c] * inv, o[8 * c + 1] * inv), pack2(o[8 * c + 2] * inv, o[8 * c + 3] * inv),
                   pack2(o[8 * c + 4] * inv, o[8 * c + 5] * inv), pack2(o[8 * c + 6] * inv, o[8 * c + 7] * inv)};
        *reinterpret_cast<u32x4 *>(dst + 8 * c) = w;
    }
}

// ---------------------------------------------------------------------------------------------------
// attention_mode other than "scaled_dot_product" (mmdit.py:50-53 -> attention_linear, mmdit.py:58-72; no shipped config selects it):
//   q <- softmax over the head channels, k <- softmax over the POSITIONS of the sequence (per channel), q <- q hd^-1/2,
//   context[d][e] = sum_n k[n][d] v[n][e],   out[n][e] = sum_d q[n][d] context[d][e]
// on the normalised + rotated q / k that linear1 leaves (bf16, q WITHOUT the softmax pre-multiplier: the host passes 1).  One 256-thread
// workgroup per (sequence, head), fp32 throughout, three passes over the unit's rows (L2-resident): channel maxima of k; the context
// (positions staged 64 at a time as exp(k - max) | v, a thread owns HDP^2 / 256 context entries); the outputs (a thread owns a position).
// S^2 never appears: 2 S hd^2 multiply-adds per unit.  Padded channels (hd < HDP) hold zeros in q / k / v and take no part in either softmax.
template <int HDP>
__global__ void __launch_bounds__(256) k_attention_linear(AttnArgs a) {
    constexpr int CH = 64, EPT = HDP * HDP / 256, TPD = HDP / EPT, G = 256 / HDP, VPR = HDP / 8;
    __shared__ float Ke[CH][HDP + 1], Vs[CH][HDP + 4], ctx[HDP][HDP + 1], red[G][HDP], kmx[HDP];
    const int tid = threadIdx.x, S = a.S, hd = a.hd;
    const size_t rs = 3 * (size_t)a.HHD;
    const float qscale = rsqrtf((float)hd);
    for (long u = blockIdx.x; u < (long)a.n_seq * a.H; u += gridDim.x) {
        const int seq = (int)(u / a.H), head = (int)(u % a.H);
        const size_t tok0 = (size_t)(seq / a.inner) * a.outer_stride + (seq % a.inner);
        const u16 *base = a.qkv + tok0 * rs + head * HDP;  // q of position 0; k at + HHD, v at + 2 HHD; position n at + n pos_stride rs
        {  // channel maxima of k over the positions
            const int d = tid % HDP, g = tid / HDP;
            float m = -INFINITY;
            for (int n = g; n < S; n += G) m = fmaxf(m, bf2f(base[a.HHD + (size_t)n * a.pos_stride * rs + d]));
            red[g][d] = m;
            __syncthreads();
            if (tid < HDP) {
#pragma unroll
                for (int j = 1; j < G; ++j) m = fmaxf(m, red[j][tid]);  // (tid < HDP: g == 0, m = red[0][tid])
                kmx[tid] = m;
            }
            __syncthreads();
        }
        const int d = tid / TPD, e0 = (tid % TPD) * EPT;
        float acc[EPT], zsum = 0.0f;
#pragma unroll
        for (int j = 0; j < EPT; ++j) acc[j] = 0.0f;
        for (int c0 = 0; c0 < S; c0 += CH) {
            for (int i = tid; i < CH * VPR; i += 256) {  // 8 channels of one position per item
                const int p = i / VPR, c8 = 8 * (i % VPR), n = c0 + p;
                u32x4 kw = {0u, 0u, 0u, 0u}, vw = {0u, 0u, 0u, 0u};
                if (n < S) {
                    const u16 *row = base + (size_t)n * a.pos_stride * rs + c8;
                    kw = *reinterpret_cast<const u32x4 *>(row + a.HHD);
                    vw = *reinterpret_cast<const u32x4 *>(row + 2 * a.HHD);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = c8 + 2 * k;
                    const float k0 = __uint_as_float(kw[k] << 16), k1 = __uint_as_float(kw[k] & 0xffff0000u);
                    Ke[p][c] = (n < S && c < hd) ? __expf(k0 - kmx[c]) : 0.0f;
                    Ke[p][c + 1] = (n < S && c + 1 < hd) ? __expf(k1 - kmx[c + 1]) : 0.0f;
                    Vs[p][c] = __uint_as_float(vw[k] << 16);
                    Vs[p][c + 1] = __uint_as_float(vw[k] & 0xffff0000u);
                }
            }
            __syncthreads();
            const int np = min(CH, S - c0);
            for (int p = 0; p < np; ++p) {
                const float kv = Ke[p][d];
                zsum += kv;
#pragma unroll
                for (int j = 0; j < EPT; ++j) acc[j] = fmaf(kv, Vs[p][e0 + j], acc[j]);
            }
            __syncthreads();
        }
        {
            const float inv = zsum > 0.0f ? 1.0f / zsum : 0.0f;  // (padded channels: every term is zero)
#pragma unroll
            for (int j = 0; j < EPT; ++j) ctx[d][e0 + j] = acc[j] * inv;
        }
        __syncthreads();
        for (int n = tid; n < S; n += 256) {
            const u16 *row = base + (size_t)n * a.pos_stride * rs;
            float q[HDP], o[HDP];
#pragma unroll
            for (int c = 0; c < VPR; ++c) {
                const u32x4 w = *reinterpret_cast<const u32x4 *>(row + 8 * c);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    q[8 * c + 2 * k] = __uint_as_float(w[k] << 16);
                    q[8 * c + 2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
                }
            }
            float m = -INFINITY, sum = 0.0f;
#pragma unroll
            for (int c = 0; c < HDP; ++c)
                if (c < hd) m = fmaxf(m, q[c]);
#pragma unroll
            for (int c = 0; c < HDP; ++c) {
                q[c] = c < hd ? __expf(q[c] - m) : 0.0f;
                sum += q[c];
                o[c] = 0.0f;
            }
#pragma unroll
            for (int c = 0; c < HDP; ++c) {
#pragma unroll
                for (int e = 0; e < HDP; ++e) o[e] = fmaf(q[c], ctx[c][e], o[e]);
            }
            const float sc = qscale / sum;
            u16 *dst = a.z + (tok0 + (size_t)n * a.pos_stride) * a.zw + head * HDP;
#pragma unroll
            for (int c = 0; c < VPR; ++c) {
                const u32x4 w = {pack2(o[8 * c] * sc, o[8 * c + 1] * sc), pack2(o[8 * c + 2] * sc, o[8 * c + 3] * sc),
                                 pack2(o[8 * c + 4] * sc, o[8 * c + 5] * sc), pack2(o[8 * c + 6] * sc, o[8 * c + 7] * sc)};
                *reinterpret_cast<u32x4 *>(dst + 8 * c) = w;
            }
        }
        __syncthreads();  // ctx, red and kmx are rewritten by the next unit
    }
}
