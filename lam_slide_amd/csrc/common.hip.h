// Shared device helpers for the gfx950 kernels of the LaM-SLidE sampling path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short u16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define LDS_PTR(T) __attribute__((address_space(3))) T *

// Timing probes (skip loads / MFMAs / epilogue phases: results WRONG when set).  The probe words are kernel ARGUMENTS; the host fills them
// from LSL_PROBE / LSL_RES_SKIP only in -DLSL_EXPERIMENTS builds (tune_int in lsl_api.hip) and with a hard 0 in the product library,
// so no environment variable can reach them there.  The run-time tests themselves stay in both builds on purpose: with them folded
// away at compile time hipcc schedules the linear1 kernel into 256 VGPRs + 52 B of scratch (376 ms per cfg-2 step) instead of 231
// VGPRs and no scratch (346 ms) - the never-taken branches bound how far it interleaves the epilogue with the main loop - and
// keeping them makes the tools' A/B builds generate the same code as the product.
#define LSL_PROBE(v, bit) ((v) & (bit))

// counted wait: all but the wave's N youngest vector-memory operations (loads, stores, LDS-DMA count together, in issue order) are done
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// 32x32x16 bf16 MFMA, fp32 accumulate.  Operand maps (lane l: r = l & 31, hf = l >> 5):
//   A[row r][k = 8 hf + j], B[k = 8 hf + j][col r], j = 0..7
//   C/D: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 hf, reg = 0..15
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ int acc_row(int reg, int hf) { return (reg & 3) + 8 * (reg >> 2) + 4 * hf; }

__device__ __forceinline__ u16 f2bf(float f) {
    __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
    return __builtin_bit_cast(u16, b);
}
__device__ __forceinline__ float bf2f(u16 v) { return __uint_as_float(((unsigned)v) << 16); }

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ unsigned pack2(float lo, float hi) {  // one v_cvt_pk_bf16_f32 (round to nearest even, like f2bf)
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ u32x4 as_u32x4(bf16x8 v) { return __builtin_bit_cast(u32x4, v); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// The same sum as ONE value for the whole wave, by DPP moves only (row shifts inside the rows of 16 lanes, then row_bcast15 / row_bcast31 across
// them; the total ends in lane 63 and comes back as a scalar): 6 vector adds + a v_readlane instead of 6 ds_bpermute round trips through the LDS
// queue (wave_sum).  For kernels with few waves per SIMD, where those round trips are exposed.  A different summation tree than wave_sum's.
#define LSL_DPP_ADD(v, ctrl, rows) ((v) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), (rows), 0xF, false)))
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v = LSL_DPP_ADD(v, 0x111, 0xF);  // row_shr:1
    v = LSL_DPP_ADD(v, 0x112, 0xF);  // row_shr:2
    v = LSL_DPP_ADD(v, 0x114, 0xF);  // row_shr:4
    v = LSL_DPP_ADD(v, 0x118, 0xF);  // row_shr:8: lane 15 of every row of 16 lanes now holds the row's sum
    v = LSL_DPP_ADD(v, 0x142, 0xA);  // row_bcast15: rows 1 and 3 add the sum of the row before them
    v = LSL_DPP_ADD(v, 0x143, 0xC);  // row_bcast31: rows 2 and 3 add the sum of rows 0 + 1
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
#undef LSL_DPP_ADD

__device__ __forceinline__ float xhalf(float v) { return __shfl_xor(v, 32, 64); }

// v + (value of lane ^ 32) as one VALU exchange (gfx950 v_permlane32_swap) + one add, instead of ds_bpermute through the LDS queue.
// v_permlane32_swap vdst, src swaps lanes 32-63 of vdst with lanes 0-31 of src; with both = v the results are (low half's value in
// every lane, high half's value in every lane), summed in the same order on both halves: the two lanes of a token get identical bits.
__device__ __forceinline__ float half_pair_sum(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto p = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)p[0]) + __builtin_bit_cast(float, (unsigned)p[1]);
}

// value of lane ^ 1 / lane ^ 2 through DPP quad permutes (a VALU move; __shfl_xor goes through ds_bpermute and the LDS queue)
__device__ __forceinline__ float quad_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_xor2(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
}

// the sum over the 8 consecutive lanes 8 g .. 8 g + 7, in every one of them: two quad permutes and row_half_mirror (lane i <-> lane 7 - i of its
// half row): three vector moves + adds, no LDS queue
__device__ __forceinline__ float oct_sum(float v) {
    v += quad_xor1(v);
    v += quad_xor2(v);
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
}

__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// erf-GELU (mmdit.py: nn.GELU() = x Phi(x)), round-3 form:  gelu(x) = max(x, 0) - |x| Phi(-|x|),  Phi(-a) = 2^q(a)  with q a degree-5
// polynomial fitted to log2 Phi(-a) on a in [0, 9] (minimax-weighted for the error of the PRODUCT a Phi(-a); tools/gelu_fit.py).
// 5 FMA + ONE transcendental + max + FMA = 8 vector instructions (the Abramowitz-Stegun 7.1.26 erfc form of rounds 1-2: 14 with a
// reciprocal and an exponential) - the linear1 epilogue of an mlp block is bound by its vector-instruction count.
//   |gelu - x Phi(x)| <= 7e-7 for every x (fp32 evaluation; 7.1.26: 2e-7), no cancellation on the negative side, and for a > 9 the
//   negative leading coefficient drives q to -inf: 2^q underflows to 0 and gelu(x) = max(x, 0) exactly.
#define LSL_GELU_Q(ax)                                                                                                  \
    fmaf(fmaf(fmaf(fmaf(fmaf(-0.0004733090754598379f, (ax), 0.0070845563896000385f), (ax), -0.051827382296323776f), (ax), \
                   -0.4599924385547638f), (ax), -1.1507878303527832f), (ax), -1.000037670135498f)
__device__ __forceinline__ float gelu_fast(float x) {
    const float ax = fabsf(x);
    const float h = __builtin_amdgcn_exp2f(LSL_GELU_Q(ax));
    // max(x, 0) through a builtin the compiler knows (v_med3_f32 x, 0, +inf, which it lowers to v_max pairs): an inline-asm v_max is
    // invisible to the hazard recognizer - placed right behind the MFMA that produces x it read the accumulator before the matrix pipe
    // had written it (no hardware interlock, no s_nop: wrong and timing-dependent results, found in k_resident in round 2)
    const float relu = __builtin_amdgcn_fmed3f(x, 0.0f, __builtin_inff());
    return fmaf(-ax, h, relu);
}

// erf-GELU of TWO accumulator values, rounded to one packed bf16 pair, with the polynomial and the final fma on the packed-fp32 pipe
// (v_pk_fma_f32: 6 instead of 12 FMAs per pair) and max(x, 0) as ONE integer maximum per value (v_max_i32 on the bit pattern: positive
// floats are positive integers, negative floats negative ones; the fp32 maximum costs hipcc a canonicalising v_max first, and an inline-asm
// v_max_f32 is invisible to the hazard recognizer right behind the MFMA that wrote x): 13 vector instructions per pair where two
// gelu_fast calls + a pack take 21.  Same polynomial and operation order per element as gelu_fast: the same values, except that a
// negative NaN gives 0 instead of NaN.
__device__ __forceinline__ unsigned gelu_pair_bf16(float x0, float x1) {
    const f32x2 x = {x0, x1};
    const f32x2 ax = __builtin_elementwise_abs(x);
    f32x2 q = f32x2{-0.0004733090754598379f, -0.0004733090754598379f};
    q = __builtin_elementwise_fma(q, ax, f32x2{0.0070845563896000385f, 0.0070845563896000385f});
    q = __builtin_elementwise_fma(q, ax, f32x2{-0.051827382296323776f, -0.051827382296323776f});
    q = __builtin_elementwise_fma(q, ax, f32x2{-0.4599924385547638f, -0.4599924385547638f});
    q = __builtin_elementwise_fma(q, ax, f32x2{-1.1507878303527832f, -1.1507878303527832f});
    q = __builtin_elementwise_fma(q, ax, f32x2{-1.000037670135498f, -1.000037670135498f});
    const f32x2 h = {__builtin_amdgcn_exp2f(q[0]), __builtin_amdgcn_exp2f(q[1])};
    const f32x2 relu = {__int_as_float(max(__float_as_int(x0), 0)), __int_as_float(max(__float_as_int(x1), 0))};
    const f32x2 r = __builtin_elementwise_fma(-ax, h, relu);
    return __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
}

// Streaming ("nt") stores: the line is written out without staying resident in L2.  The GEMM epilogues write 0.5-1.3 GB per
// launch through the 4 MiB L2 of each XCD; with plain stores that evicted the operand tiles every round (linear1 fetched
// 1.1 GB per launch for 0.25 GB of operands; with nt stores 0.32 GB).  Inline asm on purpose: when a branch selects between
// __builtin_nontemporal_store and a plain store of the same value, hipcc merges the two and drops the hint.
__device__ __forceinline__ void store16(void *p, u32x4 v, bool nt) {
    if (nt) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else *reinterpret_cast<u32x4 *>(p) = v;
}
__device__ __forceinline__ void store8(void *p, u32x2 v, bool nt) {
    if (nt) asm volatile("global_store_dwordx2 %0, %1, off nt\n\ts_nop 0" ::"v"(p), "v"(v) : "memory");
    else *reinterpret_cast<u32x2 *>(p) = v;
}

// 16-byte loads that are served by L2, past this CU's vector L1 (sc1 loads bypass L1 only: MI355X_MICROARCH.md, visibility table): for
// rows this workgroup has just rewritten while L1 may still hold their old lines.  Buffer form (aux 16 = sc1) because hipcc counts it
// in its s_waitcnt bookkeeping (an inline-asm load would not be) and drops the hint of __builtin_nontemporal_load on 16-byte vectors.
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
struct L2Reader {
    __amdgpu_buffer_rsrc_t rsrc;
    __device__ __forceinline__ explicit L2Reader(const void *base) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0xFFFFFFFFu, 0x00020000);
    }
    __device__ __forceinline__ float4 load16(unsigned byte_offset) const {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_offset, 0, 16);
        return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
    }
};

// Workgroup id remap so that consecutive logical tiles share an XCD's L2.  Hardware deals
// workgroups round-robin over the 8 XCDs; this only affects speed, never results.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
