// Do the MFMA chain of one wave and the vector work of its SIMD partner overlap?  (round 6: in k_tail and k_linear1_ts the matrix-pipe time,
// the vector-pipe time and the ring skeleton ADD UP - profiles/r06_tail_experiments.txt section 3 - where the design assumed max().)
// One workgroup of 8 waves per CU (two per SIMD: waves w and w + 4 share one).  Roles:
//   M  a chain of v_mfma_f32_32x32x16_bf16 on register operands (DEP = 1: one accumulator, every MFMA depends on the last; DEP = 0: 8 accumulators)
//   V  the GELU pair form of common.hip.h on 16 values per iteration (13 vector instructions per pair, one dependent chain per pair)
// Modes: 0 waves 0-3 run M, waves 4-7 exit;  1 waves 0-3 exit, waves 4-7 run V;  2 both at once;  3 every wave runs M then V (the in-order sum);
//        4 every wave runs M and V interleaved in ONE instruction stream (2 GELU pairs behind every 4 MFMAs).
//   build: hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/microbench_coissue.hip -o tools/_exp/mb_coissue
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "../lam_slide_amd/csrc/common.hip.h"

#ifndef VKIND
#define VKIND 0  // the vector work: 0 gelu_pair_bf16 (packed-fp32 polynomial), 1 two gelu_fast + pack (scalar FMAs), 2 13 plain v_fma_f32 per pair, 3 4 v_exp_f32 per pair, 4 both (independent)
#endif
__device__ __forceinline__ unsigned vwork(float x0, float x1) {
    if (VKIND == 0) return gelu_pair_bf16(x0, x1);
    if (VKIND == 1) return pack2(gelu_fast(x0), gelu_fast(x1));
    if (VKIND == 2) {
        float a = x0, b = x1;
#pragma unroll
        for (int k = 0; k < 6; ++k) { a = fmaf(a, 0.99f, b); b = fmaf(b, 1.01f, a); }
        return __float_as_uint(fmaf(a, b, 1.0f));
    }
    if (VKIND == 4) {  // kinds 2 and 3 together, independent of each other: do the transcendental and the plain vector instructions of ONE wave overlap?
        float a = x0, b = x1;
#pragma unroll
        for (int k = 0; k < 6; ++k) { a = fmaf(a, 0.99f, b); b = fmaf(b, 1.01f, a); }
        return __float_as_uint(fmaf(a, b, 1.0f)) ^ __float_as_uint(__builtin_amdgcn_exp2f(x0) + __builtin_amdgcn_exp2f(x1)) ^
               __float_as_uint(__builtin_amdgcn_exp2f(x0 * 0.5f) + __builtin_amdgcn_exp2f(x1 * 0.5f));
    }
    return __float_as_uint(__builtin_amdgcn_exp2f(x0) + __builtin_amdgcn_exp2f(x1)) ^ __float_as_uint(__builtin_amdgcn_exp2f(x0 * 0.5f) + __builtin_amdgcn_exp2f(x1 * 0.5f));
}

template <int MODE, int DEP>
__global__ void __launch_bounds__(512, 2) probe(float *out, const float *in, int iters) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool do_m = MODE == 0 ? wave < 4 : MODE == 1 ? false : MODE == 2 ? wave < 4 : true;
    if ((MODE == 7 || MODE == 8 || MODE == 10) && wave >= 4) return;  // modes 7 / 8 / 10 = modes 5 / 6 / 9 with ONE wave per SIMD
    const bool do_v = MODE == 0 ? false : MODE == 1 ? wave >= 4 : MODE == 2 ? wave >= 4 : true;
    if (!do_m && !do_v) return;
#ifdef VPRIO
    if (MODE == 2) {  // (uniform per wave)
        if (wave >= 4) __builtin_amdgcn_s_setprio(VPRIO);
        else __builtin_amdgcn_s_setprio(MPRIO);
    }
#endif
    bf16x8 a = as_bf16x8(u32x4{0x3f803f80u ^ (lane * 40503u & 0x00ff00ffu), 0x3f813f7fu, 0x3f7e3f82u, 0x3f803f80u});
    bf16x8 b = as_bf16x8(u32x4{0x3f7f3f81u, 0x3f803f80u ^ (lane * 977u & 0x000f000fu), 0x3f803f80u, 0x3f823f7eu});
    f32x16 acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[k][e] = 0.0f;
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = in[(tid * 16 + e) & 1023];
    unsigned sink = 0;
    unsigned long long t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        if (MODE == 9 || MODE == 10) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[DEP ? (q & 1) : 4 * (q & 1) + i] = mfma32(a, b, acc[DEP ? (q & 1) : 4 * (q & 1) + i]);
                sink ^= vwork(v[2 * q], v[2 * q + 1]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);  // then up to four vector instructions
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE == 5 || MODE == 7) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[DEP ? (q & 1) : 4 * (q & 1) + i] = mfma32(a, b, acc[DEP ? (q & 1) : 4 * (q & 1) + i]);
                sink ^= vwork(v[2 * q], v[2 * q + 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE == 6 || MODE == 8) {  // the same work, not interleaved: 32 MFMAs, then the 16 GELUs
#pragma unroll
            for (int i = 0; i < 32; ++i) acc[DEP ? (i >> 4) : i & 7] = mfma32(a, b, acc[DEP ? (i >> 4) : i & 7]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 8; ++s) sink ^= vwork(v[2 * s], v[2 * s + 1]);
            __builtin_amdgcn_sched_barrier(0);
        } else if (MODE == 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[DEP ? 0 : 4 * (q & 1) + i] = mfma32(a, b, acc[DEP ? 0 : 4 * (q & 1) + i]);
                sink ^= vwork(v[4 * q], v[4 * q + 1]);
                sink ^= vwork(v[4 * q + 2], v[4 * q + 3]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            if (do_m) {  // 16 MFMAs
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[DEP ? 0 : i & 7] = mfma32(a, b, acc[DEP ? 0 : i & 7]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (do_v) {  // the GELU of 16 values
#pragma unroll
                for (int s = 0; s < 8; ++s) sink ^= vwork(v[2 * s], v[2 * s + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] += 1e-3f;  // (keeps the vector work from being hoisted)
    }
    unsigned long long t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (blockIdx.x == 7 && lane == 0) reinterpret_cast<unsigned long long *>(out + 256 * 512)[wave] = t1 - t0;  // shader cycles of this wave's loop
    float s = __uint_as_float(sink & 0x007fffffu);
#pragma unroll
    for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][9];
    if (s == 123.456f) out[blockIdx.x * 512 + tid] = s;
}

template <int MODE, int DEP>
float run(float *out, float *in, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int round = 0; round < 3; ++round) {
        hipEventRecord(e0, 0);
        for (int it = 0; it < 5; ++it) hipLaunchKernelGGL((probe<MODE, DEP>), dim3(256), dim3(512), 0, 0, out, in, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms / 5 < best ? ms / 5 : best;
    }
    unsigned long long cyc[8];
    hipMemcpy(cyc, out + 256 * 512, 64, hipMemcpyDeviceToHost);
    printf("    [mode %d: shader cycles per iteration by s_memtime, wave 0 %.0f, wave 4 %.0f]\n", MODE, (double)cyc[0] / iters, (double)cyc[4] / iters);
    return best;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    float *out, *in;
    hipMalloc(&out, 256 * 512 * 4 + 64);
    hipMalloc(&in, 1024 * 4);
    float h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 37) % 200) / 50.0f - 2.0f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    const double cyc = 2.1e6;  // cycles per ms at ~2.1 GHz (the printed cycle figures are nominal)
#ifdef VPRIO
    printf("vector work kind %d, s_setprio: V waves %d, M waves %d\n", VKIND, VPRIO, MPRIO);
#else
    printf("vector work kind %d\n", VKIND);
#endif
    for (int dep = 1; dep >= 1; --dep) {
        const float m = dep ? run<0, 1>(out, in, iters) : run<0, 0>(out, in, iters);
        const float v = dep ? run<1, 1>(out, in, iters) : run<1, 0>(out, in, iters);
        const float both = dep ? run<2, 1>(out, in, iters) : run<2, 0>(out, in, iters);
        const float seq = dep ? run<3, 1>(out, in, iters) : run<3, 0>(out, in, iters);
        const float mix = dep ? run<4, 1>(out, in, iters) : run<4, 0>(out, in, iters);
        printf("%s MFMA chain, per iteration of 16 MFMAs / 16 GELUs (nominal cycles at 2.1 GHz):\n", dep ? "dependent" : "8-accumulator");
        printf("  M alone (one wave per SIMD)           %.3f ms  = %5.0f cycles\n", m, m * cyc / iters);
        printf("  V alone (one wave per SIMD)           %.3f ms  = %5.0f cycles\n", v, v * cyc / iters);
        printf("  M on wave w, V on wave w + 4          %.3f ms  = %5.0f cycles   (max %.0f, sum %.0f)\n", both, both * cyc / iters,
               (m > v ? m : v) * cyc / iters, (m + v) * cyc / iters);
        printf("  every wave M then V (two per SIMD)    %.3f ms  = %5.0f cycles per wave-iteration pair\n", seq, seq * cyc / iters);
        printf("  every wave M and V interleaved        %.3f ms  = %5.0f cycles\n", mix, mix * cyc / iters);
        const float t5 = run<5, 1>(out, in, iters), t6 = run<6, 1>(out, in, iters), t7 = run<7, 1>(out, in, iters), t8 = run<8, 1>(out, in, iters);
        printf("  tail block (32 MFMAs + 16 GELUs per wave), two waves per SIMD: interleaved %.0f, sequential %.0f;  one wave per SIMD: interleaved %.0f, sequential %.0f\n",
               t5 * cyc / iters, t6 * cyc / iters, t7 * cyc / iters, t8 * cyc / iters);
        const float t9 = run<9, 1>(out, in, iters), t10 = run<10, 1>(out, in, iters);
        printf("  the same with an enforced MFMA / 4 x VALU pattern (sched_group_barrier): two waves per SIMD %.0f, one wave per SIMD %.0f\n", t9 * cyc / iters, t10 * cyc / iters);
    }
    return 0;
}
