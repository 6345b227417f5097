// Ceiling probes for the GEMM inner loop on gfx950 (build: hipcc -O3 --offload-arch=gfx950 tools/microbench_mfma.hip -o /tmp/mb).
//   mode 0: bare v_mfma_f32_32x32x16_bf16, MI x NJ accumulators, operands in registers
//   mode 1: the production sub-step (MI + NJ ds_read_b128 fragments per MI*NJ MFMAs, double-buffered) on LDS-resident data
//   mode 2: mode 1 + one s_barrier per 4 sub-steps (the k-tile cadence of the product kernel)
// Reports TFLOP/s for 256 workgroups x {4, 8} waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../lam_slide_amd/csrc/k_gemm.hip.h"

template <int MI, int NJ, int MODE, int NW>
__global__ void __launch_bounds__(NW * 64) probe(float *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hf = lane >> 5;
    for (int i = tid; i < 512 * 32; i += NW * 64) reinterpret_cast<unsigned *>(smem)[i] = 0x3f803f80u ^ (i * 2654435761u & 0x00ff00ffu);
    __syncthreads();
    f32x16 acc[MI][NJ];
    for (int i = 0; i < MI; ++i)
        for (int j = 0; j < NJ; ++j)
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    bf16x8 a0[MI], b0[NJ], a1[MI], b1[NJ];
    const int wf = wave & 1, wt = wave >> 1;
    auto load = [&](int ks, bf16x8(&a)[MI], bf16x8(&b)[NJ]) {
        for (int i = 0; i < MI; ++i) a[i] = as_bf16x8(*reinterpret_cast<const u32x4 *>(smem + swz_bk<64>((wf * MI * 32 + i * 32 + r) & 255, 2 * ks + hf)));
        for (int j = 0; j < NJ; ++j) b[j] = as_bf16x8(*reinterpret_cast<const u32x4 *>(smem + swz_bk<64>(256 + ((wt * NJ * 32 + j * 32 + r) & 255), 2 * ks + hf)));
    };
    auto mm = [&](const bf16x8(&a)[MI], const bf16x8(&b)[NJ]) {
        for (int i = 0; i < MI; ++i)
            for (int j = 0; j < NJ; ++j) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
    };
    load(0, a0, b0);
    load(1, a1, b1);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 2) __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int ks = 0; ks < 4; ks += 2) {
            if (MODE >= 1) load(ks + 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            mm(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE >= 1) load((ks + 2) & 3, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            mm(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < MI; ++i)
        for (int j = 0; j < NJ; ++j)
            for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    if (s == 12345.678f) out[tid] = s;
}

// the same 128 x 64 wave tile on v_mfma_f32_16x16x32_bf16: 8 x 4 accumulator tiles of 4 VGPRs, per 32-deep step 8 + 4 ds_read_b128 and
// 32 MFMAs (MI355X_MICROARCH.md: this shape holds a higher clock under load than 32x32x16)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
template <int MODE, int NW>
__global__ void __launch_bounds__(NW * 64) probe16(float *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rr = lane & 15, kq = lane >> 4;
    for (int i = tid; i < 512 * 32; i += NW * 64) reinterpret_cast<unsigned *>(smem)[i] = 0x3f803f80u ^ (i * 2654435761u & 0x00ff00ffu);
    __syncthreads();
    f32x4_t acc[8][4];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 4; ++j)
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    bf16x8 a[8], b[4];
    const int wf = wave & 1, wt = wave >> 1;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 2) __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {  // two 32-deep steps per 64-deep k-tile
            if (MODE >= 1 || (it == 0 && ks == 0)) {
                for (int i = 0; i < 8; ++i) a[i] = as_bf16x8(*reinterpret_cast<const u32x4 *>(smem + swz_bk<64>((wf * 128 + i * 16 + rr) & 255, 4 * ks + kq)));
                for (int j = 0; j < 4; ++j) b[j] = as_bf16x8(*reinterpret_cast<const u32x4 *>(smem + swz_bk<64>(256 + ((wt * 64 + j * 16 + rr) & 255), 4 * ks + kq)));
            }
            __builtin_amdgcn_sched_barrier(0);
            for (int i = 0; i < 8; ++i)
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 4; ++j)
            for (int e = 0; e < 4; ++e) s += acc[i][j][e];
    if (s == 12345.678f) out[tid] = s;
}

template <int MODE, int NW>
void run16(const char *name, float *out) {
    const int iters = 2000, grid = 256;
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe16<MODE, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe16<MODE, NW>), dim3(grid), dim3(NW * 64), 65536, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * NW * iters * 2 * 32 * 16384.0;
    printf("%-44s %8.1f TFLOP/s  (%.3f ms)\n", name, flops / ms / 1e9, ms);
}

template <int MI, int NJ, int MODE, int NW>
void run(const char *name, float *out) {
    const int iters = 2000, grid = 256 * (NW == 4 ? 2 : 1);
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe<MI, NJ, MODE, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<MI, NJ, MODE, NW>), dim3(grid), dim3(NW * 64), 65536, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * NW * iters * 4 * MI * NJ * 32768.0;
    printf("%-44s %8.1f TFLOP/s  (%.3f ms)\n", name, flops / ms / 1e9, ms);
}

int main() {
    float *out;
    hipMalloc(&out, 1 << 20);
    run<4, 2, 0, 8>("bare MFMA 4x2 tiles, 8 waves/CU", out);
    run<2, 2, 0, 8>("bare MFMA 2x2 tiles, 8 waves/CU", out);
    run<2, 2, 0, 4>("bare MFMA 2x2 tiles, 2 WG x 4 waves/CU", out);
    run<4, 2, 1, 8>("LDS frags + MFMA 4x2, 8 waves/CU", out);
    run<2, 2, 1, 8>("LDS frags + MFMA 2x2, 8 waves/CU", out);
    run<2, 2, 1, 4>("LDS frags + MFMA 2x2, 2 WG x 4 waves/CU", out);
    run<4, 2, 2, 8>("LDS frags + MFMA 4x2 + barrier, 8 waves/CU", out);
    run<2, 2, 2, 4>("LDS frags + MFMA 2x2 + barrier, 2x4 waves", out);
    run16<0, 8>("16x16x32: bare MFMA 8x4 tiles, 8 waves/CU", out);
    run16<1, 8>("16x16x32: LDS frags + MFMA, 8 waves/CU", out);
    run16<2, 8>("16x16x32: LDS frags + MFMA + barrier, 8 waves", out);
    run<4, 2, 2, 8>("32x32x16 again: frags + MFMA 4x2 + barrier", out);
    return 0;
}
