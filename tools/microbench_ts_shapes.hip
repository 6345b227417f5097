// The token-stationary inner loop of k_linear1_ts / k_tail with two MFMA shapes (round 6, review item: "v_mfma_f32_16x16x32_bf16 in the block
// loop"): a wave keeps 32 tokens x K activations in registers as B fragments; per block of 32 features it reads the block's A fragments from
// LDS (lane-linear 1 KiB pieces, PD ahead) and accumulates one 32 x 32 output tile.  No epilogue, no DMA, no barrier: the ceiling of the loop
// body alone, on random operands (the clock the chip holds depends on the data: MI355X_MICROARCH.md).
//   shape 0: v_mfma_f32_32x32x16_bf16  - K / 16 MFMAs per block, one accumulator tile of 16 registers
//   shape 1: v_mfma_f32_16x16x32_bf16  - 2 x 2 tiles of 16 x 16 per k-step of 32: 4 MFMAs per 2 KiB of A fragments, four 4-register tiles
//   shape 2: the K-split pair (review item, second structure): a wave keeps 64 tokens x K / 2 (the same K / 4 registers), waves w and w ^ 1 share
//            the 64 tokens and split K; per block a wave reads HALF of the block's fragments and feeds each to two MFMAs (two 32 x 32 tiles),
//            then hands one partial tile to its partner through LDS (4 KiB written, 4 KiB read, 16 adds; one workgroup barrier per block,
//            which the product kernel has anyway): 16 + 8 = 24 KiB of LDS traffic per block instead of 32, the same MFMAs
// Same FLOPs, same registers per block in all; shapes 0 and 1 read the same LDS bytes.
//   build: hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/microbench_ts_shapes.hip -o tools/_exp/mb_ts
//   run:   tools/_exp/mb_ts [K] [blocks per launch] [waves per workgroup 8|4]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../lam_slide_amd/csrc/common.hip.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int K, int SHAPE, int NW>
__global__ void __launch_bounds__(NW * 64, NW / 4) probe(float *out, const unsigned *seed, int blocks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = K / 16, PD = 3;
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 32 * K / 2 * 2; i += NW * 64) reinterpret_cast<unsigned *>(smem)[i] = 0x3f803f80u ^ ((i * 2654435761u + seed[0]) & 0x00ff00ffu);  // two blocks of bf16 near 1
    __syncthreads();
    bf16x8 xreg[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xreg[ks] = as_bf16x8(u32x4{0x3f803f80u ^ (lane * 40503u & 0x00ff00ffu), 0x3f813f7fu, 0x3f7e3f82u ^ (ks * 0x1003u & 0x001f001fu), 0x3f803f80u});
    float sink = 0.0f;
    const char *sb0 = smem + lane * 16;
    if (SHAPE == 2) {
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), khalf = wave & 1;
        char *const hand = smem + 2 * 32 * K * 2;  // [2 slots][NW waves][4 KiB]
        f32x16 mine;                               // the tile this wave finishes: its own partial sum of the previous block
#pragma unroll
        for (int e = 0; e < 16; ++e) mine[e] = 0.0f;
        for (int b = 0; b < blocks; ++b) {
            const char *sb = sb0 + (b & 1) * (32 * K * 2) + khalf * (KS / 2) * 1024;
            f32x16 acc[2];
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[0][e] = acc[1][e] = 0.0f;
            // the partner's partial tile of block b - 1 (written before the barrier that ended that block)
            f32x4 theirs[4];
            const char *hp = hand + ((b + 1) & 1) * (NW * 4096) + (wave ^ 1) * 4096 + lane * 16;
            if (b > 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) theirs[q] = *reinterpret_cast<const f32x4 *>(hp + 1024 * q);
            }
            bf16x8 fr[PD];
#pragma unroll
            for (int f = 0; f < PD; ++f) fr[f] = as_bf16x8(*reinterpret_cast<const u32x4 *>(sb + f * 1024));
#pragma unroll
            for (int f = 0; f < KS / 2; ++f) {
                acc[0] = mfma32(fr[f % PD], xreg[f], acc[0]);
                acc[1] = mfma32(fr[f % PD], xreg[KS / 2 + f], acc[1]);
                if (f + PD < KS / 2) fr[f % PD] = as_bf16x8(*reinterpret_cast<const u32x4 *>(sb + (f + PD) * 1024));
            }
            if (b > 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) sink += mine[4 * q + e] + theirs[q][e];
            }
            // hand the partner's tile over (tile 1 - khalf), keep the other
            char *hw = hand + (b & 1) * (NW * 4096) + wave * 4096 + lane * 16;
            if (khalf) {  // (uniform)
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4 *>(hw + 1024 * q) = f32x4{acc[0][4 * q], acc[0][4 * q + 1], acc[0][4 * q + 2], acc[0][4 * q + 3]};
                mine = acc[1];
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4 *>(hw + 1024 * q) = f32x4{acc[1][4 * q], acc[1][4 * q + 1], acc[1][4 * q + 2], acc[1][4 * q + 3]};
                mine = acc[0];
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if (SHAPE == 0 || SHAPE == 3) {  // (3: shape 0 with the product kernel's one workgroup barrier per block)
        for (int b = 0; b < blocks; ++b) {
            const char *sb = sb0 + (b & 1) * (32 * K * 2);
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
            bf16x8 fr[PD];
#pragma unroll
            for (int f = 0; f < PD; ++f) fr[f] = as_bf16x8(*reinterpret_cast<const u32x4 *>(sb + f * 1024));
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                acc = mfma32(fr[ks % PD], xreg[ks], acc);
                if (ks + PD < KS) fr[ks % PD] = as_bf16x8(*reinterpret_cast<const u32x4 *>(sb + (ks + PD) * 1024));
            }
            sink += acc[0] + acc[7] + acc[15];
            if (SHAPE == 3) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        for (int b = 0; b < blocks; ++b) {
            const char *sb = sb0 + (b & 1) * (32 * K * 2);
            f32x4 acc[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.0f;
            bf16x8 fr[PD];  // piece f = (k-step of 32: f / 2, feature half: f & 1)
#pragma unroll
            for (int f = 0; f < PD; ++f) fr[f] = as_bf16x8(*reinterpret_cast<const u32x4 *>(sb + f * 1024));
#pragma unroll
            for (int f = 0; f < KS; ++f) {  // KS pieces of 1 KiB per block, like shape 0
                const int k2 = f >> 1, mi = f & 1;
                // the wave's two token tiles of k-step k2: registers of xreg[2 k2] and xreg[2 k2 + 1] (any fixed assignment: timing only)
                acc[mi][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[f % PD], xreg[2 * k2], acc[mi][0], 0, 0, 0);
                acc[mi][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[f % PD], xreg[2 * k2 + 1], acc[mi][1], 0, 0, 0);
                if (f + PD < KS) fr[f % PD] = as_bf16x8(*reinterpret_cast<const u32x4 *>(sb + (f + PD) * 1024));
            }
            sink += acc[0][0][0] + acc[0][1][1] + acc[1][0][2] + acc[1][1][3];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (sink == 123.456f) out[blockIdx.x * NW * 64 + tid] = sink;
}

template <int K, int SHAPE, int NW>
void run(int blocks) {
    auto kern = probe<K, SHAPE, NW>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    float *out;
    unsigned *seed;
    hipMalloc(&out, 256 * NW * 64 * 4);
    hipMalloc(&seed, 4);
    unsigned s = 12345u;
    hipMemcpy(seed, &s, 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t lds = 2 * 32 * K * 2 + (SHAPE == 2 ? 2 * NW * 4096 : 0);
    for (int round = 0; round < 3; ++round) {
        hipEventRecord(e0, 0);
        for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(kern, dim3(256), dim3(NW * 64), lds, 0, out, seed, blocks);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = 2.0 * 32 * 32 * K * (double)blocks * NW * 256;
        printf("  K=%d %s waves/wg=%d round %d: %.3f ms/launch, %.0f TFLOP/s (%.3f of 2.5 PF)\n", K, SHAPE == 3 ? "32x32x16 + barrier" : SHAPE == 2 ? "32x32x16, K-split pair" : SHAPE ? "16x16x32" : "32x32x16", NW, round, ms / 10,
               flop / (ms / 10 * 1e-3) * 1e-12, flop / (ms / 10 * 1e-3) * 1e-12 / 2500);
    }
    hipFree(out);
    hipFree(seed);
}

int main(int argc, char **argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 512, blocks = argc > 2 ? atoi(argv[2]) : 4000, nw = argc > 3 ? atoi(argv[3]) : 8;
    if (K == 512 && nw == 8) { run<512, 0, 8>(blocks); run<512, 1, 8>(blocks); run<512, 2, 8>(blocks); run<512, 3, 8>(blocks); }
    else if (K == 256 && nw == 8) { run<256, 0, 8>(blocks); run<256, 1, 8>(blocks); run<256, 2, 8>(blocks); run<256, 3, 8>(blocks); }
    else if (K == 512 && nw == 4) { run<512, 0, 4>(blocks); run<512, 1, 4>(blocks); run<512, 2, 4>(blocks); run<512, 3, 4>(blocks); }
    else if (K == 256 && nw == 4) { run<256, 0, 4>(blocks); run<256, 1, 4>(blocks); run<256, 2, 4>(blocks); run<256, 3, 4>(blocks); }
    else printf("unsupported\n");
    return 0;
}
