// HBM bandwidth ceilings on MI355X for the access patterns of the GEMM epilogues (tools only, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_hbm.hip -o gpurun_out/mb_hbm && gpurun_out/mb_hbm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void k_write(u32x4 *p, size_t n16, unsigned v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = u32x4{v, v, v, v};
}
__global__ void k_read(const u32x4 *p, size_t n16, unsigned *out) {
    u32x4 s = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    if (s[0] + s[1] + s[2] + s[3] == 0x12345u) *out = 1;
}
__global__ void k_copy(const u32x4 *a, u32x4 *b, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
// rows of `row_bytes`; a wave instruction writes SEG-byte segments of 1024/SEG consecutive rows; a workgroup of 8 waves owns a
// 256-row x 512-byte tile (like the linear1 epilogue: 8 lanes x 16 B = 128 B of 8 token rows per instruction)
template <int SEG>
__global__ void k_write_tiles(char *p, int rows, int row_bytes, unsigned v) {
    constexpr int LPS = SEG / 16, RPI = 64 / LPS;  // lanes per segment, rows per instruction
    const int tiles_f = row_bytes / 512, tiles = (rows / 256) * tiles_f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int r0 = (t / tiles_f) * 256 + (wave & 3) * 64, c0 = (t % tiles_f) * 512 + (wave >> 2) * 256;
        for (int cs = 0; cs < 256; cs += SEG)
            for (int rr = 0; rr < 64; rr += RPI) {
                char *q = p + (size_t)(r0 + rr + lane / LPS) * row_bytes + c0 + cs + (lane % LPS) * 16;
                *reinterpret_cast<u32x4 *>(q) = u32x4{v, v, v, v};
            }
    }
}
template <class F>
static double time_ms(F f, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}
int main() {
    const int rows = 245760, row_bytes = 5120;  // = one linear1 output (1.26 GB)
    const size_t bytes = (size_t)rows * row_bytes, n16 = bytes / 16;
    char *a, *b; unsigned *flag;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&flag, 4);
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    for (int grid : {1024, 4096, 16384}) {
        double w = time_ms([&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, (u32x4 *)a, n16, 7u); }, 10);
        double r = time_ms([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, (const u32x4 *)a, n16, flag); }, 10);
        double c = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, (const u32x4 *)a, (u32x4 *)b, n16); }, 10);
        printf("grid %5d: write %.0f GB/s  read %.0f GB/s  copy %.0f GB/s (r+w)\n", grid, bytes / w * 1e-6, bytes / r * 1e-6, 2.0 * bytes / c * 1e-6);
    }
    for (int grid : {256, 512, 2048}) {
        double t128 = time_ms([&] { hipLaunchKernelGGL(k_write_tiles<128>, dim3(grid), dim3(512), 0, 0, a, rows, row_bytes, 9u); }, 10);
        double t256 = time_ms([&] { hipLaunchKernelGGL(k_write_tiles<256>, dim3(grid), dim3(512), 0, 0, a, rows, row_bytes, 9u); }, 10);
        printf("tile-pattern grid %4d: 128-B segments %.0f GB/s, 256-B segments %.0f GB/s\n", grid, bytes / t128 * 1e-6, bytes / t256 * 1e-6);
    }
    double ms = time_ms([&] { hipMemsetAsync(a, 0, bytes, 0); }, 10);
    printf("hipMemsetAsync %.0f GB/s\n", bytes / ms * 1e-6);
    return 0;
}
