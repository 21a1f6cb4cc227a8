// Store-pattern microbenchmark (round 2): how fast can workgroups write 256x256 fp32 tiles of a row-major [M][N] matrix, as a GEMM
// epilogue does, depending on the shape of one wave-instruction?  hipcc --offload-arch=gfx950 -O3 store_pattern_bench.hip -o spb && ./spb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// PATTERN 0: one wave-instruction = 4 rows x 256 B (wave owns a 64-column stripe of 128 rows: the LDS-transposed epilogue)
// PATTERN 1: one wave-instruction = 1 row x 1 KB (wave owns whole tile rows)
// PATTERN 2: one wave-instruction = 32 rows x 2 x 16 B (C^T accumulator layout, untransposed)
template <int PATTERN, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void store_tiles(float* C, int tiles_n, long ldc, float v) {
    const int t = blockIdx.x, tm = t / tiles_n, tn = t % tiles_n;
    float* base = C + (long)tm * 256 * ldc + tn * 256;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const f32x4 val = {v, v + 1.f, v + 2.f, v + 3.f};
    if (PATTERN == 0) {
        const int wm = wave / 4, wn = wave % 4;           // 2 x 4 waves of 128 x 64 (8 waves), or 4 waves each doing two of them
        for (int rep = 0; rep < 8 / WAVES; ++rep) {
            const int w2 = wave + rep * WAVES, m2 = w2 / 4, n2 = w2 % 4;
            (void)wm; (void)wn;
            float* p = base + (long)(m2 * 128) * ldc + n2 * 64;
#pragma unroll 4
            for (int it = 0; it < 32; ++it) {
                const int row = it * 4 + (lane >> 4), col = (lane & 15) * 4;
                *reinterpret_cast<f32x4*>(p + (long)row * ldc + col) = val;
            }
        }
    } else if (PATTERN == 1) {
        for (int row = wave; row < 256; row += WAVES) *reinterpret_cast<f32x4*>(base + (long)row * ldc + lane * 4) = val;
    } else {
        for (int rep = 0; rep < 8 / WAVES; ++rep) {
            const int w2 = wave + rep * WAVES, m2 = w2 / 4, n2 = w2 % 4;
            float* p = base + (long)(m2 * 128) * ldc + n2 * 64;
            const int r = lane & 31, h = lane >> 5;
#pragma unroll 4
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 2; ++j)
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<f32x4*>(p + (long)(i * 32 + r) * ldc + j * 32 + 8 * q + 4 * h) = val;
        }
    }
}
template <int P, int W>
static void run(const char* name, float* C, int M, int N) {
    const int tiles_m = M / 256, tiles_n = N / 256, tiles = tiles_m * tiles_n;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((store_tiles<P, W>), dim3(tiles), dim3(W * 64), 0, 0, C, tiles_n, (long)N, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((store_tiles<P, W>), dim3(tiles), dim3(W * 64), 0, 0, C, tiles_n, (long)N, (float)i);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("%-44s M=%d N=%d tiles=%d  %7.3f ms  %6.2f TB/s\n", name, M, N, tiles, ms, (double)M * N * 4 / ms / 1e9);
}
int main() {
    const int M = 19200, N = 4096;
    float* C; hipMalloc(&C, (size_t)M * N * 4);
    for (int n : {1024, 3072, 4096}) {
        run<0, 8>("4 rows x 256 B per instruction, 8 waves", C, M, n);
        run<0, 4>("4 rows x 256 B per instruction, 4 waves", C, M, n);
        run<1, 8>("1 row x 1 KB per instruction, 8 waves", C, M, n);
        run<1, 4>("1 row x 1 KB per instruction, 4 waves", C, M, n);
        run<2, 8>("32 rows x 2 x 16 B per instruction, 8 waves", C, M, n);
    }
    return 0;
}
