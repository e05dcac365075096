// Store-pattern probe for the split GEMM's epilogue: a wave writes its 64 x 128 f32 piece of a C tile either as the accumulator
// layout gives it (every 16-byte store instruction = 32 rows x 32 bytes) or row-contiguous (every instruction = 2 rows x 512 bytes,
// full 128-byte lines).  256 workgroups of 4 waves, TILES tiles each, 128 x 256 f32 per tile.  Prints the time per tile and CU.
// Also with 64 / 32 / 8 workgroups: what ONE CU can store when the memory system is not the limit.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/store_pattern.hip -o tools/micro/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int PATTERN>
__global__ void __launch_bounds__(256, 1) k(float* C, int64_t ldc, int tiles, int tiles_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    float4 v = make_float4(lane, wave, 1.f, 2.f);
    for (int t = 0; t < tiles; ++t) {
        const int tile = (blockIdx.x + t * gridDim.x) % tiles_total;
        float* base = C + (int64_t)tile * 128 * ldc + (int64_t)wm * 64 * ldc + wn * 128;
        if (PATTERN == 0) {
            // accumulator layout: i = row block (2), jg = 16 column groups: lane -> row li, columns 32 j + 8 g + 4 lh
#pragma unroll
            for (int jg = 0; jg < 16; ++jg)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    *reinterpret_cast<float4*>(base + (int64_t)(i * 32 + li) * ldc + (jg >> 2) * 32 + 8 * (jg & 3) + 4 * lh) = v;
        } else if (PATTERN == 2) {
            // accumulator layout after a 4 x 4 transpose inside every quad of lanes (rows 4 a .. 4 a + 3 x their four 16-byte column
            // groups): instruction (g', i, j) writes row i 32 + 4 (li >> 2) + g', columns 32 j + 8 (li & 3) + 4 lh: 8 rows x 128 B
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        *reinterpret_cast<float4*>(base + (int64_t)(i * 32 + (li >> 2) * 4 + gq) * ldc + j * 32 + 8 * (li & 3) + 4 * lh) = v;
        } else if (PATTERN == 3) {
            // the same with the column blocks innermost on consecutive instructions: (g', i) x j -- a row's 512 B by 4 instructions in a row
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        *reinterpret_cast<float4*>(base + (int64_t)(i * 32 + (li >> 2) * 4 + gq) * ldc + j * 32 + 8 * (li & 3) + 4 * lh) = v;
        } else {
            // row-contiguous: instruction s writes rows 2 s, 2 s + 1: lane -> row 2 s + (lane >> 5), columns 4 (lane & 31)
#pragma unroll
            for (int s_ = 0; s_ < 32; ++s_)
                *reinterpret_cast<float4*>(base + (int64_t)(2 * s_ + (lane >> 5)) * ldc + 4 * (lane & 31)) = v;
        }
        v.x += 1.f;
    }
}

int main() {
    const int M = 1000000 / 128 * 128, N = 256, tiles_total = M / 128, tiles = 31;
    float* C;
    CK(hipMalloc(&C, (size_t)M * N * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int grid : {256, 64, 32, 8})
    for (int p = 0; p < 4; ++p)
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            if (p == 0) k<0><<<grid, 256>>>(C, N, tiles, tiles_total); else if (p == 1) k<1><<<grid, 256>>>(C, N, tiles, tiles_total);
            else if (p == 2) k<2><<<grid, 256>>>(C, N, tiles, tiles_total); else k<3><<<grid, 256>>>(C, N, tiles, tiles_total);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) printf("%3d workgroups, %s: %.3f ms for %d tiles per CU = %.2f us per tile and CU, %.2f TB/s chip-wide, %.1f B per cycle and CU at 2 GHz\n",
                                 grid, p == 0 ? "accumulator layout (32 rows x 32 B per instruction)" : p == 1 ? "row-contiguous (2 rows x 512 B per instruction)" :
                                 p == 2 ? "quad-transposed (8 rows x 128 B per instruction)" : "quad-transposed, a row's 4 instructions adjacent",
                                 ms, tiles, ms * 1e3 / tiles, (double)grid * tiles * 128 * 256 * 4 / (ms * 1e-3) / 1e12, 128.0 * 256 * 4 / (ms * 1e-3 / tiles * 2e9));
        }
    return 0;
}
