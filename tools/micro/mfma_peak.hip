// Bare v_mfma_f32_32x32x2_f32 issue-rate probe: W waves per SIMD, NACC independent accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int q = 0; q < 16; ++q) acc[n][q] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) for (int q = 0; q < 16; ++q) s += acc[n][q];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(int blocks_per_cu, int iters) {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = 256 * blocks_per_cu;
    k<NACC><<<grid, 256>>>(out, iters, 1.f, 2.f);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0); k<NACC><<<grid, 256>>>(out, iters, 1.f, 2.f); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    double flop = (double)grid * 4 /*waves*/ * iters * 4 * NACC * 4096.0;
    printf("NACC=%d waves/SIMD=%d iters=%d: %.3f ms  %.1f TF/s\n", NACC, blocks_per_cu, iters, best, flop / best / 1e9);
    hipFree(out);
}
int main() {
    run<4>(1, 20000); run<4>(2, 10000); run<4>(4, 5000); run<1>(1, 20000); run<2>(2, 10000);
    run<4>(1, 200000);   // ~1 s: sustained clock
    return 0;
}
