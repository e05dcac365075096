// Bare v_mfma_f32_32x32x2_f32 issue-rate probe: W waves per SIMD, NACC independent accumulators.
// mode 0: (nearly) constant operands; mode 1: 16 random operand pairs per lane, cycled -- the clock the
// chip holds under MFMA load depends on operand toggling (MI355X_MICROARCH.md, DVFS give-back).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); exit(1); } } while (0)
template <int NACC, int RANDOM>
__global__ void __launch_bounds__(256) k(float* out, const float* rnd, int iters) {
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int q = 0; q < 16; ++q) acc[n][q] = 0.f;
    float a[16], b[16];
    for (int i = 0; i < 16; ++i) {
        a[i] = RANDOM ? rnd[(threadIdx.x * 16 + i) * 2] : 1.f + threadIdx.x * 1e-3f;
        b[i] = RANDOM ? rnd[(threadIdx.x * 16 + i) * 2 + 1] : 2.f - threadIdx.x * 1e-3f;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], b[r], acc[n], 0, 0, 0);
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) for (int q = 0; q < 16; ++q) s += acc[n][q];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC, int RANDOM> void run(int blocks_per_cu, int iters, float* out, float* rnd) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int grid = 256 * blocks_per_cu;
    k<NACC, RANDOM><<<grid, 256>>>(out, rnd, iters);
    CK(hipDeviceSynchronize());
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0)); k<NACC, RANDOM><<<grid, 256>>>(out, rnd, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    double flop = (double)grid * 4 /*waves*/ * iters * 16.0 * NACC * 4096.0;
    printf("NACC=%d waves/SIMD=%d random=%d iters=%d: %.3f ms  %.1f TF/s\n", NACC, blocks_per_cu, RANDOM, iters, best, flop / best / 1e9);
}
int main() {
    float *out, *rnd; CK(hipMalloc(&out, 256 * 8 * 256 * 4)); CK(hipMalloc(&rnd, 256 * 32 * 4));
    std::vector<float> h(256 * 32); srand(1); for (auto& v : h) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    CK(hipMemcpy(rnd, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    run<4, 0>(1, 5000, out, rnd); run<4, 1>(1, 5000, out, rnd); run<4, 1>(2, 2500, out, rnd);
    run<4, 1>(1, 50000, out, rnd);   // ~1 s sustained
    run<4, 0>(1, 50000, out, rnd);
    return 0;
}
