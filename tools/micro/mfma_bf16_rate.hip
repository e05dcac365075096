// v_mfma_f32_32x32x16_bf16 issue-rate probe in the split kernel's consumer pattern: 8 accumulators (2 x 4 tile pairs),
// six dependent MFMAs per accumulator and k-step, fragments resident in registers (no LDS, no memory).  Variants:
// chained (six back to back on one accumulator) or interleaved over the accumulators; one or two waves per SIMD; random
// or constant operands (the clock the chip holds under MFMA load depends on operand toggling); with a co-resident wave per
// SIMD that only does packed VALU work (what the split producer does).
// Prints cycles per MFMA (s_memtime in the kernel) and the achieved TF/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); exit(1); } } while (0)

template <int CHAIN, int VALU_WAVES>
__global__ void __launch_bounds__(512, 1) k(float* out, const uint4* rnd, int iters, unsigned long long* cyc) {
    const int wave = threadIdx.x >> 6;
    if (wave >= 4) {
        if (!VALU_WAVES) return;
        // VALU-only companion: packed f32 arithmetic, about as dense as the producer's split
        f32x2v v[8];
        for (int i = 0; i < 8; ++i) v[i] = f32x2v{1.f + threadIdx.x * 1e-3f + i, 2.f - i};
        for (int it = 0; it < iters * 12; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = v[i] * f32x2v{1.0001f, 0.9999f} + f32x2v{1e-3f, -1e-3f};
        }
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
        out[blockIdx.x * 512 + threadIdx.x] = s;
        return;
    }
    bf16x8 a[2][3], b[4][3];
    for (int i = 0; i < 2; ++i) for (int p = 0; p < 3; ++p) a[i][p] = __builtin_bit_cast(bf16x8, rnd[(threadIdx.x * 18 + i * 3 + p) & 4095]);
    for (int j = 0; j < 4; ++j) for (int p = 0; p < 3; ++p) b[j][p] = __builtin_bit_cast(bf16x8, rnd[(threadIdx.x * 18 + 6 + j * 3 + p) & 4095]);
    f32x16 acc[2][4];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (CHAIN) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    f32x16 c = acc[i][j];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j][0], a[i][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j][2], a[i][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j][1], a[i][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j][0], a[i][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j][1], a[i][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j][0], a[i][0], c, 0, 0, 0);
                    acc[i][j] = c;
                    __builtin_amdgcn_sched_barrier(0);
                }
        } else {
#define P(BP, AP) _Pragma("unroll") for (int j = 0; j < 4; ++j) _Pragma("unroll") for (int i = 0; i < 2; ++i) \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j][BP], a[i][AP], acc[i][j], 0, 0, 0); __builtin_amdgcn_sched_barrier(0)
            P(0, 2); P(2, 0); P(1, 1); P(0, 1); P(1, 0); P(0, 0);
#undef P
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int q = 0; q < 16; ++q) s += acc[i][j][q];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int CHAIN, int VALU_WAVES> void run(const char* name, int iters, float* out, uint4* rnd, unsigned long long* cyc) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<CHAIN, VALU_WAVES><<<256, 512>>>(out, rnd, iters, cyc); CK(hipDeviceSynchronize());
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0)); k<CHAIN, VALU_WAVES><<<256, 512>>>(out, rnd, iters, cyc); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    const double nm = (double)iters * 48;
    printf("%-44s %.3f ms  %.1f cycles per MFMA (clock %.2f GHz)  %.0f TF/s bf16\n", name, best, c / nm, c / (best * 1e6),
           256.0 * 4 * nm * 32768.0 / best / 1e9);
}

typedef float f32x4v __attribute__((ext_vector_type(4)));
// the same 64 x 128 output tile per wave on v_mfma_f32_16x16x32_bf16: 4 x 8 tiles, six products each, k-step 32
__global__ void __launch_bounds__(512, 1) k16(float* out, const uint4* rnd, int iters, unsigned long long* cyc) {
    const int wave = threadIdx.x >> 6;
    if (wave >= 4) return;
    bf16x8 a[4][3], b[2][3];                       // A: four row blocks resident; B: streamed per column block (two in flight)
    for (int i = 0; i < 4; ++i) for (int p = 0; p < 3; ++p) a[i][p] = __builtin_bit_cast(bf16x8, rnd[(threadIdx.x * 18 + i * 3 + p) & 4095]);
    for (int j = 0; j < 2; ++j) for (int p = 0; p < 3; ++p) b[j][p] = __builtin_bit_cast(bf16x8, rnd[(threadIdx.x * 18 + 12 + j * 3 + p) & 4095]);
    f32x4v acc[4][8];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) for (int q = 0; q < 4; ++q) acc[i][j][q] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4v c = acc[i][j];
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j & 1][0], a[i][2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j & 1][2], a[i][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j & 1][1], a[i][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j & 1][0], a[i][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j & 1][1], a[i][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j & 1][0], a[i][0], c, 0, 0, 0);
                acc[i][j] = c;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) for (int q = 0; q < 4; ++q) s += acc[i][j][q];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
void run16(const char* name, int iters, float* out, uint4* rnd, unsigned long long* cyc) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k16<<<256, 512>>>(out, rnd, iters, cyc); CK(hipDeviceSynchronize());
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0)); k16<<<256, 512>>>(out, rnd, iters, cyc); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    const double nm = (double)iters * 192;
    printf("%-44s %.3f ms  %.1f cycles per MFMA (clock %.2f GHz)  %.0f TF/s bf16\n", name, best, c / nm, c / (best * 1e6),
           256.0 * 4 * nm * 16384.0 / best / 1e9);
}
int main(int argc, char** argv) {
    const int random = argc > 1 ? atoi(argv[1]) : 1;
    float* out; uint4* rnd; unsigned long long* cyc;
    CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&rnd, 4096 * 16)); CK(hipMalloc(&cyc, 8));
    std::vector<uint16_t> h(4096 * 8); srand(1);
    for (auto& v : h) { float f = random ? (rand() / (float)RAND_MAX) * 2.f - 1.f : 1.f; uint32_t u; std::memcpy(&u, &f, 4); v = (uint16_t)(u >> 16); }
    CK(hipMemcpy(rnd, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    printf("operands: %s\n", random ? "random" : "constant");
    run<1, 0>("chained, 1 wave / SIMD", 2000, out, rnd, cyc);
    run<0, 0>("interleaved, 1 wave / SIMD", 2000, out, rnd, cyc);
    run<1, 1>("chained + a VALU-only wave per SIMD", 2000, out, rnd, cyc);
    run<0, 1>("interleaved + a VALU-only wave per SIMD", 2000, out, rnd, cyc);
    run<1, 0>("chained, 1 wave / SIMD, 10x longer", 20000, out, rnd, cyc);
    run16("16x16x32, same tile, 1 wave / SIMD", 1000, out, rnd, cyc);
    run16("16x16x32, same tile, 10x longer", 10000, out, rnd, cyc);
    run<1, 0>("32x32x16 chained again, 10x longer", 20000, out, rnd, cyc);
    return 0;
}
