// Groundwork for an f32-accurate projection on HALF the matrix work (EXPERIMENTS A33): an f32 operand as TWO fp16 pieces
// (11 + 11 significant bits) and three v_mfma_f32_32x32x16_f16 products a0b0 + a0b1 + a1b0, against the six bf16 products of
// gemm_split_ws_kernel.  fp16 has 5 exponent bits, so the operands must be scaled into range by powers of two (per row of A,
// per column of B; undone in the epilogue).  This program answers, on the device:
//   1. does the matrix pipe honour fp16 SUBNORMAL inputs or flush them (decides where the scaled row maximum must sit)?
//   2. the error of the scaled fp16 x 2 product against fp64 on operands with a wide dynamic range, next to bf16 x 3 (six
//      products) and a plain f32 FMA loop;
//   3. the issue rate and clock of three f16 MFMAs against six bf16 MFMAs per tile pair on random operands.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_f16_split.hip -o tools/micro/mfma_f16_split
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ---- 1. subnormal inputs -------------------------------------------------------------------------------------------------------
__global__ void k_denorm(float* out, float aval, float bval) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)aval; b[i] = (_Float16)bval; }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}

// ---- 2. one 32 x 32 output tile, K a multiple of 16: C = A B with A [32, K] row-major, B [K, 32] row-major ---------------------
// lane l of the wave holds, for the k-block kb: A[l % 32][kb*16 + (l / 32)*8 + 0..7] and B[kb*16 + (l / 32)*8 + 0..7][l % 32];
// accumulator element i of lane l is C[8 (i / 4) + 4 (l / 32) + i % 4][l % 32]
__device__ __forceinline__ int c_row(int i, int l) { return 8 * (i / 4) + 4 * (l / 32) + (i % 4); }

// mode 0: scaled fp16 x 2, three products; mode 1: bf16 x 3, six products; mode 2: plain f32 FMA loop (one thread per element)
// sa[32]: power-of-two scale of every A row, sb[32]: of every B column (mode 0; the stored pieces are a * sa, b * sb)
__global__ void k_tile(const float* A, const float* B, int K, const float* sa, const float* sb, int mode, float* C) {
    const int l = threadIdx.x;
    if (mode == 2) {
        for (int e = l; e < 1024; e += 64) {
            const int r = e / 32, c = e % 32;
            float s = 0.f;
            for (int k = 0; k < K; ++k) s = fmaf(A[r * K + k], B[k * 32 + c], s);
            C[e] = s;
        }
        return;
    }
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int r = l % 32, kh = (l / 32) * 8;
    for (int kb = 0; kb < K / 16; ++kb) {
        float av[8], bv[8];
        for (int q = 0; q < 8; ++q) {
            av[q] = A[r * K + kb * 16 + kh + q];
            bv[q] = B[(kb * 16 + kh + q) * 32 + r];
        }
        if (mode == 0) {
            f16x8 a0, a1, b0, b1;
            for (int q = 0; q < 8; ++q) {
                const float as = av[q] * sa[r], bs = bv[q] * sb[r];
                a0[q] = (_Float16)as; a1[q] = (_Float16)(as - (float)a0[q]);
                b0[q] = (_Float16)bs; b1[q] = (_Float16)(bs - (float)b0[q]);
            }
            // (operand order as in the kernels: the first operand's rows become the tile's ROWS here -- A first)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc, 0, 0, 0);
        } else {
            bf16x8 a[3], b[3];
            for (int q = 0; q < 8; ++q) {
                float ra = av[q], rb = bv[q];
                for (int p = 0; p < 3; ++p) {
                    a[p][q] = (__bf16)ra; ra -= (float)a[p][q];
                    b[p][q] = (__bf16)rb; rb -= (float)b[p][q];
                }
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
        }
    }
    for (int i = 0; i < 16; ++i) {
        const int row = c_row(i, l), col = l % 32;
        C[row * 32 + col] = mode == 0 ? acc[i] / (sa[row] * sb[col]) : acc[i];
    }
}

// ---- 3. issue rate: NP products per tile pair and k-step on 8 accumulators, fragments resident ------------------------------
template <int F16>
__global__ void __launch_bounds__(256, 1) k_rate(float* out, const uint4* rnd, int iters, unsigned long long* cyc) {
    constexpr int NP = F16 ? 3 : 6;
    uint4 a[2][3], b[4][3];
    for (int i = 0; i < 2; ++i) for (int p = 0; p < 3; ++p) a[i][p] = rnd[(threadIdx.x * 18 + i * 3 + p) & 4095];
    for (int j = 0; j < 4; ++j) for (int p = 0; p < 3; ++p) b[j][p] = rnd[(threadIdx.x * 18 + 6 + j * 3 + p) & 4095];
    f32x16 acc[2][4];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f32x16 c = acc[i][j];
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    if (F16) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, b[j][p % 2]), __builtin_bit_cast(f16x8, a[i][(p + 1) % 2]), c, 0, 0, 0);
                    else c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b[j][p % 3]), __builtin_bit_cast(bf16x8, a[i][(p + 1) % 3]), c, 0, 0, 0);
                }
                acc[i][j] = c;
            }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int q = 0; q < 16; ++q) s += acc[i][j][q];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

static float pow2_scale(float maxabs, int target_exp) {       // 2^e with maxabs * 2^e in [2^target_exp, 2^(target_exp+1))
    if (maxabs == 0.f) return 1.f;
    int e;
    frexpf(maxabs, &e);                                       // maxabs = m 2^e, m in [0.5, 1)
    return ldexpf(1.f, target_exp + 1 - e);
}

int main() {
    float *d_out;
    CK(hipMalloc(&d_out, 1 << 20));
    // 1. subnormals: 2^-20 is an fp16 subnormal (smallest normal 2^-14); 16 products of 2^-20 * 2^10 = 2^-6 when honoured
    float h;
    k_denorm<<<1, 64>>>(d_out, ldexpf(1.f, -20), 1024.f);
    CK(hipMemcpy(&h, d_out, 4, hipMemcpyDeviceToHost));
    printf("fp16 subnormal INPUT 2^-20 x 2^10, K = 16: got %g (honoured: %g, flushed: 0)\n", h, 16 * ldexpf(1.f, -10));
    k_denorm<<<1, 64>>>(d_out, ldexpf(1.f, -14), 1024.f);
    CK(hipMemcpy(&h, d_out, 4, hipMemcpyDeviceToHost));
    printf("smallest NORMAL input 2^-14 x 2^10, K = 16: got %g (expected %g)\n", h, 16 * ldexpf(1.f, -4));

    // 2. accuracy
    const int K = 256;
    std::mt19937_64 g(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::uniform_real_distribution<float> ud(0.f, 1.f);
    float *dA, *dB, *dsa, *dsb, *dC;
    CK(hipMalloc(&dA, 32 * K * 4)); CK(hipMalloc(&dB, 32 * K * 4)); CK(hipMalloc(&dsa, 128)); CK(hipMalloc(&dsb, 128)); CK(hipMalloc(&dC, 4096));
    const char* names[3] = {"N(0,1) operands", "log-uniform magnitudes over 12 decades within every row", "rows of very different scale (1e-20 .. 1e+20)"};
    for (int dist = 0; dist < 3; ++dist) {
        std::vector<float> A(32 * K), B(32 * K), sa(32), sb(32), C(1024);
        for (int r = 0; r < 32; ++r) {
            const float rs = dist == 2 ? powf(10.f, -20.f + 40.f * ud(g)) : 1.f;
            for (int k = 0; k < K; ++k) {
                float v = nd(g);
                if (dist == 1) v *= powf(10.f, -6.f + 12.f * ud(g));
                A[r * K + k] = v * rs;
            }
        }
        for (int k = 0; k < K; ++k)
            for (int c = 0; c < 32; ++c) {
                float v = nd(g) / 16.f;
                if (dist == 1) v *= powf(10.f, -6.f + 12.f * ud(g));
                B[k * 32 + c] = v;
            }
        std::vector<double> ref(1024), scale(32, 0.0);
        for (int r = 0; r < 32; ++r)
            for (int c = 0; c < 32; ++c) {
                double s = 0, sabs = 0;
                for (int k = 0; k < K; ++k) { s += (double)A[r * K + k] * B[k * 32 + c]; sabs += fabs((double)A[r * K + k] * B[k * 32 + c]); }
                ref[r * 32 + c] = s;
                scale[r] = fmax(scale[r], fabs(s));
            }
        for (int target = 14; target >= 0; target -= 7) {          // where the scaled row / column maximum sits: 2^14, 2^7, 2^0
            for (int r = 0; r < 32; ++r) {
                float m = 0.f;
                for (int k = 0; k < K; ++k) m = fmaxf(m, fabsf(A[r * K + k]));
                sa[r] = pow2_scale(m, target);
            }
            for (int c = 0; c < 32; ++c) {
                float m = 0.f;
                for (int k = 0; k < K; ++k) m = fmaxf(m, fabsf(B[k * 32 + c]));
                sb[c] = pow2_scale(m, target);
            }
            CK(hipMemcpy(dA, A.data(), 32 * K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 32 * K * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(dsa, sa.data(), 128, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb.data(), 128, hipMemcpyHostToDevice));
            double err[3];
            for (int mode = 0; mode < 3; ++mode) {
                k_tile<<<1, 64>>>(dA, dB, K, dsa, dsb, mode, dC);
                CK(hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost));
                double e = 0;
                for (int i = 0; i < 1024; ++i) e = fmax(e, fabs((double)C[i] - ref[i]) / scale[i / 32]);   // relative to the ROW's largest |C|
                err[mode] = e;
            }
            printf("%-60s  max 2^%-2d: fp16x2 (3 products) %.2e | bf16x3 (6 products) %.2e | f32 FMA loop %.2e\n", names[dist], target, err[0], err[1], err[2]);
        }
    }

    // 3. rate
    std::vector<uint32_t> rnd(4096 * 4);
    for (auto& w : rnd) {                                     // random bf16 / fp16 pairs with exponents near 1 (no inf / nan)
        const uint32_t lo = 0x3c00u | (uint32_t)(g() & 0x3ff) | (uint32_t)((g() & 1) << 15), hi = 0x3c00u | (uint32_t)(g() & 0x3ff) | (uint32_t)((g() & 1) << 15);
        w = lo | (hi << 16);
    }
    uint4* d_rnd; unsigned long long* d_cyc;
    CK(hipMalloc(&d_rnd, rnd.size() * 4)); CK(hipMalloc(&d_cyc, 8));
    CK(hipMemcpy(d_rnd, rnd.data(), rnd.size() * 4, hipMemcpyHostToDevice));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, iters = 20000;
    for (int f16 = 0; f16 < 2; ++f16) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (f16) k_rate<1><<<cus, 256>>>(d_out, d_rnd, iters, d_cyc); else k_rate<0><<<cus, 256>>>(d_out, d_rnd, iters, d_cyc);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long cyc; CK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
        const int np = f16 ? 3 : 6;
        const double mfmas = (double)iters * 8 * np;
        const double flops = mfmas * 2.0 * 32 * 32 * 16 * 4 /*waves*/ * cus;
        printf("%s: %d products per tile pair: %.1f cycles per MFMA, %.3f ms for %d k-steps, %.2f PF/s issued, clock %.2f GHz, time per k-step tile set %.1f ns\n",
               f16 ? "fp16 x 2" : "bf16 x 3", np, (double)cyc / mfmas, ms, iters, flops / ms / 1e12, (double)cyc / (ms * 1e6), ms * 1e6 / iters);
    }
    return 0;
}
