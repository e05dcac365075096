// LDS-fed MFMA probe: the GEMM's mma_step (fragment reads + MFMAs) on an LDS-resident tile,
// no global traffic, optional barrier per step.  Isolates the inner loop of gemm_f32.hip.
#define NPI_PROBE 1
#include "../../npi_gnn_amd/csrc/gemm_f32.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace npi;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); exit(1); } } while (0)
template <int AMODE, int BMODE, int TM, int TN, int BAR>
__global__ void __launch_bounds__(256, (TM * TN > 4) ? 1 : 2) probe(float* out, const float* rnd, int iters) {
    constexpr int AF = Stage<AMODE == 0, 64 * TM>::FLOATS, BF = Stage<BMODE == 1, 64 * TN>::FLOATS;
    __shared__ __attribute__((aligned(16))) float lds[2][AF + BF];
    for (int i = threadIdx.x; i < 2 * (AF + BF); i += 256) (&lds[0][0])[i] = rnd[i % 8192];
    __syncthreads();
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    for (int it = 0; it < iters; ++it) {
        mma_step<AMODE, BMODE, TM, TN>(lds[it & 1], lds[it & 1] + AF, wave >> 1, wave & 1, lane & 31, lane >> 5, acc);
        if (BAR) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int q = 0; q < 16; ++q) s += acc[i][j][q];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int AMODE, int BMODE, int TM, int TN, int BAR> void run(const char* name, int bpc, float* out, float* rnd) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 400, grid = 256 * bpc;
    probe<AMODE, BMODE, TM, TN, BAR><<<grid, 256>>>(out, rnd, iters); CK(hipDeviceSynchronize());
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0)); probe<AMODE, BMODE, TM, TN, BAR><<<grid, 256>>>(out, rnd, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    double flop = (double)grid * iters * (64.0 * TM) * (64.0 * TN) * 32 * 2;
    printf("%-34s blocks/CU=%d barrier=%d : %.3f ms %.1f TF/s\n", name, bpc, BAR, best, flop / best / 1e9);
}
int main() {
    float *out, *rnd; CK(hipMalloc(&out, 256 * 8 * 256 * 4)); CK(hipMalloc(&rnd, 8192 * 4));
    std::vector<float> h(8192); srand(1); for (auto& v : h) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    CK(hipMemcpy(rnd, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    run<0, 0, 2, 2, 0>("NN 2x2", 1, out, rnd); run<0, 0, 2, 2, 0>("NN 2x2", 2, out, rnd); run<0, 0, 2, 2, 1>("NN 2x2", 2, out, rnd);
    run<0, 0, 2, 4, 0>("NN 2x4", 1, out, rnd); run<0, 0, 2, 4, 1>("NN 2x4", 1, out, rnd);
    run<0, 1, 2, 2, 0>("NT 2x2", 2, out, rnd); run<1, 0, 2, 2, 0>("TN 2x2", 2, out, rnd);
    return 0;
}
