// The split kernel's consumer step (ws_consume_step of gemm_f32.hip: 18 fragment reads + 48 MFMAs per k-step, software
// pipelined) on an LDS-resident image, without global traffic and without hand-over stalls (counters preset).  Variants:
// consumers alone; with four producer-like waves that only do the producer's LDS stores (split arithmetic + 12 stores per
// step); with producers that store without the split arithmetic.  Cycles per k-step against the 48 x 32.4 = 1,555 of the
// bare MFMA loop (tools/micro/mfma_bf16_rate.hip) show what the LDS traffic costs the matrix pipe.
#include "../../npi_gnn_amd/csrc/gemm_f32.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace npi;
namespace npi { void set_error(const char*, ...) {} }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); exit(1); } } while (0)

template <int PROD>     // 0: none, 1: split + stores, 2: stores only, 3: split only (no stores)
__global__ void __launch_bounds__(512, 1) probe(float* out, const float* rnd, int iters, unsigned long long* cyc) {
    constexpr int TM = 2, TN = 4, BN = 256, APL = 128 * 32, BPL = BN * 32, BUF = 3 * APL + 3 * BPL, NST = 4;
    __shared__ __attribute__((aligned(16))) char lds[NST * BUF];
    __shared__ int full[NST], empty[NST];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    for (int i = t; i < NST * BUF / 4; i += 512) reinterpret_cast<uint32_t*>(lds)[i] = __float_as_uint(rnd[i & 8191]) & 0xffff7fffu;
    if (t < NST) { full[t] = 1 << 30; empty[t] = 0; }
    __syncthreads();
    if (wave >= 4) {
        if (PROD == 0) return;
        const int pt = t - 256;
        const int ar = pt >> 2, ac = pt & 3;
        char* la0 = lds + simg(ar, ac >> 1) + (ac & 1) * 8;
        char* la1 = lds + simg(ar + 64, ac >> 1) + (ac & 1) * 8;
        char* lb0 = lds + 3 * APL + simg(pt, 0);
        char* lb1 = lds + 3 * APL + simg(pt, 1);
        f32x4r a0 = {rnd[pt], rnd[pt + 1], rnd[pt + 2], rnd[pt + 3]}, a1 = {rnd[pt + 4], rnd[pt + 5], rnd[pt + 6], rnd[pt + 7]};
        u32x4r b = {(uint32_t)pt, 2u, 3u, 4u};
        float sink = 0.f;
        for (int it = 0; it < iters; ++it) {
            const int off = (it & 3) * BUF;
            a0 = a0 * 1.0001f; a1 = a1 * 0.9999f;
            if (PROD == 1) { split3_store(a0, la0 + off, APL); split3_store(a1, la1 + off, APL); }
            if (PROD == 3) { uint32_t p0, p1, p2; split3_pair(a0.x, a0.y, p0, p1, p2); sink += __uint_as_float(p0 ^ p1 ^ p2);
                             split3_pair(a0.z, a0.w, p0, p1, p2); sink += __uint_as_float(p0 ^ p1 ^ p2);
                             split3_pair(a1.x, a1.y, p0, p1, p2); sink += __uint_as_float(p0 ^ p1 ^ p2);
                             split3_pair(a1.z, a1.w, p0, p1, p2); sink += __uint_as_float(p0 ^ p1 ^ p2); }
            if (PROD == 2) {
                for (int p = 0; p < 3; ++p) {
                    *reinterpret_cast<uint2*>(la0 + off + p * APL) = make_uint2(b.x, b.y);
                    *reinterpret_cast<uint2*>(la1 + off + p * APL) = make_uint2(b.z, b.w);
                }
            }
            if (PROD != 3) {
                for (int p = 0; p < 3; ++p) {
                    *reinterpret_cast<u32x4r*>(lb0 + off + p * BPL) = b;
                    *reinterpret_cast<u32x4r*>(lb1 + off + p * BPL) = b;
                }
            }
            signal(&empty[it & 3]);          // the producer's per-step hand-over cost (lgkmcnt(0) + one LDS atomic)
            // pace: one step per consumer step, roughly (the real producer waits for the stage)
            while (__hip_atomic_load(&empty[(it + 1) & 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 8 * ((it + 1) >> 2)) __builtin_amdgcn_s_sleep(1);
        }
        out[blockIdx.x * 512 + t] = a0.x + a1.y + sink;
        return;
    }
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    int offa[TM], offb[TN];
    for (int i = 0; i < TM; ++i) offa[i] = simg(wm * 64 + i * 32 + li, lh);
    for (int j = 0; j < TN; ++j) offb[j] = 3 * APL + simg(wn * (32 * TN) + j * 32 + li, lh);
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    frag_t af[TM][3], bf[TN][3];
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    ws_consume_first<TM, TN, APL, BPL>(lds_base, offa, offb, af, bf);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int g = 0; g < iters; ++g) ws_consume_step<TM, TN, APL, BPL, BUF, NST>(lds_base, full, empty, g, true, offa, offb, af, bf, acc);
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int q = 0; q < 16; ++q) s += acc[i][j][q];
    out[blockIdx.x * 512 + t] = s;
    if (t == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int PROD> void run(const char* name, int iters, float* out, float* rnd, unsigned long long* cyc) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    probe<PROD><<<256, 512>>>(out, rnd, iters, cyc); CK(hipDeviceSynchronize());
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0)); probe<PROD><<<256, 512>>>(out, rnd, iters, cyc); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-52s %.3f ms  %.0f cycles per k-step (48 MFMAs; clock %.2f GHz)\n", name, best, (double)c / iters, c / (best * 1e6));
}
int main() {
    float *out, *rnd; unsigned long long* cyc;
    CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&rnd, 8192 * 4 + 64)); CK(hipMalloc(&cyc, 8));
    std::vector<float> h(8192 + 16); srand(1); for (auto& v : h) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    CK(hipMemcpy(rnd, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    run<0>("consumers alone", 2000, out, rnd, cyc);
    run<1>("+ producer waves: split + LDS stores", 2000, out, rnd, cyc);
    run<2>("+ producer waves: LDS stores only", 2000, out, rnd, cyc);
    run<3>("+ producer waves: split only", 2000, out, rnd, cyc);
    return 0;
}
