// Shared helpers for the gfx950 kernels behind include/npi_gnn.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../include/npi_gnn.h"

namespace npi {

constexpr int WAVE = 64;

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return NPI_ERR_LAUNCH;
    }
    return NPI_OK;
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int64_t align_up(int64_t a, int64_t b) { return ceil_div(a, b) * b; }

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// value of `v` in lane `l` (l wave-uniform) as a scalar
__device__ __forceinline__ int bcast_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float bcast_f(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ int uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
// a wave-uniform pointer, forced into an SGPR pair
template <typename T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
    const uint64_t u = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return reinterpret_cast<T*>(((uint64_t)hi << 32) | lo);
}

// fp16 x 2 projection (NPI_GEMM_SPLIT_F16X2, gemm_f32.hip): power-of-two scale for a row / column whose largest magnitude is m,
// m * scale in [2^14, 2^15) (1 for an all-zero row and for Inf / NaN -- which then propagate as in the bf16 split --; clamped so
// that scale and 1 / scale are normal f32).  Written by npi_row_scales, by the weight preparation, and by the aggregation
// kernels for the rows they finish (npi_segsum_ex).
__device__ __forceinline__ float pow2_scale_of(float m) {
    const uint32_t eb = (__float_as_uint(m) >> 23) & 0xff;
    if (m == 0.f || eb == 255) return 1.f;
    int es = 268 - (int)eb;                                   // biased exponent of 2^(14 - (eb - 127))
    es = es > 253 ? 253 : (es < 1 ? 1 : es);
    return __uint_as_float((uint32_t)es << 23);
}
__device__ __forceinline__ float pow2_inverse(float s) { return __uint_as_float(0x7f000000u - __float_as_uint(s)); }

// Entries per work item ("item" = what one wavefront reduces).  The item size is a PROPERTY OF THE CSR: npi_csr_build /
// npi_csr_filter take it as an argument and cut item_row with it, and every consumer of item_row (npi_segsum*, npi_gat_*)
// receives the same value from the caller, next to item_row -- no launch re-derives it, and the library keeps NO process-wide
// state (ABI 3).  npi_item_edges(nnz_max) is a pure HINT for a caller that builds a CSR: 64-entry items for capacities below
// 2^22 entries -- the reference's 200-subgraph batches, the bundled full graphs and the per-rank sides of a sharded graph, four
// times as many wavefronts with a quarter of the serial chain each -- and NPI_ITEM_EDGES above; a caller that wants the other
// size passes it.  Kernels that keep no item state between calls (segscan.hip, gat_edge_grad) pick their own chunking per call
// by the same rule.
constexpr int64_t NPI_SMALL_GRAPH_ENTRIES = (int64_t)1 << 22;
inline int item_edges_for(int64_t nnz_max) { return nnz_max < NPI_SMALL_GRAPH_ENTRIES ? 64 : NPI_ITEM_EDGES; }
inline bool item_edges_ok(int64_t item) { return item == 64 || item == NPI_ITEM_EDGES; }
inline int64_t num_items_of(int64_t nnz_max, int64_t item) { return nnz_max <= 0 ? 0 : ceil_div(nnz_max, item); }

}  // namespace npi

#define NPI_REQUIRE(cond, msg)                 \
    do {                                       \
        if (!(cond)) {                         \
            npi::set_error("%s", msg);         \
            return NPI_ERR_ARG;                \
        }                                      \
    } while (0)
