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

// Entries per work item ("item" = what one wavefront reduces) as a function of the CSR's capacity: small graphs
// (the reference's 200-subgraph batches, the bundled full graphs) get 64-entry items -- four times as many
// wavefronts, a quarter of the serial chain each -- large ones NPI_ITEM_EDGES.  Every entry point derives it
// from the same nnz_max, so item_row, carry and the kernels always agree.
// Host-side decision (every kernel receives the item size as an argument).  Round 3: the switch moved from 2^20 to 2^22
// entries -- the per-rank sides of a sharded graph (1.2-2.6 M entries at C4 with 8 ranks) were launched as ~5 k wavefronts of
// 256 entries, a fraction of one wave per SIMD slot: a rank's SAGE step 1.19 -> 1.15 ms, its GAT step 1.97 -> 1.80 ms with
// 64-entry items; 2^24 changes nothing at 5 M entries and costs 4 % at 10 M.  One process-wide value (csr_build.hip):
// npi_small_graph_entries(n) sets it (tests), NPI_SMALL_GRAPH_ENTRIES=<n> in the environment presets it.
constexpr int64_t NPI_SMALL_GRAPH_ENTRIES = (int64_t)1 << 22;
int64_t small_graph_entries();
inline int item_edges_for(int64_t nnz_max) { return nnz_max < small_graph_entries() ? 64 : NPI_ITEM_EDGES; }

}  // namespace npi

#define NPI_REQUIRE(cond, msg)                 \
    do {                                       \
        if (!(cond)) {                         \
            npi::set_error("%s", msg);         \
            return NPI_ERR_ARG;                \
        }                                      \
    } while (0)
