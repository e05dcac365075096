// Plain C++ host (no Python, no torch) driving the C ABI of include/npi_gnn.h: one SAGEConv forward
//   out = mean_{j in N(i) U {i}} x_j  @ W + b            (PyG 1.4.2 SAGEConv, reference src/classes.py:62)
// on a small random graph, checked against a CPU loop.  Shows what a non-Python caller binds:
//   npi_csr_workspace_bytes / npi_csr_build_ex  ->  npi_segsum_carry_elems / npi_segsum_ex  ->  npi_linear_workspace_bytes /
//   npi_linear_fwd_ex  (ABI 3 on: every scratch buffer is the caller's; the library allocates nothing and keeps no state)
// build:  hipcc --offload-arch=gfx950 -I include examples/c_abi_demo.cpp -L npi_gnn_amd -lnpi_gnn -Wl,-rpath,$PWD/npi_gnn_amd -o /tmp/c_abi_demo
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "npi_gnn.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define NPI_CALL(x) do { int rc_ = (x); if (rc_ != NPI_OK) { std::fprintf(stderr, "%s -> %d: %s\n", #x, rc_, npi_last_error()); return 3; } } while (0)

template <typename T>
static T* to_device(const std::vector<T>& h) {
    T* d = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d), h.size() * sizeof(T) + 16) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}
template <typename T>
static T* device_alloc(size_t n) {
    T* d = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d), (n ? n : 1) * sizeof(T)) != hipSuccess) return nullptr;
    return d;
}

int main() {
    const int64_t N = 3000, E = 40000, Fin = 178, Fout = 128;
    std::srand(7);
    auto frand = [] { return (float)std::rand() / RAND_MAX * 2.f - 1.f; };
    std::vector<int64_t> src(E), dst(E);
    for (int64_t e = 0; e < E; ++e) { src[e] = std::rand() % N; dst[e] = (e < E / 4) ? 5 : std::rand() % N; }   // row 5 is a hub
    std::vector<float> x(N * Fin), W(Fin * Fout), b(Fout);
    for (auto& v : x) v = frand();
    for (auto& v : W) v = frand() / std::sqrt((float)Fin);
    for (auto& v : b) v = frand();

    // ---- device side through the C ABI ----
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    int64_t* d_src = to_device(src);  int64_t* d_dst = to_device(dst);
    float* d_x = to_device(x);  float* d_W = to_device(W);  float* d_b = to_device(b);
    const int64_t nnz_max = E + N;                                   // edges + one self loop per node
    const int64_t item_edges = npi_item_edges(nnz_max);              // the recommended item size; this CSR keeps it from here on
    const int64_t n_items = npi_num_items(nnz_max, item_edges);
    int32_t* rowptr = device_alloc<int32_t>(N + 1);
    int32_t* col = device_alloc<int32_t>(nnz_max);
    int32_t* eid = device_alloc<int32_t>(nnz_max);
    int32_t* rowidx = device_alloc<int32_t>(nnz_max);
    int32_t* item_row = device_alloc<int32_t>(n_items + 1);
    int32_t* status = device_alloc<int32_t>(1);
    const int64_t ws_bytes = npi_csr_workspace_bytes(E, N);
    void* ws = device_alloc<char>((size_t)ws_bytes);
    if (!d_src || !d_dst || !d_x || !d_W || !d_b || !rowptr || !col || !eid || !rowidx || !item_row || !status || !ws) return 2;
    // key = destination, value = source: rows are the targets, as scatter_mean(x_j, edge_index[1]) groups them
    NPI_CALL(npi_csr_build_ex(d_dst, d_src, E, N, /*n_cols=*/N, /*add_self_loops=*/1, /*loop_col_offset=*/0, NPI_CSR_DROP_EQUAL, rowptr, col, eid,
                              rowidx, item_row, item_edges, status, ws, ws_bytes, stream));
    float* agg = device_alloc<float>(N * Fin);
    float* carry = device_alloc<float>((size_t)npi_segsum_carry_elems(nnz_max, item_edges, Fin));
    float* out = device_alloc<float>(N * Fout);
    const int64_t gemm_ws_bytes = npi_linear_workspace_bytes(Fin, Fout);          // the re-laid weight matrix lives in caller memory
    void* gemm_ws = device_alloc<char>((size_t)gemm_ws_bytes);
    if (!agg || !carry || !out || !gemm_ws) return 2;
    // the scratch holds arrival counters: zero it ONCE after allocating it (every launch leaves them at zero again)
    HIP_OK(hipMemsetAsync(carry, 0, sizeof(float) * (size_t)npi_segsum_carry_elems(nnz_max, item_edges, Fin), stream));
    NPI_CALL(npi_segsum_ex(rowptr, col, item_row, item_edges, /*w=*/nullptr, N, nnz_max, d_x, Fin, /*x2=*/nullptr, /*split=*/0, agg, Fin, Fin, NPI_F32,
                           /*mean=*/1, /*bias=*/nullptr, carry, /*row_scales_out=*/nullptr, stream));
    NPI_CALL(npi_linear_fwd_ex(agg, Fin, d_W, Fout, d_b, /*rowscale=*/nullptr, out, Fout, N, Fin, Fout, /*relu=*/0, NPI_F32, /*flags=*/0, gemm_ws,
                               gemm_ws_bytes, /*a_scales=*/nullptr, stream));
    std::vector<float> got(N * Fout);
    HIP_OK(hipMemcpyAsync(got.data(), out, got.size() * sizeof(float), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));

    // ---- CPU loop: add_remaining_self_loops, mean over in-neighbours and self, then @ W + b ----
    std::vector<double> sum(N * Fin, 0.0);
    std::vector<int> cnt(N, 1);
    for (int64_t i = 0; i < N; ++i)
        for (int64_t f = 0; f < Fin; ++f) sum[i * Fin + f] = x[i * Fin + f];
    for (int64_t e = 0; e < E; ++e) {
        if (src[e] == dst[e]) continue;                              // existing self loops are replaced by exactly one
        ++cnt[dst[e]];
        for (int64_t f = 0; f < Fin; ++f) sum[dst[e] * Fin + f] += x[src[e] * Fin + f];
    }
    double max_err = 0.0, max_ref = 0.0;
    for (int64_t i = 0; i < N; ++i)
        for (int64_t o = 0; o < Fout; ++o) {
            double r = b[o];
            for (int64_t f = 0; f < Fin; ++f) r += sum[i * Fin + f] / cnt[i] * W[f * Fout + o];
            max_err = std::fmax(max_err, std::fabs(r - got[i * Fout + o]));
            max_ref = std::fmax(max_ref, std::fabs(r));
        }
    std::printf("c_abi_demo: N=%lld E=%lld %lld->%lld  abi %d  max |err| = %.3e (max |ref| = %.3f)\n", (long long)N, (long long)E,
                (long long)Fin, (long long)Fout, npi_abi_version(), max_err, max_ref);
    return max_err <= 1e-4 * std::fmax(1.0, max_ref) ? 0 : 1;
}
