// One conv LAYER CALL as one entry point (ABI 3 on): npi_conv_fwd / npi_conv_bwd issue, in order on one stream, exactly the launches
// the per-op entry points issue -- aggregation, weight preparation, projection; ReLU mask, weight gradient, dAgg GEMM, transposed
// aggregation -- and add no kernel of their own.  Why: the reference's real workload is SMALL (src/train_with_twoDataset.PY:46-57,
// batches of 200 enclosing subgraphs; configs 1-3: 5,085 / 1,992 nodes).  There a layer's launches are 5-25 us of GPU work each and
// the step is bounded by the HOST: eight trips from Python through ctypes per layer and direction cost more than the kernels
// (C2: 0.54 ms eager for 0.33 ms of GPU time).  PyG's own granularity is one call per layer (SAGEConv.forward, reference
// src/classes.py:62,66,70; its autograd backward from src/train_with_twoDataset.PY:54): this is that call.
// Large graphs (the C4 / C5 configurations) keep the per-op entry points: their backward runs on two HIP streams, arranged by the
// host (functional._SageConvFn), and launch cost is noise there.
#include "npi_common.h"

using namespace npi;

extern "C" int npi_conv_fwd(const int32_t* rowptr, const int32_t* col, const int32_t* item_row, int64_t item_edges,
                            const float* w_entry, int64_t N, int64_t nnz_max, const void* x, int64_t ldx, int64_t F, int mean,
                            void* agg, int64_t ldagg, float* carry, const void* W, int64_t ldw, const void* bias, void* out,
                            int64_t ldo, int64_t K, int64_t Nout, int relu, int dtype, int gemm_flags, int prepare_which,
                            void* ws, int64_t ws_bytes, void* stream) {
    NPI_REQUIRE(prepare_which == 0 || prepare_which == 1 || prepare_which == 3,
                "npi_conv_fwd: prepare_which must be 0 (the GEMM prepares its own copy of W), 1 (forward copy) or 3 (both copies)");
    NPI_REQUIRE(!(prepare_which != 0 && (gemm_flags & NPI_GEMM_A_ZERO_PADDED)),
                "npi_conv_fwd: a prepared workspace excludes NPI_GEMM_A_ZERO_PADDED (as in npi_linear_fwd_ex)");
    // a2-a4: gather + segmented mean / weighted sum into the first F columns of agg
    int rc = npi_segsum_ex(rowptr, col, item_row, item_edges, w_entry, N, nnz_max, x, ldx, nullptr, 0, agg, ldagg, F, dtype, mean,
                           nullptr, carry, nullptr, stream);
    if (rc != NPI_OK) return rc;
    int flags = gemm_flags;
    if (prepare_which != 0) {
        rc = npi_linear_prepare(W, ldw, K, Nout, prepare_which, dtype, ws, ws_bytes, stream);
        if (rc != NPI_OK) return rc;
        flags |= NPI_GEMM_WORKSPACE_PREPARED;
    }
    // a5: agg @ W + b (ReLU in the store epilogue on request)
    return npi_linear_fwd_ex(agg, ldagg, W, ldw, bias, nullptr, out, ldo, N, K, Nout, relu, dtype, flags, ws,
                             prepare_which == 3 ? ws_bytes / 2 : ws_bytes, nullptr, stream);
}

extern "C" int npi_conv_bwd(const void* dout, int64_t lddo, const float* out_relu, int64_t ldor, float* dz, int64_t lddz, int64_t N,
                            int64_t K, int64_t Nout, int dtype, int gemm_flags, const void* agg, int64_t ldagg, void* dW, int64_t lddw,
                            void* db, float* dw_ws, int64_t dw_ws_elems, const void* W, int64_t ldw, const float* rowscale,
                            void* dagg, int64_t lddagg, void* ws_bwd, int64_t ws_bwd_bytes, int ws_prepared,
                            const int32_t* t_rowptr, const int32_t* t_col, const int32_t* t_item_row, int64_t t_item_edges,
                            const float* t_w, int64_t t_nnz_max, void* dx, int64_t lddx, float* t_carry, void* stream) {
    const void* g = dout;
    int64_t ldg = lddo;
    int rc = NPI_OK;
    if (out_relu != nullptr) {                     // threshold_backward of the fused ReLU (f32): dz = dout where out > 0
        NPI_REQUIRE(dtype == NPI_F32 && dz != nullptr, "npi_conv_bwd: the ReLU mask is an f32 pass and needs the dz buffer");
        rc = npi_relu_backward(reinterpret_cast<const float*>(dout), lddo, out_relu, ldor, N, Nout, dz, lddz, stream);
        if (rc != NPI_OK) return rc;
        g = dz;
        ldg = lddz;
    }
    if (dW != nullptr) {                           // agg^T dOut and the column sums of dOut (db may be null)
        rc = npi_linear_bwd_weight_ex(agg, ldagg, g, ldg, dW, lddw, db, N, K, Nout, dw_ws, dw_ws_elems, dtype,
                                      gemm_flags & (NPI_GEMM_EXACT_F32 | NPI_GEMM_SPLIT_BF16 | NPI_GEMM_A_ZERO_PADDED), 0, nullptr, nullptr, stream);
        if (rc != NPI_OK) return rc;
    }
    if (dx != nullptr) {                           // dAgg = rowscale * (dOut W^T), then dX = A^T dAgg over the transposed CSR
        NPI_REQUIRE(dagg && t_rowptr && t_item_row && t_carry, "npi_conv_bwd: null pointer on the dX chain");
        int flags = gemm_flags & (NPI_GEMM_EXACT_F32 | NPI_GEMM_SPLIT_BF16);
        if (ws_prepared) flags |= NPI_GEMM_WORKSPACE_PREPARED;
        rc = npi_linear_bwd_data_ex(g, ldg, W, ldw, rowscale, dagg, lddagg, N, K, Nout, dtype, flags, ws_bwd, ws_bwd_bytes, nullptr, stream);
        if (rc != NPI_OK) return rc;
        rc = npi_segsum_ex(t_rowptr, t_col, t_item_row, t_item_edges, t_w, N, t_nnz_max, dagg, lddagg, nullptr, 0, dx, lddx, K, dtype,
                           0, nullptr, t_carry, nullptr, stream);
    }
    return rc;
}
