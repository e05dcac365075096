// Parameter block of the segmented-reduction kernel (segsum.hip), shared with gat.hip.
#pragma once
#include "npi_common.h"

namespace npi {

enum { W_NONE = 0, W_ARRAY = 1, W_GAT_DST = 2, W_GAT_SRC = 3, W_GAT_SRC_PRE = 4, W_GAT_SRC_FUSED = 5, W_GAT_DST_PRE = 6,
       // the fused backward for 2 / 4 / 8 heads (packed form only; H C <= 256, C a power of two >= 32): compile-time head counts
       W_GAT_SRC_FUSED_H2 = 7, W_GAT_SRC_FUSED_H4 = 8, W_GAT_SRC_FUSED_H8 = 9,
       // round 5, one head, F <= 256: the FORWARD aggregation that also computes every entry's score and the softmax statistics of
       // every row -- no separate statistics pass, no per-entry score array.  An item first computes the scores of its entries
       // (lane-parallel, parked in LDS); a row that lies inside the item is weighted against its exact maximum, exactly as with the
       // statistics pass in front; the parts of a row that is cut by item / workgroup boundaries carry their own (max, sum exp) and
       // are merged with the usual rescaling exp(m_part - m_row) where the cut rows are resolved (segsum.hip)
       W_GAT_DST_FUSED = 10 };
constexpr bool is_fused_mode(int m) { return m == W_GAT_SRC_FUSED || (m >= W_GAT_SRC_FUSED_H2 && m <= W_GAT_SRC_FUSED_H8); }
constexpr int fused_heads(int m) { return m == W_GAT_SRC_FUSED_H2 ? 2 : m == W_GAT_SRC_FUSED_H4 ? 4 : m == W_GAT_SRC_FUSED_H8 ? 8 : 1; }

struct SegParams {
    const int32_t* rowptr;
    const int32_t* col;
    const int32_t* item_row;
    int N, n_items;          // n_items: filled in by segsum_run from nnz_max and item
    int relu;                // != 0: max(., 0) after scale and bias (the F.relu behind a GATConv, in the row epilogue; NaN kept)
    int mean;                // != 0: divide every row sum by its entry count (scatter_mean); filled in by segsum_run
    int item;                // entries per item: the CSR's own item size (what item_row was cut with), set by the entry point
    int nt_out;              // != 0: finished rows leave with non-temporal stores (filled in by segsum_run: outputs of >= 64 MB,
                             // which no cache would hold until their next use -- they only displace gathered rows)
    const float* x;
    int64_t ldx;
    const float* x2;         // two-part table: entries with col >= split read row (col - split) of x2 (same ldx);
    int split;               // segsum_run fills in x2 = x, split = INT_MAX when x2 is null (one table)
    float* out;
    int64_t ldo;
    int F;
    float* carry;
    const float* w;          // W_ARRAY: one weight per entry; W_GAT_SRC_PRE: alpha per by-target entry;
                             // W_GAT_DST_PRE (one head): the leaky_relu score e_p of every entry (npi_gat_softmax_stats_ex)
    const int32_t* wmap;     // W_GAT_SRC_PRE: entry p takes w[wmap[p]] (by-source entry -> by-target position)
    const float* bias;       // [F] or null, added after scaling
    float* scale_out;        // [N] or null: the power-of-two row scale (pow2_scale_of) of every finished row -- what the projection
                             // GEMM behind the aggregation takes as a_scales (NPI_GEMM_SPLIT_F16X2); f32, F <= 256 only
    // GAT: H heads of C channels (F == H * C), per-node per-head scalars [N, H]
    int H, C;
    const float* a_dst;
    const float* a_src;
    const float* m;          // row max of the scores
    const float* s;          // row sum of exp(score - max)
    float slope;             // leaky_relu negative slope
    // W_GAT_SRC epilogue: out[j, h, c] += g_dst[j,h] * att[h, c] + g_src[j,h] * att[h, C + c]
    const float* g_dst;
    const float* g_src;
    const float* att;        // [H, 2C]
    // W_GAT_DST, one head: alpha of every entry is also written here (by-target entry order) when not null
    float* alpha_out;
    // W_GAT_SRC_FUSED* (F <= 256): the by-source aggregation plus, in the same pass over the gathered dOut rows, the score
    // gradient dz[q, h] = alpha_q (<dOut_i, h_j> - D_i) leaky_relu'(a_dst[i] + a_src[j]) of every by-source entry q and head
    const float* hrow;       // [N, F] features of the ROW nodes (h_j), leading dimension ldh
    int64_t ldh;
    const int32_t* rowidx;   // row of every entry (the by-source CSR's rowidx)
    float* dz_out;           // [nnz_max, H] by-source entry order
    // W_GAT_SRC_FUSED (one head), optional: the row sums of dz -- g_src -- from the lanes that hold dz anyway.  Rows inside an item are
    // written to rowsum_out [N]; the parts of a row cut by an item boundary go to rs_head / rs_tail / rs_tail_row [n_items] exactly as
    // segscan.hip's seg_items_kernel leaves them, and seg_chain_sum (segscan.hip) adds them up behind the launch
    float* rowsum_out; float* rs_head; float* rs_tail; int32_t* rs_tail_row;
    // (a_dst, m, 1 / (s + 1e-16), D) of every TARGET node AND HEAD as one float4 ([n_cols, H, 4]): one 16-byte gather per entry
    // and head, alpha recomputed by the lane that owns the entry
    const float4* tpack;
    // W_GAT_DST_FUSED: the statistics it computes, [N] each (one head): row max of the scores, row sum of exp(score - max);
    // `rowidx` above is then the by-TARGET CSR's row of every entry, a_dst / a_src the per-node scores, slope the leaky_relu's
    float* m_out;
    float* s_out;
};

// x / out / bias are stored as `dtype` (NPI_F32 or NPI_BF16; the struct's float* are reinterpreted)
int segsum_run(SegParams P, int wmode, int mean, int64_t nnz_max, int dtype, hipStream_t stream);
// out[r] = tail[i] + head[i + 1] + ... for every row r = tail_row[i] cut by an item boundary (segscan.hip; one head)
int seg_chain_sum(const int32_t* rowptr, float* head, float* tail, int32_t* tail_row, float* out, int64_t N, int64_t n_items, int item,
                  hipStream_t stream);

}  // namespace npi
