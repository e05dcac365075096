/*
 * npi_gnn.h -- C ABI of the MI355X (gfx950) message-passing engine for NPI-GNN's conv hot path.
 *
 * Drop-in boundary: the reference reaches this path through the PyG
 * `nn.Conv` module interface -- `SAGEConv(in,128)` constructed at reference src/classes.py:48,50,52
 * and called as `conv(x, edge_index)` at src/classes.py:62,66,70; backward through autograd at
 * src/train_with_twoDataset.PY:54.  The arithmetic lives in torch-geometric 1.4.2 / torch-scatter
 * (un-vendored, reference README.md:11).  The reference has no FFI of its own for this path; the
 * entry points below are what a ctypes binding inside a PyG-style `MessagePassing.propagate`
 * replacement binds.  Each entry point names the PyG-1.4.2 op it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer borrowed from the caller (PyTorch's caching allocator);
 *     the library never frees or retains the caller's memory and NO entry point allocates (ABI 3: the library
 *     imports no hipMalloc* / hipFree*): every scratch buffer is a caller workspace with a size query next to it;
 *   - the library keeps NO process-wide state and reads no environment variable (ABI 3: the setters npi_gemm_mode,
 *     npi_dw_shared, npi_small_graph_entries of ABI 2 and the entry points that consulted them are gone): the GEMM
 *     arithmetic, the dW grid regime and the item size of a CSR are arguments of the calls they concern; calls on
 *     different streams from different threads are independent.  npi_item_edges() is a pure function of its argument;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     every call is asynchronous on it and performs no host synchronisation;
 *   - return value: 0 = ok, <0 = error (text via npi_last_error(), thread-local); no C++
 *     exception crosses the ABI, nothing calls exit();
 *   - node ids in CSR arrays are int32 (N + E + 1 < 2^31); the COO input is int64 as PyG hands it
 *     over (`edge_index` LongTensor [2,E], row 0 = source j, row 1 = target i);
 *   - ABI 4 (npi_abi_version() == 4): ONE entry point per operation.  The `_ex2` forms of ABI 3 (their `_ex` / plain twins plus one
 *     optional pointer: row scales in or out) and the plain forms of npi_topk_select / npi_topk_gather / npi_filter_adj are folded
 *     into the base names, which now take that pointer / flag (NULL / 0 = the old behaviour); npi_entry_col_scale is new.
 */
#ifndef NPI_GNN_H
#define NPI_GNN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NPI_OK 0
#define NPI_ERR_ARG (-1)      /* bad argument (null pointer, negative size, unsupported width) */
#define NPI_ERR_LAUNCH (-2)   /* HIP launch / runtime error */
#define NPI_ERR_WORKSPACE (-3)/* workspace too small */

/* entries of the self-loop-augmented CSR that one wavefront ("item") reduces: a property of each CSR, 64 or NPI_ITEM_EDGES,
 * chosen by whoever builds it (npi_item_edges(nnz_max) = the recommended value) and passed as `item_edges` to the build and
 * to every entry point that takes `item_row` */
#ifndef NPI_ITEM_EDGES
#define NPI_ITEM_EDGES 256
#endif

/* data types of feature matrices */
#define NPI_F32 0
#define NPI_BF16 1

const char* npi_last_error(void);
int npi_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Graph build.  Replaces PyG `add_remaining_self_loops` (SAGEConv.forward / GCNConv.norm) and
 * the implicit "group messages by target" that torch_scatter does with atomics: a STABLE LSD
 * radix sort of the COO columns by `key` node, so that within one row the entries keep their
 * edge_index order and the appended self loop comes last -- exactly the accumulation order of
 * the reference's CPU scatter.
 *
 *   key_nodes[E], val_nodes[E] : int64 device arrays.  CSR-by-target: key = edge_index[1],
 *                                val = edge_index[0].  CSR-by-source (transpose, for backward):
 *                                swap them.
 *   Columns with key == val (existing self loops) and columns with an id outside [0,N) are
 *   dropped; the latter also set bit 0 of status[0] -- except the padding column (-1, -1) of
 *   npi_filter_adj, which is dropped silently.
 *   add_self_loops != 0 appends (i,i) as the LAST entry of every row.
 *
 *   rowptr[N+1]  : int32, rowptr[N] = nnz (device-side; nnz <= E + N)
 *   col[E+N]     : int32 neighbour (val) node per entry
 *   eid[E+N]     : int32 original column in edge_index, -1 for an appended self loop (may be NULL)
 *   rowidx[E+N]  : int32 key node per entry, i.e. the sorted COO row (may be NULL)
 *   item_row[npi_num_items(E+N, item_edges)+1] : int32 first row of every item of item_edges entries
 *   item_edges   : 64 or NPI_ITEM_EDGES -- the item size of THIS CSR; hand the same value to every call that takes item_row
 *   status[1]    : int32 device word, bit 0 = out-of-range id seen
 * ------------------------------------------------------------------------------------------ */
int64_t npi_csr_workspace_bytes(int64_t E, int64_t N);
/* recommended item size for a new CSR of capacity nnz_max: 64 below 2^22 entries, NPI_ITEM_EDGES from there on.  A pure
 * function and a hint only -- nothing that consumes a CSR calls it; a caller that wants the other size passes it. */
int64_t npi_item_edges(int64_t nnz_max);
int64_t npi_num_items(int64_t nnz_max, int64_t item_edges);   /* ceil(nnz_max / item_edges); -1 for an item size that does not exist */

/* Same build with separate row and column id spaces, for a destination-row SHARD of the graph on
 * one GPU: keys are local row ids in [0, N), values index a feature table of
 * n_cols rows (e.g. the all-gathered x of every rank); the appended self loop of row r gets
 * column r + loop_col_offset.  build_flags: NPI_CSR_DROP_EQUAL drops key == val columns (0: the caller has
 * already removed the global self loops); NPI_CSR_SORT_COLUMNS orders the entries of a row by column
 * (ties: list order) instead of list order -- a second key for the build's sort, for graphs whose rows are
 * long: the 4-byte gathers of per-node scalars (GATConv) then walk ascending addresses.  The SUMS are the
 * same in another association. */
#define NPI_CSR_DROP_EQUAL 1
#define NPI_CSR_SORT_COLUMNS 2
int npi_csr_build_ex(const int64_t* key_nodes, const int64_t* val_nodes, int64_t E, int64_t N,
                     int64_t n_cols, int add_self_loops, int64_t loop_col_offset, int build_flags,
                     int32_t* rowptr, int32_t* col, int32_t* eid, int32_t* rowidx,
                     int32_t* item_row, int64_t item_edges, int32_t* status,
                     void* workspace, int64_t workspace_bytes, void* stream);

/* inverse map for per-edge data that lives in one CSR's entry order and is needed in the other's:
 * pos_of[e] for e in [0,E) = entry index of edge column e in the CSR given by (eid, nnz_max), or -1 */
int npi_edge_positions(const int32_t* eid, const int32_t* rowptr, int64_t N, int64_t nnz_max,
                       int64_t E, int32_t* pos_of, void* stream);

/* ------------------------------------------------------------------------------------------
 * Segmented row reduction over a CSR -- replaces `torch.index_select(x, 0, edge_index[0])`
 * followed by `torch_scatter.scatter_{add,mean}(x_j, edge_index[1], dim_size=N)`
 * (PyG MessagePassing.propagate, reached from reference src/classes.py:62,66,70), without
 * materialising the [E+N, F] message tensor and without float atomics:
 *
 *     out[i, :] = scale_i * sum_{p in [rowptr[i], rowptr[i+1])}  w[p] * x[col[p], :]  (+ bias[:])
 *
 *   mean != 0  : scale_i = 1 / max(rowptr[i+1]-rowptr[i], 1)   (scatter_mean)
 *   w == NULL  : all ones (SAGE);  otherwise one f32 per CSR entry (GCN norm / GAT alpha)
 *   x, out     : [N, F] row-major, leading dimension ldx / ldo (elements), dtype f32 or bf16
 *                (bf16 storage, f32 accumulation)
 *   item_row, item_edges : as the CSR was built (npi_csr_build_ex / npi_csr_filter)
 *   carry      : f32 scratch, npi_segsum_carry_elems(nnz_max, item_edges, F) elements: the partial sums of rows cut by a
 *                workgroup boundary and the agent-scope arrival counters through which the LAST workgroup to deliver a
 *                partial of a row sums that row's chain inside the same launch (in a fixed order => bitwise reproducible;
 *                no second launch).  ZERO IT ONCE after allocating it: every launch resets the counters it used, so the
 *                buffer can be reused launch after launch (any width F' <= F, this CSR or another of no larger capacity);
 *                two launches that may run CONCURRENTLY need two buffers.
 * ------------------------------------------------------------------------------------------ */
int64_t npi_segsum_carry_elems(int64_t nnz_max, int64_t item_edges, int64_t F);
/* x2 != NULL: a TWO-PART feature table -- an entry with col[p] < split reads row col[p] of x, any other entry row
 * col[p] - split of x2 (same leading dimension and dtype).  One rank of the sharded layers (npi_gnn_amd/dist.py, SURVEY.md
 * 8(e)) gathers from [hub rows received from all ranks ; its own rows] without copying its own rows behind the received
 * ones.  x2 == NULL (split ignored): one table. */
/* row_scales_out != NULL: the launch also writes, for every finished row, the power-of-two scale the projection GEMM behind the
 * aggregation takes as `a_scales` (NPI_GEMM_SPLIT_F16X2; the same values npi_row_scales would compute from `out` in a pass of its
 * own): a wave maximum and one 4-byte store per row.  [N] or NULL.  Only f32 rows of 256 columns, 16-byte aligned, on a graph
 * with entries: npi_segsum_scales_supported(F, dtype).  (ABI 4: the separate scales-writing twin of ABI 3, folded.) */
int npi_segsum_scales_supported(int64_t F, int dtype);
int npi_segsum_ex(const int32_t* rowptr, const int32_t* col, const int32_t* item_row, int64_t item_edges,
                   const float* w, int64_t N, int64_t nnz_max, const void* x, int64_t ldx,
                   const void* x2, int64_t split, void* out, int64_t ldo, int64_t F, int dtype, int mean,
                   const float* bias, float* carry, float* row_scales_out, void* stream);

/* GCNConv.norm (PyG 1.4.2): deg[j] = sum of weights of entries in row j of the BY-SOURCE CSR
 * (deg == NULL: unweighted, the row lengths of deg_rowptr are used);
 * norm[p] = deg^-1/2[rowidx[p]] * w[p] * deg^-1/2[col[p]] for the entries of either CSR
 * (w == NULL: ones; deg^-1/2 of 0 is 0 as PyG masks inf). */
int npi_row_weight_sum(const int32_t* rowptr, const float* w, int64_t N, float* deg, void* stream);
int npi_gcn_norm(const int32_t* rowidx, const int32_t* col, const int32_t* rowptr, const float* w,
                 const float* deg, const int32_t* deg_rowptr, int64_t N, int64_t nnz_max, float* norm,
                 void* stream);
/* inv_cnt[i] = 1 / max(rowptr[i+1] - rowptr[i], 1): the scatter_mean divisor, needed again by the
 * backward (dX = A^T D^-1 dAgg) */
/* w_out[p] = table[col[p]] * w_in[p] (w_in == NULL: ones) for the entries p of one CSR, 0 behind its last entry: a per-NODE factor of
 * the gathered row as a per-entry weight.  The layers' backward uses it with table = npi_row_inv_count of the by-target CSR over the
 * by-source side: scatter_mean's divisor belongs to the TARGET, so dX = (A^T D^-1 dOut) W^T aggregates dOut itself with these
 * weights and projects afterwards (autograd of `scatter_mean` + `torch.matmul`, reference src/train_with_twoDataset.PY:54). */
int npi_entry_col_scale(const int32_t* col, const int32_t* rowptr, const float* table, const float* w_in, int64_t N, int64_t n_cols,
                        int64_t nnz_max, float* w_out, void* stream);
int npi_row_inv_count(const int32_t* rowptr, int64_t N, float* inv_cnt, void* stream);
/* per-entry weights from per-edge weights: w_entry[p] = eid[p] >= 0 ? edge_w[eid[p]] : loop_w
 * (loop weight of node i = loop_w_node[i] if given else fill) */
int npi_entry_weights(const int32_t* eid, const int32_t* rowidx, const int32_t* rowptr,
                      const float* edge_w, const float* loop_w_node, float fill,
                      int64_t N, int64_t nnz_max, float* w_entry, void* stream);
/* Backward of the ReLU that SAGEConv(..., relu=True) applies in its projection epilogue (autograd's threshold_backward):
 * dz[r, c] = y[r, c] > 0 ? dy[r, c] : 0 over M rows of F floats, y = the layer's (post-ReLU) output. */
/* Measurement support (no reference counterpart): `workgroups` workgroups that each hold 64 KB of a CU's LDS for `nanoseconds` --
 * a stand-in for the resident kernel of a collective, launched by npi_gnn_amd.virtual.StubCollectives beside a rank's step so
 * that a one-GPU run has the duration of an exchange and the CUs it sits on.  `start_word` (device memory, ZERO before the
 * launch; may be null): the time runs from the moment the first workgroup got a CU -- one that waited for a wave slot leaves
 * with the others; null: every workgroup times its own start.  Computes nothing; <= 256 workgroups, <= 100 ms. */
int npi_hold_cus(int workgroups, int64_t nanoseconds, uint64_t* start_word, void* stream);
int npi_relu_backward(const float* dy, int64_t ldd, const float* y, int64_t ldy, int64_t M, int64_t F, float* dz, int64_t ldz,
                      void* stream);

/* SAGEConv(normalize=True): y_i = x_i / max(||x_i||_2, eps) -- torch.nn.functional.normalize(out, p=2, dim=-1) at the end of
 * PyG 1.4.2 SAGEConv.update (the reference constructs its layers with the default normalize=False, src/classes.py:48-52; the
 * option is part of the layer's signature).  norm [M] receives ||x_i|| for the backward:
 * dx_i = (dy_i - y_i <dy_i, y_i>) / ||x_i||, and dy_i / eps where the norm was clamped.  One wavefront per row, fixed
 * reduction tree (bitwise reproducible); rows with pitches, 16-byte lanes when aligned. */
int npi_l2_normalize_rows(const float* x, int64_t ldx, int64_t M, int64_t F, float eps, float* y, int64_t ldy, float* norm,
                          void* stream);
int npi_l2_normalize_rows_bwd(const float* dy, int64_t ldd, const float* y, int64_t ldy, const float* norm, int64_t M, int64_t F,
                              float eps, float* dx, int64_t ldx, void* stream);

/* ------------------------------------------------------------------------------------------
 * Dense projection on the matrix cores -- replaces `torch.matmul(aggr_out, self.weight) + bias`
 * (SAGEConv.update / GCNConv.forward) and its autograd backward.  f32 in, f32 out, f32 accumulate on
 * the matrix cores, error at f32-rounding level in both arithmetics (the `flags` of the entry points below).
 *
 *   npi_linear_fwd_ex       : C[M,N]  = act( rowscale_m * (A[M,K] @ W[K,N]) + bias[N] )
 *   npi_linear_bwd_data_ex  : dA[M,K] = rowscale_m * (dC[M,N] @ W[K,N]^T)
 *   npi_linear_bwd_weight_ex: dW[K,N] = A[M,K]^T @ dC[M,N],  db[N] = colsum(dC)   (db may be NULL)
 *                             deterministic split over M; workspace f32
 * GEMM arithmetic of f32 storage: the default (flags 0) is the 3-way bf16 split of both operands on the bf16 matrix cores (six
 * v_mfma_f32_32x32x16_bf16 per product tile, f32 accumulate, error at f32-rounding level; csrc/gemm_f32.hip) -- used by
 * fwd / bwd_data on full 128 x 128 tiles when K % 32 == 0 and rows are 16-byte aligned, and by bwd_weight when K % 128 == 0,
 * N % 128 == 0, M >= 4096 (both operands split on the fly, gemm_dw_split_kernel); everything else (ragged strips, other
 * shapes) and every call with NPI_GEMM_EXACT_F32 runs the exact-f32 kernels (v_mfma_f32_32x32x2_f32).
 * ------------------------------------------------------------------------------------------ */

/* out[N] = column sums of X[M,N] (GCNConv / GATConv bias gradient).  workspace f32: npi_colsum_workspace_elems(M, N)
 * elements for the finest row chunking; ceil(M/2048)*N is the minimum that is accepted (coarser chunks, slower on
 * small M). */
int64_t npi_colsum_workspace_elems(int64_t M, int64_t N);
int npi_colsum(const float* X, int64_t ldx, int64_t M, int64_t N, float* out, float* workspace,
               int64_t workspace_elems, void* stream);
int64_t npi_linear_bwd_weight_workspace_elems(int64_t M, int64_t K, int64_t N);

/* The arithmetic and the grid regime are ARGUMENTS, the scratch for the re-laid weight matrix is a caller workspace -- no
 * process-wide switch exists, nothing is allocated.
 *   flags     : 0 = the default arithmetic (3-way bf16 split for f32 storage); NPI_GEMM_EXACT_F32 forces the exact-f32 kernels
 *               for this call (NPI_GEMM_SPLIT_BF16 names the default explicitly)
 *   workspace : npi_linear_workspace_bytes(K, N) bytes, 16-byte aligned, REQUIRED (NPI_ERR_WORKSPACE otherwise)
 *   shared    : npi_linear_bwd_weight_ex: 1 = the GEMM shares the CUs with an HBM-bound kernel on another stream
 *               (about 3 workgroups per 4 CUs), 0 = it has the GPU to itself (~4 workgroups per CU)
 *
 * NON-FINITE OPERANDS (what replaces torch.matmul at PyG 1.4.2 SAGEConv.update / its autograd):
 *   NPI_GEMM_EXACT_F32  is an fp32 fmaf chain: an Inf operand gives +-Inf in the products it takes part in (NaN against a
 *                       zero or an opposing Inf), exactly like torch.matmul.
 *   NPI_GEMM_SPLIT_BF16 (the default for f32) writes every operand as x0 + x1 + x2 with x1 = bf16(x - x0): for x = +-Inf,
 *                       and for finite |x| > 3.3895e38 (beyond the largest bf16, x0 rounds to Inf), x - x0 is NaN, so every
 *                       output element that operand takes part in is NaN -- the whole output ROW for an element of A / dC,
 *                       the whole output COLUMN for an element of W -- where fp32 matmul gives +-Inf (or, for a finite
 *                       |x| > 3.3895e38 against small weights, a finite number).  NaN operands give
 *                       NaN in both.  Finite results are unaffected (error <= 3 * 2^-24 |a||b| per product).  A model that
 *                       has diverged therefore reads NaN instead of Inf; callers that test `isinf` on activations must
 *                       test `!isfinite`, or pass NPI_GEMM_EXACT_F32.
 *                       Pinned by tests/test_gpu_parity.py::test_non_finite_operands_of_the_projection_gemms. */
#define NPI_GEMM_EXACT_F32 1
#define NPI_GEMM_SPLIT_BF16 2
/* npi_linear_fwd_ex / npi_linear_bwd_weight_ex, f32: A is stored with lda >= Kp (K rounded up to a multiple of 128) and its
 * columns K .. Kp-1 are ZERO (W and dW keep K rows): the matrix-core kernels then run on Kp instead of the guarded kernel on
 * an odd K (178 features in the reference's first layer).  Workspaces are queried with Kp. */
#define NPI_GEMM_A_ZERO_PADDED 4
/* npi_linear_fwd_ex / npi_linear_bwd_data_ex: `workspace` already holds the re-laid copy of THIS weight matrix that the call
 * would otherwise prepare in a launch of its own -- written by npi_linear_prepare (set 0 for fwd, set 1 for bwd_data) from the
 * same W / ldw / K / N / dtype, W unchanged since.  A layer prepares both copies in one launch and runs its forward GEMM,
 * its backward GEMM and any row-block split of them (dist.py: light rows / hub rows) without a preparation launch each.  Not
 * together with NPI_GEMM_A_ZERO_PADDED.  A call whose shape does not take the matrix-core kernels ignores the workspace. */
#define NPI_GEMM_WORKSPACE_PREPARED 8
/* npi_linear_fwd_ex / npi_linear_bwd_data_ex: the persistent matrix-core kernels (one workgroup per CU, a static walk over the
 * tiles) take n workgroups fewer than the chip has CUs (n a multiple of 8, at most 128).  For a GEMM launched while another
 * kernel holds CUs -- a collective's workgroups on a multi-GPU node: a workgroup that finds its CU taken starts late and
 * still owns its share of the tiles, so the launch ends when the other kernel does (measured with a stand-in that holds 16
 * CUs: 0.107 -> 0.15 ms for 125 k rows).  Costs n / 256 more tiles per workgroup when nothing else runs. */
#define NPI_GEMM_RESERVE_CUS(n) ((((n) / 8) & 0xff) << 8)
#define NPI_GEMM_RESERVED_CUS_OF(flags) ((((flags) >> 8) & 0xff) * 8)
/* npi_linear_fwd_ex / npi_linear_bwd_data_ex, f32: the matrix-core kernel runs on TWO fp16 pieces per operand (11 + 11
 * significant bits) and THREE v_mfma_f32_32x32x16_f16 per product tile instead of three bf16 pieces and six products -- half the
 * matrix work for the same f32-rounding-level result (tools/micro/mfma_f16_split.hip: 3.8e-7 of a row's largest |C| against
 * 3.9e-7 for the six bf16 products and 5.5e-7 for an f32 FMA loop; EXPERIMENTS A33 / A34).  fp16 has five exponent bits, so every
 * row of A (dC) is scaled by a power of two that puts its largest magnitude into [2^14, 2^15) -- `a_scales`, [M], written by
 * npi_row_scales -- and every column of the weight operand likewise (inside the preparation); the store epilogue undoes both
 * exactly.  Elements within 2^-28 of their row's maximum keep 22 significant bits, smaller ones an absolute error of 2^-39 of
 * that maximum (the matrix pipe honours fp16 subnormals).  Non-finite operands behave as under NPI_GEMM_SPLIT_BF16.  A workspace
 * handed over with NPI_GEMM_WORKSPACE_PREPARED must have been prepared with `which | NPI_PREPARE_F16X2`.  Shapes that do not take
 * the matrix-core kernel ignore the flag. */
#define NPI_GEMM_SPLIT_F16X2 16
#define NPI_PREPARE_F16X2 4
int64_t npi_linear_workspace_bytes(int64_t K, int64_t N);
/* scales[m] = the power of two s with max_k |A[m, k]| s in [2^14, 2^15) (1 for an all-zero row or one that holds Inf; clamped so
 * that s and 1 / s are normal f32) -- the `a_scales` of the NPI_GEMM_SPLIT_F16X2 calls.  One pass over A. */
int npi_row_scales(const float* A, int64_t lda, int64_t M, int64_t K, float* scales, void* stream);
/* The re-laid copies of W [K, N] (the `weight` of PyG's `torch.matmul(aggr_out, self.weight)`, reference call sites
 * src/classes.py:62,66,70) for the matrix-core kernels, in ONE launch: which = 1: the copy npi_linear_fwd_ex uses, 2: the one
 * npi_linear_bwd_data_ex uses, 3: both, the second npi_linear_workspace_bytes(K, N) bytes behind the first; | NPI_PREPARE_F16X2:
 * the fp16 x 2 planes (and column scales) of NPI_GEMM_SPLIT_F16X2 calls instead of the three bf16 planes.  K and N
 * multiples of 16; workspace 16-byte aligned, npi_linear_workspace_bytes(K, N) bytes per copy. */
int npi_linear_prepare(const void* W, int64_t ldw, int64_t K, int64_t N, int which, int dtype, void* workspace,
                       int64_t workspace_bytes, void* stream);
/* a_col_scales [K] / dc_col_scales [N] (16-byte aligned; NULL without the flag): NPI_GEMM_SPLIT_F16X2 for the weight gradient -- two fp16
 * pieces per operand, three matrix products per tile instead of six.  The contraction runs over the ROWS, so what factors out of
 * the sum is a power-of-two scale per COLUMN of A and of dC: npi_col_scales.  K, N multiples of 128, M >= 4096, f32. */
int npi_linear_bwd_weight_ex(const void* A, int64_t lda, const void* dC, int64_t lddc,
                             void* dW, int64_t lddw, void* db,
                             int64_t M, int64_t K, int64_t N,
                             float* workspace, int64_t workspace_elems, int dtype, int flags, int shared,
                             const float* a_col_scales, const float* dc_col_scales, void* stream);
/* scales[k] = the power of two that puts column k's largest magnitude into [2^14, 2^15).  A != NULL: from the column maxima of A
 * [M, K] (one pass over it: for a matrix that does not change between steps, once).  A == NULL: from `row_scales` [M] (npi_row_scales,
 * or the launch that wrote the matrix): the SMALLEST row scale -- the scale of the matrix's largest magnitude -- for every column
 * alike, without a pass over the matrix; elements more than 2^-18 below that magnitude then keep an absolute 2^-39 of it rather
 * than 22 relative bits.  workspace: npi_col_scales_workspace_elems(M, K) floats. */
int64_t npi_col_scales_workspace_elems(int64_t M, int64_t K);
int npi_col_scales(const float* A, int64_t lda, int64_t M, int64_t K, const float* row_scales, float* scales,
                   float* workspace, int64_t workspace_elems, void* stream);
/* a_scales / dc_scales: the row scales of the left operand for NPI_GEMM_SPLIT_F16X2 in `flags` (npi_row_scales, or the launch that
 * wrote the operand); NULL / ignored without the flag.  (ABI 4: the scale-taking twins of ABI 3, folded.) */
int npi_linear_fwd_ex(const void* A, int64_t lda, const void* W, int64_t ldw, const void* bias,
                       const float* rowscale, void* C, int64_t ldc,
                       int64_t M, int64_t K, int64_t N, int relu, int dtype, int flags,
                       void* workspace, int64_t workspace_bytes, const float* a_scales, void* stream);
int npi_linear_bwd_data_ex(const void* dC, int64_t lddc, const void* W, int64_t ldw,
                            const float* rowscale, void* dA, int64_t ldda,
                            int64_t M, int64_t K, int64_t N, int dtype, int flags,
                            void* workspace, int64_t workspace_bytes, const float* dc_scales, void* stream);

/* The same three GEMMs with A / W / C / bias / dW / db stored as `dtype` (NPI_F32 or NPI_BF16; bf16
 * storage, f32 MFMA accumulation, f32 rowscale and workspace) -- BASELINE.json configs[1]. */
/* C = A W (f32, no bias) and, from the accumulators on their way to C, the row dots sc0[m] = <C[m, :], att[:N]>,
 * sc1[m] = <C[m, :], att[N:]> (att: [2 N]; sc0 / sc1: [M]): GATConv's `x = torch.mm(x, self.weight)` together with the two
 * halves of `(torch.cat([x_i, x_j], dim=-1) * self.att).sum(dim=-1)` per NODE (PyG 1.4.2 gat_conv.py forward / message, one
 * head; not on the reference's own path: BASELINE configs[4]) -- the pass over h that npi_gat_scores makes is gone.  Only
 * shapes where one column tile of the matrix-core kernel covers N: npi_linear_fwd_scores_supported(M, K, N) (M >= 128,
 * K % 32 == 0, N = 128 or 256, default GEMM arithmetic); anything else returns NPI_ERR_ARG (the caller runs npi_linear_fwd_ex
 * + npi_gat_scores).  Operands 16-byte aligned, leading dimensions % 4 == 0; workspace as npi_linear_fwd_ex.  Fixed
 * summation order (bitwise reproducible). */
int npi_linear_fwd_scores_supported(int64_t M, int64_t K, int64_t N);
/* a_scales != NULL: the fp16 x 2 arithmetic (NPI_GEMM_SPLIT_F16X2) with the row scales of A -- npi_row_scales, ONCE for
 * a feature matrix that does not change between steps, or the launch that wrote A (npi_gat_aggregate_fused) */
int npi_linear_fwd_scores(const float* A, int64_t lda, const float* W, int64_t ldw, const float* att, float* C, int64_t ldc,
                              float* sc0, float* sc1, int64_t M, int64_t K, int64_t N, void* workspace, int64_t workspace_bytes,
                              const float* a_scales, void* stream);
/* dA = dC W^T + row0 (x) col0 + row1 (x) col1 (f32; row* are [M], col* [K] vectors): npi_linear_bwd_data_ex with a rank-2 term
 * added in the store epilogue of the matrix-core kernel -- no read-modify-write pass over dA or dC.  GATConv backward
 * (PyG 1.4.2 GATConv.message's `(x_i, x_j) * att` terms, reference call site src/classes.py:48-52 via BASELINE configs[4]): the
 * attention terms g_dst (x) W att_dst + g_src (x) W att_src of dX.  Only shapes the kernel covers completely:
 * npi_linear_bwd_data_rank2_supported(M, K, N) (M >= 128, K % 128 == 0, N % 32 == 0, default GEMM arithmetic); anything else
 * returns NPI_ERR_ARG (the caller adds the terms to dC with npi_gat_rank1_add instead).  Operands 16-byte aligned, leading
 * dimensions % 4 == 0; workspace as npi_linear_bwd_data_ex. */
int npi_linear_bwd_data_rank2_supported(int64_t M, int64_t K, int64_t N);
/* dc_scales != NULL: the fp16 x 2 arithmetic (NPI_GEMM_SPLIT_F16X2) with the row scales of dC, from npi_row_scales
 * or from the launch that wrote dC (npi_gat_backward_fused_heads) */
int npi_linear_bwd_data_rank2(const float* dC, int64_t lddc, const float* W, int64_t ldw, const float* row0,
                                  const float* row1, const float* col0, const float* col1, float* dA, int64_t ldda,
                                  int64_t M, int64_t K, int64_t N, void* workspace, int64_t workspace_bytes,
                                  const float* dc_scales, void* stream);

/* The two-row products around that epilogue, one head (att = [att_dst ; att_src], [2, C]; W [K, C]; P = x^T [g_dst g_src], [2, K] from
 * npi_gat_att_grad on x):  cols: U [2, K] = att W^T, the column vectors col0 / col1 above.  tail: dW [K, C] += P^T att (the
 * outer-product correction of the weight gradient; null: skipped) and datt [2, C] = P W (null: skipped).  Fixed summation
 * orders.  (GATConv.message's att terms, PyG 1.4.2 gat_conv.py; not on the reference's own path: BASELINE configs[4].) */
int npi_gat_rank2_cols(const float* W, int64_t ldw, const float* att, int64_t K, int64_t C, float* U, void* stream);
int npi_gat_rank2_tail(const float* P, const float* W, int64_t ldw, const float* att, int64_t K, int64_t C, float* dw, int64_t lddw,
                       float* datt, void* stream);

/* ------------------------------------------------------------------------------------------
 * One conv LAYER CALL as one entry point -- replaces `SAGEConv.forward` as the reference calls it (`self.convK(x, edge_index)`,
 * src/classes.py:62,66,70) and its autograd backward (src/train_with_twoDataset.PY:54); also GCNConv evaluated as (A_hat x) W + b.
 * They issue, in order on `stream`, exactly the launches of the per-op entry points above and add no kernel of their own; what
 * they save is the HOST's cost of eight separate calls per layer and direction, which bounds the step at the reference's real
 * sizes (batches of 200 enclosing subgraphs, the bundled full graphs).  Large graphs keep the per-op calls (their backward is
 * arranged on two streams by the caller).  Every buffer is the caller's, as everywhere.
 *
 *   npi_conv_fwd : agg[:, :F] = segsum(CSR by target, x)  (mean / per-entry weights w_entry as npi_segsum_ex; columns F .. of a
 *                  wider agg must be ZERO: NPI_GEMM_A_ZERO_PADDED in gemm_flags);  out = act(agg @ W[K, Nout] + bias).
 *                  prepare_which: 0 = the GEMM lays W out itself (ws: npi_linear_workspace_bytes(K rounded up to 128 if padded,
 *                  Nout)), 1 = npi_linear_prepare of the forward copy first, 3 = both copies (ws: two of them; the second one is
 *                  what npi_conv_bwd takes as ws_bwd with ws_prepared = 1).
 *   npi_conv_bwd : g = out_relu ? dout masked by out_relu > 0 (into dz; f32) : dout;
 *                  dW != NULL: dW[K, Nout] = agg^T g, db = colsum(g) (db may be NULL; dw_ws: npi_linear_bwd_weight_workspace_elems);
 *                  dx != NULL: dagg = rowscale * (g @ W^T), dx = segsum(transposed CSR t_*, dagg) (sum; weights t_w or NULL).
 * ------------------------------------------------------------------------------------------ */
int npi_conv_fwd(const int32_t* rowptr, const int32_t* col, const int32_t* item_row, int64_t item_edges,
                 const float* w_entry, int64_t N, int64_t nnz_max, const void* x, int64_t ldx, int64_t F, int mean,
                 void* agg, int64_t ldagg, float* carry, const void* W, int64_t ldw, const void* bias, void* out,
                 int64_t ldo, int64_t K, int64_t Nout, int relu, int dtype, int gemm_flags, int prepare_which,
                 void* ws, int64_t ws_bytes, void* stream);
int npi_conv_bwd(const void* dout, int64_t lddo, const float* out_relu, int64_t ldor, float* dz, int64_t lddz, int64_t N,
                 int64_t K, int64_t Nout, int dtype, int gemm_flags, const void* agg, int64_t ldagg, void* dW, int64_t lddw,
                 void* db, float* dw_ws, int64_t dw_ws_elems, const void* W, int64_t ldw, const float* rowscale,
                 void* dagg, int64_t lddagg, void* ws_bwd, int64_t ws_bwd_bytes, int ws_prepared,
                 const int32_t* t_rowptr, const int32_t* t_col, const int32_t* t_item_row, int64_t t_item_edges,
                 const float* t_w, int64_t t_nnz_max, void* dx, int64_t lddx, float* t_carry, void* stream);

/* ------------------------------------------------------------------------------------------
 * GATConv (PyG 1.4.2; absent from the reference tree, BASELINE.json configs[4]).  H heads of C
 * channels, hfeat = x @ W is [N, H*C]; att is [H, 2C] (first C multiply the TARGET's features).
 * The attention coefficient alpha of an entry is recomputed from four per-node, per-head scalars
 * [N, H] -- a_dst, a_src (npi_gat_scores), m, s (npi_gat_softmax_stats_ex):
 *     alpha(i <- j) = exp(leaky_relu(a_dst[i] + a_src[j]) - m[i]) / (s[i] + 1e-16)
 * The backward may keep the alpha that npi_gat_edge_grad computes anyway (alpha_out [nnz_max, H], by-target
 * entry order) and hand it to the by-source npi_gat_aggregate (alpha + alpha_map = npi_entry_transpose_map;
 * one head): reading a weight back is cheaper there than the exp and the divide per entry.  Both NULL: recompute.
 *
 *   npi_gat_scores        `(cat[x_i, x_j] * att).sum(-1)` split into its two dot products
 *   npi_gat_softmax_stats_ex `utils.softmax`: scatter_max + scatter_add(exp) per target row
 *   npi_gat_aggregate     by_source == 0: out[i] = sum_p alpha_p hfeat[col p] (+ bias)          (forward)
 *                         by_source != 0: out[j] = sum_q alpha_q x[col q] + g_dst[j] att[:C] + g_src[j] att[C:]
 *                                         over the by-source CSR                                (backward, d hfeat)
 *   npi_gat_rowdot        D[i,h] = <a[i,h,:], b[i,h,:] - bias[h,:]>      (= sum_p alpha_p dalpha_p)
 *   npi_gat_edge_grad     dz[p,h] = alpha_p (<dout_i, hfeat_j> - D_i) * leaky_relu'(z_p), by-target entry order
 *   npi_seg_rowsum_ex     out[r,h] = sum_{p in row r} vals[map ? map[p] : p, h]
 *   npi_entry_transpose_map  map[q] = by-target position of by-source entry q
 *   npi_gat_att_grad      datt[h,:C] = sum_i g_dst[i,h] hfeat[i,h,:],  datt[h,C:] likewise with g_src
 * ------------------------------------------------------------------------------------------ */
int npi_gat_scores(const float* hfeat, int64_t ldh, const float* att, int64_t N, int64_t H, int64_t C,
                   float* a_dst, float* a_src, void* stream);
/* npi_gat_aggregate over a two-part table (x2 / split as in npi_segsum_ex). */
int npi_gat_aggregate_ex(const int32_t* rowptr, const int32_t* col, const int32_t* item_row, int64_t item_edges,
                         int64_t N, int64_t nnz_max, const float* x, int64_t ldx, const float* x2, int64_t split,
                         float* out, int64_t ldo,
                         int64_t H, int64_t C, const float* a_dst, const float* a_src, const float* m,
                         const float* s, float negative_slope, int by_source, const float* bias,
                         const float* g_dst, const float* g_src, const float* att,
                         const float* alpha, const int32_t* alpha_map,
                         float* carry, void* stream);
/* Fused backward pass over the by-SOURCE CSR, H in {1, 2, 4, 8} heads (H > 1: C in {32, 64, 128}), H C <= 256 -- in ONE pass over
 * the gathered dOut rows
 *     out[j, :]  = sum_q alpha_q dout[col q, :]                                   (the aggregation half of d hfeat)
 *     dz[q, h]   = alpha_q (<dout[col q], hfeat[j]> - D[col q]) leaky_relu'(a_dst[col q] + a_src[j])   (the SDDMM)
 * replaces npi_gat_edge_grad + the by-source npi_gat_aggregate: one gather pass over the entries instead of two.  The four
 * per-TARGET scalars the pass needs -- a_dst, m, 1 / (s + 1e-16), D -- are packed into one float4 per node and head
 * (npi_gat_pack_targets over the N H flattened scalars; tpack [n_cols, H, 4], 16 MB at N = 1M: cache resident): every entry
 * makes ONE 16-byte gather per head and alpha is recomputed by the lane that owns the entry (one exp per entry and head).
 * a_src [n_rows, H], dz [nnz_max, H]; every entry's H dots are reduced inside the C / 4 lanes of their head.  dout2 / split: the
 * gathered rows come from a two-part table as in npi_segsum_ex (tpack is ONE array indexed by the column id over both parts:
 * the sharded GATConv's received hub rows + the rank's own rows).  The attention terms g_dst att[:C] + g_src att[C:] of d hfeat
 * are added afterwards by npi_gat_rank1_add, or never formed (npi_linear_bwd_data_rank2); g_dst / g_src are row sums of dz
 * (npi_seg_rowsum_ex over both orientations). */
int npi_gat_pack_targets(const float* a_dst, const float* m, const float* s, const float* D, int64_t N, float* tpack,
                         void* stream);
/* row_scales_out != NULL ([N]; heads * out_channels == 256): the launch also writes the power-of-two scale of every finished row of `out`:
 * the dc_scales of the projection behind it (npi_linear_bwd_data_rank2 / npi_linear_bwd_data_ex) without a pass over `out`.
 * g_src_out != NULL ([N]; one head): the launch also leaves the row sums of dz -- what npi_seg_rowsum_ex(vals = dz) over this CSR would
 * return -- taken by the lanes that compute dz; workspace: npi_seg_scan_workspace_elems(nnz_max, 1) floats (rows cut by an item
 * boundary are added up by a small launch behind the pass).  NULL: workspace is not used. */
int npi_gat_backward_fused_heads(const int32_t* rowptr, const int32_t* col, const int32_t* rowidx,
                                     const int32_t* item_row, int64_t item_edges, int64_t N, int64_t nnz_max,
                                     const float* dout, int64_t ldd, const float* dout2, int64_t split, const float* hfeat,
                                     int64_t ldh, float* out, int64_t ldo, int64_t H, int64_t C, const float* tpack,
                                     const float* a_src, float slope, float* dz, float* carry, float* row_scales_out,
                                     float* g_src_out, float* workspace, int64_t workspace_elems, void* stream);
int npi_gat_rank1_add(float* dh, int64_t ld, const float* g_dst, const float* g_src, const float* att,
                      int64_t N, int64_t H, int64_t C, void* stream);
int npi_gat_rowdot(const float* a, int64_t lda, const float* b, int64_t ldb, const float* bias,
                   int64_t N, int64_t H, int64_t C, float* D, void* stream);
/* npi_gat_edge_grad with a two-part gathered table (hfeat2 / split) and, swap != 0, the roles of rows and columns
 * exchanged -- the same dz seen from a by-SOURCE CSR: rows are source nodes (`dout` = their hfeat rows, a_src indexed
 * by row), columns are target nodes (`hfeat` = the gathered dOut rows; a_dst, m, s, D indexed by column).  The sharded
 * GAT backward uses it to sum dz by source without a transpose map between different ranks' entry sets. */
int npi_gat_edge_grad_ex(const int32_t* rowptr, const int32_t* col, const int32_t* rowidx,
                         int64_t N, int64_t nnz_max, const float* hfeat, int64_t ldh,
                         const float* hfeat2, int64_t split,
                         const float* dout, int64_t ldd, int64_t H, int64_t C,
                         const float* a_dst, const float* a_src, const float* m, const float* s,
                         const float* D, float negative_slope, int swap, float* dz, float* alpha_out, void* stream);
/* Per-row reductions of per-ENTRY scalars (npi_gnn_amd/csrc/segscan.hip): the entries are streamed -- a wavefront per 64 / 256
 * consecutive entries, a segmented scan keyed by the CSR's rowidx, cut rows folded by a second launch in a fixed order
 * (deterministic, no atomics).  These calls keep no item state: they take no item_row and chunk the stream per call.
 *   npi_seg_rowsum_ex        out[r, h] = sum over the entries p of row r of vals[map ? map[p] : p, h]; rows without an entry: 0
 *   npi_gat_softmax_stats_ex (m, s)[r, h] = (max, sum exp(. - max)) of e_p = leaky_relu(a_row[r] + a_col[col p]) over row r (an empty
 *                              row: m = 0, s = 0); additionally writes e_p of every entry to scores [nnz_max, H] (may be NULL),
 *                              which npi_gat_aggregate_scores reads back
 *   npi_gat_aggregate_scores = npi_gat_aggregate_ex(by_source = 0), one head, with the per-entry scores of the statistics
 *                              pass instead of two gathered per-node scalars per entry;
 *                              relu != 0: max(., 0) in the row epilogue (the F.relu behind the layer)
 *   npi_gat_rowdot_colsum    = npi_gat_rowdot AND the column sums of `a` (GATConv's bias gradient; colsum may be NULL) in
 *                              one pass over a and b; needs 16-byte aligned rows, C % 4 == 0, H C <= 1024
 * workspace: npi_seg_scan_workspace_elems(nnz_max, H) / npi_gat_rowdot_colsum_workspace_elems(N, H, C) floats. */
int64_t npi_seg_scan_workspace_elems(int64_t nnz_max, int64_t H);
int npi_seg_rowsum_ex(const int32_t* rowptr, const int32_t* rowidx, const float* vals, const int32_t* map,
                      int64_t N, int64_t nnz_max, int64_t H, float* out, float* workspace, int64_t workspace_elems,
                      void* stream);
int npi_gat_softmax_stats_ex(const int32_t* rowptr, const int32_t* col, const int32_t* rowidx, const float* a_row,
                             const float* a_col, int64_t N, int64_t nnz_max, int64_t H, float negative_slope, float* m,
                             float* s, float* scores, float* workspace, int64_t workspace_elems, void* stream);
int npi_gat_aggregate_scores(const int32_t* rowptr, const int32_t* col, const int32_t* item_row, int64_t item_edges,
                             int64_t N, int64_t nnz_max, const float* x, int64_t ldx, const float* x2, int64_t split,
                             float* out, int64_t ldo, int64_t C, const float* scores, const float* m, const float* s,
                             const float* bias, int relu, float* carry, void* stream);
/* The same forward aggregation WITHOUT the statistics pass in front of it (one head of at most 256 channels): an online softmax
 * inside the launch.  The score of an entry is e_p = leaky_relu(a_dst[row] + <x[col p], att[C:]>) -- the source's half recomputed
 * from the row that is gathered anyway (att: the layer's [2 C] attention vector, PyG layout: target half first), so no per-entry
 * score array and no 4-byte gather per entry exist; the open row keeps (max so far, sum of exp(e - max)) and rescales its
 * accumulators when the maximum moves; the parts of a row cut by an item boundary carry their (max, sum) and are merged with
 * exp(m_part - m_row) where cut rows are resolved, in a fixed order (bitwise reproducible).  Writes m[N], s[N] -- the row
 * maximum and the row sum of exp(e - m) that npi_gat_pack_targets / the backward need -- beside out.  (rowidx: unused, may be
 * NULL.)  The scores differ from npi_gat_scores' by the rounding of another summation order (1e-6 relative). */
/* row_scales_out != NULL ([N]; 256 channels): the launch also writes the power-of-two scale of every finished row of `out` -- bias and
 * ReLU applied, as stored --: the a_scales of the NEXT layer's projection (npi_linear_fwd_scores / npi_linear_fwd_ex) */
int npi_gat_aggregate_fused(const int32_t* rowptr, const int32_t* col, const int32_t* rowidx, const int32_t* item_row,
                                int64_t item_edges, int64_t N, int64_t nnz_max, const float* x, int64_t ldx, const float* x2,
                                int64_t split, float* out, int64_t ldo, int64_t C, const float* a_dst, const float* att,
                                float negative_slope, const float* bias, int relu, float* m, float* s, float* carry,
                                float* row_scales_out, void* stream);
int64_t npi_gat_rowdot_colsum_workspace_elems(int64_t N, int64_t H, int64_t C);
/* `F.relu(conv(x))` fused (npi_gat_aggregate_scores with relu != 0 applies the ReLU in the row epilogue): b is then the ReLU
 * OUTPUT, and this form first masks the incoming gradient, a' = a where b > 0 else 0 (threshold_backward), uses a' for D and
 * the column sums and writes it to a_masked [N, ldm] -- the gradient of the pre-activation the rest of the backward consumes.
 * D is unchanged by the fusion: where b > 0 the pre-activation equals b, elsewhere a' = 0. */
int npi_gat_rowdot_colsum_relu(const float* a, int64_t lda, const float* b, int64_t ldb, const float* bias,
                               int64_t N, int64_t H, int64_t C, float* D, float* colsum, float* a_masked, int64_t ldm,
                               float* workspace, int64_t workspace_elems, void* stream);
int npi_entry_transpose_map(const int32_t* src_eid, const int32_t* src_rowidx, const int32_t* src_rowptr,
                            const int32_t* dst_rowptr, const int32_t* pos_dst_of_edge, int64_t N,
                            int64_t nnz_max, int32_t* map, void* stream);
int64_t npi_gat_att_grad_workspace_elems(int64_t N, int64_t H, int64_t C);
int npi_gat_att_grad(const float* hfeat, int64_t ldh, const float* g_dst, const float* g_src,
                     int64_t N, int64_t H, int64_t C, float* datt, float* workspace,
                     int64_t workspace_elems, void* stream);

/* ------------------------------------------------------------------------------------------
 * Between-layer steps of Net_1 (reference src/classes.py:63-64,67-68,71-72),
 * forward only: PyG 1.4.2 TopKPooling(ratio) and the [global_max_pool || global_mean_pool] readout.
 * `batch` is the PyG Batch vector (int64, non-decreasing); graph_ptr[B+1] its segment starts.
 *   npi_topk_score      score_i = tanh(<x_i, w> / ||w||)
 *   npi_graph_bounds    graph_ptr from batch
 *   npi_topk_select     per graph keep ceil(ratio n_g) best (score desc, index asc): out_ptr[B+1],
 *                       perm[out_ptr[B]], remap[N] (new id or -1); status bit 1 = a graph > 16384 nodes
 *   npi_topk_gather     x' = x[perm] * score[perm], batch' = batch[perm], score' = score[perm]
 *   npi_filter_adj      surviving edges, relabelled, original order; count[0] on device
 *   npi_readout_max_mean  out[g] = [max_i x_i || mean_i x_i]  ([B, 2F])
 * ------------------------------------------------------------------------------------------ */
int npi_topk_score(const float* x, int64_t ldx, const float* w, int64_t N, int64_t F, float* score, void* stream);
int npi_graph_bounds(const int64_t* batch, int64_t N, int64_t B, int32_t* graph_ptr, void* stream);
/* max_nodes: an upper bound of the largest graph of the batch, when the caller knows one (0 = unknown) */
int npi_topk_select(const float* score, const int32_t* graph_ptr, int64_t N, int64_t B, float ratio,
                    int32_t* out_ptr, int32_t* perm, int32_t* remap, int32_t* status, int64_t max_nodes, void* stream);
/* The same selection for graphs of ANY size (npi_topk_select sorts a graph's scores in LDS: at most 16,384 nodes per graph,
 * bit 1 of its status word otherwise): two stable radix sorts of the whole batch on the device -- by score, descending, ties
 * by the lower node index; then by graph id -- and one pass that keeps the first ceil(ratio n_g) nodes of every graph.
 * Same outputs (out_ptr [B+1], perm in graph-major, score-descending order, remap), no device read; `batch` is the PyG batch
 * vector (int64, non-decreasing).  workspace: npi_topk_sorted_workspace_bytes(N) bytes. */
int64_t npi_topk_sorted_workspace_bytes(int64_t N);
int npi_topk_select_sorted(const float* score, const int64_t* batch, const int32_t* graph_ptr, int64_t N, int64_t B,
                           float ratio, int32_t* out_ptr, int32_t* perm, int32_t* remap, void* workspace,
                           int64_t workspace_bytes, void* stream);
/* perm64 (may be NULL) receives perm as int64 -- the LongTensor TopKPooling returns -- without a cast launch */
int npi_topk_gather(const float* x, int64_t ldx, const float* score, const int64_t* batch,
                    const int32_t* perm, const int32_t* out_ptr, int64_t B, int64_t F, int64_t n_out_max,
                    float* xo, int64_t ldo, int64_t* batch_o, float* score_o, int64_t* perm64, void* stream);
/* int32 workspace of npi_filter_adj: one count per tile of 2,048 edges, two spare words, then E words that receive the
 * new position of every input edge (-1: dropped) -- npi_filter_adj_newpos_offset(E) is where those start */
int64_t npi_filter_adj_workspace_elems(int64_t E);
/* pad_tail != 0: the output kept at the INPUT's length -- fills out_src / out_dst[count .. E) with -1.
 * A (-1, -1) column is padding everywhere downstream -- npi_csr_build_ex drops it without raising the out-of-range flag,
 * npi_filter_adj skips it -- so a caller that knows the node counts (TopKPooling keeps ceil(ratio n_g) per graph) never
 * has to read the surviving-edge count back and the whole Net_1 step runs without a host synchronisation (and captures
 * into a HIP graph).  Negative ids in src / dst are always treated as dropped.  Output arrays must not alias the input. */
int npi_filter_adj(const int64_t* src, const int64_t* dst, int64_t E, const int32_t* remap,
                      int64_t* out_src, int64_t* out_dst, int32_t* count, int32_t* workspace, int pad_tail, void* stream);
int64_t npi_filter_adj_newpos_offset(int64_t E);   /* npi_filter_adj: workspace[offset + e] = new position of input edge e, -1 if dropped */
/* The by-target CSR of the POOLED graph from the CSR of its parent, without a sort: row perm[r'] of the parent with the
 * entries whose source survived (remap[col] >= 0), in the parent's order, self loop last; eid through newpos
 * (npi_filter_adj's workspace tail).  Bit-identical to npi_csr_build_ex on the filtered edge list.  nnz_max_out = capacity
 * of col_o / eid_o / rowidx_o (>= the surviving entries; E_in + n_out is what a fresh build would use); item_edges: the item
 * size of the NEW CSR (independent of the parent's); workspace int32 [npi_csr_filter_workspace_elems(n_out)];
 * n_out <= npi_csr_filter_max_rows(). */
int64_t npi_csr_filter_max_rows(void);
int64_t npi_csr_filter_workspace_elems(int64_t n_out);
int npi_csr_filter(const int32_t* rowptr, const int32_t* col, const int32_t* eid, const int32_t* perm, const int32_t* remap,
                   const int32_t* newpos, int64_t n_out, int64_t nnz_max_out, int32_t* rowptr_o, int32_t* col_o,
                   int32_t* eid_o, int32_t* rowidx_o, int32_t* item_row_o, int64_t item_edges, int32_t* status_o, int32_t* workspace,
                   void* stream);
int npi_readout_max_mean(const float* x, int64_t ldx, const int32_t* graph_ptr, int64_t B, int64_t F,
                         float* out, void* stream);

/* Backward of the pooling layer (autograd of reference src/classes.py:63-72 in the train loop,
 * src/train_with_twoDataset.PY:52-54).  x is the layer INPUT, perm/score the forward's results.
 *   npi_topk_gather_bwd : per kept row i = perm[p]:  ds = <dxo[p], x[i]> (+ dscore_o[p], may be NULL),
 *                         dz = ds (1 - score_i^2),  dx[i] = dxo[p] score_i + dz w / ||w||;  dx must be
 *                         zero-filled by the caller (dropped rows get no gradient);  dzv[p] = dz,
 *                         dzz[p] = dz * (x_i . w / ||w||) feed the weight gradient.
 *   npi_topk_weight_grad: dw[f] = sum_p dzv[p] x[perm[p], f] / ||w|| - w[f] sum_p dzz[p] / ||w||^2
 *                         (deterministic two-level sum; workspace f32).
 *   npi_readout_max_mean_bwd: dx[i, c] = dout[g, F + c] / n_g + (i = first arg-max row of column c in graph g
 *                         ? dout[g, c] : 0); `out` is the forward's [B, 2F] result. */
int npi_topk_gather_bwd(const float* x, int64_t ldx, const float* score, const float* w, const int32_t* perm,
                        int64_t n_out, int64_t F, const float* dxo, int64_t lddxo, const float* dscore_o,
                        float* dx, int64_t lddx, float* dzv, float* dzz, void* stream);
/* The forms the training step uses: no fill launch in front of them.
 *   npi_topk_gather_bwd_ex     : one wave per INPUT row i of the N; remap[i] = its kept position p or -1 (npi_topk_select);
 *                                dropped rows are written as zeros, so dx may come in uninitialised.
 *   npi_readout_max_mean_bwd_ex: N = rows of x; rows outside [graph_ptr[0], graph_ptr[B]) are written as zeros. */
int npi_topk_gather_bwd_ex(const float* x, int64_t ldx, const float* score, const float* w, const int32_t* remap,
                           int64_t N, int64_t F, const float* dxo, int64_t lddxo, const float* dscore_o,
                           float* dx, int64_t lddx, float* dzv, float* dzz, void* stream);
int npi_readout_max_mean_bwd_ex(const float* x, int64_t ldx, const int32_t* graph_ptr, int64_t B, int64_t F,
                                const float* out, const float* dout, float* dx, int64_t lddx, int64_t N, void* stream);
int64_t npi_topk_weight_grad_workspace_elems(int64_t n_out, int64_t F);
int npi_topk_weight_grad(const float* x, int64_t ldx, const int32_t* perm, const float* dzv, const float* dzz,
                         int64_t n_out, int64_t F, const float* w, float* dw, float* workspace,
                         int64_t workspace_elems, void* stream);
int npi_readout_max_mean_bwd(const float* x, int64_t ldx, const int32_t* graph_ptr, int64_t B, int64_t F,
                             const float* out, const float* dout, float* dx, int64_t lddx, void* stream);

/* ---------------------------------------------------------------------------------------------
 * One-hop enclosing-subgraph extraction + PyG Batch collate.  Replaces the
 * per-sample Python loops of local_subgraph_generation (reference src/classes.py:652-733) and the
 * DataLoader collate in front of Net_1.  Interaction graph = CSR over node serial numbers:
 * ptr[N+1], nbr[nnz] partners in interaction_list order (npi_csr_build_ex keeps it), ok[nnz] = pair usable
 * (not in set_allInteractionKey_cannotUse, src/generate_dataset.py:296-299).  keys[B][2] = (rna, protein)
 * targets.  Local node order of a sample: rna, protein, usable partners of the rna, then of the protein;
 * pairs: target first, then in that same order, each in both directions ((rna, protein) first).
 *   npi_subgraph_sizes   : node_off[B+1], pair_off[B+1] (exclusive prefix sums; directed edges = 2 pairs);
 *                          workspace int32[2B]; node_off[B] = pair_off[B] = -1 when the batch exceeds 2^31 - 1 rows
 *   npi_subgraph_fill    : node_id[n], batch[n] (int64), edge_src/edge_dst[2 * pairs] (int64, batch-global ids).  n_nodes /
 *                          n_pairs: the totals the CALLER sized those arrays with; when they are not node_off[B] / pair_off[B]
 *                          (totals computed for other keys) bit 2 (value 4) of status[0] is raised -- an error the host reads at its
 *                          next device read instead of an out-of-bounds write (status may be NULL) -- and the arrays are filled,
 *                          within the caller's sizes, with values every consumer is safe on until then: node 0 / graph 0 for
 *                          every row, the (-1, -1) padding column for every edge
 *   npi_subgraph_features: x[row] = [row is a target ? 0 : 1 | feat[node_id[row]][0..Ff)]; zero rows unless node_off[B] == n
 * ------------------------------------------------------------------------------------------ */
int npi_subgraph_sizes(const int32_t* ptr, const int32_t* nbr, const uint8_t* ok, const int32_t* keys, int64_t B,
                       int32_t* node_off, int32_t* pair_off, int32_t* workspace, void* stream);
int npi_subgraph_fill(const int32_t* ptr, const int32_t* nbr, const uint8_t* ok, const int32_t* keys, int64_t B,
                      const int32_t* node_off, const int32_t* pair_off, int32_t* node_id, int64_t* batch,
                      int64_t* edge_src, int64_t* edge_dst, int64_t n_nodes, int64_t n_pairs, int32_t* status, void* stream);
int npi_subgraph_features(const float* feat, int64_t ldf, int64_t Ff, const int32_t* node_id, const int64_t* batch,
                          const int32_t* node_off, int64_t B, int64_t n, float* x, int64_t ldx, void* stream);

/* Evaluation loop (reference src/methods.py:87-105 compares one element per Python
 * iteration, one device sync each).  counts[4] += [TP, FN, TN, FP] for pred = first arg-max of scores[i, 0..C):
 * pred 1 & y 1 -> TP, pred 1 & y 0 -> FP, pred 0 & y 1 -> FN, anything else -> TN (the reference's else
 * branch).  counts is device memory, accumulated across calls; no synchronisation. */
int npi_confusion_update(const float* scores, int64_t lds, int64_t C, const int64_t* y, int64_t B, int64_t* counts,
                         void* stream);

/* ------------------------------------------------------------------------------------------
 * The readout sum and MLP head of Net_1 (reference src/classes.py:74-80):
 *     x = x1 + x2 + x3; relu(lin1) -> dropout -> relu(lin2) -> lin3 -> log_softmax
 * as one forward and two backward launches (a 200-row batch: every library GEMM / element-wise kernel of the head is
 * launch-latency bound).  Weights in torch.nn.Linear layout W [out, in], 16-byte aligned; D0 <= 1024, D1, D2 <= 256 (all
 * multiples of 4), D3 <= 32.  r2 / r3 may be NULL (fewer readouts).  mask [B, D1] holds 0 / 1 (NULL: evaluation), scale =
 * 1 / (1 - p).  Saved for the backward (may be NULL in evaluation): s = the summed input [B, D0], h1 = relu(lin1) before
 * dropout [B, D1], h2 [B, D2].  npi_mlp_head_bwd: ds (may be NULL) is the gradient of EACH readout; dW* / db* are written,
 * not accumulated; every sum over the batch runs in row order (deterministic).
 * activation: what follows lin3 -- NPI_HEAD_LOG_SOFTMAX (Net_1) or NPI_HEAD_SIGMOID (the one-output variant,
 * src/train_with_twoDataset_modelOnlyOneOutput.py:45-82: lin3 is 64 -> 1, `torch.sigmoid`, trained with binary cross entropy);
 * `logp` / `dlogp` are then the sigmoid's output and its gradient.
 * ------------------------------------------------------------------------------------------ */
#define NPI_HEAD_LOG_SOFTMAX 0
#define NPI_HEAD_SIGMOID 1
int npi_mlp_head_fwd(const float* r1, int64_t ld1, const float* r2, int64_t ld2, const float* r3, int64_t ld3,
                     int64_t B, int64_t D0, const float* W1, const float* b1, int64_t D1, const float* W2, const float* b2,
                     int64_t D2, const float* W3, const float* b3, int64_t D3, const float* mask, float scale, int activation,
                     float* s, float* h1, float* h2, float* logp, void* stream);
int64_t npi_mlp_head_workspace_elems(int64_t B, int64_t D1, int64_t D2, int64_t D3);
int npi_mlp_head_bwd(int64_t B, int64_t D0, int64_t D1, int64_t D2, int64_t D3, const float* W1, const float* W2,
                     const float* W3, const float* mask, float scale, int activation, const float* s, const float* h1, const float* h2,
                     const float* logp, const float* dlogp, float* ds, float* dW1, float* db1, float* dW2, float* db2,
                     float* dW3, float* db3, float* workspace, int64_t workspace_elems, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NPI_GNN_H */
