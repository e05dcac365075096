// COO (int64, as PyG hands it over) -> destination-sorted CSR (int32) with the self loop of
// add_remaining_self_loops appended as the LAST entry of every row.
//
// Replaces, for reference src/classes.py:62,66,70 (SAGEConv.forward -> PyG 1.4.2
// utils.add_remaining_self_loops + the per-target grouping torch_scatter does with atomics).
//
// Stable LSD radix sort on the key node id (8-bit digits, ceil(log2(N+1)/8) passes):
// entries of a row keep their edge_index order, so the reduction order equals the
// reference's CPU scatter order and every run is bitwise identical.
#include "npi_common.h"
#include <stdarg.h>

namespace npi {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

constexpr int SORT_THREADS = 256;
constexpr int SORT_WAVES = SORT_THREADS / WAVE;
constexpr int SORT_ITEMS = 16;                       // keys per lane
constexpr int SORT_WAVE_TILE = WAVE * SORT_ITEMS;    // 1024 keys per wave, striped: key j*64+lane
constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS; // 4096 keys per workgroup
constexpr int RADIX = 256;

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;
static_assert(SCAN_THREADS == SORT_THREADS && SORT_THREADS == 256, "one thread per radix digit");

// key = key node, or N (sentinel, sorts behind every row) for dropped columns
__global__ void make_keys_kernel(const int64_t* __restrict__ key_nodes,
                                 const int64_t* __restrict__ val_nodes, int64_t E, int64_t N,
                                 int64_t n_cols, int drop_equal,
                                 uint32_t* __restrict__ keys, int32_t* __restrict__ vals,
                                 int32_t* __restrict__ status) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    int64_t k = key_nodes[e], v = val_nodes[e];
    // (-1, -1) is a PADDING column: npi_filter_adj(pad_tail) keeps its output at the input's length and fills the tail with it,
    // so that the edge count never has to be read back; dropped without raising the out-of-range flag
    const bool pad = (k == -1) & (v == -1);
    bool bad = !pad & ((k < 0) | (k >= N) | (v < 0) | (v >= n_cols));
    if (bad) atomicOr(status, 1);
    keys[e] = (pad || bad || (drop_equal && k == v)) ? (uint32_t)N : (uint32_t)k;
    vals[e] = (int32_t)e;
}

// NPI_CSR_SORT_COLUMNS, first sort: key = the column node (out-of-range columns: n_cols, behind everything; make_keys_kernel's
// rules decide later what becomes of them), val = the entry's position in the caller's list
__global__ void col_keys_kernel(const int64_t* __restrict__ val_nodes, int64_t E, int64_t n_cols,
                                uint32_t* __restrict__ keys, int32_t* __restrict__ vals) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int64_t v = val_nodes[e];
    keys[e] = (v < 0 || v >= n_cols) ? (uint32_t)n_cols : (uint32_t)v;
    vals[e] = (int32_t)e;
}

// ... second sort's keys: make_keys_kernel's rule for the entry that the column sort left at position i
__global__ void rekey_kernel(const int64_t* __restrict__ key_nodes, const int64_t* __restrict__ val_nodes,
                             const int32_t* __restrict__ vals, int64_t E, int64_t N, int64_t n_cols, int drop_equal,
                             uint32_t* __restrict__ keys, int32_t* __restrict__ status) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E) return;
    const int64_t e = vals[i];
    const int64_t k = key_nodes[e], v = val_nodes[e];
    const bool pad = (k == -1) & (v == -1);
    const bool bad = !pad & ((k < 0) | (k >= N) | (v < 0) | (v >= n_cols));
    if (bad) atomicOr(status, 1);
    keys[i] = (pad || bad || (drop_equal && k == v)) ? (uint32_t)N : (uint32_t)k;
}

__global__ void __launch_bounds__(SORT_THREADS)
radix_hist_kernel(const uint32_t* __restrict__ keys, int64_t n, int shift, int nblocks,
                  int32_t* __restrict__ counts) {
    __shared__ int hist[RADIX];
    hist[threadIdx.x] = 0;
    __syncthreads();
    int64_t base = (int64_t)blockIdx.x * SORT_TILE;
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        int64_t i = base + j * SORT_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&hist[(keys[i] >> shift) & (RADIX - 1)], 1);
    }
    __syncthreads();
    counts[(int64_t)threadIdx.x * nblocks + blockIdx.x] = hist[threadIdx.x];   // digit-major
}

// ---- exclusive scan of an int32 array (three launches) -------------------------------------
__device__ __forceinline__ int block_exclusive_scan(int v, int* lds /*[SCAN_THREADS]*/, int* total) {
    lds[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < SCAN_THREADS; off <<= 1) {
        int t = (threadIdx.x >= off) ? lds[threadIdx.x - off] : 0;
        __syncthreads();
        lds[threadIdx.x] += t;
        __syncthreads();
    }
    int incl = lds[threadIdx.x];
    *total = lds[SCAN_THREADS - 1];
    __syncthreads();
    return incl - v;
}

__global__ void __launch_bounds__(SCAN_THREADS)
scan_tiles_kernel(int32_t* __restrict__ data, int64_t n, int32_t* __restrict__ tile_sums) {
    __shared__ int lds[SCAN_THREADS];
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        v[j] = (base + j < n) ? data[base + j] : 0;
        s += v[j];
    }
    int total;
    int excl = block_exclusive_scan(s, lds, &total);
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        if (base + j < n) data[base + j] = excl;
        excl += v[j];
    }
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

__global__ void __launch_bounds__(SCAN_THREADS)
scan_sums_kernel(int32_t* __restrict__ sums, int64_t n) {   // one workgroup
    __shared__ int lds[SCAN_THREADS];
    int carry = 0;
    for (int64_t base = 0; base < n; base += SCAN_THREADS) {
        int64_t i = base + threadIdx.x;
        int v = (i < n) ? sums[i] : 0;
        int total;
        int excl = block_exclusive_scan(v, lds, &total);
        if (i < n) sums[i] = carry + excl;
        carry += total;
    }
}

__global__ void __launch_bounds__(SCAN_THREADS)
scan_add_kernel(int32_t* __restrict__ data, int64_t n, const int32_t* __restrict__ tile_sums) {
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int add = tile_sums[blockIdx.x];
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j)
        if (base + j < n) data[base + j] += add;
}

// ---- stable scatter of one radix pass ----------------------------------------------------------
// SCANNED: `offsets` is the exclusive scan of the digit-major tile histograms (three scan launches before this one).
// !SCANNED (at most SMALL_SORT_TILES tiles -- the reference's 200-subgraph batches, launch-bound): `offsets` holds the
// raw histograms and every workgroup derives its own start offsets from them, two launches per pass instead of five.
constexpr int SMALL_SORT_TILES = 64;
template <bool SCANNED>
__global__ void __launch_bounds__(SORT_THREADS)
radix_scatter_kernel(const uint32_t* __restrict__ keys_in, const int32_t* __restrict__ vals_in,
                     uint32_t* __restrict__ keys_out, int32_t* __restrict__ vals_out,
                     int64_t n, int shift, int nblocks, const int32_t* __restrict__ offsets) {
    __shared__ int wcnt[SORT_WAVES][RADIX];
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < SORT_WAVES * RADIX; i += SORT_THREADS) (&wcnt[0][0])[i] = 0;
    __syncthreads();

    const int64_t base = (int64_t)blockIdx.x * SORT_TILE + (int64_t)wave * SORT_WAVE_TILE;
    uint32_t key[SORT_ITEMS];
    int32_t val[SORT_ITEMS];
    int rank[SORT_ITEMS];
    const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        int64_t i = base + j * WAVE + lane;
        bool valid = i < n;
        key[j] = valid ? keys_in[i] : 0xFFFFFFFFu;
        val[j] = valid ? vals_in[i] : 0;
        // tail lanes rank as digit 255 of the last tile: they sort behind everything and are not written
        int digit = valid ? (int)((key[j] >> shift) & (RADIX - 1)) : (RADIX - 1);
        uint64_t same = ~0ull;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            bool bit = (digit >> b) & 1;
            uint64_t m = __ballot(bit);
            same &= bit ? m : ~m;
        }
        int before = __popcll(same & lt_mask);
        int prev = wcnt[wave][digit];
        __builtin_amdgcn_wave_barrier();
        if (before == 0) wcnt[wave][digit] = prev + __popcll(same);
        __builtin_amdgcn_wave_barrier();
        rank[j] = prev + before;
    }
    __syncthreads();
    {   // thread d owns digit d: turn per-wave counts into global start offsets
        int d = threadIdx.x;
        int off;
        if (SCANNED) {
            off = offsets[(int64_t)d * nblocks + blockIdx.x];
        } else {
            __shared__ int scan_s[SCAN_THREADS];
            const int32_t* __restrict__ h = offsets + (int64_t)d * nblocks;
            int total = 0, before = 0;
            for (int t = 0; t < nblocks; ++t) {
                const int c = h[t];
                before += (t < (int)blockIdx.x) ? c : 0;
                total += c;
            }
            int all;
            off = block_exclusive_scan(total, scan_s, &all) + before;   // keys with a smaller digit + same digit in earlier tiles
        }
#pragma unroll
        for (int w = 0; w < SORT_WAVES; ++w) {
            int c = wcnt[w][d];
            wcnt[w][d] = off;
            off += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        int64_t i = base + j * WAVE + lane;
        if (i < n) {
            int digit = (int)((key[j] >> shift) & (RADIX - 1));
            int pos = wcnt[wave][digit] + rank[j];
            keys_out[pos] = key[j];
            vals_out[pos] = val[j];
        }
    }
}

__device__ __forceinline__ int64_t first_key_at_least(const uint32_t* __restrict__ keys, int64_t E, int64_t r) {
    int64_t lo = 0, hi = E;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < (uint32_t)r) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// One launch writes the whole CSR from the sorted (key, column) stream.
// Workgroups [0, entry_blocks): entry p of the sorted stream lands at p (+ key when every row gets a self loop behind it).
// Workgroups behind them: row r -- rowptr[r] = first sorted position with key >= r (+ r), and the appended self loop.
__global__ void __launch_bounds__(256)
fill_csr_kernel(const uint32_t* __restrict__ keys, const int32_t* __restrict__ vals,
                const int64_t* __restrict__ val_nodes, int64_t E, int64_t N, int loops, int32_t loop_col_offset,
                unsigned entry_blocks, int32_t* __restrict__ rowptr, int32_t* __restrict__ col,
                int32_t* __restrict__ eid, int32_t* __restrict__ rowidx) {
    if (blockIdx.x < entry_blocks) {
        int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (p >= E) return;
        uint32_t k = keys[p];
        if (k >= (uint32_t)N) return;
        int32_t e = vals[p];
        int64_t q = p + (loops ? (int64_t)k : 0);
        col[q] = (int32_t)val_nodes[e];
        if (eid) eid[q] = e;
        if (rowidx) rowidx[q] = (int32_t)k;
        return;
    }
    // row r starts at the first sorted position with key >= r; its end is the next thread's start (one search per
    // row, the workgroup's last thread searches once more)
    __shared__ int64_t lb_s[257];
    int64_t r = (int64_t)(blockIdx.x - entry_blocks) * 256 + threadIdx.x;
    const bool have = r <= N;
    const int64_t lb = have ? first_key_at_least(keys, E, r) : E;
    lb_s[threadIdx.x] = lb;
    if (threadIdx.x == 255) lb_s[256] = (r + 1 <= N) ? first_key_at_least(keys, E, r + 1) : E;
    __syncthreads();
    if (!have) return;
    rowptr[r] = (int32_t)(lb + (loops ? r : 0));
    if (loops && r < N) {
        int64_t q = lb_s[threadIdx.x + 1] + r;                  // behind the last entry of row r
        col[q] = (int32_t)r + loop_col_offset;
        if (eid) eid[q] = -1;
        if (rowidx) rowidx[q] = (int32_t)r;
    }
}

// item_row[i] = row holding entry i*item (item 0 starts at row 0 so that leading empty
// rows get written); N for items past nnz.
__global__ void item_rows_kernel(const int32_t* __restrict__ rowptr, int64_t N, int64_t n_items, int item,
                                 int32_t* __restrict__ item_row) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_items) return;
    int64_t nnz = rowptr[N];
    int64_t k = i * item;
    if (i == 0) { item_row[0] = 0; return; }
    if (k >= nnz) { item_row[i] = (int32_t)N; return; }
    // upper_bound(rowptr[0..N], k) - 1
    int64_t lo = 0, hi = N + 1;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if ((int64_t)rowptr[mid] <= k) lo = mid + 1; else hi = mid;
    }
    item_row[i] = (int32_t)(lo - 1);
}

__global__ void edge_positions_kernel(const int32_t* __restrict__ eid, const int32_t* __restrict__ rowptr,
                                      int64_t N, int64_t nnz_max, int32_t* __restrict__ pos_of) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nnz_max || p >= rowptr[N]) return;
    int32_t e = eid[p];
    if (e >= 0) pos_of[e] = (int32_t)p;
}

__global__ void fill_i32_kernel(int32_t* __restrict__ p, int64_t n, int32_t v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

struct SortLayout {
    int64_t nblocks, counts_len, ntiles;
    int64_t off_keys_a, off_keys_b, off_vals_a, off_vals_b, off_counts, off_tiles, total;
};

static SortLayout sort_layout(int64_t E, int64_t N) {
    SortLayout L;
    L.nblocks = ceil_div(E > 0 ? E : 1, SORT_TILE);
    L.counts_len = L.nblocks * RADIX;
    L.ntiles = ceil_div(L.counts_len, SCAN_TILE);
    int64_t o = 0;
    auto take = [&](int64_t bytes) { int64_t r = o; o += align_up(bytes, 256); return r; };
    int64_t Ee = E > 0 ? E : 1;
    L.off_keys_a = take(Ee * 4);
    L.off_keys_b = take(Ee * 4);
    L.off_vals_a = take(Ee * 4);
    L.off_vals_b = take(Ee * 4);
    L.off_counts = take(L.counts_len * 4);
    L.off_tiles = take(L.ntiles * 4);
    L.total = o;
    return L;
}

// stable LSD radix sort of (key, val) pairs on the low `bits` bits of the key, 8 bits per pass; the sorted pairs end up in
// (keys_a, vals_a) -- the pointers are swapped pass by pass
static void radix_sort_pairs(uint32_t*& keys_a, uint32_t*& keys_b, int32_t*& vals_a, int32_t*& vals_b, int64_t n, int bits,
                             const SortLayout& L, int32_t* counts, int32_t* tiles, hipStream_t stream) {
    const int passes = (bits + 7) / 8;
    for (int p = 0; p < passes; ++p) {
        const int shift = 8 * p;
        radix_hist_kernel<<<(unsigned)L.nblocks, SORT_THREADS, 0, stream>>>(keys_a, n, shift, (int)L.nblocks, counts);
        if (L.nblocks <= SMALL_SORT_TILES) {
            radix_scatter_kernel<false><<<(unsigned)L.nblocks, SORT_THREADS, 0, stream>>>(keys_a, vals_a, keys_b, vals_b, n, shift, (int)L.nblocks, counts);
        } else {
            scan_tiles_kernel<<<(unsigned)L.ntiles, SCAN_THREADS, 0, stream>>>(counts, L.counts_len, tiles);
            scan_sums_kernel<<<1, SCAN_THREADS, 0, stream>>>(tiles, L.ntiles);
            scan_add_kernel<<<(unsigned)L.ntiles, SCAN_THREADS, 0, stream>>>(counts, L.counts_len, tiles);
            radix_scatter_kernel<true><<<(unsigned)L.nblocks, SORT_THREADS, 0, stream>>>(keys_a, vals_a, keys_b, vals_b, n, shift, (int)L.nblocks, counts);
        }
        uint32_t* tk = keys_a; keys_a = keys_b; keys_b = tk;
        int32_t* tv = vals_a; vals_a = vals_b; vals_b = tv;
    }
}

// ---- TopKPooling selection for graphs of ANY size (pool.hip sorts a graph's scores in LDS: up to 16,384 nodes) -----------
// Two stable radix sorts of the whole batch: by score (descending; the sort is stable over the node index, so equal scores
// keep the lower index first -- torch.sort(descending=True, stable) order, as the LDS kernel), then by graph id: position
// graph_ptr[g] + r of the result is the node of graph g with the r-th highest score.
__global__ void topk_score_keys_kernel(const float* __restrict__ score, int64_t N, uint32_t* __restrict__ keys,
                                       int32_t* __restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float f = score[i];
    uint32_t u = __float_as_uint(f);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);     // ascending-orderable
    keys[i] = (f != f) ? 0u : ~u;                       // descending; NaN first, as torch.sort treats it as the largest
    vals[i] = (int32_t)i;
}
__global__ void topk_graph_keys_kernel(const int64_t* __restrict__ batch, const int32_t* __restrict__ vals, int64_t N,
                                       uint32_t* __restrict__ keys) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < N) keys[q] = (uint32_t)batch[vals[q]];
}
// out_ptr[g] = sum_{g' < g} min(ceil(ratio n_g'), n_g') in the f32 arithmetic of pool.hip's topk_counts_kernel; one workgroup
__global__ void __launch_bounds__(256)
topk_kept_offsets_kernel(const int32_t* __restrict__ graph_ptr, int B, float ratio, int32_t* __restrict__ out_ptr) {
    __shared__ int lds[256];
    int carry = 0;
    for (int base = 0; base < B; base += 256) {
        const int g = base + threadIdx.x;
        int k = 0;
        if (g < B) {
            const int n = graph_ptr[g + 1] - graph_ptr[g];
            k = min((int)ceilf(ratio * (float)n), n);
        }
        int total;
        const int excl = block_exclusive_scan(k, lds, &total);
        if (g < B) out_ptr[g] = carry + excl;
        carry += total;
    }
    if (threadIdx.x == 0) out_ptr[B] = carry;
}
__global__ void topk_take_kernel(const uint32_t* __restrict__ graph_of, const int32_t* __restrict__ node_of, int64_t N,
                                 const int32_t* __restrict__ graph_ptr, const int32_t* __restrict__ out_ptr,
                                 int32_t* __restrict__ perm, int32_t* __restrict__ remap) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= N) return;
    const int g = (int)graph_of[q], node = node_of[q];
    const int rank = (int)(q - graph_ptr[g]);
    const int k = out_ptr[g + 1] - out_ptr[g];
    const int at = rank < k ? out_ptr[g] + rank : -1;
    if (at >= 0) perm[at] = node;
    remap[node] = at;                                    // every node occurs exactly once: kept -> new id, dropped -> -1
}

}  // namespace npi

using namespace npi;

extern "C" const char* npi_last_error(void) { return npi::g_err; }
extern "C" int npi_abi_version(void) { return 4; }

extern "C" int64_t npi_csr_workspace_bytes(int64_t E, int64_t N) {
    if (E < 0 || N < 0) return -1;
    return sort_layout(E, N).total;
}

extern "C" int64_t npi_item_edges(int64_t nnz_max) { return npi::item_edges_for(nnz_max); }

extern "C" int64_t npi_num_items(int64_t nnz_max, int64_t item_edges) {
    if (!npi::item_edges_ok(item_edges)) return -1;
    return npi::num_items_of(nnz_max, item_edges);
}

extern "C" int npi_csr_build_ex(const int64_t* key_nodes, const int64_t* val_nodes, int64_t E, int64_t N,
                                int64_t n_cols, int add_self_loops, int64_t loop_col_offset, int build_flags,
                                int32_t* rowptr, int32_t* col, int32_t* eid,
                                int32_t* rowidx, int32_t* item_row, int64_t item_edges, int32_t* status,
                                void* workspace, int64_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(E >= 0 && N >= 0 && n_cols >= 0, "npi_csr_build: negative size");
    NPI_REQUIRE((build_flags & ~(NPI_CSR_DROP_EQUAL | NPI_CSR_SORT_COLUMNS)) == 0, "npi_csr_build: unknown build flag");
    NPI_REQUIRE(item_edges_ok(item_edges), "npi_csr_build: item_edges must be 64 or NPI_ITEM_EDGES (npi_item_edges gives the hint)");
    NPI_REQUIRE(E + N + 1 < (int64_t)0x7fffffff, "npi_csr_build: E + N does not fit int32");
    NPI_REQUIRE(n_cols < (int64_t)0x7fffffff && loop_col_offset >= 0 && loop_col_offset + N <= (n_cols > N ? n_cols : N),
                "npi_csr_build: column range does not fit");
    NPI_REQUIRE(rowptr && item_row && status && workspace, "npi_csr_build: null output");
    NPI_REQUIRE(E == 0 || (key_nodes && val_nodes && col), "npi_csr_build: null edge arrays");
    SortLayout L = sort_layout(E, N);
    if (workspace_bytes < L.total) {
        set_error("npi_csr_build: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)L.total);
        return NPI_ERR_WORKSPACE;
    }
    char* ws = (char*)workspace;
    uint32_t* keys_a = (uint32_t*)(ws + L.off_keys_a);
    uint32_t* keys_b = (uint32_t*)(ws + L.off_keys_b);
    int32_t* vals_a = (int32_t*)(ws + L.off_vals_a);
    int32_t* vals_b = (int32_t*)(ws + L.off_vals_b);
    int32_t* counts = (int32_t*)(ws + L.off_counts);
    int32_t* tiles = (int32_t*)(ws + L.off_tiles);

    (void)hipMemsetAsync(status, 0, sizeof(int32_t), stream);
    const int drop_equal = build_flags & NPI_CSR_DROP_EQUAL;
    if (E > 0) {
        const unsigned eb = (unsigned)ceil_div(E, 256);
        if (build_flags & NPI_CSR_SORT_COLUMNS) {
            // LSD over the pair (row, column): the stable sort by column first, then the stable sort by row keeps it inside a row
            col_keys_kernel<<<eb, 256, 0, stream>>>(val_nodes, E, n_cols, keys_a, vals_a);
            int cbits = 1;
            while (((int64_t)1 << cbits) <= n_cols) ++cbits;
            radix_sort_pairs(keys_a, keys_b, vals_a, vals_b, E, cbits, L, counts, tiles, stream);
            rekey_kernel<<<eb, 256, 0, stream>>>(key_nodes, val_nodes, vals_a, E, N, n_cols, drop_equal, keys_a, status);
        } else {
            make_keys_kernel<<<eb, 256, 0, stream>>>(key_nodes, val_nodes, E, N, n_cols, drop_equal, keys_a, vals_a, status);
        }
        int bits = 1;
        while (((int64_t)1 << bits) <= N) ++bits;       // keys lie in [0, N]
        radix_sort_pairs(keys_a, keys_b, vals_a, vals_b, E, bits, L, counts, tiles, stream);
    }
    const unsigned entry_blocks = (unsigned)ceil_div(E, 256);
    fill_csr_kernel<<<entry_blocks + (unsigned)ceil_div(N + 1, 256), 256, 0, stream>>>(
        keys_a, vals_a, val_nodes, E, N, add_self_loops, (int32_t)loop_col_offset, entry_blocks, rowptr, col, eid, rowidx);
    const int64_t nnz_max = E + (add_self_loops ? N : 0);
    int64_t n_items = num_items_of(nnz_max, item_edges);
    item_rows_kernel<<<(unsigned)ceil_div(n_items + 1, 256), 256, 0, stream>>>(rowptr, N, n_items, (int)item_edges, item_row);
    return check_launch("npi_csr_build");
}

extern "C" int64_t npi_topk_sorted_workspace_bytes(int64_t N) {
    if (N < 0) return -1;
    return sort_layout(N, 0).total;
}

extern "C" int npi_topk_select_sorted(const float* score, const int64_t* batch, const int32_t* graph_ptr, int64_t N, int64_t B,
                                      float ratio, int32_t* out_ptr, int32_t* perm, int32_t* remap, void* workspace,
                                      int64_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(N >= 0 && B >= 0 && N < (int64_t)0x7fffffff && ratio > 0.f && ratio <= 1.f, "npi_topk_select_sorted: bad argument");
    NPI_REQUIRE(out_ptr && graph_ptr && (N == 0 || (score && batch && perm && remap && workspace)), "npi_topk_select_sorted: null pointer");
    topk_kept_offsets_kernel<<<1, 256, 0, stream>>>(graph_ptr, (int)B, ratio, out_ptr);
    if (N == 0 || B == 0) return check_launch("npi_topk_select_sorted");
    SortLayout L = sort_layout(N, 0);
    if (workspace_bytes < L.total) {
        set_error("npi_topk_select_sorted: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)L.total);
        return NPI_ERR_WORKSPACE;
    }
    char* ws = (char*)workspace;
    uint32_t* keys_a = (uint32_t*)(ws + L.off_keys_a);
    uint32_t* keys_b = (uint32_t*)(ws + L.off_keys_b);
    int32_t* vals_a = (int32_t*)(ws + L.off_vals_a);
    int32_t* vals_b = (int32_t*)(ws + L.off_vals_b);
    int32_t* counts = (int32_t*)(ws + L.off_counts);
    int32_t* tiles = (int32_t*)(ws + L.off_tiles);
    const unsigned nb = (unsigned)ceil_div(N, 256);
    topk_score_keys_kernel<<<nb, 256, 0, stream>>>(score, N, keys_a, vals_a);
    radix_sort_pairs(keys_a, keys_b, vals_a, vals_b, N, 32, L, counts, tiles, stream);
    topk_graph_keys_kernel<<<nb, 256, 0, stream>>>(batch, vals_a, N, keys_a);
    int bits = 1;
    while (((int64_t)1 << bits) < B) ++bits;             // graph ids lie in [0, B)
    radix_sort_pairs(keys_a, keys_b, vals_a, vals_b, N, bits, L, counts, tiles, stream);
    topk_take_kernel<<<nb, 256, 0, stream>>>(keys_a, vals_a, N, graph_ptr, out_ptr, perm, remap);
    return check_launch("npi_topk_select_sorted");
}

// ---- the CSR of a pooled graph from the CSR of its parent: no sort ------------------------------------------------------
// TopKPooling keeps a subset of the nodes, renumbers them (perm: new -> old, remap: old -> new or -1) and filter_adj drops
// every edge that lost an endpoint while keeping the edge order.  The by-target CSR of the result is therefore the parent's,
// row perm[r'] for new row r', with the entries whose source survived, in the same order (the self loop still last):
// count per new row, ordered fill (offsets from per-tile totals), item rows -- three launches instead of the eight of a fresh radix sort.
namespace npi {
// Tiles of 256 new rows, one thread per row: the enclosing subgraphs are double stars, so almost every row has two or three
// entries and a lane walks its own row; the few long rows (the two centres of a subgraph, hundreds of entries) are taken by
// the whole wave, one after the other.  count: per-row counts and the tile's total.  fill: the tile's base is the sum of the
// totals in front of it (a few hundred integers), the rows' offsets an exclusive scan inside the tile -- no scan launch.
constexpr int CF_SHORT = 8;
__device__ __forceinline__ int cf_wave_sum(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
}
__global__ void __launch_bounds__(256)
csr_filter_count_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const int32_t* __restrict__ perm,
                        const int32_t* __restrict__ remap, int n_out, int32_t* __restrict__ cnt, int32_t* __restrict__ tile_total) {
    __shared__ int wtot[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r2 = blockIdx.x * 256 + t;
    const bool valid = r2 < n_out;
    const int r = valid ? perm[r2] : 0;
    const int b = valid ? rowptr[r] : 0, e = valid ? rowptr[r + 1] : 0;
    const bool is_long = e - b > CF_SHORT;
    int c = 0;
    if (!is_long) for (int p = b; p < e; ++p) c += remap[col[p]] >= 0 ? 1 : 0;
    uint64_t m = __ballot(is_long);
    while (m) {
        const int l = __ffsll((long long)m) - 1;
        m &= m - 1;
        const int bb = __shfl(b, l, WAVE), ee = __shfl(e, l, WAVE);
        int cc = 0;
        for (int p = bb + lane; p < ee; p += WAVE) cc += remap[col[p]] >= 0 ? 1 : 0;
        cc = cf_wave_sum(cc);
        if (lane == l) c = cc;
    }
    if (valid) cnt[r2] = c;
    const int ws_ = cf_wave_sum(c);
    if (lane == 0) wtot[wave] = ws_;
    __syncthreads();
    if (t == 0) tile_total[blockIdx.x] = wtot[0] + wtot[1] + wtot[2] + wtot[3];
}
__global__ void __launch_bounds__(256)
csr_filter_fill_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const int32_t* __restrict__ eid,
                       const int32_t* __restrict__ perm, const int32_t* __restrict__ remap, const int32_t* __restrict__ newpos,
                       int n_out, const int32_t* __restrict__ cnt, const int32_t* __restrict__ tile_total,
                       int32_t* __restrict__ rowptr_o, int32_t* __restrict__ col_o, int32_t* __restrict__ eid_o,
                       int32_t* __restrict__ rowidx_o) {
    __shared__ int wtot[4], base_s;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // entries in front of this tile
    int s = 0;
    for (int i = t; i < (int)blockIdx.x; i += 256) s += tile_total[i];
    s = cf_wave_sum(s);
    if (lane == 0) wtot[wave] = s;
    __syncthreads();
    if (t == 0) base_s = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    __syncthreads();
    const int base = base_s;
    const int r2 = blockIdx.x * 256 + t;
    const bool valid = r2 < n_out;
    const int c = valid ? cnt[r2] : 0;
    int x = c;                                               // inclusive scan inside the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int y = __shfl_up(x, off, WAVE);
        if (lane >= off) x += y;
    }
    __syncthreads();
    if (lane == 63) wtot[wave] = x;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wtot[w];
    const int start = base + woff + x - c;
    if (valid) {
        rowptr_o[r2] = start;
        if (r2 == n_out - 1) rowptr_o[n_out] = start + c;
    }
    const int r = valid ? perm[r2] : 0;
    const int b = valid ? rowptr[r] : 0, e = valid ? rowptr[r + 1] : 0;
    const bool is_long = e - b > CF_SHORT;
    if (!is_long) {
        int pos = start;
        for (int p = b; p < e; ++p) {
            const int c2 = remap[col[p]];
            if (c2 >= 0) {
                const int e0 = eid[p];
                col_o[pos] = c2;
                eid_o[pos] = e0 >= 0 ? newpos[e0] : -1;      // the self loop keeps -1
                rowidx_o[pos] = r2;
                ++pos;
            }
        }
    }
    uint64_t m = __ballot(is_long);
    while (m) {
        const int l = __ffsll((long long)m) - 1;
        m &= m - 1;
        const int bb = __shfl(b, l, WAVE), ee = __shfl(e, l, WAVE), rr = __shfl(r2, l, WAVE);
        int out = __shfl(start, l, WAVE);
        for (int pb = bb; pb < ee; pb += WAVE) {
            const int p = pb + lane;
            int c2 = -1;
            if (p < ee) c2 = remap[col[p]];
            const uint64_t k = __ballot(c2 >= 0);
            if (c2 >= 0) {
                const int pos = out + __popcll(k & ((1ull << lane) - 1ull));
                const int e0 = eid[p];
                col_o[pos] = c2;
                eid_o[pos] = e0 >= 0 ? newpos[e0] : -1;
                rowidx_o[pos] = rr;
            }
            out += __popcll(k);
        }
    }
}
}  // namespace npi

extern "C" int64_t npi_csr_filter_max_rows(void) { return (int64_t)1 << 24; }      // tile totals summed per tile: above this, sort afresh
extern "C" int64_t npi_csr_filter_workspace_elems(int64_t n_out) { return n_out < 0 ? -1 : n_out + ceil_div(n_out > 0 ? n_out : 1, 256) + 1; }

extern "C" int npi_csr_filter(const int32_t* rowptr, const int32_t* col, const int32_t* eid, const int32_t* perm,
                              const int32_t* remap, const int32_t* newpos, int64_t n_out, int64_t nnz_max_out,
                              int32_t* rowptr_o, int32_t* col_o, int32_t* eid_o, int32_t* rowidx_o, int32_t* item_row_o,
                              int64_t item_edges, int32_t* status_o, int32_t* workspace, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(n_out >= 0 && n_out <= npi_csr_filter_max_rows() && nnz_max_out >= n_out, "npi_csr_filter: bad size");
    NPI_REQUIRE(item_edges_ok(item_edges), "npi_csr_filter: item_edges must be 64 or NPI_ITEM_EDGES");
    NPI_REQUIRE(rowptr_o && item_row_o && status_o && (n_out == 0 || (rowptr && col && eid && perm && remap && newpos && col_o &&
                eid_o && rowidx_o && workspace)), "npi_csr_filter: null pointer");
    (void)hipMemsetAsync(status_o, 0, sizeof(int32_t), stream);            // ids were checked when the parent was built
    if (n_out == 0) {
        (void)hipMemsetAsync(rowptr_o, 0, sizeof(int32_t), stream);
    } else {
        const unsigned tiles = (unsigned)ceil_div(n_out, 256);
        int32_t* cnt = workspace;
        int32_t* tile_total = workspace + n_out;
        csr_filter_count_kernel<<<tiles, 256, 0, stream>>>(rowptr, col, perm, remap, (int)n_out, cnt, tile_total);
        csr_filter_fill_kernel<<<tiles, 256, 0, stream>>>(rowptr, col, eid, perm, remap, newpos, (int)n_out, cnt, tile_total,
                                                          rowptr_o, col_o, eid_o, rowidx_o);
    }
    const int64_t n_items = num_items_of(nnz_max_out, item_edges);
    item_rows_kernel<<<(unsigned)ceil_div(n_items + 1, 256), 256, 0, stream>>>(rowptr_o, n_out, n_items, (int)item_edges, item_row_o);
    return check_launch("npi_csr_filter");
}

extern "C" int npi_edge_positions(const int32_t* eid, const int32_t* rowptr, int64_t N, int64_t nnz_max,
                                  int64_t E, int32_t* pos_of, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(E >= 0 && nnz_max >= 0, "npi_edge_positions: negative size");
    if (E > 0) fill_i32_kernel<<<(unsigned)ceil_div(E, 256), 256, 0, stream>>>(pos_of, E, -1);
    if (nnz_max > 0)
        edge_positions_kernel<<<(unsigned)ceil_div(nnz_max, 256), 256, 0, stream>>>(eid, rowptr, N, nnz_max, pos_of);
    return check_launch("npi_edge_positions");
}
