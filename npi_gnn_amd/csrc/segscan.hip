// Per-row reductions of per-ENTRY scalars over the CSR, item-parallel (GATConv's small kernels):
//
//   npi_seg_rowsum_ex          out[r, h] = sum over the entries p of row r of vals[idx(p), h]       (g_dst, g_src)
//   npi_gat_softmax_stats_ex   (m, s)[r, h] = (max, sum exp(. - max)) of e_p = leaky_relu(a_row[r] + a_col[col[p]]),
//                              optionally e_p itself, per entry, for the aggregation that follows
//
// Round 2 walked the ROWS: a group of 8-64 lanes per row, each entry address derived from rowptr -- 0.28 ms for 84 MB of
// scalars at C4 (0.3 TB/s: a latency chain per row, not a stream).  Here the ENTRIES are streamed instead, exactly like
// the aggregation kernel does it: a wavefront takes one item (64 or 256 consecutive entries, whatever rows they belong
// to), lane l of a block holds entry kb + l and its row id (the CSR's `rowidx`, a coalesced load), a segmented inclusive
// scan keyed by the row id (6 shuffle steps; keys are sorted) leaves every run's total in its last lane, and
//   * a row that begins and ends inside the item is written at once,
//   * the part of a row that began in an earlier item goes to head[item], the part of a row that continues into the next
//     item to tail[item]; seg_chain_kernel adds head[i + 1 .. i_end] to it -- a lane per item for the short chains, the whole
//     wavefront striding over a long one (the 400k-entry hub row: 1,600 partials, 25 per lane; a fixed xor tree folds them) --
//     so the result does not depend on the launch order (no atomics anywhere).
// The softmax statistics use the same skeleton with (max, sum) pairs merged like an online softmax.
#include "segsum.h"

namespace npi {

namespace {

constexpr float NEG_BIG = -3.0e38f;

struct SumOp {
    struct V { float a; };
    __device__ static V identity() { return {0.f}; }
    __device__ static V merge(V x, V y) { return {x.a + y.a}; }         // x = the EARLIER part
    __device__ static V shfl_up(V v, int d) { return {__shfl_up(v.a, d, WAVE)}; }
    __device__ static V shfl_xor(V v, int d) { return {__shfl_xor(v.a, d, WAVE)}; }
    __device__ static V lane(V v, int l) { return {bcast_f(v.a, l)}; }
};

struct SoftmaxOp {
    struct V { float m, s; };                                             // s = sum of exp(. - m)
    __device__ static V identity() { return {NEG_BIG, 0.f}; }
    __device__ static V merge(V x, V y) {
        const float m = fmaxf(x.m, y.m);
        return {m, x.s * __expf(x.m - m) + y.s * __expf(y.m - m)};
    }
    __device__ static V shfl_up(V v, int d) { return {__shfl_up(v.m, d, WAVE), __shfl_up(v.s, d, WAVE)}; }
    __device__ static V shfl_xor(V v, int d) { return {__shfl_xor(v.m, d, WAVE), __shfl_xor(v.s, d, WAVE)}; }
    __device__ static V lane(V v, int l) { return {bcast_f(v.m, l), bcast_f(v.s, l)}; }
};

struct ScanArgs {
    const int32_t* rowptr;
    const int32_t* rowidx;
    const int32_t* col;        // softmax: column (source) of every entry
    const int32_t* map;        // rowsum: entry p takes vals[map[p]] (null: vals[p])
    const float* vals;         // rowsum: [nnz, H]
    const float* a_row;        // softmax: [n_rows, H]
    const float* a_col;        // softmax: [n_cols, H]
    float slope;
    float* e_out;              // softmax: leaky_relu score of every entry [nnz, H], or null
    float* out0;               // rowsum: out [N, H]; softmax: m [N, H]
    float* out1;               // softmax: s [N, H]
    float* head;               // [n_items, H] x sizeof(V) / 4
    float* tail;               // likewise
    int32_t* tail_row;         // [n_items]: row of the item's tail partial, -1 = none
    int N, H, n_items, item;
};

template <class Op> __device__ __forceinline__ void store_v(float* base, int64_t i, typename Op::V v);
template <> __device__ __forceinline__ void store_v<SumOp>(float* base, int64_t i, SumOp::V v) { base[i] = v.a; }
template <> __device__ __forceinline__ void store_v<SoftmaxOp>(float* base, int64_t i, SoftmaxOp::V v) {
    reinterpret_cast<float2*>(base)[i] = make_float2(v.m, v.s);
}
template <class Op> __device__ __forceinline__ typename Op::V load_v(const float* base, int64_t i);
template <> __device__ __forceinline__ SumOp::V load_v<SumOp>(const float* base, int64_t i) { return {base[i]}; }
template <> __device__ __forceinline__ SoftmaxOp::V load_v<SoftmaxOp>(const float* base, int64_t i) {
    const float2 t = reinterpret_cast<const float2*>(base)[i];
    return {t.x, t.y};
}
template <class Op> __device__ __forceinline__ void write_row(const ScanArgs& A, int64_t i, typename Op::V v);
template <> __device__ __forceinline__ void write_row<SumOp>(const ScanArgs& A, int64_t i, SumOp::V v) { A.out0[i] = v.a; }
template <> __device__ __forceinline__ void write_row<SoftmaxOp>(const ScanArgs& A, int64_t i, SoftmaxOp::V v) {
    A.out0[i] = v.m;
    A.out1[i] = v.s;
}

// NB = blocks of 64 entries per item (1 or 4); HH = heads handled TOGETHER (1, 2, 4, 8: the H values of an entry are one
// contiguous 4 HH-byte load and share the key shuffles of the scan; any other H runs the HH = 1 kernel head by head).  All of
// an item's index and value loads are issued before the first scan, and no row bound is ever gathered: "p is the last entry of
// its row" <=> the NEXT entry has another row id (one more coalesced load), "the row began in an earlier item" <=> it is the
// row of entry k0 - 1 (one scalar load per item).
template <class Op, int NB, int HH>
__global__ void __launch_bounds__(256)
seg_items_kernel(ScanArgs A) {
    using V = typename Op::V;
    constexpr bool SOFTMAX = sizeof(V) == 8;
    const int lane = lane_id();
    // (XCD x taking the x-th contiguous eighth of the items instead of every eighth workgroup: the mapped row sum 0.26 -> 0.33 ms at C4)
    const int item = uniform_i(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (item >= A.n_items) return;
    const int nnz = A.rowptr[A.N];
    const int k0 = item * (NB * WAVE);
    const int k1 = min(k0 + NB * WAVE, nnz);
    const int H = A.H;
    if (k0 >= nnz) {
        if (lane == 0) A.tail_row[item] = -1;
        return;
    }
    const int prev_row = k0 > 0 ? uniform_i(A.rowidx[k0 - 1]) : -1;         // the row that may run INTO this item
    const int next_row = k1 < nnz ? uniform_i(A.rowidx[k1]) : -2;            // the row of the first entry behind it
    int key[NB], src[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int p = k0 + b * WAVE + lane;
        key[b] = p < k1 ? A.rowidx[p] : 0x7fffffff;
        if constexpr (SOFTMAX) src[b] = p < k1 ? A.col[p] : 0;
        else src[b] = p < k1 ? (A.map ? A.map[p] : p) : 0;
    }
    int tail_r = -1;
    for (int h0 = 0; h0 < H; h0 += HH) {                                       // (one trip when HH == H)
        V v[NB][HH];
#pragma unroll
        for (int b = 0; b < NB; ++b) {                                        // every value load of the item in flight at once
            const int p = k0 + b * WAVE + lane;
#pragma unroll
            for (int h = 0; h < HH; ++h) {
                v[b][h] = Op::identity();
                if (p < k1) {
                    if constexpr (SOFTMAX) {
                        const float z = A.a_row[(int64_t)key[b] * H + h0 + h] + A.a_col[(int64_t)src[b] * H + h0 + h];
                        const float e = z > 0.f ? z : z * A.slope;
                        if (A.e_out) A.e_out[(int64_t)p * H + h0 + h] = e;
                        v[b][h].m = e;
                        v[b][h].s = 1.f;
                    } else {
                        v[b][h].a = A.vals[(int64_t)src[b] * H + h0 + h];
                    }
                }
            }
        }
        int open_key = -1;                          // row whose entries ran up to the end of the previous block
        V open_val[HH];
#pragma unroll
        for (int h = 0; h < HH; ++h) open_val[h] = Op::identity();
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int kb = k0 + b * WAVE;
            if (kb >= k1) break;
            const int p = kb + lane;
            const bool valid = p < k1;
            const int kk = key[b];
            V x[HH];
#pragma unroll
            for (int h = 0; h < HH; ++h) x[h] = v[b][h];
            // segmented inclusive scan: keys are sorted, so "same key at distance d" means the whole span shares it
#pragma unroll
            for (int d = 1; d < WAVE; d <<= 1) {
                const int ko = __shfl_up(kk, d, WAVE);
                const bool same = lane >= d && ko == kk;
#pragma unroll
                for (int h = 0; h < HH; ++h) {
                    const V o = Op::shfl_up(x[h], d);
                    if (same) x[h] = Op::merge(o, x[h]);
                }
            }
            if (valid && kk == open_key) {                                   // first run: the row was already open
#pragma unroll
                for (int h = 0; h < HH; ++h) x[h] = Op::merge(open_val[h], x[h]);
            }
            // row id of entry p + 1: the next lane, the next block's first lane, or the first entry behind the item
            int key_next = __shfl_down(kk, 1, WAVE);
            const int first_of_next = (b + 1 < NB && kb + WAVE < k1) ? bcast_i(key[b + 1 < NB ? b + 1 : b], 0) : next_row;
            if (lane == WAVE - 1) key_next = first_of_next;
            if (valid && p + 1 == k1) key_next = next_row;
            const bool run_end = valid && (lane == WAVE - 1 || p + 1 == k1 || key_next != kk);
            bool cont = false;                                               // this lane's run continues past the block
            if (run_end) {
                if (key_next != kk) {                                        // p is the row's last entry
#pragma unroll
                    for (int h = 0; h < HH; ++h) {
                        if (kk != prev_row) write_row<Op>(A, (int64_t)kk * H + h0 + h, x[h]);
                        else store_v<Op>(A.head, (int64_t)item * H + h0 + h, x[h]);     // it began in an earlier item
                    }
                } else {
                    cont = true;
                }
            }
            const uint64_t cm = __ballot(cont);
            if (cm) {
                const int cl = __ffsll((unsigned long long)cm) - 1;          // (at most one: the last valid lane)
                open_key = bcast_i(kk, cl);
#pragma unroll
                for (int h = 0; h < HH; ++h) open_val[h] = Op::lane(x[h], cl);
            } else {
                open_key = -1;
#pragma unroll
                for (int h = 0; h < HH; ++h) open_val[h] = Op::identity();
            }
        }
        if (open_key >= 0) {                                                  // a row runs past the end of the item
            if (open_key != prev_row) {
                if (lane == 0) {
#pragma unroll
                    for (int h = 0; h < HH; ++h) store_v<Op>(A.tail, (int64_t)item * H + h0 + h, open_val[h]);
                }
                tail_r = open_key;
            } else if (lane == 0) {
#pragma unroll
                for (int h = 0; h < HH; ++h) store_v<Op>(A.head, (int64_t)item * H + h0 + h, open_val[h]);   // the whole item lies inside one row
            }
        }
    }
    if (lane == 0) A.tail_row[item] = tail_r;
}

// row total of every row cut by an item boundary = tail[i] (+) head[i + 1] (+) ... (+) head[i_end].  Nearly every item ends inside a
// row (rows are 10 - 100 entries, items 64 or 256), and nearly every such chain is ONE or two heads long: a LANE per item walks a
// short chain by itself (a wavefront per item -- round 2 to 5 -- spent 0.1 ms on launching 98k workgroups at C5); the long chains of
// the hub rows (the 400k-entry row: 1,600 partials) are then taken one after the other by the whole wavefront, lanes striding
// over the chain and a fixed xor tree folding them.  No atomics; a row's result depends on its chain length only.
constexpr int CHAIN_SHORT = 4;
template <class Op>
__global__ void __launch_bounds__(256)
seg_chain_kernel(ScanArgs A) {
    using V = typename Op::V;
    const int lane = lane_id();
    const int item = (blockIdx.x * 4 + (threadIdx.x >> 6)) * WAVE + lane;
    const int H = A.H;
    int r = -1, n = 0;
    if (item < A.n_items) {
        r = A.tail_row[item];
        if (r >= A.N) r = -1;                                                 // (never written by seg_items: nothing to trust)
        if (r >= 0) n = min((A.rowptr[r + 1] - 1) / A.item, A.n_items - 1) - item;
    }
    if (r >= 0 && n <= CHAIN_SHORT) {
        for (int hd = 0; hd < H; ++hd) {
            V acc = load_v<Op>(A.tail, (int64_t)item * H + hd);
            for (int t = 1; t <= n; ++t) acc = Op::merge(acc, load_v<Op>(A.head, (int64_t)(item + t) * H + hd));
            write_row<Op>(A, (int64_t)r * H + hd, acc);
        }
    }
    uint64_t todo = __ballot(r >= 0 && n > CHAIN_SHORT);
    while (todo) {
        const int l = __ffsll((unsigned long long)todo) - 1;
        todo &= todo - 1;
        const int it = bcast_i(item, l), nn = bcast_i(n, l), rr = bcast_i(r, l);
        for (int hd = 0; hd < H; ++hd) {
            // four independent partial sums per lane (t mod 4 WAVE): the 10,000-link chain of the C5 hub row was one wavefront's
            // 165 dependent rounds
            V a4[4] = {Op::identity(), Op::identity(), Op::identity(), Op::identity()};
            int t = lane;
            for (; t + 3 * WAVE < nn; t += 4 * WAVE) {
#pragma unroll
                for (int q = 0; q < 4; ++q) a4[q] = Op::merge(a4[q], load_v<Op>(A.head, (int64_t)(it + 1 + t + q * WAVE) * H + hd));
            }
            for (int q = 0; t < nn; t += WAVE, ++q) a4[q] = Op::merge(a4[q], load_v<Op>(A.head, (int64_t)(it + 1 + t) * H + hd));
            V acc = Op::merge(Op::merge(a4[0], a4[1]), Op::merge(a4[2], a4[3]));
#pragma unroll
            for (int d = 1; d < WAVE; d <<= 1) {                              // fixed tree: lower lane = earlier items
                const V o = Op::shfl_xor(acc, d);
                acc = (lane & d) ? Op::merge(o, acc) : Op::merge(acc, o);
            }
            if (lane == 0) write_row<Op>(A, (int64_t)rr * H + hd, Op::merge(load_v<Op>(A.tail, (int64_t)it * H + hd), acc));
        }
    }
}

template <class Op>
int run_scan(ScanArgs A, int64_t nnz_max, float* workspace, int64_t workspace_elems, hipStream_t stream, const char* what) {
    // no item state outlives the call (head / tail / tail_row are this call's scratch): the chunking is chosen here, and the
    // workspace query below sizes for the finest one, so it fits whatever the hint is at call time
    A.item = item_edges_for(nnz_max);
    const int64_t n_items = num_items_of(nnz_max, A.item);
    const int64_t per = (int64_t)sizeof(typename Op::V) / 4 * A.H;
    if (n_items == 0) return NPI_OK;
    if (workspace == nullptr || workspace_elems < 2 * per * n_items + n_items + 2) {
        set_error("%s: workspace too small", what);
        return NPI_ERR_WORKSPACE;
    }
    float* ws = reinterpret_cast<float*>(align_up((int64_t)(uintptr_t)workspace, 8));    // float2 slots
    A.head = ws;
    A.tail = ws + per * n_items;
    A.tail_row = reinterpret_cast<int32_t*>(A.tail + per * n_items);
    A.n_items = (int)n_items;
    const unsigned grid = (unsigned)ceil_div(n_items, 4);
    constexpr int NBL = NPI_ITEM_EDGES / WAVE;
    const bool small = A.item == WAVE;
#define NPI_SCAN(HH_) do { if (small) seg_items_kernel<Op, 1, HH_><<<grid, 256, 0, stream>>>(A); \
                           else       seg_items_kernel<Op, NBL, HH_><<<grid, 256, 0, stream>>>(A); } while (0)
    switch (A.H) {                       // 2 / 4 / 8 heads travel together; any other count head by head
        case 2: NPI_SCAN(2); break;
        case 4: NPI_SCAN(4); break;
        case 8: NPI_SCAN(8); break;
        default: NPI_SCAN(1); break;
    }
#undef NPI_SCAN
    seg_chain_kernel<Op><<<(unsigned)ceil_div(n_items, 4 * WAVE), 256, 0, stream>>>(A);
    return check_launch(what);
}

}  // namespace

int seg_chain_sum(const int32_t* rowptr, float* head, float* tail, int32_t* tail_row, float* out, int64_t N, int64_t n_items, int item,
                  hipStream_t stream) {
    if (n_items == 0) return NPI_OK;
    ScanArgs A{};
    A.rowptr = rowptr; A.head = head; A.tail = tail; A.tail_row = tail_row; A.out0 = out; A.N = (int)N; A.H = 1;
    A.n_items = (int)n_items; A.item = item;
    seg_chain_kernel<SumOp><<<(unsigned)ceil_div(n_items, 4 * WAVE), 256, 0, stream>>>(A);
    return check_launch("seg_chain_sum");
}

}  // namespace npi

using namespace npi;

extern "C" int64_t npi_seg_scan_workspace_elems(int64_t nnz_max, int64_t H) {
    if (nnz_max < 0 || H <= 0) return -1;
    const int64_t items = num_items_of(nnz_max, WAVE);          // the finest chunking run_scan may choose
    return 4 * H * items + items + 4;
}

extern "C" int npi_seg_rowsum_ex(const int32_t* rowptr, const int32_t* rowidx, const float* vals, const int32_t* map,
                                 int64_t N, int64_t nnz_max, int64_t H, float* out, float* workspace, int64_t workspace_elems,
                                 void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(N >= 0 && H > 0 && H <= 64 && nnz_max >= 0 && N < 0x7fffffff && nnz_max < 0x7fffffff, "npi_seg_rowsum_ex: bad size");
    if (N == 0) return NPI_OK;
    NPI_REQUIRE(rowptr && out && (nnz_max == 0 || (rowidx && vals)), "npi_seg_rowsum_ex: null pointer");
    (void)hipMemsetAsync(out, 0, sizeof(float) * N * H, stream);              // rows without an entry
    ScanArgs A{};
    A.rowptr = rowptr; A.rowidx = rowidx; A.map = map; A.vals = vals; A.out0 = out; A.N = (int)N; A.H = (int)H;
    return run_scan<SumOp>(A, nnz_max, workspace, workspace_elems, stream, "npi_seg_rowsum_ex");
}

extern "C" int npi_gat_softmax_stats_ex(const int32_t* rowptr, const int32_t* col, const int32_t* rowidx, const float* a_row,
                                        const float* a_col, int64_t N, int64_t nnz_max, int64_t H, float slope, float* m,
                                        float* s, float* e_out, float* workspace, int64_t workspace_elems, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(N >= 0 && H > 0 && H <= 64 && nnz_max >= 0 && N < 0x7fffffff && nnz_max < 0x7fffffff, "npi_gat_softmax_stats_ex: bad size");
    if (N == 0) return NPI_OK;
    NPI_REQUIRE(rowptr && m && s && (nnz_max == 0 || (col && rowidx && a_row && a_col)), "npi_gat_softmax_stats_ex: null pointer");
    (void)hipMemsetAsync(m, 0, sizeof(float) * N * H, stream);                // an empty row: m = 0, s = 0
    (void)hipMemsetAsync(s, 0, sizeof(float) * N * H, stream);
    ScanArgs A{};
    A.rowptr = rowptr; A.rowidx = rowidx; A.col = col; A.a_row = a_row; A.a_col = a_col; A.slope = slope; A.e_out = e_out;
    A.out0 = m; A.out1 = s; A.N = (int)N; A.H = (int)H;
    return run_scan<SoftmaxOp>(A, nnz_max, workspace, workspace_elems, stream, "npi_gat_softmax_stats_ex");
}
