// Fused neighbour gather + per-destination segmented reduction over the destination-sorted CSR.
//
// Replaces  x_j = torch.index_select(x, 0, edge_index[0]);  torch_scatter.scatter_{mean,add}(x_j,
// edge_index[1], dim_size=N)  of PyG 1.4.2 MessagePassing.propagate (reached from reference
// src/classes.py:62,66,70; backward from src/train_with_twoDataset.PY:54) -- the [E+N, F] message
// tensor is never materialised and there are no float atomics.
//
// Work decomposition ("merge-path" over the entry stream): item i = entries [256 i, 256 i + 256)
// of the CSR, one item per 64-lane wavefront, whatever rows those entries belong to.  Every lane
// owns VEC consecutive feature columns of a row (F = 256 f32: one global_load_dwordx4 wave
// instruction == one 1 KiB row), accumulates in registers, and a row is written when its last
// entry has been added -- so a 540k-entry hub row and a 3-entry row cost the same per entry.
// Rows cut by an item boundary leave f32 partial sums in `carry`; the item that holds the row's
// first entry adds them up in item order (segsum_fixup_kernel) => bitwise reproducible.
#include "npi_common.h"

namespace npi {

constexpr int SEG_THREADS = 256;
constexpr int SEG_WAVES = SEG_THREADS / WAVE;
constexpr int T = NPI_ITEM_EDGES;

template <int VEC> struct vec_of;
template <> struct vec_of<1> { using type = float; };
template <> struct vec_of<2> { using type = float2; };
template <> struct vec_of<4> { using type = float4; };

template <int VEC>
__device__ __forceinline__ void load_row(const float* __restrict__ p, float (&d)[VEC]) {
    using V = typename vec_of<VEC>::type;
    V v = *reinterpret_cast<const V*>(p);
    if constexpr (VEC == 1) { d[0] = v; }
    if constexpr (VEC == 2) { d[0] = v.x; d[1] = v.y; }
    if constexpr (VEC == 4) { d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; }
}
template <int VEC>
__device__ __forceinline__ void store_row(float* __restrict__ p, const float (&d)[VEC], float s) {
    using V = typename vec_of<VEC>::type;
    V v;
    if constexpr (VEC == 1) { v = d[0] * s; }
    if constexpr (VEC == 2) { v.x = d[0] * s; v.y = d[1] * s; }
    if constexpr (VEC == 4) { v.x = d[0] * s; v.y = d[1] * s; v.z = d[2] * s; v.w = d[3] * s; }
    *reinterpret_cast<V*>(p) = v;
}

// rows in flight per wavefront: ~32 VGPRs of outstanding loads
template <int VEC, int NCH> struct inflight { static constexpr int value = (32 / (VEC * NCH)) > 8 ? 8 : ((32 / (VEC * NCH)) < 2 ? 2 : (32 / (VEC * NCH))); };

template <int VEC, int NCH, bool WEIGHTED, bool MEAN, bool EXACT>
__global__ void __launch_bounds__(SEG_THREADS)
segsum_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
              const int32_t* __restrict__ item_row, const float* __restrict__ w,
              int N, int n_items, const float* __restrict__ x, int64_t ldx,
              float* __restrict__ out, int64_t ldo, int F, float* __restrict__ carry,
              const float* __restrict__ bias) {
    constexpr int U = inflight<VEC, NCH>::value;
    const int lane = lane_id();
    const int item = uniform_i(blockIdx.x * SEG_WAVES + (threadIdx.x >> 6));
    if (item >= n_items) return;
    const int nnz = rowptr[N];
    const int k0 = item * T;
    if (k0 >= nnz) return;
    const int k1 = min(k0 + T, nnz);

    bool act[NCH];
    int foff[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        foff[c] = (c * WAVE + lane) * VEC;
        act[c] = EXACT ? true : (foff[c] < F);
    }

    int r = uniform_i(item_row[item]);
    int row_start = uniform_i(rowptr[r]);
    bool head = row_start < k0;          // row r began in an earlier item
    // window of upcoming row ends: lane l holds rowptr[r + 1 + l]
    int wbase = r;
    int rend_v = rowptr[min(wbase + 1 + lane, N)];
    int ri = 0;
    int row_end = bcast_i(rend_v, 0);

    float acc[NCH][VEC];
    float bv[NCH][VEC];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            acc[c][v] = 0.f;
            bv[c][v] = (bias && act[c]) ? bias[foff[c] + v] : 0.f;
        }

    auto write_row = [&](float* __restrict__ dst, float s) {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            if (act[c]) store_row<VEC>(dst + foff[c], acc[c], s);
    };
    auto write_out = [&](float* __restrict__ dst, float s) {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            if (act[c]) {
                float t[VEC];
#pragma unroll
                for (int v = 0; v < VEC; ++v) t[v] = fmaf(acc[c][v], s, bv[c][v]);
                store_row<VEC>(dst + foff[c], t, 1.f);
            }
    };
    // row r is complete (its last entry has been accumulated, or it is empty)
    auto close_row = [&]() {
        if (head) {
            write_row(carry + ((int64_t)item * 2 + 0) * F, 1.f);
            head = false;
        } else {
            float s = 1.f;
            if (MEAN) s = 1.f / (float)max(row_end - row_start, 1);
            write_out(out + (int64_t)r * ldo, s);
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[c][v] = 0.f;
        ++r;
        row_start = row_end;
        if (++ri == WAVE) {
            wbase = r;
            rend_v = rowptr[min(wbase + 1 + lane, N)];
            ri = 0;
        }
        row_end = bcast_i(rend_v, ri);
    };

    for (int kb = k0; kb < k1; kb += WAVE) {
        const int nb = min(WAVE, k1 - kb);
        const int cv = (lane < nb) ? col[kb + lane] : 0;
        float wv = 1.f;
        if (WEIGHTED) wv = (lane < nb) ? w[kb + lane] : 0.f;
        int j = 0;
        for (; j + U <= nb; j += U) {
            float v[U][NCH][VEC];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float* src = x + (int64_t)bcast_i(cv, j + u) * ldx;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if (act[c]) load_row<VEC>(src + foff[c], v[u][c]);
                    else {
#pragma unroll
                        for (int q = 0; q < VEC; ++q) v[u][c][q] = 0.f;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = kb + j + u;
                while (k == row_end) close_row();
                const float ws = WEIGHTED ? bcast_f(wv, j + u) : 1.f;
#pragma unroll
                for (int c = 0; c < NCH; ++c)
#pragma unroll
                    for (int q = 0; q < VEC; ++q)
                        acc[c][q] = WEIGHTED ? fmaf(ws, v[u][c][q], acc[c][q]) : (acc[c][q] + v[u][c][q]);
            }
        }
        for (; j < nb; ++j) {       // ragged tail of the last item only
            float v[NCH][VEC];
            const float* src = x + (int64_t)bcast_i(cv, j) * ldx;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (act[c]) load_row<VEC>(src + foff[c], v[c]);
                else {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) v[c][q] = 0.f;
                }
            }
            const int k = kb + j;
            while (k == row_end) close_row();
            const float ws = WEIGHTED ? bcast_f(wv, j) : 1.f;
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int q = 0; q < VEC; ++q)
                    acc[c][q] = WEIGHTED ? fmaf(ws, v[c][q], acc[c][q]) : (acc[c][q] + v[c][q]);
        }
    }
    // entries exhausted at k1
    if (row_end == k1) {
        close_row();                                     // row r ends exactly here
        while (r < N && row_end == k1) close_row();      // empty rows behind it (out = 0)
    } else if (head) {
        write_row(carry + ((int64_t)item * 2 + 0) * F, 1.f);   // one row spans the whole item
    } else {
        write_row(carry + ((int64_t)item * 2 + 1) * F, 1.f);   // row continues in the next item
    }
}

// The item holding a cut row's FIRST entry sums that row's partials: one 4-wave workgroup per item.
// Short chains (the common case: tail of item i + head of item i+1) are summed by wave 0; a long
// chain (hub row: ~1,600 partials at C4) is split into 4 contiguous slices, one per wave, 8 loads
// in flight each, and the slice sums are added in slice order -- a fixed order, so still bitwise
// reproducible.
constexpr int FIX_COOP_MIN = 16;     // chain length from which all 4 waves cooperate
constexpr int FIX_U = 8;

template <int VEC, int NCH, bool MEAN, bool EXACT>
__global__ void __launch_bounds__(SEG_THREADS)
segsum_fixup_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ item_row,
                    int N, int n_items, float* __restrict__ out, int64_t ldo, int F,
                    const float* __restrict__ carry, const float* __restrict__ bias) {
    __shared__ float red[SEG_WAVES][NCH * VEC * WAVE];
    const int lane = lane_id();
    const int wave = uniform_i(threadIdx.x >> 6);
    const int item = blockIdx.x;
    const int nnz = rowptr[N];
    const int k0 = item * T;
    const int k1 = k0 + T;
    if (k1 >= nnz) return;                               // last item: nothing continues
    const int r = uniform_i(item_row[item + 1]);         // row holding entry k1
    const int rs = uniform_i(rowptr[r]);
    if (rs >= k1 || rs < k0) return;                     // not cut here / owned by an earlier item
    const int re = uniform_i(rowptr[r + 1]);
    const int last = (re - 1) / T;
    const int L = last - item;                           // head partials to add (>= 1)
    const bool coop = L >= FIX_COOP_MIN;                 // workgroup-uniform
    if (!coop && wave != 0) return;
    const int per = coop ? (L + SEG_WAVES - 1) / SEG_WAVES : L;
    const int jb = item + 1 + wave * per;
    const int je = min(jb + per, last + 1);

    bool act[NCH];
    int foff[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        foff[c] = (c * WAVE + lane) * VEC;
        act[c] = EXACT ? true : (foff[c] < F);
    }
    float acc[NCH][VEC];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        // wave 0 starts from the owner's tail partial, the other slices from zero
        if (act[c] && wave == 0) load_row<VEC>(carry + ((int64_t)item * 2 + 1) * F + foff[c], acc[c]);
        else {
#pragma unroll
            for (int q = 0; q < VEC; ++q) acc[c][q] = 0.f;
        }
    }
    int j = jb;
    for (; j + FIX_U <= je; j += FIX_U) {
        float v[FIX_U][NCH][VEC];
#pragma unroll
        for (int u = 0; u < FIX_U; ++u) {
            const float* src = carry + ((int64_t)(j + u) * 2 + 0) * F;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (act[c]) load_row<VEC>(src + foff[c], v[u][c]);
                else {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) v[u][c][q] = 0.f;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < FIX_U; ++u)
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int q = 0; q < VEC; ++q) acc[c][q] += v[u][c][q];
    }
    for (; j < je; ++j) {
        const float* src = carry + ((int64_t)j * 2 + 0) * F;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            float v[VEC];
            if (act[c]) {
                load_row<VEC>(src + foff[c], v);
#pragma unroll
                for (int q = 0; q < VEC; ++q) acc[c][q] += v[q];
            }
        }
    }
    if (coop) {
        if (wave != 0) {
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int q = 0; q < VEC; ++q) red[wave][(c * VEC + q) * WAVE + lane] = acc[c][q];
        }
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int w = 1; w < SEG_WAVES; ++w)
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int q = 0; q < VEC; ++q) acc[c][q] += red[w][(c * VEC + q) * WAVE + lane];
    }
    float s = 1.f;
    if (MEAN) s = 1.f / (float)max(re - rs, 1);
#pragma unroll
    for (int c = 0; c < NCH; ++c)
        if (act[c]) {
            float t[VEC];
#pragma unroll
            for (int q = 0; q < VEC; ++q) t[q] = fmaf(acc[c][q], s, bias ? bias[foff[c] + q] : 0.f);
            store_row<VEC>(out + (int64_t)r * ldo + foff[c], t, 1.f);
        }
}

template <int VEC, int NCH, bool EXACT>
static int launch_segsum(const int32_t* rowptr, const int32_t* col, const int32_t* item_row,
                         const float* w, int N, int n_items, const float* x, int64_t ldx, float* out,
                         int64_t ldo, int F, int mean, float* carry, const float* bias, hipStream_t stream) {
    dim3 grid((unsigned)ceil_div(n_items, SEG_WAVES)), block(SEG_THREADS);
#define NPI_SEG_LAUNCH(WT, MN)                                                                         \
    segsum_kernel<VEC, NCH, WT, MN, EXACT><<<grid, block, 0, stream>>>(rowptr, col, item_row, w, N,    \
                                                                       n_items, x, ldx, out, ldo, F, carry, bias)
    if (w) { if (mean) NPI_SEG_LAUNCH(true, true); else NPI_SEG_LAUNCH(true, false); }
    else   { if (mean) NPI_SEG_LAUNCH(false, true); else NPI_SEG_LAUNCH(false, false); }
#undef NPI_SEG_LAUNCH
    dim3 fgrid((unsigned)n_items);      // one workgroup per item
    if (mean) segsum_fixup_kernel<VEC, NCH, true, EXACT><<<fgrid, block, 0, stream>>>(rowptr, item_row, N, n_items, out, ldo, F, carry, bias);
    else      segsum_fixup_kernel<VEC, NCH, false, EXACT><<<fgrid, block, 0, stream>>>(rowptr, item_row, N, n_items, out, ldo, F, carry, bias);
    return check_launch("npi_segsum");
}

template <int VEC>
static int dispatch_nch(const int32_t* rowptr, const int32_t* col, const int32_t* item_row,
                        const float* w, int N, int n_items, const float* x, int64_t ldx, float* out,
                        int64_t ldo, int F, int mean, float* carry, const float* bias, hipStream_t stream) {
    const int per = WAVE * VEC;
    const int nch = (int)ceil_div(F, per);
    const bool exact = (F % per) == 0;
#define NPI_SEG_CASE(NC)                                                                                  \
    case NC:                                                                                              \
        return exact ? launch_segsum<VEC, NC, true>(rowptr, col, item_row, w, N, n_items, x, ldx, out,    \
                                                    ldo, F, mean, carry, bias, stream)                          \
                     : launch_segsum<VEC, NC, false>(rowptr, col, item_row, w, N, n_items, x, ldx, out,   \
                                                     ldo, F, mean, carry, bias, stream)
    switch (nch) {
        NPI_SEG_CASE(1);
        NPI_SEG_CASE(2);
        NPI_SEG_CASE(3);
        NPI_SEG_CASE(4);
        default: break;
    }
#undef NPI_SEG_CASE
    set_error("npi_segsum: feature width %d needs %d chunks (max 4)", F, nch);
    return NPI_ERR_ARG;
}

}  // namespace npi

using namespace npi;

extern "C" int64_t npi_segsum_carry_elems(int64_t nnz_max, int64_t F) {
    int64_t items = npi_num_items(nnz_max);
    return items > 0 ? 2 * items * F : 1;
}

extern "C" int npi_segsum(const int32_t* rowptr, const int32_t* col, const int32_t* item_row,
                          const float* w, int64_t N, int64_t nnz_max, const void* x_, int64_t ldx,
                          void* out_, int64_t ldo, int64_t F, int dtype, int mean, const float* bias,
                          float* carry, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(N >= 0 && nnz_max >= 0 && F > 0, "npi_segsum: bad size");
    NPI_REQUIRE(dtype == NPI_F32, "npi_segsum: only f32 features are implemented");
    NPI_REQUIRE(ldx >= F && ldo >= F, "npi_segsum: leading dimension < F");
    if (N == 0) return NPI_OK;
    NPI_REQUIRE(rowptr && item_row && x_ && out_ && carry, "npi_segsum: null pointer");
    const float* x = (const float*)x_;
    float* out = (float*)out_;
    const int64_t n_items = npi_num_items(nnz_max);
    if (n_items == 0) {     // no entries at all: every row is empty
        NPI_REQUIRE(bias == nullptr, "npi_segsum: bias with an entry-free graph is not supported");
        (void)hipMemset2DAsync(out, ldo * sizeof(float), 0, F * sizeof(float), N, stream);
        return check_launch("npi_segsum(memset)");
    }
    NPI_REQUIRE(col != nullptr, "npi_segsum: null col");
    // widest vector the row pitch and base alignment allow
    auto aligned = [&](int v) {
        return (F % v == 0) && (ldx % v == 0) && (ldo % v == 0) &&
               (((uintptr_t)x % (4 * v)) == 0) && (((uintptr_t)out % (4 * v)) == 0) &&
               (((uintptr_t)carry % (4 * v)) == 0);
    };
    const int vec = aligned(4) ? 4 : (aligned(2) ? 2 : 1);
    // feature columns handled per launch: 4 chunks of 64 lanes x vec
    const int64_t span = (int64_t)4 * WAVE * vec;
    int rc = NPI_OK;
    for (int64_t f0 = 0; f0 < F && rc == NPI_OK; f0 += span) {
        const int Fc = (int)((F - f0 < span) ? (F - f0) : span);
        // carry rows are Fc wide for this column block
        if (vec == 4) rc = dispatch_nch<4>(rowptr, col, item_row, w, (int)N, (int)n_items, x + f0, ldx, out + f0, ldo, Fc, mean, carry, bias ? bias + f0 : nullptr, stream);
        else if (vec == 2) rc = dispatch_nch<2>(rowptr, col, item_row, w, (int)N, (int)n_items, x + f0, ldx, out + f0, ldo, Fc, mean, carry, bias ? bias + f0 : nullptr, stream);
        else rc = dispatch_nch<1>(rowptr, col, item_row, w, (int)N, (int)n_items, x + f0, ldx, out + f0, ldo, Fc, mean, carry, bias ? bias + f0 : nullptr, stream);
    }
    return rc;
}
