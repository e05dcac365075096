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
//
// The per-entry weight comes in four flavours (WMODE):
//   W_NONE     1                                   SAGEConv (mean or sum)
//   W_ARRAY    w[p]                                GCNConv norm, any per-entry weight
//   W_GAT_DST  exp(lrelu(a_dst[row] + a_src[col]) - m[row]), row scale 1/(s[row] + 1e-16)
//              = GATConv's softmax(alpha) * x_j on the by-target CSR, alpha never stored
//   W_GAT_DST_PRE  W_GAT_DST, one head, with the entry's score read back from the statistics pass (w[p]) instead of gathered
//   W_GAT_SRC_PRE  W_GAT_SRC with alpha read back (w[wmap[p]], one head) instead of recomputed: the per-entry
//              exp / divide of W_GAT_SRC costs more VALU time than the row it weighs costs memory time
//   W_GAT_SRC_FUSED[_H2/4/8]  by-source aggregation that also computes the per-entry score gradient dz from the rows it
//              gathers anyway (the SDDMM of the GATConv backward: no separate gather pass over the 104M entries); alpha is
//              recomputed by the lane that owns the entry from ONE 16-byte gather of packed per-target scalars (tpack)
//   W_GAT_SRC  the same alpha seen from the by-source CSR (backward: d h_j = sum_i alpha_ij d out_i),
//              plus the rank-1 terms of the attention-score gradient in the epilogue
#include "segsum.h"
#include <stdlib.h>

namespace npi {

// the gather reads a two-part table (SegParams.x / x2, split): the select is free (2.479 vs 2.480 ms at C4 with it compiled out)
constexpr bool TWO_PART = true;

constexpr int SEG_THREADS = 256;
constexpr int SEG_WAVES = SEG_THREADS / WAVE;

// storage element types: float, or bf16 carried as uint16_t (f32 accumulation either way)
typedef uint16_t bf16_t;
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return __uint_as_float((uint32_t)v << 16); }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) {
    return __builtin_bit_cast(bf16_t, (__bf16)v);          // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
}

template <typename T, int VEC> struct vec_of;
template <> struct vec_of<float, 1> { using type = float; };
template <> struct vec_of<float, 2> { using type = float2; };
template <> struct vec_of<float, 4> { using type = float4; };
template <> struct vec_of<bf16_t, 1> { using type = uint16_t; };
template <> struct vec_of<bf16_t, 2> { using type = uint32_t; };
template <> struct vec_of<bf16_t, 4> { using type = uint2; };

template <int VEC, typename T>
__device__ __forceinline__ void load_row(const T* __restrict__ p, float (&d)[VEC]) {
    using V = typename vec_of<T, VEC>::type;
    V v = *reinterpret_cast<const V*>(p);
    if constexpr (sizeof(T) == 4) {
        if constexpr (VEC == 1) { d[0] = v; }
        if constexpr (VEC == 2) { d[0] = v.x; d[1] = v.y; }
        if constexpr (VEC == 4) { d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; }
    } else {
        if constexpr (VEC == 1) { d[0] = to_f32((bf16_t)v); }
        if constexpr (VEC == 2) { d[0] = __uint_as_float(v << 16); d[1] = __uint_as_float(v & 0xffff0000u); }
        if constexpr (VEC == 4) {
            d[0] = __uint_as_float(v.x << 16); d[1] = __uint_as_float(v.x & 0xffff0000u);
            d[2] = __uint_as_float(v.y << 16); d[3] = __uint_as_float(v.y & 0xffff0000u);
        }
    }
}
template <int VEC, typename T>
__device__ __forceinline__ void store_row(T* __restrict__ p, const float (&d)[VEC]) {
    using V = typename vec_of<T, VEC>::type;
    V v;
    if constexpr (sizeof(T) == 4) {
        if constexpr (VEC == 1) { v = d[0]; }
        if constexpr (VEC == 2) { v.x = d[0]; v.y = d[1]; }
        if constexpr (VEC == 4) { v.x = d[0]; v.y = d[1]; v.z = d[2]; v.w = d[3]; }
    } else {
        if constexpr (VEC == 1) { v = from_f32<bf16_t>(d[0]); }
        if constexpr (VEC == 2) { v = (uint32_t)from_f32<bf16_t>(d[0]) | ((uint32_t)from_f32<bf16_t>(d[1]) << 16); }
        if constexpr (VEC == 4) {
            v.x = (uint32_t)from_f32<bf16_t>(d[0]) | ((uint32_t)from_f32<bf16_t>(d[1]) << 16);
            v.y = (uint32_t)from_f32<bf16_t>(d[2]) | ((uint32_t)from_f32<bf16_t>(d[3]) << 16);
        }
    }
    *reinterpret_cast<V*>(p) = v;
}

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

// rows in flight per wavefront: ~32 VGPRs of outstanding loads
template <int VEC, int NCH> struct inflight { static constexpr int value = (32 / (VEC * NCH)) > 8 ? 8 : ((32 / (VEC * NCH)) < 2 ? 2 : (32 / (VEC * NCH))); };

// lane geometry shared by the main and the fix-up kernel
template <int VEC, int NCH, int WMODE, bool EXACT>
struct Lanes {
    bool act[NCH];
    int foff[NCH];
    int hd[NCH];       // head of this lane's columns in chunk c (GAT modes)
    __device__ __forceinline__ void init(const SegParams& P) {
        const int lane = lane_id();
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            foff[c] = (c * WAVE + lane) * VEC;
            act[c] = EXACT ? true : (foff[c] < P.F);
            hd[c] = (WMODE >= W_GAT_DST && act[c]) ? foff[c] / P.C : 0;
        }
    }
};

// scale, bias and epilogue of a finished row r, then the store
template <typename T, int VEC, int NCH, int WMODE, bool EXACT>
__device__ __forceinline__ void finish_row(const SegParams& P, const Lanes<VEC, NCH, WMODE, EXACT>& L,
                                           const float (&acc)[NCH][VEC], int r, int row_len) {
    T* __restrict__ dst = reinterpret_cast<T*>(P.out) + (int64_t)r * P.ldo;
    const T* __restrict__ bias = reinterpret_cast<const T*>(P.bias);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (!L.act[c]) continue;
        float sc = 1.f;
        if (P.mean) sc = 1.f / (float)max(row_len, 1);       // scatter_mean's divisor (a launch argument, not a template one)
        if (WMODE == W_GAT_DST || WMODE == W_GAT_DST_PRE) sc = 1.f / (P.s[(int64_t)r * P.H + L.hd[c]] + 1e-16f);
        float t[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            t[q] = fmaf(acc[c][q], sc, bias ? to_f32(bias[L.foff[c] + q]) : 0.f);
            if (P.relu) t[q] = (t[q] < 0.f) ? 0.f : t[q];            // keeps NaN, like torch.relu
        }
        if ((WMODE == W_GAT_SRC || WMODE == W_GAT_SRC_PRE) && P.g_dst != nullptr) {
            const int h = L.hd[c];
            const float gd = P.g_dst[(int64_t)r * P.H + h], gs = P.g_src[(int64_t)r * P.H + h];
            const float* __restrict__ at = P.att + (int64_t)h * 2 * P.C + (L.foff[c] - h * P.C);
#pragma unroll
            for (int q = 0; q < VEC; ++q) t[q] += gd * at[q] + gs * at[P.C + q];
        }
        store_row<VEC, T>(dst + L.foff[c], t);
    }
}

// Workgroup b takes items 4 b .. 4 b + 3, in stream order.  (Dealing the workgroups round-robin over 2 / 4 / 8 contiguous
// parts of the entry stream, so that rows gathering cache-resident hub rows and rows gathering from HBM are in flight
// together, was measured at C4: 2.43-2.46 / 2.56 / 3.02 ms against 2.41-2.45 -- the phases do not overlap; not kept.)
static unsigned seg_grid(int n_items) { return (unsigned)ceil_div(n_items, SEG_WAVES); }
__device__ __forceinline__ int item_block(int b, int) { return b; }

template <typename T, int VEC, int NCH, int WMODE, bool EXACT>
__global__ void __launch_bounds__(SEG_THREADS)
segsum_kernel(SegParams P) {
    constexpr int U = inflight<VEC, NCH>::value;
    const int lane = lane_id();
    const int item = uniform_i(item_block(blockIdx.x, gridDim.x) * SEG_WAVES + (threadIdx.x >> 6));
    if (item >= P.n_items) return;
    const int N = P.N;
    const int nnz = P.rowptr[N];
    const int k0 = item * P.item;
    // a CSR with capacity but no entry at all (every edge dropped): item 0 still runs and closes all N empty rows
    if (k0 >= nnz && !(item == 0 && nnz == 0)) return;
    const int k1 = min(k0 + P.item, nnz);
    const int F = P.F;
    const T* __restrict__ xT = reinterpret_cast<const T*>(P.x);
    // second part of the table, biased so that it is indexed by the column id itself (sharded layers: the gathered hub
    // rows and the rank's own rows are two buffers; dist.py)
    const T* __restrict__ x2T = reinterpret_cast<const T*>(reinterpret_cast<uintptr_t>(P.x2) - (uint64_t)P.split * (uint64_t)P.ldx * sizeof(T));
    const int split = P.split;

    Lanes<VEC, NCH, WMODE, EXACT> L;
    L.init(P);

    int r = uniform_i(P.item_row[item]);
    int row_start = uniform_i(P.rowptr[r]);
    bool head = row_start < k0;          // row r began in an earlier item
    // window of upcoming row ends: lane l holds rowptr[r + 1 + l]
    int rend_v = P.rowptr[min(r + 1 + lane, N)];
    int ri = 0;
    int row_end = bcast_i(rend_v, 0);

    float acc[NCH][VEC];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[c][v] = 0.f;

    // W_GAT_SRC_FUSED: per-entry dot products of one 64-entry block are parked here, then all 64 lanes turn them into dz
    constexpr int HH = fused_heads(WMODE);           // heads of the fused GAT backward (1 in every other mode)
    __shared__ float seg_pb[SEG_WAVES][WAVE * HH];
    float* __restrict__ pb = seg_pb[threadIdx.x >> 6];
    float hr[VEC];                       // W_GAT_SRC_FUSED: this lane's columns of the open row's own features (h_j)
#pragma unroll
    for (int q = 0; q < VEC; ++q) hr[q] = 0.f;
    // per-row constants of the GAT weight (per lane: the head of its columns)
    float rs_a[NCH], rs_m[NCH], rs_i[NCH];
    auto open_row = [&]() {
        if (WMODE == W_GAT_DST || WMODE == W_GAT_DST_PRE) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int64_t i = (int64_t)min(r, N - 1) * P.H + L.hd[c];
                rs_a[c] = (WMODE == W_GAT_DST) ? P.a_dst[i] : 0.f;
                rs_m[c] = P.m[i];
                rs_i[c] = (WMODE == W_GAT_DST && P.alpha_out) ? 1.f / (P.s[i] + 1e-16f) : 0.f;
            }
        } else if (is_fused_mode(WMODE)) {
            if (L.act[0]) load_row<VEC, float>(P.hrow + (int64_t)min(r, N - 1) * P.ldh + L.foff[0], hr);
        } else if (WMODE == W_GAT_SRC) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                rs_a[c] = P.a_src[(int64_t)min(r, N - 1) * P.H + L.hd[c]];
                rs_m[c] = 0.f;
            }
        }
    };
    open_row();

    auto write_carry = [&](int slot) {
        float* __restrict__ dst = P.carry + ((int64_t)item * 2 + slot) * F;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            if (L.act[c]) store_row<VEC, float>(dst + L.foff[c], acc[c]);
    };
    // row r is complete (its last entry has been accumulated, or it is empty)
    auto close_row = [&]() {
        if (head) {
            write_carry(0);
            head = false;
        } else {
            finish_row<T, VEC, NCH, WMODE, EXACT>(P, L, acc, r, row_end - row_start);
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[c][v] = 0.f;
        ++r;
        row_start = row_end;
        if (++ri == WAVE) {
            rend_v = P.rowptr[min(r + 1 + lane, N)];
            ri = 0;
        }
        row_end = bcast_i(rend_v, ri);
        if (WMODE >= W_GAT_DST) open_row();
    };

    // one gathered row (+ its weight) into the accumulators
    auto entry_weight = [&](int c, float g0, float g1, float g2, float ws) -> float {
        if (WMODE == W_ARRAY || WMODE == W_GAT_SRC_PRE || is_fused_mode(WMODE)) return ws;
        if (WMODE == W_GAT_DST_PRE) return expf(ws - rs_m[c]);          // ws = the entry's score, computed by the statistics pass
        if (WMODE == W_GAT_DST) return expf(lrelu(rs_a[c] + g0, P.slope) - rs_m[c]);
        if (WMODE == W_GAT_SRC) return expf(lrelu(g0 + rs_a[c], P.slope) - g1) * g2;
        return 1.f;
    };

    for (int kb = k0; kb < k1; kb += WAVE) {
        const int nb = min(WAVE, k1 - kb);
        const int cv = (lane < nb) ? P.col[kb + lane] : 0;
        float wv = 1.f;
        if (WMODE == W_ARRAY || WMODE == W_GAT_DST_PRE) wv = (lane < nb) ? P.w[kb + lane] : 0.f;
        float dz_d = 0.f, dz_g = 0.f;    // W_GAT_SRC_FUSED, packed: D of the entry's target and leaky_relu' of its score
        // several heads: lane l holds alpha / D / leaky_relu' of entry kb + l for EVERY head; the lane's own head picks its weight
        float wvh[HH], dzdh[HH], dzgh[HH];
        const int myhead = L.hd[0];
        const int lph = (HH > 1) ? (P.C >> 2) : WAVE;            // lanes per head: a power of two >= 8 (the entry point checks)
        if constexpr (HH > 1) {
            const int rl = (lane < nb) ? P.rowidx[kb + lane] : 0;
#pragma unroll
            for (int h = 0; h < HH; ++h) {
                wvh[h] = dzdh[h] = dzgh[h] = 0.f;
                if (lane < nb) {
                    const float4 t = P.tpack[(int64_t)cv * HH + h];
                    const float z = t.x + P.a_src[(int64_t)rl * HH + h];
                    wvh[h] = expf(lrelu(z, P.slope) - t.y) * t.z;
                    dzdh[h] = t.w;
                    dzgh[h] = z > 0.f ? 1.f : P.slope;
                }
            }
        } else if (WMODE == W_GAT_SRC_FUSED) {
            // alpha of entry kb + l is computed BY LANE l (one exp per entry, not per lane) from the packed target scalars
            if (lane < nb) {
                const float4 t = P.tpack[cv];
                const float z = t.x + P.a_src[P.rowidx[kb + lane]];
                wv = expf(lrelu(z, P.slope) - t.y) * t.z;
                dz_d = t.w;
                dz_g = z > 0.f ? 1.f : P.slope;
            } else {
                wv = 0.f;
            }
        } else if (WMODE == W_GAT_SRC_PRE) wv = (lane < nb) ? P.w[P.wmap[kb + lane]] : 0.f;
        (void)wvh; (void)dzdh; (void)dzgh; (void)myhead; (void)lph;
        int avec = 0;                    // W_GAT_DST + alpha_out: alpha of entry kb + l collects in lane l, stored once per block
        const bool keep_alpha = WMODE == W_GAT_DST && P.alpha_out != nullptr;
        int j = 0;
        for (; j + U <= nb; j += U) {
            float v[U][NCH][VEC];
            float g0[U][NCH], g1[U][NCH], g2[U][NCH];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int cu = bcast_i(cv, j + u);
                const T* src = ((TWO_PART && cu >= split) ? x2T : xT) + (int64_t)cu * P.ldx;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if (L.act[c]) load_row<VEC, T>(src + L.foff[c], v[u][c]);
                    else {
#pragma unroll
                        for (int q = 0; q < VEC; ++q) v[u][c][q] = 0.f;
                    }
                    g0[u][c] = g1[u][c] = g2[u][c] = 0.f;
                    if (WMODE == W_GAT_DST) g0[u][c] = P.a_src[(int64_t)cu * P.H + L.hd[c]];
                    if (WMODE == W_GAT_SRC) {
                        const int64_t i = (int64_t)cu * P.H + L.hd[c];
                        g0[u][c] = P.a_dst[i];
                        g1[u][c] = P.m[i];
                        g2[u][c] = 1.f / (P.s[i] + 1e-16f);
                    }
                }
            }
            float pd[U];                 // W_GAT_SRC_FUSED: this lane's share of <gathered row, open row's own features>
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = kb + j + u;
                while (k == row_end) close_row();
                float ws = (WMODE == W_ARRAY || WMODE == W_GAT_SRC_PRE || WMODE == W_GAT_SRC_FUSED || WMODE == W_GAT_DST_PRE) ? bcast_f(wv, j + u) : 1.f;
                if constexpr (HH > 1) {                 // alpha of entry j + u for this lane's head (HH scalar reads, HH - 1 selects)
                    ws = bcast_f(wvh[0], j + u);
#pragma unroll
                    for (int h = 1; h < HH; ++h) {
                        const float t = bcast_f(wvh[h], j + u);
                        ws = (myhead == h) ? t : ws;
                    }
                }
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const float we = entry_weight(c, g0[u][c], g1[u][c], g2[u][c], ws);
                    if (keep_alpha && c == 0)       // one head: every lane holds the same weight
                        avec = (lane == j + u) ? __float_as_int(we * rs_i[0]) : avec;
#pragma unroll
                    for (int q = 0; q < VEC; ++q)
                        acc[c][q] = (WMODE == W_NONE) ? (acc[c][q] + v[u][c][q]) : fmaf(we, v[u][c][q], acc[c][q]);
                }
                pd[u] = 0.f;
                if (is_fused_mode(WMODE)) {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) pd[u] = fmaf(v[u][0][q], hr[q], pd[u]);
                }
            }
            if constexpr (HH > 1 && U == 8) {
                // the same reduce-scatter INSIDE every head's group of lph lanes: the three top bits of the position in the
                // group halve the 8 entries (xor lph/2, lph/4, lph/8), the remaining bits are plain sums
                const bool bh = (lane & (lph >> 1)) != 0, bm = (lane & (lph >> 2)) != 0, bl = (lane & (lph >> 3)) != 0;
                float w4[4], w2[2];
#pragma unroll
                for (int k = 0; k < 4; ++k) w4[k] = (bh ? pd[(k + 4) % U] : pd[k % U]) + __shfl_xor(bh ? pd[k % U] : pd[(k + 4) % U], lph >> 1, WAVE);
#pragma unroll
                for (int k = 0; k < 2; ++k) w2[k] = (bm ? w4[k + 2] : w4[k]) + __shfl_xor(bm ? w4[k] : w4[k + 2], lph >> 2, WAVE);
                float y = (bl ? w2[1] : w2[0]) + __shfl_xor(bl ? w2[0] : w2[1], lph >> 3, WAVE);
                for (int d = lph >> 4; d > 0; d >>= 1) y += __shfl_xor(y, d, WAVE);
                if ((lane & ((lph >> 3) - 1)) == 0 && L.act[0])
                    pb[(j + (bh ? 4 : 0) + (bm ? 2 : 0) + (bl ? 1 : 0)) * HH + myhead] = y;
            }
            if (WMODE == W_GAT_SRC_FUSED && U == 8) {
                // the 8 partial dots are reduce-SCATTERED: every xor step halves the entries a lane still carries
                // (4 + 2 + 1 + 3 = 10 cross-lane steps), after which lane group l >> 3 holds the sum of entry (l >> 3)
                const bool b5 = (lane & 32) != 0, b4 = (lane & 16) != 0, b3 = (lane & 8) != 0;
                float w4[4], w2[2];
#pragma unroll
                for (int k = 0; k < 4; ++k) w4[k] = (b5 ? pd[(k + 4) % U] : pd[k % U]) + __shfl_xor(b5 ? pd[k % U] : pd[(k + 4) % U], 32, WAVE);
#pragma unroll
                for (int k = 0; k < 2; ++k) w2[k] = (b4 ? w4[k + 2] : w4[k]) + __shfl_xor(b4 ? w4[k] : w4[k + 2], 16, WAVE);
                float y = (b3 ? w2[1] : w2[0]) + __shfl_xor(b3 ? w2[0] : w2[1], 8, WAVE);
                y += __shfl_xor(y, 4, WAVE);
                y += __shfl_xor(y, 2, WAVE);
                y += __shfl_xor(y, 1, WAVE);
                if ((lane & 7) == 0) pb[j + (b5 ? 4 : 0) + (b4 ? 2 : 0) + (b3 ? 1 : 0)] = y;
            }
        }
        for (; j < nb; ++j) {       // ragged tail of the last item only
            float v[NCH][VEC];
            float g0[NCH], g1[NCH], g2[NCH];
            const int cu = bcast_i(cv, j);
            const T* src = ((TWO_PART && cu >= split) ? x2T : xT) + (int64_t)cu * P.ldx;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (L.act[c]) load_row<VEC, T>(src + L.foff[c], v[c]);
                else {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) v[c][q] = 0.f;
                }
                g0[c] = g1[c] = g2[c] = 0.f;
                if (WMODE == W_GAT_DST) g0[c] = P.a_src[(int64_t)cu * P.H + L.hd[c]];
                if (WMODE == W_GAT_SRC) {
                    const int64_t i = (int64_t)cu * P.H + L.hd[c];
                    g0[c] = P.a_dst[i];
                    g1[c] = P.m[i];
                    g2[c] = 1.f / (P.s[i] + 1e-16f);
                }
            }
            const int k = kb + j;
            while (k == row_end) close_row();
            float ws = (WMODE == W_ARRAY || WMODE == W_GAT_SRC_PRE || WMODE == W_GAT_SRC_FUSED || WMODE == W_GAT_DST_PRE) ? bcast_f(wv, j) : 1.f;
            if constexpr (HH > 1) {
                ws = bcast_f(wvh[0], j);
#pragma unroll
                for (int h = 1; h < HH; ++h) {
                    const float t = bcast_f(wvh[h], j);
                    ws = (myhead == h) ? t : ws;
                }
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const float we = entry_weight(c, g0[c], g1[c], g2[c], ws);
                if (keep_alpha && c == 0)
                    avec = (lane == j) ? __float_as_int(we * rs_i[0]) : avec;
#pragma unroll
                for (int q = 0; q < VEC; ++q)
                    acc[c][q] = (WMODE == W_NONE) ? (acc[c][q] + v[c][q]) : fmaf(we, v[c][q], acc[c][q]);
            }
            if constexpr (HH > 1) {
                float p = 0.f;
#pragma unroll
                for (int q = 0; q < VEC; ++q) p = fmaf(v[0][q], hr[q], p);
                for (int off = lph >> 1; off > 0; off >>= 1) p += __shfl_xor(p, off, WAVE);
                if ((lane & (lph - 1)) == 0 && L.act[0]) pb[j * HH + myhead] = p;
            }
            if (WMODE == W_GAT_SRC_FUSED) {
                float p = 0.f;
#pragma unroll
                for (int q = 0; q < VEC; ++q) p = fmaf(v[0][q], hr[q], p);
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) p += __shfl_xor(p, off, WAVE);
                if (lane == 0) pb[j] = p;
            }
        }
        if (keep_alpha && lane < nb) P.alpha_out[kb + lane] = __int_as_float(avec);      // one coalesced store per 64 entries
        if constexpr (HH > 1) {
            __builtin_amdgcn_wave_barrier();
            if (lane < nb) {
#pragma unroll
                for (int h = 0; h < HH; ++h)
                    P.dz_out[(int64_t)(kb + lane) * HH + h] = wvh[h] * (pb[lane * HH + h] - dzdh[h]) * dzgh[h];
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (WMODE == W_GAT_SRC_FUSED) {
            // all 64 lanes turn the block's dots into dz together: dz = alpha (dot - D_i) leaky_relu'(a_dst[i] + a_src[j])
            __builtin_amdgcn_wave_barrier();
            if (lane < nb) P.dz_out[kb + lane] = wv * (pb[lane] - dz_d) * dz_g;
            __builtin_amdgcn_wave_barrier();
        }
    }
    // entries exhausted at k1
    if (row_end == k1) {
        close_row();                                     // row r ends exactly here
        while (r < N && row_end == k1) close_row();      // empty rows behind it (out = bias)
    } else if (head) {
        write_carry(0);                                  // one row spans the whole item
    } else {
        write_carry(1);                                  // row continues in the next item
    }
}

// Narrow rows (F <= 32 VEC: hidden = 128 or 64 in f32, the reference's own model width): one row is only
// half / a quarter of a wave instruction, so G = 2 or 4 ENTRIES are gathered per instruction -- lane group
// g = lane / (64 / G) fetches entry G j + g -- and every group keeps its own partial sum of the open row.
// Entries are still added in entry order (group 0, a possible row close, group 1, ...); at a row close
// the G partials are folded across the lane groups (xor shuffles) and group 0 stores.  Same items, carry
// format and fix-up kernel as the wide path; W_NONE / W_ARRAY only.
template <typename T, int VEC, int G, int WMODE>
__global__ void __launch_bounds__(SEG_THREADS)
segsum_group_kernel(SegParams P) {
    constexpr int LG = WAVE / G;                           // lanes per entry group
    constexpr int U = 8 / G < 2 ? 2 : 8 / G;               // wave instructions in flight: U * G = 8 rows (16 or 32 rows: no gain at
                                                           // 58k edges, -3 % / -40 % at 20M)
    const int lane = lane_id();
    const int item = uniform_i(item_block(blockIdx.x, gridDim.x) * SEG_WAVES + (threadIdx.x >> 6));
    if (item >= P.n_items) return;
    const int N = P.N;
    const int nnz = P.rowptr[N];
    const int k0 = item * P.item;
    // a CSR with capacity but no entry at all (every edge dropped): item 0 still runs and closes all N empty rows
    if (k0 >= nnz && !(item == 0 && nnz == 0)) return;
    const int k1 = min(k0 + P.item, nnz);
    const int F = P.F;
    const T* __restrict__ xT = reinterpret_cast<const T*>(P.x);
    const T* __restrict__ x2T = reinterpret_cast<const T*>(reinterpret_cast<uintptr_t>(P.x2) - (uint64_t)P.split * (uint64_t)P.ldx * sizeof(T));
    const int split = P.split;
    const int grp = lane / LG;
    const int foff = (lane % LG) * VEC;
    const bool act = foff < F;

    int r = uniform_i(P.item_row[item]);
    int row_start = uniform_i(P.rowptr[r]);
    bool head = row_start < k0;
    int rend_v = P.rowptr[min(r + 1 + lane, N)];
    int ri = 0;
    int row_end = bcast_i(rend_v, 0);

    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;

    auto fold = [&](float (&t)[VEC]) {                     // sum of the G group partials, in every lane
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            float v = acc[q];
#pragma unroll
            for (int off = WAVE / 2; off >= LG; off >>= 1) v += __shfl_xor(v, off, WAVE);
            t[q] = v;
        }
    };
    auto close_row = [&]() {
        float t[VEC];
        fold(t);
        if (head) {
            if (act && grp == 0) store_row<VEC, float>(P.carry + ((int64_t)item * 2 + 0) * F + foff, t);
            head = false;
        } else if (act && grp == 0) {
            const T* __restrict__ bias = reinterpret_cast<const T*>(P.bias);
            const float sc = P.mean ? 1.f / (float)max(row_end - row_start, 1) : 1.f;
#pragma unroll
            for (int q = 0; q < VEC; ++q) t[q] = fmaf(t[q], sc, bias ? to_f32(bias[foff + q]) : 0.f);
            store_row<VEC, T>(reinterpret_cast<T*>(P.out) + (int64_t)r * P.ldo + foff, t);
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
        ++r;
        row_start = row_end;
        if (++ri == WAVE) {
            rend_v = P.rowptr[min(r + 1 + lane, N)];
            ri = 0;
        }
        row_end = bcast_i(rend_v, ri);
    };

    for (int kb = k0; kb < k1; kb += WAVE) {
        const int nb = min(WAVE, k1 - kb);
        const int cv = (lane < nb) ? P.col[kb + lane] : 0;
        float wv = 1.f;
        if (WMODE == W_ARRAY) wv = (lane < nb) ? P.w[kb + lane] : 0.f;
        for (int j = 0; j < nb; j += U * G) {
            float v[U][VEC];
            float we[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = j + u * G + grp;             // this lane group's entry of wave instruction u
                const int cu = __shfl(cv, min(e, nb - 1), WAVE);
                we[u] = (WMODE == W_ARRAY) ? __shfl(wv, min(e, nb - 1), WAVE) : 1.f;
                if (act && e < nb) load_row<VEC, T>(((TWO_PART && cu >= split) ? x2T : xT) + (int64_t)cu * P.ldx + foff, v[u]);
                else {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) v[u][q] = 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int e = j + u * G + g;           // wave-uniform
                    if (e >= nb) break;
                    while (kb + e == row_end) close_row();
                    if (grp == g) {
#pragma unroll
                        for (int q = 0; q < VEC; ++q)
                            acc[q] = (WMODE == W_NONE) ? (acc[q] + v[u][q]) : fmaf(we[u], v[u][q], acc[q]);
                    }
                }
            }
        }
    }
    if (row_end == k1) {
        close_row();
        while (r < N && row_end == k1) close_row();
    } else {
        float t[VEC];
        fold(t);
        if (act && grp == 0) store_row<VEC, float>(P.carry + ((int64_t)item * 2 + (head ? 0 : 1)) * F + foff, t);
    }
}

// Fix-up: the item holding a cut row's FIRST entry (the "owner") sums that row's partials in item order.
// A workgroup takes FIX_SPAN items.  One thread per item finds out whether it owns a cut row and how long its
// chain of partials is (three dependent index loads -- paid once per span, not once per workgroup as with a
// workgroup per item); then each wave walks its share of the span's owners and adds up the SHORT chains (the
// common case: tail of item i + head of item i + 1); finally the whole workgroup takes the span's LONG
// chains (a hub row: ~1,600 partials at C4) one by one: 4 contiguous slices, one per wave with 8 loads in
// flight, slice sums added in slice order.  Every order of addition is fixed by the data, never by
// scheduling => bitwise reproducible.
constexpr int FIX_COOP_MIN = 16;     // chain length from which the whole workgroup cooperates
constexpr int FIX_U = 8;
constexpr int FIX_SPAN = 64;         // items per workgroup (16 / 32: same time -- the hub row's chain sets it: ~40 us at C4)

// (owner?, row, first entry, end entry, chain length) of `item`
__device__ __forceinline__ bool fix_owner(const SegParams& P, int item, int nnz, int& r, int& rs, int& re, int& len) {
    const int k0 = item * P.item, k1 = k0 + P.item;
    if (item >= P.n_items || k1 >= nnz) return false;            // last item: nothing continues
    r = P.item_row[item + 1];                                     // row holding entry k1
    rs = P.rowptr[r];
    if (rs >= k1 || rs < k0) return false;                        // not cut here / owned by an earlier item
    re = P.rowptr[r + 1];
    len = (re - 1) / P.item - item;                                 // head partials to add (>= 1)
    return true;
}

// Waves per fix-up workgroup: the long chain of a hub row (1,600 partial rows at C4) is summed by ONE workgroup, every wave
// a slice with FIX_U row loads in flight -- a latency chain.  With one 256-column chunk per lane the kernel needs < 64 VGPRs, so
// the workgroup is 16 waves (1,024 threads) instead of 4: the hub chain's slices shrink from 400 to 100 partials (72 -> ~25 us
// per launch at C4); wider rows keep 4 waves (their staging registers would not fit 1,024 threads).
template <int NCH> struct fix_waves { static constexpr int value = NCH == 1 ? 16 : SEG_WAVES; };

template <typename T, int VEC, int NCH, int WMODE, bool EXACT>
__global__ void __launch_bounds__(fix_waves<NCH>::value * WAVE)
segsum_fixup_kernel(SegParams P) {
    constexpr int FW = fix_waves<NCH>::value;
    __shared__ int s_row[FIX_SPAN], s_rs[FIX_SPAN], s_re[FIX_SPAN], s_len[FIX_SPAN];
    __shared__ float red[FW][NCH * VEC * WAVE];
    const int lane = lane_id();
    const int wave = uniform_i(threadIdx.x >> 6);
    const int base = blockIdx.x * FIX_SPAN;
    const int N = P.N, F = P.F;
    const int nnz = P.rowptr[N];
    if (threadIdx.x < FIX_SPAN) {
        int r = -1, rs = 0, re = 0, len = 0;
        const bool own = fix_owner(P, base + threadIdx.x, nnz, r, rs, re, len);
        s_row[threadIdx.x] = own ? r : -1;
        s_rs[threadIdx.x] = rs; s_re[threadIdx.x] = re; s_len[threadIdx.x] = own ? len : 0;
    }
    __syncthreads();
    Lanes<VEC, NCH, WMODE, EXACT> L;
    L.init(P);
    // short chains: one wave each
    for (int q = wave; q < FIX_SPAN; q += FW) {
        const int r = s_row[q], len = s_len[q];
        if (r < 0 || len >= FIX_COOP_MIN) continue;              // wave-uniform
        const int item = base + q;
        float acc[NCH][VEC];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (L.act[c]) load_row<VEC, float>(P.carry + ((int64_t)item * 2 + 1) * F + L.foff[c], acc[c]);   // the owner's tail partial
            else {
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[c][v] = 0.f;
            }
        }
        for (int j = 1; j <= len; ++j) {
            const float* src = P.carry + ((int64_t)(item + j) * 2 + 0) * F;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                float v[VEC];
                if (L.act[c]) {
                    load_row<VEC, float>(src + L.foff[c], v);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[c][k] += v[k];
                }
            }
        }
        finish_row<T, VEC, NCH, WMODE, EXACT>(P, L, acc, r, s_re[q] - s_rs[q]);
    }
    // long chains: the whole workgroup, one chain after the other (s_* are read-only from here on)
    for (int q = 0; q < FIX_SPAN; ++q) {
        const int r = s_row[q], len = s_len[q];
        if (r < 0 || len < FIX_COOP_MIN) continue;               // workgroup-uniform
        const int item = base + q;
        const int per = (len + FW - 1) / FW;
        const int jb = item + 1 + wave * per;
        const int je = min(jb + per, item + len + 1);
        float acc[NCH][VEC];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            // wave 0 starts from the owner's tail partial, the other slices from zero
            if (L.act[c] && wave == 0) load_row<VEC, float>(P.carry + ((int64_t)item * 2 + 1) * F + L.foff[c], acc[c]);
            else {
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[c][v] = 0.f;
            }
        }
        int j = jb;
        for (; j + FIX_U <= je; j += FIX_U) {
            float v[FIX_U][NCH][VEC];
#pragma unroll
            for (int u = 0; u < FIX_U; ++u) {
                const float* src = P.carry + ((int64_t)(j + u) * 2 + 0) * F;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if (L.act[c]) load_row<VEC, float>(src + L.foff[c], v[u][c]);
                    else {
#pragma unroll
                        for (int k = 0; k < VEC; ++k) v[u][c][k] = 0.f;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < FIX_U; ++u)
#pragma unroll
                for (int c = 0; c < NCH; ++c)
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[c][k] += v[u][c][k];
        }
        for (; j < je; ++j) {
            const float* src = P.carry + ((int64_t)j * 2 + 0) * F;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                float v[VEC];
                if (L.act[c]) {
                    load_row<VEC, float>(src + L.foff[c], v);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[c][k] += v[k];
                }
            }
        }
        if (wave != 0) {
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int k = 0; k < VEC; ++k) red[wave][(c * VEC + k) * WAVE + lane] = acc[c][k];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int w = 1; w < FW; ++w)
#pragma unroll
                for (int c = 0; c < NCH; ++c)
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[c][k] += red[w][(c * VEC + k) * WAVE + lane];
            finish_row<T, VEC, NCH, WMODE, EXACT>(P, L, acc, r, s_re[q] - s_rs[q]);
        }
        __syncthreads();                                          // red is reused by the next long chain
    }
}

template <typename T, int VEC, int NCH, int WMODE, bool EXACT>
static void launch_fixup(const SegParams& P, hipStream_t stream) {
    segsum_fixup_kernel<T, VEC, NCH, WMODE, EXACT><<<dim3((unsigned)ceil_div(P.n_items, FIX_SPAN)), dim3(fix_waves<NCH>::value * WAVE), 0, stream>>>(P);
}

template <typename T, int VEC, int NCH, int WMODE, bool EXACT>
static void launch_one(const SegParams& P, hipStream_t stream) {
    dim3 grid(seg_grid(P.n_items)), block(SEG_THREADS);
    segsum_kernel<T, VEC, NCH, WMODE, EXACT><<<grid, block, 0, stream>>>(P);
    launch_fixup<T, VEC, NCH, WMODE, EXACT>(P, stream);
}

template <typename T, int VEC, int G, int WMODE>
static void launch_group(const SegParams& P, hipStream_t stream) {
    dim3 grid(seg_grid(P.n_items)), block(SEG_THREADS);
    segsum_group_kernel<T, VEC, G, WMODE><<<grid, block, 0, stream>>>(P);
    launch_fixup<T, VEC, 1, WMODE, false>(P, stream);
}
template <typename T, int VEC, int G>
static int launch_group_modes(const SegParams& P, int wmode, int mean, hipStream_t stream) {
    if (wmode == W_NONE) launch_group<T, VEC, G, W_NONE>(P, stream);
    else                 launch_group<T, VEC, G, W_ARRAY>(P, stream);
    return check_launch("npi_segsum");
}

template <typename T, int VEC, int NCH, bool EXACT>
static int launch_segsum(const SegParams& P, int wmode, int mean, hipStream_t stream) {
    if constexpr (NCH == 1 && !EXACT) {
        // narrow rows: several entries per wave instruction
        if (wmode <= W_ARRAY) {
            if (P.F <= 16 * VEC) return launch_group_modes<T, VEC, 4>(P, wmode, mean, stream);
            if (P.F <= 32 * VEC) return launch_group_modes<T, VEC, 2>(P, wmode, mean, stream);
        }
    }
    if (wmode == W_NONE) launch_one<T, VEC, NCH, W_NONE, EXACT>(P, stream);
    else if (wmode == W_ARRAY) launch_one<T, VEC, NCH, W_ARRAY, EXACT>(P, stream);
    else if (wmode == W_GAT_DST) {
        if constexpr (VEC == 4 && sizeof(T) == 4) launch_one<T, VEC, NCH, W_GAT_DST, EXACT>(P, stream);
    } else if (wmode == W_GAT_DST_PRE) {
        if constexpr (VEC == 4 && sizeof(T) == 4) {
            dim3 grid(seg_grid(P.n_items)), block(SEG_THREADS);
            segsum_kernel<T, VEC, NCH, W_GAT_DST_PRE, EXACT><<<grid, block, 0, stream>>>(P);
            launch_fixup<T, VEC, NCH, W_GAT_DST, EXACT>(P, stream);           // same row epilogue (1 / (s + eps), bias)
        }
    } else if (wmode == W_GAT_SRC_PRE) {
        if constexpr (VEC == 4 && sizeof(T) == 4) launch_one<T, VEC, NCH, W_GAT_SRC_PRE, EXACT>(P, stream);
    } else if (is_fused_mode(wmode)) {
        if constexpr (VEC == 4 && sizeof(T) == 4 && NCH == 1) {
            dim3 grid(seg_grid(P.n_items)), block(SEG_THREADS);
            if (wmode == W_GAT_SRC_FUSED) segsum_kernel<T, VEC, NCH, W_GAT_SRC_FUSED, EXACT><<<grid, block, 0, stream>>>(P);
            else if (wmode == W_GAT_SRC_FUSED_H2) segsum_kernel<T, VEC, NCH, W_GAT_SRC_FUSED_H2, EXACT><<<grid, block, 0, stream>>>(P);
            else if (wmode == W_GAT_SRC_FUSED_H4) segsum_kernel<T, VEC, NCH, W_GAT_SRC_FUSED_H4, EXACT><<<grid, block, 0, stream>>>(P);
            else segsum_kernel<T, VEC, NCH, W_GAT_SRC_FUSED_H8, EXACT><<<grid, block, 0, stream>>>(P);
            launch_fixup<T, VEC, NCH, W_GAT_SRC_PRE, EXACT>(P, stream);       // same row epilogue
        } else {
            set_error("npi_gat_backward_fused: needs heads * out_channels <= 256");
            return NPI_ERR_ARG;
        }
    } else {
        if constexpr (VEC == 4 && sizeof(T) == 4) launch_one<T, VEC, NCH, W_GAT_SRC, EXACT>(P, stream);
    }
    return check_launch("npi_segsum");
}

template <typename T, int VEC>
static int dispatch_nch(const SegParams& P, int wmode, int mean, hipStream_t stream) {
    const int per = WAVE * VEC;
    const int nch = (int)ceil_div(P.F, per);
    // the unguarded (EXACT) variant exists for 16-byte lanes only -- hidden = 256 / 512 / 768 / 1024, the widths the HBM
    // roofline is quoted on; narrower vectors (odd widths, unaligned rows) always take the guarded kernel
    const bool exact = VEC == 4 && (P.F % per) == 0;
#define NPI_SEG_CASE(NC)                                                                    \
    case NC:                                                                                \
        if constexpr (VEC == 4) { if (exact) return launch_segsum<T, VEC, NC, true>(P, wmode, mean, stream); } \
        return launch_segsum<T, VEC, NC, false>(P, wmode, mean, stream)
    switch (nch) {
        NPI_SEG_CASE(1);
        NPI_SEG_CASE(2);
        NPI_SEG_CASE(3);
        NPI_SEG_CASE(4);
        default: break;
    }
#undef NPI_SEG_CASE
    set_error("npi_segsum: feature width %d needs %d chunks (max 4)", P.F, nch);
    return NPI_ERR_ARG;
}

// shared by npi_segsum and npi_gat_aggregate (gat.hip)
int segsum_run(SegParams P, int wmode, int mean, int64_t nnz_max, int dtype, hipStream_t stream) {
    const int64_t F = P.F;
    if (!item_edges_ok(P.item)) {                             // the CSR's own item size, handed over by the caller next to item_row
        set_error("npi_segsum: item_edges must be 64 or %d (the value the CSR was built with)", NPI_ITEM_EDGES);
        return NPI_ERR_ARG;
    }
    P.n_items = (int)num_items_of(nnz_max, P.item);
    P.mean = mean ? 1 : 0;
    const int es = (dtype == NPI_BF16) ? 2 : 4;              // bytes per stored element
    if (P.x2 == nullptr) { P.x2 = P.x; P.split = 0x7fffffff; }
    const char* x = reinterpret_cast<const char*>(P.x);
    const char* x2 = reinterpret_cast<const char*>(P.x2);
    char* out = reinterpret_cast<char*>(P.out);
    const char* bias = reinterpret_cast<const char*>(P.bias);
    // widest vector the row pitch and base alignment allow
    auto aligned = [&](int v) {
        return (F % v == 0) && (P.ldx % v == 0) && (P.ldo % v == 0) &&
               (((uintptr_t)x % (es * v)) == 0) && (((uintptr_t)x2 % (es * v)) == 0) && (((uintptr_t)out % (es * v)) == 0) &&
               (((uintptr_t)P.carry % (4 * v)) == 0);
    };
    const int vec = aligned(4) ? 4 : (aligned(2) ? 2 : 1);
    if (wmode >= W_GAT_DST) {
        if (dtype != NPI_F32 || vec != 4 || P.C % 4 != 0 || F > 4 * WAVE * 4) {
            set_error("npi_gat_aggregate: needs f32, 16-B aligned rows, out_channels %% 4 == 0 and heads*out_channels <= 1024");
            return NPI_ERR_ARG;
        }
    }
    // feature columns handled per launch: 4 chunks of 64 lanes x vec
    const int64_t span = (int64_t)4 * WAVE * vec;
    int rc = NPI_OK;
    for (int64_t f0 = 0; f0 < F && rc == NPI_OK; f0 += span) {
        SegParams Q = P;
        Q.F = (int)((F - f0 < span) ? (F - f0) : span);       // carry rows are Q.F wide for this column block
        Q.x = reinterpret_cast<const float*>(x + f0 * es);
        Q.x2 = reinterpret_cast<const float*>(x2 + f0 * es);
        Q.out = reinterpret_cast<float*>(out + f0 * es);
        Q.bias = bias ? reinterpret_cast<const float*>(bias + f0 * es) : nullptr;
        if (dtype == NPI_BF16) {
            if (vec == 4) rc = dispatch_nch<bf16_t, 4>(Q, wmode, mean, stream);
            else if (vec == 2) rc = dispatch_nch<bf16_t, 2>(Q, wmode, mean, stream);
            else rc = dispatch_nch<bf16_t, 1>(Q, wmode, mean, stream);
        } else {
            if (vec == 4) rc = dispatch_nch<float, 4>(Q, wmode, mean, stream);
            else if (vec == 2) rc = dispatch_nch<float, 2>(Q, wmode, mean, stream);
            else rc = dispatch_nch<float, 1>(Q, wmode, mean, stream);
        }
    }
    return rc;
}

}  // namespace npi

using namespace npi;

extern "C" int64_t npi_segsum_carry_elems(int64_t nnz_max, int64_t item_edges, int64_t F) {
    if (!item_edges_ok(item_edges) || F <= 0) return -1;
    int64_t items = num_items_of(nnz_max, item_edges);
    return items > 0 ? 2 * items * F : 1;
}

extern "C" int npi_segsum(const int32_t* rowptr, const int32_t* col, const int32_t* item_row, int64_t item_edges,
                          const float* w, int64_t N, int64_t nnz_max, const void* x_, int64_t ldx,
                          void* out_, int64_t ldo, int64_t F, int dtype, int mean, const float* bias,
                          float* carry, void* stream_) {
    return npi_segsum_ex(rowptr, col, item_row, item_edges, w, N, nnz_max, x_, ldx, nullptr, 0, out_, ldo, F, dtype, mean, bias,
                         carry, stream_);
}

extern "C" int npi_segsum_ex(const int32_t* rowptr, const int32_t* col, const int32_t* item_row, int64_t item_edges,
                             const float* w, int64_t N, int64_t nnz_max, const void* x_, int64_t ldx,
                             const void* x2_, int64_t split, void* out_, int64_t ldo, int64_t F, int dtype, int mean,
                             const float* bias, float* carry, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(x2_ == nullptr || (split >= 0 && split < 0x7fffffff), "npi_segsum_ex: bad split");
    NPI_REQUIRE(N >= 0 && nnz_max >= 0 && F > 0, "npi_segsum: bad size");
    NPI_REQUIRE(dtype == NPI_F32 || dtype == NPI_BF16, "npi_segsum: dtype must be NPI_F32 or NPI_BF16");
    NPI_REQUIRE(ldx >= F && ldo >= F, "npi_segsum: leading dimension < F");
    if (N == 0) return NPI_OK;
    NPI_REQUIRE(rowptr && item_row && x_ && out_ && carry, "npi_segsum: null pointer");
    NPI_REQUIRE(item_edges_ok(item_edges), "npi_segsum: item_edges must be 64 or NPI_ITEM_EDGES (the value the CSR was built with)");
    const int64_t n_items = num_items_of(nnz_max, item_edges);
    if (n_items == 0) {     // no entries at all: every row is empty
        NPI_REQUIRE(bias == nullptr, "npi_segsum: bias with an entry-free graph is not supported");
        const size_t es = (dtype == NPI_BF16) ? 2 : 4;
        (void)hipMemset2DAsync(out_, ldo * es, 0, F * es, N, stream);
        return check_launch("npi_segsum(memset)");
    }
    NPI_REQUIRE(col != nullptr, "npi_segsum: null col");
    SegParams P{};
    P.rowptr = rowptr; P.col = col; P.item_row = item_row;
    P.N = (int)N; P.item = (int)item_edges;
    P.x = (const float*)x_; P.ldx = ldx; P.out = (float*)out_; P.ldo = ldo; P.F = (int)F;
    P.x2 = (const float*)x2_; P.split = (int)split;
    P.carry = carry; P.w = w; P.bias = bias;
    P.H = 1; P.C = (int)F;
    return segsum_run(P, w ? W_ARRAY : W_NONE, mean, nnz_max, dtype, stream);
}
