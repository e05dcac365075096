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
// Rows cut by an item boundary are finished INSIDE the launch: partials meet in LDS when the cut lies inside a workgroup
// (4 items), in `carry` otherwise, where the last workgroup to deliver a row's partial adds the chain up in a fixed order
// (see "rows cut by an item boundary" below) => one launch per aggregation, bitwise reproducible.
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
//   W_GAT_DST_FUSED  (round 5; one head, F <= 256) W_GAT_DST_PRE without the statistics pass in front of it: the item computes
//              the scores of its own entries, every row part is weighted against the maximum of ITS entries and carries
//              (max, sum exp) next to its partial row; parts of a cut row are merged with exp(m_part - m_row) where cut rows
//              are resolved, and the row's (m, s) come out beside the output for the backward
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

// A gather in two halves: the raw load (may sit in an exec-masked block of the guarded kernels) and the widening to f32.
// With both in one masked block (round 1-3) hipcc waited vmcnt(0) behind EVERY bf16 load to convert it inside the block --
// 16 serialised round trips per batch of 8 entries (C2's F = 178 aggregation: 45 us against 24 us in f32).
template <typename V> __device__ __forceinline__ V zero_of() { V z; __builtin_memset(&z, 0, sizeof(V)); return z; }
template <int VEC, typename T>
__device__ __forceinline__ typename vec_of<T, VEC>::type load_raw(const T* __restrict__ p) {
    return *reinterpret_cast<const typename vec_of<T, VEC>::type*>(p);
}
template <int VEC, typename T>
__device__ __forceinline__ void unpack_row(const typename vec_of<T, VEC>::type& v, float (&d)[VEC]) {
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
template <int VEC, int NCH, int WMODE, int EXACT>
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

// wave-wide max, the same value in every lane.  DPP, not __shfl_xor: six VALU instructions instead of six ds_bpermute round trips
// (the fused GATConv forward takes one per row it opens: with the shuffles a 256-entry item of 11-entry rows spent ~7 us in them).
// row_shr:1/2/4/8 = an inclusive max-scan inside every row of 16 lanes; row_bcast:15 folds row 0 into row 1 and row 2 into row 3,
// row_bcast:31 rows 0-1 into rows 2-3: lane 63 then holds the maximum of all 64 (every lane must be active: wave-uniform callers).
__device__ __forceinline__ float wave_max(float v) {
    const int ident = __float_as_int(-3.0e38f);
#define NPI_DPP_MAX(CTRL, ROWS) \
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(ident, __float_as_int(v), CTRL, ROWS, 0xf, false)))
    NPI_DPP_MAX(0x111, 0xf);
    NPI_DPP_MAX(0x112, 0xf);
    NPI_DPP_MAX(0x114, 0xf);
    NPI_DPP_MAX(0x118, 0xf);
    NPI_DPP_MAX(0x142, 0xa);
    NPI_DPP_MAX(0x143, 0xc);
#undef NPI_DPP_MAX
    return bcast_f(v, WAVE - 1);
}
// wave-wide sum in the same fixed DPP order (deterministic): the partial dots of the fused GATConv forward's scores
__device__ __forceinline__ float wave_sum(float v) {
#define NPI_DPP_ADD(CTRL, ROWS) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROWS, 0xf, false))
    NPI_DPP_ADD(0x111, 0xf);
    NPI_DPP_ADD(0x112, 0xf);
    NPI_DPP_ADD(0x114, 0xf);
    NPI_DPP_ADD(0x118, 0xf);
    NPI_DPP_ADD(0x142, 0xa);
    NPI_DPP_ADD(0x143, 0xc);
#undef NPI_DPP_ADD
    return bcast_f(v, WAVE - 1);
}

// the same reduction for NON-NEGATIVE values, leaving the result in lane 63 only (no broadcast): invalid source lanes read as zero
// (bound_ctrl), so the DPP needs no identity register -- finish_row runs at the kernel's register cap
__device__ __forceinline__ float wave_max_nonneg_lane63(float v) {
#define NPI_DPP_MAX0(CTRL, ROWS) \
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROWS, 0xf, true)))
    NPI_DPP_MAX0(0x111, 0xf);
    NPI_DPP_MAX0(0x112, 0xf);
    NPI_DPP_MAX0(0x114, 0xf);
    NPI_DPP_MAX0(0x118, 0xf);
    NPI_DPP_MAX0(0x142, 0xa);
    NPI_DPP_MAX0(0x143, 0xc);
#undef NPI_DPP_MAX0
    return v;
}

// scale, bias and epilogue of a finished row r, then the store
// (m_val, s_val: W_GAT_DST_FUSED only -- the row's softmax statistics, complete; every other mode passes zeros)
template <typename T, int VEC, int NCH, int WMODE, int EXACT>
__device__ __forceinline__ void finish_row(const SegParams& P, const Lanes<VEC, NCH, WMODE, EXACT>& L,
                                           const float (&acc)[NCH][VEC], int r, int row_len, float m_val = 0.f, float s_val = 0.f) {
    T* __restrict__ dst = reinterpret_cast<T*>(P.out) + (int64_t)r * P.ldo;
    const T* __restrict__ bias = reinterpret_cast<const T*>(P.bias);
    if constexpr (WMODE == W_GAT_DST_FUSED) {
        if (lane_id() == 0) {                                // an empty row: m = 0, s = 0, as the statistics pass leaves it
            P.m_out[r] = row_len > 0 ? m_val : 0.f;
            P.s_out[r] = row_len > 0 ? s_val : 0.f;
        }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (!L.act[c]) continue;
        float sc = 1.f;
        if (P.mean) sc = 1.f / (float)max(row_len, 1);       // scatter_mean's divisor (a launch argument, not a template one)
        if (WMODE == W_GAT_DST || WMODE == W_GAT_DST_PRE) sc = 1.f / (P.s[(int64_t)r * P.H + L.hd[c]] + 1e-16f);
        if (WMODE == W_GAT_DST_FUSED) sc = 1.f / ((row_len > 0 ? s_val : 0.f) + 1e-16f);
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
        // the row's power-of-two scale for the fp16 x 2 projection behind this aggregation (F = 256: every lane of the wave holds
        // four columns): this lane's maximum now, the wave maximum and one 4-byte store per row BEHIND the row's own store -- the
        // values are dead by then (the kernel runs at its 64-register cap with a batch of gathered rows in flight)
        constexpr bool SCALES = EXACT == 2 && NCH == 1 && VEC == 4 && sizeof(T) == 4;      // (EXACT: 0 guarded, 1 unguarded, 2 unguarded + scales)
        float mloc = 0.f;
        if constexpr (SCALES) {
            if (P.scale_out != nullptr) mloc = fmaxf(fmaxf(fabsf(t[0]), fabsf(t[1])), fmaxf(fabsf(t[2]), fabsf(t[3])));
        }
        if constexpr (VEC == 4 && sizeof(T) == 4) {
            // a large output is written once and read again after gigabytes of gathers: streamed past the caches it does not
            // displace gathered rows (same-box A/B at C4: 2.618 -> 2.592 ms per launch, step 6.80 -> 6.75 ms)
            if (P.nt_out) {
                typedef float v4f __attribute__((ext_vector_type(4)));
                const v4f v = {t[0], t[1], t[2], t[3]};
                __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(dst + L.foff[c]));
                if constexpr (SCALES) {
                    if (P.scale_out != nullptr) {
                        mloc = wave_max_nonneg_lane63(mloc);
                        if (lane_id() == WAVE - 1) P.scale_out[r] = pow2_scale_of(mloc);
                    }
                }
                continue;
            }
        }
        store_row<VEC, T>(dst + L.foff[c], t);
        if constexpr (SCALES) {
            if (P.scale_out != nullptr) {
                mloc = wave_max_nonneg_lane63(mloc);
                if (lane_id() == WAVE - 1) P.scale_out[r] = pow2_scale_of(mloc);
            }
        }
    }
}

// Workgroup b takes items 4 b .. 4 b + 3, in stream order.  (Dealing the workgroups round-robin over 2 / 4 / 8 contiguous
// parts of the entry stream, so that rows gathering cache-resident hub rows and rows gathering from HBM are in flight
// together, was measured at C4: 2.43-2.46 / 2.56 / 3.02 ms against 2.41-2.45 -- the phases do not overlap; not kept.)
static unsigned seg_grid(int n_items) { return (unsigned)ceil_div(n_items, SEG_WAVES); }
__device__ __forceinline__ int item_block(int b, int) { return b; }


// ---- rows cut by an item boundary: resolved INSIDE the launch ------------------------------------------------------------------
// A wavefront that leaves a row unfinished (the row began in an earlier item: "head" partial; the row continues in the next
// item: "tail" partial) parks the f32 partial in LDS.  The wavefront of the workgroup that finishes LAST (an LDS arrival
// counter: no barrier, the other waves have left) walks the workgroup's SEG_WAVES items in order and
//   * finishes every row that lies inside the workgroup (tail of item i + heads of the items behind it, in item order),
//   * is left with at most two partials that concern OTHER workgroups: the head partial of a row that began before the
//     workgroup's first entry and the tail partial of a row that runs past its last one.
// Those go to `carry` in global memory, and the chain of a row's workgroup-level partials is summed by whichever workgroup
// delivers the LAST of them -- an agent-scope arrival counter per row, kept at the row's FIRST workgroup.  A long chain (a hub
// row: 400 workgroup partials at C4, 2,000 at C5) is summed in two levels: the heads that fall into one span of CHAIN_SPAN
// workgroups by the last arriver of that span, the span sums (+ the tail) by the last arriver of the row.  Which workgroup
// does a sum depends on timing; WHAT it adds in which order does not (item order inside a workgroup, workgroup order inside
// a span, span order inside a row): the result is bitwise reproducible.
//
// Hand-off (cdna_hip_programming.md Guideline 16 / MI355X_MICROARCH.md, inter-workgroup visibility): the partial rows are
// stored write-through (sc1), the storing wave drains its stores (s_waitcnt vmcnt(0)), ONE lane signals with an agent-scope
// atomic add whose returned value tells the last arriver; that wave makes one agent-scope acquire and then loads.  Every
// counter is reset by its last arriver, so the words are zero again when the launch ends: `carry` has to be zeroed ONCE, when
// it is allocated (npi_segsum_carry_elems), not per launch.
constexpr int CHAIN_SPAN = 64;       // workgroups per span of the two-level sum; chains up to this length are summed directly
constexpr int CHAIN_U = 8;           // partial rows in flight while a chain is summed

struct ItemMeta {                    // what a wavefront leaves behind for the resolver (LDS)
    int head_row, head_rs, head_re, head_closed;     // head_row < 0: no head partial; closed: the row ended inside the item
    int tail_row, tail_rs, tail_re;                   // tail_row < 0: no tail partial
    float head_m, head_s, tail_m, tail_s;             // W_GAT_DST_FUSED: (max, sum exp(. - max)) of the part's entries
};

struct CarryLayout {                 // global scratch of one launch (f32 words); n_wg = workgroups of the launch
    int64_t n_wg, n_span;
    __host__ __device__ CarryLayout(int64_t n_items) {
        n_wg = (n_items + SEG_WAVES - 1) / SEG_WAVES;
        n_span = (n_wg + CHAIN_SPAN - 1) / CHAIN_SPAN;
    }
    // [counters: n_wg row counters, 2 n_span span counters][pad to 64 words]
    // [ms: (max, sum exp) of every partial row below, 2 floats each: W_GAT_DST_FUSED][pad to 64 words]
    // [2 n_wg rows of F][2 n_span rows of F]
    __host__ __device__ int64_t counters() const { return ((n_wg + 2 * n_span + 63) / 64) * 64; }
    __host__ __device__ int64_t ms_words() const { return ((2 * (2 * n_wg + 2 * n_span) + 63) / 64) * 64; }
    __host__ __device__ int64_t rows_off() const { return counters() + ms_words(); }
    __host__ __device__ int64_t elems(int64_t F) const { return rows_off() + (2 * n_wg + 2 * n_span) * F; }
};

// write-through (sc1) stores of a partial row: hipcc does not count an asm store, the caller drains with drain_stores()
template <int VEC>
__device__ __forceinline__ void store_row_sc1(float* p, const float (&d)[VEC]) {
    if constexpr (VEC == 4) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 v = {d[0], d[1], d[2], d[3]};
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
    } else if constexpr (VEC == 2) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 v = {d[0], d[1]};
        asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
    } else {
        asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(d[0]) : "memory");
    }
}
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// one lane adds 1 to an agent-scope counter; true in every lane of the wave iff this arrival was the `expected`-th.  The last
// arriver resets the counter (nobody else touches it any more in this launch) and makes the acquire its loads need.
__device__ __forceinline__ bool arrive_last(int* cnt, int expected) {
    int old = 0;
    if (lane_id() == 0) old = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    old = uniform_i(old);
    if (old != expected - 1) return false;
    if (lane_id() == 0) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return true;
}

// acc += rows [j0, j1) of `base` (row stride `stride` floats), CHAIN_U rows in flight, in row order
template <int VEC, int NCH, class Geo>
__device__ __forceinline__ void add_rows(const Geo& L, float (&acc)[NCH][VEC], const float* base, int64_t stride, int j0, int j1) {
    int j = j0;
    for (; j + CHAIN_U <= j1; j += CHAIN_U) {
        float v[CHAIN_U][NCH][VEC];
#pragma unroll
        for (int u = 0; u < CHAIN_U; ++u) {
            const float* src = base + (int64_t)(j + u) * stride;
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
        for (int u = 0; u < CHAIN_U; ++u)
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[c][k] += v[u][c][k];
    }
    for (; j < j1; ++j) {
        const float* src = base + (int64_t)j * stride;
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
}

// The resolver wave hands over a workgroup-level partial of row r = entries [rs, re): `slot` 0 = head (the row began before
// this workgroup), 1 = tail (the row began in it).  Whoever delivers the last partial of the row sums the chain and finishes it.
template <int VEC, int NCH, class Geo, class Fin>
__device__ __forceinline__ void emit_partial(const SegParams& P, const Geo& L, const Fin& finish, float (&acc)[NCH][VEC],
                                             int slot, int r, int rs, int re) {
    const int F = P.F;
    const int S = P.item * SEG_WAVES;                                  // entries per workgroup
    const CarryLayout lay(P.n_items);
    int* cnt_row = reinterpret_cast<int*>(P.carry);
    int* cnt_span = cnt_row + lay.n_wg;
    float* rows = P.carry + lay.rows_off();                             // [2 n_wg, F]
    float* span_rows = rows + 2 * lay.n_wg * (int64_t)F;                // [2 n_span, F]
    const int b = blockIdx.x;
    const int fi = rs / S, li = (re - 1) / S, len = li - fi;            // workgroups of the row: fi .. li (len >= 1)
    float* mine = rows + ((int64_t)b * 2 + slot) * F;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
        if (L.act[c]) store_row_sc1<VEC>(mine + L.foff[c], acc[c]);
    drain_stores();
    auto load_tail = [&]() {                                            // the row's first partial: the tail of workgroup fi
        const float* t = rows + ((int64_t)fi * 2 + 1) * F;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (L.act[c]) load_row<VEC, float>(t + L.foff[c], acc[c]);
            else {
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[c][k] = 0.f;
            }
        }
    };
    if (len <= CHAIN_SPAN) {                                            // tail + len heads: summed directly
        if (!arrive_last(cnt_row + fi, len + 1)) return;
        load_tail();
        add_rows<VEC, NCH>(L, acc, rows + (int64_t)(fi + 1) * 2 * F, 2 * (int64_t)F, 0, len);      // heads of fi + 1 .. li
        finish(acc, r, re - rs, 0.f, 0.f);
        return;
    }
    const int g0 = (fi + 1) / CHAIN_SPAN, g1 = li / CHAIN_SPAN;          // spans that hold heads of this row
    if (slot == 0) {
        const int g = b / CHAIN_SPAN;
        const int s = fi >= g * CHAIN_SPAN ? 1 : 0;                      // the row began inside this span / before it
        const int lo = max(fi + 1, g * CHAIN_SPAN), hi = min(li, g * CHAIN_SPAN + CHAIN_SPAN - 1);
        if (!arrive_last(cnt_span + 2 * g + s, hi - lo + 1)) return;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[c][k] = 0.f;
        add_rows<VEC, NCH>(L, acc, rows + (int64_t)lo * 2 * F, 2 * (int64_t)F, 0, hi - lo + 1);
        float* sp = span_rows + ((int64_t)2 * g + s) * F;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            if (L.act[c]) store_row_sc1<VEC>(sp + L.foff[c], acc[c]);
        drain_stores();
    }
    if (!arrive_last(cnt_row + fi, 1 + (g1 - g0 + 1))) return;           // the tail + one arrival per span
    load_tail();
    // span g0 holds the row in its slot 1 when the row began inside it, in slot 0 otherwise; every later span in slot 0
    add_rows<VEC, NCH>(L, acc, span_rows + ((int64_t)2 * g0 + (fi >= g0 * CHAIN_SPAN ? 1 : 0)) * F, 0, 0, 1);
    if (g1 > g0) add_rows<VEC, NCH>(L, acc, span_rows + (int64_t)2 * (g0 + 1) * F, 2 * (int64_t)F, 0, g1 - g0);
    finish(acc, r, re - rs, 0.f, 0.f);
}

// ---- W_GAT_DST_FUSED: the same resolution for partial rows that carry softmax statistics -------------------------------------
// A partial of a row = (acc = sum_p exp(e_p - m) x[col p], m = max_p e_p, s = sum_p exp(e_p - m)) over the part's entries.  Two
// parts merge as  M = max(m1, m2),  acc = acc1 exp(m1 - M) + acc2 exp(m2 - M),  s likewise (the online softmax).  A chain is
// merged against ITS maximum: first the (m, s) pairs of the chain (lane-parallel, a wave-wide max), then the rows in chain
// order, each scaled by exp(m_k - M) -- the same fixed order as the plain sums above, so the result stays bitwise reproducible.
// max of the m of n (m, s) pairs, `stride` floats apart
__device__ __forceinline__ float chain_max(const float* ms, int64_t stride, int n) {
    float m = -3.0e38f;
    for (int j = lane_id(); j < n; j += WAVE) m = fmaxf(m, ms[(int64_t)j * stride]);
    return wave_max(m);
}
// acc += sum_j exp(m_j - M) row_j,  s_acc += sum_j exp(m_j - M) s_j  over rows [0, n) of `base` (row stride `stride` floats),
// their (m, s) pairs `ms_stride` floats apart; in row order, CHAIN_U rows in flight
constexpr int SM_U = 4;             // partial rows in flight while a chain with statistics is merged
template <int VEC, int NCH, class Geo>
__device__ __forceinline__ void add_rows_scaled(const Geo& L, float (&acc)[NCH][VEC], float& s_acc, const float* base, int64_t stride,
                                                const float* ms, int64_t ms_stride, int n, float M) {
    const int lane = lane_id();
    for (int j0 = 0; j0 < n; j0 += WAVE) {
        const int nb = min(WAVE, n - j0);
        float f = 0.f, sv = 0.f;
        if (lane < nb) {
            const float2 t = *reinterpret_cast<const float2*>(ms + (int64_t)(j0 + lane) * ms_stride);
            f = expf(t.x - M);
            sv = t.y * f;
        }
        for (int j = 0; j < nb; j += SM_U) {
            float v[SM_U][NCH][VEC];
#pragma unroll
            for (int u = 0; u < SM_U; ++u) {
                const float* src = base + (int64_t)(j0 + min(j + u, nb - 1)) * stride;       // past the end: the last row again, dropped
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
            for (int u = 0; u < SM_U; ++u) {
                if (j + u < nb) {                                // wave-uniform
                    const float fj = bcast_f(f, j + u);
                    s_acc += bcast_f(sv, j + u);
#pragma unroll
                    for (int c = 0; c < NCH; ++c)
#pragma unroll
                        for (int k = 0; k < VEC; ++k) acc[c][k] = fmaf(fj, v[u][c][k], acc[c][k]);
                }
            }
        }
    }
}

template <int VEC, int NCH, class Geo, class Fin>
__device__ __forceinline__ void emit_partial_sm(const SegParams& P, const Geo& L, const Fin& finish, float (&acc)[NCH][VEC],
                                                int slot, int r, int rs, int re, float pm, float ps) {
    const int F = P.F;
    const int S = P.item * SEG_WAVES;
    const CarryLayout lay(P.n_items);
    int* cnt_row = reinterpret_cast<int*>(P.carry);
    int* cnt_span = cnt_row + lay.n_wg;
    float* msb = P.carry + lay.counters();                              // [2 n_wg + 2 n_span][2]: (m, s) of every partial row
    float* span_ms = msb + 2 * (2 * lay.n_wg);
    float* rows = P.carry + lay.rows_off();
    float* span_rows = rows + 2 * lay.n_wg * (int64_t)F;
    const int b = blockIdx.x;
    const int fi = rs / S, li = (re - 1) / S, len = li - fi;
    float* mine = rows + ((int64_t)b * 2 + slot) * F;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
        if (L.act[c]) store_row_sc1<VEC>(mine + L.foff[c], acc[c]);
    if (lane_id() == 0) {
        const float t2[2] = {pm, ps};
        store_row_sc1<2>(msb + ((int64_t)b * 2 + slot) * 2, t2);
    }
    drain_stores();
    float s_tot = 0.f;
    // the row's first partial (the tail of workgroup fi) scaled into acc / s_tot against the chain's maximum M
    auto take_tail = [&](float M) {
        const float* t = rows + ((int64_t)fi * 2 + 1) * F;
        const float2 tms = *reinterpret_cast<const float2*>(msb + ((int64_t)fi * 2 + 1) * 2);
        const float ft = expf(tms.x - M);
        s_tot = tms.y * ft;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            float v[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) v[k] = 0.f;
            if (L.act[c]) load_row<VEC, float>(t + L.foff[c], v);
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[c][k] = v[k] * ft;
        }
    };
    const float* tail_m = msb + ((int64_t)fi * 2 + 1) * 2;
    if (len <= CHAIN_SPAN) {
        if (!arrive_last(cnt_row + fi, len + 1)) return;
        const float* hms = msb + (int64_t)(fi + 1) * 2 * 2;              // heads of fi + 1 .. li: slot 0 of consecutive workgroups
        const float M = fmaxf(chain_max(hms, 4, len), *tail_m);
        take_tail(M);
        add_rows_scaled<VEC, NCH>(L, acc, s_tot, rows + (int64_t)(fi + 1) * 2 * F, 2 * (int64_t)F, hms, 4, len, M);
        finish(acc, r, re - rs, M, s_tot);
        return;
    }
    const int g0 = (fi + 1) / CHAIN_SPAN, g1 = li / CHAIN_SPAN;
    if (slot == 0) {
        const int g = b / CHAIN_SPAN;
        const int sl = fi >= g * CHAIN_SPAN ? 1 : 0;
        const int lo = max(fi + 1, g * CHAIN_SPAN), hi = min(li, g * CHAIN_SPAN + CHAIN_SPAN - 1);
        if (!arrive_last(cnt_span + 2 * g + sl, hi - lo + 1)) return;
        const float* hms = msb + (int64_t)lo * 2 * 2;
        const float Mg = chain_max(hms, 4, hi - lo + 1);
        float sg = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[c][k] = 0.f;
        add_rows_scaled<VEC, NCH>(L, acc, sg, rows + (int64_t)lo * 2 * F, 2 * (int64_t)F, hms, 4, hi - lo + 1, Mg);
        float* sp = span_rows + ((int64_t)2 * g + sl) * F;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            if (L.act[c]) store_row_sc1<VEC>(sp + L.foff[c], acc[c]);
        if (lane_id() == 0) {
            const float t2[2] = {Mg, sg};
            store_row_sc1<2>(span_ms + ((int64_t)2 * g + sl) * 2, t2);
        }
        drain_stores();
    }
    if (!arrive_last(cnt_row + fi, 1 + (g1 - g0 + 1))) return;
    // span g0 holds the row in its slot 1 when the row began inside it, in slot 0 otherwise; every later span in slot 0
    const int64_t first = (int64_t)2 * g0 + (fi >= g0 * CHAIN_SPAN ? 1 : 0);
    const float* later_ms = span_ms + (int64_t)2 * (g0 + 1) * 2;
    float M = fmaxf(*tail_m, span_ms[first * 2]);
    if (g1 > g0) M = fmaxf(M, chain_max(later_ms, 4, g1 - g0));
    take_tail(M);
    add_rows_scaled<VEC, NCH>(L, acc, s_tot, span_rows + first * F, 0, span_ms + first * 2, 0, 1, M);
    if (g1 > g0) add_rows_scaled<VEC, NCH>(L, acc, s_tot, span_rows + (int64_t)2 * (g0 + 1) * F, 2 * (int64_t)F, later_ms, 4, g1 - g0, M);
    finish(acc, r, re - rs, M, s_tot);
}

// Every wave calls this when its item is done (inactive waves too).  `part`: [SEG_WAVES][2][ROWF] LDS rows, `meta`: [SEG_WAVES],
// both filled in by the waves themselves (lane 0 writes the meta words at the moment they are known).
// SM: the partial rows carry softmax statistics (W_GAT_DST_FUSED): merged with rescaling instead of added
template <int VEC, int NCH, int ROWF, bool SM = false, class Geo, class Fin>
__device__ __forceinline__ void resolve_block(const SegParams& P, const Geo& L, const Fin& finish, float (*part)[2][ROWF],
                                              ItemMeta* meta, int* arrived) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");                // the wave's partial rows and meta (LDS) before its arrival
    int old = 0;
    if (lane_id() == 0) old = __hip_atomic_fetch_add(arrived, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    old = uniform_i(old);
    if (old != SEG_WAVES - 1) return;                                   // not the last wave of the workgroup
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    float acc[NCH][VEC];
    bool open = false, wg_head = false;
    int orow = 0, ors = 0, ore = 0;
    float om = 0.f, os = 0.f;                                           // SM: (max, sum exp) of the open row so far
    auto load_part = [&](int w, int slot, bool add) {
        float f1 = 1.f, f2 = 1.f;
        if constexpr (SM) {
            const float pm = slot ? meta[w].tail_m : meta[w].head_m, ps = slot ? meta[w].tail_s : meta[w].head_s;
            if (add) {
                const float M = fmaxf(om, pm);
                f1 = expf(om - M);
                f2 = expf(pm - M);
                os = os * f1 + ps * f2;
                om = M;
            } else {
                om = pm;
                os = ps;
            }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            float v[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) v[k] = 0.f;
            if (L.act[c]) load_row<VEC, float>(&part[w][slot][L.foff[c]], v);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                if constexpr (SM) acc[c][k] = add ? fmaf(acc[c][k], f1, v[k] * f2) : v[k];
                else acc[c][k] = add ? acc[c][k] + v[k] : v[k];
            }
        }
    };
    auto emit = [&](int slot) {
        if constexpr (SM) emit_partial_sm<VEC, NCH>(P, L, finish, acc, slot, orow, ors, ore, om, os);
        else emit_partial<VEC, NCH>(P, L, finish, acc, slot, orow, ors, ore);
    };
    for (int w = 0; w < SEG_WAVES; ++w) {                               // (wave-uniform control flow: meta is read by every lane)
        const int m_head_row = meta[w].head_row, m_tail_row = meta[w].tail_row;
        if (m_head_row >= 0) {
            if (!open) {                                                // the row began before this workgroup
                load_part(w, 0, false);
                open = true; wg_head = true; orow = m_head_row; ors = meta[w].head_rs;
            } else {
                load_part(w, 0, true);                                  // the open row runs on through this item
            }
            ore = meta[w].head_re;
            if (meta[w].head_closed) {
                if (wg_head) emit(0);
                else finish(acc, orow, ore - ors, om, os);              // began and ended inside the workgroup
                open = false;
            }
        }
        if (m_tail_row >= 0) {
            load_part(w, 1, false);
            open = true; wg_head = false; orow = m_tail_row; ors = meta[w].tail_rs; ore = meta[w].tail_re;
        }
    }
    if (open) {
        // wg_head: the whole workgroup lies inside one row
        emit(wg_head ? 0 : 1);
    }
}

// one item: `part` = this wave's two LDS rows ([2][NCH VEC WAVE]: head partial, tail partial), `M` = what it leaves behind
template <typename T, int VEC, int NCH, int WMODE, int EXACT>
__device__ __forceinline__ void segsum_item(const SegParams& P, const Lanes<VEC, NCH, WMODE, EXACT>& L, const int item,
                                            float* __restrict__ part, ItemMeta* __restrict__ M) {
    constexpr int U = inflight<VEC, NCH>::value;
    constexpr int ROWF = NCH * VEC * WAVE;
    const int lane = lane_id();
    const int N = P.N;
    const int nnz = P.rowptr[N];
    const int k0 = item * P.item;
    // a CSR with capacity but no entry at all (every edge dropped): item 0 still runs and closes all N empty rows
    if (k0 >= nnz && !(item == 0 && nnz == 0)) {
        if (WMODE == W_GAT_SRC_FUSED && P.rowsum_out != nullptr && lane == 0) P.rs_tail_row[item] = -1;
        return;
    }
    const int k1 = min(k0 + P.item, nnz);
    const T* __restrict__ xT = reinterpret_cast<const T*>(P.x);
    // second part of the table, biased so that it is indexed by the column id itself (sharded layers: the gathered hub
    // rows and the rank's own rows are two buffers; dist.py)
    // (a pointer biased by -split rows through integer arithmetic loses its address space: every gather became a FLAT load)
    const T* __restrict__ x2T = reinterpret_cast<const T*>(P.x2);
    const int split = P.split;

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
    // W_GAT_DST_FUSED: an ONLINE softmax.  The score of an entry is e_p = leaky_relu(a_dst[row] + <h_j, att_src>): the source's
    // half is recomputed from the row h_j that is gathered anyway -- this lane's 4 columns against its 4 values of att_src, a DPP
    // sum over the wave -- instead of gathering a_src[col p] (a 4-byte gather per entry costs the memory pipeline a line request
    // of its own, 64 per wave instruction against the 16 of a row gather: the first version of this mode, with the scores
    // gathered and parked in LDS, ran the launch 6-7 % slower and gave back what the statistics pass had cost).  The open row
    // keeps (m_run, s_run) = (max so far, sum of exp(e - m_run)); a new maximum rescales the accumulators (rare after a row's
    // first few entries: a wave-uniform branch).
    constexpr bool SMX = WMODE == W_GAT_DST_FUSED;
    float m_run = -3.0e38f, s_run = 0.f;
    float at_src[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) at_src[q] = 0.f;
    if constexpr (SMX) {
        if (L.act[0]) load_row<VEC, float>(P.att + P.C + L.foff[0], at_src);
    }
    (void)m_run; (void)s_run; (void)at_src;
    float hr[VEC];                       // W_GAT_SRC_FUSED: this lane's columns of the open row's own features (h_j)
#pragma unroll
    for (int q = 0; q < VEC; ++q) hr[q] = 0.f;
    // per-row constants of the GAT weight (per lane: the head of its columns)
    float rs_a[NCH], rs_m[NCH], rs_i[NCH];
    auto open_row = [&]() {
        if constexpr (SMX) {
            m_run = -3.0e38f;
            s_run = 0.f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) { rs_a[c] = P.a_dst[min(r, N - 1)]; rs_m[c] = 0.f; rs_i[c] = 0.f; }
            return;
        }
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

    auto write_carry = [&](int slot) {                                   // to LDS: resolve_block takes it from there
        float* __restrict__ dst = part + slot * ROWF;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            if (L.act[c]) store_row<VEC, float>(dst + L.foff[c], acc[c]);
    };
    // row r is complete (its last entry has been accumulated, or it is empty)
    auto close_row = [&]() {
        if (head) {
            write_carry(0);
            if (lane == 0) {
                M->head_row = r; M->head_rs = row_start; M->head_re = row_end; M->head_closed = 1;
                if constexpr (SMX) { M->head_m = m_run; M->head_s = s_run; }
            }
            head = false;
        } else {
            finish_row<T, VEC, NCH, WMODE, EXACT>(P, L, acc, r, row_end - row_start, m_run, s_run);
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

    // W_GAT_SRC_FUSED with rowsum_out: the row sums of dz.  rs_prev = the row that runs INTO this item (its sum here is a chain
    // link, not a result); (rs_open_key, rs_open_val) = the row of the previous block's last entry and its sum so far
    const bool rs_on = WMODE == W_GAT_SRC_FUSED && P.rowsum_out != nullptr;
    const int rs_prev = head ? r : -1;
    int rs_open_key = -1;
    float rs_open_val = 0.f;
    auto rs_put = [&](int key, float val) {                              // a finished row (or the last link of a chain)
        if (key != rs_prev) P.rowsum_out[key] = val;
        else P.rs_head[item] = val;
    };
    (void)rs_prev; (void)rs_open_key; (void)rs_open_val;
    for (int kb = k0; kb < k1; kb += WAVE) {
        const int nb = min(WAVE, k1 - kb);
        const int cv = (lane < nb) ? P.col[kb + lane] : 0;
        int rlk = 0x7fffffff;            // rs_on: the row of entry kb + lane
        (void)rlk;
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
                rlk = P.rowidx[kb + lane];
                const float z = t.x + P.a_src[rlk];
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
            typename vec_of<T, VEC>::type raw[U][NCH];
            float g0[U][NCH], g1[U][NCH], g2[U][NCH];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int cu = bcast_i(cv, j + u);
                const T* src = (TWO_PART && cu >= split) ? x2T + (int64_t)(cu - split) * P.ldx : xT + (int64_t)cu * P.ldx;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    raw[u][c] = zero_of<typename vec_of<T, VEC>::type>();
                    if (L.act[c]) raw[u][c] = load_raw<VEC, T>(src + L.foff[c]);
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
            if constexpr (SMX) {
                // scores of the U entries from their gathered rows, then the entries in order: row ends, online max, accumulate
                float dt[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    unpack_row<VEC, T>(raw[u][0], v[u][0]);
                    dt[u] = 0.f;
#pragma unroll
                    for (int q = 0; q < VEC; ++q) dt[u] = fmaf(v[u][0][q], at_src[q], dt[u]);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) dt[u] = wave_sum(dt[u]);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int k = kb + j + u;
                    while (k == row_end) close_row();
                    const float z = rs_a[0] + dt[u];
                    const float e = z > 0.f ? z : z * P.slope;
                    if (e > m_run) {                      // wave-uniform: every lane holds the same e and m_run
                        const float rsc = expf(m_run - e);
                        s_run *= rsc;
#pragma unroll
                        for (int q = 0; q < VEC; ++q) acc[0][q] *= rsc;
                        m_run = e;
                    }
                    const float we = expf(e - m_run);
                    s_run += we;
#pragma unroll
                    for (int q = 0; q < VEC; ++q) acc[0][q] = fmaf(we, v[u][0][q], acc[0][q]);
                }
                continue;
            }
            float pd[U];                 // W_GAT_SRC_FUSED: this lane's share of <gathered row, open row's own features>
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = kb + j + u;
                while (k == row_end) close_row();
#pragma unroll
                for (int c = 0; c < NCH; ++c) unpack_row<VEC, T>(raw[u][c], v[u][c]);    // first use: all U NCH loads are in flight
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
            const T* src = (TWO_PART && cu >= split) ? x2T + (int64_t)(cu - split) * P.ldx : xT + (int64_t)cu * P.ldx;
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
            if constexpr (SMX) {
                float dt = 0.f;
#pragma unroll
                for (int q = 0; q < VEC; ++q) dt = fmaf(v[0][q], at_src[q], dt);
                dt = wave_sum(dt);
                const float z = rs_a[0] + dt;
                const float e = z > 0.f ? z : z * P.slope;
                if (e > m_run) {
                    const float rsc = expf(m_run - e);
                    s_run *= rsc;
#pragma unroll
                    for (int q = 0; q < VEC; ++q) acc[0][q] *= rsc;
                    m_run = e;
                }
                const float we = expf(e - m_run);
                s_run += we;
#pragma unroll
                for (int q = 0; q < VEC; ++q) acc[0][q] = fmaf(we, v[0][q], acc[0][q]);
                continue;
            }
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
            float dzv = 0.f;
            if (lane < nb) {
                dzv = wv * (pb[lane] - dz_d) * dz_g;
                P.dz_out[kb + lane] = dzv;
            }
            __builtin_amdgcn_wave_barrier();
            if (rs_on) {
                // segmented inclusive scan keyed by the row id (sorted: "same key at distance d" = the whole span shares it), as
                // segscan.hip; whether the block's LAST run is complete is only known with the next block's first key (or, behind
                // the item's last block, from rowptr), so it always stays open
                if (rs_open_key >= 0 && bcast_i(rlk, 0) != rs_open_key) {
                    if (lane == 0) rs_put(rs_open_key, rs_open_val);
                    rs_open_key = -1;
                }
                float xs = dzv;
#pragma unroll
                for (int d = 1; d < WAVE; d <<= 1) {
                    const int ko = __shfl_up(rlk, d, WAVE);
                    const float o = __shfl_up(xs, d, WAVE);
                    if (lane >= d && ko == rlk) xs += o;
                }
                if (rlk == rs_open_key) xs += rs_open_val;                  // the first run continues the open row
                const int key_next = __shfl_down(rlk, 1, WAVE);
                if (lane < nb - 1 && key_next != rlk) rs_put(rlk, xs);
                rs_open_key = bcast_i(rlk, nb - 1);
                rs_open_val = bcast_f(xs, nb - 1);
            }
        }
    }
    if (rs_on) {
        int tail_r = -1;
        if (rs_open_key >= 0) {
            const bool complete = uniform_i(P.rowptr[rs_open_key + 1]) == k1;
            if (lane == 0) {
                if (complete || rs_open_key == rs_prev) rs_put(rs_open_key, rs_open_val);   // (the whole item inside one row: a link too)
                else P.rs_tail[item] = rs_open_val;
            }
            if (!complete && rs_open_key != rs_prev) tail_r = rs_open_key;
        }
        if (lane == 0) P.rs_tail_row[item] = tail_r;
    }
    // entries exhausted at k1
    if (row_end == k1) {
        close_row();                                     // row r ends exactly here
        while (r < N && row_end == k1) close_row();      // empty rows behind it (out = bias)
    } else if (head) {
        write_carry(0);                                  // one row spans the whole item
        if (lane == 0) {
            M->head_row = r; M->head_rs = row_start; M->head_re = row_end; M->head_closed = 0;
            if constexpr (SMX) { M->head_m = m_run; M->head_s = s_run; }
        }
    } else {
        write_carry(1);                                  // row continues in the next item
        if (lane == 0) {
            M->tail_row = r; M->tail_rs = row_start; M->tail_re = row_end;
            if constexpr (SMX) { M->tail_m = m_run; M->tail_s = s_run; }
        }
    }
}

// the HBM-bound kernels of the headline (one 16-byte chunk per lane, no GAT weights) must keep 8 waves per SIMD: <= 64 VGPRs
// (W_GAT_DST_FUSED: 6 waves per SIMD -- 80 VGPRs; left alone the allocator takes 93 for the chain resolution's sake, 5 waves)
template <int VEC, int NCH, int WMODE, int EXACT> struct seg_min_waves {
    // (EXACT == 2: the variant that also writes the finished rows' power-of-two scales -- 70 registers, 7 waves; measured at C4
    // against the 8-wave one: no difference in the launch, EXPERIMENTS A34)
    static constexpr int value = (NCH == 1 && WMODE <= W_ARRAY && EXACT) ? (EXACT == 2 ? 7 : 8) : (NCH == 1 && WMODE == W_GAT_DST_FUSED) ? 6 : 1;
};

template <typename T, int VEC, int NCH, int WMODE, int EXACT>
__global__ void __launch_bounds__(SEG_THREADS, (seg_min_waves<VEC, NCH, WMODE, EXACT>::value))
segsum_kernel(SegParams P) {
    constexpr int ROWF = NCH * VEC * WAVE;
    __shared__ __attribute__((aligned(16))) float s_part[SEG_WAVES][2][ROWF];
    __shared__ ItemMeta s_meta[SEG_WAVES];
    __shared__ int s_arrived;
    if (threadIdx.x == 0) s_arrived = 0;
    if (threadIdx.x < SEG_WAVES) { s_meta[threadIdx.x].head_row = -1; s_meta[threadIdx.x].tail_row = -1; }   // "no partial"
    __syncthreads();
    const int wave = uniform_i(threadIdx.x >> 6);           // wave-uniform: LDS addresses derived from it stay in SGPRs
    const int item = uniform_i(item_block(blockIdx.x, gridDim.x) * SEG_WAVES + wave);
    Lanes<VEC, NCH, WMODE, EXACT> L;
    L.init(P);
    if (item < P.n_items) segsum_item<T, VEC, NCH, WMODE, EXACT>(P, L, item, &s_part[wave][0][0], s_meta + wave);
    auto finish = [&](const float (&acc)[NCH][VEC], int r, int row_len, float m_, float s_) {
        finish_row<T, VEC, NCH, WMODE, EXACT>(P, L, acc, r, row_len, m_, s_);
    };
    resolve_block<VEC, NCH, ROWF, WMODE == W_GAT_DST_FUSED>(P, L, finish, s_part, s_meta, &s_arrived);
}

// Narrow rows (F <= 32 VEC: hidden = 128 or 64 in f32, the reference's own model width): one row is only
// half / a quarter of a wave instruction, so G = 2 or 4 ENTRIES are gathered per instruction -- lane group
// g = lane / (64 / G) fetches entry G j + g -- and every group keeps its own partial sum of the open row.
// Entries are still added in entry order (group 0, a possible row close, group 1, ...); at a row close
// the G partials are folded across the lane groups (xor shuffles) and group 0 stores.  Same items and the same in-launch
// resolution of cut rows as the wide path (the lanes of group 0 own the columns); W_NONE / W_ARRAY only.
template <int VEC> struct GroupLanes { bool act[1]; int foff[1]; };

template <typename T, int VEC, int G, int WMODE>
__device__ __forceinline__ void segsum_group_item(const SegParams& P, const int item, float* __restrict__ part,
                                                  ItemMeta* __restrict__ M) {
    constexpr int LG = WAVE / G;                           // lanes per entry group
    constexpr int ROWF = LG * VEC;
    constexpr int U = 8 / G < 2 ? 2 : 8 / G;               // wave instructions in flight: U * G = 8 rows (16 or 32 rows: no gain at
                                                           // 58k edges, -3 % / -40 % at 20M)
    const int lane = lane_id();
    const int N = P.N;
    const int nnz = P.rowptr[N];
    const int k0 = item * P.item;
    // a CSR with capacity but no entry at all (every edge dropped): item 0 still runs and closes all N empty rows
    if (k0 >= nnz && !(item == 0 && nnz == 0)) return;
    const int k1 = min(k0 + P.item, nnz);
    const int F = P.F;
    const T* __restrict__ xT = reinterpret_cast<const T*>(P.x);
    // (a pointer biased by -split rows through integer arithmetic loses its address space: every gather became a FLAT load)
    const T* __restrict__ x2T = reinterpret_cast<const T*>(P.x2);
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
            if (act && grp == 0) store_row<VEC, float>(part + foff, t);
            if (lane == 0) { M->head_row = r; M->head_rs = row_start; M->head_re = row_end; M->head_closed = 1; }
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
            typename vec_of<T, VEC>::type raw[U];
            float we[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = j + u * G + grp;             // this lane group's entry of wave instruction u
                const int cu = __shfl(cv, min(e, nb - 1), WAVE);
                we[u] = (WMODE == W_ARRAY) ? __shfl(wv, min(e, nb - 1), WAVE) : 1.f;
                raw[u] = zero_of<typename vec_of<T, VEC>::type>();
                if (act && e < nb) raw[u] = load_raw<VEC, T>(((TWO_PART && cu >= split) ? x2T + (int64_t)(cu - split) * P.ldx : xT + (int64_t)cu * P.ldx) + foff);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) unpack_row<VEC, T>(raw[u], v[u]);       // behind the last load: all U are in flight
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
        if (act && grp == 0) store_row<VEC, float>(part + (head ? 0 : ROWF) + foff, t);
        if (lane == 0) {
            if (head) { M->head_row = r; M->head_rs = row_start; M->head_re = row_end; M->head_closed = 0; }   // one row spans the whole item
            else { M->tail_row = r; M->tail_rs = row_start; M->tail_re = row_end; }                            // row continues in the next item
        }
    }
}

template <typename T, int VEC, int G, int WMODE>
__global__ void __launch_bounds__(SEG_THREADS, 8)            // 8 waves per SIMD: <= 64 VGPRs, <= 96 SGPRs
segsum_group_kernel(SegParams P) {
    constexpr int LG = WAVE / G;
    constexpr int ROWF = LG * VEC;
    __shared__ __attribute__((aligned(16))) float s_part[SEG_WAVES][2][ROWF];
    __shared__ ItemMeta s_meta[SEG_WAVES];
    __shared__ int s_arrived;
    if (threadIdx.x == 0) s_arrived = 0;
    if (threadIdx.x < SEG_WAVES) { s_meta[threadIdx.x].head_row = -1; s_meta[threadIdx.x].tail_row = -1; }   // "no partial"
    __syncthreads();
    const int wave = uniform_i(threadIdx.x >> 6);           // wave-uniform: LDS addresses derived from it stay in SGPRs
    const int lane = lane_id();
    const int item = uniform_i(item_block(blockIdx.x, gridDim.x) * SEG_WAVES + wave);
    if (item < P.n_items) segsum_group_item<T, VEC, G, WMODE>(P, item, &s_part[wave][0][0], s_meta + wave);
    GroupLanes<VEC> L;
    L.foff[0] = (lane % LG) * VEC;
    L.act[0] = lane < LG && L.foff[0] < P.F;                 // the lanes of group 0 own the columns
    auto finish = [&](const float (&acc)[1][VEC], int r, int row_len, float, float) {
        if (!L.act[0]) return;
        const T* __restrict__ bias = reinterpret_cast<const T*>(P.bias);
        const float sc = P.mean ? 1.f / (float)max(row_len, 1) : 1.f;
        float t[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) t[q] = fmaf(acc[0][q], sc, bias ? to_f32(bias[L.foff[0] + q]) : 0.f);
        store_row<VEC, T>(reinterpret_cast<T*>(P.out) + (int64_t)r * P.ldo + L.foff[0], t);
    };
    resolve_block<VEC, 1, ROWF>(P, L, finish, s_part, s_meta, &s_arrived);
}

template <typename T, int VEC, int NCH, int WMODE, int EXACT>
static void launch_one(const SegParams& P, hipStream_t stream) {
    dim3 grid(seg_grid(P.n_items)), block(SEG_THREADS);
    segsum_kernel<T, VEC, NCH, WMODE, EXACT><<<grid, block, 0, stream>>>(P);
}

template <typename T, int VEC, int G, int WMODE>
static void launch_group(const SegParams& P, hipStream_t stream) {
    dim3 grid(seg_grid(P.n_items)), block(SEG_THREADS);
    segsum_group_kernel<T, VEC, G, WMODE><<<grid, block, 0, stream>>>(P);
}
template <typename T, int VEC, int G>
static int launch_group_modes(const SegParams& P, int wmode, int mean, hipStream_t stream) {
    if (wmode == W_NONE) launch_group<T, VEC, G, W_NONE>(P, stream);
    else                 launch_group<T, VEC, G, W_ARRAY>(P, stream);
    return check_launch("npi_segsum");
}

template <typename T, int VEC, int NCH, int EXACT>
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
        if constexpr (VEC == 4 && sizeof(T) == 4) launch_one<T, VEC, NCH, W_GAT_DST_PRE, EXACT>(P, stream);
    } else if (wmode == W_GAT_DST_FUSED) {
        if constexpr (VEC == 4 && sizeof(T) == 4 && NCH == 1) launch_one<T, VEC, NCH, W_GAT_DST_FUSED, EXACT>(P, stream);
        else {
            set_error("npi_gat_aggregate_fused: needs one head of at most 256 channels");
            return NPI_ERR_ARG;
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
        if constexpr (VEC == 4 && NC == 1 && sizeof(T) == 4) {                              \
            if (exact && P.scale_out != nullptr && (wmode <= W_ARRAY || is_fused_mode(wmode) || wmode == W_GAT_DST_FUSED)) \
                return launch_segsum<T, VEC, NC, 2>(P, wmode, mean, stream);                \
        }                                                                                   \
        if constexpr (VEC == 4) { if (exact) return launch_segsum<T, VEC, NC, 1>(P, wmode, mean, stream); } \
        return launch_segsum<T, VEC, NC, 0>(P, wmode, mean, stream)
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
    P.nt_out = ((int64_t)P.N * F * ((dtype == NPI_BF16) ? 2 : 4) >= ((int64_t)64 << 20)) ? 1 : 0;
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
    const int64_t items = num_items_of(nnz_max, item_edges);
    if (items == 0) return 64;
    // one launch handles at most 1,024 columns (wider rows: several launches over column blocks, one after the other on the
    // stream, reusing the scratch): arrival counters + two partial rows per workgroup + two span rows per CHAIN_SPAN workgroups
    return CarryLayout(items).elems(F < 1024 ? F : 1024);
}

extern "C" int npi_segsum_scales_supported(int64_t F, int dtype) { return (F == 256 && dtype == NPI_F32) ? 1 : 0; }

extern "C" int npi_segsum_ex(const int32_t* rowptr, const int32_t* col, const int32_t* item_row, int64_t item_edges,
                              const float* w, int64_t N, int64_t nnz_max, const void* x_, int64_t ldx,
                              const void* x2_, int64_t split, void* out_, int64_t ldo, int64_t F, int dtype, int mean,
                              const float* bias, float* carry, float* row_scales_out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(row_scales_out == nullptr || (npi_segsum_scales_supported(F, dtype) && nnz_max > 0 && ldx % 4 == 0 && ldo % 4 == 0 &&
                                              ((uintptr_t)x_ % 16) == 0 && ((uintptr_t)out_ % 16) == 0 &&
                                              (x2_ == nullptr || ((uintptr_t)x2_ % 16) == 0)),
                "npi_segsum_ex: row_scales_out needs f32 rows of 256 columns, 16-byte aligned (npi_segsum_scales_supported), and a "
                "graph with entries");
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
    P.scale_out = row_scales_out;
    return segsum_run(P, w ? W_ARRAY : W_NONE, mean, nnz_max, dtype, stream);
}
