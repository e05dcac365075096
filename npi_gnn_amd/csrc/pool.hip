// Between-layer steps of the reference's Net_1 (reference src/classes.py:63-64,
// 67-68, 71-72): PyG 1.4.2 TopKPooling(ratio) and the [global_max_pool || global_mean_pool] readout,
// forward (inference: src/methods.py:87-96, src/test.py, src/case_study*.py) and backward (training:
// src/train_with_twoDataset.PY:52-54).
//
//   score_i = tanh(<x_i, w> / ||w||)
//   per graph g (nodes are contiguous per graph in a PyG Batch): keep the k_g = ceil(ratio n_g) highest
//       scores, descending, ties by lower node index (a stable descending sort)
//   x' = x[perm] * score[perm],  batch' = batch[perm]
//   filter_adj: keep the edges whose two ends survive, relabelled, in their original order
//   readout[g] = [max_i x'_i || mean_i x'_i]
//
// Integer / index work is bit-exact; sums are in a fixed order (reproducible).
#include <algorithm>
#include <initializer_list>
#include "npi_common.h"

namespace npi {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
}
__device__ __forceinline__ int wsum_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
}

// score[i] = tanh(<x_i, w> / ||w||_2); one wave per node
__global__ void __launch_bounds__(256)
topk_score_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ w, int N, int F,
                  float* __restrict__ score) {
    const int lane = lane_id();
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= N) return;
    float dot = 0.f, nn = 0.f;
    for (int c = lane; c < F; c += WAVE) {
        const float wc = w[c];
        dot = fmaf(x[(int64_t)i * ldx + c], wc, dot);
        nn = fmaf(wc, wc, nn);
    }
    dot = wsum(dot);
    nn = wsum(nn);
    if (lane == 0) score[i] = tanhf(dot / sqrtf(nn));
}

// graph_ptr[g] = first node with batch >= g (batch is non-decreasing), g in [0, B]
__global__ void graph_bounds_kernel(const int64_t* __restrict__ batch, int64_t N, int64_t B,
                                    int32_t* __restrict__ graph_ptr) {
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g > B) return;
    int64_t lo = 0, hi = N;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (batch[mid] < g) lo = mid + 1; else hi = mid;
    }
    graph_ptr[g] = (int32_t)lo;
}

// out_ptr[g] = sum_{g' < g} ceil(ratio * n_g'); one workgroup, B is small (graphs per batch).  The same launch clears the
// status word and presets remap to -1 (blocks 1 ..: one element per thread) -- three launches in one.
__global__ void __launch_bounds__(256)
topk_counts_kernel(const int32_t* __restrict__ graph_ptr, int B, float ratio, int32_t* __restrict__ out_ptr,
                   int32_t* __restrict__ status, int32_t* __restrict__ remap, int64_t N) {
    if (blockIdx.x > 0) {
        const int64_t i = (int64_t)(blockIdx.x - 1) * 256 + threadIdx.x;
        if (i < N) remap[i] = -1;
        return;
    }
    if (threadIdx.x == 0) *status = 0;
    __shared__ int lds[256];
    int carry = 0;
    for (int base = 0; base < B; base += 256) {
        const int g = base + threadIdx.x;
        int k = 0;
        if (g < B) {
            const int n = graph_ptr[g + 1] - graph_ptr[g];
            k = (int)ceilf(ratio * (float)n);
            k = min(k, n);
        }
        lds[threadIdx.x] = k;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            int t = (threadIdx.x >= off) ? lds[threadIdx.x - off] : 0;
            __syncthreads();
            lds[threadIdx.x] += t;
            __syncthreads();
        }
        if (g < B) out_ptr[g] = carry + lds[threadIdx.x] - k;
        carry += lds[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) out_ptr[B] = carry;
}

// sortable key: ascending key order == (score descending, node index ascending)
__device__ __forceinline__ uint64_t topk_key(float s, uint32_t idx) {
    uint32_t u = __float_as_uint(s);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);      // ascending float order as unsigned
    return ((uint64_t)(~u) << 32) | idx;                  // ~u: descending score
}

// One workgroup per graph: bitonic sort of (score, index) keys in LDS, write the k best node ids.
// The instantiation <CAP, LO> handles graphs with LO < n_g <= CAP (CAP a power of two).
template <int CAP, int LO>
__global__ void __launch_bounds__(CAP > 1024 ? 1024 : 256)
topk_select_kernel(const float* __restrict__ score, const int32_t* __restrict__ graph_ptr,
                   const int32_t* __restrict__ out_ptr, int B, int32_t* __restrict__ perm,
                   int32_t* __restrict__ remap, int32_t* __restrict__ too_big) {
    __shared__ uint64_t keys[CAP];
    const int g = blockIdx.x;
    const int nb = graph_ptr[g], n = graph_ptr[g + 1] - nb;
    if (n > CAP) { if (CAP > 1024 && threadIdx.x == 0) atomicOr(too_big, 2); return; }
    if (n <= LO) return;                                  // the smaller instantiation handles it
    const int T = blockDim.x;
    // the network only spans the next power of two above THIS graph's size (a 140-node graph sorts 256 keys in 36 stages,
    // not 1,024 in 55): the pooled layers' graphs are a half and a quarter of the first one's
    int cap = 64;
    while (cap < n) cap <<= 1;
    for (int i = threadIdx.x; i < cap; i += T) keys[i] = (i < n) ? topk_key(score[nb + i], (uint32_t)i) : ~0ull;
    __syncthreads();
    for (int k = 2; k <= cap; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < cap; i += T) {
                const int l = i ^ j;
                if (l > i) {
                    const uint64_t a = keys[i], b = keys[l];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { keys[i] = b; keys[l] = a; }
                }
            }
            __syncthreads();
        }
    }
    const int ob = out_ptr[g], kk = out_ptr[g + 1] - ob;
    for (int i = threadIdx.x; i < kk; i += T) {
        const int node = nb + (int)(keys[i] & 0xffffffffu);
        perm[ob + i] = node;
        remap[node] = ob + i;
    }
}

// x'[q] = x[perm[q]] * score[perm[q]], batch'[q] = batch[perm[q]]; one wave per output row
__global__ void __launch_bounds__(256)
topk_gather_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ score,
                   const int64_t* __restrict__ batch, const int32_t* __restrict__ perm,
                   const int32_t* __restrict__ out_ptr, int B, int F, float* __restrict__ xo, int64_t ldo,
                   int64_t* __restrict__ batch_o, float* __restrict__ score_o, int64_t* __restrict__ perm64) {
    const int lane = lane_id();
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= out_ptr[B]) return;
    const int i = perm[q];
    const float s = score[i];
    for (int c = lane; c < F; c += WAVE) xo[(int64_t)q * ldo + c] = x[(int64_t)i * ldx + c] * s;
    if (lane == 0) {
        batch_o[q] = batch[i];
        score_o[q] = s;
        if (perm64) perm64[q] = i;                         // the LongTensor PyG returns, without a cast launch
    }
}

// ---- filter_adj: order-preserving compaction of the surviving edges -------------------------------------
constexpr int FA_TILE = 2048;      // edges per workgroup (256 threads x 8)
__global__ void __launch_bounds__(256)
filter_flag_kernel(const int64_t* __restrict__ src, const int64_t* __restrict__ dst, int64_t E,
                   const int32_t* __restrict__ remap, int32_t* __restrict__ tile_counts,
                   int64_t* __restrict__ pad_src, int64_t* __restrict__ pad_dst) {
    __shared__ int cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    int c = 0;
    const int64_t base = (int64_t)blockIdx.x * FA_TILE;
    for (int j = 0; j < 8; ++j) {
        const int64_t e = base + j * 256 + threadIdx.x;
        if (e < E) {
            const int64_t s0 = src[e], d0 = dst[e];              // negative ids: padding columns of an earlier filter
            c += (s0 >= 0 && d0 >= 0 && remap[s0] >= 0 && remap[d0] >= 0) ? 1 : 0;
            if (pad_src != nullptr) { pad_src[e] = -1; pad_dst[e] = -1; }   // filter_write overwrites the first `count`
        }
    }
    c = (int)wsum((float)c);          // <= 512 per wave: exact in f32
    if (lane_id() == 0) atomicAdd(&cnt, c);
    __syncthreads();
    if (threadIdx.x == 0) tile_counts[blockIdx.x] = cnt;
}
__global__ void __launch_bounds__(256)
scan_small_kernel(int32_t* __restrict__ v, int n, int32_t* __restrict__ total) {   // exclusive scan, one workgroup
    __shared__ int lds[256];
    int carry = 0;
    for (int base = 0; base < n; base += 256) {
        const int i = base + threadIdx.x;
        const int val = i < n ? v[i] : 0;
        lds[threadIdx.x] = val;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            int t = (threadIdx.x >= off) ? lds[threadIdx.x - off] : 0;
            __syncthreads();
            lds[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < n) v[i] = carry + lds[threadIdx.x] - val;
        carry += lds[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}
// each workgroup re-walks its tile in edge order: wave w handles edges [w*512, w*512+512) of the tile in
// 8 steps of 64 consecutive edges, so ballot ranks give the original order
__global__ void __launch_bounds__(256)
filter_write_kernel(const int64_t* __restrict__ src, const int64_t* __restrict__ dst, int64_t E,
                    const int32_t* __restrict__ remap, const int32_t* __restrict__ tile_off,
                    int64_t* __restrict__ out_src, int64_t* __restrict__ out_dst, int32_t* __restrict__ newpos,
                    int32_t* __restrict__ total, int scanned) {
    __shared__ int wcnt[4];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int64_t wbase = (int64_t)blockIdx.x * FA_TILE + wave * 512;
    int64_t s[8], d[8];
    bool keep[8];
    int c = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t e = wbase + j * 64 + lane;
        keep[j] = false;
        if (e < E) {
            const int64_t s0 = src[e], d0 = dst[e];
            if (s0 >= 0 && d0 >= 0) {
                const int rs = remap[s0], rd = remap[d0];
                keep[j] = rs >= 0 && rd >= 0;
                s[j] = rs; d[j] = rd;
            }
        }
        c += __popcll(__ballot(keep[j]));
    }
    // the tile's offset = the counts of the tiles in front of it (a few hundred integers: no scan launch in between);
    // the first workgroup also publishes the total
    __shared__ int psum[4];
    int pre = 0, all = 0;
    if (scanned) pre = threadIdx.x == 0 ? tile_off[blockIdx.x] : 0;     // many tiles: the host ran scan_small_kernel in between
    else for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) {
        const int v = tile_off[i];
        all += v;
        if (i < (int)blockIdx.x) pre += v;
    }
    pre = wsum_i(pre);
    if (lane == 0) { wcnt[wave] = c; psum[wave] = pre; }
    if (blockIdx.x == 0 && total != nullptr && !scanned) {
        __shared__ int asum[4];
        all = wsum_i(all);
        if (lane == 0) asum[wave] = all;
        __syncthreads();
        if (threadIdx.x == 0) *total = asum[0] + asum[1] + asum[2] + asum[3];
    }
    __syncthreads();
    int off = psum[0] + psum[1] + psum[2] + psum[3];
    for (int w = 0; w < wave; ++w) off += wcnt[w];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint64_t m = __ballot(keep[j]);
        const int64_t e = wbase + j * 64 + lane;
        int pos = -1;
        if (keep[j]) {
            pos = off + __popcll(m & ((1ull << lane) - 1ull));
            out_src[pos] = s[j];
            out_dst[pos] = d[j];
        }
        if (newpos != nullptr && e < E) newpos[e] = pos;      // where the edge went (-1: dropped): npi_csr_filter's eid map
        off += __popcll(m);
    }
}

// Rows x column groups: a workgroup of POOL_THREADS handles one graph and `cgb` groups of VEC adjacent columns with
// POOL_THREADS / cgb row lanes striding over the graph's rows (F = 128: 32 groups of 4 columns x 32 row lanes), so a
// 900-row graph is ~30 independent 16-byte loads per thread instead of 900 dependent ones.  The lane partials meet in
// LDS and are combined in lane order: the result does not depend on scheduling.
constexpr int POOL_THREADS = 1024;

template <int VEC> struct VecLoad;
template <> struct VecLoad<1> { typedef float T; };
template <> struct VecLoad<2> { typedef float2 T; };
template <> struct VecLoad<4> { typedef float4 T; };

template <int VEC>
__device__ __forceinline__ void load_vec(const float* __restrict__ p, float (&v)[VEC]) {
    typename VecLoad<VEC>::T t = *reinterpret_cast<const typename VecLoad<VEC>::T*>(p);
    const float* f = reinterpret_cast<const float*>(&t);
#pragma unroll
    for (int k = 0; k < VEC; ++k) v[k] = f[k];
}
template <int VEC>
__device__ __forceinline__ void store_vec(float* __restrict__ p, const float (&v)[VEC]) {
    typename VecLoad<VEC>::T t;
    float* f = reinterpret_cast<float*>(&t);
#pragma unroll
    for (int k = 0; k < VEC; ++k) f[k] = v[k];
    *reinterpret_cast<typename VecLoad<VEC>::T*>(p) = t;
}

// widest VEC in {4, 2, 1} that divides F and every leading dimension and keeps every base pointer aligned
static int pool_vec(int64_t F, std::initializer_list<int64_t> lds, std::initializer_list<const void*> ptrs) {
    for (int v = 4; v > 1; v >>= 1) {
        bool ok = F % v == 0;
        for (int64_t l : lds) ok = ok && l % v == 0;
        for (const void* p : ptrs) ok = ok && ((uintptr_t)p % (v * sizeof(float))) == 0;
        if (ok) return v;
    }
    return 1;
}
static int pool_groups_per_block(int64_t F, int vec) { return (int)std::min<int64_t>(ceil_div(F, vec), POOL_THREADS); }

// readout[g] = [max over the graph's rows || mean over the graph's rows]; an empty graph gives zeros
template <int VEC>
__global__ void __launch_bounds__(POOL_THREADS)
readout_kernel(const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ graph_ptr, int B, int F, int cgb,
               float* __restrict__ out /* [B, 2F] */) {
    __shared__ float mx_s[POOL_THREADS * VEC];
    __shared__ float sum_s[POOL_THREADS * VEC];
    const int g = blockIdx.x;
    const int lanes = POOL_THREADS / cgb;
    const int cgi = threadIdx.x % cgb, rl = threadIdx.x / cgb;
    const int c0 = (blockIdx.y * cgb + cgi) * VEC;
    const int b = graph_ptr[g], e = graph_ptr[g + 1];
    const bool live = rl < lanes && c0 < F;
    float mx[VEC], sum[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) { mx[k] = -3.0e38f; sum[k] = 0.f; }
    if (live) {
#pragma unroll 4
        for (int i = b + rl; i < e; i += lanes) {
            float v[VEC];
            load_vec<VEC>(x + (int64_t)i * ldx + c0, v);
#pragma unroll
            for (int k = 0; k < VEC; ++k) { mx[k] = fmaxf(mx[k], v[k]); sum[k] += v[k]; }
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            mx_s[(rl * cgb + cgi) * VEC + k] = mx[k];
            sum_s[(rl * cgb + cgi) * VEC + k] = sum[k];
        }
    }
    __syncthreads();
    const int cols = cgb * VEC;
    for (int t = threadIdx.x; t < cols; t += POOL_THREADS) {
        const int c = blockIdx.y * cols + t;
        if (c >= F) continue;
        float m = -3.0e38f, sm = 0.f;
        for (int r = 0; r < lanes; ++r) {
            m = fmaxf(m, mx_s[r * cols + t]);
            sm += sum_s[r * cols + t];
        }
        out[(int64_t)g * 2 * F + c] = (e > b) ? m : 0.f;
        out[(int64_t)g * 2 * F + F + c] = sm / (float)max(e - b, 1);
    }
}


// ---- backward of the gate x[perm] * score[perm] and of score = tanh(x.w / ||w||) --------------------
// one wave per kept node p (row i = perm[p]):
//   ds = <dxo[p], x[i]> (+ dscore_o[p]);  dz = ds (1 - s^2);  dx[i] = dxo[p] s + dz w / ||w||
//   dzv[p] = dz,  dzz[p] = dz z  with z = <x[i], w> / ||w||   (inputs of the weight gradient)
// rows that were not kept receive no gradient: dx must come in zero-filled.
__device__ __forceinline__ void topk_bwd_row(const float* __restrict__ x, int64_t ldx, const float* __restrict__ score,
                                             const float* __restrict__ w, int F, const float* __restrict__ dxo, int64_t lddxo,
                                             const float* __restrict__ dscore_o, float* __restrict__ dx, int64_t lddx,
                                             float* __restrict__ dzv, float* __restrict__ dzz, int p, int i, int lane) {
    const float* __restrict__ xr = x + (int64_t)i * ldx;
    const float* __restrict__ gr = dxo + (int64_t)p * lddxo;
    float ds = 0.f, dot = 0.f, nn = 0.f;
    for (int c = lane; c < F; c += WAVE) {
        const float wc = w[c], xv = xr[c];
        ds = fmaf(gr[c], xv, ds);
        dot = fmaf(xv, wc, dot);
        nn = fmaf(wc, wc, nn);
    }
    ds = wsum(ds);
    dot = wsum(dot);
    nn = wsum(nn);
    const float inv_norm = 1.f / sqrtf(nn);
    const float sc = score[i];
    if (dscore_o) ds += dscore_o[p];
    const float dz = ds * (1.f - sc * sc);
    float* __restrict__ dr = dx + (int64_t)i * lddx;
    for (int c = lane; c < F; c += WAVE) dr[c] = fmaf(gr[c], sc, dz * w[c] * inv_norm);
    if (lane == 0) {
        dzv[p] = dz;
        dzz[p] = dz * dot * inv_norm;
    }
}

__global__ void __launch_bounds__(256)
topk_gather_bwd_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ score,
                       const float* __restrict__ w, const int32_t* __restrict__ perm, int n_out, int F,
                       const float* __restrict__ dxo, int64_t lddxo, const float* __restrict__ dscore_o,
                       float* __restrict__ dx, int64_t lddx, float* __restrict__ dzv, float* __restrict__ dzz) {
    const int lane = lane_id();
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= n_out) return;
    const int i = perm[p];
    topk_bwd_row(x, ldx, score, w, F, dxo, lddxo, dscore_o, dx, lddx, dzv, dzz, p, i, lane);
}

// the same over ALL input rows (one wave per row i; remap[i] = its kept position or -1): dropped rows get their zeros here,
// so dx needs no zero-filling launch in front
__global__ void __launch_bounds__(256)
topk_gather_bwd_all_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ score,
                           const float* __restrict__ w, const int32_t* __restrict__ remap, int N, int F,
                           const float* __restrict__ dxo, int64_t lddxo, const float* __restrict__ dscore_o,
                           float* __restrict__ dx, int64_t lddx, float* __restrict__ dzv, float* __restrict__ dzz) {
    const int lane = lane_id();
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= N) return;
    const int p = remap[i];
    if (p < 0) {
        float* __restrict__ dr = dx + (int64_t)i * lddx;
        for (int c = lane; c < F; c += WAVE) dr[c] = 0.f;
        return;
    }
    topk_bwd_row(x, ldx, score, w, F, dxo, lddxo, dscore_o, dx, lddx, dzv, dzz, p, i, lane);
}

// dw partials over chunks of POOLW_ROWS kept nodes: part[chunk][f] = sum_p dzv[p] x[perm[p], f]; part[chunk][F] = sum_p dzz[p]
constexpr int POOLW_ROWS = 256;
template <int VEC>
__global__ void __launch_bounds__(POOL_THREADS)
topk_weight_grad_partial_kernel(const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ perm,
                                const float* __restrict__ dzv, const float* __restrict__ dzz, int n_out, int F, int cgb,
                                float* __restrict__ part) {
    __shared__ float s_s[POOL_THREADS * VEC];
    __shared__ float z_s[POOL_THREADS];
    const int lanes = POOL_THREADS / cgb;
    const int cgi = threadIdx.x % cgb, rl = threadIdx.x / cgb;
    const int c0 = (blockIdx.y * cgb + cgi) * VEC;
    const int pb = blockIdx.x * POOLW_ROWS, pe = min(n_out, pb + POOLW_ROWS);
    const bool live = rl < lanes && c0 < F;
    const bool scalar = blockIdx.y == 0 && cgi == 0 && rl < lanes;      // these lanes also sum dzz (column F)
    float s[VEC], z = 0.f;
#pragma unroll
    for (int k = 0; k < VEC; ++k) s[k] = 0.f;
    if (live) {
#pragma unroll 4
        for (int p = pb + rl; p < pe; p += lanes) {
            float v[VEC];
            load_vec<VEC>(x + (int64_t)perm[p] * ldx + c0, v);
            const float d = dzv[p];
#pragma unroll
            for (int k = 0; k < VEC; ++k) s[k] = fmaf(d, v[k], s[k]);
            if (scalar) z += dzz[p];
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) s_s[(rl * cgb + cgi) * VEC + k] = s[k];
    }
    if (scalar) z_s[rl] = z;
    __syncthreads();
    const int cols = cgb * VEC;
    for (int t = threadIdx.x; t < cols; t += POOL_THREADS) {
        const int c = blockIdx.y * cols + t;
        if (c >= F) continue;
        float acc = 0.f;
        for (int r = 0; r < lanes; ++r) acc += s_s[r * cols + t];
        part[(int64_t)blockIdx.x * (F + 1) + c] = acc;
    }
    if (blockIdx.y == 0 && threadIdx.x == 0) {
        float acc = 0.f;
        for (int r = 0; r < lanes; ++r) acc += z_s[r];
        part[(int64_t)blockIdx.x * (F + 1) + F] = acc;
    }
}
// dw[f] = (sum_chunks part[.][f]) / ||w|| - w[f] (sum_chunks part[.][F]) / ||w||^2
// 64 columns x 4 chunk lanes per workgroup (the chunk loop is a chain of dependent-latency loads otherwise);
// lanes are combined in lane order, so the sum order is fixed.
__global__ void __launch_bounds__(256)
topk_weight_grad_reduce_kernel(const float* __restrict__ part, int nchunks, const float* __restrict__ w, int F,
                               float* __restrict__ dw) {
    __shared__ float s_s[4][64], z_s[4], n_s[4];
    const int col = threadIdx.x & 63, ql = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + col;
    float s = 0.f, z = 0.f, nn = 0.f;
    if (c < F) {
#pragma unroll 8
        for (int q = ql; q < nchunks; q += 4) s += part[(int64_t)q * (F + 1) + c];
    }
    for (int q = ql * 64 + col; q < nchunks; q += 256) z += part[(int64_t)q * (F + 1) + F];
    for (int k = ql * 64 + col; k < F; k += 256) nn = fmaf(w[k], w[k], nn);
    z = wsum(z);
    nn = wsum(nn);
    s_s[ql][col] = s;
    if (col == 0) { z_s[ql] = z; n_s[ql] = nn; }
    __syncthreads();
    if (ql == 0 && c < F) {
        const float st = (s_s[0][col] + s_s[1][col]) + (s_s[2][col] + s_s[3][col]);
        const float zt = (z_s[0] + z_s[1]) + (z_s[2] + z_s[3]);
        const float nt = (n_s[0] + n_s[1]) + (n_s[2] + n_s[3]);
        dw[c] = st / sqrtf(nt) - w[c] * zt / nt;
    }
}

// backward of [max || mean]: dx[i, c] = dout[g, F + c] / n_g + (i is the FIRST row of graph g with x = max ? dout[g, c] : 0)
// (a single arg-max row takes the gradient, as torch_scatter's scatter_max backward does)
template <int VEC>
__global__ void __launch_bounds__(POOL_THREADS)
readout_bwd_kernel(const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ graph_ptr, int B, int F, int cgb,
                   const float* __restrict__ out, const float* __restrict__ dout, float* __restrict__ dx, int64_t lddx, int N) {
    __shared__ int first_s[POOL_THREADS * VEC];
    const int g = blockIdx.x;
    const int lanes = POOL_THREADS / cgb;
    const int cgi = threadIdx.x % cgb, rl = threadIdx.x / cgb;
    const int c0 = (blockIdx.y * cgb + cgi) * VEC;
    const int b = graph_ptr[g], e = graph_ptr[g + 1];
    const bool live = rl < lanes && c0 < F;
    if (N >= 0 && live && (g == 0 || g == B - 1)) {
        // rows outside every graph (none in a PyG batch) get no gradient: zeroed here instead of by a fill launch in front
        float z[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) z[k] = 0.f;
        if (g == 0) for (int i = rl; i < graph_ptr[0]; i += lanes) store_vec<VEC>(dx + (int64_t)i * lddx + c0, z);
        if (g == B - 1) for (int i = graph_ptr[B] + rl; i < N; i += lanes) store_vec<VEC>(dx + (int64_t)i * lddx + c0, z);
    }
    if (e <= b) return;                                    // uniform over the workgroup
    float mx[VEC], dmx[VEC], dmean[VEC];
    int first[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) first[k] = 0x7fffffff;
    if (live) {
        load_vec<VEC>(out + (int64_t)g * 2 * F + c0, mx);
        load_vec<VEC>(dout + (int64_t)g * 2 * F + c0, dmx);
        load_vec<VEC>(dout + (int64_t)g * 2 * F + F + c0, dmean);
#pragma unroll
        for (int k = 0; k < VEC; ++k) dmean[k] /= (float)(e - b);
#pragma unroll 4
        for (int i = b + rl; i < e; i += lanes) {          // first row of this lane's stride that holds the maximum
            float v[VEC];
            load_vec<VEC>(x + (int64_t)i * ldx + c0, v);
#pragma unroll
            for (int k = 0; k < VEC; ++k) first[k] = (v[k] == mx[k]) ? min(first[k], i) : first[k];
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) first_s[(rl * cgb + cgi) * VEC + k] = first[k];
    }
    __syncthreads();
    if (!live) return;
    for (int r = 0; r < lanes; ++r) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) first[k] = min(first[k], first_s[(r * cgb + cgi) * VEC + k]);
    }
#pragma unroll 4
    for (int i = b + rl; i < e; i += lanes) {
        float v[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) v[k] = dmean[k] + (i == first[k] ? dmx[k] : 0.f);
        store_vec<VEC>(dx + (int64_t)i * lddx + c0, v);
    }
}

// ---- evaluation: confusion matrix of argmax predictions (reference src/methods.py:87-105) ---------------
// pred = index of the first maximum of the row (torch .max(dim=1)[1]);  counts += [TP, FN, TN, FP] with the
// reference's rule: pred 1 & y 1 -> TP, pred 1 & y 0 -> FP, pred 0 & y 1 -> FN, everything else -> TN.
// Integer atomics: the result does not depend on the order.
__global__ void __launch_bounds__(256)
confusion_kernel(const float* __restrict__ scores, int64_t lds, int C, const int64_t* __restrict__ y, int64_t B,
                 unsigned long long* __restrict__ counts) {
    const int lane = lane_id();
    unsigned long long tp = 0, fn = 0, tn = 0, fp = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (B + WAVE - 1) / WAVE * WAVE;
         i += (int64_t)gridDim.x * blockDim.x) {
        int cls = 3;                                       // 0 TP, 1 FN, 2 TN, 3 FP; lanes past the end vote nothing
        const bool live = i < B;
        if (live) {
            const float* __restrict__ r = scores + i * lds;
            int pred = 0;
            float best = r[0];
            for (int c = 1; c < C; ++c)
                if (r[c] > best) { best = r[c]; pred = c; }
            const int64_t yy = y[i];
            cls = (pred == 1 && yy == 1) ? 0 : (pred == 1 && yy == 0) ? 3 : (pred == 0 && yy == 1) ? 1 : 2;
        }
        tp += __popcll(__ballot(live && cls == 0));
        fn += __popcll(__ballot(live && cls == 1));
        tn += __popcll(__ballot(live && cls == 2));
        fp += __popcll(__ballot(live && cls == 3));
    }
    if (lane == 0) {
        if (tp) atomicAdd(counts + 0, tp);
        if (fn) atomicAdd(counts + 1, fn);
        if (tn) atomicAdd(counts + 2, tn);
        if (fp) atomicAdd(counts + 3, fp);
    }
}

__global__ void fill_i32_pool_kernel(int32_t* __restrict__ p, int64_t n, int32_t v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace npi

using namespace npi;

extern "C" int npi_topk_score(const float* x, int64_t ldx, const float* w, int64_t N, int64_t F, float* score,
                              void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(N >= 0 && F > 0 && ldx >= F, "npi_topk_score: bad size");
    if (N == 0) return NPI_OK;
    NPI_REQUIRE(x && w && score, "npi_topk_score: null pointer");
    topk_score_kernel<<<(unsigned)ceil_div(N, 4), 256, 0, stream>>>(x, ldx, w, (int)N, (int)F, score);
    return check_launch("npi_topk_score");
}

extern "C" int npi_graph_bounds(const int64_t* batch, int64_t N, int64_t B, int32_t* graph_ptr, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(N >= 0 && B >= 0 && N < 0x7fffffff, "npi_graph_bounds: bad size");
    NPI_REQUIRE(graph_ptr && (N == 0 || batch), "npi_graph_bounds: null pointer");
    graph_bounds_kernel<<<(unsigned)ceil_div(B + 1, 256), 256, 0, stream>>>(batch, N, B, graph_ptr);
    return check_launch("npi_graph_bounds");
}

// perm[out_ptr[B]] (node ids, graph-major, score-descending), remap[N] (new id or -1), out_ptr[B+1];
// status[0] bit 1 is set when a graph has more than 16384 nodes (unsupported)
// max_nodes: an upper bound of the largest graph of the batch the CALLER knows (0 = unknown): at most 1,024 skips the launch
// for graphs of 1,025 .. 16,384 nodes (it would find none)
extern "C" int npi_topk_select(const float* score, const int32_t* graph_ptr, int64_t N, int64_t B, float ratio,
                                  int32_t* out_ptr, int32_t* perm, int32_t* remap, int32_t* status, int64_t max_nodes,
                                  void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(N >= 0 && B >= 0 && ratio > 0.f && ratio <= 1.f && max_nodes >= 0, "npi_topk_select: bad argument");
    NPI_REQUIRE(out_ptr && status && (N == 0 || (score && graph_ptr && perm && remap)), "npi_topk_select: null pointer");
    topk_counts_kernel<<<(unsigned)(1 + ceil_div(N, 256)), 256, 0, stream>>>(graph_ptr, (int)B, ratio, out_ptr, status, remap, N);
    if (B > 0) {
        topk_select_kernel<1024, -1><<<(unsigned)B, 256, 0, stream>>>(score, graph_ptr, out_ptr, (int)B, perm, remap, status);
        if (max_nodes == 0 || max_nodes > 1024)
            topk_select_kernel<16384, 1024><<<(unsigned)B, 1024, 0, stream>>>(score, graph_ptr, out_ptr, (int)B, perm, remap, status);
    }
    return check_launch("npi_topk_select");
}

extern "C" int npi_topk_gather(const float* x, int64_t ldx, const float* score, const int64_t* batch,
                                  const int32_t* perm, const int32_t* out_ptr, int64_t B, int64_t F, int64_t n_out_max,
                                  float* xo, int64_t ldo, int64_t* batch_o, float* score_o, int64_t* perm64, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(B >= 0 && F > 0 && n_out_max >= 0, "npi_topk_gather: bad size");
    if (n_out_max == 0) return NPI_OK;
    NPI_REQUIRE(x && score && batch && perm && out_ptr && xo && batch_o && score_o, "npi_topk_gather: null pointer");
    topk_gather_kernel<<<(unsigned)ceil_div(n_out_max, 4), 256, 0, stream>>>(x, ldx, score, batch, perm, out_ptr, (int)B, (int)F, xo, ldo, batch_o, score_o, perm64);
    return check_launch("npi_topk_gather");
}

// tile counts (+ 2), then -- npi_filter_adj only -- the new position of every input edge (-1: dropped), which
// npi_csr_filter takes as its `newpos`
extern "C" int64_t npi_filter_adj_workspace_elems(int64_t E) { return ceil_div(E > 0 ? E : 1, FA_TILE) + 2 + (E > 0 ? E : 0); }
extern "C" int64_t npi_filter_adj_newpos_offset(int64_t E) { return ceil_div(E > 0 ? E : 1, FA_TILE) + 2; }

// out_src/out_dst: capacity E; count[0] = number of surviving edges (device)
extern "C" int npi_filter_adj(const int64_t* src, const int64_t* dst, int64_t E, const int32_t* remap,
                                 int64_t* out_src, int64_t* out_dst, int32_t* count, int32_t* workspace,
                                 int pad_tail, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(E >= 0 && E < 0x7fffffff, "npi_filter_adj: bad size");
    NPI_REQUIRE(count && workspace, "npi_filter_adj: null pointer");
    if (E == 0) { (void)hipMemsetAsync(count, 0, sizeof(int32_t), stream); return check_launch("npi_filter_adj"); }
    NPI_REQUIRE(src && dst && remap && out_src && out_dst, "npi_filter_adj: null pointer");
    NPI_REQUIRE(!pad_tail || (out_src != src && out_dst != dst), "npi_filter_adj: pad_tail needs separate output arrays");
    const int ntiles = (int)ceil_div(E, FA_TILE);
    filter_flag_kernel<<<ntiles, 256, 0, stream>>>(src, dst, E, remap, workspace, pad_tail ? out_src : nullptr,
                                                   pad_tail ? out_dst : nullptr);
    // up to 2,048 tiles (4 M edges) the tile counts are summed inside the write kernel: two launches; above, a scan in between
    const int scanned = ntiles > 2048 ? 1 : 0;
    if (scanned) scan_small_kernel<<<1, 256, 0, stream>>>(workspace, ntiles, count);
    filter_write_kernel<<<ntiles, 256, 0, stream>>>(src, dst, E, remap, workspace, out_src, out_dst,
                                                    workspace + npi_filter_adj_newpos_offset(E), count, scanned);
    return check_launch("npi_filter_adj");
}

extern "C" int npi_readout_max_mean(const float* x, int64_t ldx, const int32_t* graph_ptr, int64_t B, int64_t F,
                                    float* out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(B >= 0 && F > 0, "npi_readout_max_mean: bad size");
    if (B == 0) return NPI_OK;
    NPI_REQUIRE(x && graph_ptr && out, "npi_readout_max_mean: null pointer");
    const int vec = pool_vec(F, {ldx}, {x});
    const int cgb = pool_groups_per_block(F, vec);
    const dim3 grid((unsigned)B, (unsigned)ceil_div(ceil_div(F, vec), cgb));
    if (vec == 4) readout_kernel<4><<<grid, POOL_THREADS, 0, stream>>>(x, ldx, graph_ptr, (int)B, (int)F, cgb, out);
    else if (vec == 2) readout_kernel<2><<<grid, POOL_THREADS, 0, stream>>>(x, ldx, graph_ptr, (int)B, (int)F, cgb, out);
    else readout_kernel<1><<<grid, POOL_THREADS, 0, stream>>>(x, ldx, graph_ptr, (int)B, (int)F, cgb, out);
    return check_launch("npi_readout_max_mean");
}

extern "C" int npi_topk_gather_bwd(const float* x, int64_t ldx, const float* score, const float* w, const int32_t* perm,
                                   int64_t n_out, int64_t F, const float* dxo, int64_t lddxo, const float* dscore_o,
                                   float* dx, int64_t lddx, float* dzv, float* dzz, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(n_out >= 0 && F > 0 && n_out < 0x7fffffff, "npi_topk_gather_bwd: bad size");
    if (n_out == 0) return NPI_OK;
    NPI_REQUIRE(x && score && w && perm && dxo && dx && dzv && dzz, "npi_topk_gather_bwd: null pointer");
    topk_gather_bwd_kernel<<<(unsigned)ceil_div(n_out, 4), 256, 0, stream>>>(x, ldx, score, w, perm, (int)n_out, (int)F, dxo,
                                                                              lddxo, dscore_o, dx, lddx, dzv, dzz);
    return check_launch("npi_topk_gather_bwd");
}

extern "C" int npi_topk_gather_bwd_ex(const float* x, int64_t ldx, const float* score, const float* w, const int32_t* remap,
                                      int64_t N, int64_t F, const float* dxo, int64_t lddxo, const float* dscore_o,
                                      float* dx, int64_t lddx, float* dzv, float* dzz, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(N >= 0 && F > 0 && N < 0x7fffffff, "npi_topk_gather_bwd_ex: bad size");
    if (N == 0) return NPI_OK;
    NPI_REQUIRE(x && score && w && remap && dxo && dx && dzv && dzz, "npi_topk_gather_bwd_ex: null pointer");
    topk_gather_bwd_all_kernel<<<(unsigned)ceil_div(N, 4), 256, 0, stream>>>(x, ldx, score, w, remap, (int)N, (int)F, dxo, lddxo,
                                                                             dscore_o, dx, lddx, dzv, dzz);
    return check_launch("npi_topk_gather_bwd_ex");
}

extern "C" int64_t npi_topk_weight_grad_workspace_elems(int64_t n_out, int64_t F) {
    return ceil_div(n_out > 0 ? n_out : 1, POOLW_ROWS) * (F + 1);
}

extern "C" int npi_topk_weight_grad(const float* x, int64_t ldx, const int32_t* perm, const float* dzv, const float* dzz,
                                    int64_t n_out, int64_t F, const float* w, float* dw, float* workspace,
                                    int64_t workspace_elems, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(n_out >= 0 && F > 0 && n_out < 0x7fffffff, "npi_topk_weight_grad: bad size");
    NPI_REQUIRE(x && perm && dzv && dzz && w && dw && workspace, "npi_topk_weight_grad: null pointer");
    if (workspace_elems < npi_topk_weight_grad_workspace_elems(n_out, F)) {
        set_error("npi_topk_weight_grad: workspace too small");
        return NPI_ERR_WORKSPACE;
    }
    const int nchunks = (int)ceil_div(n_out > 0 ? n_out : 1, POOLW_ROWS);
    const int vec = pool_vec(F, {ldx}, {x});
    const int cgb = pool_groups_per_block(F, vec);
    const dim3 grid((unsigned)nchunks, (unsigned)ceil_div(ceil_div(F, vec), cgb));
    if (vec == 4)
        topk_weight_grad_partial_kernel<4><<<grid, POOL_THREADS, 0, stream>>>(x, ldx, perm, dzv, dzz, (int)n_out, (int)F, cgb, workspace);
    else if (vec == 2)
        topk_weight_grad_partial_kernel<2><<<grid, POOL_THREADS, 0, stream>>>(x, ldx, perm, dzv, dzz, (int)n_out, (int)F, cgb, workspace);
    else
        topk_weight_grad_partial_kernel<1><<<grid, POOL_THREADS, 0, stream>>>(x, ldx, perm, dzv, dzz, (int)n_out, (int)F, cgb, workspace);
    topk_weight_grad_reduce_kernel<<<(unsigned)ceil_div(F, 64), 256, 0, stream>>>(workspace, nchunks, w, (int)F, dw);
    return check_launch("npi_topk_weight_grad");
}

extern "C" int npi_readout_max_mean_bwd(const float* x, int64_t ldx, const int32_t* graph_ptr, int64_t B, int64_t F,
                                        const float* out, const float* dout, float* dx, int64_t lddx, void* stream_) {
    return npi_readout_max_mean_bwd_ex(x, ldx, graph_ptr, B, F, out, dout, dx, lddx, -1, stream_);
}
extern "C" int npi_readout_max_mean_bwd_ex(const float* x, int64_t ldx, const int32_t* graph_ptr, int64_t B, int64_t F,
                                           const float* out, const float* dout, float* dx, int64_t lddx, int64_t N,
                                           void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(B >= 0 && F > 0 && N < 0x7fffffff, "npi_readout_max_mean_bwd: bad size");
    if (B == 0) {
        NPI_REQUIRE(N <= 0, "npi_readout_max_mean_bwd_ex: rows without a graph need B >= 1 (or zero dx yourself)");
        return NPI_OK;
    }
    NPI_REQUIRE(x && graph_ptr && out && dout && dx, "npi_readout_max_mean_bwd: null pointer");
    const int vec = pool_vec(2 * F, {ldx, lddx, F}, {x, dx, out, dout});
    const int cgb = pool_groups_per_block(F, vec);
    const dim3 grid((unsigned)B, (unsigned)ceil_div(ceil_div(F, vec), cgb));
    if (vec == 4)
        readout_bwd_kernel<4><<<grid, POOL_THREADS, 0, stream>>>(x, ldx, graph_ptr, (int)B, (int)F, cgb, out, dout, dx, lddx, (int)N);
    else if (vec == 2)
        readout_bwd_kernel<2><<<grid, POOL_THREADS, 0, stream>>>(x, ldx, graph_ptr, (int)B, (int)F, cgb, out, dout, dx, lddx, (int)N);
    else
        readout_bwd_kernel<1><<<grid, POOL_THREADS, 0, stream>>>(x, ldx, graph_ptr, (int)B, (int)F, cgb, out, dout, dx, lddx, (int)N);
    return check_launch("npi_readout_max_mean_bwd");
}

extern "C" int npi_confusion_update(const float* scores, int64_t lds, int64_t C, const int64_t* y, int64_t B,
                                    int64_t* counts, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(B >= 0 && C >= 1 && C < 0x7fffffff && lds >= C, "npi_confusion_update: bad size");
    if (B == 0) return NPI_OK;
    NPI_REQUIRE(scores && y && counts, "npi_confusion_update: null pointer");
    const int64_t blocks = ceil_div(B, 256);
    confusion_kernel<<<(unsigned)(blocks < 1024 ? blocks : 1024), 256, 0, stream>>>(
        scores, lds, (int)C, y, B, reinterpret_cast<unsigned long long*>(counts));
    return check_launch("npi_confusion_update");
}
