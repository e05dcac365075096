// GATConv (PyG 1.4.2; SURVEY.md 8(a) row a8, BASELINE.json configs[4]) on the destination-sorted CSR.
//   h = x W;  e_p = leaky_relu(<h_i, att[:C]> + <h_j, att[C:]>) for entry p = (i <- j), self loops included;
//   alpha = softmax of e over the entries of row i (exp(e - max) / (sum + 1e-16));  out_i = sum_p alpha_p h_j (+ b)
// GATConv is absent from the reference tree ("parity unpinned"); the formulas are the
// published PyG 1.4.2 ones.
//
// alpha is never stored: every kernel recomputes it from four per-node, per-head scalars
// (a_dst, a_src, row max m, row sum s), so the same numbers serve the by-target CSR (forward) and
// the by-source CSR (backward) and no per-edge array has to be permuted between the two.
// The weighted aggregation itself is segsum.hip's kernel in W_GAT_DST / W_GAT_SRC mode.
#include "segsum.h"

namespace npi {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, WAVE));
    return v;
}
__device__ __forceinline__ float lrelu_(float v, float slope) { return v > 0.f ? v : v * slope; }

// a_dst[i,h] = <h[i,h,:], att[h,:C]>, a_src[i,h] = <h[i,h,:], att[h,C:]>; one wave per node
__global__ void __launch_bounds__(256)
gat_scores_kernel(const float* __restrict__ h, int64_t ldh, const float* __restrict__ att, int N, int H, int C,
                  float* __restrict__ a_dst, float* __restrict__ a_src) {
    const int lane = lane_id();
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= N) return;
    for (int hd = 0; hd < H; ++hd) {
        const float* __restrict__ row = h + (int64_t)i * ldh + (int64_t)hd * C;
        const float* __restrict__ at = att + (int64_t)hd * 2 * C;
        float pd = 0.f, ps = 0.f;
        for (int c = lane; c < C; c += WAVE) {
            const float v = row[c];
            pd = fmaf(v, at[c], pd);
            ps = fmaf(v, at[C + c], ps);
        }
        pd = wave_sum(pd);
        ps = wave_sum(ps);
        if (lane == 0) {
            a_dst[(int64_t)i * H + hd] = pd;
            a_src[(int64_t)i * H + hd] = ps;
        }
    }
}

// D[i,h] = <a[i,h,:], b[i,h,:] - bias[h,:]>   (softmax backward: sum_p alpha_p dalpha_p = <dout_i, out_i - b>)
__global__ void __launch_bounds__(256)
gat_rowdot_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ b, int64_t ldb,
                  const float* __restrict__ bias, int N, int H, int C, float* __restrict__ D) {
    const int lane = lane_id();
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= N) return;
    for (int hd = 0; hd < H; ++hd) {
        const float* __restrict__ ra = a + (int64_t)i * lda + (int64_t)hd * C;
        const float* __restrict__ rb = b + (int64_t)i * ldb + (int64_t)hd * C;
        float p = 0.f;
        for (int c = lane; c < C; c += WAVE) p = fmaf(ra[c], rb[c] - (bias ? bias[hd * C + c] : 0.f), p);
        p = wave_sum(p);
        if (lane == 0) D[(int64_t)i * H + hd] = p;
    }
}

// 16-byte variants of the two per-node dot-product kernels (C % 4 == 0, 16-byte aligned rows): one head of a row is
// one wave instruction at C = 256, and a wave keeps ROWS_PER_WAVE rows in flight instead of one
constexpr int ROWS_PER_WAVE = 4;
__device__ __forceinline__ float dot4(const float4& a, const float4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

__global__ void __launch_bounds__(256)
gat_scores_vec_kernel(const float* __restrict__ h, int64_t ldh, const float* __restrict__ att, int N, int H, int C,
                      float* __restrict__ a_dst, float* __restrict__ a_src) {
    const int lane = lane_id();
    const int i0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS_PER_WAVE;
    if (i0 >= N) return;
    const int C4 = C >> 2;
    for (int hd = 0; hd < H; ++hd) {
        const float4* __restrict__ at = reinterpret_cast<const float4*>(att + (int64_t)hd * 2 * C);
        float pd[ROWS_PER_WAVE], ps[ROWS_PER_WAVE];
#pragma unroll
        for (int r = 0; r < ROWS_PER_WAVE; ++r) pd[r] = ps[r] = 0.f;
        for (int c = lane; c < C4; c += WAVE) {
            const float4 ad = at[c], as = at[C4 + c];
            float4 v[ROWS_PER_WAVE];
#pragma unroll
            for (int r = 0; r < ROWS_PER_WAVE; ++r)
                v[r] = *(reinterpret_cast<const float4*>(h + (int64_t)min(i0 + r, N - 1) * ldh + (int64_t)hd * C) + c);
#pragma unroll
            for (int r = 0; r < ROWS_PER_WAVE; ++r) { pd[r] += dot4(v[r], ad); ps[r] += dot4(v[r], as); }
        }
#pragma unroll
        for (int r = 0; r < ROWS_PER_WAVE; ++r) {
            const float d = wave_sum(pd[r]), sc = wave_sum(ps[r]);
            if (lane == 0 && i0 + r < N) {
                a_dst[(int64_t)(i0 + r) * H + hd] = d;
                a_src[(int64_t)(i0 + r) * H + hd] = sc;
            }
        }
    }
}

// several heads whose lane groups are powers of two (C / 4 in {8, 16, 32} lanes, H C <= 256): the whole row is ONE wave
// instruction -- lane l owns columns 4 l .. 4 l + 3 of head (4 l) / C -- and the two dot products of a head are reduced inside
// its group of lanes (the loop-over-heads kernel above keeps 16 of 64 lanes busy at C = 64)
__global__ void __launch_bounds__(256)
gat_scores_heads_kernel(const float* __restrict__ h, int64_t ldh, const float* __restrict__ att, int N, int H, int C,
                        float* __restrict__ a_dst, float* __restrict__ a_src) {
    const int lane = lane_id();
    const int i0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS_PER_WAVE;
    if (i0 >= N) return;
    const int Fw = H * C, lph = C >> 2;
    const bool on = lane * 4 < Fw;
    const int hd = on ? (lane * 4) / C : 0, cin = lane * 4 - hd * C;
    const float4 ad = on ? *reinterpret_cast<const float4*>(att + (int64_t)hd * 2 * C + cin) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 as = on ? *reinterpret_cast<const float4*>(att + (int64_t)hd * 2 * C + C + cin) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 v[ROWS_PER_WAVE];
#pragma unroll
    for (int r = 0; r < ROWS_PER_WAVE; ++r)
        v[r] = on ? *reinterpret_cast<const float4*>(h + (int64_t)min(i0 + r, N - 1) * ldh + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < ROWS_PER_WAVE; ++r) {
        float d = dot4(v[r], ad), sc = dot4(v[r], as);
        for (int off = lph >> 1; off > 0; off >>= 1) {
            d += __shfl_xor(d, off, WAVE);
            sc += __shfl_xor(sc, off, WAVE);
        }
        if (on && (lane & (lph - 1)) == 0 && i0 + r < N) {
            a_dst[(int64_t)(i0 + r) * H + hd] = d;
            a_src[(int64_t)(i0 + r) * H + hd] = sc;
        }
    }
}

__global__ void __launch_bounds__(256)
gat_rowdot_vec_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ b, int64_t ldb,
                      const float* __restrict__ bias, int N, int H, int C, float* __restrict__ D) {
    const int lane = lane_id();
    const int i0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS_PER_WAVE;
    if (i0 >= N) return;
    const int C4 = C >> 2;
    for (int hd = 0; hd < H; ++hd) {
        float p[ROWS_PER_WAVE];
#pragma unroll
        for (int r = 0; r < ROWS_PER_WAVE; ++r) p[r] = 0.f;
        for (int c = lane; c < C4; c += WAVE) {
            const float4 bs = bias ? *(reinterpret_cast<const float4*>(bias + (int64_t)hd * C) + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 va[ROWS_PER_WAVE], vb[ROWS_PER_WAVE];
#pragma unroll
            for (int r = 0; r < ROWS_PER_WAVE; ++r) {
                const int64_t i = min(i0 + r, N - 1);
                va[r] = *(reinterpret_cast<const float4*>(a + i * lda + (int64_t)hd * C) + c);
                vb[r] = *(reinterpret_cast<const float4*>(b + i * ldb + (int64_t)hd * C) + c);
            }
#pragma unroll
            for (int r = 0; r < ROWS_PER_WAVE; ++r) {
                p[r] = fmaf(va[r].x, vb[r].x - bs.x, p[r]);
                p[r] = fmaf(va[r].y, vb[r].y - bs.y, p[r]);
                p[r] = fmaf(va[r].z, vb[r].z - bs.z, p[r]);
                p[r] = fmaf(va[r].w, vb[r].w - bs.w, p[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < ROWS_PER_WAVE; ++r) {
            const float d = wave_sum(p[r]);
            if (lane == 0 && i0 + r < N) D[(int64_t)(i0 + r) * H + hd] = d;
        }
    }
}

// D[i, h] = <a[i, h, :], b[i, h, :] - bias[h, :]> AND the column sums of a, in ONE pass over a (= dOut) and b (= out): the
// backward's softmax term and GATConv's bias gradient used to stream dOut once each (gat_rowdot_vec + colsum_partial:
// 3 GB at C4), this reads 2 GB.  A workgroup takes `rows` rows (its four waves interleave groups of ROWS_PER_WAVE rows),
// lane l owns the float4 column groups l, l + 64, ... (Fw = H C <= 1024); column sums: registers -> LDS (fixed wave
// order) -> part[block, Fw], summed over the blocks in block order by slab_reduce (deterministic, no atomics).
constexpr int RDC_MAXCH = 4;
template <int NCHK>
__global__ void __launch_bounds__(256)
gat_rowdot_colsum_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ b, int64_t ldb,
                         const float* __restrict__ bias, int N, int H, int C, int rows, float* __restrict__ D,
                         float* __restrict__ part, float* __restrict__ a_masked, int64_t ldm) {
    __shared__ float red[4][NCHK * 256];
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const int Fw = H * C;
    const int rbeg = blockIdx.x * rows, rend = min(N, rbeg + rows);
    float4 cs[NCHK], bs[NCHK];
    int hd[NCHK];
#pragma unroll
    for (int k = 0; k < NCHK; ++k) {
        cs[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int c = k * 256 + lane * 4;
        const bool on = c < Fw;
        hd[k] = on ? c / C : -1;
        bs[k] = (on && bias) ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int r0 = rbeg + wave * ROWS_PER_WAVE; r0 < rend; r0 += 4 * ROWS_PER_WAVE) {
        float4 va[ROWS_PER_WAVE][NCHK], vb[ROWS_PER_WAVE][NCHK];
#pragma unroll
        for (int r = 0; r < ROWS_PER_WAVE; ++r) {
            const int64_t i = min(r0 + r, N - 1);
#pragma unroll
            for (int k = 0; k < NCHK; ++k) {
                if (hd[k] >= 0) {
                    va[r][k] = *reinterpret_cast<const float4*>(a + i * lda + k * 256 + lane * 4);
                    vb[r][k] = *reinterpret_cast<const float4*>(b + i * ldb + k * 256 + lane * 4);
                } else {
                    va[r][k] = vb[r][k] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < ROWS_PER_WAVE; ++r) {
            const bool live = r0 + r < rend;
            float pk[NCHK];
#pragma unroll
            for (int k = 0; k < NCHK; ++k) {
                float4 x = va[r][k];
                const float4 y = vb[r][k];
                if (a_masked != nullptr) {
                    // b is a ReLU OUTPUT: the gradient passes where it is positive (threshold_backward), and that masked
                    // gradient is what the rest of the backward consumes -- written here, in the pass that reads both anyway
                    x.x = y.x > 0.f ? x.x : 0.f; x.y = y.y > 0.f ? x.y : 0.f; x.z = y.z > 0.f ? x.z : 0.f; x.w = y.w > 0.f ? x.w : 0.f;
                    if (live && hd[k] >= 0) *reinterpret_cast<float4*>(a_masked + (int64_t)(r0 + r) * ldm + k * 256 + lane * 4) = x;
                }
                pk[k] = x.x * (y.x - bs[k].x) + x.y * (y.y - bs[k].y) + x.z * (y.z - bs[k].z) + x.w * (y.w - bs[k].w);
                if (live) { cs[k].x += x.x; cs[k].y += x.y; cs[k].z += x.z; cs[k].w += x.w; }
            }
            for (int h = 0; h < H; ++h) {                          // (one head: a single wave reduction per row)
                float p = 0.f;
#pragma unroll
                for (int k = 0; k < NCHK; ++k) p += (hd[k] == h) ? pk[k] : 0.f;
                p = wave_sum(p);
                if (lane == 0 && live) D[(int64_t)(r0 + r) * H + h] = p;
            }
        }
    }
    if (part == nullptr) return;
#pragma unroll
    for (int k = 0; k < NCHK; ++k) *reinterpret_cast<float4*>(&red[wave][k * 256 + lane * 4]) = cs[k];
    __syncthreads();
    for (int c = threadIdx.x; c < Fw; c += 256)
        part[(int64_t)blockIdx.x * Fw + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

// out[c] = sum over the blocks of part[k, c]: 64 columns per workgroup, its four waves take the block ranges
// [0, n/4), [n/4, n/2), ... with four independent partial sums each (k mod 4), folded in a fixed order
__global__ void __launch_bounds__(256)
colsum_blocks_kernel(const float* __restrict__ part, int nblocks, int Fw, float* __restrict__ out) {
    __shared__ float red[4][64];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int per = (nblocks + 3) / 4;
    const int kb = wave * per, ke = min(nblocks, kb + per);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < Fw) {
        int k = kb;
        for (; k + 4 <= ke; k += 4) {
            s0 += part[(int64_t)k * Fw + c];
            s1 += part[(int64_t)(k + 1) * Fw + c];
            s2 += part[(int64_t)(k + 2) * Fw + c];
            s3 += part[(int64_t)(k + 3) * Fw + c];
        }
        for (; k < ke; ++k) s0 += part[(int64_t)k * Fw + c];
    }
    red[wave][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wave == 0 && c < Fw) out[c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

static bool rows16(const void* p, int64_t ld, int64_t C) { return C % 4 == 0 && ld % 4 == 0 && ((uintptr_t)p % 16) == 0; }

// ---- segment softmax statistics: m[i,h] = max_p e_p, s[i,h] = sum_p exp(e_p - m) -------------------
constexpr int GAT_HEAVY = 4096;      // rows longer than this go to the workgroup-per-row kernel

// LG lanes per row, WAVE / LG rows per wave: the typical row of these graphs has ~20 entries (an ncRNA with its
// partners and its self loop), a third of a wavefront.  Rows much longer than the group (a protein) are then taken
// by the whole wave, one after the other; rows above GAT_HEAVY belong to the segment kernel.
template <int LG> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int off = LG / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
}
template <int LG> __device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int off = LG / 2; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, WAVE));
    return v;
}
constexpr int GROUP_WIDE = 8;        // a row longer than GROUP_WIDE * LG entries is worth the whole wave

static int row_group_lanes(int64_t nnz_max, int64_t N) {
    const int64_t avg = nnz_max / (N > 0 ? N : 1);
    return avg <= 12 ? 8 : avg <= 24 ? 16 : avg <= 48 ? 32 : 64;
}

// (max, sum of exp(. - max)) of row i, head hd, over entries [b, e) with lanes gl, gl + LG, ...
template <int LG>
__device__ __forceinline__ void softmax_row(const int32_t* __restrict__ col, const float* __restrict__ a_src, int H, int hd,
                                            float ad, int b, int e, int gl, float slope, float& mx, float& sum) {
    // the lane's first entry stays in a register: rows that fit the group (most of them) gather a_src once, not twice
    const bool has0 = b + gl < e;
    const float z0 = has0 ? lrelu_(ad + a_src[(int64_t)col[b + gl] * H + hd], slope) : -3.0e38f;
    mx = z0;
    for (int p = b + gl + LG; p < e; p += LG) mx = fmaxf(mx, lrelu_(ad + a_src[(int64_t)col[p] * H + hd], slope));
    mx = group_max<LG>(mx);
    if (e == b) mx = 0.f;
    sum = has0 ? expf(z0 - mx) : 0.f;
    for (int p = b + gl + LG; p < e; p += LG) sum += expf(lrelu_(ad + a_src[(int64_t)col[p] * H + hd], slope) - mx);
    sum = group_sum<LG>(sum);
}

template <int LG>
__global__ void __launch_bounds__(256)
gat_softmax_rows_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                        const float* __restrict__ a_dst, const float* __restrict__ a_src, int N, int H,
                        float slope, float* __restrict__ m, float* __restrict__ s) {
    constexpr int G = WAVE / LG;
    const int lane = lane_id();
    const int grp = lane / LG, gl = lane % LG;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * G;
    if (row0 >= N) return;
    const int i = row0 + grp;
    const bool have = i < N;
    const int b = have ? rowptr[i] : 0, e = have ? rowptr[i + 1] : 0;
    const bool heavy = e - b > GAT_HEAVY;
    const bool wide = LG < WAVE && !heavy && e - b > GROUP_WIDE * LG;
    if (have && !heavy && !wide) {
        for (int hd = 0; hd < H; ++hd) {
            float mx, sum;
            softmax_row<LG>(col, a_src, H, hd, a_dst[(int64_t)i * H + hd], b, e, gl, slope, mx, sum);
            if (gl == 0) {
                m[(int64_t)i * H + hd] = mx;
                s[(int64_t)i * H + hd] = sum;
            }
        }
    }
    if (LG < WAVE) {
        uint64_t todo = __ballot(wide && gl == 0);
        while (todo) {                                     // wave-uniform
            const int from = __ffsll((unsigned long long)todo) - 1;
            todo &= todo - 1;
            const int wb = __shfl(b, from, WAVE), we = __shfl(e, from, WAVE), wi = __shfl(i, from, WAVE);
            for (int hd = 0; hd < H; ++hd) {
                float mx, sum;
                softmax_row<WAVE>(col, a_src, H, hd, a_dst[(int64_t)wi * H + hd], wb, we, lane, slope, mx, sum);
                if (lane == 0) {
                    m[(int64_t)wi * H + hd] = mx;
                    s[(int64_t)wi * H + hd] = sum;
                }
            }
        }
    }
}

// Heavy rows (hub proteins: 400k entries at C4, 2M at C5) are cut into SEGMENTS of HEAVY_SEG_ITEMS items, one
// 1024-thread workgroup per segment, so a hub row is spread over dozens of CUs instead of serialising on one.
// There is one workgroup per item; it takes (a) the first segment of the heavy row that STARTS in its item (at
// most one: a heavy row is longer than an item) and (b) the continuing segment that starts at its first entry, if
// the row passing through is heavy and (item - first_item(row)) is a multiple of HEAVY_SEG_ITEMS.  A segment's
// partial goes to workspace slot 2 item + (0: first segment, 1: continuing); the workgroup that finishes LAST
// (per-row counter, indexed by the row's first item) folds the partials in segment order -- a fixed order, so
// the result does not depend on which workgroup that is.
constexpr int HEAVY_THREADS = 1024;
constexpr int HEAVY_WAVES = HEAVY_THREADS / WAVE;
constexpr int HEAVY_SEG_ITEMS = 64;
constexpr int HEAVY_GRID = 2048;     // workgroups walking the segment list (8 per CU)

struct HeavySeg {
    int row, b, e;      // row and entry range of the segment
    int fi, k, nseg;    // first item of the row, index of this segment, segments of the row
};

// segments this workgroup (item) owns: returns a bit mask (1: first segment in seg[0], 2: continuing in seg[1])
__device__ __forceinline__ int heavy_segments(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ item_row,
                                              int N, int n_items, int item, int IE, HeavySeg (&seg)[2]) {
    const int nnz = rowptr[N];
    const int64_t k0 = (int64_t)item * IE, k1 = k0 + IE;
    int mask = 0;
    if (k0 >= nnz) return 0;
    const int span = HEAVY_SEG_ITEMS * IE;
    // (b) the row that contains entry k0 and started before it
    const int r0 = item_row[item];
    const int b0 = rowptr[r0];
    if (b0 < k0) {
        const int e0 = rowptr[r0 + 1];
        if (e0 - b0 > GAT_HEAVY) {
            const int fi = b0 / IE;
            if ((item - fi) % HEAVY_SEG_ITEMS == 0) {
                seg[1].row = r0; seg[1].b = (int)k0; seg[1].e = (int)min((int64_t)e0, k0 + span);
                seg[1].fi = fi; seg[1].k = (item - fi) / HEAVY_SEG_ITEMS;
                seg[1].nseg = ((e0 - 1) / IE - fi) / HEAVY_SEG_ITEMS + 1;
                mask |= 2;
            }
        }
    }
    // (a) the row that starts in this item and reaches past its end
    if (k1 < nnz && item + 1 <= n_items) {
        const int r1 = item_row[item + 1];                 // row holding entry k1
        const int b1 = rowptr[r1];
        if (b1 >= k0 && b1 < k1) {
            const int e1 = rowptr[r1 + 1];
            if (e1 - b1 > GAT_HEAVY) {
                seg[0].row = r1; seg[0].b = b1; seg[0].e = (int)min((int64_t)e1, k0 + span);
                seg[0].fi = item; seg[0].k = 0;
                seg[0].nseg = ((e1 - 1) / IE - item) / HEAVY_SEG_ITEMS + 1;
                mask |= 1;
            }
        }
    }
    return mask;
}
__device__ __forceinline__ int heavy_slot(const HeavySeg& g, int k) {          // workspace slot of segment k of g's row
    return k == 0 ? 2 * g.fi : 2 * (g.fi + k * HEAVY_SEG_ITEMS) + 1;
}

// The heavy segments of a CSR are few (a row needs > GAT_HEAVY entries): a scout launch -- one THREAD per item --
// lists them as item * 2 + q, and the 1024-thread kernels run over that list instead of over every item
// (82 k mostly idle workgroups at C4 before: 0.3-0.5 ms per launch of pure dispatch).  The order of the list is
// whatever the atomics make it; it does not matter: every segment writes its own slot and the fold order is fixed.
__host__ __device__ inline int64_t heavy_list_capacity(int64_t nnz_max) {
    const int64_t ie = item_edges_for(nnz_max);
    const int64_t items = (nnz_max + ie - 1) / ie;
    const int64_t bound = nnz_max / (HEAVY_SEG_ITEMS * ie) + 2 * (nnz_max / GAT_HEAVY) + 4;   // sum over heavy rows of (len / span + 2)
    return bound < 2 * items ? bound : 2 * items;
}
__global__ void __launch_bounds__(256)
heavy_list_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ item_row, int N, int n_items, int item_edges,
                  int capacity, int* __restrict__ cnt, int* __restrict__ count, int* __restrict__ list) {
    const int item = blockIdx.x * 256 + threadIdx.x;
    if (item >= n_items) return;
    cnt[item] = 0;                                         // the per-row arrival counters (indexed by first item)
    HeavySeg seg[2];
    const int mask = heavy_segments(rowptr, item_row, N, n_items, item, item_edges, seg);
    for (int q = 0; q < 2; ++q) {
        if (!(mask & (1 << q))) continue;
        const int at = atomicAdd(count, 1);
        if (at < capacity) list[at] = item * 2 + q;
    }
}
// the segment workgroup `w` of the list owns (workgroup-uniform); false: nothing to do
__device__ __forceinline__ bool heavy_take(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ item_row, int N,
                                           int n_items, int item_edges, const int* __restrict__ count,
                                           const int* __restrict__ list, int w, HeavySeg& g) {
    if (w >= *count) return false;
    const int code = list[w];
    HeavySeg seg[2];
    const int mask = heavy_segments(rowptr, item_row, N, n_items, code >> 1, item_edges, seg);
    if (!(mask & (1 << (code & 1)))) return false;         // cannot happen: the scout saw the same CSR
    g = seg[code & 1];
    return true;
}

// fixed-order sum / max of one value per wave
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    if (lane_id() == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < HEAVY_WAVES; ++w) s += red[w];
    __syncthreads();
    return s;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    if (lane_id() == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = red[0];
#pragma unroll
    for (int w = 1; w < HEAVY_WAVES; ++w) s = fmaxf(s, red[w]);
    __syncthreads();
    return s;
}
// partials written by OTHER workgroups are read past this CU's L1
__device__ __forceinline__ float ld_agent(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// thread 0 publishes the segment's partial and learns whether its workgroup is the last one of the row
__device__ __forceinline__ bool heavy_arrive(int* cnt, const HeavySeg& g, int* flag) {
    if (threadIdx.x == 0) {
        __threadfence();                                   // the partial is visible before the count
        *flag = (atomicAdd(cnt + g.fi, 1) == g.nseg - 1);
        __threadfence();
    }
    __syncthreads();
    const bool last = *flag != 0;
    __syncthreads();
    return last;
}

// part[slot][H][2] = (max, sum of exp(. - max)) of a segment; the last workgroup merges them like an online softmax
__global__ void __launch_bounds__(HEAVY_THREADS)
gat_softmax_heavy_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                         const int32_t* __restrict__ item_row, const float* __restrict__ a_dst,
                         const float* __restrict__ a_src, int N, int n_items, int H, float slope,
                         float* __restrict__ m, float* __restrict__ s, int item_edges,
                         float* __restrict__ part, int* __restrict__ cnt, const int* __restrict__ count,
                         const int* __restrict__ list) {
    __shared__ float red[HEAVY_WAVES];
    __shared__ int flag;
    const int t = threadIdx.x;
    HeavySeg g;
    for (int w = blockIdx.x; heavy_take(rowptr, item_row, N, n_items, item_edges, count, list, w, g); w += gridDim.x) {   // workgroup-uniform
        const int i = g.row;
        float* __restrict__ mine = part + (int64_t)heavy_slot(g, g.k) * H * 2;
        for (int hd = 0; hd < H; ++hd) {
            const float ad = a_dst[(int64_t)i * H + hd];
            float mx = -3.0e38f;
            for (int p = g.b + t; p < g.e; p += HEAVY_THREADS) mx = fmaxf(mx, lrelu_(ad + a_src[(int64_t)col[p] * H + hd], slope));
            mx = block_max(mx, red);
            float sum = 0.f;
            for (int p = g.b + t; p < g.e; p += HEAVY_THREADS) sum += expf(lrelu_(ad + a_src[(int64_t)col[p] * H + hd], slope) - mx);
            sum = block_sum(sum, red);
            if (t == 0) { mine[hd * 2] = mx; mine[hd * 2 + 1] = sum; }
        }
        if (heavy_arrive(cnt, g, &flag) && t < H) {
            float M = -3.0e38f;
            for (int k = 0; k < g.nseg; ++k) M = fmaxf(M, ld_agent(part + ((int64_t)heavy_slot(g, k) * H + t) * 2));
            float S = 0.f;
            for (int k = 0; k < g.nseg; ++k) {
                const float* pk = part + ((int64_t)heavy_slot(g, k) * H + t) * 2;
                S += ld_agent(pk + 1) * expf(ld_agent(pk) - M);
            }
            m[(int64_t)i * H + t] = M;
            s[(int64_t)i * H + t] = S;
        }
    }
}

// row sums of per-entry scalars for the heavy rows (see seg_rowsum_scalar_kernel below for the rest)
__global__ void __launch_bounds__(HEAVY_THREADS)
seg_rowsum_heavy_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ item_row,
                        const float* __restrict__ vals, const int32_t* __restrict__ map, int N, int n_items, int H,
                        float* __restrict__ out, int item_edges, float* __restrict__ part, int* __restrict__ cnt,
                        const int* __restrict__ count, const int* __restrict__ list) {
    __shared__ float red[HEAVY_WAVES];
    __shared__ int flag;
    const int t = threadIdx.x;
    HeavySeg g;
    for (int w = blockIdx.x; heavy_take(rowptr, item_row, N, n_items, item_edges, count, list, w, g); w += gridDim.x) {
        float* __restrict__ mine = part + (int64_t)heavy_slot(g, g.k) * H;
        for (int hd = 0; hd < H; ++hd) {
            float sum = 0.f;
            for (int p = g.b + t; p < g.e; p += HEAVY_THREADS) {
                const int64_t idx = map ? map[p] : p;
                sum += vals[idx * H + hd];
            }
            sum = block_sum(sum, red);
            if (t == 0) mine[hd] = sum;
        }
        if (heavy_arrive(cnt, g, &flag) && t < H) {
            float S = 0.f;
            for (int k = 0; k < g.nseg; ++k) S += ld_agent(part + (int64_t)heavy_slot(g, k) * H + t);
            out[(int64_t)g.row * H + t] = S;
        }
    }
}

// ---- backward: per-entry score gradient over the by-target CSR ------------------------------------------
//   dalpha_p = <dout_i[h], hfeat_j[h]>;  de_p = alpha_p (dalpha_p - D_i);  dz_p = de_p * lrelu'(z_p)
// one wave per 256-entry item; dout_i is reloaded when the row changes, hfeat_j gathered per entry
constexpr int EDGE_HMAX = 8;        // heads whose per-entry dots are staged through LDS
template <int NCH>
__global__ void __launch_bounds__(256)
gat_edge_grad_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                     const int32_t* __restrict__ rowidx, int N, int n_items, int item_edges,
                     const float* __restrict__ hfeat, int64_t ldh, const float* __restrict__ dout, int64_t ldd,
                     int H, int C, const float* __restrict__ a_dst, const float* __restrict__ a_src,
                     const float* __restrict__ m, const float* __restrict__ s, const float* __restrict__ D,
                     float slope, float* __restrict__ dz, float* __restrict__ alpha_out,
                     const float* __restrict__ hfeat2, int split, int swap) {
    // swap == 0: rows are TARGETS (dout rows, a_dst / m / s / D by row), columns SOURCES (hfeat gathered, a_src by column).
    // swap != 0: the same edges seen from a by-source CSR -- rows are sources (`dout` holds their hfeat rows, a_src by
    //            row), columns targets (`hfeat` holds the gathered dOut rows; a_dst / m / s / D by column).
    // hfeat2 / split: two-part gathered table as in segsum (column >= split reads row column - split of hfeat2).
    constexpr int U = (NCH <= 2) ? 4 : 2;
    const float* __restrict__ hf2 = reinterpret_cast<const float*>(reinterpret_cast<uintptr_t>(hfeat2) - (uint64_t)split * (uint64_t)ldh * sizeof(float));
    const int lane = lane_id();
    const int item = uniform_i(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (item >= n_items) return;
    const int nnz = rowptr[N];
    const int k0 = item * item_edges;
    if (k0 >= nnz) return;
    const int k1 = min(k0 + item_edges, nnz);
    const int F = H * C;
    bool act[NCH];
    int foff[NCH], hd[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        foff[c] = (c * WAVE + lane) * 4;
        act[c] = foff[c] < F;
        hd[c] = act[c] ? foff[c] / C : -1;
    }
    float4 dr[NCH];
    int cur = -1;
    // per-entry dot products of one 64-entry block are parked in LDS, then all 64 lanes turn them
    // into dz together (exp, loads and the store leave the serial per-entry chain)
    __shared__ float pbuf[4][WAVE * EDGE_HMAX];
    float* __restrict__ pb = pbuf[threadIdx.x >> 6];
    const bool staged = H <= EDGE_HMAX;
    for (int kb = k0; kb < k1; kb += WAVE) {
        const int nb = min(WAVE, k1 - kb);
        const int cv = (lane < nb) ? col[kb + lane] : 0;
        const int rv = (lane < nb) ? rowidx[kb + lane] : 0;
        if (NCH == 1 && H == 1) {
            // One head, one chunk: 8 entries at a time.  Each lane has the partial dot product of its 4 columns for
            // each of the 8 entries; instead of 8 full wave reductions (6 cross-lane steps each) the 8 values are
            // reduce-SCATTERED -- every xor step halves the number of entries a lane still carries -- 4 + 2 + 1 + 3
            // = 10 cross-lane steps, after which lane group l >> 3 holds the sum of entry (l >> 3).
            const bool b5 = (lane & 32) != 0, b4 = (lane & 16) != 0, b3 = (lane & 8) != 0;
            const int e_of_lane = (b5 ? 4 : 0) + (b4 ? 2 : 0) + (b3 ? 1 : 0);
            for (int j = 0; j < nb; j += 8) {
                float4 hv8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int cu = bcast_i(cv, min(j + u, nb - 1));
                    hv8[u] = act[0] ? *reinterpret_cast<const float4*>((cu < split ? hfeat : hf2) + (int64_t)cu * ldh + foff[0])
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                float p[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    p[u] = 0.f;
                    if (j + u < nb) {                          // wave-uniform
                        const int i = bcast_i(rv, j + u);
                        if (i != cur) {
                            cur = i;
                            dr[0] = act[0] ? *reinterpret_cast<const float4*>(dout + (int64_t)i * ldd + foff[0])
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
                        }
                        p[u] = dr[0].x * hv8[u].x + dr[0].y * hv8[u].y + dr[0].z * hv8[u].z + dr[0].w * hv8[u].w;
                    }
                }
                float w4[4], w2[2];
#pragma unroll
                for (int k = 0; k < 4; ++k) w4[k] = (b5 ? p[k + 4] : p[k]) + __shfl_xor(b5 ? p[k] : p[k + 4], 32, WAVE);
#pragma unroll
                for (int k = 0; k < 2; ++k) w2[k] = (b4 ? w4[k + 2] : w4[k]) + __shfl_xor(b4 ? w4[k] : w4[k + 2], 16, WAVE);
                float y = (b3 ? w2[1] : w2[0]) + __shfl_xor(b3 ? w2[0] : w2[1], 8, WAVE);
                y += __shfl_xor(y, 4, WAVE);
                y += __shfl_xor(y, 2, WAVE);
                y += __shfl_xor(y, 1, WAVE);
                if ((lane & 7) == 0 && j + e_of_lane < nb) pb[j + e_of_lane] = y;
            }
        } else
        for (int j = 0; j < nb; j += U) {
            float4 hv[U][NCH];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int cu = bcast_i(cv, min(j + u, nb - 1));
#pragma unroll
                for (int c = 0; c < NCH; ++c)
                    hv[u][c] = act[c] ? *reinterpret_cast<const float4*>((cu < split ? hfeat : hf2) + (int64_t)cu * ldh + foff[c])
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (j + u >= nb) break;
                const int i = bcast_i(rv, j + u);
                const int cu = bcast_i(cv, j + u);
                if (i != cur) {
                    cur = i;
#pragma unroll
                    for (int c = 0; c < NCH; ++c)
                        dr[c] = act[c] ? *reinterpret_cast<const float4*>(dout + (int64_t)i * ldd + foff[c])
                                       : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                for (int h = 0; h < H; ++h) {
                    float p = 0.f;
#pragma unroll
                    for (int c = 0; c < NCH; ++c)
                        if (hd[c] == h)
                            p += dr[c].x * hv[u][c].x + dr[c].y * hv[u][c].y + dr[c].z * hv[u][c].z + dr[c].w * hv[u][c].w;
                    p = wave_sum(p);
                    if (staged) {
                        if (lane == 0) pb[(j + u) * H + h] = p;
                    } else if (lane == 0) {
                        const int64_t ii = (int64_t)(swap ? cu : i) * H + h;
                        const float z = a_dst[ii] + a_src[(int64_t)(swap ? i : cu) * H + h];
                        const float alpha = expf(lrelu_(z, slope) - m[ii]) / (s[ii] + 1e-16f);
                        const float de = alpha * (p - D[ii]);
                        dz[(int64_t)(kb + j + u) * H + h] = de * (z > 0.f ? 1.f : slope);
                        if (alpha_out) alpha_out[(int64_t)(kb + j + u) * H + h] = alpha;
                    }
                }
            }
        }
        if (staged) {
            __builtin_amdgcn_wave_barrier();
            for (int t0 = 0; t0 < nb * H; t0 += WAVE) {        // uniform trip count: the shuffles need every lane
                const int t = t0 + lane;
                const int en = min(t / H, nb - 1), h = t - (t / H) * H;
                const int i = __shfl(rv, en, WAVE), cu = __shfl(cv, en, WAVE);
                if (t >= nb * H) continue;
                const int64_t ii = (int64_t)(swap ? cu : i) * H + h;
                const float z = a_dst[ii] + a_src[(int64_t)(swap ? i : cu) * H + h];
                const float alpha = expf(lrelu_(z, slope) - m[ii]) / (s[ii] + 1e-16f);
                const float de = alpha * (pb[t] - D[ii]);
                dz[(int64_t)kb * H + t] = de * (z > 0.f ? 1.f : slope);
                if (alpha_out) alpha_out[(int64_t)kb * H + t] = alpha;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// out[r,h] = sum over the entries p of row r of vals[idx(p), h], idx = map ? map[p] : p; LG lanes per row like
// gat_softmax_rows_kernel
template <int LG>
__device__ __forceinline__ float rowsum_row(const float* __restrict__ vals, const int32_t* __restrict__ map, int H, int hd,
                                            int b, int e, int gl) {
    float sum = 0.f;
    for (int p = b + gl; p < e; p += LG) {
        const int64_t q = map ? map[p] : p;
        sum += vals[q * H + hd];
    }
    return group_sum<LG>(sum);
}

template <int LG>
__global__ void __launch_bounds__(256)
seg_rowsum_scalar_kernel(const int32_t* __restrict__ rowptr, const float* __restrict__ vals,
                         const int32_t* __restrict__ map, int N, int H, float* __restrict__ out, int skip_heavy) {
    constexpr int G = WAVE / LG;
    const int lane = lane_id();
    const int grp = lane / LG, gl = lane % LG;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * G;
    if (row0 >= N) return;
    const int r = row0 + grp;
    const bool have = r < N;
    const int b = have ? rowptr[r] : 0, e = have ? rowptr[r + 1] : 0;
    const bool heavy = skip_heavy && e - b > GAT_HEAVY;    // seg_rowsum_heavy_kernel owns it
    const bool wide = LG < WAVE && !heavy && e - b > GROUP_WIDE * LG;
    if (have && !heavy && !wide) {
        for (int hd = 0; hd < H; ++hd) {
            const float sum = rowsum_row<LG>(vals, map, H, hd, b, e, gl);
            if (gl == 0) out[(int64_t)r * H + hd] = sum;
        }
    }
    if (LG < WAVE) {
        uint64_t todo = __ballot(wide && gl == 0);
        while (todo) {
            const int from = __ffsll((unsigned long long)todo) - 1;
            todo &= todo - 1;
            const int wb = __shfl(b, from, WAVE), we = __shfl(e, from, WAVE), wr = __shfl(r, from, WAVE);
            for (int hd = 0; hd < H; ++hd) {
                const float sum = rowsum_row<WAVE>(vals, map, H, hd, wb, we, lane);
                if (lane == 0) out[(int64_t)wr * H + hd] = sum;
            }
        }
    }
}

// map[q] = position in the by-target CSR of by-source entry q (same directed edge; loops map to loops)
__global__ void entry_transpose_map_kernel(const int32_t* __restrict__ src_eid, const int32_t* __restrict__ src_rowidx,
                                           const int32_t* __restrict__ src_rowptr, const int32_t* __restrict__ dst_rowptr,
                                           const int32_t* __restrict__ pos_dst_of_edge, int N, int64_t nnz_max,
                                           int32_t* __restrict__ map) {
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nnz_max || q >= src_rowptr[N]) return;
    const int e = src_eid[q];
    map[q] = e >= 0 ? pos_dst_of_edge[e] : dst_rowptr[src_rowidx[q] + 1] - 1;
}

// datt partials: part[chunk][0][h*C+c] = sum_i g_dst[i,h] hfeat[i,h*C+c], part[chunk][1][..] with g_src
constexpr int ATT_ROWS = 256;       // 3,906 workgroups at C4: enough loads in flight to stream hfeat
__global__ void __launch_bounds__(256)
gat_att_grad_partial_kernel(const float* __restrict__ hfeat, int64_t ldh, const float* __restrict__ g_dst,
                            const float* __restrict__ g_src, int N, int H, int C, float* __restrict__ part) {
    const int F = H * C;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= F) return;
    const int hd = c / C;
    const int rbeg = blockIdx.y * ATT_ROWS, rend = min(N, rbeg + ATT_ROWS);
    float sd = 0.f, ss = 0.f;
#pragma unroll 8
    for (int i = rbeg; i < rend; ++i) {
        const float v = hfeat[(int64_t)i * ldh + c];
        sd = fmaf(g_dst[(int64_t)i * H + hd], v, sd);
        ss = fmaf(g_src[(int64_t)i * H + hd], v, ss);
    }
    part[((int64_t)blockIdx.y * 2 + 0) * F + c] = sd;
    part[((int64_t)blockIdx.y * 2 + 1) * F + c] = ss;
}
// 32 columns x 32 chunk lanes per workgroup; the lanes are folded in a fixed order (deterministic)
__global__ void __launch_bounds__(1024)
gat_att_grad_reduce_kernel(const float* __restrict__ part, int nchunks, int H, int C, float* __restrict__ datt) {
    __shared__ float red[32][33];
    const int F = H * C;
    const int cl = threadIdx.x & 31, zl = threadIdx.x >> 5;
    const int idx = blockIdx.x * 32 + cl;                      // over 2*F
    float sum = 0.f;
    if (idx < 2 * F) {
        const int which = idx / F, c = idx % F;
        for (int z = zl; z < nchunks; z += 32) sum += part[((int64_t)z * 2 + which) * F + c];
    }
    red[zl][cl] = sum;
    __syncthreads();
    if (zl == 0 && idx < 2 * F) {
        float tot = 0.f;
        for (int z = 0; z < 32; ++z) tot += red[z][cl];
        const int which = idx / F, c = idx % F;
        const int hd = c / C, cc = c % C;
        datt[(int64_t)hd * 2 * C + which * C + cc] = tot;
    }
}

// dh[i, h C + c] += g_dst[i, h] att[h, c] + g_src[i, h] att[h, C + c]: the rank-1 terms of the attention-score gradient,
// added once g_dst / g_src are complete (the fused backward pass cannot apply them in its row epilogue)
__global__ void __launch_bounds__(256)
gat_rank1_add_kernel(float* __restrict__ dh, int64_t ld, const float* __restrict__ g_dst, const float* __restrict__ g_src,
                     const float* __restrict__ att, int64_t N, int H, int C) {
    const int F = H * C;
    const int64_t idx = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (idx >= N * F) return;
    const int64_t i = idx / F;
    const int f = (int)(idx - i * F);                  // F % 4 == 0 and C % 4 == 0: the four columns share a head
    const int h = f / C, c = f - h * C;
    const float gd = g_dst[i * H + h], gs = g_src[i * H + h];
    const float4 ad = *reinterpret_cast<const float4*>(att + (int64_t)h * 2 * C + c);
    const float4 as = *reinterpret_cast<const float4*>(att + (int64_t)h * 2 * C + C + c);
    float4 v = *reinterpret_cast<float4*>(dh + i * ld + f);
    v.x += gd * ad.x + gs * as.x; v.y += gd * ad.y + gs * as.y;
    v.z += gd * ad.z + gs * as.z; v.w += gd * ad.w + gs * as.w;
    *reinterpret_cast<float4*>(dh + i * ld + f) = v;
}

}  // namespace npi

using namespace npi;

extern "C" int npi_gat_backward_fused(const int32_t* rowptr, const int32_t* col, const int32_t* rowidx, const int32_t* item_row,
                                      int64_t N, int64_t nnz_max, const float* dout, int64_t ldd, const float* hfeat, int64_t ldh,
                                      float* out, int64_t ldo, int64_t C, const float* a_dst, const float* a_src, const float* D,
                                      float slope, const float* alpha, const int32_t* alpha_map, float* dz, float* carry,
                                      void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(N >= 0 && nnz_max > 0 && C > 0 && C <= 256 && C % 4 == 0, "npi_gat_backward_fused: needs one head of <= 256 channels, C % 4 == 0");
    if (N == 0) return NPI_OK;
    NPI_REQUIRE(rowptr && col && rowidx && item_row && dout && hfeat && out && a_dst && a_src && D && alpha && alpha_map && dz && carry,
                "npi_gat_backward_fused: null pointer");
    NPI_REQUIRE(ldd >= C && ldh >= C && ldo >= C && ldh % 4 == 0 && ((uintptr_t)hfeat % 16) == 0,
                "npi_gat_backward_fused: leading dimension / alignment");
    SegParams P{};
    P.rowptr = rowptr; P.col = col; P.item_row = item_row;
    P.N = (int)N; P.n_items = (int)npi_num_items(nnz_max);
    P.x = dout; P.ldx = ldd; P.out = out; P.ldo = ldo; P.F = (int)C;
    P.carry = carry; P.w = alpha; P.wmap = alpha_map; P.bias = nullptr;
    P.H = 1; P.C = (int)C; P.a_dst = a_dst; P.a_src = a_src; P.m = a_dst; P.s = a_dst; P.slope = slope;   // m, s unused in this mode
    P.hrow = hfeat; P.ldh = ldh; P.Dt = D; P.rowidx = rowidx; P.dz_out = dz;
    return segsum_run(P, W_GAT_SRC_FUSED, 0, nnz_max, NPI_F32, stream);
}

__global__ void __launch_bounds__(256)
gat_pack_targets_kernel(const float* __restrict__ a_dst, const float* __restrict__ m, const float* __restrict__ s,
                        const float* __restrict__ D, int N, float4* __restrict__ t) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < N) t[i] = make_float4(a_dst[i], m[i], 1.f / (s[i] + 1e-16f), D[i]);
}

extern "C" int npi_gat_pack_targets(const float* a_dst, const float* m, const float* s, const float* D, int64_t N,
                                    float* tpack, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(N >= 0 && N < 0x7fffffff, "npi_gat_pack_targets: bad size");
    if (N == 0) return NPI_OK;
    NPI_REQUIRE(a_dst && m && s && D && tpack && ((uintptr_t)tpack % 16) == 0, "npi_gat_pack_targets: null or misaligned pointer");
    gat_pack_targets_kernel<<<(unsigned)ceil_div(N, 256), 256, 0, stream>>>(a_dst, m, s, D, (int)N, reinterpret_cast<float4*>(tpack));
    return check_launch("npi_gat_pack_targets");
}

extern "C" int npi_gat_backward_fused_packed(const int32_t* rowptr, const int32_t* col, const int32_t* rowidx,
                                             const int32_t* item_row, int64_t N, int64_t nnz_max, const float* dout, int64_t ldd,
                                             const float* hfeat, int64_t ldh, float* out, int64_t ldo, int64_t C,
                                             const float* tpack, const float* a_src, float slope, float* dz, float* carry,
                                             void* stream_) {
    return npi_gat_backward_fused_packed_ex(rowptr, col, rowidx, item_row, N, nnz_max, dout, ldd, nullptr, 0, hfeat, ldh, out, ldo,
                                            C, tpack, a_src, slope, dz, carry, stream_);
}

extern "C" int npi_gat_backward_fused_packed_ex(const int32_t* rowptr, const int32_t* col, const int32_t* rowidx,
                                                const int32_t* item_row, int64_t N, int64_t nnz_max, const float* dout,
                                                int64_t ldd, const float* dout2, int64_t split, const float* hfeat, int64_t ldh,
                                                float* out, int64_t ldo, int64_t C, const float* tpack, const float* a_src,
                                                float slope, float* dz, float* carry, void* stream_) {
    return npi_gat_backward_fused_heads(rowptr, col, rowidx, item_row, N, nnz_max, dout, ldd, dout2, split, hfeat, ldh, out, ldo, 1, C,
                                        tpack, a_src, slope, dz, carry, stream_);
}

extern "C" int npi_gat_backward_fused_heads(const int32_t* rowptr, const int32_t* col, const int32_t* rowidx,
                                            const int32_t* item_row, int64_t N, int64_t nnz_max, const float* dout,
                                            int64_t ldd, const float* dout2, int64_t split, const float* hfeat, int64_t ldh,
                                            float* out, int64_t ldo, int64_t H, int64_t C, const float* tpack, const float* a_src,
                                            float slope, float* dz, float* carry, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t F = H * C;
    NPI_REQUIRE(N >= 0 && nnz_max > 0 && C > 0 && C % 4 == 0 && F <= 256, "npi_gat_backward_fused: needs heads * out_channels <= 256, out_channels % 4 == 0");
    NPI_REQUIRE(H == 1 || ((H == 2 || H == 4 || H == 8) && C >= 32 && (C & (C - 1)) == 0),
                "npi_gat_backward_fused_heads: several heads need 2 / 4 / 8 heads of 32 / 64 / 128 channels");
    NPI_REQUIRE(dout2 == nullptr || (split >= 0 && split < 0x7fffffff), "npi_gat_backward_fused_packed_ex: bad split");
    if (N == 0) return NPI_OK;
    NPI_REQUIRE(rowptr && col && rowidx && item_row && dout && hfeat && out && tpack && a_src && dz && carry,
                "npi_gat_backward_fused_packed: null pointer");
    NPI_REQUIRE(ldd >= F && ldh >= F && ldo >= F && ldh % 4 == 0 && ((uintptr_t)hfeat % 16) == 0 && ((uintptr_t)tpack % 16) == 0,
                "npi_gat_backward_fused_packed: leading dimension / alignment");
    SegParams P{};
    P.rowptr = rowptr; P.col = col; P.item_row = item_row;
    P.N = (int)N; P.n_items = (int)npi_num_items(nnz_max);
    P.x = dout; P.ldx = ldd; P.out = out; P.ldo = ldo; P.F = (int)F;
    P.x2 = dout2; P.split = (int)split;              // rows gathered from a two-part table (the sharded layers), as npi_segsum_ex
    P.carry = carry; P.bias = nullptr;
    P.H = (int)H; P.C = (int)C; P.a_src = a_src; P.slope = slope;
    P.a_dst = a_src; P.m = a_src; P.s = a_src;                                 // unused in this mode
    P.tpack = reinterpret_cast<const float4*>(tpack);                          // [n_cols, H, 4], indexed by the COLUMN id over both parts
    P.hrow = hfeat; P.ldh = ldh; P.rowidx = rowidx; P.dz_out = dz;            // dz: [nnz_max, H]
    const int mode = H == 1 ? W_GAT_SRC_FUSED : H == 2 ? W_GAT_SRC_FUSED_H2 : H == 4 ? W_GAT_SRC_FUSED_H4 : W_GAT_SRC_FUSED_H8;
    return segsum_run(P, mode, 0, nnz_max, NPI_F32, stream);
}

extern "C" int npi_gat_rank1_add(float* dh, int64_t ld, const float* g_dst, const float* g_src, const float* att,
                                 int64_t N, int64_t H, int64_t C, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(N >= 0 && H > 0 && C > 0 && C % 4 == 0 && ld % 4 == 0, "npi_gat_rank1_add: bad size (C % 4 == 0, ld % 4 == 0)");
    if (N == 0) return NPI_OK;
    NPI_REQUIRE(dh && g_dst && g_src && att && ((uintptr_t)dh % 16) == 0 && ((uintptr_t)att % 16) == 0, "npi_gat_rank1_add: null or misaligned pointer");
    gat_rank1_add_kernel<<<(unsigned)ceil_div(N * H * C, 1024), 256, 0, stream>>>(dh, ld, g_dst, g_src, att, N, (int)H, (int)C);
    return check_launch("npi_gat_rank1_add");
}

extern "C" int npi_gat_scores(const float* h, int64_t ldh, const float* att, int64_t N, int64_t H, int64_t C,
                              float* a_dst, float* a_src, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(N >= 0 && H > 0 && C > 0 && ldh >= H * C, "npi_gat_scores: bad size");
    if (N == 0) return NPI_OK;
    NPI_REQUIRE(h && att && a_dst && a_src, "npi_gat_scores: null pointer");
    const int64_t lph = C / 4;
    if (H > 1 && H * C <= 256 && rows16(h, ldh, C) && ((uintptr_t)att % 16) == 0 && lph >= 2 && (lph & (lph - 1)) == 0)
        gat_scores_heads_kernel<<<(unsigned)ceil_div(N, 4 * ROWS_PER_WAVE), 256, 0, stream>>>(h, ldh, att, (int)N, (int)H, (int)C, a_dst, a_src);
    else if (rows16(h, ldh, C) && ((uintptr_t)att % 16) == 0)
        gat_scores_vec_kernel<<<(unsigned)ceil_div(N, 4 * ROWS_PER_WAVE), 256, 0, stream>>>(h, ldh, att, (int)N, (int)H, (int)C, a_dst, a_src);
    else
        gat_scores_kernel<<<(unsigned)ceil_div(N, 4), 256, 0, stream>>>(h, ldh, att, (int)N, (int)H, (int)C, a_dst, a_src);
    return check_launch("npi_gat_scores");
}

extern "C" int npi_gat_rowdot(const float* a, int64_t lda, const float* b, int64_t ldb, const float* bias,
                              int64_t N, int64_t H, int64_t C, float* D, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(N >= 0 && H > 0 && C > 0, "npi_gat_rowdot: bad size");
    if (N == 0) return NPI_OK;
    NPI_REQUIRE(a && b && D, "npi_gat_rowdot: null pointer");
    if (rows16(a, lda, C) && rows16(b, ldb, C) && ((uintptr_t)bias % 16) == 0)
        gat_rowdot_vec_kernel<<<(unsigned)ceil_div(N, 4 * ROWS_PER_WAVE), 256, 0, stream>>>(a, lda, b, ldb, bias, (int)N, (int)H, (int)C, D);
    else
        gat_rowdot_kernel<<<(unsigned)ceil_div(N, 4), 256, 0, stream>>>(a, lda, b, ldb, bias, (int)N, (int)H, (int)C, D);
    return check_launch("npi_gat_rowdot");
}

static int rdc_rows(int64_t N) { return N >= (1 << 18) ? 1024 : (N >= (1 << 14) ? 256 : 64); }

extern "C" int64_t npi_gat_rowdot_colsum_workspace_elems(int64_t N, int64_t H, int64_t C) {
    if (N < 0 || H <= 0 || C <= 0) return -1;
    return ceil_div(N > 0 ? N : 1, (int64_t)rdc_rows(N)) * H * C;
}

extern "C" int npi_gat_rowdot_colsum(const float* a, int64_t lda, const float* b, int64_t ldb, const float* bias,
                                     int64_t N, int64_t H, int64_t C, float* D, float* colsum, float* workspace,
                                     int64_t workspace_elems, void* stream_) {
    return npi_gat_rowdot_colsum_relu(a, lda, b, ldb, bias, N, H, C, D, colsum, nullptr, 0, workspace, workspace_elems, stream_);
}

extern "C" int npi_gat_rowdot_colsum_relu(const float* a, int64_t lda, const float* b, int64_t ldb, const float* bias,
                                          int64_t N, int64_t H, int64_t C, float* D, float* colsum, float* a_masked,
                                          int64_t ldm, float* workspace, int64_t workspace_elems, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(N >= 0 && H > 0 && C > 0 && N < 0x7fffffff, "npi_gat_rowdot_colsum: bad size");
    if (N == 0) return NPI_OK;
    NPI_REQUIRE(a && b && D, "npi_gat_rowdot_colsum: null pointer");
    const int64_t Fw = H * C;
    NPI_REQUIRE(a_masked == nullptr || (ldm >= Fw && ldm % 4 == 0 && ((uintptr_t)a_masked % 16) == 0), "npi_gat_rowdot_colsum_relu: a_masked pitch / alignment");
    if (!(rows16(a, lda, C) && rows16(b, ldb, C) && ((uintptr_t)bias % 16) == 0 && Fw <= RDC_MAXCH * 256)) {
        set_error("npi_gat_rowdot_colsum: needs 16-byte aligned rows, out_channels %% 4 == 0 and heads * out_channels <= 1024 "
                  "(use npi_gat_rowdot + npi_colsum otherwise)");
        return NPI_ERR_ARG;
    }
    const int rows = rdc_rows(N);
    const int64_t nblocks = ceil_div(N, (int64_t)rows);
    if (colsum != nullptr && (workspace == nullptr || workspace_elems < nblocks * Fw)) {
        set_error("npi_gat_rowdot_colsum: workspace too small");
        return NPI_ERR_WORKSPACE;
    }
    float* part = colsum ? workspace : nullptr;
    switch ((int)ceil_div(Fw, 256)) {
#define NPI_RDC(K) case K: gat_rowdot_colsum_kernel<K><<<(unsigned)nblocks, 256, 0, stream>>>(a, lda, b, ldb, bias, (int)N, (int)H, (int)C, rows, D, part, a_masked, ldm); break
        NPI_RDC(1); NPI_RDC(2); NPI_RDC(3); default: NPI_RDC(4);
#undef NPI_RDC
    }
    if (colsum != nullptr)
        colsum_blocks_kernel<<<(unsigned)ceil_div(Fw, 64), 256, 0, stream>>>(workspace, (int)nblocks, (int)Fw, colsum);
    return check_launch("npi_gat_rowdot_colsum");
}

extern "C" int64_t npi_gat_heavy_workspace_elems(int64_t nnz_max, int64_t H) {
    const int64_t items = npi_num_items(nnz_max);
    // segment partials (2 slots x 2 values per item and head), row counters, segment count, segment list
    return 4 * items * (H > 0 ? H : 1) + items + 1 + heavy_list_capacity(nnz_max) + 64;
}

extern "C" int npi_gat_softmax_stats(const int32_t* rowptr, const int32_t* col, const int32_t* item_row,
                                     const float* a_dst, const float* a_src, int64_t N, int64_t nnz_max, int64_t H,
                                     float slope, float* m, float* s, float* workspace, int64_t workspace_elems,
                                     void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(N >= 0 && H > 0 && H <= HEAVY_THREADS && nnz_max >= 0, "npi_gat_softmax_stats: bad size");
    if (N == 0) return NPI_OK;
    NPI_REQUIRE(rowptr && item_row && a_dst && a_src && m && s && workspace, "npi_gat_softmax_stats: null pointer");
    if (workspace_elems < npi_gat_heavy_workspace_elems(nnz_max, H)) {
        set_error("npi_gat_softmax_stats: workspace too small");
        return NPI_ERR_WORKSPACE;
    }
    switch (row_group_lanes(nnz_max, N)) {
#define NPI_ROWS(LG) gat_softmax_rows_kernel<LG><<<(unsigned)ceil_div(N, 4 * (WAVE / LG)), 256, 0, stream>>>(rowptr, col, a_dst, a_src, (int)N, (int)H, slope, m, s)
        case 8: NPI_ROWS(8); break;
        case 16: NPI_ROWS(16); break;
        case 32: NPI_ROWS(32); break;
        default: NPI_ROWS(64); break;
#undef NPI_ROWS
    }
    const int64_t n_items = npi_num_items(nnz_max);
    if (n_items > 0) {
        int* cnt = reinterpret_cast<int*>(workspace + 4 * n_items * H);
        int* count = cnt + n_items;
        int* list = count + 1;
        const int cap = (int)heavy_list_capacity(nnz_max);
        const int ie = item_edges_for(nnz_max);
        (void)hipMemsetAsync(count, 0, sizeof(int), stream);
        heavy_list_kernel<<<(unsigned)ceil_div(n_items, 256), 256, 0, stream>>>(rowptr, item_row, (int)N, (int)n_items, ie, cap, cnt, count, list);
        gat_softmax_heavy_kernel<<<(unsigned)(cap < HEAVY_GRID ? cap : HEAVY_GRID), HEAVY_THREADS, 0, stream>>>(rowptr, col, item_row, a_dst, a_src, (int)N, (int)n_items,
                                                                             (int)H, slope, m, s, ie, workspace, cnt, count, list);
    }
    return check_launch("npi_gat_softmax_stats");
}

extern "C" int npi_gat_aggregate(const int32_t* rowptr, const int32_t* col, const int32_t* item_row,
                                 int64_t N, int64_t nnz_max, const float* x, int64_t ldx, float* out, int64_t ldo,
                                 int64_t H, int64_t C, const float* a_dst, const float* a_src, const float* m,
                                 const float* s, float slope, int by_source, const float* bias,
                                 const float* g_dst, const float* g_src, const float* att,
                                 const float* alpha, const int32_t* alpha_map,
                                 float* carry, void* stream_) {
    return npi_gat_aggregate_ex(rowptr, col, item_row, N, nnz_max, x, ldx, nullptr, 0, out, ldo, H, C, a_dst, a_src, m, s,
                                slope, by_source, bias, g_dst, g_src, att, alpha, alpha_map, carry, stream_);
}

extern "C" int npi_gat_aggregate_ex(const int32_t* rowptr, const int32_t* col, const int32_t* item_row,
                                    int64_t N, int64_t nnz_max, const float* x, int64_t ldx, const float* x2, int64_t split,
                                    float* out, int64_t ldo,
                                    int64_t H, int64_t C, const float* a_dst, const float* a_src, const float* m,
                                    const float* s, float slope, int by_source, const float* bias,
                                    const float* g_dst, const float* g_src, const float* att,
                                    const float* alpha, const int32_t* alpha_map,
                                    float* carry, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(x2 == nullptr || (split >= 0 && split < 0x7fffffff), "npi_gat_aggregate_ex: bad split");
    NPI_REQUIRE(N >= 0 && nnz_max > 0 && H > 0 && C > 0, "npi_gat_aggregate: bad size");
    if (N == 0) return NPI_OK;
    NPI_REQUIRE(rowptr && col && item_row && x && out && a_dst && a_src && m && s && carry, "npi_gat_aggregate: null pointer");
    NPI_REQUIRE(ldx >= H * C && ldo >= H * C, "npi_gat_aggregate: leading dimension too small");
    SegParams P{};
    P.rowptr = rowptr; P.col = col; P.item_row = item_row;
    P.N = (int)N; P.n_items = (int)npi_num_items(nnz_max);
    P.x = x; P.ldx = ldx; P.out = out; P.ldo = ldo; P.F = (int)(H * C);
    P.x2 = x2; P.split = (int)split;
    P.carry = carry; P.w = nullptr; P.bias = bias;
    P.H = (int)H; P.C = (int)C; P.a_dst = a_dst; P.a_src = a_src; P.m = m; P.s = s; P.slope = slope;
    P.g_dst = g_dst; P.g_src = g_src; P.att = att;
    if (alpha != nullptr && !by_source) {     // forward, one head: also WRITE alpha of every entry (by-target order) for the backward
        NPI_REQUIRE(H == 1 && alpha_map == nullptr, "npi_gat_aggregate: storing alpha needs one head (and no alpha_map)");
        P.alpha_out = const_cast<float*>(alpha);
        return segsum_run(P, W_GAT_DST, 0, nnz_max, NPI_F32, stream);
    }
    if (alpha != nullptr) {       // the weights of the forward / of npi_gat_edge_grad, read back through the transpose map (one head, by source)
        NPI_REQUIRE(by_source && H == 1 && alpha_map != nullptr, "npi_gat_aggregate: alpha needs by_source, one head and alpha_map");
        P.w = alpha; P.wmap = alpha_map;
        return segsum_run(P, W_GAT_SRC_PRE, 0, nnz_max, NPI_F32, stream);
    }
    return segsum_run(P, by_source ? W_GAT_SRC : W_GAT_DST, 0, nnz_max, NPI_F32, stream);
}

extern "C" int npi_gat_aggregate_scores(const int32_t* rowptr, const int32_t* col, const int32_t* item_row,
                                        int64_t N, int64_t nnz_max, const float* x, int64_t ldx, const float* x2, int64_t split,
                                        float* out, int64_t ldo, int64_t C, const float* scores, const float* m, const float* s,
                                        const float* bias, int relu, float* alpha_out, float* carry, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(x2 == nullptr || (split >= 0 && split < 0x7fffffff), "npi_gat_aggregate_scores: bad split");
    NPI_REQUIRE(N >= 0 && nnz_max > 0 && C > 0, "npi_gat_aggregate_scores: bad size");
    if (N == 0) return NPI_OK;
    NPI_REQUIRE(rowptr && col && item_row && x && out && scores && m && s && carry, "npi_gat_aggregate_scores: null pointer");
    NPI_REQUIRE(ldx >= C && ldo >= C, "npi_gat_aggregate_scores: leading dimension too small");
    SegParams P{};
    P.rowptr = rowptr; P.col = col; P.item_row = item_row;
    P.N = (int)N; P.n_items = (int)npi_num_items(nnz_max);
    P.x = x; P.ldx = ldx; P.out = out; P.ldo = ldo; P.F = (int)C;
    P.x2 = x2; P.split = (int)split;
    P.carry = carry; P.w = scores; P.bias = bias;
    P.H = 1; P.C = (int)C; P.m = m; P.s = s; P.alpha_out = alpha_out; P.relu = relu ? 1 : 0;
    return segsum_run(P, W_GAT_DST_PRE, 0, nnz_max, NPI_F32, stream);
}

extern "C" int npi_gat_edge_grad(const int32_t* rowptr, const int32_t* col, const int32_t* rowidx,
                                 int64_t N, int64_t nnz_max, const float* hfeat, int64_t ldh,
                                 const float* dout, int64_t ldd, int64_t H, int64_t C,
                                 const float* a_dst, const float* a_src, const float* m, const float* s,
                                 const float* D, float slope, float* dz, float* alpha_out, void* stream_) {
    return npi_gat_edge_grad_ex(rowptr, col, rowidx, N, nnz_max, hfeat, ldh, nullptr, 0, dout, ldd, H, C, a_dst, a_src, m, s, D,
                                slope, 0, dz, alpha_out, stream_);
}

extern "C" int npi_gat_edge_grad_ex(const int32_t* rowptr, const int32_t* col, const int32_t* rowidx,
                                    int64_t N, int64_t nnz_max, const float* hfeat, int64_t ldh,
                                    const float* hfeat2, int64_t split,
                                    const float* dout, int64_t ldd, int64_t H, int64_t C,
                                    const float* a_dst, const float* a_src, const float* m, const float* s,
                                    const float* D, float slope, int swap, float* dz, float* alpha_out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(hfeat2 == nullptr || (split >= 0 && split < 0x7fffffff), "npi_gat_edge_grad_ex: bad split");
    if (hfeat2 == nullptr) { hfeat2 = hfeat; split = 0x7fffffff; }
    NPI_REQUIRE((uintptr_t)hfeat2 % 16 == 0, "npi_gat_edge_grad_ex: hfeat2 must be 16-B aligned");
    NPI_REQUIRE(N >= 0 && nnz_max >= 0 && H > 0 && C > 0, "npi_gat_edge_grad: bad size");
    if (N == 0 || nnz_max == 0) return NPI_OK;
    NPI_REQUIRE(rowptr && col && rowidx && hfeat && dout && a_dst && a_src && m && s && D && dz, "npi_gat_edge_grad: null pointer");
    const int64_t F = H * C;
    NPI_REQUIRE(F % 4 == 0 && C % 4 == 0 && ldh % 4 == 0 && ldd % 4 == 0 && F <= 1024 &&
                ((uintptr_t)hfeat % 16 == 0) && ((uintptr_t)dout % 16 == 0),
                "npi_gat_edge_grad: needs 16-B aligned rows, out_channels % 4 == 0, heads*out_channels <= 1024");
    const int n_items = (int)npi_num_items(nnz_max);
    const unsigned grid = (unsigned)ceil_div(n_items, 4);
    const int nch = (int)ceil_div(F, 256);
#define NPI_EG(NC) gat_edge_grad_kernel<NC><<<grid, 256, 0, stream>>>(rowptr, col, rowidx, (int)N, n_items, item_edges_for(nnz_max), hfeat, ldh, dout, ldd, (int)H, (int)C, a_dst, a_src, m, s, D, slope, dz, alpha_out, hfeat2, (int)split, swap)
    if (nch == 1) NPI_EG(1); else if (nch == 2) NPI_EG(2); else if (nch == 3) NPI_EG(3); else NPI_EG(4);
#undef NPI_EG
    return check_launch("npi_gat_edge_grad");
}

extern "C" int npi_seg_rowsum(const int32_t* rowptr, const int32_t* item_row, const float* vals, const int32_t* map,
                              int64_t N, int64_t nnz_max, int64_t H, float* out, float* workspace, int64_t workspace_elems,
                              void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(N >= 0 && H > 0 && H <= HEAVY_THREADS && nnz_max >= 0, "npi_seg_rowsum: bad size");
    if (N == 0) return NPI_OK;
    NPI_REQUIRE(rowptr && vals && out, "npi_seg_rowsum: null pointer");
    const int64_t n_items = item_row ? npi_num_items(nnz_max) : 0;
    if (n_items > 0) {
        NPI_REQUIRE(workspace != nullptr, "npi_seg_rowsum: null workspace");
        if (workspace_elems < npi_gat_heavy_workspace_elems(nnz_max, H)) {
            set_error("npi_seg_rowsum: workspace too small");
            return NPI_ERR_WORKSPACE;
        }
    }
    // rows up to 4096 entries: one wave each; longer ones: 1024-thread workgroups over 64-item segments
    switch (row_group_lanes(nnz_max, N)) {
#define NPI_ROWS(LG) seg_rowsum_scalar_kernel<LG><<<(unsigned)ceil_div(N, 4 * (WAVE / LG)), 256, 0, stream>>>(rowptr, vals, map, (int)N, (int)H, out, n_items > 0 ? 1 : 0)
        case 8: NPI_ROWS(8); break;
        case 16: NPI_ROWS(16); break;
        case 32: NPI_ROWS(32); break;
        default: NPI_ROWS(64); break;
#undef NPI_ROWS
    }
    if (n_items > 0) {
        int* cnt = reinterpret_cast<int*>(workspace + 4 * n_items * H);
        int* count = cnt + n_items;
        int* list = count + 1;
        const int cap = (int)heavy_list_capacity(nnz_max);
        const int ie = item_edges_for(nnz_max);
        (void)hipMemsetAsync(count, 0, sizeof(int), stream);
        heavy_list_kernel<<<(unsigned)ceil_div(n_items, 256), 256, 0, stream>>>(rowptr, item_row, (int)N, (int)n_items, ie, cap, cnt, count, list);
        seg_rowsum_heavy_kernel<<<(unsigned)(cap < HEAVY_GRID ? cap : HEAVY_GRID), HEAVY_THREADS, 0, stream>>>(rowptr, item_row, vals, map, (int)N, (int)n_items, (int)H, out,
                                                                            ie, workspace, cnt, count, list);
    }
    return check_launch("npi_seg_rowsum");
}

extern "C" int npi_entry_transpose_map(const int32_t* src_eid, const int32_t* src_rowidx, const int32_t* src_rowptr,
                                       const int32_t* dst_rowptr, const int32_t* pos_dst_of_edge, int64_t N,
                                       int64_t nnz_max, int32_t* map, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(N >= 0 && nnz_max >= 0, "npi_entry_transpose_map: bad size");
    if (nnz_max == 0) return NPI_OK;
    NPI_REQUIRE(src_eid && src_rowidx && src_rowptr && dst_rowptr && map, "npi_entry_transpose_map: null pointer");
    entry_transpose_map_kernel<<<(unsigned)ceil_div(nnz_max, 256), 256, 0, stream>>>(src_eid, src_rowidx, src_rowptr, dst_rowptr, pos_dst_of_edge, (int)N, nnz_max, map);
    return check_launch("npi_entry_transpose_map");
}

extern "C" int64_t npi_gat_att_grad_workspace_elems(int64_t N, int64_t H, int64_t C) {
    return ceil_div(N > 0 ? N : 1, ATT_ROWS) * 2 * H * C;
}

extern "C" int npi_gat_att_grad(const float* hfeat, int64_t ldh, const float* g_dst, const float* g_src,
                                int64_t N, int64_t H, int64_t C, float* datt, float* workspace,
                                int64_t workspace_elems, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(N >= 0 && H > 0 && C > 0, "npi_gat_att_grad: bad size");
    NPI_REQUIRE(hfeat && g_dst && g_src && datt && workspace, "npi_gat_att_grad: null pointer");
    if (workspace_elems < npi_gat_att_grad_workspace_elems(N, H, C)) {
        set_error("npi_gat_att_grad: workspace too small");
        return NPI_ERR_WORKSPACE;
    }
    const int nchunks = (int)ceil_div(N > 0 ? N : 1, ATT_ROWS);
    const int F = (int)(H * C);
    gat_att_grad_partial_kernel<<<dim3((unsigned)ceil_div(F, 256), (unsigned)nchunks), 256, 0, stream>>>(hfeat, ldh, g_dst, g_src, (int)N, (int)H, (int)C, workspace);
    gat_att_grad_reduce_kernel<<<(unsigned)ceil_div(2 * F, 32), 1024, 0, stream>>>(workspace, nchunks, (int)H, (int)C, datt);
    return check_launch("npi_gat_att_grad");
}
