// GATConv (PyG 1.4.2; SURVEY.md 8(a) row a8, BASELINE.json configs[4]) on the destination-sorted CSR.
//   h = x W;  e_p = leaky_relu(<h_i, att[:C]> + <h_j, att[C:]>) for entry p = (i <- j), self loops included;
//   alpha = softmax of e over the entries of row i (exp(e - max) / (sum + 1e-16));  out_i = sum_p alpha_p h_j (+ b)
// GATConv is absent from the reference tree ("parity unpinned"); the formulas are the
// published PyG 1.4.2 ones.
//
// alpha is never stored: every kernel recomputes it from per-node, per-head scalars (a_dst, a_src, row max m, row sum s),
// so the same numbers serve the by-target CSR (forward) and the by-source CSR (backward) and no per-edge array has to be
// permuted between the two.  This file holds the per-node dot products, the by-target SDDMM (npi_gat_edge_grad, shapes the
// fused backward does not cover), the attention-gradient reductions and the entry points of the weighted aggregations, which
// are segsum.hip's kernel in its W_GAT_* modes; the per-row statistics and row sums live in segscan.hip.
#include "segsum.h"

namespace npi {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
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

// out[y, c] = sum over slice y of the blocks of part[k, c] (gridDim.y slices of ceil(nblocks / gridDim.y) blocks): 64 columns per
// workgroup, its four waves take the quarters of the slice with four independent partial sums each (k mod 4), folded in a fixed
// order.  One slice over ~1,000 blocks was a chain of 61 dependent load rounds in four workgroups (31 us at C4, 117 us at the C5
// size): colsum_blocks() below cuts COLSUM_SLICES slices first and sums those with a second launch.
constexpr int COLSUM_SLICES = 32;
__global__ void __launch_bounds__(256)
colsum_blocks_kernel(const float* __restrict__ part, int nblocks, int Fw, float* __restrict__ out) {
    __shared__ float red[4][64];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int per_y = (nblocks + (int)gridDim.y - 1) / (int)gridDim.y;
    const int yb = blockIdx.y * per_y, ye = min(nblocks, yb + per_y);
    const int per = (max(ye - yb, 0) + 3) / 4;
    const int kb = yb + wave * per, ke = min(ye, kb + per);
    out += (int64_t)blockIdx.y * Fw;
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

// part [nblocks, Fw] -> out [Fw]; `slices` [COLSUM_SLICES, Fw] scratch behind the partials (fixed orders: deterministic)
static void colsum_blocks(const float* part, int64_t nblocks, int64_t Fw, float* out, float* slices, hipStream_t stream) {
    const unsigned gx = (unsigned)ceil_div(Fw, 64);
    if (nblocks <= 4 * COLSUM_SLICES) {
        colsum_blocks_kernel<<<gx, 256, 0, stream>>>(part, (int)nblocks, (int)Fw, out);
        return;
    }
    colsum_blocks_kernel<<<dim3(gx, COLSUM_SLICES), 256, 0, stream>>>(part, (int)nblocks, (int)Fw, slices);
    colsum_blocks_kernel<<<gx, 256, 0, stream>>>(slices, COLSUM_SLICES, (int)Fw, out);
}

static bool rows16(const void* p, int64_t ld, int64_t C) { return C % 4 == 0 && ld % 4 == 0 && ((uintptr_t)p % 16) == 0; }

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
    const float* __restrict__ hf2 = hfeat2;           // (a pointer biased through integer arithmetic turns every gather into a FLAT load)
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
                    hv8[u] = act[0] ? *reinterpret_cast<const float4*>((cu < split ? hfeat + (int64_t)cu * ldh : hf2 + (int64_t)(cu - split) * ldh) + foff[0])
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
                    hv[u][c] = act[c] ? *reinterpret_cast<const float4*>((cu < split ? hfeat + (int64_t)cu * ldh : hf2 + (int64_t)(cu - split) * ldh) + foff[c])
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
// The same partials with 16-byte lane loads (F % 4 == 0, C % 4 == 0, aligned rows, F <= 1024): a thread owns one column
// QUAD q of the row lanes rl, rl + RL, ... of its workgroup's chunk (QP = quads per row rounded up to a power of two,
// RL = 256 / QP; F = 256: one whole 1 KiB row per wave instruction, 4 rows per workgroup instruction), eight rows in flight
// per lane.  UNI (one head, QP >= 64: a wavefront stays on one row lane): g_dst / g_src of 64 rows are fetched lane-parallel
// and broadcast with v_readlane, so a row costs ONE vector-memory instruction.  The row lanes are folded through LDS in
// a fixed order (deterministic).  1 GB at C4: 0.73 ms with the 4-byte kernel above, alone.
template <bool UNI>
__global__ void __launch_bounds__(256)
gat_att_grad_partial_vec_kernel(const float* __restrict__ hfeat, int64_t ldh, const float* __restrict__ g_dst,
                                const float* __restrict__ g_src, int N, int H, int C, int qp_log2, int rows_per_wg,
                                float* __restrict__ part) {
    __shared__ float4 red[2][256];
    const int F = H * C, Q = F >> 2;
    const int QP = 1 << qp_log2, RL = 256 >> qp_log2;
    const int q = threadIdx.x & (QP - 1), rl = threadIdx.x >> qp_log2;
    const int rbeg = blockIdx.x * rows_per_wg, rend = min(N, rbeg + rows_per_wg);
    const bool live = q < Q;
    const int hd = live ? (4 * q) / C : 0;
    const float* col = hfeat + 4 * (live ? q : 0);
    float4 sd = make_float4(0.f, 0.f, 0.f, 0.f), ss = sd;
    auto acc = [&](const float4& v, float a, float b) {
        sd.x = fmaf(a, v.x, sd.x); sd.y = fmaf(a, v.y, sd.y); sd.z = fmaf(a, v.z, sd.z); sd.w = fmaf(a, v.w, sd.w);
        ss.x = fmaf(b, v.x, ss.x); ss.y = fmaf(b, v.y, ss.y); ss.z = fmaf(b, v.z, ss.z); ss.w = fmaf(b, v.w, ss.w);
    };
    int i = rbeg + rl;
    if constexpr (UNI) {
        const int lane = threadIdx.x & 63;
        for (; i + 63 * RL < rend; i += 64 * RL) {             // wave-uniform: 64 full rows of this row lane
            const int gd = __float_as_int(g_dst[i + lane * RL]), gs = __float_as_int(g_src[i + lane * RL]);
            if (live) {
#pragma unroll
                for (int k0 = 0; k0 < 64; k0 += 8) {
                    float4 v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float4*>(col + (int64_t)(i + (k0 + k) * RL) * ldh);
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        acc(v[k], __int_as_float(__builtin_amdgcn_readlane(gd, k0 + k)),
                            __int_as_float(__builtin_amdgcn_readlane(gs, k0 + k)));
                }
            }
        }
    }
    if (live) {
#pragma unroll 4
        for (; i < rend; i += RL) {
            const float4 v = *reinterpret_cast<const float4*>(col + (int64_t)i * ldh);
            acc(v, g_dst[(int64_t)i * H + hd], g_src[(int64_t)i * H + hd]);
        }
    }
    red[0][threadIdx.x] = sd;
    red[1][threadIdx.x] = ss;
    __syncthreads();
    if (rl == 0 && live) {
        for (int r = 1; r < RL; ++r) {
            const float4 a = red[0][(r << qp_log2) + q], b = red[1][(r << qp_log2) + q];
            sd.x += a.x; sd.y += a.y; sd.z += a.z; sd.w += a.w;
            ss.x += b.x; ss.y += b.y; ss.z += b.z; ss.w += b.w;
        }
        *reinterpret_cast<float4*>(part + ((int64_t)blockIdx.x * 2 + 0) * F + 4 * q) = sd;
        *reinterpret_cast<float4*>(part + ((int64_t)blockIdx.x * 2 + 1) * F + 4 * q) = ss;
    }
}
// rows per workgroup of the 16-byte kernel: at most ~2,048 chunks (their partials are summed by ONE small launch), at least
// ATT_ROWS rows (the workspace is sized for ATT_ROWS-row chunks)
static int att_vec_rows(int64_t N) { return (int)std::max<int64_t>(ATT_ROWS, ceil_div(N, 2048)); }
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

extern "C" int npi_gat_backward_fused_heads(const int32_t* rowptr, const int32_t* col, const int32_t* rowidx,
                                                const int32_t* item_row, int64_t item_edges, int64_t N, int64_t nnz_max,
                                                const float* dout, int64_t ldd, const float* dout2, int64_t split, const float* hfeat,
                                                int64_t ldh, float* out, int64_t ldo, int64_t H, int64_t C, const float* tpack,
                                                const float* a_src, float slope, float* dz, float* carry, float* row_scales_out,
                                                float* g_src_out, float* workspace, int64_t workspace_elems, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t F = H * C;
    NPI_REQUIRE(row_scales_out == nullptr || (F == 256 && ldd % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)dout % 16) == 0 &&
                                              ((uintptr_t)out % 16) == 0 && (dout2 == nullptr || ((uintptr_t)dout2 % 16) == 0)),
                "npi_gat_backward_fused_heads: row_scales_out needs heads * out_channels == 256 and 16-byte aligned rows");
    NPI_REQUIRE(N >= 0 && nnz_max > 0 && C > 0 && C % 4 == 0 && F <= 256, "npi_gat_backward_fused_heads: needs heads * out_channels <= 256, out_channels % 4 == 0");
    NPI_REQUIRE(H == 1 || ((H == 2 || H == 4 || H == 8) && C >= 32 && (C & (C - 1)) == 0),
                "npi_gat_backward_fused_heads: several heads need 2 / 4 / 8 heads of 32 / 64 / 128 channels");
    NPI_REQUIRE(dout2 == nullptr || (split >= 0 && split < 0x7fffffff), "npi_gat_backward_fused_heads: bad split");
    NPI_REQUIRE(item_edges_ok(item_edges), "npi_gat_backward_fused_heads: item_edges must be 64 or NPI_ITEM_EDGES (the value the CSR was built with)");
    if (N == 0) return NPI_OK;
    NPI_REQUIRE(rowptr && col && rowidx && item_row && dout && hfeat && out && tpack && a_src && dz && carry,
                "npi_gat_backward_fused_heads: null pointer");
    NPI_REQUIRE(ldd >= F && ldh >= F && ldo >= F && ldh % 4 == 0 && ((uintptr_t)hfeat % 16) == 0 && ((uintptr_t)tpack % 16) == 0,
                "npi_gat_backward_fused_heads: leading dimension / alignment");
    SegParams P{};
    P.rowptr = rowptr; P.col = col; P.item_row = item_row;
    P.N = (int)N; P.item = (int)item_edges;
    P.x = dout; P.ldx = ldd; P.out = out; P.ldo = ldo; P.F = (int)F;
    P.x2 = dout2; P.split = (int)split;              // rows gathered from a two-part table (the sharded layers), as npi_segsum_ex
    P.carry = carry; P.bias = nullptr;
    P.H = (int)H; P.C = (int)C; P.a_src = a_src; P.slope = slope;
    P.a_dst = a_src; P.m = a_src; P.s = a_src;                                 // unused in this mode
    P.tpack = reinterpret_cast<const float4*>(tpack);                          // [n_cols, H, 4], indexed by the COLUMN id over both parts
    P.hrow = hfeat; P.ldh = ldh; P.rowidx = rowidx; P.dz_out = dz;            // dz: [nnz_max, H]
    P.scale_out = row_scales_out;
    const int mode = H == 1 ? W_GAT_SRC_FUSED : H == 2 ? W_GAT_SRC_FUSED_H2 : H == 4 ? W_GAT_SRC_FUSED_H4 : W_GAT_SRC_FUSED_H8;
    // g_src_out [N] (one head): the row sums of dz from the same launch -- the lanes that compute dz scan it by row; rows cut by an
    // item boundary are added up by the chain kernel of segscan.hip behind it (workspace: npi_seg_scan_workspace_elems(nnz_max, 1))
    const int64_t n_items = num_items_of(nnz_max, (int)item_edges);
    if (g_src_out != nullptr) {
        NPI_REQUIRE(H == 1, "npi_gat_backward_fused_heads: g_src_out serves one head");
        if (workspace == nullptr || workspace_elems < 3 * n_items) {
            set_error("npi_gat_backward_fused_heads: workspace too small");
            return NPI_ERR_WORKSPACE;
        }
        (void)hipMemsetAsync(g_src_out, 0, sizeof(float) * N, stream);                 // rows without an entry
        P.rowsum_out = g_src_out; P.rs_head = workspace; P.rs_tail = workspace + n_items;
        P.rs_tail_row = reinterpret_cast<int32_t*>(workspace + 2 * n_items);
    }
    const int rc = segsum_run(P, mode, 0, nnz_max, NPI_F32, stream);
    if (rc != NPI_OK || g_src_out == nullptr) return rc;
    return seg_chain_sum(rowptr, P.rs_head, P.rs_tail, P.rs_tail_row, g_src_out, N, n_items, (int)item_edges, stream);
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
    return (ceil_div(N > 0 ? N : 1, (int64_t)rdc_rows(N)) + COLSUM_SLICES) * H * C;     // the blocks' partials + their slices' sums
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
    if (colsum != nullptr && (workspace == nullptr || workspace_elems < (nblocks + COLSUM_SLICES) * Fw)) {
        set_error("npi_gat_rowdot_colsum: workspace too small");
        return NPI_ERR_WORKSPACE;
    }
    float* part = colsum ? workspace : nullptr;
    switch ((int)ceil_div(Fw, 256)) {
#define NPI_RDC(K) case K: gat_rowdot_colsum_kernel<K><<<(unsigned)nblocks, 256, 0, stream>>>(a, lda, b, ldb, bias, (int)N, (int)H, (int)C, rows, D, part, a_masked, ldm); break
        NPI_RDC(1); NPI_RDC(2); NPI_RDC(3); default: NPI_RDC(4);
#undef NPI_RDC
    }
    if (colsum != nullptr) colsum_blocks(workspace, nblocks, Fw, colsum, workspace + nblocks * Fw, stream);
    return check_launch("npi_gat_rowdot_colsum");
}

extern "C" int npi_gat_aggregate_ex(const int32_t* rowptr, const int32_t* col, const int32_t* item_row, int64_t item_edges,
                                    int64_t N, int64_t nnz_max, const float* x, int64_t ldx, const float* x2, int64_t split,
                                    float* out, int64_t ldo,
                                    int64_t H, int64_t C, const float* a_dst, const float* a_src, const float* m,
                                    const float* s, float slope, int by_source, const float* bias,
                                    const float* g_dst, const float* g_src, const float* att,
                                    const float* alpha, const int32_t* alpha_map,
                                    float* carry, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(x2 == nullptr || (split >= 0 && split < 0x7fffffff), "npi_gat_aggregate_ex: bad split");
    NPI_REQUIRE(item_edges_ok(item_edges), "npi_gat_aggregate: item_edges must be 64 or NPI_ITEM_EDGES (the value the CSR was built with)");
    NPI_REQUIRE(N >= 0 && nnz_max > 0 && H > 0 && C > 0, "npi_gat_aggregate: bad size");
    if (N == 0) return NPI_OK;
    NPI_REQUIRE(rowptr && col && item_row && x && out && a_dst && a_src && m && s && carry, "npi_gat_aggregate: null pointer");
    NPI_REQUIRE(ldx >= H * C && ldo >= H * C, "npi_gat_aggregate: leading dimension too small");
    SegParams P{};
    P.rowptr = rowptr; P.col = col; P.item_row = item_row;
    P.N = (int)N; P.item = (int)item_edges;
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

extern "C" int npi_gat_aggregate_scores(const int32_t* rowptr, const int32_t* col, const int32_t* item_row, int64_t item_edges,
                                        int64_t N, int64_t nnz_max, const float* x, int64_t ldx, const float* x2, int64_t split,
                                        float* out, int64_t ldo, int64_t C, const float* scores, const float* m, const float* s,
                                        const float* bias, int relu, float* carry, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(x2 == nullptr || (split >= 0 && split < 0x7fffffff), "npi_gat_aggregate_scores: bad split");
    NPI_REQUIRE(item_edges_ok(item_edges), "npi_gat_aggregate_scores: item_edges must be 64 or NPI_ITEM_EDGES (the value the CSR was built with)");
    NPI_REQUIRE(N >= 0 && nnz_max > 0 && C > 0, "npi_gat_aggregate_scores: bad size");
    if (N == 0) return NPI_OK;
    NPI_REQUIRE(rowptr && col && item_row && x && out && scores && m && s && carry, "npi_gat_aggregate_scores: null pointer");
    NPI_REQUIRE(ldx >= C && ldo >= C, "npi_gat_aggregate_scores: leading dimension too small");
    SegParams P{};
    P.rowptr = rowptr; P.col = col; P.item_row = item_row;
    P.N = (int)N; P.item = (int)item_edges;
    P.x = x; P.ldx = ldx; P.out = out; P.ldo = ldo; P.F = (int)C;
    P.x2 = x2; P.split = (int)split;
    P.carry = carry; P.w = scores; P.bias = bias;
    P.H = 1; P.C = (int)C; P.m = m; P.s = s; P.relu = relu ? 1 : 0;
    return segsum_run(P, W_GAT_DST_PRE, 0, nnz_max, NPI_F32, stream);
}

// One head, C <= 256: the forward aggregation WITH the softmax statistics (W_GAT_DST_FUSED): out, m, s in one launch; no
// statistics pass, no per-entry score array, no gather of the sources' scores (npi_gat_softmax_stats_ex +
// npi_gat_aggregate_scores do the same in two): the source half of every score is recomputed from the gathered row
extern "C" int npi_gat_aggregate_fused(const int32_t* rowptr, const int32_t* col, const int32_t* rowidx, const int32_t* item_row,
                                           int64_t item_edges, int64_t N, int64_t nnz_max, const float* x, int64_t ldx,
                                           const float* x2, int64_t split, float* out, int64_t ldo, int64_t C, const float* a_dst,
                                           const float* att, float slope, const float* bias, int relu, float* m, float* s,
                                           float* carry, float* row_scales_out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(row_scales_out == nullptr || (C == 256 && ldx % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)x % 16) == 0 &&
                                              ((uintptr_t)out % 16) == 0 && (x2 == nullptr || ((uintptr_t)x2 % 16) == 0)),
                "npi_gat_aggregate_fused: row_scales_out needs 256 channels and 16-byte aligned rows");
    NPI_REQUIRE(x2 == nullptr || (split >= 0 && split < 0x7fffffff), "npi_gat_aggregate_fused: bad split");
    NPI_REQUIRE(item_edges_ok(item_edges), "npi_gat_aggregate_fused: item_edges must be 64 or NPI_ITEM_EDGES (the value the CSR was built with)");
    NPI_REQUIRE(N >= 0 && nnz_max > 0 && C > 0 && C <= 256 && C % 4 == 0, "npi_gat_aggregate_fused: bad size (one head of at most 256 channels, a multiple of 4)");
    if (N == 0) return NPI_OK;
    NPI_REQUIRE(rowptr && col && item_row && x && out && a_dst && att && m && s && carry, "npi_gat_aggregate_fused: null pointer");
    NPI_REQUIRE((uintptr_t)att % 16 == 0, "npi_gat_aggregate_fused: att must be 16-byte aligned");
    NPI_REQUIRE(ldx >= C && ldo >= C, "npi_gat_aggregate_fused: leading dimension too small");
    SegParams P{};
    P.rowptr = rowptr; P.col = col; P.item_row = item_row; P.rowidx = rowidx;
    P.N = (int)N; P.item = (int)item_edges;
    P.x = x; P.ldx = ldx; P.out = out; P.ldo = ldo; P.F = (int)C;
    P.x2 = x2; P.split = (int)split;
    P.carry = carry; P.bias = bias;
    P.H = 1; P.C = (int)C; P.a_dst = a_dst; P.att = att; P.slope = slope; P.m_out = m; P.s_out = s; P.relu = relu ? 1 : 0;
    P.scale_out = row_scales_out;                // the scales of the rows as they are stored: bias and ReLU applied
    return segsum_run(P, W_GAT_DST_FUSED, 0, nnz_max, NPI_F32, stream);
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
    // this kernel keeps no item state between calls (no item_row, no carry): its own chunking of the entry stream, per call
    const int chunk = item_edges_for(nnz_max);
    const int n_items = (int)num_items_of(nnz_max, chunk);
    const unsigned grid = (unsigned)ceil_div(n_items, 4);
    const int nch = (int)ceil_div(F, 256);
#define NPI_EG(NC) gat_edge_grad_kernel<NC><<<grid, 256, 0, stream>>>(rowptr, col, rowidx, (int)N, n_items, chunk, hfeat, ldh, dout, ldd, (int)H, (int)C, a_dst, a_src, m, s, D, slope, dz, alpha_out, hfeat2, (int)split, swap)
    if (nch == 1) NPI_EG(1); else if (nch == 2) NPI_EG(2); else if (nch == 3) NPI_EG(3); else NPI_EG(4);
#undef NPI_EG
    return check_launch("npi_gat_edge_grad");
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
    int nchunks = (int)ceil_div(N > 0 ? N : 1, ATT_ROWS);
    const int F = (int)(H * C);
    if (F % 4 == 0 && C % 4 == 0 && F <= 1024 && ldh % 4 == 0 && ((uintptr_t)hfeat & 15) == 0 && N > 0) {
        int qp_log2 = 0;
        while ((1 << qp_log2) < F / 4) ++qp_log2;
        const int rows = att_vec_rows(N);
        nchunks = (int)ceil_div(N, rows);
        if (H == 1 && qp_log2 >= 6)
            gat_att_grad_partial_vec_kernel<true><<<(unsigned)nchunks, 256, 0, stream>>>(hfeat, ldh, g_dst, g_src, (int)N, (int)H, (int)C, qp_log2, rows, workspace);
        else
            gat_att_grad_partial_vec_kernel<false><<<(unsigned)nchunks, 256, 0, stream>>>(hfeat, ldh, g_dst, g_src, (int)N, (int)H, (int)C, qp_log2, rows, workspace);
    } else {
        gat_att_grad_partial_kernel<<<dim3((unsigned)ceil_div(F, 256), (unsigned)nchunks), 256, 0, stream>>>(hfeat, ldh, g_dst, g_src, (int)N, (int)H, (int)C, workspace);
    }
    gat_att_grad_reduce_kernel<<<(unsigned)ceil_div(2 * F, 32), 1024, 0, stream>>>(workspace, nchunks, (int)H, (int)C, datt);
    return check_launch("npi_gat_att_grad");
}
