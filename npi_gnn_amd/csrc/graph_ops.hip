// Per-entry weight plumbing for the weighted convs: GCNConv.norm (PyG 1.4.2, a7 in SURVEY.md 8(a))
// and edge_weight handling of add_remaining_self_loops.  Integer/index work plus one rsqrt per
// entry; all arrays are in CSR entry order (see csr_build.hip).
#include "npi_common.h"

namespace npi {

// deg[r] = sum of w over row r, lanes stride the row, fixed-order wave reduction (deterministic)
__global__ void __launch_bounds__(256)
row_weight_sum_kernel(const int32_t* __restrict__ rowptr, const float* __restrict__ w, int N,
                      float* __restrict__ deg) {
    const int lane = lane_id();
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= N) return;
    const int b = rowptr[r], e = rowptr[r + 1];
    float s = 0.f;
    for (int p = b + lane; p < e; p += WAVE) s += w[p];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, WAVE);
    if (lane == 0) deg[r] = s;
}

__device__ __forceinline__ float inv_sqrt_or_zero(float d) {
    // PyG: deg.pow(-0.5); inf -> 0
    return d > 0.f ? 1.0f / sqrtf(d) : 0.f;
}

__global__ void gcn_norm_kernel(const int32_t* __restrict__ rowidx, const int32_t* __restrict__ col,
                                const int32_t* __restrict__ rowptr, const float* __restrict__ w,
                                const float* __restrict__ deg, const int32_t* __restrict__ deg_rowptr,
                                int N, int64_t nnz_max, float* __restrict__ norm) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nnz_max || p >= rowptr[N]) return;
    const int a = rowidx[p], b = col[p];
    float da, db;
    if (deg) { da = deg[a]; db = deg[b]; }
    else { da = (float)(deg_rowptr[a + 1] - deg_rowptr[a]); db = (float)(deg_rowptr[b + 1] - deg_rowptr[b]); }
    const float wp = w ? w[p] : 1.f;
    norm[p] = inv_sqrt_or_zero(db) * wp * inv_sqrt_or_zero(da);
}

__global__ void entry_weights_kernel(const int32_t* __restrict__ eid, const int32_t* __restrict__ rowidx,
                                     const int32_t* __restrict__ rowptr, const float* __restrict__ edge_w,
                                     const float* __restrict__ loop_w_node, float fill, int N,
                                     int64_t nnz_max, float* __restrict__ w_entry) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nnz_max || p >= rowptr[N]) return;
    const int e = eid[p];
    float v;
    if (e >= 0) v = edge_w ? edge_w[e] : 1.f;
    else v = loop_w_node ? loop_w_node[rowidx[p]] : fill;
    w_entry[p] = v;
}

__global__ void row_inv_count_kernel(const int32_t* __restrict__ rowptr, int N, float* __restrict__ inv) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    inv[i] = 1.f / (float)max(rowptr[i + 1] - rowptr[i], 1);
}

// w_out[p] = w_in[p] * table[col[p]] (w_in null: ones) for the entries of one CSR, 0 behind the last entry: a factor that belongs
// to the GATHERED row -- scatter_mean's 1 / in-count of the target, seen from the by-source side -- as a per-entry weight
__global__ void entry_col_scale_kernel(const int32_t* __restrict__ col, const int32_t* __restrict__ rowptr,
                                       const float* __restrict__ table, const float* __restrict__ w_in, int N, int n_cols,
                                       int64_t nnz_max, float* __restrict__ w_out) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nnz_max) return;
    if (p >= rowptr[N]) { w_out[p] = 0.f; return; }
    const int c = col[p];
    w_out[p] = ((unsigned)c < (unsigned)n_cols ? table[c] : 0.f) * (w_in ? w_in[p] : 1.f);
}

}  // namespace npi

using namespace npi;

extern "C" int npi_row_weight_sum(const int32_t* rowptr, const float* w, int64_t N, float* deg, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(N >= 0, "npi_row_weight_sum: bad size");
    if (N == 0) return NPI_OK;
    NPI_REQUIRE(rowptr && w && deg, "npi_row_weight_sum: null pointer");
    row_weight_sum_kernel<<<(unsigned)ceil_div(N, 4), 256, 0, stream>>>(rowptr, w, (int)N, deg);
    return check_launch("npi_row_weight_sum");
}

extern "C" int npi_gcn_norm(const int32_t* rowidx, const int32_t* col, const int32_t* rowptr,
                            const float* w, const float* deg, const int32_t* deg_rowptr, int64_t N,
                            int64_t nnz_max, float* norm, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(N >= 0 && nnz_max >= 0, "npi_gcn_norm: bad size");
    if (nnz_max == 0) return NPI_OK;
    NPI_REQUIRE(rowidx && col && rowptr && norm && (deg || deg_rowptr), "npi_gcn_norm: null pointer");
    gcn_norm_kernel<<<(unsigned)ceil_div(nnz_max, 256), 256, 0, stream>>>(rowidx, col, rowptr, w, deg, deg_rowptr, (int)N, nnz_max, norm);
    return check_launch("npi_gcn_norm");
}

extern "C" int npi_entry_weights(const int32_t* eid, const int32_t* rowidx, const int32_t* rowptr,
                                 const float* edge_w, const float* loop_w_node, float fill, int64_t N,
                                 int64_t nnz_max, float* w_entry, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(N >= 0 && nnz_max >= 0, "npi_entry_weights: bad size");
    if (nnz_max == 0) return NPI_OK;
    NPI_REQUIRE(eid && rowidx && rowptr && w_entry, "npi_entry_weights: null pointer");
    entry_weights_kernel<<<(unsigned)ceil_div(nnz_max, 256), 256, 0, stream>>>(eid, rowidx, rowptr, edge_w, loop_w_node, fill, (int)N, nnz_max, w_entry);
    return check_launch("npi_entry_weights");
}

extern "C" int npi_entry_col_scale(const int32_t* col, const int32_t* rowptr, const float* table, const float* w_in, int64_t N,
                                   int64_t n_cols, int64_t nnz_max, float* w_out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(N >= 0 && n_cols >= 0 && nnz_max >= 0 && N < 0x7fffffff && n_cols < 0x7fffffff, "npi_entry_col_scale: bad size");
    if (nnz_max == 0) return NPI_OK;
    NPI_REQUIRE(col && rowptr && table && w_out, "npi_entry_col_scale: null pointer");
    entry_col_scale_kernel<<<(unsigned)ceil_div(nnz_max, 256), 256, 0, stream>>>(col, rowptr, table, w_in, (int)N, (int)n_cols, nnz_max,
                                                                               w_out);
    return check_launch("npi_entry_col_scale");
}

extern "C" int npi_row_inv_count(const int32_t* rowptr, int64_t N, float* inv_cnt, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(N >= 0, "npi_row_inv_count: bad size");
    if (N == 0) return NPI_OK;
    NPI_REQUIRE(rowptr && inv_cnt, "npi_row_inv_count: null pointer");
    row_inv_count_kernel<<<(unsigned)ceil_div(N, 256), 256, 0, stream>>>(rowptr, (int)N, inv_cnt);
    return check_launch("npi_row_inv_count");
}

// grad of the pre-activation behind SAGEConv(..., relu=True): dz = dy where y > 0 else 0 (threshold_backward); y is the
// ReLU OUTPUT the layer saved.  Rows of F floats with pitches (the activations of a batch may be views); 16-byte lanes when
// everything is aligned, 4 elements per thread otherwise.
template <bool VEC4>
__global__ void __launch_bounds__(256)
relu_backward_kernel(const float* __restrict__ dy, int64_t ldd, const float* __restrict__ y, int64_t ldy, int64_t M, int64_t F,
                     float* __restrict__ dz, int64_t ldz) {
    const int64_t per = (F + 3) / 4;                                  // column groups of 4 per row
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * per) return;
    const int64_t r = i / per, c = (i % per) * 4;
    if (VEC4) {
        const float4 g = *reinterpret_cast<const float4*>(dy + r * ldd + c);
        const float4 v = *reinterpret_cast<const float4*>(y + r * ldy + c);
        *reinterpret_cast<float4*>(dz + r * ldz + c) = make_float4(v.x > 0.f ? g.x : 0.f, v.y > 0.f ? g.y : 0.f,
                                                                    v.z > 0.f ? g.z : 0.f, v.w > 0.f ? g.w : 0.f);
    } else {
        for (int q = 0; q < 4 && c + q < F; ++q) dz[r * ldz + c + q] = y[r * ldy + c + q] > 0.f ? dy[r * ldd + c + q] : 0.f;
    }
}

extern "C" int npi_relu_backward(const float* dy, int64_t ldd, const float* y, int64_t ldy, int64_t M, int64_t F, float* dz,
                                 int64_t ldz, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(M >= 0 && F > 0 && ldd >= F && ldy >= F && ldz >= F, "npi_relu_backward: bad size");
    if (M == 0) return NPI_OK;
    NPI_REQUIRE(dy && y && dz, "npi_relu_backward: null pointer");
    const int64_t n = M * ((F + 3) / 4);
    const bool v4 = F % 4 == 0 && ldd % 4 == 0 && ldy % 4 == 0 && ldz % 4 == 0 && ((uintptr_t)dy % 16) == 0 &&
                    ((uintptr_t)y % 16) == 0 && ((uintptr_t)dz % 16) == 0;
    if (v4) relu_backward_kernel<true><<<(unsigned)ceil_div(n, 256), 256, 0, stream>>>(dy, ldd, y, ldy, M, F, dz, ldz);
    else    relu_backward_kernel<false><<<(unsigned)ceil_div(n, 256), 256, 0, stream>>>(dy, ldd, y, ldy, M, F, dz, ldz);
    return check_launch("npi_relu_backward");
}

// SAGEConv(normalize=True): y_i = x_i / max(||x_i||_2, eps)  (torch.nn.functional.normalize(p = 2, dim = -1), PyG 1.4.2
// sage_conv.update).  One wavefront per row: lanes stride over the row in 16-byte pieces (or single floats when the row is not
// 16-byte aligned), a fixed xor tree adds the squares -- the same order on every run.  The backward of y = x / n:
// dx = (dy - y <dy, y>) / n where n > eps, dy / eps below it (the clamp is then the constant divisor), as torch's.
template <bool VEC4>
__global__ void __launch_bounds__(256)
l2_normalize_rows_kernel(const float* __restrict__ x, int64_t ldx, int64_t M, int F, float eps, float* __restrict__ y, int64_t ldy,
                         float* __restrict__ nrm) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= M) return;                                               // wave-uniform
    const int lane = lane_id();
    const float* xr = x + r * ldx;
    float ss = 0.f;
    if (VEC4) {
        for (int c = lane * 4; c < F; c += 4 * WAVE) {
            const float4 v = *reinterpret_cast<const float4*>(xr + c);
            ss = fmaf(v.x, v.x, ss); ss = fmaf(v.y, v.y, ss); ss = fmaf(v.z, v.z, ss); ss = fmaf(v.w, v.w, ss);
        }
    } else {
        for (int c = lane; c < F; c += WAVE) ss = fmaf(xr[c], xr[c], ss);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    const float n = sqrtf(ss);
    const float inv = 1.f / fmaxf(n, eps);
    if (lane == 0) nrm[r] = n;
    float* yr = y + r * ldy;
    if (VEC4) {
        for (int c = lane * 4; c < F; c += 4 * WAVE) {
            const float4 v = *reinterpret_cast<const float4*>(xr + c);
            *reinterpret_cast<float4*>(yr + c) = make_float4(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
        }
    } else {
        for (int c = lane; c < F; c += WAVE) yr[c] = xr[c] * inv;
    }
}

template <bool VEC4>
__global__ void __launch_bounds__(256)
l2_normalize_rows_bwd_kernel(const float* __restrict__ dy, int64_t ldd, const float* __restrict__ y, int64_t ldy,
                             const float* __restrict__ nrm, int64_t M, int F, float eps, float* __restrict__ dx, int64_t ldx) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= M) return;
    const int lane = lane_id();
    const float* gr = dy + r * ldd;
    const float* yr = y + r * ldy;
    const float n = nrm[r];
    const bool clamped = !(n > eps);                                  // the divisor was the constant eps: no projection term
    float dot = 0.f;
    if (!clamped) {
        if (VEC4) {
            for (int c = lane * 4; c < F; c += 4 * WAVE) {
                const float4 g = *reinterpret_cast<const float4*>(gr + c);
                const float4 v = *reinterpret_cast<const float4*>(yr + c);
                dot = fmaf(g.x, v.x, dot); dot = fmaf(g.y, v.y, dot); dot = fmaf(g.z, v.z, dot); dot = fmaf(g.w, v.w, dot);
            }
        } else {
            for (int c = lane; c < F; c += WAVE) dot = fmaf(gr[c], yr[c], dot);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
    }
    const float inv = 1.f / fmaxf(n, eps);
    float* xr = dx + r * ldx;
    if (VEC4) {
        for (int c = lane * 4; c < F; c += 4 * WAVE) {
            const float4 g = *reinterpret_cast<const float4*>(gr + c);
            const float4 v = *reinterpret_cast<const float4*>(yr + c);
            *reinterpret_cast<float4*>(xr + c) = make_float4((g.x - v.x * dot) * inv, (g.y - v.y * dot) * inv,
                                                             (g.z - v.z * dot) * inv, (g.w - v.w * dot) * inv);
        }
    } else {
        for (int c = lane; c < F; c += WAVE) xr[c] = (gr[c] - yr[c] * dot) * inv;
    }
}

static bool rows_vec4(const void* p, int64_t ld, int64_t F) { return F % 4 == 0 && ld % 4 == 0 && ((uintptr_t)p % 16) == 0; }

extern "C" int npi_l2_normalize_rows(const float* x, int64_t ldx, int64_t M, int64_t F, float eps, float* y, int64_t ldy,
                                     float* norm, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(M >= 0 && F > 0 && F < 0x7fffffff && ldx >= F && ldy >= F && eps > 0.f, "npi_l2_normalize_rows: bad size");
    if (M == 0) return NPI_OK;
    NPI_REQUIRE(x && y && norm, "npi_l2_normalize_rows: null pointer");
    const unsigned grid = (unsigned)ceil_div(M, 4);
    if (rows_vec4(x, ldx, F) && rows_vec4(y, ldy, F)) l2_normalize_rows_kernel<true><<<grid, 256, 0, stream>>>(x, ldx, M, (int)F, eps, y, ldy, norm);
    else                                              l2_normalize_rows_kernel<false><<<grid, 256, 0, stream>>>(x, ldx, M, (int)F, eps, y, ldy, norm);
    return check_launch("npi_l2_normalize_rows");
}

extern "C" int npi_l2_normalize_rows_bwd(const float* dy, int64_t ldd, const float* y, int64_t ldy, const float* norm, int64_t M,
                                         int64_t F, float eps, float* dx, int64_t ldx, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(M >= 0 && F > 0 && F < 0x7fffffff && ldd >= F && ldy >= F && ldx >= F && eps > 0.f, "npi_l2_normalize_rows_bwd: bad size");
    if (M == 0) return NPI_OK;
    NPI_REQUIRE(dy && y && norm && dx, "npi_l2_normalize_rows_bwd: null pointer");
    const unsigned grid = (unsigned)ceil_div(M, 4);
    if (rows_vec4(dy, ldd, F) && rows_vec4(y, ldy, F) && rows_vec4(dx, ldx, F))
        l2_normalize_rows_bwd_kernel<true><<<grid, 256, 0, stream>>>(dy, ldd, y, ldy, norm, M, (int)F, eps, dx, ldx);
    else
        l2_normalize_rows_bwd_kernel<false><<<grid, 256, 0, stream>>>(dy, ldd, y, ldy, norm, M, (int)F, eps, dx, ldx);
    return check_launch("npi_l2_normalize_rows_bwd");
}

// The [2, .] products around GATConv's rank-2 store epilogue (npi_linear_bwd_data_rank2), one head, as two small launches
// instead of three guarded GEMMs of two rows each (23 us apiece: a 128 x 128 tile walk for 2 x 256 outputs):
//   cols:  U[r, k] = sum_c W[k, c] att[r, c]                 (r = 0: att_dst, 1: att_src)  -- the column vectors of the epilogue
//   tail:  dW[k, c] += P[0, k] att[0, c] + P[1, k] att[1, c]   and   datt[r, c] = sum_k P[r, k] W[k, c]      (P = x^T [g_dst g_src])
// Fixed summation orders (a lane strides over c / a thread walks k), so run-to-run identical.
__global__ void __launch_bounds__(256)
gat_rank2_cols_kernel(const float* __restrict__ W, int64_t ldw, const float* __restrict__ att, int K, int C, float* __restrict__ U) {
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);               // one wavefront per row of W
    if (k >= K) return;
    const int lane = lane_id();
    float u0 = 0.f, u1 = 0.f;
    for (int c = lane; c < C; c += WAVE) {
        const float w = W[(int64_t)k * ldw + c];
        u0 = fmaf(w, att[c], u0);
        u1 = fmaf(w, att[C + c], u1);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { u0 += __shfl_xor(u0, o); u1 += __shfl_xor(u1, o); }
    if (lane == 0) { U[k] = u0; U[K + k] = u1; }
}

__global__ void __launch_bounds__(256)
gat_rank2_tail_kernel(const float* __restrict__ P, const float* __restrict__ W, int64_t ldw, const float* __restrict__ att, int K, int C,
                      float* __restrict__ dw, int64_t lddw, float* __restrict__ datt) {
    const int cb = (C + 255) / 256;                                   // column blocks
    if ((int)blockIdx.x < K * cb) {
        if (dw == nullptr) return;
        const int k = blockIdx.x / cb, c = (blockIdx.x % cb) * 256 + threadIdx.x;
        if (c < C) dw[(int64_t)k * lddw + c] += fmaf(P[k], att[c], P[K + k] * att[C + c]);
        return;
    }
    if (datt == nullptr) return;
    // one wavefront per column: lanes stride over k (W is small and cache resident), then a fixed xor tree
    const int c = ((int)blockIdx.x - K * cb) * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    const int lane = lane_id();
    float d0 = 0.f, d1 = 0.f;
    for (int k = lane; k < K; k += WAVE) {
        const float w = W[(int64_t)k * ldw + c];
        d0 = fmaf(P[k], w, d0);
        d1 = fmaf(P[K + k], w, d1);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { d0 += __shfl_xor(d0, o); d1 += __shfl_xor(d1, o); }
    if (lane == 0) { datt[c] = d0; datt[C + c] = d1; }
}

extern "C" int npi_gat_rank2_cols(const float* W, int64_t ldw, const float* att, int64_t K, int64_t C, float* U, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(K > 0 && C > 0 && K < 0x7fffffff && C < 0x7fffffff && ldw >= C, "npi_gat_rank2_cols: bad size");
    NPI_REQUIRE(W && att && U, "npi_gat_rank2_cols: null pointer");
    gat_rank2_cols_kernel<<<(unsigned)ceil_div(K, 4), 256, 0, stream>>>(W, ldw, att, (int)K, (int)C, U);
    return check_launch("npi_gat_rank2_cols");
}

extern "C" int npi_gat_rank2_tail(const float* P, const float* W, int64_t ldw, const float* att, int64_t K, int64_t C, float* dw,
                                  int64_t lddw, float* datt, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(K > 0 && C > 0 && K < (1 << 20) && C < (1 << 20) && ldw >= C && (dw == nullptr || lddw >= C), "npi_gat_rank2_tail: bad size");
    NPI_REQUIRE(P && W && att && (dw || datt), "npi_gat_rank2_tail: null pointer");
    const int64_t cb = ceil_div(C, 256);
    gat_rank2_tail_kernel<<<(unsigned)(K * cb + ceil_div(C, 4)), 256, 0, stream>>>(P, W, ldw, att, (int)K, (int)C, dw, lddw, datt);
    return check_launch("npi_gat_rank2_tail");
}

// ---- measurement support: a stand-in for a collective's RESIDENT kernel ----------------------------------------------------
// `workgroups` workgroups that each hold 64 KB of a CU's LDS (no 152 KB GEMM workgroup fits beside one) for `nanoseconds` of
// wall-clock time (wall_clock64: 100 MHz).  npi_gnn_amd.virtual.StubCollectives launches it on its copy stream to give a
// one-GPU run of a rank's step the duration of an exchange over xGMI and the CUs RCCL's kernel would sit on (bench.py,
// C4_w8_virtual.hubs_sage.emulated_wire); tools/occupy_probe.py times a GEMM beside it.  Computes nothing.
namespace npi {
__global__ void __launch_bounds__(256) hold_cus_kernel(long long ticks, unsigned long long* __restrict__ start_word) {
    __shared__ int h[16384];                           // 64 KB that nothing is computed with (kept alive by the asm at the end)
    __shared__ long long t0_s;
    h[threadIdx.x] = (int)threadIdx.x;
    if (threadIdx.x == 0) {
        long long t0 = wall_clock64();
        if (start_word != nullptr) {
            // the exchange lasts `ticks` from the moment its FIRST workgroup got a CU: a workgroup that had to wait for a wave
            // slot joins late and leaves with the others (each timing its own start would stretch the kernel by the wait)
            const unsigned long long old = atomicCAS(start_word, 0ull, (unsigned long long)t0);
            if (old != 0ull) t0 = (long long)old;
        }
        t0_s = t0;
    }
    __syncthreads();
    const long long t0 = t0_s;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
    asm volatile("; keep %0" :: "v"(h[(threadIdx.x * 7) & 255]));
}
}  // namespace npi
extern "C" int npi_hold_cus(int workgroups, int64_t nanoseconds, uint64_t* start_word, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(workgroups >= 0 && workgroups <= 256 && nanoseconds >= 0 && nanoseconds <= 100000000,
                "npi_hold_cus: 0..256 workgroups, at most 100 ms");
    if (workgroups == 0 || nanoseconds == 0) return NPI_OK;
    npi::hold_cus_kernel<<<(unsigned)workgroups, 256, 0, stream>>>((long long)(nanoseconds / 10),
                                                                  reinterpret_cast<unsigned long long*>(start_word));
    return check_launch("npi_hold_cus");
}

