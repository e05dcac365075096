// Dense feature projection on the CDNA4 matrix cores, exact f32 (v_mfma_f32_32x32x2_f32).
//
// Replaces `torch.matmul(aggr_out, self.weight) + self.bias` of PyG 1.4.2 SAGEConv.update /
// GCNConv.forward (reached from reference src/classes.py:62,66,70) and its autograd backward
// (src/train_with_twoDataset.PY:54):  dA = dC W^T,  dW = A^T dC,  db = colsum(dC).
//
// One kernel template, three operand layouts.  128x128 output tile per 256-thread workgroup,
// 2x2 wavefronts, each wavefront a 2x2 grid of 32x32 MFMA tiles (64 accumulator VGPRs), BK = 32,
// double-buffered LDS with the next tile's global loads issued before the current tile's MFMAs.
// LDS images are chosen so that fragment reads are bank-conflict free:
//   operand contiguous along K in memory -> image [row][BK+4], fragment = one ds_read_b128 holding
//       k = 8g + 4h + {0,1,2,3}  (h = lane>>5) -- the k order inside a group of 8 is permuted the
//       same way for A and B, which a sum over k does not care about;
//   operand contiguous along M/N in memory -> image [k][128+4], fragment = ds_read_b32 per k.
#include "npi_common.h"

namespace npi {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int GEMM_THREADS = 256;
constexpr int KPITCH = BK + 4;      // [row][k] image
constexpr int RPITCH = 128 + 4;     // [k][row] image
constexpr int TILE_FLOATS = 128 * KPITCH;   // 4608 >= 32 * RPITCH (4224)

// ---- global -> register staging -------------------------------------------------------------
// K-contiguous operand: element (row, k) at base[row * ld + k]; tile rows [r0, r0+128), k [k0, k0+32)
template <bool VEC4>
__device__ __forceinline__ void gload_kcontig(const float* __restrict__ base, int64_t ld, int r0,
                                              int rmax, int k0, int kmax, float4 (&reg)[4]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int row = r0 + (t >> 3) + 32 * p;
        const int k = k0 + (t & 7) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < rmax) {
            const float* src = base + (int64_t)row * ld + k;
            if (VEC4) {
                if (k < kmax) v = *reinterpret_cast<const float4*>(src);
            } else {
                if (k + 0 < kmax) v.x = src[0];
                if (k + 1 < kmax) v.y = src[1];
                if (k + 2 < kmax) v.z = src[2];
                if (k + 3 < kmax) v.w = src[3];
            }
        }
        reg[p] = v;
    }
}
__device__ __forceinline__ void lstore_kcontig(float* __restrict__ img, const float4 (&reg)[4]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int row = (t >> 3) + 32 * p;
        *reinterpret_cast<float4*>(img + row * KPITCH + (t & 7) * 4) = reg[p];
    }
}
// row-contiguous operand: element (k, c) at base[k * ld + c]; tile k [k0,k0+32), c [c0, c0+128)
template <bool VEC4>
__device__ __forceinline__ void gload_rcontig(const float* __restrict__ base, int64_t ld, int c0,
                                              int cmax, int k0, int kmax, float4 (&reg)[4]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int k = k0 + (t >> 5) + 8 * p;
        const int c = c0 + (t & 31) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < kmax) {
            const float* src = base + (int64_t)k * ld + c;
            if (VEC4) {
                if (c < cmax) v = *reinterpret_cast<const float4*>(src);
            } else {
                if (c + 0 < cmax) v.x = src[0];
                if (c + 1 < cmax) v.y = src[1];
                if (c + 2 < cmax) v.z = src[2];
                if (c + 3 < cmax) v.w = src[3];
            }
        }
        reg[p] = v;
    }
}
__device__ __forceinline__ void lstore_rcontig(float* __restrict__ img, const float4 (&reg)[4]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int k = (t >> 5) + 8 * p;
        *reinterpret_cast<float4*>(img + k * RPITCH + (t & 31) * 4) = reg[p];
    }
}

struct Epilogue {
    const float* bias;       // [N] or null
    const float* rowscale;   // [M] or null
    int relu;
    float* colsum;           // BMODE 0 only: per-split column sums of B, [gridDim.z][N], or null
};

// C[M,N] (+ split-K slabs) = A(m,k) * B(k,n)
//   AMODE 0: A(m,k) = A[m*lda + k]     AMODE 1: A(m,k) = A[k*lda + m]
//   BMODE 0: B(k,n) = B[k*ldb + n]     BMODE 1: B(k,n) = B[n*ldb + k]
template <int AMODE, int BMODE, bool VEC4>
__global__ void __launch_bounds__(GEMM_THREADS)
gemm_f32_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
                float* __restrict__ C, int64_t ldc, int M, int N, int K, int kchunk,
                int64_t slab_stride, Epilogue ep) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][TILE_FLOATS];
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int n0 = blockIdx.x * BN;
    const int m0 = blockIdx.y * BM;
    const int kbeg = blockIdx.z * kchunk;
    const int kend = min(K, kbeg + kchunk);
    C += (int64_t)blockIdx.z * slab_stride;
    const bool do_colsum = (ep.colsum != nullptr) && (blockIdx.y == 0);
    float csum = 0.f;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    float4 ra[4], rb[4];
    auto gload = [&](int k0) {
        if (AMODE == 0) gload_kcontig<VEC4>(A, lda, m0, M, k0, kend, ra);
        else            gload_rcontig<VEC4>(A, lda, m0, M, k0, kend, ra);
        if (BMODE == 0) gload_rcontig<VEC4>(B, ldb, n0, N, k0, kend, rb);
        else            gload_kcontig<VEC4>(B, ldb, n0, N, k0, kend, rb);
    };
    auto lstore = [&](int buf) {
        if (AMODE == 0) lstore_kcontig(lds[buf][0], ra); else lstore_rcontig(lds[buf][0], ra);
        if (BMODE == 0) lstore_rcontig(lds[buf][1], rb); else lstore_kcontig(lds[buf][1], rb);
    };

    const int nk = (kend > kbeg) ? (kend - kbeg + BK - 1) / BK : 0;
    if (nk > 0) {
        gload(kbeg);
        lstore(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kbeg + (kt + 1) * BK);
        const float* as = lds[buf][0];
        const float* bs = lds[buf][1];
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            float af[2][4], bf[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wm * 64 + i * 32 + li;
                if (AMODE == 0) {
                    float4 v = *reinterpret_cast<const float4*>(as + row * KPITCH + g * 8 + lh * 4);
                    af[i][0] = v.x; af[i][1] = v.y; af[i][2] = v.z; af[i][3] = v.w;
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) af[i][s] = as[(g * 8 + lh * 4 + s) * RPITCH + row];
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c = wn * 64 + j * 32 + li;
                if (BMODE == 0) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) bf[j][s] = bs[(g * 8 + lh * 4 + s) * RPITCH + c];
                } else {
                    float4 v = *reinterpret_cast<const float4*>(bs + c * KPITCH + g * 8 + lh * 4);
                    bf[j][0] = v.x; bf[j][1] = v.y; bf[j][2] = v.z; bf[j][3] = v.w;
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
        }
        if (BMODE == 0 && do_colsum && threadIdx.x < BN) {   // db: column sums of the staged dC tile
#pragma unroll 8
            for (int k = 0; k < BK; ++k) csum += bs[k * RPITCH + threadIdx.x];
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }
    if (BMODE == 0 && do_colsum && threadIdx.x < BN && n0 + (int)threadIdx.x < N)
        ep.colsum[(int64_t)blockIdx.z * N + n0 + threadIdx.x] = csum;

    // C/D map of the 32x32 MFMA: col = lane & 31, row = (q & 3) + 8 (q >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = n0 + wn * 64 + j * 32 + li;
            if (c >= N) continue;
            const float b = ep.bias ? ep.bias[c] : 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int r = m0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh;
                if (r < M) {
                    float v = acc[i][j][q];
                    if (ep.rowscale) v *= ep.rowscale[r];
                    v += b;
                    if (ep.relu) v = fmaxf(v, 0.f);
                    C[(int64_t)r * ldc + c] = v;
                }
            }
        }
}

// out[r, c] = sum_z slabs[z][r, c]  (z ascending: deterministic)
__global__ void slab_reduce_kernel(const float* __restrict__ slabs, int64_t slab_stride, int nslab,
                                   int rows, int cols, int64_t ld_slab, float* __restrict__ out, int64_t ldo) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows * cols) return;
    int r = (int)(i / cols), c = (int)(i % cols);
    float s = 0.f;
    for (int z = 0; z < nslab; ++z) s += slabs[(int64_t)z * slab_stride + (int64_t)r * ld_slab + c];
    out[(int64_t)r * ldo + c] = s;
}

// partial column sums of X[M, N] over row chunks: part[z][c]
__global__ void __launch_bounds__(256)
colsum_partial_kernel(const float* __restrict__ X, int64_t ldx, int M, int N, int rows_per_block,
                      float* __restrict__ part) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int sub = threadIdx.x >> 6;
    const int rbeg = blockIdx.y * rows_per_block;
    const int rend = min(M, rbeg + rows_per_block);
    float s = 0.f;
    if (c < N)
        for (int r = rbeg + sub; r < rend; r += 4) s += X[(int64_t)r * ldx + c];
    red[sub][threadIdx.x & 63] = s;
    __syncthreads();
    if (sub == 0 && c < N)
        part[(int64_t)blockIdx.y * N + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

static bool vec4_ok(const void* p, int64_t ld, int64_t inner_extent) {
    return ((uintptr_t)p % 16 == 0) && (ld % 4 == 0) && (inner_extent % 4 == 0);
}

constexpr int COLSUM_ROWS = 4096;

static int pick_splits(int64_t M, int64_t tiles) {
    // dW: reduction over M (nodes).  Aim for ~4 workgroups per CU, at least 8 K-tiles each.
    int64_t want = ceil_div(1024, tiles);
    int64_t maxs = ceil_div(M, (int64_t)BK * 8);
    int64_t s = want < maxs ? want : maxs;
    return (int)(s < 1 ? 1 : s);
}

}  // namespace npi

using namespace npi;

extern "C" int npi_linear_fwd(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias,
                              const float* rowscale, float* C, int64_t ldc, int64_t M, int64_t K,
                              int64_t N, int relu, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(M >= 0 && K > 0 && N > 0, "npi_linear_fwd: bad size");
    NPI_REQUIRE(M < 0x7fffffff && K < 0x7fffffff && N < 0x7fffffff, "npi_linear_fwd: size > int32");
    if (M == 0) return NPI_OK;
    NPI_REQUIRE(A && W && C, "npi_linear_fwd: null pointer");
    NPI_REQUIRE(lda >= K && ldw >= N && ldc >= N, "npi_linear_fwd: leading dimension too small");
    dim3 grid((unsigned)ceil_div(N, BN), (unsigned)ceil_div(M, BM), 1);
    Epilogue ep{bias, rowscale, relu, nullptr};
    const bool v4 = vec4_ok(A, lda, K) && vec4_ok(W, ldw, N);
    if (v4) gemm_f32_kernel<0, 0, true><<<grid, GEMM_THREADS, 0, stream>>>(A, lda, W, ldw, C, ldc, (int)M, (int)N, (int)K, (int)K, 0, ep);
    else    gemm_f32_kernel<0, 0, false><<<grid, GEMM_THREADS, 0, stream>>>(A, lda, W, ldw, C, ldc, (int)M, (int)N, (int)K, (int)K, 0, ep);
    return check_launch("npi_linear_fwd");
}

// dA[M,K] = rowscale * (dC[M,N] @ W[K,N]^T): GEMM with "K" = N (contracted), output width K
extern "C" int npi_linear_bwd_data(const float* dC, int64_t lddc, const float* W, int64_t ldw,
                                   const float* rowscale, float* dA, int64_t ldda, int64_t M, int64_t K,
                                   int64_t N, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(M >= 0 && K > 0 && N > 0, "npi_linear_bwd_data: bad size");
    NPI_REQUIRE(M < 0x7fffffff && K < 0x7fffffff && N < 0x7fffffff, "npi_linear_bwd_data: size > int32");
    if (M == 0) return NPI_OK;
    NPI_REQUIRE(dC && W && dA, "npi_linear_bwd_data: null pointer");
    NPI_REQUIRE(lddc >= N && ldw >= N && ldda >= K, "npi_linear_bwd_data: leading dimension too small");
    dim3 grid((unsigned)ceil_div(K, BN), (unsigned)ceil_div(M, BM), 1);
    Epilogue ep{nullptr, rowscale, 0, nullptr};
    // B(k = n_contract, n = k_out) = W[k_out * ldw + n_contract]  -> BMODE 1
    const bool v4 = vec4_ok(dC, lddc, N) && vec4_ok(W, ldw, N);
    if (v4) gemm_f32_kernel<0, 1, true><<<grid, GEMM_THREADS, 0, stream>>>(dC, lddc, W, ldw, dA, ldda, (int)M, (int)K, (int)N, (int)N, 0, ep);
    else    gemm_f32_kernel<0, 1, false><<<grid, GEMM_THREADS, 0, stream>>>(dC, lddc, W, ldw, dA, ldda, (int)M, (int)K, (int)N, (int)N, 0, ep);
    return check_launch("npi_linear_bwd_data");
}

extern "C" int npi_colsum(const float* X, int64_t ldx, int64_t M, int64_t N, float* out, float* workspace,
                          int64_t workspace_elems, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(M >= 0 && N > 0 && M < 0x7fffffff && N < 0x7fffffff, "npi_colsum: bad size");
    NPI_REQUIRE(X && out && workspace && ldx >= N, "npi_colsum: bad argument");
    const int nchunks = (int)ceil_div(M > 0 ? M : 1, COLSUM_ROWS);
    if (workspace_elems < (int64_t)nchunks * N) {
        set_error("npi_colsum: workspace too small");
        return NPI_ERR_WORKSPACE;
    }
    dim3 cg((unsigned)ceil_div(N, 64), (unsigned)nchunks);
    colsum_partial_kernel<<<cg, 256, 0, stream>>>(X, ldx, (int)M, (int)N, COLSUM_ROWS, workspace);
    slab_reduce_kernel<<<(unsigned)ceil_div(N, 256), 256, 0, stream>>>(workspace, N, nchunks, 1, (int)N, N, out, N);
    return check_launch("npi_colsum");
}

extern "C" int64_t npi_linear_bwd_weight_workspace_elems(int64_t M, int64_t K, int64_t N) {
    if (M < 0 || K <= 0 || N <= 0) return -1;
    int64_t tiles = ceil_div(K, BM) * ceil_div(N, BN);
    int64_t splits = pick_splits(M, tiles);
    return splits * K * N + splits * N + 64;      // dW slabs, then db slabs
}

// dW[K,N] = A[M,K]^T @ dC[M,N] (contract over M), db[N] = colsum(dC)
extern "C" int npi_linear_bwd_weight(const float* A, int64_t lda, const float* dC, int64_t lddc,
                                     float* dW, int64_t lddw, float* db, int64_t M, int64_t K, int64_t N,
                                     float* workspace, int64_t workspace_elems, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(M >= 0 && K > 0 && N > 0, "npi_linear_bwd_weight: bad size");
    NPI_REQUIRE(M < 0x7fffffff && K < 0x7fffffff && N < 0x7fffffff, "npi_linear_bwd_weight: size > int32");
    NPI_REQUIRE(A && dC && dW && workspace, "npi_linear_bwd_weight: null pointer");
    NPI_REQUIRE(lda >= K && lddc >= N && lddw >= N, "npi_linear_bwd_weight: leading dimension too small");
    if (workspace_elems < npi_linear_bwd_weight_workspace_elems(M, K, N)) {
        set_error("npi_linear_bwd_weight: workspace too small");
        return NPI_ERR_WORKSPACE;
    }
    const int64_t tiles = ceil_div(K, BM) * ceil_div(N, BN);
    const int splits = pick_splits(M, tiles);
    const int kchunk = (int)(ceil_div(ceil_div(M > 0 ? M : 1, splits), BK) * BK);
    dim3 grid((unsigned)ceil_div(N, BN), (unsigned)ceil_div(K, BM), (unsigned)splits);
    float* db_slabs = workspace + (int64_t)splits * K * N;
    Epilogue ep{nullptr, nullptr, 0, db ? db_slabs : nullptr};   // db fused: colsum of the staged dC tiles
    // output rows = K (features of A), cols = N, contraction over M:  A(m=k_feat, k=node) = A[node*lda + k_feat]
    const bool v4 = vec4_ok(A, lda, K) && vec4_ok(dC, lddc, N);
    if (v4) gemm_f32_kernel<1, 0, true><<<grid, GEMM_THREADS, 0, stream>>>(A, lda, dC, lddc, workspace, N, (int)K, (int)N, (int)M, kchunk, K * N, ep);
    else    gemm_f32_kernel<1, 0, false><<<grid, GEMM_THREADS, 0, stream>>>(A, lda, dC, lddc, workspace, N, (int)K, (int)N, (int)M, kchunk, K * N, ep);
    slab_reduce_kernel<<<(unsigned)ceil_div(K * N, 256), 256, 0, stream>>>(workspace, K * N, splits, (int)K, (int)N, N, dW, lddw);
    if (db) slab_reduce_kernel<<<(unsigned)ceil_div(N, 256), 256, 0, stream>>>(db_slabs, N, splits, 1, (int)N, N, db, N);
    return check_launch("npi_linear_bwd_weight");
}
