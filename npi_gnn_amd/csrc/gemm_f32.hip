// Dense feature projection on the CDNA4 matrix cores, exact f32 (v_mfma_f32_32x32x2_f32).
//
// Replaces `torch.matmul(aggr_out, self.weight) + self.bias` of PyG 1.4.2 SAGEConv.update /
// GCNConv.forward (reached from reference src/classes.py:62,66,70) and its autograd backward
// (src/train_with_twoDataset.PY:54):  dA = dC W^T,  dW = A^T dC,  db = colsum(dC).
//
// 128x128 output tile per 256-thread workgroup, 2x2 wavefronts, each a 2x2 grid of 32x32 MFMA
// tiles (64 accumulator VGPRs), BK = 32, double-buffered LDS, 2 workgroups per CU.
// Two kernels share the tile code:
//   FAST  : full interior tiles with K % 32 == 0 and 16-B aligned rows.  Every staging load is
//           `global_load_dwordx4 v, v_off, s[base]` -- a per-lane 32-bit byte offset computed once
//           and a scalar base bumped per k-step -- so a k-step is ONE basic block without address
//           VALU or exec-mask branches and the loads / LDS stores interleave with the 64 MFMAs.
//           (A bare MFMA loop sustains 155 TF on this chip; what separates a tiled kernel from it
//           is the issue time of everything that is not an MFMA, not bandwidth or latency.)
//   EDGE  : the same tile with guarded loads/stores, for the ragged strips at the matrix edges,
//           K tails and unaligned operands (F = 178, 65).
// LDS images are chosen so that fragment reads are bank-conflict free (SQ_LDS_BANK_CONFLICT = 0):
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

struct Epilogue {
    const float* bias;       // [N] or null
    const float* rowscale;   // [M] or null
    int relu;
    float* colsum;           // BMODE 0 only: per-split column sums of B, [splits][N], or null
};

// ---- guarded global -> register staging (EDGE kernel) ---------------------------------------------
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
__device__ __forceinline__ void lstore_kcontig(float* __restrict__ img, const float4 (&reg)[4]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int p = 0; p < 4; ++p)
        *reinterpret_cast<float4*>(img + ((t >> 3) + 32 * p) * KPITCH + (t & 7) * 4) = reg[p];
}
__device__ __forceinline__ void lstore_rcontig(float* __restrict__ img, const float4 (&reg)[4]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int p = 0; p < 4; ++p)
        *reinterpret_cast<float4*>(img + ((t >> 5) + 8 * p) * RPITCH + (t & 31) * 4) = reg[p];
}

// ---- one k-step of MFMAs on the staged tile ----------------------------------------------------------
template <int AMODE, int BMODE>
__device__ __forceinline__ void mma_step(const float* __restrict__ as, const float* __restrict__ bs,
                                         int wm, int wn, int li, int lh, f32x16 (&acc)[2][2]) {
    float af[2][2][4], bf[2][2][4];
    auto read_frags = [&](int g, float (&fa)[2][4], float (&fb)[2][4]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wm * 64 + i * 32 + li;
            if (AMODE == 0) {
                float4 v = *reinterpret_cast<const float4*>(as + row * KPITCH + g * 8 + lh * 4);
                fa[i][0] = v.x; fa[i][1] = v.y; fa[i][2] = v.z; fa[i][3] = v.w;
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) fa[i][s] = as[(g * 8 + lh * 4 + s) * RPITCH + row];
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = wn * 64 + j * 32 + li;
            if (BMODE == 0) {
#pragma unroll
                for (int s = 0; s < 4; ++s) fb[j][s] = bs[(g * 8 + lh * 4 + s) * RPITCH + c];
            } else {
                float4 v = *reinterpret_cast<const float4*>(bs + c * KPITCH + g * 8 + lh * 4);
                fb[j][0] = v.x; fb[j][1] = v.y; fb[j][2] = v.z; fb[j][3] = v.w;
            }
        }
    };
    read_frags(0, af[0], bf[0]);
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
        const int cur = g & 1;
        if (g + 1 < BK / 8) read_frags(g + 1, af[cur ^ 1], bf[cur ^ 1]);   // one k-group ahead
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i][s], bf[cur][j][s], acc[i][j], 0, 0, 0);
    }
}

// ---- epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (q & 3) + 8 (q >> 2) + 4 (lane >> 5).
// Every operand is fetched BEFORE the store loop: a load inside it makes hipcc wait vmcnt(0) per
// element, which also drains the preceding store (64 serialised stores per lane).
template <bool GUARD>
__device__ __forceinline__ void store_tile(float* __restrict__ C, int64_t ldc, int M, int N, int m0, int n0,
                                           int wm, int wn, int li, int lh, const f32x16 (&acc)[2][2],
                                           const Epilogue& ep) {
    float rsv[2][16];
    float bv[2];
    if (ep.rowscale != nullptr) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int r = m0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh;
                rsv[i][q] = ep.rowscale[GUARD ? min(r, M - 1) : r];
            }
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) rsv[i][q] = 1.f;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = n0 + wn * 64 + j * 32 + li;
        bv[j] = (ep.bias != nullptr) ? ep.bias[GUARD ? min(c, N - 1) : c] : 0.f;
    }
    const bool relu_on = ep.relu != 0;
    float* __restrict__ cbase = C + (int64_t)(m0 + wm * 64 + 4 * lh) * ldc + (n0 + wn * 64 + li);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int ro = i * 32 + (q & 3) + 8 * (q >> 2);
                float v = fmaf(acc[i][j][q], rsv[i][q], bv[j]);
                v = (relu_on && v < 0.f) ? 0.f : v;          // keeps NaN, like torch.relu
                if (!GUARD) {
                    cbase[(int64_t)ro * ldc + j * 32] = v;
                } else {
                    const int r = m0 + wm * 64 + 4 * lh + ro, c = n0 + wn * 64 + j * 32 + li;
                    if (r < M && c < N) cbase[(int64_t)ro * ldc + j * 32] = v;
                }
            }
}

// C[M,N] (+ split-K slabs) = A(m,k) * B(k,n) over the tile grid starting at (tm0, tn0)
//   AMODE 0: A(m,k) = A[m*lda + k]     AMODE 1: A(m,k) = A[k*lda + m]
//   BMODE 0: B(k,n) = B[k*ldb + n]     BMODE 1: B(k,n) = B[n*ldb + k]
//   split z = blockIdx.z contracts k in [z*kchunk, min(K, (z+1)*kchunk)) into slab z
struct GemmArgs {
    const float* A; int64_t lda;
    const float* B; int64_t ldb;
    float* C; int64_t ldc;
    int M, N, K;            // extents of the m / n / contraction index
    int kchunk;             // multiple of BK
    int tm0, tn0;           // first tile of this launch's grid
    int64_t slab_stride;
    Epilogue ep;
};

template <int AMODE, int BMODE>
__global__ void __launch_bounds__(GEMM_THREADS, 2)
gemm_fast_kernel(GemmArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][TILE_FLOATS];
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int n0 = (a.tn0 + (int)blockIdx.x) * BN;
    const int m0 = (a.tm0 + (int)blockIdx.y) * BM;
    const int z = blockIdx.z;
    const int kbeg = z * a.kchunk;
    const int kend = min(a.K, kbeg + a.kchunk);
    const int nk = max(kend - kbeg, 0) / BK;                 // K % BK == 0 on this path

    // per-lane byte offsets inside a tile slab (computed once) + scalar bases (bumped per k-step)
    uint32_t voa[4], vob[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        voa[p] = (AMODE == 0) ? (uint32_t)((((t >> 3) + 32 * p) * a.lda + (t & 7) * 4) * 4)
                              : (uint32_t)((((t >> 5) + 8 * p) * a.lda + (t & 31) * 4) * 4);
        vob[p] = (BMODE == 0) ? (uint32_t)((((t >> 5) + 8 * p) * a.ldb + (t & 31) * 4) * 4)
                              : (uint32_t)((((t >> 3) + 32 * p) * a.ldb + (t & 7) * 4) * 4);
    }
    const char* sa = reinterpret_cast<const char*>(
        (AMODE == 0) ? a.A + (int64_t)m0 * a.lda + kbeg : a.A + (int64_t)kbeg * a.lda + m0);
    const char* sb = reinterpret_cast<const char*>(
        (BMODE == 0) ? a.B + (int64_t)kbeg * a.ldb + n0 : a.B + (int64_t)n0 * a.ldb + kbeg);
    const int64_t step_a = ((AMODE == 0) ? (int64_t)BK : (int64_t)BK * a.lda) * 4;
    const int64_t step_b = ((BMODE == 0) ? (int64_t)BK * a.ldb : (int64_t)BK) * 4;

    // Staging registers are named scalars (not arrays behind a lambda) so they stay in VGPRs.
    // readfirstlane keeps the bases in SGPRs; loop strength reduction would otherwise give every
    // load address its own 64-bit VGPR induction variable.
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define NPI_GLOAD()                                                                   \
    do {                                                                              \
        const char* ua = uniform_ptr(sa);                                             \
        const char* ub = uniform_ptr(sb);                                             \
        ra0 = *reinterpret_cast<const float4*>(ua + voa[0]);                          \
        ra1 = *reinterpret_cast<const float4*>(ua + voa[1]);                          \
        ra2 = *reinterpret_cast<const float4*>(ua + voa[2]);                          \
        ra3 = *reinterpret_cast<const float4*>(ua + voa[3]);                          \
        rb0 = *reinterpret_cast<const float4*>(ub + vob[0]);                          \
        rb1 = *reinterpret_cast<const float4*>(ub + vob[1]);                          \
        rb2 = *reinterpret_cast<const float4*>(ub + vob[2]);                          \
        rb3 = *reinterpret_cast<const float4*>(ub + vob[3]);                          \
        sa += step_a;                                                                 \
        sb += step_b;                                                                 \
    } while (0)
    // LDS byte offsets of this lane's four 16-B staging slots per operand (fixed for the kernel)
    const int la = (AMODE == 0) ? ((t >> 3) * KPITCH + (t & 7) * 4) : ((t >> 5) * RPITCH + (t & 31) * 4);
    const int lb = (BMODE == 0) ? ((t >> 5) * RPITCH + (t & 31) * 4) : ((t >> 3) * KPITCH + (t & 7) * 4);
    constexpr int LSA = (AMODE == 0) ? 32 * KPITCH : 8 * RPITCH;
    constexpr int LSB = (BMODE == 0) ? 8 * RPITCH : 32 * KPITCH;
#define NPI_LSTORE(buf)                                                               \
    do {                                                                              \
        float* pa = lds[buf][0] + la;                                                 \
        float* pb = lds[buf][1] + lb;                                                 \
        *reinterpret_cast<float4*>(pa + 0 * LSA) = ra0;                               \
        *reinterpret_cast<float4*>(pa + 1 * LSA) = ra1;                               \
        *reinterpret_cast<float4*>(pa + 2 * LSA) = ra2;                               \
        *reinterpret_cast<float4*>(pa + 3 * LSA) = ra3;                               \
        *reinterpret_cast<float4*>(pb + 0 * LSB) = rb0;                               \
        *reinterpret_cast<float4*>(pb + 1 * LSB) = rb1;                               \
        *reinterpret_cast<float4*>(pb + 2 * LSB) = rb2;                               \
        *reinterpret_cast<float4*>(pb + 3 * LSB) = rb3;                               \
    } while (0)

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    const bool do_colsum = (BMODE == 0) && (a.ep.colsum != nullptr) && (m0 == 0);
    float csum = 0.f;

    if (nk > 0) {
        NPI_GLOAD();
        NPI_LSTORE(0);
    }
    __syncthreads();
    int buf = 0;
    for (int kt = 0; kt + 1 < nk; ++kt) {            // steady state
        NPI_GLOAD();
        // keep the loads at the TOP of the k-step: hipcc otherwise sinks them below the MFMAs and
        // then waits for them at once, exposing the whole memory latency every step
        __builtin_amdgcn_sched_barrier(0);
        mma_step<AMODE, BMODE>(lds[buf][0], lds[buf][1], wm, wn, li, lh, acc);
        if (do_colsum && t < BN) {
#pragma unroll 8
            for (int k = 0; k < BK; ++k) csum += lds[buf][1][k * RPITCH + t];
        }
        __builtin_amdgcn_sched_barrier(0);
        NPI_LSTORE(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
#undef NPI_GLOAD
#undef NPI_LSTORE
    if (nk > 0) {                                    // last k-step: nothing left to stage
        mma_step<AMODE, BMODE>(lds[buf][0], lds[buf][1], wm, wn, li, lh, acc);
        if (do_colsum && t < BN) {
#pragma unroll 8
            for (int k = 0; k < BK; ++k) csum += lds[buf][1][k * RPITCH + t];
        }
    }
    if (do_colsum && t < BN) a.ep.colsum[(int64_t)z * a.N + n0 + t] = csum;
    store_tile<false>(a.C + (int64_t)z * a.slab_stride, a.ldc, a.M, a.N, m0, n0, wm, wn, li, lh, acc, a.ep);
}

template <int AMODE, int BMODE, bool VEC4>
__global__ void __launch_bounds__(GEMM_THREADS, 2)
gemm_edge_kernel(GemmArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][TILE_FLOATS];
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int n0 = (a.tn0 + (int)blockIdx.x) * BN;
    const int m0 = (a.tm0 + (int)blockIdx.y) * BM;
    const int z = blockIdx.z;
    const int kbeg = z * a.kchunk;
    const int kend = min(a.K, kbeg + a.kchunk);
    const int nk = (kend > kbeg) ? (kend - kbeg + BK - 1) / BK : 0;

    float4 ra[4], rb[4];
    auto gload = [&](int k0) {
        if (AMODE == 0) gload_kcontig<VEC4>(a.A, a.lda, m0, a.M, k0, kend, ra);
        else            gload_rcontig<VEC4>(a.A, a.lda, m0, a.M, k0, kend, ra);
        if (BMODE == 0) gload_rcontig<VEC4>(a.B, a.ldb, n0, a.N, k0, kend, rb);
        else            gload_kcontig<VEC4>(a.B, a.ldb, n0, a.N, k0, kend, rb);
    };
    auto lstore = [&](int buf) {
        if (AMODE == 0) lstore_kcontig(lds[buf][0], ra); else lstore_rcontig(lds[buf][0], ra);
        if (BMODE == 0) lstore_rcontig(lds[buf][1], rb); else lstore_kcontig(lds[buf][1], rb);
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    const bool do_colsum = (BMODE == 0) && (a.ep.colsum != nullptr) && (m0 == 0);
    float csum = 0.f;

    if (nk > 0) {
        gload(kbeg);
        lstore(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kbeg + (kt + 1) * BK);
        mma_step<AMODE, BMODE>(lds[buf][0], lds[buf][1], wm, wn, li, lh, acc);
        if (do_colsum && t < BN) {
#pragma unroll 8
            for (int k = 0; k < BK; ++k) csum += lds[buf][1][k * RPITCH + t];
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }
    if (do_colsum && t < BN && n0 + t < a.N) a.ep.colsum[(int64_t)z * a.N + n0 + t] = csum;
    store_tile<true>(a.C + (int64_t)z * a.slab_stride, a.ldc, a.M, a.N, m0, n0, wm, wn, li, lh, acc, a.ep);
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

// partial column sums of X[M, N] over row chunks: part[chunk][c].  Lanes walk a row 16 B each
// (a 256-column row is one 1 KiB wave instruction); the 4 waves of a workgroup take rows r, r+1, ...
constexpr int COLSUM_ROWS = 2048;
template <bool VEC4>
__global__ void __launch_bounds__(256)
colsum_partial_kernel(const float* __restrict__ X, int64_t ldx, int M, int N, float* __restrict__ part) {
    __shared__ float red[4][256];
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + lane * 4;
    const int rbeg = blockIdx.y * COLSUM_ROWS;
    const int rend = min(M, rbeg + COLSUM_ROWS);
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < N) {
        for (int r = rbeg + wave; r < rend; r += 4) {
            const float* src = X + (int64_t)r * ldx + c;
            if (VEC4) {
                float4 v = *reinterpret_cast<const float4*>(src);
                s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) if (c + q < N) s[q] += src[q];
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) red[wave][lane * 4 + q] = s[q];
    __syncthreads();
    const int cc = blockIdx.x * 256 + threadIdx.x;
    if (cc < N)
        part[(int64_t)blockIdx.y * N + cc] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

static bool vec4_ok(const void* p, int64_t ld, int64_t inner_extent) {
    return ((uintptr_t)p % 16 == 0) && (ld % 4 == 0) && (inner_extent % 4 == 0);
}

static int pick_splits(int64_t M, int64_t tiles) {
    // dW: reduction over M (nodes).  Aim for ~4 tiles per CU, at least 8 k-steps each.
    int64_t want = ceil_div(1024, tiles);
    int64_t maxs = ceil_div(M, (int64_t)BK * 8);
    int64_t s = want < maxs ? want : maxs;
    return (int)(s < 1 ? 1 : s);
}

// Cover the (M x N) tile grid with the FAST kernel on full tiles and the EDGE kernel on the two
// ragged strips.  `splits` slabs along the contraction; the fast path needs K % BK == 0.
template <int AMODE, int BMODE>
static void launch_gemm(bool v4, GemmArgs a, int splits, hipStream_t stream) {
    const int tm = (int)ceil_div(a.M, BM), tn = (int)ceil_div(a.N, BN);
    const bool fast_ok = v4 && (a.K % BK == 0) && (a.K > 0);
    const int fm = fast_ok ? a.M / BM : 0, fn = fast_ok ? a.N / BN : 0;    // full tiles
    if (fm > 0 && fn > 0) {
        GemmArgs f = a;
        f.tm0 = 0; f.tn0 = 0;
        gemm_fast_kernel<AMODE, BMODE><<<dim3(fn, fm, splits), GEMM_THREADS, 0, stream>>>(f);
    }
    auto edge = [&](int tm0, int tn0, int nm, int nn) {
        if (nm <= 0 || nn <= 0) return;
        GemmArgs e = a;
        e.tm0 = tm0; e.tn0 = tn0;
        if (v4) gemm_edge_kernel<AMODE, BMODE, true><<<dim3(nn, nm, splits), GEMM_THREADS, 0, stream>>>(e);
        else    gemm_edge_kernel<AMODE, BMODE, false><<<dim3(nn, nm, splits), GEMM_THREADS, 0, stream>>>(e);
    };
    if (fm > 0 && fn > 0) {
        edge(fm, 0, tm - fm, tn);        // bottom strip (all columns)
        edge(0, fn, fm, tn - fn);        // right strip (full rows only)
    } else {
        edge(0, 0, tm, tn);
    }
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
    GemmArgs a{A, lda, W, ldw, C, ldc, (int)M, (int)N, (int)K, (int)align_up(K, BK), 0, 0, 0,
               Epilogue{bias, rowscale, relu, nullptr}};
    launch_gemm<0, 0>(vec4_ok(A, lda, K) && vec4_ok(W, ldw, N), a, 1, stream);
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
    // B(k = n_contract, n = k_out) = W[k_out * ldw + n_contract]  -> BMODE 1
    GemmArgs a{dC, lddc, W, ldw, dA, ldda, (int)M, (int)K, (int)N, (int)align_up(N, BK), 0, 0, 0,
               Epilogue{nullptr, rowscale, 0, nullptr}};
    launch_gemm<0, 1>(vec4_ok(dC, lddc, N) && vec4_ok(W, ldw, N), a, 1, stream);
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
    dim3 cg((unsigned)ceil_div(N, 256), (unsigned)nchunks);
    if (vec4_ok(X, ldx, N)) colsum_partial_kernel<true><<<cg, 256, 0, stream>>>(X, ldx, (int)M, (int)N, workspace);
    else                    colsum_partial_kernel<false><<<cg, 256, 0, stream>>>(X, ldx, (int)M, (int)N, workspace);
    slab_reduce_kernel<<<(unsigned)ceil_div(N, 256), 256, 0, stream>>>(workspace, N, nchunks, 1, (int)N, N, out, N);
    return check_launch("npi_colsum");
}

// dW: the contraction runs over the nodes.  Node count is arbitrary, so the part that is a
// multiple of BK goes through `splits` slabs (fast path) and the < BK remainder through one
// extra slab (guarded).
static void bwd_weight_plan(int64_t M, int64_t K, int64_t N, int& splits, int& kchunk, int64_t& m_main) {
    const int64_t tiles = ceil_div(K, BM) * ceil_div(N, BN);
    m_main = (M / BK) * BK;
    splits = pick_splits(m_main > 0 ? m_main : 1, tiles);
    kchunk = (int)(ceil_div(ceil_div(m_main > 0 ? m_main : 1, splits), BK) * BK);
}

extern "C" int64_t npi_linear_bwd_weight_workspace_elems(int64_t M, int64_t K, int64_t N) {
    if (M < 0 || K <= 0 || N <= 0) return -1;
    int splits, kchunk;
    int64_t m_main;
    bwd_weight_plan(M, K, N, splits, kchunk, m_main);
    return (int64_t)(splits + 1) * K * N + (int64_t)(splits + 1) * N + 64;      // dW slabs, then db slabs
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
    int splits, kchunk;
    int64_t m_main;
    bwd_weight_plan(M, K, N, splits, kchunk, m_main);
    const int nslab = splits + 1;
    float* db_slabs = workspace + (int64_t)nslab * K * N;
    const bool v4 = vec4_ok(A, lda, K) && vec4_ok(dC, lddc, N);
    // output rows = K (features of A), cols = N; A(m = feature, k = node) = A[node*lda + feature]
    // main part: nodes [0, m_main) in `splits` slabs
    GemmArgs a{A, lda, dC, lddc, workspace, N, (int)K, (int)N, (int)m_main, kchunk, 0, 0, K * N,
               Epilogue{nullptr, nullptr, 0, db ? db_slabs : nullptr}};
    launch_gemm<1, 0>(v4, a, splits, stream);
    // remainder: nodes [m_main, M) into slab `splits` (a zero slab when there is none)
    GemmArgs r{A + m_main * lda, lda, dC + m_main * lddc, lddc, workspace + (int64_t)splits * K * N, N,
               (int)K, (int)N, (int)(M - m_main), BK, 0, 0, K * N,
               Epilogue{nullptr, nullptr, 0, db ? db_slabs + (int64_t)splits * N : nullptr}};
    launch_gemm<1, 0>(false, r, 1, stream);
    slab_reduce_kernel<<<(unsigned)ceil_div(K * N, 256), 256, 0, stream>>>(workspace, K * N, nslab, (int)K, (int)N, N, dW, lddw);
    if (db) slab_reduce_kernel<<<(unsigned)ceil_div(N, 256), 256, 0, stream>>>(db_slabs, N, nslab, 1, (int)N, N, db, N);
    return check_launch("npi_linear_bwd_weight");
}
