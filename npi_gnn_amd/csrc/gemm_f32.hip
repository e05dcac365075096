// Dense feature projection on the CDNA4 matrix cores: exact f32 (v_mfma_f32_32x32x2_f32) kernels, and -- for
// fwd / bwd_data interior tiles, the default -- an f32-accurate 3-way bf16 split on v_mfma_f32_32x32x16_bf16
// (second half of this file).
//
// Replaces `torch.matmul(aggr_out, self.weight) + self.bias` of PyG 1.4.2 SAGEConv.update /
// GCNConv.forward (reached from reference src/classes.py:62,66,70) and its autograd backward
// (src/train_with_twoDataset.PY:54):  dA = dC W^T,  dW = A^T dC,  db = colsum(dC).
//
// 256-thread workgroup = 2x2 wavefronts, each wavefront a TM x TN grid of 32x32 MFMA tiles
// (block tile 64 TM x 64 TN), BK = 32, double-buffered LDS.
// Two kernels share the tile code:
//   FAST  : full interior tiles with K % 32 == 0 and 16-B aligned rows.  Every staging load is a
//           per-lane 32-bit byte offset computed once plus a scalar base bumped per k-step, pinned
//           at the top of the k-step, so a k-step has no address VALU and no exec-mask branches.
//           (A bare MFMA loop sustains 155 TF on this chip; what separates a tiled kernel from it
//           is the issue time of everything that is not an MFMA, not bandwidth or latency.)
//           Tile 128x256 (TM=2, TN=4; 128 accumulator VGPRs, one workgroup per CU) when the
//           output is >= 256 wide: A is read once and there are half as many LDS reads, stores,
//           barriers and staging instructions per MFMA as with 128x128.
//   EDGE  : 128x128 tile with guarded loads/stores, for the ragged strips at the matrix edges,
//           K tails and unaligned operands (F = 178, 65).
// LDS images are chosen so that fragment reads are bank-conflict free (SQ_LDS_BANK_CONFLICT = 0):
//   operand contiguous along K in memory -> image [row][BK+4], fragment = one ds_read_b128 holding
//       k = 8g + 4h + {0,1,2,3}  (h = lane>>5) -- the k order inside a group of 8 is permuted the
//       same way for A and B, which a sum over k does not care about;
//   operand contiguous along M/N in memory -> image [k][rows+4], fragment = ds_read_b32 per k.
#include "npi_common.h"
#include <stdlib.h>
#include <stdio.h>

namespace npi {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int GEMM_THREADS = 256;
constexpr int KPITCH = BK + 4;      // [row][k] image

struct Epilogue {
    const float* bias;       // [N] or null
    const float* rowscale;   // [M] or null
    int relu;
    float* colsum;           // BMODE 0 only: per-split column sums of B, [splits][N], or null
    // rank-2 update of the output, C += r2_row0 (x) r2_col0 + r2_row1 (x) r2_col1 ([M] and [N] vectors; all four or none):
    // only the split kernel applies it, and only when it covers the whole output (npi_linear_bwd_data_rank2)
    const float* r2_row0; const float* r2_row1; const float* r2_col0; const float* r2_col1;
    // row dots of the STORED output with two column vectors, sc0[m] = <C[m, :], r2_col0>, sc1[m] = <C[m, :], r2_col1> ([M] outputs;
    // both or none; excludes the rank-2 term): only the split kernel when ONE column tile covers N (npi_linear_fwd_scores)
    float* sc0; float* sc1;
};

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
    int tm0, tn0;           // first tile of this launch's grid (in units of the kernel's tile)
    int64_t slab_stride;
    Epilogue ep;
};

// Up to 8 staging registers with compile-time slot access.  (A float4 ARRAY indexed in an unrolled
// loop is demoted to scratch / LDS by hipcc once a sched_barrier sits between its writes and reads.)
struct Slots {
    float4 v0, v1, v2, v3, v4, v5, v6, v7;
    template <int P> __device__ __forceinline__ float4& at() {
        if constexpr (P == 0) return v0; else if constexpr (P == 1) return v1;
        else if constexpr (P == 2) return v2; else if constexpr (P == 3) return v3;
        else if constexpr (P == 4) return v4; else if constexpr (P == 5) return v5;
        else if constexpr (P == 6) return v6; else return v7;
    }
};
template <int I> struct IC { static constexpr int value = I; };
template <int P, int NP, class Fn>
__device__ __forceinline__ void static_for(Fn&& f) {
    if constexpr (P < NP) {
        f(IC<P>{});
        static_for<P + 1, NP>(f);
    }
}

// geometry of one operand's staging: ROWS = tile extent along m (or n)
template <int MODE_KCONTIG, int ROWS>
struct Stage {
    // K-contiguous: image [ROWS][KPITCH]; a thread moves float4 #(t&7) of rows (t>>3) + 32 p
    // row-contiguous: image [BK][ROWS+4]; a thread moves float4 #(t % RQ) of k-rows t / RQ + KR p
    static constexpr int RQ = ROWS / 4;                 // float4 per k-row (row-contiguous)
    static constexpr int KR = GEMM_THREADS / RQ;        // k-rows per pass
    static constexpr int NP = MODE_KCONTIG ? ROWS / 32 : BK / KR;   // float4 per thread
    static constexpr int PITCH = MODE_KCONTIG ? KPITCH : ROWS + 4;
    static constexpr int FLOATS = MODE_KCONTIG ? ROWS * KPITCH : BK * (ROWS + 4);
    __device__ static __forceinline__ int row_of(int t, int p) { return MODE_KCONTIG ? (t >> 3) + 32 * p : (t % RQ) * 4; }
    __device__ static __forceinline__ int k_of(int t, int p) { return MODE_KCONTIG ? (t & 7) * 4 : t / RQ + KR * p; }
    // float offset inside the LDS image
    __device__ static __forceinline__ int lds_off(int t, int p) {
        return MODE_KCONTIG ? ((t >> 3) + 32 * p) * KPITCH + (t & 7) * 4 : (t / RQ + KR * p) * (ROWS + 4) + (t % RQ) * 4;
    }
    // byte offset from the tile's scalar base in global memory
    __device__ static __forceinline__ uint32_t gl_off(int t, int p, int64_t ld) {
        return MODE_KCONTIG ? (uint32_t)((((t >> 3) + 32 * p) * ld + (t & 7) * 4) * 4)
                            : (uint32_t)(((t / RQ + KR * p) * ld + (t % RQ) * 4) * 4);
    }
};

// storage element types of the EDGE kernel: float, or bf16 carried as uint16_t (f32 MFMA accumulation)
typedef uint16_t bf16_t;
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const bf16_t* p) { return __uint_as_float((uint32_t)*p << 16); }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const bf16_t* p) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u),
                       __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
}
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16_t* p, float v) { *p = __builtin_bit_cast(bf16_t, (__bf16)v); }
__device__ __forceinline__ float elem_f32(float v) { return v; }
__device__ __forceinline__ float elem_f32(bf16_t v) { return __uint_as_float((uint32_t)v << 16); }
template <typename T> __device__ __forceinline__ T elem_from_f32(float v);
template <> __device__ __forceinline__ float elem_from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t elem_from_f32<bf16_t>(float v) { return __builtin_bit_cast(bf16_t, (__bf16)v); }

// guarded load of staging slot p (EDGE kernel): rows beyond rmax / k beyond kmax read as zero
template <int MODE_KCONTIG, int ROWS, bool VEC4, typename T>
__device__ __forceinline__ float4 guarded_load(const T* __restrict__ base, int64_t ld, int r0, int rmax,
                                               int k0, int kmax, int t, int p) {
    using S = Stage<MODE_KCONTIG, ROWS>;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const int r = r0 + S::row_of(t, p);
    const int k = k0 + S::k_of(t, p);
    if (MODE_KCONTIG) {
        if (r < rmax) {
            const T* src = base + (int64_t)r * ld + k;
            if (VEC4) { if (k < kmax) v = ld4(src); }
            else {
                if (k + 0 < kmax) v.x = ld1(src + 0);
                if (k + 1 < kmax) v.y = ld1(src + 1);
                if (k + 2 < kmax) v.z = ld1(src + 2);
                if (k + 3 < kmax) v.w = ld1(src + 3);
            }
        }
    } else {
        if (k < kmax) {
            const T* src = base + (int64_t)k * ld + r;
            if (VEC4) { if (r < rmax) v = ld4(src); }
            else {
                if (r + 0 < rmax) v.x = ld1(src + 0);
                if (r + 1 < rmax) v.y = ld1(src + 1);
                if (r + 2 < rmax) v.z = ld1(src + 2);
                if (r + 3 < rmax) v.w = ld1(src + 3);
            }
        }
    }
    return v;
}

// ---- one k-step of MFMAs on the staged tile ----------------------------------------------------------
template <int AMODE, int BMODE, int TM, int TN>
__device__ __forceinline__ void mma_step(const float* __restrict__ as, const float* __restrict__ bs,
                                         int wm, int wn, int li, int lh, f32x16 (&acc)[TM][TN]) {
    constexpr int APITCH = Stage<AMODE == 0, 64 * TM>::PITCH;
    constexpr int BPITCH = Stage<BMODE == 1, 64 * TN>::PITCH;
    float a0[TM][4], a1[TM][4], b0[TN][4], b1[TN][4];
    auto read_frags = [&](int g, float (&fa)[TM][4], float (&fb)[TN][4]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = wm * (32 * TM) + i * 32 + li;
            if (AMODE == 0) {
                float4 v = *reinterpret_cast<const float4*>(as + row * APITCH + g * 8 + lh * 4);
                fa[i][0] = v.x; fa[i][1] = v.y; fa[i][2] = v.z; fa[i][3] = v.w;
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) fa[i][s] = as[(g * 8 + lh * 4 + s) * APITCH + row];
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int c = wn * (32 * TN) + j * 32 + li;
            if (BMODE == 0) {
#pragma unroll
                for (int s = 0; s < 4; ++s) fb[j][s] = bs[(g * 8 + lh * 4 + s) * BPITCH + c];
            } else {
                float4 v = *reinterpret_cast<const float4*>(bs + c * BPITCH + g * 8 + lh * 4);
                fb[j][0] = v.x; fb[j][1] = v.y; fb[j][2] = v.z; fb[j][3] = v.w;
            }
        }
    };
    auto mma_group = [&](const float (&fa)[TM][4], const float (&fb)[TN][4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
    };
    // BK / 8 = 4 k-groups, fragments read one group ahead; the two buffers are named (not indexed by
    // g & 1): a runtime-looking index makes hipcc demote the arrays to LDS / scratch
    static_assert(BK / 8 == 4, "k-group pipeline below is written for BK = 32");
    read_frags(0, a0, b0);
    read_frags(1, a1, b1);
    mma_group(a0, b0);
    read_frags(2, a0, b0);
    mma_group(a1, b1);
    read_frags(3, a1, b1);
    mma_group(a0, b0);
    mma_group(a1, b1);
}

// ---- epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (q & 3) + 8 (q >> 2) + 4 (lane >> 5).
// Every operand is fetched BEFORE the store loop: a load inside it makes hipcc wait vmcnt(0) per
// element, which also drains the preceding store (64 serialised stores per lane).
template <bool GUARD, int TM, int TN, typename T = float>
__device__ __forceinline__ void store_tile(T* __restrict__ C, int64_t ldc, int M, int N, int m0, int n0,
                                           int wm, int wn, int li, int lh, const f32x16 (&acc)[TM][TN],
                                           const Epilogue& ep) {
    float rsv[TM][16];
    float bv[TN];
    const int mw = m0 + wm * (32 * TM), nw = n0 + wn * (32 * TN);
    if (ep.rowscale != nullptr) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int r = mw + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh;
                rsv[i][q] = ep.rowscale[GUARD ? min(r, M - 1) : r];
            }
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) rsv[i][q] = 1.f;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int c = nw + j * 32 + li;
        bv[j] = (ep.bias != nullptr) ? ld1(reinterpret_cast<const T*>(ep.bias) + (GUARD ? min(c, N - 1) : c)) : 0.f;
    }
    const bool relu_on = ep.relu != 0;
    T* __restrict__ cbase = C + (int64_t)(mw + 4 * lh) * ldc + (nw + li);
    // the common case (no row scale, no relu) is one add per element: two separate store loops under a
    // wave-uniform branch (written as one loop hipcc turns the choice into per-element selects)
    auto put = [&](int ro, int j, float v) {
        if (!GUARD) {
            st1(cbase + (int64_t)ro * ldc + j * 32, v);
        } else {
            const int r = mw + 4 * lh + ro, c = nw + j * 32 + li;
            if (r < M && c < N) st1(cbase + (int64_t)ro * ldc + j * 32, v);
        }
    };
    if (!relu_on && ep.rowscale == nullptr) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q)
#pragma unroll
                for (int j = 0; j < TN; ++j) put(i * 32 + (q & 3) + 8 * (q >> 2), j, acc[i][j][q] + bv[j]);
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    float v = fmaf(acc[i][j][q], rsv[i][q], bv[j]);
                    v = (relu_on && v < 0.f) ? 0.f : v;      // keeps NaN, like torch.relu
                    put(i * 32 + (q & 3) + 8 * (q >> 2), j, v);
                }
    }
}

// TI: storage type of A and B (float, or bf16 converted to f32 on its way into LDS -- the dW of the bf16 path)
template <int AMODE, int BMODE, int TM, int TN, typename TI = float>
__global__ void __launch_bounds__(GEMM_THREADS, (TM * TN > 4) ? 1 : 2)
gemm_fast_kernel(GemmArgs a) {
    constexpr int ES = sizeof(TI);
    constexpr int BM = 64 * TM, BN = 64 * TN;
    using SA = Stage<AMODE == 0, BM>;
    using SB = Stage<BMODE == 1, BN>;
    constexpr int AF = SA::FLOATS, BF = SB::FLOATS;
    __shared__ __attribute__((aligned(16))) float lds[2][AF + BF];
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

    // scalar bases, bumped per k-step; per-lane offsets are loop-invariant expressions of t
    const char* sa = reinterpret_cast<const char*>(a.A) +
        ((AMODE == 0) ? (int64_t)m0 * a.lda + kbeg : (int64_t)kbeg * a.lda + m0) * ES;
    const char* sb = reinterpret_cast<const char*>(a.B) +
        ((BMODE == 0) ? (int64_t)kbeg * a.ldb + n0 : (int64_t)n0 * a.ldb + kbeg) * ES;
    const int64_t step_a = ((AMODE == 0) ? (int64_t)BK : (int64_t)BK * a.lda) * ES;
    const int64_t step_b = ((BMODE == 0) ? (int64_t)BK * a.ldb : (int64_t)BK) * ES;

    // Staging registers: fully unrolled constant indices keep them in VGPRs.  readfirstlane keeps
    // the bases in SGPRs (loop strength reduction would otherwise give every load address its own
    // 64-bit VGPR induction variable).
    // Staging registers are plain named locals moved by macros: arrays, structs or lambdas that
    // capture them by reference end up in scratch / LDS once a sched_barrier sits between the
    // loads and the LDS stores (seen in the ISA, round 1).
    static_assert(SA::NP == 4 && (SB::NP == 4 || SB::NP == 8), "staging macros below assume these counts");
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3, rb4, rb5, rb6, rb7;
#define NPI_LD(P, base, S, ld) ld4(reinterpret_cast<const TI*>((base) + S::gl_off(t, P, ld) / (4 / ES)))
#define NPI_GLOAD()                                                                  \
    do {                                                                             \
        const char* ua = uniform_ptr(sa);                                            \
        const char* ub = uniform_ptr(sb);                                            \
        ra0 = NPI_LD(0, ua, SA, a.lda); ra1 = NPI_LD(1, ua, SA, a.lda);              \
        ra2 = NPI_LD(2, ua, SA, a.lda); ra3 = NPI_LD(3, ua, SA, a.lda);              \
        rb0 = NPI_LD(0, ub, SB, a.ldb); rb1 = NPI_LD(1, ub, SB, a.ldb);              \
        rb2 = NPI_LD(2, ub, SB, a.ldb); rb3 = NPI_LD(3, ub, SB, a.ldb);              \
        if constexpr (SB::NP == 8) {                                                 \
            rb4 = NPI_LD(4, ub, SB, a.ldb); rb5 = NPI_LD(5, ub, SB, a.ldb);          \
            rb6 = NPI_LD(6, ub, SB, a.ldb); rb7 = NPI_LD(7, ub, SB, a.ldb);          \
        }                                                                            \
        sa += step_a;                                                                \
        sb += step_b;                                                                \
    } while (0)
#define NPI_ST(P, buf, off0, S, v) (*reinterpret_cast<float4*>(lds[buf] + (off0) + S::lds_off(t, P)) = (v))
#define NPI_LSTORE(buf)                                                              \
    do {                                                                             \
        NPI_ST(0, buf, 0, SA, ra0); NPI_ST(1, buf, 0, SA, ra1);                      \
        NPI_ST(2, buf, 0, SA, ra2); NPI_ST(3, buf, 0, SA, ra3);                      \
        NPI_ST(0, buf, AF, SB, rb0); NPI_ST(1, buf, AF, SB, rb1);                    \
        NPI_ST(2, buf, AF, SB, rb2); NPI_ST(3, buf, AF, SB, rb3);                    \
        if constexpr (SB::NP == 8) {                                                 \
            NPI_ST(4, buf, AF, SB, rb4); NPI_ST(5, buf, AF, SB, rb5);                \
            NPI_ST(6, buf, AF, SB, rb6); NPI_ST(7, buf, AF, SB, rb7);                \
        }                                                                            \
    } while (0)

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    const bool do_colsum = (BMODE == 0) && (a.ep.colsum != nullptr) && (m0 == 0);
    float csum = 0.f;
    constexpr int BPITCH = SB::PITCH;

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
        mma_step<AMODE, BMODE, TM, TN>(lds[buf], lds[buf] + AF, wm, wn, li, lh, acc);
        if (do_colsum && t < BN) {
#pragma unroll 8
            for (int k = 0; k < BK; ++k) csum += lds[buf][AF + k * BPITCH + t];
        }
        __builtin_amdgcn_sched_barrier(0);
        NPI_LSTORE(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
#undef NPI_GLOAD
#undef NPI_LSTORE
#undef NPI_LD
#undef NPI_ST
    if (nk > 0) {                                    // last k-step: nothing left to stage
        mma_step<AMODE, BMODE, TM, TN>(lds[buf], lds[buf] + AF, wm, wn, li, lh, acc);
        if (do_colsum && t < BN) {
#pragma unroll 8
            for (int k = 0; k < BK; ++k) csum += lds[buf][AF + k * BPITCH + t];
        }
    }
    if (do_colsum && t < BN) a.ep.colsum[(int64_t)z * a.N + n0 + t] = csum;
    store_tile<false, TM, TN>(a.C + (int64_t)z * a.slab_stride, a.ldc, a.M, a.N, m0, n0, wm, wn, li, lh, acc, a.ep);
}

// TI: element type of A and B; TO: element type of C and bias (f32 slabs for the split-K dW)
template <int AMODE, int BMODE, bool VEC4, typename TI = float, typename TO = float>
__global__ void __launch_bounds__(GEMM_THREADS, 2)
gemm_edge_kernel(GemmArgs a) {
    constexpr int TM = 2, TN = 2, BM = 128, BN = 128;
    using SA = Stage<AMODE == 0, BM>;
    using SB = Stage<BMODE == 1, BN>;
    constexpr int AF = SA::FLOATS, BF = SB::FLOATS;
    __shared__ __attribute__((aligned(16))) float lds[2][AF + BF];
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

    float4 ra[SA::NP], rb[SB::NP];
    auto gload = [&](int k0) {
#pragma unroll
        for (int p = 0; p < SA::NP; ++p) ra[p] = guarded_load<AMODE == 0, BM, VEC4, TI>(reinterpret_cast<const TI*>(a.A), a.lda, m0, a.M, k0, kend, t, p);
#pragma unroll
        for (int p = 0; p < SB::NP; ++p) rb[p] = guarded_load<BMODE == 1, BN, VEC4, TI>(reinterpret_cast<const TI*>(a.B), a.ldb, n0, a.N, k0, kend, t, p);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int p = 0; p < SA::NP; ++p) *reinterpret_cast<float4*>(lds[buf] + SA::lds_off(t, p)) = ra[p];
#pragma unroll
        for (int p = 0; p < SB::NP; ++p) *reinterpret_cast<float4*>(lds[buf] + AF + SB::lds_off(t, p)) = rb[p];
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
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
        mma_step<AMODE, BMODE, TM, TN>(lds[buf], lds[buf] + AF, wm, wn, li, lh, acc);
        if (do_colsum && t < BN) {
#pragma unroll 8
            for (int k = 0; k < BK; ++k) csum += lds[buf][AF + k * SB::PITCH + t];
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }
    if (do_colsum && t < BN && n0 + t < a.N) a.ep.colsum[(int64_t)z * a.N + n0 + t] = csum;
    store_tile<true, TM, TN, TO>(reinterpret_cast<TO*>(a.C) + (int64_t)z * a.slab_stride, a.ldc, a.M, a.N, m0, n0, wm, wn, li, lh, acc, a.ep);
}

// out[r, c] = sum_z slabs[z][r, c]  (z ascending: deterministic)
template <typename TO>
__global__ void slab_reduce_kernel(const float* __restrict__ slabs, int64_t slab_stride, int nslab,
                                   int rows, int cols, int64_t ld_slab, TO* __restrict__ out, int64_t ldo) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows * cols) return;
    int r = (int)(i / cols), c = (int)(i % cols);
    const float* __restrict__ p = slabs + (int64_t)r * ld_slab + c;
    float s = 0.f;
    int z = 0;
    for (; z + 8 <= nslab; z += 8) {                      // 8 loads in flight; added in slab order all the same
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(z + u) * slab_stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; z < nslab; ++z) s += p[(int64_t)z * slab_stride];
    st1(out + (int64_t)r * ldo + c, s);
}

// The f32 dW epilogue in ONE launch: dW[r, c] = sum_z slabs[z][r, c] (z ascending) + the < 32 trailing nodes the slab
// kernels do not cover, sum_m A[m, r] dC[m, c] (plain f32 FMAs, node order), and db[c] = sum_z db_slabs[z][c] + sum_m dC[m, c].
// (Separate launches for the remainder GEMM and the db reduction were 2 of the 4 launches of a small layer's dW.)
// Blocks [0, ceil(rows cols / 256)) take dW, one element per thread; the blocks behind them db: 16 columns x 16 partial sums
// over every 16th slab row, folded in a fixed order (up to 512 db rows: one thread per column would walk them serially).
constexpr int DWF_COLS = 16;
static unsigned dw_finish_grid(int64_t rows, int64_t cols, bool with_db) {
    return (unsigned)(ceil_div(rows * cols, 256) + (with_db ? ceil_div(cols, DWF_COLS) : 0));
}
// TS: storage of A / dC (the trailing nodes) and of dW / db: float, or bf16 (widened on load, rounded once on store)
template <typename TS>
__global__ void __launch_bounds__(256)
dw_finish_kernel(const float* __restrict__ slabs, int64_t slab_stride, int nslab, int rows, int cols,
                 TS* __restrict__ dW, int64_t lddw, const float* __restrict__ db_slabs, int n_db,
                 TS* __restrict__ db, const TS* __restrict__ A_rem, int64_t lda,
                 const TS* __restrict__ dC_rem, int64_t lddc, int n_rem) {
    const int64_t n_w = (int64_t)rows * cols;
    const int nb_w = (int)((n_w + 255) / 256);
    const int t = threadIdx.x;
    if ((int)blockIdx.x < nb_w) {
        const int64_t i = (int64_t)blockIdx.x * 256 + t;
        if (i >= n_w) return;
        const int r = (int)(i / cols), c = (int)(i % cols);
        const float* __restrict__ p = slabs + (int64_t)r * cols + c;
        float s = 0.f;
        int z = 0;
        for (; z + 8 <= nslab; z += 8) {                      // 8 loads in flight; added in slab order all the same
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(z + u) * slab_stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; z < nslab; ++z) s += p[(int64_t)z * slab_stride];
        for (int m0 = 0; m0 < n_rem; m0 += 8) {               // likewise the trailing nodes
            // raw loads first (unconditional, the row index clamped), widening behind the last of them: a bf16 element converted
            // inside its own guard made hipcc wait for every load separately -- 58 serialised round trips, 13 us against 5 us
            TS av[8], cv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int m = min(m0 + u, n_rem - 1);
                av[u] = A_rem[(int64_t)m * lda + r];
                cv[u] = dC_rem[(int64_t)m * lddc + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s = (m0 + u < n_rem) ? fmaf(elem_f32(av[u]), elem_f32(cv[u]), s) : s;
        }
        dW[(int64_t)r * lddw + c] = elem_from_f32<TS>(s);
        return;
    }
    if (db == nullptr) return;
    __shared__ float part[256 / DWF_COLS][DWF_COLS];
    const int cl = t % DWF_COLS, zl = t / DWF_COLS;
    const int c = ((int)blockIdx.x - nb_w) * DWF_COLS + cl;
    float s = 0.f;
    if (c < cols) {
        constexpr int ZL = 256 / DWF_COLS;
        int z = zl;
        for (; z + 3 * ZL < n_db; z += 4 * ZL) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = db_slabs[(int64_t)(z + u * ZL) * cols + c];
#pragma unroll
            for (int u = 0; u < 4; ++u) s += v[u];
        }
        for (; z < n_db; z += ZL) s += db_slabs[(int64_t)z * cols + c];
    }
    part[zl][cl] = s;
    __syncthreads();
    if (zl == 0 && c < cols) {
        float r = 0.f;
#pragma unroll
        for (int q = 0; q < 256 / DWF_COLS; ++q) r += part[q][cl];
        for (int m0 = 0; m0 < n_rem; m0 += 8) {
            TS cv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) cv[u] = dC_rem[(int64_t)min(m0 + u, n_rem - 1) * lddc + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) r += m0 + u < n_rem ? elem_f32(cv[u]) : 0.f;
        }
        db[c] = elem_from_f32<TS>(r);
    }
}

// partial column sums of X[M, N] over row chunks: part[chunk][c].  Lanes walk a row 16 B each
// (a 256-column row is one 1 KiB wave instruction); the 4 waves of a workgroup take rows r, r+1, ...
constexpr int COLSUM_ROWS = 2048;      // most rows per workgroup (and the chunking the documented minimum workspace implies)
// rows per workgroup: about 1,024 workgroups per 256 columns, 16..COLSUM_ROWS rows each (a 5,085-row matrix is 3 workgroups
// of 2,048 rows otherwise: 30 us of serial loads for 5 MB)
static int colsum_rows(int64_t M) {
    const int64_t r = ceil_div(M > 0 ? M : 1, 1024);
    return (int)(r < 16 ? 16 : r > COLSUM_ROWS ? COLSUM_ROWS : r);
}
template <bool VEC4>
__global__ void __launch_bounds__(256)
colsum_partial_kernel(const float* __restrict__ X, int64_t ldx, int M, int N, int rows, float* __restrict__ part) {
    __shared__ float red[4][256];
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + lane * 4;
    const int rbeg = blockIdx.y * rows;
    const int rend = min(M, rbeg + rows);
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < N) {
#pragma unroll 8
        for (int r = rbeg + wave; r < rend; r += 4) {                      // independent loads: keep 8 rows in flight
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

// Power-of-two COLUMN scales for the fp16 x 2 weight-gradient GEMM (the contraction runs over the rows, so only a column scale
// factors out of the sum).  Two sources: the column maxima of a matrix (one pass over it: static input features, once), or --
// no pass at all -- the smallest of the ROW scales its producer wrote, i.e. the scale of the whole matrix's largest magnitude, for
// every column alike (what an aggregation or the fused GATConv pass leaves behind; elements more than 2^-18 below that magnitude
// then keep an absolute 2^-39 of it instead of 22 relative bits).
__global__ void __launch_bounds__(256)
colmax_partial_kernel(const float* __restrict__ X, int64_t ldx, int M, int N, int rows, float* __restrict__ part) {
    __shared__ float red[4][256];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + lane * 4;
    const int rbeg = blockIdx.y * rows, rend = min(M, rbeg + rows);
    float m[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < N) {
#pragma unroll 8
        for (int r = rbeg + wave; r < rend; r += 4) {
            const float* src = X + (int64_t)r * ldx + c;
#pragma unroll
            for (int q = 0; q < 4; ++q) if (c + q < N) m[q] = fmaxf(m[q], fabsf(src[q]));
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) red[wave][lane * 4 + q] = m[q];
    __syncthreads();
    const int cc = blockIdx.x * 256 + threadIdx.x;
    if (cc < N) part[(int64_t)blockIdx.y * N + cc] = fmaxf(fmaxf(red[0][threadIdx.x], red[1][threadIdx.x]), fmaxf(red[2][threadIdx.x], red[3][threadIdx.x]));
}
__global__ void colmax_finish_kernel(const float* __restrict__ part, int nchunks, int N, float* __restrict__ scales) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    float m = 0.f;
    for (int z = 0; z < nchunks; ++z) m = fmaxf(m, part[(int64_t)z * N + c]);
    scales[c] = pow2_scale_of(m);
}
__global__ void __launch_bounds__(256)
minscale_partial_kernel(const float* __restrict__ rs, int64_t M, float* __restrict__ part) {
    __shared__ float red[4];
    float m = 3.0e38f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < M; i += (int64_t)gridDim.x * 256) m = fminf(m, rs[i]);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = fminf(m, __shfl_xor(m, d, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
}
__global__ void __launch_bounds__(256)
minscale_finish_kernel(const float* __restrict__ part, int nparts, int K, float* __restrict__ scales) {
    __shared__ float red[4];
    float m = 3.0e38f;
    for (int i = threadIdx.x; i < nparts; i += 256) m = fminf(m, part[i]);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = fminf(m, __shfl_xor(m, d, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
    if (!(m < 3.0e38f)) m = 1.f;                          // no row at all
    for (int k = threadIdx.x; k < K; k += 256) scales[k] = m;
}

// ---- f32 GEMM on the bf16 matrix cores: 3-way split ------------------------------------------------
// x = x0 + x1 + x2 with x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1): three bf16 carry the 24
// significand bits of an f32 (the remainder is <= 2^-25 |x|; exponent range is the f32 one).  The
// product a*b is then the six bf16 x bf16 terms of order <= 2,
//     a0 b0 + (a0 b1 + a1 b0) + (a1 b1 + a0 b2 + a2 b0),      dropped terms <= 3 * 2^-24 |a b|,
// each exact in the f32 accumulator of v_mfma_f32_32x32x16_bf16.  Six bf16 MFMAs cost 6/16 of the
// f32 MFMA time for the same tile, and the result carries f32-rounding-level error (measured in
// tools/gemm_accuracy.py against an fp64 product, next to the exact-f32 kernel).
// Limits: an Inf in A or B gives NaN (Inf - Inf in the split), not Inf.
//
// Kernel (fwd, bwd_data): C[M, N] = A[M, K] B(K, N).  A is f32, K-contiguous, split in registers on its
// way into LDS; B (the small weight matrix) comes as three pre-split bf16 planes (split_planes_kernel).
// Persistent and wave-specialised: a workgroup = 4 consumer waves (fragment reads + MFMAs + epilogue,
// a 2 x 2 grid of 64 x 32 TN wave tiles) + 4 producer waves (global loads, the split, LDS stores), one of
// each per SIMD, so the matrix pipe and the VALU / LDS-store work overlap by construction rather than
// by instruction scheduling.  Tile 128 x 64 TN (TN = 4 when N % 256 == 0: every A element is then split
// once for 256 output columns), BK = 16, FOUR LDS stages handed over through per-stage LDS counters (no barrier in
// the main loop: the producer runs up to three k-steps ahead, so a consumer's epilogue does not stop it); the
// producer keeps a ring of four k-steps of loads in flight and runs across tile boundaries, so only the
// first tile of a workgroup exposes a load latency.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef float f32x4r __attribute__((ext_vector_type(4)));      // native 128-bit register values (inline-asm operands)
typedef uint32_t u32x4r __attribute__((ext_vector_type(4)));

constexpr int SK = 16;                             // k-step of the split kernel = one 32x32x16 MFMA block
// LDS plane image: [row][16 bf16] = 32-byte rows, no padding; the two 16-byte halves of a row are swapped
// on rows with bit 3 set, which makes the ds_read_b128 fragment reads (lane -> row, half lane>>5)
// bank-conflict free for the 16-lane groups the hardware forms
__device__ __forceinline__ int simg(int row, int half) { return row * 32 + ((half ^ ((row >> 3) & 1)) << 4); }

__device__ __forceinline__ uint32_t pack_bf16(float x, float y) {
    f32x2v v = {x, y};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2v));    // x in the low half
}
__device__ __forceinline__ void split3_pair(float x, float y, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
    // two-element vectors: the subtractions become one v_pk_add_f32 per stage
    f32x2v v = {x, y};
    p0 = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2v));
    f32x2v h0 = {__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xffff0000u)};
    v = v - h0;
    p1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2v));
    f32x2v h1 = {__uint_as_float(p1 << 16), __uint_as_float(p1 & 0xffff0000u)};
    v = v - h1;
    p2 = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2v));
}
// the three planes of four consecutive k, as 8-byte LDS stores; `plane` = bytes between plane images
__device__ __forceinline__ void split3_store(f32x4r v, char* img, int plane) {
    uint32_t a0, a1, a2, b0, b1, b2;
    split3_pair(v.x, v.y, a0, a1, a2);
    split3_pair(v.z, v.w, b0, b1, b2);
    *reinterpret_cast<uint2*>(img) = make_uint2(a0, b0);
    *reinterpret_cast<uint2*>(img + plane) = make_uint2(a1, b1);
    *reinterpret_cast<uint2*>(img + 2 * plane) = make_uint2(a2, b2);
}

// ---- fp16 x 2 (NPI_GEMM_SPLIT_F16X2): x s = h0 + h1 with h0 = fp16(x s), h1 = fp16(x s - h0), s a power of two that puts the
// largest |x| of the ROW (A) / COLUMN (B) into [2^14, 2^15): two fp16 carry 22 significant bits of every element that is within
// 2^-28 of that maximum and an ABSOLUTE error <= 2^-39 of it below (the matrix pipe honours fp16 subnormals on gfx950:
// tools/micro/mfma_f16_split.hip), and a b = a0 b0 + a0 b1 + a1 b0 + O(2^-22 |a b|): THREE v_mfma_f32_32x32x16_f16 instead of six
// bf16 ones for the same f32-rounding-level result (measured there: 3.8e-7 of the row's largest |C| against 3.9e-7 for the six
// bf16 products and 5.5e-7 for an f32 FMA loop).  The epilogue undoes the scales (exact: powers of two).
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2_pair(float x, float y, uint32_t& p0, uint32_t& p1) {
    f32x2v v = {x, y};
    const f16x2v h0 = __builtin_convertvector(v, f16x2v);
    p0 = __builtin_bit_cast(uint32_t, h0);
    v = v - __builtin_convertvector(h0, f32x2v);
    p1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2v));
}
__device__ __forceinline__ void split2_store(f32x4r v, float sc, char* img, int plane) {
    uint32_t a0, a1, b0, b1;
    split2_pair(v.x * sc, v.y * sc, a0, a1);
    split2_pair(v.z * sc, v.w * sc, b0, b1);
    *reinterpret_cast<uint2*>(img) = make_uint2(a0, b0);
    *reinterpret_cast<uint2*>(img + plane) = make_uint2(a1, b1);
}
// (pow2_scale_of / pow2_inverse: npi_common.h -- the aggregation kernels write the same scales for the rows they finish)

// scale[r] for every row of A [M, K] (npi_row_scales): one wave per row
__global__ void __launch_bounds__(256)
row_scale_kernel(const float* __restrict__ A, int64_t lda, int M, int K, float* __restrict__ scale) {
    const int r = (int)blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= M) return;
    const float* __restrict__ row = A + (int64_t)r * lda;
    float m = 0.f;                       // (fmaxf drops a NaN: the row is scaled by its finite values and the NaN stays a NaN)
    if (((uintptr_t)row % 16) == 0 && K % 4 == 0) {
        for (int k = lane * 4; k < K; k += 256) {
            const float4 v = *reinterpret_cast<const float4*>(row + k);
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
    } else {
        for (int k = lane; k < K; k += 64) m = fmaxf(m, fabsf(row[k]));
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
    if (lane == 0) scale[r] = pow2_scale_of(m);
}

// planes[p][k / 16][n][k % 16] (bf16, K % 16 == 0) of B(k, n): BMODE 0: B[k*ldb + n], BMODE 1: B[n*ldb + k].
// k-step major: the BN x 16 tile of one k-step is BN * 32 contiguous bytes per plane, so the producer's
// loads are whole cache lines.
__global__ void split_planes_kernel(const float* __restrict__ B, int64_t ldb, int K, int N, int bmode,
                                    uint16_t* __restrict__ planes, int k_valid) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * K) return;
    const int n = (int)(i / K), k = (int)(i % K);
    // k >= k_valid: the zero rows that go with the zero pad columns of A (NPI_GEMM_A_ZERO_PADDED)
    const float x = k >= k_valid ? 0.f : (bmode == 0 ? B[(int64_t)k * ldb + n] : B[(int64_t)n * ldb + k]);
    uint32_t p0, p1, p2;
    split3_pair(x, 0.f, p0, p1, p2);
    const int64_t o = ((int64_t)(k / SK) * N + n) * SK + (k % SK);
    planes[o] = (uint16_t)p0;
    planes[(int64_t)N * K + o] = (uint16_t)p1;
    planes[2 * (int64_t)N * K + o] = (uint16_t)p2;
}

// fp16 x 2: planes[p][k / 16][n][k % 16] (two fp16 planes) of B(k, n) s_n with s_n the power-of-two scale of COLUMN n, and
// inv[n] = 1 / s_n.  One workgroup per column.
__global__ void __launch_bounds__(256)
split_planes_f16_kernel(const float* __restrict__ B, int64_t ldb, int K, int N, int bmode, uint16_t* __restrict__ planes,
                        float* __restrict__ inv) {
    const int n = (int)blockIdx.x;
    if (n >= N) return;
    __shared__ float red[4];
    auto elem = [&](int k) { return bmode == 0 ? B[(int64_t)k * ldb + n] : B[(int64_t)n * ldb + k]; };
    float m = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) m = fmaxf(m, fabsf(elem(k)));
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    const float sc = pow2_scale_of(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
    for (int k = threadIdx.x; k < K; k += 256) {
        uint32_t p0, p1;
        split2_pair(elem(k) * sc, 0.f, p0, p1);
        const int64_t o = ((int64_t)(k / SK) * N + n) * SK + (k % SK);
        planes[o] = (uint16_t)p0;
        planes[(int64_t)N * K + o] = (uint16_t)p1;
    }
    if (threadIdx.x == 0) inv[n] = pow2_inverse(sc);
}
// where the column scales' inverses live in a weight workspace: behind the two planes (npi_linear_workspace_bytes covers three)
static inline float* f16_inv_of(void* planes, int64_t K, int64_t N) { return reinterpret_cast<float*>(reinterpret_cast<char*>(planes) + 4 * K * N); }

// Epilogue of the split kernel.  Its MFMAs are issued with the operands swapped (B fragment first), so an
// accumulator tile is C^T: the lane owns ONE row of C (m = lane & 31) and its registers run along the
// columns, n = 8 (q >> 2) + 4 (lane >> 5) + (q & 3): four consecutive columns per register quad, i.e.
// 16-byte stores.  (With the natural orientation a lane owns a column and every store is 4 bytes; a wave
// may have 64 stores in flight, so the 1 GB of C then drains at 64 x 256 B per write round trip per wave --
// measured 0.45 ms of a 0.9 ms kernel.)
// EPI: 0 plain, 1 (R2) the rank-2 term, 2 (SC) this lane's share of the row dots of the stored values with two column
// vectors (same LDS layout as the bias), returned in (*g0)[i] / (*g1)[i]
// CS (fp16 x 2): the accumulators carry the operands' power-of-two scales -- cs_lds holds 1 / (column scale) like the bias, the
// row's 1 / (row scale) is folded into rs by the caller; both multiplications are exact
template <int TM, int TN, int EPI = 0, bool CS = false>
__device__ __forceinline__ void store_tile_t(float* __restrict__ C, int64_t ldc, int mw, int nw, int li, int lh,
                                             const f32x16 (&acc)[TM][TN], const float* __restrict__ bias_lds /* this wave's
                                             first column (zeros without a bias) */, const float (&rs)[TM], float floor_,
                                             const float* __restrict__ u0_lds = nullptr, const float* __restrict__ u1_lds = nullptr,
                                             float (*g0)[TM] = nullptr, float (*g1)[TM] = nullptr,
                                             const float* __restrict__ cs_lds = nullptr) {
    constexpr bool R2 = EPI == 1, SC = EPI == 2;
    if constexpr (SC) {
#pragma unroll
        for (int i = 0; i < TM; ++i) { (*g0)[i] = 0.f; (*g1)[i] = 0.f; }
    }
    // bias comes from an LDS copy made once per workgroup and the row scales were fetched at the start of the
    // tile: a global load here would put a full memory round trip in front of every tile's stores.  No branch in
    // here (a missing bias is a row of zeros, no ReLU is a floor of -inf): with one, every 16-byte store waited for
    // its own LDS read of the bias -- 32 round trips per tile.
    float* __restrict__ crow[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) crow[i] = C + (int64_t)(mw + i * 32 + li) * ldc + nw + 4 * lh;
    // one bias read per 16-byte column group, shared by the row blocks and issued one group ahead of its use
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (!SC) b = *reinterpret_cast<const float4*>(bias_lds + 4 * lh);
#pragma unroll
    for (int jg = 0; jg < 4 * TN; ++jg) {
        const int j = jg >> 2, g = jg & 3;
        const int jn = (jg + 1) >> 2, gn = (jg + 1) & 3;
        float4 bn = b;
        if constexpr (!SC)
            if (jg + 1 < 4 * TN) bn = *reinterpret_cast<const float4*>(bias_lds + jn * 32 + 8 * gn + 4 * lh);
        float4 u0 = make_float4(0.f, 0.f, 0.f, 0.f), u1 = u0;
        if constexpr (R2 || SC) {         // the column vectors of the rank-2 term / the row dots, same LDS layout as the bias
            u0 = *reinterpret_cast<const float4*>(u0_lds + j * 32 + 8 * g + 4 * lh);
            u1 = *reinterpret_cast<const float4*>(u1_lds + j * 32 + 8 * g + 4 * lh);
        }
        float4 cs = make_float4(1.f, 1.f, 1.f, 1.f);
        if constexpr (CS) cs = *reinterpret_cast<const float4*>(cs_lds + j * 32 + 8 * g + 4 * lh);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float4 v;
            float4 q = make_float4(acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
            if constexpr (CS) { q.x *= cs.x; q.y *= cs.y; q.z *= cs.z; q.w *= cs.w; }
            if constexpr (SC) {           // plain C = A B (no bias, row scale or ReLU: npi_linear_fwd_scores): the registers go to the dots
                v = q;
                if constexpr (CS) { v.x *= rs[i]; v.y *= rs[i]; v.z *= rs[i]; v.w *= rs[i]; }
                (*g0)[i] = fmaf(v.w, u0.w, fmaf(v.z, u0.z, fmaf(v.y, u0.y, fmaf(v.x, u0.x, (*g0)[i]))));
                (*g1)[i] = fmaf(v.w, u1.w, fmaf(v.z, u1.z, fmaf(v.y, u1.y, fmaf(v.x, u1.x, (*g1)[i]))));
                *reinterpret_cast<float4*>(crow[i] + j * 32 + 8 * g) = v;
                continue;
            }
            v.x = fmaf(q.x, rs[i], b.x);
            v.y = fmaf(q.y, rs[i], b.y);
            v.z = fmaf(q.z, rs[i], b.z);
            v.w = fmaf(q.w, rs[i], b.w);
            if constexpr (R2) {
                const float a0 = (*g0)[i], a1 = (*g1)[i];
                v.x = fmaf(a1, u1.x, fmaf(a0, u0.x, v.x)); v.y = fmaf(a1, u1.y, fmaf(a0, u0.y, v.y));
                v.z = fmaf(a1, u1.z, fmaf(a0, u0.z, v.z)); v.w = fmaf(a1, u1.w, fmaf(a0, u0.w, v.w));
            }
            // keeps NaN, like torch.relu
            v.x = v.x < floor_ ? floor_ : v.x; v.y = v.y < floor_ ? floor_ : v.y;
            v.z = v.z < floor_ ? floor_ : v.z; v.w = v.w < floor_ ? floor_ : v.w;
            *reinterpret_cast<float4*>(crow[i] + j * 32 + 8 * g) = v;        // (non-temporal stores: 13 % slower)
        }
        b = bn;
    }
}

// The same epilogue with FULL 128-byte lines per store instruction (round 6; EPI 0).  In the accumulator layout a wave's store
// instruction writes 32 rows x 32 bytes (the lane pair li / li + 32 of a row): a quarter of 32 different lines -- tools/micro/
// store_pattern.hip: 18 bytes per cycle and CU, against 51 when an instruction writes 8 rows x 128 bytes and 58 for whole rows.
// So the four rows li = 4 a .. 4 a + 3 of a quad of lanes exchange their four 16-byte column groups first -- a 4 x 4 transpose of
// 16-byte elements inside every quad, two butterfly stages of quad_perm DPP moves and selects (4 VALU instructions per register,
// in a phase whose VALU is otherwise idle) -- after which lane r of the quad holds column group r of ALL four rows and instruction
// g' stores row 4 a + g': the 8 lanes (r, lh) of a row write its 128 contiguous bytes.  The lane's columns no longer depend on the
// register, so bias / column scale are read once per 32-column block; the row scales of the quad's four rows are broadcast inside
// the quad.  The same operations on the same values in the same order as store_tile_t: bit-identical.
__device__ __forceinline__ float qp_xor2(float x) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x4E, 0xf, 0xf, true)); }   // quad_perm [2,3,0,1]
__device__ __forceinline__ float qp_xor1(float x) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xB1, 0xf, 0xf, true)); }   // quad_perm [1,0,3,2]
// EPI 1 (R2, fp16 x 2 only): the rank-2 term's row factors take the same quad broadcast as the row scales, its column vectors are read
// like the bias (the launch has none: its four registers go to them); 0.61 -> 0.57 ms at 1M x 256 x 256.  The row-dot epilogue (EPI 2)
// and bf16 x 3 with either epilogue keep store_tile_t: the exchange's 16 temporaries on top of their own state do not fit in 256
// registers (92 - 632 bytes of scratch in every arrangement tried: dots before / after the exchange, reduce-scattered per block).
__device__ __forceinline__ float quad_lane(float x, int g) {               // g is a constant after unrolling
    const int v = __float_as_int(x);
    return __int_as_float(g == 0 ? __builtin_amdgcn_mov_dpp(v, 0x00, 0xf, 0xf, true) : g == 1 ? __builtin_amdgcn_mov_dpp(v, 0x55, 0xf, 0xf, true)
                        : g == 2 ? __builtin_amdgcn_mov_dpp(v, 0xAA, 0xf, 0xf, true) : __builtin_amdgcn_mov_dpp(v, 0xFF, 0xf, 0xf, true));
}
template <int TM, int TN, bool CS, int EPI = 0>
__device__ __forceinline__ void store_tile_q(float* __restrict__ C, int64_t ldc, int mw, int nw, int li, int lh,
                                             const f32x16 (&acc)[TM][TN], const float* __restrict__ bias_lds, const float (&rs)[TM],
                                             float floor_, const float* __restrict__ cs_lds,
                                             const float* __restrict__ u0_lds = nullptr, const float* __restrict__ u1_lds = nullptr,
                                             float (*g0)[TM] = nullptr, float (*g1)[TM] = nullptr) {
    static_assert(EPI == 0 || EPI == 1, "the row-dot epilogue stays on store_tile_t");
    constexpr bool R2 = EPI == 1;
    const bool b1 = (li & 2) != 0, b0 = (li & 1) != 0;
    const int r = li & 3;
    const int co = 8 * r + 4 * lh;                    // this lane's four columns inside every 32-column block, after the exchange
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        float* __restrict__ rowbase = C + (int64_t)(mw + i * 32 + (li & ~3)) * ldc + nw + co;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (EPI == 0) b = *reinterpret_cast<const float4*>(bias_lds + j * 32 + co);     // (R2: launched without a bias)
            float4 cs = make_float4(1.f, 1.f, 1.f, 1.f);
            if constexpr (CS) cs = *reinterpret_cast<const float4*>(cs_lds + j * 32 + co);
            float4 u0 = make_float4(0.f, 0.f, 0.f, 0.f), u1 = u0;
            if constexpr (R2) {
                u0 = *reinterpret_cast<const float4*>(u0_lds + j * 32 + co);
                u1 = *reinterpret_cast<const float4*>(u1_lds + j * 32 + co);
            }
            float e[4][4];                            // [column group g -> row g' after the exchange][dword]
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int c = 0; c < 4; ++c) e[g][c] = acc[i][j][4 * g + c];
            // (a scheduling fence behind every exchange: left to itself hipcc hoists all 32 DPP moves of a block -- and of the next
            // blocks -- to the front and needs 60 registers more than the kernel has: 224 bytes of scratch in the first build)
#pragma unroll
            for (int g = 0; g < 2; ++g)               // lanes r <-> r ^ 2, groups g <-> g ^ 2
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float lo = e[g][c], hi = e[g + 2][c];
                    const float tlo = qp_xor2(lo), thi = qp_xor2(hi);
                    e[g][c] = b1 ? thi : lo;
                    e[g + 2][c] = b1 ? hi : tlo;
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
            for (int g = 0; g < 4; g += 2)            // lanes r <-> r ^ 1, groups g <-> g ^ 1
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float lo = e[g][c], hi = e[g + 1][c];
                    const float tlo = qp_xor1(lo), thi = qp_xor1(hi);
                    e[g][c] = b0 ? thi : lo;
                    e[g + 1][c] = b0 ? hi : tlo;
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
            for (int g = 0; g < 4; ++g) {             // row 4 a + g: its columns 32 j + 8 r + 4 lh .. + 3
                float4 q = make_float4(e[g][0], e[g][1], e[g][2], e[g][3]), v;
                const float rsg = quad_lane(rs[i], g);       // the row scale of row 4 a + g (lane g of the quad holds its own)
                if constexpr (CS) { q.x *= cs.x; q.y *= cs.y; q.z *= cs.z; q.w *= cs.w; }
                v.x = fmaf(q.x, rsg, b.x); v.y = fmaf(q.y, rsg, b.y);
                v.z = fmaf(q.z, rsg, b.z); v.w = fmaf(q.w, rsg, b.w);
                if constexpr (R2) {               // the row factors of row 4 a + g: broadcast where they are used (8 registers less)
                    const float a0 = quad_lane((*g0)[i], g), a1 = quad_lane((*g1)[i], g);
                    v.x = fmaf(a1, u1.x, fmaf(a0, u0.x, v.x)); v.y = fmaf(a1, u1.y, fmaf(a0, u0.y, v.y));
                    v.z = fmaf(a1, u1.z, fmaf(a0, u0.z, v.z)); v.w = fmaf(a1, u1.w, fmaf(a0, u0.w, v.w));
                }
                v.x = v.x < floor_ ? floor_ : v.x; v.y = v.y < floor_ ? floor_ : v.y;      // keeps NaN, like torch.relu
                v.z = v.z < floor_ ? floor_ : v.z; v.w = v.w < floor_ ? floor_ : v.w;
                *reinterpret_cast<float4*>(rowbase + (int64_t)g * ldc + j * 32) = v;
            }
        }
    }
}

struct SplitArgs {
    const float* A; int64_t lda;
    const uint16_t* Bp;      // [3][K/16][N][16]
    float* C; int64_t ldc;
    int M, N, K;
    Epilogue ep;
    int tiles_m, tiles_n;    // full 128 x (64 TN) tiles to cover
    // fp16 x 2 only: the power-of-two scale of every row of A ([M], npi_row_scales) and 1 / scale of every column of B ([N], written
    // behind the planes by the preparation kernel); Bp then holds [2][K/16][N][16] fp16
    const float* a_scale; const float* b_inv;
};

constexpr int WS_THREADS = 512;

struct TileWalk {
    int q, kt;          // position: tile sequence number, k-step
    int mt, nt;
    int qend, G, nk, tiles_m, tiles_n;
    int last_row0;      // first row of the LAST m-tile: M - 128 when M is not a multiple of 128 -- that tile then overlaps its
                        // neighbour instead of hanging over the edge; the shared rows are computed twice from the same
                        // inputs in the same order and stored twice with the same bits (no guarded strip launch)
    __device__ __forceinline__ int row0() const { return mt == tiles_m - 1 ? last_row0 : mt * 128; }
    __device__ __forceinline__ void decode() {
        // q -> (m-tile, n-tile): q & 7 = XCD of the workgroup (round-robin dispatch), consecutive slots of an
        // XCD take the n-tiles of one m-tile (their A rows then hit that XCD's L2)
        // (mt, nt) only ever hold a VALID tile: the producers' loads run a few k-steps past the end of the walk and drop the
        // data -- from a slot of the rounded-up grid beyond the last row tile they would read up to 7 x 128 rows past the end
        // of A, which is a memory fault when A ends where its mapping ends
        for (;;) {
            if (q >= qend) return;
            const int nt_ = (q >> 3) % tiles_n;
            const int mt_ = ((q >> 3) / tiles_n) * 8 + (q & 7);
            if (mt_ < tiles_m) { mt = mt_; nt = nt_; return; }
            q += G;
        }
    }
    __device__ __forceinline__ void init(int b, int G_, int nk_, int tm, int tn, int M = -1) {
        G = G_; nk = nk_; tiles_m = tm; tiles_n = tn;
        last_row0 = M >= 128 ? M - 128 : (tm - 1) * 128;
        mt = 0; nt = 0;
        qend = ((tm + 7) / 8) * 8 * tn;
        q = b; kt = 0;
        decode();
    }
    __device__ __forceinline__ bool valid() const { return q < qend; }
    __device__ __forceinline__ void next() {
        if (++kt == nk) { kt = 0; q += G; decode(); }
    }
};

// LDS hand-over counters of the split kernel (workgroup scope)
__device__ __forceinline__ void wait_ge(int* flag, int target) {
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
}
__device__ __forceinline__ void signal(int* flag) {      // one count per wave, after the wave's LDS traffic is complete
    __builtin_amdgcn_s_waitcnt(0xc07f);                     // lgkmcnt(0)
    if (lane_id() == 0) __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// One k-step of a consumer wave of the split kernels, software-pipelined by hand.  The six products of a 32 x 32 tile pair
// are 192 matrix-pipe cycles, a k-step 48 MFMAs = 1,536; the fragment reads of the NEXT step are issued between the blocks
// of this one, into the very registers the finished block just freed (no second fragment set: 128 accumulator + 72
// fragment registers is what two waves per SIMD allow), so that their LDS round trip -- and the poll of the next stage's
// counter -- runs under MFMAs instead of in front of them.  (Reads first, then 48 MFMAs, as hipcc schedules the plain loop:
// 2,060 cycles per step, the matrix pipe idle for a quarter of it.)  Column block j's B fragments are dead after the
// block; the A fragments of row block i after ITS six MFMAs of the last column block, so A(0) is refilled half a block
// before the step ends and A(1) at its end -- the next step starts with the row-block-0 MFMAs, which need only A(0).
// The six products of a tile pair keep their order (small terms first): results are bit-identical to the plain loop.
//   entry: this step's fragments are in registers or in flight;  exit: the next step's (`more`), stage handed back.
// The fragment reads and their waits are assembly: hipcc's own lgkmcnt bookkeeping joins the loop's state with the
// tile-start path (scalar loads pending there) and falls back to lgkmcnt(0) in front of the first MFMA of every step,
// i.e. waits for the reads issued a few cycles earlier.  LDS operations complete in order, so "at most n younger ones
// outstanding" is exact.  Issue order per step: B(0) | B(1) | ... | A(0) | A(1), B(TN-1)  (three planes each).
// The fragments live in registers as four opaque dwords (frag_t) and become bf16x8 only as MFMA operands: kept as a vector of
// eight bf16 across basic blocks, hipcc "re-packs" them between the asm read and the asm wait (v_lshrrev + v_perm pairs that
// are the identity on arrived data) -- on registers whose LDS data is still in flight that mixes stale and new halves.  It
// showed only under contention (the first overlapped backward, while the by-source CSR was being sorted on the other stream):
// dW off by 3 %.  tests: test_split_gemms_are_exact_under_concurrent_load.
typedef u32x4r frag_t;
#define FRAG(x) __builtin_bit_cast(bf16x8, x)
#define NPI_DSR(dst, addr, IMM) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(IMM) : "memory")
// NPL plane images per operand: 3 (bf16 x 3) or 2 (fp16 x 2, F16 below)
template <int PL, int NPL = 3>
__device__ __forceinline__ void ws_read3(frag_t (&f)[NPL], uint32_t addr) {
    NPI_DSR(f[0], addr, 0); NPI_DSR(f[1], addr, PL);
    if constexpr (NPL == 3) NPI_DSR(f[2], addr, 2 * PL);
}
// at most N fragment reads still outstanding; the fragments are in/out operands so that no use moves above the wait
#define NPI_LGKM_WAIT(N, F) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]) : : "memory")
#define NPI_LGKM_WAIT2(N, F) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(F[0]), "+v"(F[1]) : : "memory")
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define FRAGH(x) __builtin_bit_cast(f16x8, x)

// F16 (NPL = 2): the operands are TWO fp16 pieces (scaled into fp16's range by the producer), three products a1 b0 + a0 b1 + a0 b0
template <int TM, int TN, int APL, int BPL, int BUF, int NST, bool ONE_MMA = false, int NPL = 3, bool F16 = false>
__device__ __forceinline__ void ws_consume_step(uint32_t lds_base, int* full, int* empty, int g, bool more,
                                                const int (&offa)[TM], const int (&offb)[TN], frag_t (&af)[TM][NPL],
                                                frag_t (&bf)[TN][NPL], f32x16 (&acc)[TM][TN]) {
    static_assert(TM == 2, "the wait counts below assume two row blocks");
    static_assert(NPL == (F16 ? 2 : 3), "bf16 x 3 has three plane images per operand, fp16 x 2 two");
    const int stg = g & (NST - 1), stn = (g + 1) & (NST - 1);
    const uint32_t nx = lds_base + (uint32_t)(stn * BUF);
    // the poll of the next stage's counter is issued here and looked at after the first column block: its LDS round trip
    // (150-250 cycles behind the fragment reads of four waves) then runs under 12 MFMAs instead of stalling the pipe
    int fullv;
    asm volatile("ds_read_b32 %0, %1" : "=v"(fullv) : "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) int*)&full[stn]) : "memory");
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if constexpr (NPL == 3) {
                if (j == 0 && i == 0) NPI_LGKM_WAIT(7, af[0]);    // younger: A(1) and B(TN-1) of this step, the poll
                if (j == 0 && i == 1) NPI_LGKM_WAIT(4, af[1]);    // younger: B(TN-1), the poll
            } else {                                              // (two reads per fragment set)
                if (j == 0 && i == 0) NPI_LGKM_WAIT2(5, af[0]);
                if (j == 0 && i == 1) NPI_LGKM_WAIT2(3, af[1]);
            }
            // B fragment as the MFMA's first operand: the tile comes out transposed (see store_tile_t)
            f32x16 c = acc[i][j];
            if constexpr (F16) {
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(FRAGH(bf[j][0]), FRAGH(af[i][1]), c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(FRAGH(bf[j][1]), FRAGH(af[i][0]), c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(FRAGH(bf[j][0]), FRAGH(af[i][0]), c, 0, 0, 0);
            } else {
            if (!ONE_MMA) {
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FRAG(bf[j][0]), FRAG(af[i][2]), c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FRAG(bf[j][2]), FRAG(af[i][0]), c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FRAG(bf[j][1]), FRAG(af[i][1]), c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FRAG(bf[j][0]), FRAG(af[i][1]), c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FRAG(bf[j][1]), FRAG(af[i][0]), c, 0, 0, 0);
            }
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FRAG(bf[j][0]), FRAG(af[i][0]), c, 0, 0, 0);
            }
            acc[i][j] = c;
            __builtin_amdgcn_sched_barrier(0);           // row block by row block: the first one of a step needs only A(0)
            if (j == TN - 1 && more) ws_read3<APL, NPL>(af[i], nx + (uint32_t)offa[i]);   // row block i is finished: refill its A
        }
        if (j == 0) {
            // every read of this stage was issued during the previous step: this wait is free by now (it is also what
            // makes B(1..TN-1) of this step safe to use, and the poll's value)
            if constexpr (NPL == 3)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bf[TN - 1][0]), "+v"(bf[TN - 1][1]), "+v"(bf[TN - 1][2]), "+v"(fullv) : : "memory");
            else
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bf[TN - 1][0]), "+v"(bf[TN - 1][1]), "+v"(fullv) : : "memory");
            signal(&empty[stg]);
            const int target = 4 * (((g + 1) >> 2) + 1);
            if (more && __builtin_amdgcn_readfirstlane(fullv) < target) wait_ge(&full[stn], target);   // rare: the producer is behind
            // (the acquire for the stage's data: wait_ge's atomic load on the slow path, program order after the poll's
            // wait on the fast one -- LDS operations of a wave execute in order)
        }
        if (more) ws_read3<BPL, NPL>(bf[j], nx + (uint32_t)offb[j]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// first step's fragments (prologue of the pipeline above)
template <int TM, int TN, int APL, int BPL, int NPL = 3>
__device__ __forceinline__ void ws_consume_first(uint32_t st, const int (&offa)[TM], const int (&offb)[TN],
                                                 frag_t (&af)[TM][NPL], frag_t (&bf)[TN][NPL]) {
#pragma unroll
    for (int j = 0; j < TN; ++j) ws_read3<BPL, NPL>(bf[j], st + (uint32_t)offb[j]);
#pragma unroll
    for (int i = 0; i < TM; ++i) ws_read3<APL, NPL>(af[i], st + (uint32_t)offa[i]);
    if constexpr (NPL == 3) {
#pragma unroll
        for (int j = 0; j < TN; ++j) NPI_LGKM_WAIT(0, bf[j]);
#pragma unroll
        for (int i = 0; i < TM; ++i) NPI_LGKM_WAIT(0, af[i]);
    } else {
#pragma unroll
        for (int j = 0; j < TN; ++j) NPI_LGKM_WAIT2(0, bf[j]);
#pragma unroll
        for (int i = 0; i < TM; ++i) NPI_LGKM_WAIT2(0, af[i]);
    }
}

template <int TN, int EPI = 0, bool F16 = false>
__global__ void __launch_bounds__(WS_THREADS, 1)
gemm_split_ws_kernel(SplitArgs a) {
    constexpr bool R2 = EPI == 1, SC = EPI == 2;
    constexpr bool UV = R2 || SC;                     // two column vectors kept in LDS beside the bias
    constexpr int TM = 2;
    constexpr int NPL = F16 ? 2 : 3;                  // plane images per operand: bf16 x 3, or fp16 x 2 (F16)
    constexpr int BN = 64 * TN;                       // 128 or 256 output columns per tile
    constexpr int APL = 128 * 32, BPL = BN * 32;      // bytes of one A / B plane image
    constexpr int BUF = NPL * APL + NPL * BPL;        // one stage: A planes 0.., B planes 0..
    constexpr int NB = BN / 128;                      // 16-byte B chunks per producer thread and plane
    constexpr int NST = 4;                            // LDS stages (4 x 36 KiB at TN = 4)
    __shared__ __attribute__((aligned(16))) char lds[NST * BUF];
    __shared__ __attribute__((aligned(16))) float bias_s[4][2][32 * TN];   // per consumer wave: bias of its columns, by tile parity
    // R2: the two column vectors of the rank-2 term, kept exactly like the bias
    __shared__ __attribute__((aligned(16))) float r2_s[UV ? 4 : 1][2][2][UV ? 32 * TN : 4];
    __shared__ __attribute__((aligned(16))) float cs_s[F16 ? 4 : 1][2][F16 ? 32 * TN : 4];     // F16: 1 / (column scale), kept like the bias
    // SC: the two consumer waves that share a row block (wn = 0 / 1: the two halves of the columns) meet here -- wave wn = 1
    // parks its half of the 64 rows' dots and counts up sc_flag[wm]; wave wn = 0 waits for the count of ITS tile, adds its own
    // half (fixed order: columns low + high) and stores the rows' two scalars.  By tile parity: wave 1 can be at most NST
    // k-steps ahead of wave 0, never a whole tile.
    __shared__ __attribute__((aligned(8))) float sc_s[SC ? 2 : 1][2][SC ? 64 : 1][2];
    __shared__ int sc_flag[2];
    // hand-over counters, one pair per stage, only ever incremented: the 4 producer waves add to full[s] when their
    // part of a k-step is in stage s, the 4 consumer waves add to empty[s] when their fragments are in registers.
    // No barrier in the main loop: the producer runs up to NST - 1 k-steps ahead, so a consumer epilogue (10 k
    // cycles of stores) no longer stops it, and the consumer finds the next stages ready when it returns.
    __shared__ int full[NST], empty[NST];
    const int t = threadIdx.x;
    const int wave = uniform_i(t >> 6);
    const int nk = a.K / SK;
    TileWalk w;
    w.init((int)blockIdx.x, (int)gridDim.x, nk, a.tiles_m, a.tiles_n, a.M);
    if (!w.valid()) return;                                 // no tile for this workgroup (uniform)
    if (t < NST) { full[t] = 0; empty[t] = 0; }
    if (t < 2) sc_flag[t] = 0;
    __syncthreads();

    if (wave >= 4) {
        // ---------------- producer ----------------
        const int pt = t - 256;
        // Every address is a wave-uniform base (SGPR pair, recomputed per step from the walk) plus a per-thread
        // 32-bit byte offset fixed for the whole kernel.
        // A: float4 #(pt & 3) of rows (pt >> 2) and (pt >> 2) + 64;  B: the NB consecutive 16-B chunks NB pt ... of each plane's tile
        const int64_t b_plane = (int64_t)a.N * a.K * 2;
        const uint32_t oa0 = (uint32_t)(((int64_t)(pt >> 2) * a.lda + (pt & 3) * 4) * 4);
        const uint32_t oa1 = oa0 + (uint32_t)(a.lda * 64 * 4);
        const uint32_t ob = (uint32_t)pt * (16 * NB);
        const int ar = pt >> 2, ac = pt & 3;
        char* la0 = lds + simg(ar, ac >> 1) + (ac & 1) * 8;
        char* la1 = lds + simg(ar + 64, ac >> 1) + (ac & 1) * 8;
        // chunk c -> row c / 2, half c % 2
        char* lb0 = lds + NPL * APL + (NB == 2 ? simg(pt, 0) : simg(pt >> 1, pt & 1));
        char* lb1 = lds + NPL * APL + simg(pt, 1);
        const uint32_t os0 = (uint32_t)(ar * 4), os1 = (uint32_t)((ar + 64) * 4);      // F16: this thread's two rows in a.a_scale
        (void)os0; (void)os1;
        TileWalk wl = w;                                    // load position (runs ahead of the store position w)
        // Register ring of 4 k-steps: three steps of loads are in flight while the fourth is split and stored.
        // The loads and their waits are written in assembly: hipcc's own vmcnt bookkeeping loses the ring at the
        // loop back-edge (it drained to vmcnt(2..6) in one of the four steps) and parks address temporaries in
        // ring registers.  global_load dst, v_off, s[base]: wave-uniform base, fixed per-thread offset.
        // Loads are UNCONDITIONAL: past the last step the walk keeps pointing at a valid tile and the data is
        // dropped (with `if (more) load` the number of loads in flight is not a compile-time fact).
#define NPI_WDECL(S) f32x4r S##a0, S##a1; u32x4r S##b0, S##b1, S##b2, S##b3, S##b4, S##b5; float S##s0 = 1.f, S##s1 = 1.f
        NPI_WDECL(r0); NPI_WDECL(r1); NPI_WDECL(r2); NPI_WDECL(r3);
#define NPI_GL(dst, off, base, IMM)                                                                    \
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #IMM : "=v"(dst) : "v"(off), "s"(base) : "memory")
#define NPI_GL1(dst, off, base)                                                                        \
        asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(off), "s"(base) : "memory")
#define NPI_WLOAD(S)                                                                                   \
    do {                                                                                               \
        const char* ga = uniform_ptr(reinterpret_cast<const char*>(a.A) + ((int64_t)wl.row0() * a.lda + wl.kt * SK) * 4);        \
        const char* gb0 = uniform_ptr(reinterpret_cast<const char*>(a.Bp) + ((int64_t)wl.kt * a.N + wl.nt * BN) * (SK * 2));     \
        const char* gb1 = uniform_ptr(gb0 + b_plane);                                                  \
        const char* gb2 = uniform_ptr(gb0 + 2 * b_plane);                                              \
        if constexpr (F16) {                                                                           \
            /* two planes, and the scales of this thread's two rows: every set has the same number of loads (the waits count them) */ \
            const char* gs = uniform_ptr(reinterpret_cast<const char*>(a.a_scale) + (int64_t)wl.row0() * 4);                       \
            NPI_GL(S##a0, oa0, ga, 0); NPI_GL(S##a1, oa1, ga, 0);                                      \
            NPI_GL(S##b0, ob, gb0, 0); NPI_GL(S##b1, ob, gb1, 0);                                      \
            if constexpr (NB == 2) { NPI_GL(S##b3, ob, gb0, 16); NPI_GL(S##b4, ob, gb1, 16); }         \
            NPI_GL1(S##s0, os0, gs); NPI_GL1(S##s1, os1, gs);                                          \
            (void)gb2;                                                                                 \
        } else {                                                                                       \
        NPI_GL(S##a0, oa0, ga, 0); NPI_GL(S##a1, oa1, ga, 0);                                          \
        NPI_GL(S##b0, ob, gb0, 0); NPI_GL(S##b1, ob, gb1, 0); NPI_GL(S##b2, ob, gb2, 0);               \
        if constexpr (NB == 2) {                                                                       \
            NPI_GL(S##b3, ob, gb0, 16); NPI_GL(S##b4, ob, gb1, 16); NPI_GL(S##b5, ob, gb2, 16);        \
        }                                                                                              \
        }                                                                                              \
        wl.next();                                                                                     \
    } while (0)
        // wait until only the loads of the three younger sets are in flight; the set is an in/out operand so
        // that no use of it can be scheduled above the wait
#define NPI_WWAIT(S)                                                                                   \
    do {                                                                                               \
        if constexpr (F16 && NB == 2)        /* 2 A + 4 B + 2 scales = 8 loads per set */              \
            asm volatile("s_waitcnt vmcnt(24)" : "+v"(S##a0), "+v"(S##a1), "+v"(S##b0), "+v"(S##b1), "+v"(S##b3), "+v"(S##b4), \
                         "+v"(S##s0), "+v"(S##s1) : : "memory");                                       \
        else if constexpr (F16)              /* 2 A + 2 B + 2 scales = 6 */                            \
            asm volatile("s_waitcnt vmcnt(18)" : "+v"(S##a0), "+v"(S##a1), "+v"(S##b0), "+v"(S##b1), "+v"(S##s0), "+v"(S##s1) : : "memory"); \
        else if constexpr (NB == 2)                                                                    \
            asm volatile("s_waitcnt vmcnt(24)" : "+v"(S##a0), "+v"(S##a1), "+v"(S##b0), "+v"(S##b1), "+v"(S##b2), \
                         "+v"(S##b3), "+v"(S##b4), "+v"(S##b5) : : "memory");                          \
        else                                                                                           \
            asm volatile("s_waitcnt vmcnt(15)" : "+v"(S##a0), "+v"(S##a1), "+v"(S##b0), "+v"(S##b1), "+v"(S##b2) : : "memory"); \
    } while (0)
#define NPI_WSTORE(OFF, S)                                                                             \
    do {                                                                                               \
        if constexpr (F16) {                                                                           \
            split2_store(S##a0, S##s0, la0 + (OFF), APL); split2_store(S##a1, S##s1, la1 + (OFF), APL); \
            *reinterpret_cast<u32x4r*>(lb0 + (OFF)) = S##b0;                                           \
            *reinterpret_cast<u32x4r*>(lb0 + (OFF) + BPL) = S##b1;                                     \
            if constexpr (NB == 2) {                                                                   \
                *reinterpret_cast<u32x4r*>(lb1 + (OFF)) = S##b3;                                       \
                *reinterpret_cast<u32x4r*>(lb1 + (OFF) + BPL) = S##b4;                                 \
            }                                                                                          \
            break;                                                                                     \
        }                                                                                              \
        split3_store(S##a0, la0 + (OFF), APL); split3_store(S##a1, la1 + (OFF), APL);                  \
        *reinterpret_cast<u32x4r*>(lb0 + (OFF)) = S##b0;                                               \
        *reinterpret_cast<u32x4r*>(lb0 + (OFF) + BPL) = S##b1;                                         \
        *reinterpret_cast<u32x4r*>(lb0 + (OFF) + 2 * BPL) = S##b2;                                     \
        if constexpr (NB == 2) {                                                                       \
            *reinterpret_cast<u32x4r*>(lb1 + (OFF)) = S##b3;                                           \
            *reinterpret_cast<u32x4r*>(lb1 + (OFF) + BPL) = S##b4;                                     \
            *reinterpret_cast<u32x4r*>(lb1 + (OFF) + 2 * BPL) = S##b5;                                 \
        }                                                                                              \
    } while (0)
        // one k-step: refill the set freed by the previous step, split + store set CUR
#define NPI_WSTEP(ST, CUR, FREE)                                                                       \
        NPI_WLOAD(FREE);                                                                               \
        if (round > 0) wait_ge(&empty[ST], 4 * round);       /* the consumers are done with this stage's previous use */ \
        NPI_WWAIT(CUR);                                                                                \
        NPI_WSTORE((ST) * BUF, CUR);                                                                   \
        signal(&full[ST]);                                                                             \
        w.next();                                                                                      \
        if (!w.valid()) break
        NPI_WLOAD(r0);
        NPI_WLOAD(r1);
        NPI_WLOAD(r2);
        for (int round = 0; w.valid(); ++round) {
            NPI_WSTEP(0, r0, r3);
            NPI_WSTEP(1, r1, r0);
            NPI_WSTEP(2, r2, r1);
            NPI_WSTEP(3, r3, r2);
        }
#undef NPI_WSTEP
#undef NPI_WDECL
#undef NPI_WWAIT
#undef NPI_GL
#undef NPI_GL1
#undef NPI_WLOAD
#undef NPI_WSTORE
        return;
    }

    // ---------------- consumer ----------------
    const int lane = t & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    int offa[TM], offb[TN];                                 // byte offsets of this lane's fragments inside a plane image
#pragma unroll
    for (int i = 0; i < TM; ++i) offa[i] = simg(wm * 64 + i * 32 + li, lh);
#pragma unroll
    for (int j = 0; j < TN; ++j) offb[j] = NPL * APL + simg(wn * (32 * TN) + j * 32 + li, lh);
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    int tsel = 1, bias_nt = -1;                             // bias_s half in use, and the n-tile whose bias it holds
    float rs[TM], g0[TM], g1[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) { rs[i] = 1.f; g0[i] = 0.f; g1[i] = 0.f; }
    const float floor_ = a.ep.relu != 0 ? 0.f : -__builtin_huge_valf();
    int g = 0;                                              // k-steps consumed so far: stage g % NST, use g / NST
    int tq = 0;                                             // SC: tiles this workgroup has finished
    (void)tq;
    frag_t af[TM][NPL], bf[TN][NPL];
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    wait_ge(&full[0], 4);
    ws_consume_first<TM, TN, APL, BPL, NPL>(lds_base, offa, offb, af, bf);
    while (w.valid()) {
        if (w.kt == 0) {
            // start of a tile: the row scales of this lane's rows and the bias of this wave's columns (into the
            // wave's own LDS slot: written and, 16 k-steps later, read by the same wave -- in order, no hand-over)
            if (a.ep.rowscale) {
#pragma unroll
                for (int i = 0; i < TM; ++i) rs[i] = a.ep.rowscale[w.row0() + wm * 64 + i * 32 + li];
            }
            if constexpr (F16) {          // the row's power-of-two scale comes out again (exact)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    rs[i] = (a.ep.rowscale ? rs[i] : 1.f) * pow2_inverse(a.a_scale[w.row0() + wm * 64 + i * 32 + li]);
            }
            if constexpr (R2) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    g0[i] = a.ep.r2_row0[w.row0() + wm * 64 + i * 32 + li];
                    g1[i] = a.ep.r2_row1[w.row0() + wm * 64 + i * 32 + li];
                }
            }
            if (w.nt != bias_nt) {                          // only when the column block changes (once, for N = 64 TN): the
                tsel ^= 1;                                  // copy waits on vmcnt(0), i.e. also on the previous tile's stores
                bias_nt = w.nt;
                for (int c = lane; c < 32 * TN; c += WAVE) {
                    bias_s[wave][tsel][c] = a.ep.bias ? a.ep.bias[w.nt * BN + wn * (32 * TN) + c] : 0.f;
                    if constexpr (UV) {
                        r2_s[wave][tsel][0][c] = a.ep.r2_col0[w.nt * BN + wn * (32 * TN) + c];
                        r2_s[wave][tsel][1][c] = a.ep.r2_col1[w.nt * BN + wn * (32 * TN) + c];
                    }
                    if constexpr (F16) cs_s[wave][tsel][c] = a.b_inv[w.nt * BN + wn * (32 * TN) + c];
                }
            }
        }
        TileWalk wn_ = w;
        wn_.next();
        ws_consume_step<TM, TN, APL, BPL, BUF, NST, false, NPL, F16>(lds_base, full, empty, g, wn_.valid(), offa, offb, af, bf, acc);
        ++g;
        if (w.kt == nk - 1) {
            // full 128-byte lines per store instruction (store_tile_q) where its registers fit: plain, and fp16 x 2 with the rank-2 term
            if constexpr ((EPI != 0 && !F16) || EPI == 2)
                store_tile_t<TM, TN, EPI, F16>(a.C, a.ldc, w.row0() + wm * 64, w.nt * BN + wn * (32 * TN), li, lh, acc,
                                               bias_s[wave][tsel], rs, floor_, r2_s[UV ? wave : 0][tsel][0], r2_s[UV ? wave : 0][tsel][1], &g0, &g1,
                                               cs_s[F16 ? wave : 0][tsel]);
            else
            store_tile_q<TM, TN, F16, EPI>(a.C, a.ldc, w.row0() + wm * 64, w.nt * BN + wn * (32 * TN), li, lh, acc, bias_s[wave][tsel], rs, floor_,
                                           cs_s[F16 ? wave : 0][tsel], r2_s[UV ? wave : 0][tsel][0], r2_s[UV ? wave : 0][tsel][1], &g0, &g1);
            if constexpr (SC) {
                const int par = tq & 1;
#pragma unroll
                for (int i = 0; i < TM; ++i) {              // the other 16 of every 32 columns sit in lane ^ 32
                    g0[i] += __shfl_xor(g0[i], 32, WAVE);
                    g1[i] += __shfl_xor(g1[i], 32, WAVE);
                }
                if (wn == 1) {
                    if (lh == 0) {
#pragma unroll
                        for (int i = 0; i < TM; ++i)
                            *reinterpret_cast<float2*>(&sc_s[wm][par][i * 32 + li][0]) = make_float2(g0[i], g1[i]);
                    }
                    signal(&sc_flag[wm]);
                } else {
                    wait_ge(&sc_flag[wm], tq + 1);
                    if (lh == 0) {
#pragma unroll
                        for (int i = 0; i < TM; ++i) {
                            const float2 o = *reinterpret_cast<const float2*>(&sc_s[wm][par][i * 32 + li][0]);
                            const int row = w.row0() + wm * 64 + i * 32 + li;
                            a.ep.sc0[row] = g0[i] + o.x;
                            a.ep.sc1[row] = g1[i] + o.y;
                        }
                    }
                }
                ++tq;
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
        }
        w = wn_;
    }
}

// ---- dW on the bf16 matrix cores: both operands split on the fly --------------------------------------------------
// dW[K, N] = A[M, K]^T dC[M, N]: the contraction runs over the NODES, which is the slow (row) dimension of both
// operands in memory, while a 32x32x16 MFMA wants 8 consecutive contraction values per lane.  The producer therefore
// loads in FRAGMENT ORDER: a thread takes 4 adjacent columns of 8 consecutive node rows -- eight 16-byte loads, each
// wave instruction still whole 512 B / 1 KiB row segments -- so it holds, per column, the 8 k-values of one fragment
// half; it splits them (3-way, as above) and stores one 16-byte half row per plane into the SAME [row][16 bf16] LDS
// image the split kernel uses.  The consumer waves are therefore unchanged: ds_read_b128 fragments, six MFMAs per
// product tile, hand-over through the per-stage LDS counters.  Six bf16 MFMAs per f32 product = 6/16 of the exact
// kernel's matrix-pipe time; error at f32-rounding level (tests: against fp64).
// One workgroup = one 128 (features of A) x 64 TN (columns of dC) tile of dW over one SLAB of the node range; the slabs
// are summed by slab_reduce_kernel in slab order (deterministic).  db = column sums of dC ride on the B loads of the
// workgroups with m-tile 0 (two partial rows per slab: the two 8-node halves of a k-step).
// The 16-byte LDS stores of 8 adjacent lanes go to 8 different bank quads: lane cg stores its columns in the rotated
// order rho(cg) + c (see dw_rot), which with the image's half swizzle covers all 32 banks.
struct DwArgs {
    const float* A; int64_t lda;     // [M, K]
    const float* dC; int64_t ldc;    // [M, N]
    float* slabs;                    // [nslab][K][N] partial dW
    float* db_slabs;                 // [2 nslab][N] partial column sums of dC, or null
    int K, N;
    int64_t m_main;                  // nodes covered (multiple of 16)
    int64_t per;                     // nodes per slab (multiple of 16)
    int tiles_m, tiles_n, nslab;
    // fp16 x 2 (F16): the power-of-two scale of every COLUMN of A ([K]) and of dC ([N]) -- the contraction runs over the rows, so
    // only column scales factor out of the sum; null otherwise
    const float* a_cs; const float* b_cs;
};

// LDS row of column 4 cg + c of the tile: inside every block of 32 columns the 8 x 4 (cg, c) grid is stored TRANSPOSED,
// row 8 c + cg -- the lanes of a store instruction (fixed c, consecutive cg) then hit consecutive rows.  The consumer's
// fragment of LDS row r therefore belongs to column dw_col(r); only the epilogue has to know (dw_store_tile).
__device__ __forceinline__ int dw_row(int cg, int c) { return 32 * (cg >> 3) + 8 * c + (cg & 7); }
__device__ __forceinline__ int dw_col(int r) { return (r & ~31) + 4 * (r & 7) + ((r >> 3) & 3); }

// BF16IN (round 5): A and dC are stored as bf16.  Their values ARE bf16, so there is nothing to split: the producer loads 8 bytes
// per row (the same 4 columns x 8 node rows per thread), re-pairs the halves into the fragment order and stores ONE plane; the
// consumer issues one MFMA per product tile (it still reads the three plane slots of a fragment -- the hand-tuned wait counts
// of ws_consume_step assume them -- and ignores two).  Half the loader's bytes and a sixth of the matrix work: the kernel is
// HBM-bound.  (Rounds 1-4 ran bf16 dW on the exact-f32 tile kernel: 1.8 ms at C4 against 0.93 for f32 storage.)
// F16 (round 6; f32 operands): two fp16 pieces per operand, every COLUMN of A and of dC scaled into fp16's range by a power of two
// (a_cs / b_cs), three products per tile pair, the slab store undoes both scales (exact).  Where dW is EXPOSED -- GATConv's
// backward, whose big HBM-bound pass produces dW's own operand -- the kernel is bound by its consumer step (2,300 cycles for
// 1,536 of MFMAs on three planes; EXPERIMENTS A36), which two planes shorten as they do in the forward kernel.
template <int TN, bool BF16IN = false, bool F16 = false>
__global__ void __launch_bounds__(WS_THREADS, 1)
gemm_dw_split_kernel(DwArgs a) {
    static_assert(!(BF16IN && F16), "bf16 operands need no split");
    constexpr int TM = 2;
    constexpr int ES = BF16IN ? 2 : 4;                      // bytes per stored operand element
    constexpr int BN = 64 * TN;
    constexpr int APL = 128 * 32, BPL = BN * 32;
    constexpr int NPL = F16 ? 2 : 3;                        // plane images per operand
    constexpr int BUF = NPL * APL + NPL * BPL;
    constexpr int NST = 4;
    constexpr int BSETS = BN / 2;                           // B producer threads: BN / 4 column groups x 2 node halves
    __shared__ __attribute__((aligned(16))) char lds[NST * BUF];
    __shared__ int full[NST], empty[NST];
    const int t = threadIdx.x;
    const int wave = uniform_i(t >> 6);
    // workgroup -> (slab, m-tile, n-tile); the tiles of one slab are neighbours (they share the slab's dC rows in L2)
    // blockIdx -> XCD is round-robin (b % 8): the tiles of one slab take slots 8 apart, i.e. the SAME XCD, so the slab's
    // dC rows (read by every m-tile) and A rows (read by every n-tile) are fetched into that XCD's L2 once
    const int tiles = a.tiles_m * a.tiles_n;
    const int b = (int)blockIdx.x;
    const int slab = (b / (8 * tiles)) * 8 + (b & 7);
    const int tile = (b >> 3) % tiles;
    const int mt = tile / a.tiles_n, nt = tile % a.tiles_n;
    if (slab >= a.nslab) return;
    const int64_t node0 = (int64_t)slab * a.per;
    const int64_t node1 = node0 + a.per < a.m_main ? node0 + a.per : a.m_main;
    const int nk = node1 > node0 ? (int)((node1 - node0) / SK) : 0;      // k-steps of 16 nodes (uniform)
    if (t < NST) { full[t] = 0; empty[t] = 0; }
    __syncthreads();

    if (wave >= 4) {
        // ---------------- producer ----------------
        const int pt = t - 256;
        const bool isA = pt < 64;
        const bool isB = pt >= 64 && pt < 64 + BSETS;
        const int q = isA ? pt : pt - 64;
        // adjacent lanes take the two 8-node halves of the same 4 columns: 8 consecutive lanes then store 4 consecutive LDS
        // rows x 2 halves = 8 different 16-byte bank groups (see dw_row): conflict-free without any per-lane rotation
        const int half = q & 1, cg = q >> 1;
        const int64_t ld = isA ? a.lda : a.ldc;
        char* img = lds + (isA ? 0 : NPL * APL);
        const int plane = isA ? APL : BPL;
        const bool do_db = isB && a.db_slabs != nullptr && mt == 0;
        float dbs[4] = {0.f, 0.f, 0.f, 0.f};
        float cs[4] = {1.f, 1.f, 1.f, 1.f};                 // F16: the scales of this thread's four columns, for the whole launch
        if constexpr (F16) {
            if (isA || isB) {
                const float4 c4 = *reinterpret_cast<const float4*>((isA ? a.a_cs + mt * 128 : a.b_cs + nt * BN) + 4 * cg);
                cs[0] = c4.x; cs[1] = c4.y; cs[2] = c4.z; cs[3] = c4.w;
            }
        }
        (void)cs;
        // Three register sets: two k-steps of loads (16 x 16 B per thread) stay in flight while the third is split and
        // stored -- with one set ahead the kernel was latency-bound (2.1 us per k-step, 2.9 TB/s).  As in the split kernel
        // above the loads and their waits are written in assembly: hipcc's own vmcnt bookkeeping drains the ring at the
        // hand-over spin loop and at the loop back-edge (it waited vmcnt(0..7) with 24 loads meant to be in flight).
        // global_load dst, v_off, s[base]: wave-uniform row base (one per node row), fixed per-thread byte offset.
        // (wave-uniform values are forced into SGPRs, so that the row bases below are scalar arithmetic: a base produced
        // by VALU + v_readfirstlane right in front of the load that reads it violates the VALU-writes-SGPR -> VMEM-reads
        // hazard -- 5 wait states -- which hipcc does not pad for an inline-asm consumer: a memory fault in the first build)
        const char* srcb = uniform_ptr(isA ? reinterpret_cast<const char*>(a.A) + (int64_t)mt * 128 * ES
                                           : reinterpret_cast<const char*>(a.dC) + (int64_t)nt * BN * ES);
        const int64_t ldb = ((int64_t)uniform_i((int)(ld >> 32)) << 32 | (uint32_t)uniform_i((int)(ld & 0xffffffff))) * ES;   // row pitch in bytes
        const uint32_t voff = (uint32_t)(((int64_t)8 * half * ld + 4 * cg) * ES);
        // every slab walks its node range from a different starting step (and wraps): 256 workgroups that all start on a
        // slab boundary -- addresses a multiple of 16 KiB apart -- otherwise sweep the memory channels in lockstep
        const int phase = nk < 2 ? 0 : (int)(((int64_t)slab * 37) % nk);
        const bool active = isA || isB;
        if constexpr (BF16IN) {
            // ---- bf16 operands: dwordx2 loads (4 bf16 columns of one node row), one plane, no split ----
            typedef uint32_t u32x2r __attribute__((ext_vector_type(2)));
#define DWB_GL(dst, base) asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %1, %2 nt" : "=v"(dst) : "v"(voff), "s"(base) : "memory")
#define DWB_DECL(S) u32x2r S##0, S##1, S##2, S##3, S##4, S##5, S##6, S##7
#define DWB_LOAD(S, KS)                                                                                \
    do {                                                                                               \
        const int kc_ = (KS) < nk ? (KS) : nk - 1;   /* past the end: reload the last step, dropped */ \
        const int kk_ = kc_ + phase < nk ? kc_ + phase : kc_ + phase - nk;   /* staggered start, wraps */ \
        const char* g_ = srcb + (node0 + (int64_t)kk_ * SK) * ldb;                                     \
        DWB_GL(S##0, g_);           DWB_GL(S##1, g_ + ldb);                                            \
        DWB_GL(S##2, g_ + 2 * ldb); DWB_GL(S##3, g_ + 3 * ldb);                                        \
        DWB_GL(S##4, g_ + 4 * ldb); DWB_GL(S##5, g_ + 5 * ldb);                                        \
        DWB_GL(S##6, g_ + 6 * ldb); DWB_GL(S##7, g_ + 7 * ldb);                                        \
    } while (0)
#define DWB_WAIT16(S)                                                                                  \
        asm volatile("s_waitcnt vmcnt(16)" : "+v"(S##0), "+v"(S##1), "+v"(S##2), "+v"(S##3), "+v"(S##4), "+v"(S##5), \
                     "+v"(S##6), "+v"(S##7) : : "memory")
            auto process_b = [&](int ks, u32x2r c0, u32x2r c1, u32x2r c2, u32x2r c3, u32x2r c4, u32x2r c5, u32x2r c6, u32x2r c7) {
                const int stg = ks & (NST - 1), round = ks >> 2;
                if (round > 0) wait_ge(&empty[stg], 4 * round);
                char* st = img + stg * BUF;
                const u32x2r rows[8] = {c0, c1, c2, c3, c4, c5, c6, c7};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    // column c of the thread's 4: the low (c even) or high (c odd) half of word c / 2 of every node row
                    uint32_t w[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) w[r] = rows[r][c >> 1];
                    uint32_t p0[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i)                 // node rows 2 i and 2 i + 1 -> one dword of the fragment half
                        p0[i] = (c & 1) ? ((w[2 * i] >> 16) | (w[2 * i + 1] & 0xffff0000u)) : ((w[2 * i] & 0xffffu) | (w[2 * i + 1] << 16));
                    if (do_db) {
                        float f[8];
#pragma unroll
                        for (int r = 0; r < 8; ++r) f[r] = __uint_as_float((c & 1) ? (w[r] & 0xffff0000u) : (w[r] << 16));
                        dbs[c] += ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
                    }
                    *reinterpret_cast<uint4*>(st + simg(dw_row(cg, c), half)) = make_uint4(p0[0], p0[1], p0[2], p0[3]);
                }
                signal(&full[stg]);
            };
            if (active && nk > 0) {                            // wave-uniform
                DWB_DECL(ra); DWB_DECL(rb); DWB_DECL(rc);
                DWB_LOAD(ra, 0);
                DWB_LOAD(rb, 1);
                for (int ks = 0; ks < nk; ks += 3) {
                    DWB_LOAD(rc, ks + 2);
                    DWB_WAIT16(ra);
                    process_b(ks, ra0, ra1, ra2, ra3, ra4, ra5, ra6, ra7);
                    if (ks + 1 >= nk) break;
                    DWB_LOAD(ra, ks + 3);
                    DWB_WAIT16(rb);
                    process_b(ks + 1, rb0, rb1, rb2, rb3, rb4, rb5, rb6, rb7);
                    if (ks + 2 >= nk) break;
                    DWB_LOAD(rb, ks + 4);
                    DWB_WAIT16(rc);
                    process_b(ks + 2, rc0, rc1, rc2, rc3, rc4, rc5, rc6, rc7);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                for (int ks = 0; ks < nk; ++ks) {
                    const int stg = ks & (NST - 1), round = ks >> 2;
                    if (round > 0) wait_ge(&empty[stg], 4 * round);
                    signal(&full[stg]);
                }
            }
#undef DWB_GL
#undef DWB_DECL
#undef DWB_LOAD
#undef DWB_WAIT16
        } else {
// The operands of dW are read ONCE (1 GB each at C4) while the backward aggregation beside it lives off what the caches hold of
// its gathered rows: non-temporal loads (same-box A/B builds at C4, twice: step 6.753 / 6.764 -> 6.692 / 6.681 ms; the same hint
// on the A loads of the forward / bwd_data kernel, whose rows the aggregation has just written or will just read: +0.05 ms)
#define DW_GL(dst, base) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 nt" : "=v"(dst) : "v"(voff), "s"(base) : "memory")
#define DW_DECL(S) f32x4r S##0, S##1, S##2, S##3, S##4, S##5, S##6, S##7
#define DW_LOAD(S, KS)                                                                                 \
    do {                                                                                               \
        const int kc_ = (KS) < nk ? (KS) : nk - 1;   /* past the end: reload the last step, dropped */ \
        const int kk_ = kc_ + phase < nk ? kc_ + phase : kc_ + phase - nk;   /* staggered start, wraps */ \
        const char* g_ = srcb + (node0 + (int64_t)kk_ * SK) * ldb;                                     \
        DW_GL(S##0, g_);           DW_GL(S##1, g_ + ldb);                                              \
        DW_GL(S##2, g_ + 2 * ldb); DW_GL(S##3, g_ + 3 * ldb);                                          \
        DW_GL(S##4, g_ + 4 * ldb); DW_GL(S##5, g_ + 5 * ldb);                                          \
        DW_GL(S##6, g_ + 6 * ldb); DW_GL(S##7, g_ + 7 * ldb);                                          \
    } while (0)
        // the set is an in/out operand of the wait, so that no use of it can be scheduled above the wait
#define DW_WAIT16(S)                                                                                   \
        asm volatile("s_waitcnt vmcnt(16)" : "+v"(S##0), "+v"(S##1), "+v"(S##2), "+v"(S##3), "+v"(S##4), "+v"(S##5), \
                     "+v"(S##6), "+v"(S##7) : : "memory")
        auto process = [&](int ks, f32x4r c0, f32x4r c1, f32x4r c2, f32x4r c3, f32x4r c4, f32x4r c5, f32x4r c6, f32x4r c7) {
            const int stg = ks & (NST - 1), round = ks >> 2;
            if (round > 0) wait_ge(&empty[stg], 4 * round);
            char* st = img + stg * BUF;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float v[8] = {c0[c], c1[c], c2[c], c3[c], c4[c], c5[c], c6[c], c7[c]};
                if (do_db) dbs[c] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                char* dst = st + simg(dw_row(cg, c), half);
                if constexpr (F16) {
                    uint32_t p0[4], p1[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) split2_pair(v[2 * i] * cs[c], v[2 * i + 1] * cs[c], p0[i], p1[i]);
                    *reinterpret_cast<uint4*>(dst) = make_uint4(p0[0], p0[1], p0[2], p0[3]);
                    *reinterpret_cast<uint4*>(dst + plane) = make_uint4(p1[0], p1[1], p1[2], p1[3]);
                } else {
                uint32_t p0[4], p1[4], p2[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    split3_pair(v[2 * i], v[2 * i + 1], p0[i], p1[i], p2[i]);
                }
                *reinterpret_cast<uint4*>(dst) = make_uint4(p0[0], p0[1], p0[2], p0[3]);
                *reinterpret_cast<uint4*>(dst + plane) = make_uint4(p1[0], p1[1], p1[2], p1[3]);
                *reinterpret_cast<uint4*>(dst + 2 * plane) = make_uint4(p2[0], p2[1], p2[2], p2[3]);
                }
            }
            signal(&full[stg]);
        };
        if (active && nk > 0) {                                // wave-uniform
            DW_DECL(ra); DW_DECL(rb); DW_DECL(rc);
            DW_LOAD(ra, 0);
            DW_LOAD(rb, 1);
            for (int ks = 0; ks < nk; ks += 3) {
                DW_LOAD(rc, ks + 2);
                DW_WAIT16(ra);
                process(ks, ra0, ra1, ra2, ra3, ra4, ra5, ra6, ra7);
                if (ks + 1 >= nk) break;
                DW_LOAD(ra, ks + 3);
                DW_WAIT16(rb);
                process(ks + 1, rb0, rb1, rb2, rb3, rb4, rb5, rb6, rb7);
                if (ks + 2 >= nk) break;
                DW_LOAD(rb, ks + 4);
                DW_WAIT16(rc);
                process(ks + 2, rc0, rc1, rc2, rc3, rc4, rc5, rc6, rc7);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the dropped tail loads land before the registers are reused
        } else {
            // a producer wave without a share of this tile shape only keeps the hand-over counters in step
            for (int ks = 0; ks < nk; ++ks) {
                const int stg = ks & (NST - 1), round = ks >> 2;
                if (round > 0) wait_ge(&empty[stg], 4 * round);
                signal(&full[stg]);
            }
        }
#undef DW_GL
#undef DW_DECL
#undef DW_LOAD
#undef DW_WAIT16
        }   // f32 operands
        if (do_db) {
            float* o = a.db_slabs + ((int64_t)slab * 2 + half) * a.N + (int64_t)nt * BN + 4 * cg;
            *reinterpret_cast<float4*>(o) = make_float4(dbs[0], dbs[1], dbs[2], dbs[3]);
        }
        return;
    }

    // ---------------- consumer: the split kernel's loop on one tile ----------------
    const int lane = t & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    int offa[TM], offb[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) offa[i] = simg(wm * 64 + i * 32 + li, lh);
#pragma unroll
    for (int j = 0; j < TN; ++j) offb[j] = NPL * APL + simg(wn * (32 * TN) + j * 32 + li, lh);
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    if (nk > 0) {
        frag_t af[TM][NPL], bf[TN][NPL];
        const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
        wait_ge(&full[0], 4);
        ws_consume_first<TM, TN, APL, BPL, NPL>(lds_base, offa, offb, af, bf);
        for (int g = 0; g < nk; ++g)
            ws_consume_step<TM, TN, APL, BPL, BUF, NST, BF16IN, NPL, F16>(lds_base, full, empty, g, g + 1 < nk, offa, offb, af, bf, acc);
    }
    // the tile of this slab (zeros when the slab holds no node: slab_reduce adds every slab)
    // Accumulator tile = C^T of the LDS-row grid: the lane owns LDS row li of the A image, its registers run along LDS rows
    // 8 (q >> 2) + 4 lh + (q & 3) of the B image; both are permuted columns (dw_col).  128 KiB per workgroup, once: 4-byte stores.
    float* __restrict__ C = a.slabs + (int64_t)slab * a.K * a.N;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = mt * 128 + dw_col(wm * 64 + i * 32 + li);
        float ia = 1.f;                                     // F16: 1 / (scale of A's column = this row of dW); exact
        if constexpr (F16) ia = pow2_inverse(a.a_cs[row]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int col = nt * BN + dw_col(wn * (32 * TN) + j * 32 + 8 * (q >> 2) + 4 * lh + (q & 3));
                float v = acc[i][j][q];
                if constexpr (F16) v = (v * ia) * pow2_inverse(a.b_cs[col]);     // one after the other: no underflow on the way out
                C[(int64_t)row * a.N + col] = v;
            }
    }
}

// ---- bf16 storage: the same persistent producer / consumer pipeline on plain bf16 MFMAs ---------------------
// A [M, K] and C [M, N] are bf16, accumulation f32 (v_mfma_f32_32x32x16_bf16, one product per tile pair).
// A k-step is 64 wide: a stage holds FOUR 16-wide k-blocks in the 32-byte-row image of the split kernel, so the
// work between two hand-overs is 32 MFMAs per consumer wave; three stages.  The producer only copies
// (global_load_dwordx4 -> ds_write_b128): the kernel is HBM-bound.  B = the weight matrix, re-laid once per call
// into k-block-major order by bf16_blocks_kernel.
__global__ void bf16_blocks_kernel(const uint16_t* __restrict__ B, int64_t ldb, int K, int N, int bmode,
                                   uint16_t* __restrict__ blocks) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * K) return;
    const int n = (int)(i / K), k = (int)(i % K);
    blocks[((int64_t)(k / SK) * N + n) * SK + (k % SK)] = bmode == 0 ? B[(int64_t)k * ldb + n] : B[(int64_t)n * ldb + k];
}

struct Bf16Args {
    const uint16_t* A; int64_t lda;
    const uint16_t* Bp;      // [K/16][N][16]
    uint16_t* C; int64_t ldc;
    int M, N, K;
    const uint16_t* bias;    // bf16 [N] or null
    const float* rowscale;   // f32 [M] or null
    int relu;
    int tiles_m, tiles_n;
};

__device__ __forceinline__ uint32_t pack2_bf16(float x, float y) {
    f32x2v v = {x, y};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2v));
}

template <int TN>
__global__ void __launch_bounds__(WS_THREADS, 1)
gemm_bf16_ws_kernel(Bf16Args a) {
    constexpr int TM = 2;
    constexpr int BN = 64 * TN;
    constexpr int APL = 128 * 32, BPL = BN * 32;      // bytes of one 16-wide k-block image of A / B
    constexpr int KB = 4;                             // k-blocks per stage (k-step 64)
    constexpr int BUF = KB * (APL + BPL);
    constexpr int NST = 3;
    constexpr int NB = BN / 128;                      // 16-byte B chunks per producer thread and k-block
    __shared__ __attribute__((aligned(16))) char lds[NST * BUF];
    __shared__ __attribute__((aligned(16))) float bias_s[4][2][32 * TN];
    __shared__ int full[NST], empty[NST];
    const int t = threadIdx.x;
    const int wave = uniform_i(t >> 6);
    const int nk = a.K / (SK * KB);
    TileWalk w;
    w.init((int)blockIdx.x, (int)gridDim.x, nk, a.tiles_m, a.tiles_n, a.M);    // a ragged last row tile overlaps its neighbour (TileWalk::row0)
    if (!w.valid()) return;
    if (t < NST) { full[t] = 0; empty[t] = 0; }
    __syncthreads();

    if (wave >= 4) {
        // ---------------- producer: copy ----------------
        const int pt = t - 256;
        // A: the 64 contiguous bytes (pt & 1) of row pt >> 1 = chunks 4 (pt & 1) .. + 3 of its 128-byte k-step;
        // chunk c lies in k-block c >> 1, half c & 1.   B: NB chunks of every k-block.
        const uint32_t oa = (uint32_t)(((int64_t)(pt >> 1) * a.lda) * 2 + (pt & 1) * 64);
        const uint32_t ob = (uint32_t)pt * (16 * NB);
        const int64_t b_blk = (int64_t)a.N * SK * 2;                 // bytes between consecutive k-blocks of Bp
        char* la = lds + ((pt & 1) * 2) * APL + simg(pt >> 1, 0);    // chunk i of the thread: k-block (pt & 1) * 2 + (i >> 1), half i & 1
        char* la_h = lds + ((pt & 1) * 2) * APL + simg(pt >> 1, 1);
        char* lb0 = lds + KB * APL + (NB == 2 ? simg(pt, 0) : simg(pt >> 1, pt & 1));
        char* lb1 = lds + KB * APL + simg(pt, 1);
        TileWalk wl = w;
#define NPI_BDECL(S) u32x4r S##a0, S##a1, S##a2, S##a3, S##b0, S##b1, S##b2, S##b3, S##b4, S##b5, S##b6, S##b7
        NPI_BDECL(r0); NPI_BDECL(r1); NPI_BDECL(r2);
#define NPI_GL(dst, off, base, IMM)                                                                    \
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #IMM : "=v"(dst) : "v"(off), "s"(base) : "memory")
#define NPI_BLOAD(S)                                                                                   \
    do {                                                                                               \
        const char* ga = uniform_ptr(reinterpret_cast<const char*>(a.A) + ((int64_t)wl.row0() * a.lda + wl.kt * (SK * KB)) * 2);   \
        const char* g0 = uniform_ptr(reinterpret_cast<const char*>(a.Bp) + ((int64_t)wl.kt * KB * a.N + wl.nt * BN) * (SK * 2));     \
        const char* g1 = uniform_ptr(g0 + b_blk);                                                      \
        const char* g2 = uniform_ptr(g0 + 2 * b_blk);                                                  \
        const char* g3 = uniform_ptr(g0 + 3 * b_blk);                                                  \
        NPI_GL(S##a0, oa, ga, 0); NPI_GL(S##a1, oa, ga, 16); NPI_GL(S##a2, oa, ga, 32); NPI_GL(S##a3, oa, ga, 48); \
        NPI_GL(S##b0, ob, g0, 0); NPI_GL(S##b1, ob, g1, 0); NPI_GL(S##b2, ob, g2, 0); NPI_GL(S##b3, ob, g3, 0);     \
        if constexpr (NB == 2) {                                                                       \
            NPI_GL(S##b4, ob, g0, 16); NPI_GL(S##b5, ob, g1, 16); NPI_GL(S##b6, ob, g2, 16); NPI_GL(S##b7, ob, g3, 16); \
        }                                                                                              \
        wl.next();                                                                                     \
    } while (0)
#define NPI_BWAIT(S)                                                                                   \
    do {                                                                                               \
        if constexpr (NB == 2)                                                                         \
            asm volatile("s_waitcnt vmcnt(24)" : "+v"(S##a0), "+v"(S##a1), "+v"(S##a2), "+v"(S##a3), "+v"(S##b0), "+v"(S##b1), \
                         "+v"(S##b2), "+v"(S##b3), "+v"(S##b4), "+v"(S##b5), "+v"(S##b6), "+v"(S##b7) : : "memory");   \
        else                                                                                           \
            asm volatile("s_waitcnt vmcnt(16)" : "+v"(S##a0), "+v"(S##a1), "+v"(S##a2), "+v"(S##a3), "+v"(S##b0), "+v"(S##b1), \
                         "+v"(S##b2), "+v"(S##b3) : : "memory");                                       \
    } while (0)
#define NPI_BSTORE(OFF, S)                                                                             \
    do {                                                                                               \
        *reinterpret_cast<u32x4r*>(la + (OFF)) = S##a0;                                                \
        *reinterpret_cast<u32x4r*>(la_h + (OFF)) = S##a1;                                              \
        *reinterpret_cast<u32x4r*>(la + (OFF) + APL) = S##a2;                                          \
        *reinterpret_cast<u32x4r*>(la_h + (OFF) + APL) = S##a3;                                        \
        *reinterpret_cast<u32x4r*>(lb0 + (OFF)) = S##b0;                                               \
        *reinterpret_cast<u32x4r*>(lb0 + (OFF) + BPL) = S##b1;                                         \
        *reinterpret_cast<u32x4r*>(lb0 + (OFF) + 2 * BPL) = S##b2;                                     \
        *reinterpret_cast<u32x4r*>(lb0 + (OFF) + 3 * BPL) = S##b3;                                     \
        if constexpr (NB == 2) {                                                                       \
            *reinterpret_cast<u32x4r*>(lb1 + (OFF)) = S##b4;                                           \
            *reinterpret_cast<u32x4r*>(lb1 + (OFF) + BPL) = S##b5;                                     \
            *reinterpret_cast<u32x4r*>(lb1 + (OFF) + 2 * BPL) = S##b6;                                 \
            *reinterpret_cast<u32x4r*>(lb1 + (OFF) + 3 * BPL) = S##b7;                                 \
        }                                                                                              \
    } while (0)
#define NPI_BSTEP(ST, CUR, FREE)                                                                       \
        NPI_BLOAD(FREE);                                                                               \
        if (round > 0) wait_ge(&empty[ST], 4 * round);                                                 \
        NPI_BWAIT(CUR);                                                                                \
        NPI_BSTORE((ST) * BUF, CUR);                                                                   \
        signal(&full[ST]);                                                                             \
        w.next();                                                                                      \
        if (!w.valid()) break
        NPI_BLOAD(r0);
        NPI_BLOAD(r1);
        for (int round = 0; w.valid(); ++round) {
            NPI_BSTEP(0, r0, r2);
            NPI_BSTEP(1, r1, r0);
            NPI_BSTEP(2, r2, r1);
        }
#undef NPI_BSTEP
#undef NPI_BSTORE
#undef NPI_BWAIT
#undef NPI_BLOAD
#undef NPI_GL
#undef NPI_BDECL
        return;
    }

    // ---------------- consumer ----------------
    const int lane = t & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    int offa[TM], offb[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) offa[i] = simg(wm * 64 + i * 32 + li, lh);
#pragma unroll
    for (int j = 0; j < TN; ++j) offb[j] = KB * APL + simg(wn * (32 * TN) + j * 32 + li, lh);
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    int tsel = 1, bias_nt = -1;
    float rs[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) rs[i] = 1.f;
    const bool relu_on = a.relu != 0;
    int g = 0, stg = 0;
    while (w.valid()) {
        if (w.kt == 0) {
            if (a.rowscale) {
#pragma unroll
                for (int i = 0; i < TM; ++i) rs[i] = a.rowscale[w.row0() + wm * 64 + i * 32 + li];
            }
            if (a.bias && w.nt != bias_nt) {
                tsel ^= 1;
                bias_nt = w.nt;
                for (int c = lane; c < 32 * TN; c += WAVE)
                    bias_s[wave][tsel][c] = __uint_as_float((uint32_t)a.bias[w.nt * BN + wn * (32 * TN) + c] << 16);
            }
        }
        wait_ge(&full[stg], 4 * (g / NST + 1));
        const char* st = lds + stg * BUF;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            bf16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(st + offa[i] + kb * APL);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(st + offb[j] + kb * BPL);
            if (kb == KB - 1) {                              // every read of the stage has been issued: hand it back
                asm volatile("" ::: "memory");
                signal(&empty[stg]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);   // transposed tile
        }
        ++g;
        stg = (stg + 1 == NST) ? 0 : stg + 1;
        if (w.kt == nk - 1) {
            // a lane owns row li of each 32-row tile; registers 4 g .. 4 g + 3 are four consecutive columns: 8-byte stores
            const float* __restrict__ bl = a.bias ? bias_s[wave][tsel] : nullptr;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                uint16_t* __restrict__ crow = a.C + (int64_t)(w.row0() + wm * 64 + i * 32 + li) * a.ldc + w.nt * BN + wn * (32 * TN) + 4 * lh;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const float4 b = bl ? *reinterpret_cast<const float4*>(bl + j * 32 + 8 * q4 + 4 * lh) : make_float4(0.f, 0.f, 0.f, 0.f);
                        float v0 = fmaf(acc[i][j][4 * q4 + 0], rs[i], b.x), v1 = fmaf(acc[i][j][4 * q4 + 1], rs[i], b.y);
                        float v2 = fmaf(acc[i][j][4 * q4 + 2], rs[i], b.z), v3 = fmaf(acc[i][j][4 * q4 + 3], rs[i], b.w);
                        if (relu_on) {
                            v0 = v0 < 0.f ? 0.f : v0; v1 = v1 < 0.f ? 0.f : v1; v2 = v2 < 0.f ? 0.f : v2; v3 = v3 < 0.f ? 0.f : v3;
                        }
                        *reinterpret_cast<uint2*>(crow + j * 32 + 8 * q4) = make_uint2(pack2_bf16(v0, v1), pack2_bf16(v2, v3));
                    }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
        }
        w.next();
    }
}

static bool vec4_ok(const void* p, int64_t ld, int64_t inner_extent, int es = 4) {
    return ((uintptr_t)p % (4 * es) == 0) && (ld % 4 == 0) && (inner_extent % 4 == 0);
}

// Workgroups of the dW GEMM (reduction over the nodes, split into slabs).  Two regimes:
//   alone   : ~4 workgroups per CU (1,024): fastest when nothing else runs (1.30 ms at C4);
//   shared  : about three workgroups per four CUs -- as a light resident it leaves room for the backward
//             aggregation to co-run on every CU (functional.py, OVERLAP_STREAMS).  Step time at C4 by workgroup
//             count: 128: 7.50, 160-224: 7.11-7.12, 256: 7.22, 384: 7.38, 512: 7.55 ms; alone that grid takes 1.77 ms.
// The caller says which one applies (npi_linear_bwd_weight_ex's `shared`).
// CUs of the current device (256 on MI355X), asked once: the persistent kernels launch one workgroup per CU
static int device_cus() {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8)
            n = 256;
        return (n / 8) * 8;
    }();
    return cus;
}
static int64_t dw_workgroups(bool shared) {
    if (!shared) return 1024;
    static const int64_t ctas = [] {
        long v = 0;
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) v = (3 * cus) / 4;
        if (v > 1024) v = 1024;                          // the workspace is sized for the 1,024-workgroup plan
        return (int64_t)(v > 0 ? v : 192);
    }();
    return ctas;
}
static int pick_splits(int64_t M, int64_t tiles, bool shared) {
    int64_t want = ceil_div(dw_workgroups(shared), tiles);
    int64_t maxs = ceil_div(M, (int64_t)BK * 8);                 // at least 8 k-steps per slab
    int64_t s = want < maxs ? want : maxs;
    return (int)(s < 1 ? 1 : s);
}

// Cover the (M x N) output with the FAST kernel on full tiles and the 128x128 EDGE kernel on the
// two ragged strips.  `splits` slabs along the contraction; the fast path needs K % BK == 0.
// dtype_in: storage of A and B; dtype_out: storage of C and bias.  bf16 runs the guarded kernel only.
template <int AMODE, int BMODE>
// `mode`: 0 = exact-f32 MFMA kernels, 1 = bf16 matrix cores where the tile shape allows (3-way split for f32 storage).
// `scratch`: caller memory for the re-laid weight matrix (npi_linear_workspace_bytes).  Nothing is allocated here: a shape that
// takes the matrix-core kernels without it is an error (ABI 3; the entry points require the workspace).
// `k_valid` > 0 (AMODE 0, f32): A has a.K columns of which only the first k_valid are data, the rest ZERO, and B has only
// k_valid rows (NPI_GEMM_A_ZERO_PADDED): the split kernel runs on a.K with the weight planes zero-extended, everything
// else (guarded strips, the exact kernels) on k_valid.
static int launch_gemm(bool v4, GemmArgs a, int splits, hipStream_t stream, int dtype_in = NPI_F32, int dtype_out = NPI_F32,
                       int mode = 1, void* scratch = nullptr, int k_valid = 0, bool prepared = false, int reserve_cus = 0,
                       const float* a_scales = nullptr) {
    // `a_scales` (NPI_GEMM_SPLIT_F16X2): the power-of-two scale of every row of A (npi_row_scales): the split kernel then runs its
    // fp16 x 2 variant -- three matrix products per tile pair instead of six, the same f32-level accuracy -- on fp16 weight planes
    // `reserve_cus`: the persistent kernels take that many workgroups fewer than CUs (a multiple of 8: one per XCD), so that a
    // kernel resident beside them -- a collective's -- holds CUs they do not wait for (NPI_GEMM_RESERVE_CUS)
    const int cu_slots = device_cus() - ((reserve_cus < 0 ? 0 : reserve_cus > 128 ? 128 : reserve_cus) / 8) * 8;
    // `prepared`: `scratch` already holds the re-laid weight matrix of THIS B / K / N / BMODE (npi_linear_prepare): no
    // preparation launch in front of the GEMM
    const bool bf16_in = dtype_in == NPI_BF16 && dtype_out == NPI_F32;          // dW of the bf16 path: bf16 operands, f32 slabs
    // (a 128 x 256 tile of the exact-f32 kernel, one workgroup per CU, measured 3-10 % SLOWER than 128 x 128 at C4 in round 1:
    // 1.49 / 1.32 / 1.67 ms vs 1.41 / 1.28 / 1.52 ms -- not built)
    const int kv = k_valid > 0 ? k_valid : a.K;
    if (kv != a.K) {
        const bool can_split = AMODE == 0 && v4 && (a.K % BK == 0) && dtype_in == NPI_F32 && dtype_out == NPI_F32 && splits == 1 &&
                               mode != 0 && a.ep.colsum == nullptr && a.M >= 128 && a.N >= 128 && ((uintptr_t)a.C % 16 == 0) &&
                               (a.ldc % 4 == 0) && ((uintptr_t)a.ep.bias % 16 == 0);
        if (!can_split) { a.K = kv; a.kchunk = (int)align_up(kv, BK); }          // the pad columns are zero: dropping them is exact
    }
    const bool fast_ok = v4 && (a.K % BK == 0) && (a.K > 0) && ((dtype_in == NPI_F32 && dtype_out == NPI_F32) || bf16_in);
    const int bm = 128, bn = 128;
    const int fm = fast_ok ? a.M / bm : 0, fn = fast_ok ? a.N / bn : 0;    // full tiles
    // bf16 storage: interior tiles on the bf16 MFMA pipeline (K % 64 == 0, 16-byte aligned rows); the rest guarded
    const bool bf16_ws = AMODE == 0 && splits == 1 && dtype_in == NPI_BF16 && dtype_out == NPI_BF16 && a.ep.colsum == nullptr &&
                         a.K % 64 == 0 && a.N % 128 == 0 && a.M >= 128 && (a.lda % 8 == 0) && (a.ldc % 4 == 0) &&
                         ((uintptr_t)a.A % 16 == 0) && ((uintptr_t)a.C % 8 == 0) && mode != 0;
    if (bf16_ws) {
        uint16_t* blocks = reinterpret_cast<uint16_t*>(scratch);
        const int64_t nel = (int64_t)a.N * a.K;
        if (blocks == nullptr) {
            set_error("gemm: the bf16 matrix-core kernel needs the caller's workspace (npi_linear_workspace_bytes)");
            return NPI_ERR_WORKSPACE;
        }
        if (!(prepared && scratch != nullptr))
            bf16_blocks_kernel<<<(unsigned)ceil_div(nel, 256), 256, 0, stream>>>(reinterpret_cast<const uint16_t*>(a.B), a.ldb, a.K, a.N, BMODE, blocks);
        const int bfm = (int)ceil_div(a.M, 128);             // the last row tile starts at M - 128: no guarded strip launch
        const bool wide_n = (a.N % 256 == 0);
        const int btn = wide_n ? a.N / 256 : a.N / 128;
        Bf16Args ba{reinterpret_cast<const uint16_t*>(a.A), a.lda, blocks, reinterpret_cast<uint16_t*>(a.C), a.ldc, a.M, a.N, a.K,
                    reinterpret_cast<const uint16_t*>(a.ep.bias), a.ep.rowscale, a.ep.relu, bfm, btn};
        const int64_t ntiles = (int64_t)bfm * btn;
        const int grid = (int)(ntiles < cu_slots ? ((ntiles + 7) / 8) * 8 : cu_slots);
        if (wide_n) gemm_bf16_ws_kernel<4><<<grid, WS_THREADS, 0, stream>>>(ba);
        else        gemm_bf16_ws_kernel<2><<<grid, WS_THREADS, 0, stream>>>(ba);
        return NPI_OK;
    }
    const bool split = fast_ok && AMODE == 0 && splits == 1 && mode != 0 && a.ep.colsum == nullptr &&
                       ((uintptr_t)a.C % 16 == 0) && (a.ldc % 4 == 0) && ((uintptr_t)a.ep.bias % 16 == 0);
    int split_tm = 0;                                        // m-tiles the split kernel covered (all of them, when it ran)
    if (a.ep.r2_row0 != nullptr && !(fm > 0 && fn > 0 && split && a.N % 128 == 0 && kv == a.K && a.ep.bias == nullptr)) {
        set_error("gemm: the rank-2 epilogue needs the split kernel over the whole output (npi_linear_bwd_data_rank2_supported)");
        return NPI_ERR_ARG;
    }
    if (a.ep.sc0 != nullptr && !(fm > 0 && fn > 0 && split && (a.N == 128 || a.N == 256) && kv == a.K && a.ep.r2_row0 == nullptr)) {
        set_error("gemm: the row-dot epilogue needs the split kernel with one column tile (npi_linear_fwd_scores_supported)");
        return NPI_ERR_ARG;
    }
    if (fm > 0 && fn > 0 && split) {
        // the three bf16 planes of B live in the caller's workspace (W is small: 3 * 2 * K * N bytes)
        uint16_t* planes = reinterpret_cast<uint16_t*>(scratch);
        const int64_t nel = (int64_t)a.N * a.K;
        if (planes == nullptr) {
            set_error("gemm: the split kernel needs the caller's workspace (npi_linear_workspace_bytes)");
            return NPI_ERR_WORKSPACE;
        }
        const bool f16 = a_scales != nullptr && kv == a.K;   // (zero-padded operands keep the bf16 x 3 planes)
        if (!(prepared && scratch != nullptr)) {
            if (f16) split_planes_f16_kernel<<<(unsigned)a.N, 256, 0, stream>>>(a.B, a.ldb, a.K, a.N, BMODE, planes, f16_inv_of(planes, a.K, a.N));
            else split_planes_kernel<<<(unsigned)ceil_div(nel, 256), 256, 0, stream>>>(a.B, a.ldb, a.K, a.N, BMODE, planes, kv);
        }
        const bool wide_n = (a.N % 256 == 0);                // 128 x 256 tiles: each A element is split once
        const int tn = wide_n ? a.N / 256 : fn;
        split_tm = (int)ceil_div(a.M, 128);                  // a ragged last m-tile overlaps its neighbour (TileWalk::row0)
        SplitArgs sa{a.A, a.lda, planes, a.C, a.ldc, a.M, a.N, a.K, a.ep, split_tm, tn, f16 ? a_scales : nullptr,
                     f16 ? f16_inv_of(planes, a.K, a.N) : nullptr};
        const int64_t ntiles = (int64_t)split_tm * tn;
        const int grid = (int)(ntiles < cu_slots ? ((ntiles + 7) / 8) * 8 : cu_slots);      // one workgroup per CU, multiple of 8 (XCDs)
        if (f16) {
            if (a.ep.r2_row0 != nullptr) {
                if (wide_n) gemm_split_ws_kernel<4, 1, true><<<grid, WS_THREADS, 0, stream>>>(sa);
                else        gemm_split_ws_kernel<2, 1, true><<<grid, WS_THREADS, 0, stream>>>(sa);
            } else if (a.ep.sc0 != nullptr) {
                if (wide_n) gemm_split_ws_kernel<4, 2, true><<<grid, WS_THREADS, 0, stream>>>(sa);
                else        gemm_split_ws_kernel<2, 2, true><<<grid, WS_THREADS, 0, stream>>>(sa);
            } else {
                if (wide_n) gemm_split_ws_kernel<4, 0, true><<<grid, WS_THREADS, 0, stream>>>(sa);
                else        gemm_split_ws_kernel<2, 0, true><<<grid, WS_THREADS, 0, stream>>>(sa);
            }
        } else if (a.ep.r2_row0 != nullptr) {
            if (wide_n) gemm_split_ws_kernel<4, 1><<<grid, WS_THREADS, 0, stream>>>(sa);
            else        gemm_split_ws_kernel<2, 1><<<grid, WS_THREADS, 0, stream>>>(sa);
        } else if (a.ep.sc0 != nullptr) {
            if (wide_n) gemm_split_ws_kernel<4, 2><<<grid, WS_THREADS, 0, stream>>>(sa);
            else        gemm_split_ws_kernel<2, 2><<<grid, WS_THREADS, 0, stream>>>(sa);
        } else {
            if (wide_n) gemm_split_ws_kernel<4><<<grid, WS_THREADS, 0, stream>>>(sa);
            else        gemm_split_ws_kernel<2><<<grid, WS_THREADS, 0, stream>>>(sa);
        }
    }
    // edge strips in 128 x 128 tiles
    const int tm = (int)ceil_div(a.M, 128), tn = (int)ceil_div(a.N, 128);
    // a ragged output the matrix-core-bf16 kernels did not take (the reference's 178-wide first layer): when ALL its tiles fit
    // the chip at once, ONE guarded launch covers it -- the unguarded kernel on the full tiles plus a launch per ragged strip were
    // three launches of 20-29 us each for a [1,992 x 178] output, each of them a single round of latency
    const bool one_guarded = split_tm == 0 && fm > 0 && fn > 0 && (a.M % bm != 0 || a.N % bn != 0) &&
                             (int64_t)tm * tn * splits <= 256;
    if (split_tm == 0 && fm > 0 && fn > 0 && !one_guarded) {
        GemmArgs f = a;
        f.tm0 = 0; f.tn0 = 0;
        if (bf16_in)   gemm_fast_kernel<AMODE, BMODE, 2, 2, bf16_t><<<dim3(fn, fm, splits), GEMM_THREADS, 0, stream>>>(f);
        else           gemm_fast_kernel<AMODE, BMODE, 2, 2><<<dim3(fn, fm, splits), GEMM_THREADS, 0, stream>>>(f);
    }
    const bool fast_ran = fm > 0 && fn > 0 && !one_guarded;
    const int em = split_tm > 0 ? split_tm : (fast_ran ? fm * bm / 128 : 0);       // first edge tile row
    const int en = fast_ran ? fn * bn / 128 : 0;                                    // first edge tile column
    auto edge = [&](int tm0, int tn0, int nm, int nn) {
        if (nm <= 0 || nn <= 0) return;
        GemmArgs e = a;
        e.tm0 = tm0; e.tn0 = tn0;
        if (kv != a.K) { e.K = kv; e.kchunk = (int)align_up(kv, BK); }          // B has kv rows only
        const dim3 g(nn, nm, splits);
        if (dtype_in == NPI_BF16 && dtype_out == NPI_BF16) {
            if (v4) gemm_edge_kernel<AMODE, BMODE, true, bf16_t, bf16_t><<<g, GEMM_THREADS, 0, stream>>>(e);
            else    gemm_edge_kernel<AMODE, BMODE, false, bf16_t, bf16_t><<<g, GEMM_THREADS, 0, stream>>>(e);
        } else if (dtype_in == NPI_BF16) {
            if (v4) gemm_edge_kernel<AMODE, BMODE, true, bf16_t, float><<<g, GEMM_THREADS, 0, stream>>>(e);
            else    gemm_edge_kernel<AMODE, BMODE, false, bf16_t, float><<<g, GEMM_THREADS, 0, stream>>>(e);
        } else {
            if (v4) gemm_edge_kernel<AMODE, BMODE, true><<<g, GEMM_THREADS, 0, stream>>>(e);
            else    gemm_edge_kernel<AMODE, BMODE, false><<<g, GEMM_THREADS, 0, stream>>>(e);
        }
    };
    if (em > 0 && en > 0) {
        edge(em, 0, tm - em, tn);        // bottom strip (all columns)
        edge(0, en, em, tn - en);        // right strip (full rows only)
    } else {
        edge(0, 0, tm, tn);
    }
    return NPI_OK;
}

}  // namespace npi

using namespace npi;

static inline const float* fp(const void* p) { return reinterpret_cast<const float*>(p); }
static inline const void* advance(const void* p, int64_t elems, int es) { return reinterpret_cast<const char*>(p) + elems * es; }


// per-call arithmetic of the *_ex entry points -> launch_gemm's mode
// (no process-wide default any more: 0 = the 3-way bf16 split wherever the shape takes it, NPI_GEMM_EXACT_F32 = the f32 MFMA kernels)
static int gemm_mode_of(int flags) { return (flags & NPI_GEMM_EXACT_F32) ? 0 : 1; }
static bool scratch_ok(void* ws, int64_t ws_bytes, int64_t K, int64_t N) {
    return ws != nullptr && ws_bytes >= npi_linear_workspace_bytes(K, N) && ((uintptr_t)ws % 16) == 0;
}

extern "C" int64_t npi_linear_workspace_bytes(int64_t K, int64_t N) {
    if (K <= 0 || N <= 0) return -1;
    return 6 * K * N + 256;                                   // three bf16 planes of the weight matrix
}

// Both re-laid copies of a weight matrix W [K, N] in ONE launch: set 0 = what npi_linear_fwd_ex prepares (B = W), set 1 = what
// npi_linear_bwd_data_ex prepares (B = W^T); blockIdx.y walks the requested sets.  f32: three bf16 planes each; bf16: one
// k-block-major copy each.
template <typename T>
__global__ void prepare_weight_kernel(const T* __restrict__ W, int64_t ldw, int K, int N, int first_set, char* __restrict__ ws,
                                      int64_t set_bytes) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * K) return;
    const int set = first_set + (int)blockIdx.y;                  // 0: contraction over K (rows of W), 1: over N (columns of W)
    const int Kg = set == 0 ? K : N, Ng = set == 0 ? N : K;
    const int n = (int)(i / Kg), k = (int)(i % Kg);
    const T x = set == 0 ? W[(int64_t)k * ldw + n] : W[(int64_t)n * ldw + k];
    uint16_t* __restrict__ out = reinterpret_cast<uint16_t*>(ws + (int64_t)blockIdx.y * set_bytes);
    const int64_t o = ((int64_t)(k / SK) * Ng + n) * SK + (k % SK);
    if constexpr (sizeof(T) == 4) {
        uint32_t p0, p1, p2;
        split3_pair(x, 0.f, p0, p1, p2);
        out[o] = (uint16_t)p0;
        out[(int64_t)N * K + o] = (uint16_t)p1;
        out[2 * (int64_t)N * K + o] = (uint16_t)p2;
    } else {
        out[o] = x;
    }
}

extern "C" int npi_linear_prepare(const void* W, int64_t ldw, int64_t K, int64_t N, int which, int dtype, void* workspace,
                                  int64_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(K > 0 && N > 0 && K < 0x7fffffff && N < 0x7fffffff, "npi_linear_prepare: bad size");
    const bool f16 = (which & NPI_PREPARE_F16X2) != 0;        // fp16 x 2 planes (+ column scales) for NPI_GEMM_SPLIT_F16X2 calls
    which &= ~NPI_PREPARE_F16X2;
    NPI_REQUIRE(which >= 1 && which <= 3, "npi_linear_prepare: which must be 1 (forward), 2 (backward) or 3 (both) [| NPI_PREPARE_F16X2]");
    NPI_REQUIRE(!f16 || dtype == NPI_F32, "npi_linear_prepare: NPI_PREPARE_F16X2 is for f32 weights");
    NPI_REQUIRE(dtype == NPI_F32 || dtype == NPI_BF16, "npi_linear_prepare: bad dtype");
    NPI_REQUIRE(W && workspace && ldw >= N, "npi_linear_prepare: null pointer or leading dimension too small");
    NPI_REQUIRE(K % SK == 0 && N % SK == 0, "npi_linear_prepare: K and N must be multiples of 16 (the matrix-core kernels' k-step)");
    const int64_t one = npi_linear_workspace_bytes(K, N);
    const int sets = which == 3 ? 2 : 1;
    if (workspace_bytes < sets * one || ((uintptr_t)workspace % 16) != 0) {
        set_error("npi_linear_prepare: workspace too small (npi_linear_workspace_bytes per set) or not 16-byte aligned");
        return NPI_ERR_WORKSPACE;
    }
    const dim3 grid((unsigned)ceil_div(K * N, 256), (unsigned)sets);
    const int first = which == 2 ? 1 : 0;
    if (f16) {
        for (int s_ = 0; s_ < sets; ++s_) {                  // set 0: B = W (contraction over K); set 1: B = W^T (contraction over N)
            const int set = first + s_;
            uint16_t* planes = reinterpret_cast<uint16_t*>((char*)workspace + (int64_t)s_ * one);
            const int Kg = set == 0 ? (int)K : (int)N, Ng = set == 0 ? (int)N : (int)K;
            split_planes_f16_kernel<<<(unsigned)Ng, 256, 0, stream>>>(fp(W), ldw, Kg, Ng, set, planes, f16_inv_of(planes, K, N));
        }
        return check_launch("npi_linear_prepare");
    }
    if (dtype == NPI_F32)
        prepare_weight_kernel<float><<<grid, 256, 0, stream>>>(fp(W), ldw, (int)K, (int)N, first, (char*)workspace, one);
    else
        prepare_weight_kernel<uint16_t><<<grid, 256, 0, stream>>>(reinterpret_cast<const uint16_t*>(W), ldw, (int)K, (int)N, first,
                                                                  (char*)workspace, one);
    return check_launch("npi_linear_prepare");
}

extern "C" int npi_row_scales(const float* A, int64_t lda, int64_t M, int64_t K, float* scales, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(M >= 0 && K > 0 && M < 0x7fffffff && K < 0x7fffffff && lda >= K, "npi_row_scales: bad size");
    if (M == 0) return NPI_OK;
    NPI_REQUIRE(A && scales, "npi_row_scales: null pointer");
    row_scale_kernel<<<(unsigned)ceil_div(M, 4), 256, 0, stream>>>(A, lda, (int)M, (int)K, scales);
    return check_launch("npi_row_scales");
}

extern "C" int npi_linear_fwd_ex(const void* A, int64_t lda, const void* W, int64_t ldw, const void* bias,
                                  const float* rowscale, void* C, int64_t ldc, int64_t M, int64_t K,
                                  int64_t N, int relu, int dtype, int flags, void* workspace, int64_t workspace_bytes,
                                  const float* a_scales, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const bool f16 = (flags & NPI_GEMM_SPLIT_F16X2) != 0;
    NPI_REQUIRE(!f16 || (a_scales != nullptr && dtype == NPI_F32 && !(flags & (NPI_GEMM_EXACT_F32 | NPI_GEMM_A_ZERO_PADDED))),
                "npi_linear_fwd_ex: NPI_GEMM_SPLIT_F16X2 needs a_scales (npi_row_scales), f32 storage, and excludes "
                "NPI_GEMM_EXACT_F32 / NPI_GEMM_A_ZERO_PADDED");
    NPI_REQUIRE(M >= 0 && K > 0 && N > 0, "npi_linear_fwd: bad size");
    NPI_REQUIRE(M < 0x7fffffff && K < 0x7fffffff && N < 0x7fffffff, "npi_linear_fwd: size > int32");
    NPI_REQUIRE(dtype == NPI_F32 || dtype == NPI_BF16, "npi_linear_fwd: bad dtype");
    if (M == 0) return NPI_OK;
    NPI_REQUIRE(A && W && C, "npi_linear_fwd: null pointer");
    NPI_REQUIRE(lda >= K && ldw >= N && ldc >= N, "npi_linear_fwd: leading dimension too small");
    // A stored with zero pad columns up to a multiple of 128 (a 178-wide aggregate kept 256 wide): the bf16 matrix-core
    // kernel runs on the padded width with the weight planes zero-extended instead of the guarded kernel on 178
    const bool padded = (flags & NPI_GEMM_A_ZERO_PADDED) != 0 && dtype == NPI_F32 && K % 128 != 0;
    const int64_t Kp = padded ? align_up(K, 128) : K;
    NPI_REQUIRE(lda >= Kp, "npi_linear_fwd: NPI_GEMM_A_ZERO_PADDED needs lda >= K rounded up to 128");
    if (!scratch_ok(workspace, workspace_bytes, Kp, N)) {
        set_error("npi_linear_fwd_ex: workspace too small or not 16-byte aligned");
        return NPI_ERR_WORKSPACE;
    }
    const int es = dtype == NPI_BF16 ? 2 : 4;
    GemmArgs a{fp(A), lda, fp(W), ldw, (float*)C, ldc, (int)M, (int)N, (int)Kp, (int)align_up(Kp, BK), 0, 0, 0,
               Epilogue{fp(bias), rowscale, relu, nullptr}};
    const bool prepared = (flags & NPI_GEMM_WORKSPACE_PREPARED) != 0;
    NPI_REQUIRE(!prepared || (workspace != nullptr && !padded), "npi_linear_fwd_ex: NPI_GEMM_WORKSPACE_PREPARED needs the workspace "
                "npi_linear_prepare filled and excludes NPI_GEMM_A_ZERO_PADDED");
    const int rc = launch_gemm<0, 0>(vec4_ok(A, lda, Kp, es) && vec4_ok(W, ldw, N, es), a, 1, stream, dtype, dtype,
                                     gemm_mode_of(flags), workspace, padded ? (int)K : 0, prepared, NPI_GEMM_RESERVED_CUS_OF(flags),
                                     f16 ? a_scales : nullptr);
    return rc != NPI_OK ? rc : check_launch("npi_linear_fwd");
}

// dA[M,K] = rowscale * (dC[M,N] @ W[K,N]^T): GEMM with "K" = N (contracted), output width K
extern "C" int npi_linear_bwd_data_ex(const void* dC, int64_t lddc, const void* W, int64_t ldw,
                                       const float* rowscale, void* dA, int64_t ldda, int64_t M, int64_t K,
                                       int64_t N, int dtype, int flags, void* workspace, int64_t workspace_bytes,
                                       const float* dc_scales, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const bool f16 = (flags & NPI_GEMM_SPLIT_F16X2) != 0;
    NPI_REQUIRE(!f16 || (dc_scales != nullptr && dtype == NPI_F32 && !(flags & NPI_GEMM_EXACT_F32)),
                "npi_linear_bwd_data_ex: NPI_GEMM_SPLIT_F16X2 needs dc_scales (npi_row_scales of dC), f32 storage, and excludes "
                "NPI_GEMM_EXACT_F32");
    NPI_REQUIRE(M >= 0 && K > 0 && N > 0, "npi_linear_bwd_data: bad size");
    NPI_REQUIRE(M < 0x7fffffff && K < 0x7fffffff && N < 0x7fffffff, "npi_linear_bwd_data: size > int32");
    NPI_REQUIRE(dtype == NPI_F32 || dtype == NPI_BF16, "npi_linear_bwd_data: bad dtype");
    if (M == 0) return NPI_OK;
    NPI_REQUIRE(dC && W && dA, "npi_linear_bwd_data: null pointer");
    NPI_REQUIRE(lddc >= N && ldw >= N && ldda >= K, "npi_linear_bwd_data: leading dimension too small");
    if (!scratch_ok(workspace, workspace_bytes, K, N)) {
        set_error("npi_linear_bwd_data_ex: workspace too small or not 16-byte aligned");
        return NPI_ERR_WORKSPACE;
    }
    const int es = dtype == NPI_BF16 ? 2 : 4;
    // B(k = n_contract, n = k_out) = W[k_out * ldw + n_contract]  -> BMODE 1
    GemmArgs a{fp(dC), lddc, fp(W), ldw, (float*)dA, ldda, (int)M, (int)K, (int)N, (int)align_up(N, BK), 0, 0, 0,
               Epilogue{nullptr, rowscale, 0, nullptr}};
    const bool prepared = (flags & NPI_GEMM_WORKSPACE_PREPARED) != 0;
    NPI_REQUIRE(!prepared || workspace != nullptr, "npi_linear_bwd_data_ex: NPI_GEMM_WORKSPACE_PREPARED needs the workspace "
                "npi_linear_prepare filled");
    const int rc = launch_gemm<0, 1>(vec4_ok(dC, lddc, N, es) && vec4_ok(W, ldw, N, es), a, 1, stream, dtype, dtype,
                                     gemm_mode_of(flags), workspace, 0, prepared, NPI_GEMM_RESERVED_CUS_OF(flags),
                                     f16 ? dc_scales : nullptr);
    return rc != NPI_OK ? rc : check_launch("npi_linear_bwd_data");
}
// C = A W and, from the accumulators on their way out, sc0[m] = <C[m, :], att[:N]>, sc1[m] = <C[m, :], att[N:]>: GATConv's
// h = x W with the two attention scores of every node in the GEMM's store epilogue (one head) instead of a pass over h
extern "C" int npi_linear_fwd_scores_supported(int64_t M, int64_t K, int64_t N) {
    // K >= 64: the row-dot epilogue double-buffers its LDS exchange by tile parity, which holds only when a tile has at least
    // NST = 4 k-steps of 16 (with 2 k-steps wave wn = 1 could be a whole tile ahead of wave wn = 0 and overwrite its slot)
    return (M >= 128 && M < 0x7fffffff && K >= 2 * BK && K % BK == 0 && (N == 128 || N == 256)) ? 1 : 0;
}
// a_scales != NULL: the fp16 x 2 arithmetic (NPI_GEMM_SPLIT_F16X2) with the row scales of A (npi_row_scales -- once, for a feature
// matrix that does not change between steps -- or the launch that wrote A: npi_gat_aggregate_fused)
extern "C" int npi_linear_fwd_scores(const float* A, int64_t lda, const float* W, int64_t ldw, const float* att, float* C,
                                         int64_t ldc, float* sc0, float* sc1, int64_t M, int64_t K, int64_t N, void* workspace,
                                         int64_t workspace_bytes, const float* a_scales, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(npi_linear_fwd_scores_supported(M, K, N), "npi_linear_fwd_scores: shape outside the split kernel's one-column-"
                "tile coverage (M >= 128, K >= 64, K % 32 == 0, N = 128 or 256)");
    NPI_REQUIRE(A && W && att && C && sc0 && sc1, "npi_linear_fwd_scores: null pointer");
    NPI_REQUIRE(lda >= K && ldw >= N && ldc >= N, "npi_linear_fwd_scores: leading dimension too small");
    NPI_REQUIRE(vec4_ok(A, lda, K, 4) && vec4_ok(W, ldw, N, 4) && ((uintptr_t)C % 16 == 0) && (ldc % 4 == 0),
                "npi_linear_fwd_scores: operands must be 16-byte aligned with leading dimensions % 4 == 0");
    if (!scratch_ok(workspace, workspace_bytes, K, N)) {
        set_error("npi_linear_fwd_scores: workspace too small or not 16-byte aligned");
        return NPI_ERR_WORKSPACE;
    }
    GemmArgs a{A, lda, W, ldw, C, ldc, (int)M, (int)N, (int)K, (int)align_up(K, BK), 0, 0, 0,
               Epilogue{nullptr, nullptr, 0, nullptr, nullptr, nullptr, att, att + N, sc0, sc1}};
    const int rc = launch_gemm<0, 0>(true, a, 1, stream, NPI_F32, NPI_F32, 1, workspace, 0, false, 0, a_scales);
    return rc != NPI_OK ? rc : check_launch("npi_linear_fwd_scores");
}
// dA = dC W^T + row0 (x) col0 + row1 (x) col1, the rank-2 term added in the split kernel's store epilogue (GATConv backward:
// the attention terms g_dst (x) W att_dst + g_src (x) W att_src of dX, without a read-modify-write pass over d hfeat)
extern "C" int npi_linear_bwd_data_rank2_supported(int64_t M, int64_t K, int64_t N) {
    return (M >= 128 && M < 0x7fffffff && K >= 128 && K % 128 == 0 && N >= BK && N % BK == 0 &&
            N % 4 == 0) ? 1 : 0;
}
// dc_scales != NULL: the fp16 x 2 arithmetic (NPI_GEMM_SPLIT_F16X2), the row scales of dC from npi_row_scales or from the launch that
// wrote dC (npi_gat_backward_fused_heads)
extern "C" int npi_linear_bwd_data_rank2(const float* dC, int64_t lddc, const float* W, int64_t ldw, const float* row0,
                                             const float* row1, const float* col0, const float* col1, float* dA, int64_t ldda,
                                             int64_t M, int64_t K, int64_t N, void* workspace, int64_t workspace_bytes,
                                             const float* dc_scales, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(npi_linear_bwd_data_rank2_supported(M, K, N), "npi_linear_bwd_data_rank2: shape outside the split kernel's "
                "full coverage (M >= 128, K % 128 == 0, N % 32 == 0)");
    NPI_REQUIRE(dC && W && dA && row0 && row1 && col0 && col1, "npi_linear_bwd_data_rank2: null pointer");
    NPI_REQUIRE(lddc >= N && ldw >= N && ldda >= K, "npi_linear_bwd_data_rank2: leading dimension too small");
    NPI_REQUIRE(vec4_ok(dC, lddc, N, 4) && vec4_ok(W, ldw, N, 4) && ((uintptr_t)dA % 16 == 0) && (ldda % 4 == 0),
                "npi_linear_bwd_data_rank2: operands must be 16-byte aligned with leading dimensions % 4 == 0");
    if (!scratch_ok(workspace, workspace_bytes, K, N)) {
        set_error("npi_linear_bwd_data_rank2: workspace too small or not 16-byte aligned");
        return NPI_ERR_WORKSPACE;
    }
    GemmArgs a{dC, lddc, W, ldw, dA, ldda, (int)M, (int)K, (int)N, (int)align_up(N, BK), 0, 0, 0,
               Epilogue{nullptr, nullptr, 0, nullptr, row0, row1, col0, col1}};
    const int rc = launch_gemm<0, 1>(true, a, 1, stream, NPI_F32, NPI_F32, 1, workspace, 0, false, 0, dc_scales);
    return rc != NPI_OK ? rc : check_launch("npi_linear_bwd_data_rank2");
}

extern "C" int64_t npi_colsum_workspace_elems(int64_t M, int64_t N) {
    if (M < 0 || N <= 0) return -1;
    return ceil_div(M > 0 ? M : 1, (int64_t)colsum_rows(M)) * N;
}

extern "C" int npi_colsum(const float* X, int64_t ldx, int64_t M, int64_t N, float* out, float* workspace,
                          int64_t workspace_elems, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(M >= 0 && N > 0 && M < 0x7fffffff && N < 0x7fffffff, "npi_colsum: bad size");
    NPI_REQUIRE(X && out && workspace && ldx >= N, "npi_colsum: bad argument");
    // the finest chunking the caller's workspace allows, down to colsum_rows(M) (npi_colsum_workspace_elems asks for that)
    int rows = colsum_rows(M);
    if (workspace_elems / N < ceil_div(M > 0 ? M : 1, rows)) rows = COLSUM_ROWS;
    const int nchunks = (int)ceil_div(M > 0 ? M : 1, rows);
    if (workspace_elems < (int64_t)nchunks * N) {
        set_error("npi_colsum: workspace too small");
        return NPI_ERR_WORKSPACE;
    }
    dim3 cg((unsigned)ceil_div(N, 256), (unsigned)nchunks);
    if (vec4_ok(X, ldx, N)) colsum_partial_kernel<true><<<cg, 256, 0, stream>>>(X, ldx, (int)M, (int)N, rows, workspace);
    else                    colsum_partial_kernel<false><<<cg, 256, 0, stream>>>(X, ldx, (int)M, (int)N, rows, workspace);
    slab_reduce_kernel<float><<<(unsigned)ceil_div(N, 256), 256, 0, stream>>>(workspace, N, nchunks, 1, (int)N, N, out, N);
    return check_launch("npi_colsum");
}

// dW: the contraction runs over the nodes.  Node count is arbitrary, so the part that is a
// multiple of BK goes through `splits` slabs (fast path) and the < BK remainder through one
// extra slab (guarded).
static void bwd_weight_plan(int64_t M, int64_t K, int64_t N, bool shared, int& splits, int& kchunk, int64_t& m_main) {
    const int64_t tiles = ceil_div(K, 128) * ceil_div(N, 128);
    m_main = (M / BK) * BK;
    splits = pick_splits(m_main > 0 ? m_main : 1, tiles, shared);
    kchunk = (int)(ceil_div(ceil_div(m_main > 0 ? m_main : 1, splits), BK) * BK);
}

// Plan of the split-bf16 dW kernel (gemm_dw_split_kernel): one workgroup per CU when it has the GPU to itself, about
// three per four CUs when it shares them with the backward aggregation; `per` nodes per slab (multiple of 16).
static bool dw_split_shape_ok(int64_t M, int64_t K, int64_t N) { return K % 128 == 0 && N % 128 == 0 && M >= 4096; }
static void dw_split_plan(int64_t m_main, int64_t K, int64_t N, bool shared, int& nslab, int64_t& per, int& tiles_m, int& tiles_n,
                          bool& wide) {
    wide = (N % 256 == 0);
    tiles_m = (int)(K / 128);
    tiles_n = (int)(wide ? N / 256 : N / 128);
    // shared regime: the 512-thread workgroup holds 2 x 208 VGPRs per SIMD and leaves the aggregation one wave slot there, so
    // it is kept to about 3 of 8 CUs (step at C4 by workgroup count: 64: 7.86, 96: 7.11, 128: 7.15, 192: 7.25 ms)
    const int64_t wgs = shared ? dw_workgroups(true) / 2 : 256;
    int64_t ns = wgs / ((int64_t)tiles_m * tiles_n);
    if (ns < 1) ns = 1;
    const int64_t maxs = ceil_div(m_main, (int64_t)SK * 16);                 // at least 16 k-steps per slab
    if (ns > maxs) ns = maxs;
    if (ns > 256) ns = 256;
    nslab = (int)ns;
    per = ceil_div(ceil_div(m_main, ns), (int64_t)SK) * SK;
}

extern "C" int64_t npi_linear_bwd_weight_workspace_elems(int64_t M, int64_t K, int64_t N) {
    if (M < 0 || K <= 0 || N <= 0) return -1;
    int splits, kchunk;
    int64_t m_main;
    bwd_weight_plan(M, K, N, /*shared=*/false, splits, kchunk, m_main);          // the regime with more slabs: enough for both
    int64_t slabs = splits + 1, dbs = splits + 1;
    if (dw_split_shape_ok(M, K, N)) {                                             // the split-bf16 plan: <= 256 slabs, 2 db rows each
        if (slabs < 257) slabs = 257;
        if (dbs < 2 * 256 + 1) dbs = 2 * 256 + 1;
    }
    return slabs * K * N + dbs * N + 64;                                          // dW slabs, then db slabs
}

// dW[K,N] = A[M,K]^T @ dC[M,N] (contract over M), db[N] = colsum(dC); A, dC, dW, db stored as `dtype`
extern "C" int64_t npi_col_scales_workspace_elems(int64_t M, int64_t K) {
    if (M < 0 || K <= 0) return -1;
    const int64_t a = ceil_div(M > 0 ? M : 1, (int64_t)colsum_rows(M)) * K;
    return a > 256 ? a : 256;
}

extern "C" int npi_col_scales(const float* A, int64_t lda, int64_t M, int64_t K, const float* row_scales, float* scales,
                              float* workspace, int64_t workspace_elems, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(M >= 0 && K > 0 && M < 0x7fffffff && K < 0x7fffffff, "npi_col_scales: bad size");
    NPI_REQUIRE(scales && workspace && (A != nullptr || row_scales != nullptr), "npi_col_scales: null pointer (A or row_scales)");
    if (workspace_elems < npi_col_scales_workspace_elems(M, K)) {
        set_error("npi_col_scales: workspace too small (npi_col_scales_workspace_elems)");
        return NPI_ERR_WORKSPACE;
    }
    if (A == nullptr) {                  // the matrix's global scale from its row scales, for every column
        const int nparts = (int)(M < 256 * 256 ? ceil_div(M > 0 ? M : 1, 256) : 256);
        minscale_partial_kernel<<<nparts, 256, 0, stream>>>(row_scales, M, workspace);
        minscale_finish_kernel<<<1, 256, 0, stream>>>(workspace, nparts, (int)K, scales);
        return check_launch("npi_col_scales");
    }
    NPI_REQUIRE(lda >= K, "npi_col_scales: leading dimension too small");
    const int rows = colsum_rows(M);
    const int nchunks = (int)ceil_div(M > 0 ? M : 1, rows);
    dim3 cg((unsigned)ceil_div(K, 256), (unsigned)nchunks);
    colmax_partial_kernel<<<cg, 256, 0, stream>>>(A, lda, (int)M, (int)K, rows, workspace);
    colmax_finish_kernel<<<(unsigned)ceil_div(K, 256), 256, 0, stream>>>(workspace, nchunks, (int)K, scales);
    return check_launch("npi_col_scales");
}

extern "C" int npi_linear_bwd_weight_ex(const void* A, int64_t lda, const void* dC, int64_t lddc,
                                        void* dW, int64_t lddw, void* db, int64_t M, int64_t K, int64_t N,
                                        float* workspace, int64_t workspace_elems, int dtype, int flags, int shared,
                                        const float* a_col_scales, const float* dc_col_scales, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const bool f16 = (flags & NPI_GEMM_SPLIT_F16X2) != 0;
    NPI_REQUIRE(!f16 || (a_col_scales && dc_col_scales && dtype == NPI_F32 && !(flags & (NPI_GEMM_EXACT_F32 | NPI_GEMM_A_ZERO_PADDED)) &&
                         K % 128 == 0 && N % 128 == 0 && M >= 4096 && ((uintptr_t)a_col_scales % 16) == 0 && ((uintptr_t)dc_col_scales % 16) == 0),
                "npi_linear_bwd_weight_ex: NPI_GEMM_SPLIT_F16X2 needs both column-scale vectors (npi_col_scales; 16-byte aligned), f32 "
                "storage, K and N multiples of 128, M >= 4096, and excludes NPI_GEMM_EXACT_F32 / NPI_GEMM_A_ZERO_PADDED");
    NPI_REQUIRE(M >= 0 && K > 0 && N > 0, "npi_linear_bwd_weight: bad size");
    NPI_REQUIRE(M < 0x7fffffff && K < 0x7fffffff && N < 0x7fffffff, "npi_linear_bwd_weight: size > int32");
    NPI_REQUIRE(dtype == NPI_F32 || dtype == NPI_BF16, "npi_linear_bwd_weight: bad dtype");
    NPI_REQUIRE(A && dC && dW && workspace, "npi_linear_bwd_weight: null pointer");
    NPI_REQUIRE(lda >= K && lddc >= N && lddw >= N, "npi_linear_bwd_weight: leading dimension too small");
    if (workspace_elems < npi_linear_bwd_weight_workspace_elems(M, K, N)) {
        set_error("npi_linear_bwd_weight: workspace too small");
        return NPI_ERR_WORKSPACE;
    }
    const int es = dtype == NPI_BF16 ? 2 : 4;
    // A with zero pad columns up to a multiple of 128 (see npi_linear_fwd_ex): the split kernel runs on the padded width --
    // the pad rows of its slabs are zero and never read -- instead of the guarded kernel on K
    const bool padded = (flags & NPI_GEMM_A_ZERO_PADDED) != 0 && dtype == NPI_F32 && K % 128 != 0;
    const int64_t Kp = padded ? align_up(K, 128) : K;
    NPI_REQUIRE(lda >= Kp, "npi_linear_bwd_weight: NPI_GEMM_A_ZERO_PADDED needs lda >= K rounded up to 128");
    if (padded && workspace_elems < npi_linear_bwd_weight_workspace_elems(M, Kp, N)) {
        set_error("npi_linear_bwd_weight: workspace too small (query it with K rounded up to 128 for a zero-padded A)");
        return NPI_ERR_WORKSPACE;
    }
    const bool v4 = vec4_ok(A, lda, Kp, es) && vec4_ok(dC, lddc, N, es);
    const unsigned gw = (unsigned)ceil_div(K * N, 256), gb = (unsigned)ceil_div(N, 256);
    const unsigned gwb = dw_finish_grid(K, N, db != nullptr);
    // ---- f32 storage on the bf16 matrix cores: both operands split on the fly (gemm_dw_split_kernel) ----
    if (dtype == NPI_F32 && gemm_mode_of(flags) != 0 && v4 && dw_split_shape_ok(M, Kp, N)) {
        const int64_t m16 = (M / SK) * SK;
        int nslab, tm, tn;
        int64_t per;
        bool wide;
        dw_split_plan(m16, Kp, N, shared != 0, nslab, per, tm, tn, wide);
        float* db_slabs = workspace + (int64_t)(nslab + 1) * Kp * N;
        DwArgs d{fp(A), lda, fp(dC), lddc, workspace, db ? db_slabs : nullptr, (int)Kp, (int)N, m16, per, tm, tn, nslab,
                 f16 ? a_col_scales : nullptr, f16 ? dc_col_scales : nullptr};
        const unsigned grid = (unsigned)(ceil_div(nslab, 8) * 8 * tm * tn);     // slots of 8 slabs (one per XCD) x tiles
        if (f16) {
            if (wide) gemm_dw_split_kernel<4, false, true><<<grid, WS_THREADS, 0, stream>>>(d);
            else      gemm_dw_split_kernel<2, false, true><<<grid, WS_THREADS, 0, stream>>>(d);
        } else if (wide) gemm_dw_split_kernel<4><<<grid, WS_THREADS, 0, stream>>>(d);
        else             gemm_dw_split_kernel<2><<<grid, WS_THREADS, 0, stream>>>(d);
        // slabs in slab order, the < 16 trailing nodes and db in one launch
        dw_finish_kernel<float><<<gwb, 256, 0, stream>>>(workspace, Kp * N, nslab, (int)K, (int)N, (float*)dW, lddw, db_slabs, 2 * nslab,
                                                  (float*)db, fp(advance(A, m16 * lda, es)), lda, fp(advance(dC, m16 * lddc, es)), lddc,
                                                  (int)(M - m16));
        return check_launch("npi_linear_bwd_weight");
    }
    // ---- bf16 storage: the same kernel without the split (one plane, one MFMA per product tile); slabs and the finish in f32 ----
    if (dtype == NPI_BF16 && gemm_mode_of(flags) != 0 && v4 && dw_split_shape_ok(M, K, N)) {
        const int64_t m16 = (M / SK) * SK;
        int nslab, tm, tn;
        int64_t per;
        bool wide;
        dw_split_plan(m16, K, N, shared != 0, nslab, per, tm, tn, wide);
        float* db_slabs = workspace + (int64_t)(nslab + 1) * K * N;
        DwArgs d{fp(A), lda, fp(dC), lddc, workspace, db ? db_slabs : nullptr, (int)K, (int)N, m16, per, tm, tn, nslab, nullptr, nullptr};
        const unsigned grid = (unsigned)(ceil_div(nslab, 8) * 8 * tm * tn);
        if (wide) gemm_dw_split_kernel<4, true><<<grid, WS_THREADS, 0, stream>>>(d);
        else      gemm_dw_split_kernel<2, true><<<grid, WS_THREADS, 0, stream>>>(d);
        dw_finish_kernel<bf16_t><<<gwb, 256, 0, stream>>>(workspace, K * N, nslab, (int)K, (int)N, (bf16_t*)dW, lddw, db_slabs, 2 * nslab,
                                                         (bf16_t*)db, reinterpret_cast<const bf16_t*>(advance(A, m16 * lda, es)), lda,
                                                         reinterpret_cast<const bf16_t*>(advance(dC, m16 * lddc, es)), lddc, (int)(M - m16));
        return check_launch("npi_linear_bwd_weight");
    }
    int splits, kchunk;
    int64_t m_main;
    bwd_weight_plan(M, K, N, shared != 0, splits, kchunk, m_main);
    const bool has_rem = M > m_main || m_main == 0;
    float* db_slabs = workspace + (int64_t)(splits + 1) * K * N;
    // output rows = K (features of A), cols = N; A(m = feature, k = node) = A[node*lda + feature]
    // main part: nodes [0, m_main) in `splits` f32 slabs
    if (m_main > 0) {
        GemmArgs a{fp(A), lda, fp(dC), lddc, workspace, N, (int)K, (int)N, (int)m_main, kchunk, 0, 0, K * N,
                   Epilogue{nullptr, nullptr, 0, db ? db_slabs : nullptr}};
        (void)launch_gemm<1, 0>(v4, a, splits, stream, dtype, NPI_F32);          // AMODE 1 never takes the allocating path
    }
    if (dtype == NPI_F32) {
        // slabs in slab order, the < 32 trailing nodes and db in one launch (a node count that is a multiple of 32 used to
        // pay a zero-filling launch here, a remainder its own guarded GEMM)
        const int ns = m_main > 0 ? splits : 0;
        dw_finish_kernel<float><<<gwb, 256, 0, stream>>>(workspace, K * N, ns, (int)K, (int)N, (float*)dW, lddw, db_slabs, ns, (float*)db,
                                                  fp(advance(A, m_main * lda, es)), lda, fp(advance(dC, m_main * lddc, es)), lddc,
                                                  (int)(M - m_main));
        return check_launch("npi_linear_bwd_weight");
    }
    if (M - m_main < 32 && m_main > 0) {
        // bf16 storage, the usual case: the < 32 trailing nodes, the slab sums, db and the rounding to bf16 in the one finishing
        // launch (rounds 1-3: a guarded GEMM for the trailing nodes and two reductions -- 4 launches per dW against 2 for f32)
        dw_finish_kernel<bf16_t><<<gwb, 256, 0, stream>>>(workspace, K * N, splits, (int)K, (int)N, (bf16_t*)dW, lddw, db_slabs, splits,
                                                         (bf16_t*)db, reinterpret_cast<const bf16_t*>(advance(A, m_main * lda, es)), lda,
                                                         reinterpret_cast<const bf16_t*>(advance(dC, m_main * lddc, es)), lddc,
                                                         (int)(M - m_main));
        return check_launch("npi_linear_bwd_weight");
    }
    // bf16 storage: remainder nodes [m_main, M) into slab `splits`, then the two reductions
    const int nslab = (m_main > 0 ? splits : 0) + (has_rem ? 1 : 0);
    if (has_rem) {
        GemmArgs r{fp(advance(A, m_main * lda, es)), lda, fp(advance(dC, m_main * lddc, es)), lddc,
                   workspace + (int64_t)(m_main > 0 ? splits : 0) * K * N, N, (int)K, (int)N, (int)(M - m_main), BK, 0, 0, K * N,
                   Epilogue{nullptr, nullptr, 0, db ? db_slabs + (int64_t)(m_main > 0 ? splits : 0) * N : nullptr}};
        (void)launch_gemm<1, 0>(false, r, 1, stream, dtype, NPI_F32);
    }
    slab_reduce_kernel<bf16_t><<<gw, 256, 0, stream>>>(workspace, K * N, nslab, (int)K, (int)N, N, (bf16_t*)dW, lddw);
    if (db) slab_reduce_kernel<bf16_t><<<gb, 256, 0, stream>>>(db_slabs, N, nslab, 1, (int)N, N, (bf16_t*)db, N);
    return check_launch("npi_linear_bwd_weight");
}
