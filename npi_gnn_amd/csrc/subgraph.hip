// One-hop enclosing-subgraph extraction + PyG `Batch` collate on the device.
// Replaces the per-sample Python dict / set loops of `local_subgraph_generation`
// (reference src/classes.py:652-733) and the DataLoader collate that feed Net_1.
//
// The interaction graph is a CSR over node serial numbers: ptr[N+1], nbr[nnz] = partners in
// interaction_list order (RNA rows hold proteins, protein rows hold RNAs; npi_csr_build's stable sort
// keeps that order), ok[nnz] = 1 when the pair may be used (not a test key of the fold).
// A sample is a target pair (l, p).  Its local nodes: 0 = l, 1 = p, then the usable partners of l in
// list order (p itself skipped), then the usable partners of p (l skipped); its undirected pairs: the
// target, (0, partner of l) ..., (partner of p, 1) ...; every pair is emitted in both directions,
// (rna, protein) first, like the reference.  Pairs must be unique and the graph bipartite (checked by the
// host wrapper): then no partner can have been numbered before and one ballot-ranked pass suffices.
// All work is integer / byte movement, one wavefront per sample (hub proteins have 10^3..10^5 partners).
#include "npi_common.h"

namespace npi {

__device__ __forceinline__ int lanes_below(uint64_t mask, int lane) {
    return __popcll(mask & ((1ull << lane) - 1ull));
}

// cnt[g] = usable partners of l other than p, cnt[B + g] = usable partners of p other than l
__global__ void __launch_bounds__(256)
subgraph_count_kernel(const int32_t* __restrict__ ptr, const int32_t* __restrict__ nbr, const uint8_t* __restrict__ ok,
                      const int32_t* __restrict__ keys, int B, int32_t* __restrict__ cnt) {
    const int lane = lane_id();
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= B) return;
    const int l = keys[2 * g], p = keys[2 * g + 1];
    int c[2];
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const int row = side ? p : l, other = side ? l : p;
        const int b = ptr[row], e = ptr[row + 1];
        int n = 0;
        for (int k = b; k < e; k += WAVE) {
            const int i = k + lane;
            const bool take = i < e && ok[i] != 0 && nbr[i] != other;
            n += __popcll(__ballot(take));
        }
        c[side] = n;
    }
    if (lane == 0) {
        cnt[g] = c[0];
        cnt[B + g] = c[1];
    }
}

// node_off[g] = sum_{h<g} (2 + cl_h + cp_h), pair_off[g] = sum_{h<g} (1 + cl_h + cp_h); one workgroup
__global__ void __launch_bounds__(1024)
subgraph_scan_kernel(const int32_t* __restrict__ cnt, int B, int32_t* __restrict__ node_off, int32_t* __restrict__ pair_off) {
    __shared__ long long part[1024];
    const int t = threadIdx.x;
    const int per = (B + 1023) / 1024;
    const int b = min(B, t * per), e = min(B, b + per);
    long long s = 0;
    for (int g = b; g < e; ++g) s += (long long)cnt[g] + cnt[B + g];
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {                   // Hillis-Steele inclusive scan of the thread sums
        const long long v = (t >= d) ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    long long run = (t > 0) ? part[t - 1] : 0;
    for (int g = b; g < e; ++g) {
        node_off[g] = (int)(run + 2 * g);
        pair_off[g] = (int)(run + g);
        run += (long long)cnt[g] + cnt[B + g];
    }
    if (t == 1023) {
        const long long tot = part[1023] + 2ll * B;        // nodes >= pairs: one check covers both
        node_off[B] = tot > 0x7fffffffll ? -1 : (int)tot;  // -1: the batch does not fit 32-bit row ids
        pair_off[B] = tot > 0x7fffffffll ? -1 : (int)(part[1023] + B);
    }
}

__global__ void __launch_bounds__(256)
subgraph_fill_kernel(const int32_t* __restrict__ ptr, const int32_t* __restrict__ nbr, const uint8_t* __restrict__ ok,
                     const int32_t* __restrict__ keys, int B, const int32_t* __restrict__ node_off,
                     const int32_t* __restrict__ pair_off, int32_t* __restrict__ node_id, int64_t* __restrict__ batch,
                     int64_t* __restrict__ esrc, int64_t* __restrict__ edst, int n_nodes, int n_pairs,
                     int32_t* __restrict__ status) {
    const int lane = lane_id();
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    // the arrays were sized from the CALLER's totals; the offsets below are the device's.  Totals that belong to other keys
    // would make every store below an out-of-bounds write: bit 2 of the status word is raised instead, and the arrays are
    // filled -- within the caller's sizes -- with values every consumer is safe on until the status is read (which may be many
    // launches later and never inside a capture): graph 0 / node 0 for every row, the (-1, -1) padding column for every edge
    if (node_off[B] != n_nodes || pair_off[B] != n_pairs) {
        if (g == 0 && lane == 0 && status != nullptr) atomicOr(status, 4);
        const int64_t nt = (int64_t)gridDim.x * blockDim.x;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_nodes; i += nt) { node_id[i] = 0; batch[i] = 0; }
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * (int64_t)n_pairs; i += nt) { esrc[i] = -1; edst[i] = -1; }
        return;
    }
    if (g >= B) return;
    const int l = keys[2 * g], p = keys[2 * g + 1];
    const int64_t n0 = node_off[g];
    const int64_t e0 = 2 * (int64_t)pair_off[g];
    if (lane == 0) {
        node_id[n0] = l;
        node_id[n0 + 1] = p;
        batch[n0] = g;
        batch[n0 + 1] = g;
        esrc[e0] = n0;     edst[e0] = n0 + 1;             // (rna, protein), then the reverse
        esrc[e0 + 1] = n0 + 1; edst[e0 + 1] = n0;
    }
    int c = 0;                                             // partners placed so far (local id 2 + c, pair 1 + c)
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const int row = side ? p : l, other = side ? l : p;
        const int b = ptr[row], e = ptr[row + 1];
        for (int k = b; k < e; k += WAVE) {
            const int i = k + lane;
            const int q = i < e ? nbr[i] : -1;
            const bool take = i < e && ok[i] != 0 && q != other;
            const uint64_t m = __ballot(take);
            if (take) {
                const int pos = c + lanes_below(m, lane);
                const int64_t nn = n0 + 2 + pos;           // the partner's row in the batch
                const int64_t ee = e0 + 2 * (int64_t)(1 + pos);
                node_id[nn] = q;
                batch[nn] = g;
                const int64_t rna = side ? nn : n0, prot = side ? n0 + 1 : nn;
                esrc[ee] = rna;      edst[ee] = prot;
                esrc[ee + 1] = prot; edst[ee + 1] = rna;
            }
            c += __popcll(m);
        }
    }
}

// x[row] = [structural label | feat[node_id[row]]]; label 0 for the two target nodes of a sample; columns Ff + 1 .. ldx - 1
// (a row pitch padded for the GEMMs, NPI_GEMM_A_ZERO_PADDED) are set to zero
__global__ void __launch_bounds__(256)
subgraph_features_kernel(const float* __restrict__ feat, int64_t ldf, int Ff, const int32_t* __restrict__ node_id,
                         const int64_t* __restrict__ batch, const int32_t* __restrict__ node_off, int B, int64_t n,
                         float* __restrict__ x, int64_t ldx) {
    const int lane = lane_id();
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    if (node_off[B] != n) {                                // npi_subgraph_fill raised the status bit: node_id is a sentinel -- zero rows
        for (int c = lane; c < ldx; c += WAVE) x[row * ldx + c] = 0.f;
        return;
    }
    const float* __restrict__ src = feat + (int64_t)node_id[row] * ldf;
    float* __restrict__ dst = x + row * ldx;
    if (lane == 0) dst[0] = (row - node_off[batch[row]] < 2) ? 0.f : 1.f;
    for (int c = lane; 1 + c < ldx; c += WAVE) dst[1 + c] = c < Ff ? src[c] : 0.f;
}

}  // namespace npi

using namespace npi;

extern "C" int npi_subgraph_sizes(const int32_t* ptr, const int32_t* nbr, const uint8_t* ok, const int32_t* keys,
                                  int64_t B, int32_t* node_off, int32_t* pair_off, int32_t* workspace, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(B >= 0 && B < 0x3fffffff, "npi_subgraph_sizes: bad size");
    NPI_REQUIRE(node_off && pair_off, "npi_subgraph_sizes: null pointer");
    if (B > 0) {
        NPI_REQUIRE(ptr && nbr && ok && keys && workspace, "npi_subgraph_sizes: null pointer");
        subgraph_count_kernel<<<(unsigned)ceil_div(B, 4), 256, 0, stream>>>(ptr, nbr, ok, keys, (int)B, workspace);
    }
    subgraph_scan_kernel<<<1, 1024, 0, stream>>>(workspace, (int)B, node_off, pair_off);
    return check_launch("npi_subgraph_sizes");
}

extern "C" int npi_subgraph_fill(const int32_t* ptr, const int32_t* nbr, const uint8_t* ok, const int32_t* keys,
                                 int64_t B, const int32_t* node_off, const int32_t* pair_off, int32_t* node_id,
                                 int64_t* batch, int64_t* edge_src, int64_t* edge_dst, int64_t n_nodes, int64_t n_pairs,
                                 int32_t* status, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(B >= 0 && B < 0x3fffffff, "npi_subgraph_fill: bad size");
    NPI_REQUIRE(n_nodes >= 0 && n_pairs >= 0 && n_nodes <= 0x7fffffff && n_pairs <= 0x7fffffff, "npi_subgraph_fill: bad totals");
    if (status != nullptr) (void)hipMemsetAsync(status, 0, sizeof(int32_t), stream);
    if (B == 0) return NPI_OK;
    NPI_REQUIRE(ptr && nbr && ok && keys && node_off && pair_off && node_id && batch && edge_src && edge_dst,
                "npi_subgraph_fill: null pointer");
    subgraph_fill_kernel<<<(unsigned)ceil_div(B, 4), 256, 0, stream>>>(ptr, nbr, ok, keys, (int)B, node_off, pair_off, node_id,
                                                                       batch, edge_src, edge_dst, (int)n_nodes, (int)n_pairs, status);
    return check_launch("npi_subgraph_fill");
}

extern "C" int npi_subgraph_features(const float* feat, int64_t ldf, int64_t Ff, const int32_t* node_id,
                                     const int64_t* batch, const int32_t* node_off, int64_t B, int64_t n, float* x, int64_t ldx,
                                     void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(n >= 0 && B >= 0 && B < 0x3fffffff && Ff > 0 && ldf >= Ff && ldx >= Ff + 1, "npi_subgraph_features: bad size");
    if (n == 0) return NPI_OK;
    NPI_REQUIRE(feat && node_id && batch && node_off && x, "npi_subgraph_features: null pointer");
    subgraph_features_kernel<<<(unsigned)ceil_div(n, 4), 256, 0, stream>>>(feat, ldf, (int)Ff, node_id, batch, node_off, (int)B, n, x, ldx);
    return check_launch("npi_subgraph_features");
}
