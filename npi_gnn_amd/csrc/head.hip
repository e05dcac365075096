// The readout sum and the MLP head of Net_1 as three launches (reference src/classes.py:74-80):
//     x = x1 + x2 + x3;  x = relu(lin1(x));  x = dropout(x, 0.5);  x = relu(lin2(x));  x = lin3(x);  log_softmax(x, -1)
// The batch has 200 rows and the layers are 256 -> 128 -> 64 -> 2: as library GEMMs, bias / ReLU / dropout / softmax
// element-wise kernels and their autograd the head was 30 launches and 0.3 ms of a 1.6 ms training step, every one of them
// launch-latency bound (hipBLASLt picks 256-row tiles for a 200-row product: 36 us).  Here: one forward kernel (a
// workgroup per 8 rows: the rows sit in LDS, a thread owns an output column and streams its weight row), one row-parallel
// backward kernel (dz3, dz2, dz1 and the input gradient, which is the gradient of all three readouts) and one
// column-parallel kernel for the weight and bias gradients (a block per weight row, the 200 rows summed in row order:
// deterministic, no atomics).  f32 throughout; weights in torch.nn.Linear layout [out, in].
#include "npi_common.h"

namespace npi {

constexpr int HEAD_R = 8;          // rows per workgroup
constexpr int HEAD_T = 256;        // threads per workgroup
constexpr int HEAD_D0 = 1024;      // widest input the LDS row buffer takes
constexpr int HEAD_DH = 256;       // widest hidden layer (one thread per column)
constexpr int HEAD_D3 = 32;        // most classes

struct HeadArgs {
    const float* r[3]; int64_t ldr[3]; int nr;      // readouts, summed on the way in
    int B, D0, D1, D2, D3;
    const float *W1, *b1, *W2, *b2, *W3, *b3;      // [D1, D0], [D1], [D2, D1], [D2], [D3, D2], [D3]
    const float* mask;                               // [B, D1] of 0 / 1, or null (evaluation)
    float scale;                                     // 1 / (1 - p)
    float *s, *h1, *h2, *logp;                       // saved for the backward: [B, D0], [B, D1] (ReLU output, before dropout), [B, D2]; out [B, D3]
    int act;                                         // NPI_HEAD_LOG_SOFTMAX (Net_1) or NPI_HEAD_SIGMOID (Net_1_onlyOneOutput): what follows lin3
};

// out[r][j] = sum_k in[r][k] W[j][k] for the HEAD_R rows in LDS; thread -> column j = t % Dout, row group t / Dout
template <int MAXR>
__device__ __forceinline__ void head_layer(const float* __restrict__ in, int ldi, int Din, const float* __restrict__ W, int Dout,
                                           float (&acc)[MAXR], int& j, int& g, int& G) {
    const int t = threadIdx.x;
    G = HEAD_T / Dout;
    if (G > HEAD_R) G = HEAD_R;
    if (G < 1) G = 1;
    j = t % Dout;
    g = t / Dout;
#pragma unroll
    for (int q = 0; q < MAXR; ++q) acc[q] = 0.f;
    if (g >= G) return;
    const float* __restrict__ w = W + (int64_t)j * Din;
    // eight 16-byte pieces of the weight row in flight (one at a time the loop is a chain of L2 round trips: 43 us for the
    // 200 x 256 x 128 layer)
    for (int k0 = 0; k0 < Din; k0 += 32) {
        float4 wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            wv[u] = (k0 + 4 * u < Din) ? *reinterpret_cast<const float4*>(w + k0 + 4 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + 4 * u;
            if (k < Din) {
#pragma unroll
                for (int q = 0; q < MAXR; ++q) {
                    const int r = g + q * G;
                    if (r < HEAD_R) {
                        const float4 xv = *reinterpret_cast<const float4*>(in + r * ldi + k);     // one address per wave: broadcast
                        acc[q] = fmaf(wv[u].x, xv.x, fmaf(wv[u].y, xv.y, fmaf(wv[u].z, xv.z, fmaf(wv[u].w, xv.w, acc[q]))));
                    }
                }
            }
        }
    }
}

__global__ void __launch_bounds__(HEAD_T)
head_fwd_kernel(HeadArgs a) {
    __shared__ __attribute__((aligned(16))) float xs[HEAD_R * HEAD_D0];
    __shared__ __attribute__((aligned(16))) float h1s[HEAD_R * HEAD_DH];
    __shared__ __attribute__((aligned(16))) float h2s[HEAD_R * HEAD_DH];
    __shared__ float z3[HEAD_R * HEAD_D3];
    const int t = threadIdx.x;
    const int row0 = blockIdx.x * HEAD_R;
    for (int idx = t; idx < HEAD_R * a.D0; idx += HEAD_T) {
        const int r = idx / a.D0, k = idx - r * a.D0;
        const int row = row0 + r;
        float v = 0.f;
        if (row < a.B) {
            v = a.r[0][(int64_t)row * a.ldr[0] + k];
            if (a.nr > 1) v += a.r[1][(int64_t)row * a.ldr[1] + k];          // (x1 + x2) + x3, as the reference adds them
            if (a.nr > 2) v += a.r[2][(int64_t)row * a.ldr[2] + k];
            if (a.s) a.s[(int64_t)row * a.D0 + k] = v;
        }
        xs[r * a.D0 + k] = v;
    }
    __syncthreads();
    float acc[HEAD_R];
    int j, g, G;
    head_layer<HEAD_R>(xs, a.D0, a.D0, a.W1, a.D1, acc, j, g, G);
    if (g < G) {
        const float b = a.b1[j];
#pragma unroll
        for (int q = 0; q < HEAD_R; ++q) {
            const int r = g + q * G, row = row0 + r;
            if (r < HEAD_R) {
                float h = acc[q] + b;
                h = h < 0.f ? 0.f : h;
                float hd = h;
                if (row < a.B) {
                    if (a.h1) a.h1[(int64_t)row * a.D1 + j] = h;
                    if (a.mask) hd = h * a.mask[(int64_t)row * a.D1 + j] * a.scale;
                }
                h1s[r * a.D1 + j] = hd;
            }
        }
    }
    __syncthreads();
    head_layer<HEAD_R>(h1s, a.D1, a.D1, a.W2, a.D2, acc, j, g, G);
    if (g < G) {
        const float b = a.b2[j];
#pragma unroll
        for (int q = 0; q < HEAD_R; ++q) {
            const int r = g + q * G, row = row0 + r;
            if (r < HEAD_R) {
                float h = acc[q] + b;
                h = h < 0.f ? 0.f : h;
                if (row < a.B && a.h2) a.h2[(int64_t)row * a.D2 + j] = h;
                h2s[r * a.D2 + j] = h;
            }
        }
    }
    __syncthreads();
    head_layer<HEAD_R>(h2s, a.D2, a.D2, a.W3, a.D3, acc, j, g, G);
    if (g < G) {
        const float b = a.b3[j];
#pragma unroll
        for (int q = 0; q < HEAD_R; ++q) {
            const int r = g + q * G;
            if (r < HEAD_R) z3[r * a.D3 + j] = acc[q] + b;
        }
    }
    __syncthreads();
    if (t < HEAD_R && row0 + t < a.B && a.act == NPI_HEAD_SIGMOID) {      // torch.sigmoid(lin3(x)): the one-output variant
        // (reference src/train_with_twoDataset_modelOnlyOneOutput.py:80-81)
        for (int c = 0; c < a.D3; ++c) a.logp[(int64_t)(row0 + t) * a.D3 + c] = 1.f / (1.f + expf(-z3[t * a.D3 + c]));
    } else if (t < HEAD_R && row0 + t < a.B) {             // log_softmax of one row (D3 = 2 in the reference)
        const float* z = z3 + t * a.D3;
        float m = z[0];
        for (int c = 1; c < a.D3; ++c) m = fmaxf(m, z[c]);
        float se = 0.f;
        for (int c = 0; c < a.D3; ++c) se += expf(z[c] - m);
        const float lse = m + logf(se);
        for (int c = 0; c < a.D3; ++c) a.logp[(int64_t)(row0 + t) * a.D3 + c] = z[c] - lse;
    }
}

struct HeadBwdArgs {
    int B, D0, D1, D2, D3;
    const float *W1, *W2, *W3;
    const float* mask; float scale;
    const float *s, *h1, *h2, *logp, *dlogp;
    float *dz1, *dz2, *dz3;          // [B, D1], [B, D2], [B, D3]
    float* ds;                       // [B, D0] or null
    float *dW1, *db1, *dW2, *db2, *dW3, *db3;
    int act;                         // as HeadArgs.act (`logp` then holds the sigmoid's output)
};

// rows: dz3 = dlogp - softmax * sum(dlogp);  dz2 = (dz3 W3) [h2 > 0];  dz1 = (dz2 W2) mask scale [h1 > 0];  ds = dz1 W1
__global__ void __launch_bounds__(HEAD_T)
head_bwd_rows_kernel(HeadBwdArgs a) {
    __shared__ float z3s[HEAD_R * HEAD_D3];
    __shared__ float z2s[HEAD_R * HEAD_DH];
    __shared__ float z1s[HEAD_R * HEAD_DH];
    const int t = threadIdx.x;
    const int row0 = blockIdx.x * HEAD_R;
    if (t < HEAD_R) {
        const int row = row0 + t;
        float sum = 0.f;
        if (row < a.B) for (int c = 0; c < a.D3; ++c) sum += a.dlogp[(int64_t)row * a.D3 + c];
        for (int c = 0; c < a.D3; ++c) {
            float v = 0.f;
            if (row < a.B) {
                const float o = a.logp[(int64_t)row * a.D3 + c], d = a.dlogp[(int64_t)row * a.D3 + c];
                v = a.act == NPI_HEAD_SIGMOID ? d * o * (1.f - o) : d - expf(o) * sum;      // sigmoid' = y (1 - y)
                a.dz3[(int64_t)row * a.D3 + c] = v;
            }
            z3s[t * a.D3 + c] = v;
        }
    }
    __syncthreads();
    for (int idx = t; idx < HEAD_R * a.D2; idx += HEAD_T) {          // consecutive threads: consecutive columns of W3 rows
        const int r = idx / a.D2, j = idx - r * a.D2, row = row0 + r;
        float v = 0.f;
        if (row < a.B) {
            for (int c = 0; c < a.D3; ++c) v = fmaf(z3s[r * a.D3 + c], a.W3[(int64_t)c * a.D2 + j], v);
            v = a.h2[(int64_t)row * a.D2 + j] > 0.f ? v : 0.f;
            a.dz2[(int64_t)row * a.D2 + j] = v;
        }
        z2s[r * a.D2 + j] = v;
    }
    __syncthreads();
    for (int idx = t; idx < HEAD_R * a.D1; idx += HEAD_T) {
        const int r = idx / a.D1, j = idx - r * a.D1, row = row0 + r;
        float v = 0.f;
        if (row < a.B) {
#pragma unroll 8
            for (int m = 0; m < a.D2; ++m) v = fmaf(z2s[r * a.D2 + m], a.W2[(int64_t)m * a.D1 + j], v);
            if (a.mask) v *= a.mask[(int64_t)row * a.D1 + j] * a.scale;
            v = a.h1[(int64_t)row * a.D1 + j] > 0.f ? v : 0.f;
            a.dz1[(int64_t)row * a.D1 + j] = v;
        }
        z1s[r * a.D1 + j] = v;
    }
    if (a.ds == nullptr) return;
    __syncthreads();
    for (int k = t; k < a.D0; k += HEAD_T) {
        float acc[HEAD_R];
#pragma unroll
        for (int r = 0; r < HEAD_R; ++r) acc[r] = 0.f;
#pragma unroll 8
        for (int j = 0; j < a.D1; ++j) {
            const float w = a.W1[(int64_t)j * a.D0 + k];
#pragma unroll
            for (int r = 0; r < HEAD_R; ++r) acc[r] = fmaf(z1s[r * a.D1 + j], w, acc[r]);
        }
#pragma unroll
        for (int r = 0; r < HEAD_R; ++r)
            if (row0 + r < a.B) a.ds[(int64_t)(row0 + r) * a.D0 + k] = acc[r];
    }
}

// weight and bias gradients, out[i][j] = sum_r L[r][i] R[r][j] over the B rows: a block takes one i and 32 consecutive j, its
// 8 x 32 threads sum every 8th row and the 8 partials are folded in a fixed order (deterministic, no atomics; one thread
// per element walking all B rows was a chain of 200 round trips: 98 us).  Block ranges: dW1 (D1 x D0 / 32 blocks), dW2, dW3,
// then the three biases (L = 1).
constexpr int HW_J = 32, HW_G = HEAD_T / HW_J;
__device__ __forceinline__ void head_outer(const float* __restrict__ L, int ldl, int i, const float* __restrict__ R, int ldr,
                                           const float* __restrict__ Rmask, float scale, int j0, int Dj, int B,
                                           float* __restrict__ out) {
    __shared__ float part[HW_G][HW_J];
    const int t = threadIdx.x, jl = t % HW_J, rg = t / HW_J, j = j0 + jl;
    float s = 0.f;
    if (j < Dj) {
#pragma unroll 4
        for (int r = rg; r < B; r += HW_G) {
            float v = R[(int64_t)r * ldr + j];
            if (Rmask) v *= Rmask[(int64_t)r * ldr + j] * scale;
            s = fmaf(L ? L[(int64_t)r * ldl + i] : 1.f, v, s);
        }
    }
    part[rg][jl] = s;
    __syncthreads();
    if (rg == 0 && j < Dj) {
        float o = 0.f;
#pragma unroll
        for (int q = 0; q < HW_G; ++q) o += part[q][jl];
        out[j] = o;
    }
}
__host__ __device__ inline int head_jblocks(int D) { return (D + HW_J - 1) / HW_J; }
__global__ void __launch_bounds__(HEAD_T)
head_bwd_weights_kernel(HeadBwdArgs a) {
    int b = blockIdx.x;
    const int n1 = a.D1 * head_jblocks(a.D0), n2 = a.D2 * head_jblocks(a.D1), n3 = a.D3 * head_jblocks(a.D2);
    if (b < n1) {
        const int i = b / head_jblocks(a.D0), jb = b % head_jblocks(a.D0);
        head_outer(a.dz1, a.D1, i, a.s, a.D0, nullptr, 1.f, jb * HW_J, a.D0, a.B, a.dW1 + (int64_t)i * a.D0);
        return;
    }
    b -= n1;
    if (b < n2) {
        const int i = b / head_jblocks(a.D1), jb = b % head_jblocks(a.D1);
        head_outer(a.dz2, a.D2, i, a.h1, a.D1, a.mask, a.scale, jb * HW_J, a.D1, a.B, a.dW2 + (int64_t)i * a.D1);
        return;
    }
    b -= n2;
    if (b < n3) {
        const int i = b / head_jblocks(a.D2), jb = b % head_jblocks(a.D2);
        head_outer(a.dz3, a.D3, i, a.h2, a.D2, nullptr, 1.f, jb * HW_J, a.D2, a.B, a.dW3 + (int64_t)i * a.D2);
        return;
    }
    b -= n3;
    if (b < head_jblocks(a.D1)) { head_outer(nullptr, 0, 0, a.dz1, a.D1, nullptr, 1.f, b * HW_J, a.D1, a.B, a.db1); return; }
    b -= head_jblocks(a.D1);
    if (b < head_jblocks(a.D2)) { head_outer(nullptr, 0, 0, a.dz2, a.D2, nullptr, 1.f, b * HW_J, a.D2, a.B, a.db2); return; }
    b -= head_jblocks(a.D2);
    head_outer(nullptr, 0, 0, a.dz3, a.D3, nullptr, 1.f, b * HW_J, a.D3, a.B, a.db3);
}

static bool head_dims_ok(int64_t B, int64_t D0, int64_t D1, int64_t D2, int64_t D3) {
    return B >= 0 && B < 0x7fffffff && D0 > 0 && D0 <= HEAD_D0 && D0 % 4 == 0 && D1 > 0 && D1 <= HEAD_DH && D1 % 4 == 0 &&
           D2 > 0 && D2 <= HEAD_DH && D2 % 4 == 0 && D3 > 0 && D3 <= HEAD_D3;
}

}  // namespace npi

using namespace npi;

extern "C" int npi_mlp_head_fwd(const float* r1, int64_t ld1, const float* r2, int64_t ld2, const float* r3, int64_t ld3,
                                int64_t B, int64_t D0, const float* W1, const float* b1, int64_t D1, const float* W2,
                                const float* b2, int64_t D2, const float* W3, const float* b3, int64_t D3, const float* mask,
                                float scale, int activation, float* s, float* h1, float* h2, float* logp, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(head_dims_ok(B, D0, D1, D2, D3), "npi_mlp_head_fwd: bad size (D0 <= 1024, D1, D2 <= 256, D3 <= 32, widths multiples of 4)");
    if (B == 0) return NPI_OK;
    NPI_REQUIRE(activation == NPI_HEAD_LOG_SOFTMAX || activation == NPI_HEAD_SIGMOID, "npi_mlp_head_fwd: unknown activation");
    NPI_REQUIRE(r1 && W1 && b1 && W2 && b2 && W3 && b3 && logp, "npi_mlp_head_fwd: null pointer");
    NPI_REQUIRE(ld1 >= D0 && (!r2 || ld2 >= D0) && (!r3 || ld3 >= D0) && (r2 || !r3), "npi_mlp_head_fwd: bad readout arguments");
    NPI_REQUIRE(((uintptr_t)W1 | (uintptr_t)W2 | (uintptr_t)W3) % 16 == 0, "npi_mlp_head_fwd: weights must be 16-byte aligned");
    HeadArgs a{{r1, r2, r3}, {ld1, ld2, ld3}, r3 ? 3 : (r2 ? 2 : 1), (int)B, (int)D0, (int)D1, (int)D2, (int)D3,
               W1, b1, W2, b2, W3, b3, mask, scale, s, h1, h2, logp, activation};
    head_fwd_kernel<<<(unsigned)ceil_div(B, HEAD_R), HEAD_T, 0, stream>>>(a);
    return check_launch("npi_mlp_head_fwd");
}

extern "C" int64_t npi_mlp_head_workspace_elems(int64_t B, int64_t D1, int64_t D2, int64_t D3) {
    return B < 0 ? -1 : B * (D1 + D2 + D3) + 16;
}

extern "C" int npi_mlp_head_bwd(int64_t B, int64_t D0, int64_t D1, int64_t D2, int64_t D3, const float* W1, const float* W2,
                                const float* W3, const float* mask, float scale, int activation, const float* s, const float* h1,
                                const float* h2, const float* logp, const float* dlogp, float* ds, float* dW1, float* db1,
                                float* dW2, float* db2, float* dW3, float* db3, float* workspace, int64_t workspace_elems,
                                void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NPI_REQUIRE(head_dims_ok(B, D0, D1, D2, D3), "npi_mlp_head_bwd: bad size");
    NPI_REQUIRE(activation == NPI_HEAD_LOG_SOFTMAX || activation == NPI_HEAD_SIGMOID, "npi_mlp_head_bwd: unknown activation");
    NPI_REQUIRE(W1 && W2 && W3 && s && h1 && h2 && logp && dlogp && dW1 && db1 && dW2 && db2 && dW3 && db3 && workspace,
                "npi_mlp_head_bwd: null pointer");
    if (workspace_elems < npi_mlp_head_workspace_elems(B, D1, D2, D3)) {
        set_error("npi_mlp_head_bwd: workspace too small");
        return NPI_ERR_WORKSPACE;
    }
    HeadBwdArgs a{(int)B, (int)D0, (int)D1, (int)D2, (int)D3, W1, W2, W3, mask, scale, s, h1, h2, logp, dlogp,
                  workspace, workspace + B * D1, workspace + B * (D1 + D2), ds, dW1, db1, dW2, db2, dW3, db3, activation};
    if (B > 0) head_bwd_rows_kernel<<<(unsigned)ceil_div(B, HEAD_R), HEAD_T, 0, stream>>>(a);
    const unsigned nb = (unsigned)(D1 * head_jblocks((int)D0) + D2 * head_jblocks((int)D1) + D3 * head_jblocks((int)D2) +
                                   head_jblocks((int)D1) + head_jblocks((int)D2) + head_jblocks((int)D3));
    head_bwd_weights_kernel<<<nb, HEAD_T, 0, stream>>>(a);
    return check_launch("npi_mlp_head_bwd");
}
