// AM-softmax head + cross-entropy, forward and backward, for gfx950.
//
// Replaces the reference's scripts/loss.py:37-52 (AMSoftmax.forward: row / column L2 normalisation, x_norm @ w_norm, a one-hot
// built ON THE HOST and copied back (loss.py:46-48), margin, annealing blend, scale) followed by train.py:219
// (nn.CrossEntropyLoss): six eager ops, a [B,N] one-hot, a separate log-softmax and a host round trip per step become
//   forward   amsoftmax_fwd_kernel       cos(theta) = <x_b, W_n> / (|x_b| |W_n|) as ONE tiled fp32 product over the raw operands,
//                                        the norms accumulated from the same tiles; margin by a label compare; logits; per
//                                        (row, column tile) partial (max, sum e^logit) of the cross-entropy
//             amsoftmax_ce_finish_kernel log-sum-exp per row from the partials (fixed order), loss = mean_b (lse_b - logit_b,label)
//   backward  amsoftmax_dcos_kernel      d loss / d cos = s (softmax - onehot) g / B on the fly from cos, lse, label (fused-loss
//                                        form) — or the caller's upstream gradients of (cos, logits) (module form)
//             amsoftmax_bwd_w_kernel     dW = (x_hat^T dcos - W_hat diag(t)) / |W_n|,  t_n = sum_b dcos_bn cos_bn
//             amsoftmax_bwd_x_kernel(+reduce)  dx = (dcos W_hat^T - x_hat diag(u)) / |x_b|,  u_b = sum_n dcos_bn cos_bn
// (the projection terms of the two normalisations are t and u: <W_hat_n, dW_hat_n> = sum_b dcos_bn cos_bn, no second pass).
// Sizes are tiny next to the front-end (B=64, E=400, N=5994: 0.3 GFLOP per product, W = 9.6 MB), so the products run on the
// fp32 vector units — exact fp32 like the reference, no operand rounding — in 64 x 64 tiles through LDS; everything is
// deterministic (fixed-order reductions, no float atomics).
// d logits / d cos = s whatever the margin and the annealing factor: logits = s (cos - m onehot / (1 + alpha)).
#include "dasv_common.hpp"
#include "../../include/dasv_hip.h"

namespace dasv {

typedef float af4 __attribute__((ext_vector_type(4)));

constexpr int AT = 64;    // tile edge (rows and columns per workgroup)
constexpr int AK = 64;    // reduction chunk of the two backward products (16 in round 3: four times the exposed round trips)
constexpr float kAmEps = 1e-12f;   // loss.py:40,42: norms clamped at 1e-12

// One 64 x 64 tile of C = A (64 x K, row r at a + r * lda, k contiguous) * B (K x 64, row k at b + k * ldb, columns contiguous),
// rows < ra and columns < cb valid, K any.  256 threads as 16 x 16, thread (ty, tx) owns rows ty*4.. and columns tx*4...
// asq / bsq (optional): sums of squares of A's rows / B's columns, accumulated from the staged tiles (every thread gets the sums
// of ITS four rows / columns).
// Chunks of FK = 16 with the loads TWO chunks ahead in registers: written as load - barrier - FMAs - barrier the 25 chunks of E = 400
// were 25 exposed memory round trips (82 us for a 0.3 GFLOP product on 94 workgroups); one chunk ahead 44 us; 64-wide chunks 65 us.
constexpr int FK = 16;
__device__ __forceinline__ void tile_product(const float* __restrict__ a, long long lda, int ra, const float* __restrict__ b,
                                             long long ldb, int cb, int K, float (&acc)[4][4], float* asq, float* bsq,
                                             float (*As)[FK + 1], float (*Bs)[AT + 4]) {
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    const int ar = tid >> 2, akk = (tid & 3) * 4;          // A chunk: 64 rows x 16 k, four consecutive k of row ar per thread
    const int bc = tid & 63, bk0 = tid >> 6;               // B chunk: 16 k x 64 columns, column bc of rows bk0, bk0 + 4, ...
    float pa0[4], pb0[4], pa1[4], pb1[4];
    float sa[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};     // sums of squares, in registers (copied out below)
    auto fetch = [&](float (&qa)[4], float (&qb)[4], int k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) qa[q] = (ar < ra && k0 + akk + q < K) ? a[(long long)ar * lda + k0 + akk + q] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) qb[q] = (k0 + q * 4 + bk0 < K && bc < cb) ? b[(long long)(k0 + q * 4 + bk0) * ldb + bc] : 0.f;
    };
    auto chunk = [&](float (&qa)[4], float (&qb)[4], int k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) As[ar][akk + q] = qa[q];
#pragma unroll
        for (int q = 0; q < 4; ++q) Bs[q * 4 + bk0][bc] = qb[q];
        __syncthreads();
        if (k0 + 2 * FK < K) fetch(qa, qb, k0 + 2 * FK);     // this register set is free again: two chunks ahead
#pragma unroll
        for (int kk = 0; kk < FK; ++kk) {
            float av[4], bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = As[ty * 4 + i][kk];
            const af4 b4 = *reinterpret_cast<const af4*>(&Bs[kk][tx * 4]);
            bv[0] = b4[0]; bv[1] = b4[1]; bv[2] = b4[2]; bv[3] = b4[3];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
#pragma unroll
            for (int i = 0; i < 4; ++i) sa[i] = fmaf(av[i], av[i], sa[i]);
#pragma unroll
            for (int j = 0; j < 4; ++j) sb[j] = fmaf(bv[j], bv[j], sb[j]);
        }
        __syncthreads();
    };
    fetch(pa0, pb0, 0);
    if (FK < K) fetch(pa1, pb1, FK);
    for (int k0 = 0; k0 < K; k0 += 2 * FK) {
        chunk(pa0, pb0, k0);
        if (k0 + FK < K) chunk(pa1, pb1, k0 + FK);
    }
    if (asq != nullptr) { asq[0] += sa[0]; asq[1] += sa[1]; asq[2] += sa[2]; asq[3] += sa[3]; }
    if (bsq != nullptr) { bsq[0] += sb[0]; bsq[1] += sb[1]; bsq[2] += sb[2]; bsq[3] += sb[3]; }
}

struct AmParams {
    const float* x;        // [B,E]
    const float* W;        // [E,N]
    const long long* label;  // [B]
    float* costh;          // [B,N]
    float* logits;         // [B,N] or null
    float* xinv;           // [B]   1 / max(|x_b|, eps)
    float* winv;           // [N]   1 / max(|W_n|, eps)
    float* part;           // [B][ntile][2] (max, sum e^(logit - max)) of the logits of each column tile, or null
    float* lab_logit;      // [B] logit in the label column, or null
    int B, E, N, ntile;
    float m_eff, s;        // margin / (1 + alpha), scale
};

__global__ void __launch_bounds__(256) amsoftmax_fwd_kernel(const AmParams p) {
    __shared__ float As[AT][FK + 1];
    __shared__ __attribute__((aligned(16))) float Bs[FK][AT + 4];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int n0 = blockIdx.x * AT, b0 = blockIdx.y * AT;
    float acc[4][4], asq[4] = {0.f, 0.f, 0.f, 0.f}, bsq[4] = {0.f, 0.f, 0.f, 0.f};
    tile_product(p.x + (long long)b0 * p.E, p.E, p.B - b0, p.W + n0, p.N, p.N - n0, p.E, acc, asq, bsq, As, Bs);
    float xi[4], wi[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) xi[i] = 1.0f / fmaxf(sqrtf(asq[i]), kAmEps);
#pragma unroll
    for (int j = 0; j < 4; ++j) wi[j] = 1.0f / fmaxf(sqrtf(bsq[j]), kAmEps);
    if (blockIdx.x == 0 && tx == 0)
#pragma unroll
        for (int i = 0; i < 4; ++i) if (b0 + ty * 4 + i < p.B) p.xinv[b0 + ty * 4 + i] = xi[i];
    if (blockIdx.y == 0 && ty == 0)
#pragma unroll
        for (int j = 0; j < 4; ++j) if (n0 + tx * 4 + j < p.N) p.winv[n0 + tx * 4 + j] = wi[j];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int b = b0 + ty * 4 + i;
        const long long lab = b < p.B ? p.label[b] : -1;
        float lg[4], mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            const float c = acc[i][j] * xi[i] * wi[j];
            lg[j] = p.s * (c - ((long long)n == lab ? p.m_eff : 0.f));
            if (b < p.B && n < p.N) {
                p.costh[(long long)b * p.N + n] = c;
                if (p.logits != nullptr) p.logits[(long long)b * p.N + n] = lg[j];
                if ((long long)n == lab && p.lab_logit != nullptr) p.lab_logit[b] = lg[j];
                mx = fmaxf(mx, lg[j]);
            } else {
                lg[j] = -INFINITY;
            }
        }
        if (p.part != nullptr) {
            // (max, sum) over the tile's 64 columns of row b: the 16 threads tx = 0..15 of this ty are 16 consecutive lanes
            mx = group_max<16>(mx);
            float se = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) se += (lg[j] == -INFINITY) ? 0.f : __expf(lg[j] - mx);
            se = group_sum<16>(se);
            if (tx == 0 && b < p.B) {
                p.part[((long long)b * p.ntile + blockIdx.x) * 2] = mx;
                p.part[((long long)b * p.ntile + blockIdx.x) * 2 + 1] = se;
            }
        }
    }
}

// lse_b from the column-tile partials (fixed order), loss = mean_b (lse_b - logit_b,label)
// A label outside [0, N) has no column: nothing wrote lab_logit[b].  The reference raises there (scatter_ at loss.py:47,
// CrossEntropyLoss at train.py:219); a launch cannot raise, so the loss becomes NaN — loudly wrong instead of a finite value made
// of uninitialised workspace.
__global__ void __launch_bounds__(256) amsoftmax_ce_finish_kernel(const float* __restrict__ part, const float* __restrict__ lab_logit,
                                                                  const long long* __restrict__ label, int N,
                                                                  float* __restrict__ lse, float* __restrict__ loss, int B, int ntile) {
    __shared__ float red[256];
    float local = 0.f;
    // four lanes per row share its column tiles (one thread per row walked the ~94 tiles twice, load after load: 28 us of a step)
    const int sub = threadIdx.x & 3;
    for (int b0 = 0; b0 < B; b0 += 64) {
        const int b = b0 + (threadIdx.x >> 2);
        const bool live = b < B;
        float mx = -INFINITY;
        if (live)
            for (int t = sub; t < ntile; t += 4) mx = fmaxf(mx, part[((long long)b * ntile + t) * 2]);
        mx = group_max<4>(mx);
        float se = 0.f;
        if (live)
            for (int t = sub; t < ntile; t += 4) se += part[((long long)b * ntile + t) * 2 + 1] * __expf(part[((long long)b * ntile + t) * 2] - mx);
        se = group_sum<4>(se);
        if (live && sub == 0) {
            const float l = mx + __logf(se);
            lse[b] = l;
            const long long lab = label[b];
            local += l - ((lab >= 0 && lab < (long long)N) ? lab_logit[b] : __builtin_nanf(""));
        }
    }
    red[threadIdx.x] = local;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = red[0] / (float)B;
}

// dcos[b][n] = s * gscale * (e^(logit - lse_b) - [n == label_b])  (+ nothing else: fused-loss form); gscale = g_loss / B.
// Module form (lse == null): dcos = g_costh + s * g_logits (either may be null).
__global__ void __launch_bounds__(256) amsoftmax_dcos_kernel(const float* __restrict__ costh, const long long* __restrict__ label,
                                                             const float* __restrict__ lse, const float* __restrict__ g_loss,
                                                             const float* __restrict__ g_costh, const float* __restrict__ g_logits,
                                                             float* __restrict__ dcos, int B, int N, float m_eff, float s) {
    const long long n_el = (long long)B * N;
    const float gs = (lse != nullptr) ? (g_loss != nullptr ? g_loss[0] : 1.0f) / (float)B : 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_el; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / N), n = (int)(i - (long long)b * N);
        float v;
        if (lse != nullptr) {
            const bool hit = (long long)n == label[b];
            const float lg = s * (costh[i] - (hit ? m_eff : 0.f));
            v = s * gs * (__expf(lg - lse[b]) - (hit ? 1.f : 0.f));
        } else {
            v = (g_costh != nullptr ? g_costh[i] : 0.f) + (g_logits != nullptr ? s * g_logits[i] : 0.f);
        }
        dcos[i] = v;
    }
}

// u[b] = sum_n dcos[b][n] * costh[b][n]   (one workgroup per row; fixed order)
__global__ void __launch_bounds__(256) amsoftmax_rowdot_kernel(const float* __restrict__ dcos, const float* __restrict__ costh,
                                                               float* __restrict__ u, int N) {
    __shared__ float red[4];
    const long long base = (long long)blockIdx.x * N;
    float v = 0.f;
    for (int n = threadIdx.x; n < N; n += 256) v = fmaf(dcos[base + n], costh[base + n], v);
    v = group_sum<64>(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) u[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// gW[e][n] = winv[n] * (sum_b x_hat[b][e] dcos[b][n] - W_hat[e][n] t[n]),  t[n] = sum_b dcos[b][n] cos[b][n].
// grid (N tiles, E tiles): A = x^T scaled by xinv (64 e x K=B), B = dcos (K=B x 64 n).
__global__ void __launch_bounds__(256) amsoftmax_bwd_w_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                              const float* __restrict__ dcos, const float* __restrict__ costh,
                                                              const float* __restrict__ xinv, const float* __restrict__ winv,
                                                              float* __restrict__ gW, int B, int E, int N) {
    __shared__ float As[AT][AK + 1];
    __shared__ __attribute__((aligned(16))) float Bs[AK][AT + 4];
    __shared__ __attribute__((aligned(16))) float Cs[AK][AT + 4];      // the cos chunk beside the dcos chunk (t[n] = sum_b dcos cos)
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int n0 = blockIdx.x * AT, e0 = blockIdx.y * AT;
    float acc[4][4], t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int k0 = 0; k0 < B; k0 += AK) {
        {   // A chunk (64 e x 16 b) = x[b][e] * xinv[b]: a wave reads 64 consecutive e of one b
#pragma unroll
            for (int q = 0; q < AK / 4; ++q) {
                const int kk = q * 4 + (tid >> 6), e = tid & 63;
                As[e][kk] = (k0 + kk < B && e0 + e < E) ? x[(long long)(k0 + kk) * E + e0 + e] * xinv[k0 + kk] : 0.f;
            }
        }
#pragma unroll
        for (int q = 0; q < AK / 4; ++q) {
            const int kk = q * 4 + (tid >> 6), c = tid & 63;
            const bool ok = k0 + kk < B && n0 + c < N;
            Bs[kk][c] = ok ? dcos[(long long)(k0 + kk) * N + n0 + c] : 0.f;
            // (round 3 read cos straight from memory inside the FMA loop: 64 dependent scalar loads per chunk and thread — 114 us)
            Cs[kk][c] = ok ? costh[(long long)(k0 + kk) * N + n0 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < AK; ++kk) {
            float av[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = As[ty * 4 + i][kk];
            const af4 b4 = *reinterpret_cast<const af4*>(&Bs[kk][tx * 4]);
            const af4 c4 = *reinterpret_cast<const af4*>(&Cs[kk][tx * 4]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], b4[j], acc[i][j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = fmaf(b4[j], c4[j], t[j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = e0 + ty * 4 + i;
        if (e >= E) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n >= N) continue;
            const float wi = winv[n];
            gW[(long long)e * N + n] = wi * (acc[i][j] - W[(long long)e * N + n] * wi * t[j]);
        }
    }
}

// partial gx_hat[chunk][b][e] = sum_{n in chunk} dcos[b][n] * W[e][n] * winv[n]: grid (E tiles, B tiles, N chunks)
__global__ void __launch_bounds__(256) amsoftmax_bwd_x_kernel(const float* __restrict__ W, const float* __restrict__ dcos,
                                                              const float* __restrict__ winv, float* __restrict__ part,
                                                              int B, int E, int N, int nchunk_cols) {
    __shared__ float As[AT][AK + 1];
    __shared__ __attribute__((aligned(16))) float Bs[AK][AT + 4];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int e0 = blockIdx.x * AT, b0 = blockIdx.y * AT;
    const int nlo = blockIdx.z * nchunk_cols, nhi = min(N, nlo + nchunk_cols);
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int k0 = nlo; k0 < nhi; k0 += AK) {
        {   // A chunk (64 b x 16 n) = dcos[b][n] * winv[n]
#pragma unroll
            for (int q = 0; q < AK / 4; ++q) {
                const int idx = q * 256 + tid, r = idx / AK, n = k0 + idx % AK;
                As[r][idx % AK] = (b0 + r < B && n < nhi) ? dcos[(long long)(b0 + r) * N + n] * winv[n] : 0.f;
            }
        }
        {   // B chunk (16 n x 64 e) = W[e][n]: element (kk, e); consecutive lanes take consecutive n of one e (n is contiguous in W)
#pragma unroll
            for (int q = 0; q < AK / 4; ++q) {
                const int idx = q * 256 + tid, e = idx / AK, kk = idx % AK;
                Bs[kk][e] = (e0 + e < E && k0 + kk < nhi) ? W[(long long)(e0 + e) * N + k0 + kk] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < AK; ++kk) {
            float av[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = As[ty * 4 + i][kk];
            const af4 b4 = *reinterpret_cast<const af4*>(&Bs[kk][tx * 4]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], b4[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int b = b0 + ty * 4 + i;
        if (b >= B) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = e0 + tx * 4 + j;
            if (e < E) part[((long long)blockIdx.z * B + b) * E + e] = acc[i][j];
        }
    }
}

// gx[b][e] = xinv[b] * (sum_chunk part[chunk][b][e] - x[b][e] * xinv[b] * u[b])
__global__ void __launch_bounds__(256) amsoftmax_bwd_x_reduce(const float* __restrict__ part, const float* __restrict__ x,
                                                              const float* __restrict__ xinv, const float* __restrict__ u,
                                                              float* __restrict__ gx, int B, int E, int nchunks) {
    const long long n_el = (long long)B * E;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_el; i += (long long)gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int c = 0; c < nchunks; ++c) v += part[(long long)c * n_el + i];
        const int b = (int)(i / E);
        const float xi = xinv[b];
        gx[i] = xi * (v - x[i] * xi * u[b]);
    }
}

// columns per workgroup of the dx product: 128 (47 chunks x 7 tiles = 329 workgroups for N = 5994; with 512 the launch was 84 workgroups
// of 32 sequential LDS chunks each: 92 us)
constexpr int kAmChunkCols = 128;
static int am_chunks(int N) { return (N + kAmChunkCols - 1) / kAmChunkCols; }

}  // namespace dasv

extern "C" {

size_t dasv_amsoftmax_workspace_bytes(int B, int E, int N) {
    if (B <= 0 || E <= 0 || N <= 0) return 0;
    const size_t ntile = (size_t)(N + dasv::AT - 1) / dasv::AT;
    // forward: [B][ntile][2] partials | [B] label logits;  backward: [B][N] dcos | [B] u | [chunks][B][E] partial gx
    const size_t fwd = ((size_t)B * ntile * 2 + B) * sizeof(float);
    const size_t bwd = ((size_t)B * N + B + (size_t)dasv::am_chunks(N) * B * E) * sizeof(float);
    return fwd > bwd ? fwd : bwd;
}

int dasv_amsoftmax_fwd(const float* x, const float* W, const int64_t* label, float* costh, float* logits, float* loss, float* lse,
                       float* xinv, float* winv, void* workspace, size_t workspace_bytes, int B, int E, int N, float margin,
                       float scale, float alpha, dasv_stream_t stream) {
    if (x == nullptr || W == nullptr || label == nullptr || costh == nullptr || xinv == nullptr || winv == nullptr) return DASV_E_NULL;
    if ((loss == nullptr) != (lse == nullptr)) return DASV_E_NULL;
    if (B <= 0 || E <= 0 || N <= 0) return DASV_E_SHAPE;
    const int ntile = (N + dasv::AT - 1) / dasv::AT;
    dasv::AmParams p;
    p.x = x; p.W = W; p.label = reinterpret_cast<const long long*>(label); p.costh = costh; p.logits = logits; p.xinv = xinv; p.winv = winv;
    p.part = nullptr; p.lab_logit = nullptr;
    p.B = B; p.E = E; p.N = N; p.ntile = ntile; p.s = scale;
    // loss.py:31-35: ((cos - m onehot) + alpha cos) / (1 + alpha) = cos - m onehot / (1 + alpha)
    p.m_eff = margin / (1.0f + alpha);
    if (loss != nullptr) {
        if (workspace == nullptr || workspace_bytes < dasv_amsoftmax_workspace_bytes(B, E, N)) return DASV_E_WORKSPACE;
        p.part = static_cast<float*>(workspace);
        p.lab_logit = p.part + (size_t)B * ntile * 2;
    }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(dasv::amsoftmax_fwd_kernel, dim3((unsigned)ntile, (unsigned)((B + dasv::AT - 1) / dasv::AT)), dim3(256), 0, s, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    if (loss != nullptr) {
        hipLaunchKernelGGL(dasv::amsoftmax_ce_finish_kernel, dim3(1), dim3(256), 0, s, p.part, p.lab_logit, p.label, N, lse, loss, B, ntile);
        e = hipGetLastError();
    }
    return (int)e;
}

int dasv_amsoftmax_bwd(const float* x, const float* W, const int64_t* label, const float* costh, const float* lse, const float* xinv,
                       const float* winv, const float* g_loss, const float* g_costh, const float* g_logits, float* gx, float* gW,
                       void* workspace, size_t workspace_bytes, int B, int E, int N, float margin, float scale, float alpha,
                       dasv_stream_t stream) {
    if (x == nullptr || W == nullptr || costh == nullptr || xinv == nullptr || winv == nullptr || gx == nullptr || gW == nullptr) return DASV_E_NULL;
    if (lse != nullptr && label == nullptr) return DASV_E_NULL;
    if (lse == nullptr && g_costh == nullptr && g_logits == nullptr) return DASV_E_NULL;
    if (B <= 0 || E <= 0 || N <= 0) return DASV_E_SHAPE;
    if (workspace == nullptr || workspace_bytes < dasv_amsoftmax_workspace_bytes(B, E, N)) return DASV_E_WORKSPACE;
    float* dcos = static_cast<float*>(workspace);
    float* u = dcos + (size_t)B * N;
    float* part = u + B;
    const int nch = dasv::am_chunks(N);
    hipStream_t s = (hipStream_t)stream;
    const long long n_el = (long long)B * N;
    const int blocks = (int)((n_el + 255) / 256 < 2048 ? (n_el + 255) / 256 : 2048);
    hipLaunchKernelGGL(dasv::amsoftmax_dcos_kernel, dim3(blocks), dim3(256), 0, s, costh, reinterpret_cast<const long long*>(label), lse,
                       g_loss, g_costh, g_logits, dcos, B, N, margin / (1.0f + alpha), scale);
    hipLaunchKernelGGL(dasv::amsoftmax_rowdot_kernel, dim3((unsigned)B), dim3(256), 0, s, dcos, costh, u, N);
    hipLaunchKernelGGL(dasv::amsoftmax_bwd_w_kernel, dim3((unsigned)((N + dasv::AT - 1) / dasv::AT), (unsigned)((E + dasv::AT - 1) / dasv::AT)),
                       dim3(256), 0, s, x, W, dcos, costh, xinv, winv, gW, B, E, N);
    hipLaunchKernelGGL(dasv::amsoftmax_bwd_x_kernel,
                       dim3((unsigned)((E + dasv::AT - 1) / dasv::AT), (unsigned)((B + dasv::AT - 1) / dasv::AT), (unsigned)nch), dim3(256), 0, s,
                       W, dcos, winv, part, B, E, N, dasv::kAmChunkCols);
    const long long nx = (long long)B * E;
    const int bx = (int)((nx + 255) / 256 < 1024 ? (nx + 255) / 256 : 1024);
    hipLaunchKernelGGL(dasv::amsoftmax_bwd_x_reduce, dim3(bx), dim3(256), 0, s, part, x, xinv, u, gx, B, E, nch);
    return (int)hipGetLastError();
}

}  // extern "C"
