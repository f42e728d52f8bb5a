// Fused double multi-head attention pooling, backward, for gfx950 (MI355X).
//
// What torch autograd derives from the reference's scripts/poolings.py:73-80, 45-51, 61-71,
// restated in closed form so that x is read once and grad_x written once:
//   gw[h]  = <go, c_h>;  gu[h] = w_h (gw[h] - <go, out>)           (sum_h w_h gw[h] == <go, out>)
//   gc_h   = w_h go + gu[h] a;   gcc_h = <gc_h, c_h>;   ga = sum_{b,h} gu[h] c_h
//   per frame t:  p = exp2(s_t - M) / L,  gp = <gc_h, x_t>,  gs = p (gp - gcc_h)
//                 gx_t = p gc_h + gs q_h / sqrt(H);   gq_h += gs x_t / sqrt(H)
// No softmax reduction is needed (M, L saved by the forward launch), so every row is independent and
// the kernel is a pure HBM stream with the same lane mapping as the forward kernel.  grad_query and
// grad_att are summed in a fixed order (per-utterance partial rows in the workspace, then a small
// reduction launch), so results are bitwise reproducible — no float atomics.
#include "dasv_common.hpp"
#include "../../include/dasv_hip.h"

#include <math.h>

namespace dasv {

struct BwdParams {
    const void* x;
    const float* query;
    const float* att;       // null: head stage skipped, `go` is grad_ctx [B,H,dh]
    const int32_t* lengths;
    const float* head_w;
    const float* ctx;
    const float* ml;
    const float* out;
    const float* go;
    void* gx;
    float* part_q;          // [B][H*dh] per-utterance partial sums of gs*x
    float* gu;              // [B][H]
    float* part_a;          // [B*arrivals][dh] per-workgroup partial sums of gu*c (vector kernel) / [B*H][dh] (generic)
    long long sb, st;
    long long seg;          // NHWC frames (PERM kernels): element stride between the F' segments of a frame (= channels C)
    int B, T, H, dh;
    int wpu, upw, arrivals;
    int t_pad;
    float qscale;           // log2(e)/sqrt(H)
    float rs;               // 1/sqrt(H)
};

__device__ __forceinline__ bool bwd_block_to_unit(int arrivals, int B, int& b, int& a) {
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    b = (k / arrivals) * 8 + xcd;
    a = k % arrivals;
    return b < B;
}

// PERM: x / grad_x are the front-end's NHWC map [B, T', F', C]; same re-addressing as in the forward kernel (dmha_fwd.hip)
template <bool PERM, int G, int NI, int EPV>
__device__ __forceinline__ int bwd_feat_of(int c, int j, int e) {
    return PERM ? (c * EPV + e) * NI + j : (c + G * j) * EPV + e;
}

template <typename XT, int G, int NI, int R, bool PERM = false>
__global__ void __launch_bounds__(512) dmha_bwd_vec(const BwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using E = Elem<XT>;
    constexpr int EPV = E::kPerVec;
    constexpr int GPW = 64 / G;
    const size_t vec_step = PERM ? (size_t)p.seg * E::kBytes : (size_t)G * 16;

    int b, a;
    if (!bwd_block_to_unit(p.arrivals, p.B, b, a)) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wpu = p.wpu, H = p.H, dh = p.dh, T = p.T;
    const int uu = w / wpu, wu = w - uu * wpu;
    const int h = a * p.upw + uu;
    const int c = lane % G, gw = lane / G;
    const int P = wpu * GPW;
    const int phase = wu * GPW + gw;
    int len = T;
    if (p.lengths != nullptr) len = max(0, min(T, p.lengths[b]));

    const size_t unit_off = ((size_t)b * p.sb + (size_t)h * (PERM ? G * EPV : dh)) * E::kBytes + (size_t)c * 16;
    const char* xrow = static_cast<const char*>(p.x) + unit_off;
    char* gxrow = static_cast<char*>(p.gx) + unit_off;
    const size_t row_bytes = (size_t)p.st * E::kBytes;

    float gq[NI][EPV];
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int e = 0; e < EPV; ++e) gq[j][e] = 0.f;

    const int rows_per_pass = P * R;
    const int npass = (T + rows_per_pass - 1) / rows_per_pass;

    auto load_pass = [&](uint4 (&buf)[R][NI], int pass) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int t = phase + P * (pass * R + i);
            const bool valid = t < len;
            const char* rp = xrow + (size_t)t * row_bytes;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (valid) v = *reinterpret_cast<const uint4*>(rp + (size_t)j * vec_step);
                buf[i][j] = v;
            }
        }
    };
    // memory operations return in issue order: the first rows of x go out before the (dependent) prologue loads
    uint4 bufA[R][NI], bufB[R][NI];
    if (npass > 0) load_pass(bufA, 0);

    // per-lane fragments of q, gc for this lane's columns
    float q[NI][EPV], gc[NI][EPV];
    float gcc = 0.f;
    {
        const float* ch = p.ctx + ((size_t)b * H + h) * dh;
        float cfrag[NI][EPV];
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
                const int col = bwd_feat_of<PERM, G, NI, EPV>(c, j, e);
                q[j][e] = p.query[(size_t)col * H + h];
                cfrag[j][e] = ch[col];
            }
        if (p.att != nullptr) {
            float gwp = 0.f, gdp = 0.f;
            float gof[NI][EPV];
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int e = 0; e < EPV; ++e) {
                    const int col = bwd_feat_of<PERM, G, NI, EPV>(c, j, e);
                    gof[j][e] = p.go[(size_t)b * dh + col];
                    gwp = fmaf(gof[j][e], cfrag[j][e], gwp);
                    gdp = fmaf(gof[j][e], p.out[(size_t)b * dh + col], gdp);
                }
            const float gwh = group_sum<G>(gwp), gdot = group_sum<G>(gdp);
            const float wh = p.head_w[(size_t)b * H + h];
            const float guh = wh * (gwh - gdot);
            if (wu == 0 && lane == 0) {
                p.gu[(size_t)b * H + h] = guh;
                smem[(size_t)p.upw * P * dh + uu] = guh;   // read back after the stream for the grad_att partial row
            }
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int e = 0; e < EPV; ++e) gc[j][e] = wh * gof[j][e] + guh * p.att[bwd_feat_of<PERM, G, NI, EPV>(c, j, e)];
        } else {
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int e = 0; e < EPV; ++e) gc[j][e] = p.go[((size_t)b * H + h) * dh + bwd_feat_of<PERM, G, NI, EPV>(c, j, e)];
        }
        float part = 0.f;
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < EPV; ++e) part = fmaf(gc[j][e], cfrag[j][e], part);
        gcc = group_sum<G>(part);
    }
    const float M = p.ml[((size_t)b * H + h) * 2];
    const float Lsum = p.ml[((size_t)b * H + h) * 2 + 1];
    const float Mu = (M == -INFINITY) ? 0.f : M;
    const float invL = (Lsum > 0.f) ? 1.0f / Lsum : 0.f;

    auto compute_pass = [&](const uint4 (&buf)[R][NI], int pass) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int t = phase + P * (pass * R + i);
            float f[NI][EPV];
            float ds = 0.f, dp = 0.f;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                E::unpack(buf[i][j], f[j]);
#pragma unroll
                for (int e = 0; e < EPV; ++e) { ds = fmaf(f[j][e], q[j][e], ds); dp = fmaf(f[j][e], gc[j][e], dp); }
            }
            ds = group_sum<G>(ds);
            dp = group_sum<G>(dp);
            const bool valid = t < len;
            const float pr = valid ? fast_exp2(ds * p.qscale - Mu) * invL : 0.f;
            const float gs = pr * (dp - gcc);
            const float gsr = gs * p.rs;
            if (t < T) {
                char* wp = gxrow + (size_t)t * row_bytes;
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    float o[EPV];
#pragma unroll
                    for (int e = 0; e < EPV; ++e) {
                        o[e] = fmaf(pr, gc[j][e], gsr * q[j][e]);
                        gq[j][e] = fmaf(gs, f[j][e], gq[j][e]);
                    }
                    *reinterpret_cast<uint4*>(wp + (size_t)j * vec_step) = E::pack(o);
                }
            }
        }
    };

    if (npass > 1) load_pass(bufB, 1);
    for (int pass = 0; pass < npass; pass += 2) {
        compute_pass(bufA, pass);
        if (pass + 2 < npass) load_pass(bufA, pass + 2);
        if (pass + 1 < npass) {
            compute_pass(bufB, pass + 1);
            if (pass + 3 < npass) load_pass(bufB, pass + 3);
        }
    }

    // every row phase parks its partial sum in its own LDS slot; the unit's first wave adds them in a fixed order
    float* red = smem;  // [upw*P][dh]
    {
        float* rs = red + (size_t)(uu * P + phase) * dh;
        typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            if constexpr (PERM) {
#pragma unroll
                for (int e = 0; e < EPV; ++e) rs[bwd_feat_of<true, G, NI, EPV>(c, j, e)] = gq[j][e];
            } else {
#pragma unroll
                for (int e = 0; e < EPV; e += 4)
                    *reinterpret_cast<f4*>(rs + (c + G * j) * EPV + e) = f4{gq[j][e], gq[j][e + 1], gq[j][e + 2], gq[j][e + 3]};
            }
        }
    }
    __syncthreads();
    if (wu == 0) {
        const float* ru = red + (size_t)uu * P * dh;
        for (int d = lane; d < dh; d += 64) {
            float v = 0.f;
#pragma unroll 8
            for (int k2 = 0; k2 < P; ++k2) v += ru[(size_t)k2 * dh + d];
            p.part_q[(size_t)b * H * dh + (size_t)h * dh + d] = v;
        }
    }
    // grad_att: this workgroup's heads pre-reduced, sum_h gu[b,h] * c[b,h,:]  (one row per workgroup)
    if (p.att != nullptr) {
        const float* gus = smem + (size_t)p.upw * P * dh;
        for (int d = tid; d < dh; d += blockDim.x) {
            float v = 0.f;
            for (int k2 = 0; k2 < p.upw; ++k2) v = fmaf(gus[k2], p.ctx[((size_t)b * H + a * p.upw + k2) * dh + d], v);
            p.part_a[((size_t)b * p.arrivals + a) * dh + d] = v;
        }
    }
}

// shape-agnostic backward: one 256-thread workgroup per unit
template <typename XT>
__global__ void __launch_bounds__(256) dmha_bwd_generic(const BwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using E = Elem<XT>;
    int b, h;
    if (!bwd_block_to_unit(p.arrivals, p.B, b, h)) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nthr = blockDim.x, nw = nthr >> 6;
    const int H = p.H, dh = p.dh, T = p.T;
    int len = T;
    if (p.lengths != nullptr) len = max(0, min(T, p.lengths[b]));
    float* pr_l = smem;               // [t_pad]
    float* gs_l = smem + p.t_pad;     // [t_pad]
    float* qs = gs_l + p.t_pad;       // [dh]
    float* gcs = qs + dh;             // [dh]
    float* red = gcs + dh;            // [3*nw]

    const float* ch = p.ctx + ((size_t)b * H + h) * dh;
    float gwp = 0.f, gdp = 0.f;
    if (p.att != nullptr)
        for (int d = tid; d < dh; d += nthr) {
            const float g = p.go[(size_t)b * dh + d];
            gwp = fmaf(g, ch[d], gwp);
            gdp = fmaf(g, p.out[(size_t)b * dh + d], gdp);
        }
    gwp = group_sum<64>(gwp);
    gdp = group_sum<64>(gdp);
    if (lane == 0) { red[w] = gwp; red[nw + w] = gdp; }
    __syncthreads();
    float gwh = 0.f, gdot = 0.f;
    for (int k = 0; k < nw; ++k) { gwh += red[k]; gdot += red[nw + k]; }
    float wh = 0.f, guh = 0.f;
    if (p.att != nullptr) {
        wh = p.head_w[(size_t)b * H + h];
        guh = wh * (gwh - gdot);
        if (tid == 0) p.gu[(size_t)b * H + h] = guh;
    }
    float gpart = 0.f;
    for (int d = tid; d < dh; d += nthr) {
        qs[d] = p.query[(size_t)d * H + h];
        const float g = (p.att != nullptr) ? wh * p.go[(size_t)b * dh + d] + guh * p.att[d]
                                           : p.go[((size_t)b * H + h) * dh + d];
        gcs[d] = g;
        gpart = fmaf(g, ch[d], gpart);
    }
    gpart = group_sum<64>(gpart);
    if (lane == 0) red[2 * nw + w] = gpart;
    __syncthreads();
    float gcc = 0.f;
    for (int k = 0; k < nw; ++k) gcc += red[2 * nw + k];
    const float M = p.ml[((size_t)b * H + h) * 2];
    const float Lsum = p.ml[((size_t)b * H + h) * 2 + 1];
    const float Mu = (M == -INFINITY) ? 0.f : M;
    const float invL = (Lsum > 0.f) ? 1.0f / Lsum : 0.f;
    const long long base = (long long)b * p.sb + (long long)h * dh;
    for (int t = tid; t < T; t += nthr) {
        float pr = 0.f, gs = 0.f;
        if (t < len) {
            float ds = 0.f, dp = 0.f;
            for (int d = 0; d < dh; ++d) {
                const float xv = E::load1(p.x, base + (long long)t * p.st + d);
                ds = fmaf(xv, qs[d], ds);
                dp = fmaf(xv, gcs[d], dp);
            }
            pr = fast_exp2(ds * p.qscale - Mu) * invL;
            gs = pr * (dp - gcc);
        }
        pr_l[t] = pr;
        gs_l[t] = gs;
    }
    __syncthreads();
    for (int d = tid; d < dh; d += nthr) {
        float gq = 0.f;
        const float qd = qs[d] * p.rs, gd = gcs[d];
        for (int t = 0; t < T; ++t) {
            const long long idx = base + (long long)t * p.st + d;
            const float xv = (t < len) ? E::load1(p.x, idx) : 0.f;
            E::store1(p.gx, idx, fmaf(pr_l[t], gd, gs_l[t] * qd));
            gq = fmaf(gs_l[t], xv, gq);
        }
        p.part_q[(size_t)b * H * dh + (size_t)h * dh + d] = gq;
        if (p.att != nullptr) p.part_a[((size_t)b * H + h) * dh + d] = guh * ch[d];
    }
}

// grad_query[d,h] = rs * sum_b part_q[b][h*dh+d];  grad_att[d] = sum_rows part_a[row][d]
// 1024 threads = 64 columns x 16 row phases; each thread adds rows ph, ph+16, ... with independent loads in flight.
struct ReduceParams {
    const float* part_q; const float* part_a;
    float* grad_query; float* grad_att;
    int B, H, dh, nq_blocks, rows_a; float rs;
};
__global__ void __launch_bounds__(1024) dmha_bwd_reduce(const ReduceParams p) {
    __shared__ float red[16][64];
    const int col_l = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int D = p.H * p.dh;
    const bool is_q = (int)blockIdx.x < p.nq_blocks;
    const int col = (is_q ? blockIdx.x : blockIdx.x - p.nq_blocks) * 64 + col_l;
    const int ncol = is_q ? D : p.dh, nrow = is_q ? p.B : p.rows_a;
    const float* src = is_q ? p.part_q : p.part_a;
    float v = 0.f;
    if (col < ncol) {
#pragma unroll 4
        for (int r = ph; r < nrow; r += 16) v += src[(size_t)r * ncol + col];
    }
    red[ph][col_l] = v;
    __syncthreads();
    if (ph == 0 && col < ncol) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][col_l];
        if (is_q) {
            const int h = col / p.dh, d = col - h * p.dh;
            p.grad_query[(size_t)d * p.H + h] = s * p.rs;
        } else {
            p.grad_att[col] = s;
        }
    }
}

constexpr int bwd_rows_per_pass_for(int ni) { return ni == 1 ? 4 : (ni <= 3 ? 2 : 1); }

template <typename XT, int G, int NI>
static hipError_t launch_bwd_vec(const BwdParams& p, dim3 grid, dim3 block, size_t lds, hipStream_t s) {
    constexpr int R = bwd_rows_per_pass_for(NI);
    hipLaunchKernelGGL((dmha_bwd_vec<XT, G, NI, R>), grid, block, lds, s, p);
    return hipGetLastError();
}

static bool dispatch_bwd_vec_nhwc(int G, int NI, const BwdParams& p, dim3 grid, dim3 block, size_t lds, hipStream_t s, hipError_t& err) {
#define DASV_CASE(g) if (G == g && NI == 5) { hipLaunchKernelGGL((dmha_bwd_vec<F32, g, 5, 1, true>), grid, block, lds, s, p); err = hipGetLastError(); return true; }
    DASV_CASE(4) DASV_CASE(8) DASV_CASE(16)
#undef DASV_CASE
    return false;
}

template <typename XT>
static bool dispatch_bwd_vec(int G, int NI, const BwdParams& p, dim3 grid, dim3 block, size_t lds, hipStream_t s, hipError_t& err) {
#define DASV_CASE(g, n) if (G == g && NI == n) { err = launch_bwd_vec<XT, g, n>(p, grid, block, lds, s); return true; }
    DASV_CASE(16, 1) DASV_CASE(16, 2) DASV_CASE(16, 3) DASV_CASE(16, 4)
    DASV_CASE(16, 5) DASV_CASE(16, 6) DASV_CASE(16, 7) DASV_CASE(16, 8)
    DASV_CASE(8, 1) DASV_CASE(8, 3) DASV_CASE(8, 5) DASV_CASE(8, 7)
    DASV_CASE(4, 1) DASV_CASE(4, 3) DASV_CASE(4, 5) DASV_CASE(4, 7)
#undef DASV_CASE
    return false;
}

static int bwd_largest_divisor_leq(int n, int cap) {
    for (int d = cap; d >= 1; --d) if (n % d == 0) return d;
    return 1;
}

template <typename XT>
static int bwd_impl(const void* x, const float* query, const float* att, const int32_t* lengths, const float* head_w,
                    const float* ctx, const float* ml, const float* out, const float* go, void* gx, float* grad_query,
                    float* grad_att, void* workspace, size_t workspace_bytes, int B, int T, int H, int dh, int64_t sb,
                    int64_t st, hipStream_t stream, int perm_f = 0, int64_t seg = 0) {
    using E = Elem<XT>;
    if (x == nullptr || query == nullptr || ctx == nullptr || ml == nullptr || go == nullptr || gx == nullptr ||
        grad_query == nullptr)
        return DASV_E_NULL;
    if (att != nullptr && (head_w == nullptr || out == nullptr || grad_att == nullptr)) return DASV_E_NULL;
    if (B <= 0 || T <= 0 || H <= 0 || dh <= 0) return DASV_E_SHAPE;
    if ((long long)H * dh > (1ll << 24) || (long long)B > (1ll << 24)) return DASV_E_SHAPE;
    if (st < (int64_t)H * dh || sb < 0) return DASV_E_STRIDE;
    if (workspace == nullptr || workspace_bytes < dasv_dmha_bwd_workspace_bytes(B, T, H, dh)) return DASV_E_WORKSPACE;

    BwdParams p;
    p.x = x; p.query = query; p.att = att; p.lengths = lengths; p.head_w = head_w; p.ctx = ctx; p.ml = ml;
    p.out = out; p.go = go; p.gx = gx;
    p.part_q = static_cast<float*>(workspace);
    p.part_a = p.part_q + (size_t)B * H * dh;
    p.gu = p.part_a + (size_t)B * H * dh;
    p.sb = sb; p.st = st; p.seg = seg; p.B = B; p.T = T; p.H = H; p.dh = dh;
    p.t_pad = round_up_i(T, 4);
    p.qscale = (float)(1.4426950408889634 / sqrt((double)H));
    p.rs = (float)(1.0 / sqrt((double)H));

    const bool aligned = ((uintptr_t)x % 16 == 0) && ((uintptr_t)gx % 16 == 0) && ((sb * E::kBytes) % 16 == 0) &&
                         ((st * E::kBytes) % 16 == 0) && (dh % E::kPerVec == 0);
    int G = 0, NI = 0;
    if (perm_f > 0) {
        const int cph = dh / perm_f;
        if (!aligned || cph * perm_f != dh || cph % E::kPerVec != 0 || (seg * E::kBytes) % 16 != 0) return DASV_E_SHAPE;
        G = cph / E::kPerVec;
        NI = perm_f;
        if (!(G == 4 || G == 8 || G == 16) || NI != 5 || XT::kCode != 0) return DASV_E_SHAPE;
    } else if (aligned) {
        const int vecs = dh / E::kPerVec;
        G = 16;
        while (G > 1 && vecs % G != 0) G >>= 1;
        NI = vecs / G;
        if (G < 4 || NI > 8) G = 0;
    }
    const int groups8 = (B + 7) / 8 * 8;
    bool launched = false;
    hipError_t err = hipSuccess;
    if (G != 0) {
        const int gpw = 64 / G;
        const int rpp = bwd_rows_per_pass_for(NI);
        int wpu = 1;
        while (wpu < 8 && wpu * gpw * rpp * 5 < T) wpu <<= 1;
        int nw = 8;
        if (nw < wpu) nw = wpu;
        int upw = bwd_largest_divisor_leq(H, nw / wpu);
        while (upw > 1 && (long long)groups8 * (H / upw) < 256) upw = bwd_largest_divisor_leq(H, upw - 1);
        while (upw * wpu > 4 && upw > 1 && (long long)groups8 * (H / upw) > 1024) upw = bwd_largest_divisor_leq(H, upw - 1);
        nw = upw * wpu;
        p.wpu = wpu; p.upw = upw; p.arrivals = H / upw;
        const size_t lds = ((size_t)upw * wpu * gpw * dh + 8) * sizeof(float);
        if (lds <= 64 * 1024) {
            dim3 grid((unsigned)(groups8 * p.arrivals)), block((unsigned)(nw * 64));
            launched = perm_f > 0 ? dispatch_bwd_vec_nhwc(G, NI, p, grid, block, lds, stream, err)
                                  : dispatch_bwd_vec<XT>(G, NI, p, grid, block, lds, stream, err);
        }
    }
    if (!launched && perm_f > 0) return DASV_E_SHAPE;
    if (!launched) {
        p.wpu = 4; p.upw = 1; p.arrivals = H;
        const size_t lds = ((size_t)2 * p.t_pad + 2 * dh + 16) * sizeof(float);
        if (lds > 160 * 1024) return DASV_E_SHAPE;   // ~20 k pooled frames on the unaligned correctness path
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dmha_bwd_generic<XT>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        dim3 grid((unsigned)(groups8 * H)), block(256);
        hipLaunchKernelGGL((dmha_bwd_generic<XT>), grid, block, lds, stream, p);
        err = hipGetLastError();
    }
    if (err != hipSuccess) return (int)err;

    ReduceParams r;
    r.part_q = p.part_q; r.part_a = p.part_a; r.grad_query = grad_query;
    r.grad_att = (att != nullptr) ? grad_att : nullptr;
    r.B = B; r.H = H; r.dh = dh; r.rs = p.rs;
    r.rows_a = B * p.arrivals;
    r.nq_blocks = (H * dh + 63) / 64;
    const int na_blocks = (att != nullptr) ? (dh + 63) / 64 : 0;
    hipLaunchKernelGGL(dmha_bwd_reduce, dim3((unsigned)(r.nq_blocks + na_blocks)), dim3(1024), 0, stream, r);
    return (int)hipGetLastError();
}

}  // namespace dasv

extern "C" {

size_t dasv_dmha_bwd_workspace_bytes(int B, int T, int H, int dh) {
    (void)T;
    if (B <= 0 || H <= 0 || dh <= 0) return 0;
    // per-utterance grad_query partial rows | per-workgroup grad_att partial rows | gu
    return ((size_t)2 * B * H * dh + (size_t)dasv::round_up_i(B * H, 4)) * sizeof(float);
}

int dasv_dmha_bwd(int x_dtype, const void* x, const float* query, const float* att, const uint8_t* head_keep,
                  const int32_t* lengths, const float* head_w, const float* ctx, const float* ml, const float* out,
                  const float* grad_out, void* grad_x, float* grad_query, float* grad_att, void* workspace,
                  size_t workspace_bytes, int B, int T, int H, int dh, int64_t x_stride_b, int64_t x_stride_t,
                  dasv_stream_t stream) {
    (void)head_keep;  // masked heads already carry head_w == 0 from the forward launch
    hipStream_t s = (hipStream_t)stream;
    switch (x_dtype) {
        case DASV_F32:
            return dasv::bwd_impl<dasv::F32>(x, query, att, lengths, head_w, ctx, ml, out, grad_out, grad_x, grad_query,
                                             grad_att, workspace, workspace_bytes, B, T, H, dh, x_stride_b, x_stride_t, s);
        case DASV_BF16:
            return dasv::bwd_impl<dasv::BF16>(x, query, att, lengths, head_w, ctx, ml, out, grad_out, grad_x, grad_query,
                                              grad_att, workspace, workspace_bytes, B, T, H, dh, x_stride_b, x_stride_t, s);
        case DASV_F16:
            return dasv::bwd_impl<dasv::F16>(x, query, att, lengths, head_w, ctx, ml, out, grad_out, grad_x, grad_query,
                                             grad_att, workspace, workspace_bytes, B, T, H, dh, x_stride_b, x_stride_t, s);
        default:
            return DASV_E_DTYPE;
    }
}

int dasv_dmha_bwd_nhwc_f32(const float* x, const float* query, const float* att, const int32_t* lengths, const float* head_w,
                           const float* ctx, const float* ml, const float* out, const float* grad_out, float* grad_x,
                           float* grad_query, float* grad_att, void* workspace, size_t workspace_bytes, int B, int T, int F,
                           int C, int H, dasv_stream_t stream) {
    if (F != 5 || C <= 0 || H <= 0 || C % H != 0) return DASV_E_SHAPE;
    const int dh = C / H * F;
    return dasv::bwd_impl<dasv::F32>(x, query, att, lengths, head_w, ctx, ml, out, grad_out, grad_x, grad_query, grad_att,
                                     workspace, workspace_bytes, B, T, H, dh, (int64_t)T * F * C, (int64_t)F * C,
                                     (hipStream_t)stream, F, (int64_t)C);
}

}  // extern "C"
