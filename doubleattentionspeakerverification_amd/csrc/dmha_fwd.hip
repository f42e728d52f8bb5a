// Fused double multi-head attention pooling, forward, for gfx950 (MI355X).
//
// Replaces the eager op chain of the reference's scripts/poolings.py
//   :73-80  innerKeyValueAttention  (matmul -> diagonal -> softmax over T -> broadcast mul -> sum)
//   :45-51, :61-71 HeadAttention    (matvec -> [mask] -> softmax over H -> weighted sum)
//   :126-129 DoubleMHA.forward
// with one launch that streams x[B,T,D] from HBM exactly once.
//
// Work decomposition (HBM-bound: ~1 FLOP/byte, so no MFMA; the job is to keep
// enough 16-byte loads in flight on all 256 CUs):
//   unit      = one (utterance b, head h): T rows of dh contiguous elements.
//   group     = G lanes that own one row of a unit at a time; a lane loads NI
//               16-byte vectors per row (vector c + G*j of the row), so one
//               wave-instruction fetches 64/G whole G*16-byte row segments.
//   score     = per-lane partial dot with the (pre-scaled) query fragment,
//               summed over the group with DPP row operations.
//   softmax   = per-group online (max, sum, weighted accumulator) in registers;
//               groups of a wave are merged by xor-shuffles, waves of a unit
//               through LDS.  Raw scores are parked in LDS so the normalised
//               alignments are written once, after the unit's (max, sum) is known.
//   head stage= softmax over heads + weighted sum is an ASSOCIATIVE reduction over heads
//               (online softmax: (max, sum, weighted vector)), so every workgroup first
//               folds ITS heads into one partial {M_w, S_w, V_w[dh]} and publishes only
//               that (write-through, agent scope) beside its heads' scores u_h; a
//               per-utterance arrival counter elects the last workgroup of each utterance,
//               which pulls the `arrivals` partials (dh + 4 floats each — not the H context
//               vectors: 2.6 KB instead of 20 KB for the model's 32 x 160 pooling) with
//               agent-scope loads and writes out / head_w.  ctx[B,H,dh] is a plain output.
//               Workgroups of one utterance share blockIdx % 8 (one XCD's L2 under
//               round-robin placement): speed only, correctness never depends on it.
#include "dasv_common.hpp"
#include "../../include/dasv_hip.h"

#include <math.h>
#include <string.h>

namespace dasv {

struct FwdParams {
    const void* x;
    const float* query;    // [dh,H]
    const float* att;      // [dh] or null
    const uint8_t* keep;   // [B,H] or null
    const int32_t* lengths;  // [B] or null
    float* out;            // [B,dh]
    float* align;          // [B,T,H] or null
    float* head_w;         // [B,H] or null
    float* ctx;            // [B,H,dh] or null
    float* ml;             // [B,H,2] or null
    int* counters;         // [B] arrival counters (zero between launches)
    float* exch_u;         // [B][round_up(H,4)] head scores <c_h, att> (masked heads: -inf), published beside the partials
    float* exch_p;         // [B][arrivals][4 + round_up(dh,4)] per-workgroup partials {M_w, S_w, 0, 0 | V_w[dh]}
    long long sb, st;      // element strides of x
    long long seg;         // NHWC frames (PERM kernels): element stride between the F' segments of a frame (= channels C)
    int B, T, H, dh;
    int wpu;               // waves cooperating on one unit
    int upw;               // units per workgroup
    int arrivals;          // workgroups per utterance = H / upw
    int nparts;            // partials per utterance the head stage merges: `arrivals` (folded per workgroup) or H (one per head)
    int t_pad;             // score row pitch (floats)
    float* s_glob;         // generic kernel, T too long for LDS: [B*H][t_pad] raw scores in the workspace (else null)
    float qscale;          // log2(e) / sqrt(H)
#ifdef DASV_STAMPS
    unsigned long long* stamps;  // diagnostic build only: [gridDim.x][8] s_memrealtime stamps (100 MHz)
#endif
};

#ifdef DASV_STAMPS
#define DASV_STAMP(p, k)                                                                          \
    do {                                                                                          \
        if (threadIdx.x == 0 && (p).stamps != nullptr)                                            \
            (p).stamps[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime();          \
    } while (0)
#else
#define DASV_STAMP(p, k) do { } while (0)
#endif

// LDS carve (floats).  Stream phase: [upw][t_pad] raw scores | [upw*wpu][2] per-wave (m,l) | [upw*P][dh] per-group
// partial accumulators (already scaled to their wave's max).  The head stage (last arriver only, after every
// other use) aliases the same bytes from offset 0: [round_up(H,4)] head weights | [ceil(H/8)][dh] partial
// outputs.  One flag word at the end.
// ... | [upw][dh] the workgroup's context vectors | [round_up(upw,4)] their head scores (the fold of the workgroup's heads)
__host__ __device__ inline size_t fwd_lds_stream_floats(int upw, int t_pad, int wpu, int P, int dh) {
    return (size_t)upw * t_pad + (size_t)round_up_i(upw * wpu * 2, 4) + (size_t)upw * P * dh + (size_t)upw * dh + (size_t)round_up_i(upw, 4);
}
__host__ __device__ inline size_t fwd_lds_head_floats(int H, int dh) {
    return (size_t)round_up_i(H, 4) + 16;   // head weights | {M, 1/S}
}
__host__ __device__ inline size_t fwd_lds_floats(int upw, int t_pad, int wpu, int P, int H, int dh) {
    const size_t a = fwd_lds_stream_floats(upw, t_pad, wpu, P, dh), b = fwd_lds_head_floats(H, dh);
    return (a > b ? a : b) + 4;
}

// block -> (utterance, head block): the `arrivals` workgroups of an utterance share blockIdx % 8.
__device__ __forceinline__ bool block_to_unit(int arrivals, int B, int& b, int& a) {
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    b = (k / arrivals) * 8 + xcd;
    a = k % arrivals;
    return b < B;
}

// ---- head stage -------------------------------------------------------------------------------------
// out = sum_h softmax(u)_h c_h is an online-softmax reduction over heads, so it folds: a workgroup that owns heads h0 .. h0+upw-1
// publishes  M_w = max u_h,  S_w = sum e^(u_h - M_w),  V_w = sum e^(u_h - M_w) c_h  (a masked head has u = -inf: weight 0), and
//   out = sum_w e^(M_w - M) V_w / sum_w e^(M_w - M) S_w,   head_w[h] = e^(u_h - M) / S,   M = max_w M_w.
// All heads masked: every S_w = 0 and out = 0/0 = NaN, as the reference's softmax over -inf scores (poolings.py:47-50).
__host__ __device__ inline int part_pitch(int dh) { return 4 + round_up_i(dh, 4); }   // floats per partial: {M, S, 0, 0 | V[dh]}

typedef float hs_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_agent4(float* base, int byte_off, int bytes, hs_f4 v) {   // 16-byte write-through (sc1) store
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, bytes, 0x00020000);
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), r, byte_off, 0, 16);
}
__device__ __forceinline__ hs_f4 load_agent4(const float* base, int byte_off, int bytes) {       // 16-byte L1-bypassing (sc1) load
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000);
    return __builtin_bit_cast(hs_f4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16));
}

// Fold and publish, by ONE wave (the caller's wave 0; all 64 lanes active): cbuf [upw][dh] context vectors and ubuf [upw] masked
// head scores of the workgroup's heads (LDS, behind a barrier).  Everything another workgroup will read is stored write-through
// by this wave, which then drains its stores — so its lane 0 may take the arrival ticket right after, with no workgroup barrier.
template <bool VEC4>
__device__ __forceinline__ void fold_publish(const FwdParams& p, int b, int a, const float* cbuf, const float* ubuf) {
    const int lane = threadIdx.x & 63, dh = p.dh, upw = p.upw, pitch = part_pitch(dh);
    // at most 8 heads per workgroup (upw <= nw <= 8, the launch planner): their scores are read once, with the index clamped instead of a
    // `k < upw` branch per read, the weights e^(u_k - M) computed once, and a feature's `upw` context values fetched together —
    // written as loops over the runtime `upw` every read waited for the one before it (1.7 us from the unit's result to the
    // published partial in the model's shape)
    float uk[8], wk[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) uk[k] = ubuf[k < upw ? k : 0];
    float M = -INFINITY;
#pragma unroll
    for (int k = 0; k < 8; ++k) M = fmaxf(M, k < upw ? uk[k] : -INFINITY);
    const float Mu = (M == -INFINITY) ? 0.f : M;
    float S = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { wk[k] = k < upw ? fast_exp2((uk[k] - Mu) * kLog2e) : 0.f; S += wk[k]; }
    float* pp = p.exch_p + ((size_t)b * p.nparts + a) * pitch;
    if (lane < upw) store_agent(p.exch_u + (size_t)b * round_up_i(p.H, 4) + a * upw + lane, ubuf[lane]);
    if constexpr (VEC4) {
        if (lane == 0) store_agent4(pp, 0, pitch * 4, hs_f4{M, S, 0.f, 0.f});
        for (int q = lane; q < (dh >> 2); q += 64) {
            hs_f4 cv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) cv[k] = *reinterpret_cast<const hs_f4*>(cbuf + (size_t)(k < upw ? k : 0) * dh + q * 4);
            hs_f4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 8; ++k) v += wk[k] * cv[k];
            store_agent4(pp, (4 + q * 4) * 4, pitch * 4, v);
        }
    } else {
        if (lane == 0) { store_agent(pp, M); store_agent(pp + 1, S); }
        for (int d = lane; d < dh; d += 64) {
            float v = 0.f;
            for (int k = 0; k < upw; ++k) v = fmaf(fast_exp2((ubuf[k] - Mu) * kLog2e), cbuf[(size_t)k * dh + d], v);
            store_agent(pp + 4 + d, v);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's write-through stores have left: the ticket may follow
}

// The last arriver of utterance b: pull the `arrivals` partials and the H scores with agent-scope loads only, merge, write
// out / head_w.  All threads of the workgroup call this (uniform control flow).  lds: fwd_lds_head_floats() floats.
template <bool VEC4>
__device__ __forceinline__ void head_stage(const FwdParams& p, int b, float* lds) {
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63;
    const int H = p.H, dh = p.dh, A = p.nparts, hpad = round_up_i(H, 4), pitch = part_pitch(dh);
    float* fw = lds;                 // [A <= H] e^(M_a - M)
    float* stat = lds + hpad;        // {1/S}
    const float* pg = p.exch_p + (size_t)b * A * pitch;
    const float* ug = p.exch_u + (size_t)b * hpad;
    constexpr int CH = 8;            // partial vectors a thread keeps in flight
    hs_f4 pv[CH];
    const int nq = dh >> 2;
    if constexpr (VEC4) {            // the one round of vector loads goes out before anything waits
        if (tid < nq) {
#pragma unroll
            for (int k = 0; k < CH; ++k)
                if (k < A) pv[k] = load_agent4(pg, (k * pitch + 4 + tid * 4) * 4, A * pitch * 4);
        }
    }
    if (tid < 64) {
        // every load of this wave goes out before anything waits: lane k the header {M_k, S_k} of partial k, lane h the score u_h
        // (one round trip; more than 64 partials / heads take further rounds)
        float mk = -INFINITY, sk = 0.f, uh = -INFINITY;
        if (lane < A) { mk = load_agent(pg + (size_t)lane * pitch); sk = load_agent(pg + (size_t)lane * pitch + 1); }
        if (lane < H) uh = load_agent(ug + lane);
        float mx = group_max<64>(mk);
        for (int a0 = 64; a0 < A; a0 += 64)
            mx = fmaxf(mx, group_max<64>((a0 + lane < A) ? load_agent(pg + (size_t)(a0 + lane) * pitch) : -INFINITY));
        const float Mu = (mx == -INFINITY) ? 0.f : mx;
        const float f0 = (lane < A) ? fast_exp2((mk - Mu) * kLog2e) : 0.f;
        if (lane < A) fw[lane] = f0;
        float den = group_sum<64>(f0 * sk);
        for (int a0 = 64; a0 < A; a0 += 64) {
            float f = 0.f, sv = 0.f;
            if (a0 + lane < A) {
                f = fast_exp2((load_agent(pg + (size_t)(a0 + lane) * pitch) - Mu) * kLog2e);
                sv = load_agent(pg + (size_t)(a0 + lane) * pitch + 1);
                fw[a0 + lane] = f;
            }
            den += group_sum<64>(f * sv);
        }
        const float inv = 1.0f / den;                       // all heads masked: 1/0, and out = 0 * inf = NaN like the reference
        if (lane == 0) stat[0] = inv;
        if (p.head_w != nullptr) {
            if (lane < H) p.head_w[(size_t)b * H + lane] = fast_exp2((uh - Mu) * kLog2e) * inv;
            for (int h0 = 64; h0 < H; h0 += 64)
                if (h0 + lane < H)
                    p.head_w[(size_t)b * H + h0 + lane] = fast_exp2((load_agent(ug + h0 + lane) - Mu) * kLog2e) * inv;
        }
    }
    __syncthreads();
    const float inv = stat[0];
    if constexpr (VEC4) {
        for (int q = tid; q < nq; q += nthr) {
            hs_f4 o = {0.f, 0.f, 0.f, 0.f};
            for (int a0 = 0; a0 < A; a0 += CH) {
                if (a0 > 0 || q != tid) {
#pragma unroll
                    for (int k = 0; k < CH; ++k)
                        if (a0 + k < A) pv[k] = load_agent4(pg, ((a0 + k) * pitch + 4 + q * 4) * 4, A * pitch * 4);
                }
#pragma unroll
                for (int k = 0; k < CH; ++k)
                    if (a0 + k < A) o += fw[a0 + k] * pv[k];
            }
            *reinterpret_cast<hs_f4*>(p.out + (size_t)b * dh + q * 4) = o * inv;
        }
    } else {
        for (int d = tid; d < dh; d += nthr) {
            float o = 0.f;
            for (int k = 0; k < A; ++k) o = fmaf(fw[k], load_agent(pg + (size_t)k * pitch + 4 + d), o);
            p.out[(size_t)b * dh + d] = o * inv;
        }
    }
}

// The workgroup's tail: fold + publish + ticket by wave 0, the election result to everybody through LDS, the head stage by
// the workgroup that drew the last ticket.  Called by all threads behind the barrier that made cbuf / ubuf visible; `between`
// runs after the ticket has been requested (work whose latency should hide the ticket's: the alignment stores).
template <bool VEC4, typename Between>
__device__ __forceinline__ void publish_and_finish(const FwdParams& p, int b, int a, const float* cbuf, const float* ubuf,
                                                   float* lds_head, int* lds_flag, Between between) {
    int ticket = 0;
    if (threadIdx.x < 64) {
        fold_publish<VEC4>(p, b, a, cbuf, ubuf);
        DASV_STAMP(p, 4);
        if (threadIdx.x == 0) ticket = __hip_atomic_fetch_add(p.counters + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    between();
    if (threadIdx.x == 0) {
        const int last = (ticket == p.arrivals - 1);
        if (last) __hip_atomic_store(p.counters + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // leave it zeroed
        *lds_flag = last;
    }
    __syncthreads();   // also: every wave is past its last use of the LDS the head stage aliases
    DASV_STAMP(p, 5);
    if (*lds_flag == 0) return;
    head_stage<VEC4>(p, b, lds_head);
}

// ---- the streaming kernel ---------------------------------------------------------------------
// PERM: x is the front-end's own NHWC map [B, T', F', C] (CNNs.py:88-89's transpose/contiguous/view folded into the
// addressing).  The reference's feature index is d = c*F' + f, so head h owns channels [h*C/H, (h+1)*C/H) of every one of
// the F' = NI segments of a frame: vector j of a lane is 16 bytes of segment f = j (G = C/H/4 lanes per segment), and
// element e of it is feature (4*c + e)*F' + j of the head.  Only the addresses and the gathers of q / the LDS parking
// change; everything after the parking works on the reference's feature order.
template <bool PERM, int G, int NI, int EPV>
__device__ __forceinline__ int feat_of(int c, int j, int e) {
    return PERM ? (c * EPV + e) * NI + j : (c + G * j) * EPV + e;
}

template <typename XT, int G, int NI, int R, bool PERM = false>
__global__ void __launch_bounds__(512) dmha_fwd_vec(const FwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using E = Elem<XT>;
    constexpr int EPV = E::kPerVec;
    constexpr int GPW = 64 / G;
    constexpr int DH = G * NI * EPV;
    constexpr int ND = (DH + 63) / 64;
    const size_t vec_step = PERM ? (size_t)p.seg * E::kBytes : (size_t)G * 16;   // bytes between a lane's vectors j, j+1

    int b, a;
    if (!block_to_unit(p.arrivals, p.B, b, a)) return;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wpu = p.wpu, upw = p.upw, H = p.H, T = p.T;
    const int uu = w / wpu, wu = w - uu * wpu;
    const int h = a * upw + uu;
    const int c = lane % G, gw = lane / G;
    const int P = wpu * GPW;
    const int phase = wu * GPW + gw;
    int len = T;
    if (p.lengths != nullptr) len = max(0, min(T, p.lengths[b]));

    float* s_lds = smem + (size_t)uu * p.t_pad;
    float* red_ml = smem + (size_t)upw * p.t_pad;             // [upw*wpu][2]
    float* red_acc = red_ml + (size_t)round_up_i(upw * wpu * 2, 4);   // [upw*P][DH], 16-byte aligned
    int* flag = reinterpret_cast<int*>(smem + fwd_lds_floats(upw, p.t_pad, wpu, P, H, DH) - 4);

    const char* xrow = static_cast<const char*>(p.x) +
                       ((size_t)b * p.sb + (size_t)h * (PERM ? G * EPV : DH)) * E::kBytes + (size_t)c * 16;
    const size_t row_bytes = (size_t)p.st * E::kBytes;
    const int rows_per_pass = P * R;
    const int npass = (len + rows_per_pass - 1) / rows_per_pass;

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

    // (1) memory ops return in issue order, so: first pass of x, then the query fragment the first pass needs
    //     (its latency hides behind the x loads already in flight), then the second pass of x, then the
    //     head-attention vector (needed only after the stream)
    DASV_STAMP(p, 0);
    uint4 bufA[R][NI], bufB[R][NI];
    if (npass > 0) load_pass(bufA, 0);
    float q[NI][EPV];
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int e = 0; e < EPV; ++e) q[j][e] = p.query[(size_t)feat_of<PERM, G, NI, EPV>(c, j, e) * H + h];
    if (npass > 1) load_pass(bufB, 1);
    float attv[ND];
#pragma unroll
    for (int k = 0; k < ND; ++k) attv[k] = (p.att != nullptr && wu == 0 && lane + 64 * k < DH) ? p.att[lane + 64 * k] : 0.f;
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int e = 0; e < EPV; ++e) q[j][e] *= p.qscale;   // pre-scaled: scores come out in the log2 domain

    float m = -INFINITY, l = 0.f;
    float acc[NI][EPV];
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int e = 0; e < EPV; ++e) acc[j][e] = 0.f;

    auto compute_pass = [&](const uint4 (&buf)[R][NI], int pass) {
        float s[R];
        float pm = -INFINITY;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            float d = 0.f;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                float f[EPV];
                E::unpack(buf[i][j], f);
#pragma unroll
                for (int e = 0; e < EPV; ++e) d = fmaf(f[e], q[j][e], d);
            }
            d = group_sum<G>(d);
            const int t = phase + P * (pass * R + i);
            s[i] = (t < len) ? d : -INFINITY;
            pm = fmaxf(pm, s[i]);
            if (c == 0 && t < len) s_lds[t] = d;
        }
        const float m_new = fmaxf(m, pm);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float corr = fast_exp2(m - m_use);
        l *= corr;
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < EPV; ++e) acc[j][e] *= corr;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const float pr = fast_exp2(s[i] - m_use);
            l += pr;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                float f[EPV];
                E::unpack(buf[i][j], f);
#pragma unroll
                for (int e = 0; e < EPV; ++e) acc[j][e] = fmaf(pr, f[e], acc[j][e]);
            }
        }
        m = m_new;
    };

    // (3) the stream: two passes always in flight
    for (int pass = 0; pass < npass; pass += 2) {
        compute_pass(bufA, pass);
        if (pass == 0) DASV_STAMP(p, 1);
        if (pass + 2 < npass) load_pass(bufA, pass + 2);
        if (pass + 1 < npass) {
            compute_pass(bufB, pass + 1);
            if (pass + 3 < npass) load_pass(bufB, pass + 3);
        }
    }
    DASV_STAMP(p, 2);
#ifdef DASV_STAMPS
    if (lane == 0 && p.stamps != nullptr)
        p.stamps[(size_t)gridDim.x * 8 + (size_t)blockIdx.x * 8 + w] = __builtin_amdgcn_s_memrealtime();
#endif

    // (4) wave-level (max, sum) by xor butterflies over the groups (2 values only); every group rescales its
    //     accumulator to the wave's max and parks it in its own LDS slot
    {
        // (m, l) are uniform over each group of G lanes: the wave's maximum and the sum over its groups by DPP (each group
        // counted once), not through the LDS crossbar (four dependent ds_bpermute round trips in round 2)
        const float mw = wave_max_all<G>(m);
        const float mwu = (mw == -INFINITY) ? 0.f : mw;
        const float sc = fast_exp2(m - mwu);
        const float lw = wave_sum_all<G>(l * sc);
        if (lane == 0) { red_ml[2 * w] = mw; red_ml[2 * w + 1] = lw; }
        float* ra = red_acc + (size_t)(uu * P + phase) * DH;
        typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            if constexpr (PERM) {
#pragma unroll
                for (int e = 0; e < EPV; ++e) ra[feat_of<true, G, NI, EPV>(c, j, e)] = acc[j][e] * sc;
            } else {
#pragma unroll
                for (int e = 0; e < EPV; e += 4)
                    *reinterpret_cast<f4*>(ra + (c + G * j) * EPV + e) =
                        f4{acc[j][e] * sc, acc[j][e + 1] * sc, acc[j][e + 2] * sc, acc[j][e + 3] * sc};
            }
        }
    }
    __syncthreads();
    DASV_STAMP(p, 7);

    // (5) finish the unit: (M, L) by every thread (<= 8 independent LDS reads); context vector + head score by
    //     the unit's first wave (P independent LDS reads per feature)
    // (The waves-per-unit count is a launch parameter, and written as `k < wpu` tests inside unrolled loops this section compiled
    // into a chain of scalar branches with ONE LDS read — and its wait — per branch: 0.96 us from the barrier to the unit's result in
    // BASELINE config 2, 1.32 us in the model's shape (tools/stamps_dmha.py).  The reads carry no dependence: one straight-line
    // copy per wpu in {1, 2, 4, 8} issues them all before anything waits.)
    const float* uml = red_ml + (size_t)uu * wpu * 2;
    float mwv[8], lwv[8];
    auto read_ml = [&](auto wpu_c) {
        constexpr int WPU = decltype(wpu_c)::value;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            mwv[k] = (k < WPU) ? uml[2 * k] : -INFINITY;
            lwv[k] = (k < WPU) ? uml[2 * k + 1] : 0.f;
        }
    };
    switch (wpu) {
        case 1: read_ml(std::integral_constant<int, 1>{}); break;
        case 2: read_ml(std::integral_constant<int, 2>{}); break;
        case 4: read_ml(std::integral_constant<int, 4>{}); break;
        case 8: read_ml(std::integral_constant<int, 8>{}); break;
        default:
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                mwv[k] = (k < wpu) ? uml[2 * k] : -INFINITY;
                lwv[k] = (k < wpu) ? uml[2 * k + 1] : 0.f;
            }
    }
    float M = -INFINITY;
#pragma unroll
    for (int k = 0; k < 8; ++k) M = fmaxf(M, mwv[k]);
    const float Mu = (M == -INFINITY) ? 0.f : M;
    float fw[8];
    float L = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { fw[k] = fast_exp2(mwv[k] - Mu); L = fmaf(lwv[k], fw[k], L); }
    const float invL = (L > 0.f) ? 1.0f / L : 0.f;
    float* cbuf = red_acc + (size_t)upw * P * DH;            // [upw][DH] the workgroup's context vectors
    float* ubuf = cbuf + (size_t)upw * DH;                   // [upw] their (masked) head scores
    const bool direct = p.nparts == H && p.arrivals != H;    // per-head partials published by the units' first waves themselves
    if (wu == 0) {
        float* cg = p.ctx != nullptr ? p.ctx + ((size_t)b * H + h) * DH : nullptr;
        const float* ua = red_acc + (size_t)uu * P * DH;
        float upart = 0.f;
#pragma unroll
        for (int kd = 0; kd < ND; ++kd) {
            const int d = lane + 64 * kd;
            if (d < DH) {
                float v = 0.f;
                auto sum_groups = [&](auto wpu_c) {
                    constexpr int WPU = decltype(wpu_c)::value;
                    float sg[WPU];
#pragma unroll
                    for (int k = 0; k < WPU; ++k) {
                        sg[k] = 0.f;
#pragma unroll
                        for (int g2 = 0; g2 < GPW; ++g2) sg[k] += ua[(size_t)(k * GPW + g2) * DH + d];
                    }
#pragma unroll
                    for (int k = 0; k < WPU; ++k) v = fmaf(sg[k], fw[k], v);
                };
                switch (wpu) {
                    case 1: sum_groups(std::integral_constant<int, 1>{}); break;
                    case 2: sum_groups(std::integral_constant<int, 2>{}); break;
                    case 4: sum_groups(std::integral_constant<int, 4>{}); break;
                    case 8: sum_groups(std::integral_constant<int, 8>{}); break;
                    default:
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            if (k < wpu) {
                                float sg = 0.f;
#pragma unroll
                                for (int g2 = 0; g2 < GPW; ++g2) sg += ua[(size_t)(k * GPW + g2) * DH + d];
                                v = fmaf(sg, fw[k], v);
                            }
                        }
                }
                const float cv = v * invL;
                if (cg != nullptr) cg[d] = cv;               // a plain output: nothing in this launch reads it back
                if (direct) store_agent(p.exch_p + ((size_t)b * H + h) * part_pitch(DH) + 4 + d, cv);   // V of this head's own partial
                else cbuf[(size_t)uu * DH + d] = cv;
                upart = fmaf(cv, attv[kd], upart);
            }
        }
        if (p.att != nullptr) {
            upart = wave_sum_all<1>(upart);
            const float um = (p.keep != nullptr && p.keep[(size_t)b * H + h] == 0) ? -INFINITY : upart;
            if (direct) {
                // one partial per HEAD {M = u_h, S = [not masked], V = c_h}: no fold over the workgroup's heads, no second barrier —
                // for small exchanges (BASELINE config 2: 8 x 68 floats per utterance) the fold's barrier + publish cost more than
                // the last arriver saves by pulling fewer partials
                if (lane == 0) {
                    float* pp = p.exch_p + ((size_t)b * H + h) * part_pitch(DH);
                    store_agent(pp, um);
                    store_agent(pp + 1, um == -INFINITY ? 0.f : 1.f);
                    store_agent(p.exch_u + (size_t)b * round_up_i(H, 4) + h, um);
                }
            } else if (lane == 0) ubuf[uu] = um;
        }
        if (p.ml != nullptr && lane == 0) {
            p.ml[((size_t)b * H + h) * 2] = M;
            p.ml[((size_t)b * H + h) * 2 + 1] = L;
        }
    }
    DASV_STAMP(p, 3);
    auto write_align = [&]() {
        if (p.align != nullptr) {
            const int tu = wu * 64 + lane, nthr_u = wpu * 64;
            float* ag = p.align + (size_t)b * T * H + h;
            for (int t = tu; t < T; t += nthr_u) ag[(size_t)t * H] = (t < len) ? fast_exp2(s_lds[t] - Mu) * invL : 0.f;
        }
    };
    if (p.att == nullptr) { write_align(); return; }
    if (direct) {
        // every publishing wave drains its write-through stores, the workgroup joins, ONE lane takes the ticket
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        DASV_STAMP(p, 4);
        int ticket = 0;
        if (tid == 0) ticket = __hip_atomic_fetch_add(p.counters + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        write_align();
        if (tid == 0) {
            const int last = (ticket == p.arrivals - 1);
            if (last) __hip_atomic_store(p.counters + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *flag = last;
        }
        __syncthreads();
        DASV_STAMP(p, 5);
        if (*flag != 0) head_stage<true>(p, b, smem);
        DASV_STAMP(p, 6);
        return;
    }
    __syncthreads();                                         // cbuf / ubuf of every unit are in LDS
    publish_and_finish<true>(p, b, a, cbuf, ubuf, smem, flag, write_align);
    DASV_STAMP(p, 6);
}

// ---- shape-agnostic kernel: any dh, any alignment, any stride ---------------------------------
// One 256-thread workgroup per unit; lane <-> frame for the scores, lane <-> feature for the sums.
// LDS: [t_pad] scores | [16] reductions | [dh] scaled query   (head stage aliases from 0) | flag at the end.
// Utterances whose scores do not fit (t_lds = 0) park them in the workspace instead: any length pools, like the
// reference (poolings.py:73-80 has no length limit).
__host__ __device__ inline size_t fwd_lds_generic_floats(int t_lds, int H, int dh) {
    const size_t a = (size_t)t_lds + 16 + dh, b = fwd_lds_head_floats(H, dh);
    return (a > b ? a : b) + 4;
}
constexpr size_t kGenericLdsCap = 160 * 1024;
__host__ inline bool generic_scores_fit_lds(int T, int H, int dh) {
    return fwd_lds_generic_floats(round_up_i(T, 4), H, dh) * sizeof(float) <= kGenericLdsCap;
}

template <typename XT>
__global__ void __launch_bounds__(256) dmha_fwd_generic(const FwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using E = Elem<XT>;
    int b, h;
    if (!block_to_unit(p.arrivals, p.B, b, h)) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nthr = blockDim.x, nw = nthr >> 6;
    const int H = p.H, dh = p.dh, T = p.T;
    int len = T;
    if (p.lengths != nullptr) len = max(0, min(T, p.lengths[b]));

    const int t_lds = p.s_glob != nullptr ? 0 : p.t_pad;
    float* s_lds = p.s_glob != nullptr ? p.s_glob + ((size_t)b * H + h) * p.t_pad : smem;   // [t_pad]
    float* red = smem + t_lds;                 // [16]
    float* qs = red + 16;                      // [dh]
    int* flag = reinterpret_cast<int*>(smem + fwd_lds_generic_floats(t_lds, H, dh) - 4);

    for (int d = tid; d < dh; d += nthr) qs[d] = p.query[(size_t)d * H + h] * p.qscale;
    __syncthreads();
    const long long base = (long long)b * p.sb + (long long)h * dh;
    float mloc = -INFINITY;
    for (int t = tid; t < len; t += nthr) {
        float s = 0.f;
        for (int d = 0; d < dh; ++d) s = fmaf(E::load1(p.x, base + (long long)t * p.st + d), qs[d], s);
        s_lds[t] = s;
        mloc = fmaxf(mloc, s);
    }
    mloc = group_max<64>(mloc);
    if (lane == 0) red[w] = mloc;
    __syncthreads();
    float M = -INFINITY;
    for (int k = 0; k < nw; ++k) M = fmaxf(M, red[k]);
    const float Mu = (M == -INFINITY) ? 0.f : M;
    float lloc = 0.f;
    for (int t = tid; t < len; t += nthr) lloc += fast_exp2(s_lds[t] - Mu);
    lloc = group_sum<64>(lloc);
    if (lane == 0) red[4 + w] = lloc;
    __syncthreads();
    float L = 0.f;
    for (int k = 0; k < nw; ++k) L += red[4 + k];
    const float invL = (L > 0.f) ? 1.0f / L : 0.f;
    float* cg = p.ctx != nullptr ? p.ctx + ((size_t)b * H + h) * dh : nullptr;
    float* cbuf = qs;          // the scaled query is no longer needed: [dh] context vector of this workgroup's one head
    float* ubuf = red + 12;    // its (masked) head score
    float upart = 0.f;
    for (int d = tid; d < dh; d += nthr) {
        float v = 0.f;
        for (int t = 0; t < len; ++t) v = fmaf(fast_exp2(s_lds[t] - Mu), E::load1(p.x, base + (long long)t * p.st + d), v);
        v *= invL;
        if (cg != nullptr) cg[d] = v;
        cbuf[d] = v;
        if (p.att != nullptr) upart = fmaf(v, p.att[d], upart);
    }
    if (p.att != nullptr) {   // head score <c_h, att>: waves -> LDS -> thread 0
        upart = group_sum<64>(upart);
        if (lane == 0) red[8 + w] = upart;
        __syncthreads();
        if (tid == 0) {
            float u = 0.f;
            for (int k = 0; k < nw; ++k) u += red[8 + k];
            ubuf[0] = (p.keep != nullptr && p.keep[(size_t)b * H + h] == 0) ? -INFINITY : u;
        }
    }
    if (p.ml != nullptr && tid == 0) {
        p.ml[((size_t)b * H + h) * 2] = M;
        p.ml[((size_t)b * H + h) * 2 + 1] = L;
    }
    auto write_align = [&]() {
        if (p.align != nullptr) {
            float* ag = p.align + (size_t)b * T * H + h;
            for (int t = tid; t < T; t += nthr) ag[(size_t)t * H] = (t < len) ? fast_exp2(s_lds[t] - Mu) * invL : 0.f;
        }
    };
    if (p.att == nullptr) { write_align(); return; }
    __syncthreads();           // cbuf / ubuf are in LDS
    if ((dh & 3) == 0) publish_and_finish<true>(p, b, h, cbuf, ubuf, smem, flag, write_align);
    else publish_and_finish<false>(p, b, h, cbuf, ubuf, smem, flag, write_align);
}

#ifdef DASV_DEBUG
// ---- read floor (measurement aid: DASV_DEBUG builds only, libdasv_hip_dbg.so) ---------------------
// The simplest kernel that reads `nvec` 16-byte vectors once: eight independent loads in flight per thread, grid-stride,
// one atomic per wave to keep the loads alive.  bench.py times it on the same rotating buffers as the fused kernel: what
// it needs is the time NO kernel that streams x once can beat on this chip at this size (tools/read_floor.hip swept the
// grid and depth; 2048 x 256 threads, depth 8 was the fastest on 39 MB).
__global__ void __launch_bounds__(256) read_floor_kernel(const float4* __restrict__ x, long long nvec, float* __restrict__ out) {
    constexpr int DEPTH = 8;
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (; i + (DEPTH - 1) * stride < nvec; i += DEPTH * stride) {
        float4 v[DEPTH];
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) v[k] = x[i + k * stride];
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) acc += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
    for (; i < nvec; i += stride) acc += (x[i].x + x[i].y) + (x[i].z + x[i].w);
    acc = group_sum<64>(acc);
    if ((threadIdx.x & 63) == 0) atomicAdd(out + blockIdx.x, acc);
}
#endif

// ---- host side --------------------------------------------------------------------------------
struct Tuning { int nw = 0, wpu = 0, lds_pad = 0, force_generic = 0, rows = 0; };
#ifdef DASV_STAMPS
static unsigned long long* g_stamps = nullptr;
#endif
#ifdef DASV_DEBUG
static Tuning g_tuning;            // sweeps only (dasv_debug_set_tuning)
#else
static constexpr Tuning g_tuning{};   // the product library has no knobs: every choice below is the automatic one
#endif
static thread_local char g_last_variant[96] = "none";

constexpr int rows_per_pass_for(int ni) { return ni == 1 ? 4 : (ni <= 3 ? 2 : 1); }

template <typename XT, int G, int NI, int R, bool PERM = false>
static hipError_t launch_vec(const FwdParams& p, dim3 grid, dim3 block, size_t lds, hipStream_t s) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dmha_fwd_vec<XT, G, NI, R, PERM>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((dmha_fwd_vec<XT, G, NI, R, PERM>), grid, block, lds, s, p);
    return hipGetLastError();
}

// NHWC frames: fp32 maps of the VGG4L front-end (F' = 5 segments per frame), 16 / 32 / 64 channels per head
static bool dispatch_vec_nhwc(int G, int NI, const FwdParams& p, dim3 grid, dim3 block, size_t lds, hipStream_t s, hipError_t& err) {
#define DASV_CASE(g) if (G == g && NI == 5) { err = launch_vec<F32, g, 5, 1, true>(p, grid, block, lds, s); return true; }
    DASV_CASE(4) DASV_CASE(8) DASV_CASE(16)
#undef DASV_CASE
    return false;
}

// (G, NI) -> kernel; R (rows per lane per pass) defaults per NI, a few alternatives exist for sweeps.
template <typename XT>
static bool dispatch_vec(int G, int NI, int R, const FwdParams& p, dim3 grid, dim3 block, size_t lds, hipStream_t s, hipError_t& err) {
#define DASV_CASE(g, n, r) if (G == g && NI == n && R == r) { err = launch_vec<XT, g, n, r>(p, grid, block, lds, s); return true; }
    DASV_CASE(16, 1, 4) DASV_CASE(16, 1, 2) DASV_CASE(16, 1, 8)
    DASV_CASE(16, 2, 2) DASV_CASE(16, 2, 4) DASV_CASE(16, 3, 2) DASV_CASE(16, 4, 1) DASV_CASE(16, 4, 2)
    DASV_CASE(16, 5, 1) DASV_CASE(16, 5, 2) DASV_CASE(16, 5, 3) DASV_CASE(16, 6, 1) DASV_CASE(16, 7, 1) DASV_CASE(16, 8, 1)
    DASV_CASE(8, 1, 4) DASV_CASE(8, 1, 8) DASV_CASE(8, 3, 2) DASV_CASE(8, 5, 1) DASV_CASE(8, 5, 2) DASV_CASE(8, 5, 3) DASV_CASE(8, 7, 1)
    DASV_CASE(4, 1, 4) DASV_CASE(4, 1, 8) DASV_CASE(4, 3, 2) DASV_CASE(4, 5, 1) DASV_CASE(4, 5, 2) DASV_CASE(4, 5, 3) DASV_CASE(4, 7, 1)
#undef DASV_CASE
    return false;
}

static int largest_divisor_leq(int n, int cap) {
    for (int d = cap; d >= 1; --d) if (n % d == 0) return d;
    return 1;
}

// Workspace layout: [kCounterSlots] arrival counters (zero between launches) | [B][round_up(H,4)] head scores | long-T scores.
// The counter region has a FIXED size, so where the counters live does not depend on B: a workspace that served a launch of
// one batch size can serve any other (callers cache one buffer per stream and reuse it for whatever fits).  With a
// B-dependent split, the head scores / raw scores written by a small-B launch sat where a later large-B launch expects
// zeroed counters: no last arriver was ever elected and those rows of `out` stayed uninitialised.  Batches beyond
// kCounterSlots utterances run as several launches (fwd_impl).
constexpr int kCounterSlots = 16384;
static size_t scores_workspace_bytes(int B, int H) {
    const int bc = B < kCounterSlots ? B : kCounterSlots;
    return (size_t)bc * round_up_i(H, 4) * sizeof(float);
}
// counters | [B][round_up(H,4)] head scores | [B][<= H][4 + round_up(dh,4)] per-workgroup partials (sized for one workgroup per head)
static size_t fixed_workspace_bytes(int B, int H, int dh) {
    const int bc = B < kCounterSlots ? B : kCounterSlots;
    return (size_t)kCounterSlots * sizeof(int) + scores_workspace_bytes(B, H) + (size_t)bc * H * part_pitch(dh) * sizeof(float);
}

template <typename XT>
static int fwd_impl(const void* x, const float* query, const float* att, const uint8_t* keep, const int32_t* lengths,
                    float* out, float* align, float* head_w, float* ctx, float* ml, void* workspace,
                    size_t workspace_bytes, int B, int T, int H, int dh, int64_t sb, int64_t st, hipStream_t stream,
                    int perm_f = 0, int64_t seg = 0) {
    using E = Elem<XT>;
    if (x == nullptr || query == nullptr || (ctx == nullptr && att == nullptr)) return DASV_E_NULL;
    if (att != nullptr && out == nullptr) return DASV_E_NULL;
    if (B <= 0 || T <= 0 || H <= 0 || dh <= 0) return DASV_E_SHAPE;
    if ((long long)H * dh > (1ll << 24) || (long long)B > (1ll << 24)) return DASV_E_SHAPE;
    if (st < (int64_t)H * dh || sb < 0) return DASV_E_STRIDE;
    if (att != nullptr) {
        if (workspace == nullptr || workspace_bytes < dasv_dmha_workspace_bytes(B, T, H, dh)) return DASV_E_WORKSPACE;
    }
    if (B > kCounterSlots) {   // one counter slot per utterance of a launch: larger batches go in slices (same stream: in order)
        for (int b0 = 0; b0 < B; b0 += kCounterSlots) {
            const int nb = B - b0 < kCounterSlots ? B - b0 : kCounterSlots;
            const int rc = fwd_impl<XT>(static_cast<const char*>(x) + (size_t)b0 * sb * E::kBytes, query, att,
                                        keep != nullptr ? keep + (size_t)b0 * H : nullptr, lengths != nullptr ? lengths + b0 : nullptr,
                                        out != nullptr ? out + (size_t)b0 * dh : nullptr,
                                        align != nullptr ? align + (size_t)b0 * T * H : nullptr,
                                        head_w != nullptr ? head_w + (size_t)b0 * H : nullptr,
                                        ctx != nullptr ? ctx + (size_t)b0 * H * dh : nullptr,
                                        ml != nullptr ? ml + (size_t)b0 * H * 2 : nullptr, workspace, workspace_bytes, nb, T, H, dh,
                                        sb, st, stream, perm_f, seg);
            if (rc != 0) return rc;
        }
        return 0;
    }

    FwdParams p;
    p.x = x; p.query = query; p.att = att; p.keep = keep; p.lengths = lengths;
    p.out = out; p.align = align; p.head_w = head_w; p.ctx = ctx; p.ml = ml;
    p.counters = static_cast<int*>(workspace);
    p.exch_u = reinterpret_cast<float*>(static_cast<char*>(workspace) + (size_t)kCounterSlots * sizeof(int));
    p.exch_p = reinterpret_cast<float*>(static_cast<char*>(workspace) + (size_t)kCounterSlots * sizeof(int) + scores_workspace_bytes(B, H));
    p.sb = sb; p.st = st; p.seg = seg; p.B = B; p.T = T; p.H = H; p.dh = dh;
    p.t_pad = round_up_i(T, 4);
    p.s_glob = nullptr;
    p.qscale = (float)(1.4426950408889634 / sqrt((double)H));
#ifdef DASV_STAMPS
    p.stamps = g_stamps;
#endif

    // fast path needs 16-byte-aligned rows
    const bool aligned = ((uintptr_t)x % 16 == 0) && ((sb * E::kBytes) % 16 == 0) && ((st * E::kBytes) % 16 == 0) &&
                         (dh % E::kPerVec == 0);
    int G = 0, NI = 0;
    if (perm_f > 0) {
        // NHWC frames: G lanes x 16 bytes = one head's channels of one segment, NI = segments per frame
        const int cph = dh / perm_f;
        if (!aligned || cph * perm_f != dh || cph % E::kPerVec != 0 || (seg * E::kBytes) % 16 != 0) return DASV_E_SHAPE;
        G = cph / E::kPerVec;
        NI = perm_f;
        if (!(G == 4 || G == 8 || G == 16) || NI != 5 || XT::kCode != 0) return DASV_E_SHAPE;
    } else if (aligned && !g_tuning.force_generic) {
        const int vecs = dh / E::kPerVec;
        G = 16;
        while (G > 1 && vecs % G != 0) G >>= 1;
        NI = vecs / G;
        if (G < 4 || NI > 8) G = 0;
    }
    const int groups8 = (B + 7) / 8 * 8;
    if (G != 0) {
        const int gpw = 64 / G;
        // waves per unit: enough row phases that one unit is a handful of passes, at most 8 waves
        int R = rows_per_pass_for(NI);
        // waves per unit: enough row phases that a unit is about five passes; workgroups of up to 8 waves
        // that carry several heads of one utterance (fewer arrivals), but never fewer workgroups than CUs.
        int wpu = g_tuning.wpu;
        if (wpu <= 0) {
            wpu = 1;
            while (wpu < 8 && wpu * gpw * R * 5 < T) wpu <<= 1;
        }
        if (wpu > 8) wpu = 8;
        int nw = g_tuning.nw > 0 ? g_tuning.nw : 8;
        if (nw > 8) nw = 8;
        if (nw < wpu) nw = wpu;
        int upw = largest_divisor_leq(H, nw / wpu);
        if (g_tuning.nw <= 0) {
            // never fewer workgroups than CUs; and once there are several rounds of workgroups per CU, smaller
            // workgroups (more of them resident) hide each other's finish latency better
            while (upw > 1 && (long long)groups8 * (H / upw) < 256) upw = largest_divisor_leq(H, upw - 1);
            while (upw * wpu > 4 && upw > 1 && (long long)groups8 * (H / upw) > 1024) upw = largest_divisor_leq(H, upw - 1);
        }
        nw = upw * wpu;
        p.wpu = wpu; p.upw = upw; p.arrivals = H / upw;
        // small exchanges: one partial per head, published by the units themselves (no fold); large ones (the model's 32 x 160
        // pooling): one folded partial per workgroup
        p.nparts = (upw > 1 && (size_t)H * part_pitch(dh) * sizeof(float) <= 4096) ? H : p.arrivals;
        const size_t lds = fwd_lds_floats(upw, p.t_pad, wpu, wpu * gpw, H, dh) * sizeof(float) + (size_t)g_tuning.lds_pad;
        if (lds <= 160 * 1024) {
            dim3 grid((unsigned)(groups8 * p.arrivals)), block((unsigned)(nw * 64));
            hipError_t err = hipSuccess;
            bool ok = false;
            if (perm_f > 0) {
                ok = dispatch_vec_nhwc(G, NI, p, grid, block, lds, stream, err);
            } else {
                ok = g_tuning.rows > 0 && dispatch_vec<XT>(G, NI, g_tuning.rows, p, grid, block, lds, stream, err);
                if (ok) R = g_tuning.rows;
                if (!ok) ok = dispatch_vec<XT>(G, NI, R, p, grid, block, lds, stream, err);
            }
            if (ok) {
                snprintf(g_last_variant, sizeof(g_last_variant), "vec<dtype=%d,G=%d,NI=%d,R=%d%s> nw=%d wpu=%d upw=%d lds=%zu",
                         XT::kCode, G, NI, R, perm_f > 0 ? ",nhwc" : "", nw, wpu, upw, lds);
                return (int)err;
            }
        }
    }
    if (perm_f > 0) return DASV_E_SHAPE;   // no shape-agnostic form of the NHWC addressing: the caller permutes instead
    // shape-agnostic path
    p.wpu = 4; p.upw = 1; p.arrivals = H; p.nparts = H;
    const bool in_lds = generic_scores_fit_lds(T, H, dh);
    if (!in_lds) {
        if (workspace == nullptr || workspace_bytes < dasv_dmha_workspace_bytes(B, T, H, dh)) return DASV_E_WORKSPACE;
        p.s_glob = reinterpret_cast<float*>(static_cast<char*>(workspace) + fixed_workspace_bytes(B, H, dh));
    }
    const size_t lds = fwd_lds_generic_floats(in_lds ? p.t_pad : 0, H, dh) * sizeof(float);
    if (lds > kGenericLdsCap) return DASV_E_SHAPE;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dmha_fwd_generic<XT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    dim3 grid((unsigned)(groups8 * H)), block(256);
    hipLaunchKernelGGL((dmha_fwd_generic<XT>), grid, block, lds, stream, p);
    snprintf(g_last_variant, sizeof(g_last_variant), "generic<dtype=%d> lds=%zu", XT::kCode, lds);
    return (int)hipGetLastError();
}

}  // namespace dasv

// ---- C ABI ------------------------------------------------------------------------------------
extern "C" {

int dasv_abi_version(void) { return DASV_ABI_VERSION; }

const char* dasv_error_string(int code) {
    switch (code) {
        case DASV_OK: return "ok";
        case DASV_E_NULL: return "required pointer is NULL";
        case DASV_E_SHAPE: return "shape out of range";
        case DASV_E_WORKSPACE: return "workspace missing or too small";
        case DASV_E_STRIDE: return "invalid stride";
        case DASV_E_DTYPE: return "unknown element type";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}

size_t dasv_dmha_workspace_bytes(int B, int T, int H, int dh) {
    if (B <= 0 || H <= 0) return 0;
    // [kCounterSlots] arrival counters (must be zero between launches; fixed size, see fixed_workspace_bytes) | head scores
    // | per-workgroup partials | only for utterances too long for the LDS score row: [B*H][round_up(T,4)] raw scores
    if (dh <= 0) return 0;
    size_t n = dasv::fixed_workspace_bytes(B, H, dh);
    const int bc = B < dasv::kCounterSlots ? B : dasv::kCounterSlots;
    if (T > 0 && dh > 0 && !dasv::generic_scores_fit_lds(T, H, dh)) n += (size_t)bc * H * dasv::round_up_i(T, 4) * sizeof(float);
    return n;
}

int dasv_dmha_workspace_init(void* workspace, size_t workspace_bytes, dasv_stream_t stream) {
    if (workspace == nullptr) return DASV_E_NULL;
    return (int)hipMemsetAsync(workspace, 0, workspace_bytes, (hipStream_t)stream);
}

int dasv_dmha_fwd(int x_dtype, const void* x, const float* query, const float* att, const uint8_t* head_keep,
                  const int32_t* lengths, float* out, float* align, float* head_w, float* ctx, float* ml,
                  void* workspace, size_t workspace_bytes, int B, int T, int H, int dh, int64_t x_stride_b,
                  int64_t x_stride_t, dasv_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    switch (x_dtype) {
        case DASV_F32:
            return dasv::fwd_impl<dasv::F32>(x, query, att, head_keep, lengths, out, align, head_w, ctx, ml, workspace,
                                             workspace_bytes, B, T, H, dh, x_stride_b, x_stride_t, s);
        case DASV_BF16:
            return dasv::fwd_impl<dasv::BF16>(x, query, att, head_keep, lengths, out, align, head_w, ctx, ml, workspace,
                                              workspace_bytes, B, T, H, dh, x_stride_b, x_stride_t, s);
        case DASV_F16:
            return dasv::fwd_impl<dasv::F16>(x, query, att, head_keep, lengths, out, align, head_w, ctx, ml, workspace,
                                             workspace_bytes, B, T, H, dh, x_stride_b, x_stride_t, s);
        default:
            return DASV_E_DTYPE;
    }
}

#define DASV_TYPED(name, code, xtype)                                                                               \
    int name(xtype x, const float* query, const float* att, const uint8_t* head_keep, const int32_t* lengths,      \
             float* out, float* align, float* head_w, float* ctx, float* ml, void* workspace,                      \
             size_t workspace_bytes, int B, int T, int H, int dh, int64_t x_stride_b, int64_t x_stride_t,          \
             dasv_stream_t stream) {                                                                                \
        return dasv_dmha_fwd(code, x, query, att, head_keep, lengths, out, align, head_w, ctx, ml, workspace,      \
                             workspace_bytes, B, T, H, dh, x_stride_b, x_stride_t, stream);                        \
    }
DASV_TYPED(dasv_dmha_fwd_f32, DASV_F32, const float*)
DASV_TYPED(dasv_dmha_fwd_bf16, DASV_BF16, const void*)
DASV_TYPED(dasv_dmha_fwd_f16, DASV_F16, const void*)
#undef DASV_TYPED

int dasv_dmha_nhwc_supported(int F, int C, int H) {
    if (F != 5 || C <= 0 || H <= 0 || C % H != 0) return 0;
    const int cph = C / H;
    return cph == 16 || cph == 32 || cph == 64;
}

int dasv_dmha_fwd_nhwc_f32(const float* x, const float* query, const float* att, const uint8_t* head_keep,
                           const int32_t* lengths, float* out, float* align, float* head_w, float* ctx, float* ml,
                           void* workspace, size_t workspace_bytes, int B, int T, int F, int C, int H, dasv_stream_t stream) {
    if (!dasv_dmha_nhwc_supported(F, C, H)) return DASV_E_SHAPE;
    const int dh = C / H * F;
    return dasv::fwd_impl<dasv::F32>(x, query, att, head_keep, lengths, out, align, head_w, ctx, ml, workspace, workspace_bytes,
                                     B, T, H, dh, (int64_t)T * F * C, (int64_t)F * C, (hipStream_t)stream, F, (int64_t)C);
}

const char* dasv_dmha_last_variant(void) { return dasv::g_last_variant; }

#ifdef DASV_DEBUG
void dasv_debug_set_tuning(int waves_per_group, int waves_per_head, int lds_pad_bytes, int force_generic,
                           int rows_per_pass) {
    dasv::g_tuning.rows = rows_per_pass;
    dasv::g_tuning.nw = waves_per_group;
    dasv::g_tuning.wpu = waves_per_head;
    dasv::g_tuning.lds_pad = lds_pad_bytes;
    dasv::g_tuning.force_generic = force_generic;
}

int dasv_debug_read_floor(const void* x, size_t bytes, float* scratch, dasv_stream_t stream) {
    if (x == nullptr || scratch == nullptr) return DASV_E_NULL;
    if (bytes < 16 || (uintptr_t)x % 16 != 0) return DASV_E_SHAPE;
    hipLaunchKernelGGL(dasv::read_floor_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const float4*>(x), (long long)(bytes / 16), scratch);
    return (int)hipGetLastError();
}

#endif  // DASV_DEBUG

#ifdef DASV_STAMPS
void dasv_debug_set_stamps(void* p) { dasv::g_stamps = static_cast<unsigned long long*>(p); }
#endif

}  // extern "C"
