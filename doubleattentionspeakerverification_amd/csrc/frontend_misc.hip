// Memory-bound pieces of the VGG front-end on NHWC fp32 activations (inference path):
//   conv3x3 with ONE input channel + bias + ReLU   (conv11 of CNNs.py:38 / 72: `F.relu(self.conv11(x))`)
//   2x2 stride-2 ceil-mode max-pool                (CNNs.py:40,44,48 / 74,78,82,86: `F.max_pool2d(x, 2, stride=2, ceil_mode=True)`)
// Both are pure HBM streams (conv11 writes 786 MB for a [64,300,80] batch and computes 9 FMAs per output), so the
// layout is the whole design: a lane owns four consecutive channels of one pixel, 16-byte loads/stores, channel-fastest.
#include "dasv_common.hpp"
#include "../../include/dasv_hip.h"

namespace dasv {

typedef float f4 __attribute__((ext_vector_type(4)));

// x [B,H,W] (one channel), w [Cout,1,3,3], y [B,H,W,Cout]
// x_rec / y_rec: range records (dasv_common.hpp).  y_rec[0] receives max |y|.  pack: y is written PACKED, scaled by 2^E with E from
// the bound |y| <= amax_x * sum_t |w[c][t]| + |b[c]| (nine taps: the bound is within a few binades of the true maximum); every
// workgroup holds all Cout channels' weights, so every workgroup derives the same E; block 0 leaves it in y_rec[1].
__global__ void __launch_bounds__(256) conv3x3_c1_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y,
                                                         int H, int W, int Cout, long long npix, int relu,
                                                         const int* __restrict__ row_span, int pack,
                                                         const uint32_t* __restrict__ x_rec, uint32_t* __restrict__ y_rec) {
    // A block walks chunks of C1_CHUNK consecutive pixels.  The one-channel input of a chunk (its pixels + W+1 of halo on each
    // side, one contiguous run) is staged in LDS once; every lane then takes its nine taps from LDS.  (Round 1 had each of the
    // Cout/4 lanes of a pixel issue the nine global loads itself: nine load instructions per 16-byte store kept the kernel at
    // 2.8 TB/s of its 786 MB write for a [64,300,80] batch.)
    extern __shared__ float c1_in[];                // [C1_CHUNK + 2 W + 2]
    constexpr int C1_CHUNK = 512;
    const int quads = Cout >> 2;                 // lanes per pixel
    const int ppb = blockDim.x / quads;          // pixels per block iteration
    const int q = threadIdx.x % quads, pl = threadIdx.x / quads;
    float wr[4][9];
    f4 bv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int t = 0; t < 9; ++t) wr[c][t] = w[(size_t)(q * 4 + c) * 9 + t];
        if (bias != nullptr) bv[c] = bias[q * 4 + c];
    }
    float sy = 1.f;
    if (pack == 1 || pack == 2) {
        __shared__ float c1_red[4];
        const float ax = __uint_as_float(x_rec[0] & 0x7fffffffu);
        float bnd = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float l1 = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) l1 += fabsf(wr[c][t]);
            const float bc = ax * l1 + fabsf(bv[c]);
            bnd = !(bc <= bnd) ? bc : bnd;          // keeps a NaN
        }
        uint32_t bb = __float_as_uint(bnd);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)bb, off); bb = o > bb ? o : bb; }
        if ((threadIdx.x & 63) == 0) c1_red[threadIdx.x >> 6] = __uint_as_float(bb);
        __syncthreads();
        uint32_t m = 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const uint32_t o = __float_as_uint(c1_red[k]); m = o > m ? o : m; }
        int Ey = x8_exp_for(__float_as_uint(__uint_as_float(m) * 1.001f));
        if (pack == 2) Ey = Ey + 7 > 126 ? 126 : Ey + 7;      // a half-precision tensor: the bound goes to [2^14, 2^15) (fp16 holds 65504)
        sy = x8_pow2(Ey);
        if (blockIdx.x == 0 && threadIdx.x == 0) y_rec[1] = (uint32_t)Ey;
    }
    uint32_t amax_u = 0u;
    __shared__ uint32_t s_amax;
    if (threadIdx.x == 0) s_amax = 0u;      // visible behind the barriers of the chunk loop (every block runs at least one chunk)
    const int halo = W + 1, span = C1_CHUNK + 2 * halo;
    // Per chunk: the raw input run goes to LDS, then ONE thread per pixel gathers the pixel's nine taps — border tests, the two
    // integer divisions that locate the pixel in its image, zeros outside — into a row of nine floats; the Cout/4 lanes of a pixel
    // then read that row (a broadcast) and do nothing but 36 FMAs.  (Round 2 / early round 3 had every lane of every pixel redo
    // the divisions and the nine border selects: ~100 of its ~140 vector instructions per output quad, and the kernel sat at
    // 167-185 us for 786 / 393 MB whatever the output format.)
    float* c1_cols = c1_in + ((span + 3) & ~3);      // [C1_CHUNK][9]
    for (long long p0 = (long long)blockIdx.x * C1_CHUNK; p0 < npix; p0 += (long long)gridDim.x * C1_CHUNK) {
        __syncthreads();                             // the previous chunk's readers are done
        for (int i = threadIdx.x; i < span; i += blockDim.x) {
            const long long g = p0 - halo + i;
            c1_in[i] = (g >= 0 && g < npix) ? x[g] : 0.f;
        }
        __syncthreads();
        const int nloc = (int)((npix - p0) < C1_CHUNK ? (npix - p0) : C1_CHUNK);
        for (int l = threadIdx.x; l < nloc; l += blockDim.x) {
            const long long pix = p0 + l;
            const int w_ = (int)(pix % W);
            int h_ = (int)((pix / W) % H), hend = H;
            if (row_span != nullptr) {   // ragged batch: image borders from the row table
                const int r = (int)(pix / W), lo = row_span[2 * r];
                h_ = r - lo;
                hend = row_span[2 * r + 1] - lo;
            }
            const float* ctr = c1_in + l + halo;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int hh = h_ + t / 3 - 1, ww = w_ + t % 3 - 1;
                c1_cols[l * 9 + t] = ((unsigned)hh < (unsigned)hend && (unsigned)ww < (unsigned)W) ? ctr[(t / 3 - 1) * W + (t % 3 - 1)] : 0.f;
            }
        }
        __syncthreads();
        if (pl < ppb) {
            for (int l = pl; l < nloc; l += ppb) {
                const long long pix = p0 + l;
                float in[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) in[t] = c1_cols[l * 9 + t];
                f4 o = bv;
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int t = 0; t < 9; ++t) o[c] = fmaf(in[t], wr[c][t], o[c]);
                if (relu) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) o[c] = relu_nan(o[c]);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) amax_u = max(amax_u, __float_as_uint(o[c]) & 0x7fffffffu);
                // PACKED output for a split8 consumer (dasv_common.hpp) / half-precision output for the fp16 chain
                if (pack == 2) {
                    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    const h2 a = __builtin_convertvector(f2{o[0] * sy, o[1] * sy}, h2), b2 = __builtin_convertvector(f2{o[2] * sy, o[3] * sy}, h2);
                    *reinterpret_cast<uint2*>(reinterpret_cast<char*>(y) + ((size_t)pix * Cout + q * 4) * 2) =
                        make_uint2(__builtin_bit_cast(uint32_t, a), __builtin_bit_cast(uint32_t, b2));
                } else if (pack == 3) {       // bf16 storage chain (bf16-operand mode): round to nearest even, no scale
                    typedef __bf16 b2t __attribute__((ext_vector_type(2)));
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    const b2t a = __builtin_convertvector(f2{o[0], o[1]}, b2t), b2 = __builtin_convertvector(f2{o[2], o[3]}, b2t);
                    *reinterpret_cast<uint2*>(reinterpret_cast<char*>(y) + ((size_t)pix * Cout + q * 4) * 2) =
                        make_uint2(__builtin_bit_cast(uint32_t, a), __builtin_bit_cast(uint32_t, b2));
                } else if (pack) store_x8(y, pix, Cout, q * 4, o, sy);
                else *reinterpret_cast<f4*>(y + (size_t)pix * Cout + q * 4) = o;
            }
        }
    }
    if (y_rec != nullptr) range_note_block(y_rec, amax_u, &s_amax);
}

// max |x| over n floats into rec[0] (atomic maximum of the sign-cleared bits; rec zeroed by the caller): the range record of a
// tensor that no kernel of this library produced (the input features, a caller's activation tensor)
__global__ void __launch_bounds__(256) range_absmax_kernel(const float* __restrict__ x, long long n, uint32_t* __restrict__ rec) {
    uint32_t m = 0u;
    __shared__ uint32_t s_amax;
    if (threadIdx.x == 0) s_amax = 0u;
    __syncthreads();
    const long long nv = n >> 2, stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
        const f4 v = *reinterpret_cast<const f4*>(x + i * 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) m = max(m, __float_as_uint(v[c]) & 0x7fffffffu);
    }
    if (blockIdx.x == 0 && (long long)threadIdx.x < n - nv * 4) m = max(m, __float_as_uint(x[nv * 4 + threadIdx.x]) & 0x7fffffffu);
    range_note_block(rec, m, &s_amax);
}

// element i of a pooled map's [windows][C / 4] index space -> channel quad q, window op = (b, ho, wo).  The five 64-bit divisions this
// used to be cost more VALU time than the window's 16 bytes of traffic (a 64-bit division is ~150 instructions; the kernels ran at
// 2.0-2.6 TB/s): 32-bit arithmetic whenever the index space fits (always, in practice), a shift for the usual power-of-two C.
__device__ __forceinline__ void pool_window_of(long long i, int cq, int Wo, int Ho, long long nvec, int& q, long long& op, int& wo, int& ho,
                                               long long& b) {
    if (nvec < (1ll << 31)) {
        const uint32_t ii = (uint32_t)i;
        uint32_t o;
        if ((cq & (cq - 1)) == 0) { o = ii >> (31 - __builtin_clz((uint32_t)cq)); q = (int)(ii & (uint32_t)(cq - 1)); }
        else { o = ii / (uint32_t)cq; q = (int)(ii - o * (uint32_t)cq); }
        const uint32_t r = o / (uint32_t)Wo;
        wo = (int)(o - r * (uint32_t)Wo);
        const uint32_t bb = r / (uint32_t)Ho;
        ho = (int)(r - bb * (uint32_t)Ho);
        op = (long long)o;
        b = (long long)bb;
        return;
    }
    q = (int)(i % cq);
    op = i / cq;
    wo = (int)(op % Wo);
    ho = (int)((op / Wo) % Ho);
    b = op / ((long long)Wo * Ho);
}

// x [B,H,W,C] -> y [B,Ho,Wo,C], Ho = ceil(H/2), Wo = ceil(W/2); windows clipped at the bottom/right edge (ceil_mode)
// ragged batch (pool_src != null): B == 1, Ho = total output rows; pool_src[ro] = {first input row, 1 if the window has a second row}
// x_rec / y_rec (optional; both required with pack): the pooled map's range record is its input's — every input element lies in a
// window, so the maximum is the same (for signed inputs max |y| <= max |x|: still a bound) — and a packed output is scaled by it
__global__ void __launch_bounds__(256) maxpool2x2_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                              int H, int W, int C, int Ho, int Wo, long long nvec,
                                                              const int* __restrict__ pool_src, int pack,
                                                              const uint32_t* __restrict__ x_rec, uint32_t* __restrict__ y_rec, int in_f16,
                                                              uint8_t* __restrict__ idx) {
    // idx (optional, fp32 input only): [outputs][C] bytes {arg-max position 0-3 | 4 if the maximum is > 0} for the backward pass
    // (training): it then needs neither x nor a second look at the window
    // in_f16: x is a half-precision tensor holding v * 2^Ex (fp16 activation chain).  pack 2: y is one too, with the SAME exponent
    // (the maximum commutes with a positive scale: the halves are compared and copied, nothing is rounded); pack 0 with in_f16:
    // fp32 out, unscaled.
    const int cq = C >> 2;
    float sy = 1.f;
    if (x_rec != nullptr && in_f16 != 2) {
        const uint32_t xa = x_rec[0];
        int Ey = x8_exp_for(xa);
        if (in_f16) Ey = (int)x_rec[1];
        sy = in_f16 ? x8_pow2(-Ey) : x8_pow2(Ey);          // fp16 in: the factor that UNDOES the input's scale (fp32 out)
        if (y_rec != nullptr && blockIdx.x == 0 && threadIdx.x == 0) { y_rec[0] = xa; y_rec[1] = (uint32_t)Ey; }
    }
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
        int q, wo, ho;
        long long op, b;
        pool_window_of(i, cq, Wo, Ho, nvec, q, op, wo, ho, b);
        int h0 = 2 * ho;
        const int w0 = 2 * wo;
        bool hh = h0 + 1 < H;
        if (pool_src != nullptr) { h0 = pool_src[2 * ho]; hh = pool_src[2 * ho + 1] != 0; }
        if (in_f16 == 2) {
            // bf16 tensors (the bf16 storage chain): the values ARE the top halves of floats — widen, compare, narrow (exact)
            const char* hb = reinterpret_cast<const char*>(x) + ((((size_t)b * H + h0) * W + w0) * C + q * 4) * 2;
            auto ld = [&](size_t off) {
                const uint2 u = *reinterpret_cast<const uint2*>(hb + off * 2);
                return f4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
            };
            auto mxh = [](f4 a, f4 v) {
#pragma unroll
                for (int c = 0; c < 4; ++c) a[c] = (v[c] > a[c] || v[c] != v[c]) ? v[c] : a[c];
                return a;
            };
            f4 m = ld(0);
            const bool hw2 = w0 + 1 < W;
            if (hw2) m = mxh(m, ld(C));
            if (hh) m = mxh(m, ld((size_t)W * C));
            if (hw2 && hh) m = mxh(m, ld((size_t)W * C + C));
            if (pack == 3) {
                *reinterpret_cast<uint2*>(reinterpret_cast<char*>(y) + ((size_t)op * C + q * 4) * 2) =
                    make_uint2((__float_as_uint(m[0]) >> 16) | (__float_as_uint(m[1]) & 0xffff0000u),
                               (__float_as_uint(m[2]) >> 16) | (__float_as_uint(m[3]) & 0xffff0000u));
            } else {
                *reinterpret_cast<f4*>(y + (size_t)op * C + q * 4) = m;
            }
            continue;
        }
        if (in_f16) {
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            const char* hb = reinterpret_cast<const char*>(x) + ((((size_t)b * H + h0) * W + w0) * C + q * 4) * 2;
            auto ld = [&](size_t off) { return __builtin_convertvector(*reinterpret_cast<const h4*>(hb + off * 2), f4); };
            auto mxh = [](f4 a, f4 v) {
#pragma unroll
                for (int c = 0; c < 4; ++c) a[c] = (v[c] > a[c] || v[c] != v[c]) ? v[c] : a[c];
                return a;
            };
            f4 m = ld(0);
            const bool hw2 = w0 + 1 < W;
            if (hw2) m = mxh(m, ld(C));
            if (hh) m = mxh(m, ld((size_t)W * C));
            if (hw2 && hh) m = mxh(m, ld((size_t)W * C + C));
            if (pack == 2) {
                *reinterpret_cast<h4*>(reinterpret_cast<char*>(y) + ((size_t)op * C + q * 4) * 2) = __builtin_convertvector(m, h4);
            } else {
                *reinterpret_cast<f4*>(y + (size_t)op * C + q * 4) = m * sy;
            }
            continue;
        }
        const float* base = x + (((size_t)b * H + h0) * W + w0) * C + q * 4;
        f4 m = *reinterpret_cast<const f4*>(base);
        const bool hw = w0 + 1 < W;
        // a NaN in the window comes out as NaN, as in F.max_pool2d (v_max_f32 would return the other operand)
        auto mx = [](f4 a, f4 v) {
#pragma unroll
            for (int c = 0; c < 4; ++c) a[c] = (v[c] > a[c] || v[c] != v[c]) ? v[c] : a[c];
            return a;
        };
        if (idx != nullptr) {
            // row-major scan, strict '>' — the first maximum wins, as ATen's kernel does (and a NaN takes over)
            int arg[4] = {0, 0, 0, 0};
            auto mxa = [&](f4 v, int pos) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (v[c] > m[c] || v[c] != v[c]) { m[c] = v[c]; arg[c] = pos; }
            };
            if (hw) mxa(*reinterpret_cast<const f4*>(base + C), 1);
            if (hh) mxa(*reinterpret_cast<const f4*>(base + (size_t)W * C), 2);
            if (hw && hh) mxa(*reinterpret_cast<const f4*>(base + (size_t)W * C + C), 3);
            uint32_t code = 0u;
#pragma unroll
            for (int c = 0; c < 4; ++c) code |= (uint32_t)(arg[c] | (m[c] > 0.f ? 4 : 0)) << (8 * c);
            *reinterpret_cast<uint32_t*>(idx + (size_t)op * C + q * 4) = code;
        } else {
            if (hw) m = mx(m, *reinterpret_cast<const f4*>(base + C));
            if (hh) m = mx(m, *reinterpret_cast<const f4*>(base + (size_t)W * C));
            if (hw && hh) m = mx(m, *reinterpret_cast<const f4*>(base + (size_t)W * C + C));
        }
        if (pack) store_x8(y, op, C, q * 4, m, sy);
        else *reinterpret_cast<f4*>(y + (size_t)op * C + q * 4) = m;
    }
}

// backward of the 2x2 ceil-mode max-pool: every input pixel belongs to exactly one window, so the thread that owns a
// window recomputes its arg-max (row-major scan, strict '>' — the first maximum wins, as ATen's kernel does) and writes
// all of the window's input gradients: gy to the arg-max, zero elsewhere.  No indices tensor, no atomics, no memset.
__global__ void __launch_bounds__(256) maxpool2x2_bwd_nhwc_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                  float* __restrict__ gx, int H, int W, int C, int Ho, int Wo,
                                                                  long long nvec, int relu_mask, float* __restrict__ colsum,
                                                                  const uint8_t* __restrict__ idx, const int* __restrict__ pool_src,
                                                                  const uint32_t* __restrict__ gy_rec, uint32_t* __restrict__ gx_rec) {
    // gy_rec / gx_rec (both or neither): gx is written as a PACKED tensor (dasv_common.hpp, store_x8) scaled by the
    // exponent of gy's recorded maximum — the scattered gradient has the same maximum — and gx_rec receives {that maximum, the exponent}:
    // the data-gradient and weight-gradient launches that consume gx then copy its pieces instead of converting fp32 values
    // pool_src (ragged batch: B == 1, H = total input rows, Ho = total output rows): {first input row, has-second-row} per output row
    // idx (optional): the forward pass's {arg-max | 4 if maximum > 0} bytes — x is then not read (x may be null)
    // colsum (optional): [gridDim.x][C] per-channel sums of the gradient this workgroup scattered (= the bias gradient of
    // the conv whose ReLU output was pooled).  Needs (C/4) | 256 so that a thread keeps the same four channels throughout.
    __shared__ f4 red[256];
    f4 csum = {0.f, 0.f, 0.f, 0.f};
    const int cq = C >> 2;
    float sp = 1.f;
    if (gx_rec != nullptr) {
        const uint32_t ga = gy_rec[0];
        const int Eg = x8_exp_for(ga);
        sp = x8_pow2(Eg);
        if (blockIdx.x == 0 && threadIdx.x == 0) { gx_rec[0] = ga; gx_rec[1] = (uint32_t)Eg; }
    }
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
        int q, wo, ho;
        long long op, b;
        pool_window_of(i, cq, Wo, Ho, nvec, q, op, wo, ho, b);
        int h0 = 2 * ho;
        const int w0 = 2 * wo;
        bool hh = h0 + 1 < H;
        if (pool_src != nullptr) { h0 = pool_src[2 * ho]; hh = pool_src[2 * ho + 1] != 0; }
        const size_t base = (((size_t)b * H + h0) * W + w0) * C + q * 4;
        const bool hw = w0 + 1 < W;
        const size_t o1 = base + C, o2 = base + (size_t)W * C, o3 = o2 + C;
        f4 g = *reinterpret_cast<const f4*>(gy + (size_t)op * C + q * 4);
        int arg[4] = {0, 0, 0, 0};
        f4 m = {1.f, 1.f, 1.f, 1.f};
        if (idx != nullptr) {
            const uint32_t code = *reinterpret_cast<const uint32_t*>(idx + (size_t)op * C + q * 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                arg[c] = (int)((code >> (8 * c)) & 3u);
                m[c] = ((code >> (8 * c)) & 4u) ? 1.f : 0.f;
            }
        } else {
        const f4 v0 = *reinterpret_cast<const f4*>(x + base);
        m = v0;
        f4 v1, v2, v3;
        if (hw) { v1 = *reinterpret_cast<const f4*>(x + o1);
#pragma unroll
            for (int c = 0; c < 4; ++c) if (v1[c] > m[c] || v1[c] != v1[c]) { m[c] = v1[c]; arg[c] = 1; } }
        if (hh) { v2 = *reinterpret_cast<const f4*>(x + o2);
#pragma unroll
            for (int c = 0; c < 4; ++c) if (v2[c] > m[c] || v2[c] != v2[c]) { m[c] = v2[c]; arg[c] = 2; } }
        if (hw && hh) { v3 = *reinterpret_cast<const f4*>(x + o3);
#pragma unroll
            for (int c = 0; c < 4; ++c) if (v3[c] > m[c] || v3[c] != v3[c]) { m[c] = v3[c]; arg[c] = 3; } }
        }
        f4 r0, r1, r2, r3;
        if (relu_mask) {   // x is a ReLU's output: pass the gradient through that ReLU too (zero where the maximum is not > 0)
#pragma unroll
            for (int c = 0; c < 4; ++c) g[c] = m[c] > 0.f ? g[c] : 0.f;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            r0[c] = arg[c] == 0 ? g[c] : 0.f; r1[c] = arg[c] == 1 ? g[c] : 0.f;
            r2[c] = arg[c] == 2 ? g[c] : 0.f; r3[c] = arg[c] == 3 ? g[c] : 0.f;
        }
        if (gx_rec != nullptr) {
            const long long p0 = ((long long)b * H + h0) * W + w0;
            store_x8(gx, p0, C, q * 4, r0, sp);
            if (hw) store_x8(gx, p0 + 1, C, q * 4, r1, sp);
            if (hh) store_x8(gx, p0 + W, C, q * 4, r2, sp);
            if (hw && hh) store_x8(gx, p0 + W + 1, C, q * 4, r3, sp);
        } else {
            *reinterpret_cast<f4*>(gx + base) = r0;
            if (hw) *reinterpret_cast<f4*>(gx + o1) = r1;
            if (hh) *reinterpret_cast<f4*>(gx + o2) = r2;
            if (hw && hh) *reinterpret_cast<f4*>(gx + o3) = r3;
        }
        csum += g;   // exactly one of the window's inputs received g
    }
    if (colsum != nullptr) {
        red[threadIdx.x] = csum;
        __syncthreads();
        if ((int)threadIdx.x < cq) {     // threads t, t + cq, t + 2 cq, ... hold the same channels: fixed-order sum
            f4 t = red[threadIdx.x];
            for (int k = threadIdx.x + cq; k < 256; k += cq) t += red[k];
            *reinterpret_cast<f4*>(colsum + (size_t)blockIdx.x * C + threadIdx.x * 4) = t;
        }
    }
}

// backward of the one-input-channel conv3x3 + bias + ReLU: gw[n][tap] = sum_pix g[pix][n] * x[pix + tap], gb[n] = sum g,
// with g = gy * (y > 0).  A pure HBM read stream of gy and y: each workgroup reduces a pixel range into 10 values per
// channel (registers -> LDS -> one partial row), a second launch adds the rows in a fixed order.
// part: [nblk][Cout][10]
__global__ void __launch_bounds__(256) conv3x3_c1_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                             const float* __restrict__ gy, float* __restrict__ part,
                                                             int H, int W, int Cout, long long npix, int pix_per_block,
                                                             const int* __restrict__ row_span) {
    // row_span (ragged batch: one tall image of H = total rows): {first row, one past the last row} of the utterance that owns each
    // row — a tap's source row must lie inside it
    extern __shared__ float red[];   // [ppb][Cout][10]
    const int quads = Cout >> 2, ppb = blockDim.x / quads;
    const int q = threadIdx.x % quads, pl = threadIdx.x / quads;
    float acc[4][10];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < 10; ++t) acc[c][t] = 0.f;
    const long long p0 = (long long)blockIdx.x * pix_per_block;
    const long long p1 = p0 + pix_per_block < npix ? p0 + pix_per_block : npix;
    if (pl < ppb) {
        // (the pixel's image coordinates are carried from one pixel to the next instead of two 64-bit divisions per 16 bytes of gradient)
        int w_ = 0, h_ = 0;
        if (p0 + pl < p1) { w_ = (int)((p0 + pl) % W); h_ = (int)(((p0 + pl) / W) % H); }
        const int dh_ = ppb / W, dw_ = ppb - dh_ * W;          // one step of ppb pixels = dh_ rows + dw_ columns
        // Four pixels per trip, the gradients of the NEXT trip requested before this one is computed: with one 16-byte load in flight
        // per thread and 12 waves per CU (the reduction's LDS) the launch was bound by memory latency, not bandwidth — 274 us
        // whether or not y was read.
        constexpr int U = 4;
        auto loadg = [&](long long pix) {
            f4 g = {0.f, 0.f, 0.f, 0.f};
            if (pix < p1) {
                g = *reinterpret_cast<const f4*>(gy + (size_t)pix * Cout + q * 4);
                if (y != nullptr) {      // (null: the producer of gy applied this layer's ReLU mask already — half the traffic)
                    const f4 yy = *reinterpret_cast<const f4*>(y + (size_t)pix * Cout + q * 4);
#pragma unroll
                    for (int c = 0; c < 4; ++c) g[c] = yy[c] > 0.f ? g[c] : 0.f;
                }
            }
            return g;
        };
        f4 gnext[U];
#pragma unroll
        for (int u = 0; u < U; ++u) gnext[u] = loadg(p0 + pl + (long long)u * ppb);
        for (long long pix = p0 + pl; pix < p1; pix += (long long)U * ppb) {
            f4 gcur[U];
#pragma unroll
            for (int u = 0; u < U; ++u) gcur[u] = gnext[u];
#pragma unroll
            for (int u = 0; u < U; ++u) gnext[u] = loadg(pix + (long long)(U + u) * ppb);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long pu = pix + (long long)u * ppb;
                if (pu < p1) {
                    float in[9];
                    int hlo = 0, hhi = H;
                    if (row_span != nullptr) { hlo = row_span[2 * h_]; hhi = row_span[2 * h_ + 1]; }
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const int hh = h_ + t / 3 - 1, ww = w_ + t % 3 - 1;
                        in[t] = (hh >= hlo && hh < hhi && (unsigned)ww < (unsigned)W) ? x[pu + (t / 3 - 1) * W + (t % 3 - 1)] : 0.f;
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
#pragma unroll
                        for (int t = 0; t < 9; ++t) acc[c][t] = fmaf(gcur[u][c], in[t], acc[c][t]);
                        acc[c][9] += gcur[u][c];
                    }
                }
                w_ += dw_; h_ += dh_;
                if (w_ >= W) { w_ -= W; ++h_; }
                while (h_ >= H) h_ -= H;
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < 10; ++t) red[((size_t)pl * Cout + q * 4 + c) * 10 + t] = acc[c][t];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < Cout * 10; i += blockDim.x) {
        float v = 0.f;
        for (int k = 0; k < ppb; ++k) v += red[(size_t)k * Cout * 10 + i];
        part[(size_t)blockIdx.x * Cout * 10 + i] = v;
    }
}

// One workgroup per SIXTEEN consecutive (n, t) values — one 64-byte line of every partial row: thread (r, q) sums quad q of rows r, r + 64,
// ... (independent 16-byte loads), then a tree over the 64 row lanes — fixed order.  (Round 3's form, 16 lanes per value, was latency-bound
// once the partial rows came from conv12's 6,080 patch tiles: ~380 dependent round trips per lane; four values per workgroup used a
// quarter of every line it touched.)
__global__ void __launch_bounds__(256) conv3x3_c1_bwd_reduce(const float* __restrict__ part, float* __restrict__ gw,
                                                             float* __restrict__ gb, int nblk, int Cout) {
    __shared__ f4 red[256];
    const int tid = threadIdx.x, q = tid & 3, r = tid >> 2;
    const int i0 = blockIdx.x * 16 + q * 4;                     // values i0 .. i0 + 3 of the [Cout][10] row
    const int nval = Cout * 10;
    f4 v = {0.f, 0.f, 0.f, 0.f};
    if (i0 < nval) {
#pragma unroll 4
        for (int k = r; k < nblk; k += 64) v += *reinterpret_cast<const f4*>(part + (size_t)k * nval + i0);
    }
    red[tid] = v;
    __syncthreads();
#pragma unroll
    for (int sft = 128; sft >= 4; sft >>= 1) {                  // over the row lanes (tid >> 2); the quad index stays
        if (tid < sft) red[tid] += red[tid + sft];
        __syncthreads();
    }
    if (tid < 4 && i0 < nval) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = i0 + c, n = i / 10, t = i - n * 10;
            const float rv = red[tid][c];
            if (t < 9) gw[n * 9 + t] = rv;
            else if (gb != nullptr) gb[n] = rv;
        }
    }
}

// out[c] = sum_k part[k][c], fixed order — the per-workgroup column sums the pool's backward and the masked data-gradient launches leave
// behind (bias gradients), a few thousand rows.  One workgroup per sixteen columns: thread (r, q) adds quad q of rows r, r + 64, ...
// (independent 16-byte loads), then a tree over the 64 row lanes.  (16 lanes per column looping over ~500 rows each was latency-bound:
// 0.2-0.5 ms per call.)
__global__ void __launch_bounds__(256) colsum_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int rows, int C) {
    __shared__ f4 red[256];
    const int tid = threadIdx.x, q = tid & 3, r = tid >> 2;     // sixteen columns (one 64-byte line of every row) per workgroup
    const int c0 = blockIdx.x * 16 + q * 4;
    f4 v = {0.f, 0.f, 0.f, 0.f};
    if (c0 < C) {
#pragma unroll 4
        for (int k = r; k < rows; k += 64) v += *reinterpret_cast<const f4*>(part + (size_t)k * C + c0);
    }
    red[tid] = v;
    __syncthreads();
#pragma unroll
    for (int sft = 128; sft >= 4; sft >>= 1) {                  // over the 64 row lanes; the quad index stays
        if (tid < sft) red[tid] += red[tid + sft];
        __syncthreads();
    }
    if (tid < 4 && c0 < C) *reinterpret_cast<f4*>(out + c0) = red[tid];
}

// Border tables of a ragged batch (ragged.py), made on the device from the utterance lengths alone: blockIdx.y = table
// (2 l: spans of level l — {first row, one past the last row} of the utterance owning each row; 2 l + 1: the pool closing level l —
// {first input row, has-second-row} per output row; 2 levels: {owner, row inside the utterance} of every output frame, int64 for
// torch indexing).  Every workgroup rebuilds the (tiny) per-utterance prefix sums of its level in LDS and finds a row's owner by
// binary search.  One launch replaces ~60 cumsum / repeat_interleave / arange / gather launches per group of utterances.
struct RaggedTablesParams {
    int B, levels;
    int off[17];      // int32 offset of table k in `out` (k < 2 * levels)
    int rows[9];      // rows at level l (rows[levels] = output frames)
};
constexpr int kRaggedMaxB = 4096;

__global__ void __launch_bounds__(256) ragged_tables_kernel(const int* __restrict__ lens, const RaggedTablesParams p, int* __restrict__ out,
                                                            long long* __restrict__ fin) {
    __shared__ int len_a[kRaggedMaxB], st_a[kRaggedMaxB + 1], st_b[kRaggedMaxB + 1], part[256];
    const int k = blockIdx.y, tid = threadIdx.x, B = p.B;
    const bool final_tab = k == 2 * p.levels;
    const int l = final_tab ? p.levels : k >> 1;
    const bool pool = !final_tab && (k & 1);
    const int nrows = pool ? p.rows[l + 1] : p.rows[l];
    if ((long long)blockIdx.x * 256 >= nrows) return;
    // lengths at level l (len_a) and, for a pool table, at level l + 1; exclusive prefix sums st_a (level l) / st_b (level l + 1)
    const int E = (B + 255) / 256;
    auto scan = [&](int shift, int* st) {
        int acc = 0;
        for (int e = 0; e < E; ++e) {
            const int b = tid * E + e;
            if (b < B) {
                int n = lens[b];
                for (int q = 0; q < shift; ++q) n = (n + 1) >> 1;
                if (st == st_a) len_a[b] = n;
                acc += n;
            }
        }
        part[tid] = acc;
        __syncthreads();
        if (tid == 0) {
            int run = 0;
            for (int t = 0; t < 256; ++t) { const int v = part[t]; part[t] = run; run += v; }
        }
        __syncthreads();
        int run = part[tid];
        for (int e = 0; e < E; ++e) {
            const int b = tid * E + e;
            if (b < B) {
                int n = lens[b];
                for (int q = 0; q < shift; ++q) n = (n + 1) >> 1;
                st[b] = run;
                run += n;
            }
        }
        if (tid == 255) st[B] = run;      // (threads past the last utterance carry the total)
        __syncthreads();
    };
    scan(l, st_a);
    if (pool) scan(l + 1, st_b);
    const int r = blockIdx.x * 256 + tid;
    if (r >= nrows) return;
    const int* st = pool ? st_b : st_a;
    int lo = 0, hi = B - 1;               // largest b with st[b] <= r
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (st[mid] <= r) lo = mid; else hi = mid - 1;
    }
    if (final_tab) {
        fin[r] = lo;
        fin[(size_t)nrows + r] = r - st_a[lo];
    } else if (!pool) {
        out[p.off[k] + 2 * r] = st_a[lo];
        out[p.off[k] + 2 * r + 1] = st_a[lo] + len_a[lo];
    } else {
        const int local = r - st_b[lo];
        out[p.off[k] + 2 * r] = st_a[lo] + 2 * local;
        out[p.off[k] + 2 * r + 1] = (2 * local + 1 < len_a[lo]) ? 1 : 0;
    }
}

}  // namespace dasv

extern "C" {

int dasv_ragged_tables(const int* lengths_dev, const int* lengths_host, int B, int levels, int* tables, long long* frame_index,
                       dasv_stream_t stream) {
    if (lengths_dev == nullptr || lengths_host == nullptr || tables == nullptr) return DASV_E_NULL;
    if (B <= 0 || B > dasv::kRaggedMaxB || levels <= 0 || levels > 8) return DASV_E_SHAPE;
    dasv::RaggedTablesParams p;
    p.B = B; p.levels = levels;
    long long off = 0;
    for (int l = 0; l <= levels; ++l) {
        long long rows = 0;
        for (int b = 0; b < B; ++b) {
            int n = lengths_host[b];
            if (n <= 0) return DASV_E_SHAPE;
            for (int q = 0; q < l; ++q) n = (n + 1) >> 1;
            rows += n;
        }
        if (rows > 0x3fffffff) return DASV_E_SHAPE;
        p.rows[l] = (int)rows;
    }
    for (int l = 0; l < levels; ++l) {
        p.off[2 * l] = (int)off;     off += 2LL * p.rows[l];
        p.off[2 * l + 1] = (int)off; off += 2LL * p.rows[l + 1];
        if (off > 0x7fffffff) return DASV_E_SHAPE;
    }
    const dim3 grid((unsigned)((p.rows[0] + 255) / 256), (unsigned)(2 * levels + (frame_index != nullptr ? 1 : 0)));
    hipLaunchKernelGGL(dasv::ragged_tables_kernel, grid, dim3(256), 0, (hipStream_t)stream, lengths_dev, p, tables, frame_index);
    return (int)hipGetLastError();
}


static int c1_launch(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int Cout, int relu,
                     const int* row_span, dasv_stream_t stream, int pack, const uint32_t* x_range, uint32_t* y_range) {
    if (x == nullptr || w == nullptr || y == nullptr) return DASV_E_NULL;
    if ((pack == 1 || pack == 2) && (x_range == nullptr || y_range == nullptr)) return DASV_E_NULL;   // a packed output is scaled from the input's range
    if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cout % 4 != 0 || Cout > 1024 || 256 % (Cout / 4) != 0) return DASV_E_SHAPE;
    if (pack == 1 && Cout % 64 != 0) return DASV_E_SHAPE;
    const long long npix = (long long)B * H * W;
    if (W > 4000) return DASV_E_SHAPE;                                   // the chunk's input run (512 + 2 W + 2 floats) lives in LDS
    long long blocks = (npix + 511) / 512;                               // 512-pixel chunks (C1_CHUNK)
    if (blocks > 256 * 16) blocks = 256 * 16;
    const size_t lds = (size_t)(((512 + 2 * W + 2 + 3) & ~3) + 512 * 9) * sizeof(float);   // the input run | [512][9] gathered taps
    hipLaunchKernelGGL(dasv::conv3x3_c1_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, x, w, bias, y, H, W,
                       Cout, npix, relu, row_span, pack, x_range, y_range);
    return (int)hipGetLastError();
}

int dasv_range_absmax_f32(const float* x, size_t n, uint32_t* range, dasv_stream_t stream) {
    if (x == nullptr || range == nullptr) return DASV_E_NULL;
    if (n == 0 || (uintptr_t)x % 16 != 0) return DASV_E_SHAPE;
    long long blocks = (long long)((n / 4 + 255) / 256);
    if (blocks < 1) blocks = 1;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(dasv::range_absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (long long)n, range);
    return (int)hipGetLastError();
}

int dasv_conv3x3_c1_bias_relu_x8_nhwc(const float* x, const uint32_t* x_range, const float* w, const float* bias, void* y_packed,
                                      uint32_t* y_range, int B, int H, int W, int Cout, int relu, dasv_stream_t stream) {
    return c1_launch(x, w, bias, static_cast<float*>(y_packed), B, H, W, Cout, relu, nullptr, stream, 1, x_range, y_range);
}

int dasv_conv3x3_c1_bias_relu_nhwc(const float* x, const float* w, const float* bias, float* y, uint32_t* y_range, int B, int H,
                                   int W, int Cout, int relu, dasv_stream_t stream) {
    return c1_launch(x, w, bias, y, B, H, W, Cout, relu, nullptr, stream, 0, nullptr, y_range);
}

int dasv_conv3x3_c1_bias_relu_ragged_nhwc(const float* x, const float* w, const float* bias, float* y, uint32_t* y_range,
                                          const int* row_span, int rows, int W, int Cout, int relu, dasv_stream_t stream) {
    if (row_span == nullptr) return DASV_E_NULL;
    return c1_launch(x, w, bias, y, 1, rows, W, Cout, relu, row_span, stream, 0, nullptr, y_range);
}

int dasv_conv3x3_c1_bias_relu_ragged_x8_nhwc(const float* x, const uint32_t* x_range, const float* w, const float* bias,
                                             void* y_packed, uint32_t* y_range, const int* row_span, int rows, int W, int Cout,
                                             int relu, dasv_stream_t stream) {
    if (row_span == nullptr) return DASV_E_NULL;
    return c1_launch(x, w, bias, static_cast<float*>(y_packed), 1, rows, W, Cout, relu, row_span, stream, 1, x_range, y_range);
}

static int pool_launch(const float* x, float* y, const int* pool_src, int B, int H, int Ho, int W, int C, int pack,
                       const uint32_t* x_range, uint32_t* y_range, dasv_stream_t stream, int in_f16 = 0, uint8_t* idx = nullptr) {
    if (x == nullptr || y == nullptr) return DASV_E_NULL;
    if (idx != nullptr && in_f16) return DASV_E_DTYPE;
    if ((pack == 1 || pack == 2) && (x_range == nullptr || y_range == nullptr)) return DASV_E_NULL;
    if (in_f16 == 1 && (x_range == nullptr || pack == 1)) return x_range == nullptr ? DASV_E_NULL : DASV_E_SHAPE;
    if (pack == 2 && in_f16 != 1) return DASV_E_SHAPE;
    if ((pack == 3) != (in_f16 == 2 && pack != 0)) return DASV_E_SHAPE;      // bf16 in -> bf16 (pack 3) or fp32 (pack 0) out
    if (B <= 0 || H <= 0 || Ho <= 0 || W <= 0 || C <= 0 || C % (pack == 1 ? 64 : 4) != 0) return DASV_E_SHAPE;
    const int Wo = (W + 1) / 2;
    const long long nvec = (long long)B * Ho * Wo * (C / 4);
    long long blocks = (nvec + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(dasv::maxpool2x2_nhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, H, W, C, Ho, Wo,
                       nvec, pool_src, pack, x_range, y_range, in_f16, idx);
    return (int)hipGetLastError();
}

int dasv_maxpool2x2_ceil_f16_nhwc(const void* x_f16, const uint32_t* x_range, void* y, int y_f16, uint32_t* y_range, int B, int H,
                                  int W, int C, dasv_stream_t stream) {
    return pool_launch(static_cast<const float*>(x_f16), static_cast<float*>(y), nullptr, B, H, (H + 1) / 2, W, C, y_f16 ? 2 : 0,
                       x_range, y_range, stream, 1);
}

int dasv_conv3x3_c1_bias_relu_f16_nhwc(const float* x, const uint32_t* x_range, const float* w, const float* bias, void* y_f16,
                                       uint32_t* y_range, int B, int H, int W, int Cout, int relu, dasv_stream_t stream) {
    return c1_launch(x, w, bias, static_cast<float*>(y_f16), B, H, W, Cout, relu, nullptr, stream, 2, x_range, y_range);
}

int dasv_maxpool2x2_ceil_ragged_f16_nhwc(const void* x_f16, const uint32_t* x_range, void* y, int y_f16, uint32_t* y_range,
                                         const int* pool_src, int rows_in, int rows_out, int W, int C, dasv_stream_t stream) {
    if (pool_src == nullptr) return DASV_E_NULL;
    return pool_launch(static_cast<const float*>(x_f16), static_cast<float*>(y), pool_src, 1, rows_in, rows_out, W, C, y_f16 ? 2 : 0,
                       x_range, y_range, stream, 1);
}

int dasv_conv3x3_c1_bias_relu_ragged_f16_nhwc(const float* x, const uint32_t* x_range, const float* w, const float* bias, void* y_f16,
                                              uint32_t* y_range, const int* row_span, int rows, int W, int Cout, int relu,
                                              dasv_stream_t stream) {
    if (row_span == nullptr) return DASV_E_NULL;
    return c1_launch(x, w, bias, static_cast<float*>(y_f16), 1, rows, W, Cout, relu, row_span, stream, 2, x_range, y_range);
}

int dasv_maxpool2x2_ceil_nhwc(const float* x, const uint32_t* x_range, float* y, uint32_t* y_range, int B, int H, int W, int C,
                              dasv_stream_t stream) {
    return pool_launch(x, y, nullptr, B, H, (H + 1) / 2, W, C, 0, x_range, y_range, stream);
}

int dasv_maxpool2x2_ceil_x8_nhwc(const float* x, const uint32_t* x_range, void* y_packed, uint32_t* y_range, int B, int H, int W,
                                 int C, dasv_stream_t stream) {
    return pool_launch(x, static_cast<float*>(y_packed), nullptr, B, H, (H + 1) / 2, W, C, 1, x_range, y_range, stream);
}

int dasv_maxpool2x2_ceil_ragged_nhwc(const float* x, const uint32_t* x_range, float* y, uint32_t* y_range, const int* pool_src,
                                     int rows_in, int rows_out, int W, int C, dasv_stream_t stream) {
    if (pool_src == nullptr) return DASV_E_NULL;
    return pool_launch(x, y, pool_src, 1, rows_in, rows_out, W, C, 0, x_range, y_range, stream);
}

int dasv_maxpool2x2_ceil_ragged_x8_nhwc(const float* x, const uint32_t* x_range, void* y_packed, uint32_t* y_range,
                                        const int* pool_src, int rows_in, int rows_out, int W, int C, dasv_stream_t stream) {
    if (pool_src == nullptr) return DASV_E_NULL;
    return pool_launch(x, static_cast<float*>(y_packed), pool_src, 1, rows_in, rows_out, W, C, 1, x_range, y_range, stream);
}

static long long maxpool_bwd_blocks(int B, int H, int W, int C) {
    const long long nvec = (long long)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
    const long long blocks = (nvec + 255) / 256;
    // (2,048 workgroups of grid-stride loops stream as fast as 8,192 and leave a quarter of the column-sum partial rows behind: the
    // reduction of a block's partials took 16 us for up to 25 MB of them)
    return blocks > 256 * 8 ? 256 * 8 : blocks;
}

static long long maxpool_bwd_blocks_out(long long out_rows, int W, int C) {
    const long long nvec = out_rows * ((W + 1) / 2) * (C / 4);
    const long long blocks = (nvec + 255) / 256;
    return blocks > 256 * 8 ? 256 * 8 : blocks;
}

static int maxpool_bwd_launch(const float* x, const float* gy, float* gx, int B, int H, int W, int C, int relu_mask,
                              float* colsum, dasv_stream_t stream, const uint8_t* idx = nullptr, const int* pool_src = nullptr,
                              int rows_out = 0, const uint32_t* gy_rec = nullptr, uint32_t* gx_rec = nullptr) {
    if ((gy_rec == nullptr) != (gx_rec == nullptr)) return DASV_E_NULL;
    if (gx_rec != nullptr && C % 64 != 0) return DASV_E_SHAPE;
    if (colsum != nullptr && (C / 4 > 256 || 256 % (C / 4) != 0)) return DASV_E_SHAPE;
    if ((x == nullptr && idx == nullptr) || gy == nullptr || gx == nullptr) return DASV_E_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 != 0) return DASV_E_SHAPE;
    if (pool_src != nullptr && (B != 1 || rows_out <= 0)) return DASV_E_SHAPE;
    const int Ho = pool_src != nullptr ? rows_out : (H + 1) / 2, Wo = (W + 1) / 2;
    const long long nvec = (long long)B * Ho * Wo * (C / 4);
    const long long blocks = pool_src != nullptr ? maxpool_bwd_blocks_out(rows_out, W, C) : maxpool_bwd_blocks(B, H, W, C);
    hipLaunchKernelGGL(dasv::maxpool2x2_bwd_nhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, gy, gx, H, W,
                       C, Ho, Wo, nvec, relu_mask, colsum, idx, pool_src, gy_rec, gx_rec);
    return (int)hipGetLastError();
}

int dasv_maxpool2x2_ceil_bwd_nhwc(const float* x, const float* gy, float* gx, int B, int H, int W, int C, dasv_stream_t stream) {
    return maxpool_bwd_launch(x, gy, gx, B, H, W, C, 0, nullptr, stream);
}

int dasv_maxpool2x2_ceil_relu_bwd_nhwc(const float* x, const float* gy, float* gx, float* colsum_partial, int B, int H, int W,
                                       int C, dasv_stream_t stream) {
    return maxpool_bwd_launch(x, gy, gx, B, H, W, C, 1, colsum_partial, stream);
}

/* training: the forward pool also writes one byte per pooled element {arg-max position 0-3 | 4 if the maximum is > 0}; the backward
 * pass then reads those bytes and gy only — not the full-resolution map (786 MB for the first block of a [64,300,80] batch) */
int dasv_maxpool2x2_ceil_idx_nhwc(const float* x, const uint32_t* x_range, float* y, uint32_t* y_range, uint8_t* idx, int B, int H,
                                  int W, int C, dasv_stream_t stream) {
    if (idx == nullptr) return DASV_E_NULL;
    return pool_launch(x, y, nullptr, B, H, (H + 1) / 2, W, C, 0, x_range, y_range, stream, 0, idx);
}

int dasv_maxpool2x2_ceil_bwd_idx_nhwc(const uint8_t* idx, const float* gy, float* gx, float* colsum_partial, int relu_mask, int B,
                                      int H, int W, int C, dasv_stream_t stream) {
    if (idx == nullptr) return DASV_E_NULL;
    return maxpool_bwd_launch(nullptr, gy, gx, B, H, W, C, relu_mask ? 1 : 0, colsum_partial, stream, idx);
}

int dasv_maxpool2x2_ceil_bwd_idx_x8_nhwc(const uint8_t* idx, const float* gy, const uint32_t* gy_range, void* gx_packed, uint32_t* gx_range,
                                         float* colsum_partial, int relu_mask, int B, int H, int W, int C, dasv_stream_t stream) {
    if (idx == nullptr || gy_range == nullptr || gx_range == nullptr) return DASV_E_NULL;
    return maxpool_bwd_launch(nullptr, gy, static_cast<float*>(gx_packed), B, H, W, C, relu_mask ? 1 : 0, colsum_partial, stream, idx, nullptr, 0,
                              gy_range, gx_range);
}

/* the packed training chain over a ragged batch: the pool writes its output PACKED next to the arg-max bytes, its backward the
 * scattered gradient */
int dasv_maxpool2x2_ceil_idx_ragged_x8_nhwc(const float* x, const uint32_t* x_range, void* y_packed, uint32_t* y_range, uint8_t* idx,
                                            const int* pool_src, int rows_in, int rows_out, int W, int C, dasv_stream_t stream) {
    if (idx == nullptr || pool_src == nullptr) return DASV_E_NULL;
    return pool_launch(x, static_cast<float*>(y_packed), pool_src, 1, rows_in, rows_out, W, C, 1, x_range, y_range, stream, 0, idx);
}

int dasv_maxpool2x2_ceil_bwd_idx_ragged_x8_nhwc(const uint8_t* idx, const float* gy, const uint32_t* gy_range, void* gx_packed,
                                                uint32_t* gx_range, float* colsum_partial, int relu_mask, const int* pool_src, int rows_in,
                                                int rows_out, int W, int C, dasv_stream_t stream) {
    if (idx == nullptr || pool_src == nullptr || gy_range == nullptr || gx_range == nullptr) return DASV_E_NULL;
    return maxpool_bwd_launch(nullptr, gy, static_cast<float*>(gx_packed), 1, rows_in, W, C, relu_mask ? 1 : 0, colsum_partial, stream, idx,
                              pool_src, rows_out, gy_range, gx_range);
}

/* the training pool and its backward over a ragged batch (pool_src as in dasv_maxpool2x2_ceil_ragged_nhwc): utterances of different
 * lengths in ONE training minibatch (BASELINE.json north_star; the reference draws one length per batch, train.py:205-207) */
int dasv_maxpool2x2_ceil_idx_ragged_nhwc(const float* x, const uint32_t* x_range, float* y, uint32_t* y_range, uint8_t* idx,
                                         const int* pool_src, int rows_in, int rows_out, int W, int C, dasv_stream_t stream) {
    if (idx == nullptr || pool_src == nullptr) return DASV_E_NULL;
    return pool_launch(x, y, pool_src, 1, rows_in, rows_out, W, C, 0, x_range, y_range, stream, 0, idx);
}

int dasv_maxpool2x2_ceil_bwd_idx_ragged_nhwc(const uint8_t* idx, const float* gy, float* gx, float* colsum_partial, int relu_mask,
                                             const int* pool_src, int rows_in, int rows_out, int W, int C, dasv_stream_t stream) {
    if (idx == nullptr || pool_src == nullptr) return DASV_E_NULL;
    return maxpool_bwd_launch(nullptr, gy, gx, 1, rows_in, W, C, relu_mask ? 1 : 0, colsum_partial, stream, idx, pool_src, rows_out);
}

int dasv_maxpool2x2_bwd_colsum_rows_ragged(int rows_out, int W, int C) {
    if (rows_out <= 0 || W <= 0 || C <= 0 || C % 4 != 0 || C / 4 > 256 || 256 % (C / 4) != 0) return 0;
    return (int)maxpool_bwd_blocks_out(rows_out, W, C);
}

int dasv_maxpool2x2_bwd_colsum_rows(int B, int H, int W, int C) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 != 0 || C / 4 > 256 || 256 % (C / 4) != 0) return 0;
    return (int)maxpool_bwd_blocks(B, H, W, C);
}

size_t dasv_conv3x3_c1_bwd_workspace_bytes(int B, int H, int W, int Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0) return 0;
    return (size_t)2048 * Cout * 10 * sizeof(float);
}

static int c1_bwd_launch(const float* x, const float* y, const float* gy, float* gw, float* gb, void* workspace,
                         size_t workspace_bytes, int B, int H, int W, int Cout, dasv_stream_t stream, const int* row_span);

int dasv_conv3x3_c1_bwd_nhwc(const float* x, const float* y, const float* gy, float* gw, float* gb, void* workspace,
                             size_t workspace_bytes, int B, int H, int W, int Cout, dasv_stream_t stream) {
    return c1_bwd_launch(x, y, gy, gw, gb, workspace, workspace_bytes, B, H, W, Cout, stream, nullptr);
}

int dasv_conv3x3_c1_bwd_ragged_nhwc(const float* x, const float* y, const float* gy, float* gw, float* gb, void* workspace,
                                    size_t workspace_bytes, const int* row_span, int rows, int W, int Cout, dasv_stream_t stream) {
    if (row_span == nullptr) return DASV_E_NULL;
    return c1_bwd_launch(x, y, gy, gw, gb, workspace, workspace_bytes, 1, rows, W, Cout, stream, row_span);
}

/* the bf16 STORAGE chain of the bf16-operand mode (BASELINE config 3 names bf16): conv11 and the pool on bf16 NHWC tensors — plain
 * round-to-nearest-even values, no scale and no range record (bf16 has fp32's exponent range) */
int dasv_conv3x3_c1_bias_relu_bf16_nhwc(const float* x, const float* w, const float* bias, void* y_bf16, int B, int H, int W, int Cout,
                                        int relu, dasv_stream_t stream) {
    return c1_launch(x, w, bias, static_cast<float*>(y_bf16), B, H, W, Cout, relu, nullptr, stream, 3, nullptr, nullptr);
}

int dasv_maxpool2x2_ceil_bf16_nhwc(const void* x_bf16, void* y, int y_bf16, int B, int H, int W, int C, dasv_stream_t stream) {
    return pool_launch(static_cast<const float*>(x_bf16), static_cast<float*>(y), nullptr, B, H, (H + 1) / 2, W, C, y_bf16 ? 3 : 0, nullptr,
                       nullptr, stream, 2);
}

int dasv_colsum_reduce(const float* partial, int rows, int C, float* out, dasv_stream_t stream) {
    if (partial == nullptr || out == nullptr) return DASV_E_NULL;
    if (rows <= 0 || C <= 0 || C % 4 != 0) return DASV_E_SHAPE;
    hipLaunchKernelGGL(dasv::colsum_reduce_kernel, dim3((unsigned)((C + 15) / 16)), dim3(256), 0, (hipStream_t)stream, partial, out, rows, C);
    return (int)hipGetLastError();
}

int dasv_conv3x3_c1_bwd_reduce_parts(const float* partial, int rows, float* gw, float* gb, int Cout, dasv_stream_t stream) {
    if (partial == nullptr || gw == nullptr) return DASV_E_NULL;
    if (rows <= 0 || Cout <= 0) return DASV_E_SHAPE;
    if (Cout % 2 != 0) return DASV_E_SHAPE;
    hipLaunchKernelGGL(dasv::conv3x3_c1_bwd_reduce, dim3((unsigned)((Cout * 10 + 15) / 16)), dim3(256), 0, (hipStream_t)stream, partial, gw, gb, rows, Cout);
    return (int)hipGetLastError();
}

static int c1_bwd_launch(const float* x, const float* y, const float* gy, float* gw, float* gb, void* workspace,
                         size_t workspace_bytes, int B, int H, int W, int Cout, dasv_stream_t stream, const int* row_span) {
    if (x == nullptr || gy == nullptr || gw == nullptr || workspace == nullptr) return DASV_E_NULL;      // y may be NULL: gy pre-masked
    if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cout % 4 != 0 || Cout > 1024 || 256 % (Cout / 4) != 0) return DASV_E_SHAPE;
    if (workspace_bytes < dasv_conv3x3_c1_bwd_workspace_bytes(B, H, W, Cout)) return DASV_E_WORKSPACE;
    const long long npix = (long long)B * H * W;
    const int ppb = 256 / (Cout / 4);
    int nblk = (int)((npix + ppb - 1) / ppb < 2048 ? (npix + ppb - 1) / ppb : 2048);
    const int pix_per_block = (int)((npix + nblk - 1) / nblk);
    nblk = (int)((npix + pix_per_block - 1) / pix_per_block);
    const size_t lds = (size_t)ppb * Cout * 10 * sizeof(float);
    if (lds > 64 * 1024) return DASV_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    float* part = static_cast<float*>(workspace);
    hipLaunchKernelGGL(dasv::conv3x3_c1_bwd_kernel, dim3((unsigned)nblk), dim3(256), lds, s, x, y, gy, part, H, W, Cout, npix,
                       pix_per_block, row_span);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(dasv::conv3x3_c1_bwd_reduce, dim3((unsigned)((Cout * 10 + 15) / 16)), dim3(256), 0, s, part, gw, gb, nblk, Cout);
    return (int)hipGetLastError();
}

}  // extern "C"
