// conv3x3 (stride 1, pad 1) + bias + ReLU as an implicit GEMM on the gfx950 matrix cores.
//
// Replaces `F.relu(self.convXY(x))` of the reference's scripts/CNNs.py:38-48 / 72-86 for the layers with
// Cin % 64 == 0 and Cout % 128 == 0 (conv12 ... conv42 of VGG3L/VGG4L at the reference's kernel sizes).
//
// GEMM view:  M = B*H*W output pixels,  N = Cout,  K = 9*Cin ordered (tap, cin).
//   A[m][k]  = x[b, h+ky-1, w+kx-1, c]      (NHWC fp32 in HBM, zero outside the image)
//   Bt[n][k] = w[n, c, ky, kx]              (pre-packed once to [Cout][9][Cin] bf16 planes)
// Precision modes:
//   SPLIT (default): every fp32 operand v is split v = hi + lo with hi = bf16(v), lo = bf16(v - hi); the product
//     is accumulated as hi*hi + hi*lo + lo*hi in fp32 on v_mfma_f32_16x16x32_bf16 (three MFMAs per product,
//     dropping lo*lo ~ 2^-18): ~2^-16 relative error per product, i.e. fp32-grade for the 1e-3 embedding bar,
//     at a third of the bf16 MFMA rate — which is still ~5x the fp32 MFMA rate.
//   BF16: hi only (one MFMA per product), 8 mantissa bits.
//   F16:  operands rounded to fp16 (11 mantissa bits, v_mfma_f32_16x16x32_f16), one MFMA per product; activations
//     saturate at +-65504 and values below 6e-8 flush, so this is a forward/inference mode (weights come from
//     dasv_conv3x3_pack_weights_f16).
//   X8 (forward / inference mode of the fp32-grade path, window kernel only): v = hi + r with hi = fp16(v); the product is
//     hi*hi on v_mfma_f32_16x16x32_f16 plus BOTH cross terms hi*r' + r*hi' on ONE block-scaled e4m3 MFMA per 64 channels
//     (v_mfma_scale_f32_16x16x128_f8f6f4, twice the bf16 rate): the cross terms only need ~4 significant bits because r is
//     already 2^-11 of v.  Operands of that MFMA: e4m3(v) and e4m3(r * 2^12) for activations (fixed exact power-of-two
//     scale, undone by the instruction's E8M0 block scale), e4m3(w * 2^Sh[n]) and e4m3(r_w * 2^Sw[n]) for weights (scales
//     per output channel, chosen when the weights are packed).  Pipe cycles per 64-channel tap: 2 x 16 + 32 = 64 instead of
//     6 x 16 = 96 (tools/mfma_mix_probe.hip: 1.51x in a bare loop); measured error vs a float64 convolution ~1e-5 of the
//     output scale for activations between 0.01 and 100 (split-bf16: 5e-6; plain fp16: 3e-4), degrading towards the fp16
//     figure outside that range.  Needs fp16's exponent range on the hi part: activations, not gradients.
// Tile: 128 pixels x 128 channels x 64 (one tap's channel slab) per 256-thread workgroup; 2x2 waves, each
// 64x64 = 4x4 MFMA tiles.  Global -> registers (next K-step, issued before the MFMAs of the current one) ->
// LDS (XOR-swizzled 16-byte slots so ds_read_b128 fragment reads spread over the banks) -> MFMA.
#include "dasv_common.hpp"
#include "../../include/dasv_hip.h"

#include <stdlib.h>

namespace dasv {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CBM = 128, CBN = 128, CBK = 64;

// timing experiments only (tools/conv_diag.py builds this file with -DDASV_CONV_DIAG=n; results are wrong then):
// 1 = the window kernel stages only the first channel slab's window, 2 = no MFMA section, 3 = weights staged once
#ifndef DASV_CONV_DIAG
#define DASV_CONV_DIAG 0
#endif
// -DDASV_CONV_STAMPS=1 (tools/conv_stamps.py): thread 0 of the launch's middle workgroup leaves 100 MHz timestamps at the phase
// boundaries of its tile in dasv_conv_stamps (split8 pipelined loop only); read with dasv_debug_conv_stamps
#ifndef DASV_CONV_STAMPS
#define DASV_CONV_STAMPS 0
#endif
#if DASV_CONV_STAMPS
__device__ unsigned long long dasv_conv_stamps[128];
// slots 0-63: the 100 MHz clock; 64-127: the shader-cycle counter at the same points (their ratio is the clock the CU held)
#define DASV_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) { dasv_conv_stamps[k] = __builtin_amdgcn_s_memrealtime(); \
                                                                                  dasv_conv_stamps[64 + (k)] = __builtin_readcyclecounter(); } } while (0)
// per-slab stamps: slots 2 ... 59 (the first 14 slabs; slots 60-63 are the epilogue's, 64+ the cycle counters — a 16-slab layer
// used to write its last slabs' stamps over them: the conv42 row of profiles/r03_summary.md)
#define DASV_STAMP_SLAB(k) do { if ((k) < 58) DASV_STAMP(k); } while (0)
#else
#define DASV_STAMP(k) do { } while (0)
#define DASV_STAMP_SLAB(k) do { } while (0)
#endif
#ifndef DASV_CONV_SCHED
#define DASV_CONV_SCHED 1   // reads-first scheduling hint (tools/conv_diag.py A/B)
#endif

__device__ __forceinline__ uint32_t bf16_rne(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;   // inputs are finite activations/weights
}
// (a, b) -> packed bf16 pairs hi = bf16(v) and lo = bf16(v - hi); plain casts lower to v_cvt_pk_bf16_f32 (RNE).  The
// subtraction compiles to one v_pk_add_f32; forcing two scalar v_sub_f32 (a 7 % win in the weight-gradient kernel, whose
// staging waves share a SIMD with MFMA waves) was measured here too, interleaved in one run: 1.7 % slower — staging in this
// kernel never runs beside MFMAs.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
    const f32x2_t v = {a, b};
    const bf16x2_t h = __builtin_convertvector(v, bf16x2_t);
    const f32x2_t hf = __builtin_convertvector(h, f32x2_t);
    const bf16x2_t l = __builtin_convertvector(v - hf, bf16x2_t);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, l);
}

// precision of the MFMA operands: split-bf16 (three products), bf16, fp16
constexpr int PREC_SPLIT = 0, PREC_BF16 = 1, PREC_F16 = 2, PREC_X8 = 3;
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// staging conversion of two fp32 values: `hi` gets the packed 16-bit operands of the mode, `lo` the split residuals
template <int PREC>
__device__ __forceinline__ void cvt2(float a, float b, uint32_t& hi, uint32_t& lo) {
    if (PREC == PREC_F16) {
        // RNE, saturating at the largest finite half (an activation beyond +-65504 must not turn the map into inf/NaN)
        const f32x2_t v = {__builtin_amdgcn_fmed3f(a, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(b, -65504.f, 65504.f)};
        hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
        lo = 0;
    } else {
        split2(a, b, hi, lo);
    }
}
// four fp32 values -> the 16-bit STORAGE form of a half-precision activation chain: fp16 of v * s (PREC_F16: s = 2^Ey from the range
// record) or bf16 of v (PREC_BF16: round to nearest even, no scale — bf16 has fp32's exponent range; a NaN stays a NaN)
template <int PREC>
__device__ __forceinline__ uint2 pack16x4(f32x4 v, float s) {
    if (PREC == PREC_F16) {
        const f32x4 q = v * s;
        const f16x2_t a = __builtin_convertvector(f32x2_t{q[0], q[1]}, f16x2_t), b2 = __builtin_convertvector(f32x2_t{q[2], q[3]}, f16x2_t);
        return make_uint2(__builtin_bit_cast(uint32_t, a), __builtin_bit_cast(uint32_t, b2));
    }
    const bf16x2_t a = __builtin_convertvector(f32x2_t{v[0], v[1]}, bf16x2_t), b2 = __builtin_convertvector(f32x2_t{v[2], v[3]}, bf16x2_t);
    return make_uint2(__builtin_bit_cast(uint32_t, a), __builtin_bit_cast(uint32_t, b2));
}
template <int PREC>
__device__ __forceinline__ f32x4 mma16(bf16x8 a, bf16x8 b, f32x4 c) {
    if (PREC == PREC_F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

typedef int i32x8 __attribute__((ext_vector_type(8)));

// weights [Cout][Cin][3][3] fp32 (torch) -> X8 planes.  Every output channel n is scaled by its own exact power of two,
// c = w * 2^Ew[n] with max |c| in [128, 256) (undone by the epilogue), so that weights of ANY magnitude sit in the normal range of
// the fp16 plane:  fp16(c) as [Cout][9][Cin] and, per (channel n, tap, 64-channel slab), 128 bytes
// { e4m3(r_c * 2^Sw[n]) x 64 | e4m3(c * 2^Sh[n]) x 64 }.  Behind the e4m3 planes: [Cout][4] scale bytes {127 - Sw, 127 - Sh, 0, 0}
// (the E8M0 block scales of the MFMA) | [Cout] floats 2^-Ew[n] | { max_n sum_k |w[n][k]|, max_n |bias[n]|, 0, 0 } — the last two bound
// the layer's output from its input's maximum (|y| <= amax_x * L1 + |b|), which is what a PACKED output is scaled by before any of
// it exists.  One workgroup per output channel.
// dynamic LDS of the pack kernel: one channel's 9 * Cin floats when that fits the 64 KB a launch may ask for without opting in
static inline size_t pack_row_lds(int Cin) { return (size_t)Cin * 36 <= 65536 - 64 ? (size_t)Cin * 36 : 0; }
__device__ __forceinline__ float block_max256(float v, float* red) {
    v = group_max<64>(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__global__ void __launch_bounds__(256) conv3x3_pack_weights_x8_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                                                       uint16_t* __restrict__ wh, uint8_t* __restrict__ w8, int Cout, int Cin,
                                                                       int transposed, int staged) {
    __shared__ float red[4];
    extern __shared__ float wrow[];          // staged != 0: the channel's 9 * Cin source values, in (c, tap) order
    const int o = blockIdx.x, tid = threadIdx.x, n = 9 * Cin;
    // transposed: w is the weight [Cin][Cout][3][3] of the layer whose DATA GRADIENT these planes compute — the same convolution with
    // the weights transposed over (out, in) and flipped over (ky, kx): plane element (o, tap, c) = w[c][o][8 - tap]
    // (transposed == 2: w is already the transposed, flipped tensor — read like a forward weight, treated like data-gradient planes)
    auto gsrc = [&](int c, int tap) -> float {
        return transposed == 1 ? w[((size_t)c * Cout + o) * 9 + (8 - tap)] : w[((size_t)o * Cin + c) * 9 + tap];
    };
    // the three passes below each wait out the memory latency of their own loads when they read `w` (a workgroup has ~18 dependent
    // trips per pass and the grid is two workgroups per CU): with the row staged in LDS first — all of its loads in flight at once —
    // they read LDS instead.  Same values in the same order: the planes are bitwise those of the unstaged form.
    if (staged) {
        if (transposed != 1)
            for (int i = tid * 4; i < n; i += 1024)
                *reinterpret_cast<float4*>(wrow + i) = *reinterpret_cast<const float4*>(w + (size_t)o * n + i);
        else
            for (int i = tid; i < n; i += 256) wrow[i] = gsrc(i / 9, i % 9);
        __syncthreads();
    }
    auto src = [&](int c, int tap) -> float { return staged ? wrow[c * 9 + tap] : gsrc(c, tap); };
    uint8_t* aux = w8 + (size_t)Cout * n * 2;
    float* winv = reinterpret_cast<float*>(aux + (size_t)Cout * 4);
    uint32_t* tail = reinterpret_cast<uint32_t*>(aux + (size_t)Cout * 8);   // zeroed by the host wrapper before this launch
    float mw = 0.f, l1 = 0.f;
    for (int i = tid; i < n; i += 256) {
        const float v = src(i / 9, i % 9);
        mw = fmaxf(mw, fabsf(v));
        l1 += fabsf(v);
    }
    mw = block_max256(mw, red);
    l1 = group_sum<64>(l1);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = l1;
    __syncthreads();
    l1 = (red[0] + red[1]) + (red[2] + red[3]);
    // (a NaN weight: fmaxf drops it, the sum keeps it — it reaches the bound as a NaN, and the planes carry it anyway)
    const int Ew = x8_exp_for(__float_as_uint(mw));
    const float sw = x8_pow2(Ew);
    float mr = 0.f;
    for (int i = tid; i < n; i += 256) {
        const float c = src(i / 9, i % 9) * sw;
        mr = fmaxf(mr, fabsf(c - (float)(_Float16)c));
    }
    mr = block_max256(mr, red);
    const float mc = mw * sw;
    // e4m3 scale exponents: max * 2^S in [64, 128) (well inside e4m3's 448); 0 for an all-zero row / exactly representable weights
    int eh = 0, el = 0;
    if (mc > 0.f && mc < INFINITY) { (void)frexpf(mc, &eh); eh = 7 - eh; }      // mc = f * 2^e, f in [0.5, 1)  ->  mc * 2^(7 - e) in [64, 128)
    if (mr > 0.f && mr < INFINITY) { (void)frexpf(mr, &el); el = 7 - el; }
    eh = max(-100, min(100, eh)); el = max(-100, min(100, el));
    const float sh = ldexpf(1.0f, eh), sl = ldexpf(1.0f, el);
    if (tid == 0) {
        aux[o * 4 + 0] = (uint8_t)(127 - el); aux[o * 4 + 1] = (uint8_t)(127 - eh); aux[o * 4 + 2] = 0; aux[o * 4 + 3] = 0;
        winv[o] = x8_pow2(-Ew);
        if (!transposed) atomicMax(tail, __float_as_uint(l1) & 0x7fffffffu);
        // (data-gradient planes: no launch scales a packed OUTPUT by them — NaN bounds instead of a memset and Cout atomics per step)
        else if (o == 0) { tail[0] = 0x7fc00000u; tail[1] = 0x7fc00000u; }
    }
    if (o == 0 && !transposed) {
        float mb = 0.f;
        uint32_t bb = 0u;
        if (bias != nullptr)
            for (int i = tid; i < Cout; i += 256) {
                const uint32_t b = __float_as_uint(bias[i]) & 0x7fffffffu;
                bb = b > bb ? b : bb;
            }
        mb = block_max256(__uint_as_float(bb <= 0x7f800000u ? bb : 0u), red);
        // (a NaN bias: recorded as such)
        uint32_t any_nan = bb > 0x7f800000u ? bb : 0u;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)any_nan, off); any_nan = t > any_nan ? t : any_nan; }
        if ((tid & 63) == 0 && any_nan != 0u) atomicMax(tail + 1, any_nan);
        if (tid == 0) atomicMax(tail + 1, __float_as_uint(mb));
    }
    const int nslab = Cin / 64;
    // four consecutive input channels of one tap per trip (destination order (tap, cin)): one 8-byte store of halves and two 4-byte
    // stores of e4m3 bytes instead of four 2-byte and eight 1-byte ones
    for (int i = tid * 4; i < n; i += 1024) {
        const int c = i % Cin, tap = i / Cin;
        float v[4], r[4];
        uint16_t hb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = src(c + k, tap) * sw;
            const _Float16 h = (_Float16)v[k];
            hb[k] = __builtin_bit_cast(uint16_t, h);
            r[k] = (v[k] - (float)h) * sl;
        }
        *reinterpret_cast<uint2*>(wh + (size_t)o * n + i) = make_uint2((uint32_t)hb[0] | ((uint32_t)hb[1] << 16), (uint32_t)hb[2] | ((uint32_t)hb[3] << 16));
        int r8 = __builtin_amdgcn_cvt_pk_fp8_f32(r[0], r[1], 0, false);
        r8 = __builtin_amdgcn_cvt_pk_fp8_f32(r[2], r[3], r8, true);
        int v8 = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * sh, v[1] * sh, 0, false);
        v8 = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * sh, v[3] * sh, v8, true);
        uint8_t* row = w8 + (((size_t)o * 9 + tap) * nslab + c / 64) * 128;
        *reinterpret_cast<uint32_t*>(row + c % 64) = (uint32_t)r8;
        *reinterpret_cast<uint32_t*>(row + 64 + c % 64) = (uint32_t)v8;
    }
}

// wt[c][o][8 - tap] = w[o][c][tap]: the weight of a layer's DATA-GRADIENT convolution (transposed over (out, in), flipped over (ky, kx)) as
// an fp32 tensor in torch layout [Cin][Cout][3][3], through a 32 x 32 x 9 LDS tile so that both the reads (288 consecutive floats of one
// output channel) and the writes (288 consecutive floats of one input channel) are contiguous.  One launch instead of torch's flip + copy.
__global__ void __launch_bounds__(256) conv3x3_transpose_flip_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout, int Cin) {
    __shared__ float tile[32][32 * 9 + 1];
    const int o0 = blockIdx.x * 32, c0 = blockIdx.y * 32, tid = threadIdx.x;
    for (int i = tid; i < 32 * 288; i += 256) {
        const int ol = i / 288, r = i - ol * 288;                 // r = cl * 9 + tap
        tile[ol][r] = (o0 + ol < Cout && c0 + r / 9 < Cin) ? w[((size_t)(o0 + ol) * Cin + c0) * 9 + r] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < 32 * 288; i += 256) {
        const int cl = i / 288, r = i - cl * 288, ol = r / 9, tp = r - ol * 9;
        if (c0 + cl < Cin && o0 + ol < Cout) wt[((size_t)(c0 + cl) * Cout + o0 + ol) * 9 + tp] = tile[ol][cl * 9 + 8 - tp];
    }
}

// weights [Cout][Cin][3][3] fp32 (torch) -> fp16 plane [Cout][9][Cin]
__global__ void conv3x3_pack_weights_f16_kernel(const float* __restrict__ w, uint16_t* __restrict__ wh, int Cout, int Cin) {
    const long long n = (long long)Cout * 9 * Cin;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cin);
        const int tap = (int)((i / Cin) % 9);
        const int o = (int)(i / ((long long)Cin * 9));
        wh[i] = __builtin_bit_cast(uint16_t, (_Float16)w[((long long)o * Cin + c) * 9 + tap]);
    }
}

// { max_n sum_k |w[n][k]|, max_n |bias[n]| } behind a weight plane (tail zeroed by the host wrapper): what bounds a layer's output
// from its input's maximum — the scale of a half-precision OUTPUT tensor is fixed before any of it exists (fp16 activation chain)
__global__ void __launch_bounds__(256) conv3x3_weight_bounds_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                                                    uint32_t* __restrict__ tail, int Cout, int Cin) {
    __shared__ float red[4];
    const int o = blockIdx.x, n = 9 * Cin;
    float l1 = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) l1 += fabsf(w[(size_t)o * n + i]);
    l1 = group_sum<64>(l1);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = l1;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMax(tail, __float_as_uint((red[0] + red[1]) + (red[2] + red[3])) & 0x7fffffffu);
        if (bias != nullptr) atomicMax(tail + 1, __float_as_uint(bias[o]) & 0x7fffffffu);
    }
}

// weights [Cout][Cin][3][3] fp32 (torch) -> hi/lo planes [Cout][9][Cin] bf16
__global__ void conv3x3_pack_weights_kernel(const float* __restrict__ w, uint16_t* __restrict__ whi, uint16_t* __restrict__ wlo,
                                            int Cout, int Cin) {
    const long long n = (long long)Cout * 9 * Cin;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cin);
        const int tap = (int)((i / Cin) % 9);
        const int o = (int)(i / ((long long)Cin * 9));
        const float v = w[((long long)o * Cin + c) * 9 + tap];
        const uint32_t h = bf16_rne(v);
        whi[i] = (uint16_t)h;
        wlo[i] = (uint16_t)bf16_rne(v - __uint_as_float(h << 16));
    }
}

struct ConvParams {
    const float* x;        // [B,H,W,Cin]
    const uint16_t* whi;   // [Cout][9][Cin]
    const uint16_t* wlo;
    const float* bias;     // [Cout] or null
    const float* mask;     // [B,H,W,Cout] or null: outputs are zeroed where mask <= 0 (a ReLU's backward, fused)
    int mask_packed;       // mask is a PACKED activation tensor (dasv_common.hpp): the sign is read from its fp16 plane (pipelined launches)
    float* colsum;         // [mtiles][Cout] or null (window kernel): per-channel sums of the stored tile (bias gradients)
    const int* row_span;   // ragged batch (B == 1, H == total rows): [H][2] = first / one-past-last row of the utterance
                           // that owns each row; image borders come from this table instead of row % H
    float* part;           // split-K (ksplit > 1): [ksplit][M][Cout] raw partial sums; bias/ReLU/mask happen in the reduction
    int ksplit;            // window kernel: the channel slabs are divided over `ksplit` workgroups per output tile
    float* y;              // [B,H,W,Cout]
    int B, H, W, Cin, Cout;
    int M;                 // B*H*W
    int mtiles, ntiles, tiles_per_xcd;
    int relu;
    // fused one-channel producer (window kernel, C1 variant): x is not read; its values are
    // relu(b1[c] + sum_tap w1[c][tap] * mel[pixel + tap]) computed while the window is staged
    int x_packed, y_packed;   // 1 (split8 launches): x / y are PACKED activation tensors (dasv_common.hpp, store_x8) instead of fp32;
                              // 2 (fp16-operand launches): x / y are fp16 NHWC tensors holding v * 2^E, E in the range record
    const uint32_t* x_rec;    // range record of x (dasv_common.hpp): required in the split8 mode — the operand planes are scaled by it
    uint32_t* y_rec;          // range record of y or null: [0] receives max |y| (atomic maximum), [1] the exponent of a packed y
    const float* mel;      // [B,H,W]
    const float* w1;       // [Cin][9]
    const float* b1;       // [Cin] or null
    // fused 2x2 stride-2 ceil-mode max-pool (CNNs.py:40,44,48 / 74,78,82,86 behind convX2: `pool(relu(convX2(.)))` in one launch,
    // pipelined launches only): y is the POOLED tensor [B, ceil(H/2), ceil(W/2), Cout] in the form y_packed names.  A tile then
    // covers `pool_units` whole ROW PAIRS (units; the last unit of an image with an odd H is a single row) instead of TBM raster
    // pixels, so that every 2x2 window lies inside one tile: pool_units * 2 * W <= TBM, pool_U = ceil(H/2) units per image.
    int pool, pool_units, pool_U;
    // data gradient of the SECOND layer with the FIRST layer's backward in its write-out pass (split8, pipelined launches): the first
    // layer has one input channel (CNNs.py:38 / 72, conv11), so its weight gradient is gw1[c][tap] = sum_pix g1[pix][c] mel[pix + tap]
    // with g1 = this launch's masked output — 9 FMAs per element while the tile is in hand — and its bias gradient the column sum.
    // c1_part: [mtiles][Cout][10] per-tile sums {9 taps, bias} (dasv_conv3x3_c1_bwd_reduce_parts adds them up); y may then be null:
    // the gradient at the first layer's output (786 MB for a [64,300,80] batch) is needed by nobody else
    const float* c1_mel;   // [B,H,W]
    float* c1_part;
    uint8_t* pool_idx;     // (training, fp32 y) [B, ceil(H/2), ceil(W/2), Cout] bytes {arg-max position 0-3 | 4 if the maximum is > 0}: what
                           // the pool's backward needs (frontend_misc.hip, maxpool2x2_bwd_nhwc_kernel) — the full-resolution map is
                           // then needed by nobody
};

// LDS: four planes of 128 rows x 128 bytes (64 bf16): A_hi | A_lo | B_hi | B_lo.  16-byte slot s of row r lives at
// r*128 + ((s ^ (r & 7)) * 16).
__device__ __forceinline__ int lds_off(int row, int slot) { return row * 128 + ((slot ^ (row & 7)) << 4); }

template <int PREC>
__global__ void __launch_bounds__(256) conv3x3_igemm_kernel(const ConvParams p) {
    constexpr bool SPLIT = PREC == PREC_SPLIT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ahi = smem;
    char* Alo = smem + 16384;
    char* Bhi = smem + 32768;
    char* Blo = smem + 49152;

    // XCD-aware tile order: workgroups that share blockIdx % 8 walk consecutive tiles (neighbouring tiles share
    // the input rows / the weight slab in that XCD's L2).  Speed only.
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int tile = xcd * p.tiles_per_xcd + idx;
    if (idx >= p.tiles_per_xcd || tile >= p.mtiles * p.ntiles) return;
    const int mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
    const int m0 = mt * CBM, n0 = nt * CBN;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int H = p.H, W = p.W, Cin = p.Cin;

    // ---- A staging: thread handles fp32 chunk (16 B = 4 channels) `ach` of rows arow0 + 16*i, i < 8
    const int ach = tid & 15, arow0 = tid >> 4;
    const float* aptr[8];   // &x[pixel][ach*4] of the output pixel (centre tap)
    unsigned amask[8];      // bit `tap` set: that tap's source pixel is inside the image (0 beyond M)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = m0 + arow0 + 16 * i;
        unsigned mask = 0;
        if (m < p.M) {
            const int w_ = m % W, h_ = (m / W) % H;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int hh = h_ + tap / 3 - 1, ww = w_ + tap % 3 - 1;
                if ((unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W) mask |= 1u << tap;
            }
        }
        amask[i] = mask;
        aptr[i] = p.x + (size_t)(m < p.M ? m : 0) * Cin + ach * 4;
    }
    // ---- B staging: thread handles 16-byte chunk `bch` (8 bf16) of rows brow0 + 32*i, i < 4 (per plane)
    const int bch = tid & 7, brow0 = tid >> 3;

    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    f32x4 ra[8];
    u32x4 rbh[4], rbl[4];

    auto load_step = [&](int ks) {
        const int csteps = Cin / CBK;
        const int tap = ks / csteps, c0 = (ks - tap * csteps) * CBK;
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        const long long aoff = (long long)(dy * W + dx) * Cin + c0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((amask[i] >> tap) & 1u) v = *reinterpret_cast<const f32x4*>(aptr[i] + aoff);
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const size_t off = ((size_t)(n0 + brow0 + 32 * i) * 9 + tap) * Cin + c0 + bch * 8;
            rbh[i] = *reinterpret_cast<const u32x4*>(p.whi + off);
            if (SPLIT) rbl[i] = *reinterpret_cast<const u32x4*>(p.wlo + off);
        }
    };
    auto store_step = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = arow0 + 16 * i;
            uint32_t h0, l0, h1, l1;
            cvt2<PREC>(ra[i][0], ra[i][1], h0, l0);
            cvt2<PREC>(ra[i][2], ra[i][3], h1, l1);
            const int o = lds_off(row, ach >> 1) + (ach & 1) * 8;
            *reinterpret_cast<uint2*>(Ahi + o) = make_uint2(h0, h1);
            if (SPLIT) *reinterpret_cast<uint2*>(Alo + o) = make_uint2(l0, l1);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = brow0 + 32 * i;
            const int o = lds_off(row, bch);
            *reinterpret_cast<u32x4*>(Bhi + o) = rbh[i];
            if (SPLIT) *reinterpret_cast<u32x4*>(Blo + o) = rbl[i];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ksteps = 9 * (Cin / CBK);
    const int frow = lane & 15, fq = lane >> 4;   // fragment row within a 16-row block, k-quarter
    load_step(0);
    for (int ks = 0; ks < ksteps; ++ks) {
        store_step();
        __syncthreads();
        if (ks + 1 < ksteps) load_step(ks + 1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int slot = kk * 4 + fq;
            bf16x8 bh[4], bl[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int o = lds_off(wn * 64 + j * 16 + frow, slot);
                bh[j] = *reinterpret_cast<const bf16x8*>(Bhi + o);
                if (SPLIT) bl[j] = *reinterpret_cast<const bf16x8*>(Blo + o);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int o = lds_off(wm * 64 + i * 16 + frow, slot);
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(Ahi + o);
                bf16x8 al;
                if (SPLIT) al = *reinterpret_cast<const bf16x8*>(Alo + o);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[i][j] = mma16<PREC>(ah, bh[j], acc[i][j]);
                    if (SPLIT) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[j], acc[i][j], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    }

    // epilogue: C/D layout of the 16x16 tile: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + (lane & 15);
        const float bv = p.bias != nullptr ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
                if (m < p.M) {
                    float v = acc[i][j][r] + bv;
                    if (p.relu) v = relu_nan(v);
                    if (p.mask != nullptr && !(p.mask[(size_t)m * p.Cout + n] > 0.f)) v = 0.f;
                    p.y[(size_t)m * p.Cout + n] = v;
                }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------
// Window variant (the fast path for W <= 80): per 64-channel slab the workgroup stages ONE contiguous raster
// window of the input — its 256 output pixels plus W+1 pixels of halo on each side — and runs all nine taps out
// of LDS: the A fragment of tap (dy,dx) for output pixel m is simply window row (m - m0) + (W+1) + dy*W + dx, and a
// 9-bit per-pixel mask zeroes taps that fall outside the image (left/right edge, top/bottom, neighbouring image).
// Each input element is fetched from HBM/L2, split and written to LDS once per slab instead of once per tap, and the
// weights of a tap are staged once per 256 pixels.  8 waves (4 x 2), each 64 x 64 outputs.
// LDS: A_hi | A_lo windows of `wrows` x 128 B, then B_hi | B_lo tiles of 128 x 128 B.
constexpr int WBM = 256, WMAXROWS = WBM + 2 * 80 + 2, WACH = (WMAXROWS * 16 + 511) / 512;

// C1: the input activations are produced on the fly by a one-input-channel conv3x3+bias+ReLU over `mel` (conv11 fused
// into conv12: its 786 MB output never exists in HBM).
// IB = 16-pixel blocks per wave along the pixels: 4 -> 256-pixel tiles; 3 -> 192-pixel tiles, whose window leaves room for both
// tile buffers on 80-column maps (90.6 KB + 64 KB), so conv12 runs the pipelined double-buffered loops too.
// WM = waves along the pixels (x NWN along the 128 output channels).  WM = 2, NWN = 4, IB = 5: eight waves of 80 pixels x 32 channels
// = 160-pixel tiles — a whole ROW PAIR of an 80-column map (and 2 / 4 / 8 pairs of the 40- / 20- / 10-column ones), the tile shape of
// the fused max-pool epilogue (ConvParams::pool) that wastes no pixel rows of the matrix work.
// GEOM = 1 (PATCH tiles, maps whose width is a multiple of 16 — the 80-column first block): a tile is a 16 x 16 pixel patch of ONE
// image instead of 256 raster pixels.  Its window is the 18 x 18 patch around it — 324 rows instead of 256 + 2 W + 2 = 418 on an
// 80-column map — which leaves room for both weight-tile buffers beside a 256-pixel tile (the raster form needs 192-pixel tiles
// there), and every 2 x 2 pooling window lies inside the tile, so the fused max-pool wastes no pixel rows.  A 16-pixel MFMA block is
// one patch row: the A fragment of tap (dy, dx) for block y is window rows (y + 1 + dy) * 18 + 1 + dx ... + 15, blocks 18 rows
// apart; the LDS swizzle key is the window COLUMN (the same for every block, so block addresses stay base + constant).
template <int PREC, bool DBUF, bool C1, int NWN, int IB = 4, int WM = 4, int GEOM = 0>
__global__ void __launch_bounds__(64 * WM * NWN) conv3x3_window_kernel(const ConvParams p) {
    static_assert(GEOM == 0 || (WM == 4 && IB == 4 && NWN == 2 && DBUF && !C1), "patch tiles: the pipelined 4 x 2 wave layout");
    constexpr int PW = 18;                  // patch window pitch (GEOM = 1)
    constexpr int ABLK = GEOM ? PW * 128 : 2048;      // bytes between the A fragments of consecutive 16-pixel blocks of a wave
    static_assert((WM == 4 && (IB == 4 || (IB == 3 && NWN == 2 && !C1))) || (WM == 2 && NWN == 4 && IB == 5 && !C1 && DBUF),
                  "256-pixel tiles; 192-pixel tiles: 8-wave layout, activations from memory; 160-pixel tiles: 2 x 4 waves, pipelined");
    constexpr int TBM = 16 * IB * WM;   // pixels per workgroup tile (WM waves along the pixels x IB blocks of 16)
    // NWN waves along the 128 output channels (x 4 along the 256 pixels): 2 -> 8 waves of 64 x 64 outputs, 4 -> 16 waves of
    // 64 x 32 (more waves per SIMD to overlap each other's barrier / fragment-read waits, at more LDS reads per MFMA)
    static_assert(NWN == 2 || NWN == 4, "wave layout");
    static_assert(!C1 || NWN == 2, "the fused first layer stages with the 512-thread layout");
    constexpr int NT = 64 * WM * NWN, JT = 8 / NWN, NPW = 128 / NWN, WACHN = ((TBM + 2 * 80 + 2) * 16 + NT - 1) / NT, BCH = 1024 / NT;
    constexpr bool W8 = WM * NWN == 8;      // eight waves: the pipelined loops (the tile's DMA is divided over eight waves)
    constexpr bool X8 = PREC == PREC_X8;
    constexpr bool SPLIT = PREC == PREC_SPLIT;
    constexpr bool H16 = PREC == PREC_F16 || PREC == PREC_BF16;      // modes with a half-precision STORAGE chain (x_packed / y_packed == 2)
    constexpr bool TWO = SPLIT || X8;      // two LDS planes per operand: bf16 hi | lo, or fp16 hi | the two e4m3 planes
    static_assert(!X8 || (W8 && !C1), "the fp16 + e4m3 mode exists for the 8-wave layouts, activations from memory");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int W = p.W, H = p.H, Cin = p.Cin;
    const int wrows = GEOM ? PW * PW : TBM + 2 * W + 2;
    char* Ahi = smem;
    char* Alo = Ahi + (size_t)wrows * 128;   // X8: rows of { e4m3(v) x 64 | e4m3(r * 2^12) x 64 }
    char* Bbase = Alo + (TWO ? (size_t)wrows * 128 : 0);   // [DBUF ? 2 : 1] x { B_hi 16 KB | B_lo 16 KB }
    // 128 zero bytes behind the tile buffers (pipelined loops): a border tap reads its fragment from here — ONE address select per
    // fragment instead of a data select per fragment register (probe: the selects cost 45 ns of a 1.9 us tap)
    char* Zrow = Bbase + (DBUF ? 65536 : 32768);
    if (DBUF && W8 && !C1 && threadIdx.x < 8)
        *reinterpret_cast<u32x4*>(Zrow + threadIdx.x * 16) = u32x4{0u, 0u, 0u, 0u};   // visible after the first barrier of the loop

    const int per_split = p.tiles_per_xcd * 8;
    const int ks = blockIdx.x / per_split, inner = blockIdx.x - ks * per_split;
    const int xcd = inner & 7;
    // (Measured and dropped, round 2: PERSISTENT tiles — one workgroup per CU walking its XCD's tiles in a loop, the thread index
    // re-read behind an empty asm per tile so that nothing is hoisted out of the tile loop (otherwise ~50 vector registers are and
    // the tap loops spill).  Bit-identical, no spills, and 0-7 % SLOWER per layer in the same-box A/B: the 3.5-4.5 us that
    // tools/conv_stamps.py shows between a tile's last stamp and the launch time per round are not idle CU time — the stamping wave
    // is not the slowest one, and the hardware dispatcher's dynamic placement beats the static stride.)
    const int idx = inner >> 3;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv / NWN, wn = wv % NWN;
    const int frow = lane & 15, fq = lane >> 4;
    const int tile = xcd * p.tiles_per_xcd + idx;
    if (idx >= p.tiles_per_xcd || tile >= p.mtiles * p.ntiles) return;
#ifndef DASV_CONV_ORDER
#define DASV_CONV_ORDER 0
#endif
#if DASV_CONV_ORDER == 1     // experiment: pixel tiles fastest (an XCD sweeps the pixel tiles of one weight slab set)
    const int nt = tile / p.mtiles, mt = tile - nt * p.mtiles;
#elif DASV_CONV_ORDER == 2   // experiment: groups of 4 pixel tiles x all channel tiles
    const int grp = tile / (4 * p.ntiles), rem = tile - grp * 4 * p.ntiles;
    const int gsz = min(4, p.mtiles - grp * 4);
    const int nt = rem / gsz, mt = grp * 4 + rem - nt * gsz;
#else
    const int mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
#endif
    int m0 = mt * TBM, mend = p.M;
    const int n0 = nt * CBN;
    int pu0 = 0, pu1 = 0;          // fused pool: the tile's units [pu0, pu1)
    int gb = 0, gh0 = 0, gw0 = 0;  // patch tiles: image, first row and first column of the 16 x 16 patch
    if constexpr (GEOM == 1) {
        const int txn = W >> 4, per_img = ((H + 15) >> 4) * txn;
        gb = mt / per_img;
        const int r = mt - gb * per_img, ty = r / txn;
        gh0 = ty << 4;
        gw0 = (r - ty * txn) << 4;
        m0 = (gb * H + gh0) * W + gw0;
    } else if (p.pool) {
        pu0 = mt * p.pool_units;
        pu1 = min(pu0 + p.pool_units, p.B * p.pool_U);
        const int b0 = pu0 / p.pool_U, b1 = pu1 / p.pool_U;
        m0 = (b0 * H + 2 * (pu0 - b0 * p.pool_U)) * W;
        mend = pu1 == p.B * p.pool_U ? p.M : (b1 * H + 2 * (pu1 - b1 * p.pool_U)) * W;
    }
    DASV_STAMP(0);

    // border masks of this lane's four A-fragment pixels
    unsigned amask[IB];
#pragma unroll
    for (int i = 0; i < IB; ++i) {
        if constexpr (GEOM == 1) {
            const int h_ = gh0 + wm * IB + i, w_ = gw0 + frow;      // block i of this wave = patch row wm * IB + i
            unsigned mask = 0;
            if (h_ < H) {
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int hh = h_ + tap / 3 - 1, ww = w_ + tap % 3 - 1;
                    if ((unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W) mask |= 1u << tap;
                }
            }
            amask[i] = mask;
            continue;
        }
        const int m = m0 + wm * (16 * IB) + i * 16 + frow;
        unsigned mask = 0;
        if (m < mend) {
            const int w_ = m % W;
            int h_ = (m / W) % H, hend = H;
            if (p.row_span != nullptr) {      // ragged: row m / W belongs to the utterance spanning rows [lo, hi)
                const int r = m / W, lo = p.row_span[2 * r];
                h_ = r - lo;
                hend = p.row_span[2 * r + 1] - lo;
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int hh = h_ + tap / 3 - 1, ww = w_ + tap % 3 - 1;
                if ((unsigned)hh < (unsigned)hend && (unsigned)ww < (unsigned)W) mask |= 1u << tap;
            }
        }
        amask[i] = mask;
    }

    // split8: power-of-two scales from the range records (wave-uniform scalar loads).  sx = 2^Ex scales the activations into the
    // operand planes' range (a packed x was made with the exponent in its record; an fp32 x is scaled while it is staged, by the
    // exponent of its recorded maximum); the accumulators come out in units of 2^-(Ex + Ew[n]).  A packed OUTPUT needs its scale
    // before any of it exists: |y| <= amax_x * max_n sum_k |w[n][k]| + max |b| (both left behind the weight planes by the pack kernel).
    float sx = 1.f, sxinv = 1.f, sy = 1.f;
    const float* winv = nullptr;
    if constexpr (X8) {
        const uint8_t* aux = reinterpret_cast<const uint8_t*>(p.wlo) + (size_t)p.Cout * 9 * Cin * 2;
        winv = reinterpret_cast<const float*>(aux + (size_t)p.Cout * 4);
        const uint32_t xa = p.x_rec[0];
        const int Ex = p.x_packed ? (int)p.x_rec[1] : x8_exp_for(xa);
        sx = x8_pow2(Ex);
        sxinv = x8_pow2(-Ex);
        if (p.y_packed) {
            const float* tail = reinterpret_cast<const float*>(aux + (size_t)p.Cout * 8);
            const float bound = (__uint_as_float(xa & 0x7fffffffu) * tail[0] + tail[1]) * 1.001f;   // margin: the sums are rounded
            const int Ey = x8_exp_for(__float_as_uint(bound));
            sy = x8_pow2(Ey);
            if (blockIdx.x == 0 && threadIdx.x == 0) p.y_rec[1] = (uint32_t)Ey;
        }
    }
    if constexpr (PREC == PREC_F16) {
        // fp16 activation chain: a half-precision x holds v * 2^Ex (Ex in its record), a half-precision y is written as v * 2^Ey
        // with Ey from the bound on the layer's output (see above; the bounds sit behind the fp16 weight plane)
        if (p.x_packed == 2) sxinv = x8_pow2(-(int)p.x_rec[1]);
        if (p.y_packed == 2) {
            const float* tail = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.whi) + (size_t)p.Cout * 9 * Cin * 2);
            const float bound = (__uint_as_float(p.x_rec[0] & 0x7fffffffu) * tail[0] + tail[1]) * 1.001f;
            // fp16 holds 65504: the bound goes to [2^14, 2^15) — the true maximum sits a few binades below it
            const int Ey = x8_exp_for(__float_as_uint(bound)) + 7;
            sy = x8_pow2(Ey > 126 ? 126 : Ey);
            if (blockIdx.x == 0 && threadIdx.x == 0) p.y_rec[1] = (uint32_t)(Ey > 126 ? 126 : Ey);
        }
    }
    uint32_t amax_u = 0u;   // max of the sign-cleared bits of the values this lane stores (range record of y)
    // what the record held when this tile began: only a larger maximum is posted (dasv_common.hpp, range_note_block)
    __shared__ uint32_t s_amax;
    if (threadIdx.x == 0) s_amax = 0u;      // visible behind the first barrier of the loops
    const uint32_t rec_seen = p.y_rec != nullptr ? range_seen(p.y_rec) : 0xffffffffu;
    auto note4 = [&](const f32x4& v) {
        const uint32_t a = __float_as_uint(v[0]) & 0x7fffffffu, b = __float_as_uint(v[1]) & 0x7fffffffu;
        const uint32_t c = __float_as_uint(v[2]) & 0x7fffffffu, d = __float_as_uint(v[3]) & 0x7fffffffu;
        amax_u = max(max(amax_u, a), max(b, max(c, d)));
    };
    auto relu4 = [&](f32x4 v) { return f32x4{relu_nan(v[0]), relu_nan(v[1]), relu_nan(v[2]), relu_nan(v[3])}; };

    // A window staging: chunk q = tid + NT*k covers window row q/16, fp32 chunk q%16 (4 channels).  NT is a multiple of 16, so a
    // thread's chunks all sit in channel column tid & 15 and in rows (tid >> 4) + (NT/16) k: every global address is one running
    // pointer + k * (NT/16) * Cin floats and every LDS address one base + k * (NT/16) * 128 bytes (row & 7, which the swizzle uses,
    // does not change either).  Written that way on purpose: as independent expressions the compiler hoisted ~5 registers of
    // addresses per chunk out of the slab loop (70 registers for 14 chunks) and spilled them once the tap loop needed more.
    constexpr int RSTEP = NT / 16;
    const int wchunks = wrows * 16;
    const int wach = tid & 15, wr0 = tid >> 4;
    const long long wbase = (long long)m0 - W - 1;   // global pixel index of window row 0
    const long long pix0 = wbase + wr0;
    const float* xrun = C1 ? nullptr : p.x + pix0 * Cin + wach * 4;   // advanced by 64 channels per slab; dereferenced only under `valid`
    const size_t xstep = (size_t)RSTEP * Cin;
    const int whi0 = lds_off(wr0, wach >> 1) + (wach & 1) * 8;
    const int wl00 = lds_off(wr0, wach >> 2) + (wach & 3) * 4, wl10 = lds_off(wr0, 4 + (wach >> 2)) + (wach & 3) * 4;
    f32x4 ra[WACHN];
    // Rows of the window in groups of RSTEP: every thread takes part in the first `nfull` groups, the threads with wr0 < wrows %
    // RSTEP in one more: per chunk one wave-uniform test instead of a per-lane predicate with its exec-mask branch.  Window rows
    // outside the tensor (first / last tiles) are never used — a tap that falls outside the image reads the zero row or is
    // deselected — so their loads only need a mapped address: the pixel index is clamped.  The index is re-read behind an empty asm
    // per slab: as a slab-invariant expression the compiler computes all the clamped addresses once per tile and holds 2 registers
    // per chunk across the tap loops, which have none to spare.
    const int nfull = wrows / RSTEP;
    const bool wtail = wr0 < wrows - nfull * RSTEP;
    // patch tiles: window row wr = py * 18 + px <-> image pixel (gh0 - 1 + py, gw0 - 1 + px), clamped into the image (a tap whose
    // source pixel lies outside the image never reads its window row: border mask -> zero row); (wr * 3641) >> 16 == wr / 18 below 600
    auto patch_pix = [&](int wr, int& px) {
        const int py = (wr * 3641) >> 16;
        px = wr - PW * py;
        const int gh = min(max(gh0 - 1 + py, 0), H - 1), gw = min(max(gw0 - 1 + px, 0), W - 1);
        return (gb * H + gh) * W + gw;
    };
    auto load_window = [&](const float* xs) {
        int pb = (int)pix0;
        asm volatile("" : "+v"(pb));
        const float* xc = xs - pix0 * Cin;          // = x + c0 + wach * 4
        if constexpr (GEOM == 1) {
            int w0r = wr0;
            asm volatile("" : "+v"(w0r));           // (per slab, not hoisted: see below)
            const bool half = H16 && p.x_packed == 2;      // 16-bit tensor: a pixel's slab is 128 bytes = chunks 0-7
            const char* xb = reinterpret_cast<const char*>(p.x) + (half ? ((size_t)((int)(xc - p.x) - wach * 4) * 2 + (size_t)(wach & 7) * 16) : 0);
#pragma unroll
            for (int k = 0; k < WACHN; ++k) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (k < nfull || (k == nfull && wtail)) {
                    int px;
                    const int pix = patch_pix(w0r + RSTEP * k, px);
                    if (half) { if (wach < 8) v = *reinterpret_cast<const f32x4*>(xb + (size_t)pix * Cin * 2); }
                    else v = *reinterpret_cast<const f32x4*>(xc + (size_t)pix * Cin);
                }
                ra[k] = v;
            }
            return;
        }
        if constexpr (H16) {
            if (p.x_packed == 2) {
                // half-precision x: a pixel's 64-channel slab is 128 bytes = chunks 0-7; the staging is a copy (threads with
                // wach >= 8 have nothing to fetch)
                const int c0 = (int)(xc - p.x) - wach * 4;
                const char* xb = reinterpret_cast<const char*>(p.x) + (size_t)c0 * 2 + (size_t)(wach & 7) * 16;
#pragma unroll
                for (int k = 0; k < WACHN; ++k) {
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (wach < 8 && (k < nfull || (k == nfull && wtail))) {
                        const int pix = min(max(pb + RSTEP * k, 0), p.M - 1);
                        v = *reinterpret_cast<const f32x4*>(xb + (size_t)pix * Cin * 2);
                    }
                    ra[k] = v;
                }
                return;
            }
        }
#pragma unroll
        for (int k = 0; k < WACHN; ++k) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};      // (defined on every path: a conditionally loaded ra[k] would stay live across the tap loops)
            if (k < nfull || (k == nfull && wtail)) {
                const int pix = min(max(pb + RSTEP * k, 0), p.M - 1);
                v = *reinterpret_cast<const f32x4*>(xc + (size_t)pix * Cin);
            }
            ra[k] = v;
        }
    };
    auto store_window = [&]() {
        if constexpr (GEOM == 1) {
            // LDS row = window row, 16-byte slot XORed with the window COLUMN (px & 7): the 16 lanes of a fragment read consecutive
            // columns, and every block of a wave reads the same columns
            int w0r = wr0;
            asm volatile("" : "+v"(w0r));           // (recomputed per slab: hoisted, the eleven row offsets would live across the tap loops)
#pragma unroll
            for (int k = 0; k < WACHN; ++k) {
                if (k < nfull || (k == nfull && wtail)) {
                    const int wr = w0r + RSTEP * k, py = (wr * 3641) >> 16, key = (wr - PW * py) & 7;
                    char* rowh = Ahi + wr * 128;
                    char* rowl = Alo + wr * 128;
                    if (X8 && p.x_packed) {
                        *reinterpret_cast<f32x4*>((wach < 8 ? rowh : rowl) + (((wach & 7) ^ key) << 4)) = ra[k];
                    } else if (H16 && p.x_packed == 2) {
                        if (wach < 8) *reinterpret_cast<f32x4*>(rowh + ((wach ^ key) << 4)) = ra[k];
                    } else {
                        uint32_t h0, l0, h1, l1;
                        if constexpr (X8) {
                            cvt4_x8(ra[k], sx, h0, h1, l0, l1);
                            *reinterpret_cast<uint32_t*>(rowl + (((wach >> 2) ^ key) << 4) + (wach & 3) * 4) = l0;
                            *reinterpret_cast<uint32_t*>(rowl + (((4 + (wach >> 2)) ^ key) << 4) + (wach & 3) * 4) = l1;
                        } else {
                            cvt2<PREC>(ra[k][0], ra[k][1], h0, l0);
                            cvt2<PREC>(ra[k][2], ra[k][3], h1, l1);
                            if (SPLIT) *reinterpret_cast<uint2*>(rowl + (((wach >> 1) ^ key) << 4) + (wach & 1) * 8) = make_uint2(l0, l1);
                        }
                        *reinterpret_cast<uint2*>(rowh + (((wach >> 1) ^ key) << 4) + (wach & 1) * 8) = make_uint2(h0, h1);
                    }
                }
            }
            return;
        }
        if (X8 && p.x_packed) {
            // the producer already made the planes: chunks 0-7 of a pixel's slab are the fp16 row, 8-11 / 12-15 the e4m3 rows.
            // (Measured and dropped, round 2: the packed window by LDS-DMA, issued EARLY — behind the last tap's MFMAs, once
            // every wave holds that tap's fragments, one extra barrier per slab.  Bit-identical once out-of-tensor rows were
            // zero-filled, but 0-5 % SLOWER per layer: a wave issues ~10 one-KB pieces per slab in the middle of its MFMA
            // stream, and their issue cost — 60-185 cycles each beside MFMAs and fragment reads — eats the latency it hides.)
            char* dst = (wach < 8 ? Ahi : Alo) + lds_off(wr0, wach & 7);
#pragma unroll
            for (int k = 0; k < WACHN; ++k)
                if (k < nfull || (k == nfull && wtail)) *reinterpret_cast<f32x4*>(dst + k * (RSTEP * 128)) = ra[k];
            return;
        }
        if constexpr (H16) {
            if (p.x_packed == 2) {
                if (wach < 8) {
                    char* dst = Ahi + lds_off(wr0, wach);
#pragma unroll
                    for (int k = 0; k < WACHN; ++k)
                        if (k < nfull || (k == nfull && wtail)) *reinterpret_cast<f32x4*>(dst + k * (RSTEP * 128)) = ra[k];
                }
                return;
            }
        }
#pragma unroll
        for (int k = 0; k < WACHN; ++k) {
            if (k < nfull || (k == nfull && wtail)) {
                uint32_t h0, l0, h1, l1;
                if constexpr (X8) {
                    cvt4_x8(ra[k], sx, h0, h1, l0, l1);
                    *reinterpret_cast<uint32_t*>(Alo + wl00 + k * (RSTEP * 128)) = l0;
                    *reinterpret_cast<uint32_t*>(Alo + wl10 + k * (RSTEP * 128)) = l1;
                } else {
                    cvt2<PREC>(ra[k][0], ra[k][1], h0, l0);
                    cvt2<PREC>(ra[k][2], ra[k][3], h1, l1);
                    if (SPLIT) *reinterpret_cast<uint2*>(Alo + whi0 + k * (RSTEP * 128)) = make_uint2(l0, l1);
                }
                *reinterpret_cast<uint2*>(Ahi + whi0 + k * (RSTEP * 128)) = make_uint2(h0, h1);
            }
        }
    };
    // C1: compute this thread's window chunks from the one-channel input and park them directly (no big register
    // batch: the nine mel taps are L1/L2 hits).  The thread's four channels are the same for all its chunks.
    auto stage_window_c1 = [&](int c0) {
        const int ach = tid & 15;
        float wr[4][9];
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int t = 0; t < 9; ++t) wr[c][t] = p.w1[(size_t)(c0 + ach * 4 + c) * 9 + t];
            if (p.b1 != nullptr) bv[c] = p.b1[c0 + ach * 4 + c];
        }
        // window row of chunk k is (tid >> 4) + 32 k; track (h, w) of its pixel incrementally
        long long pix = wbase + (tid >> 4);
        int w_ = 0, h_ = 0;
        if (pix >= 0) { w_ = (int)(pix % W); h_ = (int)((pix / W) % H); }
#pragma unroll 2
        for (int k = 0; k < WACH; ++k) {
            const int q = tid + 512 * k;
            if (q < wchunks) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (pix >= 0 && pix < p.M) {
                    float in[9];
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const int hh = h_ + t / 3 - 1, ww = w_ + t % 3 - 1;
                        in[t] = ((unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W) ? p.mel[pix + (t / 3 - 1) * W + (t % 3 - 1)] : 0.f;
                    }
                    v = bv;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int t = 0; t < 9; ++t) v[c] = fmaf(in[t], wr[c][t], v[c]);
                    v = __builtin_elementwise_max(v, f32x4{0.f, 0.f, 0.f, 0.f});
                }
                uint32_t h0, l0, h1, l1;
                cvt2<PREC>(v[0], v[1], h0, l0);
                cvt2<PREC>(v[2], v[3], h1, l1);
                const int o = lds_off(q >> 4, ach >> 1) + (ach & 1) * 8;
                *reinterpret_cast<uint2*>(Ahi + o) = make_uint2(h0, h1);
                if (SPLIT) *reinterpret_cast<uint2*>(Alo + o) = make_uint2(l0, l1);
            }
            if (pix >= 0) {
                w_ += 32;
                while (w_ >= W) { w_ -= W; if (++h_ == H) h_ = 0; }
                pix += 32;
            } else {
                pix += 32;
                if (pix >= 0) { w_ = (int)(pix % W); h_ = (int)((pix / W) % H); }
            }
        }
    };
    // B staging: 128 rows x 8 chunks per plane = 1024 chunks; BCH chunks per thread
    u32x4 rbh[BCH], rbl[BCH];
    auto load_b = [&](int tap, int c0) {
#pragma unroll
        for (int k = 0; k < BCH; ++k) {
            const int q = tid + NT * k;
            const size_t off = ((size_t)(n0 + (q >> 3)) * 9 + tap) * Cin + c0 + (q & 7) * 8;
            rbh[k] = *reinterpret_cast<const u32x4*>(p.whi + off);
            if (TWO) rbl[k] = *reinterpret_cast<const u32x4*>(p.wlo + off);   // X8: the 128-byte e4m3 rows have the same pitch
        }
    };
    auto store_b = [&](char* Bhi) {
#pragma unroll
        for (int k = 0; k < BCH; ++k) {
            const int q = tid + NT * k;
            const int o = lds_off(q >> 3, q & 7);
            *reinterpret_cast<u32x4*>(Bhi + o) = rbh[k];
            if (TWO) *reinterpret_cast<u32x4*>(Bhi + 16384 + o) = rbl[k];
        }
    };
    // Pipelined loops: the weight tile goes global -> LDS by LDS-DMA (global_load_lds_dwordx4: no registers, no ds_write; the
    // probe put the tile stores at 75 ns of a tap).  A wave-instruction writes 1 KB = 8 rows x 8 physical slots in lane order, so
    // the XOR swizzle moves to the SOURCE: lane L lands at (row L >> 3, slot L & 7) and fetches logical chunk (L & 7) ^ (L >> 3).
    // Wave w moves pieces 2w, 2w+1 of both planes (rows 16w ... 16w+15).  The DMA is counted in vmcnt: the __syncthreads() in
    // front of the barrier after which the tile is read drains it.
    // Every wave waits for its OWN pieces in front of the barrier behind which the tile is read: `s_waitcnt vmcnt(0)` (gfx9 encoding:
    // vmcnt 0, expcnt 7, lgkmcnt 15).  The compiler places the same wait for the barrier's fence while the DMA sits in straight-line
    // code, but not once it is under a wave-uniform branch (seen in a round-3 experiment) — the protocol must not depend on that.
    auto dma_drain = [&]() { __builtin_amdgcn_s_waitcnt(0x0F70); };
#ifndef DASV_DMA_ASM
#define DASV_DMA_ASM 0
#endif
    // One 1-KB piece: lane L's 16 bytes at `src` land at `dst` + 16 L.  DASV_DMA_ASM=1 issues it as inline assembly (M0 = LDS base),
    // which the compiler does not book as a FLAT access with an LDS side: while one of ITS OWN LDS-DMAs is pending, every lgkmcnt
    // wait it places is a full lgkmcnt(0), so with a tile always in flight the tap loops drain their fragment reads completely in
    // front of every MFMA group; with the assembly form the waits become counted (lgkmcnt(9), (12), ...).  Measured, same box: the
    // counted waits are SLOWER — fp16 mode +3.9 % (tools/_diag/f16_lib_ab.py), split8 +1.1 % (tools/conv_ab_r3.py); the builtin stays.
    auto dma_piece = [&](const void* src, char* dst) {
#if DASV_DMA_ASM
        const uint32_t l = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)dst;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(l), "v"(src) : "memory");
#else
        typedef __attribute__((address_space(3))) void* lds_ptr_t;
        typedef const __attribute__((address_space(1))) void* glb_ptr_t;
        __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)dst, 16, 0, 0);
#endif
    };
    auto dma_b = [&](char* Bdst, int tap, int c0) {
        typedef __attribute__((address_space(3))) void* lds_ptr_t;
        typedef const __attribute__((address_space(1))) void* glb_ptr_t;
        const int wu = __builtin_amdgcn_readfirstlane(wv);
        const int lrow = lane >> 3, chunk = (lane & 7) ^ lrow;
        if constexpr (!TWO) {
            // one-plane modes: a wave-uniform base (tile, tap, slab: scalar arithmetic) + a 32-bit lane offset — 4 vector instructions
            // per tap instead of the ~24 of the 64-bit per-lane products below (+3.1 % in the fp16 mode, same-box A/B).  In the
            // two-plane modes the cheaper loop body makes the unroller take all nine taps apart (100 spilled registers in split8), and
            // pinned to today's three taps per trip it still spills 35: they keep the long form.
            const char* bh = reinterpret_cast<const char*>(p.whi) + (((size_t)n0 * 9 + tap) * Cin + c0) * 2;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int piece = wu * 2 + k;
                const unsigned vo = (((unsigned)piece * 8u + (unsigned)lrow) * (unsigned)(9 * Cin) + (unsigned)chunk * 8u) * 2u;
                dma_piece(bh + vo, Bdst + piece * 1024);
            }
            return;
        }
        // (Measured and dropped, round 3: the tile moved by waves 0-3 or 0-1 only, so that the two waves of a SIMD stop issuing their
        // DMA pieces at the same time: 1 % slower (tools/conv_ab_r3.py, same box) — and in that form the compiler no longer
        // drained vmcnt in front of the loop's barrier, hence the explicit wait there.  The scalar-base form above for both planes:
        // no spills any more, 8 vector instructions fewer per tap, no measurable change.)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int piece = wu * 2 + k;
            const size_t off = ((size_t)(n0 + piece * 8 + lrow) * 9 + tap) * Cin + c0 + chunk * 8;
            dma_piece(p.whi + off, Bdst + piece * 1024);
            if constexpr (TWO) dma_piece(p.wlo + off, Bdst + 16384 + piece * 1024);
        }
    };
    // X8: E8M0 block scales of the cross-term MFMA.  Block b of its K = 128 takes the scale byte of lane group b: blocks
    // 0 / 2 pair e4m3(x) with the weight residuals, blocks 1 / 3 the activation residuals with e4m3(w) (tools/fp8_mfma_check.hip)
    int wsc[JT];
    const int xsc = (fq & 1) ? 127 - X8_ACT_SHIFT : 127;
    if constexpr (X8) {
        const uint8_t* sc = reinterpret_cast<const uint8_t*>(p.wlo) + (size_t)p.Cout * 9 * Cin * 2;
#pragma unroll
        for (int j = 0; j < JT; ++j) wsc[j] = sc[(size_t)(n0 + wn * NPW + j * 16 + frow) * 4 + (fq & 1)];
    }

    // byte offset (inside a window plane) of this lane's A fragment of tap `tap`, 16-byte slot `slot`, for its wave's block 0
    auto a_off = [&](int tap, int slot) {
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        if constexpr (GEOM == 1) {
            const int col = 1 + dx + frow;
            return ((wm * IB + 1 + dy) * PW + col) * 128 + ((slot ^ (col & 7)) << 4);
        } else {
            return lds_off(wm * (16 * IB) + frow + (W + 1) + dy * W + dx, slot);
        }
    };
    f32x4 acc[IB][JT];
#pragma unroll
    for (int i = 0; i < IB; ++i)
#pragma unroll
        for (int j = 0; j < JT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // split-K: this workgroup covers channel slabs [sl_lo, nslab) of the Cin / 64
    const int sl_lo = (Cin / CBK) * ks / p.ksplit, nslab = (Cin / CBK) * (ks + 1) / p.ksplit;
    constexpr bool PIPELINED = DBUF && W8 && !C1 && DASV_CONV_DIAG == 0;
    if constexpr (PIPELINED) dma_b(Bbase, 0, sl_lo * CBK);
    else load_b(0, sl_lo * CBK);
    int it = 0;
    // PIPE (split-bf16, double-buffered tile, 8 waves, activations from memory): the fragments of a 32-channel half are read one
    // half ahead of the MFMAs that use them, across taps: while half 0 of a tap is multiplied the fragments of half 1 arrive;
    // while half 1 is multiplied the NEXT tap's first-half pixel fragments (same window) are read before the tap's barrier and its
    // weight fragments (new tile) right after it.  The two waves of a SIMD run in step, so without this both read, then both
    // multiply: two exposed read phases per tap.  tools/conv_core_probe.hip (same loop, LDS only): 2049 -> 1882 ns per tap.
    // Needs 128 fragment registers: possible since the window staging addresses stopped occupying ~60 registers (above).
    // (the single-MFMA modes, bf16 / fp16 operands, run the same loop on their one plane per operand)
    constexpr bool PIPE = !X8 && DBUF && W8 && !C1 && DASV_CONV_DIAG == 0;
    if constexpr (PIPE) {
        struct Frags { bf16x8 bh[JT], bl[SPLIT ? JT : 1], ah[IB], al[SPLIT ? IB : 1]; };
        const int brow = wn * NPW + frow;
        const int boff0 = lds_off(brow, fq), boff1 = lds_off(brow, 4 + fq);   // tiles are 16 rows = 2048 bytes apart
        auto read_b = [&](Frags& f, const char* Bhi, int kk) {
            const char* bp = Bhi + (kk ? boff1 : boff0);
#pragma unroll
            for (int j = 0; j < JT; ++j) {
                f.bh[j] = *reinterpret_cast<const bf16x8*>(bp + j * 2048);
                if constexpr (SPLIT) f.bl[j] = *reinterpret_cast<const bf16x8*>(bp + 16384 + j * 2048);
            }
        };
        auto read_a = [&](Frags& f, int tap, int kk) {
            const int o = a_off(tap, kk * 4 + fq);
#pragma unroll
            for (int i = 0; i < IB; ++i) {
                const bool ok = (amask[i] >> tap) & 1u;
                f.ah[i] = *reinterpret_cast<const bf16x8*>(ok ? Ahi + o + i * ABLK : Zrow);
                if constexpr (SPLIT) f.al[i] = *reinterpret_cast<const bf16x8*>(ok ? Alo + o + i * ABLK : Zrow);
            }
        };
        auto mma_rows = [&](const Frags& f, int i0, int i1) {
#pragma unroll
            for (int i = 0; i < IB; ++i) {
                if (i < i0 || i >= i1) continue;
#pragma unroll
                for (int j = 0; j < JT; ++j) {
                    acc[i][j] = mma16<PREC>(f.bh[j], f.ah[i], acc[i][j]);
                    if constexpr (SPLIT) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.bl[j], f.ah[i], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.bh[j], f.al[i], acc[i][j], 0, 0, 0);
                    }
                }
            }
        };
        Frags fa, fb;
        // (Measured and dropped, round 2: the last tap of a slab peeled off the tap loop, with the NEXT slab's window loads issued
        // behind its second-half MFMAs, where half of the fragment registers are dead — tools/conv_stamps.py shows ~1.2 us of global
        // latency exposed at every slab top.  The 256-pixel variants spill 32-44 registers for it and run 7-20 % slower; the
        // 192-pixel ones fit and gain nothing measurable.)
        // (Measured and dropped, round 3: in the one-plane modes a tap's tile is 16 KB, so a 32-KB buffer can take TWO taps and the
        // workgroup then meets at four tile barriers per slab instead of eight — stages {0,1} {2,3} {4,5} {6,7} {8}, bit-identical,
        // and 1.0 % SLOWER over the seven layers (tools/_diag/f16_lib_ab.py; 4 % slower with the pair loop kept rolled).  Like every
        // other change of the tap's schedule this round it moves nothing, because these launches run against the chip's power
        // management, not against their own stalls: tools/_diag/conv_power_test.py — the same launches on all-zero data, same
        // instruction stream and traffic, finish 10-18 % sooner, and the in-kernel clock reads 1.9-2.0 GHz of 2.4 on real data.)
        for (int sl = sl_lo; sl < nslab; ++sl) {
            const int c0 = sl * CBK;
            load_window(xrun + c0);
            if (sl > sl_lo) __syncthreads();   // every wave is done reading the previous slab's window
            store_window();
            {   // first tap of the slab: its tile is on its way by DMA (issued during the previous slab / before the loop)
                char* Bcur = Bbase + ((it & 1) ? 32768 : 0);
                dma_drain();
                __syncthreads();
                dma_b(Bbase + ((it & 1) ? 0 : 32768), 1, c0);
                read_b(fa, Bcur, 0);
                read_a(fa, 0, 0);
            }
            for (int tap = 0; tap < 9; ++tap, ++it) {
                char* Bcur = Bbase + ((it & 1) ? 32768 : 0);
                char* Bnxt = Bbase + ((it & 1) ? 0 : 32768);
                read_b(fb, Bcur, 1);
                read_a(fb, tap, 1);
                mma_rows(fa, 0, IB);
                if (tap < 8) read_a(fa, tap + 1, 0);    // same window: no need to wait for the barrier
                mma_rows(fb, 0, 2);
                // (round 3: forcing the fragment reads between these MFMAs with sched_group_barrier, as in the split8 loop below,
                // costs the fp16 mode 1.7 % — 16 spilled registers — tools/_diag/f16_lib_ab.py)
                if (tap < 8) {
                    dma_drain();
                    __syncthreads();           // the next tap's tile has landed (DMA issued one tap ago, drained just above)
                    // the tile after next goes into the buffer this tap has finished reading
                    if (tap < 7) dma_b(Bcur, tap + 2, c0);
                    else if (sl + 1 < nslab) dma_b(Bcur, 0, c0 + CBK);
                    read_b(fa, Bnxt, 0);
                }
                mma_rows(fb, 2, IB);
            }
        }
    } else if constexpr (X8 && DBUF && W8 && DASV_CONV_DIAG == 0) {
        // The same software pipeline for the fp16 + e4m3 mode.  Per tap three MFMA groups: fp16 half 0, fp16 half 1 (16 MFMAs of 16
        // cycles each), e4m3 cross terms (16 MFMAs of 32 cycles).  The fragments of fp16 half 1 and the e4m3 fragments are read at
        // the top of the tap, while half 0 (fetched during the previous tap) is multiplied; the NEXT tap's half-0 fragments while
        // the e4m3 group runs — pixel part before the tap's barrier, weight part after it.  (Carrying both fp16 halves across the
        // barrier needs 32 more registers than there are: spill reloads inside the tap loop, 3 % slower than no pipeline.)
        const int s1 = 4 * (fq >> 1) + (fq & 1), s2 = s1 + 2;
        const int brow = wn * NPW + frow;
        const int bo0 = lds_off(brow, fq), bo1 = lds_off(brow, 4 + fq);
        const int bq0 = 16384 + lds_off(brow, s1), bq1 = 16384 + lds_off(brow, s2);
        bf16x8 bh0[JT], bh1[JT], ah0[IB], ah1[IB];
        u32x4 wl[JT], wh[JT], xl[IB], xh[IB];
        auto read_f16_b = [&](bf16x8 (&bh)[JT], const char* Bhi, int kk) {
#pragma unroll
            for (int j = 0; j < JT; ++j) bh[j] = *reinterpret_cast<const bf16x8*>(Bhi + (kk ? bo1 : bo0) + j * 2048);
        };
        auto read_f16_a = [&](bf16x8 (&ah)[IB], int tap, int kk) {
            const int o = a_off(tap, kk * 4 + fq);
#pragma unroll
            for (int i = 0; i < IB; ++i)
                ah[i] = *reinterpret_cast<const bf16x8*>(((amask[i] >> tap) & 1u) ? Ahi + o + i * ABLK : Zrow);
        };
        auto read_e4m3 = [&](const char* Bhi, int tap) {
            const int o0 = a_off(tap, s1), o1 = a_off(tap, s2);
#pragma unroll
            for (int j = 0; j < JT; ++j) {
                wl[j] = *reinterpret_cast<const u32x4*>(Bhi + bq0 + j * 2048);
                wh[j] = *reinterpret_cast<const u32x4*>(Bhi + bq1 + j * 2048);
            }
#pragma unroll
            for (int i = 0; i < IB; ++i) {
                const bool ok = (amask[i] >> tap) & 1u;
                xl[i] = *reinterpret_cast<const u32x4*>(ok ? Alo + o0 + i * ABLK : Zrow);
                xh[i] = *reinterpret_cast<const u32x4*>(ok ? Alo + o1 + i * ABLK : Zrow);
            }
        };
        auto mma_f16 = [&](const bf16x8 (&b)[JT], const bf16x8 (&a)[IB]) {
#pragma unroll
            for (int i = 0; i < IB; ++i)
#pragma unroll
                for (int j = 0; j < JT; ++j) acc[i][j] = mma16<PREC_F16>(b[j], a[i], acc[i][j]);
        };
        auto mma_e4m3 = [&](int i0, int i1) {
#pragma unroll
            for (int i = 0; i < IB; ++i) {
                if (i < i0 || i >= i1) continue;
                const i32x8 x8 = {(int)xl[i][0], (int)xl[i][1], (int)xl[i][2], (int)xl[i][3], (int)xh[i][0], (int)xh[i][1], (int)xh[i][2], (int)xh[i][3]};
#pragma unroll
                for (int j = 0; j < JT; ++j) {
                    const i32x8 w8 = {(int)wl[j][0], (int)wl[j][1], (int)wl[j][2], (int)wl[j][3], (int)wh[j][0], (int)wh[j][1], (int)wh[j][2], (int)wh[j][3]};
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w8, x8, acc[i][j], 0, 0, 0, wsc[j], 0, xsc);
                }
            }
        };
        DASV_STAMP(1);
        for (int sl = sl_lo; sl < nslab; ++sl) {
            const int c0 = sl * CBK;
            DASV_STAMP_SLAB(2 + 4 * (sl - sl_lo));
            load_window(xrun + c0);
            if (sl > sl_lo) __syncthreads();   // every wave is done reading the previous slab's window
            DASV_STAMP_SLAB(3 + 4 * (sl - sl_lo));
            store_window();
            {
                char* Bcur = Bbase + ((it & 1) ? 32768 : 0);
                dma_drain();
                __syncthreads();
                DASV_STAMP_SLAB(4 + 4 * (sl - sl_lo));
                dma_b(Bbase + ((it & 1) ? 0 : 32768), 1, c0);
                read_f16_b(bh0, Bcur, 0);
                read_f16_a(ah0, 0, 0);
            }
#ifndef DASV_X8_SCHED
#define DASV_X8_SCHED 1   // 0: the scheduler's own order (A/B with tools/conv_ab_r3.py: the hints are worth 1-2 % over the seven layers)
#endif
            // (Measured and dropped, round 3, same-box A/B with tools/conv_ab_r3.py: the tap as a HAND-ORDERED stream — every MFMA and
            // every fragment read pinned in place with sched_barrier, one read behind each MFMA of the group in front of its consumer,
            // counted lgkmcnt waits only, the tile's DMA branch-free — runs within 0.1 % of the hinted order below; the same stream with
            // the two waves of a SIMD issuing their DMA pieces at different points of the tap (per-role copies of the loop) spills
            // inside the loop and is 24 % slower.  The tap is not waiting for fragment reads: SQ counters over these launches give
            // 32 % of the wave cycles issuing, 36 % stalled at issue behind the busy matrix pipe, 33 % parked at waits and barriers.)
            for (int tap = 0; tap < 9; ++tap, ++it) {
                char* Bcur = Bbase + ((it & 1) ? 32768 : 0);
                char* Bnxt = Bbase + ((it & 1) ? 0 : 32768);
                read_f16_b(bh1, Bcur, 1);
                read_f16_a(ah1, tap, 1);
                read_e4m3(Bcur, tap);
                mma_f16(bh0, ah0);
                mma_f16(bh1, ah1);
                if (tap < 8) read_f16_a(ah0, tap + 1, 0);
                mma_e4m3(0, 2);
#if DASV_X8_SCHED == 1
                // the tap's fragment reads go out BETWEEN the MFMAs of the groups in front of their consumers: one read behind each
                // of the 16 first-half MFMAs (second-half fp16 fragments, then the weight e4m3 fragments), one per two MFMAs
                // after that (pixel e4m3 fragments; the next tap's first-half pixel fragments beside the e4m3 group).  Left alone
                // the scheduler emits each group as a block behind a block of reads and an lgkmcnt(0): both waves of a SIMD read,
                // then both multiply
                if constexpr (WM == 4) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
#pragma unroll
                    for (int k = 0; k < 12; ++k) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                } else {
                    // 5 x 2 tiles per wave: 10 MFMAs per group, 21 reads at the top of the tap + 5 for the next tap's pixels
#pragma unroll
                    for (int k = 0; k < 3 * (IB + JT) + IB; ++k) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
#endif
                if (tap < 8) {
                    dma_drain();
                    __syncthreads();
                    if (tap < 7) dma_b(Bcur, tap + 2, c0);
                    else if (sl + 1 < nslab) dma_b(Bcur, 0, c0 + CBK);
                    read_f16_b(bh0, Bnxt, 0);
                }
                mma_e4m3(2, IB);
            }
            DASV_STAMP_SLAB(5 + 4 * (sl - sl_lo));
        }
    } else
    for (int sl = sl_lo; sl < nslab; ++sl) {
        const int c0 = sl * CBK;
        if (!C1 && (DASV_CONV_DIAG != 1 || sl == sl_lo)) load_window(xrun + c0);
        if (DBUF && sl > sl_lo) __syncthreads();   // every wave is done reading the previous slab's window
        if (C1) stage_window_c1(c0);
        else if (DASV_CONV_DIAG != 1 || sl == sl_lo) store_window();
        for (int tap = 0; tap < 9; ++tap, ++it) {
            // DBUF: the buffer written now was last read two iterations ago, and every wave passed the barrier of
            // the previous iteration only after finishing that read -> ONE barrier per tap.
            char* Bhi = Bbase + ((DBUF && (it & 1)) ? 32768 : 0);
            if (DASV_CONV_DIAG != 3 || it == 0) store_b(Bhi);
            __syncthreads();
            if (DASV_CONV_DIAG == 3) {
            } else if (tap < 8) load_b(tap + 1, c0);
            else if (sl + 1 < nslab) load_b(0, c0 + CBK);
            // Measured alternatives that did NOT help: pinning these loads here with a sched_barrier (the scheduler sinks
            // them to the end of the MFMA section): no change; a two-deep software pipeline (tile of tap+2 fetched at the
            // top, tile of tap+1 written to the other buffer between the two MFMA halves, so no store sits between a
            // barrier and the first fragment reads): 4.5 % slower at 255 registers; fetching the NEXT tap's first-half pixel
            // fragments before the barrier (the window is unchanged within a slab): 2.5 % slower at 251 registers; 128-pixel
            // tiles with a single weight buffer (<= 80 KB of LDS on maps up to 20 columns, so two workgroups share a CU and
            // desynchronise): 4-7 % slower on those layers (128 registers with spills, more halo and fragment reads per MFMA).
            // Round 2, same-box A/B against this code (tools/conv_lib_ab.py): a half-tap stagger of waves 4-7 against their
            // SIMD partners (MI355X_MICROARCH.md, two waves per SIMD, item 9: second-half fragments read before the tap's
            // barrier and held in registers, their MFMAs issued after it beside the partner's reads; bit-identical results,
            // 248 registers, no spills): 3-5 % SLOWER on every double-buffered layer with the phases fenced or left to the
            // scheduler, 6-10 % slower with reads-first hints — the matrix pipe is not waiting for fragment reads here.
            const int shift = (W + 1) + (tap / 3 - 1) * W + (tap % 3 - 1);
            if constexpr (X8) {
                // The fragment reads of the tap go out ahead of the MFMAs that use them, in consumption order: both fp16 halves and
                // the weight e4m3 fragments first (24 ds_read_b128, 96 registers), the pixel e4m3 fragments into the registers the
                // first half's MFMAs have freed.  Only the first group's LDS latency is exposed.  With a read phase in front of each
                // of the three MFMA groups the two waves of a SIMD, which run in step, left the matrix pipe idle three times per
                // tap (1.91 us per tap against 2.09 for split-bf16: the saved pipe cycles were lost again).
                // Addresses: tiles are 16 rows apart and the swizzle uses row & 7, so every fragment is base(operand, slot) + a
                // compile-time tile * 2048.
                const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
                const u32x4 z4 = {0u, 0u, 0u, 0u};
                const int s1 = 4 * (fq >> 1) + (fq & 1), s2 = s1 + 2;
                const int brow = wn * NPW + frow, arow = wm * (16 * IB) + frow + shift;
                const char* bp0 = Bhi + lds_off(brow, fq);
                const char* bp1 = Bhi + lds_off(brow, 4 + fq);
                const char* bq0 = Bhi + 16384 + lds_off(brow, s1);
                const char* bq1 = Bhi + 16384 + lds_off(brow, s2);
                const char* ap0 = Ahi + lds_off(arow, fq);
                const char* ap1 = Ahi + lds_off(arow, 4 + fq);
                const char* aq0 = Alo + lds_off(arow, s1);
                const char* aq1 = Alo + lds_off(arow, s2);
                bf16x8 bh0[JT], bh1[JT], ah0[IB], ah1[IB];
                u32x4 wl[JT], wh[JT], xl[IB], xh[IB];
#pragma unroll
                for (int j = 0; j < JT; ++j) bh0[j] = *reinterpret_cast<const bf16x8*>(bp0 + j * 2048);
#pragma unroll
                for (int i = 0; i < IB; ++i) ah0[i] = *reinterpret_cast<const bf16x8*>(ap0 + i * 2048);
#pragma unroll
                for (int j = 0; j < JT; ++j) bh1[j] = *reinterpret_cast<const bf16x8*>(bp1 + j * 2048);
#pragma unroll
                for (int i = 0; i < IB; ++i) ah1[i] = *reinterpret_cast<const bf16x8*>(ap1 + i * 2048);
#pragma unroll
                for (int j = 0; j < JT; ++j) {
                    wl[j] = *reinterpret_cast<const u32x4*>(bq0 + j * 2048);
                    wh[j] = *reinterpret_cast<const u32x4*>(bq1 + j * 2048);
                }
                // hi * hi, first 32-channel half
#pragma unroll
                for (int i = 0; i < IB; ++i) {
                    const bf16x8 a = ((amask[i] >> tap) & 1u) ? ah0[i] : zero;
#pragma unroll
                    for (int j = 0; j < JT; ++j) acc[i][j] = mma16<PREC_F16>(bh0[j], a, acc[i][j]);
                }
#pragma unroll
                for (int i = 0; i < IB; ++i) {
                    xl[i] = *reinterpret_cast<const u32x4*>(aq0 + i * 2048);
                    xh[i] = *reinterpret_cast<const u32x4*>(aq1 + i * 2048);
                }
                // hi * hi, second half
#pragma unroll
                for (int i = 0; i < IB; ++i) {
                    const bf16x8 a = ((amask[i] >> tap) & 1u) ? ah1[i] : zero;
#pragma unroll
                    for (int j = 0; j < JT; ++j) acc[i][j] = mma16<PREC_F16>(bh1[j], a, acc[i][j]);
                }
                // both cross terms of both halves: one K = 128 block-scaled e4m3 MFMA per tile.  A lane's eight operand registers are
                // 16 bytes of block fq >> 1 (channels 16 (fq & 1) ...) and 16 bytes of block 2 + (fq >> 1) (channels 32 + 16 (fq & 1) ...)
#pragma unroll
                for (int i = 0; i < IB; ++i) {
                    const bool ok = (amask[i] >> tap) & 1u;
                    const u32x4 lo = ok ? xl[i] : z4, hi = ok ? xh[i] : z4;
                    const i32x8 x8 = {(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
#pragma unroll
                    for (int j = 0; j < JT; ++j) {
                        const i32x8 w8 = {(int)wl[j][0], (int)wl[j][1], (int)wl[j][2], (int)wl[j][3], (int)wh[j][0], (int)wh[j][1], (int)wh[j][2], (int)wh[j][3]};
                        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w8, x8, acc[i][j], 0, 0, 0, wsc[j], 0, xsc);
                    }
                }
                // reads -> MFMA groups in source order: 24 reads | 4 selects + 16 MFMAs | 8 reads | 4 + 16 | 8 selects + 16
                __builtin_amdgcn_sched_group_barrier(0x100, 24, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 32, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            }
#pragma unroll
            for (int kk = 0; kk < ((DASV_CONV_DIAG == 2 || X8) ? 0 : 2); ++kk) {
                const int slot = kk * 4 + fq;
                bf16x8 bh[JT], bl[JT];
#pragma unroll
                for (int j = 0; j < JT; ++j) {
                    const int o = lds_off(wn * NPW + j * 16 + frow, slot);
                    bh[j] = *reinterpret_cast<const bf16x8*>(Bhi + o);
                    if (SPLIT) bl[j] = *reinterpret_cast<const bf16x8*>(Bhi + 16384 + o);
                }
#pragma unroll
                for (int i = 0; i < IB; ++i) {
                    const int o = lds_off(wm * (16 * IB) + i * 16 + frow + shift, slot);
                    const bool ok = (amask[i] >> tap) & 1u;
                    bf16x8 ah = *reinterpret_cast<const bf16x8*>(Ahi + o);
                    bf16x8 al;
                    if (SPLIT) al = *reinterpret_cast<const bf16x8*>(Alo + o);
                    const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
                    // (skipping these selects behind a wave-uniform "some lane is at a border" test was measured: 7 % slower —
                    // the scalar branches break up the MFMA schedule)
                    ah = ok ? ah : zero;
                    if (SPLIT) al = ok ? al : zero;
#pragma unroll
                    for (int j = 0; j < JT; ++j) {
                        // weights as the A operand, pixels as B: the 16x16 result has channels on its rows, so a
                        // lane ends up with 4 consecutive channels of one pixel (16-byte stores in the epilogue)
                        acc[i][j] = mma16<PREC>(bh[j], ah, acc[i][j]);
                        if (SPLIT) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al, acc[i][j], 0, 0, 0);
                        }
                    }
                }
                // scheduling hint for this 32-channel half in the 8-wave layouts: all sixteen fragment reads first, then the
                // selects and the MFMAs — 0-3 % on the double-buffered split layers (interleaved A/B, two runs); a {12 MFMAs, next
                // block's reads} pipeline was 2 % slower, in the 16-wave layout either hint costs 5-7 %, and in the single-MFMA
                // modes this one costs 5 %
                if constexpr (SPLIT && NWN == 2 && DASV_CONV_SCHED != 0) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 32, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 48, 0);
                }
            }
            if (!DBUF) __syncthreads();
        }
    }

    if (p.ksplit > 1) {   // partial sums (split8: already in output units); conv3x3_splitk_reduce adds them up and applies bias / ReLU / mask
        float* part = p.part + (size_t)ks * p.M * p.Cout;
#pragma unroll
        for (int j = 0; j < JT; ++j) {
            const int n = n0 + wn * NPW + j * 16 + (lane >> 4) * 4;
            f32x4 sc = {1.f, 1.f, 1.f, 1.f};
            if constexpr (X8) sc = *reinterpret_cast<const f32x4*>(winv + n) * sxinv;
            if constexpr (PREC == PREC_F16) sc = f32x4{sxinv, sxinv, sxinv, sxinv};
#pragma unroll
            for (int i = 0; i < IB; ++i) {
                const int m = m0 + wm * (16 * IB) + i * 16 + (lane & 15);
                if (m < p.M) *reinterpret_cast<f32x4*>(part + (size_t)m * p.Cout + n) = (X8 || PREC == PREC_F16) ? acc[i][j] * sc : acc[i][j];
            }
        }
        return;
    }
    // Coalescing epilogue (pipelined launches, plain output): a lane holds 4 channels of 16-pixel-strided pixels, so a direct store
    // instruction writes 16 separate 64-byte pieces; measured (stores removed in a timing build) that epilogue costs 6.5 % of
    // the seven layers and 11-13 % of the two-slab ones (3.5-5.4 us per tile).  Here the tile goes through LDS — [pixel][128
    // channels] rows of 528 bytes (the 16-byte pad makes the b128 writes of 8 pixels hit 32 distinct banks) — and is written out
    // as whole 512-byte pixel rows, two per wave-instruction.
    if constexpr (PIPELINED) {
        // (fp16 operands: no masked launch exists in the product — its gradients run on the bf16 / split kernels — so those kernels keep
        // the plain write-out only and hand a masked call to the direct epilogue below; same-process A/B of the two builds: the masked
        // write-out's code in the fp16 kernels cost the fp16 inference chain 1.4 %)
        if (PREC != PREC_F16 || (p.mask == nullptr && p.colsum == nullptr)) {
            constexpr int OP = 528;
            DASV_STAMP(60);
            __syncthreads();                     // every wave is done with the window and the tiles
            DASV_STAMP(61);
#pragma unroll
            for (int j = 0; j < JT; ++j) {
                const int nl = wn * NPW + j * 16 + (lane >> 4) * 4;
                f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                if (p.bias != nullptr) bv = *reinterpret_cast<const f32x4*>(p.bias + n0 + nl);
                f32x4 sc = {1.f, 1.f, 1.f, 1.f};
                if constexpr (X8) sc = *reinterpret_cast<const f32x4*>(winv + n0 + nl) * sxinv;
                constexpr bool SCALED = X8 || PREC == PREC_F16;     // (fp16 chain: sxinv undoes a half-precision input's scale)
                if constexpr (PREC == PREC_F16) sc = f32x4{sxinv, sxinv, sxinv, sxinv};
#pragma unroll
                for (int i = 0; i < IB; ++i) {
                    f32x4 v = (SCALED ? acc[i][j] * sc : acc[i][j]) + bv;
                    if (p.relu) v = relu4(v);
                    // (a masked output is noted where the mask is applied: the write-out pass below)
                    if (p.mask == nullptr && (GEOM == 1 ? gh0 + wm * IB + i < H : m0 + wm * (16 * IB) + i * 16 + (lane & 15) < mend)) note4(v);
                    char* row = smem + (wm * (16 * IB) + i * 16 + (lane & 15)) * OP;
                    if (p.pool) {
                        // fused max-pool: the image holds the fp32 values; the output form is applied to the pooled values below
                        *reinterpret_cast<f32x4*>(row + nl * 4) = v;
                    } else if (H16 && p.y_packed == 2) {
                        *reinterpret_cast<uint2*>(row + nl * 2) = pack16x4<PREC>(v, sy);
                    } else if (p.y_packed) {
                        // PACKED output (dasv_common.hpp): the row image is the pixel's two 256-byte slabs; the small stores that
                        // made a packed epilogue cost the layer 18 % when they went to memory are LDS stores here
                        uint32_t h01, h23, v8, r8;
                        cvt4_x8(v, sy, h01, h23, v8, r8);
                        char* slab = row + (nl >> 6) * 256;
                        const int cs = nl & 63;
                        *reinterpret_cast<uint2*>(slab + cs * 2) = make_uint2(h01, h23);
                        *reinterpret_cast<uint32_t*>(slab + 128 + cs) = v8;
                        *reinterpret_cast<uint32_t*>(slab + 192 + cs) = r8;
                    } else {
                        *reinterpret_cast<f32x4*>(row + nl * 4) = v;
                    }
                }
            }
            if (p.y_rec != nullptr) {            // workgroup maximum through one LDS word (joined by the barrier below)
                amax_u = wave_umax(amax_u);
                if (lane == 0 && amax_u > rec_seen) atomicMax(&s_amax, amax_u);
            }
            // (first-layer backward in the write-out pass, patch tiles: the 18 x 18 window of the input features behind the image —
            // the nine taps of every pixel then come out of LDS instead of nine cache reads in the exposed epilogue)
            float* melw = reinterpret_cast<float*>(smem + TBM * OP);
            if constexpr (X8 && GEOM == 1) {
                if (p.c1_part != nullptr && tid < PW * PW) {
                    const int r = tid / PW, c = tid - r * PW, hh = gh0 - 1 + r, ww = gw0 - 1 + c;
                    melw[tid] = ((unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W) ? p.c1_mel[((size_t)gb * H + hh) * W + ww] : 0.f;
                }
            }
            if constexpr (X8 && GEOM == 0) {      // raster tiles: the tile's pixels plus a row and a pixel on either side, contiguous
                if (p.c1_part != nullptr && tid < TBM + 2 * W + 2) {
                    const long long gi = (long long)m0 - W - 1 + tid;
                    melw[tid] = (gi >= 0 && gi < (long long)p.M) ? p.c1_mel[gi] : 0.f;
                }
            }
            __syncthreads();
            DASV_STAMP(62);
            if (p.y_rec != nullptr && tid == 0 && s_amax > rec_seen) atomicMax(p.y_rec, s_amax);
            if (p.pool) {
                // 2x2 stride-2 ceil-mode max-pool of the tile's row pairs out of the LDS image (F.max_pool2d(., 2, stride=2,
                // ceil_mode=True), CNNs.py:74-86): a thread takes four channels of one pooled pixel — 32 threads a pixel's 128
                // channels — compares the window's up to four pixels (a NaN takes over, as in torch; strict '>': values only, no
                // indices here) and stores the pooled pixel in the output form: fp32, PACKED (same conversion the stand-alone pool
                // applies, scaled by the exponent of THIS layer's output bound since the true maximum is not known before the last
                // tile has finished) or fp16.  The full-resolution map never reaches memory.
                const int Wo = (W + 1) >> 1, Ho = (H + 1) >> 1, U = p.pool_U;
                const int npool = GEOM == 1 ? 64 : (pu1 - pu0) * Wo;     // patch tiles: the 8 x 8 pooled patch
                const int col = tid & 31;
                auto mx4 = [](f32x4 a, f32x4 v) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) a[c] = (v[c] > a[c] || v[c] != v[c]) ? v[c] : a[c];
                    return a;
                };
                for (int q = tid >> 5; q < npool; q += NT / 32) {
                    int lp, rowp = W;             // first pixel of the window inside the tile's image; pixels per tile row
                    bool c1, r1;
                    long long op;
                    if constexpr (GEOM == 1) {
                        const int r = q >> 3, c = q & 7;                     // pooled pixel (r, c) of the patch
                        if (gh0 + 2 * r >= H) continue;                      // (a patch that hangs over the bottom of the image)
                        lp = 32 * r + 2 * c;
                        rowp = 16;
                        c1 = true;                                           // W % 16 == 0: the window's second column exists
                        r1 = gh0 + 2 * r + 1 < H;
                        op = ((long long)gb * Ho + (gh0 >> 1) + r) * Wo + (gw0 >> 1) + c;
                    } else {
                        const int j = q / Wo, c = q - j * Wo;
                        const int u = pu0 + j, b = u / U, k = u - b * U;
                        lp = (b * H + 2 * k) * W + 2 * c - m0;
                        c1 = 2 * c + 1 < W;
                        r1 = 2 * k + 1 < H;
                        op = ((long long)b * Ho + k) * Wo + c;
                    }
                    const char* src = smem + lp * OP + col * 16;
                    f32x4 v = *reinterpret_cast<const f32x4*>(src);
                    if (p.pool_idx != nullptr) {
                        // row-major scan, strict '>': the first maximum wins, as ATen's kernel does (and a NaN takes over) — the
                        // bytes dasv_maxpool2x2_ceil_idx_nhwc writes
                        int arg[4] = {0, 0, 0, 0};
                        auto mxa = [&](f32x4 t, int pos) {
#pragma unroll
                            for (int c = 0; c < 4; ++c)
                                if (t[c] > v[c] || t[c] != t[c]) { v[c] = t[c]; arg[c] = pos; }
                        };
                        if (c1) mxa(*reinterpret_cast<const f32x4*>(src + OP), 1);
                        if (r1) mxa(*reinterpret_cast<const f32x4*>(src + rowp * OP), 2);
                        if (c1 && r1) mxa(*reinterpret_cast<const f32x4*>(src + (rowp + 1) * OP), 3);
                        uint32_t code = 0u;
#pragma unroll
                        for (int c = 0; c < 4; ++c) code |= (uint32_t)(arg[c] | (v[c] > 0.f ? 4 : 0)) << (8 * c);
                        *reinterpret_cast<uint32_t*>(p.pool_idx + (size_t)op * p.Cout + n0 + col * 4) = code;
                    } else {
                        if (c1) v = mx4(v, *reinterpret_cast<const f32x4*>(src + OP));
                        if (r1) v = mx4(v, *reinterpret_cast<const f32x4*>(src + rowp * OP));
                        if (c1 && r1) v = mx4(v, *reinterpret_cast<const f32x4*>(src + (rowp + 1) * OP));
                    }
                    if (H16 && p.y_packed == 2) {
                        *reinterpret_cast<uint2*>(reinterpret_cast<char*>(p.y) + ((size_t)op * p.Cout + n0 + col * 4) * 2) = pack16x4<PREC>(v, sy);
                    } else if (p.y_packed) {
                        store_x8(p.y, op, p.Cout, n0 + col * 4, v, sy);
                    } else {
                        *reinterpret_cast<f32x4*>(p.y + (size_t)op * p.Cout + n0 + col * 4) = v;
                    }
                }
                DASV_STAMP(63);
                return;
            }
            if (H16 && p.y_packed == 2) {
                // half-precision rows: 256 bytes per pixel and 128-channel tile, 32 pixels per pass of the workgroup
                const int col16 = tid & 15;
#pragma unroll
                for (int k = 0; k < TBM / 32; ++k) {
                    const int pix = (tid >> 4) + 32 * k;
                    int m = m0 + pix;
                    if constexpr (GEOM == 1) m = gh0 + (pix >> 4) < H ? (gb * H + gh0 + (pix >> 4)) * W + gw0 + (pix & 15) : p.M;
                    if (m < p.M)
                        *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(p.y) + ((size_t)m * p.Cout + n0) * 2 + col16 * 16) =
                            *reinterpret_cast<const f32x4*>(smem + pix * OP + col16 * 16);
                }
                DASV_STAMP(63);
                return;
            }
            const int col = tid & 31;            // 16-byte chunk of a pixel's 512-byte row; 16 pixels per pass of the workgroup
            if (p.mask == nullptr && p.colsum == nullptr) {
#pragma unroll
                for (int k = 0; k < TBM / 16; ++k) {
                    const int pix = (tid >> 5) + 16 * k;
                    int m = m0 + pix;
                    if constexpr (GEOM == 1) m = gh0 + (pix >> 4) < H ? (gb * H + gh0 + (pix >> 4)) * W + gw0 + (pix & 15) : p.M;
                    if (m < p.M)
                        *reinterpret_cast<f32x4*>(p.y + (size_t)m * p.Cout + n0 + col * 4) = *reinterpret_cast<const f32x4*>(smem + pix * OP + col * 16);
                }
                DASV_STAMP(63);
                return;
            }
            // data-gradient launches (a ReLU's backward fused: outputs zeroed where mask <= 0; per-channel sums of the stored rows =
            // the bias gradient of the layer below): the mask is read, and the output written, in the same whole 512-byte pixel rows;
            // the mask rows of the first passes are requested before the image is read
            if constexpr (PREC != PREC_F16) {
            static_assert(NT == 512, "16 pixel rows per pass");
            f32x4 csum = {0.f, 0.f, 0.f, 0.f};
            const bool c1f = X8 && p.c1_part != nullptr;
            f32x4 wacc[9];                      // (c1f) the first layer's weight-gradient sums of this thread's four channels
#pragma unroll
            for (int t = 0; t < 9; ++t) wacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            constexpr int KB = 4;               // passes whose mask rows are in flight together
#pragma unroll
            for (int k0 = 0; k0 < TBM / 16; k0 += KB) {
                f32x4 mk[KB];
                int mm[KB];
#pragma unroll
                for (int k = 0; k < KB; ++k) {
                    const int pix = (tid >> 5) + 16 * (k0 + k);
                    int m = m0 + pix;
                    if constexpr (GEOM == 1) m = gh0 + (pix >> 4) < H ? (gb * H + gh0 + (pix >> 4)) * W + gw0 + (pix & 15) : p.M;
                    if (k0 + k >= TBM / 16) m = p.M;
                    mm[k] = m;
                    mk[k] = f32x4{1.f, 1.f, 1.f, 1.f};
                    if (p.mask != nullptr && m < p.M) {
                        if (p.mask_packed) {      // four halves of the pixel's fp16 plane (c = v * 2^E: the same sign)
                            const int n = n0 + col * 4;
                            const uint2 hb = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(p.mask) + (size_t)m * p.Cout * 4 +
                                                                             (n >> 6) * 256 + (n & 63) * 2);
                            const f16x2_t a = __builtin_bit_cast(f16x2_t, hb.x), b2 = __builtin_bit_cast(f16x2_t, hb.y);
                            mk[k] = f32x4{(float)a[0], (float)a[1], (float)b2[0], (float)b2[1]};
                        } else {
                            mk[k] = *reinterpret_cast<const f32x4*>(p.mask + (size_t)m * p.Cout + n0 + col * 4);
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < KB; ++k) {
                    if (mm[k] < p.M) {
                        const int pix = (tid >> 5) + 16 * (k0 + k);
                        f32x4 v = *reinterpret_cast<const f32x4*>(smem + pix * OP + col * 16);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = mk[k][r] > 0.f ? v[r] : 0.f;
                        if (p.mask != nullptr) note4(v);
                        if (p.y != nullptr) *reinterpret_cast<f32x4*>(p.y + (size_t)mm[k] * p.Cout + n0 + col * 4) = v;
                        csum += v;
                        if constexpr (X8) {
                            if (c1f) {
                                if constexpr (GEOM == 1) {
                                    const float* mw = melw + (pix >> 4) * PW + (pix & 15);
#pragma unroll
                                    for (int t = 0; t < 9; ++t) wacc[t] += v * mw[(t / 3) * PW + t % 3];
                                } else {
                                    const int w_ = mm[k] % W, rr = mm[k] / W;
                                    int h_ = rr % H, hlo = 0, hhi = H;
                                    if (p.row_span != nullptr) { h_ = rr; hlo = p.row_span[2 * rr]; hhi = p.row_span[2 * rr + 1]; }   // ragged: the utterance's rows
                                    const float* mp = melw + pix + W + 1;              // the pixel's place in the staged window
#pragma unroll
                                    for (int t = 0; t < 9; ++t) {
                                        const int hh = h_ + t / 3 - 1, ww = w_ + t % 3 - 1;
                                        const float in = (hh >= hlo && hh < hhi && (unsigned)ww < (unsigned)W) ? mp[(t / 3 - 1) * W + (t % 3 - 1)] : 0.f;
                                        wacc[t] += v * in;
                                    }
                                }
                            }
                        }
                    }
                }
            }
            if (p.mask != nullptr && p.y_rec != nullptr) range_note(p.y_rec, amax_u, rec_seen);
            if constexpr (X8) {
                if (c1f) {                       // fixed order: a thread's pixel rows (above), then the 16 pixel-row groups
                    float* red = reinterpret_cast<float*>(smem);       // [16][128 channels][10]
                    __syncthreads();             // every thread is done with the image
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float* dst = red + ((tid >> 5) * 128 + col * 4 + c) * 10;
#pragma unroll
                        for (int t = 0; t < 9; ++t) dst[t] = wacc[t][c];
                        dst[9] = csum[c];
                    }
                    __syncthreads();
                    for (int i = tid; i < 1280; i += NT) {
                        float t = 0.f;
#pragma unroll
                        for (int g = 0; g < 16; ++g) t += red[g * 1280 + i];
                        p.c1_part[((size_t)mt * p.Cout + n0) * 10 + i] = t;
                    }
                    DASV_STAMP(63);
                    return;
                }
            }
            if (p.colsum != nullptr) {           // fixed order: a thread's pixel rows (above), then the 16 pixel-row groups
                float* red = reinterpret_cast<float*>(smem);
                __syncthreads();                 // every thread is done with the image
                *reinterpret_cast<f32x4*>(red + (tid >> 5) * 128 + col * 4) = csum;
                __syncthreads();
                if (tid < 128) {
                    float t = 0.f;
#pragma unroll
                    for (int g = 0; g < 16; ++g) t += red[g * 128 + tid];
                    p.colsum[(size_t)mt * p.Cout + n0 + tid] = t;
                }
            }
            DASV_STAMP(63);
            return;
            }
        }
    }
    // epilogue: D[row = channel][col = pixel]: lane holds channels n .. n+3 of pixel m
    float* red = reinterpret_cast<float*>(smem);   // [4 wm][128 channels] partial column sums (colsum only)
    if (p.colsum != nullptr) __syncthreads();      // every wave is done with the LDS tiles
#pragma unroll
    for (int j = 0; j < JT; ++j) {
        const int n = n0 + wn * NPW + j * 16 + (lane >> 4) * 4;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias != nullptr) bv = *reinterpret_cast<const f32x4*>(p.bias + n);
        f32x4 sc = {1.f, 1.f, 1.f, 1.f};
        if constexpr (X8) sc = *reinterpret_cast<const f32x4*>(winv + n) * sxinv;
        if constexpr (PREC == PREC_F16) sc = f32x4{sxinv, sxinv, sxinv, sxinv};
        f32x4 csum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < IB; ++i) {
            const int m = m0 + wm * (16 * IB) + i * 16 + (lane & 15);
            if (m < p.M) {
                f32x4 v = ((X8 || PREC == PREC_F16) ? acc[i][j] * sc : acc[i][j]) + bv;
                if (p.relu) v = relu4(v);
                if (p.mask != nullptr) {
                    const f32x4 mk = *reinterpret_cast<const f32x4*>(p.mask + (size_t)m * p.Cout + n);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = mk[r] > 0.f ? v[r] : 0.f;
                }
                note4(v);
                if (H16 && p.y_packed == 2) {
                    *reinterpret_cast<uint2*>(reinterpret_cast<char*>(p.y) + ((size_t)m * p.Cout + n) * 2) = pack16x4<PREC>(v, sy);
                } else if (p.y_packed) store_x8(p.y, m, p.Cout, n, v, sy);
                else *reinterpret_cast<f32x4*>(p.y + (size_t)m * p.Cout + n) = v;
                csum += v;
            }
        }
        if (p.colsum != nullptr) {   // fixed-order reduction: 4 pixel blocks (above), 16 lanes (DPP), then the 4 pixel waves
#pragma unroll
            for (int r = 0; r < 4; ++r) csum[r] = group_sum<16>(csum[r]);
            if ((lane & 15) == 0) *reinterpret_cast<f32x4*>(red + wm * 128 + wn * NPW + j * 16 + (lane >> 4) * 4) = csum;
        }
    }
    if (p.colsum != nullptr) {
        __syncthreads();
        if (tid < 128) p.colsum[(size_t)mt * p.Cout + n0 + tid] = (red[tid] + red[128 + tid]) + (red[256 + tid] + red[384 + tid]);
    }
    if (p.y_rec != nullptr) range_note(p.y_rec, amax_u, rec_seen);
}

// (Round 2 kept an experimental four-wave form of the split8 kernel here — one wave of 128 x 64 outputs per SIMD, bit-identical,
// 7-14 % slower in place: DESIGN.md section 4.4d has the measurements.  Removed in round 3 with the rest of the debug surface; it is
// in the history at 469c692.)

// y = f(sum_ks part[ks] + bias): the second launch of a split-K convolution (fixed summation order)
__global__ void __launch_bounds__(256) conv3x3_splitk_reduce(const float* __restrict__ part, const float* __restrict__ bias,
                                                             const float* __restrict__ mask, float* __restrict__ y,
                                                             long long nvec, int Cout, int S, int relu, uint32_t* __restrict__ y_rec,
                                                             int mask_packed) {
    const long long plane = nvec * 4;
    uint32_t amax_u = 0u;
    __shared__ uint32_t s_amax;
    if (threadIdx.x == 0) s_amax = 0u;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
        f32x4 v = *reinterpret_cast<const f32x4*>(part + i * 4);
        for (int k = 1; k < S; ++k) v += *reinterpret_cast<const f32x4*>(part + k * plane + i * 4);
        if (bias != nullptr) v += *reinterpret_cast<const f32x4*>(bias + (int)((i * 4) % Cout));
        if (relu) v = f32x4{relu_nan(v[0]), relu_nan(v[1]), relu_nan(v[2]), relu_nan(v[3])};
        if (mask != nullptr) {
            f32x4 mk;
            if (mask_packed) {       // a packed activation tensor: the sign of the fp16 plane (four halves of the pixel's slab)
                const int n = (int)((i * 4) % Cout);
                const uint2 hb = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(mask) + (i * 4 - n) * 4 + (n >> 6) * 256 + (n & 63) * 2);
                const f16x2_t a = __builtin_bit_cast(f16x2_t, hb.x), b2 = __builtin_bit_cast(f16x2_t, hb.y);
                mk = f32x4{(float)a[0], (float)a[1], (float)b2[0], (float)b2[1]};
            } else {
                mk = *reinterpret_cast<const f32x4*>(mask + i * 4);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = mk[r] > 0.f ? v[r] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) amax_u = max(amax_u, __float_as_uint(v[r]) & 0x7fffffffu);
        *reinterpret_cast<f32x4*>(y + i * 4) = v;
    }
    if (y_rec != nullptr) range_note_block(y_rec, amax_u, &s_amax);
}

#ifdef DASV_DEBUG
static int g_patch_tiles = 1;            // developer build only (dasv_debug_conv_patch_tiles): A/B of the patch-tile geometry
#else
static constexpr int g_patch_tiles = 1;
#endif

// split-K factor of the window kernel: when the map has too few output tiles to fill the chip (small batches), the
// channel slabs of a tile are divided over several workgroups
static int conv3x3_ksplit(int B, int H, int W, int Cin, int Cout) {
    if (W > 80) return 1;
    const long long M = (long long)B * H * W;
    const long long tiles = ((M + WBM - 1) / WBM) * (Cout / CBN);
    const int nslab = Cin / CBK;
    if (tiles >= 192 || nslab < 2) return 1;
    long long S = (256 + tiles - 1) / tiles;
    if (S > nslab) S = nslab;
    return (int)S;
}

}  // namespace dasv

extern "C" {

size_t dasv_conv3x3_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % dasv::CBK != 0 || Cout % dasv::CBN != 0) return 0;
    const int S = dasv::conv3x3_ksplit(B, H, W, Cin, Cout);
    return S > 1 ? (size_t)S * B * H * W * Cout * sizeof(float) : 0;
}

size_t dasv_conv3x3_packed_weight_bytes(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0) return 0;
    return (size_t)Cout * 9 * Cin * sizeof(uint16_t);
}

int dasv_conv3x3_pack_weights(const float* w, void* w_hi, void* w_lo, int Cout, int Cin, dasv_stream_t stream) {
    if (w == nullptr || w_hi == nullptr || w_lo == nullptr) return DASV_E_NULL;
    if (Cout <= 0 || Cin <= 0) return DASV_E_SHAPE;
    const long long n = (long long)Cout * 9 * Cin;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(dasv::conv3x3_pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w,
                       static_cast<uint16_t*>(w_hi), static_cast<uint16_t*>(w_lo), Cout, Cin);
    return (int)hipGetLastError();
}

size_t dasv_conv3x3_packed_weight_bytes_x8(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0) return 0;
    // the e4m3 planes | [Cout][4] scale bytes | [Cout] floats 2^-Ew | { L1 bound, max |bias|, 0, 0 }
    return (size_t)Cout * 9 * Cin * sizeof(uint16_t) + (size_t)Cout * 8 + 16;
}

int dasv_conv3x3_pack_weights_x8(const float* w, const float* bias, void* w_f16, void* w_e4m3, int Cout, int Cin, dasv_stream_t stream) {
    if (w == nullptr || w_f16 == nullptr || w_e4m3 == nullptr) return DASV_E_NULL;
    if (Cout <= 0 || Cin <= 0 || Cin % 64 != 0) return DASV_E_SHAPE;
    hipError_t e = hipMemsetAsync(static_cast<char*>(w_e4m3) + (size_t)Cout * 9 * Cin * 2 + (size_t)Cout * 8, 0, 16, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    const size_t row = dasv::pack_row_lds(Cin);
    hipLaunchKernelGGL(dasv::conv3x3_pack_weights_x8_kernel, dim3((unsigned)Cout), dim3(256), row, (hipStream_t)stream, w, bias,
                       static_cast<uint16_t*>(w_f16), static_cast<uint8_t*>(w_e4m3), Cout, Cin, 0, row != 0 ? 1 : 0);
    return (int)hipGetLastError();
}

int dasv_conv3x3_pack_weights_x8_dgrad(const float* w, void* w_f16, void* w_e4m3, int Cout, int Cin, dasv_stream_t stream) {
    return dasv_conv3x3_pack_weights_x8_dgrad_ws(w, nullptr, w_f16, w_e4m3, Cout, Cin, stream);
}

int dasv_conv3x3_pack_weights_x8_dgrad_ws(const float* w, float* scratch, void* w_f16, void* w_e4m3, int Cout, int Cin, dasv_stream_t stream) {
    if (w == nullptr || w_f16 == nullptr || w_e4m3 == nullptr) return DASV_E_NULL;
    if (Cout <= 0 || Cin <= 0 || Cout % 64 != 0) return DASV_E_SHAPE;
    const size_t row = dasv::pack_row_lds(Cout);       // (the planes' input channels are this layer's output channels)
    if (scratch != nullptr) {
        // large layers: the transposed, flipped weight goes through `scratch` ([Cin][Cout][3][3] floats) with contiguous reads and writes,
        // then the row-wise pack reads it like a forward weight (mode 2: no output bounds) — the index arithmetic of the direct form reads
        // 36-byte pieces a whole row apart, which loses above ~256 x 256 channels
        hipLaunchKernelGGL(dasv::conv3x3_transpose_flip_kernel, dim3((unsigned)((Cout + 31) / 32), (unsigned)((Cin + 31) / 32)), dim3(256), 0,
                           (hipStream_t)stream, w, scratch, Cout, Cin);
        hipLaunchKernelGGL(dasv::conv3x3_pack_weights_x8_kernel, dim3((unsigned)Cin), dim3(256), row, (hipStream_t)stream, scratch, nullptr,
                           static_cast<uint16_t*>(w_f16), static_cast<uint8_t*>(w_e4m3), Cin, Cout, 2, row != 0 ? 1 : 0);
        return (int)hipGetLastError();
    }
    // planes of the convolution with Cin outputs and Cout inputs
    hipLaunchKernelGGL(dasv::conv3x3_pack_weights_x8_kernel, dim3((unsigned)Cin), dim3(256), row, (hipStream_t)stream, w, nullptr,
                       static_cast<uint16_t*>(w_f16), static_cast<uint8_t*>(w_e4m3), Cin, Cout, 1, row != 0 ? 1 : 0);
    return (int)hipGetLastError();
}

size_t dasv_conv3x3_packed_weight_bytes_f16(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0) return 0;
    return (size_t)Cout * 9 * Cin * sizeof(uint16_t) + 16;     // the fp16 plane | { L1 bound, max |bias|, 0, 0 }
}

int dasv_conv3x3_pack_weights_f16(const float* w, const float* bias, void* w_f16, int Cout, int Cin, dasv_stream_t stream) {
    if (w == nullptr || w_f16 == nullptr) return DASV_E_NULL;
    if (Cout <= 0 || Cin <= 0) return DASV_E_SHAPE;
    const long long n = (long long)Cout * 9 * Cin;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(dasv::conv3x3_pack_weights_f16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w,
                       static_cast<uint16_t*>(w_f16), Cout, Cin);
    hipError_t e = hipMemsetAsync(static_cast<char*>(w_f16) + (size_t)n * 2, 0, 16, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(dasv::conv3x3_weight_bounds_kernel, dim3((unsigned)Cout), dim3(256), 0, (hipStream_t)stream, w, bias,
                       reinterpret_cast<uint32_t*>(static_cast<char*>(w_f16) + (size_t)n * 2), Cout, Cin);
    return (int)hipGetLastError();
}


static int conv3x3_launch(int mode, const float* x, const float* mel, const float* w1, const float* b1, const void* w_hi,
                          const void* w_lo, const float* bias, const float* mask, float* colsum, float* y, void* workspace,
                          size_t workspace_bytes, int B, int H, int W, int Cin, int Cout, int relu, dasv_stream_t stream,
                          const int* row_span = nullptr, int x_packed = 0, int y_packed = 0, const uint32_t* x_rec = nullptr,
                          uint32_t* y_rec = nullptr, int pool = 0, int pool_tile = 0, uint8_t* pool_idx = nullptr, int mask_packed = 0,
                          const float* c1_mel = nullptr, float* c1_part = nullptr) {
    if ((x == nullptr && mel == nullptr) || w_hi == nullptr || (y == nullptr && c1_part == nullptr) || ((mode == 0 || mode == 3) && w_lo == nullptr)) return DASV_E_NULL;
    if (c1_part != nullptr && (c1_mel == nullptr || mode != 3 || mask == nullptr || colsum != nullptr || workspace != nullptr || W > 80 || pool))
        return DASV_E_SHAPE;
    if (mel != nullptr && (w1 == nullptr || W > 80)) return mel != nullptr && w1 == nullptr ? DASV_E_NULL : DASV_E_SHAPE;
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return DASV_E_SHAPE;
    if (Cin % dasv::CBK != 0 || Cout % dasv::CBN != 0) return DASV_E_SHAPE;
    if ((long long)B * H * W > 0x7fffffffLL / 2) return DASV_E_SHAPE;
    if (mode < 0 || mode > 3) return DASV_E_DTYPE;
    if (mode == 3 && (W > 80 || mel != nullptr)) return DASV_E_SHAPE;   // fp16 + e4m3 mode: window kernel, activations from memory
    dasv::ConvParams p;
    p.x = x; p.whi = static_cast<const uint16_t*>(w_hi); p.wlo = static_cast<const uint16_t*>(w_lo); p.bias = bias; p.mask = mask; p.colsum = colsum; p.y = y;
    if (colsum != nullptr && W > 80) return DASV_E_SHAPE;   // only the window kernel carries the column sums
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.M = B * H * W; p.relu = relu;
    p.mel = mel; p.w1 = w1; p.b1 = b1;
    p.part = nullptr; p.ksplit = 1;
    p.x_packed = x_packed; p.y_packed = y_packed;
    p.x_rec = x_rec; p.y_rec = y_rec;
    p.mask_packed = mask_packed;
    p.c1_mel = c1_mel; p.c1_part = c1_part;
    if (y_packed && (mask != nullptr || colsum != nullptr)) return DASV_E_SHAPE;
    if ((x_packed || y_packed) && mel != nullptr) return DASV_E_SHAPE;
    if ((x_packed == 1 || y_packed == 1) && mode != 3) return DASV_E_SHAPE;        // packed split8 planes
    if ((x_packed == 2 || y_packed == 2) && ((mode != 2 && mode != 1) || W > 80)) return DASV_E_SHAPE;   // 16-bit tensors: the fp16 / bf16 window kernels
    if (mode == 3 && (x_rec == nullptr || (y_packed && y_rec == nullptr))) return DASV_E_NULL;   // split8 scales its operands by the records
    if (mode == 2 && (x_packed == 2 || y_packed == 2) && (x_rec == nullptr || (y_packed == 2 && y_rec == nullptr))) return DASV_E_NULL;
    p.row_span = row_span;
    p.pool = 0; p.pool_units = 0; p.pool_U = 0; p.pool_idx = pool_idx;
    if (pool_idx != nullptr && !pool) return DASV_E_SHAPE;
    if (row_span != nullptr && (B != 1 || W > 80 || mel != nullptr)) return DASV_E_SHAPE;   // ragged: one tall image, window kernel
    // fused max-pool: dense batches, plain epilogue (no mask / column sums / split-K), relu'd outputs
    if (pool && (row_span != nullptr || mask != nullptr || colsum != nullptr || mel != nullptr || W > 80 || !relu)) return DASV_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    p.ntiles = Cout / dasv::CBN;
    if (W <= 80) {
        // split-K needs the caller's workspace (and has no column-sum / fused-first-layer form): without it, one pass
        // (a packed output is written by the convolution's own epilogue: no split-K form)
        const int S = (colsum == nullptr && mel == nullptr && workspace != nullptr && !y_packed && !pool) ? dasv::conv3x3_ksplit(B, H, W, Cin, Cout) : 1;
        const bool two = mode == 0 || mode == 3;
        const size_t lds256 = (size_t)(dasv::WBM + 2 * W + 2) * 128 * (two ? 2 : 1) + 32768;
        const size_t lds192 = (size_t)(192 + 2 * W + 2) * 128 * 2 + 32768;
        // 192-pixel tiles where the 256-pixel window leaves no room for the second tile buffer (80-column maps) but the 192-pixel
        // one does: the launch then runs the pipelined double-buffered loops.  Not for split-K launches.
        // (dasv_conv3x3_colsum_rows repeats this choice of tiles: keep the two in step)
        const bool t192 = two && lds256 + 32768 + 128 > 160 * 1024 && lds192 + 32768 + 128 <= 160 * 1024 && S == 1 && mel == nullptr;
        const int tbm = t192 ? 192 : dasv::WBM;
        p.mtiles = (p.M + tbm - 1) / tbm;
        const size_t lds1 = t192 ? lds192 : lds256;
        // both tile buffers (stride 32 KB also for the one-plane modes) whenever they fit: the launch then runs the pipelined loops
        // (round 1 measured the double buffer alone as a loss for bf16 operands — with the un-pipelined loop)
        const bool dbuf = (two || mel == nullptr) && lds1 + 32768 + 128 <= 160 * 1024;
        // (a packed mask is read by the pipelined launches' write-out pass or by the split-K reduction)
        if (mask_packed && S == 1 && !dbuf) return DASV_E_SHAPE;
        if (c1_part != nullptr && !dbuf) return DASV_E_SHAPE;
        // patch tiles (GEOM = 1): maps whose width is a multiple of 16 and whose raster window would be larger than the 18 x 18
        // patch window (80 / 64 / 48 columns; the 80-column first block in this model) — pipelined split8 launches (forward and
        // masked data gradient) without ragged borders or split-K
        // (measured, tools/_diag/pool_fuse_ab.py, profiles/r04_pool_fusion_ab.txt: split8 on [64,300,80] -1.7 % of the front-end's time
        // for the plain layer and -5.6 % with the fused pool; with fp16 operands, whose raster form already has 256-pixel tiles there,
        // the patch form is 4.5 % SLOWER — more address arithmetic per staged chunk — so it is a split8 geometry)
        const bool patch = mode == 3 && W % 16 == 0 && W >= 48 && mel == nullptr && row_span == nullptr && S == 1 && pool_tile == 0 &&
                           dasv::g_patch_tiles;
        if (patch) {
            p.mtiles = B * ((H + 15) / 16) * (W / 16);
            p.tiles_per_xcd = (p.mtiles * p.ntiles + 7) / 8;
            if (pool) { p.pool = 1; p.pool_units = 0; p.pool_U = (H + 1) / 2; }
            size_t ldsp = (size_t)324 * 128 * (two ? 2 : 1) + 65536 + 128;
            if (ldsp < (size_t)256 * 528) ldsp = (size_t)256 * 528;
            const dim3 gridp((unsigned)(p.tiles_per_xcd * 8));
#define DASV_LAUNCH_PATCH(SP)                                                                                       \
            do {                                                                                                    \
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dasv::conv3x3_window_kernel<SP, true, false, 2, 4, 4, 1>), \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp);          \
                if (e != hipSuccess) return (int)e;                                                                 \
                hipLaunchKernelGGL((dasv::conv3x3_window_kernel<SP, true, false, 2, 4, 4, 1>), gridp, dim3(512), ldsp, s, p); \
            } while (0)
            DASV_LAUNCH_PATCH(3);
#undef DASV_LAUNCH_PATCH
            return (int)hipGetLastError();
        }
        bool t160 = false;
        if (pool) {
            // tiles of whole row pairs (the pooled tile is formed in the coalescing epilogue's LDS image: pipelined launches only).
            // Tile shape: 256 (192) pixels of which `big` hold whole row pairs, or 160 pixels (2 x 4 waves of 80 x 32 outputs),
            // which hold the row pairs of 80-, 40-, 20- and 10-column maps exactly.  Measured (tools/_diag/pool_fuse_ab.py,
            // profiles/r04_pool_fusion_ab.txt): per pixel the 160-pixel layout is ~17 % slower than the 64 x 64 wave tiles (more
            // fragment reads per MFMA, the weight tile staged per 160 pixels), so it only pays where the big tile would waste more
            // than a fifth of its rows (fp16 operands on 80 columns: 160 of 256).  pool_tile forces one (0: automatic).
            const int big = (dbuf && 2 * W <= tbm) ? tbm / (2 * W) * 2 * W : 0;            // pixels used of the big tile
            const int small = (mode == 2 || mode == 3) && 2 * W <= 160 ? 160 / (2 * W) * 2 * W : 0;
            if (pool_tile == 160) t160 = small > 0;
            else if (pool_tile == 0) t160 = small > 0 && (big == 0 || big * 5 < tbm * 4);
            if (pool_tile != 0 && pool_tile != 160 && pool_tile != tbm) return DASV_E_SHAPE;
            if (!t160 && big == 0) return DASV_E_SHAPE;
            if (pool_tile == 160 && !t160) return DASV_E_SHAPE;
            p.pool = 1;
            p.pool_units = (t160 ? 160 : tbm) / (2 * W);
            p.pool_U = (H + 1) / 2;
            p.mtiles = (int)(((long long)B * p.pool_U + p.pool_units - 1) / p.pool_units);
        }
        p.tiles_per_xcd = (p.mtiles * p.ntiles + 7) / 8;
        size_t lds = lds1 + (dbuf ? 32768 + 128 : 0);   // + the zero row of the pipelined loops
        if (dbuf && lds < (size_t)tbm * 528) lds = (size_t)tbm * 528;   // the coalescing epilogue's [pixel][528 B] image (narrow maps)
        if (c1_part != nullptr && lds < (size_t)tbm * 528 + (size_t)(tbm + 2 * W + 2) * 4) lds = (size_t)tbm * 528 + (size_t)(tbm + 2 * W + 2) * 4;
        if (t160) {
            lds = (size_t)(160 + 2 * W + 2) * 128 * (two ? 2 : 1) + 65536 + 128;
            if (lds < (size_t)160 * 528) lds = (size_t)160 * 528;
        }
        if (S > 1) {
            if (workspace_bytes < (size_t)S * p.M * Cout * sizeof(float)) return DASV_E_WORKSPACE;
            p.part = static_cast<float*>(workspace);
            p.ksplit = S;
        }
        const dim3 grid((unsigned)(p.tiles_per_xcd * 8 * p.ksplit)), block(512);
#define DASV_LAUNCH_WINDOW(SP, DB, CC, NW, ...)                                                                      \
        do {                                                                                                         \
            constexpr int nwn_ = NW;                                                                                 \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dasv::conv3x3_window_kernel<SP, DB, CC, nwn_, ##__VA_ARGS__>), \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                \
            if (e != hipSuccess) return (int)e;                                                                      \
            hipLaunchKernelGGL((dasv::conv3x3_window_kernel<SP, DB, CC, nwn_, ##__VA_ARGS__>), grid, dim3(t160 ? 512 : 256 * nwn_), lds, s, p); \
        } while (0)
        // wave layout: 8 waves (64 x 64 outputs each) everywhere except the single-buffered split launches (80-column maps:
        // two barriers per tap), where 16 waves of 64 x 32 hide each other's waits: conv12 1.32 -> 1.20 ms (interleaved A/B)
        if (t160 && mode == 3) DASV_LAUNCH_WINDOW(3, true, false, 4, 5, 2);
        else if (t160) DASV_LAUNCH_WINDOW(2, true, false, 4, 5, 2);
        else if (p.mel != nullptr) {
            if (mode == 0 && dbuf) DASV_LAUNCH_WINDOW(0, true, true, 2);
            else if (mode == 0) DASV_LAUNCH_WINDOW(0, false, true, 2);
            else if (mode == 1) DASV_LAUNCH_WINDOW(1, false, true, 2);
            else DASV_LAUNCH_WINDOW(2, false, true, 2);
        } else if (t192 && mode == 3) DASV_LAUNCH_WINDOW(3, true, false, 2, 3);
        else if (t192) DASV_LAUNCH_WINDOW(0, true, false, 2, 3);
        else if (mode == 3 && dbuf) DASV_LAUNCH_WINDOW(3, true, false, 2);
        else if (mode == 3) DASV_LAUNCH_WINDOW(3, false, false, 2);
        else if (mode == 0 && dbuf) DASV_LAUNCH_WINDOW(0, true, false, 2);
        else if (mode == 0) DASV_LAUNCH_WINDOW(0, false, false, 4);
        else if (mode == 1 && dbuf) DASV_LAUNCH_WINDOW(1, true, false, 2);
        else if (mode == 1) DASV_LAUNCH_WINDOW(1, false, false, 2);
        else if (dbuf) DASV_LAUNCH_WINDOW(2, true, false, 2);
        else DASV_LAUNCH_WINDOW(2, false, false, 2);
#undef DASV_LAUNCH_WINDOW
        if (p.ksplit > 1) {
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return (int)e;
            const long long nvec = (long long)p.M * Cout / 4;
            const int blocks = (int)((nvec + 255) / 256 < 4096 ? (nvec + 255) / 256 : 4096);
            hipLaunchKernelGGL(dasv::conv3x3_splitk_reduce, dim3(blocks), dim3(256), 0, s, p.part, bias, mask, y, nvec, Cout,
                               p.ksplit, relu, y_rec, mask_packed);
        }
        return (int)hipGetLastError();
    }
    p.mtiles = (p.M + dasv::CBM - 1) / dasv::CBM;
    const int tiles = p.mtiles * p.ntiles;
    p.tiles_per_xcd = (tiles + 7) / 8;
    const dim3 grid((unsigned)(p.tiles_per_xcd * 8)), block(256);
    const size_t lds = 65536;
#define DASV_LAUNCH_TILE(PR)                                                                                        \
    do {                                                                                                            \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dasv::conv3x3_igemm_kernel<PR>),           \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
        if (e != hipSuccess) return (int)e;                                                                         \
        hipLaunchKernelGGL(dasv::conv3x3_igemm_kernel<PR>, grid, block, lds, s, p);                                  \
    } while (0)
    if (mode == 0) DASV_LAUNCH_TILE(0);
    else if (mode == 1) DASV_LAUNCH_TILE(1);
    else DASV_LAUNCH_TILE(2);
#undef DASV_LAUNCH_TILE
    return (int)hipGetLastError();
}

int dasv_conv3x3_bias_relu_nhwc(int mode, const float* x, const void* w_hi, const void* w_lo, const float* bias, float* y,
                                int B, int H, int W, int Cin, int Cout, int relu, dasv_stream_t stream) {
    return conv3x3_launch(mode, x, nullptr, nullptr, nullptr, w_hi, w_lo, bias, nullptr, nullptr, y, nullptr, 0, B, H, W, Cin, Cout,
                          relu, stream);
}

int dasv_conv3x3_bias_relu_nhwc_ws(int mode, const float* x, const void* w_hi, const void* w_lo, const float* bias, float* y,
                                   void* workspace, size_t workspace_bytes, int B, int H, int W, int Cin, int Cout, int relu,
                                   dasv_stream_t stream) {
    return conv3x3_launch(mode, x, nullptr, nullptr, nullptr, w_hi, w_lo, bias, nullptr, nullptr, y, workspace, workspace_bytes, B,
                          H, W, Cin, Cout, relu, stream);
}

int dasv_conv3x3_bias_relu_x8_nhwc(const void* x, int x_packed, const uint32_t* x_range, const void* w_f16, const void* w_e4m3,
                                   const float* bias, void* y, int y_packed, uint32_t* y_range, void* workspace, size_t workspace_bytes,
                                   int B, int H, int W, int Cin, int Cout, int relu, dasv_stream_t stream) {
    if (W > 80) return DASV_E_SHAPE;
    return conv3x3_launch(3, static_cast<const float*>(x), nullptr, nullptr, nullptr, w_f16, w_e4m3, bias, nullptr, nullptr,
                          static_cast<float*>(y), workspace, workspace_bytes, B, H, W, Cin, Cout, relu, stream, nullptr,
                          x_packed ? 1 : 0, y_packed ? 1 : 0, x_range, y_range);
}

int dasv_conv3x3_bias_relu_bf16_nhwc(const void* x, int x_bf16, const void* w_bf16, const float* bias, void* y, int y_bf16, void* workspace,
                                     size_t workspace_bytes, int B, int H, int W, int Cin, int Cout, int relu, dasv_stream_t stream) {
    if (W > 80) return DASV_E_SHAPE;
    return conv3x3_launch(1, static_cast<const float*>(x), nullptr, nullptr, nullptr, w_bf16, nullptr, bias, nullptr, nullptr,
                          static_cast<float*>(y), y_bf16 ? nullptr : workspace, y_bf16 ? 0 : workspace_bytes, B, H, W, Cin, Cout, relu, stream,
                          nullptr, x_bf16 ? 2 : 0, y_bf16 ? 2 : 0, nullptr, nullptr);
}

int dasv_conv3x3_pool_supported(int mode, int B, int H, int W, int Cin, int Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || W > 80 || Cin <= 0 || Cout <= 0 || Cin % dasv::CBK != 0 || Cout % dasv::CBN != 0) return 0;
    if (mode != 2 && mode != 3) return 0;
    // few tiles (single utterances): those launches divide their channel slabs over workgroups instead (split-K, no fused pool)
    if (dasv::conv3x3_ksplit(B, H, W, Cin, Cout) > 1) return 0;
    const bool two = mode == 3;
    const size_t lds256 = (size_t)(dasv::WBM + 2 * W + 2) * 128 * (two ? 2 : 1) + 32768;
    const size_t lds192 = (size_t)(192 + 2 * W + 2) * 128 * 2 + 32768;
    const bool t192 = two && lds256 + 32768 + 128 > 160 * 1024 && lds192 + 32768 + 128 <= 160 * 1024;
    const size_t lds1 = t192 ? lds192 : lds256;
    return 2 * W <= 160 || (lds1 + 32768 + 128 <= 160 * 1024 && 2 * W <= (t192 ? 192 : dasv::WBM));
}

int dasv_conv3x3_bias_relu_pool_x8_nhwc(const void* x, int x_packed, const uint32_t* x_range, const void* w_f16, const void* w_e4m3,
                                        const float* bias, void* y, int y_packed, uint32_t* y_range, int B, int H, int W, int Cin,
                                        int Cout, int tile_pixels, dasv_stream_t stream) {
    if (!dasv_conv3x3_pool_supported(3, B, H, W, Cin, Cout)) return DASV_E_SHAPE;
    return conv3x3_launch(3, static_cast<const float*>(x), nullptr, nullptr, nullptr, w_f16, w_e4m3, bias, nullptr, nullptr,
                          static_cast<float*>(y), nullptr, 0, B, H, W, Cin, Cout, 1, stream, nullptr, x_packed ? 1 : 0,
                          y_packed ? 1 : 0, x_range, y_range, 1, tile_pixels);
}

int dasv_conv3x3_bias_relu_pool_idx_x8_nhwc(const void* x, int x_packed, const uint32_t* x_range, const void* w_f16, const void* w_e4m3,
                                            const float* bias, void* y, int y_packed, uint32_t* y_range, uint8_t* idx, int B, int H, int W,
                                            int Cin, int Cout, dasv_stream_t stream) {
    if (idx == nullptr) return DASV_E_NULL;
    if (!dasv_conv3x3_pool_supported(3, B, H, W, Cin, Cout)) return DASV_E_SHAPE;
    return conv3x3_launch(3, static_cast<const float*>(x), nullptr, nullptr, nullptr, w_f16, w_e4m3, bias, nullptr, nullptr,
                          static_cast<float*>(y), nullptr, 0, B, H, W, Cin, Cout, 1, stream, nullptr, x_packed ? 1 : 0, y_packed ? 1 : 0,
                          x_range, y_range, 1, 0, idx);
}

int dasv_conv3x3_masked_x8p_nhwc(const void* x, int x_packed, const uint32_t* x_range, const void* w_f16, const void* w_e4m3,
                                 const void* mask_packed, float* y, uint32_t* y_range, float* colsum_partial, void* workspace,
                                 size_t workspace_bytes, int B, int H, int W, int Cin, int Cout, dasv_stream_t stream) {
    if (mask_packed == nullptr) return DASV_E_NULL;
    if (W > 80) return DASV_E_SHAPE;
    return conv3x3_launch(3, static_cast<const float*>(x), nullptr, nullptr, nullptr, w_f16, w_e4m3, nullptr, static_cast<const float*>(mask_packed),
                          colsum_partial, y, workspace, workspace_bytes, B, H, W, Cin, Cout, 0, stream, nullptr, x_packed ? 1 : 0, 0, x_range,
                          y_range, 0, 0, nullptr, 1);
}

int dasv_conv3x3_bias_relu_pool_f16_nhwc(const void* x, int x_f16, const uint32_t* x_range, const void* w_f16, const float* bias,
                                         void* y, int y_f16, uint32_t* y_range, int B, int H, int W, int Cin, int Cout,
                                         int tile_pixels, dasv_stream_t stream) {
    if (!dasv_conv3x3_pool_supported(2, B, H, W, Cin, Cout)) return DASV_E_SHAPE;
    return conv3x3_launch(2, static_cast<const float*>(x), nullptr, nullptr, nullptr, w_f16, nullptr, bias, nullptr, nullptr,
                          static_cast<float*>(y), nullptr, 0, B, H, W, Cin, Cout, 1, stream, nullptr, x_f16 ? 2 : 0,
                          y_f16 ? 2 : 0, x_range, y_range, 1, tile_pixels);
}

#ifdef DASV_DEBUG
void dasv_debug_conv_patch_tiles(int on) { dasv::g_patch_tiles = on; }
#endif

#if DASV_CONV_STAMPS
int dasv_debug_conv_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dasv::dasv_conv_stamps), sizeof(unsigned long long) * 128);
}
#endif

int dasv_conv3x3_bias_relu_f16_nhwc(const void* x, int x_f16, const uint32_t* x_range, const void* w_f16, const float* bias, void* y,
                                    int y_f16, uint32_t* y_range, void* workspace, size_t workspace_bytes, int B, int H, int W,
                                    int Cin, int Cout, int relu, dasv_stream_t stream) {
    return conv3x3_launch(2, static_cast<const float*>(x), nullptr, nullptr, nullptr, w_f16, nullptr, bias, nullptr, nullptr,
                          static_cast<float*>(y), workspace, workspace_bytes, B, H, W, Cin, Cout, relu, stream, nullptr,
                          x_f16 ? 2 : 0, y_f16 ? 2 : 0, x_range, y_range);
}

int dasv_conv3x3_bias_relu_ragged_f16_nhwc(const void* x, int x_f16, const uint32_t* x_range, const void* w_f16, const float* bias,
                                           void* y, int y_f16, uint32_t* y_range, const int* row_span, void* workspace,
                                           size_t workspace_bytes, int rows, int W, int Cin, int Cout, int relu, dasv_stream_t stream) {
    if (row_span == nullptr) return DASV_E_NULL;
    return conv3x3_launch(2, static_cast<const float*>(x), nullptr, nullptr, nullptr, w_f16, nullptr, bias, nullptr, nullptr,
                          static_cast<float*>(y), workspace, workspace_bytes, 1, rows, W, Cin, Cout, relu, stream, row_span,
                          x_f16 ? 2 : 0, y_f16 ? 2 : 0, x_range, y_range);
}

int dasv_conv3x3_bias_relu_ragged_nhwc(int mode, const float* x, const void* w_hi, const void* w_lo, const float* bias, float* y,
                                       const int* row_span, void* workspace, size_t workspace_bytes, int rows, int W, int Cin,
                                       int Cout, int relu, dasv_stream_t stream) {
    if (row_span == nullptr) return DASV_E_NULL;
    return conv3x3_launch(mode, x, nullptr, nullptr, nullptr, w_hi, w_lo, bias, nullptr, nullptr, y, workspace, workspace_bytes, 1,
                          rows, W, Cin, Cout, relu, stream, row_span);
}

int dasv_conv3x3_bias_relu_ragged_x8_nhwc(const void* x, int x_packed, const uint32_t* x_range, const void* w_f16, const void* w_e4m3,
                                          const float* bias, void* y, int y_packed, uint32_t* y_range, const int* row_span, void* workspace,
                                          size_t workspace_bytes, int rows, int W, int Cin, int Cout, int relu, dasv_stream_t stream) {
    if (row_span == nullptr) return DASV_E_NULL;
    if (W > 80) return DASV_E_SHAPE;
    return conv3x3_launch(3, static_cast<const float*>(x), nullptr, nullptr, nullptr, w_f16, w_e4m3, bias, nullptr, nullptr,
                          static_cast<float*>(y), workspace, workspace_bytes, 1, rows, W, Cin, Cout, relu, stream, row_span,
                          x_packed ? 1 : 0, y_packed ? 1 : 0, x_range, y_range);
}

int dasv_conv3x3_masked_nhwc(int mode, const float* x, const void* w_hi, const void* w_lo, const float* mask, float* y,
                             float* colsum_partial, int B, int H, int W, int Cin, int Cout, dasv_stream_t stream) {
    if (mask == nullptr) return DASV_E_NULL;
    return conv3x3_launch(mode, x, nullptr, nullptr, nullptr, w_hi, w_lo, nullptr, mask, colsum_partial, y, nullptr, 0, B, H, W,
                          Cin, Cout, 0, stream);
}

int dasv_conv3x3_masked_x8_nhwc(const float* x, const uint32_t* x_range, const void* w_f16, const void* w_e4m3, const float* mask,
                                float* y, uint32_t* y_range, float* colsum_partial, int B, int H, int W, int Cin, int Cout,
                                dasv_stream_t stream) {
    if (mask == nullptr) return DASV_E_NULL;
    if (W > 80) return DASV_E_SHAPE;
    return conv3x3_launch(3, x, nullptr, nullptr, nullptr, w_f16, w_e4m3, nullptr, mask, colsum_partial, y, nullptr, 0, B, H, W,
                          Cin, Cout, 0, stream, nullptr, 0, 0, x_range, y_range);
}

int dasv_conv3x3_masked_ragged_x8p_nhwc(const void* x, int x_packed, const uint32_t* x_range, const void* w_f16, const void* w_e4m3,
                                        const void* mask_packed, float* y, uint32_t* y_range, float* colsum_partial, const int* row_span,
                                        int rows, int W, int Cin, int Cout, dasv_stream_t stream) {
    if (mask_packed == nullptr || row_span == nullptr) return DASV_E_NULL;
    if (W > 80) return DASV_E_SHAPE;
    return conv3x3_launch(3, static_cast<const float*>(x), nullptr, nullptr, nullptr, w_f16, w_e4m3, nullptr, static_cast<const float*>(mask_packed),
                          colsum_partial, y, nullptr, 0, 1, rows, W, Cin, Cout, 0, stream, row_span, x_packed ? 1 : 0, 0, x_range, y_range, 0, 0,
                          nullptr, 1);
}

int dasv_conv3x3_masked_c1_x8p_nhwc(const void* x, int x_packed, const uint32_t* x_range, const void* w_f16, const void* w_e4m3, const void* mask,
                                    int mask_packed, const float* mel, float* c1_partial, int B, int H, int W, int Cin, int Cout,
                                    dasv_stream_t stream) {
    if (mask == nullptr || mel == nullptr || c1_partial == nullptr) return DASV_E_NULL;
    if (W > 80) return DASV_E_SHAPE;
    return conv3x3_launch(3, static_cast<const float*>(x), nullptr, nullptr, nullptr, w_f16, w_e4m3, nullptr, static_cast<const float*>(mask),
                          nullptr, nullptr, nullptr, 0, B, H, W, Cin, Cout, 0, stream, nullptr, x_packed ? 1 : 0, 0, x_range, nullptr, 0, 0,
                          nullptr, mask_packed ? 1 : 0, mel, c1_partial);
}

int dasv_conv3x3_masked_c1_ragged_x8p_nhwc(const void* x, int x_packed, const uint32_t* x_range, const void* w_f16, const void* w_e4m3,
                                           const void* mask, int mask_packed, const float* mel, float* c1_partial, const int* row_span, int rows,
                                           int W, int Cin, int Cout, dasv_stream_t stream) {
    if (mask == nullptr || mel == nullptr || c1_partial == nullptr || row_span == nullptr) return DASV_E_NULL;
    if (W > 80) return DASV_E_SHAPE;
    return conv3x3_launch(3, static_cast<const float*>(x), nullptr, nullptr, nullptr, w_f16, w_e4m3, nullptr, static_cast<const float*>(mask),
                          nullptr, nullptr, nullptr, 0, 1, rows, W, Cin, Cout, 0, stream, row_span, x_packed ? 1 : 0, 0, x_range, nullptr, 0, 0,
                          nullptr, mask_packed ? 1 : 0, mel, c1_partial);
}

int dasv_conv3x3_masked_ragged_x8_nhwc(const float* x, const uint32_t* x_range, const void* w_f16, const void* w_e4m3,
                                       const float* mask, float* y, uint32_t* y_range, float* colsum_partial, const int* row_span,
                                       int rows, int W, int Cin, int Cout, dasv_stream_t stream) {
    if (mask == nullptr || row_span == nullptr) return DASV_E_NULL;
    if (W > 80) return DASV_E_SHAPE;
    return conv3x3_launch(3, x, nullptr, nullptr, nullptr, w_f16, w_e4m3, nullptr, mask, colsum_partial, y, nullptr, 0, 1, rows, W,
                          Cin, Cout, 0, stream, row_span, 0, 0, x_range, y_range);
}

int dasv_conv3x3_colsum_rows(int mode, int B, int H, int W, int ragged) {
    if (B <= 0 || H <= 0 || W <= 0 || W > 80 || mode < 0 || mode > 3 || (ragged && B != 1)) return 0;
    // the pixel tiles of a column-sum launch (never split-K): conv3x3_launch's choice, repeated
    const bool two = mode == 0 || mode == 3;
    const size_t lds256 = (size_t)(dasv::WBM + 2 * W + 2) * 128 * (two ? 2 : 1) + 32768;
    const size_t lds192 = (size_t)(192 + 2 * W + 2) * 128 * 2 + 32768;
    const bool t192 = two && lds256 + 32768 + 128 > 160 * 1024 && lds192 + 32768 + 128 <= 160 * 1024;
    if (mode == 3 && W % 16 == 0 && W >= 48 && !ragged && dasv::g_patch_tiles) return B * ((H + 15) / 16) * (W / 16);
    const int tbm = t192 ? 192 : dasv::WBM;
    return (int)(((long long)B * H * W + tbm - 1) / tbm);
}

int dasv_conv3x3_c1_conv3x3_bias_relu_nhwc(int mode, const float* mel, const float* w1, const float* b1, const void* w_hi,
                                           const void* w_lo, const float* bias, float* y, int B, int H, int W, int Cmid,
                                           int Cout, dasv_stream_t stream) {
    return conv3x3_launch(mode, nullptr, mel, w1, b1, w_hi, w_lo, bias, nullptr, nullptr, y, nullptr, 0, B, H, W, Cmid,
                          Cout, 1, stream);
}

}  // extern "C"
