// Shared device helpers for the gfx950 DoubleMHA kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace dasv {

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

// ---- element types of the frame embeddings -------------------------------
struct F32 { using storage = float; static constexpr int kCode = 0; };
struct BF16 { using storage = uint16_t; static constexpr int kCode = 1; };
struct F16 { using storage = uint16_t; static constexpr int kCode = 2; };

template <typename XT> struct Elem;
template <> struct Elem<F32> {
    static constexpr int kBytes = 4;
    static constexpr int kPerVec = 4;  // elements in one 16-byte vector
    __device__ static __forceinline__ float load1(const void* p, long long i) { return static_cast<const float*>(p)[i]; }
    __device__ static __forceinline__ void store1(void* p, long long i, float v) { static_cast<float*>(p)[i] = v; }
    __device__ static __forceinline__ void unpack(const uint4& v, float* f) {
        f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y);
        f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
    }
    __device__ static __forceinline__ uint4 pack(const float* f) {
        return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    }
};
template <> struct Elem<BF16> {
    static constexpr int kBytes = 2;
    static constexpr int kPerVec = 8;
    __device__ static __forceinline__ float cvt(uint32_t lo16) { return __uint_as_float(lo16 << 16); }
    __device__ static __forceinline__ uint32_t rnd(float f) {  // round-to-nearest-even, NaN stays NaN
        uint32_t u = __float_as_uint(f);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
        return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
    }
    __device__ static __forceinline__ float load1(const void* p, long long i) { return cvt(static_cast<const uint16_t*>(p)[i]); }
    __device__ static __forceinline__ void store1(void* p, long long i, float v) { static_cast<uint16_t*>(p)[i] = (uint16_t)rnd(v); }
    __device__ static __forceinline__ void unpack(const uint4& v, float* f) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) { f[2 * k] = __uint_as_float(w[k] << 16); f[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u); }
    }
    __device__ static __forceinline__ uint4 pack(const float* f) {
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = rnd(f[2 * k]) | (rnd(f[2 * k + 1]) << 16);
        return make_uint4(w[0], w[1], w[2], w[3]);
    }
};
template <> struct Elem<F16> {
    static constexpr int kBytes = 2;
    static constexpr int kPerVec = 8;
    __device__ static __forceinline__ float cvt(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
    __device__ static __forceinline__ uint16_t rnd(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
    __device__ static __forceinline__ float load1(const void* p, long long i) { return cvt(static_cast<const uint16_t*>(p)[i]); }
    __device__ static __forceinline__ void store1(void* p, long long i, float v) { static_cast<uint16_t*>(p)[i] = rnd(v); }
    __device__ static __forceinline__ void unpack(const uint4& v, float* f) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) { f[2 * k] = cvt((uint16_t)(w[k] & 0xffffu)); f[2 * k + 1] = cvt((uint16_t)(w[k] >> 16)); }
    }
    __device__ static __forceinline__ uint4 pack(const float* f) {
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = (uint32_t)rnd(f[2 * k]) | ((uint32_t)rnd(f[2 * k + 1]) << 16);
        return make_uint4(w[0], w[1], w[2], w[3]);
    }
};

// ---- cross-lane reductions (all lanes of the wave must be active) --------
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// sum over aligned groups of G lanes (G a power of two <= 64); every lane gets its group's sum.
template <int G> __device__ __forceinline__ float group_sum(float v) {
    if constexpr (G >= 2) v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
    if constexpr (G >= 4) v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    if constexpr (G >= 8) v += dpp_mov<0x141>(v);   // row_half_mirror
    if constexpr (G >= 16) v += dpp_mov<0x140>(v);  // row_mirror
    if constexpr (G >= 32) v += __shfl_xor(v, 16);
    if constexpr (G >= 64) v += __shfl_xor(v, 32);
    return v;
}
template <int G> __device__ __forceinline__ float group_max(float v) {
    if constexpr (G >= 2) v = fmaxf(v, dpp_mov<0xB1>(v));
    if constexpr (G >= 4) v = fmaxf(v, dpp_mov<0x4E>(v));
    if constexpr (G >= 8) v = fmaxf(v, dpp_mov<0x141>(v));
    if constexpr (G >= 16) v = fmaxf(v, dpp_mov<0x140>(v));
    if constexpr (G >= 32) v = fmaxf(v, __shfl_xor(v, 16));
    if constexpr (G >= 64) v = fmaxf(v, __shfl_xor(v, 32));
    return v;
}

// Reductions over the whole wave WITHOUT the LDS crossbar (ds_bpermute, ~100+ cycles of latency per step): inside a row of 16
// lanes by DPP permutations, across the four rows by row_bcast:15 / row_bcast:31 (GFX9 DPP controls: lane 15 of a row to the next
// row, lane 31 to rows 2 and 3), the result read from lane 63 into a scalar register — every lane gets it.  `ROW_UNIFORM_G`: the
// value is already uniform over aligned groups of that many lanes (1 = arbitrary per-lane values).
template <int ROW_UNIFORM_G> __device__ __forceinline__ float wave_sum_all(float v) {
    if constexpr (ROW_UNIFORM_G < 2) v += dpp_mov<0xB1>(v);
    if constexpr (ROW_UNIFORM_G < 4) v += dpp_mov<0x4E>(v);
    if constexpr (ROW_UNIFORM_G < 8) v += dpp_mov<0x141>(v);
    if constexpr (ROW_UNIFORM_G < 16) v += dpp_mov<0x140>(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));   // rows 1, 3 += rows 0, 2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, false));   // rows 2, 3 += row 1
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
template <int ROW_UNIFORM_G> __device__ __forceinline__ float wave_max_all(float v) {
    if constexpr (ROW_UNIFORM_G < 2) v = fmaxf(v, dpp_mov<0xB1>(v));
    if constexpr (ROW_UNIFORM_G < 4) v = fmaxf(v, dpp_mov<0x4E>(v));
    if constexpr (ROW_UNIFORM_G < 8) v = fmaxf(v, dpp_mov<0x141>(v));
    if constexpr (ROW_UNIFORM_G < 16) v = fmaxf(v, dpp_mov<0x140>(v));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x142, 0xA, 0xF, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x143, 0xC, 0xF, false)));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }  // v_exp_f32

// write-through (agent-scope) accesses for data exchanged between workgroups inside one launch
__device__ __forceinline__ void store_agent(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float load_agent(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__host__ __device__ inline int round_up_i(int a, int m) { return (a + m - 1) / m * m; }

// ---- range records ------------------------------------------------------------------------------------------------------------
// A RANGE RECORD is two 32-bit words in device memory that travel beside an activation tensor (split8 convolutions, conv3x3.hip):
//   [0]  bit pattern of max |v| over the tensor, kept with an unsigned atomic maximum of the sign-cleared bits by the kernel that
//        writes the tensor (a NaN sorts above every number, so one NaN anywhere is recorded as such); zero before that kernel runs;
//   [1]  for a PACKED tensor: the power-of-two exponent E its planes were made with (they hold v * 2^E), written by its producer.
// The fp16 / e4m3 operand planes of the split8 mode only have fp16's and e4m3's exponent range; the record is what lets every
// tensor be scaled into that range by an exact power of two chosen ON THE DEVICE (no host round trip), whatever its magnitude.
__device__ __forceinline__ int x8_exp_for(uint32_t amax_bits) {   // E with amax * 2^E in [128, 256); 0 for Inf / NaN (they then propagate)
    const int be = (int)((amax_bits >> 23) & 0xffu);
    if (be == 255) return 0;
    const int E = 134 - be;
    return E > 126 ? 126 : E;
}
__device__ __forceinline__ float x8_pow2(int E) { return __uint_as_float((uint32_t)(E + 127) << 23); }   // E in [-126, 127]
// Recording a maximum.  Atomics to ONE address execute one after the other at the memory side (~12 ns each): a launch of 8,000
// workgroups in which every wave posts its maximum keeps that address busy for most of a millisecond and slows the whole channel
// (measured: the split8 layers +2 ... +11 %, conv11 +60 %, before this was thinned).  So: reduce inside the wave (DPP), then inside the
// workgroup through one LDS word, and post to memory only what exceeds the value already there — after the first round of
// workgroups almost nothing does.
__device__ __forceinline__ uint32_t wave_umax(uint32_t m) {
    auto dpp = [](uint32_t v, auto ctrl) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, decltype(ctrl)::value, 0xF, 0xF, true); };
    m = max(m, dpp(m, std::integral_constant<int, 0xB1>{}));    // quad_perm [1,0,3,2]
    m = max(m, dpp(m, std::integral_constant<int, 0x4E>{}));    // quad_perm [2,3,0,1]
    m = max(m, dpp(m, std::integral_constant<int, 0x141>{}));   // row_half_mirror
    m = max(m, dpp(m, std::integral_constant<int, 0x140>{}));   // row_mirror
    m = max(m, (uint32_t)__shfl_xor((int)m, 16));
    m = max(m, (uint32_t)__shfl_xor((int)m, 32));
    return m;
}
__device__ __forceinline__ uint32_t range_seen(const uint32_t* rec) {   // what the record holds now (L2-served: other CUs' atomics are visible)
    return __hip_atomic_load(rec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every thread of the workgroup calls this once, with uniform control flow; `slot` is a __shared__ word that was zeroed behind an
// earlier barrier.  `seen`: a value read from the record earlier in the launch (range_seen), or 0 to read it now.
__device__ __forceinline__ void range_note_block(uint32_t* rec, uint32_t m, uint32_t* slot, uint32_t seen = 0u) {
    m = wave_umax(m);
    if ((threadIdx.x & 63) == 0 && m != 0u) atomicMax(slot, m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t v = *slot;
        if (v != 0u && v > (seen != 0u ? seen : range_seen(rec))) atomicMax(rec, v);
    }
}
// the same per wave (kernels whose epilogue has no workgroup barrier)
__device__ __forceinline__ void range_note(uint32_t* rec, uint32_t m, uint32_t seen) {
    m = wave_umax(m);
    if ((threadIdx.x & 63) == 0 && m > seen) atomicMax(rec, m);
}
// relu that keeps a NaN a NaN (v_max_f32 returns the other operand): torch.relu(NaN) is NaN, and a diverging run must show it
__device__ __forceinline__ float relu_nan(float v) { return !(v <= 0.f) ? v : 0.f; }

// ---- "split8" operand planes (csrc/conv3x3.hip, PREC_X8): scaled value c = v * 2^E = fp16 hi + residual r --------------------
constexpr int X8_ACT_SHIFT = 12;   // activation residuals are stored as e4m3(r * 2^12)
typedef float x8_f32x4 __attribute__((ext_vector_type(4)));
typedef float x8_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 x8_f16x2 __attribute__((ext_vector_type(2)));
// X8 staging of four fp32 values scaled by s = 2^E (E from the tensor's range record: |v| * s < 256): fp16 hi (two words), e4m3 of
// the scaled values, e4m3 of the scaled residuals.  Nothing is clamped: inside the recorded range nothing can overflow (fp16 holds
// 65504, e4m3 448; |c| < 256 and |r| * 2^12 <= 256), and a NaN / Inf input stays one (v_cvt_pk_fp8_f32 turns |v| >= 480 into NaN:
// tools/fp8_mfma_check.hip) — round 2 clamped here with v_med3, which returns a finite value for a NaN operand and so hid
// diverged activations.  Values below 2^-14 / s lose bits in the fp16 plane (subnormal): they are 2^-22 of the tensor's maximum.
__device__ __forceinline__ void cvt4_x8(x8_f32x4 v, float s, uint32_t& h01, uint32_t& h23, uint32_t& v8, uint32_t& r8) {
    const x8_f32x2 ss = {s, s};
    const x8_f32x2 c01 = x8_f32x2{v[0], v[1]} * ss, c23 = x8_f32x2{v[2], v[3]} * ss;
    const x8_f16x2 p0 = __builtin_convertvector(c01, x8_f16x2), p1 = __builtin_convertvector(c23, x8_f16x2);
    const x8_f32x2 rs = {(float)(1 << X8_ACT_SHIFT), (float)(1 << X8_ACT_SHIFT)};
    const x8_f32x2 r01 = (c01 - __builtin_convertvector(p0, x8_f32x2)) * rs, r23 = (c23 - __builtin_convertvector(p1, x8_f32x2)) * rs;
    h01 = __builtin_bit_cast(uint32_t, p0);
    h23 = __builtin_bit_cast(uint32_t, p1);
    const int a = __builtin_amdgcn_cvt_pk_fp8_f32(c01[0], c01[1], (int)h01, false);
    v8 = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(c23[0], c23[1], a, true);
    const int b = __builtin_amdgcn_cvt_pk_fp8_f32(r01[0], r01[1], (int)h23, false);
    r8 = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(r23[0], r23[1], b, true);
}

// A PACKED activation tensor has the geometry of the fp32 NHWC tensor it replaces — [pixels][C] floats, C % 64 == 0 — but every
// 256-byte slab of 64 channels holds { fp16(c) x 64 | e4m3(c) x 64 | e4m3(r * 2^12) x 64 } with c = v * 2^E and E in the tensor's
// range record: what the split8 convolution would make of the fp32 values while staging its window, made ONCE by the producer (a
// conv epilogue, conv11, a pool).  The consumer's staging is then a 16-byte copy with the addresses it already uses.  Four
// consecutive channels n .. n+3 (n % 4 == 0) of pixel `pix`:
__device__ __forceinline__ void store_x8(float* y, long long pix, int C, int n, x8_f32x4 v, float s) {
    uint32_t h01, h23, v8, r8;
    cvt4_x8(v, s, h01, h23, v8, r8);
    char* slab = reinterpret_cast<char*>(y + (size_t)pix * C + (n & ~63));
    const int cs = n & 63;
    *reinterpret_cast<uint2*>(slab + cs * 2) = make_uint2(h01, h23);
    *reinterpret_cast<uint32_t*>(slab + 128 + cs) = v8;
    *reinterpret_cast<uint32_t*>(slab + 192 + cs) = r8;
}

}  // namespace dasv
