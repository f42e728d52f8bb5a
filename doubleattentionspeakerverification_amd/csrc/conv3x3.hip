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
//   BF16: hi only (one MFMA per product).
// Tile: 128 pixels x 128 channels x 64 (one tap's channel slab) per 256-thread workgroup; 2x2 waves, each
// 64x64 = 4x4 MFMA tiles.  Global -> registers (next K-step, issued before the MFMAs of the current one) ->
// LDS (XOR-swizzled 16-byte slots so ds_read_b128 fragment reads spread over the banks) -> MFMA.
#include "dasv_common.hpp"
#include "../../include/dasv_hip.h"

namespace dasv {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CBM = 128, CBN = 128, CBK = 64;

__device__ __forceinline__ uint32_t bf16_rne(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;   // inputs are finite activations/weights
}
__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
    const uint32_t ha = bf16_rne(a), hb = bf16_rne(b);
    const float ra = a - __uint_as_float(ha << 16), rb = b - __uint_as_float(hb << 16);
    hi = ha | (hb << 16);
    lo = bf16_rne(ra) | (bf16_rne(rb) << 16);
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
    float* y;              // [B,H,W,Cout]
    int B, H, W, Cin, Cout;
    int M;                 // B*H*W
    int mtiles, ntiles, tiles_per_xcd;
    int relu;
};

// LDS: four planes of 128 rows x 128 bytes (64 bf16): A_hi | A_lo | B_hi | B_lo.  16-byte slot s of row r lives at
// r*128 + ((s ^ (r & 7)) * 16).
__device__ __forceinline__ int lds_off(int row, int slot) { return row * 128 + ((slot ^ (row & 7)) << 4); }

template <bool SPLIT>
__global__ void __launch_bounds__(256) conv3x3_igemm_kernel(const ConvParams p) {
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
    int apix[8];      // (b*H + h)*W + w of the output pixel, or -1 beyond M
    int ahw[8];       // h << 16 | w
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = m0 + arow0 + 16 * i;
        if (m < p.M) {
            const int w_ = m % W, t2 = m / W, h_ = t2 % H;
            apix[i] = m;
            ahw[i] = (h_ << 16) | w_;
        } else {
            apix[i] = -1;
            ahw[i] = 0;
        }
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
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int h_ = (ahw[i] >> 16) + dy, w_ = (ahw[i] & 0xffff) + dx;
            const bool ok = apix[i] >= 0 && (unsigned)h_ < (unsigned)H && (unsigned)w_ < (unsigned)W;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4*>(p.x + ((size_t)(apix[i] + dy * W + dx)) * Cin + c0 + ach * 4);
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
            split2(ra[i][0], ra[i][1], h0, l0);
            split2(ra[i][2], ra[i][3], h1, l1);
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
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[j], acc[i][j], 0, 0, 0);
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
                    if (p.relu) v = fmaxf(v, 0.f);
                    p.y[(size_t)m * p.Cout + n] = v;
                }
            }
        }
    }
}

}  // namespace dasv

extern "C" {

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

int dasv_conv3x3_bias_relu_nhwc(int mode, const float* x, const void* w_hi, const void* w_lo, const float* bias, float* y,
                                int B, int H, int W, int Cin, int Cout, int relu, dasv_stream_t stream) {
    if (x == nullptr || w_hi == nullptr || y == nullptr || (mode == 0 && w_lo == nullptr)) return DASV_E_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return DASV_E_SHAPE;
    if (Cin % dasv::CBK != 0 || Cout % dasv::CBN != 0 || H > 32767 || W > 65535) return DASV_E_SHAPE;
    if ((long long)B * H * W > 0x7fffffffLL / 2) return DASV_E_SHAPE;
    if (mode != 0 && mode != 1) return DASV_E_DTYPE;
    dasv::ConvParams p;
    p.x = x; p.whi = static_cast<const uint16_t*>(w_hi); p.wlo = static_cast<const uint16_t*>(w_lo); p.bias = bias; p.y = y;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.M = B * H * W; p.relu = relu;
    p.mtiles = (p.M + dasv::CBM - 1) / dasv::CBM;
    p.ntiles = Cout / dasv::CBN;
    const int tiles = p.mtiles * p.ntiles;
    p.tiles_per_xcd = (tiles + 7) / 8;
    const dim3 grid((unsigned)(p.tiles_per_xcd * 8)), block(256);
    const size_t lds = 65536;
    hipStream_t s = (hipStream_t)stream;
    if (mode == 0) {
        static bool attr0 = false;
        if (!attr0) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dasv::conv3x3_igemm_kernel<true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
            attr0 = true;
        }
        hipLaunchKernelGGL(dasv::conv3x3_igemm_kernel<true>, grid, block, lds, s, p);
    } else {
        static bool attr1 = false;
        if (!attr1) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dasv::conv3x3_igemm_kernel<false>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
            attr1 = true;
        }
        hipLaunchKernelGGL(dasv::conv3x3_igemm_kernel<false>, grid, block, lds, s, p);
    }
    return (int)hipGetLastError();
}

}  // extern "C"
