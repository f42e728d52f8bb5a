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
__global__ void __launch_bounds__(256) conv3x3_c1_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y,
                                                         int H, int W, int Cout, long long npix, int relu) {
    const int quads = Cout >> 2;                 // lanes per pixel
    const int ppb = blockDim.x / quads;          // pixels per block iteration
    const int q = threadIdx.x % quads, pl = threadIdx.x / quads;
    if (pl >= ppb) return;
    float wr[4][9];
    f4 bv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int t = 0; t < 9; ++t) wr[c][t] = w[(size_t)(q * 4 + c) * 9 + t];
        if (bias != nullptr) bv[c] = bias[q * 4 + c];
    }
    for (long long pix = (long long)blockIdx.x * ppb + pl; pix < npix; pix += (long long)gridDim.x * ppb) {
        const int w_ = (int)(pix % W), h_ = (int)((pix / W) % H);
        float in[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int hh = h_ + t / 3 - 1, ww = w_ + t % 3 - 1;
            in[t] = ((unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W) ? x[pix + (t / 3 - 1) * W + (t % 3 - 1)] : 0.f;
        }
        f4 o = bv;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < 9; ++t) o[c] = fmaf(in[t], wr[c][t], o[c]);
        if (relu) {
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = fmaxf(o[c], 0.f);
        }
        *reinterpret_cast<f4*>(y + (size_t)pix * Cout + q * 4) = o;
    }
}

// x [B,H,W,C] -> y [B,Ho,Wo,C], Ho = ceil(H/2), Wo = ceil(W/2); windows clipped at the bottom/right edge (ceil_mode)
__global__ void __launch_bounds__(256) maxpool2x2_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                              int H, int W, int C, int Ho, int Wo, long long nvec) {
    const int cq = C >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(i % cq);
        const long long op = i / cq;
        const int wo = (int)(op % Wo), ho = (int)((op / Wo) % Ho);
        const long long b = op / ((long long)Wo * Ho);
        const int h0 = 2 * ho, w0 = 2 * wo;
        const float* base = x + (((size_t)b * H + h0) * W + w0) * C + q * 4;
        f4 m = *reinterpret_cast<const f4*>(base);
        const bool hw = w0 + 1 < W, hh = h0 + 1 < H;
        if (hw) { const f4 v = *reinterpret_cast<const f4*>(base + C); m = __builtin_elementwise_max(m, v); }
        if (hh) { const f4 v = *reinterpret_cast<const f4*>(base + (size_t)W * C); m = __builtin_elementwise_max(m, v); }
        if (hw && hh) { const f4 v = *reinterpret_cast<const f4*>(base + (size_t)W * C + C); m = __builtin_elementwise_max(m, v); }
        *reinterpret_cast<f4*>(y + (size_t)op * C + q * 4) = m;
    }
}

}  // namespace dasv

extern "C" {

int dasv_conv3x3_c1_bias_relu_nhwc(const float* x, const float* w, const float* bias, float* y, int B, int H, int W,
                                   int Cout, int relu, dasv_stream_t stream) {
    if (x == nullptr || w == nullptr || y == nullptr) return DASV_E_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cout % 4 != 0 || Cout > 1024 || 256 % (Cout / 4) != 0) return DASV_E_SHAPE;
    const long long npix = (long long)B * H * W;
    const int ppb = 256 / (Cout / 4);
    long long blocks = (npix + ppb - 1) / ppb;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(dasv::conv3x3_c1_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, H, W,
                       Cout, npix, relu);
    return (int)hipGetLastError();
}

int dasv_maxpool2x2_ceil_nhwc(const float* x, float* y, int B, int H, int W, int C, dasv_stream_t stream) {
    if (x == nullptr || y == nullptr) return DASV_E_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 != 0) return DASV_E_SHAPE;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const long long nvec = (long long)B * Ho * Wo * (C / 4);
    long long blocks = (nvec + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(dasv::maxpool2x2_nhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, H, W, C, Ho, Wo, nvec);
    return (int)hipGetLastError();
}

}  // extern "C"
