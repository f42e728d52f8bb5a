// Adam step over FLAT fp32 buffers (parameters, gradients, both moments in one layout): what `optimizer.step()` does at train.py:224 with
// the optimizer of train.py:82-84 (`optim.Adam(parameters, lr, weight_decay)`: L2 weight decay added to the gradient, no amsgrad), for
// every parameter at once — one pass over 4 x 86 MB for the default model instead of torch's per-tensor-list launches, and the gradient
// buffer is cleared on the way (the trainer's memset).  Parameters that received no gradient (model.py:46,50: b1.*, b3.*) keep a slot in
// the buffers but are skipped — no update, no weight decay, as in the reference, where their `.grad` is None: `active` has one byte per
// 64-element block (slots are 64-aligned), 0 = skip.  A pure HBM stream: 7 accesses per element.
#include "dasv_common.hpp"
#include "../../include/dasv_hip.h"

namespace dasv {

typedef float af4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) adam_step_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, const uint8_t* __restrict__ active, long long nvec,
                                                        float lr_over_bc1, float b1, float b2, float eps, float wd, float rsqrt_bc2,
                                                        int zero_grad) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
        if (active != nullptr && active[i >> 4] == 0) continue;          // 16 float4 per 64-element block
        const af4 pp = reinterpret_cast<const af4*>(p)[i];
        af4 gg = reinterpret_cast<const af4*>(g)[i];
        af4 mm = reinterpret_cast<const af4*>(m)[i], vv = reinterpret_cast<const af4*>(v)[i];
        af4 out;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float gr = fmaf(wd, pp[c], gg[c]);                      // grad = grad + weight_decay * param
            mm[c] = fmaf(b1, mm[c], (1.f - b1) * gr);                     // exp_avg
            vv[c] = fmaf(b2, vv[c], (1.f - b2) * gr * gr);                // exp_avg_sq
            const float denom = sqrtf(vv[c]) * rsqrt_bc2 + eps;           // sqrt(v) / sqrt(1 - b2^t) + eps
            out[c] = pp[c] - lr_over_bc1 * (mm[c] / denom);               // param -= lr / (1 - b1^t) * m / denom
        }
        reinterpret_cast<af4*>(p)[i] = out;
        reinterpret_cast<af4*>(m)[i] = mm;
        reinterpret_cast<af4*>(v)[i] = vv;
        if (zero_grad) reinterpret_cast<af4*>(g)[i] = af4{0.f, 0.f, 0.f, 0.f};
    }
}

}  // namespace dasv

extern "C" {

int dasv_adam_step_f32(float* param, float* grad, float* exp_avg, float* exp_avg_sq, const uint8_t* active_blocks, long long n, float lr,
                       float beta1, float beta2, float eps, float weight_decay, long long step, int zero_grad, dasv_stream_t stream) {
    if (param == nullptr || grad == nullptr || exp_avg == nullptr || exp_avg_sq == nullptr) return DASV_E_NULL;
    if (n <= 0 || n % 64 != 0 || step < 1) return DASV_E_SHAPE;
    // bias corrections in double on the host (torch computes them per step from python floats)
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const long long nvec = n / 4;
    const int blocks = (int)((nvec + 255) / 256 < 8192 ? (nvec + 255) / 256 : 8192);
    hipLaunchKernelGGL(dasv::adam_step_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
                       active_blocks, nvec, (float)((double)lr / bc1), beta1, beta2, eps, weight_decay, (float)(1.0 / sqrt(bc2)), zero_grad);
    return (int)hipGetLastError();
}

}  // extern "C"
