/* A maintainer's-eye view of the boundary: plain C, no Python, no torch.  Allocates device buffers with the HIP runtime,
 * calls the fused pooling through include/dasv_hip.h and checks the result against the plain-C restatement
 * (oracle/dmha_oracle.c, linked in as the checker).  Built and run by tests/test_cabi_native_gpu.py.
 * Shapes: the reference's `DoubleMHA(512, 8)` on [B,T,512] (BASELINE config 2 at a size the scalar checker finishes fast),
 * then one conv3x3+bias+ReLU layer against a naive convolution. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "dasv_hip.h"

int dmha_oracle_forward(const float* x, const float* query, const float* att, const uint8_t* keep, float* out, float* align,
                        float* head_w, float* ctx, int B, int T, int H, int dh);

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "HIP error %d at line %d\n", (int)r_, __LINE__); return 2; } } while (0)

static float frand(uint32_t* s) {   /* xorshift, uniform in [-1, 1) */
    *s ^= *s << 13; *s ^= *s >> 17; *s ^= *s << 5;
    return (float)(*s & 0xffffff) / 8388608.0f - 1.0f;
}

int main(void) {
    const int B = 5, T = 77, H = 8, dh = 64, D = H * dh;
    const size_t nx = (size_t)B * T * D;
    float *x = malloc(nx * 4), *q = malloc((size_t)dh * H * 4), *a = malloc((size_t)dh * 4);
    float *out = malloc((size_t)B * dh * 4), *align = malloc((size_t)B * T * H * 4), *hw = malloc((size_t)B * H * 4);
    float *ctx = malloc((size_t)B * H * dh * 4);
    float *r_out = malloc((size_t)B * dh * 4), *r_align = malloc((size_t)B * T * H * 4), *r_hw = malloc((size_t)B * H * 4);
    float *r_ctx = malloc((size_t)B * H * dh * 4);
    uint32_t seed = 12345u;
    for (size_t i = 0; i < nx; ++i) x[i] = frand(&seed);
    for (int i = 0; i < dh * H; ++i) q[i] = 0.2f * frand(&seed);
    for (int i = 0; i < dh; ++i) a[i] = 0.2f * frand(&seed);
    if (dmha_oracle_forward(x, q, a, NULL, r_out, r_align, r_hw, r_ctx, B, T, H, dh) != 0) return 3;

    float *dx, *dq, *da, *dout, *dalign, *dhw, *dctx, *dml;
    void* ws;
    const size_t wsb = dasv_dmha_workspace_bytes(B, T, H, dh);
    CHECK(hipMalloc((void**)&dx, nx * 4)); CHECK(hipMalloc((void**)&dq, (size_t)dh * H * 4)); CHECK(hipMalloc((void**)&da, (size_t)dh * 4));
    CHECK(hipMalloc((void**)&dout, (size_t)B * dh * 4)); CHECK(hipMalloc((void**)&dalign, (size_t)B * T * H * 4));
    CHECK(hipMalloc((void**)&dhw, (size_t)B * H * 4)); CHECK(hipMalloc((void**)&dctx, (size_t)B * H * dh * 4));
    CHECK(hipMalloc((void**)&dml, (size_t)B * H * 2 * 4)); CHECK(hipMalloc(&ws, wsb));
    CHECK(hipMemcpy(dx, x, nx * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dq, q, (size_t)dh * H * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(da, a, (size_t)dh * 4, hipMemcpyHostToDevice));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    int rc = dasv_dmha_workspace_init(ws, wsb, st);
    if (rc) { fprintf(stderr, "workspace_init: %s\n", dasv_error_string(rc)); return 4; }
    for (int rep = 0; rep < 3; ++rep) {   /* relaunch on the same workspace: the kernel leaves it zeroed */
        rc = dasv_dmha_fwd_f32(dx, dq, da, NULL, NULL, dout, dalign, dhw, dctx, dml, ws, wsb, B, T, H, dh, (int64_t)T * D, D, st);
        if (rc) { fprintf(stderr, "dasv_dmha_fwd_f32: %s\n", dasv_error_string(rc)); return 4; }
    }
    CHECK(hipStreamSynchronize(st));
    CHECK(hipMemcpy(out, dout, (size_t)B * dh * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(align, dalign, (size_t)B * T * H * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hw, dhw, (size_t)B * H * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(ctx, dctx, (size_t)B * H * dh * 4, hipMemcpyDeviceToHost));
    double e_out = 0, e_al = 0, e_hw = 0, e_ctx = 0;
    for (int i = 0; i < B * dh; ++i) e_out = fmax(e_out, fabs((double)out[i] - r_out[i]));
    for (int i = 0; i < B * T * H; ++i) e_al = fmax(e_al, fabs((double)align[i] - r_align[i]));
    for (int i = 0; i < B * H; ++i) e_hw = fmax(e_hw, fabs((double)hw[i] - r_hw[i]));
    for (int i = 0; i < B * H * dh; ++i) e_ctx = fmax(e_ctx, fabs((double)ctx[i] - r_ctx[i]));
    /* argument errors come back as codes, never as a crash */
    const int e_null = dasv_dmha_fwd_f32(NULL, dq, da, NULL, NULL, dout, dalign, dhw, dctx, dml, ws, wsb, B, T, H, dh, (int64_t)T * D, D, st);
    const int e_ws = dasv_dmha_fwd_f32(dx, dq, da, NULL, NULL, dout, dalign, dhw, dctx, dml, ws, 4, B, T, H, dh, (int64_t)T * D, D, st);
    printf("abi %d max|err| out %.3g align %.3g head_w %.3g ctx %.3g; null-> %d, small workspace -> %d\n", dasv_abi_version(),
           e_out, e_al, e_hw, e_ctx, e_null, e_ws);

    /* ---- conv3x3 + bias + ReLU (CNNs.py:38-48) through the same boundary: pack the torch-layout weights, run the
     * split-bf16 kernel on an NHWC map, compare with a naive double-precision convolution written right here ---- */
    const int cB = 2, cH = 6, cW = 5, ci = 64, co = 128;
    const size_t nxc = (size_t)cB * cH * cW * ci, nyc = (size_t)cB * cH * cW * co, nwc = (size_t)co * ci * 9;
    float *cx = malloc(nxc * 4), *cw = malloc(nwc * 4), *cb = malloc((size_t)co * 4), *cy = malloc(nyc * 4);
    for (size_t i = 0; i < nxc; ++i) cx[i] = frand(&seed);
    for (size_t i = 0; i < nwc; ++i) cw[i] = 0.06f * frand(&seed);
    for (int i = 0; i < co; ++i) cb[i] = 0.1f * frand(&seed);
    float *dcx, *dcw, *dcb, *dcy;
    void *whi, *wlo;
    const size_t pwb = dasv_conv3x3_packed_weight_bytes(co, ci);
    CHECK(hipMalloc((void**)&dcx, nxc * 4)); CHECK(hipMalloc((void**)&dcw, nwc * 4)); CHECK(hipMalloc((void**)&dcb, (size_t)co * 4));
    CHECK(hipMalloc((void**)&dcy, nyc * 4)); CHECK(hipMalloc(&whi, pwb)); CHECK(hipMalloc(&wlo, pwb));
    CHECK(hipMemcpy(dcx, cx, nxc * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dcw, cw, nwc * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dcb, cb, (size_t)co * 4, hipMemcpyHostToDevice));
    rc = dasv_conv3x3_pack_weights(dcw, whi, wlo, co, ci, st);
    if (!rc) rc = dasv_conv3x3_bias_relu_nhwc(0, dcx, whi, wlo, dcb, dcy, cB, cH, cW, ci, co, 1, st);
    if (rc) { fprintf(stderr, "conv: %s\n", dasv_error_string(rc)); return 4; }
    CHECK(hipStreamSynchronize(st));
    CHECK(hipMemcpy(cy, dcy, nyc * 4, hipMemcpyDeviceToHost));
    double e_conv = 0, s_conv = 0;
    for (int b = 0; b < cB; ++b) for (int h = 0; h < cH; ++h) for (int w = 0; w < cW; ++w) for (int n = 0; n < co; ++n) {
        double acc = cb[n];
        for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) {
            const int hh = h + ky - 1, ww = w + kx - 1;
            if (hh < 0 || hh >= cH || ww < 0 || ww >= cW) continue;
            const float* xp = cx + (((size_t)b * cH + hh) * cW + ww) * ci;           /* NHWC */
            for (int c = 0; c < ci; ++c) acc += (double)xp[c] * cw[((size_t)n * ci + c) * 9 + ky * 3 + kx];   /* [Cout,Cin,3,3] */
        }
        if (acc < 0) acc = 0;
        const double got = cy[(((size_t)b * cH + h) * cW + w) * co + n];
        e_conv = fmax(e_conv, fabs(got - acc));
        s_conv = fmax(s_conv, fabs(acc));
    }
    const int e_shape = dasv_conv3x3_bias_relu_nhwc(0, dcx, whi, wlo, dcb, dcy, cB, cH, cW, 48, co, 1, st);   /* Cin % 64 != 0 */
    printf("conv3x3: max|err| %.3g of scale %.3g; unsupported Cin -> %d\n", e_conv, s_conv, e_shape);
    const int ok = e_out < 1e-5 && e_al < 1e-5 && e_hw < 1e-5 && e_ctx < 1e-5 && e_null == DASV_E_NULL && e_ws == DASV_E_WORKSPACE &&
                   e_conv < 2e-5 * s_conv && e_shape == DASV_E_SHAPE;
    puts(ok ? "CABI_NATIVE_OK" : "CABI_NATIVE_FAIL");
    return ok ? 0 : 1;
}
