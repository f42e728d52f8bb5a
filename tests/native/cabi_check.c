/* A maintainer's-eye view of the boundary: plain C, no Python, no torch.  Allocates device buffers with the HIP runtime,
 * calls the fused pooling through include/dasv_hip.h and checks the result against the plain-C restatement
 * (oracle/dmha_oracle.c, linked in as the checker).  Built and run by tests/test_cabi_native_gpu.py.
 * Shapes: the reference's `DoubleMHA(512, 8)` on [B,T,512] (BASELINE config 2 at a size the scalar checker finishes fast). */
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
    const int ok = e_out < 1e-5 && e_al < 1e-5 && e_hw < 1e-5 && e_ctx < 1e-5 && e_null == DASV_E_NULL && e_ws == DASV_E_WORKSPACE;
    puts(ok ? "CABI_NATIVE_OK" : "CABI_NATIVE_FAIL");
    return ok ? 0 : 1;
}
