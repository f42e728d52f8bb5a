/* A maintainer's-eye view of the boundary: plain C, no Python, no torch.  Allocates device buffers with the HIP runtime,
 * calls the fused pooling through include/dasv_hip.h and checks the result against the plain-C restatement
 * (oracle/dmha_oracle.c, linked in as the checker).  Built and run by tests/test_cabi_native_gpu.py.
 * Shapes: the reference's `DoubleMHA(512, 8)` on [B,T,512] (BASELINE config 2 at a size the scalar checker finishes fast),
 * then one conv3x3+bias+ReLU layer against a naive convolution, then the DEFAULT path of the front-end — range record ->
 * split8 convolution -> max-pool, and convolution + pool as one launch — the AM-softmax + cross-entropy head, forward and
 * backward, and the TRAINING entry points of the default arithmetic (conv + pool with arg-max bytes, pool backward, masked data
 * gradient, weight gradient, column sums, flat Adam), each against a plain-C double-precision restatement written right here. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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
    int ok = e_out < 1e-5 && e_al < 1e-5 && e_hw < 1e-5 && e_ctx < 1e-5 && e_null == DASV_E_NULL && e_ws == DASV_E_WORKSPACE &&
             e_conv < 2e-5 * s_conv && e_shape == DASV_E_SHAPE;

    /* ---- the DEFAULT path of the front-end (ABI v2 / v3): range record -> split8 convolution -> max-pool, and the same
     * convolution + pool as ONE launch, against the naive double-precision result of the first part (relu'd map `ref`) ---- */
    double* ref = malloc(nyc * sizeof(double));
    float amax_x = 0.f;
    for (size_t i = 0; i < nxc; ++i) amax_x = fmaxf(amax_x, fabsf(cx[i]));
    for (int b = 0; b < cB; ++b) for (int h = 0; h < cH; ++h) for (int w = 0; w < cW; ++w) for (int n = 0; n < co; ++n) {
        double acc = cb[n];
        for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) {
            const int hh = h + ky - 1, ww = w + kx - 1;
            if (hh < 0 || hh >= cH || ww < 0 || ww >= cW) continue;
            const float* xp = cx + (((size_t)b * cH + hh) * cW + ww) * ci;
            for (int c = 0; c < ci; ++c) acc += (double)xp[c] * cw[((size_t)n * ci + c) * 9 + ky * 3 + kx];
        }
        ref[(((size_t)b * cH + h) * cW + w) * co + n] = acc < 0 ? 0 : acc;
    }
    uint32_t* drec;                                   /* four range records: x, y, pooled (two-launch), pooled (one launch) */
    CHECK(hipMalloc((void**)&drec, 8 * sizeof(uint32_t)));
    CHECK(hipMemsetAsync(drec, 0, 8 * sizeof(uint32_t), st));
    void *w16, *w8;
    CHECK(hipMalloc(&w16, dasv_conv3x3_packed_weight_bytes_f16(co, ci)));
    CHECK(hipMalloc(&w8, dasv_conv3x3_packed_weight_bytes_x8(co, ci)));
    const int pH = (cH + 1) / 2, pW = (cW + 1) / 2;
    const size_t npool = (size_t)cB * pH * pW * co;
    float *dpy, *dpy1, *py = malloc(npool * 4), *py1 = malloc(npool * 4);
    CHECK(hipMalloc((void**)&dpy, npool * 4)); CHECK(hipMalloc((void**)&dpy1, npool * 4));
    rc = dasv_range_absmax_f32(dcx, nxc, drec, st);
    if (!rc) rc = dasv_conv3x3_pack_weights_x8(dcw, dcb, w16, w8, co, ci, st);
    if (!rc) rc = dasv_conv3x3_bias_relu_x8_nhwc(dcx, 0, drec, w16, w8, dcb, dcy, 0, drec + 2, NULL, 0, cB, cH, cW, ci, co, 1, st);
    if (!rc) rc = dasv_maxpool2x2_ceil_nhwc(dcy, drec + 2, dpy, drec + 4, cB, cH, cW, co, st);
    if (!rc) rc = dasv_conv3x3_pool_supported(3, cB, cH, cW, ci, co) ? 0 : DASV_E_SHAPE;
    if (!rc) rc = dasv_conv3x3_bias_relu_pool_x8_nhwc(dcx, 0, drec, w16, w8, dcb, dpy1, 0, drec + 6, cB, cH, cW, ci, co, 0, st);
    if (rc) { fprintf(stderr, "split8 path: %s\n", dasv_error_string(rc)); return 4; }
    uint32_t rec[8];
    CHECK(hipStreamSynchronize(st));
    CHECK(hipMemcpy(rec, drec, sizeof(rec), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(cy, dcy, nyc * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(py, dpy, npool * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(py1, dpy1, npool * 4, hipMemcpyDeviceToHost));
    double e_x8 = 0, e_pool = 0, amax_y = 0;
    int same = 1;
    for (size_t i = 0; i < nyc; ++i) { e_x8 = fmax(e_x8, fabs((double)cy[i] - ref[i])); amax_y = fmax(amax_y, (double)cy[i]); }
    for (int b = 0; b < cB; ++b) for (int h = 0; h < pH; ++h) for (int w = 0; w < pW; ++w) for (int n = 0; n < co; ++n) {
        double m = 0;                                /* relu'd inputs: the window maximum is >= 0 */
        for (int dy = 0; dy < 2; ++dy) for (int dx2 = 0; dx2 < 2; ++dx2) {
            const int hh = 2 * h + dy, ww = 2 * w + dx2;
            if (hh < cH && ww < cW) m = fmax(m, ref[(((size_t)b * cH + hh) * cW + ww) * co + n]);
        }
        const size_t o = (((size_t)b * pH + h) * pW + w) * co + n;
        e_pool = fmax(e_pool, fabs((double)py1[o] - m));
        if (py[o] != py1[o]) same = 0;               /* one launch == two launches, bit for bit */
    }
    float fx, fy;
    memcpy(&fx, &rec[0], 4); memcpy(&fy, &rec[2], 4);
    printf("split8 conv: max|err| %.3g of scale %.3g; records x %.6g (true %.6g) y %.6g (true %.6g) pooled %08x / %08x; conv+pool in one "
           "launch: max|err| %.3g, identical to two launches: %d\n", e_x8, s_conv, fx, amax_x, fy, amax_y, rec[4], rec[6], e_pool, same);
    ok = ok && e_x8 < 4e-5 * s_conv && e_pool < 4e-5 * s_conv && same && fx == amax_x && (double)fy == amax_y && rec[4] == rec[2] &&
         rec[6] == rec[2];

    /* ---- AM-softmax + cross-entropy head (loss.py:37-52, train.py:219): forward and backward against the closed forms in double ---- */
    {
        const int aB = 5, aE = 24, aN = 37;
        const float margin = 0.4f, scale = 30.0f;
        float *ax = malloc((size_t)aB * aE * 4), *aW = malloc((size_t)aE * aN * 4);
        long long lab[5] = {3, 36, 0, 17, 3};
        for (int i = 0; i < aB * aE; ++i) ax[i] = 2.0f * frand(&seed);
        for (int i = 0; i < aE * aN; ++i) aW[i] = frand(&seed);
        double xn[5], wn[37], cosr[5 * 37], lossr = 0, dcos[5 * 37];
        for (int b = 0; b < aB; ++b) { double q2 = 0; for (int e = 0; e < aE; ++e) q2 += (double)ax[b * aE + e] * ax[b * aE + e]; xn[b] = fmax(sqrt(q2), 1e-12); }
        for (int n = 0; n < aN; ++n) { double q2 = 0; for (int e = 0; e < aE; ++e) q2 += (double)aW[e * aN + n] * aW[e * aN + n]; wn[n] = fmax(sqrt(q2), 1e-12); }
        for (int b = 0; b < aB; ++b) {
            double mx = -1e300, se = 0;
            for (int n = 0; n < aN; ++n) {
                double d = 0;
                for (int e = 0; e < aE; ++e) d += (double)ax[b * aE + e] * aW[e * aN + n];
                cosr[b * aN + n] = d / (xn[b] * wn[n]);
                const double lg = scale * (cosr[b * aN + n] - (n == lab[b] ? margin : 0.0));
                mx = fmax(mx, lg);
            }
            for (int n = 0; n < aN; ++n) se += exp(scale * (cosr[b * aN + n] - (n == lab[b] ? margin : 0.0)) - mx);
            for (int n = 0; n < aN; ++n) {
                const double lg = scale * (cosr[b * aN + n] - (n == lab[b] ? margin : 0.0));
                dcos[b * aN + n] = scale * (exp(lg - mx) / se - (n == lab[b] ? 1.0 : 0.0)) / aB;
            }
            lossr += (mx + log(se)) - scale * (cosr[b * aN + lab[b]] - margin);
        }
        lossr /= aB;
        float *dax, *daW, *dcosth, *dloss, *dlse, *dxinv, *dwinv, *dgx, *dgW;
        long long* dlab;
        void* aws;
        const size_t awb = dasv_amsoftmax_workspace_bytes(aB, aE, aN);
        CHECK(hipMalloc((void**)&dax, (size_t)aB * aE * 4)); CHECK(hipMalloc((void**)&daW, (size_t)aE * aN * 4));
        CHECK(hipMalloc((void**)&dcosth, (size_t)aB * aN * 4)); CHECK(hipMalloc((void**)&dloss, 4)); CHECK(hipMalloc((void**)&dlse, aB * 4));
        CHECK(hipMalloc((void**)&dxinv, aB * 4)); CHECK(hipMalloc((void**)&dwinv, aN * 4)); CHECK(hipMalloc((void**)&dgx, (size_t)aB * aE * 4));
        CHECK(hipMalloc((void**)&dgW, (size_t)aE * aN * 4)); CHECK(hipMalloc((void**)&dlab, aB * 8)); CHECK(hipMalloc(&aws, awb));
        CHECK(hipMemcpy(dax, ax, (size_t)aB * aE * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(daW, aW, (size_t)aE * aN * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dlab, lab, aB * 8, hipMemcpyHostToDevice));
        rc = dasv_amsoftmax_fwd(dax, daW, (const int64_t*)dlab, dcosth, NULL, dloss, dlse, dxinv, dwinv, aws, awb, aB, aE, aN, margin, scale, 0.f, st);
        if (!rc) rc = dasv_amsoftmax_bwd(dax, daW, (const int64_t*)dlab, dcosth, dlse, dxinv, dwinv, NULL, NULL, NULL, dgx, dgW, aws, awb,
                                         aB, aE, aN, margin, scale, 0.f, st);
        if (rc) { fprintf(stderr, "amsoftmax: %s\n", dasv_error_string(rc)); return 4; }
        float loss_h, *costh_h = malloc((size_t)aB * aN * 4), *gx_h = malloc((size_t)aB * aE * 4), *gW_h = malloc((size_t)aE * aN * 4);
        CHECK(hipStreamSynchronize(st));
        CHECK(hipMemcpy(&loss_h, dloss, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(costh_h, dcosth, (size_t)aB * aN * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(gx_h, dgx, (size_t)aB * aE * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(gW_h, dgW, (size_t)aE * aN * 4, hipMemcpyDeviceToHost));
        double e_cos = 0, e_gx = 0, s_gx = 0, e_gW = 0, s_gW = 0;
        for (int i = 0; i < aB * aN; ++i) e_cos = fmax(e_cos, fabs((double)costh_h[i] - cosr[i]));
        for (int b = 0; b < aB; ++b) {                       /* dx = (d xhat - xhat <xhat, d xhat>) / |x| */
            double u = 0;
            for (int n = 0; n < aN; ++n) u += dcos[b * aN + n] * cosr[b * aN + n];
            for (int e = 0; e < aE; ++e) {
                double v = 0;
                for (int n = 0; n < aN; ++n) v += dcos[b * aN + n] * aW[e * aN + n] / wn[n];
                const double g = (v - ax[b * aE + e] / xn[b] * u) / xn[b];
                e_gx = fmax(e_gx, fabs((double)gx_h[b * aE + e] - g)); s_gx = fmax(s_gx, fabs(g));
            }
        }
        for (int n = 0; n < aN; ++n) {                       /* dW_n = (d What_n - What_n <What_n, d What_n>) / |W_n| */
            double t = 0;
            for (int b = 0; b < aB; ++b) t += dcos[b * aN + n] * cosr[b * aN + n];
            for (int e = 0; e < aE; ++e) {
                double v = 0;
                for (int b = 0; b < aB; ++b) v += dcos[b * aN + n] * ax[b * aE + e] / xn[b];
                const double g = (v - aW[e * aN + n] / wn[n] * t) / wn[n];
                e_gW = fmax(e_gW, fabs((double)gW_h[e * aN + n] - g)); s_gW = fmax(s_gW, fabs(g));
            }
        }
        printf("amsoftmax: loss %.7g (ref %.7g), max|err| costh %.3g, gx %.3g of %.3g, gW %.3g of %.3g\n", loss_h, lossr, e_cos, e_gx, s_gx,
               e_gW, s_gW);
        ok = ok && fabs(loss_h - lossr) < 2e-5 * fabs(lossr) && e_cos < 2e-6 && e_gx < 1e-4 * s_gx && e_gW < 1e-4 * s_gW;
    }
    /* ---- the TRAINING entry points of the default arithmetic (train.py:215-224 through CNNs.py:72-74): conv + ReLU + pool in one
     * launch with arg-max bytes, the pool's backward from those bytes, the masked data gradient with planes packed straight from the
     * forward weight, the weight gradient, the column-sum reduction and the flat Adam step — each against plain C in double ---- */
    {
        const int tB = 2, tH = 6, tW = 5, tc = 64, tn = 128, qH = 3, qW = 3;     /* layer A: conv + pool, pool backward */
        const size_t nX = (size_t)tB * tH * tW * tc, nY = (size_t)tB * tH * tW * tn, nW = (size_t)tn * tc * 9, nP = (size_t)tB * qH * qW * tn;
        float *tx = malloc(nX * 4), *tw = malloc(nW * 4), *tb = malloc((size_t)tn * 4), *tgy = malloc(nP * 4);
        for (size_t i = 0; i < nX; ++i) { tx[i] = frand(&seed); if (tx[i] < 0) tx[i] = 0; }          /* a ReLU output */
        for (size_t i = 0; i < nW; ++i) tw[i] = 0.05f * frand(&seed);
        for (int i = 0; i < tn; ++i) tb[i] = 0.1f * frand(&seed);
        for (size_t i = 0; i < nP; ++i) tgy[i] = 1e-3f * frand(&seed);
        /* forward in double: z (pre-ReLU map), then per window the first maximum in row-major order */
        double* z = malloc(nY * sizeof(double));
        for (int b = 0; b < tB; ++b) for (int h = 0; h < tH; ++h) for (int w = 0; w < tW; ++w) for (int n = 0; n < tn; ++n) {
            double acc = tb[n];
            for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) {
                const int hh = h + ky - 1, ww = w + kx - 1;
                if (hh < 0 || hh >= tH || ww < 0 || ww >= tW) continue;
                const float* xp = tx + (((size_t)b * tH + hh) * tW + ww) * tc;
                for (int c = 0; c < tc; ++c) acc += (double)xp[c] * tw[((size_t)n * tc + c) * 9 + ky * 3 + kx];
            }
            z[(((size_t)b * tH + h) * tW + w) * tn + n] = acc < 0 ? 0 : acc;
        }
        float *dtx, *dtw, *dtb, *dty, *dpool, *dgy, *dg, *dgx, *dgw, *dcol, *dcolsum;
        uint8_t* didx;
        uint32_t* trec;
        void *tw16, *tw8, *dw16, *dw8, *wws;
        const size_t wwb = dasv_conv3x3_wgrad_workspace_bytes(tB, tH, tW, 128, tn);
        const int crow = dasv_maxpool2x2_bwd_colsum_rows(tB, tH, tW, tn);
        CHECK(hipMalloc((void**)&dtx, nX * 4)); CHECK(hipMalloc((void**)&dtw, nW * 4)); CHECK(hipMalloc((void**)&dtb, (size_t)tn * 4));
        CHECK(hipMalloc((void**)&dty, nY * 4)); CHECK(hipMalloc((void**)&dpool, nP * 4)); CHECK(hipMalloc((void**)&dgy, nP * 4));
        CHECK(hipMalloc((void**)&dg, nY * 4)); CHECK(hipMalloc((void**)&dgx, 2 * nX * 4)); CHECK(hipMalloc((void**)&dgw, 2 * nW * 4));
        CHECK(hipMalloc((void**)&dcol, (size_t)crow * tn * 4)); CHECK(hipMalloc((void**)&dcolsum, (size_t)tn * 4));
        CHECK(hipMalloc((void**)&didx, nP)); CHECK(hipMalloc((void**)&trec, 16 * sizeof(uint32_t)));
        CHECK(hipMalloc(&tw16, dasv_conv3x3_packed_weight_bytes_f16(tn, tc))); CHECK(hipMalloc(&tw8, dasv_conv3x3_packed_weight_bytes_x8(tn, tc)));
        CHECK(hipMalloc(&dw16, dasv_conv3x3_packed_weight_bytes_f16(128, tn))); CHECK(hipMalloc(&dw8, dasv_conv3x3_packed_weight_bytes_x8(128, tn)));
        CHECK(hipMalloc(&wws, wwb));
        CHECK(hipMemcpy(dtx, tx, nX * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dtw, tw, nW * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dtb, tb, (size_t)tn * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dgy, tgy, nP * 4, hipMemcpyHostToDevice));
        CHECK(hipMemsetAsync(trec, 0, 16 * sizeof(uint32_t), st));
        rc = dasv_range_absmax_f32(dtx, nX, trec, st);                                   /* record 0: x */
        if (!rc) rc = dasv_range_absmax_f32(dgy, nP, trec + 2, st);                      /* record 1: the pooled gradient */
        if (!rc) rc = dasv_conv3x3_pack_weights_x8(dtw, dtb, tw16, tw8, tn, tc, st);
        if (!rc) rc = dasv_conv3x3_pool_supported(3, tB, tH, tW, tc, tn) ? 0 : DASV_E_SHAPE;
        if (!rc) rc = dasv_conv3x3_bias_relu_pool_idx_x8_nhwc(dtx, 0, trec, tw16, tw8, dtb, dpool, 0, trec + 4, didx, tB, tH, tW, tc, tn, st);
        if (!rc) rc = dasv_maxpool2x2_ceil_bwd_idx_nhwc(didx, dgy, dg, dcol, 1, tB, tH, tW, tn, st);      /* g = gradient at conv's pre-activation */
        if (!rc) rc = dasv_colsum_reduce(dcol, crow, tn, dcolsum, st);                                  /* = the bias gradient */
        /* layer B (128 -> 128; the data-gradient convolution writes 128-channel tiles): masked data gradient and weight gradient of the
         * gradient tensor g made above, against an input xb and weights wb of their own */
        const int bc = 128;
        const size_t nXb = (size_t)tB * tH * tW * bc, nWb = (size_t)tn * bc * 9;
        float *xb = malloc(nXb * 4), *wb = malloc(nWb * 4), *dxb, *dwb;
        for (size_t i = 0; i < nXb; ++i) { xb[i] = frand(&seed); if (xb[i] < 0) xb[i] = 0; }
        for (size_t i = 0; i < nWb; ++i) wb[i] = 0.05f * frand(&seed);
        CHECK(hipMalloc((void**)&dxb, nXb * 4)); CHECK(hipMalloc((void**)&dwb, nWb * 4));
        CHECK(hipMemcpy(dxb, xb, nXb * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dwb, wb, nWb * 4, hipMemcpyHostToDevice));
        if (!rc) rc = dasv_range_absmax_f32(dxb, nXb, trec + 8, st);
        if (!rc) rc = dasv_conv3x3_pack_weights_x8_dgrad(dwb, dw16, dw8, tn, bc, st);
        if (!rc) rc = dasv_conv3x3_masked_x8_nhwc(dg, trec + 2, dw16, dw8, dxb, dgx, trec + 6, NULL, tB, tH, tW, tn, bc, st);
        if (!rc) rc = dasv_conv3x3_wgrad_x8_nhwc(dxb, trec + 8, dg, trec + 2, dgw, wws, wwb, tB, tH, tW, bc, tn, st);
        if (rc) { fprintf(stderr, "training path: %s\n", dasv_error_string(rc)); return 4; }
        float *pool_h = malloc(nP * 4), *g_h = malloc(nY * 4), *gx_h = malloc(nXb * 4), *gw_h = malloc(nWb * 4), colsum_h[128];
        uint8_t* idx_h = malloc(nP);
        CHECK(hipStreamSynchronize(st));
        CHECK(hipMemcpy(pool_h, dpool, nP * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(idx_h, didx, nP, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(g_h, dg, nY * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(gx_h, dgx, nXb * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(gw_h, dgw, nWb * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(colsum_h, dcolsum, (size_t)tn * 4, hipMemcpyDeviceToHost));
        /* pooled map and arg-max bytes; a window whose two largest values agree to 1e-5 may hand on either: counted, must be rare */
        double e_pl = 0, s_pl = 0, *gref = calloc(nY, sizeof(double)), gbr[128] = {0};
        int near_ties = 0, bad_idx = 0;
        for (int b = 0; b < tB; ++b) for (int h = 0; h < qH; ++h) for (int w = 0; w < qW; ++w) for (int n = 0; n < tn; ++n) {
            double m = -1, second = -1;
            int arg = 0;
            for (int pos = 0; pos < 4; ++pos) {
                const int hh = 2 * h + (pos >> 1), ww = 2 * w + (pos & 1);
                if (hh >= tH || ww >= tW) continue;
                const double v = z[(((size_t)b * tH + hh) * tW + ww) * tn + n];
                if (v > m) { second = m; m = v; arg = pos; } else if (v > second) second = v;
            }
            const size_t o = (((size_t)b * qH + h) * qW + w) * tn + n;
            e_pl = fmax(e_pl, fabs((double)pool_h[o] - m)); s_pl = fmax(s_pl, m);
            const int code = idx_h[o], got_arg = code & 3, got_pos = (code & 4) != 0;
            if (got_arg != arg || got_pos != (m > 0)) { if (m - second < 1e-4 || m < 1e-4) ++near_ties; else ++bad_idx; }
            /* the backward follows the LAUNCH's decisions (its own bytes): g at the byte's position, if the byte says positive */
            const int hh = 2 * h + (got_arg >> 1), ww = 2 * w + (got_arg & 1);
            if (got_pos) { gref[(((size_t)b * tH + hh) * tW + ww) * tn + n] = tgy[o]; gbr[n] += tgy[o]; }
        }
        double e_g = 0, e_gb = 0, s_gb = 0, e_gx = 0, s_gx = 0, e_gw = 0, s_gw = 0;
        for (size_t i = 0; i < nY; ++i) e_g = fmax(e_g, fabs((double)g_h[i] - gref[i]));      /* a scatter: exact */
        for (int n = 0; n < tn; ++n) { e_gb = fmax(e_gb, fabs(colsum_h[n] - gbr[n])); s_gb = fmax(s_gb, fabs(gbr[n])); }
        for (int b = 0; b < tB; ++b) for (int h = 0; h < tH; ++h) for (int w = 0; w < tW; ++w) for (int c = 0; c < bc; ++c) {
            double acc = 0;                            /* gx = sum_n,ky,kx g[h - ky + 1][w - kx + 1][n] W[n][c][ky][kx], where x > 0 */
            for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) {
                const int hh = h - ky + 1, ww = w - kx + 1;
                if (hh < 0 || hh >= tH || ww < 0 || ww >= tW) continue;
                const double* gp = gref + (((size_t)b * tH + hh) * tW + ww) * tn;
                for (int n = 0; n < tn; ++n) acc += gp[n] * wb[((size_t)n * bc + c) * 9 + ky * 3 + kx];
            }
            const size_t o = (((size_t)b * tH + h) * tW + w) * bc + c;
            if (!(xb[o] > 0)) acc = 0;
            e_gx = fmax(e_gx, fabs((double)gx_h[o] - acc)); s_gx = fmax(s_gx, fabs(acc));
        }
        for (int n = 0; n < tn; ++n) for (int c = 0; c < bc; ++c) for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) {
            double acc = 0;                            /* gw = sum_pix g[pix][n] x[pix + tap][c] */
            for (int b = 0; b < tB; ++b) for (int h = 0; h < tH; ++h) for (int w = 0; w < tW; ++w) {
                const int hh = h + ky - 1, ww = w + kx - 1;
                if (hh < 0 || hh >= tH || ww < 0 || ww >= tW) continue;
                acc += gref[(((size_t)b * tH + h) * tW + w) * tn + n] * xb[(((size_t)b * tH + hh) * tW + ww) * bc + c];
            }
            const size_t o = ((size_t)n * bc + c) * 9 + ky * 3 + kx;
            e_gw = fmax(e_gw, fabs((double)gw_h[o] - acc)); s_gw = fmax(s_gw, fabs(acc));
        }
        /* flat Adam (train.py:82-84, 224): three steps on 256 values, the second 64-element block frozen */
        enum { AN = 256 };
        float ap[AN], ag[AN], am[AN] = {0}, av[AN] = {0}, *dap, *dag, *dam, *dav;
        double rp[AN], rm[AN] = {0}, rv[AN] = {0};
        uint8_t act[AN / 64] = {1, 0, 1, 1}, *dact;
        for (int i = 0; i < AN; ++i) { ap[i] = frand(&seed); rp[i] = ap[i]; }
        CHECK(hipMalloc((void**)&dap, AN * 4)); CHECK(hipMalloc((void**)&dag, AN * 4)); CHECK(hipMalloc((void**)&dam, AN * 4));
        CHECK(hipMalloc((void**)&dav, AN * 4)); CHECK(hipMalloc((void**)&dact, AN / 64));
        CHECK(hipMemcpy(dap, ap, AN * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dam, am, AN * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dav, av, AN * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dact, act, AN / 64, hipMemcpyHostToDevice));
        const double lr = 1e-2, b1 = 0.9, b2 = 0.999, eps = 1e-8, wd = 1e-3;
        for (int step = 1; step <= 3; ++step) {
            for (int i = 0; i < AN; ++i) ag[i] = 0.1f * frand(&seed);
            CHECK(hipMemcpy(dag, ag, AN * 4, hipMemcpyHostToDevice));
            rc = dasv_adam_step_f32(dap, dag, dam, dav, dact, AN, (float)lr, (float)b1, (float)b2, (float)eps, (float)wd, step, 1, st);
            if (rc) { fprintf(stderr, "adam: %s\n", dasv_error_string(rc)); return 4; }
            for (int i = 0; i < AN; ++i) {
                if (!act[i / 64]) continue;
                const double gr = ag[i] + wd * rp[i];
                rm[i] = b1 * rm[i] + (1 - b1) * gr; rv[i] = b2 * rv[i] + (1 - b2) * gr * gr;
                rp[i] -= lr / (1 - pow(b1, step)) * rm[i] / (sqrt(rv[i]) / sqrt(1 - pow(b2, step)) + eps);
            }
        }
        CHECK(hipStreamSynchronize(st));
        CHECK(hipMemcpy(ap, dap, AN * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(ag, dag, AN * 4, hipMemcpyDeviceToHost));
        double e_ad = 0, g_left = 0;
        for (int i = 0; i < AN; ++i) { e_ad = fmax(e_ad, fabs(ap[i] - rp[i])); if (act[i / 64]) g_left = fmax(g_left, fabs(ag[i])); }
        printf("training: conv+pool max|err| %.3g of %.3g, arg-max bytes: %d wrong, %d near ties of %zu; pool backward max|err| %.3g; bias gradient "
               "%.3g of %.3g; data gradient %.3g of %.3g; weight gradient %.3g of %.3g; adam max|err| %.3g, gradient left %.3g\n",
               e_pl, s_pl, bad_idx, near_ties, nP, e_g, e_gb, s_gb, e_gx, s_gx, e_gw, s_gw, e_ad, g_left);
        ok = ok && e_pl < 4e-5 * s_pl && bad_idx == 0 && near_ties <= 4 && e_g == 0 && e_gb < 1e-5 * s_gb && e_gx < 5e-5 * s_gx &&
             e_gw < 8e-5 * s_gw && e_ad < 2e-6 && g_left == 0;
    }
    puts(ok ? "CABI_NATIVE_OK" : "CABI_NATIVE_FAIL");
    return ok ? 0 : 1;
}
