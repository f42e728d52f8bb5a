/*
 * Plain-C restatement of the reference's DoubleMHA pooling (TEST INFRASTRUCTURE ONLY — see oracle/__init__.py).
 * Scalar, single-threaded, fp32 storage with double accumulation, no shortcuts: a second, independent
 * implementation of the maths in oracle/dmha_oracle.py, pinned to the same golden vectors
 * (tests/test_oracle_golden.py::test_c_restatement_*).
 *
 * Follows /root/reference/scripts/poolings.py:
 *   :73-80   scores = diag(key @ query) / sqrt(query.size(-1) = H); softmax over frames; weighted sum
 *   :45-51   head scores = ct @ att; optional -inf mask; softmax over heads
 *   :61-71, :126-129  out = sum_h w_h * ct_h
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

/* x [B,T,H*dh], query [dh,H], att [dh], keep [B,H] or NULL  ->  out [B,dh], align [B,T,H], head_w [B,H], ctx [B,H,dh] */
int dmha_oracle_forward(const float* x, const float* query, const float* att, const uint8_t* keep,
                        float* out, float* align, float* head_w, float* ctx,
                        int B, int T, int H, int dh) {
    const int D = H * dh;
    const double scale = 1.0 / sqrt((double)H); /* poolings.py:75 — heads, not head size */
    double* s = (double*)malloc(sizeof(double) * (size_t)T);
    double* u = (double*)malloc(sizeof(double) * (size_t)H);
    if (!s || !u) { free(s); free(u); return -1; }
    for (int b = 0; b < B; ++b) {
        for (int h = 0; h < H; ++h) {
            double mx = -INFINITY;
            for (int t = 0; t < T; ++t) {
                const float* row = x + ((size_t)b * T + t) * D + (size_t)h * dh;
                double acc = 0.0;
                for (int d = 0; d < dh; ++d) acc += (double)row[d] * (double)query[(size_t)d * H + h];
                s[t] = acc * scale;
                if (s[t] > mx) mx = s[t];
            }
            double den = 0.0;
            for (int t = 0; t < T; ++t) { s[t] = exp(s[t] - mx); den += s[t]; }
            for (int t = 0; t < T; ++t) {
                s[t] /= den;
                align[((size_t)b * T + t) * H + h] = (float)s[t];
            }
            for (int d = 0; d < dh; ++d) {
                double acc = 0.0;
                for (int t = 0; t < T; ++t) acc += s[t] * (double)x[((size_t)b * T + t) * D + (size_t)h * dh + d];
                ctx[((size_t)b * H + h) * dh + d] = (float)acc;
            }
        }
        double mx = -INFINITY;
        for (int h = 0; h < H; ++h) {
            double acc = 0.0;
            for (int d = 0; d < dh; ++d) acc += (double)ctx[((size_t)b * H + h) * dh + d] * (double)att[d];
            if (keep && !keep[(size_t)b * H + h]) acc = -INFINITY;
            u[h] = acc;
            if (acc > mx) mx = acc;
        }
        double den = 0.0;
        for (int h = 0; h < H; ++h) { u[h] = exp(u[h] - mx); den += u[h]; } /* all masked: NaN, as the reference */
        for (int h = 0; h < H; ++h) { u[h] /= den; head_w[(size_t)b * H + h] = (float)u[h]; }
        for (int d = 0; d < dh; ++d) {
            double acc = 0.0;
            for (int h = 0; h < H; ++h) acc += u[h] * (double)ctx[((size_t)b * H + h) * dh + d];
            out[(size_t)b * dh + d] = (float)acc;
        }
    }
    free(s);
    free(u);
    return 0;
}
