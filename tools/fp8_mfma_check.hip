// Exact-data check of what the fp16 + fp8-cross-term conv mode relies on (no ISA manual in this image):
//  1. v_cvt_pk_fp8_f32 (__builtin_amdgcn_cvt_pk_fp8_f32) on gfx950: OCP e4m3fn codes, rounding, behaviour beyond +-448
//  2. v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 x e4m3: lane l holds row l & 15, k = 32 * (l >> 4) + byte index of its
//     eight operand registers; the scale operand's byte 0 (op_sel 0) of lane l scales THAT lane's 32-element block by 2^(s-127)
//  3. v_mfma_f32_16x16x32_f16 operand map (as the bf16 form)
// Every value is a small dyadic number, so the fp32 results must match the host sum EXACTLY.
//   hipcc -O3 --offload-arch=gfx950 tools/fp8_mfma_check.hip -o tools/_diag/fp8_mfma_check
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void cvt_kernel(const float* in, unsigned* out, int n) {
    const int i = threadIdx.x;
    if (i < n) out[i] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(in[i], 0.f, 0, false) & 0xffu;
}

__global__ void mfma8_kernel(const unsigned char* A, const unsigned char* B, const unsigned char* sa, const unsigned char* sb, float* D) {
    const int l = threadIdx.x, r = l & 15, kb = l >> 4;
    i32x8 a, b;
    for (int w = 0; w < 8; ++w) {
        unsigned va = 0, vb = 0;
        for (int q = 0; q < 4; ++q) {
            va |= (unsigned)A[r * 128 + kb * 32 + w * 4 + q] << (8 * q);
            vb |= (unsigned)B[r * 128 + kb * 32 + w * 4 + q] << (8 * q);
        }
        a[w] = (int)va; b[w] = (int)vb;
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    // D[i][j] = sum_k A[i][k] B[j][k] with A as src0 rows, B as src1 columns
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, (int)sa[r * 4 + kb], 0, (int)sb[r * 4 + kb]);
    for (int q = 0; q < 4; ++q) D[((l >> 4) * 4 + q) * 16 + (l & 15)] = c[q];   // row = (l>>4)*4+q, col = l&15
}

__global__ void mfma16_kernel(const _Float16* A, const _Float16* B, float* D) {
    const int l = threadIdx.x, r = l & 15, kb = l >> 4;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = A[r * 32 + kb * 8 + j]; b[j] = B[r * 32 + kb * 8 + j]; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int q = 0; q < 4; ++q) D[((l >> 4) * 4 + q) * 16 + (l & 15)] = c[q];
}

static float e4m3_to_float(unsigned char v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f;
    if (e == 15 && m == 7) return NAN;
    if (e == 0) f = ldexpf((float)m, -9); else f = ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -f : f;
}

int main() {
    // ---- 1. conversions
    const float tests[] = {0.f, 1.f, -1.f, 1.0625f, 1.1875f, 1.125f, 0.0625f, 1.0f / 64, 1.0f / 128, 1.0f / 512, 1.0f / 1024, 3.0f / 1024,
                           416.f, 448.f, 464.f, 480.f, 500.f, 1e6f, -1e6f, INFINITY, 0.3f, 17.3f, 240.f, 247.9f, 248.1f};
    const int n = sizeof(tests) / sizeof(float);
    float* din; unsigned* dout; CK(hipMalloc(&din, n * 4)); CK(hipMalloc(&dout, n * 4));
    CK(hipMemcpy(din, tests, n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(cvt_kernel, dim3(1), dim3(64), 0, 0, din, dout, n);
    std::vector<unsigned> codes(n); CK(hipMemcpy(codes.data(), dout, n * 4, hipMemcpyDeviceToHost));
    printf("v_cvt_pk_fp8_f32:\n");
    for (int i = 0; i < n; ++i) printf("  %12g -> 0x%02x = %g\n", tests[i], codes[i], e4m3_to_float((unsigned char)codes[i]));
    // ---- 2. scaled fp8 MFMA
    std::vector<unsigned char> A(16 * 128), B(16 * 128), sa(64), sb(64);
    srand(3);
    const unsigned char vals[] = {0x00, 0x38, 0xb8, 0x40, 0x30, 0x3c, 0xc0, 0x28, 0x44, 0xbc};   // 0, 1, -1, 2, .5, 1.5, -2, .25, 3, -1.5
    for (auto& v : A) v = vals[rand() % 10];
    for (auto& v : B) v = vals[rand() % 10];
    for (int i = 0; i < 64; ++i) { sa[i] = 127 - (rand() % 5); sb[i] = 127 + (rand() % 4) - 6; }
    unsigned char *dA, *dB, *dsa, *dsb; float* dD;
    CK(hipMalloc(&dA, A.size())); CK(hipMalloc(&dB, B.size())); CK(hipMalloc(&dsa, 64)); CK(hipMalloc(&dsb, 64)); CK(hipMalloc(&dD, 256 * 4));
    CK(hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dsa, sa.data(), 64, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb.data(), 64, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mfma8_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
    std::vector<float> D(256); CK(hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double ref = 0;
            for (int kb = 0; kb < 4; ++kb) {
                double s = 0;
                for (int k = 0; k < 32; ++k) s += (double)e4m3_to_float(A[i * 128 + kb * 32 + k]) * e4m3_to_float(B[j * 128 + kb * 32 + k]);
                ref += ldexp(s, (int)sa[i * 4 + kb] - 127 + (int)sb[j * 4 + kb] - 127);
            }
            worst = fmax(worst, fabs(ref - D[i * 16 + j]));
        }
    printf("scaled fp8 16x16x128 (lane = row, 32-k block; per-lane scale byte 0): max |gpu - host| = %g  (D[0][0..3] = %g %g %g %g)\n", worst, D[0], D[1], D[2], D[3]);
    // which 32 products does the scale byte of lane (row, kb) apply to?  position (kb, w, j) = lane group, register pair, byte in pair
    for (int hyp = 1; hyp <= 4; ++hyp) {
        double wh = 0;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double ref = 0;
                for (int kb = 0; kb < 4; ++kb)
                    for (int w = 0; w < 4; ++w) {
                        int blk = kb;
                        if (hyp == 2) blk = w;
                        if (hyp == 3) blk = 2 * (w >> 1) + (kb >> 1);
                        if (hyp == 4) blk = 2 * (kb >> 1) + (w >> 1);
                        double sum = 0;
                        for (int q = 0; q < 8; ++q) sum += (double)e4m3_to_float(A[i * 128 + kb * 32 + w * 8 + q]) * e4m3_to_float(B[j * 128 + kb * 32 + w * 8 + q]);
                        ref += ldexp(sum, (int)sa[i * 4 + blk] - 127 + (int)sb[j * 4 + blk] - 127);
                    }
                wh = fmax(wh, fabs(ref - D[i * 16 + j]));
            }
        printf("  hypothesis %d (block of (lane group kb, register pair w) = %s): max |gpu - host| = %g\n", hyp,
               hyp == 1 ? "kb" : hyp == 2 ? "w" : hyp == 3 ? "2*(w>>1)+(kb>>1)" : "2*(kb>>1)+(w>>1)", wh);
    }
    // ---- 2b. the same with every scale = 127 (1.0): isolates the operand layout from the scale semantics
    {
        std::vector<unsigned char> one(64, 127);
        CK(hipMemcpy(dsa, one.data(), 64, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, one.data(), 64, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(mfma8_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
        CK(hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost));
        double w2 = 0;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double ref = 0;
                for (int k = 0; k < 128; ++k) ref += (double)e4m3_to_float(A[i * 128 + k]) * e4m3_to_float(B[j * 128 + k]);
                w2 = fmax(w2, fabs(ref - D[i * 16 + j]));
            }
        printf("  all scales 127: max |gpu - host| = %g\n", w2);
        // ---- 2c. which products does ONE lane's scale byte touch?  A = B = all ones (D = 128 everywhere); set sa of one
        // (row r0, block kb0) lane to 128 (x2): under the assumed semantics only row r0 changes, to 128 + 32 = 160
        std::vector<unsigned char> ones(16 * 128, 0x38);
        CK(hipMemcpy(dA, ones.data(), ones.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dB, ones.data(), ones.size(), hipMemcpyHostToDevice));
        for (int trial = 0; trial < 3; ++trial) {
            const int r0 = 3 + trial * 4, kb0 = trial + 1;
            std::vector<unsigned char> s2(64, 127);
            s2[r0 * 4 + kb0] = 128;
            CK(hipMemcpy(dsa, s2.data(), 64, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(mfma8_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
            CK(hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost));
            printf("  sa(row %d, block %d) = 128: rows with D != 128:", r0, kb0);
            for (int i = 0; i < 16; ++i) if (D[i * 16] != 128.f) printf(" row %d -> %g", i, D[i * 16]);
            printf("   | cols of row 0:");
            for (int j = 0; j < 16; ++j) if (D[j] != 128.f) printf(" col %d -> %g", j, D[j]);
            printf("\n");
        }
        // the same for sb
        std::vector<unsigned char> one2(64, 127);
        CK(hipMemcpy(dsa, one2.data(), 64, hipMemcpyHostToDevice));
        for (int trial = 0; trial < 2; ++trial) {
            const int r0 = 5 + trial * 6, kb0 = trial * 3;
            std::vector<unsigned char> s2(64, 127);
            s2[r0 * 4 + kb0] = 129;
            CK(hipMemcpy(dsb, s2.data(), 64, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(mfma8_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
            CK(hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost));
            printf("  sb(row %d, block %d) = 129: cols of row 0 with D != 128:", r0, kb0);
            for (int j = 0; j < 16; ++j) if (D[j] != 128.f) printf(" col %d -> %g", j, D[j]);
            printf("\n");
        }
    }
    // ---- 3. fp16 16x16x32
    std::vector<_Float16> A16(16 * 32), B16(16 * 32);
    for (auto& v : A16) v = (_Float16)((rand() % 17 - 8) * 0.25f);
    for (auto& v : B16) v = (_Float16)((rand() % 17 - 8) * 0.5f);
    _Float16 *dA16, *dB16; CK(hipMalloc(&dA16, 1024)); CK(hipMalloc(&dB16, 1024));
    CK(hipMemcpy(dA16, A16.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB16, B16.data(), 1024, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mfma16_kernel, dim3(1), dim3(64), 0, 0, dA16, dB16, dD);
    CK(hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost));
    worst = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double ref = 0;
            for (int k = 0; k < 32; ++k) ref += (double)(float)A16[i * 32 + k] * (float)B16[j * 32 + k];
            worst = fmax(worst, fabs(ref - D[i * 16 + j]));
        }
    printf("fp16 16x16x32: max |gpu - host| = %g\n", worst);
    CK(hipGetLastError());
    return 0;
}
