// Exact-data probe of v_mfma_scale_f32_16x16x128_f8f6f4 with fp6 (e2m3) operands on gfx950 (cbsz = blgp = 2), groundwork for
// the block-scaled fp6 cross terms of DESIGN.md section 8 item 3 (no ISA manual in this image; companion of fp8_mfma_check.hip).
// A lane supplies six operand registers = 32 six-bit fields.  Questions:
//   E0  all fields 1.0 (code 0x08), all scales 127: D = 128 everywhere?
//   E1  are the fields packed little-endian, field f at bits [6f, 6f+5] of the 192-bit operand?  (one field = 2.0 -> its row + 1)
//   E2  whose products does the scale byte of lane (row r, lane group g) multiply?
//   E3  which fields (lane group, register half) belong to the block that lane group g's scale byte scales?
//   hipcc -O3 --offload-arch=gfx950 tools/fp6_mfma_check.hip -o tools/_diag/fp6_mfma_check
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

__global__ void mfma6_kernel(const unsigned* A, const unsigned* B, const unsigned char* sa, const unsigned char* sb, float* D) {
    const int l = threadIdx.x;
    i32x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int w = 0; w < 6; ++w) { a[w] = (int)A[l * 6 + w]; b[w] = (int)B[l * 6 + w]; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 2, 2, 0, (int)sa[l], 0, (int)sb[l]);
    for (int q = 0; q < 4; ++q) D[((l >> 4) * 4 + q) * 16 + (l & 15)] = c[q];   // row = (l>>4)*4+q, col = l&15
}

static void set_field(std::vector<unsigned>& v, int lane, int f, unsigned code) {
    const int bit = 6 * f, w = bit / 32, sh = bit % 32;
    unsigned long long cur = v[lane * 6 + w];
    if (w + 1 < 6) cur |= (unsigned long long)v[lane * 6 + w + 1] << 32;
    cur = (cur & ~(0x3full << sh)) | ((unsigned long long)code << sh);
    v[lane * 6 + w] = (unsigned)cur;
    if (w + 1 < 6) v[lane * 6 + w + 1] = (unsigned)(cur >> 32);
}

int main() {
    unsigned *dA, *dB; unsigned char *dsa, *dsb; float* dD;
    CK(hipMalloc(&dA, 64 * 6 * 4)); CK(hipMalloc(&dB, 64 * 6 * 4)); CK(hipMalloc(&dsa, 64)); CK(hipMalloc(&dsb, 64)); CK(hipMalloc(&dD, 256 * 4));
    std::vector<float> D(256);
    auto run = [&](const std::vector<unsigned>& A, const std::vector<unsigned>& B, const std::vector<unsigned char>& sa, const std::vector<unsigned char>& sb) {
        CK(hipMemcpy(dA, A.data(), 64 * 6 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 64 * 6 * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dsa, sa.data(), 64, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb.data(), 64, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(mfma6_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
        CK(hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost));
    };
    std::vector<unsigned> ones(64 * 6, 0);
    for (int l = 0; l < 64; ++l) for (int f = 0; f < 32; ++f) set_field(ones, l, f, 0x08);
    std::vector<unsigned char> s127(64, 127);
    // E0
    run(ones, ones, s127, s127);
    { float mn = 1e9f, mx = -1e9f; for (float v : D) { mn = v < mn ? v : mn; mx = v > mx ? v : mx; } printf("E0 all ones: D in [%g, %g] (expect 128)\n", mn, mx); }
    // E1: one field of A = 2.0
    for (int trial = 0; trial < 6; ++trial) {
        const int lane = 5 + 16 * (trial & 3), f = (trial * 7 + 5) % 32;
        std::vector<unsigned> A = ones; set_field(A, lane, f, 0x10);
        run(A, ones, s127, s127);
        printf("E1 A lane %2d (row %d, group %d) field %2d = 2.0: rows != 128:", lane, lane & 15, lane >> 4, f);
        for (int i = 0; i < 16; ++i) if (D[i * 16] != 128.f) printf(" row %d -> %g", i, D[i * 16]);
        printf("\n");
    }
    // E1b: code table: field value v for every 6-bit code (row 0, lane 0, field 0; rest zero except B ones)
    {
        printf("E1b codes:");
        for (unsigned code = 0; code < 64; ++code) {
            std::vector<unsigned> A(64 * 6, 0); set_field(A, 0, 0, code);
            run(A, ones, s127, s127);
            printf(" %02x=%g", code, D[0]);
        }
        printf("\n");
    }
    // E2: scale byte of one lane of A
    for (int g = 0; g < 4; ++g) {
        std::vector<unsigned char> sa = s127; sa[7 + 16 * g] = 128;
        run(ones, ones, sa, s127);
        printf("E2 sa(lane %2d: row 7, group %d) = 128: rows != 128:", 7 + 16 * g, g);
        for (int i = 0; i < 16; ++i) if (D[i * 16] != 128.f) printf(" row %d -> %g", i, D[i * 16]);
        printf("\n");
    }
    // E3: which (lane group, register half) of row 7 sit in the block scaled by group g's byte: zero 16 fields of A, scale x2
    for (int g = 0; g < 4; ++g) {
        std::vector<unsigned char> sa = s127; sa[7 + 16 * g] = 128;
        printf("E3 sa group %d x2; zeroing (group, half) -> D[7][0]:", g);
        for (int g2 = 0; g2 < 4; ++g2)
            for (int h = 0; h < 2; ++h) {
                std::vector<unsigned> A = ones;
                for (int f = 16 * h; f < 16 * h + 16; ++f) set_field(A, 7 + 16 * g2, f, 0);
                run(A, ones, sa, s127);
                printf(" (%d,%d) %g", g2, h, D[7 * 16]);
            }
        printf("\n");
    }
    // E4: the same for B's scale and fields (column 9)
    for (int g = 0; g < 4; ++g) {
        std::vector<unsigned char> sb = s127; sb[9 + 16 * g] = 129;
        printf("E4 sb group %d x4; zeroing B (group, half) -> D[0][9]:", g);
        for (int g2 = 0; g2 < 4; ++g2)
            for (int h = 0; h < 2; ++h) {
                std::vector<unsigned> B = ones;
                for (int f = 16 * h; f < 16 * h + 16; ++f) set_field(B, 9 + 16 * g2, f, 0);
                run(ones, B, s127, sb);
                printf(" (%d,%d) %g", g2, h, D[9]);
            }
        printf("\n");
    }
    // E5: k pairing between A and B: A one field = 2.0 at (group ga, field fa); B zero everywhere except (group gb, field fb) = 1:
    // D[row][col] = 2 only when the two positions are the same k.  Scan B positions for a few A positions.
    for (int trial = 0; trial < 4; ++trial) {
        const int ga = trial, fa = (11 * trial + 3) % 32;
        std::vector<unsigned> A(64 * 6, 0); set_field(A, 2 + 16 * ga, fa, 0x10);
        printf("E5 A(group %d, field %2d) pairs with B", ga, fa);
        for (int gb = 0; gb < 4; ++gb)
            for (int fb = 0; fb < 32; ++fb) {
                std::vector<unsigned> B(64 * 6, 0); set_field(B, 4 + 16 * gb, fb, 0x08);
                run(A, B, s127, s127);
                if (D[2 * 16 + 4] != 0.f) printf(" (group %d, field %2d: %g)", gb, fb, D[2 * 16 + 4]);
            }
        printf("\n");
    }
    CK(hipGetLastError());
    return 0;
}
