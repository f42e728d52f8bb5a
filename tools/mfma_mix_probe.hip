// Bare-MFMA probe for the conv3x3 kernel's per-tap matrix work (one wave tile = 64 pixels x 64 channels x 64 input channels):
//   A  split-bf16, what the kernel issues today: 16 tiles x 2 k-steps x 3 = 96 x v_mfma_f32_16x16x32_bf16
//   B  candidate: hi*hi on fp16 (32 x v_mfma_f32_16x16x32_f16) + BOTH cross terms of both k-steps on one block-scaled
//      fp8 MFMA per tile (16 x v_mfma_scale_f32_16x16x128_f8f6f4, e4m3 x e4m3)
//   C  single bf16 (32 MFMAs): the one-MFMA modes
// Operands live in registers (random bits, finite), 8 waves per workgroup (two per SIMD) on every CU, so what is measured
// is the matrix pipe under the clock the chip holds on random data — the ceiling the LDS-fed kernel can approach.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_mix_probe.hip -o tools/_diag/mfma_mix_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ void __launch_bounds__(512) probe(const unsigned* __restrict__ seed, float* __restrict__ out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    // operand fragments: 4 pixel blocks (hi, lo), 4 channel blocks (hi, lo), two k-steps share them (timing only)
    unsigned r[40];
#pragma unroll
    for (int i = 0; i < 40; ++i) r[i] = seed[(tid * 40 + i) & 0xfffff];
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 ah[4], al[4], bh[4], bl[4];
    i32x8 a8[4], b8[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // bf16 / fp16 bit patterns with exponent forced near 1.0 (0x3c00 / 0x3f80 range): finite, random mantissa + sign
        unsigned m0 = MODE == 1 ? 0x83ff83ffu : 0x807f807fu, e0 = MODE == 1 ? 0x38003800u : 0x3f003f00u;
        unsigned w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = (r[i * 4 + k] & m0) | e0;
        ah[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<uint4*>(w));
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = (r[16 + i * 4 + k] & m0) | e0;
        bh[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<uint4*>(w));
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = (r[(i * 4 + k + 7) % 40] & m0) | e0;
        al[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<uint4*>(w));
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = (r[(i * 4 + k + 23) % 40] & m0) | e0;
        bl[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<uint4*>(w));
        // e4m3 bytes: sign + 3 mantissa bits random, exponent field 0b0111 (1.0 .. 1.875)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            a8[i][k] = (int)((r[(i * 8 + k) % 40] & 0x87878787u) | 0x38383838u);
            b8[i][k] = (int)((r[(i * 8 + k + 11) % 40] & 0x87878787u) | 0x38383838u);
        }
    }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {          // A: 96 bf16 MFMAs
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah[i], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al[i], acc[i][j], 0, 0, 0);
                    }
        } else if (MODE == 1) {   // B: 32 fp16 + 16 scaled fp8 (K = 128)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bh[j]), __builtin_bit_cast(f16x8, ah[i]), acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b8[j], a8[i], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        } else if (MODE == 3 || MODE == 4) {   // D / E: as B with the cross terms on fp6 (e2m3) / fp4 operands — is the K = 128 MFMA faster there?
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bh[j]), __builtin_bit_cast(f16x8, ah[i]), acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = MODE == 3 ? __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b8[j], a8[i], acc[i][j], 2, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f)
                                          : __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b8[j], a8[i], acc[i][j], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        } else if (MODE == 5) {   // F: the scaled fp8 MFMAs alone (16 per tap)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b8[j], a8[i], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        } else if (MODE == 6) {   // G: the scaled fp6 MFMAs alone
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b8[j], a8[i], acc[i][j], 2, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        } else {                   // C: 32 bf16 MFMAs
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
        }
    }
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j];
    out[tid] = s[0] + s[1] + s[2] + s[3];
}

int main() {
    const int WG = 256, NT = 512, iters = 20000;
    std::vector<unsigned> h(1 << 20);
    srand(7);
    for (auto& v : h) v = ((unsigned)rand() << 16) ^ (unsigned)rand();
    unsigned* seed; CK(hipMalloc(&seed, h.size() * 4)); CK(hipMemcpy(seed, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    float* out; CK(hipMalloc(&out, WG * NT * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto kern, double flop_per_iter_per_wave) {
        std::vector<float> ms;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(WG), dim3(NT), 0, 0, seed, out, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        const double waves = (double)WG * NT / 64;
        // algorithmic work of one iteration of one wave = one tap of a 64 x 64 x 64 tile = 2*64*64*64 FLOP
        printf("%-46s %8.2f ms  %7.1f ns per tap per wave  executed %7.1f TFLOP/s  algorithmic %6.1f TFLOP/s\n", name, ms[1], ms[1] * 1e6 / iters,
               flop_per_iter_per_wave * waves * iters / ms[1] / 1e9, 2.0 * 64 * 64 * 64 * waves * iters / ms[1] / 1e9);
    };
    for (int round = 0; round < 2; ++round) {
        run("A split-bf16: 96 x 16x16x32 bf16", probe<0>, 96.0 * 2 * 16 * 16 * 32);
        run("B fp16 hi*hi + scaled fp8 cross: 32 + 16 (K=128)", probe<1>, 32.0 * 2 * 16 * 16 * 32 + 16.0 * 2 * 16 * 16 * 128);
        run("C single bf16: 32 x 16x16x32", probe<2>, 32.0 * 2 * 16 * 16 * 32);
        run("D fp16 hi*hi + scaled fp6 (e2m3) cross: 32 + 16", probe<3>, 32.0 * 2 * 16 * 16 * 32 + 16.0 * 2 * 16 * 16 * 128);
        run("E fp16 hi*hi + scaled fp4 cross: 32 + 16", probe<4>, 32.0 * 2 * 16 * 16 * 32 + 16.0 * 2 * 16 * 16 * 128);
        run("F 16 x scaled fp8 K=128 alone", probe<5>, 16.0 * 2 * 16 * 16 * 128);
        run("G 16 x scaled fp6 K=128 alone", probe<6>, 16.0 * 2 * 16 * 16 * 128);
    }
    CK(hipGetLastError());
    return 0;
}
