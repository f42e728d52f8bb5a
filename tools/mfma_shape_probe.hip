// MFMA SHAPE probe for the split8 tap (cdna_hip_programming.md rule 28 / MI355X_MICROARCH.md give-back item 7):
// the same wave tile — 64 pixels x 64 output channels x 64 input channels per tap, fp16 hi*hi products + block-scaled e4m3 cross
// terms — issued on the 16x16 forms the kernel uses today and on the 32x32 forms, operands in registers, on random data, with the
// in-kernel clock stamped.  Cycles per tap are equal by construction (16x16: 32 x 16 + 16 x 32; 32x32: 16 x 32 + 8 x 64 = 1024 pipe
// cycles); what can differ is the clock the chip holds on each shape.
//   S16  32 x v_mfma_f32_16x16x32_f16 + 16 x v_mfma_scale_f32_16x16x128_f8f6f4          (csrc/conv3x3.hip, PREC_X8)
//   S32  16 x v_mfma_f32_32x32x16_f16 +  8 x v_mfma_scale_f32_32x32x64_f8f6f4
// One and two waves per SIMD, every CU busy.   hipcc -O3 --offload-arch=gfx950 tools/mfma_shape_probe.hip -o tools/_diag/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int SHAPE, int NT>
__global__ void __launch_bounds__(NT) probe(const unsigned* __restrict__ seed, float* __restrict__ out, unsigned long long* __restrict__ clk,
                                            int iters, int zeros) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned r[40];
#pragma unroll
    for (int i = 0; i < 40; ++i) r[i] = zeros ? 0u : seed[(tid * 40 + i) & 0xfffff];
    f16x8 ah[4], bh[4];
    i32x8 a8[4], b8[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = zeros ? 0u : ((r[i * 4 + k] & 0x83ff83ffu) | 0x38003800u);     // fp16 near 1.0, random sign / mantissa
        ah[i] = __builtin_bit_cast(f16x8, *reinterpret_cast<uint4*>(w));
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = zeros ? 0u : ((r[16 + i * 4 + k] & 0x83ff83ffu) | 0x38003800u);
        bh[i] = __builtin_bit_cast(f16x8, *reinterpret_cast<uint4*>(w));
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            a8[i][k] = zeros ? 0 : (int)((r[(i * 8 + k) % 40] & 0x87878787u) | 0x38383838u);                // e4m3 1.0 .. 1.875, random sign
            b8[i][k] = zeros ? 0 : (int)((r[(i * 8 + k + 11) % 40] & 0x87878787u) | 0x38383838u);
        }
    }
    f32x4 acc[4][4];
    f32x16 big[2][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) big[i][j][k] = 0.f;
    const unsigned long long c0 = __builtin_readcyclecounter(), t0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (SHAPE == 16) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[(i + kk) & 3], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b8[j], a8[i], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        } else {
            // 2 x 2 tiles of 32 x 32; K = 64 as four k-steps of 16 (fp16) and the 128-deep cross-term sum as two K = 64 scaled MFMAs
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        big[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[(2 * j + kk) & 3], ah[(2 * i + kk) & 3], big[i][j], 0, 0, 0);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        big[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b8[2 * j + kk], a8[2 * i + kk], big[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), t1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) s += big[i][j][k];
    out[tid] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = t1 - t0; }
}

int main() {
    const int WG = 256, iters = 40000;
    std::vector<unsigned> h(1 << 20);
    srand(7);
    for (auto& v : h) v = ((unsigned)rand() << 16) ^ (unsigned)rand();
    unsigned* seed; CK(hipMalloc(&seed, h.size() * 4)); CK(hipMemcpy(seed, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    float* out; CK(hipMalloc(&out, WG * 512 * 4));
    unsigned long long* clk; CK(hipMalloc(&clk, WG * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto kern, int nt, int zeros) {
        std::vector<float> ms;
        std::vector<unsigned long long> hc(WG * 2);
        double ghz = 0;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(WG), dim3(nt), 0, 0, seed, out, clk, iters, zeros);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
            CK(hipMemcpy(hc.data(), clk, WG * 16, hipMemcpyDeviceToHost));
            std::vector<double> g;
            for (int b = 0; b < WG; ++b) g.push_back((double)hc[2 * b] / (double)hc[2 * b + 1] * 0.1);   // cycles per 10 ns tick -> GHz
            std::sort(g.begin(), g.end());
            ghz = g[WG / 2];
        }
        std::sort(ms.begin(), ms.end());
        const double waves = (double)WG * nt / 64;
        printf("%-58s %8.2f ms  %7.1f ns per tap per wave  algorithmic %6.1f TFLOP/s  in-kernel clock %.2f GHz\n", name, ms[2], ms[2] * 1e6 / iters,
               2.0 * 64 * 64 * 64 * waves * iters / ms[2] / 1e9, ghz);
    };
    for (int round = 0; round < 3; ++round) {
        run("S16 2 waves/SIMD  random: 32 x 16x16x32 f16 + 16 x scaled 16x16x128", probe<16, 512>, 512, 0);
        run("S32 2 waves/SIMD  random: 16 x 32x32x16 f16 +  8 x scaled 32x32x64", probe<32, 512>, 512, 0);
        run("S16 1 wave/SIMD   random", probe<16, 256>, 256, 0);
        run("S32 1 wave/SIMD   random", probe<32, 256>, 256, 0);
        run("S16 2 waves/SIMD  zeros", probe<16, 512>, 512, 1);
        run("S32 2 waves/SIMD  zeros", probe<32, 512>, 512, 1);
    }
    CK(hipGetLastError());
    return 0;
}
