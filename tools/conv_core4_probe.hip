// Synthetic probe, companion of tools/conv_core_probe.hip: the same 256-pixel x 128-channel tile, the same LDS image and the same
// split-bf16 arithmetic (3 MFMAs per product), but carried by FOUR waves (one per SIMD, up to 512 registers each) of 128 pixels x 64
// channels instead of eight waves of 64 x 64 (two per SIMD, running in step).  One wave per SIMD has no partner to fill its
// stalls, so everything rests on the software pipeline inside the wave: a tap is 16 steps (2 halves x 8 pixel rows) of 12 MFMAs;
// the pixel fragments of step s+2 and a quarter of the next half's weight fragments are read at the top of step s.
//   W0  full: tile stores of the next tap (LDS only) + one barrier per tap
//   W1  no tile stores
//   W2  no tile stores, no barrier (reads + MFMAs)
//   W3  W0 with lean addressing of the pixel fragments (3 vector instructions per step instead of ~12); W4 = W3 without tile stores
// Ideal (matrix pipe only): 192 MFMAs x 16 cycles = 3072 cycles per tap = the 1432 ns of tools/mfma_mix_probe.
//   hipcc -O3 --offload-arch=gfx950 tools/conv_core4_probe.hip -o tools/_diag/conv_core4_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int lds_off(int row, int slot) { return row * 128 + ((slot ^ (row & 7)) << 4); }

template <int V>
__global__ void __launch_bounds__(256) core4(const unsigned* __restrict__ seed, float* __restrict__ out, int slabs, int W) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wrows = 256 + 2 * W + 2;
    char* Ahi = smem;                                  // wrows rows + one zero row
    const int D = (wrows + 1) * 128;
    char* Alo = Ahi + D;
    char* Bbase = Alo + D;   // 2 x { B_hi 16 KB | B_lo 16 KB }
    char* Zrow = Ahi + wrows * 128;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1, frow = lane & 15, fq = lane >> 4;
    const int total = (2 * D + 65536) / 4;
    for (int i = tid; i < total; i += 256) reinterpret_cast<unsigned*>(smem)[i] = (seed[(blockIdx.x * 977 + i) & 0xfffff] & 0x807f807fu) | 0x3f003f00u;
    __syncthreads();
    if (tid < 32) { reinterpret_cast<unsigned*>(Zrow)[tid] = 0u; reinterpret_cast<unsigned*>(Zrow + D)[tid] = 0u; }
    unsigned amask[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) amask[i] = 0x1ff ^ ((seed[(tid + i) & 0xfffff] & 7u) == 0 ? 1u << (tid % 9) : 0u);
    u32x4 rbh[4], rbl[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { rbh[k] = u32x4{0x3f803f80u, 0x3f003f00u, 0xbf803f00u, 0x3f80bf00u}; rbl[k] = rbh[k]; }
    __syncthreads();

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 bh[2][4], bl[2][4];   // weight fragments of a half, double-buffered by half parity
    bf16x8 ah[4], al[4];         // pixel fragments: ring of four steps
    const int bo[2] = {lds_off(wn * 64 + frow, fq), lds_off(wn * 64 + frow, 4 + fq)};
    auto rd_b = [&](int buf, const char* Bhi, int kk, int j) {
        bh[buf][j] = *reinterpret_cast<const bf16x8*>(Bhi + bo[kk] + j * 2048);
        bl[buf][j] = *reinterpret_cast<const bf16x8*>(Bhi + 16384 + bo[kk] + j * 2048);
    };
    // lean form (V >= 3): the tap's two swizzled bases once per tap; per step one select between the base and (zero row - i * 2048),
    // the row step i * 2048 and the plane step D as instruction offsets
    int ab[2];
    auto tap_bases = [&](int tap) {
        const int arow = wm * 128 + frow + (W + 1) + (tap / 3 - 1) * W + (tap % 3 - 1);
        ab[0] = lds_off(arow, fq);
        ab[1] = lds_off(arow, 4 + fq);
    };
    const int zrel = wrows * 128;
    auto rd_a_lean = [&](int r, unsigned tapbit, int kk, int i) {
        const int o = (amask[i] & tapbit) ? ab[kk] : zrel - i * 2048;
        ah[r] = *reinterpret_cast<const bf16x8*>(Ahi + o + i * 2048);
        al[r] = *reinterpret_cast<const bf16x8*>(Ahi + o + i * 2048 + D);
    };
    auto rd_a = [&](int r, int tap, int kk, int i) {
        const int arow = wm * 128 + frow + (W + 1) + (tap / 3 - 1) * W + (tap % 3 - 1);
        const int o = lds_off(arow, kk * 4 + fq) + i * 2048;
        const bool ok = (amask[i] >> tap) & 1u;
        ah[r] = *reinterpret_cast<const bf16x8*>(ok ? Ahi + o : Zrow);
        al[r] = *reinterpret_cast<const bf16x8*>(ok ? Alo + o : Zrow);
    };
    auto store_b = [&](char* Bhi, int k) {
        const int q = tid + 256 * k;
        const int o = lds_off(q >> 3, q & 7);
        *reinterpret_cast<u32x4*>(Bhi + o) = rbh[k];
        *reinterpret_cast<u32x4*>(Bhi + 16384 + o) = rbl[k];
    };
    auto mma_step = [&](int buf, int r, int i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[buf][j], ah[r], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[buf][j], ah[r], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[buf][j], al[r], acc[i][j], 0, 0, 0);
    };

    {
#pragma unroll
        for (int k = 0; k < 4; ++k) store_b(Bbase, k);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) rd_b(0, Bbase, 0, j);
        rd_a(0, 0, 0, 0);
        rd_a(1, 0, 0, 1);
    }
    const int ntaps = slabs * 9;
    int tap = 0;
    for (int it = 0; it < ntaps; ++it) {
        const int tapn = tap == 8 ? 0 : tap + 1;
        char* Bhi = Bbase + ((it & 1) ? 32768 : 0);
        char* Bnx = Bbase + ((it & 1) ? 0 : 32768);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int kk = s >> 3, i = s & 7, s2 = s + 2;
            if constexpr (V >= 3) {
                if (s == 0) tap_bases(tap);          // steps 14 / 15 of the previous tap read this tap's rows 0 / 1 with the bases set below
                if (s == 14) tap_bases(tapn);
                if (s2 < 16) rd_a_lean(s2 & 3, 1u << tap, s2 >> 3, s2 & 7);
                else rd_a_lean(s2 & 3, 1u << tapn, 0, s2 - 16);
            } else {
                if (s2 < 16) rd_a(s2 & 3, tap, s2 >> 3, s2 & 7);
                else rd_a(s2 & 3, tapn, 0, s2 - 16);
            }
            if (kk == 0 && i < 4) rd_b(1, Bhi, 1, i);
            if (kk == 0 && i >= 4 && (V == 0 || V == 3)) store_b(Bnx, i - 4);
            if (kk == 1 && i == 4 && V != 2) __syncthreads();
            if (kk == 1 && i >= 4) rd_b(0, Bnx, 0, i - 4);
            mma_step(kk, s & 3, i);
            __builtin_amdgcn_sched_barrier(0);
        }
        tap = tapn;
    }
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + tid] = s[0] + s[1] + s[2] + s[3];
}

int main(int argc, char** argv) {
    const int W = argc > 1 ? atoi(argv[1]) : 10;
    const int WG = 256, slabs = 400;
    std::vector<unsigned> h(1 << 20);
    srand(11);
    for (auto& v : h) v = ((unsigned)rand() << 16) ^ (unsigned)rand();
    unsigned* seed; CK(hipMalloc(&seed, h.size() * 4)); CK(hipMemcpy(seed, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    float* out; CK(hipMalloc(&out, WG * 512 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t lds = (size_t)(256 + 2 * W + 3) * 256 + 65536;
    auto run = [&](const char* name, auto kern) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        std::vector<float> ms;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(WG), dim3(256), lds, 0, seed, out, slabs, W);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        const double taps = slabs * 9.0;
        printf("%-60s %8.2f ms  %7.1f ns per tap  executed %7.1f TFLOP/s\n", name, ms[1], ms[1] * 1e6 / taps,
               192.0 * 2 * 16 * 16 * 32 * 4 * WG * taps / ms[1] / 1e9);
    };
    printf("W = %d, LDS %zu bytes\n", W, lds);
    for (int round = 0; round < 2; ++round) {
        run("W0 four waves of 128x64: stores + barrier", core4<0>);
        run("W1 no tile stores", core4<1>);
        run("W2 no tile stores, no barrier", core4<2>);
        run("W3 W0 with lean fragment addressing", core4<3>);
        run("W4 W3 without tile stores", core4<4>);
    }
    CK(hipGetLastError());
    return 0;
}
