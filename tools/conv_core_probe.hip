// Synthetic probe of the conv3x3 window kernel's CORE loop: the same workgroup (8 waves as 4 x 2, each 64 pixels x 64
// channels), the same XOR-swizzled LDS image (A window rows + tap shift, B tile), one barrier per tap, the tile write of the
// next tap (LDS stores only, no global traffic), split-bf16 = 3 MFMAs per product.  No staging from memory: this isolates how
// the fragment reads, the MFMAs and the barrier of two waves per SIMD overlap, for different orderings of the same work:
//   V0  today's order: per 32-channel half all 16 fragment reads, then the selects and 48 MFMAs (reads-first hint)
//   V1  fragments double-buffered in registers: the reads of half h+1 are issued before the MFMAs of half h (also across
//       the tap's barrier for the pixel fragments; weight fragments of the next tap's first half are read after the barrier)
//   V2  V1 with the 16 reads of the next half issued in four groups of four between the MFMA rows of the current half
//   V3  V0 without the tile stores (what the stores + their barrier cost)
// Reference: tools/mfma_mix_probe (registers only) needs 1432 ns per tap per wave for the same 96 MFMAs x 2 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/conv_core_probe.hip -o tools/_diag/conv_core_probe
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

struct Frags { bf16x8 bh[4], bl[4], ah[4], al[4]; };

// ablations on top of V2: V4 no per-tap barrier, V5 no border selects, V6 no tile stores, V7 all three off (reads + MFMAs only)
template <int V>
__global__ void __launch_bounds__(512) core(const unsigned* __restrict__ seed, float* __restrict__ out, int slabs, int W) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wrows = 256 + 2 * W + 2;
    char* Ahi = smem;
    char* Alo = Ahi + (size_t)wrows * 128;
    char* Bbase = Alo + (size_t)wrows * 128;   // 2 x { B_hi 16 KB | B_lo 16 KB }
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1, frow = lane & 15, fq = lane >> 4;
    // fill LDS with finite random bf16 (exponent near 1.0)
    const int total = (wrows * 256 + 65536) / 4;
    for (int i = tid; i < total; i += 512) reinterpret_cast<unsigned*>(smem)[i] = (seed[(blockIdx.x * 977 + i) & 0xfffff] & 0x807f807fu) | 0x3f003f00u;
    unsigned amask[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) amask[i] = 0x1ff ^ ((seed[(tid + i) & 0xfffff] & 7u) == 0 ? 1u << (tid % 9) : 0u);   // a few border taps zeroed
    u32x4 rbh[2], rbl[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) { rbh[k] = u32x4{0x3f803f80u, 0x3f003f00u, 0xbf803f00u, 0x3f80bf00u}; rbl[k] = rbh[k]; }
    __syncthreads();

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto read_b = [&](Frags& f, const char* Bhi, int kk) {
        const int slot = kk * 4 + fq;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int o = lds_off(wn * 64 + j * 16 + frow, slot);
            f.bh[j] = *reinterpret_cast<const bf16x8*>(Bhi + o);
            f.bl[j] = *reinterpret_cast<const bf16x8*>(Bhi + 16384 + o);
        }
    };
    auto read_a = [&](Frags& f, int tap, int kk) {
        const int shift = (W + 1) + (tap / 3 - 1) * W + (tap % 3 - 1);
        const int slot = kk * 4 + fq;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = lds_off(wm * 64 + i * 16 + frow + shift, slot);
            const bool ok = (V == 5 || V == 7) ? true : (bool)((amask[i] >> tap) & 1u);
            const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(Ahi + o);
            const bf16x8 al = *reinterpret_cast<const bf16x8*>(Alo + o);
            f.ah[i] = ok ? ah : zero;
            f.al[i] = ok ? al : zero;
        }
    };
    auto mma_row = [&](const Frags& f, int i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.bh[j], f.ah[i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.bl[j], f.ah[i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.bh[j], f.al[i], acc[i][j], 0, 0, 0);
        }
    };
    auto store_b = [&](char* Bhi) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int q = tid + 512 * k;
            const int o = lds_off(q >> 3, q & 7);
            *reinterpret_cast<u32x4*>(Bhi + o) = rbh[k];
            *reinterpret_cast<u32x4*>(Bhi + 16384 + o) = rbl[k];
        }
    };

    int it = 0;
    if (V == 0 || V == 3) {
        for (int sl = 0; sl < slabs; ++sl)
            for (int tap = 0; tap < 9; ++tap, ++it) {
                char* Bhi = Bbase + ((it & 1) ? 32768 : 0);
                if (V == 0) store_b(Bhi);
                __syncthreads();
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    Frags f;
                    read_b(f, Bhi, kk);
                    read_a(f, tap, kk);
#pragma unroll
                    for (int i = 0; i < 4; ++i) mma_row(f, i);
                    __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 32, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 48, 0);
                }
            }
    } else {
        // fragments double-buffered: cur = half being multiplied, nxt = half being read
        Frags fa, fb;
        {
            char* Bhi = Bbase;
            store_b(Bhi);
            __syncthreads();
            read_b(fa, Bhi, 0);
            read_a(fa, 0, 0);
        }
        const int ntaps = slabs * 9;
        for (it = 0; it < ntaps; ++it) {
            const int tap = it % 9, tapn = (it + 1) % 9;
            char* Bhi = Bbase + ((it & 1) ? 32768 : 0);
            char* Bnx = Bbase + ((it & 1) ? 0 : 32768);
            // half 0 of this tap is in fa; read half 1 into fb while multiplying fa
            if (V == 1) {
                read_b(fb, Bhi, 1);
                read_a(fb, tap, 1);
#pragma unroll
                for (int i = 0; i < 4; ++i) mma_row(fa, i);
                __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 48, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 32, 0);
            } else {
                // V2: next half's reads in four groups between the MFMA rows
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int slot = 4 + fq;
                    {
                        const int o = lds_off(wn * 64 + i * 16 + frow, slot);
                        fb.bh[i] = *reinterpret_cast<const bf16x8*>(Bhi + o);
                        fb.bl[i] = *reinterpret_cast<const bf16x8*>(Bhi + 16384 + o);
                        const int shift = (W + 1) + (tap / 3 - 1) * W + (tap % 3 - 1);
                        const int oa = lds_off(wm * 64 + i * 16 + frow + shift, slot);
                        const bool ok = (V == 5 || V == 7) ? true : (bool)((amask[i] >> tap) & 1u);
                        const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
                        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(Ahi + oa);
                        const bf16x8 al = *reinterpret_cast<const bf16x8*>(Alo + oa);
                        fb.ah[i] = ok ? ah : zero;
                        fb.al[i] = ok ? al : zero;
                    }
                    mma_row(fa, i);
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                }
            }
            // write the next tap's tile (other buffer: last read one tap ago, before the previous barrier)
            if (V != 6 && V != 7) store_b(Bnx);
            // the next tap's pixel fragments do not depend on the barrier (same window): read them now, multiply fb meanwhile
            read_a(fa, tapn, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) mma_row(fb, i);
            if (V != 4 && V != 7) __syncthreads();
            read_b(fa, Bnx, 0);
#pragma unroll
            for (int i = 2; i < 4; ++i) mma_row(fb, i);
        }
    }
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j];
    out[blockIdx.x * 512 + tid] = s[0] + s[1] + s[2] + s[3];
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
    const size_t lds = (size_t)(256 + 2 * W + 2) * 256 + 65536;
    auto run = [&](const char* name, auto kern) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        std::vector<float> ms;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(WG), dim3(512), lds, 0, seed, out, slabs, W);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        const double taps = slabs * 9.0;
        printf("%-60s %8.2f ms  %7.1f ns per tap  executed %7.1f TFLOP/s\n", name, ms[1], ms[1] * 1e6 / taps,
               96.0 * 2 * 16 * 16 * 32 * 8 * WG * taps / ms[1] / 1e9);
    };
    printf("W = %d, LDS %zu bytes\n", W, lds);
    for (int round = 0; round < 2; ++round) {
        run("V0 reads-first per half (today), tile stores + barrier", core<0>);
        run("V3 V0 without the tile stores", core<3>);
        run("V1 fragments double-buffered, reads one half ahead", core<1>);
        run("V2 V1 with the reads in 4 groups between MFMA rows", core<2>);
        run("V4 V2 without the per-tap barrier", core<4>);
        run("V5 V2 without the border selects", core<5>);
        run("V6 V2 without the tile stores", core<6>);
        run("V7 V2 without barrier, selects, stores (reads + MFMAs)", core<7>);
    }
    CK(hipGetLastError());
    return 0;
}
