// Does the WIDTH of a lane's store matter for a kernel that only writes?  The packed activation format (dasv_common.hpp: per pixel and
// 64-channel slab 128 B fp16 | 64 B e4m3 | 64 B e4m3) is written by store_x8 as 8 + 4 + 4 bytes per lane (four channels per lane); the
// kernels that do little else (the pool's backward scatter, conv11) run at 2.0-2.7 TB/s.  This probe writes 393 MB (the [64,300,80,64]
// map) five ways: plain 16-byte stores; the 8/4/4 pattern pixel by pixel; the same slabs with 16 channels per lane (four 16-byte stores);
// and both in the pool-backward order (a lane owns a 2x2 window: four pixels in two rows).
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/store_width_probe.hip -o /tmp/swp && /tmp/swp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { if ((x) != hipSuccess) { printf("HIP error at line %d\n", __LINE__); return; } } while (0)
#define CKM(x) do { if ((x) != hipSuccess) { printf("HIP error at line %d\n", __LINE__); return 1; } } while (0)

constexpr int C = 64, W = 80, H = 300, B = 64;
constexpr long long NPIX = (long long)B * H * W;

__global__ void __launch_bounds__(256) k_plain16(float* y, long long nvec) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256)
        reinterpret_cast<float4*>(y)[i] = float4{(float)i, 1.f, 2.f, 3.f};
}
__global__ void __launch_bounds__(256) k_844(float* y, long long nvec) {           // nvec = pixels * 16
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        const int q = (int)(i & 15);
        char* slab = reinterpret_cast<char*>(y) + (i >> 4) * 256;
        *reinterpret_cast<uint2*>(slab + q * 8) = make_uint2((uint32_t)i, 7u);
        *reinterpret_cast<uint32_t*>(slab + 128 + q * 4) = (uint32_t)i;
        *reinterpret_cast<uint32_t*>(slab + 192 + q * 4) = (uint32_t)i + 1u;
    }
}
__global__ void __launch_bounds__(256) k_16x4(float* y, long long nvec) {          // nvec = pixels * 4
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        const int j = (int)(i & 3);
        char* slab = reinterpret_cast<char*>(y) + (i >> 2) * 256;
        const uint4 v = make_uint4((uint32_t)i, 1u, 2u, 3u);
        *reinterpret_cast<uint4*>(slab + j * 32) = v;
        *reinterpret_cast<uint4*>(slab + j * 32 + 16) = v;
        *reinterpret_cast<uint4*>(slab + 128 + j * 16) = v;
        *reinterpret_cast<uint4*>(slab + 192 + j * 16) = v;
    }
}
// pool-backward order: element i -> (window, lane part); the window's four pixels (two rows) are written by the same lane
template <int PARTS>   // 16: 8/4/4-byte stores, 4: 16-byte stores
__global__ void __launch_bounds__(256) k_window(float* y, long long nvec) {
    constexpr int Wo = W / 2, Ho = H / 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        const uint32_t ii = (uint32_t)i, p = ii % PARTS, win = ii / PARTS, wo = win % Wo, r = win / Wo, ho = r % Ho, b = r / Ho;
        const long long p0 = ((long long)b * H + 2 * ho) * W + 2 * wo;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            char* slab = reinterpret_cast<char*>(y) + (p0 + (k >> 1) * W + (k & 1)) * 256;
            if (PARTS == 16) {
                *reinterpret_cast<uint2*>(slab + p * 8) = make_uint2(ii, 7u);
                *reinterpret_cast<uint32_t*>(slab + 128 + p * 4) = ii;
                *reinterpret_cast<uint32_t*>(slab + 192 + p * 4) = ii + 1u;
            } else {
                const uint4 v = make_uint4(ii, 1u, 2u, 3u);
                *reinterpret_cast<uint4*>(slab + p * 32) = v;
                *reinterpret_cast<uint4*>(slab + p * 32 + 16) = v;
                *reinterpret_cast<uint4*>(slab + 128 + p * 16) = v;
                *reinterpret_cast<uint4*>(slab + 192 + p * 16) = v;
            }
        }
    }
}

// the pool backward itself: 16 bytes of pooled gradient + 4 arg-max bytes read per lane, then the window's four pixels written (8/4/4);
// U windows per trip, all of a trip's loads issued before its stores
template <int U>
__global__ void __launch_bounds__(256) k_window_rd(float* y, const float4* __restrict__ g, const uint32_t* __restrict__ idx, long long nvec) {
    constexpr int Wo = W / 2, Ho = H / 2;
    const long long stride = (long long)gridDim.x * 256;
    for (long long i0 = (long long)blockIdx.x * 256 + threadIdx.x; i0 < nvec; i0 += stride * U) {
        float4 gv[U]; uint32_t cv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long i = i0 + u * stride;
            if (i < nvec) { gv[u] = g[i]; cv[u] = idx[i]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long i = i0 + u * stride;
            if (i >= nvec) break;
            const uint32_t ii = (uint32_t)i, p = ii % 16, win = ii / 16, wo = win % Wo, r = win / Wo, ho = r % Ho, b = r / Ho;
            const long long p0 = ((long long)b * H + 2 * ho) * W + 2 * wo;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                char* slab = reinterpret_cast<char*>(y) + (p0 + (k >> 1) * W + (k & 1)) * 256;
                const bool on = (cv[u] & 3u) == (uint32_t)k;
                *reinterpret_cast<uint2*>(slab + p * 8) = on ? make_uint2(__float_as_uint(gv[u].x), __float_as_uint(gv[u].y)) : make_uint2(0u, 0u);
                *reinterpret_cast<uint32_t*>(slab + 128 + p * 4) = on ? __float_as_uint(gv[u].z) : 0u;
                *reinterpret_cast<uint32_t*>(slab + 192 + p * 4) = on ? __float_as_uint(gv[u].w) : 0u;
            }
        }
    }
}

template <typename F>
static void run(const char* name, F launch, float* y, size_t bytes) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 3; ++r) launch(y + (r & 1) * (bytes / 4));
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) launch(y + (r & 1) * (bytes / 4));
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s %7.1f us  %5.2f TB/s\n", name, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
}

int main() {
    const size_t bytes = (size_t)NPIX * C * 4;
    float* y; CKM(hipMalloc(&y, 2 * bytes)); CKM(hipMemset(y, 0, 2 * bytes));
    float4* g; uint32_t* idx;
    CKM(hipMalloc(&g, bytes / 4)); CKM(hipMalloc(&idx, bytes / 16));
    CKM(hipMemset(g, 0, bytes / 4)); CKM(hipMemset(idx, 1, bytes / 16));
    for (int blocks : {2048, 8192, 65536}) {
        printf("grid %d workgroups of 256\n", blocks);
        run("plain 16-byte stores", [&](float* d) { hipLaunchKernelGGL(k_plain16, dim3(blocks), dim3(256), 0, 0, d, NPIX * 16); }, y, bytes);
        run("slabs, 8/4/4 bytes per lane", [&](float* d) { hipLaunchKernelGGL(k_844, dim3(blocks), dim3(256), 0, 0, d, NPIX * 16); }, y, bytes);
        run("slabs, 4 x 16 bytes per lane", [&](float* d) { hipLaunchKernelGGL(k_16x4, dim3(blocks), dim3(256), 0, 0, d, NPIX * 4); }, y, bytes);
        run("2x2 windows, 8/4/4 bytes", [&](float* d) { hipLaunchKernelGGL(k_window<16>, dim3(blocks), dim3(256), 0, 0, d, NPIX / 4 * 16); }, y, bytes);
        run("pool backward (reads), U = 1", [&](float* d) { hipLaunchKernelGGL(k_window_rd<1>, dim3(blocks), dim3(256), 0, 0, d, g, idx, NPIX / 4 * 16); }, y, bytes);
        run("pool backward (reads), U = 2", [&](float* d) { hipLaunchKernelGGL(k_window_rd<2>, dim3(blocks), dim3(256), 0, 0, d, g, idx, NPIX / 4 * 16); }, y, bytes);
        run("pool backward (reads), U = 4", [&](float* d) { hipLaunchKernelGGL(k_window_rd<4>, dim3(blocks), dim3(256), 0, 0, d, g, idx, NPIX / 4 * 16); }, y, bytes);
        run("2x2 windows, 4 x 16 bytes", [&](float* d) { hipLaunchKernelGGL(k_window<4>, dim3(blocks), dim3(256), 0, 0, d, NPIX / 4 * 4); }, y, bytes);
    }
    return 0;
}
