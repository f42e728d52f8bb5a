// How fast can ONE workgroup per CU stream its own contiguous slice?  (Why the fused DoubleMHA kernel spreads an utterance over several
// CUs and pays an exchange, instead of giving each utterance to one workgroup: tools/_diag)
// N workgroups of 512 threads, each sums `bytes_per_wg` contiguous bytes (16-byte loads, UNROLL in flight per thread); N = 64, 128, 256,
// 512.  Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/cu_stream_probe.hip -o tools/_diag/cu_stream_probe && tools/_diag/cu_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int UNROLL>
__global__ void __launch_bounds__(512) stream_kernel(const float4* __restrict__ x, long long vec_per_wg, float* __restrict__ out) {
    const float4* base = x + (long long)blockIdx.x * vec_per_wg;
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    for (long long i = threadIdx.x; i < vec_per_wg; i += 512LL * UNROLL) {
        float4 v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) v[k] = (i + 512LL * k < vec_per_wg) ? base[i + 512LL * k] : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 1.2345f) out[blockIdx.x] = acc.x;
}

int main() {
    const size_t total = 512ull << 20;                 // rotate over 512 MB: beyond the Infinity Cache
    float4* x; float* out;
    hipMalloc(&x, total); hipMalloc(&out, 4096 * 4);
    hipMemset(x, 0, total);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t sizes[] = {389120, 614400};            // the model's pooling input per utterance; BASELINE config 2's
    for (size_t bytes : sizes)
        for (int nwg : {64, 128, 256, 512}) {
            const long long vec = bytes / 16;
            const size_t span = (size_t)nwg * bytes;
            const int reps = 200;
            // warm
            for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(stream_kernel<12>, dim3(nwg), dim3(512), 0, 0, x, vec, out);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < reps; ++r) {
                const size_t off = ((size_t)r * span) % (total - span);
                hipLaunchKernelGGL(stream_kernel<12>, dim3(nwg), dim3(512), 0, 0, x + off / 16, vec, out);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1e3 / reps;
            printf("%6zu KB per workgroup x %3d workgroups: %6.2f us per launch = %6.1f GB/s total, %5.1f GB/s per workgroup\n", bytes / 1024, nwg, us,
                   span / us / 1e3, bytes / us / 1e3);
        }
    return 0;
}
