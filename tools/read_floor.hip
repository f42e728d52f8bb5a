// Floor probe for the fused DoubleMHA kernel (BASELINE config 2: x = [64,300,512] fp32 = 39,321,600 B per launch).
// What does the SIMPLEST kernel that only reads those bytes once need on this chip?  No softmax, no exchange between
// workgroups, no outputs beyond one float per workgroup.  Same measurement rules as bench.py's `pool` workload: 8 rotating
// input buffers (315 MB > the 256 MiB Infinity Cache), back-to-back launches on one stream, HIP events around the batch;
// run it under `rocprofv3 --kernel-trace --stats` for per-kernel durations (profiles/r02_read_floor*.csv).
//
//   hipcc -O3 --offload-arch=gfx950 tools/read_floor.hip -o tools/_diag/read_floor && tools/_diag/read_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

// each thread: DEPTH independent 16-byte loads in flight per round, grid-stride over the buffer
template <int DEPTH>
__global__ void __launch_bounds__(256) read_sum(const f4* __restrict__ x, long long nvec, float* __restrict__ out) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (; i + (DEPTH - 1) * stride < nvec; i += DEPTH * stride) {
        f4 v[DEPTH];
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) v[k] = x[i + k * stride];
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) acc += v[k];
    }
    for (; i < nvec; i += stride) acc += x[i];
    float s = acc[0] + acc[1] + acc[2] + acc[3];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(out + blockIdx.x, s);   // keeps the loads alive; one atomic per wave
}

// the pooling kernel's own access shape: one 512-thread workgroup per CU-sized slice (2 heads x 300 rows x 256 B of one
// utterance), 16-lane groups own 256-byte rows, 4 rows per lane per pass, two passes in flight — minus all arithmetic
__global__ void __launch_bounds__(512) read_like_dmha(const char* __restrict__ x, float* __restrict__ out) {
    const int b = blockIdx.x >> 2, a = blockIdx.x & 3;     // 64 utterances x 4 head pairs
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int uu = w >> 2, wu = w & 3, h = a * 2 + uu;
    const int c = lane & 15, phase = wu * 4 + (lane >> 4);
    const char* row0 = x + ((size_t)b * 300 * 512 + (size_t)h * 64) * 4 + c * 16;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    f4 bufA[4], bufB[4];
    auto load = [&](f4 (&buf)[4], int pass) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = phase + 16 * (pass * 4 + i);
            f4 v = {0.f, 0.f, 0.f, 0.f};
            if (t < 300) v = *reinterpret_cast<const f4*>(row0 + (size_t)t * 2048);
            buf[i] = v;
        }
    };
    load(bufA, 0); load(bufB, 1);
    for (int pass = 0; pass < 5; pass += 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc += bufA[i];
        if (pass + 2 < 5) load(bufA, pass + 2);
        if (pass + 1 < 5) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc += bufB[i];
            if (pass + 3 < 5) load(bufB, pass + 3);
        }
    }
    float s = acc[0] + acc[1] + acc[2] + acc[3];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) atomicAdd(out + blockIdx.x, s);
}

// whole-utterance workgroups: NWG workgroups of 1024 threads, each reads its own contiguous 1/NWG of the buffer (NWG = 64: one
// utterance per workgroup, all heads on one CU — no exchange between CUs would be needed, but only 64 CUs pull)
template <int DEPTH>
__global__ void __launch_bounds__(1024) read_chunk(const f4* __restrict__ x, long long nvec, float* __restrict__ out) {
    const long long per = nvec / gridDim.x;
    const f4* p = x + per * blockIdx.x;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    long long i = threadIdx.x;
    for (; i + (DEPTH - 1) * 1024 < per; i += DEPTH * 1024) {
        f4 v[DEPTH];
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) v[k] = p[i + k * 1024];
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) acc += v[k];
    }
    for (; i < per; i += 1024) acc += p[i];
    float s = acc[0] + acc[1] + acc[2] + acc[3];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(out + blockIdx.x, s);
}

int main() {
    const size_t bytes = 64ull * 300 * 512 * 4;
    const int NBUF = 8, REP = 200;
    std::vector<char*> xs(NBUF);
    for (auto& p : xs) { CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 0x3c, bytes)); }
    float* out; CK(hipMalloc(&out, 1 << 20)); CK(hipMemset(out, 0, 1 << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long long nvec = bytes / 16;
    auto time_it = [&](const char* name, auto launch) {
        std::vector<float> us;
        for (int rep = 0; rep < 5; ++rep) {
            for (int i = 0; i < 16; ++i) launch(i % NBUF);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < REP; ++i) launch(i % NBUF);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            us.push_back(ms * 1e3f / REP);
        }
        std::sort(us.begin(), us.end());
        printf("%-44s median %6.2f us  (min %6.2f)  -> %5.0f GB/s = %4.1f %% of 8 TB/s\n", name, us[2], us[0], bytes / us[2] / 1e3,
               100.0 * bytes / us[2] / 1e3 / 8000.0);
    };
#define RS(G, D) time_it("read_sum grid " #G " x256, depth " #D, [&](int k) { hipLaunchKernelGGL(read_sum<D>, dim3(G), dim3(256), 0, 0, (const f4*)xs[k], nvec, out); })
    RS(256, 8); RS(512, 8); RS(1024, 4); RS(1024, 8); RS(2048, 4); RS(2048, 8); RS(4096, 2); RS(4096, 4); RS(8192, 2);
    time_it("read_like_dmha 256 x512 (pooling access shape)", [&](int k) { hipLaunchKernelGGL(read_like_dmha, dim3(256), dim3(512), 0, 0, xs[k], out); });
#define RC(G, D) time_it("read_chunk " #G " x1024 (whole-utterance WGs), depth " #D, [&](int k) { hipLaunchKernelGGL(read_chunk<D>, dim3(G), dim3(1024), 0, 0, (const f4*)xs[k], nvec, out); })
    RC(64, 8); RC(64, 16); RC(64, 32); RC(128, 8); RC(128, 16); RC(256, 8); RC(256, 16);
    // the same with the input hot in the Infinity Cache (one buffer): separates HBM from the CU load path
    time_it("read_like_dmha, ONE buffer (cache-hot)", [&](int) { hipLaunchKernelGGL(read_like_dmha, dim3(256), dim3(512), 0, 0, xs[0], out); });
    time_it("read_sum 2048 x256 d4, ONE buffer (cache-hot)", [&](int) { hipLaunchKernelGGL(read_sum<4>, dim3(2048), dim3(256), 0, 0, (const f4*)xs[0], nvec, out); });
    CK(hipGetLastError());
    return 0;
}
