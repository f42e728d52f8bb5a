// Hardware check of the LDS transposed read the weight-gradient kernel is built on (csrc/conv3x3_wgrad.hip, tr_frag()):
// `ds_read_b64_tr_b16` with the kernel's lane addressing must hand lane l the eight K-consecutive bf16 values
// image[8*(l>>4) + t][l & 15], t = 0..7 — the 16x16x32 MFMA operand fragment of a K-strided (row = K) LDS image.
//   hipcc --offload-arch=gfx950 -O2 tools/tr_check.hip -o tools/_diag/tr_check && tools/_diag/tr_check
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef short bf16x4 __attribute__((ext_vector_type(4)));

__global__ void tr_kernel(uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t img[32][64];   // [k row][column], 128-byte rows like the kernel's images
    for (int i = threadIdx.x; i < 32 * 64; i += 64) img[i / 64][i % 64] = (uint16_t)((i / 64) * 64 + i % 64);
    __syncthreads();
    const int lane = threadIdx.x, grp = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    typedef bf16x4 __attribute__((address_space(3))) * lds4;
    const char* base = reinterpret_cast<const char*>(&img[0][0]);
    const int j0 = 8 * grp + q, j1 = j0 + 4, col = 4 * pp * 2;      // exactly the kernel's addressing (16-column block 0)
    const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(base + j0 * 128 + col));
    const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(base + j1 * 128 + col));
    for (int t = 0; t < 4; ++t) { out[lane * 8 + t] = (uint16_t)a[t]; out[lane * 8 + 4 + t] = (uint16_t)b[t]; }
}

int main() {
    uint16_t *d, h[64 * 8];
    if (hipMalloc((void**)&d, sizeof(h)) != hipSuccess) return 2;
    hipLaunchKernelGGL(tr_kernel, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int t = 0; t < 8; ++t) {
            const int want = (8 * (l >> 4) + t) * 64 + (l & 15);
            if (h[l * 8 + t] != want) { if (bad < 8) printf("lane %d t %d: got %d want %d\n", l, t, h[l * 8 + t], want); ++bad; }
        }
    puts(bad ? "TR_CHECK_FAIL" : "TR_CHECK_OK");
    return bad ? 1 : 0;
}
