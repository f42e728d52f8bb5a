// Probe of ds_read_b64_tr_b8 on gfx950 (no ISA manual in this image): what does lane L receive?
// Every lane passes the address of an 8-byte group of an LDS image whose byte at offset a holds (a & 0xff) and whose second plane
// holds (a >> 8); the result bytes tell which (lane address, byte) each destination byte came from.
//   hipcc --offload-arch=gfx950 -O2 tools/tr8_check.hip -o tools/_diag/tr8_check && tools/_diag/tr8_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int i32x2 __attribute__((ext_vector_type(2)));

__global__ void probe(uint32_t* out_lo, uint32_t* out_hi, int stride) {
    __shared__ __attribute__((aligned(16))) unsigned char img[2][16384];
    for (int i = threadIdx.x; i < 16384; i += 64) { img[0][i] = (unsigned char)(i & 0xff); img[1][i] = (unsigned char)(i >> 8); }
    __syncthreads();
    const int lane = threadIdx.x;
    // lane L addresses row L (stride bytes apart), column 0
    typedef i32x2 __attribute__((address_space(3))) * lds2;
    const i32x2 a = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds2)(&img[0][lane * stride]));
    const i32x2 b = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds2)(&img[1][lane * stride]));
    out_lo[lane * 2] = a[0]; out_lo[lane * 2 + 1] = a[1];
    out_hi[lane * 2] = b[0]; out_hi[lane * 2 + 1] = b[1];
}

int main() {
    uint32_t *lo, *hi;
    hipMalloc(&lo, 512); hipMalloc(&hi, 512);
    const int stride = 64;
    probe<<<1, 64>>>(lo, hi, stride);
    uint32_t hl[128], hh[128];
    hipMemcpy(hl, lo, 512, hipMemcpyDeviceToHost); hipMemcpy(hh, hi, 512, hipMemcpyDeviceToHost);
    printf("lane L addresses image offset L*%d.  dst byte k of lane L <- (source lane, source byte):\n", stride);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int k = 0; k < 8; ++k) {
            const int a = ((hl[l * 2 + k / 4] >> (8 * (k % 4))) & 0xff) | (((hh[l * 2 + k / 4] >> (8 * (k % 4))) & 0xff) << 8);
            printf(" (%2d,%d)", a / stride, a % stride);
        }
        printf("\n");
    }
    return 0;
}
