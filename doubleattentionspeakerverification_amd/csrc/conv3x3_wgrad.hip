// Weight gradient of conv3x3 (stride 1, pad 1) on NHWC fp32 activations, on the gfx950 matrix cores.
//
// What torch autograd derives for `self.convXY.weight` from `F.relu(self.convXY(x))` (reference scripts/CNNs.py:38-48 /
// 72-86, backward of train.py:220):   gw[n][c][ky][kx] = sum_{b,h,w} g[b,h,w,n] * x[b,h+ky-1,w+kx-1,c]   (zero outside).
//
// GEMM view per tap: D[n][c] = sum over pixels m of G^T[n][m] * X_tap[m][c] — the reduction runs over PIXELS, and both
// operands are pixel-strided in memory ([pixel][channel]).  ds_read_b64_tr_b16 reads a 4-row x 16-column block of a
// row-major LDS image and hands it to the lanes column-major, which is exactly the MFMA fragment of a K-strided operand:
// so g and x are staged row-major (as the forward kernel does) and BOTH operands are fetched with transposed reads.
// The nine taps are row shifts of one zero-padded image of the x window: in LDS every image row gets one extra (zero)
// column and every image one extra (zero) row, so a tap that leaves the image lands on zeros and no masking is needed.
// Operands are split bf16 (hi + lo, three MFMAs per product, fp32 accumulate) exactly like the forward kernel.
//
// Workgroup (4 waves): 64 output channels x 64 input channels x 9 taps; wave w owns input-channel block w (16 channels)
// for all four 16-channel output blocks: 36 accumulator tiles.  K is walked in chunks of 64 pixels that never straddle
// an utterance; the pixel range is split over `S` workgroups per tile whose partial tiles are summed in a fixed order
// by conv3x3_wgrad_reduce (no float atomics: bitwise reproducible).
#include "dasv_common.hpp"
#include "../../include/dasv_hip.h"

namespace dasv {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

constexpr int WKC = 64;   // pixels per K chunk

__device__ __forceinline__ void wsplit2(float a, float b, uint32_t& hi, uint32_t& lo) {
    const f32x2_t v = {a, b};
    const bf16x2_t h = __builtin_convertvector(v, bf16x2_t);
    const f32x2_t hf = __builtin_convertvector(h, f32x2_t);
    const bf16x2_t l = __builtin_convertvector(v - hf, bf16x2_t);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, l);
}

struct WgradParams {
    const float* x;     // [B,H,W,Cin]
    const float* g;     // [B,H,W,Cout]  gradient w.r.t. the convolution output (ReLU mask already applied)
    float* part;        // [S][Cout][9][Cin]
    int B, H, W, Cin, Cout;
    int ctiles;         // Cin / 64
    int ntiles;         // Cout / 64
    int chunks_per_image;   // ceil(H*W / WKC)
    int chunks_per_split;
    int S;
    int xrows;          // rows of the padded x window image in LDS
};

__host__ __device__ inline int wgrad_xrows(int W) { return WKC + WKC / W + 2 + 2 * (W + 2) + 1; }

// LDS images have 128-byte rows = four 32-byte segments (one 16-channel block each).  A row-dependent XOR of the segment
// index (against bank conflicts of the transposed reads) was measured twice: no change — the kernel is bound by the
// staging VALU work, not by LDS reads — so the plain layout stays.
__device__ __forceinline__ int wg_off(int row, int seg) { return row * 128 + (seg << 5); }

// transposed fragment: 8 consecutive K rows (two 4-row blocks) of a 16-column block; `p0`/`p1` are this lane's row addresses
__device__ __forceinline__ bf16x8 tr_frag(const char* p0, const char* p1) {
    typedef bf16x4 __attribute__((address_space(3))) * lds4;
    const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(p0));
    const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(p1));
    return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

// NX: x-window rows a thread stages per chunk (ceil(xrows / 16)); the next chunk's g and x are fetched into registers
// before the MFMA section of the current one and written to LDS after it, so HBM/L2 latency hides behind the MFMAs.
template <bool SPLIT, int NX>
__global__ void __launch_bounds__(256) conv3x3_wgrad_kernel(const WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int W = p.W, H = p.H, Cin = p.Cin, Cout = p.Cout;
    const int HW = H * W;
    char* Ghi = smem;                        // [WKC][64] bf16
    char* Glo = Ghi + WKC * 128;
    char* Xhi = Glo + WKC * 128;             // [xrows][64] bf16, zero-padded image coordinates
    char* Xlo = Xhi + (size_t)p.xrows * 128;
    int* Ltab = reinterpret_cast<int*>(Xlo + (size_t)p.xrows * 128);   // [WKC] window row of each chunk pixel (tap 0,0)

    const int tiles = p.ntiles * p.ctiles;
    const int tile = blockIdx.x % tiles, split = blockIdx.x / tiles;
    const int n0 = (tile / p.ctiles) * 64, c0 = (tile % p.ctiles) * 64;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;   // wave = input-channel block
    const int grp = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int scc = tid & 15, srow = tid >> 4;                     // staging: 4-channel chunk, first row

    f32x4 acc[4][9];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int chunk_lo = split * p.chunks_per_split;
    const int total_chunks = p.B * p.chunks_per_image;
    const int chunk_hi = min(total_chunks, chunk_lo + p.chunks_per_split);

    f32x4 gr[4], xr[NX];
    int w0_next = 0;
    // issue the global loads of chunk `ch` (rows beyond the utterance / outside the image become zeros)
    auto fetch = [&](int ch) {
        const int b = ch / p.chunks_per_image;
        const int off = (ch - b * p.chunks_per_image) * WKC;
        const int npx = min(WKC, HW - off);
        const long long m0 = (long long)b * HW + off;
        const int h0 = off / W, w0 = off - h0 * W;
        w0_next = w0;
        // padded index of (b,h,w): ((b*(H+1) + h) * (W+1)) + w; window row 0 <-> padded index P(m0) - (W+2)
        const long long pbase = ((long long)b * (H + 1) + h0) * (W + 1) + w0 - (W + 2);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = srow + 16 * k;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row < npx) v = *reinterpret_cast<const f32x4*>(p.g + (size_t)(m0 + row) * Cout + n0 + scc * 4);
            gr[k] = v;
        }
        // decode this thread's first window row once, then step by 16 padded positions with carries
        long long pidx = pbase + srow;
        long long t2 = pidx >= 0 ? pidx / (W + 1) : 0;
        int wq = pidx >= 0 ? (int)(pidx - t2 * (W + 1)) : 0;
        int hq = pidx >= 0 ? (int)(t2 % (H + 1)) : 0;
        long long bq = pidx >= 0 ? t2 / (H + 1) : 0;
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (srow + 16 * k < p.xrows && pidx >= 0 && wq < W && hq < H && bq < p.B)
                v = *reinterpret_cast<const f32x4*>(p.x + ((size_t)(bq * H + hq) * W + wq) * Cin + c0 + scc * 4);
            xr[k] = v;
            if (pidx >= 0) {
                wq += 16;
                while (wq >= W + 1) {
                    wq -= W + 1;
                    if (++hq == H + 1) { hq = 0; ++bq; }
                }
                pidx += 16;
            } else {
                pidx += 16;
                if (pidx >= 0) {
                    t2 = pidx / (W + 1);
                    wq = (int)(pidx - t2 * (W + 1));
                    hq = (int)(t2 % (H + 1));
                    bq = t2 / (H + 1);
                }
            }
        }
    };
    // split and park the fetched chunk in LDS
    auto park = [&]() {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t h0_, l0_, h1_, l1_;
            wsplit2(gr[k][0], gr[k][1], h0_, l0_);
            wsplit2(gr[k][2], gr[k][3], h1_, l1_);
            const int o = wg_off(srow + 16 * k, scc >> 2) + (scc & 3) * 8;
            *reinterpret_cast<uint2*>(Ghi + o) = make_uint2(h0_, h1_);
            if (SPLIT) *reinterpret_cast<uint2*>(Glo + o) = make_uint2(l0_, l1_);
        }
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int r = srow + 16 * k;
            if (r < p.xrows) {
                uint32_t h0_, l0_, h1_, l1_;
                wsplit2(xr[k][0], xr[k][1], h0_, l0_);
                wsplit2(xr[k][2], xr[k][3], h1_, l1_);
                const int o = wg_off(r, scc >> 2) + (scc & 3) * 8;
                *reinterpret_cast<uint2*>(Xhi + o) = make_uint2(h0_, h1_);
                if (SPLIT) *reinterpret_cast<uint2*>(Xlo + o) = make_uint2(l0_, l1_);
            }
        }
        if (tid < WKC) Ltab[tid] = tid + (w0_next + tid) / W + (W + 2);   // chunk pixel j -> window row (row ends add one)
    };

    if (chunk_lo < chunk_hi) fetch(chunk_lo);
    for (int ch = chunk_lo; ch < chunk_hi; ++ch) {
        park();
        __syncthreads();
        if (ch + 1 < chunk_hi) fetch(ch + 1);     // in flight during the MFMA section

        // ---- MFMA: two K steps of 32 pixels; the fragments of tap t+1 are requested before tap t's MFMAs issue
#pragma unroll
        for (int k0 = 0; k0 < WKC; k0 += 32) {
            const int j0 = k0 + 8 * grp + q, j1 = j0 + 4;            // this lane's rows of the two 4-row blocks
            const int col = 4 * pp * 2;                               // byte offset of this lane's 4 columns in a 16-col block
            bf16x8 gh[4], gl[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int o0 = wg_off(j0, i) + col, o1 = wg_off(j1, i) + col;
                gh[i] = tr_frag(Ghi + o0, Ghi + o1);
                if (SPLIT) gl[i] = tr_frag(Glo + o0, Glo + o1);
            }
            const int r0 = Ltab[j0], r1 = Ltab[j1];
            auto xfrag = [&](int t, bf16x8& xh, bf16x8& xl) {
                const int toff = (t / 3 - 1) * (W + 1) + (t % 3 - 1);
                const char* x0 = Xhi + wg_off(r0 + toff, wv) + col;
                const char* x1 = Xhi + wg_off(r1 + toff, wv) + col;
                xh = tr_frag(x0, x1);
                if (SPLIT) xl = tr_frag(x0 + (size_t)p.xrows * 128, x1 + (size_t)p.xrows * 128);
            };
            bf16x8 xh[2], xl[2];
            xfrag(0, xh[0], xl[0]);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                if (t + 1 < 9) xfrag(t + 1, xh[(t + 1) & 1], xl[(t + 1) & 1]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh[i], xh[t & 1], acc[i][t], 0, 0, 0);
                    if (SPLIT) {
                        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gl[i], xh[t & 1], acc[i][t], 0, 0, 0);
                        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh[i], xl[t & 1], acc[i][t], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    }

    // partial tile -> part[split][n][tap][c]; D layout: col (c) = lane & 15, row (n) = (lane >> 4) * 4 + reg
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + i * 16 + (lane >> 4) * 4 + r;
                p.part[(((size_t)split * Cout + n) * 9 + t) * Cin + c0 + wv * 16 + (lane & 15)] = acc[i][t][r];
            }
}

// gw[n][c][tap] (torch layout [Cout][Cin][3][3]) = sum_s part[s][n][tap][c], fixed order
__global__ void __launch_bounds__(256) conv3x3_wgrad_reduce(const float* __restrict__ part, float* __restrict__ gw, int S,
                                                            int Cout, int Cin) {
    const long long n_el = (long long)Cout * 9 * Cin;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_el; i += (long long)gridDim.x * blockDim.x) {
        float v = 0.f;
#pragma unroll 4
        for (int s = 0; s < S; ++s) v += part[(size_t)s * n_el + i];
        const int c = (int)(i % Cin);
        const int tap = (int)((i / Cin) % 9);
        const long long n = i / ((long long)Cin * 9);
        gw[(n * Cin + c) * 9 + tap] = v;
    }
}

static void wgrad_plan(int B, int H, int W, int Cin, int Cout, WgradParams& p) {
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.ctiles = Cin / 64; p.ntiles = Cout / 64;
    p.chunks_per_image = (H * W + WKC - 1) / WKC;
    const int total = B * p.chunks_per_image, tiles = p.ctiles * p.ntiles;
    int S = (1024 + tiles - 1) / tiles;
    if (S > total) S = total;
    if (S < 1) S = 1;
    p.chunks_per_split = (total + S - 1) / S;
    p.S = (total + p.chunks_per_split - 1) / p.chunks_per_split;
    p.xrows = wgrad_xrows(W);
}

}  // namespace dasv

extern "C" {

size_t dasv_conv3x3_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % 64 != 0 || Cout % 64 != 0) return 0;
    dasv::WgradParams p;
    dasv::wgrad_plan(B, H, W, Cin, Cout, p);
    return (size_t)p.S * Cout * 9 * Cin * sizeof(float);
}

int dasv_conv3x3_wgrad_nhwc(int mode, const float* x, const float* g, float* gw, void* workspace, size_t workspace_bytes,
                            int B, int H, int W, int Cin, int Cout, dasv_stream_t stream) {
    if (x == nullptr || g == nullptr || gw == nullptr || workspace == nullptr) return DASV_E_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % 64 != 0 || Cout % 64 != 0) return DASV_E_SHAPE;
    if ((long long)B * (H + 1) * (W + 1) > 0x7fffffffLL / 4) return DASV_E_SHAPE;
    if (mode != 0 && mode != 1) return DASV_E_DTYPE;
    dasv::WgradParams p;
    dasv::wgrad_plan(B, H, W, Cin, Cout, p);
    if (workspace_bytes < (size_t)p.S * Cout * 9 * Cin * sizeof(float)) return DASV_E_WORKSPACE;
    p.x = x; p.g = g; p.part = static_cast<float*>(workspace);
    // both planes are always carved (the bf16 mode simply leaves the lo planes unused)
    const size_t lds = 2 * (size_t)dasv::WKC * 128 + 2 * (size_t)p.xrows * 128 + dasv::WKC * sizeof(int) + 16;
    if (lds > 160 * 1024 || W > 80) return DASV_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)(p.ctiles * p.ntiles * p.S)), block(256);
    const int nx = (p.xrows + 15) / 16;
#define DASV_LAUNCH_WGRAD(SP, NXV)                                                                                 \
    do {                                                                                                           \
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&dasv::conv3x3_wgrad_kernel<SP, NXV>),    \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                 \
        if (e2 != hipSuccess) return (int)e2;                                                                      \
        hipLaunchKernelGGL((dasv::conv3x3_wgrad_kernel<SP, NXV>), grid, block, lds, s, p);                          \
    } while (0)
    if (nx > 15) return DASV_E_SHAPE;   // maps wider than 80 columns
    if (mode == 0) {
        if (nx <= 8) DASV_LAUNCH_WGRAD(true, 8);
        else if (nx <= 10) DASV_LAUNCH_WGRAD(true, 10);
        else DASV_LAUNCH_WGRAD(true, 15);
    } else {
        if (nx <= 8) DASV_LAUNCH_WGRAD(false, 8);
        else if (nx <= 10) DASV_LAUNCH_WGRAD(false, 10);
        else DASV_LAUNCH_WGRAD(false, 15);
    }
#undef DASV_LAUNCH_WGRAD
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    const long long n_el = (long long)Cout * 9 * Cin;
    const int blocks = (int)((n_el + 255) / 256 < 4096 ? (n_el + 255) / 256 : 4096);
    hipLaunchKernelGGL(dasv::conv3x3_wgrad_reduce, dim3(blocks), dim3(256), 0, s, p.part, gw, p.S, Cout, Cin);
    return (int)hipGetLastError();
}

}  // extern "C"
