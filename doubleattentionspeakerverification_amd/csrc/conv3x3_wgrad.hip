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

#include <stdlib.h>

namespace dasv {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

constexpr int WKC = 64;   // pixels per K chunk

// timing experiments only (tools/wgrad_diag.py builds this file with -DDASV_WGRAD_DIAG=n; results are wrong then):
// 1 = stage only the first two chunks, 2 = no MFMA section, 3 = the x window parked as fetched (one 16-byte store per piece, no
// conversion: what a PACKED x operand would cost the producers), 4 = x and g parked as fetched
#ifndef DASV_WGRAD_DIAG
#define DASV_WGRAD_DIAG 0
#endif

#ifndef DASV_WGRAD_XCD
#define DASV_WGRAD_XCD 1    // XCD-aware (split, tile) order: 6.53 -> 6.47 ms (interleaved A/B, tools/wgrad_diag.py)
#endif
#ifndef DASV_WGRAD_NOPK
#define DASV_WGRAD_NOPK 1   // measured in one run (tools/wgrad_diag.py, "packed subs"): 7.33 ms with v_pk_add_f32, 6.79 ms without
#endif
__device__ __forceinline__ void wsplit2(float a, float b, uint32_t& hi, uint32_t& lo) {
    const f32x2_t v = {a, b};
    const bf16x2_t h = __builtin_convertvector(v, bf16x2_t);
    const f32x2_t hf = __builtin_convertvector(h, f32x2_t);
#if DASV_WGRAD_NOPK
    // two scalar subtractions: the compiler otherwise SLP-packs them into v_pk_add_f32, which is expensive to issue beside
    // another wave's MFMAs (MI355X_MICROARCH.md, constants table)
    float l0, l1;
    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(l0) : "v"(a), "v"(hf[0]));
    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(l1) : "v"(b), "v"(hf[1]));
    const f32x2_t d = {l0, l1};
    const bf16x2_t l = __builtin_convertvector(d, bf16x2_t);
#else
    const bf16x2_t l = __builtin_convertvector(v - hf, bf16x2_t);
#endif
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
    int magic_w;        // floor(65536 / W) + 1: (v * magic_w) >> 16 == v / W for v * W < 65536
    int hcarries;       // image-row carries a 16-position step of the window walk can cause: ceil((16/(W+1) + 1) / (H+1))
    int patch;          // 1: chunks are 8 x 8 pixel patches (W % 8 == 0); chunks_per_image = ceil(H/8) * (W/8)
    int ph;             // patch == 2: whole image rows per chunk (64 / W)
    const uint32_t* x_rec;   // X8: range records of x and g (dasv_hip.h): [0] = bits of max |v|
    const uint32_t* g_rec;
};

__host__ __device__ inline int wgrad_xrows(int W) { return WKC + WKC / W + 2 + 2 * (W + 2) + 1; }

// LDS images have 128-byte rows = four 32-byte segments (one 16-channel block each).  A row-dependent XOR of the segment
// index (seg ^ ((row >> 1) & 3), against bank conflicts of the transposed reads) was measured again on the
// producer/consumer kernel, both layouts in one run: consumers alone 1.5 % faster, the full kernel 3 % slower — the plain
// layout stays.
__device__ __forceinline__ int wg_off(int row, int seg) { return row * 128 + (seg << 5); }
// e4m3 planes (X8): 64-byte rows of four 16-byte channel blocks.  A transposed read's 16 lanes address 8 consecutive rows x 16 bytes:
// rows r and r + 4 share their banks, so the block index is XORed with (row >> 2) & 3 (rows r, r + 4 then land in different blocks)
__device__ __forceinline__ int w8_off(int row, int blk) { return row * 64 + ((blk ^ ((row >> 2) & 3)) << 4); }

// transposed fragment: 8 consecutive K rows (two 4-row blocks) of a 16-column block; `p0`/`p1` are this lane's row addresses
__device__ __forceinline__ bf16x8 tr_frag(const char* p0, const char* p1) {
    typedef bf16x4 __attribute__((address_space(3))) * lds4;
    const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(p0));
    const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(p1));
    return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

// One LDS buffer: G_hi | G_lo ([WKC][64] bf16) | X_hi | X_lo ([16 * NX][64] bf16, zero-padded image coordinates; the
// window's `xrows` rows rounded up to what the 16 staging rows per step cover) | Ltab[WKC]
__host__ __device__ constexpr size_t wgrad_buffer_bytes(int nx) {
    return 2 * (size_t)WKC * 128 + 2 * (size_t)(16 * nx) * 128 + WKC * sizeof(int);
}

// Workgroup = 8 waves with two roles (the accumulators alone take 144 registers, so a SIMD holds two such waves at most;
// two copies of one program in lock step between barriers would leave the matrix cores idle while both stage):
//   waves 0-3 (one per SIMD)  CONSUMERS: MFMAs only, out of the LDS buffer of the current chunk;
//   waves 4-7 (one per SIMD)  PRODUCERS: fetch chunk i+2 from global memory, split chunk i+1 and park it in the other
//                             LDS buffer — their VALU/memory work runs in the issue slots the partner's MFMAs leave free.
// One barrier per chunk swaps the buffers.  NX: x-window rows a producer thread stages per chunk (ceil(xrows / 16)).
//
// PATCH (maps whose width is a multiple of 8): a chunk is an 8 x 8 pixel patch instead of 64 raster pixels.  Its window is
// the 10 x 10 patch around it (100 rows instead of 64 + 2 W + ... = 231 for an 80-column map), window row of patch pixel
// (jy, jx) is (jy+1)*10 + jx+1 and a tap is a shift by dy*10 + dx — no table, no pad columns (pixels outside the image are
// fetched as zeros) — and a producer thread's row offsets inside the patch are constants of the launch.
// X8 (the fp16 + e4m3 form of the forward kernel's `split8` mode, csrc/conv3x3.hip): g = (gh + gl) 2^-Eg, x = (xh + xl) 2^-Ex with fp16
// gh, xh and the scales from the tensors' range records; gh xh on the fp16 MFMA, both cross terms of a 64-pixel chunk on ONE block-scaled
// e4m3 MFMA (K = 128 = 64 pixels of e4m3(g) e4m3(xl 2^12) | 64 pixels of e4m3(gl 2^12) e4m3(x)): 64 matrix-pipe cycles per 16 x 16 x 64
// instead of the 96 of three bf16 MFMAs.  The hi planes keep their place in the buffers; each lo plane's room holds the two e4m3 planes
// ([row][64 channels] bytes), which the consumers fetch with ds_read_b64_tr_b8 (tools/tr8_check.hip: in a group of 16 lanes, byte k of
// lane d comes from the 8 bytes addressed by lane 2k + (d >> 3), byte d & 7 — so lanes 2k / 2k + 1 address channels 0-7 / 8-15 of
// pixel k, and every lane receives eight consecutive pixels of its own channel).
//
// PATCH == 2 (ROWS; narrow maps whose width is not a multiple of 8): a chunk is `ph` = 64 / W whole image rows (ph W <= 64 pixels, the
// rest of the 64 are dummies with g = 0).  Its window is the (ph + 2) x (W + 2) block around them — window row of chunk pixel j is
// (j / W + 1)(W + 2) + j % W + 1, a tap a shift by dy (W + 2) + dx — so, like the patches, no table and no carried coordinates:
// a producer thread's window positions are constants of the launch.  For the 20- and 10-column maps of the default model the
// raster form's producers (coordinate carries per fetched row, 114 / 97 window rows) were what the fp16 + e4m3 consumers waited for.
// XP (with X8): x is a PACKED activation tensor (dasv_common.hpp, store_x8 — what the split8 forward chain writes): a 16-byte piece of a
// pixel's 256-byte slab is already a piece of one of the three LDS planes, so the producers park the x window as fetched — no
// conversion, one 16-byte LDS store per piece (the same global addresses: a slab of 64 packed channels is as long as 64 floats).
// Measured with the pieces parked raw (tools/wgrad_diag.py, mode 3): 5.41 -> 4.34 ms over the seven layers of VGG4L/1024.
// GP (with XP): g is a packed tensor too (the pool's backward writes it that way, dasv_maxpool2x2_ceil_bwd_idx_x8_nhwc).
template <bool SPLIT, int NX, int PATCH, bool X8 = false, bool XP = false, bool GP = false>
__global__ void __launch_bounds__(512) conv3x3_wgrad_kernel(const WgradParams p) {
    static_assert(!XP || X8, "a packed x operand is a split8 operand");
    static_assert(!GP || XP, "packed g: the packed training chain (x packed as well)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int W = p.W, H = p.H, Cin = p.Cin, Cout = p.Cout;
    const int HW = H * W;
    constexpr size_t bufsz = wgrad_buffer_bytes(NX);
    constexpr size_t xplane = (size_t)(16 * NX) * 128;

    const int tiles = p.ntiles * p.ctiles;
#if DASV_WGRAD_XCD
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs, so the `tiles` workgroups that walk the SAME pixel
    // range (one g chunk is used by every input-channel tile, one x window by every output-channel tile) are given
    // consecutive slots of ONE XCD and share its L2
    const int per_xcd = (tiles * p.S + 7) >> 3;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int linear = xcd * per_xcd + slot;            // split-major: an XCD owns a contiguous range of (split, tile)
    if (slot >= per_xcd || linear >= tiles * p.S) return;
    const int split = linear / tiles, tile = linear - split * tiles;
#else
    const int tile = blockIdx.x % tiles, split = blockIdx.x / tiles;
#endif
    const int n0 = (tile / p.ctiles) * 64, c0 = (tile % p.ctiles) * 64;
    const int chunk_lo = split * p.chunks_per_split;
    const int total_chunks = p.B * p.chunks_per_image;
    const int chunk_hi = min(total_chunks, chunk_lo + p.chunks_per_split);
    const int nch = chunk_hi - chunk_lo;
    const int tid = threadIdx.x;
    // X8: power-of-two scales from the range records (wave-uniform): max |v| 2^E in [128, 256)
    float sg = 1.f, sx = 1.f, sout = 1.f;
    if constexpr (X8) {
        // (a packed operand was made with the exponent in its record)
        const int Eg = GP ? (int)p.g_rec[1] : x8_exp_for(p.g_rec[0]), Ex = XP ? (int)p.x_rec[1] : x8_exp_for(p.x_rec[0]);
        sg = x8_pow2(Eg);
        sx = x8_pow2(Ex);
        sout = x8_pow2(-Eg > -126 ? -Eg : -126) * x8_pow2(-Ex > -126 ? -Ex : -126);
    }

    if (tid >= 256) {
        // ------------------------------------------------------------------ producers
        const int ptid = tid - 256;
        const int scc = ptid & 15, srow = ptid >> 4;   // staging: 4-channel chunk, first row
        f32x4 gr[4], xr[NX];
        int w0_next = 0;
        // issue the global loads of chunk `ch` (rows beyond the utterance / outside the image become zeros).  The
        // producers share each SIMD's issue slots with a consumer's MFMAs, so the address arithmetic is kept to a few
        // 32-bit operations per load: per-thread offsets are precomputed and coordinates are carried, never divided.
        // Loads are raw buffer loads: a lane whose row is outside the utterance / image passes an offset beyond the
        // buffer and gets zeros from the hardware range check — no branches around the loads.
        const unsigned g_rowbytes = (unsigned)Cout * 4u, x_rowbytes = (unsigned)Cin * 4u;
        const unsigned g_bytes = (unsigned)p.B * (unsigned)HW * g_rowbytes, x_bytes = (unsigned)p.B * (unsigned)HW * x_rowbytes;
        const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.g), 0, g_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, x_bytes, 0x00020000);
        const unsigned g_thread = (unsigned)(n0 + scc * 4) * 4u + (unsigned)srow * g_rowbytes;
        const unsigned x_thread = (unsigned)(c0 + scc * 4) * 4u;
        const int a16 = 16 / (W + 1), r16 = 16 - a16 * (W + 1);   // 16 padded positions = a16 rows + r16 columns
        const unsigned step_bytes = (unsigned)(r16 + a16 * W) * x_rowbytes, w_bytes = (unsigned)W * x_rowbytes;
        // PATCH: this thread's G row inside the patch is (srow >> 3) + 2k, column srow & 7; its window rows are
        // r = srow + 16k -> (u, v) = (r / 10, r % 10), constants of the launch
        int pu[NX], pv[NX];
        unsigned poff[NX];
        if constexpr (PATCH == 1) {
#pragma unroll
            for (int k = 0; k < NX; ++k) {
                const int r = srow + 16 * k;
                pu[k] = r / 10;
                pv[k] = r - pu[k] * 10;
                poff[k] = (unsigned)(pu[k] * W + pv[k]) * x_rowbytes;
            }
        }
        if constexpr (PATCH == 2) {
#pragma unroll
            for (int k = 0; k < NX; ++k) {
                const int r = srow + 16 * k;
                pu[k] = r / (W + 2);
                pv[k] = r - pu[k] * (W + 2);
                poff[k] = (unsigned)(pu[k] * W + pv[k]) * x_rowbytes;
            }
        }
        const unsigned g_thread_patch = (unsigned)(n0 + scc * 4) * 4u + (unsigned)((srow >> 3) * W + (srow & 7)) * g_rowbytes;
        const int patches_per_row = W >> 3;
        // Chunks are fetched in order, so the chunk -> (utterance, row, column) decode is carried from one fetch to the next
        // (wave-uniform scalar adds) instead of being divided out again each time.
        int cb = chunk_lo / p.chunks_per_image;                 // utterance of the next chunk to fetch
        int cpi = chunk_lo - cb * p.chunks_per_image;           // its index inside the utterance
        int ch0 = PATCH == 2 ? cpi * p.ph : PATCH ? (cpi / patches_per_row) * 8 : (cpi * WKC) / W;                     // first row
        int cw0 = PATCH == 2 ? 0 : PATCH ? (cpi - (cpi / patches_per_row) * patches_per_row) * 8 : cpi * WKC - ((cpi * WKC) / W) * W;   // first column
        const int dh64 = WKC / W, dw64 = WKC - dh64 * W;        // raster: 64 pixels = dh64 rows + dw64 columns
        auto advance = [&]() {
            if (++cpi == p.chunks_per_image) { cpi = 0; ++cb; ch0 = 0; cw0 = 0; return; }
            if constexpr (PATCH == 2) {
                ch0 += p.ph;
            } else if constexpr (PATCH == 1) {
                cw0 += 8;
                if (cw0 >= W) { cw0 = 0; ch0 += 8; }
            } else {
                cw0 += dw64;
                ch0 += dh64;
                if (cw0 >= W) { cw0 -= W; ++ch0; }
            }
        };
        auto fetch = [&](int) {
            if constexpr (PATCH == 2) {
                const int b = cb, h0 = ch0;
                advance();
                const int npx = min(p.ph, H - h0) * W;            // real pixels of the chunk: whole rows, raster-contiguous
                const unsigned gbase = (unsigned)((b * H + h0) * W) * g_rowbytes + g_thread;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned o = srow + 16 * k < npx ? gbase + (unsigned)(16 * k) * g_rowbytes : g_bytes;
                    gr[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, (int)o, 0, 0));
                }
                // window origin = pixel (h0 - 1, -1)
                const unsigned xbase = (unsigned)((b * H + h0 - 1) * W - 1) * x_rowbytes + x_thread;
#pragma unroll
                for (int k = 0; k < NX; ++k) {
                    const bool ok = ((unsigned)(h0 - 1 + pu[k]) < (unsigned)H) & ((unsigned)(pv[k] - 1) < (unsigned)W) & (pu[k] < p.ph + 2);
                    xr[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, (int)(ok ? xbase + poff[k] : x_bytes), 0, 0));
                }
                return;
            }
            if constexpr (PATCH == 1) {
                const int b = cb, h0 = ch0, w0 = cw0;
                advance();
                const unsigned gbase = (unsigned)((b * H + h0) * W + w0) * g_rowbytes + g_thread_patch;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned o = h0 + (srow >> 3) + 2 * k < H ? gbase + (unsigned)(2 * k * W) * g_rowbytes : g_bytes;
                    gr[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, (int)o, 0, 0));
                }
                // window origin = pixel (h0 - 1, w0 - 1); offsets are taken mod 2^32 and only used where the pixel exists
                const unsigned xbase = (unsigned)((b * H + h0 - 1) * W + w0 - 1) * x_rowbytes + x_thread;
#pragma unroll
                for (int k = 0; k < NX; ++k) {
                    const bool ok = ((unsigned)(h0 - 1 + pu[k]) < (unsigned)H) & ((unsigned)(w0 - 1 + pv[k]) < (unsigned)W) & (pu[k] < 10);
                    xr[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, (int)(ok ? xbase + poff[k] : x_bytes), 0, 0));
                }
                return;
            }
            const int b = cb, off = cpi * WKC, h0 = ch0, w0 = cw0;
            const int npx = min(WKC, HW - off);
            advance();
            w0_next = w0;
            const unsigned gbase = (unsigned)(b * HW + off) * g_rowbytes + g_thread;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned o = srow + 16 * k < npx ? gbase + (unsigned)(16 * k) * g_rowbytes : g_bytes;
                gr[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, (int)o, 0, 0));
            }
            // padded coordinates: image rows have W+1 columns (column W is zero), images H+1 rows (row H is zero).
            // Window row 0 is the chunk's first pixel moved up one padded row and left one column; this thread starts
            // `srow` further and then steps by 16 padded positions with carries; positions before the first image
            // have bq < 0.
            int bq = b, hq = h0 - 1, wq = w0 - 1 + srow;
            if (wq < 0) { wq += W + 1; --hq; }
            while (wq >= W + 1) { wq -= W + 1; ++hq; }
            if (hq < 0) { hq += H + 1; --bq; }
            while (hq >= H + 1) { hq -= H + 1; ++bq; }
            // byte offset of (bq, hq, wq) carried along (mod 2^32; only used where the position is a real pixel): a column
            // carry skips one pad position, a row carry a pad row
            unsigned ob = (unsigned)((bq * H + hq) * W + wq) * x_rowbytes + x_thread;
            if (p.hcarries == 1) {     // every map with at least 16 / (W + 1) rows: at most one row carry per step
#pragma unroll
                for (int k = 0; k < NX; ++k) {
                    const bool ok = (wq < W) & (hq < H) & ((unsigned)bq < (unsigned)p.B);
                    xr[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, (int)(ok ? ob : x_bytes), 0, 0));
                    wq += r16;
                    const bool cw = wq >= W + 1;
                    wq -= cw ? W + 1 : 0;
                    hq += a16 + (cw ? 1 : 0);
                    const bool chh = hq >= H + 1;
                    hq -= chh ? H + 1 : 0;
                    bq += chh ? 1 : 0;
                    ob += step_bytes - (cw ? x_rowbytes : 0u) - (chh ? w_bytes : 0u);
                }
            } else {
#pragma unroll
                for (int k = 0; k < NX; ++k) {
                    const bool ok = (wq < W) & (hq < H) & ((unsigned)bq < (unsigned)p.B);
                    xr[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, (int)(ok ? ob : x_bytes), 0, 0));
                    wq += r16;
                    const bool cw = wq >= W + 1;
                    wq -= cw ? W + 1 : 0;
                    hq += a16 + (cw ? 1 : 0);
                    unsigned rows_skipped = 0;
                    for (int c = 0; c < p.hcarries; ++c) {      // degenerate maps (a handful of rows): several carries
                        const bool chh = hq >= H + 1;
                        hq -= chh ? H + 1 : 0;
                        bq += chh ? 1 : 0;
                        rows_skipped += chh ? w_bytes : 0u;
                    }
                    ob += step_bytes - (cw ? x_rowbytes : 0u) - rows_skipped;
                }
            }
        };
        // split and park the fetched chunk in LDS buffer `buf`
        auto park = [&](char* buf) {
            char* Ghi = buf;
            char* Glo = Ghi + WKC * 128;
            char* Xhi = Glo + WKC * 128;
            char* Xlo = Xhi + xplane;
            int* Ltab = reinterpret_cast<int*>(Xlo + xplane);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t h0_, l0_, h1_, l1_;
                const int o = wg_off(srow + 16 * k, scc >> 2) + (scc & 3) * 8;
                if constexpr (DASV_WGRAD_DIAG == 4) { *reinterpret_cast<f32x4*>(Ghi + (srow + 16 * k) * 256 + scc * 16) = gr[k]; continue; }
                if constexpr (GP) {           // piece scc of the pixel's slab: fp16 plane | e4m3(g) | e4m3(gl 2^12)  (see XP below)
                    const int row = srow + 16 * k;
                    const int op = scc < 8 ? wg_off(row, scc >> 1) + (scc & 1) * 16
                                           : WKC * 128 + (scc < 12 ? 0 : WKC * 64) + w8_off(row, scc & 3);
                    *reinterpret_cast<f32x4*>(Ghi + op) = gr[k];
                    continue;
                }
                if constexpr (X8) {
                    cvt4_x8(gr[k], sg, h0_, h1_, l0_, l1_);
                    const int o8 = w8_off(srow + 16 * k, scc >> 2) + (scc & 3) * 4;
                    *reinterpret_cast<uint32_t*>(Glo + o8) = l0_;                       // e4m3(g)
                    *reinterpret_cast<uint32_t*>(Glo + WKC * 64 + o8) = l1_;            // e4m3(gl 2^12)
                } else {
                    wsplit2(gr[k][0], gr[k][1], h0_, l0_);
                    wsplit2(gr[k][2], gr[k][3], h1_, l1_);
                    if (SPLIT) *reinterpret_cast<uint2*>(Glo + o) = make_uint2(l0_, l1_);
                }
                *reinterpret_cast<uint2*>(Ghi + o) = make_uint2(h0_, h1_);
            }
#pragma unroll
            for (int k = 0; k < NX; ++k) {
                uint32_t h0_, l0_, h1_, l1_;
                const int o = wg_off(srow + 16 * k, scc >> 2) + (scc & 3) * 8;
                if constexpr (DASV_WGRAD_DIAG == 3 || DASV_WGRAD_DIAG == 4) { *reinterpret_cast<f32x4*>(Xhi + (srow + 16 * k) * 256 + scc * 16) = xr[k]; continue; }
                if constexpr (XP) {
                    // piece scc of the slab: 0-7 the fp16 plane (8 channels each), 8-11 e4m3(x), 12-15 e4m3(xl 2^12) (16 channels each)
                    const int row = srow + 16 * k;
                    const int op = scc < 8 ? wg_off(row, scc >> 1) + (scc & 1) * 16
                                           : (int)xplane + (scc < 12 ? 0 : (int)(xplane / 2)) + w8_off(row, scc & 3);
                    *reinterpret_cast<f32x4*>(Xhi + op) = xr[k];
                    continue;
                }
                if constexpr (X8) {
                    cvt4_x8(xr[k], sx, h0_, h1_, l0_, l1_);
                    const int o8 = w8_off(srow + 16 * k, scc >> 2) + (scc & 3) * 4;
                    *reinterpret_cast<uint32_t*>(Xlo + o8) = l0_;                       // e4m3(x)
                    *reinterpret_cast<uint32_t*>(Xlo + xplane / 2 + o8) = l1_;          // e4m3(xl 2^12)
                } else {
                    wsplit2(xr[k][0], xr[k][1], h0_, l0_);
                    wsplit2(xr[k][2], xr[k][3], h1_, l1_);
                    if (SPLIT) *reinterpret_cast<uint2*>(Xlo + o) = make_uint2(l0_, l1_);
                }
                *reinterpret_cast<uint2*>(Xhi + o) = make_uint2(h0_, h1_);
            }
            // chunk pixel j -> window row (every image-row end crossed adds one pad position)
            if (PATCH == 0 && ptid < WKC) Ltab[ptid] = ptid + (int)(((unsigned)(w0_next + ptid) * (unsigned)p.magic_w) >> 16) + (W + 2);
        };

        if (nch > 0) {
            fetch(chunk_lo);
            park(smem);
            if (nch > 1) fetch(chunk_lo + 1);
        }
        __syncthreads();
        for (int i = 0; i < nch; ++i) {
            if (i + 1 < nch && (DASV_WGRAD_DIAG != 1 || i == 0)) {
                park(smem + ((i + 1) & 1) * bufsz);
                if (i + 2 < nch) fetch(chunk_lo + i + 2);     // in flight while the consumers work on chunk i+1
            }
            __syncthreads();
        }
        return;
    }

    // ---------------------------------------------------------------------- consumers
    // (s_setprio 3 for these waves against the staging wave on their SIMD was measured: no change)
    const int lane = tid & 63, wv = tid >> 6;   // wave = input-channel block
    const int grp = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    f32x4 acc[4][9];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // window row of chunk pixel j (PATCH 1: 10 x 10 window of an 8 x 8 patch; PATCH 2: (ph + 2) x (W + 2) window of ph whole rows, a dummy
    // pixel behind them borrows pixel 0's row — its g is zero; raster: the chunk's table)
    const int tstride = PATCH == 1 ? 10 : PATCH == 2 ? W + 2 : W + 1;
    auto wrow = [&](int j, const int* Ltab) {
        if constexpr (PATCH == 1) return (j >> 3) * 10 + (j & 7) + 11;
        else if constexpr (PATCH == 2) {
            const int jy = (int)(((unsigned)j * (unsigned)p.magic_w) >> 16);      // j / W (j < 64 <= 65536 / W)
            return j < p.ph * W ? (jy + 1) * (W + 2) + (j - jy * W) + 1 : W + 3;
        } else return Ltab[j];
    };
    __syncthreads();
    for (int ci = 0; ci < nch; ++ci) {
        const char* Ghi = smem + (ci & 1) * bufsz;
        const char* Glo = Ghi + WKC * 128;
        const char* Xhi = Glo + WKC * 128;
        const int* Ltab = reinterpret_cast<const int*>(Xhi + 2 * xplane);
        if constexpr (X8 && DASV_WGRAD_DIAG == 2) { __syncthreads(); continue; }     // timing build: producers alone
        // (Measured, tools/wgrad_diag.py with WGRAD_X8=1, seven layers: full 5.46 ms, consumers alone 4.00 ms — against 2.5 ms of
        // matrix-pipe time — producers alone 2.98 ms.  Fragments requested two taps ahead instead of one: consumers alone 0.70 ->
        // 0.71 ms on the patch layers (they are not waiting for LDS round trips) and 25 spilled registers in the whole-row variant.)
        if constexpr (X8) {
            typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
            typedef int i32x8 __attribute__((ext_vector_type(8)));
            typedef int i32x2 __attribute__((ext_vector_type(2)));
            typedef i32x2 __attribute__((address_space(3))) * lds2;
            // (window rows of the pixels this lane addresses in the e4m3 section: fetched here, ahead of the fp16 section, so that the table
            // read's latency is not in front of the first transposed reads there)
            const int s16 = lane & 15;
            const int pA = 16 * grp + (s16 >> 1), pB = pA + 8;
            const int rA = wrow(pA, Ltab), rB = wrow(pB, Ltab);
            // ---- gh xh: two K steps of 32 pixels on the fp16 MFMA (the bf16 loop below, one plane)
#pragma unroll
            for (int k0 = 0; k0 < WKC; k0 += 32) {
                const int j0 = k0 + 8 * grp + q, j1 = j0 + 4;
                const int col = 4 * pp * 2;
                bf16x8 gh[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) gh[i] = tr_frag(Ghi + wg_off(j0, i) + col, Ghi + wg_off(j1, i) + col);
                const int r0 = wrow(j0, Ltab), r1 = wrow(j1, Ltab);
                auto xfrag = [&](int t) {
                    const int toff = (t / 3 - 1) * tstride + (t % 3 - 1);
                    return tr_frag(Xhi + wg_off(r0 + toff, wv) + col, Xhi + wg_off(r1 + toff, wv) + col);
                };
                bf16x8 xh[2];
                xh[0] = xfrag(0);
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    if (t + 1 < 9) xh[(t + 1) & 1] = xfrag(t + 1);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, gh[i]), __builtin_bit_cast(f16x8, xh[t & 1]),
                                                                           acc[i][t], 0, 0, 0);
                }
            }
            // ---- both cross terms of the chunk: one K = 128 block-scaled e4m3 MFMA per tile.  A lane's registers 0-3 are pixels
            // 16 grp ... 16 grp + 15 of term one (e4m3(g) x e4m3(xl 2^12)), registers 4-7 the same pixels of term two
            // (e4m3(gl 2^12) x e4m3(x)); MX block b takes its E8M0 scale from lane group b: blocks 0, 1 are term one, 2, 3 term two
            {
                const int hb = (s16 & 1) * 8;                                // pA, pB: the pixels this lane ADDRESSES (see tr8 above)
                const char* G8 = Glo;                                        // e4m3(g) | e4m3(gl 2^12), [64][64] bytes each
                const char* X8v = Xhi + xplane;                              // e4m3(x) | e4m3(xl 2^12), [16 NX][64] bytes each
                auto tr8 = [&](const char* a0, const char* a1, const char* b0, const char* b1) {
                    const i32x2 u0 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds2)a0), u1 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds2)a1);
                    const i32x2 v0 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds2)b0), v1 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds2)b1);
                    return i32x8{u0[0], u0[1], u1[0], u1[1], v0[0], v0[1], v1[0], v1[1]};
                };
                i32x8 g8[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const char* a = G8 + hb;
                    const int oA = w8_off(pA, i), oB = w8_off(pB, i);
                    g8[i] = tr8(a + oA, a + oB, a + WKC * 64 + oA, a + WKC * 64 + oB);
                }
                const int gsc = grp < 2 ? 127 : 127 - X8_ACT_SHIFT, xsc = grp < 2 ? 127 - X8_ACT_SHIFT : 127;
                auto xfrag8 = [&](int t) {
                    const int toff = (t / 3 - 1) * tstride + (t % 3 - 1);
                    const char* a = X8v + hb;
                    const int oA = w8_off(rA + toff, wv), oB = w8_off(rB + toff, wv);
                    // term one wants the residual plane, term two the value plane
                    return tr8(a + xplane / 2 + oA, a + xplane / 2 + oB, a + oA, a + oB);
                };
                i32x8 x8[2];
                x8[0] = xfrag8(0);
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    if (t + 1 < 9) x8[(t + 1) & 1] = xfrag8(t + 1);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[i][t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(g8[i], x8[t & 1], acc[i][t], 0, 0, 0, gsc, 0, xsc);
                }
            }
            __syncthreads();
            continue;
        }
        // ---- MFMA: two K steps of 32 pixels; the fragments of tap t+1 are requested before tap t's MFMAs issue
#pragma unroll
        for (int k0 = 0; k0 < (DASV_WGRAD_DIAG == 2 ? 0 : WKC); k0 += 32) {
            const int j0 = k0 + 8 * grp + q, j1 = j0 + 4;            // this lane's rows of the two 4-row blocks
            const int col = 4 * pp * 2;                               // byte offset of this lane's 4 columns in a 16-col block
            bf16x8 gh[4], gl[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int o0 = wg_off(j0, i) + col, o1 = wg_off(j1, i) + col;
                gh[i] = tr_frag(Ghi + o0, Ghi + o1);
                if (SPLIT) gl[i] = tr_frag(Glo + o0, Glo + o1);
            }
            const int r0 = wrow(j0, Ltab), r1 = wrow(j1, Ltab);
            auto xfrag = [&](int t, bf16x8& xh, bf16x8& xl) {
                const int toff = (t / 3 - 1) * tstride + (t % 3 - 1);
                const char* x0 = Xhi + wg_off(r0 + toff, wv) + col;
                const char* x1 = Xhi + wg_off(r1 + toff, wv) + col;
                xh = tr_frag(x0, x1);
                if (SPLIT) xl = tr_frag(x0 + xplane, x1 + xplane);
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
                p.part[(((size_t)split * Cout + n) * 9 + t) * Cin + c0 + wv * 16 + (lane & 15)] = X8 ? acc[i][t][r] * sout : acc[i][t][r];
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

static void wgrad_plan(int B, int H, int W, int Cin, int Cout, WgradParams& p, bool x8) {
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.ctiles = Cin / 64; p.ntiles = Cout / 64;
#ifndef DASV_WGRAD_RASTER
#define DASV_WGRAD_RASTER 0   // timing builds only (tools/wgrad_diag.py): 1 forces raster chunks
#endif
    p.patch = (W % 8 == 0 && !DASV_WGRAD_RASTER) ? 1 : 0;
    // whole-row chunks: at least 3/4 of the 64 chunk positions real pixels, window within the widest staging variant
    p.ph = W <= WKC ? WKC / W : 0;
    // (measured, tools/_diag/wgrad_x8_ab.py: the fp16 + e4m3 consumers gain from it on every narrow layer — 38 x 10: -12 %, 75 x 20:
    // -24 % against raster chunks — the split-bf16 ones only where it does not add chunks: 75 x 20 -4 %, 38 x 10 (7 chunks per
    // utterance instead of 6) +6 %)
    const int rows_chunks = p.ph >= 1 ? (H + p.ph - 1) / p.ph : 0, raster_chunks = (H * W + WKC - 1) / WKC;
    if (!p.patch && !DASV_WGRAD_RASTER && p.ph >= 1 && p.ph * W * 4 >= WKC * 3 && (p.ph + 2) * (W + 2) <= 16 * 10 &&
        (x8 || rows_chunks * 16 <= raster_chunks * 17))
        p.patch = 2;
    p.chunks_per_image = p.patch == 2 ? (H + p.ph - 1) / p.ph : p.patch ? ((H + 7) / 8) * (W / 8) : (H * W + WKC - 1) / WKC;
    const int total = B * p.chunks_per_image, tiles = p.ctiles * p.ntiles;
    // workgroups per launch the pixel range is split for: one workgroup per CU is resident, so whole multiples of 256 avoid a
    // partial last round; measured sweep over the VGG4L/1024 layers (tools/_diag/wgrad_target_sweep.py, interleaved):
    // 256: 6.43  384: 7.92  512: 6.41 | 512: 6.23  768: 6.36  1024: 6.44  1536: 6.72  2048: 6.94 ms
#ifndef DASV_WGRAD_TARGET
#define DASV_WGRAD_TARGET 512   // (a compile-time constant: sweeps build their own objects, tools/_diag/wgrad_target_sweep.py)
#endif
    const int target = DASV_WGRAD_TARGET;
    int S = (target + tiles - 1) / tiles;
    if (S > total) S = total;
    if (S < 1) S = 1;
    p.chunks_per_split = (total + S - 1) / S;
    p.S = (total + p.chunks_per_split - 1) / p.chunks_per_split;
    p.xrows = p.patch == 2 ? (p.ph + 2) * (W + 2) : wgrad_xrows(W);
    // the per-lane division by W (window-row table) has numerators below W + 64 and W <= 80: numerator * W < 65536, where
    // the multiply-shift form is exact
    p.magic_w = 65536 / W + 1;
    p.hcarries = (16 / (W + 1) + 1 + H) / (H + 1);
}

}  // namespace dasv

extern "C" {

size_t dasv_conv3x3_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % 64 != 0 || Cout % 64 != 0) return 0;
    dasv::WgradParams p, q;     // (the chunk form, and with it the split count, may differ between the operand modes)
    dasv::wgrad_plan(B, H, W, Cin, Cout, p, false);
    dasv::wgrad_plan(B, H, W, Cin, Cout, q, true);
    return (size_t)(p.S > q.S ? p.S : q.S) * Cout * 9 * Cin * sizeof(float);
}

static int wgrad_launch(int mode, const float* x, const uint32_t* x_range, const float* g, const uint32_t* g_range, float* gw,
                        void* workspace, size_t workspace_bytes, int B, int H, int W, int Cin, int Cout, dasv_stream_t stream);

int dasv_conv3x3_wgrad_nhwc(int mode, const float* x, const float* g, float* gw, void* workspace, size_t workspace_bytes,
                            int B, int H, int W, int Cin, int Cout, dasv_stream_t stream) {
    if (mode != 0 && mode != 1) return DASV_E_DTYPE;
    return wgrad_launch(mode, x, nullptr, g, nullptr, gw, workspace, workspace_bytes, B, H, W, Cin, Cout, stream);
}

int dasv_conv3x3_wgrad_x8_nhwc(const float* x, const uint32_t* x_range, const float* g, const uint32_t* g_range, float* gw,
                               void* workspace, size_t workspace_bytes, int B, int H, int W, int Cin, int Cout, dasv_stream_t stream) {
    if (x_range == nullptr || g_range == nullptr) return DASV_E_NULL;     // the operands are scaled from their range records
    return wgrad_launch(3, x, x_range, g, g_range, gw, workspace, workspace_bytes, B, H, W, Cin, Cout, stream);
}

int dasv_conv3x3_wgrad_x8p_nhwc(const void* x_packed, const uint32_t* x_range, const void* g, int g_packed, const uint32_t* g_range, float* gw,
                                void* workspace, size_t workspace_bytes, int B, int H, int W, int Cin, int Cout, dasv_stream_t stream) {
    if (x_range == nullptr || g_range == nullptr) return DASV_E_NULL;
    return wgrad_launch(g_packed ? 5 : 4, static_cast<const float*>(x_packed), x_range, static_cast<const float*>(g), g_range, gw, workspace,
                        workspace_bytes, B, H, W, Cin, Cout, stream);
}

static int wgrad_launch(int mode, const float* x, const uint32_t* x_range, const float* g, const uint32_t* g_range, float* gw,
                        void* workspace, size_t workspace_bytes, int B, int H, int W, int Cin, int Cout, dasv_stream_t stream) {
    if (x == nullptr || g == nullptr || gw == nullptr || workspace == nullptr) return DASV_E_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % 64 != 0 || Cout % 64 != 0) return DASV_E_SHAPE;
    if ((long long)B * (H + 1) * (W + 1) > 0x7fffffffLL / 4) return DASV_E_SHAPE;
    // operands are addressed through 32-bit buffer offsets
    if ((long long)B * H * W * (Cin > Cout ? Cin : Cout) * 4 >= 0xffffffffLL) return DASV_E_SHAPE;
    dasv::WgradParams p;
    dasv::wgrad_plan(B, H, W, Cin, Cout, p, mode >= 3);
    if (workspace_bytes < (size_t)p.S * Cout * 9 * Cin * sizeof(float)) return DASV_E_WORKSPACE;
    p.x = x; p.g = g; p.part = static_cast<float*>(workspace);
    p.x_rec = x_range; p.g_rec = g_range;
    // two buffers; both planes are always carved (the bf16 mode simply leaves the lo planes unused)
    if (W > 80) return DASV_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)(DASV_WGRAD_XCD ? ((p.ctiles * p.ntiles * p.S + 7) / 8) * 8 : p.ctiles * p.ntiles * p.S)), block(512);
    const int nx = (p.xrows + 15) / 16;
#define DASV_LAUNCH_WGRAD(SP, NXV, PT, ...)                                                                            \
    do {                                                                                                           \
        const size_t lds = 2 * dasv::wgrad_buffer_bytes(NXV);                                                      \
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&dasv::conv3x3_wgrad_kernel<SP, NXV, PT __VA_OPT__(,) __VA_ARGS__>), \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                 \
        if (e2 != hipSuccess) return (int)e2;                                                                      \
        hipLaunchKernelGGL((dasv::conv3x3_wgrad_kernel<SP, NXV, PT __VA_OPT__(,) __VA_ARGS__>), grid, block, lds, s, p); \
    } while (0)
    if (nx > 15) return DASV_E_SHAPE;   // maps wider than 80 columns
    if (mode == 5) {       // split8, x and g packed
        if (p.patch == 1) DASV_LAUNCH_WGRAD(false, 7, 1, true, true, true);
        else if (p.patch == 2 && nx <= 7) DASV_LAUNCH_WGRAD(false, 7, 2, true, true, true);
        else if (p.patch == 2) DASV_LAUNCH_WGRAD(false, 10, 2, true, true, true);
        else if (nx <= 8) DASV_LAUNCH_WGRAD(false, 8, 0, true, true, true);
        else if (nx <= 10) DASV_LAUNCH_WGRAD(false, 10, 0, true, true, true);
        else DASV_LAUNCH_WGRAD(false, 15, 0, true, true, true);
    } else if (mode == 4) {       // split8, x packed
        if (p.patch == 1) DASV_LAUNCH_WGRAD(false, 7, 1, true, true);
        else if (p.patch == 2 && nx <= 7) DASV_LAUNCH_WGRAD(false, 7, 2, true, true);
        else if (p.patch == 2) DASV_LAUNCH_WGRAD(false, 10, 2, true, true);
        else if (nx <= 8) DASV_LAUNCH_WGRAD(false, 8, 0, true, true);
        else if (nx <= 10) DASV_LAUNCH_WGRAD(false, 10, 0, true, true);
        else DASV_LAUNCH_WGRAD(false, 15, 0, true, true);
    } else if (mode == 3) {
        if (p.patch == 1) DASV_LAUNCH_WGRAD(false, 7, 1, true);
        else if (p.patch == 2 && nx <= 7) DASV_LAUNCH_WGRAD(false, 7, 2, true);
        else if (p.patch == 2) DASV_LAUNCH_WGRAD(false, 10, 2, true);
        else if (nx <= 8) DASV_LAUNCH_WGRAD(false, 8, 0, true);
        else if (nx <= 10) DASV_LAUNCH_WGRAD(false, 10, 0, true);
        else DASV_LAUNCH_WGRAD(false, 15, 0, true);
    } else if (mode == 0) {
        if (p.patch == 1) DASV_LAUNCH_WGRAD(true, 7, 1);
        else if (p.patch == 2 && nx <= 7) DASV_LAUNCH_WGRAD(true, 7, 2);
        else if (p.patch == 2) DASV_LAUNCH_WGRAD(true, 10, 2);
        else if (nx <= 8) DASV_LAUNCH_WGRAD(true, 8, 0);
        else if (nx <= 10) DASV_LAUNCH_WGRAD(true, 10, 0);
        else DASV_LAUNCH_WGRAD(true, 15, 0);
    } else {
        if (p.patch == 1) DASV_LAUNCH_WGRAD(false, 7, 1);
        else if (p.patch == 2 && nx <= 7) DASV_LAUNCH_WGRAD(false, 7, 2);
        else if (p.patch == 2) DASV_LAUNCH_WGRAD(false, 10, 2);
        else if (nx <= 8) DASV_LAUNCH_WGRAD(false, 8, 0);
        else if (nx <= 10) DASV_LAUNCH_WGRAD(false, 10, 0);
        else DASV_LAUNCH_WGRAD(false, 15, 0);
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
