#!/usr/bin/env python3
"""CPU error model for the next operand mode of the convolution: the cross terms of split8 on
block-scaled e2m3 (fp6) instead of e4m3 operands.  Emulates, in float64 with exact minifloat rounding, one GEMM of the conv's
shape statistics (K = 9 * Cin products per output; activations relu(N(0,1)), He-initialised weights) and prints the error of
each scheme against the exact product, relative to the output scale (max |y|):

  split    bf16 hi/lo, three products (hi*hi + hi*lo + lo*hi)
  split8   fp16 hi*hi + e4m3(x) * e4m3(r_w * 2^Sw[n]) + e4m3(r_x * 2^12) * e4m3(w * 2^Sh[n])     (what runs today)
  split6   fp16 hi*hi + the same two cross terms on e2m3 with one power-of-two scale per 32-channel block and operand
  fp16     fp16 hi*hi only

No GPU needed.  python tools/split6_error_model.py [Cin]"""
import sys
import numpy as np


def minifloat(x, ebits, mbits, bias, maxval):
    """Round-to-nearest-even onto a sign/exponent/mantissa grid with subnormals, saturating at maxval."""
    x = np.asarray(x, dtype=np.float64)
    a = np.minimum(np.abs(x), maxval)
    e = np.floor(np.log2(np.maximum(a, 1e-300)))
    e = np.maximum(e, 1 - bias)                      # subnormal range shares the smallest normal exponent
    step = 2.0 ** (e - mbits)
    q = np.round(a / step) * step                    # np.round is round-half-even
    return np.sign(x) * np.minimum(q, maxval)


e4m3 = lambda v: minifloat(v, 4, 3, 7, 448.0)
e2m3 = lambda v: minifloat(v, 2, 3, 1, 7.5)
fp16 = lambda v: np.asarray(v, dtype=np.float64).astype(np.float16).astype(np.float64)


def bf16(v):
    u = np.asarray(v, dtype=np.float32).view(np.uint32)
    u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000
    return u.view(np.float32).astype(np.float64)


def block_scaled(v, quant, top, block=32):
    """Per row and per `block` consecutive columns: power-of-two scale so that the block maximum lands in (top/2, top]."""
    r, k = v.shape
    b = v.reshape(r, k // block, block)
    m = np.abs(b).max(axis=2, keepdims=True)
    s = np.where(m > 0, 2.0 ** np.ceil(np.log2(np.maximum(m, 1e-300) / top)), 1.0)
    return (quant(b / s) * s).reshape(r, k)


def main():
    cin = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    K, M, N = 9 * cin, 384, 128
    g = np.random.RandomState(0)
    for name, x in (("relu(N(0,1)) activations", np.maximum(g.standard_normal((M, K)), 0)),
                    ("heavy-tailed activations (relu(N)^3)", np.maximum(g.standard_normal((M, K)), 0) ** 3)):
        w = g.standard_normal((N, K)) * np.sqrt(2.0 / K)
        x = x.astype(np.float32).astype(np.float64)
        w = w.astype(np.float32).astype(np.float64)
        y = x @ w.T
        scale = np.abs(y).max()
        err = lambda z: np.abs(z - y).max() / scale
        # split-bf16
        xh, wh = bf16(x), bf16(w)
        xl, wl = bf16(x - xh), bf16(w - wh)
        r_split = err(xh @ wh.T + xh @ wl.T + xl @ wh.T)
        # fp16 hi parts and residuals
        hx, hw = fp16(x), fp16(w)
        rx, rw = x - hx, w - hw
        r_fp16 = err(hx @ hw.T)
        # split8: fixed activation residual scale 2^12, per-output-channel weight scales (max * 2^S in (64, 128])
        sh = 2.0 ** (7 - np.ceil(np.log2(np.abs(w).max(axis=1, keepdims=True))))
        sw = 2.0 ** (7 - np.ceil(np.log2(np.maximum(np.abs(rw).max(axis=1, keepdims=True), 1e-300))))
        c8 = e4m3(x) @ (e4m3(rw * sw) / sw).T + (e4m3(rx * 4096.0) / 4096.0) @ (e4m3(w * sh) / sh).T
        r_x8 = err(hx @ hw.T + c8)
        # split6: e2m3 with a scale per (row, 32-channel block) for all four cross-term operands
        q = lambda v: block_scaled(v, e2m3, 7.5)
        c6 = q(x) @ q(rw).T + q(rx) @ q(w).T
        r_x6 = err(hx @ hw.T + c6)
        # split6 with e4m3-like headroom variant: top of the block at 6 (keeps 4 bits for elements down to max/12)
        q2 = lambda v: block_scaled(v, e2m3, 6.0)
        r_x6b = err(hx @ hw.T + q2(x) @ q2(rw).T + q2(rx) @ q2(w).T)
        print("%s, Cin %d (K = %d):" % (name, cin, K))
        print("   split-bf16 %.2e | split8 (e4m3) %.2e | split6 (e2m3, block max -> 7.5) %.2e | (-> 6.0) %.2e | fp16 only %.2e"
              % (r_split, r_x8, r_x6, r_x6b, r_fp16))


if __name__ == "__main__":
    main()
