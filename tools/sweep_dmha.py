#!/usr/bin/env python3
"""Time the fused DoubleMHA forward/backward launches on one MI355X over workgroup-shape knobs.

Device-side timing: events on the launch stream around N back-to-back launches, rotating over enough
distinct input buffers (> 256 MiB in total) that the Infinity Cache cannot serve the re-reads.
Usage: python tools/sweep_dmha.py [--shape B,T,D,H] [--dtype f32|bf16|f16] [--bwd] [--quick]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from doubleattentionspeakerverification_amd import _lib, poolings  # noqa: E402


def algorithmic_bytes(B, T, D, H, esize, with_align=True):
    dh = D // H
    return esize * B * T * D + (4 * B * T * H if with_align else 0) + 4 * B * dh + 4 * (dh * H + dh)


def time_launches(fn, nbuf, iters, reps=5):
    torch.cuda.synchronize()
    for i in range(nbuf):
        fn(i)
    torch.cuda.synchronize()
    best = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(iters):
            fn(i % nbuf)
        e.record()
        torch.cuda.synchronize()
        best.append(s.elapsed_time(e) * 1e3 / iters)
    return float(np.median(best)), float(np.min(best))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="64,300,512,8")
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--bwd", action="store_true")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--no-align", action="store_true")
    ap.add_argument("--nbuf", type=int, default=0, help="number of rotating input buffers (default: enough for > 300 MB)")
    ap.add_argument("--no-att", action="store_true", help="per-head stage only (no in-launch head exchange)")
    ap.add_argument("--configs", default="", help="semicolon list of nw,wpu,pad,rows tuples to time instead of the default sweep")
    args = ap.parse_args()
    B, T, D, H = (int(v) for v in args.shape.split(","))
    dh = D // H
    dt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[args.dtype]
    esize = 4 if dt == torch.float32 else 2
    dev = torch.device("cuda:0")
    lib = _lib.debug_probe()       # the developer build (libdasv_hip_dbg.so): the same kernels plus dasv_debug_set_tuning
    nbuf = args.nbuf if args.nbuf > 0 else max(8, int(np.ceil(300e6 / (B * T * D * esize))))
    g = torch.Generator(device=dev).manual_seed(1234)
    xs = [torch.randn(B, T, D, device=dev, generator=g).to(dt) for _ in range(nbuf)]
    q = torch.randn(dh, H, device=dev, generator=g) * (2.0 / (dh + H)) ** 0.5
    a = torch.randn(dh, 1, device=dev, generator=g) * (2.0 / (dh + 1)) ** 0.5
    abytes = algorithmic_bytes(B, T, D, H, esize, not args.no_align)
    print("shape B=%d T=%d D=%d H=%d dh=%d dtype=%s  nbuf=%d (%.0f MB)  algorithmic bytes/launch=%d" %
          (B, T, D, H, dh, args.dtype, nbuf, nbuf * B * T * D * esize / 1e6, abytes))

    # raw launch through the C ABI with preallocated outputs (no allocator noise)
    out = torch.empty(B, dh, device=dev)
    align = None if args.no_align else torch.empty(B, T, H, device=dev)
    hw = torch.empty(B, H, device=dev)
    ctx = torch.empty(B, H, dh, device=dev)
    ml = torch.empty(B, H, 2, device=dev)
    ws = torch.zeros(max(4096, lib.dasv_dmha_workspace_bytes(B, T, H, dh)), dtype=torch.uint8, device=dev)
    code = _lib.DTYPE_CODES[str(dt).split(".")[-1]]
    stream = torch.cuda.current_stream().cuda_stream
    ap_ = None if align is None else align.data_ptr()
    att_p = None if args.no_att else a.data_ptr()

    def fwd(i):
        rc = lib.dasv_dmha_fwd(code, xs[i].data_ptr(), q.data_ptr(), att_p, None, None, out.data_ptr(), ap_,
                               hw.data_ptr(), ctx.data_ptr(), ml.data_ptr(), ws.data_ptr(), ws.numel(), B, T, H, dh,
                               T * D, D, stream)
        assert rc == 0, rc

    configs = [(0, 0, 0, 0)]
    if args.configs:
        configs = [tuple(int(v) for v in c.split(",")) for c in args.configs.split(";")]
    elif not args.quick:
        configs += [(nw, wpu, 0, r) for nw in (2, 4, 8) for wpu in (1, 2, 4, 8) if wpu <= nw for r in (2, 4, 8)]
    print("%-28s %10s %10s %9s %9s" % ("config (nw,wpu,ldspad,rows)", "med us", "min us", "GB/s med", "%of 8TB/s"))
    for nw, wpu, pad, rows in configs:
        lib.dasv_debug_set_tuning(nw, wpu, pad, 0, rows)
        try:
            med, mn = time_launches(fwd, nbuf, args.iters)
        except AssertionError as e:
            print("(%d,%d,%d,%d) failed rc=%s" % (nw, wpu, pad, rows, e))
            continue
        var = lib.dasv_dmha_last_variant().decode()
        print("%-30s %10.2f %10.2f %9.0f %8.1f%%   %s" % ((nw, wpu, pad, rows), med, mn, abytes / med / 1e3, abytes / med / 1e3 / 80.0, var))
    lib.dasv_debug_set_tuning(0, 0, 0, 0, 0)

    if args.bwd:
        go = torch.randn(B, dh, device=dev, generator=g)
        gx = torch.empty_like(xs[0])
        gq = torch.empty(dh, H, device=dev)
        ga = torch.empty(dh, device=dev)
        wsb = torch.empty(lib.dasv_dmha_bwd_workspace_bytes(B, T, H, dh), dtype=torch.uint8, device=dev)
        fwd(0)

        def bwd(i):
            rc = lib.dasv_dmha_bwd(code, xs[i].data_ptr(), q.data_ptr(), a.data_ptr(), None, None, hw.data_ptr(),
                                   ctx.data_ptr(), ml.data_ptr(), out.data_ptr(), go.data_ptr(), gx.data_ptr(),
                                   gq.data_ptr(), ga.data_ptr(), wsb.data_ptr(), wsb.numel(), B, T, H, dh, T * D, D, stream)
            assert rc == 0, rc
        med, mn = time_launches(bwd, nbuf, args.iters)
        bb = 2 * esize * B * T * D
        print("backward (2 launches): med %.2f us  min %.2f us  %.0f GB/s (%.1f%% of 8 TB/s) on %d algorithmic bytes" %
              (med, mn, bb / med / 1e3, bb / med / 1e3 / 80.0, bb))

    # reference point: plain device-to-device copy of one input buffer (read+write) and a read-only reduction
    y = torch.empty_like(xs[0])
    med, mn = time_launches(lambda i: y.copy_(xs[i]), nbuf, args.iters)
    cb = 2 * B * T * D * esize
    print("torch copy_ (read+write %d B): med %.2f us -> %.0f GB/s" % (cb, med, cb / med / 1e3))
    med, mn = time_launches(lambda i: torch.sum(xs[i]), nbuf, args.iters)
    print("torch sum (read %d B): med %.2f us -> %.0f GB/s" % (cb // 2, med, cb / 2 / med / 1e3))


if __name__ == "__main__":
    main()
