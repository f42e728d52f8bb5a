#!/usr/bin/env python3
"""Diagnostic: per-workgroup phase timeline of the fused forward kernel (separate -DDASV_STAMPS build;
the product library contains no stamp code).  Build: python tools/stamps_dmha.py --build  (CPU box is fine).
Run on the GPU box: python tools/stamps_dmha.py [--shape B,T,D,H] [--tuning nw,wpu,pad]"""
import argparse
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DIAG = os.path.join(ROOT, "tools", "_diag")
SO = os.path.join(DIAG, "libdasv_diag.so")


def build():
    os.makedirs(DIAG, exist_ok=True)
    src = os.path.join(ROOT, "doubleattentionspeakerverification_amd", "csrc", "dmha_fwd.hip")
    subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-DDASV_STAMPS", "-DDASV_DEBUG",
                           "-o", SO, src])
    print("built", SO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--shape", default="64,300,512,8")
    ap.add_argument("--tuning", default="0,0,0,0,0", help="nw,wpu,ldspad,rows,force_generic")
    ap.add_argument("--nhwc", action="store_true", help="the front-end's NHWC map [B,T,5,C] pooled in place (D = 5 C)")
    ap.add_argument("--no-att", action="store_true")
    ap.add_argument("--no-align", action="store_true")
    ap.add_argument("--nbuf", type=int, default=8)
    args = ap.parse_args()
    if args.build:
        build()
        return
    import torch
    lib = ctypes.CDLL(SO)
    lib.dasv_dmha_fwd.restype = ctypes.c_int
    vp, ci, c64, csz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t
    lib.dasv_dmha_fwd.argtypes = [ci] + [vp] * 10 + [vp, csz, ci, ci, ci, ci, c64, c64, vp]
    lib.dasv_debug_set_stamps.argtypes = [vp]
    lib.dasv_debug_set_tuning.argtypes = [ci] * 5
    lib.dasv_dmha_last_variant.restype = ctypes.c_char_p
    B, T, D, H = (int(v) for v in args.shape.split(","))
    dh = D // H
    dev = torch.device("cuda:0")
    nbuf = args.nbuf
    xs = [torch.randn(B, T, D, device=dev) for _ in range(nbuf)]
    q = torch.randn(dh, H, device=dev) * 0.2
    a = torch.randn(dh, device=dev) * 0.2
    out, hw = torch.empty(B, dh, device=dev), torch.empty(B, H, device=dev)
    align = torch.empty(B, T, H, device=dev)
    ctx, ml = torch.empty(B, H, dh, device=dev), torch.empty(B, H, 2, device=dev)
    ws = torch.zeros(1 << 26, dtype=torch.uint8, device=dev)
    nwg_max = 8 * ((B + 7) // 8) * H
    stamps = torch.zeros(2 * nwg_max, 8, dtype=torch.int64, device=dev)  # [grid][8] phase stamps, then [grid][8] per-wave loop-done
    tn = [int(v) for v in args.tuning.split(",")] + [0]
    lib.dasv_debug_set_tuning(tn[0], tn[1], tn[2], tn[4], tn[3])
    if args.nhwc:
        lib.dasv_dmha_fwd_nhwc_f32.restype = ctypes.c_int
        lib.dasv_dmha_fwd_nhwc_f32.argtypes = [vp] * 10 + [vp, csz, ci, ci, ci, ci, ci, vp]
    GRID = [0]
    stream = torch.cuda.current_stream().cuda_stream

    def launch(i, st):
        lib.dasv_debug_set_stamps(st)
        if args.nhwc:
            rc = lib.dasv_dmha_fwd_nhwc_f32(xs[i % nbuf].data_ptr(), q.data_ptr(), a.data_ptr(), None, None, out.data_ptr(),
                                            None if args.no_align else align.data_ptr(), hw.data_ptr(), None, None,
                                            ws.data_ptr(), ws.numel(), B, T, 5, D // 5, H, stream)
        else:
            rc = lib.dasv_dmha_fwd(0, xs[i % nbuf].data_ptr(), q.data_ptr(), None if args.no_att else a.data_ptr(), None, None,
                               out.data_ptr(), None if args.no_align else align.data_ptr(), hw.data_ptr(), ctx.data_ptr(),
                               ml.data_ptr(), ws.data_ptr(), ws.numel(), B, T, H, dh, T * D, D, stream)
        assert rc == 0, rc
        import re
        mm = re.search(r"upw=(\d+)", lib.dasv_dmha_last_variant().decode())
        GRID[0] = 8 * ((B + 7) // 8) * (H // int(mm.group(1))) if mm else nwg_max
    for i in range(20):
        launch(i, None)
    torch.cuda.synchronize()
    names = ["start", "pass0 done", "loop done", "unit done", "published", "ticket known", "head stage done"]
    agg, ivals, skew = [], [], []
    for rep in range(7):
        stamps.zero_()
        for i in range(8):          # keep the chip in steady state, stamp the last launch only
            launch(rep * 8 + i, None)
        launch(rep * 8 + 8, stamps.data_ptr())
        torch.cuda.synchronize()
        sall = stamps.cpu().numpy().astype(np.float64)
        nblk = GRID[0]
        s = sall[:nblk]
        wv = sall[nblk:2 * nblk]
        used = s[:, 0] > 0
        wv = wv[used]
        s = s[used]
        wmax = wv.max(axis=1)
        wmin = np.where(wv > 0, wv, np.inf).min(axis=1)
        skew.append((np.median(wmax - wmin) * 0.01, np.median(wmax - s[:, 2]) * 0.01, np.median(s[:, 3] - wmax) * 0.01,
                     np.median(s[:, 7] - wmax) * 0.01, np.median(s[:, 3] - s[:, 7]) * 0.01))
        t0 = s[:, 0].min()
        rows = []
        for k in range(7):
            col = s[:, k]
            col = col[col > 0]
            if len(col) == 0:
                rows.append((np.nan,) * 4)
                continue
            rel = (col - t0) * 0.01  # 100 MHz -> us
            rows.append((rel.min(), np.median(rel), rel.max(), len(col)))
        agg.append(rows)
        iv = []
        for k in range(6):
            ok = (s[:, k] > 0) & (s[:, k + 1] > 0)
            d = (s[ok, k + 1] - s[ok, k]) * 0.01
            iv.append((np.percentile(d, 10), np.median(d), np.percentile(d, 90), d.max(), ok.sum()) if ok.any() else (np.nan,) * 5)
        ivals.append(iv)
    print("variant:", lib.dasv_dmha_last_variant().decode(), " workgroups:", int(used.sum()))
    print("%-18s %8s %8s %8s %6s   (us since first workgroup start; median over stamped launches)" % ("phase", "min", "median", "max", "n"))
    agg = np.array(agg, dtype=np.float64)
    for k, nm in enumerate(names):
        n = agg[0, k, 3]
        print("%-18s %8.2f %8.2f %8.2f %6d" % (nm, np.median(agg[:, k, 0]), np.median(agg[:, k, 1]), np.median(agg[:, k, 2]), 0 if np.isnan(n) else int(n)))
    sk = np.median(np.array(skew), axis=0)
    print("per-wave loop-done: median spread within a workgroup %.2f us; slowest wave after wave 0: %.2f us; slowest wave -> unit done: %.2f us "
          "(of which slowest wave -> partials parked + barrier %.2f, barrier -> unit done %.2f)" % tuple(sk))
    ivals = np.array(ivals, dtype=np.float64)
    print("per-workgroup intervals (us): p10 / median / p90 / max / n")
    for k in range(6):
        v = np.nanmedian(ivals[:, k, :], axis=0)
        print("  %-16s -> %-16s %6.2f %6.2f %6.2f %6.2f %5.0f" % (names[k], names[k + 1], v[0], v[1], v[2], v[3], v[4]))


if __name__ == "__main__":
    main()
