#!/usr/bin/env python3
"""Diagnostic: where the conv3x3 weight-gradient kernel's time goes.  Builds csrc/conv3x3_wgrad.hip four times
(-DDASV_WGRAD_DIAG=0..2: full / producers stage the first two chunks only / no MFMA section) into tools/_diag and
times each on the VGG4L/1024 layer shapes (DASV_WGRAD_RASTER=1 in the environment forces raster chunks).  `build` cross-compiles (no GPU needed); `run` needs the GPU."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIAG = os.path.join(ROOT, "tools", "_diag")
SRC = os.path.join(ROOT, "doubleattentionspeakerverification_amd", "csrc", "conv3x3_wgrad.hip")
NAMES = {0: "full", 1: "no re-staging", 2: "no MFMA", 3: "x parked raw", 4: "x and g parked raw"}


def so(n):
    return os.path.join(DIAG, "libwgrad_diag%d.so" % n)


def build():
    os.makedirs(DIAG, exist_ok=True)
    for n in NAMES:
        subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared",
                               "-DDASV_WGRAD_DIAG=%d" % n, SRC, "-o", so(n)])


def time_round_robin(gos, rounds=3, reps=4):
    """Best-of-`rounds` average launch time per variant, variants interleaved (clock ramps and box-to-box differences hit
    every variant alike), after a warm-up burst that brings the clocks up."""
    import torch
    for _ in range(3):
        for go in gos.values():
            go()
    torch.cuda.synchronize()
    best = {n: float("inf") for n in gos}
    for _ in range(rounds):
        for n, go in gos.items():
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(reps):
                go()
            e.record()
            torch.cuda.synchronize()
            best[n] = min(best[n], s.elapsed_time(e) / reps)
    return best


X8 = os.environ.get("WGRAD_X8") == "1"


def run():
    import torch
    sys.path.insert(0, ROOT)
    global hc
    from doubleattentionspeakerverification_amd import conv as hc
    dev = torch.device("cuda:0")
    layers = [("conv12", 128, 128, 300, 80), ("conv21", 128, 256, 150, 40), ("conv22", 256, 256, 150, 40),
              ("conv31", 256, 512, 75, 20), ("conv32", 512, 512, 75, 20), ("conv41", 512, 1024, 38, 10), ("conv42", 1024, 1024, 38, 10)]
    libs = {}
    for n in NAMES:
        lib = ctypes.CDLL(so(n))
        lib.dasv_conv3x3_wgrad_workspace_bytes.restype = ctypes.c_size_t
        lib.dasv_conv3x3_wgrad_workspace_bytes.argtypes = [ctypes.c_int] * 5
        lib.dasv_conv3x3_wgrad_nhwc.restype = ctypes.c_int
        lib.dasv_conv3x3_wgrad_nhwc.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_size_t] + [ctypes.c_int] * 5 + [ctypes.c_void_p]
        lib.dasv_conv3x3_wgrad_x8_nhwc.restype = ctypes.c_int
        lib.dasv_conv3x3_wgrad_x8_nhwc.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_size_t] + [ctypes.c_int] * 5 + [ctypes.c_void_p]
        libs[n] = lib
    B = 64
    tot = {n: 0.0 for n in NAMES}
    for name, cin, cout, H, W in layers:
        x = torch.relu(torch.randn(B, cin, H, W, device=dev)).contiguous(memory_format=torch.channels_last)
        g = torch.randn(B, cout, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        gw = torch.empty(cout, cin, 3, 3, device=dev)
        xr, gr = hc.range_of(x), hc.range_of(g)
        line = "%s %4d->%4d %3dx%2d |" % (name, cin, cout, H, W)
        gos = {}
        for n, lib in libs.items():
            nb = lib.dasv_conv3x3_wgrad_workspace_bytes(B, H, W, cin, cout)
            ws = torch.empty(nb, dtype=torch.uint8, device=dev)
            st = torch.cuda.current_stream().cuda_stream

            def go(lib=lib, ws=ws, nb=nb):
                if X8:      # WGRAD_X8=1 in the environment: the fp16 + e4m3 form
                    rc = lib.dasv_conv3x3_wgrad_x8_nhwc(x.data_ptr(), xr.data_ptr(), g.data_ptr(), gr.data_ptr(), gw.data_ptr(), ws.data_ptr(), nb,
                                                        B, H, W, cin, cout, st)
                else:
                    rc = lib.dasv_conv3x3_wgrad_nhwc(0, x.data_ptr(), g.data_ptr(), gw.data_ptr(), ws.data_ptr(), nb, B, H, W, cin, cout, st)
                assert rc == 0, rc
            gos[n] = go
        best = time_round_robin(gos)
        for n in NAMES:
            tot[n] += best[n]
            line += " %s %6.3f ms |" % (NAMES[n], best[n])
        print(line, flush=True)
    print("total: " + " | ".join("%s %.2f ms" % (NAMES[n], tot[n]) for n in NAMES))


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
