#!/usr/bin/env python3
"""Diagnostic: where the conv3x3 forward / data-gradient window kernel's time goes.  Builds csrc/conv3x3.hip four times
(-DDASV_CONV_DIAG=0..3: full / window staged for the first slab only / no MFMA section / weight tile staged once) into tools/_diag and
times each on the VGG4L/1024 layer shapes.  `build` cross-compiles (no GPU needed); `run` needs the GPU."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIAG = os.path.join(ROOT, "tools", "_diag")
SRC = os.path.join(ROOT, "doubleattentionspeakerverification_amd", "csrc", "conv3x3.hip")
NAMES = {0: "full", 1: "window staged once", 2: "no MFMA", 10: "full, no scheduling hint"}


def so(n):
    return os.path.join(DIAG, "libconv_diag%d.so" % n)


def build():
    os.makedirs(DIAG, exist_ok=True)
    for n in NAMES:
        subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared",
                               "-DDASV_CONV_DIAG=%d" % (n % 10), "-DDASV_CONV_SCHED=%d" % (1 - n // 10), SRC, "-o", so(n)])


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


def run():
    import torch
    dev = torch.device("cuda:0")
    layers = [("conv12", 128, 128, 300, 80), ("conv21", 128, 256, 150, 40), ("conv22", 256, 256, 150, 40),
              ("conv31", 256, 512, 75, 20), ("conv32", 512, 512, 75, 20), ("conv41", 512, 1024, 38, 10), ("conv42", 1024, 1024, 38, 10)]
    libs = {}
    ci, vp, cs = ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t
    for n in NAMES:
        lib = ctypes.CDLL(so(n))
        lib.dasv_conv3x3_packed_weight_bytes.restype = cs
        lib.dasv_conv3x3_packed_weight_bytes.argtypes = [ci, ci]
        lib.dasv_conv3x3_pack_weights.restype = ci
        lib.dasv_conv3x3_pack_weights.argtypes = [vp, vp, vp, ci, ci, vp]
        lib.dasv_conv3x3_bias_relu_nhwc.restype = ci
        lib.dasv_conv3x3_bias_relu_nhwc.argtypes = [ci, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]
        libs[n] = lib
    B = 64
    mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    tot = {n: 0.0 for n in NAMES}
    for name, cin, cout, H, W in layers:
        x = torch.relu(torch.randn(B, cin, H, W, device=dev)).contiguous(memory_format=torch.channels_last)
        w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
        bias = torch.zeros(cout, device=dev)
        y = torch.empty(B, cout, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        line = "%s %4d->%4d %3dx%2d |" % (name, cin, cout, H, W)
        gos = {}
        for n, lib in libs.items():
            nb = lib.dasv_conv3x3_packed_weight_bytes(cout, cin)
            hi, lo = torch.empty(nb, dtype=torch.uint8, device=dev), torch.empty(nb, dtype=torch.uint8, device=dev)
            st = torch.cuda.current_stream().cuda_stream
            assert lib.dasv_conv3x3_pack_weights(w.data_ptr(), hi.data_ptr(), lo.data_ptr(), cout, cin, st) == 0

            def go(lib=lib, hi=hi, lo=lo):
                rc = lib.dasv_conv3x3_bias_relu_nhwc(mode, x.data_ptr(), hi.data_ptr(), lo.data_ptr(), bias.data_ptr(), y.data_ptr(),
                                                     B, H, W, cin, cout, 1, st)
                assert rc == 0, rc
            gos[n] = go
        best = time_round_robin(gos)
        for n in NAMES:
            tot[n] += best[n]
            line += " %s %6.3f ms |" % (NAMES[n], best[n])
        print(line, flush=True)
    print("total: " + " | ".join("%s %.2f ms" % (NAMES[n], tot[n]) for n in NAMES))


if __name__ == "__main__":
    build() if sys.argv[1:2] == ["build"] else run()
