#!/usr/bin/env python3
"""Where a tile's time goes in the split8 window kernel (packed input): builds csrc/conv3x3.hip with -DDASV_CONV_STAMPS=1 into
tools/_diag/libconv_stamps.so (`build`, no GPU needed) and prints, per VGG4L/1024 layer, the phase boundaries that thread 0 of the
launch's middle workgroup stamped (100 MHz clock) and the launch time divided by the rounds of workgroups (`run`)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "_diag", "libconv_stamps.so")
SRC = os.path.join(ROOT, "doubleattentionspeakerverification_amd", "csrc", "conv3x3.hip")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-shared", "-DDASV_CONV_STAMPS=1", SRC, "-o", SO])
    sys.exit(0)
import torch
sys.path.insert(0, ROOT)
from doubleattentionspeakerverification_amd import conv as hc
lib = ctypes.CDLL(SO)
vp, ci, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
lib.dasv_conv3x3_bias_relu_x8_nhwc.restype = ci
lib.dasv_conv3x3_bias_relu_x8_nhwc.argtypes = [vp, ci, vp, vp, vp, vp, vp, ci, vp, vp, sz, ci, ci, ci, ci, ci, ci, vp]
lib.dasv_debug_conv_stamps.restype = ci
lib.dasv_debug_conv_stamps.argtypes = [vp]
dev = torch.device("cuda:0")
B = 64
layers = [("conv12", 128, 128, 300, 80), ("conv21", 128, 256, 150, 40), ("conv22", 256, 256, 150, 40),
          ("conv31", 256, 512, 75, 20), ("conv32", 512, 512, 75, 20), ("conv41", 512, 1024, 38, 10), ("conv42", 1024, 1024, 38, 10)]
stream = torch.cuda.current_stream().cuda_stream
for name, cin, cout, H, W in layers:
    xf = torch.relu(torch.randn(B, cin, 2 * H, 2 * W, device=dev)).contiguous(memory_format=torch.channels_last)
    recs = hc.new_ranges(dev, 3)
    recs[0] = hc.range_of(xf)
    x = hc.maxpool2x2_ceil(xf, pack=True, x_rec=recs[0], y_rec=recs[1])
    del xf
    w = torch.randn(cout, cin, 3, 3, device=dev) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout, device=dev) * 0.1
    hi, lo = hc.PackedConvWeights().get(w, "split8", b)
    y = torch.zeros(B, H, W, cout, device=dev)

    def run():
        assert lib.dasv_conv3x3_bias_relu_x8_nhwc(x.data_ptr(), 1, recs[1].data_ptr(), hi.data_ptr(), lo.data_ptr(), b.data_ptr(), y.data_ptr(), 1,
                                                  recs[2].data_ptr(), None, 0, B, H, W, cin, cout, 1, stream) == 0
    for _ in range(3): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): run()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    st = (ctypes.c_ulonglong * 128)()
    assert lib.dasv_debug_conv_stamps(st) == 0
    t = [int(v) for v in st]
    tbm = 192 if W > 40 else 256
    tiles = ((B * H * W + tbm - 1) // tbm) * (cout // 128)
    rounds = tiles / 256.0
    us = lambda a, b_: (t[b_] - t[a]) / 100.0
    nslab = cin // 64
    line = "%s: launch %.3f ms / %.2f rounds = %.1f us per tile | setup %.2f" % (name, ms, rounds, ms * 1e3 / rounds, us(0, 1))
    for sl in range(min(nslab, 3)):
        k = 2 + 4 * sl
        line += " | slab%d: load+wait %.2f store+sync %.2f taps %.2f" % (sl, us(k, k + 1), us(k + 1, k + 2), us(k + 2, k + 3))
    last = 2 + 4 * (nslab - 1) + 3 if nslab <= 14 else None
    line += " | loop total %.2f | epi: wait %.2f image %.2f stores %.2f | tile total %.2f" % (us(1, 60), us(60, 61), us(61, 62), us(62, 63), us(0, 63))
    # the clock the CU held over the tile: shader cycles (slots 64+) per 100 MHz tick; and the matrix pipe's share of those cycles
    # (two waves per SIMD, each IB x 4 tiles x (2 x 16 + 32) cycles per tap)
    cyc = t[64 + 63] - t[64 + 0]
    ghz = cyc / max(1, t[63] - t[0]) * 0.1
    mfma = 2 * (tbm // 64) * 4 * 64 * 9 * nslab
    line += " | clock %.2f GHz, tile %d cycles, matrix pipe %d = %.0f %%" % (ghz, cyc, mfma, 100.0 * mfma / cyc)
    lc = t[64 + 60] - t[64 + 1]
    line += " (loop alone %d cycles: %.0f %%)" % (lc, 100.0 * mfma / max(1, lc))
    print(line, flush=True)
