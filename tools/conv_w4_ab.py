#!/usr/bin/env python3
"""Same-box A/B of the packed-input split8 convolution: another build of csrc/conv3x3.hip (default tools/_diag/ab/libconv_a.so)
against the in-tree library, both through dasv_conv3x3_bias_relu_x8_nhwc on the same packed input, interleaved rounds.
Prints per layer the best time of each and whether the outputs (fp32 and packed) are bit-identical."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from doubleattentionspeakerverification_amd import _lib
from doubleattentionspeakerverification_amd import conv as hc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
other = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tools", "_diag", "ab", "libconv_a.so")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
libs = [("A", ctypes.CDLL(other)), ("B", _lib.load())]
vp, ci, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
for _, l in libs:
    l.dasv_conv3x3_bias_relu_x8_nhwc.restype = ci
    l.dasv_conv3x3_bias_relu_x8_nhwc.argtypes = [vp, ci, vp, vp, vp, vp, ci, vp, sz, ci, ci, ci, ci, ci, ci, vp]
dev = torch.device("cuda:0")
torch.manual_seed(0)
layers = [("conv12", 128, 128, 300, 80), ("conv21", 128, 256, 150, 40), ("conv22", 256, 256, 150, 40),
          ("conv31", 256, 512, 75, 20), ("conv32", 512, 512, 75, 20), ("conv41", 512, 1024, 38, 10), ("conv42", 1024, 1024, 38, 10)]
stream = torch.cuda.current_stream().cuda_stream
tot = [0.0, 0.0]
for name, cin, cout, H, W in layers:
    xf = torch.relu(torch.randn(B, cin, 2 * H, 2 * W, device=dev)).contiguous(memory_format=torch.channels_last)
    x = hc.maxpool2x2_ceil(xf, pack=True)          # packed [B,H,W,cin]
    del xf
    w = torch.randn(cout, cin, 3, 3, device=dev) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout, device=dev) * 0.1
    pk = hc.PackedConvWeights()
    hi, lo = pk.get(w, "split8")
    ys = [[torch.zeros(B, H, W, cout, device=dev) for _ in range(2)] for _ in libs]

    def run(k, yp):
        rc = libs[k][1].dasv_conv3x3_bias_relu_x8_nhwc(x.data_ptr(), 1, hi.data_ptr(), lo.data_ptr(), b.data_ptr(), ys[k][yp].data_ptr(), yp,
                                                      None, 0, B, H, W, cin, cout, 1, stream)
        assert rc == 0, rc

    best = [1e9, 1e9]
    for rnd in range(5):
        for k in (0, 1):
            run(k, 0); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(6): run(k, 0)
            e.record(); torch.cuda.synchronize()
            best[k] = min(best[k], s.elapsed_time(e) / 6)
    for k in (0, 1): run(k, 1)
    torch.cuda.synchronize()
    fl = 2 * 9 * cin * cout * H * W * B
    tot[0] += best[0]; tot[1] += best[1]
    print("%s %4d->%4d %3dx%2d | A %6.3f ms %6.1f TF | B %6.3f ms %6.1f TF | B vs A %+5.1f %% | identical fp32 %s packed %s"
          % (name, cin, cout, H, W, best[0], fl / best[0] / 1e9, best[1], fl / best[1] / 1e9, 100 * (best[0] / best[1] - 1),
             bool(torch.equal(ys[0][0], ys[1][0])), bool(torch.equal(ys[0][1].view(torch.int32), ys[1][1].view(torch.int32)))), flush=True)
print("total: A %.3f ms | B %.3f ms | B vs A %+.1f %%" % (tot[0], tot[1], 100 * (tot[0] / tot[1] - 1)))
