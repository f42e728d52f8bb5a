"""Same-box A/B of the split8 convolution: a reference build of csrc/conv3x3.hip + frontend_misc.hip (any earlier commit,
compiled to tools/_diag/libdasv_r2conv.so with the round-2 ABI: no range records) against the in-tree library, layer by layer on
the [64,300,80] shapes of the default model, packed input, interleaved rounds on the same box (boxes differ by up to 12 % on
MFMA-bound kernels, so only same-process comparisons mean anything).

    python tools/conv_ab_r3.py [old_so]
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from doubleattentionspeakerverification_amd import _lib, conv as hc  # noqa: E402

vp, ci, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
old = ctypes.CDLL(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tools", "_diag", "libdasv_r2conv.so"))
# AB_SAME_ABI=1: the reference library is a build of the CURRENT sources (range records in the ABI): it shares the packed input,
# weights and records with the in-tree library and only the convolution launch differs
SAME = os.environ.get("AB_SAME_ABI") == "1"
old.dasv_conv3x3_bias_relu_x8_nhwc.restype = ci
if SAME:
    old.dasv_conv3x3_bias_relu_x8_nhwc.argtypes = [vp, ci, vp, vp, vp, vp, vp, ci, vp, vp, sz, ci, ci, ci, ci, ci, ci, vp]
else:
    old.dasv_conv3x3_bias_relu_x8_nhwc.argtypes = [vp, ci, vp, vp, vp, vp, ci, vp, sz, ci, ci, ci, ci, ci, ci, vp]
    old.dasv_maxpool2x2_ceil_x8_nhwc.restype = ci
    old.dasv_maxpool2x2_ceil_x8_nhwc.argtypes = [vp, vp, ci, ci, ci, ci, vp]
    old.dasv_conv3x3_pack_weights_x8.restype = ci
    old.dasv_conv3x3_pack_weights_x8.argtypes = [vp, vp, vp, ci, ci, vp]
    old.dasv_conv3x3_packed_weight_bytes_x8.restype = sz
    old.dasv_conv3x3_packed_weight_bytes_x8.argtypes = [ci, ci]
new = _lib.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
B = int(os.environ.get("AB_BATCH", "64"))
LAYERS = [("conv12", 128, 128, 300, 80, 0), ("conv21", 128, 256, 150, 40, 1), ("conv22", 256, 256, 150, 40, 0),
          ("conv31", 256, 512, 75, 20, 1), ("conv32", 512, 512, 75, 20, 0), ("conv41", 512, 1024, 38, 10, 1),
          ("conv42", 1024, 1024, 38, 10, 0)]
torch.manual_seed(0)
tot = {"old": 0.0, "new": 0.0}
for name, cin, cout, H, W, ypk in LAYERS:
    xf = torch.relu(torch.randn(B, 2 * H, 2 * W, cin, device=dev))
    w = torch.randn(cout, cin, 3, 3, device=dev) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout, device=dev) * 0.1
    # old library: its own packed input and weights
    yo = torch.empty(B, H, W, cout, device=dev)
    if not SAME:
        xo = torch.empty(B, H, W, cin, device=dev)
        assert old.dasv_maxpool2x2_ceil_x8_nhwc(xf.data_ptr(), xo.data_ptr(), B, 2 * H, 2 * W, cin, st) == 0
        n16 = cout * 9 * cin * 2
        who = torch.empty(n16, dtype=torch.uint8, device=dev)
        wlo = torch.empty(max(n16, old.dasv_conv3x3_packed_weight_bytes_x8(cout, cin)), dtype=torch.uint8, device=dev)
        assert old.dasv_conv3x3_pack_weights_x8(w.data_ptr(), who.data_ptr(), wlo.data_ptr(), cout, cin, st) == 0
    # new library
    recs = hc.new_ranges(dev, 3)
    recs[0] = hc.range_of(xf)
    xn = torch.empty(B, H, W, cin, device=dev)
    assert new.dasv_maxpool2x2_ceil_x8_nhwc(xf.data_ptr(), recs[0].data_ptr(), xn.data_ptr(), recs[1].data_ptr(), B, 2 * H, 2 * W, cin, st) == 0
    pk = hc.PackedConvWeights()
    whn, wln = pk.get(w, "split8", b)
    yn = torch.empty(B, H, W, cout, device=dev)

    def run_old():
        if SAME:
            return old.dasv_conv3x3_bias_relu_x8_nhwc(xn.data_ptr(), 1, recs[1].data_ptr(), whn.data_ptr(), wln.data_ptr(), b.data_ptr(),
                                                      yo.data_ptr(), ypk, recs[2].data_ptr(), None, 0, B, H, W, cin, cout, 1, st)
        return old.dasv_conv3x3_bias_relu_x8_nhwc(xo.data_ptr(), 1, who.data_ptr(), wlo.data_ptr(), b.data_ptr(), yo.data_ptr(), ypk, None, 0,
                                                  B, H, W, cin, cout, 1, st)

    def run_new():
        return new.dasv_conv3x3_bias_relu_x8_nhwc(xn.data_ptr(), 1, recs[1].data_ptr(), whn.data_ptr(), wln.data_ptr(), b.data_ptr(),
                                                  yn.data_ptr(), ypk, recs[2].data_ptr(), None, 0, B, H, W, cin, cout, 1, st)
    assert run_old() == 0 and run_new() == 0
    torch.cuda.synchronize()
    if not ypk:
        err = float((yo - yn).abs().max() / yo.abs().max())
    else:
        err = float("nan")
    best = {"old": 1e9, "new": 1e9}
    for rnd in range(4):
        for tag, fn in (("old", run_old), ("new", run_new)) if rnd % 2 == 0 else (("new", run_new), ("old", run_old)):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best[tag] = min(best[tag], e0.elapsed_time(e1) / 10)
    tot["old"] += best["old"]
    tot["new"] += best["new"]
    print("%s %4d->%4d %3dx%2d ypk=%d  old %.4f ms  new %.4f ms  (%+.1f %%)  old-vs-new max rel diff %.1e"
          % (name, cin, cout, H, W, ypk, best["old"], best["new"], (best["new"] / best["old"] - 1) * 100, err), flush=True)
print("seven layers: old %.3f ms  new %.3f ms  (%+.1f %%)" % (tot["old"], tot["new"], (tot["new"] / tot["old"] - 1) * 100))
