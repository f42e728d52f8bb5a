"""Same-box A/B of the fp16-operand convolution on half-precision activations between a variant build (argv[1]: a library made from
the current sources with other -D flags) and the in-tree library, layer by layer on the [64,300,80] shapes of the default model."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from doubleattentionspeakerverification_amd import _lib, conv as hc  # noqa: E402

vp, ci, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
old = ctypes.CDLL(sys.argv[1])
old.dasv_conv3x3_bias_relu_f16_nhwc.restype = ci
old.dasv_conv3x3_bias_relu_f16_nhwc.argtypes = [vp, ci, vp, vp, vp, vp, ci, vp, vp, sz, ci, ci, ci, ci, ci, ci, vp]
new = _lib.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
B = 64
LAYERS = [("conv12", 128, 128, 300, 80), ("conv21", 128, 256, 150, 40), ("conv22", 256, 256, 150, 40), ("conv31", 256, 512, 75, 20),
          ("conv32", 512, 512, 75, 20), ("conv41", 512, 1024, 38, 10), ("conv42", 1024, 1024, 38, 10)]
torch.manual_seed(0)
tot = {"old": 0.0, "new": 0.0}
for name, cin, cout, H, W in LAYERS:
    x = torch.relu(torch.randn(B, H, W, cin, device=dev)).half()
    w = torch.randn(cout, cin, 3, 3, device=dev) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout, device=dev) * 0.1
    recs = hc.new_ranges(dev, 2)
    recs[0][0] = int(x.float().abs().max().view(torch.int32))
    hi, _ = hc.PackedConvWeights().get(w, "fp16", b)
    ys = {"old": torch.empty(B, H, W, cout, device=dev, dtype=torch.float16), "new": torch.empty(B, H, W, cout, device=dev, dtype=torch.float16)}

    def run(tag):
        lib = old if tag == "old" else new
        return lib.dasv_conv3x3_bias_relu_f16_nhwc(x.data_ptr(), 1, recs[0].data_ptr(), hi.data_ptr(), b.data_ptr(), ys[tag].data_ptr(), 1,
                                                   recs[1].data_ptr(), None, 0, B, H, W, cin, cout, 1, st)
    assert run("old") == 0 and run("new") == 0
    torch.cuda.synchronize()
    same = bool((ys["old"] == ys["new"]).all())
    best = {"old": 1e9, "new": 1e9}
    for rnd in range(4):
        for tag in (("old", "new") if rnd % 2 == 0 else ("new", "old")):
            for _ in range(3):
                run(tag)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run(tag)
            e1.record()
            torch.cuda.synchronize()
            best[tag] = min(best[tag], e0.elapsed_time(e1) / 10)
    tot["old"] += best["old"]
    tot["new"] += best["new"]
    print("%s %4d->%4d %3dx%2d  variant %.4f ms  in-tree %.4f ms  (in-tree %+.1f %%)  identical %s"
          % (name, cin, cout, H, W, best["old"], best["new"], (best["new"] / best["old"] - 1) * 100, same), flush=True)
print("seven layers: variant %.3f ms  in-tree %.3f ms  (in-tree %+.1f %%)" % (tot["old"], tot["new"], (tot["new"] / tot["old"] - 1) * 100))
