#!/usr/bin/env python3
"""Same-box A/B of two builds of csrc/conv3x3.hip: the in-tree library against another shared object built from a
different revision of the source (default tools/_diag/ab/libconv_a.so), both called through the C ABI on the same
buffers, interleaved rounds, best of the rounds per layer.  Boxes of the pool differ by several percent, so kernel
edits are judged this way and never across runs.

    hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -fno-gpu-rdc -shared <other conv3x3.hip> -o tools/_diag/ab/libconv_a.so
"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from doubleattentionspeakerverification_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
other = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tools", "_diag", "ab", "libconv_a.so")
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0     # 0 split-bf16, 1 bf16, 2 fp16 (pack not wired here), 3 split8
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
libs = [("A (other)", ctypes.CDLL(other)), ("B (in-tree)", _lib.load())]
vp, ci = ctypes.c_void_p, ctypes.c_int
for _, l in libs:
    l.dasv_conv3x3_bias_relu_nhwc.restype = ci
    l.dasv_conv3x3_bias_relu_nhwc.argtypes = [ci, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]
    l.dasv_conv3x3_pack_weights.restype = ci
    l.dasv_conv3x3_pack_weights.argtypes = [vp, vp, vp, ci, ci, vp]
    if mode == 3:
        l.dasv_conv3x3_pack_weights_x8.restype = ci
        l.dasv_conv3x3_pack_weights_x8.argtypes = [vp, vp, vp, ci, ci, vp]
dev = torch.device("cuda:0")
torch.manual_seed(0)
layers = [("conv12", 128, 128, 300, 80), ("conv21", 128, 256, 150, 40), ("conv22", 256, 256, 150, 40),
          ("conv31", 256, 512, 75, 20), ("conv32", 512, 512, 75, 20), ("conv41", 512, 1024, 38, 10), ("conv42", 1024, 1024, 38, 10)]
stream = torch.cuda.current_stream().cuda_stream
tot = [0.0, 0.0]
for name, cin, cout, H, W in layers:
    x = torch.relu(torch.randn(B, H, W, cin, device=dev))
    w = torch.randn(cout, cin, 3, 3, device=dev) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout, device=dev) * 0.1
    hi = torch.empty(cout * 9 * cin * 2, dtype=torch.uint8, device=dev)
    lo = torch.empty(cout * 9 * cin * 2 + 4 * cout, dtype=torch.uint8, device=dev)
    pack = libs[1][1].dasv_conv3x3_pack_weights_x8 if mode == 3 else libs[1][1].dasv_conv3x3_pack_weights
    assert pack(w.data_ptr(), hi.data_ptr(), lo.data_ptr(), cout, cin, stream) == 0
    ys = [torch.empty(B, H, W, cout, device=dev) for _ in libs]

    def run(k):
        rc = libs[k][1].dasv_conv3x3_bias_relu_nhwc(mode, x.data_ptr(), hi.data_ptr(), lo.data_ptr(), b.data_ptr(), ys[k].data_ptr(),
                                                    B, H, W, cin, cout, 1, stream)
        assert rc == 0, rc

    best = [1e9, 1e9]
    for rnd in range(5):
        for k in (0, 1):
            run(k); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(6): run(k)
            e.record(); torch.cuda.synchronize()
            best[k] = min(best[k], s.elapsed_time(e) / 6)
    fl = 2 * 9 * cin * cout * H * W * B
    tot[0] += best[0]; tot[1] += best[1]
    print("%s %4d->%4d %3dx%2d | A %6.3f ms %6.1f TF | B %6.3f ms %6.1f TF | B vs A %+5.1f %% | identical %s"
          % (name, cin, cout, H, W, best[0], fl / best[0] / 1e9, best[1], fl / best[1] / 1e9, 100 * (best[0] / best[1] - 1),
             bool(torch.equal(ys[0], ys[1]))), flush=True)
print("total: A %.3f ms | B %.3f ms | B vs A %+.1f %%" % (tot[0], tot[1], 100 * (tot[0] / tot[1] - 1)))
