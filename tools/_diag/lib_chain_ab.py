#!/usr/bin/env python3
"""Same-process A/B of TWO BUILDS of the whole library on the inference front-end (VGG4L/1024, [64,300,80], split8 packed chain and the
fp16 chain): the in-tree libdasv_hip.so against another shared object built from an older revision of csrc/ (argv[1]; entry points the
old revision lacks are skipped when binding — the inference chains use none of them).  Interleaved rounds; boxes of the pool differ by
several percent, so "did this round's kernel edits cost the headline anything" is judged this way.  GPU box only."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from doubleattentionspeakerverification_amd import _lib
from doubleattentionspeakerverification_amd.CNNs import VGG4L

other_path = sys.argv[1]
new = _lib.load()
old = ctypes.CDLL(other_path)
for name, (res, args) in _lib.SIGNATURES.items():
    fn = getattr(old, name, None)
    if fn is not None:
        fn.restype, fn.argtypes = res, args
dev = torch.device("cuda:0")
torch.manual_seed(1234)
net = VGG4L(1024).to(dev).eval()
x = 2.0 * torch.randn(64, 300, 80, device=dev)


def timeit(n=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        net(x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for mode in ("split8", "fp16"):
    net.conv_mode = mode
    res = {"old": [], "new": []}
    with torch.no_grad():
        outs = {}
        for tag, lib in (("old", old), ("new", new)):
            _lib._lib = lib
            net(x); net(x)
            outs[tag] = net(x).clone()
        for rnd in range(5):
            for tag, lib in (("old", old), ("new", new)):
                _lib._lib = lib
                res[tag].append(timeit())
    _lib._lib = new
    a, b = sorted(res["old"])[2], sorted(res["new"])[2]
    print("%-6s front-end: old %.3f ms (min %.3f) | new %.3f ms (min %.3f) | %+.1f %%; outputs equal: %s"
          % (mode, a, min(res["old"]), b, min(res["new"]), 100 * (b / a - 1), bool(torch.equal(outs["old"], outs["new"]))))
