#!/usr/bin/env python3
"""Same-process A/B of the inference front-end with the blocks' max-pools fused into their convX2 launches against the
two-launch form, and of the PATCH-tile geometry of the 80-column layer (16 x 16 pixel tiles) against raster tiles: [64,300,80]
through VGG4L/1024, interleaved rounds.  Runs through the developer library (dasv_debug_conv_patch_tiles).  GPU box only."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from doubleattentionspeakerverification_amd import _lib
lib = _lib.use_debug_library()
from doubleattentionspeakerverification_amd.CNNs import VGG4L

mode = sys.argv[1] if len(sys.argv) > 1 else "split8"
dev = torch.device("cuda:0")
torch.manual_seed(1)
net = VGG4L(1024).to(dev).eval()
net.conv_mode = mode
x = torch.randn(64, 300, 80, device=dev)
# (fused blocks, patch tiles on the 80-column layer, forced raster tile of the fused launches)
variants = {"none raster": (set(), 0, 0), "none patch": (set(), 1, 0), "b1 raster": ({1}, 0, 0), "b1 patch": ({1}, 1, 0),
            "b2 raster256": ({2}, 1, 0), "b12 raster": ({1, 2}, 0, 0), "b12 patch": ({1, 2}, 1, 0), "b123 patch": ({1, 2, 3}, 1, 0),
            "all patch": ({1, 2, 3, 4}, 1, 0)}


def run(blocks, patch, tile, n=20):
    net.fuse_pool_layers = {(b, mode) for b in blocks}
    net.pool_tile = tile
    lib.dasv_debug_conv_patch_tiles(patch)
    with torch.no_grad():
        for _ in range(3):
            net.forward_map(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            net.forward_map(x)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


res = {}
for rnd in range(5):
    for k, (blocks, patch, tile) in variants.items():
        res.setdefault(k, []).append(run(blocks, patch, tile))
base = sorted(res["none raster"])[2]
for k in res:
    v = sorted(res[k])
    print("%-14s median %.3f ms  min %.3f ms   vs none raster %+.1f %%" % (k, v[2], v[0], (v[2] / base - 1) * 100))
