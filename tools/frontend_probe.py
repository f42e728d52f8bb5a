#!/usr/bin/env python3
"""Probe: time the VGG4L/1024 front-end (PyTorch-ROCm / MIOpen) under different settings on [64,300,80]."""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from doubleattentionspeakerverification_amd.CNNs import VGG4L

def run(tag, net, x, autocast=None, iters=5):
    with torch.no_grad(), torch.autocast("cuda", dtype=autocast or torch.bfloat16, enabled=autocast is not None):
        for _ in range(3): y = net(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(iters): y = net(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
    print("%-40s %8.2f ms  %7.1f TFLOP/s  out %s %s" % (tag, dt * 1e3, 39.13e9 * x.size(0) / dt / 1e12, tuple(y.shape), y.dtype), flush=True)
    return y

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = VGG4L(1024).to(dev).eval()
x = torch.randn(64, 300, 80, device=dev)
for bench in (False, True):
    torch.backends.cudnn.benchmark = bench
    for cl in (True, False):
        net.channels_last = cl
        y32 = run("fp32 benchmark=%s channels_last=%s" % (bench, cl), net, x)
        y16 = run("bf16 benchmark=%s channels_last=%s" % (bench, cl), net, x, torch.bfloat16)
        yh = run("fp16 benchmark=%s channels_last=%s" % (bench, cl), net, x, torch.float16)
        if bench and cl:
            ref = y32.float()
            print("   bf16 vs fp32 rel err %.3e ; fp16 vs fp32 rel err %.3e" % (float((y16.float() - ref).abs().max() / ref.abs().max()), float((yh.float() - ref).abs().max() / ref.abs().max())))
