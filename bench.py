#!/usr/bin/env python3
"""Headline benchmark: utterance-embeddings/sec on synthetic [B=64, T=300, 80-mel] batches.

    python bench.py --gpus N --steps K --warmup W [--regions R] [--workload embed|pool|train|varlen] [--dtype fp32|bf16]

A step is one pass of the hot path over one batch per GPU, inputs already resident in HBM:
  embed (default)  SpeakerClassifier.getEmbedding (VGG4L/1024 front-end -> fused DoubleMHA -> fc/BN),
                   the reference's default configuration (train.py:263-291), random-init weights.
  pool             the fused DoubleMHA launch alone on [64,300,512], H=8 (BASELINE config 2), rotating over
                   enough input buffers to defeat the 256 MiB Infinity Cache.
  train            forward + AM-softmax/CE + backward + Adam step, gradients all-reduced over RCCL (config 4).
For N > 1 the driver launches one rank per GPU (torch.distributed.run); every rank processes its own batch
(weak scaling, replicas only for embed/pool — the path has no data exchange — and one gradient all-reduce
per step for train).  Rank 0 prints ONE JSON line.

`roofline` describes the kernel that dominates the step: for `embed`/`train` the hand-written conv3x3 implicit GEMM
(MFMA-bound: algorithmic FLOPs of its launches over their event-timed duration, against the dense bf16 MFMA peak;
split-bf16 operands execute 3 MFMA FLOPs per algorithmic FLOP, reported separately), for `pool` the fused DoubleMHA
kernel (HBM-bound: algorithmic bytes per launch).  `roofline_pooling` always describes the DoubleMHA
kernel.  `cpu_baseline` times the
op-sequence-faithful CPU restatement (oracle/, kind "port") on a bounded sample on rank 0 at N=1.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_BF16_PEAK_TFS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16 (the conv's MFMAs are bf16, fp32 accumulate)


def _lib_mod():
    from doubleattentionspeakerverification_amd import _lib
    return _lib


def default_params():
    return argparse.Namespace(front_end="VGG4L", kernel_size=1024, pooling_method="DoubleMHA", heads_number=32,
                              mask_prob=0.3, embedding_size=400, num_spkrs=5994, scalingFactor=30.0,
                              marginFactor=0.4, annealing=False)


def pooling_bytes(B, T, D, H, esize, with_align):
    """read x once + write alignments + output + parameters."""
    dh = D // H
    return esize * B * T * D + (4 * B * T * H if with_align else 0) + 4 * B * dh + 4 * (dh * H + dh)


class PoolTimer:
    """Events around every fused-pooling forward launch inside the timed region, recorded on the stream the
    kernel is launched on (the host-side wrapper `poolings._launch_fwd` is what every module path calls)."""

    def __init__(self, poolings):
        self.pairs = []
        self.active = False

        def wrap(inner):
            def timed(x, *a, **k):
                if not self.active:
                    return inner(x, *a, **k)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(torch.cuda.current_stream(x.device))
                r = inner(x, *a, **k)
                e.record(torch.cuda.current_stream(x.device))
                self.pairs.append((s, e))
                return r
            return timed
        poolings._launch_fwd = wrap(poolings._launch_fwd)
        poolings._launch_fwd_nhwc = wrap(poolings._launch_fwd_nhwc)   # the model's path: the front-end's NHWC map pooled in place

    def mean_us(self):
        d = [s.elapsed_time(e) * 1e3 for s, e in self.pairs]
        return float(np.mean(d)) if d else None


class ConvTimer:
    """Events around every hand-written conv3x3 launch inside the timed region (launch stream), with the
    algorithmic FLOPs of each launch (2*9*Cin*Cout*B*H*W)."""

    def __init__(self, convmod):
        self.rec = []
        self.active = False
        inner = convmod._launch

        def timed(x, hi, lo, bias, cout, mode, *a, **k):
            if not self.active:
                return inner(x, hi, lo, bias, cout, mode, *a, **k)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(torch.cuda.current_stream(x.device))
            y = inner(x, hi, lo, bias, cout, mode, *a, **k)
            e.record(torch.cuda.current_stream(x.device))
            B, cin, H, W = x.shape
            self.rec.append((s, e, 2.0 * 9 * cin * cout * B * H * W))
            return y
        convmod._launch = timed
        # the packed training chain's other two launch sites: conv + ReLU + pool with arg-max bytes, and conv12's data gradient with
        # conv11's backward in its write-out pass (both: x [B, cin, H, W], cout as fifth argument)
        for name in ("_launch_pool_idx", "_launch_dgrad_c1"):
            def wrap(inner_fn):
                def timed_fn(x, hi, lo, a4, a5, *a, **k):
                    if not self.active:
                        return inner_fn(x, hi, lo, a4, a5, *a, **k)
                    cout = a5 if inner_fn.__name__ == "_launch_pool_idx" else a4
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record(torch.cuda.current_stream(x.device))
                    y = inner_fn(x, hi, lo, a4, a5, *a, **k)
                    e.record(torch.cuda.current_stream(x.device))
                    B, cin, H, W = x.shape
                    self.rec.append((s, e, 2.0 * 9 * cin * cout * B * H * W))
                    return y
                return timed_fn
            setattr(convmod, name, wrap(getattr(convmod, name)))
        inner8 = convmod.conv3x3_x8       # the packed-activation inference chain (CNNs._forward_map_packed) launches through this one

        def timed8(x, weight, *a, **k):
            if not self.active:
                return inner8(x, weight, *a, **k)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(torch.cuda.current_stream(x.device))
            y = inner8(x, weight, *a, **k)
            e.record(torch.cuda.current_stream(x.device))
            B, cin, H, W = x.shape
            self.rec.append((s, e, 2.0 * 9 * cin * weight.shape[0] * B * H * W))
            return y
        convmod.conv3x3_x8 = timed8
        inner16 = convmod.conv3x3_f16     # the fp16 activation chain (CNNs._forward_map_f16)

        def timed16(x, weight, *a, **k):
            if not self.active:
                return inner16(x, weight, *a, **k)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(torch.cuda.current_stream(x.device))
            y = inner16(x, weight, *a, **k)
            e.record(torch.cuda.current_stream(x.device))
            B, cin, H, W = x.shape
            self.rec.append((s, e, 2.0 * 9 * cin * weight.shape[0] * B * H * W))
            return y
        convmod.conv3x3_f16 = timed16
        try:                                  # the ragged-batch launches (--workload varlen) go through ragged._conv_ragged
            from doubleattentionspeakerverification_amd import ragged as ragmod
        except ImportError:
            ragmod = None
        if ragmod is not None:
            innerr = ragmod._conv_ragged

            def timedr(x, conv, *a, **k):
                if not self.active:
                    return innerr(x, conv, *a, **k)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(torch.cuda.current_stream(x.device))
                y = innerr(x, conv, *a, **k)
                e.record(torch.cuda.current_stream(x.device))
                rows, W, cin = x.shape
                self.rec.append((s, e, 2.0 * 9 * cin * conv.out_channels * rows * W))
                return y
            ragmod._conv_ragged = timedr
            innerr16 = ragmod._conv_ragged_f16

            def timedr16(x, conv, *a, **k):
                if not self.active:
                    return innerr16(x, conv, *a, **k)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(torch.cuda.current_stream(x.device))
                y = innerr16(x, conv, *a, **k)
                e.record(torch.cuda.current_stream(x.device))
                rows, W, cin = x.shape
                self.rec.append((s, e, 2.0 * 9 * cin * conv.out_channels * rows * W))
                return y
            ragmod._conv_ragged_f16 = timedr16

    def summary(self):
        if not self.rec:
            return None
        us = sum(s.elapsed_time(e) * 1e3 for s, e, _ in self.rec)
        fl = sum(f for _, _, f in self.rec)
        return us / len(self.rec), fl / len(self.rec), fl / us / 1e6   # avg us, avg flops, TFLOP/s


class WgradTimer:
    """Events around every hand-written weight-gradient launch (kernel + its fixed-order reduction) inside the timed region, with
    the algorithmic FLOPs of each (2*9*Cin*Cout*B*H*W, as the forward layer)."""

    def __init__(self, convmod):
        self.rec, self.modes = [], set()
        self.active = False
        inner = convmod.conv3x3_wgrad

        def timed(x, g, mode="split", *a, **k):
            if not self.active:
                return inner(x, g, mode, *a, **k)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(torch.cuda.current_stream(x.device))
            y = inner(x, g, mode, *a, **k)
            e.record(torch.cuda.current_stream(x.device))
            B, cin, H, W = x.shape
            self.rec.append((s, e, 2.0 * 9 * cin * g.shape[1] * B * H * W))
            self.modes.add(mode)
            return y
        convmod.conv3x3_wgrad = timed

    summary = ConvTimer.summary


def pooling_config2(poolings, lib, dev, launches=200, reps=5):
    """BASELINE config 2 (DoubleMHA pooling only, B=64 T=300 D=512 heads=8, fp32): the fused kernel through the
    allocation-free plan launcher, back to back on torch's current stream, rotating over 8 input buffers (315 MB > the
    256 MiB Infinity Cache), HIP events around the batch; median of `reps`.  Beside it, on the same buffers, the read floor:
    the simplest kernel that reads x once (dasv_debug_read_floor) — the time no kernel that streams x can beat here."""
    B, T, D, H = 64, 300, 512, 8
    g = torch.Generator(device=dev).manual_seed(4321)
    xs = [torch.randn(B, T, D, device=dev, generator=g) for _ in range(8)]
    mod = poolings.DoubleMHA(D, H).to(dev).eval()
    plan = poolings.DoubleMHAPlan(mod, B, T, dtype=torch.float32, device=dev, want_align=True)
    scratch = torch.zeros(4096, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    nbytes = B * T * D * 4

    def timed(fn):
        out = []
        for _ in range(reps):
            for i in range(16):
                fn(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(launches):
                fn(i)
            e1.record()
            torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) * 1e3 / launches)
        return sorted(out)[len(out) // 2]

    pool_us = timed(lambda i: plan(xs[i % 8]))
    variant = lib.dasv_dmha_last_variant().decode()
    probe = _lib_mod().debug_probe()      # the read-floor kernel lives in the developer build of the library only

    def floor(i):
        rc = probe.dasv_debug_read_floor(xs[i % 8].data_ptr(), nbytes, scratch.data_ptr(), stream)
        if rc != 0:
            raise RuntimeError("dasv_debug_read_floor rc=%d" % rc)
    floor_us = timed(floor)
    # the same kernel on a batch that amortises its fixed costs (launch ramp, merge, head exchange): B = 512, two rotating buffers
    # of 315 MB each — context for the B = 64 figure, never the quoted one
    del xs
    big = None
    try:
        Bb = 512
        xb = [torch.randn(Bb, T, D, device=dev, generator=g) for _ in range(2)]
        planb = poolings.DoubleMHAPlan(mod, Bb, T, dtype=torch.float32, device=dev, want_align=True)
        nbb = Bb * T * D * 4

        def big_floor(i):
            probe.dasv_debug_read_floor(xb[i % 2].data_ptr(), nbb, scratch.data_ptr(), stream)
        launches = 50
        pb = timed(lambda i: planb(xb[i % 2]))
        fb = timed(big_floor)
        ab = pooling_bytes(Bb, T, D, H, 4, True)
        big = {"B": Bb, "avg_launch_us": round(pb, 2), "achieved_GBps": round(ab / pb / 1e3, 1), "frac": round(ab / pb / 1e3 / HBM_PEAK_GBS, 4),
               "read_floor_us": round(fb, 2), "frac_of_read_floor": round(fb / pb, 4)}
    except RuntimeError:
        big = None
    return (B, T, D, H), pool_us, floor_us, variant, big


def pmc_traffic(workload, dtype):
    """HBM bytes per launch of the fused kernel from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json:
    FETCH_SIZE and WRITE_SIZE collected in separate passes, FETCH_SIZE doubled per the gfx950 note in
    MI355X_MICROARCH.md).  bench.py cannot run the profiler on itself; null when no pass was recorded."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        return rec.get("%s_%s" % (workload, dtype), {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def conv_power_probe(dev, B):
    """Does the chip's power management hold the split8 convolution back?  One layer of the default model (conv32: 512 -> 512 on
    [B,75,20]) launched on random ReLU activations / random weights and on ALL-ZERO activations and weights: same instruction
    stream, same addresses and traffic, but the matrix pipe draws far less on zeros.  A launch that finishes sooner on zeros was
    running at a lowered clock (MI355X_MICROARCH.md, 'DVFS give-back'); its roofline fraction against the 2.4 GHz peak cannot
    reach what a kernel limited by its own stalls could.  (tools/_diag/conv_power_test.py: the four wide layers.)"""
    import torch
    from doubleattentionspeakerverification_amd import _lib, conv as hc
    lib = _lib.load()
    st = torch.cuda.current_stream(dev).cuda_stream
    cin = cout = 512
    H, W = 75, 20
    g = torch.Generator(device=dev).manual_seed(7)
    runs, keep = {}, []
    for tag in ("random", "zeros"):
        xf = torch.relu(torch.randn(B, 2 * H, 2 * W, cin, device=dev, generator=g))
        w = torch.randn(cout, cin, 3, 3, device=dev, generator=g) * (2.0 / (9 * cin)) ** 0.5
        b = torch.randn(cout, device=dev, generator=g) * 0.1
        if tag == "zeros":
            xf.zero_()
            w.zero_()
        recs = hc.new_ranges(dev, 3)
        if tag == "random":
            recs[0] = hc.range_of(xf)
        xn = torch.empty(B, H, W, cin, device=dev)
        _lib.check(lib.dasv_maxpool2x2_ceil_x8_nhwc(xf.data_ptr(), recs[0].data_ptr(), xn.data_ptr(), recs[1].data_ptr(), B, 2 * H, 2 * W, cin, st),
                   "dasv_maxpool2x2_ceil_x8_nhwc")
        wh, wl = hc.PackedConvWeights().get(w, "split8", b)
        y = torch.empty(B, H, W, cout, device=dev)
        keep.append((xn, wh, wl, b, y, recs))

        def run(xn=xn, wh=wh, wl=wl, b=b, y=y, recs=recs):
            _lib.check(lib.dasv_conv3x3_bias_relu_x8_nhwc(xn.data_ptr(), 1, recs[1].data_ptr(), wh.data_ptr(), wl.data_ptr(), b.data_ptr(),
                                                          y.data_ptr(), 0, recs[2].data_ptr(), None, 0, B, H, W, cin, cout, 1, st),
                       "dasv_conv3x3_bias_relu_x8_nhwc")
        runs[tag] = run
        del xf
    best = {"random": 1e9, "zeros": 1e9}
    for rnd in range(4):
        for tag in (("random", "zeros") if rnd % 2 == 0 else ("zeros", "random")):
            for _ in range(5):
                runs[tag]()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                runs[tag]()
            e1.record()
            torch.cuda.synchronize(dev)
            best[tag] = min(best[tag], e0.elapsed_time(e1) / 20)
    return {"layer": "conv32 split8, 512->512 on [%d,75,20], packed input" % B, "ms_random_data": round(best["random"], 4),
            "ms_all_zero_data": round(best["zeros"], 4), "zero_over_random": round(best["zeros"] / best["random"], 4),
            "reading": "same launches, same instruction stream and traffic: the share by which zeros finish sooner is clock the chip "
                       "withholds on real data"}


def mfma_util_pmc():
    """Matrix-pipe busy share of the conv kernels from the committed `rocprofv3 --pmc MfmaUtil` passes (profiles/)."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get("mfma_util_percent") or None
    except (OSError, ValueError):
        return None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def companion_workloads():
    """BASELINE configs 3/4 and 5 inside the driver-timed line: after the headline's timed regions (never inside them, never in
    `value`), this script runs itself as a child process for the training step (`--workload train`: forward + AM-softmax/CE + backward +
    Adam on [64,300,80], 5994 speakers; and the same on a minibatch of utterances of different lengths) and for the variable-length pass on
    the fp16 MFMA path (`--workload varlen --conv fp16`: 1,024
    utterances of 150..800 frames, ragged packing), and keeps each child's value, step time and roofline.  The parent's GPU work is
    finished and synchronised before a child starts; a child that fails is reported as such, not hidden."""
    import subprocess
    out = {}
    runs = {"train (BASELINE config 3/4 on one GPU)": ["--workload", "train", "--steps", "20", "--warmup", "5", "--regions", "5"],
            "train_varied_lengths (north_star: variable-length training minibatches, one GPU)":
                ["--workload", "train", "--train-lengths", "varied", "--steps", "10", "--warmup", "3", "--regions", "3"],
            "varlen_fp16 (BASELINE config 5 on one GPU)": ["--workload", "varlen", "--conv", "fp16", "--steps", "3", "--warmup", "1", "--regions", "3"]}
    for name, extra in runs.items():
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--no-cpu-baseline", "--no-companions", "--no-power-probe"] + extra
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not line:
                out[name] = {"error": "rc %d: %s" % (r.returncode, (r.stderr or r.stdout)[-300:])}
                continue
            d = json.loads(line[-1])
            keep = {k: d[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype") if k in d}
            keep["regions_elapsed_s"] = d.get("regions", {}).get("elapsed_s")
            keep["workload"] = d.get("config", {}).get("workload")
            roof = d.get("roofline") or {}
            keep["roofline"] = {k: roof[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_us", "launches_per_step",
                                                     "mfma_executed_frac") if k in roof}
            if "weight_gradient" in roof:
                keep["roofline"]["weight_gradient"] = {k: roof["weight_gradient"][k] for k in ("kernel", "achieved", "frac", "avg_launch_us",
                                                                                               "launches_per_step") if k in roof["weight_gradient"]}
            keep["child_wall_s"] = round(time.perf_counter() - t0, 1)
            out[name] = keep
        except (subprocess.TimeoutExpired, ValueError, OSError) as e:
            out[name] = {"error": repr(e)[:300]}
    return out


def host_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(np.ceil(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline_embed(params, frames, seconds=12.0):
    import oracle
    from doubleattentionspeakerverification_amd.model import SpeakerClassifier
    torch.manual_seed(1234)
    net = SpeakerClassifier(params, torch.device("cpu")).eval()
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    rs = np.random.RandomState(1234)
    Bc = 4
    x = torch.from_numpy(rs.standard_normal((Bc, frames, 80)).astype(np.float32))
    n_blocks = 4 if params.front_end == "VGG4L" else 3
    with torch.no_grad():
        oracle.embedding_forward(x[:1], sd, n_blocks, params.heads_number)  # warm-up
        t0, n = time.perf_counter(), 0
        while True:
            oracle.embedding_forward(x, sd, n_blocks, params.heads_number)
            n += Bc
            if time.perf_counter() - t0 > seconds:
                break
        dt = time.perf_counter() - t0
    return dict(value=n / dt, unit="utterance-embeddings/s", cores=torch.get_num_threads(), kind="port", cpu_model=cpu_model(),
                sample="%d utterances of [%d,80] through oracle.embedding_forward (torch CPU fp32, op-sequence "
                       "restatement of model.py:52-59), %.1f s" % (n, frames, dt))


def cpu_baseline_pool(B, T, D, H, seconds=10.0):
    import oracle
    rs = np.random.RandomState(1234)
    dh = D // H
    x = torch.from_numpy(rs.standard_normal((B, T, D)).astype(np.float32))
    q = torch.from_numpy((rs.standard_normal((dh, H)) * np.sqrt(2.0 / (dh + H))).astype(np.float32))
    a = torch.from_numpy((rs.standard_normal((dh, 1)) * np.sqrt(2.0 / (dh + 1))).astype(np.float32))
    with torch.no_grad():
        oracle.dmha_opseq(x, q, a)
        t0, n = time.perf_counter(), 0
        while True:
            oracle.dmha_opseq(x, q, a)
            n += B
            if time.perf_counter() - t0 > seconds:
                break
        dt = time.perf_counter() - t0
    return dict(value=n / dt, unit="utterance-embeddings/s", cores=torch.get_num_threads(), kind="port", cpu_model=cpu_model(),
                sample="%d utterances of [%d,%d] through oracle.dmha_opseq (torch CPU fp32, the reference's op "
                       "sequence poolings.py:73-80,45-51), %.1f s" % (n, T, D, dt))


def timed_regions(run_steps, warmup, steps, regions, dist, dev):
    """The driver's timing contract, `regions` times: W untimed warm-up steps once, then per region EXACTLY `steps` steps bracketed
    by barrier + synchronize on both sides, the MAX over ranks of its wall time; returns (elapsed of the MEDIAN region in seconds,
    that region's HIP-event milliseconds on this rank's stream or None on the CPU, all regions' elapsed).  Device-agnostic so
    that the multi-rank path is exercised on gloo/CPU by tests/test_dist_cpu.py."""
    cuda = torch.device(dev).type == "cuda"

    def barrier():
        if dist is not None:
            dist.barrier()
        if cuda:
            torch.cuda.synchronize()

    run_steps(0, warmup)
    regs = []
    first = warmup
    for _ in range(max(1, regions)):
        barrier()
        ev0 = ev1 = None
        if cuda:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        if cuda:
            ev0.record()
        run_steps(first, steps)
        if cuda:
            ev1.record()
        barrier()
        el = time.perf_counter() - t0
        first += steps
        regs.append((el, ev0.elapsed_time(ev1) if cuda else None))
    els = torch.tensor([r[0] for r in regs], device=dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(els, op=dist.ReduceOp.MAX)
    els = [float(v) for v in els.tolist()]
    mid = sorted(range(len(els)), key=lambda i: els[i])[len(els) // 2]
    return els[mid], regs[mid][1], els


def headline(metric, unit, units_per_rank_step, world, steps, warmup, elapsed, els, dist):
    """The fields of the JSON line that every workload shares: `value` is the WHOLE-JOB rate (all ranks' units over the median
    region's max-over-ranks time); `dist` says what torch.distributed reports, so that an N-GPU run can be checked to have had N ranks."""
    return {"metric": metric, "value": round(units_per_rank_step * world * steps / elapsed, 2), "unit": unit,
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "regions": {"count": len(els), "steps_each": steps, "elapsed_s": [round(v, 5) for v in els], "quoted": "median"},
            "dist": {"world_size": (dist.get_world_size() if dist is not None else 1),
                     "backend": (dist.get_backend() if dist is not None else None)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--regions", type=int, default=5,
                    help="timed regions of --steps steps each (every one bracketed by barrier + synchronize); the line reports the "
                         "MEDIAN region, so ms_per_step x steps is the time of one real region")
    ap.add_argument("--workload", default="embed", choices=["embed", "pool", "train", "varlen"])
    ap.add_argument("--train-lengths", default="equal", choices=["equal", "varied"],
                    help="--workload train: `varied` = a minibatch of utterances of different lengths (a list), T uniform in [2T/3, 4T/3]")
    ap.add_argument("--utterances", type=int, default=1024, help="varlen: utterances per rank and step")
    ap.add_argument("--packing", default="ragged", choices=["ragged", "bucket"],
                    help="varlen: ragged = utterances stacked along time, borders from tables (ragged.py); bucket = "
                         "equal-length buckets (batching.py)")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--conv", default="split8", choices=["split8", "split", "bf16", "fp16", "torch"],
                    help="front-end convolutions: hand-written MFMA implicit GEMM, fp32-grade with fp16 products + block-scaled e4m3 "
                         "cross terms (split8, default: ~1e-5 of the output scale per layer) or split-bf16 operands (split: ~5e-6), "
                         "single-MFMA bf16 / fp16 operands, or PyTorch-ROCm/MIOpen")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-companions", action="store_true",
                    help="skip `companion_workloads` (the training step and the variable-length fp16 pass timed by child runs of this "
                         "script after the headline's regions; profiling runs and the children themselves pass this)")
    ap.add_argument("--no-power-probe", action="store_true", help="skip roofline.power_probe (profiling runs: its 200 launches would mix into the kernel statistics)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL over xGMI) for real runs; gloo only to rehearse the multi-rank path on one GPU")
    ap.add_argument("--share-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start one rank per GPU as child processes (this parent never
        # touches the GPU) and leave with their exit code
        import socket
        import subprocess
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE)" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the measured path)")
    if args.share_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist_mod.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist_mod.init_process_group("gloo", rank=rank, world_size=world)
        dist = dist_mod

    from doubleattentionspeakerverification_amd import _lib, poolings, conv as convmod
    from doubleattentionspeakerverification_amd.model import SpeakerClassifier
    lib = _lib.load()  # raises if the HIP library is missing

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    B, T = args.batch, args.frames
    rs = np.random.RandomState(1234 + rank)
    timer = PoolTimer(poolings)
    ctimer = ConvTimer(convmod)
    wtimer = WgradTimer(convmod)
    compute_dtype = torch.float32 if args.dtype == "fp32" else torch.bfloat16
    config = {}

    if args.workload == "pool":
        D, H = 512, 8
        dh = D // H
        esize = 4 if compute_dtype == torch.float32 else 2
        nbuf = max(8, int(np.ceil(300e6 / (B * T * D * esize))))
        g = torch.Generator(device=dev).manual_seed(1234 + rank)
        xs = [torch.randn(B, T, D, device=dev, generator=g).to(compute_dtype) for _ in range(nbuf)]
        mod = poolings.DoubleMHA(D, H).to(dev).eval()
        pool_shape = (B, T, D, H)
        with_align = True
        # The launch is ~14 us, so the serving-style allocation-free launcher is used (one ctypes call per step,
        # outputs preallocated): the GPU, not Python, is the bottleneck and elapsed/K is the launch time.
        plan = poolings.DoubleMHAPlan(mod, B, T, dtype=compute_dtype, device=dev, want_align=True)

        def step(i):
            return plan(xs[i % nbuf])[0]
        config = {"workload": "DoubleMHA pooling only, B=%d T=%d D=%d heads=%d (BASELINE config 2)" % (B, T, D, H),
                  "global_batch": B * world, "frames": T, "input_buffers_rotated": nbuf, "parallelism": "replicas x%d" % world}
    else:
        params = default_params()
        torch.manual_seed(1234)
        net = SpeakerClassifier(params, dev).to(dev)
        net.front_end.conv_mode = args.conv
        x = torch.from_numpy((2.0 * rs.standard_normal((B, T, 80))).astype(np.float32))
        x = (x - x.mean(dim=1, keepdim=True)).to(dev)  # CMN, as featureExtractor.py:25-26
        Tp = T
        for _ in range(4):
            Tp = (Tp + 1) // 2
        pool_shape = (B, Tp, 5120, 32)
        esize = 4 if compute_dtype == torch.float32 else 2
        with_align = False
        if args.workload == "varlen":
            # BASELINE config 5: utterances of T in [150, 800] frames, packed into equal-length buckets (the only packing
            # that is exact for this model: no padding anywhere, batching.py), this rank's share resident in HBM
            from doubleattentionspeakerverification_amd.batching import embed_utterances
            from doubleattentionspeakerverification_amd.ragged import embed_ragged_batches
            net.eval()
            lens = rs.randint(150, 801, size=args.utterances)
            utts = [torch.from_numpy((2.0 * rs.standard_normal((int(n), 80))).astype(np.float32)).to(dev) for n in lens]
            utts = [u - u.mean(dim=0, keepdim=True) for u in utts]
            B = args.utterances

            def step(i):
                if args.packing == "ragged":
                    return embed_ragged_batches(net, utts, max_frames=args.batch * 300)
                return embed_utterances(net, utts, max_batch=args.batch)
            config = {"workload": "variable-length embedding extraction: %d utterances per GPU and step, T uniform in [150,800], "
                                  "%s (BASELINE config 5), VGG4L/1024 + DoubleMHA"
                                  % (B, "stacked along time in groups of <= %d frames, utterance borders from tables" % (args.batch * 300)
                                     if args.packing == "ragged" else "equal-length buckets of up to %d" % args.batch),
                      "global_batch": B * world, "frames": "150..800 (mean %.0f)" % float(np.mean(lens)),
                      "parallelism": "replicas x%d (no data exchange)" % world, "conv": args.conv}
        elif args.workload == "embed":
            net.eval()

            def step(i):
                with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=(compute_dtype == torch.bfloat16)):
                    return net.getEmbedding(x)
            config = {"workload": "SpeakerClassifier.getEmbedding, VGG4L/1024 + DoubleMHA(32 heads) + fc, emb 400, "
                                  "input [%d,%d,80] per GPU (reference default config, train.py:263-291)" % (B, T),
                      "global_batch": B * world, "frames": T, "parallelism": "replicas x%d (no data exchange)" % world,
                      "conv": {"split8": "hand-written MFMA implicit GEMM, fp32-grade: every fp32 operand v = fp16 hi + residual, hi*hi on "
                                         "v_mfma_f32_16x16x32_f16, both cross terms on one block-scaled e4m3 MFMA per 64 channels "
                                         "(v_mfma_scale_f32_16x16x128_f8f6f4), fp32 accumulate, fp32 activations; measured ~1e-5 of the "
                                         "output scale per layer vs a float64 convolution (tests/test_model_gpu.py)",
                               "split": "hand-written MFMA implicit GEMM, fp32 operands split into bf16 hi+lo (3 MFMAs per "
                                        "product, fp32 accumulate; ~5e-6 relative error vs fp32 conv)",
                               "bf16": "hand-written MFMA implicit GEMM, bf16 operands",
                               "fp16": "hand-written MFMA implicit GEMM, fp16 operands (fp32 accumulate)", "torch": "PyTorch-ROCm / MIOpen"}[args.conv]
                              if compute_dtype == torch.float32 else "torch.autocast(bf16): hand-written conv kernels with bf16 operands (fp32 accumulate, "
                              "fp32 activations), dense layers in bf16 through PyTorch-ROCm"}
        else:
            from doubleattentionspeakerverification_amd.dist import DataParallelTrainer
            net.train()
            labels = torch.from_numpy(rs.randint(0, params.num_spkrs, size=(B,))).to(dev)
            trainer = DataParallelTrainer(net, lr=1e-4, weight_decay=1e-3, autocast_dtype=(None if compute_dtype == torch.float32 else torch.bfloat16))

            xs = x
            if args.train_lengths == "varied":
                # north_star: minibatches of VARIABLE-LENGTH utterances — T uniform in [2T/3, 4T/3] (mean T: the dense batch's frame
                # count), handed to the trainer as a list: one set of launches with border tables
                lens = rs.randint(2 * T // 3, 4 * T // 3 + 1, size=B)
                xs = [torch.from_numpy(rs.standard_normal((int(n), 80)).astype(np.float32)).to(dev) for n in lens]

            def step(i):
                return trainer.train_step(xs, labels, step=i)
            config = {"workload": "training step: forward + AM-softmax/CE + backward + Adam, input [%d,%d,80] per GPU, "
                                  "5994 speakers (BASELINE config 3/4)" % (B, T),
                      "global_batch": B * world, "frames": T, "parallelism": "dp%d (RCCL all-reduce of flat fp32 gradients)" % world}
            if args.train_lengths == "varied":
                config["workload"] = ("training step on a VARIABLE-LENGTH minibatch: %d utterances per GPU, T uniform in [%d,%d] (%d frames, "
                                      "%d distinct lengths), each processed at its own length in one set of launches; forward + AM-softmax/CE "
                                      "+ backward + Adam, 5994 speakers (BASELINE.json north_star)"
                                      % (B, 2 * T // 3, 4 * T // 3, int(lens.sum()), len(set(lens.tolist()))))

    def run_steps(first, count):
        for i in range(count):
            step(first + i)

    # warm-up, then `--regions` timed regions of exactly `--steps` steps (timed_regions); the line quotes the MEDIAN region (one
    # real region: ms_per_step x steps is its wall time).  Several regions because one region of the embedding workload is
    # 0.1 s — too short for a sampler outside the process to see the GPU busy, and a single region inherits whatever clock
    # state the warm-up left.
    run_steps(0, args.warmup)
    timer.active = True
    ctimer.active = wtimer.active = True
    elapsed, region_ms, els = timed_regions(lambda first, count: run_steps(args.warmup + first, count), 0, args.steps, args.regions, dist, dev)
    timer.active = False
    ctimer.active = wtimer.active = False
    nstep_timed = args.steps * len(els)

    if rank == 0:
        Bp, Tp, Dp, Hp = pool_shape
        abytes = pooling_bytes(Bp, Tp, Dp, Hp, esize, with_align)
        pool_us = timer.mean_us()
        if args.workload == "pool":   # back-to-back launches on one stream: events around the whole timed region
            pool_us = region_ms * 1e3 / args.steps
        pool_roof = None
        if pool_us:
            ach = abytes / pool_us / 1e3  # GB/s
            pool_roof = {"kernel": "dasv::dmha_fwd_vec (fused DoubleMHA forward)", "bound": "hbm", "achieved": round(ach, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                         "traffic": pmc_traffic(args.workload, args.dtype),
                         "algorithmic_bytes_per_launch": abytes, "avg_launch_us": round(pool_us, 3),
                         "pool_shape": {"B": Bp, "T": Tp, "D": Dp, "H": Hp}, "variant": lib.dasv_dmha_last_variant().decode()}
            if args.workload == "pool" and compute_dtype == torch.float32:
                # the read floor on the same rotating buffers: the simplest kernel that reads x once
                scratch = torch.zeros(4096, device=dev)
                probe = _lib_mod().debug_probe()
                sp = torch.cuda.current_stream(dev).cuda_stream
                nb = Bp * Tp * Dp * 4
                fl = []
                for _ in range(5):
                    for i in range(16):
                        probe.dasv_debug_read_floor(xs[i % nbuf].data_ptr(), nb, scratch.data_ptr(), sp)
                    torch.cuda.synchronize()
                    f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    f0.record()
                    for i in range(200):
                        probe.dasv_debug_read_floor(xs[i % nbuf].data_ptr(), nb, scratch.data_ptr(), sp)
                    f1.record()
                    torch.cuda.synchronize()
                    fl.append(f0.elapsed_time(f1) * 1e3 / 200)
                floor_us = sorted(fl)[2]
                pool_roof.update({"read_floor_us": round(floor_us, 3),
                                  "read_floor_frac_of_peak": round(nb / floor_us / 1e3 / HBM_PEAK_GBS, 4),
                                  "frac_of_read_floor": round(floor_us / pool_us, 4)})
        in_model = None
        if args.workload != "pool":
            # `roofline_pooling` is quoted on BASELINE config 2, the configuration the north star sizes the fused kernel on; the
            # launches inside this workload's step (a smaller [B,T',5120] map, events around single launches) are kept beside it
            in_model = pool_roof
            (Bc, Tc, Dc, Hc), c2_us, floor_us, variant, big = pooling_config2(poolings, lib, dev)
            ab2 = pooling_bytes(Bc, Tc, Dc, Hc, 4, True)
            ach = ab2 / c2_us / 1e3
            pool_roof = {"kernel": "dasv::dmha_fwd_vec (fused DoubleMHA forward)", "bound": "hbm", "achieved": round(ach, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                         "traffic": pmc_traffic("pool", "fp32"), "traffic_source": "profiles/pmc_traffic.json — NOT measured by this run",
                         "algorithmic_bytes_per_launch": ab2,
                         "avg_launch_us": round(c2_us, 3), "pool_shape": {"B": Bc, "T": Tc, "D": Dc, "H": Hc},
                         "workload": "BASELINE config 2: DoubleMHA pooling only, fp32, 8 rotating input buffers, 200 launches back to back",
                         "variant": variant,
                         "read_floor_us": round(floor_us, 3), "read_floor_frac_of_peak": round(Bc * Tc * Dc * 4 / floor_us / 1e3 / HBM_PEAK_GBS, 4),
                         "frac_of_read_floor": round(floor_us / c2_us, 4),
                         "large_batch_context": big,
                         "in_model_launch": in_model}
        roofline = pool_roof
        csum = ctimer.summary()
        if csum is not None:
            # the step is dominated (~90 %) by the hand-written conv3x3 launches: MFMA-bound
            avg_us, avg_fl, tfs = csum
            # matrix-pipe time per algorithmic FLOP, in units of a plain bf16 MFMA: split = 3 bf16 MFMAs per product; split8 = one
            # fp16 MFMA + one e4m3 MFMA of twice the K at twice the rate per 64 channels = 2
            mult = {"split": 3, "split8": 2}.get(args.conv, 1)
            roofline = {"kernel": "dasv::conv3x3_window_kernel (implicit-GEMM conv3x3+bias+ReLU, %s operands)" % args.conv,
                        "bound": "mfma", "achieved": round(tfs, 1), "peak": MFMA_BF16_PEAK_TFS, "unit": "TFLOP/s",
                        "frac": round(tfs / MFMA_BF16_PEAK_TFS, 4),
                        "traffic": pmc_traffic("embed_conv", args.conv) if args.workload == "embed" else None,
                        "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run of this "
                                          "command, FETCH doubled per the gfx950 note; bench.py cannot profile itself) — NOT measured by this run",
                        "algorithmic_flops_per_launch": avg_fl, "avg_launch_us": round(avg_us, 2),
                        "launches_per_step": len(ctimer.rec) // max(1, nstep_timed),
                        "matrix_pipe_time_per_algorithmic_flop_in_bf16_mfma_units": mult,
                        "mfma_executed_tflops": round(tfs * mult, 1), "mfma_executed_frac": round(tfs * mult / MFMA_BF16_PEAK_TFS, 4),
                        "mfma_util_pmc_percent": mfma_util_pmc(),
                        "mfma_util_source": "profiles/pmc_traffic.json (rocprofv3 --pmc MfmaUtil pass) — NOT measured by this run"}
            wsum = wtimer.summary()
            if wsum is not None:
                wus, wfl, wtfs = wsum
                wmode = "split8" if "split8" in wtimer.modes else sorted(wtimer.modes)[0]
                wmult = {"split": 3, "split8": 2}.get(wmode, 1)
                roofline["weight_gradient"] = {
                    "kernel": "dasv::conv3x3_wgrad_kernel + conv3x3_wgrad_reduce (%s operands)" % wmode, "bound": "mfma",
                    "achieved": round(wtfs, 1), "peak": MFMA_BF16_PEAK_TFS, "unit": "TFLOP/s", "frac": round(wtfs / MFMA_BF16_PEAK_TFS, 4),
                    "avg_launch_us": round(wus, 2), "algorithmic_flops_per_launch": wfl,
                    "launches_per_step": len(wtimer.rec) // max(1, nstep_timed),
                    "matrix_pipe_time_per_algorithmic_flop_in_bf16_mfma_units": wmult,
                    "mfma_executed_frac": round(wtfs * wmult / MFMA_BF16_PEAK_TFS, 4),
                    "note": "each 64 x 64 tile streams its g and x channels for every pixel: the launch moves Cout/64 x |x| + Cin/64 x |g| "
                            "bytes through the L2 (3.1 GB for conv12); in the packed training chain the producer waves copy x (and, behind a "
                            "pool, g) instead of converting them, and the launch is held by its transposed LDS reads (~85 % of the matrix "
                            "pipe's time per chunk) — DESIGN.md 4.4b, 4.4i"}
            if world == 1 and args.workload == "embed" and args.conv == "split8" and not args.no_power_probe:
                roofline["power_probe"] = conv_power_probe(dev, B)
        # companions (never the headline `value`): the same step with the other operand modes of the convolutions, each with its
        # measured embedding error against the CPU oracle's fp32 embeddings of the same batch
        companions = None
        if world == 1 and args.workload == "embed" and args.conv in ("split8", "split") and compute_dtype == torch.float32:
            import oracle
            nref = 4
            companions, embs = {}, {}
            for mode in ("split8", "split", "fp16", "bf16"):     # timings first (the CPU oracle below lets the GPU clock down)
                net.front_end.conv_mode = mode
                if mode == args.conv:                    # the headline's own mode: the timed region above IS its measurement
                    embs[mode] = step(0).float()[:nref]
                    companions[mode] = {"ms_per_step": round(elapsed / args.steps * 1e3, 4), "value": round(B * args.steps / elapsed, 2),
                                        "unit": "utterance-embeddings/s", "this_is_the_headline_mode": True}
                    continue
                run_steps(0, 5)
                torch.cuda.synchronize()
                r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                r0.record()
                run_steps(0, args.steps)
                r1.record()
                torch.cuda.synchronize()
                embs[mode] = step(0).float()[:nref]
                msm = r0.elapsed_time(r1) / args.steps
                companions[mode] = {"ms_per_step": round(msm, 4), "value": round(B / msm * 1e3, 2), "unit": "utterance-embeddings/s"}
            sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
            ref_emb = oracle.embedding_forward(x[:nref].cpu(), sd, 4, params.heads_number).to(dev)
            for mode, emb in embs.items():
                companions[mode]["embedding_max_rel_err_vs_cpu_fp32_oracle"] = float((emb - ref_emb).abs().max() / ref_emb.abs().max())
                companions[mode]["min_cosine_vs_cpu_fp32_oracle"] = float(torch.nn.functional.cosine_similarity(emb, ref_emb, dim=-1).min())
            net.front_end.conv_mode = args.conv
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            torch.set_num_threads(host_cores())
            cpu = cpu_baseline_pool(B, T, 512, 8) if args.workload == "pool" else cpu_baseline_embed(default_params(), T)
        out = headline("utterance-embeddings/sec" if args.workload != "train" else "training utterances/sec",
                       "utterance-embeddings/s" if args.workload != "train" else "utterances/s", B, world, args.steps, args.warmup,
                       elapsed, els, dist)
        out.update({
            # the arithmetic type of the multiplicands (I/O and accumulation are fp32 everywhere): what the matrix cores multiply
            "dtype": ("f32" if args.workload == "pool" else
                      {"split8": "f32-io/f16+e4m3-split", "split": "f32-io/3xbf16-split", "torch": "f32", "fp16": "f32-io/f16",
                       "bf16": "f32-io/bf16"}[args.conv]) if compute_dtype == torch.float32 else "bf16",
            "data": "synthetic (np.random.RandomState(1234), random-init weights under torch.manual_seed(1234))",
            "config": config, "roofline": roofline, "roofline_pooling": pool_roof, "cpu_baseline": cpu,
        })
        if companions is not None:
            out["conv_operand_modes"] = companions
        if world == 1 and args.workload == "embed" and args.conv == "split8" and compute_dtype == torch.float32 and not args.no_companions:
            torch.cuda.synchronize()
            out["companion_workloads"] = companion_workloads()
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
