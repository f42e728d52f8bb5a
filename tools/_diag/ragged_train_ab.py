#!/usr/bin/env python3
"""Training step on a minibatch of 64 utterances of DIFFERENT lengths (T uniform in [200,400], mean 300 frames: the frame count of
the dense [64,300,80] batch), default model, 5994 speakers: the stacked path (one set of launches, border tables) against one dense
pass per distinct length, and the dense equal-length step for scale.  GPU box only."""
import sys, os, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from doubleattentionspeakerverification_amd.model import SpeakerClassifier
from doubleattentionspeakerverification_amd.dist import DataParallelTrainer

dev = torch.device("cuda:0")
p = argparse.Namespace(front_end="VGG4L", kernel_size=1024, pooling_method="DoubleMHA", heads_number=32, mask_prob=0.3, embedding_size=400,
                       num_spkrs=5994, scalingFactor=30.0, marginFactor=0.4, annealing=False)
torch.manual_seed(1234)
net = SpeakerClassifier(p, dev).to(dev).train()
tr = DataParallelTrainer(net, lr=1e-4, weight_decay=1e-3)
rs = np.random.RandomState(3)
lens = rs.randint(200, 401, size=64)
utts = [torch.randn(int(n), 80, device=dev) for n in lens]
lab = torch.from_numpy(rs.randint(0, 5994, size=64)).to(dev)
dense = torch.randn(64, 300, 80, device=dev)


def timeit(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print("frames: %d (dense batch: 19200), %d distinct lengths" % (int(lens.sum()), len(set(lens.tolist()))))
res = {}
for rnd in range(3):
    net.ragged_training = "stacked"
    res.setdefault("stacked", []).append(timeit(lambda: tr.train_step(utts, lab, step=0)))
    res.setdefault("dense [64,300,80]", []).append(timeit(lambda: tr.train_step(dense, lab, step=0)))
    if rnd == 0:
        net.ragged_training = "buckets"
        res.setdefault("one dense pass per length", []).append(timeit(lambda: tr.train_step(utts, lab, step=0), n=2))
for k, v in res.items():
    print("%-28s %8.2f ms per step (min %.2f)" % (k, sorted(v)[len(v) // 2], min(v)))
