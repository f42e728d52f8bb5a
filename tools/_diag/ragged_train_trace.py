#!/usr/bin/env python3
"""Three stacked training steps on 64 utterances of different lengths (for rocprofv3 --kernel-trace; tools/train_step_seq.py reads
the last step).  GPU box only."""
import sys, os, argparse
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
tr = DataParallelTrainer(net, lr=1e-4, weight_decay=1e-3, )
rs = np.random.RandomState(3)
lens = rs.randint(200, 401, size=64)
utts = [torch.randn(int(n), 80, device=dev) for n in lens]
lab = torch.from_numpy(rs.randint(0, 5994, size=64)).to(dev)
for _ in range(4):
    tr.train_step(utts, lab, step=0)
torch.cuda.synchronize()
