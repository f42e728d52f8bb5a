"""Diagnostic: the [64,300,80] embedding step eager against a hipGraph replay (graphs.EmbeddingGraph), same process."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from doubleattentionspeakerverification_amd.model import SpeakerClassifier
from doubleattentionspeakerverification_amd.graphs import EmbeddingGraph
p = argparse.Namespace(front_end="VGG4L", kernel_size=1024, pooling_method="DoubleMHA", heads_number=32, mask_prob=0.3,
                       embedding_size=400, num_spkrs=5994, scalingFactor=30.0, marginFactor=0.4, annealing=False)
torch.manual_seed(1234)
dev = torch.device("cuda:0")
net = SpeakerClassifier(p, dev).to(dev).eval()
x = torch.randn(64, 300, 80, device=dev)
g = EmbeddingGraph(net, 64, 300)
with torch.no_grad():
    ref = net.getEmbedding(x).clone()
out = g(x)
torch.cuda.synchronize()
print("graph == eager:", bool(torch.equal(out, ref)), float((out - ref).abs().max()))
best = {"eager": 1e9, "graph": 1e9}
with torch.no_grad():
    for rnd in range(6):
        for tag in (("eager", "graph") if rnd % 2 == 0 else ("graph", "eager")):
            fn = (lambda: net.getEmbedding(x)) if tag == "eager" else (lambda: g(x))
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best[tag] = min(best[tag], e0.elapsed_time(e1) / 20)
print("eager %.3f ms (%.0f utt/s)   graph %.3f ms (%.0f utt/s)" % (best["eager"], 64e3 / best["eager"], best["graph"], 64e3 / best["graph"]))
