"""Diagnostic: where the time of the ragged embedding path goes (border tables, front-end, pooling) vs a dense batch."""
import sys, os, time, argparse, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from doubleattentionspeakerverification_amd.model import SpeakerClassifier
from doubleattentionspeakerverification_amd import ragged
dev = torch.device("cuda:0")
p = argparse.Namespace(front_end="VGG4L", kernel_size=1024, pooling_method="DoubleMHA", heads_number=32, mask_prob=0.3,
                       embedding_size=400, num_spkrs=5994, scalingFactor=30.0, marginFactor=0.4, annealing=False)
torch.manual_seed(0)
net = SpeakerClassifier(p, dev).to(dev).eval()
rs = np.random.RandomState(0)
lens = rs.randint(150, 801, size=40)
utts = [torch.randn(int(n), 80, device=dev) for n in lens]
print("frames", int(lens.sum()))
def T(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("tables %.2f ms" % T(lambda: ragged.ragged_tables([int(u.shape[0]) for u in utts], 4, dev)))
print("front_end_ragged %.2f ms" % T(lambda: ragged.front_end_ragged(net.front_end, utts)))
print("embed_ragged %.2f ms" % T(lambda: ragged.embed_ragged(net, utts)))
x = torch.randn(64, 300, 80, device=dev)
with torch.no_grad():
    print("dense getEmbedding [64,300] %.2f ms" % T(lambda: net.getEmbedding(x)))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); ragged.embed_ragged(net, utts); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
