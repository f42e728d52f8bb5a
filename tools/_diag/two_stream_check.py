import argparse, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from doubleattentionspeakerverification_amd.model import SpeakerClassifier
from doubleattentionspeakerverification_amd import ragged
DEV = torch.device("cuda:0")
p = argparse.Namespace(front_end="VGG4L", kernel_size=1024, pooling_method="DoubleMHA", heads_number=32, mask_prob=0.3,
                       embedding_size=400, num_spkrs=50, scalingFactor=30.0, marginFactor=0.4, annealing=False)
torch.manual_seed(1234)
net = SpeakerClassifier(p, DEV).to(DEV).eval()
g = np.random.RandomState(1)
groups = [[torch.randn(int(n), 80, device=DEV) for n in g.randint(150, 801, size=40)] for _ in range(6)]
def seq(fn):
    return [fn(gr) for gr in groups]
def two(fn):
    cur = torch.cuda.current_stream(DEV)
    side = [torch.cuda.Stream(DEV), torch.cuda.Stream(DEV)]
    for s in side: s.wait_stream(cur)
    outs = []
    for k, gr in enumerate(groups):
        with torch.cuda.stream(side[k & 1]):
            outs.append(fn(gr))
    for s in side: cur.wait_stream(s)
    torch.cuda.synchronize()
    return outs
def cmp(a, b):
    return [bool(torch.equal(x, y)) for x, y in zip(a, b)]
with torch.no_grad():
    fe = lambda gr: ragged.front_end_ragged(net.front_end, gr)[0]
    ref = seq(fe); torch.cuda.synchronize()
    print("front-end, two streams vs sequential:", cmp(ref, two(fe)))
    for flag in (False,):
        net.front_end.packed_activations = flag
        ref2 = seq(fe); torch.cuda.synchronize()
        print("front-end fp32 chain, two streams:", cmp(ref2, two(fe)))
    net.front_end.packed_activations = True
    em = lambda gr: ragged.embed_ragged(net, gr)
    r3 = seq(em); torch.cuda.synchronize()
    print("embed, two streams:", cmp(r3, two(em)))
    # pooling alone on fixed frames
    frames = [torch.randn(40, 30, 5120, device=DEV) for _ in groups]
    lens = torch.full((40,), 30, dtype=torch.int32, device=DEV)
    k = [0]
    def pl(gr):
        i = k[0] % len(frames); k[0] += 1
        return net.poolingLayer.pool(frames[i], lengths=lens)
    r4 = seq(pl); torch.cuda.synchronize(); k[0] = 0
    print("pooling, two streams:", cmp(r4, two(pl)))
