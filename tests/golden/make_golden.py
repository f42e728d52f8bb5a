#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Build-container only: imports /root/reference/scripts/{poolings,CNNs,model,loss,utils}.py
read-only (no bytecode written, nothing copied) and records inputs/weights/outputs as
small .npz fixtures.  /root/reference does not exist on the GPU box; tests read only the
.npz files.  Re-run:  python tests/golden/make_golden.py

Conventions: inputs come from np.random.RandomState(seed).standard_normal(shape) cast
to float32 (the frozen legacy stream), stored only when small; weights and outputs are
always stored.  Train-mode head masks (poolings.py:41 draws on torch.cuda.FloatTensor,
unavailable without a GPU) are produced by aliasing torch.cuda.FloatTensor to
torch.FloatTensor for the duration of the call and seeding the CPU generator; the mask
is stored explicitly so no test depends on the generator.
"""
import argparse
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference/scripts"
sys.path.insert(0, REF)
sys.modules.setdefault("soundfile", types.ModuleType("soundfile"))  # imported-but-unused by data.py/train.py

import torch  # noqa: E402

import poolings as R_pool  # noqa: E402
import CNNs as R_cnn  # noqa: E402
import loss as R_loss  # noqa: E402
import model as R_model  # noqa: E402
import utils as R_utils  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(4)


def rnd(seed, *shape, scale=1.0):
    return (np.random.RandomState(seed).standard_normal(shape) * scale).astype(np.float32)


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print("wrote %-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def tiny_params(front_end="VGG4L", kernel_size=16, heads=4, emb=24, spk=11, pooling="DoubleMHA",
                annealing=False, mask_prob=0.3):
    return argparse.Namespace(front_end=front_end, kernel_size=kernel_size, pooling_method=pooling,
                              heads_number=heads, mask_prob=mask_prob, embedding_size=emb, num_spkrs=spk,
                              scalingFactor=30.0, marginFactor=0.4, annealing=annealing)


def sd_np(sd, prefix=""):
    return {prefix + k.replace(".", "__"): v.detach().cpu().numpy() for k, v in sd.items()}


# ---------------------------------------------------------------- G1: pooling eval forward
def g1():
    cases = [(1, 1, 8, 2), (3, 17, 32, 4), (2, 37, 512, 8), (1, 19, 5120, 32), (4, 50, 640, 4), (2, 301, 96, 8)]
    out = {}
    for i, (B, T, D, H) in enumerate(cases):
        torch.manual_seed(100 + i)
        m = R_pool.DoubleMHA(D, H).eval()
        x = torch.from_numpy(rnd(200 + i, B, T, D))
        with torch.no_grad():
            o, al = m(x)
            al2, hw = m.getAlignments(x)
            ctx, _ = m.utteranceAttention(x)
        assert torch.equal(al, al2)
        out["c%d_shape" % i] = np.array([B, T, D, H])
        out["c%d_seed" % i] = np.array(200 + i)
        if x.numel() <= 40000:
            out["c%d_x" % i] = x.numpy()
        out["c%d_query" % i] = m.utteranceAttention.query.detach().numpy()
        out["c%d_att" % i] = m.headsAttention.att.detach().numpy()
        out["c%d_out" % i] = o.numpy()
        out["c%d_align" % i] = al.numpy()
        out["c%d_head_w" % i] = hw.numpy()
        out["c%d_ctx" % i] = ctx.numpy()
    out["n"] = np.array(len(cases))
    # stress: large-magnitude scores (softmax max-subtraction) and non-negative (front-end-like) input
    torch.manual_seed(150)
    m = R_pool.DoubleMHA(64, 4).eval()
    for tag, x in (("big", rnd(250, 2, 40, 64, scale=30.0)), ("relu", np.maximum(rnd(251, 2, 40, 64), 0))):
        with torch.no_grad():
            o, al = m(torch.from_numpy(x))
        out["s_%s_x" % tag], out["s_%s_out" % tag], out["s_%s_align" % tag] = x, o.numpy(), al.numpy()
    out["s_query"] = m.utteranceAttention.query.detach().numpy()
    out["s_att"] = m.headsAttention.att.detach().numpy()
    # single-head Attention pooling (poolings.py:14-27)
    torch.manual_seed(160)
    a = R_pool.Attention(48).eval()
    x = rnd(260, 3, 23, 48)
    with torch.no_grad():
        ct, sc = a(torch.from_numpy(x))
    out["a_x"], out["a_att"], out["a_ct"], out["a_score"] = x, a.att.detach().numpy(), ct.numpy(), sc.numpy()
    save("g1_pooling_eval", **out)


# ---------------------------------------------------------------- G2: train-mode fwd + bwd with explicit mask
def g2():
    out = {}
    saved = torch.cuda.FloatTensor
    torch.cuda.FloatTensor = torch.FloatTensor
    try:
        cases = [(3, 17, 32, 4, 0.3), (5, 29, 128, 8, 0.2), (2, 19, 640, 4, 0.25)]
        for i, (B, T, D, H, mp) in enumerate(cases):
            torch.manual_seed(300 + i)
            m = R_pool.DoubleMHA(D, H, mask_prob=mp).train()
            x = torch.from_numpy(rnd(400 + i, B, T, D)).requires_grad_(True)
            go = torch.from_numpy(rnd(500 + i, B, D // H))
            k = int(1 / mp)
            for seed in range(1000 + i, 1100 + i):  # pick a draw with no fully-masked row (NaN row otherwise)
                torch.manual_seed(seed)
                keep = torch.FloatTensor(B, H).random_(k) > 0
                if bool(keep.any(dim=1).all()) and not bool(keep.all()):
                    break
            torch.manual_seed(seed)
            o, al = m(x)
            (o * go).sum().backward()
            out["c%d_shape" % i] = np.array([B, T, D, H])
            out["c%d_mask_prob" % i] = np.array(mp)
            out["c%d_mask_seed" % i] = np.array(seed)
            out["c%d_x" % i] = x.detach().numpy()
            out["c%d_query" % i] = m.utteranceAttention.query.detach().numpy()
            out["c%d_att" % i] = m.headsAttention.att.detach().numpy()
            out["c%d_keep" % i] = keep.numpy()
            out["c%d_grad_out" % i] = go.numpy()
            out["c%d_out" % i] = o.detach().numpy()
            out["c%d_align" % i] = al.detach().numpy()
            out["c%d_grad_x" % i] = x.grad.numpy()
            out["c%d_grad_query" % i] = m.utteranceAttention.query.grad.numpy()
            out["c%d_grad_att" % i] = m.headsAttention.att.grad.numpy()
        out["n"] = np.array(len(cases))
        # eval-mode backward (no mask) for one case
        torch.manual_seed(350)
        m = R_pool.DoubleMHA(64, 4).eval()
        x = torch.from_numpy(rnd(450, 2, 11, 64)).requires_grad_(True)
        go = torch.from_numpy(rnd(550, 2, 16))
        o, _ = m(x)
        (o * go).sum().backward()
        out["e_x"], out["e_grad_out"] = x.detach().numpy(), go.numpy()
        out["e_query"], out["e_att"] = m.utteranceAttention.query.detach().numpy(), m.headsAttention.att.detach().numpy()
        out["e_out"], out["e_grad_x"] = o.detach().numpy(), x.grad.numpy()
        out["e_grad_query"], out["e_grad_att"] = m.utteranceAttention.query.grad.numpy(), m.headsAttention.att.grad.numpy()
        # all-heads-masked row -> NaN row
        torch.manual_seed(351)
        m = R_pool.HeadAttention(16, 4, mask_prob=0.9).train()   # k = 1 -> everything masked
        with torch.no_grad():
            ct, w = m(torch.from_numpy(rnd(451, 2, 4, 4)))
        out["allmasked_isnan"] = np.array(bool(torch.isnan(ct).all()))
    finally:
        torch.cuda.FloatTensor = saved
    save("g2_pooling_train", **out)


# ---------------------------------------------------------------- G3: VGG front-ends
def g3():
    out = {}
    for L, cls in ((3, R_cnn.VGG3L), (4, R_cnn.VGG4L)):
        for ks in (16, 32):
            torch.manual_seed(600 + L + ks)
            net = cls(ks).eval()
            out.update(sd_np(net.state_dict(), "L%d_k%d_" % (L, ks)))
            for T in (8, 31, 50):
                x = rnd(700 + T, 2, T, 80)
                with torch.no_grad():
                    y = net(torch.from_numpy(x))
                out["L%d_k%d_T%d_y" % (L, ks, T)] = y.numpy()
    # output-shape table (frames in -> frames out) and output dims
    Ts = [1, 2, 15, 16, 17, 150, 201, 299, 300, 301, 350, 800]
    net3, net4 = R_cnn.VGG3L(8).eval(), R_cnn.VGG4L(8).eval()
    with torch.no_grad():
        out["shape_T"] = np.array(Ts)
        out["shape_T3"] = np.array([net3(torch.zeros(1, t, 80)).shape[1] for t in Ts])
        out["shape_T4"] = np.array([net4(torch.zeros(1, t, 80)).shape[1] for t in Ts])
    out["dim3_80_1024"] = np.array(R_cnn.getVGG3LOutputDimension(80, outputChannel=1024))
    out["dim4_80_1024"] = np.array(R_cnn.getVGG4LOutputDimension(80, outputChannel=1024))
    out["dim4_80_512"] = np.array(R_cnn.getVGG4LOutputDimension(80, outputChannel=512))
    out["dim3_40_128"] = np.array(R_cnn.getVGG3LOutputDimension(40, outputChannel=128))
    # input gradient + weight gradient for one small case
    torch.manual_seed(650)
    net = R_cnn.VGG4L(16).train()
    x = torch.from_numpy(rnd(750, 2, 21, 80)).requires_grad_(True)
    y = net(x)
    gy = torch.from_numpy(rnd(751, *y.shape))
    (y * gy).sum().backward()
    out.update(sd_np(net.state_dict(), "g_"))
    out["g_x"], out["g_gy"], out["g_y"], out["g_grad_x"] = x.detach().numpy(), gy.numpy(), y.detach().numpy(), x.grad.numpy()
    out["g_grad_conv11_w"] = net.conv11.weight.grad.numpy()
    out["g_grad_conv42_w"] = net.conv42.weight.grad.numpy()
    out["g_grad_conv42_b"] = net.conv42.bias.grad.numpy()
    save("g3_vgg", **out)


# ---------------------------------------------------------------- G4: SpeakerClassifier
def g4():
    out = {}
    for tag, p in (("v4", tiny_params("VGG4L", 16, 4)), ("v3", tiny_params("VGG3L", 16, 8)),
                   ("mha", tiny_params("VGG4L", 16, 4, pooling="MHA")), ("att", tiny_params("VGG4L", 16, 4, pooling="Attention"))):
        torch.manual_seed(800)
        net = R_model.SpeakerClassifier(p, torch.device("cpu"))
        # give BatchNorm non-trivial running stats and affine terms
        with torch.no_grad():
            for bi, bn in enumerate((net.b1, net.b2, net.b3)):
                bn.running_mean.copy_(torch.from_numpy(rnd(810 + bi, p.embedding_size, scale=0.3)))
                bn.running_var.copy_(torch.from_numpy(np.abs(rnd(813 + bi, p.embedding_size)) + 0.5))
                bn.weight.copy_(torch.from_numpy(1 + rnd(816 + bi, p.embedding_size, scale=0.1)))
                bn.bias.copy_(torch.from_numpy(rnd(819 + bi, p.embedding_size, scale=0.1)))
        net.eval()
        x = torch.from_numpy(rnd(820, 3, 57, 80))
        lab = torch.tensor([1, 5, 10])
        with torch.no_grad():
            emb = net.getEmbedding(x)
            costh, logits = net(x, label=lab, step=0)
        out.update(sd_np(net.state_dict(), tag + "_sd_"))
        out[tag + "_emb"], out[tag + "_costh"], out[tag + "_logits"] = emb.numpy(), costh.numpy(), logits.numpy()
        out[tag + "_feature_size"] = np.array(p.feature_size)
    out["x"], out["label"] = rnd(820, 3, 57, 80), np.array([1, 5, 10])
    # two single-utterance embeddings -> cosine score (train.py:117-133)
    p = tiny_params("VGG4L", 16, 4)
    torch.manual_seed(800)
    net = R_model.SpeakerClassifier(p, torch.device("cpu")).eval()
    x1, x2 = torch.from_numpy(rnd(830, 1, 123, 80)), torch.from_numpy(rnd(831, 1, 77, 80))
    with torch.no_grad():
        e1, e2 = net.getEmbedding(x1), net.getEmbedding(x2)
        out["score_e1"], out["score_e2"] = e1.numpy(), e2.numpy()
        out["score_cos"] = R_utils.scoreCosineDistance(e1, e2).numpy()
    out.update(sd_np(net.state_dict(), "score_sd_"))
    # default-config (C3) state_dict key/shape contract
    pc3 = argparse.Namespace(front_end="VGG4L", kernel_size=1024, pooling_method="DoubleMHA", heads_number=32,
                             mask_prob=0.3, embedding_size=400, num_spkrs=5994, scalingFactor=30.0,
                             marginFactor=0.4, annealing=False)
    net = R_model.SpeakerClassifier(pc3, torch.device("cpu"))
    keys = list(net.state_dict().keys())
    out["c3_keys"] = np.array(keys)
    out["c3_shapes"] = np.array([",".join(map(str, net.state_dict()[k].shape)) for k in keys])
    out["c3_nparams"] = np.array(sum(p_.numel() for p_ in net.parameters()))
    out["c3_param_names"] = np.array([n for n, _ in net.named_parameters()])
    save("g4_model", **out)


# ---------------------------------------------------------------- G5: AM-softmax
def g5():
    out = {}
    torch.manual_seed(900)
    for ann in (False, True):
        head = R_loss.AMSoftmax(24, 37, m=0.4, s=30.0, annealing=ann)
        x = torch.from_numpy(rnd(910, 5, 24)).requires_grad_(True)
        lab = torch.tensor([0, 36, 7, 7, 12])
        tag = "ann" if ann else "plain"
        out[tag + "_W"] = head.W.detach().numpy()
        for step in (0, 100, 5000, 1000000):
            head.zero_grad()
            x.grad = None
            costh, logits = head(x, lab, step)
            lossv = torch.nn.functional.cross_entropy(logits, lab)
            lossv.backward()
            out["%s_s%d_costh" % (tag, step)] = costh.detach().numpy()
            out["%s_s%d_logits" % (tag, step)] = logits.detach().numpy()
            out["%s_s%d_loss" % (tag, step)] = lossv.detach().numpy()
            out["%s_s%d_grad_x" % (tag, step)] = x.grad.numpy().copy()
            out["%s_s%d_grad_W" % (tag, step)] = head.W.grad.numpy().copy()
            out["%s_s%d_factor" % (tag, step)] = np.array(head.getAnnealedFactor(step))
    out["x"], out["label"] = rnd(910, 5, 24), np.array([0, 36, 7, 7, 12])
    save("g5_amsoftmax", **out)


# ---------------------------------------------------------------- G6: scoring helpers
def g6():
    out = {}
    rs = np.random.RandomState(0)
    cl = rs.normal(0.6, 0.15, 2000)
    im = rs.normal(0.1, 0.15, 2000)
    sys.argv = ["x"]
    import train as R_train  # noqa: E402  (argparse only runs under __main__)
    tr = R_train.Trainer.__new__(R_train.Trainer)
    out["eer_clients"], out["eer_impostors"] = cl.astype(np.float32), im.astype(np.float32)
    out["eer"] = np.array(tr._Trainer__calculate_EER(cl.astype(np.float32), im.astype(np.float32)))
    out["eer_separated"] = np.array(tr._Trainer__calculate_EER(np.full(10, 0.9), np.full(10, -0.9)))
    ths = np.array([-0.5, 0.0, 0.3, 0.35, 0.7])
    out["score_th"] = ths
    out["score_far"] = np.array([R_utils.Score(im, t, "FAR") for t in ths])
    out["score_frr"] = np.array([R_utils.Score(cl, t, "FRR") for t in ths])
    e1, e2 = rnd(10, 4, 24), rnd(11, 4, 24)
    out["cos_e1"], out["cos_e2"] = e1, e2
    out["cos"] = R_utils.scoreCosineDistance(torch.from_numpy(e1), torch.from_numpy(e2)).numpy()
    pred = torch.from_numpy(rnd(12, 9, 5))
    lab = torch.tensor([0, 1, 2, 3, 4, 0, 1, 2, 3])
    out["acc_pred"], out["acc_label"], out["acc"] = pred.numpy(), lab.numpy(), np.array(R_utils.Accuracy(pred, lab))
    ns = argparse.Namespace(model_name="CNN", front_end="VGG4L", window_size=3.5, batch_size=64, gradientAccumulation=2,
                            learning_rate=0.0001, weight_decay=0.001, kernel_size=1024, embedding_size=400,
                            scalingFactor=30.0, marginFactor=0.4, pooling_method="DoubleMHA", heads_number=32)
    out["model_name"] = np.array(R_utils.getModelName(ns))
    save("g6_scoring", **out)


# ---------------------------------------------------------------- G7: one full training step (train.py:215-226)
def g7():
    out = {}
    saved = torch.cuda.FloatTensor
    torch.cuda.FloatTensor = torch.FloatTensor
    try:
        for opt_name in ("Adam", "SGD"):
            p = tiny_params("VGG4L", 16, 4)
            torch.manual_seed(1234)
            net = R_model.SpeakerClassifier(p, torch.device("cpu")).train()
            init = {k: v.clone() for k, v in net.state_dict().items()}
            if opt_name == "Adam":
                opt = torch.optim.Adam(net.parameters(), lr=1e-4, weight_decay=1e-3)
            else:
                opt = torch.optim.SGD(net.parameters(), lr=1e-2, weight_decay=1e-3)
            crit = torch.nn.CrossEntropyLoss()
            losses, keeps = [], []
            k = int(1 / p.mask_prob)
            for mb in range(2):  # gradientAccumulation = 2, grads summed (train.py:219-226)
                x = torch.from_numpy(rnd(1300 + mb, 4, 40, 80))
                lab = torch.tensor([(3 * mb + j) % p.num_spkrs for j in range(4)])
                torch.manual_seed(1400 + mb)
                keeps.append((torch.FloatTensor(4, p.heads_number).random_(k) > 0).numpy())
                torch.manual_seed(1400 + mb)
                pred, am = net(x, label=lab, step=0)
                loss = crit(am, lab)
                loss.backward()
                losses.append(loss.item())
            grads = {n: (None if q.grad is None else q.grad.clone()) for n, q in net.named_parameters()}
            opt.step()
            tag = opt_name.lower()
            out.update(sd_np(init, tag + "_init_"))
            out.update(sd_np(net.state_dict(), tag + "_post_"))
            out[tag + "_losses"] = np.array(losses)
            out[tag + "_keep"] = np.stack(keeps)
            out[tag + "_nograd"] = np.array([n for n, g in grads.items() if g is None])
            for n, g in grads.items():
                if g is not None:
                    out[tag + "_grad_" + n.replace(".", "__")] = g.numpy()
    finally:
        torch.cuda.FloatTensor = saved
    save("g7_train_step", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7"]
    for name in which:
        globals()[name]()
