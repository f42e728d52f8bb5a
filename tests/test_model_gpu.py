"""GPU parity of the full path (front-end -> fused pooling -> fc/BN -> AM-softmax) against golden vectors
from the reference and the CPU oracle; BASELINE.json's bar: embeddings within 1e-3 relative (fp32),
cosine scores within 1e-4."""
import argparse

import numpy as np
import pytest
import torch

import oracle
from conftest import load_golden, rnd, sd_from_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _params(fe="VGG4L", ks=16, H=4, pool="DoubleMHA", emb=24, spk=11):
    return argparse.Namespace(front_end=fe, kernel_size=ks, pooling_method=pool, heads_number=H, mask_prob=0.3,
                              embedding_size=emb, num_spkrs=spk, scalingFactor=30.0, marginFactor=0.4, annealing=False)


def _net(tag, fe, H, pool):
    from doubleattentionspeakerverification_amd.model import SpeakerClassifier
    g = load_golden("g4_model")
    net = SpeakerClassifier(_params(fe, 16, H, pool), torch.device(DEV))
    net.load_state_dict(sd_from_golden(g, tag + "_sd_"), strict=True)
    return net.to(DEV).eval(), g


@pytest.mark.parametrize("tag,fe,H,pool", [("v4", "VGG4L", 4, "DoubleMHA"), ("v3", "VGG3L", 8, "DoubleMHA"),
                                           ("mha", "VGG4L", 4, "MHA"), ("att", "VGG4L", 4, "Attention")])
def test_golden_embedding_and_logits(tag, fe, H, pool):
    net, g = _net(tag, fe, H, pool)
    x = torch.from_numpy(g["x"]).to(DEV)
    lab = torch.from_numpy(g["label"]).to(DEV)
    with torch.no_grad():
        emb = net.getEmbedding(x)
        costh, logits = net(x, label=lab, step=0)
    ref = g[tag + "_emb"]
    assert np.abs(emb.cpu().numpy() - ref).max() / np.abs(ref).max() < 1e-4   # bar: 1e-3
    np.testing.assert_allclose(costh.cpu().numpy(), g[tag + "_costh"], atol=2e-5)
    np.testing.assert_allclose(logits.cpu().numpy(), g[tag + "_logits"], atol=1e-3)


def test_cosine_score_within_1e4_of_reference():
    from doubleattentionspeakerverification_amd.model import SpeakerClassifier
    g = load_golden("g4_model")
    net = SpeakerClassifier(_params(), torch.device(DEV))
    net.load_state_dict(sd_from_golden(g, "score_sd_"), strict=True)
    net = net.to(DEV).eval()
    with torch.no_grad():   # train.py:117-133: one utterance per forward, full length
        e1 = net.getEmbedding(torch.from_numpy(rnd(830, 1, 123, 80)).to(DEV))
        e2 = net.getEmbedding(torch.from_numpy(rnd(831, 1, 77, 80)).to(DEV))
    cos = torch.nn.functional.cosine_similarity(e1, e2, dim=-1, eps=1e-8)
    assert abs(float(cos) - float(g["score_cos"].reshape(-1)[0])) < 1e-4


def test_train_step_matches_reference_adam_and_sgd(monkeypatch):
    """Fixture G7: two accumulated micro-batches (summed grads), explicit head masks, one optimizer step.

    Gradients are held element by element, front-end included.  A ReLU network's gradient is only comparable between two
    forwards that took the same DECISIONS (ReLU signs, max-pool arg-maxes): the hand-written forward differs from the
    reference's fp32 one by ~1e-5, so a few of the ~190k decisions of this tiny model may fall the other way, and one flip moves
    the gradients of its own channel and of every layer below it by percents.  So (1) the decisions the HIP forward took are
    captured and compared with the float64 forward's own — the flips are COUNTED and printed; (2) every gradient tensor is
    compared with the float64 gradient under the HIP forward's decisions (oracle.train_step_grads, pinned to this same fixture
    by tests/test_oracle_golden.py::test_g7_train_step_gradients): no flips by construction, rtol 2e-3 on every element;
    (3) against the fixture's own gradients every tensor that no flip can reach — all of them when there is no flip, else the
    layers above the topmost flipped one — at the same rtol 2e-3."""
    from doubleattentionspeakerverification_amd.model import SpeakerClassifier
    from _decisions import DecisionCapture
    g = load_golden("g7_train_step")
    order = oracle.vgg_layer_names(4)
    for tag, mk in (("adam", lambda ps: torch.optim.Adam(ps, lr=1e-4, weight_decay=1e-3)),
                    ("sgd", lambda ps: torch.optim.SGD(ps, lr=1e-2, weight_decay=1e-3))):
        p = _params()
        net = SpeakerClassifier(p, torch.device(DEV))
        net.load_state_dict(sd_from_golden(g, tag + "_init_"), strict=True)
        net = net.to(DEV).train()
        assert net.front_end.conv_mode == "split8"      # the shipped training default
        opt = mk(net.parameters())
        crit = torch.nn.CrossEntropyLoss()
        losses, decs, batches = [], [], []
        with DecisionCapture(monkeypatch, net.front_end) as cap:
            for mb in range(2):
                cap.reset()
                x = torch.from_numpy(rnd(1300 + mb, 4, 40, 80)).to(DEV)
                lab = torch.tensor([(3 * mb + j) % p.num_spkrs for j in range(4)], device=DEV)
                keep = torch.from_numpy(g[tag + "_keep"][mb]).to(DEV)
                # forward with the reference's mask draw made explicit
                enc = net.front_end(x)
                pooled, _ = net.poolingLayer(enc, head_keep=keep)
                e1 = torch.relu(net.fc1(pooled))
                e2 = net.b2(torch.relu(net.fc2(e1)))
                _, am = net.predictionLayer(net.preLayer(e2), lab, 0)
                loss = crit(am, lab)
                loss.backward()
                losses.append(float(loss.detach()))
                decs.append(cap.decisions())
                batches.append((x.cpu(), lab.cpu(), keep.cpu()))
        np.testing.assert_allclose(losses, g[tag + "_losses"], rtol=2e-5)
        nograd = sorted(n for n, q in net.named_parameters() if q.grad is None)
        assert nograd == sorted(str(n) for n in g[tag + "_nograd"])
        sd0 = sd_from_golden(g, tag + "_init_")
        _, _, own = oracle.train_step_grads(sd0, batches, 4, p.marginFactor, p.scalingFactor)
        flips = [oracle.count_flips(own[i], decs[i], 4) for i in range(2)]
        n_dec = sum(int(v.numel()) for k, v in decs[0].items()) * 2
        total = sum(sum(f.values()) for f in flips)
        top = max([order.index(k) for f in flips for k, v in f.items() if v], default=-1)
        print("G7 %s: %d of %d forward decisions differ from the float64 forward's: %s" % (tag, total, n_dec, flips))
        assert total <= 40, flips          # ~1e-5 forward differences flip a handful, not hundreds
        _, g64, _ = oracle.train_step_grads(sd0, batches, 4, p.marginFactor, p.scalingFactor, decisions=decs)
        # Gradients: behind train-mode BatchNorm the per-feature gradient sums to zero over the batch, so bias-like
        # (1-D) gradients are the rounding residue of O(10) terms cancelling — the reference's own values there
        # are multiples of 2^-15 — and are compared with an absolute bound; weights relative to their scale.
        for n, q in net.named_parameters():
            if q.grad is not None:
                ref = g[tag + "_grad_" + n.replace(".", "__")]
                atol = 2e-3 if ref.ndim == 1 else 2e-5 * max(1.0, np.abs(ref).max())
                if n.startswith("front_end.") and ref.ndim > 1:
                    # split8 arithmetic: a product is good to 1-2.5e-5 of its tensor's scale (tests/test_split8_gpu.py holds the
                    # data gradient to 4e-5 and the weight gradient to 2e-5 / 6.2e-5 of the output scale), forward and backward
                    # through eight layers: small elements of a gradient tensor are held to 1e-4 of the tensor's largest
                    atol = 1e-4 * max(1.0, np.abs(ref).max())
                got = q.grad.cpu().numpy()
                np.testing.assert_allclose(got, g64[n].numpy(), rtol=2e-3, atol=atol, err_msg=n + " (same decisions)")
                layer = n.split(".")[1] if n.startswith("front_end.") else None
                if layer is None or order.index(layer) > top:
                    np.testing.assert_allclose(got, ref, rtol=2e-3, atol=atol, err_msg=n)
        init = sd_from_golden(g, tag + "_init_")
        opt.step()
        post = sd_from_golden(g, tag + "_post_")
        for k, v in net.state_dict().items():
            if not v.dtype.is_floating_point:
                assert int(v) == int(post[k]), k
                continue
            upd, upd_ref = v.cpu().numpy() - init[k].numpy(), post[k].numpy() - init[k].numpy()
            if tag == "sgd":   # update is linear in the gradient: lr * (grad + wd * p)
                np.testing.assert_allclose(upd, upd_ref, rtol=2e-3, atol=1e-2 * 2e-3 + 1e-7, err_msg=k)
            else:              # Adam's first step is lr * sign-like(grad): compare where the gradient is well above noise
                gname = tag + "_grad_" + k.replace(".", "__")
                if gname in g.files:
                    strong = np.abs(g[gname]) > 2e-2
                    np.testing.assert_allclose(upd[strong], upd_ref[strong], rtol=0, atol=2e-5, err_msg=k)
                assert np.abs(upd - upd_ref).max() <= 2.05e-4, k   # never more than 2 * lr apart
                if k.startswith(("b1.", "b3.")):
                    assert np.abs(upd).max() == 0.0, k                # no grad -> untouched


def test_variable_length_minibatch_train_step_vs_float64(monkeypatch):
    """BASELINE.json north_star: "training shards minibatches of variable-length log-mel utterances".  One training step on a
    minibatch of utterances of DIFFERENT lengths (a list): every utterance goes through front-end and pooling at its own length
    — no padding, which the reference's mask-free model could not undo (it can only run unequal utterances one per forward,
    train.py:117-133) — and the dense tail, BatchNorm statistics and the AM-softmax head see ONE batch (model.py:56-71).  Loss and
    every gradient against the float64 restatement of exactly that (oracle.classifier_train_forward on a list), under the
    decisions the HIP forward took; the decisions compared with the float64 forward's own and the flips counted."""
    from doubleattentionspeakerverification_amd.model import SpeakerClassifier
    from doubleattentionspeakerverification_amd.batching import bucket_by_length
    from _decisions import DecisionCapture
    p = _params()
    torch.manual_seed(77)
    net = SpeakerClassifier(p, torch.device(DEV)).to(DEV).train()
    lens = [40, 57, 40, 33, 57, 64, 40]
    utts = [torch.from_numpy(rnd(2100 + i, n, 80)) for i, n in enumerate(lens)]
    lab = torch.tensor([3, 0, 7, 7, 10, 1, 5])
    keep = torch.rand(len(lens), p.heads_number) > 0.3
    keep[:, 1] = True
    buckets = bucket_by_length(lens, max_batch=len(lens))
    assert sorted(len(b) for b in buckets) == [1, 1, 2, 3]
    queue = [keep[torch.tensor(b)].to(DEV) for b in buckets]
    monkeypatch.setattr(net.poolingLayer.headsAttention, "draw_head_keep", lambda batch, heads, device: queue.pop(0))
    with DecisionCapture(monkeypatch, net.front_end) as cap:
        loss, _ = net.forward_loss([u.to(DEV) for u in utts], lab.to(DEV), 0)
        loss.backward()
        per_pass = cap.passes()
    assert not queue and net.last_ragged_buckets == buckets and len(per_pass) == len(buckets)
    decs = [None] * len(lens)
    for b, d in zip(buckets, per_pass):
        for j, i in enumerate(b):
            decs[i] = {k: v[j:j + 1] for k, v in d.items()}
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    l_own, _, own = oracle.train_step_grads(sd, [(utts, lab, keep)], 4, p.marginFactor, p.scalingFactor)
    flips = sum(sum(oracle.count_flips(own[0][i], decs[i], 4).values()) for i in range(len(lens)))
    l64, g64, _ = oracle.train_step_grads(sd, [(utts, lab, keep)], 4, p.marginFactor, p.scalingFactor, decisions=[decs])
    print("variable-length minibatch: loss %.6f (float64 %.6f), %d forward decisions differ from the float64 forward's" % (float(loss.detach()), l64[0], flips))
    assert abs(float(loss) - l64[0]) < 2e-5 * abs(l64[0]) and flips <= 40
    for n, q in net.named_parameters():
        if n.startswith(("b1.", "b3.")):
            assert q.grad is None
            continue
        ref = g64[n].numpy()
        # (1-D gradients behind the train-mode BatchNorm — fc2.bias above all — are exactly zero in exact arithmetic: what an fp32
        # run leaves there is the cancellation residue of O(10) terms, multiples of 2^-12 at this loss scale; absolute bound)
        atol = 4e-3 if ref.ndim == 1 else 1e-4 * max(1.0, np.abs(ref).max())
        np.testing.assert_allclose(q.grad.cpu().numpy(), ref, rtol=2e-3, atol=atol, err_msg=n)
    # the same list through the data-parallel trainer (one rank): a step is taken, the loss is the same
    from doubleattentionspeakerverification_amd.dist import DataParallelTrainer
    queue.extend(keep[torch.tensor(b)].to(DEV) for b in buckets)
    monkeypatch.setattr(net.poolingLayer.headsAttention, "draw_head_keep", lambda batch, heads, device: queue.pop(0))   # (undone with the capture)
    net.zero_grad(set_to_none=True)
    tr = DataParallelTrainer(net, lr=1e-3, weight_decay=0.0, optimizer="SGD")
    before = net.fc1.weight.detach().clone()
    loss2, _ = tr.train_step([u.to(DEV) for u in utts], lab.to(DEV), step=0)
    assert abs(float(loss2) - float(loss)) < 1e-5 * abs(float(loss)) and float((net.fc1.weight - before).abs().max()) > 0


@pytest.mark.parametrize("wiring", ["packed, first block as one node", "packed", "fp32"])
def test_variable_length_minibatch_in_one_set_of_launches_vs_float64_and_vs_buckets(wiring, monkeypatch):
    """The same semantics as the test above — every utterance of a training minibatch at its own length, one dense tail — but as
    ONE set of launches: the utterances stacked along time, the front-end's forward, data-gradient, weight-gradient, pool and conv11
    kernels taking the utterance borders from tables (ragged.utterance_vectors_ragged_train; the weight gradient subtracts the
    cross-border products).  VGG4L at the reference's default width (the stacked path needs channel counts the kernels take without
    padding).  (1) Loss and every gradient element against the float64 oracle on the same list under the decisions the HIP forward
    took, flips counted; (2) against the one-dense-pass-per-length path of the same model: same loss, gradients to 4e-5.
    packed: the stacked batch on PACKED activations (CNNs._forward_map_packed_ragged_train: packed tensors between the layers, the pool's
    backward writing a packed gradient, packed operands in the weight gradients and the packed mask in the data gradients — the shipped
    wiring — with conv11 + conv12 as one autograd node whose backward forms conv11's gradients inside conv12's data-gradient launch,
    conv.first_block_ragged_train, or with conv11 as its own node) or on fp32 activations; (2) holds element by element only between
    forwards of the same arithmetic (the fp32 wiring and the per-length passes), with the packed chain the loss."""
    packed = wiring != "fp32"
    from doubleattentionspeakerverification_amd.model import SpeakerClassifier
    from _decisions import DecisionCapture
    p = argparse.Namespace(front_end="VGG4L", kernel_size=1024, pooling_method="DoubleMHA", heads_number=32, mask_prob=0.3,
                           embedding_size=64, num_spkrs=20, scalingFactor=30.0, marginFactor=0.4, annealing=False)
    torch.manual_seed(5)
    net = SpeakerClassifier(p, torch.device(DEV)).to(DEV).train()
    lens = [24, 17, 24, 33, 9]
    utts = [torch.from_numpy(rnd(2300 + i, n, 80)) for i, n in enumerate(lens)]
    lab = torch.tensor([3, 0, 7, 19, 1])
    keep = torch.rand(len(lens), p.heads_number) > 0.3
    keep[:, 1] = True
    assert net.ragged_training == "stacked" and net.front_end.ragged_training_ok()
    net.front_end.packed_training = packed
    net.front_end.fuse_first_backward = wiring.endswith("one node")
    took = []
    inner = net.front_end._forward_map_packed_ragged_train
    monkeypatch.setattr(net.front_end, "_forward_map_packed_ragged_train", lambda *a: (took.append(1), inner(*a))[1])
    monkeypatch.setattr(net.poolingLayer.headsAttention, "draw_head_keep", lambda batch, heads, device: keep.to(device))
    with DecisionCapture(monkeypatch, net.front_end) as cap:
        loss, _ = net.forward_loss([u.to(DEV) for u in utts], lab.to(DEV), 0)
        loss.backward()
        assert net.last_ragged_buckets is None and len(cap.events) == 12          # ONE pass over the stacked batch
        assert len(took) == (1 if packed else 0)
        decs = cap.decisions_stacked(lens)
    grads = {n: q.grad.detach().clone() for n, q in net.named_parameters() if q.grad is not None}
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    torch.set_num_threads(8)
    _, _, own = oracle.train_step_grads(sd, [(utts, lab, keep)], 4, p.marginFactor, p.scalingFactor)
    flips = sum(sum(oracle.count_flips(own[0][i], decs[i], 4).values()) for i in range(len(lens)))
    l64, g64, _ = oracle.train_step_grads(sd, [(utts, lab, keep)], 4, p.marginFactor, p.scalingFactor, decisions=[decs])
    print("stacked variable-length minibatch: loss %.6f (float64 %.6f), %d decisions differ from the float64 forward's"
          % (float(loss.detach()), l64[0], flips))
    assert abs(float(loss.detach()) - l64[0]) < 2e-5 * abs(l64[0]) and flips <= 40
    for n, g in grads.items():
        ref = g64[n].numpy()
        atol = 4e-3 if ref.ndim == 1 else 1e-4 * max(1.0, np.abs(ref).max())
        np.testing.assert_allclose(g.cpu().numpy(), ref, rtol=2e-3, atol=atol, err_msg=n)
    # (2) the bucketed path of the same model: one dense pass per distinct length
    from doubleattentionspeakerverification_amd.batching import bucket_by_length
    buckets = bucket_by_length(lens, max_batch=len(lens))
    queue = [keep[torch.tensor(b)].to(DEV) for b in buckets]
    monkeypatch.setattr(net.poolingLayer.headsAttention, "draw_head_keep", lambda batch, heads, device: queue.pop(0))
    net.ragged_training = "buckets"
    net.zero_grad(set_to_none=True)
    loss_b, _ = net.forward_loss([u.to(DEV) for u in utts], lab.to(DEV), 0)
    loss_b.backward()
    assert net.last_ragged_buckets == buckets and abs(float(loss_b.detach()) - float(loss.detach())) < 1e-5 * abs(float(loss.detach()))
    for n, q in net.named_parameters():
        if q.grad is not None and not packed:
            a, b = grads[n], q.grad
            assert float((a - b).abs().max()) <= 4e-5 * max(1.0, float(b.abs().max())) + (4e-3 if b.dim() == 1 else 0.0), n


def test_default_config_embedding_vs_oracle_and_bf16_error():
    """Reference default model (VGG4L/1024, 32 heads, emb 400) on a small batch: fp32 within 1e-3 relative of
    the CPU oracle; the bf16-autocast path's error is reported, not hidden."""
    from doubleattentionspeakerverification_amd.model import SpeakerClassifier
    p = argparse.Namespace(front_end="VGG4L", kernel_size=1024, pooling_method="DoubleMHA", heads_number=32, mask_prob=0.3,
                           embedding_size=400, num_spkrs=5994, scalingFactor=30.0, marginFactor=0.4, annealing=False)
    torch.manual_seed(1234)
    net = SpeakerClassifier(p, torch.device(DEV)).to(DEV).eval()
    x = torch.from_numpy(rnd(5, 2, 150, 80))
    with torch.no_grad():
        emb = net.getEmbedding(x.to(DEV)).cpu()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            emb16 = net.getEmbedding(x.to(DEV)).float().cpu()
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    torch.set_num_threads(8)
    with torch.no_grad():
        ref = oracle.embedding_forward(x, sd, 4, 32)
    rel = float((emb - ref).abs().max() / ref.abs().max())
    rel16 = float((emb16 - ref).abs().max() / ref.abs().max())
    cos = float(torch.nn.functional.cosine_similarity(emb, ref, dim=-1).min())
    cos16 = float(torch.nn.functional.cosine_similarity(emb16, ref, dim=-1).min())
    print("default-config embedding: fp32 rel err %.3e (cos %.7f); bf16 autocast rel err %.3e (cos %.6f)" % (rel, cos, rel16, cos16))
    assert rel < 1e-3 and 1 - cos < 1e-4
    assert rel16 < 5e-2 and 1 - cos16 < 1e-2   # bf16: stated, looser bar


def test_embedding_example_flow_from_reference_style_checkpoint(tmp_path):
    """BASELINE config 1 / SURVEY row G1: the flow of getEmbeddingExample.py:23-39 — torch.load of a checkpoint dict
    laid out by utils.py:33-40 ({'model','optimizer','settings','epoch','step'}, settings = argparse.Namespace),
    SpeakerClassifier(settings, device), strict load_state_dict, eval, getEmbedding on one [1,297,80] utterance
    (3 s at 16 kHz, hop 160, n_fft 512, center=False; CMN as featureExtractor.py:25-26)."""
    import sys
    sys.path.append("./scripts/")
    g = load_golden("g4_model")
    sd = sd_from_golden(g, "score_sd_")
    settings = _params()
    ckpt = {"model": sd, "optimizer": {"state": {}, "param_groups": []}, "settings": settings, "epoch": 3, "step": 1234}
    path = str(tmp_path / "CNN_VGG4L_1234.chkpt")
    torch.save(ckpt, path)
    from model import SpeakerClassifier as SC   # the scripts/ shim, as `from model import *` would give
    device = torch.device(DEV)
    net_dict = torch.load(path, map_location=device)   # weights_only default: Namespace allow-listed by our model.py
    opt = net_dict["settings"]
    net = SC(opt, device)
    net.load_state_dict(net_dict["model"])
    net.to(device)
    net.eval()
    feats = 2.0 * rnd(77, 297, 80)
    feats = feats - feats.mean(axis=0, keepdims=True)
    with torch.no_grad():
        inputs = torch.FloatTensor(feats).to(device).unsqueeze(0)
        emb = net.getEmbedding(inputs)
    assert emb.shape == (1, 24)
    with torch.no_grad():
        ref = oracle.embedding_forward(torch.from_numpy(feats).unsqueeze(0), sd, 4, 4)
    assert float((emb.cpu() - ref).abs().max() / ref.abs().max()) < 1e-4


def test_variable_length_bucketed_embeddings_equal_one_at_a_time_runs():
    """BASELINE config 5 semantics: T in [150,800], equal-length buckets == the reference's one-utterance-per-forward
    scoring (train.py:107-133); also the fp16-autocast path's error is stated."""
    from doubleattentionspeakerverification_amd.batching import embed_utterances
    from doubleattentionspeakerverification_amd.model import SpeakerClassifier
    torch.manual_seed(7)
    net = SpeakerClassifier(_params(ks=32, H=8, emb=32), torch.device(DEV)).to(DEV).eval()
    rs = np.random.RandomState(11)
    pool = [150, 151, 299, 300, 800]
    lengths = [pool[i % 5] for i in range(23)]
    utts = [torch.from_numpy(rnd(900 + i, n, 80)) for i, n in enumerate(lengths)]
    got = embed_utterances(net, utts, max_batch=8)
    assert sorted(got) == list(range(23))
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    worst = 0.0
    for i in (0, 1, 2, 3, 4, 7, 13, 22):
        with torch.no_grad():
            one = net.getEmbedding(utts[i].unsqueeze(0).to(DEV))[0].cpu()
            ref = oracle.embedding_forward(utts[i].unsqueeze(0), sd, 4, 8)[0]
        worst = max(worst, float((got[i].cpu() - one).abs().max() / one.abs().max()))
        assert float((got[i].cpu() - ref).abs().max() / ref.abs().max()) < 1e-4
    assert worst < 1e-5, worst
    # two ranks own disjoint halves and together cover everything
    a = embed_utterances(net, utts, max_batch=8, rank=0, world=2)
    b = embed_utterances(net, utts, max_batch=8, rank=1, world=2)
    assert set(a) | set(b) == set(range(23)) and not (set(a) & set(b))
    h = embed_utterances(net, utts, max_batch=8, autocast_dtype=torch.float16)
    rel16 = max(float((h[i].cpu() - got[i].cpu()).abs().max() / got[i].cpu().abs().max()) for i in range(23))
    print("fp16-autocast embeddings vs fp32: max rel err %.3e" % rel16)
    assert rel16 < 2e-2


@pytest.mark.parametrize("cin,cout,H,W,B", [(64, 128, 9, 7, 3), (128, 128, 33, 80, 2), (256, 512, 19, 5, 4), (128, 256, 1, 1, 1),
                                           (1024, 1024, 38, 10, 1), (512, 512, 75, 20, 1), (192, 128, 5, 3, 1),
                                           (64, 64, 21, 10, 2), (32, 64, 9, 20, 2), (64, 32, 9, 20, 2), (4, 8, 7, 5, 1), (2, 2, 6, 5, 2)])
def test_hip_conv3x3_bias_relu_vs_fp64_conv(cin, cout, H, W, B):
    """Hand-written implicit-GEMM conv (csrc/conv3x3.hip) against a float64 convolution: split mode is fp32-grade
    (<= 2e-5 of the output scale), bf16 mode is stated (<= 1e-2); includes image borders, M tails, 1x1 maps and the
    single-utterance shapes that take the split-K path (channel slabs divided over workgroups, 3 slabs over 3 included)."""
    from doubleattentionspeakerverification_amd import conv
    x = torch.from_numpy(rnd(1, B, cin, H, W)).to(DEV).contiguous(memory_format=torch.channels_last)
    w = torch.from_numpy(rnd(2, cout, cin, 3, 3, scale=(2.0 / (9 * cin)) ** 0.5)).to(DEV)
    b = torch.from_numpy(rnd(3, cout, scale=0.1)).to(DEV)
    ref = torch.relu(torch.nn.functional.conv2d(x.double().cpu(), w.double().cpu(), b.double().cpu(), padding=1))
    scale = float(ref.abs().max())
    with torch.no_grad():
        # channel counts that are not multiples of (64, 128) run as the zero-padded layer: the result carries padded_channels(cout)
        # channels, the surplus identically zero (the reference takes any width, CNNs.py:27-32 / 59-66)
        full = conv.conv3x3_bias_relu(x, w, b, None, "split8")
        assert full.shape[1] == conv.padded_channels(cout) and (cout == full.shape[1] or float(full[:, cout:].abs().max()) == 0.0)
        ys = conv.conv3x3_bias_relu(x, w, b, None, "split")[:, :cout].cpu().double()
        y8 = full[:, :cout].cpu().double()
        yb = conv.conv3x3_bias_relu(x, w, b, None, "bf16")[:, :cout].cpu().double()
        yh = conv.conv3x3_bias_relu(x, w, b, None, "fp16")[:, :cout].cpu().double()
    assert ys.shape == ref.shape
    assert float((ys - ref).abs().max()) / scale < 2e-5
    # fp16 products + block-scaled e4m3 cross terms: fp32-grade too (emulated and measured ~1e-5; plain fp16 is ~3e-4)
    assert y8.shape == ref.shape and float((y8 - ref).abs().max()) / scale < 4e-5
    assert float((yb - ref).abs().max()) / scale < 1e-2
    assert float((yh - ref).abs().max()) / scale < 1.25e-3   # fp16 operands: 8x tighter than bf16
    if cin == 64 and cout == 128:   # a 1e6 activation: the fp16 mode saturates instead of overflowing (finite map, stated); split8 scales the
        xb = x.clone()   # tensor by its recorded maximum and stays fp32-grade (the reference, CNNs.py:72-86, is exact fp32)
        xb[0, 0, 0, 0] = 1e6
        refb = torch.relu(torch.nn.functional.conv2d(xb.double().cpu(), w.double().cpu(), b.double().cpu(), padding=1))
        with torch.no_grad():
            assert torch.isfinite(conv.conv3x3_bias_relu(xb, w, b, None, "fp16")).all()
            y8b = conv.conv3x3_bias_relu(xb, w, b, None, "split8")[:, :cout].cpu().double()
        assert float((y8b - refb).abs().max()) / float(refb.abs().max()) < 4e-5


def test_split8_wide_map_fallback():
    """Maps wider than 80 columns run in split mode (the fp16 + e4m3 form exists in the window kernel only).  The accuracy of
    split8 over activation / weight magnitudes is held to the fp64 convolution in tests/test_split8_gpu.py."""
    from doubleattentionspeakerverification_amd import conv
    xw = torch.from_numpy(rnd(4, 1, 64, 3, 96)).to(DEV).contiguous(memory_format=torch.channels_last)
    ww = torch.from_numpy(rnd(5, 128, 64, 3, 3, scale=0.05)).to(DEV)
    with torch.no_grad():
        assert torch.equal(conv.conv3x3_bias_relu(xw, ww, None, None, "split8"), conv.conv3x3_bias_relu(xw, ww, None, None, "split"))


def test_hip_conv_backward_matches_torch_autograd():
    from doubleattentionspeakerverification_amd import conv
    x = torch.from_numpy(rnd(4, 2, 64, 10, 6)).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = torch.from_numpy(rnd(5, 128, 64, 3, 3, scale=0.05)).to(DEV).requires_grad_(True)
    b = torch.from_numpy(rnd(6, 128, scale=0.1)).to(DEV).requires_grad_(True)
    gy = torch.from_numpy(rnd(7, 2, 128, 10, 6)).to(DEV)
    y = conv.conv3x3_bias_relu(x, w, b, None, "split")
    (y * gy).sum().backward()
    g1 = [t.grad.clone() for t in (x, w, b)]
    for t in (x, w, b):
        t.grad = None
    yr = torch.relu(torch.nn.functional.conv2d(x, w, b, padding=1))
    (yr * gy).sum().backward()
    for a, r in zip(g1, (x.grad, w.grad, b.grad)):
        np.testing.assert_allclose(a.cpu().numpy(), r.cpu().numpy(), rtol=1e-3, atol=1e-3 * float(r.abs().max()))


def test_hip_conv11_and_maxpool_match_torch():
    from doubleattentionspeakerverification_amd import conv
    x = torch.from_numpy(rnd(21, 3, 37, 80)).to(DEV)
    w = torch.from_numpy(rnd(22, 128, 1, 3, 3, scale=0.3)).to(DEV)
    b = torch.from_numpy(rnd(23, 128, scale=0.1)).to(DEV)
    ref = torch.relu(torch.nn.functional.conv2d(x.unsqueeze(1), w, b, padding=1))
    y = conv.conv3x3_c1_bias_relu(x, w, b)
    np.testing.assert_allclose(y.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=1e-5)
    for shape in ((2, 128, 37, 80), (3, 64, 19, 5), (1, 16, 1, 1), (2, 32, 10, 3)):
        t = torch.from_numpy(rnd(24, *shape)).to(DEV).contiguous(memory_format=torch.channels_last)
        pr = torch.nn.functional.max_pool2d(t, 2, stride=2, ceil_mode=True)
        pm = conv.maxpool2x2_ceil(t)
        assert pm.shape == pr.shape and torch.equal(pm, pr)


def test_hip_conv_wide_map_uses_tile_kernel_and_matches():
    """W > 80 does not fit the window kernel's LDS budget: the per-tap tile kernel takes over, same results."""
    from doubleattentionspeakerverification_amd import conv
    x = torch.relu(torch.from_numpy(rnd(31, 2, 64, 5, 97))).to(DEV).contiguous(memory_format=torch.channels_last)
    w = torch.from_numpy(rnd(32, 128, 64, 3, 3, scale=0.06)).to(DEV)
    b = torch.from_numpy(rnd(33, 128, scale=0.1)).to(DEV)
    ref = torch.relu(torch.nn.functional.conv2d(x.double().cpu(), w.double().cpu(), b.double().cpu(), padding=1))
    with torch.no_grad():
        y = conv.conv3x3_bias_relu(x, w, b, None, "split").cpu().double()
    assert float((y - ref).abs().max()) / float(ref.abs().max()) < 2e-5


def test_frontend_hip_path_equals_torch_path_on_default_width():
    """VGG4L at kernel_size 1024 (the reference default): all-HIP inference front-end vs PyTorch-ROCm convs."""
    from doubleattentionspeakerverification_amd.CNNs import VGG4L
    torch.manual_seed(3)
    net = VGG4L(1024).to(DEV).eval()
    x = torch.from_numpy(rnd(41, 2, 77, 80)).to(DEV)
    with torch.no_grad():
        net.conv_mode = "torch"
        ref = net(x)
        net.conv_mode = "split"
        y = net(x)
        net.conv_mode = "bf16"
        yb = net(x)
        net.conv_mode = "fp16"
        yh = net(x)
    assert y.shape == ref.shape == (2, 5, 5120)
    rel = float((y - ref).abs().max() / ref.abs().max())
    relb = float((yb - ref).abs().max() / ref.abs().max())
    relh = float((yh - ref).abs().max() / ref.abs().max())
    print("front-end output: split-vs-torch rel err %.2e, bf16-vs-torch %.2e, fp16-vs-torch %.2e" % (rel, relb, relh))
    assert rel < 5e-5 and relb < 3e-2 and relh < 4e-3


def test_fp16_mfma_mode_embedding_error_and_training_step():
    """BASELINE config 5's "fp16 MFMA path": conv operands rounded to fp16 (fp32 accumulate, fp32 activations in HBM).
    The embedding error against the fp32-grade path is stated, and a training step runs (gradients: bf16 operands)."""
    from doubleattentionspeakerverification_amd.model import SpeakerClassifier
    p = argparse.Namespace(front_end="VGG4L", kernel_size=1024, pooling_method="DoubleMHA", heads_number=32, mask_prob=0.3,
                           embedding_size=400, num_spkrs=100, scalingFactor=30.0, marginFactor=0.4, annealing=False)
    torch.manual_seed(7)
    net = SpeakerClassifier(p, torch.device(DEV)).to(DEV).eval()
    x = torch.from_numpy(rnd(51, 3, 150, 80)).to(DEV)
    with torch.no_grad():
        ref = net.getEmbedding(x)
        net.front_end.conv_mode = "fp16"
        emb = net.getEmbedding(x)
        net.front_end.conv_mode = "bf16"
        embb = net.getEmbedding(x)
    rel = float((emb - ref).abs().max() / ref.abs().max())
    relb = float((embb - ref).abs().max() / ref.abs().max())
    cos = float(torch.nn.functional.cosine_similarity(emb, ref, dim=-1).min())
    print("fp16-operand front-end: embedding rel err %.3e vs split mode (cos %.7f); bf16 operands %.3e" % (rel, cos, relb))
    assert rel < 1e-3 and 1 - cos < 1e-4 and relb < 1e-2
    net.train()
    y = torch.tensor([1, 5, 7], device=DEV)
    grads = {}
    for mode in ("fp16", "split"):
        net.front_end.conv_mode = mode
        net.zero_grad()
        torch.manual_seed(11)                      # same train-time head mask in both runs
        pred, _ = net(x, label=y, step=0)
        torch.nn.functional.cross_entropy(pred, y).backward()
        grads[mode] = torch.cat([net.front_end.conv32.weight.grad.flatten(), net.front_end.conv12.weight.grad.flatten()]).clone()
    assert torch.isfinite(grads["fp16"]).all()
    gcos = float(torch.nn.functional.cosine_similarity(grads["fp16"], grads["split"], dim=0))
    print("fp16-mode conv weight gradients vs split mode: cosine %.5f" % gcos)
    assert gcos > 0.995


def test_data_parallel_trainer_single_rank_matches_plain_optimizer_step():
    """world_size 1 on the GPU: DataParallelTrainer (flat gradient buffer, hidden unused parameters, summed accumulation)
    must take exactly the step a plain torch loop takes on the same model (train.py:215-226 semantics)."""
    import copy
    from doubleattentionspeakerverification_amd.dist import DataParallelTrainer
    from doubleattentionspeakerverification_amd.model import SpeakerClassifier
    torch.manual_seed(11)
    p = _params(ks=64, H=8, emb=32, spk=20)
    net_a = SpeakerClassifier(p, torch.device(DEV)).to(DEV).train()
    net_b = copy.deepcopy(net_a)
    xs = [torch.from_numpy(rnd(50 + i, 6, 48, 80)).to(DEV) for i in range(2)]
    ys = [torch.tensor([(3 * i + j) % 20 for j in range(6)], device=DEV) for i in range(2)]
    keep = [(torch.rand(6, 8, device=DEV) > 0.3) for _ in range(2)]
    for k in keep:
        k[:, 0] = True
    def fwd(net, x, y, k):   # forward with an explicit head mask so both nets see the same draw
        enc = net.front_end(x)
        pooled, _ = net.poolingLayer(enc, head_keep=k)
        e2 = net.b2(torch.relu(net.fc2(torch.relu(net.fc1(pooled)))))
        return net.predictionLayer(net.preLayer(e2), y, 0)
    # plain loop
    opt = torch.optim.Adam(net_b.parameters(), lr=1e-3, weight_decay=1e-3)
    for x, y, k in zip(xs, ys, keep):
        torch.nn.functional.cross_entropy(fwd(net_b, x, y, k)[1], y).backward()
    opt.step()
    # trainer (accum = 2), same masks via a patched forward
    tr = DataParallelTrainer(net_a, lr=1e-3, weight_decay=1e-3, optimizer="Adam", accum=2, bucket_bytes=1 << 16)
    tr.fused_loss = False      # same head arithmetic as the plain step: Adam's first update is sign-like, so rounding-level
    #                            gradient differences (the fused head agrees to ~1e-7) would show as 2 * lr
    it = iter(keep)
    net_a.forward = lambda x, label=None, step=0: fwd(net_a, x, label, next(it))
    for x, y in zip(xs, ys):
        loss, _ = tr.train_step(x, y, step=0)
    assert bool(torch.isfinite(loss))
    for (n, a), (_, b) in zip(net_a.named_parameters(), net_b.named_parameters()):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=1e-5, atol=1e-6, err_msg=n)
    assert float(tr.reducer.flat.abs().sum()) == 0.0     # gradients cleared for the next step


def test_a_fused_optimizer_step_is_seen_by_the_next_forward():
    """torch's fused optimizers (`Adam(..., fused=True)`, what DataParallelTrainer uses on the GPU) update the parameters WITHOUT
    moving their version counters, which the cache of packed convolution weights is keyed by: a training forward therefore packs
    the planes from the parameters it is handed, every time, and a no-grad forward after the step must see the new weights too.
    Three steps of train.py:215-226 with a large learning rate; after every step the front-end's output (training forward and
    no-grad forward) must equal the output of a FRESH module loaded with the current state_dict."""
    from doubleattentionspeakerverification_amd.CNNs import VGG4L
    torch.manual_seed(5)
    net = VGG4L(128).to(DEV).train()
    opt = torch.optim.Adam(net.parameters(), lr=5e-2, fused=True)
    x = torch.from_numpy(rnd(77, 3, 40, 80)).to(DEV)
    with torch.no_grad():
        before = net(x).clone()          # (leaves cached planes behind: they must not survive the step)
    for step in range(3):
        v0 = net.conv21.weight._version
        net(x).square().mean().backward()
        opt.step()
        opt.zero_grad()
        fresh = VGG4L(128).to(DEV).train()
        fresh.load_state_dict(net.state_dict())
        want = fresh(x).detach()
        assert float((want - before).abs().max()) > 1e-3 * float(before.abs().max())      # the step moved the output
        with torch.no_grad():
            got_eval = net(x)
        got_train = net(x).detach()
        scale = float(want.abs().max())
        assert float((got_eval - want).abs().max()) <= 2e-5 * scale, (step, net.conv21.weight._version - v0)
        assert float((got_train - want).abs().max()) <= 2e-5 * scale, step


@pytest.mark.parametrize("cin,cout,H,W,B", [(64, 64, 9, 7, 3), (128, 64, 38, 10, 5), (64, 128, 33, 80, 2), (64, 64, 1, 1, 2),
                                           (128, 128, 19, 5, 7), (64, 64, 13, 8, 3), (128, 64, 10, 24, 2), (64, 64, 8, 40, 2)])
def test_hip_conv3x3_wgrad_vs_fp64(cin, cout, H, W, B):
    """Hand-written weight-gradient kernel (transposed LDS reads, zero-padded window image, split-K) against a float64
    reference; includes chunks that end mid-utterance, maps narrower than a chunk, 1x1 maps, and widths that are multiples
    of 8 (8x8-patch chunks, with partial patches at the bottom edge)."""
    from doubleattentionspeakerverification_amd import conv
    x = torch.from_numpy(rnd(51, B, cin, H, W)).to(DEV).contiguous(memory_format=torch.channels_last)
    g = torch.from_numpy(rnd(52, B, cout, H, W)).to(DEV).contiguous(memory_format=torch.channels_last)
    xd, gd = x.double().cpu(), g.double().cpu()
    w0 = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    (torch.nn.functional.conv2d(xd, w0, None, padding=1) * gd).sum().backward()
    ref = w0.grad
    gw = conv.conv3x3_wgrad(x, g, "split").cpu().double()
    gwb = conv.conv3x3_wgrad(x, g, "bf16").cpu().double()
    scale = float(ref.abs().max())
    assert float((gw - ref).abs().max()) / scale < 2e-5
    assert float((gwb - ref).abs().max()) / scale < 2e-2
    gw2 = conv.conv3x3_wgrad(x, g, "split").cpu().double()
    assert torch.equal(gw, gw2)   # fixed-order reduction: bitwise reproducible
    # the fp16 + e4m3 form (split8): fp32-grade at any operand scale (range records), records that only BOUND the maximum, and
    # a gradient with the heavy tail real ones have
    gw8 = conv.conv3x3_wgrad(x, g, "split8").cpu().double()
    assert float((gw8 - ref).abs().max()) / scale < 2e-5, float((gw8 - ref).abs().max()) / scale
    assert torch.equal(gw8, conv.conv3x3_wgrad(x, g, "split8").cpu().double())
    for fx, fg in ((3e4, 1e-7), (2e-6, 5e5)):
        gws = conv.conv3x3_wgrad(x * fx, g * fg, "split8").cpu().double() / (fx * fg)
        assert float((gws - ref).abs().max()) / scale < 2e-5, (fx, fg)
    loose = conv.conv3x3_wgrad(x, g, "split8", x_rec=conv.range_of(x * 7.0), g_rec=conv.range_of(g * 30.0)).cpu().double()
    assert float((loose - ref).abs().max()) / scale < 2e-5
    gt = g * (1.0 + 3999.0 * (torch.rand_like(g) < 1e-3))
    w1 = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    (torch.nn.functional.conv2d(xd, w1, None, padding=1) * gt.double().cpu()).sum().backward()
    gwt = conv.conv3x3_wgrad(x, gt, "split8").cpu().double()
    # (a sum that a few outliers dominate does not average the cross terms' e4m3 rounding — 2^-15 of a product — down: bar 2^-14)
    assert float((gwt - w1.grad).abs().max()) / float(w1.grad.abs().max()) < 6.2e-5


def test_fused_first_block_matches_unfused_and_torch():
    """conv11 fused into conv12's window staging (one launch) vs the two-launch HIP path vs PyTorch-ROCm."""
    from doubleattentionspeakerverification_amd.CNNs import VGG4L
    torch.manual_seed(5)
    net = VGG4L(1024).to(DEV).eval()
    x = torch.from_numpy(rnd(61, 3, 53, 80)).to(DEV)
    with torch.no_grad():
        net.conv_mode = "torch"
        ref = net(x)
        net.conv_mode = "split"
        assert net.fuse_first_block is False      # default: measured slower than two launches
        two = net(x)
        net.fuse_first_block = True
        one = net(x)
    assert float((one - two).abs().max() / two.abs().max()) < 1e-5
    assert float((one - ref).abs().max() / ref.abs().max()) < 5e-5


def test_hip_maxpool_and_conv11_backward_match_torch():
    from doubleattentionspeakerverification_amd import conv
    # max-pool: ReLU-like inputs full of ties (zeros) — the arg-max choice must be ATen's (first maximum wins)
    for shape in ((2, 64, 37, 80), (3, 32, 19, 5), (1, 16, 1, 1), (2, 128, 10, 3)):
        x = torch.relu(torch.from_numpy(rnd(71, *shape))).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        x2 = x.detach().clone().requires_grad_(True)
        y = conv.maxpool2x2_ceil_autograd(x)
        yr = torch.nn.functional.max_pool2d(x2, 2, stride=2, ceil_mode=True)
        gy = torch.from_numpy(rnd(72, *yr.shape)).to(DEV)
        (y * gy).sum().backward()
        (yr * gy).sum().backward()
        assert torch.equal(y, yr) and torch.equal(x.grad, x2.grad)
    # first layer
    x = torch.from_numpy(rnd(73, 3, 41, 80)).to(DEV)
    w = torch.from_numpy(rnd(74, 128, 1, 3, 3, scale=0.3)).to(DEV).requires_grad_(True)
    b = torch.from_numpy(rnd(75, 128, scale=0.1)).to(DEV).requires_grad_(True)
    gy = torch.from_numpy(rnd(76, 3, 128, 41, 80)).to(DEV)
    y = conv.conv3x3_c1_bias_relu_autograd(x, w, b)
    (y * gy).sum().backward()
    g1 = (w.grad.clone(), b.grad.clone())
    w.grad = b.grad = None
    yr = torch.relu(torch.nn.functional.conv2d(x.unsqueeze(1), w, b, padding=1))
    (yr * gy).sum().backward()
    np.testing.assert_allclose(g1[0].cpu().numpy(), w.grad.cpu().numpy(), rtol=2e-4, atol=2e-4 * float(w.grad.abs().max()))
    np.testing.assert_allclose(g1[1].cpu().numpy(), b.grad.cpu().numpy(), rtol=2e-4, atol=2e-4 * float(b.grad.abs().max()))
    # a gradient that already carries the layer's ReLU mask (what conv12's masked data-gradient kernel sends back): the launch
    # that skips reading y must give the same sums bit for bit (odd sizes: partial last trips of its four-pixel loop)
    for shp in ((3, 41, 80), (2, 7, 5), (1, 300, 80)):
        xx = torch.from_numpy(rnd(77, *shp)).to(DEV)
        gg = torch.from_numpy(rnd(78, shp[0], 128, shp[1], shp[2])).to(DEV)
        res = []
        for pre in (False, True):
            w.grad = b.grad = None
            yy = conv.conv3x3_c1_bias_relu_autograd(xx, w, b, None, pre)
            yy.backward(gg * (yy.detach() > 0) if pre else gg)
            res.append((w.grad.clone(), b.grad.clone()))
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), shp


@pytest.mark.parametrize("cmode", ["split8", "split"])
@pytest.mark.parametrize("ks", [1024, 512])
def test_frontend_training_backward_hip_equals_torch(ks, cmode, monkeypatch):
    """All gradients of the VGG4L front-end through the HIP path — convs, pools, conv11, with the ReLU masks fused into
    the neighbouring backward kernels — in the shipped default arithmetic (`split8`: data AND weight gradients on fp16 products +
    e4m3 cross terms, operands scaled by range records that ride down the backward chain) and in the 3 x bf16 `split` mode.
    kernel_size 512 runs its first block as the zero-padded network (conv12 has 64 output channels).  Three checks:
      (a) fused and unfused HIP backward agree to 2e-5 (same forward, so no arg-max / ReLU-sign flips; only summation orders
          differ);
      (b) every HIP conv layer's weight / bias / DATA gradient equals ATen's convolution_backward on the SAME saved tensors (the
          incoming gradient and the layer's input as the HIP path saw them: no flips possible) to 2e-5 (split) / 4e-5 (split8)
          of the gradient's scale;
      (c) against PyTorch-ROCm's own path the gradient vectors agree in direction (a single arg-max flip caused by the
          5e-6 forward difference moves one pixel's gradient — percent-level for one channel in a 2-utterance batch —
          so an element-wise bound against a different forward is not meaningful here)."""
    from doubleattentionspeakerverification_amd.CNNs import VGG4L
    from doubleattentionspeakerverification_amd import conv as C
    torch.manual_seed(5)
    net = VGG4L(ks).to(DEV).train()
    # (the per-layer wiring: one autograd node per convolution and per pool, fp32 tensors between them — what small batches, ragged
    # batches and narrow kernel sizes run; the one-launch conv + pool nodes and the packed chain: tests/test_split8_gpu.py)
    net.fuse_pool_training, net.packed_training = set(), False
    x = torch.from_numpy(rnd(61, 2, 45, 80)).to(DEV)
    orig_conv, orig_pool, orig_c1 = C.conv3x3_bias_relu, C.maxpool2x2_ceil_autograd, C.conv3x3_c1_bias_relu_autograd
    cap = {}

    def capturing(xx, w, b, packed=None, mode="split", premasked=False, mask_in=False, link_in=None, link_out=None, x_rec=None,
                  y_rec=None, span=None, borders=None, fuse=True):
        y = orig_conv(xx, w, b, packed, mode, premasked and fuse, mask_in and fuse, link_in if fuse else None,
                      link_out if fuse else None, x_rec, y_rec, span, borders)
        name = [k for k, v in net.named_parameters() if v is w][0].split(".")[0]
        cap[name] = dict(x=xx.detach(), y=y.detach(), premasked=premasked and fuse, mask_in=mask_in and fuse)
        y.register_hook(lambda g, name=name: cap[name].__setitem__("gy", g.detach().clone()))
        if xx.requires_grad:      # (this layer is the only consumer of xx: the hook sees this layer's data gradient)
            xx.register_hook(lambda g, name=name: cap[name].__setitem__("gx", g.detach().clone()))
        return y

    gout, grads = None, {}
    for mode in ("torch", "fused", "unfused"):
        fuse = mode == "fused"
        monkeypatch.setattr(C, "conv3x3_bias_relu", lambda *a, fuse=fuse, **k: capturing(*a, fuse=fuse, **k))
        monkeypatch.setattr(C, "maxpool2x2_ceil_autograd", lambda t, relu_input=False, link_out=None, x_rec=None, y_rec=None, pool_src=None, fuse=fuse:
                            orig_pool(t, relu_input and fuse, link_out if fuse else None, x_rec, y_rec, pool_src))
        # (conv11 leaves its ReLU mask to conv12's data-gradient kernel in the fused wiring only)
        monkeypatch.setattr(C, "conv3x3_c1_bias_relu_autograd", lambda t, w, b, y_rec=None, grad_premasked=False, span=None, fuse=fuse:
                            orig_c1(t, w, b, y_rec, grad_premasked and fuse, span))
        net.conv_mode = "torch" if mode == "torch" else cmode
        net.zero_grad()
        cap.clear()
        y = net(x)
        if gout is None:
            gout = torch.from_numpy(rnd(62, *y.shape)).to(DEV)
        (y * gout).sum().backward()
        grads[mode] = {k: v.grad.clone() for k, v in net.named_parameters()}
        if mode == "torch":
            continue
        assert len(cap) == 7      # every layer behind conv11 is captured: none leaves the hand-written kernels
        for name, d in cap.items():                                   # (b)
            conv_ = getattr(net, name)
            if d["premasked"]:
                assert torch.equal(d["gy"], d["gy"] * (d["y"] > 0)), name      # the consumer really applied the mask
            # (kernel_size 512: conv12's 64 output channels run as a zero-padded 128-channel layer — the real channels come first)
            g = (d["gy"] * (d["y"] > 0))[:, :conv_.out_channels].contiguous(memory_format=torch.channels_last)
            xin = d["x"][:, :conv_.in_channels].contiguous(memory_format=torch.channels_last)
            gx, gw, gb = torch.ops.aten.convolution_backward(g, xin, conv_.weight.detach(), [conv_.out_channels], [1, 1], [1, 1],
                                                             [1, 1], False, [0, 0], 1, [True, True, True])
            bar = 4e-5 if cmode == "split8" else 2e-5
            assert float((conv_.weight.grad - gw).abs().max()) <= bar * float(gw.abs().max()), name
            assert float((conv_.bias.grad - gb).abs().max()) <= 2e-5 * float(gb.abs().max()), name
            assert "gx" in d, name
            if d["mask_in"]:          # the data-gradient kernel applied the ReLU mask of the layer below in its epilogue
                gx = gx * (xin > 0)
            got = d["gx"][:, :conv_.in_channels]
            assert float((got - gx).abs().max()) <= bar * float(gx.abs().max()), (name, "data gradient")
    for k in grads["torch"]:
        # (a) same forward, so no arg-max / ReLU-sign flips: only summation orders differ (bias gradients from the
        # epilogues' per-workgroup partial sums; split-K data gradients on this small map in the unfused path)
        assert float((grads["fused"][k] - grads["unfused"][k]).abs().max()) <= 2e-5 * float(grads["unfused"][k].abs().max()), k
        cosv = float(torch.nn.functional.cosine_similarity(grads["fused"][k].flatten().double(),
                                                           grads["torch"][k].flatten().double(), dim=0))
        assert cosv > 0.9995, (k, cosv)                                                                       # (c)


def test_masked_conv_and_relu_pool_backward_entry_points():
    """dasv_conv3x3_masked_nhwc (data gradient with the lower layer's ReLU mask in the epilogue) and
    dasv_maxpool2x2_ceil_relu_bwd_nhwc (pool backward composed with the ReLU below it) against the two-pass torch forms."""
    from doubleattentionspeakerverification_amd import conv, _lib
    lib = _lib.load()
    g = torch.from_numpy(rnd(71, 2, 128, 11, 9)).to(DEV).contiguous(memory_format=torch.channels_last)
    w = torch.from_numpy(rnd(72, 128, 128, 3, 3, scale=0.05)).to(DEV)
    xin = torch.relu(torch.from_numpy(rnd(73, 2, 128, 11, 9))).to(DEV).contiguous(memory_format=torch.channels_last)
    wt = w.transpose(0, 1).flip(2, 3).contiguous()
    hi, lo = conv.PackedConvWeights().get(wt, "split")
    got = conv._launch(g, hi, lo, None, 128, "split", relu=0, mask=xin)
    ref = torch.nn.functional.conv2d(g.double().cpu(), wt.double().cpu(), None, padding=1) * (xin.cpu() > 0)
    assert float((got.cpu().double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    assert torch.equal(got == 0, (got == 0) | (xin <= 0)) and bool(((got != 0) <= (xin > 0)).all())
    # pool backward through a ReLU output full of zeros
    y = torch.relu(torch.from_numpy(rnd(74, 2, 64, 9, 7))).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    pre = y.detach().clone().requires_grad_(True)
    gy = torch.from_numpy(rnd(75, 2, 64, 5, 4)).to(DEV)
    torch.nn.functional.max_pool2d(torch.relu(pre), 2, stride=2, ceil_mode=True).backward(gy)
    gx = torch.empty_like(y, memory_format=torch.channels_last)
    gyc = gy.contiguous(memory_format=torch.channels_last)
    rows = lib.dasv_maxpool2x2_bwd_colsum_rows(2, 9, 7, 64)
    part = torch.empty(rows, 64, device=DEV)
    rc = lib.dasv_maxpool2x2_ceil_relu_bwd_nhwc(y.data_ptr(), gyc.data_ptr(), gx.data_ptr(), part.data_ptr(), 2, 9, 7, 64,
                                                torch.cuda.current_stream().cuda_stream)
    assert rc == 0 and rows > 0
    assert torch.equal(gx, pre.grad)
    np.testing.assert_allclose(part.sum(0).cpu().numpy(), pre.grad.sum(dim=(0, 2, 3)).cpu().numpy(), rtol=1e-5, atol=1e-5)
    # the same on an 80-column map (single-buffered 16-wave launch): masked values and their column sums
    g80 = torch.from_numpy(rnd(76, 1, 128, 7, 80)).to(DEV).contiguous(memory_format=torch.channels_last)
    x80 = torch.relu(torch.from_numpy(rnd(77, 1, 128, 7, 80))).to(DEV).contiguous(memory_format=torch.channels_last)
    l80 = {}
    got80 = conv._launch(g80, hi, lo, None, 128, "split", relu=0, mask=x80, colsum=l80)
    ref80 = torch.nn.functional.conv2d(g80.double().cpu(), wt.double().cpu(), None, padding=1) * (x80.cpu() > 0)
    assert float((got80.cpu().double() - ref80).abs().max()) <= 2e-5 * float(ref80.abs().max())
    np.testing.assert_allclose(l80["gb"].cpu().numpy(), got80.sum(dim=(0, 2, 3)).cpu().numpy(), rtol=1e-5,
                               atol=1e-5 * float(got80.sum(dim=(0, 2, 3)).abs().max()))
    # column sums of the masked data gradient (the bias gradient of the layer below)
    link = {}
    got2 = conv._launch(g, hi, lo, None, 128, "split", relu=0, mask=xin, colsum=link)
    assert torch.equal(got2, got)
    np.testing.assert_allclose(link["gb"].cpu().numpy(), got.sum(dim=(0, 2, 3)).cpu().numpy(), rtol=1e-5,
                               atol=1e-5 * float(got.sum(dim=(0, 2, 3)).abs().max()))


def test_vgg3l_default_width_and_long_utterance_on_the_hip_path():
    """VGG3L at kernel_size 1024 (256/256/512/512/1024/1024 channels, three pools) and an 800-frame utterance (BASELINE
    config 5's upper length) through the HIP front-end against PyTorch-ROCm; and the weight gradient on an 800 x 80 map."""
    from doubleattentionspeakerverification_amd.CNNs import VGG3L, VGG4L
    from doubleattentionspeakerverification_amd import conv
    torch.manual_seed(9)
    net = VGG3L(1024).to(DEV).eval()
    x = torch.from_numpy(rnd(81, 2, 61, 80)).to(DEV)
    with torch.no_grad():
        net.conv_mode = "torch"
        ref = net(x)
        net.conv_mode = "split"
        y = net(x)
    assert y.shape == ref.shape == (2, 8, 10240)
    assert float((y - ref).abs().max() / ref.abs().max()) < 5e-5
    net4 = VGG4L(1024).to(DEV).eval()
    xl = torch.from_numpy(rnd(82, 1, 800, 80)).to(DEV)
    with torch.no_grad():
        net4.conv_mode = "torch"
        ref = net4(xl)
        net4.conv_mode = "split"
        y = net4(xl)
    assert y.shape == ref.shape == (1, 50, 5120)
    assert float((y - ref).abs().max() / ref.abs().max()) < 5e-5
    a = torch.relu(torch.from_numpy(rnd(83, 1, 64, 800, 80))).to(DEV).contiguous(memory_format=torch.channels_last)
    g = torch.from_numpy(rnd(84, 1, 64, 800, 80)).to(DEV).contiguous(memory_format=torch.channels_last)
    w = torch.zeros(64, 64, 3, 3, device=DEV)
    gw_ref = torch.ops.aten.convolution_backward(g, a, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    gw = conv.conv3x3_wgrad(a, g, "split")
    assert float((gw - gw_ref).abs().max()) <= 5e-5 * float(gw_ref.abs().max())


def test_embedding_graph_replays_the_eager_result_and_cuts_single_utterance_latency():
    """hipGraph capture of getEmbedding at B=1 (the reference's one-utterance flows): same embedding as the eager path,
    for different inputs through the same graph; latency of both reported."""
    import time
    from doubleattentionspeakerverification_amd.model import SpeakerClassifier
    from doubleattentionspeakerverification_amd.graphs import EmbeddingGraph
    p = argparse.Namespace(front_end="VGG4L", kernel_size=1024, pooling_method="DoubleMHA", heads_number=32, mask_prob=0.3,
                           embedding_size=400, num_spkrs=100, scalingFactor=30.0, marginFactor=0.4, annealing=False)
    torch.manual_seed(3)
    net = SpeakerClassifier(p, torch.device(DEV)).to(DEV).eval()
    xs = [torch.from_numpy(rnd(90 + i, 1, 297, 80)).to(DEV) for i in range(3)]
    with torch.no_grad():
        ref = [net.getEmbedding(x).clone() for x in xs]
    g = EmbeddingGraph(net, 1, 297)
    for x, r in zip(xs, ref):
        out = g(x)
        assert out.shape == r.shape and float((out - r).abs().max()) <= 1e-6 * float(r.abs().max())
    with pytest.raises(ValueError):
        g(torch.zeros(2, 297, 80, device=DEV))

    def timeit(fn, n=50):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    with torch.no_grad():
        eager = timeit(lambda: net.getEmbedding(xs[0]))
    graph = timeit(lambda: g(xs[0]))
    print("single-utterance getEmbedding: eager %.3f ms, hipGraph replay %.3f ms" % (eager, graph))
    assert graph < eager * 1.05


def test_ragged_batch_equals_one_utterance_per_forward():
    """Utterances of different lengths stacked along time and run as ONE set of launches (ragged.py) against the
    reference's only exact way to handle them: one utterance per forward (train.py:117-133).  Odd lengths exercise the
    clipped ceil-mode pool windows at every level, and neighbouring utterances must not leak through the 3x3 taps."""
    from doubleattentionspeakerverification_amd.model import SpeakerClassifier
    from doubleattentionspeakerverification_amd.ragged import embed_ragged, front_end_ragged
    p = argparse.Namespace(front_end="VGG4L", kernel_size=1024, pooling_method="DoubleMHA", heads_number=32, mask_prob=0.3,
                           embedding_size=400, num_spkrs=50, scalingFactor=30.0, marginFactor=0.4, annealing=False)
    torch.manual_seed(21)
    net = SpeakerClassifier(p, torch.device(DEV)).to(DEV).eval()
    lens = [150, 297, 301, 163, 800, 151, 17, 1]
    utts = [torch.from_numpy(rnd(100 + i, n, 80)).to(DEV) for i, n in enumerate(lens)]
    with torch.no_grad():
        ref_frames = [net.front_end(u.unsqueeze(0))[0] for u in utts]
        ref = torch.cat([net.getEmbedding(u.unsqueeze(0)) for u in utts])
    frames, tl = front_end_ragged(net.front_end, utts)
    assert tl == [f.shape[0] for f in ref_frames] and frames.shape == (sum(tl), 5120)
    off = 0
    for f, n in zip(ref_frames, tl):
        got = frames[off:off + n]
        assert float((got - f).abs().max()) <= 2e-5 * float(f.abs().max()) + 1e-7
        off += n
    emb = embed_ragged(net, utts)
    assert emb.shape == ref.shape
    assert float((emb - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    # the other pooling heads take the same path
    for method in ("MHA", "Attention"):
        p2 = argparse.Namespace(**{**vars(p), "pooling_method": method, "heads_number": 8})
        torch.manual_seed(22)
        net2 = SpeakerClassifier(p2, torch.device(DEV)).to(DEV).eval()
        with torch.no_grad():
            r2 = torch.cat([net2.getEmbedding(u.unsqueeze(0)) for u in utts[:4]])
        e2 = embed_ragged(net2, utts[:4])
        assert float((e2 - r2).abs().max()) <= 2e-5 * float(r2.abs().max()), method


def test_packed_activation_chain_vgg3l_and_small_kernel_size():
    """The packed chain is generic over the number of blocks and the channel widths: VGG3L (three blocks, F' = 10) and a
    narrower VGG4L must also reproduce their fp32 chains (a packed tensor is scaled by a bound on its maximum, an fp32 one by
    its maximum: the operand planes agree except where values leave e4m3's normal range, far below the 1e-5 of the mode)."""
    from doubleattentionspeakerverification_amd.CNNs import VGG3L, VGG4L
    torch.manual_seed(11)
    mel = torch.randn(32, 200, 80, device=DEV) * 2.0
    for fe in (VGG3L(1024).to(DEV).eval(), VGG4L(512).to(DEV).eval()):
        with torch.no_grad():
            used = fe._packed_chain_ok(80, 32, 200)
            a = fe.forward_map(mel)
            fe.packed_activations = False
            b = fe.forward_map(mel)
            fe.packed_activations = True
        assert a.shape == b.shape
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()), used
    assert VGG3L(1024).to(DEV)._packed_chain_ok(80, 32, 200)      # the wide three-block model does take the packed chain here


def test_ragged_packed_activation_chain_matches_the_fp32_chain():
    """A ragged group big enough to fill the chip without split-K runs the split8 layers on packed activations (ragged.py,
    the `_x8` ragged entry points): the frame vectors must equal the fp32 ragged chain's within the mode's accuracy, and a
    few utterances are checked against their own one-utterance forward."""
    from doubleattentionspeakerverification_amd.CNNs import VGG4L
    from doubleattentionspeakerverification_amd import ragged
    torch.manual_seed(5)
    fe = VGG4L(1024).to(DEV).eval()
    g = np.random.RandomState(77)
    lens = [int(n) for n in g.randint(150, 801, size=40)] + [151, 17]
    utts = [torch.from_numpy(rnd(300 + i, n, 80)).to(DEV) for i, n in enumerate(lens)]
    rows = sum(lens)
    spans, pools, _ = ragged.ragged_tables(lens, fe.n_blocks, torch.device(DEV))
    assert fe.conv_mode == "split8" and fe.packed_activations and ragged._packed_chain_ok(fe, rows, 80, pools)
    a, ta = ragged.front_end_ragged(fe, utts)
    fe.packed_activations = False
    b, tb = ragged.front_end_ragged(fe, utts)
    fe.packed_activations = True
    assert ta == tb and float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())
    off = 0
    with torch.no_grad():
        for i, n in enumerate(ta):
            if i in (0, 7, 40, 41):
                f = fe(utts[i].unsqueeze(0))[0]
                assert float((a[off:off + n] - f).abs().max()) <= 2e-5 * float(f.abs().max()) + 1e-7, i
            off += n
    # a small group keeps the fp32 chain (its narrow maps want split-K)
    spans2, pools2, _ = ragged.ragged_tables(lens[:2], fe.n_blocks, torch.device(DEV))
    assert not ragged._packed_chain_ok(fe, sum(lens[:2]), 80, pools2)


def test_two_streams_on_one_device_share_no_scratch():
    """launches go to torch's current stream and no scratch is shared between calls.  The same module driven from
    two side streams at once (groups of utterances alternating between them, so convolutions, pools and the fused pooling
    kernel of different groups overlap in time) must give exactly what one stream gives."""
    from doubleattentionspeakerverification_amd.model import SpeakerClassifier
    from doubleattentionspeakerverification_amd import ragged
    p = argparse.Namespace(front_end="VGG4L", kernel_size=1024, pooling_method="DoubleMHA", heads_number=32, mask_prob=0.3,
                           embedding_size=400, num_spkrs=50, scalingFactor=30.0, marginFactor=0.4, annealing=False)
    torch.manual_seed(1234)
    dev = torch.device(DEV)
    net = SpeakerClassifier(p, dev).to(DEV).eval()
    g = np.random.RandomState(1)
    groups = [[torch.from_numpy(rnd(900 + 50 * k + i, int(n), 80)).to(DEV) for i, n in enumerate(g.randint(150, 801, size=24))]
              for k in range(4)]
    with torch.no_grad():
        ref = [ragged.embed_ragged(net, gr) for gr in groups]
        dense_ref = net.getEmbedding(torch.stack([gr[0][:150] for gr in groups]))
        torch.cuda.synchronize()
        cur = torch.cuda.current_stream(dev)
        side = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
        for st in side:
            st.wait_stream(cur)
        got, dense = [], []
        for k, gr in enumerate(groups):
            with torch.cuda.stream(side[k & 1]):
                got.append(ragged.embed_ragged(net, gr))
                dense.append(net.getEmbedding(torch.stack([q[0][:150] for q in groups])))
        for st in side:
            cur.wait_stream(st)
        torch.cuda.synchronize()
    for a, b in zip(ref, got):
        assert torch.equal(a, b)
    for d in dense:
        assert torch.equal(d, dense_ref)


def test_config3_full_size_properties():
    """BASELINE config 3 at its full size (default model, [64,300,80]), checked through properties that need no CPU run of
    39 GFLOP per utterance: (1) utterances are independent of their batch-mates — rows of the batched embedding equal the
    one-utterance embeddings (the reference's own scoring path is B=1, train.py:117-133); (2) duplicated utterances give
    identical rows; (3) the front-end commutes with a shift in time away from the borders (conv/pool stack with total
    stride 16: a 16-frame shift of the input moves the frame embeddings by one row)."""
    from doubleattentionspeakerverification_amd.model import SpeakerClassifier
    p = argparse.Namespace(front_end="VGG4L", kernel_size=1024, pooling_method="DoubleMHA", heads_number=32, mask_prob=0.3,
                           embedding_size=400, num_spkrs=5994, scalingFactor=30.0, marginFactor=0.4, annealing=False)
    torch.manual_seed(1234)
    net = SpeakerClassifier(p, torch.device(DEV)).to(DEV).eval()
    x = torch.from_numpy(rnd(123, 64, 300, 80)).to(DEV)
    x[17] = x[3]
    with torch.no_grad():
        emb = net.getEmbedding(x)
        assert emb.shape == (64, 400) and torch.isfinite(emb).all()
        assert torch.equal(emb[17], emb[3])                                            # (2)
        for i in (0, 3, 40, 63):                                                       # (1)
            one = net.getEmbedding(x[i:i + 1])
            assert float((one[0] - emb[i]).abs().max()) <= 2e-5 * float(emb.abs().max()), i
        frames = net.front_end(x[:4])                                                  # [4, 19, 5120]
        shifted = net.front_end(torch.roll(x[:4], shifts=16, dims=1))
        # rows 3..17 of the shifted map see exactly the samples rows 2..16 saw (receptive field of a row is ~+-46 input frames;
        # keep a margin from both ends where the wrap-around / zero padding differs)
        a, b = shifted[:, 5:15], frames[:, 4:14]
        assert float((a - b).abs().max()) <= 2e-5 * float(frames.abs().max())


def test_training_trajectory_hip_path_tracks_pytorch_path():
    """Three SGD steps of the default architecture on a fixed tiny batch: the losses of the all-HIP training path (conv
    forward / data gradient / weight gradient, pools, conv11, fused masks and bias sums, pooling forward/backward) must
    track the PyTorch-ROCm path — a wrong gradient anywhere shows up in the loss right after the first update.  (SGD, not
    Adam: Adam's first steps move every parameter by +-lr whatever the gradient's size, so noise-level gradients turn
    rounding differences into full-size update differences.  Only three steps: with a batch of four through train-mode
    BatchNorm and a x30 logit scale the dynamics amplify any 1e-6 difference ~50x per step in BOTH implementations.)"""
    import copy
    from doubleattentionspeakerverification_amd.model import SpeakerClassifier
    p = argparse.Namespace(front_end="VGG4L", kernel_size=1024, pooling_method="DoubleMHA", heads_number=32, mask_prob=0.3,
                           embedding_size=400, num_spkrs=16, scalingFactor=30.0, marginFactor=0.4, annealing=False)
    torch.manual_seed(31)
    net_a = SpeakerClassifier(p, torch.device(DEV)).to(DEV).train()
    net_b = copy.deepcopy(net_a)
    net_b.front_end.conv_mode = "torch"
    x = torch.from_numpy(rnd(131, 4, 61, 80)).to(DEV)
    y = torch.tensor([1, 5, 5, 12], device=DEV)
    losses = {}
    for tag, net in (("hip", net_a), ("torch", net_b)):
        opt = torch.optim.SGD(net.parameters(), lr=0.002)
        cur = []
        for step in range(3):
            torch.manual_seed(1000 + step)            # same train-time head mask in both runs
            opt.zero_grad(set_to_none=True)
            _, logits = net(x, label=y, step=step)
            loss = torch.nn.functional.cross_entropy(logits, y)
            loss.backward()
            opt.step()
            cur.append(float(loss.detach()))
        losses[tag] = cur
    print("loss curves:", [round(v, 4) for v in losses["hip"]], [round(v, 4) for v in losses["torch"]])
    assert abs(losses["hip"][1] - losses["hip"][0]) > 0.5              # the update really changed the loss
    for (a, b), tol in zip(zip(losses["hip"], losses["torch"]), (1e-5, 2e-4, 4e-3)):
        assert abs(a - b) <= tol * abs(b), (losses["hip"], losses["torch"])


def test_packed_conv_weights_follow_replicas_and_data_writes():
    """the packed bf16 planes were cached under (address, version).  A DataParallel replica (train.py:68-70)
    is a fresh plain tensor with version 0 at a recycled address, and a write through `.data` leaves the version alone —
    both must still reach the convolution."""
    from doubleattentionspeakerverification_amd import conv as hc
    torch.manual_seed(3)
    x = torch.randn(2, 64, 9, 10, device=DEV).contiguous(memory_format=torch.channels_last)
    bias = torch.zeros(128, device=DEV)
    w = torch.nn.Parameter(torch.randn(128, 64, 3, 3, device=DEV) * 0.05)
    pk = hc.PackedConvWeights()

    def ref(wt):
        return torch.relu(torch.nn.functional.conv2d(x.double(), wt.detach().double(), bias.double(), padding=1)).float()

    def close(y, wt):
        r = ref(wt)
        return float((y - r).abs().max()) <= 5e-5 * float(r.abs().max())

    with torch.no_grad():
        y0 = hc.conv3x3_bias_relu(x, w, bias, pk)
        assert close(y0, w)
        key0 = pk.key
        hc.conv3x3_bias_relu(x, w, bias, pk)
        assert pk.key == key0 and pk.owner() is w                    # the leaf Parameter itself: cached
        # (1) a replica: new plain tensor, version 0, same storage address, different contents
        replica = torch.empty(0)
        addr = None
        for _ in range(4):                                            # let the caching allocator hand the block back
            del replica
            replica = (torch.randn(128, 64, 3, 3, device=DEV) * 0.05)
            addr = replica.data_ptr()
            y1 = hc.conv3x3_bias_relu(x, replica, bias, pk)
            assert close(y1, replica) and pk.key is None              # never cached
        # simulated worst case: same address AND version as the cached Parameter
        pk2 = hc.PackedConvWeights()
        hc.conv3x3_bias_relu(x, w, bias, pk2)
        alias = w.detach()                                            # same address, shares the version counter
        w_old = w.detach().clone()
        alias.data.mul_(-1.0)                                         # contents change, w._version does not
        assert pk2.key[:3] == (w.data_ptr(), w._version, w.device)
        stale = hc.conv3x3_bias_relu(x, w, bias, pk2)                 # documented: `.data` writes need invalidate()
        assert close(stale, w_old)
        pk2.invalidate()
        assert close(hc.conv3x3_bias_relu(x, w, bias, pk2), w)
        y_alias = hc.conv3x3_bias_relu(x, alias, bias, pk2)           # a non-Parameter view is packed per call
        assert close(y_alias, w)
    # (2) module level: invalidate_packed_weights reaches the front-end's packed copies
    from doubleattentionspeakerverification_amd.CNNs import VGG4L
    fe = VGG4L(1024).to(DEV).eval()
    mel = torch.randn(2, 40, 80, device=DEV)
    with torch.no_grad():
        a = fe(mel)
        fe.conv42.weight.data.mul_(0.5)
        hc.invalidate_packed_weights(fe)
        b = fe(mel)
        fe.conv42.weight.mul_(2.0)                                    # ordinary in-place update: version moves
        c = fe(mel)
    assert float((b - a).abs().max()) > 1e-3 * float(a.abs().max())
    assert float((c - b).abs().max()) > 1e-3 * float(b.abs().max())


def test_data_parallel_replicas_see_current_weights():
    """nn.DataParallel on one device (device_ids=[0, 0] is rejected by torch, so replicate by hand like DataParallel.forward
    does): the replica's forward must use the weights of THIS step."""
    from torch.nn.parallel import replicate
    from doubleattentionspeakerverification_amd.CNNs import VGG4L
    fe = VGG4L(1024).to(DEV).eval()
    mel = torch.randn(2, 33, 80, device=DEV)
    with torch.no_grad():
        outs = []
        for it in range(3):
            rep = replicate(fe, [0], detach=True)[0]
            outs.append((rep(mel), fe(mel)))
            for p in fe.parameters():
                p.mul_(1.05)
            del rep
        for r, m in outs:
            assert float((r - m).abs().max()) <= 1e-5 * float(m.abs().max())
    assert float((outs[0][1] - outs[1][1]).abs().max()) > 0


@pytest.mark.parametrize("B,T", [(24, 300), (64, 150), (2, 61)])
def test_packed_activation_chain_matches_the_fp32_chain_and_the_oracle(B, T):
    """Inference in the split8 mode writes the activations between the layers once, in the consumer's operand form (fp16 hi |
    e4m3 | e4m3 residual per 64-channel slab, scaled by the exponent in the tensor's range record): the front-end's output must
    equal the fp32 chain's within the mode's accuracy, and both the CPU oracle's (CNNs.py:68-91).  Also the single pieces."""
    from doubleattentionspeakerverification_amd.CNNs import VGG4L
    from doubleattentionspeakerverification_amd import conv as hc
    torch.manual_seed(7)
    fe = VGG4L(1024).to(DEV).eval()
    mel = torch.randn(B, T, 80, device=DEV) * 2.0
    with torch.no_grad():
        # active for batches whose smallest maps fill the chip without split-K (a packed output has no split-K form); the
        # single-utterance flow keeps the fp32 chain
        assert fe.packed_activations and fe._packed_chain_ok(80, B, T) == (B >= 16)
        a = fe.forward_map(mel)
        fe.packed_activations = False
        b = fe.forward_map(mel)
        fe.packed_activations = True
        assert a.shape == b.shape and float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())
        nb = min(B, 4)
        sd = {"front_end." + k: v.detach().cpu() for k, v in fe.state_dict().items()}
        torch.set_num_threads(8)
        ref = oracle.vgg_forward(mel[:nb].cpu(), sd, 4)                       # [nb, T', C*F']
        got = a[:nb].permute(0, 2, 1, 3).reshape(nb, a.size(2), -1).cpu()
        assert float((got - ref).abs().max()) <= 5e-5 * float(ref.abs().max())
        # pieces: packed conv11 -> conv12 vs fp32 conv11 -> conv12; pool(pack) -> conv21(packed in / packed out) -> conv22
        recs = hc.new_ranges(mel.device, 8)
        h11p = hc.conv3x3_c1_bias_relu(mel, fe.conv11.weight, fe.conv11.bias, pack=True, y_rec=recs[0])
        h11 = hc.conv3x3_c1_bias_relu(mel, fe.conv11.weight, fe.conv11.bias, y_rec=recs[1])
        assert int(recs[0, 0]) == int(recs[1, 0]) == int(h11.abs().max().view(torch.int32))      # both record max |y| exactly
        assert float(h11.abs().max()) * 2.0 ** int(recs[0, 1]) < 256.0                           # the bound-derived exponent is safe
        y_p = hc.conv3x3_x8(h11p, fe.conv12.weight, fe.conv12.bias, None, x_packed=True, x_rec=recs[0], y_rec=recs[2])
        y_f = hc.conv3x3_bias_relu(h11, fe.conv12.weight, fe.conv12.bias, None, "split8", x_rec=recs[1], y_rec=recs[3])
        assert float((y_p - y_f).abs().max()) <= 2e-5 * float(y_f.abs().max())
        assert int(recs[3, 0]) == int(y_f.abs().max().view(torch.int32))
        pp, pf = hc.maxpool2x2_ceil(y_f, pack=True, x_rec=recs[3], y_rec=recs[4]), hc.maxpool2x2_ceil(y_f)
        z_p = hc.conv3x3_x8(hc.conv3x3_x8(pp, fe.conv21.weight, fe.conv21.bias, None, x_packed=True, y_packed=True, x_rec=recs[4], y_rec=recs[5]),
                            fe.conv22.weight, fe.conv22.bias, None, x_packed=True, x_rec=recs[5])
        z_f = hc.conv3x3_bias_relu(hc.conv3x3_bias_relu(pf, fe.conv21.weight, fe.conv21.bias, None, "split8"),
                                   fe.conv22.weight, fe.conv22.bias, None, "split8")
        assert float((z_p - z_f).abs().max()) < 2e-5 * float(z_f.abs().max())


@pytest.mark.parametrize("ann", [False, True])
def test_amsoftmax_hip_head_matches_golden_g5(ann):
    """Fixture G5 (the reference's AMSoftmax + CrossEntropyLoss, loss.py:37-52 / train.py:219, annealing on and off, steps 0 ... 1e6)
    against the hand-written head (csrc/amsoftmax.hip): the module form (costh, logits -> torch CE) and the fused form
    (fused_loss: no logits, no one-hot, no log-softmax) — values and both gradients."""
    from doubleattentionspeakerverification_amd.loss import AMSoftmax
    g = load_golden("g5_amsoftmax")
    tag = "ann" if ann else "plain"
    head = AMSoftmax(24, 37, m=0.4, s=30.0, annealing=ann).to(DEV)
    with torch.no_grad():
        head.W.copy_(torch.from_numpy(g[tag + "_W"]))
    lab = torch.from_numpy(g["label"]).to(DEV)
    for step in (0, 100, 5000, 1000000):
        k = "%s_s%d_" % (tag, step)
        for fused in (False, True):
            x = torch.from_numpy(g["x"]).to(DEV).requires_grad_(True)
            head.W.grad = None
            if fused:
                loss, costh = head.fused_loss(x, lab, step)
            else:
                costh, logits = head(x, lab, step)
                np.testing.assert_allclose(logits.detach().cpu().numpy(), g[k + "logits"], rtol=1e-5, atol=2e-5)
                loss = torch.nn.functional.cross_entropy(logits, lab)
            loss.backward()
            np.testing.assert_allclose(costh.detach().cpu().numpy(), g[k + "costh"], rtol=0, atol=2e-6)
            np.testing.assert_allclose(float(loss.detach()), float(g[k + "loss"]), rtol=2e-6)
            np.testing.assert_allclose(x.grad.cpu().numpy(), g[k + "grad_x"], rtol=1e-4, atol=1e-6 * np.abs(g[k + "grad_x"]).max() + 1e-7)
            np.testing.assert_allclose(head.W.grad.cpu().numpy(), g[k + "grad_W"], rtol=1e-4, atol=1e-6 * np.abs(g[k + "grad_W"]).max() + 1e-7)
        assert abs(head.getAnnealedFactor(step) - float(g[k + "factor"])) < 1e-12


def test_amsoftmax_hip_head_default_size_vs_float64():
    """The default head (emb 400 -> 5994 speakers, B = 64: tiles with ragged edges in every dimension) against the reference's op
    sequence in float64 on the CPU: loss, costh, gradients; the fused and the module form agree; a label at the last column."""
    from doubleattentionspeakerverification_amd.loss import AMSoftmax
    torch.manual_seed(3)
    B, E, N = 64, 400, 5994
    head = AMSoftmax(E, N, m=0.4, s=30.0).to(DEV)
    x0 = torch.randn(B, E) * 3.0
    lab = torch.randint(0, N, (B,))
    lab[0], lab[1] = N - 1, 0
    Wd = head.W.detach().cpu().double()
    xd = x0.double().requires_grad_(True)
    Wd.requires_grad_(True)
    rows = xd / xd.norm(dim=1, keepdim=True).clamp(min=1e-12)
    cols = Wd / Wd.norm(dim=0, keepdim=True).clamp(min=1e-12)
    costh_ref = rows @ cols
    logits_ref = 30.0 * (costh_ref - 0.4 * torch.nn.functional.one_hot(lab, N).double())
    loss_ref = torch.nn.functional.cross_entropy(logits_ref, lab)
    loss_ref.backward()
    res = {}
    for fused in (True, False):
        x = x0.to(DEV).requires_grad_(True)
        head.W.grad = None
        if fused:
            loss, costh = head.fused_loss(x, lab.to(DEV), 0)
        else:
            costh, logits = head(x, lab.to(DEV), 0)
            loss = torch.nn.functional.cross_entropy(logits, lab.to(DEV))
        loss.backward()
        res[fused] = (float(loss.detach()), costh.detach().cpu(), x.grad.cpu(), head.W.grad.cpu())
        assert abs(res[fused][0] - float(loss_ref)) < 2e-6 * abs(float(loss_ref))
        assert float((res[fused][1].double() - costh_ref.detach()).abs().max()) < 2e-6
        assert float((res[fused][2].double() - xd.grad).abs().max()) <= 2e-5 * float(xd.grad.abs().max())
        assert float((res[fused][3].double() - Wd.grad).abs().max()) <= 2e-5 * float(Wd.grad.abs().max())
    assert float((res[True][2] - res[False][2]).abs().max()) <= 1e-5 * float(res[False][2].abs().max())
    # a label outside [0, N): the reference raises (scatter_, loss.py:47 / CrossEntropyLoss, train.py:219); a launch cannot, so the
    # fused loss must be NaN — never a finite number made of whatever the workspace held — launch after launch
    for bad in (N, -1, 1 << 40):
        lb = lab.clone()
        lb[5] = bad
        for _ in range(2):
            loss, _ = head.fused_loss(x0.to(DEV), lb.to(DEV), 0)
            assert bool(torch.isnan(loss)), bad
    loss, _ = head.fused_loss(x0.to(DEV), lab.to(DEV), 0)
    assert abs(float(loss) - float(loss_ref)) < 2e-6 * abs(float(loss_ref))


@pytest.mark.parametrize("lens,levels", [([1], 4), ([17, 1, 800, 2, 3], 4), ([5] * 300 + [799, 1, 2], 3), ([301, 150, 299], 1)])
def test_ragged_tables_kernel_equals_the_host_walk(lens, levels):
    """dasv_ragged_tables (one launch per group of utterances) against the host arithmetic that specifies it (ragged.ragged_tables on
    CPU tensors, itself checked against a naive walk in tests/test_host_cpu.py): spans, pools and the frame index, bit for bit."""
    from doubleattentionspeakerverification_amd.ragged import ragged_tables
    sd, pd, ld, fd = ragged_tables(lens, levels, torch.device(DEV), with_index=True)
    sh, ph, lh, fh = ragged_tables(lens, levels, "cpu", with_index=True)
    assert ld == lh
    for a, b in zip(sd + pd + [fd], sh + ph + [fh]):
        assert a.dtype == b.dtype and torch.equal(a.cpu(), b)
