"""Pin the CPU oracle (oracle/) to golden vectors captured from the real reference.

Runs without a GPU.  Each test names the fixture group of SURVEY.md section 8c it covers.
"""
import numpy as np
import pytest
import torch

import oracle
from conftest import load_golden, rnd, sd_from_golden


def _x(g, i):
    key = "c%d_x" % i
    if key in g.files:
        return g[key]
    B, T, D, H = g["c%d_shape" % i]
    return rnd(int(g["c%d_seed" % i]), int(B), int(T), int(D))


# ------------------------------------------------------------------ G1
@pytest.mark.parametrize("i", range(6))
def test_g1_dmha_eval_opseq_and_closed(i):
    g = load_golden("g1_pooling_eval")
    x, q, a = _x(g, i), g["c%d_query" % i], g["c%d_att" % i]
    out, al, hw = oracle.dmha_opseq(torch.from_numpy(x), torch.from_numpy(q), torch.from_numpy(a))
    np.testing.assert_allclose(out.numpy(), g["c%d_out" % i], rtol=0, atol=1e-6)
    np.testing.assert_allclose(al.numpy(), g["c%d_align" % i], rtol=0, atol=1e-7)
    np.testing.assert_allclose(hw.numpy(), g["c%d_head_w" % i], rtol=0, atol=1e-7)
    c = oracle.dmha_closed(x, q, a)
    np.testing.assert_allclose(c["out"], g["c%d_out" % i], rtol=0, atol=2e-6)
    np.testing.assert_allclose(c["align"], g["c%d_align" % i], rtol=0, atol=1e-6)
    np.testing.assert_allclose(c["head_w"][..., None], g["c%d_head_w" % i], rtol=0, atol=1e-6)
    B, T, D, H = g["c%d_shape" % i]
    np.testing.assert_allclose(c["ctx"].reshape(B, D), g["c%d_ctx" % i], rtol=0, atol=2e-6)


def test_g1_scale_is_sqrt_heads_not_sqrt_head_dim():
    """poolings.py:75: d_k = query.size(-1) = H.  A sqrt(dh) oracle must NOT match."""
    g = load_golden("g1_pooling_eval")
    x, q = _x(g, 1), g["c1_query"]
    B, T, D, H = g["c1_shape"]
    dh = D // H
    xv = x.reshape(B, T, H, dh).astype(np.float64)
    s_wrong = np.einsum("bthd,dh->bth", xv, q.astype(np.float64)) / np.sqrt(dh)
    e = np.exp(s_wrong - s_wrong.max(1, keepdims=True))
    assert np.abs(e / e.sum(1, keepdims=True) - g["c1_align"]).max() > 1e-3


@pytest.mark.parametrize("tag", ["big", "relu"])
def test_g1_stress_inputs(tag):
    g = load_golden("g1_pooling_eval")
    c = oracle.dmha_closed(g["s_%s_x" % tag], g["s_query"], g["s_att"])
    np.testing.assert_allclose(c["out"], g["s_%s_out" % tag], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(c["align"], g["s_%s_align" % tag], rtol=0, atol=2e-6)


def test_g1_single_head_attention():
    g = load_golden("g1_pooling_eval")
    ct, sc = oracle.attention_pool_opseq(torch.from_numpy(g["a_x"]), torch.from_numpy(g["a_att"]))
    np.testing.assert_allclose(ct.numpy(), g["a_ct"], atol=1e-6)
    np.testing.assert_allclose(sc.numpy(), g["a_score"], atol=1e-7)
    ct2, sc2 = oracle.attention_pool_closed(g["a_x"], g["a_att"])
    np.testing.assert_allclose(ct2, g["a_ct"], atol=2e-6)
    np.testing.assert_allclose(sc2, g["a_score"], atol=1e-6)


# ------------------------------------------------------------------ G2
@pytest.mark.parametrize("i", range(3))
def test_g2_train_mode_masked_forward_backward(i):
    g = load_golden("g2_pooling_train")
    x, q, a, keep, go = (g["c%d_%s" % (i, k)] for k in ("x", "query", "att", "keep", "grad_out"))
    f = oracle.dmha_closed(x, q, a, head_keep=keep)
    np.testing.assert_allclose(f["out"], g["c%d_out" % i], atol=2e-6)
    np.testing.assert_allclose(f["align"], g["c%d_align" % i], atol=1e-6)
    b = oracle.dmha_backward_closed(x, q, a, go, head_keep=keep)
    np.testing.assert_allclose(b["grad_x"], g["c%d_grad_x" % i], atol=2e-6)
    np.testing.assert_allclose(b["grad_query"], g["c%d_grad_query" % i], atol=5e-6)
    np.testing.assert_allclose(b["grad_att"], g["c%d_grad_att" % i], atol=5e-6)
    # op-sequence version + torch autograd agrees too
    xt = torch.from_numpy(x).requires_grad_(True)
    qt = torch.from_numpy(q).requires_grad_(True)
    at = torch.from_numpy(a).requires_grad_(True)
    o, _, _ = oracle.dmha_opseq(xt, qt, at, torch.from_numpy(keep))
    (o * torch.from_numpy(go)).sum().backward()
    np.testing.assert_allclose(xt.grad.numpy(), g["c%d_grad_x" % i], atol=1e-6)
    np.testing.assert_allclose(qt.grad.numpy(), g["c%d_grad_query" % i], atol=2e-6)
    np.testing.assert_allclose(at.grad.numpy(), g["c%d_grad_att" % i], atol=2e-6)


def test_g2_mask_draw_reproduces_reference_rng():
    g = load_golden("g2_pooling_train")
    for i in range(int(g["n"])):
        B, T, D, H = g["c%d_shape" % i]
        keep = oracle.head_keep_from_torch_rng(int(B), int(H), float(g["c%d_mask_prob" % i]), int(g["c%d_mask_seed" % i]))
        assert np.array_equal(keep.numpy(), g["c%d_keep" % i])


def test_g2_eval_backward_and_all_masked_nan():
    g = load_golden("g2_pooling_train")
    b = oracle.dmha_backward_closed(g["e_x"], g["e_query"], g["e_att"], g["e_grad_out"])
    np.testing.assert_allclose(b["grad_x"], g["e_grad_x"], atol=2e-6)
    np.testing.assert_allclose(b["grad_query"], g["e_grad_query"], atol=5e-6)
    np.testing.assert_allclose(b["grad_att"], g["e_grad_att"], atol=5e-6)
    assert bool(g["allmasked_isnan"])
    f = oracle.dmha_closed(rnd(1, 2, 5, 16), rnd(2, 4, 4), rnd(3, 4, 1), head_keep=np.zeros((2, 4), bool))
    assert np.isnan(f["out"]).all()


# ------------------------------------------------------------------ G3
@pytest.mark.parametrize("L", [3, 4])
@pytest.mark.parametrize("ks", [16, 32])
def test_g3_vgg_forward(L, ks):
    g = load_golden("g3_vgg")
    sd = sd_from_golden(g, "L%d_k%d_" % (L, ks))
    sd = {"front_end." + k: v for k, v in sd.items() if k.startswith("conv")}
    for T in (8, 31, 50):
        y = oracle.vgg_forward(torch.from_numpy(rnd(700 + T, 2, T, 80)), sd, L)
        ref = g["L%d_k%d_T%d_y" % (L, ks, T)]
        assert y.shape == ref.shape
        np.testing.assert_allclose(y.numpy(), ref, atol=1e-5, rtol=1e-5)


def test_g3_shape_table_and_dims():
    g = load_golden("g3_vgg")
    for T, t3, t4 in zip(g["shape_T"], g["shape_T3"], g["shape_T4"]):
        assert oracle.vgg_out_frames(int(T), 3) == int(t3)
        assert oracle.vgg_out_frames(int(T), 4) == int(t4)
    assert oracle.vgg_output_dim(80, 1024, 3) == int(g["dim3_80_1024"]) == 10240
    assert oracle.vgg_output_dim(80, 1024, 4) == int(g["dim4_80_1024"]) == 5120
    assert oracle.vgg_output_dim(80, 512, 4) == int(g["dim4_80_512"])
    assert oracle.vgg_output_dim(40, 128, 3) == int(g["dim3_40_128"])


# ------------------------------------------------------------------ G4
@pytest.mark.parametrize("tag,L,H", [("v4", 4, 4), ("v3", 3, 8)])
def test_g4_embedding_and_classifier(tag, L, H):
    g = load_golden("g4_model")
    sd = sd_from_golden(g, tag + "_sd_")
    x = torch.from_numpy(g["x"])
    emb = oracle.embedding_forward(x, sd, L, H)
    np.testing.assert_allclose(emb.numpy(), g[tag + "_emb"], atol=2e-5, rtol=1e-5)
    costh, logits = oracle.classifier_forward(x, sd, L, H, torch.from_numpy(g["label"]), 0.4, 30.0)
    np.testing.assert_allclose(costh.numpy(), g[tag + "_costh"], atol=1e-5)
    np.testing.assert_allclose(logits.numpy(), g[tag + "_logits"], atol=3e-4)


def test_g4_cosine_score_of_two_utterances():
    g = load_golden("g4_model")
    sd = sd_from_golden(g, "score_sd_")
    e1 = oracle.embedding_forward(torch.from_numpy(rnd(830, 1, 123, 80)), sd, 4, 4)
    e2 = oracle.embedding_forward(torch.from_numpy(rnd(831, 1, 77, 80)), sd, 4, 4)
    np.testing.assert_allclose(e1.numpy(), g["score_e1"], atol=2e-5)
    assert abs(float(oracle.cosine_score(e1, e2)) - float(g["score_cos"].reshape(-1)[0])) < 1e-5


# ------------------------------------------------------------------ G5
@pytest.mark.parametrize("tag,ann", [("plain", False), ("ann", True)])
def test_g5_amsoftmax(tag, ann):
    g = load_golden("g5_amsoftmax")
    W = torch.from_numpy(g[tag + "_W"]).requires_grad_(True)
    lab = torch.from_numpy(g["label"])
    for step in (0, 100, 5000, 1000000):
        x = torch.from_numpy(g["x"]).requires_grad_(True)
        W.grad = None
        costh, logits = oracle.amsoftmax_forward(x, W, lab, 0.4, 30.0, ann, step)
        np.testing.assert_allclose(costh.detach().numpy(), g["%s_s%d_costh" % (tag, step)], atol=1e-6)
        np.testing.assert_allclose(logits.detach().numpy(), g["%s_s%d_logits" % (tag, step)], atol=2e-5)
        torch.nn.functional.cross_entropy(logits, lab).backward()
        np.testing.assert_allclose(x.grad.numpy(), g["%s_s%d_grad_x" % (tag, step)], atol=1e-6)
        np.testing.assert_allclose(W.grad.numpy(), g["%s_s%d_grad_W" % (tag, step)], atol=1e-6)
        assert abs(oracle.annealed_factor(step, ann) - float(g["%s_s%d_factor" % (tag, step)])) < 1e-12


# ------------------------------------------------------------------ G6
def test_g6_scoring_helpers():
    g = load_golden("g6_scoring")
    assert oracle.eer_from_scores(g["eer_clients"], g["eer_impostors"]) == float(g["eer"])
    assert oracle.eer_from_scores(np.full(10, 0.9), np.full(10, -0.9)) == float(g["eer_separated"])
    cl, im = g["eer_clients"].astype(np.float64), g["eer_impostors"].astype(np.float64)
    rs = np.random.RandomState(0)
    cl, im = rs.normal(0.6, 0.15, 2000), rs.normal(0.1, 0.15, 2000)
    for t, far, frr in zip(g["score_th"], g["score_far"], g["score_frr"]):
        assert oracle.far_frr(im, t, "FAR") == far
        assert oracle.far_frr(cl, t, "FRR") == frr
    cos = oracle.cosine_score(torch.from_numpy(g["cos_e1"]), torch.from_numpy(g["cos_e2"]))
    np.testing.assert_allclose(cos.numpy(), g["cos"], atol=1e-7)


# ------------------------------------------------------------------ plain-C restatement
@pytest.fixture(scope="module")
def c_oracle():
    import os
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")])
    return oracle.dmha_c


@pytest.mark.parametrize("i", range(6))
def test_c_restatement_matches_reference_vectors(c_oracle, i):
    g = load_golden("g1_pooling_eval")
    f = c_oracle(_x(g, i), g["c%d_query" % i], g["c%d_att" % i])
    np.testing.assert_allclose(f["out"], g["c%d_out" % i], rtol=0, atol=2e-6)
    np.testing.assert_allclose(f["align"], g["c%d_align" % i], rtol=0, atol=1e-6)
    np.testing.assert_allclose(f["head_w"][..., None], g["c%d_head_w" % i], rtol=0, atol=1e-6)


def test_c_restatement_masked_heads(c_oracle):
    g = load_golden("g2_pooling_train")
    for i in range(int(g["n"])):
        f = c_oracle(g["c%d_x" % i], g["c%d_query" % i], g["c%d_att" % i], g["c%d_keep" % i])
        np.testing.assert_allclose(f["out"], g["c%d_out" % i], atol=2e-6)
        np.testing.assert_allclose(f["align"], g["c%d_align" % i], atol=1e-6)


# ------------------------------------------------------------------ G7 (the training-step restatement, oracle/train_oracle.py)
def g7_batches(g, tag, num_spkrs=11):
    """The two micro-batches of fixture G7 (tests/golden/make_golden.py g7): inputs, labels, the explicit head masks."""
    out = []
    for mb in range(2):
        x = torch.from_numpy(rnd(1300 + mb, 4, 40, 80))
        lab = torch.tensor([(3 * mb + j) % num_spkrs for j in range(4)])
        out.append((x, lab, torch.from_numpy(g[tag + "_keep"][mb])))
    return out


@pytest.mark.parametrize("tag", ["adam", "sgd"])
def test_g7_train_step_gradients(tag):
    """train.py:215-226 restated in float64 under its own ReLU / max-pool decisions: the reference's two losses and every
    gradient tensor of fixture G7 (the reference ran fp32: 1-D gradients behind train-mode BatchNorm are rounding residue of
    cancelling terms there — absolute bound, as in tests/test_model_gpu.py)."""
    g = load_golden("g7_train_step")
    sd = sd_from_golden(g, tag + "_init_")
    losses, grads, used = oracle.train_step_grads(sd, g7_batches(g, tag), 4, 0.4, 30.0)
    np.testing.assert_allclose(losses, g[tag + "_losses"], rtol=2e-6)
    assert sorted(set(k for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k) - set(grads)) == \
        sorted(str(n) for n in g[tag + "_nograd"])
    for n, gr in grads.items():
        ref = g[tag + "_grad_" + n.replace(".", "__")]
        atol = 2e-3 if ref.ndim == 1 else 2e-5 * max(1.0, np.abs(ref).max())
        np.testing.assert_allclose(gr.numpy(), ref, rtol=2e-3, atol=atol, err_msg=n)
    # a forward that takes the SAME decisions has no flips; one perturbed pre-activation sign is counted
    rec = {}
    p64 = {k: v.double() for k, v in sd.items()}
    oracle.vgg_forward_decided(g7_batches(g, tag)[0][0].double(), p64, 4, None, rec)
    assert sum(oracle.count_flips(rec, rec, 4).values()) == 0
    other = {k: v.clone() for k, v in rec.items()}
    other["conv21"][0, 0, 0, 0] ^= True
    assert oracle.count_flips(rec, other, 4)["conv21"] == 1
