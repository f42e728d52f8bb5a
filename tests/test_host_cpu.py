"""CPU-only checks of the host-side mirror of the reference interface: names, constructor signatures,
state_dict contract, star-import shims, and that the compute path refuses CPU tensors (no fallback)."""
import argparse
import inspect
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _c3_params():
    return argparse.Namespace(front_end="VGG4L", kernel_size=1024, pooling_method="DoubleMHA", heads_number=32,
                              mask_prob=0.3, embedding_size=400, num_spkrs=5994, scalingFactor=30.0,
                              marginFactor=0.4, annealing=False)


def test_state_dict_contract_matches_reference_default_config():
    from doubleattentionspeakerverification_amd.model import SpeakerClassifier
    g = load_golden("g4_model")
    p = _c3_params()
    net = SpeakerClassifier(p, torch.device("cpu"))
    sd = net.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["c3_keys"]]
    assert [",".join(map(str, v.shape)) for v in sd.values()] == [str(s) for s in g["c3_shapes"]]
    assert sum(q.numel() for q in net.parameters()) == int(g["c3_nparams"]) == 21522384
    assert [n for n, _ in net.named_parameters()] == [str(n) for n in g["c3_param_names"]]
    assert p.feature_size == 80  # model.py:13 side effect


@pytest.mark.parametrize("tag,fe,H,pool", [("v4", "VGG4L", 4, "DoubleMHA"), ("v3", "VGG3L", 8, "DoubleMHA"),
                                           ("mha", "VGG4L", 4, "MHA"), ("att", "VGG4L", 4, "Attention")])
def test_reference_state_dicts_load_strict(tag, fe, H, pool):
    from conftest import sd_from_golden
    from doubleattentionspeakerverification_amd.model import SpeakerClassifier
    g = load_golden("g4_model")
    p = argparse.Namespace(front_end=fe, kernel_size=16, pooling_method=pool, heads_number=H, mask_prob=0.3,
                           embedding_size=24, num_spkrs=11, scalingFactor=30.0, marginFactor=0.4, annealing=False)
    net = SpeakerClassifier(p, torch.device("cpu"))
    net.load_state_dict(sd_from_golden(g, tag + "_sd_"), strict=True)


def test_constructor_and_forward_signatures():
    from doubleattentionspeakerverification_amd import poolings, CNNs, loss, model
    assert list(inspect.signature(poolings.DoubleMHA.__init__).parameters)[1:] == ["encoder_size", "heads_number", "mask_prob"]
    assert inspect.signature(poolings.DoubleMHA.__init__).parameters["mask_prob"].default == 0.2
    assert list(inspect.signature(poolings.MultiHeadAttention.__init__).parameters)[1:] == ["encoder_size", "heads_number"]
    assert list(inspect.signature(poolings.HeadAttention.__init__).parameters)[1:] == ["encoder_size", "heads_number", "mask_prob", "attentionSmoothing"]
    assert list(inspect.signature(poolings.DoubleMHA.forward).parameters)[:2] == ["self", "x"]
    assert list(inspect.signature(loss.AMSoftmax.__init__).parameters)[1:] == ["in_feats", "n_classes", "m", "s", "annealing"]
    sig = inspect.signature(model.SpeakerClassifier.forward).parameters
    assert list(sig) == ["self", "x", "label", "step"] and sig["step"].default == 0
    assert CNNs.getVGG4LOutputDimension(80, outputChannel=1024) == 5120
    assert CNNs.getVGG3LOutputDimension(80, outputChannel=1024) == 10240
    with pytest.raises(AssertionError):
        poolings.MultiHeadAttention(10, 3)  # poolings.py:87
    m = poolings.DoubleMHA(64, 4, mask_prob=0.3)
    assert m.utteranceAttention.query.shape == (16, 4) and m.headsAttention.att.shape == (16, 1)
    assert m.headsAttention.mask_prob == 3  # k = int(1/mask_prob)


def test_compute_path_has_no_cpu_fallback():
    from doubleattentionspeakerverification_amd import poolings
    m = poolings.DoubleMHA(64, 4).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(2, 5, 64))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        poolings.Attention(8)(torch.zeros(2, 5, 8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        poolings.MultiHeadAttention(8, 2)(torch.zeros(2, 5, 8))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "doubleattentionspeakerverification_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f


def test_amsoftmax_and_frontend_cpu_math_match_golden():
    """AM-softmax and the VGG front-end are plain torch ops (device-agnostic); check them against the reference's
    vectors here so the GPU run only has to confirm device numerics."""
    from conftest import rnd, sd_from_golden
    from doubleattentionspeakerverification_amd.loss import AMSoftmax
    from doubleattentionspeakerverification_amd.CNNs import VGG3L, VGG4L
    g = load_golden("g5_amsoftmax")
    for tag, ann in (("plain", False), ("ann", True)):
        head = AMSoftmax(24, 37, m=0.4, s=30.0, annealing=ann)
        with torch.no_grad():
            head.W.copy_(torch.from_numpy(g[tag + "_W"]))
        for step in (0, 100, 5000, 1000000):
            costh, logits = head(torch.from_numpy(g["x"]), torch.from_numpy(g["label"]), step)
            np.testing.assert_allclose(costh.detach().numpy(), g["%s_s%d_costh" % (tag, step)], atol=1e-6)
            np.testing.assert_allclose(logits.detach().numpy(), g["%s_s%d_logits" % (tag, step)], atol=2e-5)
            assert abs(head.getAnnealedFactor(step) - float(g["%s_s%d_factor" % (tag, step)])) < 1e-12
    g3 = load_golden("g3_vgg")
    for L, cls in ((3, VGG3L), (4, VGG4L)):
        for ks in (16, 32):
            net = cls(ks).eval()
            sd = {k: v for k, v in sd_from_golden(g3, "L%d_k%d_" % (L, ks)).items() if k.startswith("conv")}
            net.load_state_dict(sd, strict=True)
            for T in (8, 31, 50):
                with torch.no_grad():
                    y = net(torch.from_numpy(rnd(700 + T, 2, T, 80)))
                ref = g3["L%d_k%d_T%d_y" % (L, ks, T)]
                assert y.shape == ref.shape
                np.testing.assert_allclose(y.numpy(), ref, atol=1e-5, rtol=1e-5)


def test_scripts_shims_star_import_like_the_reference():
    code = ("import sys; sys.path.append('./scripts/')\n"
            "from model import *\n"
            "from poolings import *\n"
            "from CNNs import *\n"
            "from loss import *\n"
            "assert SpeakerClassifier and DoubleMHA and VGG4L and AMSoftmax and torch and nn and F\n"
            "print('ok')\n")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr


def test_reference_style_checkpoint_loads_under_weights_only_default(tmp_path):
    """getEmbeddingExample.py:27 calls torch.load(path, map_location=device) with no weights_only argument; the
    checkpoint holds an argparse.Namespace (utils.py:36).  Importing our model module must make that work."""
    from doubleattentionspeakerverification_amd import model  # noqa: F401  (allow-lists Namespace on import)
    ns = argparse.Namespace(front_end="VGG4L", kernel_size=16, heads_number=4)
    path = str(tmp_path / "x.chkpt")
    torch.save({"model": {"w": torch.ones(2)}, "optimizer": {}, "settings": ns, "epoch": 1, "step": 2}, path)
    d = torch.load(path, map_location=torch.device("cpu"))
    assert d["settings"].kernel_size == 16 and d["step"] == 2


def test_length_bucketing_and_dealing():
    from doubleattentionspeakerverification_amd.batching import bucket_by_length, deal_buckets
    rs = np.random.RandomState(3)
    lengths = list(rs.randint(150, 801, size=400)) + [300] * 150
    buckets = bucket_by_length(lengths, max_batch=64, max_frames=64 * 300)
    seen = sorted(i for b in buckets for i in b)
    assert seen == list(range(len(lengths)))                      # a partition
    for b in buckets:
        assert len({lengths[i] for i in b}) == 1                   # equal lengths only
        assert len(b) <= 64 and len(b) * lengths[b[0]] <= 64 * 300 or len(b) == 1
    assert [lengths[b[0]] for b in buckets] == sorted((lengths[b[0]] for b in buckets), reverse=True)
    parts = deal_buckets(buckets, lengths, 8)
    loads = [sum(lengths[i] for b in p for i in b) for p in parts]
    assert sorted(i for p in parts for b in p for i in b) == list(range(len(lengths)))
    assert (max(loads) - min(loads)) / (sum(loads) / 8) < 0.05    # balanced to a few percent
    assert deal_buckets(buckets, lengths, 8) == parts               # deterministic
    with pytest.raises(ValueError):
        bucket_by_length([10, 0, 3])


def test_utils_helpers_match_reference_vectors(tmp_path):
    from doubleattentionspeakerverification_amd import utils
    g = load_golden("g6_scoring")
    assert utils.calculate_EER(g["eer_clients"], g["eer_impostors"]) == float(g["eer"])
    assert utils.calculate_EER(np.full(10, 0.9), np.full(10, -0.9)) == float(g["eer_separated"])
    rs = np.random.RandomState(0)
    cl, im = rs.normal(0.6, 0.15, 2000), rs.normal(0.1, 0.15, 2000)
    for t, far, frr in zip(g["score_th"], g["score_far"], g["score_frr"]):
        assert utils.Score(im, t, "FAR") == far and utils.Score(list(cl), t, "FRR") == frr
    cos = utils.scoreCosineDistance(torch.from_numpy(g["cos_e1"]), torch.from_numpy(g["cos_e2"]))
    np.testing.assert_allclose(cos.numpy(), g["cos"], atol=1e-7)
    assert utils.Accuracy(torch.from_numpy(g["acc_pred"]), torch.from_numpy(g["acc_label"])) == pytest.approx(float(g["acc"]))
    ns = argparse.Namespace(model_name="CNN", front_end="VGG4L", window_size=3.5, batch_size=64, gradientAccumulation=2,
                            learning_rate=0.0001, weight_decay=0.001, kernel_size=1024, embedding_size=400,
                            scalingFactor=30.0, marginFactor=0.4, pooling_method="DoubleMHA", heads_number=32)
    assert utils.getModelName(ns) == str(g["model_name"])
    # checkpoint round trip in the reference's layout and naming
    ns.out_dir, ns.model_name = str(tmp_path), "m"
    net = torch.nn.Linear(3, 2)
    opt = torch.optim.Adam(net.parameters())
    path = utils.chkptsave(ns, net, opt, 4, 99)
    assert path.endswith("m_99.chkpt")
    d = utils.load_checkpoint(path, torch.device("cpu"))
    assert sorted(d) == ["epoch", "model", "optimizer", "settings", "step"] and d["settings"].kernel_size == 1024
    lab = tmp_path / "labels.ndx"
    lab.write_text("a/1 spk1 0\nb/2 spk2 1\nc/3 spk1 0\n")
    assert utils.getNumberOfSpeakers(str(lab)) == 2


def test_ragged_tables_match_a_naive_walk():
    """Border tables of the ragged path (ragged.py): every row knows its utterance's span, every pooled row its two source
    rows, and lengths halve with ceil at every level."""
    from doubleattentionspeakerverification_amd.ragged import ragged_tables
    lengths = [5, 2, 3, 1, 8]
    spans, pools, lens = ragged_tables(lengths, 3, "cpu")
    cur = list(lengths)
    for lvl in range(3):
        assert list(lens[lvl]) == cur
        rows, start = [], 0
        for n in cur:
            rows += [[start, start + n]] * n
            start += n
        assert spans[lvl].tolist() == rows
        src, start = [], 0
        for n in cur:
            for o in range((n + 1) // 2):
                src.append([start + 2 * o, 1 if 2 * o + 1 < n else 0])
            start += n
        assert pools[lvl].tolist() == src
        cur = [(n + 1) // 2 for n in cur]
    assert list(lens[3]) == cur
    import pytest
    with pytest.raises(ValueError):
        ragged_tables([3, 0], 2, "cpu")
