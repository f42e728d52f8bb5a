"""Scoring / bookkeeping helpers with the reference's names (scripts/utils.py, scripts/train.py:135-150), vectorised
and free of per-sample host synchronisations.

    Score(SC, th, rate)                 utils.py:5-15    FAR / FRR in percent, rounded to 4 decimals
    scoreCosineDistance(emb1, emb2)     utils.py:18-21   cosine similarity (dim -1, eps 1e-8)
    chkptsave(opt, model, optimizer, epoch, step)   utils.py:23-40   same dict layout and file name
    Accuracy(pred, labels)              utils.py:42-51   the reference loops over the batch with two .item() calls per
                                                         sample (2B host syncs per step); here one device reduction
    getNumberOfSpeakers, getModelName   utils.py:53-69
    calculate_EER(CL, IM)               train.py:135-150 thresholds -1..1 step 0.01, first sign change of FAR-FRR
"""
import numpy as np
import torch
from torch.nn import functional as F

__all__ = ["Score", "scoreCosineDistance", "chkptsave", "Accuracy", "accuracy_tensor", "getNumberOfSpeakers",
           "getModelName", "calculate_EER", "load_checkpoint"]


def Score(SC, th, rate):
    sc = np.asarray([float(s) for s in SC] if not isinstance(SC, np.ndarray) else SC, dtype=np.float64)
    th = float(th)
    if rate == "FAR":
        count = np.count_nonzero(sc >= th)
    elif rate == "FRR":
        count = np.count_nonzero(sc < th)
    else:
        count = 0
    return round(float(count) * 100 / float(len(sc)), 4)


def scoreCosineDistance(emb1, emb2):
    return F.cosine_similarity(emb1, emb2, dim=-1, eps=1e-08)


def accuracy_tensor(pred, labels):
    """Fraction of rows whose arg-max equals the label, as a 0-dim device tensor (no host sync)."""
    return (torch.max(pred, 1)[1] == labels).float().mean()


def Accuracy(pred, labels):
    return float(accuracy_tensor(pred, labels))


def chkptsave(opt, model, optimizer, epoch, step):
    """Write '{out_dir}/{model_name}_{step}.chkpt' holding {'model','optimizer','settings','epoch','step'}; a wrapped
    (DataParallel-style `.module`) model is unwrapped like the reference does."""
    net = model.module if hasattr(model, "module") else model
    checkpoint = {"model": net.state_dict(), "optimizer": optimizer.state_dict(), "settings": opt, "epoch": epoch, "step": step}
    path = "{}/{}_{}.chkpt".format(opt.out_dir, opt.model_name, step)
    torch.save(checkpoint, path)
    return path


def load_checkpoint(path, device):
    """The read side (train.py:37-45, getEmbeddingExample.py:27): returns the checkpoint dict."""
    from . import model as _model  # noqa: F401  (allow-lists argparse.Namespace for weights_only loading)
    return torch.load(path, map_location=device)


def getNumberOfSpeakers(labelsFilePath):
    speakers = set()
    with open(labelsFilePath, "r") as f:
        for line in f:
            parts = line.split()
            if len(parts) > 1:
                speakers.add(parts[1])
    return len(speakers)


def getModelName(params):
    name = params.model_name
    name += "_{}_{}".format(params.front_end, params.window_size)
    name += "_{}batchSize".format(params.batch_size * params.gradientAccumulation)
    name += "_{}lr_{}weightDecay_{}kernel".format(params.learning_rate, params.weight_decay, params.kernel_size)
    name += "_{}embSize_{}s_{}m".format(params.embedding_size, params.scalingFactor, params.marginFactor)
    name += "_{}_{}".format(params.pooling_method, params.heads_number)
    return name


def calculate_EER(CL, IM):
    """Equal error rate in percent from client / impostor score lists (train.py:135-150), all thresholds at once."""
    cl = np.asarray(CL, dtype=np.float64).reshape(-1)
    im = np.asarray(IM, dtype=np.float64).reshape(-1)
    th = np.arange(-1, 1, 0.01)
    frr = np.round((cl[None, :] < th[:, None]).sum(axis=1) * 100.0 / len(cl), 4)
    far = np.round((im[None, :] >= th[:, None]).sum(axis=1) * 100.0 / len(im), 4)
    idx = np.argwhere(np.diff(np.sign(far - frr)) != 0).reshape(-1)
    if len(idx) == 0:
        return 50.00
    i = int(idx[0])
    return round((far[i] + frr[i]) / 2, 4)
