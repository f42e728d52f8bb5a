"""Variable-length utterances: length-bucketed packing for the embedding path (BASELINE config 5).

The reference handles unequal lengths in exactly one place — validation scoring — by running ONE utterance per
forward at its full length (train.py:107-133): no padding, no masks anywhere in the convolutions or the pooling.
Zero-padding a batch is therefore NOT equivalent (ReLU(bias) != 0 in the pad region feeds the next conv; ceil-mode
pooling on odd lengths), so parity-preserving batching means: put utterances of EQUAL frame count in one batch.
`bucket_by_length` does that; `embed_utterances` runs a model over the buckets and returns embeddings in the input
order (bit-for-bit what B=1 runs give, up to the convolution library's batch-size-dependent algorithm choice);
`deal_buckets` spreads whole buckets over data-parallel ranks balancing total frames (longest-processing-time first).
The pooling kernel additionally accepts per-utterance `lengths` for zero-padded batches; that is exact for the
pooling stage alone (tests/test_dmha_gpu.py::test_lengths_equal_running_each_utterance_alone).
"""
import torch

__all__ = ["bucket_by_length", "deal_buckets", "embed_utterances"]


def bucket_by_length(lengths, max_batch=64, max_frames=None):
    """Group utterance indices by identical length.  Returns a list of index lists (each <= max_batch utterances and,
    if given, <= max_frames total frames), longest utterances first so the biggest launches go out first."""
    by_len = {}
    for i, n in enumerate(lengths):
        n = int(n)
        if n <= 0:
            raise ValueError("utterance %d has no frames" % i)
        by_len.setdefault(n, []).append(i)
    buckets = []
    for n in sorted(by_len, reverse=True):
        idx = by_len[n]
        cap = max_batch if max_frames is None else max(1, min(max_batch, max_frames // n))
        for k in range(0, len(idx), cap):
            buckets.append(idx[k:k + cap])
    return buckets


def deal_buckets(buckets, lengths, world):
    """Assign whole buckets to `world` ranks, balancing the sum of frames (LPT greedy).  Returns one list of buckets
    per rank; deterministic, so every rank can compute the same assignment without communication."""
    cost = [(sum(int(lengths[i]) for i in b), k) for k, b in enumerate(buckets)]
    load = [0] * world
    out = [[] for _ in range(world)]
    for c, k in sorted(cost, key=lambda t: (-t[0], t[1])):
        r = min(range(world), key=lambda j: (load[j], j))
        load[r] += c
        out[r].append(buckets[k])
    return out


@torch.no_grad()
def embed_utterances(net, utterances, max_batch=64, max_frames=None, autocast_dtype=None, rank=0, world=1):
    """utterances: list of [T_i, 80] tensors (any device).  Returns {index: embedding [emb]} for the utterances this
    rank owns (all of them when world == 1), computed bucket by bucket with net.getEmbedding."""
    lengths = [int(u.shape[0]) for u in utterances]
    buckets = bucket_by_length(lengths, max_batch, max_frames)
    mine = deal_buckets(buckets, lengths, world)[rank] if world > 1 else buckets
    dev = next(net.parameters()).device
    out = {}
    for b in mine:
        x = torch.stack([utterances[i] for i in b]).to(dev, non_blocking=True).float()
        with torch.autocast(dev.type, dtype=autocast_dtype or torch.float16, enabled=autocast_dtype is not None):
            emb = net.getEmbedding(x)
        for j, i in enumerate(b):
            out[i] = emb[j].float()
    return out
