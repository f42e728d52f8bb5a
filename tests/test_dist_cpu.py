"""world_size-2 gloo tests (CPU) of the data-parallel trainer: the flat-buffer bucketed all-reduce must give
the same parameters as one process training on the whole batch, with the reference's step semantics
(global-mean cross-entropy, summed gradient accumulation, parameters that never get a gradient)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class TinyClassifier(nn.Module):
    """Same call signature as SpeakerClassifier.forward (x, label, step) -> (costh, logits); has parameters that
    never receive gradients, like b1/b3 in the reference model."""

    def __init__(self):
        super().__init__()
        self.fc1 = nn.Linear(12, 16)
        self.unused = nn.BatchNorm1d(16)      # never called
        self.fc2 = nn.Linear(16, 16)
        self.bn = nn.BatchNorm1d(16)
        self.head = nn.Linear(16, 7, bias=False)

    def forward(self, x, label=None, step=0):
        h = torch.relu(self.fc1(x.mean(dim=1)))
        h = torch.relu(self.fc2(h))
        z = self.head(h)
        return z, 3.0 * z


def _data():
    rs = np.random.RandomState(7)
    x = torch.from_numpy(rs.standard_normal((2, 8, 5, 12)).astype(np.float32))   # [micro, batch, frames, feat]
    y = torch.from_numpy(rs.randint(0, 7, size=(2, 8)))
    return x, y


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, optimizer, accum, bucket_bytes, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from doubleattentionspeakerverification_amd.dist import DataParallelTrainer, init_distributed, shard_batch
    r, w, dev = init_distributed("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)          # different init per rank: broadcast_state must fix it
    net = TinyClassifier().train()
    net.bn.eval()                           # batch statistics differ between shards by design; keep the maths comparable
    tr = DataParallelTrainer(net, lr=1e-2, weight_decay=1e-3, optimizer=optimizer, accum=accum, bucket_bytes=bucket_bytes)
    x, y = _data()
    for mb in range(accum):
        lo, hi = shard_batch(x.size(1), rank, world)
        tr.train_step(x[mb, lo:hi], y[mb, lo:hi], step=0)
    torch.save({k: v.clone() for k, v in net.state_dict().items()}, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.destroy_process_group()


def _single_process(optimizer, accum):
    torch.manual_seed(100)                  # rank 0's init
    net = TinyClassifier().train()
    net.bn.eval()
    params = list(net.parameters())
    opt = {"Adam": lambda: torch.optim.Adam(params, lr=1e-2, weight_decay=1e-3),
           "SGD": lambda: torch.optim.SGD(params, lr=1e-2, weight_decay=1e-3)}[optimizer]()
    crit = nn.CrossEntropyLoss()
    x, y = _data()
    for mb in range(accum):                 # train.py:219-226: grads of micro-batches are summed
        _, logits = net(x[mb], label=y[mb])
        crit(logits, y[mb]).backward()
    opt.step()
    return net.state_dict()


@pytest.mark.parametrize("optimizer,accum,bucket_bytes", [("Adam", 1, 1 << 20), ("SGD", 2, 256), ("Adam", 2, 64)])
def test_two_rank_gloo_step_matches_single_process(tmp_path, optimizer, accum, bucket_bytes):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, optimizer, accum, bucket_bytes, str(tmp_path)), nprocs=2, join=True)
    ref = _single_process(optimizer, accum)
    r0 = torch.load(os.path.join(str(tmp_path), "rank0.pt"))
    r1 = torch.load(os.path.join(str(tmp_path), "rank1.pt"))
    for k in ref:
        assert torch.equal(r0[k], r1[k]), "ranks diverged on %s" % k
        np.testing.assert_allclose(r0[k].numpy(), ref[k].numpy(), rtol=2e-5, atol=2e-6, err_msg=k)
    # the never-used BatchNorm is untouched: no update, no weight decay (grad is None in the reference too)
    assert float((r0["unused.weight"] - 1.0).abs().max()) == 0.0


def test_flat_buffer_layout_and_buckets():
    from doubleattentionspeakerverification_amd.dist import FlatGradReducer
    net = TinyClassifier()
    red = FlatGradReducer(net.parameters(), bucket_bytes=512)
    params = [p for p in net.parameters()]
    # every grad is a view into the flat buffer, 256-byte aligned, last-registered parameter first
    base = red.flat.data_ptr()
    offs = [(p.grad.data_ptr() - base) for p in params]
    assert all(o % 256 == 0 for o in offs)
    assert offs[-1] == 0 and sorted(offs, reverse=True) == offs
    covered = sorted((lo, hi) for lo, hi, _ in red.buckets)
    assert covered[0][0] == 0 and covered[-1][1] == red.flat.numel()
    assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    assert sorted(i for _, _, mem in red.buckets for i in mem) == list(range(len(params)))
    # single process: arm/finish is a no-op that still leaves summed grads in place
    red.arm()
    z, logits = net(torch.randn(4, 3, 12), label=None)
    logits.sum().backward()
    red.finish()
    assert float(net.fc1.weight.grad.abs().sum()) > 0 and float(net.unused.weight.grad.abs().sum()) == 0.0
    red.zero()
    assert float(red.flat.abs().sum()) == 0.0


def test_shard_batch():
    from doubleattentionspeakerverification_amd.dist import shard_batch
    assert [shard_batch(64, r, 8) for r in (0, 7)] == [(0, 8), (56, 64)]
    with pytest.raises(ValueError):
        shard_batch(10, 0, 4)


class LateBloomer(TinyClassifier):
    """A branch whose parameters get their first gradient only from the second step on."""

    def __init__(self):
        super().__init__()
        self.extra = nn.Linear(16, 16)
        self.use_extra = False

    def forward(self, x, label=None, step=0):
        h = torch.relu(self.fc1(x.mean(dim=1)))
        h = torch.relu(self.fc2(h))
        if self.use_extra:
            h = h + torch.tanh(self.extra(h))
        z = self.head(h)
        return z, 3.0 * z


def _worker_steps(rank, world, port, out_dir, lopsided=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from doubleattentionspeakerverification_amd.dist import DataParallelTrainer, init_distributed, shard_batch
    init_distributed("gloo")
    torch.manual_seed(100)
    net = LateBloomer().train()
    net.bn.eval()
    tr = DataParallelTrainer(net, lr=1e-2, weight_decay=0.0, optimizer="SGD", bucket_bytes=256)
    x, y = _data()
    lo, hi = shard_batch(x.size(1), rank, world)
    from_hook = []
    for step in range(3):
        # lopsided: from step 1 on the late layer fires on rank 0 ONLY (a data-dependent graph)
        net.use_extra = step >= 1 and (rank == 0 or not lopsided)
        tr.train_step(x[step % 2, lo:hi], y[step % 2, lo:hi], step=step)
        from_hook.append(list(tr.reducer.launched_from_hook))
    cold = sorted(tr.reducer._cold)
    names = [n for n, _ in net.named_parameters()]
    torch.save({"state": {k: v.clone() for k, v in net.state_dict().items()}, "from_hook": from_hook,
                "cold": [names[i] for i in cold]}, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.destroy_process_group()


def test_unused_parameters_do_not_hold_back_buckets_and_late_gradients_are_reduced(tmp_path):
    """a bucket holding parameters that never fire was only reduced in finish().  After the first reduced step
    every bucket must be launched from an autograd hook; a parameter whose first gradient comes later is still averaged
    over the ranks."""
    port = _free_port()
    mp.spawn(_worker_steps, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    torch.manual_seed(100)
    net = LateBloomer().train()
    net.bn.eval()
    opt = torch.optim.SGD([p for p in net.parameters()], lr=1e-2)
    x, y = _data()
    for step in range(3):
        net.use_extra = step >= 1
        opt.zero_grad(set_to_none=True)
        _, logits = net(x[step % 2], label=y[step % 2])
        nn.CrossEntropyLoss()(logits, y[step % 2]).backward()
        opt.step()
    r0 = torch.load(os.path.join(str(tmp_path), "rank0.pt"))
    r1 = torch.load(os.path.join(str(tmp_path), "rank1.pt"))
    for k, v in net.state_dict().items():
        assert torch.equal(r0["state"][k], r1["state"][k]), k
        np.testing.assert_allclose(r0["state"][k].numpy(), v.numpy(), rtol=2e-5, atol=2e-6, err_msg=k)
    assert float((r0["state"]["extra.weight"] - LateBloomerInit.extra_weight()).abs().max()) > 0   # it did train
    assert set(r0["cold"]) == {"unused.weight", "unused.bias", "bn.weight", "bn.bias", "extra.weight", "extra.bias"}
    assert not all(r0["from_hook"][0])          # first step: some bucket waited for the parameters that never fire
    assert all(r0["from_hook"][1]) and all(r0["from_hook"][2])


def test_a_gradient_that_fires_on_one_rank_only_keeps_the_ranks_in_step(tmp_path):
    """which parameters fired is agreed over the ranks before anything is retired to the cold buffers or hidden from the
    optimizer.  A layer that (from step 1 on) takes part in the graph on rank 0 only must neither change the sequence of
    collectives (a hang) nor be updated on one rank and skipped on the other (diverging weights): both ranks apply the reduced
    gradient — rank 0's contribution over the two ranks."""
    port = _free_port()
    mp.spawn(_worker_steps, args=(2, port, str(tmp_path), True), nprocs=2, join=True)
    r0 = torch.load(os.path.join(str(tmp_path), "rank0.pt"))
    r1 = torch.load(os.path.join(str(tmp_path), "rank1.pt"))
    for k in r0["state"]:
        assert torch.equal(r0["state"][k], r1["state"][k]), k
    assert float((r0["state"]["extra.weight"] - LateBloomerInit.extra_weight()).abs().max()) > 0   # it did train, on both
    assert r0["cold"] == r1["cold"]


def _ragged_data():
    rs = np.random.RandomState(17)
    lens = [5, 9, 5, 3, 7, 9, 4, 6]
    return [torch.from_numpy(rs.standard_normal((n, 12)).astype(np.float32)) for n in lens], torch.from_numpy(rs.randint(0, 7, size=8))


class RaggedTiny(TinyClassifier):
    """x may be a LIST of [T_i, feat] utterances (a variable-length shard), like SpeakerClassifier.forward."""

    def forward(self, x, label=None, step=0):
        m = torch.stack([u.mean(dim=0) for u in x]) if isinstance(x, (list, tuple)) else x.mean(dim=1)
        h = torch.relu(self.fc2(torch.relu(self.fc1(m))))
        z = self.head(h)
        return z, 3.0 * z


def _worker_ragged(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from doubleattentionspeakerverification_amd.dist import DataParallelTrainer, init_distributed, shard_batch
    init_distributed("gloo")
    torch.manual_seed(100)
    net = RaggedTiny().train()
    net.bn.eval()
    tr = DataParallelTrainer(net, lr=1e-2, weight_decay=1e-3, optimizer="SGD", bucket_bytes=256)
    utts, y = _ragged_data()
    lo, hi = shard_batch(len(utts), rank, world)
    tr.train_step(utts[lo:hi], y[lo:hi], step=0)          # a LIST: this rank's utterances, each at its own length
    torch.save({k: v.clone() for k, v in net.state_dict().items()}, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.destroy_process_group()


def test_variable_length_shards_on_two_gloo_ranks_match_the_full_minibatch(tmp_path):
    """BASELINE.json north_star: minibatches of variable-length utterances sharded over data-parallel ranks.  Each rank gets a LIST
    of utterances of different lengths (equal COUNT per rank, so that the global-batch mean of train.py:219's CrossEntropyLoss is
    the mean of the rank means); one all-reduced step must equal the single-process step on the whole list."""
    port = _free_port()
    mp.spawn(_worker_ragged, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    torch.manual_seed(100)
    net = RaggedTiny().train()
    net.bn.eval()
    opt = torch.optim.SGD(list(net.parameters()), lr=1e-2, weight_decay=1e-3)
    utts, y = _ragged_data()
    nn.CrossEntropyLoss()(net(utts, label=y)[1], y).backward()
    opt.step()
    r0 = torch.load(os.path.join(str(tmp_path), "rank0.pt"))
    r1 = torch.load(os.path.join(str(tmp_path), "rank1.pt"))
    for k, v in net.state_dict().items():
        assert torch.equal(r0[k], r1[k]), k
        np.testing.assert_allclose(r0[k].numpy(), v.numpy(), rtol=2e-5, atol=2e-6, err_msg=k)


class LateBloomerInit:
    @staticmethod
    def extra_weight():
        torch.manual_seed(100)
        return LateBloomer().extra.weight.detach().clone()


# ------------------------------------------------------------------ bench.py's multi-rank path, and config 5's dealing
def _bench_worker(rank, world, port, out_dir):
    """bench.py's own rank code path (timed_regions: barrier + max-over-ranks regions, median region; headline: whole-job
    value, world size and backend as torch.distributed reports them) on two gloo ranks, around a data-parallel training step."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import json
    import time
    import bench
    from doubleattentionspeakerverification_amd.dist import DataParallelTrainer, init_distributed, shard_batch
    r, w, dev = init_distributed("gloo")
    torch.manual_seed(100 + rank)
    net = TinyClassifier().train()
    tr = DataParallelTrainer(net, lr=1e-2, weight_decay=1e-3, optimizer="SGD", accum=1, bucket_bytes=256)
    x, y = _data()
    lo, hi = shard_batch(x.size(1), rank, world)
    calls = []

    def run_steps(first, count):
        for i in range(count):
            calls.append(first + i)
            tr.train_step(x[0, lo:hi], y[0, lo:hi], step=first + i)
            if rank == 1:
                time.sleep(0.002)          # the slower rank sets the pace: the MAX over ranks must show it on rank 0

    elapsed, ev_ms, els = bench.timed_regions(run_steps, 2, 3, 4, dist, torch.device("cpu"))
    assert ev_ms is None and len(els) == 4 and sorted(els)[2] == elapsed
    assert calls == list(range(2 + 3 * 4))                      # W warm-up steps once, then exactly K steps per region
    line = bench.headline("training utterances/sec", "utterances/s", hi - lo, world, 3, 2, elapsed, els, dist)
    with open(os.path.join(out_dir, "bench%d.json" % rank), "w") as f:
        json.dump(line, f)
    dist.destroy_process_group()


def test_bench_rank_code_path_on_two_gloo_ranks(tmp_path):
    import json
    port = _free_port()
    mp.spawn(_bench_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = (json.load(open(os.path.join(str(tmp_path), "bench%d.json" % r))) for r in (0, 1))
    assert a["n_gpus"] == 2 and a["dist"] == {"world_size": 2, "backend": "gloo"} and a["scaling"] == "weak"
    assert a["regions"]["elapsed_s"] == b["regions"]["elapsed_s"]        # max over ranks: both ranks hold the same numbers
    assert min(a["regions"]["elapsed_s"]) >= 3 * 0.002                   # and they include the slower rank's time
    assert abs(a["value"] - 4 * 2 * 3 / (a["ms_per_step"] * 3 / 1e3)) <= 0.01 * a["value"]   # whole-job: both ranks' 4 utterances per step
    assert a["ms_per_step"] * a["steps"] / 1e3 == pytest.approx(sorted(a["regions"]["elapsed_s"])[2], rel=1e-3)


@pytest.mark.parametrize("world", [2, 8])
def test_ragged_groups_partition_the_set_and_balance_the_frames(world):
    """BASELINE config 5 across ranks (ragged packing): every utterance is embedded by exactly one rank, every group respects
    the frame budget, and the ranks' total frames agree within 5 % (VERDICT r2 task 8)."""
    from doubleattentionspeakerverification_amd.ragged import plan_ragged_groups
    rs = np.random.RandomState(1234)
    for n in (1024, 37):
        lens = rs.randint(150, 801, size=n).tolist()
        plans = [plan_ragged_groups(lens, 64 * 300, r, world) for r in range(world)]
        owned = [i for p in plans for g in p for i in g]
        assert sorted(owned) == list(range(n))
        for p in plans:
            for g in p:
                assert sum(lens[i] for i in g) <= 64 * 300 or len(g) == 1
        load = [sum(lens[i] for g in p for i in g) for p in plans]
        if n >= 64 * world:
            assert (max(load) - min(load)) <= 0.05 * (sum(load) / world), load
        else:
            assert max(load) - min(load) <= max(lens)            # LPT: never more than one utterance apart
    assert plan_ragged_groups([300, 10, 299], 400, 0, 1) == [[0], [2, 1]]
