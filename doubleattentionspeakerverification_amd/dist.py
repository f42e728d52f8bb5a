"""Data-parallel training for the speaker classifier: one process per GPU, RCCL over xGMI.

Replaces the reference's single-process ``nn.DataParallel`` wrap (train.py:63-70), which re-broadcasts all
21.5 M parameters (86 MB) to every GPU on every forward and reduces gradients onto GPU 0 each backward.
Here parameters stay resident on every rank and the ONLY per-step traffic is one all-reduce of the flat
fp32 gradient buffer, cut into buckets that are launched from autograd hooks as soon as their gradients
exist, so the reduction of the big late-layer tensors (conv42/conv41, 56 MB) overlaps the rest of backward.
xGMI is a point-to-point mesh (7 links x ~153 GB/s per GPU): a few large buckets keep every link busy;
many small ones would be launch-bound.

Step semantics follow train.py:215-226: cross-entropy is a mean over the GLOBAL batch (equal shards =>
mean of rank means => gradients are averaged over ranks); gradient accumulation SUMS micro-batch
gradients (no division); the optimizer is Adam(lr, weight_decay) / SGD / RMSprop as train.py:82-88.
Parameters that never receive a gradient (b1.*, b3.* — model.py:46,50) keep a zero slot in the buffer and
are hidden from the optimizer (grad None), so like in the reference they get no update and no weight decay.
"""
import os

import torch
import torch.distributed as dist
from torch import nn

__all__ = ["init_distributed", "FlatGradReducer", "FlatAdam", "DataParallelTrainer", "shard_batch"]


def init_distributed(backend=None):
    """Initialise torch.distributed from the torchrun environment.  Returns (rank, world, device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    device = torch.device("cuda", local) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if use_gpu else "gloo")  # "nccl" IS RCCL on ROCm
        kw = {"device_id": device} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, device


def shard_batch(n_items, rank, world):
    """Contiguous equal shard [lo, hi) of a global batch; requires divisibility so the CE mean stays global."""
    if n_items % world != 0:
        raise ValueError("global batch %d is not divisible by %d ranks" % (n_items, world))
    per = n_items // world
    return rank * per, (rank + 1) * per


class FlatGradReducer:
    """All parameter gradients live in ONE contiguous fp32 buffer; buckets of it are all-reduced
    asynchronously the moment every gradient in the bucket has been accumulated."""

    def __init__(self, params, bucket_bytes=32 << 20, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, torch.float32
        sizes = [p.numel() for p in self.params]
        # 64-element (256 B) aligned slots, laid out in REVERSE registration order: backward produces the
        # gradients of the last layers first, so bucket 0 fills first.
        order = list(range(len(self.params)))[::-1]
        offs, total = {}, 0
        for i in order:
            offs[i] = total
            total += (sizes[i] + 63) // 64 * 64
        self.flat = torch.zeros(total, dtype=dt, device=dev)
        self.offs, self.sizes = offs, sizes      # slot of parameter i: flat[offs[i] : offs[i] + sizes[i]] (FlatAdam shares the layout)
        self.buckets = []  # (lo, hi, [param indices])
        cap = max(1, bucket_bytes // 4)
        lo, members = 0, []
        for i in order:
            members.append(i)
            hi = offs[i] + (sizes[i] + 63) // 64 * 64
            if hi - lo >= cap:
                self.buckets.append((lo, hi, members))
                lo, members = hi, []
        if members:
            self.buckets.append((lo, total, members))
        self.bucket_of = {}
        for bi, (_, _, mem) in enumerate(self.buckets):
            for i in mem:
                self.bucket_of[i] = bi
        self.views = []
        for i, p in enumerate(self.params):
            if p.dtype != dt:
                raise TypeError("FlatGradReducer expects fp32 master parameters")
            self.views.append(self.flat[offs[i]:offs[i] + sizes[i]].view_as(p))
            p.grad = self.views[i]
        self._fired = [False] * len(self.params)
        # COLD parameters: those the first reduced backward produced no gradient for (b1.* / b3.*, model.py:46,50, never
        # fire).  A bucket that waited for them would only be reduced in finish(), after the whole backward, so after that
        # first step their gradients are re-pointed at a separate small buffer: the buckets stop counting them, and should
        # one fire in a later step it accumulates there — never into a range whose all-reduce is already in flight — and
        # is reduced in finish().
        self._cold = set()
        self._cold_bufs = []
        self._cold_fired = False
        self._fired_armed = [False] * len(self.params)
        self._pending = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self.launched_from_hook = [False] * len(self.buckets)
        self._in_finish = False
        self._works = []
        self._armed = False
        for i, p in enumerate(self.params):
            p.register_post_accumulate_grad_hook(self._make_hook(i))

    def _make_hook(self, i):
        def hook(_p):
            self._fired[i] = True
            if not self._armed:
                return
            self._fired_armed[i] = True
            if i in self._cold:
                self._cold_fired = True
                return
            bi = self.bucket_of[i]
            self._pending[bi] -= 1
            if self._pending[bi] == 0:
                self._launch(bi)
        return hook

    def _launch(self, bi):
        if not self._launched[bi]:
            self.launched_from_hook[bi] = not self._in_finish
        if self._launched[bi] or self.world == 1:
            self._launched[bi] = True
            return
        lo, hi, _ = self.buckets[bi]
        self._launched[bi] = True
        self._works.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def arm(self):
        """Call before the backward pass whose gradients are to be reduced (the LAST micro-batch)."""
        self._armed = True
        self._pending = [sum(1 for i in mem if i not in self._cold) for _, _, mem in self.buckets]
        # a bucket of cold parameters only is a dead range of the flat buffer (always zero): nothing to reduce
        self._launched = [n == 0 for n in self._pending]
        self.launched_from_hook = [n == 0 for n in self._pending]
        self._fired_armed = [False] * len(self.params)
        self._cold_fired = False
        self._works = []

    def _retire_unfired(self):
        """Move the parameters the reduced backward left without a gradient to the cold buffer (same on every rank: the
        graph is)."""
        new = [i for i in range(len(self.params)) if i not in self._cold and not self._fired_armed[i]]
        if not new:
            return
        sizes = [(self.params[i].numel() + 63) // 64 * 64 for i in new]
        buf = torch.zeros(sum(sizes), dtype=self.flat.dtype, device=self.flat.device)
        off = 0
        for i, n in zip(new, sizes):
            v = buf[off:off + self.params[i].numel()].view_as(self.params[i])
            v.copy_(self.views[i])       # non-zero only if an un-reduced micro-batch produced it
            self.views[i].zero_()
            self.views[i] = v
            self.params[i].grad = v
            self._cold.add(i)
            off += n
        self._cold_bufs.append(buf)

    def finish(self):
        """After backward: reduce buckets whose hooks never all fired, wait.  The buffers then hold the SUM over the ranks: the
        caller scales its loss by `grad_scale` (= 1 / world) before backward — DataParallelTrainer does — so that no extra pass
        over the 86 MB buffer is needed to turn the sum into the global-batch mean (train.py:219: CrossEntropyLoss is a mean)."""
        self._in_finish = True
        try:
            for bi in range(len(self.buckets)):
                if not self._launched[bi]:
                    self._launch(bi)
        finally:
            self._in_finish = False
        # the cold buffers are reduced whenever they exist — not only when THIS rank saw one of their parameters fire: a gradient
        # that appears on one rank only (data-dependent graph) would otherwise make the ranks issue different collectives
        cold = bool(self._cold_bufs) and self.world > 1
        if cold:
            for buf in self._cold_bufs:
                self._works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        was_armed, self._armed = self._armed, False
        if self.world > 1 and was_armed:
            # Which parameters fired is agreed over the ranks (MAX) before anything is decided from it: what is retired to the cold
            # buffers (a rank-local decision could leave the ranks with different numbers of cold buffers, i.e. different collective
            # sequences: a hang) and what hide_unused() hides from the optimizer (a parameter hidden on one rank and updated on
            # another: diverging weights) — ADVICE r3.  One tiny collective and one host read per optimizer step.
            flags = torch.tensor([self._fired, self._fired_armed], dtype=torch.int32).to(self.flat.device)
            self._works.append(dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=self.group, async_op=True))
        for w in self._works:
            w.wait()
        self._works = []
        if self.world > 1 and was_armed:
            fl = flags.cpu().tolist()
            self._fired = [bool(v) for v in fl[0]]
            self._fired_armed = [bool(v) for v in fl[1]]
        if was_armed:      # finish() without arm() saw no backward: nothing may be retired on its evidence
            self._retire_unfired()

    @property
    def grad_scale(self):
        """What the caller multiplies its (mean) loss by so that the summed buffers hold the global-batch mean's gradient."""
        return 1.0 / self.world

    def zero(self, flat_cleared=False):
        if not flat_cleared:         # (FlatAdam clears the gradient buffer in its own pass)
            self.flat.zero_()
        for buf in self._cold_bufs:
            buf.zero_()
        self._fired = [False] * len(self.params)

    def hide_unused(self):
        """Parameters that received no gradient since the last zero() must look like `grad is None` to the
        optimizer, exactly as in the reference (b1.*, b3.*: no update, no weight decay)."""
        hidden = [i for i, f in enumerate(self._fired) if not f]
        for i in hidden:
            self.params[i].grad = None
        return hidden

    def restore(self, hidden):
        for i in hidden:
            self.params[i].grad = self.views[i]


class FlatAdam:
    """`optim.Adam(net.parameters(), lr, weight_decay)` of train.py:82-84 as ONE hand-written launch per step (csrc/adam.hip,
    dasv_adam_step_f32): parameters, gradients (FlatGradReducer's buffer) and both moments share one flat layout, so `optimizer.step()`
    (train.py:224) is a single pass over 4 x 86 MB that also clears the gradient buffer.  The parameters of the module are re-pointed at
    views of the flat parameter buffer (same values, same shapes: state_dict / checkpoints are unaffected).  Parameters that received no
    gradient are skipped — no update, no weight decay — like `grad is None` in the reference (model.py:46,50: b1.*, b3.*).
    `param_groups[0]` carries lr / weight_decay / betas / eps and is read at every step (a scheduler or train.py's manual decay may
    rewrite it).  `state_dict()` (what utils.chkptsave stores under 'optimizer', utils.py:23-40) is this class's own layout: the step
    count and the two flat moment buffers.  GPU only; torch's arithmetic."""

    def __init__(self, reducer, lr=1e-4, weight_decay=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.reducer = reducer
        flat = reducer.flat
        if not flat.is_cuda:
            raise RuntimeError("FlatAdam runs on the GPU only (libdasv_hip.so); use torch.optim.Adam on the CPU")
        self.p = torch.zeros_like(flat)
        self.exp_avg, self.exp_avg_sq = torch.zeros_like(flat), torch.zeros_like(flat)
        with torch.no_grad():
            for i, prm in enumerate(reducer.params):
                view = self.p[reducer.offs[i]:reducer.offs[i] + reducer.sizes[i]].view_as(prm)
                view.copy_(prm)
                prm.data = view
        self.param_groups = [{"lr": lr, "weight_decay": weight_decay, "betas": tuple(betas), "eps": eps}]
        self.step_count = 0
        self._active_key, self._active = None, None

    def _active_blocks(self, hidden):
        key = frozenset(hidden)
        if key != self._active_key:
            act = torch.ones(self.p.numel() // 64, dtype=torch.uint8)
            for i in hidden:
                lo = self.reducer.offs[i] // 64
                act[lo:lo + (self.reducer.sizes[i] + 63) // 64] = 0
            self._active, self._active_key = act.to(self.p.device), key
        return self._active

    def step(self, hidden=()):
        """hidden: indices (into reducer.params) of the parameters without a gradient this step (FlatGradReducer.hide_unused).  Clears
        the reducer's flat gradient buffer."""
        from . import _lib
        r = self.reducer
        hidden = set(hidden)
        with torch.no_grad():
            for i in r._cold:             # a retired parameter that fired after all: its gradient sits in a cold buffer
                if i not in hidden:
                    r.flat[r.offs[i]:r.offs[i] + r.sizes[i]].copy_(r.views[i].reshape(-1))
        act = self._active_blocks(hidden)
        g = self.param_groups[0]
        self.step_count += 1
        with torch.cuda.device(self.p.device):
            rc = _lib.load().dasv_adam_step_f32(self.p.data_ptr(), r.flat.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                                act.data_ptr(), self.p.numel(), float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                                                float(g["eps"]), float(g["weight_decay"]), self.step_count, 1,
                                                torch.cuda.current_stream(self.p.device).cuda_stream)
        _lib.check(rc, "dasv_adam_step_f32")

    def zero_grad(self, set_to_none=False):
        self.reducer.zero()

    def state_dict(self):
        return {"flat_adam": True, "step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "param_groups": [dict(g) for g in self.param_groups]}

    def load_state_dict(self, sd):
        if not sd.get("flat_adam"):
            raise ValueError("FlatAdam.load_state_dict: not a FlatAdam state (a torch.optim.Adam checkpoint keeps per-parameter moments; "
                             "construct the trainer with flat_adam=False to resume from it)")
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.param_groups = [dict(g) for g in sd["param_groups"]]


class DataParallelTrainer:
    """One training step per call, with the reference's step semantics (train.py:186-198, 215-226)."""

    def __init__(self, net, lr=1e-4, weight_decay=1e-3, optimizer="Adam", accum=1, bucket_bytes=32 << 20,
                 autocast_dtype=None, group=None, flat_adam=True):
        self.net = net
        self.accum = int(accum)
        self.autocast_dtype = autocast_dtype
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if self.world > 1:
            self.broadcast_state()
        self.reducer = FlatGradReducer(net.parameters(), bucket_bytes=bucket_bytes, group=group)
        params = list(net.parameters())
        on_gpu = params[0].is_cuda
        if optimizer == "Adam" and on_gpu and flat_adam and all(p.dtype == torch.float32 for p in params):
            self.opt = FlatAdam(self.reducer, lr=lr, weight_decay=weight_decay)      # one hand-written launch per step
        elif optimizer == "Adam":
            self.opt = torch.optim.Adam(params, lr=lr, weight_decay=weight_decay, fused=on_gpu)
        elif optimizer == "SGD":
            self.opt = torch.optim.SGD(params, lr=lr, weight_decay=weight_decay)
        elif optimizer == "RMSprop":
            self.opt = torch.optim.RMSprop(params, lr=lr, weight_decay=weight_decay)
        else:
            raise ValueError(optimizer)
        self.criterion = nn.CrossEntropyLoss()
        self.fused_loss = True     # use net.forward_loss (fused AM-softmax + CE head) when the model has one
        self._micro = 0

    def sync_buffers(self):
        """Rank 0's BUFFERS (BatchNorm running statistics) to every rank.  Policy across ranks: the reference's nn.DataParallel
        keeps the statistics of the replica on GPU 0 — the replicas' buffer updates are dropped at the end of every forward
        (train.py:68-70) — i.e. those of the first shard of every batch.  Here every rank updates its own module
        from its own shard, so rank 0's module holds exactly what the reference's would: checkpoints are written by rank 0 from
        its own state (utils.py:23-40) and need nothing else; ranks > 0 call this before they evaluate (validation, scoring), or
        their drifted statistics would be used.  Training itself never reads the running statistics (b2 normalises with batch
        statistics in train mode, model.py:57)."""
        if self.world > 1:
            with torch.no_grad():
                for t in self.net.buffers():
                    dist.broadcast(t, src=0, group=self.group)

    def broadcast_state(self):
        """Rank 0's parameters and buffers to everyone (initialisation)."""
        from .conv import invalidate_packed_weights
        with torch.no_grad():   # in place on the tensors themselves, so their version counters move
            for t in list(self.net.parameters()) + list(self.net.buffers()):
                dist.broadcast(t, src=0, group=self.group)
        invalidate_packed_weights(self.net)

    def train_step(self, x, labels, step=0):
        """Forward/backward on this rank's shard.  Every `accum`-th call all-reduces and steps the optimizer.
        Returns (loss, costh) as device tensors — no host synchronisation.
        x: [B,T,80], or a LIST of [T_i,80] utterances of different lengths (a variable-length shard: the model processes every
        utterance at its own length, model.SpeakerClassifier._utterance_vectors_ragged; every rank must hold the same NUMBER of
        utterances so that the mean over the global batch stays the mean of the rank means)."""
        self._micro += 1
        last = (self._micro % self.accum == 0)
        if last:
            self.reducer.arm()
        dev_type = "cuda" if (x[0] if isinstance(x, (list, tuple)) else x).is_cuda else "cpu"
        fused = getattr(self.net, "forward_loss", None) if self.fused_loss else None
        with torch.autocast(dev_type, dtype=self.autocast_dtype or torch.bfloat16, enabled=self.autocast_dtype is not None):
            if fused is not None:
                # AM-softmax + cross-entropy as one fused head (csrc/amsoftmax.hip): no [B,N] logits, one-hot or log-softmax
                loss, costh = fused(x, labels, step)
            else:
                costh, logits = self.net(x, label=labels, step=step)
        if fused is None:
            loss = self.criterion(logits.float(), labels)
        # the all-reduce SUMS over the ranks: 1 / world goes into the loss (exact for 2, 4, 8 ranks) instead of a pass over the
        # flat gradient buffer afterwards; every micro-batch of an accumulation is scaled, they are summed into the same buffer
        scale = self.reducer.grad_scale
        (loss if scale == 1.0 else loss * scale).backward()
        if last:
            self.reducer.finish()
            hidden = self.reducer.hide_unused()
            own = isinstance(self.opt, FlatAdam)
            if own:
                self.opt.step(hidden)
            else:
                self.opt.step()
            # (the fused optimizers do not move the parameters' version counters: packed copies of the convolution weights made by
            # a no-grad forward before this step must not survive it)
            from .conv import invalidate_packed_weights
            invalidate_packed_weights(self.net)
            self.reducer.restore(hidden)
            self.reducer.zero(flat_cleared=own)  # grads are views of the flat buffer: one memset (or none) instead of zero_grad()
        return loss.detach(), costh.detach()
