"""Data-parallel glue: one process per GPU, minibatches sharded across ranks, ONE collective per
step -- an all-reduce (average) of the flat fp32 gradient buffer over RCCL/xGMI.

The reference has no multi-device path at all (selene/__init__.py:98-100 leaves data_parallel
commented out); semantics here are those `torch DDP` would give the reference module: per-shard
BatchNorm statistics, averaged gradients, rank-0 parameters broadcast at start.
"""
import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def shard_bounds(n_items, world_size, r):
    """Contiguous shard [lo, hi) of rank r; the first n_items % world_size ranks get one more."""
    base, extra = divmod(n_items, world_size)
    lo = r * base + min(r, extra)
    return lo, lo + base + (1 if r < extra else 0)


def shard_batch(x, y):
    lo, hi = shard_bounds(x.shape[0], world(), rank())
    return x[lo:hi], y[lo:hi]


def broadcast_parameters(model, src=0):
    """Make every rank start from rank `src`'s parameters and BatchNorm buffers."""
    if world() == 1:
        return
    tensors = list(model.parameters()) + list(model.buffers())
    for t in tensors:
        dist.broadcast(t.data, src)
    # the broadcast wrote through `.data`, which torch's version counters do not see: tell the
    # model (cached eval tables, explainn_params.version) that the values moved
    torch.autograd.graph.increment_version(tensors)
    if hasattr(model, "invalidate"):
        model.invalidate()


def average_gradients(flat):
    """All-reduce (average) of a flat gradient buffer: the `grad_sync` callable of the autograd
    path (ExplaiNN._launch_backward calls it with the buffer all 14 gradients are views of)."""
    if world() > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(world())
    return flat


class GradAllReduce:
    """Averages one flat gradient buffer across ranks: a single collective per step.  At C2 the
    buffer is 3.7 MB; RCCL picks its own algorithm."""

    def __init__(self, flat, split=0, force=False):
        """split: number of leading elements (conv_w, conv_b, bn1_w, bn1_b in explainn_grads order)
        that only the last part of the backward produces; everything behind it is final earlier and
        can be reduced while that part still runs (`start_tail` / `finish`, used by StepEngine)."""
        self.flat = flat
        self.split = int(split)
        self.n = world()
        # force: issue the collectives even in a one-rank group (they are identities there); lets a
        # single-GPU test drive the asynchronous RCCL path
        self.single = self.n == 1 and not force
        self.native_avg = False
        if not self.single and dist.get_backend() == "nccl":
            # ReduceOp.AVG saves the scaling pass; probe it once (all ranks take the same branch)
            try:
                probe = torch.ones(1, device=flat.device)
                dist.all_reduce(probe, op=dist.ReduceOp.AVG)
                self.native_avg = abs(float(probe.item()) - 1.0) < 1e-6
            except Exception:
                self.native_avg = False

    def _reduce(self, t, async_op=False):
        op = dist.ReduceOp.AVG if self.native_avg else dist.ReduceOp.SUM
        return dist.all_reduce(t, op=op, async_op=async_op)

    def start_tail(self):
        """Enqueue the all-reduce of flat[split:] (the per-unit FC / head gradients) without making
        the launch stream wait for it; returns the work handle for `finish`."""
        if self.single or self.split <= 0 or self.split >= self.flat.numel():
            return None
        return self._reduce(self.flat[self.split:], async_op=True)

    def finish(self, work):
        """All-reduce what `start_tail` left (everything if it returned None), then make the launch
        stream wait for both."""
        if self.single:
            return self.flat
        if work is None:
            return self()
        # the head slice (conv_w, conv_b, bn1_w, bn1_b) goes to RCCL's stream as well: enqueued there
        # behind an event of the launch stream, so the tail reduce still in flight and this one
        # queue up on RCCL's side while the launch stream only waits once, for both
        head = self._reduce(self.flat[:self.split], async_op=True)
        work.wait()
        head.wait()
        if not self.native_avg:
            self.flat.div_(self.n)
        return self.flat

    def __call__(self, flat=None):
        t = self.flat if flat is None else flat
        if self.single:
            return t
        if self.native_avg:
            dist.all_reduce(t, op=dist.ReduceOp.AVG)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            t.div_(self.n)
        return t
