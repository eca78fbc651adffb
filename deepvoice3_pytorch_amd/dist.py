# coding: utf-8
"""Data-parallel gradient exchange: one process per GPU, RCCL (torch.distributed "nccl" backend on
ROCm) all-reduce over xGMI of the flat gradient arena, in buckets, on a side HIP stream that
overlaps the rest of backward.

The reference has no multi-GPU code at all; this is the new capability the
north-star asks for.  Samples are independent, so the only collective is the gradient sum.

Bucketing: the arena is laid out in parameter registration order (encoder, decoder, converter,
speaker table); backward produces gradients roughly in reverse, so buckets are cut from the tail.
xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce of S bytes moves
2*(N-1)/N*S per GPU over one link pair, ~25 MB buckets keep each collective well above the
latency floor while leaving >= 4 of them to pipeline behind the converter/decoder backward.
"""
import torch
import torch.distributed as dist


class BucketedAllReduce(object):
    def __init__(self, arena, process_group=None, bucket_mb=25.0):
        self.arena = arena
        self.pg = process_group
        self.side = torch.cuda.Stream() if arena.grad.is_cuda else None
        cap = max(1, int(bucket_mb * (1 << 20) / 4))
        # cut buckets from the tail of the arena
        self.buckets = []   # (lo, hi, [param indices])
        hi = arena.total
        lo = hi
        cur = []
        for i in range(len(arena.params) - 1, -1, -1):
            o = arena.offsets[i]
            if hi - o > cap and cur:
                self.buckets.append((lo, hi, cur))
                hi, cur = lo, []
            lo = o
            cur.append(i)
        if cur:
            self.buckets.append((lo, hi, cur))
        self.bucket_of = {}
        for b, (_, _, plist) in enumerate(self.buckets):
            for i in plist:
                self.bucket_of[i] = b
        self.notified = [False] * len(arena.params)
        self.pending = [0] * len(self.buckets)
        self.launched = [False] * len(self.buckets)
        self._armed = False
        # measurement (bench.py): when a list, finish() appends an (event, event) pair bracketing the point where
        # the step stream joins the collective stream -- their distance is the all-reduce time NOT hidden behind
        # backward.  Timing events can not be recorded while a hipGraph is being captured.
        self.exposed_events = None
        # a parameter reports its gradient final either through autograd (AccumulateGrad hook) or, for
        # the conv-layer parameters whose gradient is accumulated in place, through ops.grad_ready_hooks
        self._index_of = {id(p): i for i, p in enumerate(arena.params)}
        self._handles = [p.register_post_accumulate_grad_hook(self._make_hook(i))
                         for i, p in enumerate(arena.params)]
        from . import ops as _ops
        self._ops_hook = self._on_inplace_grad
        _ops.grad_ready_hooks.append(self._ops_hook)

    def close(self):
        """Detach from the parameters and from ops.grad_ready_hooks (a second Trainer in the same process
        must not leave this one's hooks -- and through them its arena -- alive)."""
        from . import ops as _ops
        if self._ops_hook in _ops.grad_ready_hooks:
            _ops.grad_ready_hooks.remove(self._ops_hook)
        for h in self._handles:
            h.remove()
        self._handles = []
        self._armed = False

    def _make_hook(self, i):
        def hook(param):
            # idempotent per step: a conv-layer parameter reports through ops.grad_ready_hooks when its
            # in-place gradient is final, and autograd's AccumulateGrad hook may fire for it as well
            # (measured on torch 2.10: it does, even though the Function returns no gradient for it)
            if not self._armed or self.notified[i]:
                return
            self.notified[i] = True
            b = self.bucket_of[i]
            self.pending[b] -= 1
            if self.pending[b] == 0:
                self._launch(b)
        return hook

    def _on_inplace_grad(self, param):
        i = self._index_of.get(id(param))
        if i is not None:
            self._make_hook(i)(param)

    def arm(self):
        """Call right before backward."""
        for b, (_, _, plist) in enumerate(self.buckets):
            self.pending[b] = len(plist)
            self.launched[b] = False
        self.notified = [False] * len(self.arena.params)
        self._armed = True

    def _launch(self, b):
        lo, hi, _ = self.buckets[b]
        view = self.arena.grad[lo:hi]
        self.launched[b] = True
        if self.side is None:   # CPU / gloo (tests): synchronous
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg)
            return
        # the bucket's gradients were written on the step stream (autograd) and, for the conv layers, on the
        # weight-gradient stream (ops.SideStream): the collective waits for both
        from . import ops as _ops
        streams = {torch.cuda.current_stream()}
        for st in (_ops.SideStream.stream, _ops.SideStream.main):
            if st is not None:
                streams.add(st)
        for st in streams:
            ev = torch.cuda.Event()
            ev.record(st)
            self.side.wait_event(ev)
        with torch.cuda.stream(self.side):
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg)

    def finish(self):
        """Launch whatever is left (parameters that received no gradient) and join the side stream."""
        self._armed = False
        for b in range(len(self.buckets)):
            if not self.launched[b]:
                self._launch(b)
        if self.side is not None:
            cur = torch.cuda.current_stream()
            timed = self.exposed_events is not None and not torch.cuda.is_current_stream_capturing()
            if timed:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record(cur)
            cur.wait_stream(self.side)
            if timed:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record(cur)
                self.exposed_events.append((e0, e1))

    def exposed_ms(self):
        """mean GPU time per step the step stream spent waiting for the collective stream (call after a
        synchronize); clears the record"""
        ev, self.exposed_events = self.exposed_events or [], []
        return sum(a.elapsed_time(b) for a, b in ev) / len(ev) if ev else None


def init_from_env(backend=None, allow_single=False):
    """torch.distributed init from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun).
    -> (process_group | None, rank, world, local_rank).  A single process gets no group unless `allow_single`
    (a world-size-1 group exercises the collective path -- RCCL on a GPU box -- with one device)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and not (allow_single and "MASTER_PORT" in os.environ):
        return None, 0, 1, 0
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return dist.group.WORLD, rank, world, local_rank
