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
    """bucket_mb: size of the buckets cut from the arena's tail (the converter / decoder gradients, final early in
    backward).  The gradients that become final LAST -- the head of the arena: the encoder, 54-63 % of the parameters
     -- sit in smaller buckets of `last_bucket_mb` over the first `last_span_mb` of the arena:
    clip + Adam need the whole reduced arena, so the all-reduce of the bucket that closes last is never hidden behind
    backward and its size is the exposed time (one encoder layer is 6.3 MB; an 8 MB ring all-reduce over xGMI is
    still bandwidth- rather than latency-bound).  last_bucket_mb=None: one size everywhere.
    isolate: arena indices of parameters that get a bucket of their own -- tables every layer contributes to (the speaker
    embedding: registered last, final only when the ENCODER's backward is done; in a shared bucket it held the 25 MB of
    converter gradients back until the end of backward: scripts/r4_group_replay_check.py).
    boundaries: arena indices at which a bucket must START (a group of parameters that become final together and late --
    the speaker projections of the fused block path -- is kept out of its neighbours' buckets)."""

    def __init__(self, arena, process_group=None, bucket_mb=25.0, last_bucket_mb=8.0, last_span_mb=40.0, isolate=(),
                 beside=(), boundaries=(), issue_stream=None):
        """issue_stream (round 5): the stream the all-reduces are ISSUED from -- the step's second backward stream
        (ops.SideStream), on which the weight-norm backward has just written the bucket's gradients.  The calls are
        asynchronous (c10d runs the collective on its own communicator stream, which waits for the issuing stream's
        position and nothing else) and the step stream picks their completion up in join(): no collective stream of our
        own, so the step runs on three streams (step, weight-gradient, c10d's) instead of four -- HIP maps streams onto
        GPU_MAX_HW_QUEUES = 4 hardware queues, and with step + weight-gradient + collective + prefetch + c10d's there
        were more streams than queues: whichever two shared one serialised, cross-stream waits and all (BENCH_r04:
        deepvoice3_vctk replay +16 % under a world-1 group on the driver's box, +1.4 % on another).  None: a collective
        stream of this object's own, picked to run beside `beside` (rounds 2-4)."""
        self.arena = arena
        self.standin = process_group if getattr(process_group, "is_standin", False) else None
        self.pg = None if self.standin is not None else process_group
        if self.standin is not None and self.standin.stream is None and arena.grad.is_cuda:
            self.standin.bind(list(beside) + [issue_stream])
        # the collective stream: one whose work really overlaps with the streams backward runs on (`beside`; see
        # ops.concurrent_stream -- HIP streams share a few hardware queues)
        self.side = None
        self.async_issue = False
        self._works = []
        self._sync_issued = False
        if arena.grad.is_cuda:
            from . import ops as _ops
            if issue_stream is not None:
                self.side, self.async_issue = issue_stream, True
            else:
                self.side = _ops.concurrent_stream(list(beside), role="collective") if beside else torch.cuda.Stream()
        cap = max(1, int(bucket_mb * (1 << 20) / 4))
        cap_last = cap if last_bucket_mb is None else max(1, min(cap, int(last_bucket_mb * (1 << 20) / 4)))
        span_last = 0 if last_bucket_mb is None else int(last_span_mb * (1 << 20) / 4)
        # cut buckets from the tail of the arena
        self.buckets = []   # (lo, hi, [param indices])
        hi = arena.total
        lo = hi
        cur = []
        isolate, boundaries = set(isolate), set(boundaries)
        for i in range(len(arena.params) - 1, -1, -1):
            o = arena.offsets[i]
            c = cap_last if hi <= span_last else cap
            if cur and (hi - o > c or i in isolate or cur[-1] in isolate or (i + 1) in boundaries):
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
        self._completed = []
        # measurement (bench.py): when a list, finish() appends an (event, event) pair bracketing the point where
        # the step stream joins the collective stream -- their distance is the all-reduce time NOT hidden behind
        # backward.  Timing events can not be recorded while a hipGraph is being captured.
        self.exposed_events = None
        # A parameter reports its gradient final either through autograd (AccumulateGrad hook) or, for the
        # conv-layer parameters whose gradient is accumulated in place, through ops.grad_ready_hooks -- ONE call per
        # layer from the layer's own backward.  Autograd fires its hook for the in-place parameters as well (measured
        # on torch 2.10, although the Function returns no gradient for them): 130-190 Python callbacks per step that
        # report nothing new.  After the first armed backward the autograd hooks of the parameters that reported in
        # place are therefore removed (`prune_hooks`); what stays is one call per conv layer + one autograd hook per
        # remaining parameter (embedding tables, speaker embedding: 3-6 per model).
        self._index_of = {id(p): i for i, p in enumerate(arena.params)}
        self._handles = {i: p.register_post_accumulate_grad_hook(self._make_hook(i))
                         for i, p in enumerate(arena.params)}
        self._inplace_seen = set()
        self._pruned = False
        from . import ops as _ops
        self._ops_hook = self._on_inplace_grad
        _ops.grad_ready_hooks.append(self._ops_hook)

    def close(self):
        """Detach from the parameters and from ops.grad_ready_hooks (a second Trainer in the same process
        must not leave this one's hooks -- and through them its arena -- alive)."""
        from . import ops as _ops
        if self._ops_hook in _ops.grad_ready_hooks:
            _ops.grad_ready_hooks.remove(self._ops_hook)
        for h in self._handles.values():
            h.remove()
        self._handles = {}
        self._armed = False

    def _make_hook(self, i):
        def hook(param):
            # idempotent per step: a conv-layer parameter reports through ops.grad_ready_hooks when its
            # in-place gradient is final, and autograd's AccumulateGrad hook may fire for it as well
            self._report(i)
        return hook

    def _report(self, i):
        if not self._armed or self.notified[i]:
            return
        self.notified[i] = True
        b = self.bucket_of[i]
        self.pending[b] -= 1
        if self.pending[b] == 0:
            from . import ops as _ops
            if _ops.SideStream.split_capture:
                # the step is being captured as segment graphs (train_step.GraphedTrainer): nothing of the process
                # group goes into a capture -- the bucket is remembered with the segment that completes it and the
                # replay issues its all-reduce from the host, between two segment launches
                self._completed.append(b)
                return
            self._launch(b)

    def _on_inplace_grad(self, *params):
        """one call per conv layer (ops.ConvLayerFn.backward) with the layer's parameters (v, g, bias; None skipped)"""
        for p in params:
            if p is None:
                continue
            i = self._index_of.get(id(p))
            if i is not None:
                if not self._pruned:
                    self._inplace_seen.add(i)
                self._report(i)

    def prune_hooks(self):
        """drop the autograd hooks of the parameters that report in place (see __init__); called by finish() after
        the first armed backward.  -> number of autograd hooks left"""
        if not self._pruned:
            for i in self._inplace_seen:
                h = self._handles.pop(i, None)
                if h is not None:
                    h.remove()
            self._pruned = True
        return len(self._handles)

    def arm(self):
        """Call right before backward."""
        for b, (_, _, plist) in enumerate(self.buckets):
            self.pending[b] = len(plist)
            self.launched[b] = False
        self.notified = [False] * len(self.arena.params)
        self._completed = []
        self._armed = True

    def take_completed(self):
        """the buckets that became complete since the last call (segment capture only; see _report)"""
        done, self._completed = self._completed, []
        return done

    def disarm(self):
        self._armed = False

    def launch_after(self, bucket_ids, streams):
        """all-reduce `bucket_ids` on the collective stream once everything enqueued on `streams` so far has run
        (the replay of a segmented step: called between two segment launches, never inside a capture)"""
        if self.side is None:   # CPU / gloo (tests): synchronous
            for b in bucket_ids:
                lo, hi, _ = self.buckets[b]
                dist.all_reduce(self.arena.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.pg)
            return
        self._wait_for(streams)
        with torch.cuda.stream(self.side):
            for b in bucket_ids:
                lo, hi, _ = self.buckets[b]
                self._all_reduce(self.arena.grad[lo:hi])

    def _wait_for(self, streams):
        """the issuing stream waits for everything enqueued so far on `streams` (itself excepted)"""
        for st in streams:
            if st is None or st == self.side:
                continue
            ev = torch.cuda.Event()
            ev.record(st)
            self.side.wait_event(ev)

    def _all_reduce(self, view):
        """current stream = the issuing stream.  async_issue: the call returns a handle and the issuing stream is NOT made
        to wait for the collective (its next kernels -- the following layers' weight gradients -- run beside it); join()
        makes the step stream wait.  Inside a capture (single-graph replay) the call stays synchronous."""
        if self.standin is not None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("RingStandin: the stand-in collectives are issued from the host (segmented replay or eager)")
            self._works.append(self.standin.all_reduce(view))
            if not self.async_issue:
                self._sync_issued = True
        elif self.async_issue and not torch.cuda.is_current_stream_capturing():
            self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        else:
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg)
            self._sync_issued = True      # no handle: join() has to make the step stream wait for the issuing stream

    def join(self, cur=None):
        """the step stream waits for the collectives (timed when exposed_events is a list)"""
        if self.side is None:
            return
        cur = cur or torch.cuda.current_stream()
        timed = self.exposed_events is not None and not torch.cuda.is_current_stream_capturing()
        if timed:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record(cur)
        works, self._works = self._works, []
        if works:
            with torch.cuda.stream(cur):
                for w in works:
                    w.wait()                 # the CURRENT stream waits for the communicator stream's end-of-collective event
        #  a collective issued synchronously on the issuing stream -- the non-async mode, and EVERY collective
        # issued inside a capture, e.g. the buckets finish() launches for parameters that got no gradient, after
        # SideStream.join() has already re-joined the side stream -- left no handle: without this wait a single-graph
        # capture would end with the issuing stream forked and unjoined, or clip + Adam would read un-reduced gradients
        if not self.async_issue or self._sync_issued or torch.cuda.is_current_stream_capturing():
            cur.wait_stream(self.side)
        self._sync_issued = False
        if timed:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(cur)
            self.exposed_events.append((e0, e1))

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
        self._wait_for(streams)
        with torch.cuda.stream(self.side):
            self._all_reduce(view)

    def finish(self):
        """Launch whatever is left (parameters that received no gradient) and join the side stream."""
        was_armed, self._armed = self._armed, False
        if was_armed and not self._pruned and self._inplace_seen:
            self.prune_hooks()
        for b in range(len(self.buckets)):
            if not self.launched[b]:
                self._launch(b)
        if self.side is not None:
            self.join()

    def exposed_ms(self):
        """mean GPU time per step the step stream spent waiting for the collective stream (call after a
        synchronize); clears the record"""
        ev, self.exposed_events = self.exposed_events or [], []
        return sum(a.elapsed_time(b) for a, b in ev) / len(ev) if ev else None


class RingStandin(object):
    """An n-rank RCCL ring all-reduce as ONE GPU sees it -- a MEASUREMENT stand-in, not a collective (csrc/standin.hip):
    `channels` persistent workgroups of `threads` threads that stream 2 (n - 1) / n x S bytes through HBM at the pace the
    links would set (2 (n - 1) / n x S / busbw), on a communicator stream of its own that waits for the issuing stream's
    position -- c10d's stream discipline.  Pass it to `train_step.Trainer(process_group=RingStandin(...))`: the bucket
    schedule, the notifications and the issue points are the data-parallel step's; the gradients stay the single-GPU
    step's (nothing is summed; world = 1 in the update).  What it measures: how much a step stretches when a third tenant
    (after the input-gradient and weight-gradient queues) holds CUs and memory bandwidth while backward runs, and how
    long the step stream waits for the last buckets (`allreduce_exposed_ms`).

    busbw_gbps: the all-reduce "bus bandwidth" (rccl-tests' busbw = S / t x 2 (n - 1) / n) assumed for the node's xGMI
    ring(s); MI355X: 7 links x ~153 GB/s per GPU, a ring is bound by one link per direction, RCCL runs several rings."""
    is_standin = True

    def __init__(self, world=8, channels=16, threads=256, busbw_gbps=150.0, device=None, avoid=()):
        from . import ops as _ops
        self.world, self.channels, self.threads, self.busbw_gbps = int(world), int(channels), int(threads), float(busbw_gbps)
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.stream = None
        if avoid:
            self.bind(avoid)
        self.scratch = torch.empty(32 << 20, dtype=torch.uint8, device=self.device)
        self.launches, self.bytes = 0, 0

    def bind(self, avoid):
        """pick the communicator stream: one whose kernels really run BESIDE the streams backward runs on (HIP maps its
        streams onto a few hardware queues, ops.concurrent_stream) -- a stand-in that shares a queue with the step stream
        would measure that serialisation, not a ring's footprint (first sweep of round 6: step inflation 1.03 .. 1.29 for
        one configuration, depending on which pool stream torch handed out)"""
        from . import ops as _ops
        with torch.cuda.device(self.device):
            uniq = []
            for st in avoid:
                if st is not None and all(st != u for u in uniq):
                    uniq.append(st)
            self.stream = _ops.concurrent_stream(uniq, role="communicator stand-in")

    class _Work(object):
        def __init__(self, ev):
            self.ev = ev

        def wait(self):                      # c10d Work.wait(): the CURRENT stream waits for the collective's end
            torch.cuda.current_stream().wait_event(self.ev)

    def all_reduce(self, view):
        """current stream = the issuing stream: the communicator stream waits for its position, runs the stand-in, and the
        returned handle's wait() makes a stream wait for its end"""
        from . import _lib
        if self.stream is None:
            self.bind([torch.cuda.current_stream()])
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.stream.wait_event(ev)
        n = view.numel() * view.element_size()
        moved = int(2 * (self.world - 1) * n // self.world)
        _lib.call("dv3_ring_standin", view.data_ptr(), n, self.scratch.data_ptr(), self.scratch.numel(), moved,
                  self.channels, self.threads, self.busbw_gbps * 1e3, self.stream.cuda_stream)
        done = torch.cuda.Event()
        done.record(self.stream)
        self.launches += 1
        self.bytes += moved
        return RingStandin._Work(done)

    def ideal_ms(self, n_bytes):
        """duration the links alone give an all-reduce of n_bytes"""
        return 2.0 * (self.world - 1) / self.world * n_bytes / (self.busbw_gbps * 1e9) * 1e3


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
