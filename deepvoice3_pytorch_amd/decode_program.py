# coding: utf-8
"""Per-step launch program of an autoregressive decoder on the fused step kernels (csrc/decode_step.hip).

A decoder step of the reference (deepvoice3.py:397-461, nyanko.py:283-321) is ~100 tiny module calls; here it is a
flat list of descriptors built ONCE per utterance batch -- one dv3_conv_step_f32 per conv / projection layer (ring
buffer on a device step counter, k-tap GEMV, the whole layer tail) and one dv3_attn_step_f32 per attention read --
that the host replays launch by launch (optionally as one hipGraph per step), or that ONE persistent launch walks for
the whole utterance (dv3_decode_program_run: the loop, the ring buffers, the stop rule and the layer-to-layer
hand-over all stay on the device; opt-in, see StepProgram.decode).  `StepProgram` owns the buffers the descriptors point at.
"""
import ctypes
import os

import torch

from . import ops
from ._lib import STRUCTS


class StepProgram(object):
    def __init__(self, B, device):
        self.B, self.dev = B, device
        self.f32 = dict(dtype=torch.float32, device=device)
        self.t_dev = torch.zeros(1, dtype=torch.int32, device=device)      # the step counter every launch reads
        self.keep, self.prog = [self.t_dev], []

    def buffer(self, *shape):
        t = torch.zeros(*shape, **self.f32)
        self.keep.append(t)
        return t

    def conv_step(self, layer, x, mode, Cout, k=1, dil=1, gated=False, residual=False, spk=None, r=None, r2=None,
                  post_add=None, y=None, y_act=None, y_pre=None, out_seq=None):
        """one incremental conv layer (conv.py:17-46) with its tail; x (B, Cin) view (row stride free) -> y (B, Cout)"""
        B = self.B
        pk = layer.packed(glu_cg=Cout if gated else 0)
        if y is None:
            y = torch.empty(B, Cout, **self.f32)
        Cin = x.size(1)
        d = STRUCTS["dv3_conv_step_desc"]()
        d.x, d.x_bs = x.data_ptr(), x.stride(0)
        if k > 1:
            L = (k - 1) * dil + 1
            ring = self.buffer(L, B, Cin)
            d.ring, d.L = ring.data_ptr(), L
        d.t = self.t_dev.data_ptr()
        tiles = getattr(pk, "step_tiles", None)         # the weights in step-tile order, built once per packed image
        if tiles is None:
            Cg, M = (Cout if gated else 0), (2 * Cout if gated else Cout)
            n = ops._lib.lib().dv3_conv_step_pack_floats(k * Cin, M, Cg)
            tiles = torch.empty(n, **self.f32)
            ops._lib.call("dv3_conv_step_pack_f32", pk.fwd.data_ptr(), pk.lda, pk.a_half, k * Cin, M, Cg,
                          tiles.data_ptr(), ops._stream())
            pk.step_tiles = tiles
        d.a, d.lda, d.a_half = tiles.data_ptr(), pk.lda, pk.a_half
        d.bias = layer.bias.data_ptr() if layer.bias is not None else None
        if spk is not None:
            d.spk, d.spk_bs = spk.data_ptr(), spk.stride(0)
        if r is not None:
            d.r, d.r_bs = r.data_ptr(), r.stride(0)
        if r2 is not None:
            d.r2, d.r2_bs = r2.data_ptr(), r2.stride(0)
        if post_add is not None:
            d.post_add, d.post_add_ts, d.post_add_bs = post_add.data_ptr(), post_add.stride(0), post_add.stride(1)
        d.y, d.y_bs = y.data_ptr(), y.stride(0)
        if y_act is not None:
            d.y_act, d.y_act_bs = y_act.data_ptr(), y_act.stride(0)
        if y_pre is not None:
            d.y_pre, d.y_pre_bs = y_pre.data_ptr(), y_pre.stride(0)
        if out_seq is not None:
            d.out_seq, d.out_seq_ts, d.out_seq_bs = out_seq.data_ptr(), out_seq.stride(0), out_seq.stride(1)
        d.B, d.Cin, d.M = B, Cin, (2 * Cout if gated else Cout)
        d.Cg, d.J, d.dil, d.mode, d.residual = (Cout if gated else 0), k, dil, mode, int(residual)
        self.keep.extend([pk, x, y, spk, r, r2, post_add, y_act, y_pre, out_seq])
        self.prog.append(("dv3_conv_step_f32", d))
        return y

    def attn_step(self, q, k, v, window_backward, window_ahead, monotonic, attn_seq=None):
        """one attention read over (B, E, Tk) keys / values (deepvoice3.py:143-171 at Tq = 1, no padding mask)"""
        B = self.B
        E, Tk = k.size(1), k.size(2)
        ctx = torch.empty(B, E, **self.f32)
        la = self.buffer(2).to(torch.int32) if monotonic else None
        if la is not None:
            self.keep.append(la)
        # one-frame reads want a key / value ROW contiguous: (B, Tk, E), transposed once per utterance batch
        kt, vt = ops.transpose(k.contiguous()), ops.transpose(v.contiguous())
        self.keep.extend([kt, vt])
        a = STRUCTS["dv3_attn_step_desc"]()
        a.q, a.q_bs, a.k, a.v, a.kv_tke = q.data_ptr(), q.stride(0), kt.data_ptr(), vt.data_ptr(), 1
        a.last_attended = la.data_ptr() if la is not None else None
        a.win_back, a.win_ahead, a.t = window_backward, window_ahead, self.t_dev.data_ptr()
        a.ctx, a.ctx_bs = ctx.data_ptr(), ctx.stride(0)
        if attn_seq is not None:
            a.attn_seq, a.attn_seq_ts = attn_seq.data_ptr(), attn_seq.stride(0)
        a.B, a.E, a.Tk = B, E, Tk
        self.keep.extend([q, k, v, ctx, attn_seq])
        self.prog.append(("dv3_attn_step_f32", a))
        return ctx

    def run_step(self):
        s = ops._stream()
        for name, d in self.prog:
            ops._lib.call(name, ctypes.byref(d), s)
        self.t_dev.add_(1)

    def _entries(self, cur_in, test_inputs):
        """the program as a host array of dv3_decode_entry (teacher forcing: the entries that read the decoder input
        buffer read frame t of test_inputs instead, deepvoice3.py:411-415)"""
        B = self.B
        Entry = STRUCTS["dv3_decode_entry"]
        arr = (Entry * len(self.prog))()
        ti = None
        if test_inputs is not None:
            ti = test_inputs.to(torch.float32).reshape(B, test_inputs.size(1), -1).contiguous()
            if ti.size(2) != cur_in.size(1):
                raise RuntimeError("decode program: test_inputs frames carry %d values, the decoder input %d" % (
                    ti.size(2), cur_in.size(1)))
            self.keep.append(ti)
        fed = 0
        for i, (name, d) in enumerate(self.prog):
            if name == "dv3_conv_step_f32":
                arr[i].kind = 0
                arr[i].conv = d
                if ti is not None and d.x == cur_in.data_ptr():
                    arr[i].conv.x, arr[i].conv.x_bs, arr[i].conv.x_ts = ti.data_ptr(), ti.stride(0), ti.stride(1)
                    fed += 1
            else:
                arr[i].kind = 1
                arr[i].attn = d
        if ti is not None and fed == 0:
            raise RuntimeError("decode program: no entry reads the decoder input buffer")
        return arr, ti

    def decode_persistent(self, cur_in, test_inputs, dones_seq, min_steps, max_steps):
        """the whole loop as one launch of the persistent program kernel (include/dv3hip.h: dv3_decode_program_run)
        -> number of steps taken"""
        B, dev = self.B, self.dev
        free_running = test_inputs is None
        Prog = STRUCTS["dv3_decode_program"]
        arr, ti = self._entries(cur_in, test_inputs)
        host = bytearray(bytes(arr))
        entries = torch.frombuffer(host, dtype=torch.uint8).to(dev)
        n_sync = ops._lib.lib().dv3_decode_program_sync_ints(B)
        sync = torch.empty(n_sync, dtype=torch.int32, device=dev)
        steps_out = torch.zeros(1, dtype=torch.int32, device=dev)
        p = Prog()
        p.entries = entries.data_ptr()
        p.entries_host = ctypes.addressof(arr)
        p.n_entries, p.B = len(self.prog), B
        p.t0 = 0
        p.n_steps = ti.size(1) if ti is not None else max_steps + 1
        if free_running:
            p.done_seq, p.done_ts = dones_seq.data_ptr(), dones_seq.stride(0)
        p.min_steps, p.max_steps = min_steps, max_steps
        p.reserved = int(os.environ.get("DV3_DECODE_ABLATE", "0"))      # developer knob: timing ablations of the barriers
        p.wg_per_group = int(os.environ.get("DV3_DECODE_WG", "0"))
        p.sync, p.steps_out = sync.data_ptr(), steps_out.data_ptr()
        ops._lib.call("dv3_decode_program_run", ctypes.byref(p), ops._stream())
        t = int(steps_out.item())
        if t < 0:
            raise RuntimeError("decode program: a device barrier timed out (the persistent grid was not co-resident?)")
        self.t_dev.fill_(t)
        return t

    def decode_launched(self, cur_in, test_inputs, dones_seq, min_steps, max_steps, chunk=8):
        """the loop with the launches issued by the library (dv3_decode_program_launch: one call per chunk of steps, the
        step index in the descriptors).  Free running, the done flags are read once per chunk and the steps after the
        stopping one are dropped -- later steps never change earlier outputs.  -> number of steps taken"""
        Prog = STRUCTS["dv3_decode_program"]
        arr, ti = self._entries(cur_in, test_inputs)
        p = Prog()
        p.entries_host = ctypes.addressof(arr)
        p.n_entries, p.B = len(self.prog), self.B
        stream = ops._stream()
        if ti is not None:
            p.t0, p.n_steps = 0, ti.size(1)
            if p.n_steps > 0:
                ops._lib.call("dv3_decode_program_launch", ctypes.byref(p), stream)
            return int(ti.size(1))
        limit, t = max_steps + 1, 0
        while True:
            n = min(min_steps + 1 if t == 0 else chunk, limit - t)
            p.t0, p.n_steps = t, n
            ops._lib.call("dv3_decode_program_launch", ctypes.byref(p), stream)
            done = (dones_seq[t:t + n].reshape(n, -1) > 0.5).all(dim=1).tolist()
            for k in range(n):
                if t + k + 1 > min_steps and done[k]:
                    return t + k + 1
            t += n
            if t >= limit:
                return t

    def decode(self, cur_in, test_inputs, dones_seq, min_steps, max_steps, use_graph, persistent=None, launched=None):
        """the decoder loop (deepvoice3.py:397-473 / nyanko.py:277-331): teacher-forced over test_inputs (B, n, D), or
        free running until every item's done flag passed 0.5 after min_steps, at most max_steps + 1 steps.
        -> number of steps taken.  Default (launched): one launch per program entry per step, issued by the library in
        chunks of steps (decode_launched: ~3 us of host time per launch).  launched=False: the same launches from
        Python + ctypes, optionally replayed as a per-step hipGraph (~100 us of host time per step either way: about
        what the GPU needs for the step, so the host is on the critical path).  persistent=True (or
        DV3_DECODE_PERSISTENT=1): ONE launch for the whole loop -- bit-identical, and the host drops out entirely,
        but on MI355X the device-wide barrier between layers (agent-scope release + acquire: L2 write-back /
        invalidate across the 8 XCDs, ~3.5 us) costs more than a kernel boundary does (scripts/decode_time.py), so
        it is opt-in."""
        if persistent is None:
            persistent = os.environ.get("DV3_DECODE_PERSISTENT", "0") == "1"
        if persistent:
            return self.decode_persistent(cur_in, test_inputs, dones_seq, min_steps, max_steps)
        if launched is None:
            launched = os.environ.get("DV3_DECODE_LAUNCHED", "1") != "0"
        if launched:
            return self.decode_launched(cur_in, test_inputs, dones_seq, min_steps, max_steps)
        free_running = test_inputs is None
        B = self.B
        graphed = bool(use_graph) and free_running
        graph, t = None, 0
        while True:
            if not free_running:
                if t >= test_inputs.size(1):
                    break
                cur_in.copy_(test_inputs[:, t, :].reshape(B, -1))
            if graphed and t >= 1:
                if graph is None:
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        self.run_step()
                graph.replay()
            else:
                self.run_step()
            t += 1
            if free_running:
                if t > min_steps and bool((dones_seq[t - 1] > 0.5).all()):
                    break
                elif t > max_steps:
                    break
        return t


class StepTrace(object):
    """Per-step results of the module-by-module decode loops (the configurations the step program does not take):
    each step's (B, 1, .) tensors are copied into stacked (B, capacity, .) buffers that double when full -- the host-side
    shape of what the step program writes on the device -- so the loops end with three slices instead of three lists of
    per-step tensors to squeeze, stack and transpose.  `stop` is the reference's rule (deepvoice3.py:469-473)."""

    def __init__(self, min_steps, max_steps, teacher_forced):
        self.min_steps, self.max_steps, self.teacher_forced = min_steps, max_steps, teacher_forced
        self.n = 0
        self.dones = []
        self._bufs = None

    def _room(self, parts):
        if self._bufs is None:
            cap = 64
            self._bufs = [p.new_empty((p.size(0), cap) + tuple(p.shape[2:])) for p in parts]
        elif self.n == self._bufs[0].size(1):
            self._bufs = [torch.cat((b, torch.empty_like(b)), dim=1) for b in self._bufs]

    def push(self, output, alignment, state, done):
        parts = (output, alignment, state)
        self._room(parts)
        for b, p in zip(self._bufs, parts):
            b[:, self.n:self.n + 1].copy_(p)
        self.dones.append(done)
        self.n += 1

    @property
    def last_output(self):
        return self._bufs[0][:, self.n - 1:self.n].contiguous()

    def stop(self, done):
        """after push: the free-running loop ends once every item signalled done past min_steps, or past max_steps"""
        if self.teacher_forced:
            return False
        return bool((done > 0.5).all() and self.n > self.min_steps) or self.n > self.max_steps

    def result(self):
        out, ali, st = (b[:, :self.n] for b in self._bufs)
        return out.contiguous(), ali, self.dones, st.contiguous()
