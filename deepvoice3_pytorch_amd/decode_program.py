# coding: utf-8
"""Per-step launch program of an autoregressive decoder on the fused step kernels (csrc/decode_step.hip).

A decoder step of the reference (deepvoice3.py:397-461, nyanko.py:283-321) is ~100 tiny module calls; here it is a
flat list of descriptors built ONCE per utterance batch -- one dv3_conv_step_f32 per conv / projection layer (ring
buffer on a device step counter, k-tap GEMV, the whole layer tail) and one dv3_attn_step_f32 per attention read --
that the host replays per step (optionally as one hipGraph).  `StepProgram` owns the buffers the descriptors point at.
"""
import ctypes

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
        d.a, d.lda, d.a_half = pk.fwd.data_ptr(), pk.lda, pk.a_half
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
        ctx = torch.empty(B, k.size(1), **self.f32)
        la = self.buffer(2).to(torch.int32) if monotonic else None
        if la is not None:
            self.keep.append(la)
        a = STRUCTS["dv3_attn_step_desc"]()
        a.q, a.q_bs, a.k, a.v = q.data_ptr(), q.stride(0), k.data_ptr(), v.data_ptr()
        a.last_attended = la.data_ptr() if la is not None else None
        a.win_back, a.win_ahead, a.t = window_backward, window_ahead, self.t_dev.data_ptr()
        a.ctx, a.ctx_bs = ctx.data_ptr(), ctx.stride(0)
        if attn_seq is not None:
            a.attn_seq, a.attn_seq_ts = attn_seq.data_ptr(), attn_seq.stride(0)
        a.B, a.E, a.Tk = B, k.size(1), k.size(2)
        self.keep.extend([q, k, v, ctx, attn_seq])
        self.prog.append(("dv3_attn_step_f32", a))
        return ctx

    def run_step(self):
        s = ops._stream()
        for name, d in self.prog:
            ops._lib.call(name, ctypes.byref(d), s)
        self.t_dev.add_(1)

    def decode(self, cur_in, test_inputs, dones_seq, min_steps, max_steps, use_graph):
        """the decoder loop (deepvoice3.py:397-473 / nyanko.py:277-331): teacher-forced over test_inputs (B, n, D), or
        free running until every item's done flag passed 0.5 after min_steps, at most max_steps + 1 steps.
        -> number of steps taken"""
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
