# coding: utf-8
"""DeepVoice3 blocks (Encoder / AttentionLayer / Decoder / Converter) on HIP kernels.

API mirror of the reference's deepvoice3_pytorch/deepvoice3.py: same class names, constructor
arguments, module/parameter names (so reference checkpoints load), same forward signatures,
layouts of inputs and outputs, and error behaviour.  Internally every activation stays in BCT
(batch, channel, time) and each layer is one fused launch:
  Conv1d + ReLU pairs            -> one tap-GEMM with a ReLU epilogue
  Conv1dGLU                      -> one tap-GEMM (dropout, bias, speaker bias, gate, residual)
  attention                      -> 1x1 tap-GEMMs + score GEMM + fused mask/softmax/dropout
                                    + context GEMM + out-projection with both residual adds fused
"""
import math

import numpy as np
import torch
from torch import nn

from . import ops
from .modules import Conv1d, ConvTranspose1d, Embedding, Linear
from .modules import key_lengths_i32, get_mask_from_lengths, SinusoidalEncoding, Conv1dGLU
from . import conv as _conv
from .decode_program import StepTrace


def expand_speaker_embed(inputs_btc, speaker_embed=None, tdim=1):
    """(B, N) -> (B, T, N) view (deepvoice3.py:13-21)."""
    if speaker_embed is None:
        return None
    ss = speaker_embed.size()
    return speaker_embed.unsqueeze(1).expand(ss[0], inputs_btc.size(tdim), ss[-1])


def _drop_speaker_embed(speaker_embed, T, p, training, site):
    """F.dropout(expand_speaker_embed(...)) as the reference applies it (one mask shared by every
    consumer in the block).  Eval / p == 0: the stride-0 expanded view (constant over time)."""
    if speaker_embed is None:
        return None
    e = speaker_embed.unsqueeze(1).expand(speaker_embed.size(0), T, speaker_embed.size(-1))
    if not training or p <= 0:
        return e
    # (B, T, N) dropped with the HIP dropout kernel on the BCT image, returned as a BTC view
    bct = ops.dropout(speaker_embed.unsqueeze(-1).expand(-1, -1, T).contiguous(), p, True, site)
    return bct.transpose(1, 2)


def _fuse_speaker_biases(speaker_embed_btc, modules):
    """Training with dropout on the expanded speaker embedding (deepvoice3.py:78-81, 292-294): the per-frame biases
    softsign(speaker_proj(e)) (modules.py:158-162) of every Conv1dGLU in `modules` -- the layers that share this dropped
    embedding -- in ONE launch (ops.speaker_bias_block); Conv1dGLU.speaker_bias picks them up from the tensor."""
    if speaker_embed_btc is None or not ops.fused_speaker_bias or not speaker_embed_btc.is_cuda:
        return
    if speaker_embed_btc.stride(1) == 0 or speaker_embed_btc.size(2) > 16:     # constant over time: a (B, C) bias
        return
    glus = [m for m in modules if isinstance(m, Conv1dGLU) and m.speaker_proj is not None]
    if len(glus) < 2:
        return
    layers = []
    for m in glus:
        v, g = m.speaker_proj.wn_params()
        layers.append((v, g, m.speaker_proj.bias))
    outs = ops.speaker_bias_block(speaker_embed_btc.transpose(1, 2), layers)
    speaker_embed_btc._dv3_block_bias = {id(m): o for m, o in zip(glus, outs)}


def _c8_enter(x):
    """bf16 storage (ops.storage_c8: the bf16 GEMM mode): fp32 (B, C, T) -> channel-blocked bf16 at a stack entry"""
    if ops.storage_c8() and not ops.is_c8(x):
        return ops.to_c8(x)
    return x


def _c8_leave(x, C=None):
    return ops.from_c8(x, C) if ops.is_c8(x) else x


def _conv1d_c8(f, x, last, **kw):
    """a plain Conv1d inside a stack: its output follows the storage mode -- c8 between the layers, fp32 (B, C, T)
    for the last layer of the stack (what the callers consume)"""
    if not (ops.is_c8(x) or ops.storage_c8()):
        return f(x, **kw)
    return f(x, out_c8=not last, **kw)


def _run_stack(modules, x, speaker_embed_btc, first=0, keep_c8=False, valid_axis=None):
    """Run a ModuleList of {Conv1d, nn.ReLU, Conv1dGLU, ConvTranspose1d} on BCT x, fusing each
    Conv1d + ReLU pair into one launch.  In the bf16 GEMM mode the activations between the layers are
    channel-blocked bf16 tensors (ops.to_c8); the result is fp32 (B, C, T) unless keep_c8.
    valid_axis = (device int32[1], most surplus columns) for a NON-CAUSAL stack on a batch padded beyond its own
    maximum (ops.ValidLengths): every layer's output -- and, in backward, its gradient -- is zero beyond that maximum,
    as the zero padding of the reference's nn.Conv1d is."""
    n = len(modules)
    i = first
    C = x.size(1)               # channel count of x (a c8 tensor pads it to a multiple of 32)
    x = _c8_enter(x)
    if valid_axis is not None:
        x = ops.zero_tail(x, *valid_axis)
    while i < n:
        f = modules[i]
        if isinstance(f, Conv1dGLU):
            x = f(x, speaker_embed_btc)
            C = f.conv.out_channels // 2
            if i + 1 < n:
                ops.mark_sole_consumer(x)      # the next layer of the stack is x's only consumer (ops.GateFuse)
        elif isinstance(f, _conv.Conv1d):
            relu = i + 1 < n and isinstance(modules[i + 1], nn.ReLU)
            last = (i + (2 if relu else 1) >= n) and not keep_c8
            x = _conv1d_c8(f, x, last, mode=ops.EPI_RELU if relu else ops.EPI_LINEAR)
            C = f.out_channels
            i += int(relu)
        elif isinstance(f, nn.ReLU):
            x = torch.relu(x)
        elif ops.is_c8(x):          # ConvTranspose1d: fp32 layer between two conversions
            x = _c8_enter(f(_c8_leave(x, C)))
            C = f.out_channels
        else:
            x = f(x)
            C = getattr(f, "out_channels", C)
        if valid_axis is not None:
            x = ops.zero_tail(x, *valid_axis)
        i += 1
    return x if keep_c8 else _c8_leave(x, C)


def _name_sites(root, prefix):
    """Give every dropout site a stable name (module path) for mask recording in tests."""
    for name, m in root.named_modules():
        m._dv3_site = prefix + ("." + name if name else "")


class Encoder(nn.Module):
    def __init__(self, n_vocab, embed_dim, n_speakers, speaker_embed_dim, padding_idx=None,
                 embedding_weight_std=0.1, convolutions=((64, 5, .1),) * 7, max_positions=512,
                 dropout=0.1, apply_grad_scaling=False):
        super(Encoder, self).__init__()
        self.dropout = dropout
        self.num_attention_layers = None
        self.apply_grad_scaling = apply_grad_scaling
        if apply_grad_scaling:
            raise NotImplementedError("apply_grad_scaling (dead code in the reference: "
                                      "modules.py:67-77 uses removed torch APIs) is not supported")
        self.embed_tokens = Embedding(n_vocab, embed_dim, padding_idx, embedding_weight_std)
        if n_speakers > 1:
            self.speaker_fc1 = Linear(speaker_embed_dim, embed_dim, dropout=dropout)
            self.speaker_fc2 = Linear(speaker_embed_dim, embed_dim, dropout=dropout)
        self.n_speakers = n_speakers

        in_channels = embed_dim
        self.convolutions = nn.ModuleList()
        std_mul = 1.0
        for (out_channels, kernel_size, dilation) in convolutions:
            if in_channels != out_channels:
                self.convolutions.append(Conv1d(in_channels, out_channels, kernel_size=1, padding=0,
                                                dilation=1, std_mul=std_mul))
                self.convolutions.append(nn.ReLU(inplace=True))
                in_channels = out_channels
                std_mul = 2.0
            self.convolutions.append(Conv1dGLU(n_speakers, speaker_embed_dim, in_channels, out_channels,
                                               kernel_size, causal=False, dilation=dilation,
                                               dropout=dropout, std_mul=std_mul, residual=True))
            in_channels = out_channels
            std_mul = 4.0
        self.convolutions.append(Conv1d(in_channels, embed_dim, kernel_size=1, padding=0, dilation=1,
                                        std_mul=std_mul, dropout=dropout))

    def forward(self, text_sequences, text_positions=None, lengths=None, speaker_embed=None):
        assert self.n_speakers == 1 or speaker_embed is not None
        B, T = text_sequences.shape
        # embed + dropout straight into BCT
        x = ops.embedding_bct(text_sequences, self.embed_tokens.weight, self.dropout, self.training,
                              self.embed_tokens.padding_idx, getattr(self, "_dv3_site", "enc") + ".embed_tokens")
        speaker_embed_btc = _drop_speaker_embed(speaker_embed, T, self.dropout, self.training,
                                                getattr(self, "_dv3_site", "enc") + ".speaker_embed")
        if speaker_embed_btc is not None:
            x = x + self._speaker_term(self.speaker_fc1, speaker_embed_btc)
        vl = ops.valid              # the batch is padded beyond its longest text: zeros beyond it, layer by layer
        if vl is not None:
            x = ops.zero_tail(x, *vl.text())
        input_embedding = x
        _fuse_speaker_biases(speaker_embed_btc, self.convolutions)
        x = _run_stack(self.convolutions, x, speaker_embed_btc, valid_axis=vl.text() if vl is not None else None)
        keys = x
        if speaker_embed_btc is not None:
            keys = keys + self._speaker_term(self.speaker_fc2, speaker_embed_btc)
        values = _ScaledAdd.apply(keys, input_embedding, math.sqrt(0.5))
        # reference layout (B, T, C): views of the BCT tensors
        return keys.transpose(1, 2), values.transpose(1, 2)

    @staticmethod
    def _speaker_term(fc, speaker_embed_btc):
        """softsign(fc(speaker_embed_btc)) in BCT, (B, C, T) or broadcastable (B, C, 1)."""
        if speaker_embed_btc.stride(1) == 0:
            return fc.forward_bct(speaker_embed_btc[:, :1, :].transpose(1, 2).contiguous(), ops.EPI_SOFTSIGN)
        return fc.forward_bct(speaker_embed_btc.transpose(1, 2).contiguous(), ops.EPI_SOFTSIGN)


class _ScaledAdd(torch.autograd.Function):
    """alpha * (a + b) in one HIP launch (values = (keys + input_embedding) * sqrt(0.5),
    deepvoice3.py:103)."""

    @staticmethod
    def forward(ctx, a, b, alpha):
        ctx.alpha = alpha
        return ops.axpby(a, b, alpha)

    @staticmethod
    def backward(ctx, dy):
        g = ops.axpby(dy, None, ctx.alpha)
        return g, g, None


class AttentionLayer(nn.Module):
    def __init__(self, conv_channels, embed_dim, dropout=0.1, window_ahead=3, window_backward=1,
                 key_projection=True, value_projection=True):
        super(AttentionLayer, self).__init__()
        self.query_projection = Linear(conv_channels, embed_dim)
        if key_projection:
            self.key_projection = Linear(embed_dim, embed_dim)
            # NB the reference tries to tie key/query init (deepvoice3.py:114-118) by assigning
            # `.weight.data` AFTER weight_norm, which only touches the derived tensor and is
            # overwritten by the next forward pre-hook: weight_g / weight_v stay independent.
            # Same here: no tying.
        else:
            self.key_projection = None
        self.value_projection = Linear(embed_dim, embed_dim) if value_projection else None
        self.out_projection = Linear(embed_dim, conv_channels)
        self.dropout = dropout
        self.window_ahead = window_ahead
        self.window_backward = window_backward

    def forward(self, query, encoder_out, mask=None, last_attended=None):
        """Reference layouts: query (B,Tq,C); encoder_out = (keys (B,E,Tk), values (B,Tk,E));
        mask (B,Tk) bool, True = padded; last_attended python int / device int32[1] / None.
        -> (x (B,Tq,C), attn_scores (B,Tq,Tk))."""
        keys, values = encoder_out
        key_len = None
        if mask is not None:
            key_len = (~mask.view(query.size(0), -1)).sum(dim=1).to(torch.int32)
        x, attn = self.forward_bct(query.transpose(1, 2), keys, values.transpose(1, 2), key_len,
                                   last_attended)
        return x.transpose(1, 2), attn

    def forward_bct(self, query, keys, values, key_len=None, last_attended=None, outer_residual=None):
        """All BCT: query (B,C,Tq), keys (B,E,Tk), values (B,E,Tk).  outer_residual: the decoder's
        `(x + residual) * sqrt(0.5)` (deepvoice3.py:348-349) fused into the out-projection."""
        query = query.contiguous()
        c8 = ops.is_c8(query)       # bf16 storage: c8 query / residuals, fp32 attention core, c8 result
        if self.value_projection is not None:
            values = self.value_projection.forward_bct(values.contiguous())
        if self.key_projection is not None:
            keys = self.key_projection.forward_bct(keys.contiguous())
        q = self.query_projection.forward_bct(query, out_c8=False if c8 else None)
        la = last_attended
        if la is not None and not torch.is_tensor(la):
            la = torch.tensor([int(la)], dtype=torch.int32, device=query.device)
        ctx, attn = ops.attention_core(q, keys, values, key_len, la, self.dropout, self.training,
                                       self.window_backward, self.window_ahead,
                                       getattr(self, "_dv3_site", None))
        x = self.out_projection.forward_bct(ctx, r=query, r2=outer_residual, out_c8=True if c8 else None)
        return x, attn


class Decoder(nn.Module):
    def __init__(self, embed_dim, n_speakers, speaker_embed_dim, in_dim=80, r=5, max_positions=512,
                 padding_idx=None, preattention=((128, 5, 1),) * 4, convolutions=((128, 5, 1),) * 4,
                 attention=True, dropout=0.1, use_memory_mask=False, force_monotonic_attention=False,
                 query_position_rate=1.0, key_position_rate=1.29, window_ahead=3, window_backward=1,
                 key_projection=True, value_projection=True):
        super(Decoder, self).__init__()
        self.dropout = dropout
        self.in_dim = in_dim
        self.r = r
        self.query_position_rate = query_position_rate
        self.key_position_rate = key_position_rate

        if isinstance(attention, bool):
            attention = [attention] * len(convolutions)

        self.embed_query_positions = SinusoidalEncoding(max_positions, convolutions[0][0])
        self.embed_keys_positions = SinusoidalEncoding(max_positions, embed_dim)
        if n_speakers > 1:
            self.speaker_proj1 = Linear(speaker_embed_dim, 1, dropout=dropout)
            self.speaker_proj2 = Linear(speaker_embed_dim, 1, dropout=dropout)
        else:
            self.speaker_proj1, self.speaker_proj2 = None, None

        self.preattention = nn.ModuleList()
        in_channels = in_dim * r
        std_mul = 1.0
        for out_channels, kernel_size, dilation in preattention:
            if in_channels != out_channels:
                self.preattention.append(Conv1d(in_channels, out_channels, kernel_size=1, padding=0,
                                                dilation=1, std_mul=std_mul))
                self.preattention.append(nn.ReLU(inplace=True))
                in_channels = out_channels
                std_mul = 2.0
            self.preattention.append(Conv1dGLU(n_speakers, speaker_embed_dim, in_channels, out_channels,
                                               kernel_size, causal=True, dilation=dilation,
                                               dropout=dropout, std_mul=std_mul, residual=True))
            in_channels = out_channels
            std_mul = 4.0

        self.convolutions = nn.ModuleList()
        self.attention = nn.ModuleList()
        for i, (out_channels, kernel_size, dilation) in enumerate(convolutions):
            assert in_channels == out_channels
            self.convolutions.append(Conv1dGLU(n_speakers, speaker_embed_dim, in_channels, out_channels,
                                               kernel_size, causal=True, dilation=dilation,
                                               dropout=dropout, std_mul=std_mul, residual=False))
            self.attention.append(AttentionLayer(out_channels, embed_dim, dropout=dropout,
                                                 window_ahead=window_ahead,
                                                 window_backward=window_backward,
                                                 key_projection=key_projection,
                                                 value_projection=value_projection)
                                  if attention[i] else None)
            in_channels = out_channels
            std_mul = 4.0
        self.last_conv = Conv1d(in_channels, in_dim * r, kernel_size=1, padding=0, dilation=1,
                                std_mul=std_mul, dropout=dropout)
        self.fc = Linear(in_dim * r, 1)

        self.max_decoder_steps = 200
        self.min_decoder_steps = 10
        self.use_step_graph = False     # free-running decode: replay one hipGraph per step
        self.fast_decode = True         # incremental_forward on the fused decode-step kernels (17 launches per step)
        self.use_memory_mask = use_memory_mask
        if isinstance(force_monotonic_attention, bool):
            self.force_monotonic_attention = [force_monotonic_attention] * len(convolutions)
        else:
            self.force_monotonic_attention = force_monotonic_attention

    # -- position rates (deepvoice3.py:304-315) ---------------------------------------------
    def _rate(self, base, proj, speaker_embed):
        if proj is None:
            return base
        s = proj.forward_bct(speaker_embed.unsqueeze(-1).contiguous(), ops.EPI_SIGMOID)  # (B,1,1)
        return base * s.view(-1)

    def forward(self, encoder_out, inputs=None, text_positions=None, frame_positions=None,
                speaker_embed=None, lengths=None):
        if inputs is None:
            assert text_positions is not None
            self.start_fresh_sequence()
            return self.incremental_forward(encoder_out, text_positions, speaker_embed)

        if inputs.size(-1) == self.in_dim:
            inputs = inputs.reshape(inputs.size(0), inputs.size(1) // self.r, -1)
        assert inputs.size(-1) == self.in_dim * self.r
        B, Td = inputs.size(0), inputs.size(1)
        site = getattr(self, "_dv3_site", "dec")

        speaker_embed_btc = _drop_speaker_embed(speaker_embed, Td, self.dropout, self.training,
                                                site + ".speaker_embed")
        keys, values = encoder_out
        keys_bct, values_bct = keys.transpose(1, 2), values.transpose(1, 2)
        Tk = keys_bct.size(-1)

        key_len = None
        if self.use_memory_mask and lengths is not None:
            key_len = key_lengths_i32(lengths, keys_bct.device)
        elif ops.valid is not None:
            # keys beyond the batch's longest text are padding of the padded SHAPE, not of the batch: out of the softmax
            # (the reference's softmax runs over the batch's own padded length, deepvoice3.py:159-163)
            key_len = ops.valid.key_valid

        if text_positions is not None:
            w = self._rate(self.key_position_rate, self.speaker_proj1, speaker_embed)
            keys_bct = self.embed_keys_positions.forward_bct(text_positions, w, base=keys_bct.contiguous())
        frame_pos_embed = None
        if frame_positions is not None:
            w = self._rate(self.query_position_rate, self.speaker_proj2, speaker_embed)
            frame_pos_embed = self.embed_query_positions.forward_bct(frame_positions, w)

        # (B, Td, C) -> BCT, then the input dropout (deepvoice3.py:320-324)
        x = inputs.transpose(1, 2).contiguous()
        x = ops.dropout(x, self.dropout, self.training, site + ".inputs")

        _fuse_speaker_biases(speaker_embed_btc, list(self.preattention) + list(self.convolutions))
        x = _run_stack(self.preattention, x, speaker_embed_btc, keep_c8=True)
        if ops.is_c8(x) and frame_pos_embed is not None:
            frame_pos_embed = ops.to_c8(frame_pos_embed)

        alignments = []
        for f, attention in zip(self.convolutions, self.attention):
            if attention is None:
                # GLU then (x + residual)*sqrt(.5) == the GLU's own fused residual epilogue
                x = f._run_with_residual(x, speaker_embed_btc)
            else:
                residual = x
                x = f(x, speaker_embed_btc)
                if frame_pos_embed is not None:
                    x = x + frame_pos_embed
                x, alignment = attention.forward_bct(x, keys_bct, values_bct, key_len,
                                                     outer_residual=residual)
                alignments += [alignment]

        x8 = x
        x = _c8_leave(x, self.last_conv.in_channels)
        decoder_states = x.transpose(1, 2).contiguous()
        decoder_states._dv3_bct = x
        c8o = False if ops.is_c8(x8) else None
        outputs = self.last_conv(x8, mode=ops.EPI_SIGMOID, out_c8=c8o).transpose(1, 2)
        pre = self.last_conv(x8, out_c8=c8o)
        done = self.fc.forward_bct(pre, ops.EPI_SIGMOID).transpose(1, 2)
        return outputs, torch.stack(alignments), done, decoder_states

    def incremental_forward(self, encoder_out, text_positions, speaker_embed=None, initial_input=None,
                            test_inputs=None):
        """Greedy autoregressive decode (deepvoice3.py:367-485).  The reference's quirks are
        kept: last_attended comes from batch item 0 (:445), no padding mask, and the running
        `ave_alignment + ave_alignment` (:449).  last_attended stays on the device (no host sync
        per attention layer); the stop rule is checked on the host like the reference does."""
        keys, values = encoder_out
        B = keys.size(0)
        dev = keys.device
        if self.training:
            raise RuntimeError('incremental_forward only supports eval mode')
        if getattr(self, "fast_decode", False) and keys.is_cuda and self._fast_decode_eligible(keys.size(1)):
            return self._incremental_fast(encoder_out, text_positions, speaker_embed, initial_input, test_inputs)
        w = self._rate(self.key_position_rate, self.speaker_proj1, speaker_embed)
        keys_bct = self.embed_keys_positions.forward_bct(text_positions, w,
                                                         base=keys.transpose(1, 2).contiguous())
        values_bct = values.transpose(1, 2).contiguous()
        Tk = keys_bct.size(-1)
        # key / value projections do not depend on the step: hoist them out of the loop
        proj = []
        for att in self.attention:
            if att is None:
                proj.append(None)
                continue
            k = att.key_projection.forward_bct(keys_bct) if att.key_projection is not None else keys_bct
            v = att.value_projection.forward_bct(values_bct) if att.value_projection is not None else values_bct
            proj.append((k, v))

        trace = StepTrace(self.min_decoder_steps, self.max_decoder_steps, test_inputs is not None)
        last_attended = [torch.zeros(1, dtype=torch.int32, device=dev) if v else None
                         for v in self.force_monotonic_attention]
        num_attention_layers = sum([layer is not None for layer in self.attention])
        wq = self._rate(self.query_position_rate, self.speaker_proj2, speaker_embed)
        if initial_input is None:
            initial_input = keys.new_zeros(B, 1, self.in_dim * self.r)

        def step(x, frame_pos):
            """one decoder step (deepvoice3.py:397-461): x (B,1,in_dim*r), frame_pos (B,1) long"""
            frame_pos_embed = self.embed_query_positions.forward_bct(frame_pos, wq)   # (B, C, 1)
            x = self._incremental_stack(self.preattention, x, speaker_embed)
            ave_alignment = None
            for idx, (f, attention) in enumerate(zip(self.convolutions, self.attention)):
                residual = x
                x = f.incremental_forward(x, speaker_embed)
                if attention is not None:
                    xq = x.transpose(1, 2) + frame_pos_embed                       # (B, C, 1)
                    k, v = proj[idx]
                    q = attention.query_projection.forward_bct(xq.contiguous())
                    ctx, alignment = ops.attention_core(q, k, v, None, last_attended[idx], 0.0, False,
                                                        attention.window_backward, attention.window_ahead)
                    xo = attention.out_projection.forward_bct(ctx, r=xq.contiguous())
                    x = xo.transpose(1, 2)
                    if self.force_monotonic_attention[idx]:
                        ops._lib.call("dv3_attn_argmax_i32", alignment.data_ptr(), Tk,
                                      last_attended[idx].data_ptr(), ops._stream())
                    if ave_alignment is None:
                        ave_alignment = alignment
                    else:
                        ave_alignment = ave_alignment + ave_alignment
                x = (x + residual) * math.sqrt(0.5)
            decoder_state = x
            x = self.last_conv.incremental_forward(x)
            ave_alignment = ave_alignment / num_attention_layers
            return torch.sigmoid(x), torch.sigmoid(self.fc(x)), decoder_state, ave_alignment

        # Free-running decode can replay ONE captured hipGraph per step (~100 launches): every address in
        # the step is static (in-place conv windows, device-side last_attended and position counter).
        graphed = bool(getattr(self, "use_step_graph", False)) and test_inputs is None and keys.is_cuda
        t = 0
        if graphed:
            step_in = initial_input.clone()
            step_pos = torch.ones((B, 1), dtype=torch.long, device=dev)
            graph, gout = None, None
        current_input = initial_input
        n_forced = test_inputs.size(1) if test_inputs is not None else None
        while n_forced is None or t < n_forced:
            if n_forced is not None:
                current_input = test_inputs[:, t:t + 1, :]
            elif t > 0 and not graphed:
                current_input = trace.last_output

            if not graphed:
                frame_pos = torch.full((B, 1), t + 1, dtype=torch.long, device=dev)
                output, done, decoder_state, ave_alignment = step(current_input, frame_pos)
            else:
                if t == 0:        # eager: also fills the packed-weight caches the graph will reuse
                    res = step(step_in, step_pos)
                    step_in.copy_(res[0])
                    step_pos.add_(1)
                else:
                    if graph is None:
                        graph = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(graph):
                            gout = step(step_in, step_pos)
                            step_in.copy_(gout[0])
                            step_pos.add_(1)
                    graph.replay()
                    res = gout
                output, done, decoder_state, ave_alignment = res[0], res[1].clone(), res[2], res[3]

            trace.push(output, ave_alignment, decoder_state, done)      # (copies: the graph's outputs are reused)
            t += 1
            if trace.stop(done):
                break
        return trace.result()

    def _fast_decode_eligible(self, Tk):
        """What the fused step program (csrc/decode_step.hip) takes; anything else runs the module-by-module
        path below, which handles every configuration the reference does."""
        def fits(conv):      # dv3_conv_step_f32 stages the k-tap window of a batch group + its reduction stages in LDS
            return ops.conv_step_fits(conv.kernel_size[0], conv.in_channels)
        mods, i = list(self.preattention), 0
        while i < len(mods):
            f = mods[i]
            if isinstance(f, Conv1dGLU):
                if not fits(f.conv):
                    return False
            elif isinstance(f, _conv.Conv1d):
                if f.kernel_size[0] != 1 or not fits(f):
                    return False
                i += int(i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU))
            else:
                return False
            i += 1
        for f, att in zip(self.convolutions, self.attention):
            if not isinstance(f, Conv1dGLU) or not fits(f.conv):
                return False
            if att is not None:
                E = att.query_projection.out_features
                if (E + Tk) * 4 > 64 * 1024:
                    return False
        return True

    # -- the same decode on the fused step kernels (csrc/decode_step.hip) -------------------------------
    def _incremental_fast(self, encoder_out, text_positions, speaker_embed=None, initial_input=None,
                          test_inputs=None):
        """incremental_forward as a flat per-step program (decode_program.StepProgram): one conv-step entry per conv /
        projection layer and one attention-step entry per attention read (17 per step for the ljspeech preset instead
        of ~100 module calls), walked by ONE persistent launch for the whole utterance (or launch by launch):
        ring buffers indexed by the step counter (no window shifting), every layer tail fused, the per-step
        outputs written straight into the stacked result tensors.  Same quirks as the module-by-module path
        (last_attended from batch item 0, no padding mask, `ave_alignment + ave_alignment`)."""
        from .decode_program import StepProgram
        keys, values = encoder_out
        B, dev = keys.size(0), keys.device
        with torch.no_grad():
            w = self._rate(self.key_position_rate, self.speaker_proj1, speaker_embed)
            keys_bct = self.embed_keys_positions.forward_bct(text_positions, w, base=keys.transpose(1, 2).contiguous())
            values_bct = values.transpose(1, 2).contiguous()
            Tk = keys_bct.size(-1)
            proj = []
            for att in self.attention:
                if att is None:
                    proj.append(None)
                    continue
                k = att.key_projection.forward_bct(keys_bct) if att.key_projection is not None else keys_bct
                v = att.value_projection.forward_bct(values_bct) if att.value_projection is not None else values_bct
                proj.append((k.contiguous(), v.contiguous()))
            D = self.in_dim * self.r
            n_max = test_inputs.size(1) if test_inputs is not None else self.max_decoder_steps + 1
            wq = self._rate(self.query_position_rate, self.speaker_proj2, speaker_embed)
            pos = torch.arange(1, n_max + 1, device=dev, dtype=torch.long)[None].expand(B, n_max).contiguous()
            pe_all = self.embed_query_positions.forward_bct(pos, wq).permute(2, 0, 1).contiguous()   # (n_max, B, C)
            P = StepProgram(B, dev)
            cur_in = (initial_input.reshape(B, D).clone().float() if initial_input is not None else P.buffer(B, D))
            free_running = test_inputs is None
            nxt_in = cur_in if free_running else P.buffer(B, D)
            n_att = sum(1 for a in self.attention if a is not None)
            Cs = self.convolutions[-1].conv.out_channels // 2
            outs, dones_seq = P.buffer(n_max, B, D), P.buffer(n_max, B, 1)
            states, aligns = P.buffer(n_max, B, Cs), P.buffer(n_max, B, Tk)
            P.keep.extend([keys_bct, values_bct, proj, pe_all, cur_in, nxt_in])

            def glu_step(f, x, residual, **kw):
                spk = None
                if f.speaker_proj is not None:
                    se = speaker_embed if speaker_embed.dim() == 2 else speaker_embed[:, 0, :]
                    spk = f.speaker_bias(se).contiguous()
                return P.conv_step(f.conv, x, ops.EPI_GLU, f.conv.out_channels // 2, k=f.conv.kernel_size[0],
                                   dil=f.conv.dilation[0], gated=True, residual=residual, spk=spk, **kw)

            # ---- the step program (deepvoice3.py:397-461) ----
            x = cur_in
            mods, i = list(self.preattention), 0
            while i < len(mods):
                f = mods[i]
                if isinstance(f, Conv1dGLU):
                    x = glu_step(f, x, f.residual)
                elif isinstance(f, _conv.Conv1d):
                    relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                    if f.kernel_size[0] != 1:
                        raise RuntimeError("fast decode: only 1x1 plain convolutions in the pre-attention stack")
                    x = P.conv_step(f, x, ops.EPI_RELU if relu else ops.EPI_LINEAR, f.out_channels)
                    i += int(relu)
                else:
                    raise RuntimeError("fast decode: unsupported pre-attention module %r" % type(f))
                i += 1
            first_att = True
            n_conv = len(self.convolutions)
            for idx, (f, attention) in enumerate(zip(self.convolutions, self.attention)):
                residual = x
                st = states if idx == n_conv - 1 else None
                if attention is None:
                    x = glu_step(f, x, f.residual, r2=residual, out_seq=st)   # (glu + residual) * sqrt(.5)
                    continue
                xq = glu_step(f, x, f.residual, post_add=pe_all)              # conv output + the step's position code
                kp, vp = proj[idx]
                q = P.conv_step(attention.query_projection, xq, ops.EPI_LINEAR, attention.query_projection.out_features)
                # ave_alignment = the FIRST layer's alignment * 2^(n-1)/n (the reference's `ave + ave`)
                ctx = P.attn_step(q, kp, vp, attention.window_backward, attention.window_ahead,
                                  self.force_monotonic_attention[idx], attn_seq=aligns if first_att else None)
                first_att = False
                x = P.conv_step(attention.out_projection, ctx, ops.EPI_LINEAR, attention.out_projection.out_features,
                                r=xq, r2=residual, out_seq=st)
            pre = P.conv_step(self.last_conv, x, ops.EPI_LINEAR, D, y_act=nxt_in, out_seq=outs)
            P.conv_step(self.fc, pre, ops.EPI_SIGMOID, 1, out_seq=dones_seq)
            t = P.decode(cur_in, test_inputs, dones_seq, self.min_decoder_steps, self.max_decoder_steps,
                         getattr(self, "use_step_graph", False), getattr(self, "persistent_decode", None),
                         getattr(self, "launched_decode", None))
            scale = float(2 ** (n_att - 1)) / n_att if n_att else 1.0
            alignments = aligns[:t].transpose(0, 1)
            if scale != 1.0:
                alignments = alignments * scale
            decoder_states = states[:t].transpose(0, 1).contiguous()
            outputs = outs[:t].transpose(0, 1).contiguous()
            dones = [dones_seq[i].view(B, 1, 1) for i in range(t)]
        return outputs, alignments, dones, decoder_states

    @staticmethod
    def _incremental_stack(modules, x, speaker_embed):
        n = len(modules)
        i = 0
        while i < n:
            f = modules[i]
            if isinstance(f, Conv1dGLU):
                x = f.incremental_forward(x, speaker_embed)
            elif isinstance(f, _conv.Conv1d):
                if i + 1 < n and isinstance(modules[i + 1], nn.ReLU):
                    x = f.incremental_forward(x, _gate=dict(mode=ops.EPI_RELU))
                    i += 1
                else:
                    x = f.incremental_forward(x)
            else:
                x = f(x)
            i += 1
        return x

    def start_fresh_sequence(self):
        _clear_modules(self.preattention)
        _clear_modules(self.convolutions)
        self.last_conv.clear_buffer()


def _clear_modules(modules):
    for m in modules:
        try:
            m.clear_buffer()
        except AttributeError:
            pass


class Converter(nn.Module):
    def __init__(self, n_speakers, speaker_embed_dim, in_dim, out_dim, convolutions=((256, 5, 1),) * 4,
                 time_upsampling=1, dropout=0.1):
        super(Converter, self).__init__()
        self.dropout = dropout
        self.in_dim = in_dim
        self.out_dim = out_dim
        self.n_speakers = n_speakers

        in_channels = convolutions[0][0]

        def glu(dilation, std_mul):
            return Conv1dGLU(n_speakers, speaker_embed_dim, in_channels, in_channels, kernel_size=3,
                             causal=False, dilation=dilation, dropout=dropout, std_mul=std_mul,
                             residual=True)

        def up(std_mul):
            return ConvTranspose1d(in_channels, in_channels, kernel_size=2, padding=0, stride=2,
                                   std_mul=std_mul)

        first = Conv1d(in_dim, in_channels, kernel_size=1, padding=0, dilation=1, std_mul=1.0)
        if time_upsampling == 4:
            self.convolutions = nn.ModuleList([first, up(1.0), glu(1, 1.0), glu(3, 4.0),
                                               up(4.0), glu(1, 1.0), glu(3, 4.0)])
        elif time_upsampling == 2:
            self.convolutions = nn.ModuleList([first, up(1.0), glu(1, 1.0), glu(3, 4.0)])
        elif time_upsampling == 1:
            self.convolutions = nn.ModuleList([first, glu(3, 4.0)])
        else:
            raise ValueError("Not supported")

        std_mul = 4.0
        for (out_channels, kernel_size, dilation) in convolutions:
            if in_channels != out_channels:
                self.convolutions.append(Conv1d(in_channels, out_channels, kernel_size=1, padding=0,
                                                dilation=1, std_mul=std_mul))
                self.convolutions.append(nn.ReLU(inplace=True))
                in_channels = out_channels
                std_mul = 2.0
            self.convolutions.append(Conv1dGLU(n_speakers, speaker_embed_dim, in_channels, out_channels,
                                               kernel_size, causal=False, dilation=dilation,
                                               dropout=dropout, std_mul=std_mul, residual=True))
            in_channels = out_channels
            std_mul = 4.0
        self.convolutions.append(Conv1d(in_channels, out_dim, kernel_size=1, padding=0, dilation=1,
                                        std_mul=std_mul, dropout=dropout))

    def forward(self, x, speaker_embed=None):
        """x (B, T, in_dim) -> (B, T*time_upsampling, out_dim), sigmoid applied."""
        assert self.n_speakers == 1 or speaker_embed is not None
        site = getattr(self, "_dv3_site", "postnet")
        bct = getattr(x, "_dv3_bct", None)
        x = bct if bct is not None else x.transpose(1, 2).contiguous()
        speaker_embed_btc = _drop_speaker_embed(speaker_embed, x.size(2), self.dropout, self.training,
                                                "%s.speaker_embed.t%d" % (site, x.size(2)))
        mods = self.convolutions
        n = len(mods)
        # the Conv1dGLU layers by time resolution (each ConvTranspose1d doubles it; the embedding is dropped anew there)
        by_t, cur_t = {}, x.size(2)
        for f in mods:
            if isinstance(f, _conv.ConvTranspose1d):
                cur_t *= 2
            elif isinstance(f, Conv1dGLU):
                by_t.setdefault(cur_t, []).append(f)
        _fuse_speaker_biases(speaker_embed_btc, by_t.get(x.size(2), ()))
        i = 0
        x = _c8_enter(x)            # bf16 storage: channel-blocked bf16 between the layers (see _run_stack)
        vl = ops.valid              # a batch padded beyond its own maximum: zeros beyond it after every layer (_run_stack)

        def ztail(x):
            if vl is None:
                return x
            ptr, tail, mult = vl.axis_for(x.size(2))
            return ops.zero_tail(x, ptr, tail, mult)

        x = ztail(x)
        while i < n:
            f = mods[i]
            if speaker_embed_btc is not None and speaker_embed_btc.size(1) != x.size(2):
                speaker_embed_btc = _drop_speaker_embed(speaker_embed, x.size(2), self.dropout, self.training,
                                                        "%s.speaker_embed.t%d" % (site, x.size(2)))
                _fuse_speaker_biases(speaker_embed_btc, by_t.get(x.size(2), ()))
            last = (i == n - 1)
            if isinstance(f, Conv1dGLU):
                x = f(x, speaker_embed_btc)
                if not last:
                    ops.mark_sole_consumer(x)  # the next layer of the stack is x's only consumer (ops.GateFuse)
            elif isinstance(f, _conv.Conv1d):
                if last:
                    x = _conv1d_c8(f, x, True, mode=ops.EPI_SIGMOID)      # torch.sigmoid(x) of deepvoice3.py:604, fused
                elif i + 1 < n and isinstance(mods[i + 1], nn.ReLU):
                    x = _conv1d_c8(f, x, False, mode=ops.EPI_RELU)
                    i += 1
                else:
                    x = _conv1d_c8(f, x, False)
            elif ops.is_c8(x):      # ConvTranspose1d: fp32 layer between two conversions
                x = _c8_enter(f(_c8_leave(x, f.in_channels)))
            else:
                x = f(x)
            if not last:
                x = ztail(x)
            i += 1
        return _c8_leave(x, self.out_dim).transpose(1, 2)
