# coding: utf-8
"""Nyanko (DCTTS-style) blocks on HIP kernels: HighwayConv1d stacks with dilations 1,3,9,27.

API mirror of the reference's deepvoice3_pytorch/nyanko.py (class names, constructor arguments,
module and parameter names, forward signatures).  BCT activations, one fused tap-GEMM launch
per HighwayConv1d / Conv1d(+ReLU / +Sigmoid).
"""
import numpy as np
import torch
from torch import nn

from . import ops
from . import conv as _conv
from .modules import Embedding, Linear, Conv1d, ConvTranspose1d
from .modules import HighwayConv1d, get_mask_from_lengths, key_lengths_i32
from .modules import position_encoding_init
from .deepvoice3 import AttentionLayer, _c8_enter, _c8_leave, _conv1d_c8
from .decode_program import StepTrace


def _run_seq(mods, x, valid_axis=None):
    """nn.Sequential / ModuleList of {Conv1d, ReLU, Sigmoid, HighwayConv1d, ConvTranspose1d} on
    BCT x with Conv1d+ReLU / Conv1d+Sigmoid fused.
    valid_axis (NON-CAUSAL sequences on a batch padded beyond its own maximum, ops.ValidLengths): a function
    T -> (device int32[1], most surplus columns, multiplier) -- the surplus columns of every layer's output, and of its
    gradient in backward, are zeroed (deepvoice3._run_stack)."""
    mods = list(mods)
    n = len(mods)
    i = 0
    C = x.size(1)           # channel count of x (a c8 tensor pads it to a multiple of 32)
    x = _c8_enter(x)        # bf16 GEMM mode: channel-blocked bf16 between the layers, fp32 (B, C, T) result

    def ztail(x):
        if valid_axis is None:
            return x
        ptr, tail, mult = valid_axis(x.size(2))
        return ops.zero_tail(x, ptr, tail, mult)

    x = ztail(x)
    while i < n:
        f = mods[i]
        C = getattr(f, "out_channels", C) if not isinstance(f, HighwayConv1d) else C
        if isinstance(f, _conv.Conv1d):
            nxt = mods[i + 1] if i + 1 < n else None
            fused = isinstance(nxt, (nn.ReLU, nn.Sigmoid))
            last = i + (2 if fused else 1) >= n
            mode = ops.EPI_RELU if isinstance(nxt, nn.ReLU) else ops.EPI_SIGMOID if isinstance(nxt, nn.Sigmoid) \
                else ops.EPI_LINEAR
            x = _conv1d_c8(f, x, last, mode=mode)
            i += int(fused)
        elif isinstance(f, nn.ReLU):
            x = torch.relu(x)
        elif isinstance(f, nn.Sigmoid):
            x = torch.sigmoid(x)
        elif isinstance(f, _conv.ConvTranspose1d) and ops.is_c8(x):   # fp32 layer between two conversions
            x = _c8_enter(f(_c8_leave(x, f.in_channels)))
        else:
            x = f(x)
            if isinstance(f, HighwayConv1d) and i + 1 < n:
                ops.mark_sole_consumer(x)      # the next layer of the sequence is x's only consumer (ops.GateFuse)
        i += 1
        if i < n:
            x = ztail(x)
    return _c8_leave(x, C)


def _run_seq_incremental(mods, x):
    mods = list(mods)
    n = len(mods)
    i = 0
    while i < n:
        f = mods[i]
        if isinstance(f, _conv.Conv1d):
            if i + 1 < n and isinstance(mods[i + 1], nn.ReLU):
                x = f.incremental_forward(x, _gate=dict(mode=ops.EPI_RELU))
                i += 1
            else:
                x = f.incremental_forward(x)
        elif isinstance(f, HighwayConv1d):
            x = f.incremental_forward(x)
        else:
            x = f(x)
        i += 1
    return x


class Encoder(nn.Module):
    def __init__(self, n_vocab, embed_dim, channels, kernel_size=3, n_speakers=1, speaker_embed_dim=16,
                 embedding_weight_std=0.01, padding_idx=None, dropout=0.1):
        super(Encoder, self).__init__()
        self.dropout = dropout
        self.embed_tokens = Embedding(n_vocab, embed_dim, padding_idx, embedding_weight_std)
        E, D = embed_dim, channels

        def hw(k, d):
            return HighwayConv1d(2 * D, 2 * D, kernel_size=k, padding=None, dilation=d, std_mul=1.0,
                                 dropout=dropout)

        self.convnet = nn.Sequential(
            Conv1d(E, 2 * D, kernel_size=1, padding=0, dilation=1, std_mul=1.0),
            nn.ReLU(inplace=True),
            Conv1d(2 * D, 2 * D, kernel_size=1, padding=0, dilation=1, std_mul=2.0),
            hw(kernel_size, 1), hw(kernel_size, 3), hw(kernel_size, 9), hw(kernel_size, 27),
            hw(kernel_size, 1), hw(kernel_size, 3), hw(kernel_size, 9), hw(kernel_size, 27),
            hw(kernel_size, 1), hw(kernel_size, 1),
            HighwayConv1d(2 * D, 2 * D, kernel_size=1, padding=0, dilation=1, std_mul=1.0,
                          dropout=dropout),
        )

    def forward(self, text_sequences, text_positions=None, lengths=None, speaker_embed=None):
        x = ops.embedding_bct(text_sequences, self.embed_tokens.weight, 0.0, False,
                              self.embed_tokens.padding_idx)
        vl = ops.valid          # the batch is padded beyond its longest text: zeros beyond it, layer by layer
        x = _run_seq(self.convnet, x, (lambda T: vl.text() + (1,)) if vl is not None else None)     # (B, 2D, T)
        D = x.size(1) // 2
        keys, values = x[:, :D, :], x[:, D:, :]             # channel halves (nyanko.py:69)
        return keys.transpose(1, 2), values.transpose(1, 2)


class Decoder(nn.Module):
    def __init__(self, embed_dim, in_dim=80, r=5, channels=256, kernel_size=3, n_speakers=1,
                 speaker_embed_dim=16, max_positions=512, padding_idx=None, dropout=0.1,
                 use_memory_mask=False, force_monotonic_attention=False, query_position_rate=1.0,
                 key_position_rate=1.29, window_ahead=3, window_backward=1, key_projection=False,
                 value_projection=False):
        super(Decoder, self).__init__()
        self.dropout = dropout
        self.in_dim = in_dim
        self.r = r
        D = channels
        F = in_dim * r

        def hw(d):
            return HighwayConv1d(D, D, kernel_size=kernel_size, padding=None, dilation=d, causal=True,
                                 std_mul=1.0, dropout=dropout)

        self.audio_encoder_modules = nn.ModuleList([
            Conv1d(F, D, kernel_size=1, padding=0, dilation=1, std_mul=1.0),
            nn.ReLU(inplace=True),
            Conv1d(D, D, kernel_size=1, padding=0, dilation=1, std_mul=2.0),
            nn.ReLU(inplace=True),
            Conv1d(D, D, kernel_size=1, padding=0, dilation=1, std_mul=2.0),
            hw(1), hw(3), hw(9), hw(27), hw(1), hw(3), hw(9), hw(27), hw(3), hw(3),
        ])
        self.attention = AttentionLayer(D, D, dropout=dropout, window_ahead=window_ahead,
                                        window_backward=window_backward, key_projection=key_projection,
                                        value_projection=value_projection)
        self.audio_decoder_modules = nn.ModuleList([
            Conv1d(2 * D, D, kernel_size=1, padding=0, dilation=1, std_mul=1.0),
            hw(1), hw(3), hw(9), hw(27), hw(1), hw(1),
            Conv1d(D, D, kernel_size=1, padding=0, dilation=1, std_mul=1.0),
            nn.ReLU(inplace=True),
            Conv1d(D, D, kernel_size=1, padding=0, dilation=1, std_mul=2.0),
            nn.ReLU(inplace=True),
            Conv1d(D, D, kernel_size=1, padding=0, dilation=1, std_mul=2.0),
            nn.ReLU(inplace=True),
        ])
        self.last_conv = Conv1d(D, F, kernel_size=1, padding=0, dilation=1, std_mul=2.0)
        self.fc = Linear(F, 1)

        # Position encodings: frozen tables with the rate baked in (nyanko.py:161-169)
        self.embed_query_positions = Embedding(max_positions, D, padding_idx)
        self.embed_query_positions.weight.data = position_encoding_init(
            max_positions, D, position_rate=query_position_rate, sinusoidal=True)
        self.embed_keys_positions = Embedding(max_positions, D, padding_idx)
        self.embed_keys_positions.weight.data = position_encoding_init(
            max_positions, D, position_rate=key_position_rate, sinusoidal=True)

        self.max_decoder_steps = 200
        self.min_decoder_steps = 10
        self.use_memory_mask = use_memory_mask
        self.force_monotonic_attention = force_monotonic_attention

    def forward(self, encoder_out, inputs=None, text_positions=None, frame_positions=None,
                speaker_embed=None, lengths=None):
        if inputs is None:
            assert text_positions is not None
            self.start_fresh_sequence()
            return self.incremental_forward(encoder_out, text_positions)

        if inputs.size(-1) == self.in_dim:
            inputs = inputs.reshape(inputs.size(0), inputs.size(1) // self.r, -1)
        assert inputs.size(-1) == self.in_dim * self.r

        keys, values = encoder_out
        keys_bct, values_bct = keys.transpose(1, 2), values.transpose(1, 2)
        key_len = None
        if self.use_memory_mask and lengths is not None:
            key_len = key_lengths_i32(lengths, keys_bct.device)
        elif ops.valid is not None:       # keys beyond the batch's longest text: out of the softmax (deepvoice3.Decoder)
            key_len = ops.valid.key_valid

        if text_positions is not None:
            keys_bct = ops.add_position_encoding(keys_bct.contiguous(), text_positions,
                                                 self.embed_keys_positions.weight, None, False)

        x = _run_seq(self.audio_encoder_modules, inputs.transpose(1, 2).contiguous())
        Q = x
        if frame_positions is not None:
            x = ops.add_position_encoding(x, frame_positions, self.embed_query_positions.weight, None, False)
        R, alignments = self.attention.forward_bct(x, keys_bct, values_bct, key_len)

        x = torch.cat((R, Q), dim=1)
        x = _run_seq(self.audio_decoder_modules, x)
        decoder_states = x.transpose(1, 2).contiguous()
        decoder_states._dv3_bct = x
        outputs = self.last_conv(x, mode=ops.EPI_SIGMOID).transpose(1, 2)
        done = self.fc.forward_bct(self.last_conv(x), ops.EPI_SIGMOID).transpose(1, 2)
        return outputs, alignments.unsqueeze(0), done, decoder_states

    def _fast_decode_eligible(self, Tk):
        """what the fused step program below takes (csrc/decode_step.hip stages a layer's k-tap window of 4 batch items
        and the attention scores in 64 KB of LDS); anything else runs the module-by-module path"""
        def fits(conv):
            return ops.conv_step_fits(conv.kernel_size[0], conv.in_channels)
        for mods in (self.audio_encoder_modules, self.audio_decoder_modules):
            for f in mods:
                if isinstance(f, HighwayConv1d):
                    if f.glu or not fits(f.conv):
                        return False
                elif isinstance(f, _conv.Conv1d):
                    if f.kernel_size[0] != 1 or not fits(f):
                        return False
                elif not isinstance(f, nn.ReLU):
                    return False
        E = self.attention.query_projection.out_features
        return (E + Tk) * 4 <= 64 * 1024 and fits(self.last_conv)

    def _incremental_fast(self, encoder_out, text_positions, initial_input=None, test_inputs=None):
        """incremental_forward (nyanko.py:250-338) as a flat per-step launch program (decode_program.StepProgram):
        28 conv-step launches + one attention-step launch per decoder step instead of ~150 module calls; the
        concat [R, Q] (nyanko.py:308) is one (B, 2D) buffer both producers write into."""
        from .decode_program import StepProgram
        keys, values = encoder_out
        B, dev = keys.size(0), keys.device
        with torch.no_grad():
            keys_bct = keys.transpose(1, 2).contiguous()
            if text_positions is not None:
                keys_bct = ops.add_position_encoding(keys_bct, text_positions, self.embed_keys_positions.weight,
                                                     None, False)
            values_bct = values.transpose(1, 2).contiguous()
            att = self.attention
            k = (att.key_projection.forward_bct(keys_bct) if att.key_projection is not None else keys_bct).contiguous()
            v = (att.value_projection.forward_bct(values_bct) if att.value_projection is not None else values_bct).contiguous()
            Tk = k.size(-1)
            F = self.in_dim * self.r
            n_max = test_inputs.size(1) if test_inputs is not None else self.max_decoder_steps + 1
            P = StepProgram(B, dev)
            # position code of step t = row t + 1 of the frozen table (nyanko.py:162-166), the same for every item
            pe_all = self.embed_query_positions.weight[1:n_max + 1].detach().contiguous()      # (n_max, D)
            if pe_all.size(0) < n_max:
                raise RuntimeError("decoder steps exceed max_positions")
            pe_all = pe_all[:, None, :].expand(n_max, B, pe_all.size(1))                          # batch stride 0
            cur_in = (initial_input.reshape(B, F).clone().float() if initial_input is not None else P.buffer(B, F))
            free_running = test_inputs is None
            nxt_in = cur_in if free_running else P.buffer(B, F)
            D = self.last_conv.in_channels
            outs, dones_seq = P.buffer(n_max, B, F), P.buffer(n_max, B, 1)
            states, aligns = P.buffer(n_max, B, D), P.buffer(n_max, B, Tk)
            cat = P.buffer(B, 2 * D)                       # [R | Q]
            xq = P.buffer(B, D)
            P.keep.extend([k, v, pe_all, cur_in, nxt_in])

            def run(mods, x, last_kw=None):
                """a Conv1d(+ReLU) / HighwayConv1d stack, one launch per layer; last_kw: extra outputs of the last layer"""
                mods = list(mods)
                i = 0
                while i < len(mods):
                    f = mods[i]
                    relu = isinstance(f, _conv.Conv1d) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                    last = (i + int(relu)) == len(mods) - 1
                    kw = dict(last_kw) if (last and last_kw) else {}
                    if isinstance(f, HighwayConv1d):
                        x = P.conv_step(f.conv, x, ops.EPI_HIGHWAY, f.conv.out_channels // 2, k=f.conv.kernel_size[0],
                                        dil=f.conv.dilation[0], gated=True, **kw)
                    else:
                        x = P.conv_step(f, x, ops.EPI_RELU if relu else ops.EPI_LINEAR, f.out_channels, **kw)
                        i += int(relu)
                    i += 1
                return x

            # audio encoder: Q (into the concat buffer) and Q + position code (the attention query input)
            run(self.audio_encoder_modules, cur_in, dict(y=xq, post_add=pe_all, y_pre=cat[:, D:]))
            q = P.conv_step(att.query_projection, xq, ops.EPI_LINEAR, att.query_projection.out_features)
            ctx = P.attn_step(q, k, v, att.window_backward, att.window_ahead, self.force_monotonic_attention,
                              attn_seq=aligns)
            P.conv_step(att.out_projection, ctx, ops.EPI_LINEAR, att.out_projection.out_features, r=xq, y=cat[:, :D])
            x = run(self.audio_decoder_modules, cat, dict(out_seq=states))
            pre = P.conv_step(self.last_conv, x, ops.EPI_LINEAR, F, y_act=nxt_in, out_seq=outs)
            P.conv_step(self.fc, pre, ops.EPI_SIGMOID, 1, out_seq=dones_seq)
            t = P.decode(cur_in, test_inputs, dones_seq, self.min_decoder_steps, self.max_decoder_steps,
                         getattr(self, "use_step_graph", False), getattr(self, "persistent_decode", None),
                         getattr(self, "launched_decode", None))
            alignments = aligns[:t].transpose(0, 1)
            decoder_states = states[:t].transpose(0, 1).contiguous()
            outputs = outs[:t].transpose(0, 1).contiguous()
            dones = [dones_seq[i].view(B, 1, 1) for i in range(t)]
        return outputs, alignments, dones, decoder_states

    def incremental_forward(self, encoder_out, text_positions, initial_input=None, test_inputs=None):
        """nyanko.py:250-338."""
        keys, values = encoder_out
        if getattr(self, "fast_decode", True) and keys.is_cuda and self._fast_decode_eligible(keys.size(1)):
            return self._incremental_fast(encoder_out, text_positions, initial_input, test_inputs)
        B = keys.size(0)
        dev = keys.device
        keys_bct = keys.transpose(1, 2).contiguous()
        if text_positions is not None:
            keys_bct = ops.add_position_encoding(keys_bct, text_positions, self.embed_keys_positions.weight,
                                                 None, False)
        values_bct = values.transpose(1, 2).contiguous()
        Tk = keys_bct.size(-1)
        att = self.attention
        k = att.key_projection.forward_bct(keys_bct) if att.key_projection is not None else keys_bct
        v = att.value_projection.forward_bct(values_bct) if att.value_projection is not None else values_bct

        trace = StepTrace(self.min_decoder_steps, self.max_decoder_steps, test_inputs is not None)
        last_attended = torch.zeros(1, dtype=torch.int32, device=dev) if self.force_monotonic_attention else None
        if initial_input is None:
            initial_input = keys.new_zeros(B, 1, self.in_dim * self.r)
        n_forced = test_inputs.size(1) if test_inputs is not None else None
        t = 0
        while n_forced is None or t < n_forced:
            frame_pos = torch.full((B, 1), t + 1, dtype=torch.long, device=dev)
            if n_forced is not None:
                current_input = test_inputs[:, t:t + 1, :]
            else:
                current_input = trace.last_output if t > 0 else initial_input
            x = _run_seq_incremental(self.audio_encoder_modules, current_input)      # (B, 1, D)
            Q = x
            xq = ops.add_position_encoding(x.transpose(1, 2).contiguous(), frame_pos,
                                           self.embed_query_positions.weight, None, False)
            q = att.query_projection.forward_bct(xq)
            ctx, alignment = ops.attention_core(q, k, v, None, last_attended, 0.0, False,
                                                att.window_backward, att.window_ahead)
            R = att.out_projection.forward_bct(ctx, r=xq).transpose(1, 2)
            if self.force_monotonic_attention:
                ops._lib.call("dv3_attn_argmax_i32", alignment.data_ptr(), Tk, last_attended.data_ptr(),
                              ops._stream())
            x = torch.cat((R, Q), dim=-1)
            x = _run_seq_incremental(self.audio_decoder_modules, x)
            decoder_state = x
            x = self.last_conv.incremental_forward(x)
            done = torch.sigmoid(self.fc(x))
            trace.push(torch.sigmoid(x), alignment, decoder_state, done)
            t += 1
            if trace.stop(done):
                break
        return trace.result()

    def start_fresh_sequence(self):
        """forget the incremental state of every layer that keeps one (nyanko.py:340-343)"""
        stateful = [m for seq in (self.audio_encoder_modules, self.audio_decoder_modules) for m in seq
                    if hasattr(m, "clear_buffer")] + [self.last_conv]
        for m in stateful:
            m.clear_buffer()


class Converter(nn.Module):
    def __init__(self, in_dim, out_dim, channels=512, kernel_size=3, dropout=0.1):
        super(Converter, self).__init__()
        self.dropout = dropout
        self.in_dim = in_dim
        self.out_dim = out_dim
        F, Fd, C = in_dim, out_dim, channels

        def hw(c, d):
            return HighwayConv1d(c, c, kernel_size=kernel_size, padding=None, dilation=d, std_mul=1.0,
                                 dropout=dropout)

        def c11(i, o, std_mul):
            return Conv1d(i, o, kernel_size=1, padding=0, dilation=1, std_mul=std_mul)

        self.convnet = nn.Sequential(
            c11(F, C, 1.0),
            hw(C, 1), hw(C, 3),
            ConvTranspose1d(C, C, kernel_size=2, padding=0, stride=2, std_mul=1.0),
            hw(C, 1), hw(C, 3),
            ConvTranspose1d(C, C, kernel_size=2, padding=0, stride=2, std_mul=1.0),
            hw(C, 1), hw(C, 3),
            c11(C, 2 * C, 1.0),
            hw(2 * C, 1), hw(2 * C, 1),
            c11(2 * C, Fd, 1.0),
            c11(Fd, Fd, 1.0), nn.ReLU(inplace=True),
            c11(Fd, Fd, 2.0), nn.ReLU(inplace=True),
            c11(Fd, Fd, 2.0), nn.Sigmoid(),
        )

    def forward(self, x, speaker_embed=None):
        bct = getattr(x, "_dv3_bct", None)
        x = bct if bct is not None else x.transpose(1, 2).contiguous()
        vl = ops.valid
        return _run_seq(self.convnet, x, vl.axis_for if vl is not None else None).transpose(1, 2)
