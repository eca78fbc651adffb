# coding: utf-8
"""Primitive modules of the DeepVoice3 / Nyanko hot path on HIP kernels.

API mirror of the reference's deepvoice3_pytorch/modules.py (same names, constructor
arguments, parameter names and error behaviour); the arithmetic is one fused tap-GEMM launch
per layer (csrc/conv_gemm.hip) instead of ~8 torch ops.
"""
import math

import numpy as np
import torch
from torch import nn

from . import ops
from . import conv as _conv


def position_encoding_init(n_position, d_pos_vec, position_rate=1.0, sinusoidal=True):
    """Sinusoid position encoding table; row 0 (padding position) is zero.
    Same values as the reference (modules.py:10-24): float64 angles rounded to float32,
    then sin (even dims) / cos (odd dims) evaluated in float32."""
    pos = np.arange(n_position, dtype=np.float64).reshape(-1, 1)
    dims = np.arange(d_pos_vec)
    angles = position_rate * pos / np.power(10000, 2 * (dims // 2) / d_pos_vec).reshape(1, -1)
    angles[0, :] = 0.0
    table = torch.from_numpy(angles).float()
    if sinusoidal:
        table[1:, 0::2] = torch.sin(table[1:, 0::2])
        table[1:, 1::2] = torch.cos(table[1:, 1::2])
    return table


class SinusoidalEncoding(nn.Embedding):
    """Raw-angle table (a frozen Parameter named `weight`, as in the reference's state_dict);
    forward evaluates sin/cos(w * angle) only at the gathered positions, on the device, for a
    scalar or per-batch rate w (reference: modules.py:34-64, which rebuilds the whole table
    per call and loops over the batch in Python)."""

    def __init__(self, num_embeddings, embedding_dim, *args, **kwargs):
        super(SinusoidalEncoding, self).__init__(num_embeddings, embedding_dim, padding_idx=0,
                                                 *args, **kwargs)
        self.weight.data = position_encoding_init(num_embeddings, embedding_dim, position_rate=1.0,
                                                  sinusoidal=False)

    def forward(self, x, w=1.0):
        """x (B, T) long -> (B, T, D), reference layout."""
        return self.forward_bct(x, w).transpose(1, 2)

    def forward_bct(self, x, w=1.0, base=None):
        """-> (B, D, T); optionally added to `base` (B, D, T) in the same launch."""
        if not (np.isscalar(w) or torch.is_tensor(w)):
            w = float(w)
        if base is None:
            return ops.position_encoding(x, self.weight, w, True)
        return ops.add_position_encoding(base, x, self.weight, w, True)


def Linear(in_features, out_features, dropout=0):
    """Weight-normalized Linear layer (input: N x T x C) -- modules.py:80-85."""
    m = _conv.Linear(in_features, out_features)
    m.weight.data.normal_(mean=0, std=math.sqrt((1 - dropout) / in_features))
    m.bias.data.zero_()
    return m.apply_weight_norm_()


def Embedding(num_embeddings, embedding_dim, padding_idx, std=0.01):
    m = nn.Embedding(num_embeddings, embedding_dim, padding_idx=padding_idx)
    m.weight.data.normal_(0, std)
    return m


def Conv1d(in_channels, out_channels, kernel_size, dropout=0, std_mul=4.0, **kwargs):
    """Weight-normalized Conv1d, init as modules.py:94-100."""
    m = _conv.Conv1d(in_channels, out_channels, kernel_size, **kwargs)
    std = math.sqrt((std_mul * (1.0 - dropout)) / (m.kernel_size[0] * in_channels))
    m.weight.data.normal_(mean=0, std=std)
    m.bias.data.zero_()
    return m.apply_weight_norm_()


def ConvTranspose1d(in_channels, out_channels, kernel_size, dropout=0, std_mul=1.0, **kwargs):
    """Weight-normalized ConvTranspose1d, init as modules.py:103-109 (norm per input channel)."""
    m = _conv.ConvTranspose1d(in_channels, out_channels, kernel_size, **kwargs)
    std = math.sqrt((std_mul * (1.0 - dropout)) / (m.kernel_size[0] * in_channels))
    m.weight.data.normal_(mean=0, std=std)
    m.bias.data.zero_()
    return m.apply_weight_norm_()


def _site(module):
    return getattr(module, "_dv3_site", None)


class _GatedConv(nn.Module):
    """Shared body of Conv1dGLU / HighwayConv1d: dropout -> dilated conv -> (trim) -> gate ->
    residual, ONE tap-GEMM launch (modules.py:145-164 / 205-226)."""

    mode = ops.EPI_GLU

    def _layer_cfg(self, residual=None):
        k, d = self.conv.kernel_size[0], self.conv.dilation[0]
        return ops.LayerCfg(k=k, dil=d, causal=self.causal, mode=self.mode,
                            residual=self._fused_residual() if residual is None else residual,
                            p=self.dropout, training=self.training, site=_site(self))

    def _check_padding(self):
        k, d = self.conv.kernel_size[0], self.conv.dilation[0]
        want = (k - 1) * d if self.causal else (k - 1) // 2 * d
        if self.conv.padding[0] != want or (not self.causal and k % 2 == 0):
            raise ValueError("only the reference's default padding is supported "
                             "(causal: (k-1)*d, else (k-1)//2*d with odd k)")

    def _run(self, x, spk_bias, residual=None):
        v, g = self.conv.wn_params()
        Cg = self.conv.out_channels // 2
        return ops.conv_layer(x, v, g, self.conv.bias, self._layer_cfg(residual), spk=spk_bias,
                              packed=self.conv.packed(glu_cg=Cg))

    def clear_buffer(self):
        self.conv.clear_buffer()


class Conv1dGLU(_GatedConv):
    """(Dilated) Conv1d + Gated linear unit + (optionally) speaker embedding (modules.py:112-167)."""

    def __init__(self, n_speakers, speaker_embed_dim, in_channels, out_channels, kernel_size,
                 dropout, padding=None, dilation=1, causal=False, residual=False, *args, **kwargs):
        super(Conv1dGLU, self).__init__()
        self.dropout = dropout
        self.residual = residual
        if padding is None:
            padding = (kernel_size - 1) * dilation if causal else (kernel_size - 1) // 2 * dilation
        self.causal = causal
        self.conv = Conv1d(in_channels, 2 * out_channels, kernel_size, dropout=dropout,
                           padding=padding, dilation=dilation, *args, **kwargs)
        self._check_padding()
        if residual and in_channels != out_channels:
            raise ValueError("residual Conv1dGLU needs in_channels == out_channels")
        self.speaker_proj = Linear(speaker_embed_dim, out_channels) if n_speakers > 1 else None

    def _fused_residual(self):
        return self.residual

    def speaker_bias(self, speaker_embed):
        """softsign(speaker_proj(speaker_embed)) -> (B, C) [2-D embed] or (B, C, T) [B,T,E embed]."""
        if self.speaker_proj is None:
            return None
        pre = getattr(speaker_embed, "_dv3_block_bias", None)     # the block computed its layers' biases in one launch
        if pre is not None and id(self) in pre:
            return pre[id(self)]
        if speaker_embed.dim() == 2:
            e = speaker_embed.unsqueeze(-1)                                   # (B, E, 1)
            return self.speaker_proj.forward_bct(e.contiguous(), ops.EPI_SOFTSIGN).squeeze(-1)
        e = speaker_embed
        if e.stride(1) == 0:      # time-expanded view of a (B,E) embedding: constant over time
            return self.speaker_proj.forward_bct(e[:, :1, :].transpose(1, 2).contiguous(),
                                                 ops.EPI_SOFTSIGN).squeeze(-1)
        return self.speaker_proj.forward_bct(e.transpose(1, 2).contiguous(), ops.EPI_SOFTSIGN)

    def forward(self, x, speaker_embed=None):
        return self._run(x, self.speaker_bias(speaker_embed) if self.speaker_proj is not None else None)

    def _run_with_residual(self, x, speaker_embed=None):
        """GLU followed by the caller's `(y + x) * sqrt(0.5)` (decoder, deepvoice3.py:348-349),
        which is exactly this layer's fused residual epilogue."""
        if self.conv.in_channels != self.conv.out_channels // 2:
            raise ValueError("fused outer residual needs in_channels == out_channels")
        return self._run(x, self.speaker_bias(speaker_embed) if self.speaker_proj is not None else None,
                         residual=True)

    def incremental_forward(self, x, speaker_embed=None):
        """x (B, 1, C) -> (B, 1, C)."""
        spk = None
        if self.speaker_proj is not None:
            se = speaker_embed if speaker_embed.dim() == 2 else speaker_embed[:, 0, :]
            spk = self.speaker_bias(se).contiguous()
        r = x[:, -1, :].unsqueeze(-1).contiguous() if self.residual else None
        return self.conv.incremental_forward(x, _gate=dict(mode=ops.EPI_GLU, spk=spk, r=r,
                                                           residual=self.residual))


class HighwayConv1d(_GatedConv):
    """Weight normalized Conv1d + Highway network, incremental forward supported
    (modules.py:170-229)."""

    def __init__(self, in_channels, out_channels, kernel_size=1, padding=None, dilation=1,
                 causal=False, dropout=0, std_mul=None, glu=False):
        super(HighwayConv1d, self).__init__()
        if std_mul is None:
            std_mul = 4.0 if glu else 1.0
        if padding is None:
            padding = (kernel_size - 1) * dilation if causal else (kernel_size - 1) // 2 * dilation
        self.causal = causal
        self.dropout = dropout
        self.glu = glu
        self.mode = ops.EPI_GLU if glu else ops.EPI_HIGHWAY
        if in_channels != out_channels:
            raise ValueError("HighwayConv1d needs in_channels == out_channels")
        self.conv = Conv1d(in_channels, 2 * out_channels, kernel_size=kernel_size, padding=padding,
                           dilation=dilation, dropout=dropout, std_mul=std_mul)
        self._check_padding()

    def _fused_residual(self):
        return bool(self.glu)   # modules.py:219-221: glu branch adds the residual and scales

    def forward(self, x):
        return self._run(x, None)

    def incremental_forward(self, x):
        r = x[:, -1, :].unsqueeze(-1).contiguous()
        return self.conv.incremental_forward(x, _gate=dict(mode=self.mode, r=r, residual=self.glu))


def key_lengths_i32(lengths, device):
    """`input_lengths` as the int32 device vector the attention kernels read.  A tensor already on
    the device is used as is (no host round trip: keeps the step hipGraph-capturable)."""
    if torch.is_tensor(lengths):
        return lengths.to(device=device, dtype=torch.int32)
    return torch.as_tensor(np.asarray(lengths), dtype=torch.int32).to(device)


def get_mask_from_lengths(memory, memory_lengths):
    """Mask tensor from a list of lengths, True where PADDED (modules.py:232-241)."""
    max_len = max(memory_lengths)
    mask = torch.arange(max_len).expand(memory.size(0), max_len) < torch.as_tensor(
        np.asarray(memory_lengths)).unsqueeze(-1)
    return ~mask.to(memory.device)
