# coding: utf-8
"""Conv1d / Linear / ConvTranspose1d parameter holders backed by the HIP tap-GEMM.

Mirror of the reference's deepvoice3_pytorch/conv.py:7-65 (Conv1d with incremental_forward /
clear_buffer) plus the weight-norm parametrisation nn.utils.weight_norm gives the reference
layers (deepvoice3_pytorch/modules.py:80-109), so `state_dict()` has the same keys and shapes:
`weight_g`, `weight_v`, `bias` (or `weight`, `bias` after make_generation_fast_()).
"""
import math

import torch
from torch import nn

from . import ops


def _norm_except_dim0(v):
    return v.reshape(v.size(0), -1).norm(dim=1).reshape([-1] + [1] * (v.dim() - 1))


class _WNLayer(nn.Module):
    """A weight tensor that is either plain (`weight`) or weight-normed (`weight_g`, `weight_v`),
    packed on demand into the tap-GEMM operand layouts (cached while parameters are unchanged)."""

    transposed = False  # ConvTranspose1d: weight is (in, out, k) and the norm is per INPUT channel

    def _init_weight(self, shape, bias_len, bias=True):
        self.weight = nn.Parameter(torch.empty(*shape))
        self.bias = nn.Parameter(torch.zeros(bias_len)) if bias else None
        self._pack_cache = None

    # -- weight norm (nn.utils.weight_norm / remove_weight_norm semantics, dim=0) --------------
    def apply_weight_norm_(self):
        if "weight" not in self._parameters:
            return self
        w = self._parameters.pop("weight")
        self.weight_g = nn.Parameter(_norm_except_dim0(w.data))
        self.weight_v = nn.Parameter(w.data)
        self._pack_cache = None
        return self

    def remove_weight_norm_(self):
        if "weight_g" not in self._parameters:
            raise ValueError("weight_norm of 'weight' not found in {}".format(self))
        g, v = self._parameters.pop("weight_g"), self._parameters.pop("weight_v")
        self.weight = nn.Parameter((g.data * v.data / _norm_except_dim0(v.data)))
        self._pack_cache = None
        return self

    def wn_params(self):
        """-> (v, g) with g None for a plain weight."""
        if "weight_g" in self._parameters:
            return self.weight_v, self.weight_g
        return self.weight, None

    def effective_weight(self):
        v, g = self.wn_params()
        if g is None:
            return v
        return g * v / _norm_except_dim0(v)

    def packed(self, glu_cg=0, need_bwd=False):
        """Packed operands, cached in eval/no-grad use; None => pack inside the autograd op."""
        v, g = self.wn_params()
        if torch.is_grad_enabled() and (v.requires_grad or (g is not None and g.requires_grad)):
            return None
        key = (v.data_ptr(), v._version, None if g is None else (g.data_ptr(), g._version), glu_cg,
               ops.param_epoch, ops.gemm_precision())
        if self._pack_cache is None or self._pack_cache[0] != key:
            pk = ops.pack_weights(v.detach(), None if g is None else g.detach(), glu_cg=glu_cg,
                                  transposed=self.transposed, need_bwd=False)
            self._pack_cache = (key, pk)
        return self._pack_cache[1]


class Conv1d(_WNLayer):
    """Extended Conv1d for incremental dilated convolutions (reference conv.py:7-65).
    Input/Output: (B, C, T).  stride 1 only (all the reference ever builds)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 bias=True):
        super(Conv1d, self).__init__()
        def one(x):
            return x[0] if isinstance(x, (tuple, list)) else x
        if one(stride) != 1:
            raise ValueError("only stride 1 is supported")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = (one(kernel_size),), (1,)
        self.padding, self.dilation = (one(padding),), (one(dilation),)
        self._init_weight((out_channels, in_channels, self.kernel_size[0]), out_channels, bias)
        bound = 1.0 / math.sqrt(in_channels * self.kernel_size[0])
        nn.init.uniform_(self.weight, -bound, bound)
        if self.bias is not None:
            nn.init.uniform_(self.bias, -bound, bound)
        self.clear_buffer()

    def extra_repr(self):
        return "{}, {}, kernel_size={}, padding={}, dilation={}".format(
            self.in_channels, self.out_channels, self.kernel_size, self.padding, self.dilation)

    def forward(self, x, mode=ops.EPI_LINEAR, r=None, r2=None, out_c8=None):
        """nn.Conv1d forward: symmetric zero padding `self.padding`, output length
        T + 2*pad - dil*(k-1).  x may be a channel-blocked bf16 tensor (ops.to_c8); out_c8 forces the output
        layout (None: like the input)."""
        k, d, pad = self.kernel_size[0], self.dilation[0], self.padding[0]
        T = x.size(2)
        cfg = ops.LayerCfg(k=k, dil=d, mode=mode, out_c8=out_c8)
        cfg.pad_left, cfg.t_out = pad, T + 2 * pad - d * (k - 1)
        v, g = self.wn_params()
        return ops.conv_layer(x, v, g, self.bias, cfg, r=r, r2=r2, packed=self.packed())

    # -- incremental (autoregressive) path: conv.py:17-49 ------------------------------------
    def incremental_forward(self, input, _gate=None):
        """input (B, 1, C) -> (B, 1, out).  Keeps the last k+(k-1)(d-1) frames in a device
        buffer and evaluates the conv at the newest frame with the same MFMA tap-GEMM."""
        if self.training:
            raise RuntimeError('incremental_forward only supports eval mode')
        k, d = self.kernel_size[0], self.dilation[0]
        B = input.size(0)
        x_t = input[:, -1, :]                       # (B, C)
        v, g = self.wn_params()
        if k == 1:
            xb = x_t.reshape(B, self.in_channels, 1)
            lbuf = 1
        else:
            # the last k + (k-1)(d-1) frames, newest last; shifted in place by one HIP launch per step
            # (static addresses: the whole decode step can be captured as a hipGraph)
            lbuf = k + (k - 1) * (d - 1)
            if self.input_buffer is None or self.input_buffer.size(0) != B:
                self.input_buffer = x_t.new_zeros(B, self.in_channels, lbuf)
            xc = x_t if x_t.stride(-1) == 1 and x_t.stride(0) == self.in_channels else x_t.contiguous()
            ops._lib.call("dv3_shift_append_f32", self.input_buffer.data_ptr(), xc.data_ptr(),
                          B * self.in_channels, lbuf, 1, ops._stream())
            xb = self.input_buffer
        with torch.no_grad():
            gate = _gate or {}
            mode = gate.get("mode", ops.EPI_LINEAR)
            gated = mode in (ops.EPI_GLU, ops.EPI_HIGHWAY)
            Cg = self.out_channels // 2 if gated else 0
            pk = self.packed(glu_cg=Cg)
            spk = gate.get("spk")
            y = ops.conv_gemm(xb, pk.fwd, pk.lda, pk.a_half, B=B, Cin=self.in_channels, Tin=lbuf,
                              M=self.out_channels, Tout=1, J=k, dil=d, padL=0, mode=mode, Cg=Cg,
                              bias=self.bias, spk=spk,
                              spk_strides=(spk.stride(0), 1, 0) if spk is not None else (0, 0, 0),
                              r=gate.get("r"), residual=int(gate.get("residual", 0)),
                              x_bs=xb.stride(0), x_rs=xb.stride(1))
        return y.transpose(1, 2)                    # (B, 1, out) view of (B, out, 1)

    def clear_buffer(self):
        """forget the window (reference conv.py:48-49).  An existing buffer is zeroed in place rather
        than dropped, so a captured decode-step graph keeps pointing at live memory."""
        buf = self.__dict__.get("input_buffer")
        if buf is not None:
            buf.zero_()
        else:
            self.input_buffer = None


class Linear(_WNLayer):
    """nn.Linear parameter holder ((out,in) weight).  Input (B, T, C) like the reference's
    Linear (modules.py:80-85); computed as a 1x1 tap-GEMM on the BCT view."""

    def __init__(self, in_features, out_features, bias=True):
        super(Linear, self).__init__()
        self.in_features, self.out_features = in_features, out_features
        self._init_weight((out_features, in_features), out_features, bias)
        bound = 1.0 / math.sqrt(in_features)
        nn.init.uniform_(self.weight, -bound, bound)
        if self.bias is not None:
            nn.init.uniform_(self.bias, -bound, bound)

    def extra_repr(self):
        return "in_features={}, out_features={}".format(self.in_features, self.out_features)

    def forward_bct(self, x, mode=ops.EPI_LINEAR, r=None, r2=None, out_c8=None):
        """x (B, in, T) -> (B, out, T); x / the output may be channel-blocked bf16 tensors (see Conv1d.forward)."""
        cfg = ops.LayerCfg(k=1, dil=1, mode=mode, out_c8=out_c8)
        v, g = self.wn_params()
        return ops.conv_layer(x, v, g, self.bias, cfg, r=r, r2=r2, packed=self.packed())

    def forward(self, x, mode=ops.EPI_LINEAR):
        """(..., in) -> (..., out), reference layout."""
        shp = x.shape
        x3 = x.reshape(-1, shp[-1]) if x.dim() != 3 else x
        if x.dim() != 3:
            y = self.forward_bct(x3.t().unsqueeze(0).contiguous(), mode)   # (1, out, N)
            return y[0].t().reshape(*shp[:-1], self.out_features)
        y = self.forward_bct(x.transpose(1, 2).contiguous(), mode)
        return y.transpose(1, 2)


class ConvTranspose1d(_WNLayer):
    """nn.ConvTranspose1d(kernel_size=2, stride=2, padding=0) -- the only form the reference
    builds (deepvoice3.py:519-520,527-528; nyanko.py:372,377): exact x2 time upsampling."""

    transposed = True

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True):
        super(ConvTranspose1d, self).__init__()
        def one(x):
            return x[0] if isinstance(x, (tuple, list)) else x
        if one(kernel_size) != 2 or one(stride) != 2 or one(padding) != 0:
            raise ValueError("only kernel_size=2, stride=2, padding=0 is supported")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = (2,), (2,), (0,)
        self._init_weight((in_channels, out_channels, 2), out_channels, bias)
        bound = 1.0 / math.sqrt(out_channels * 2)
        nn.init.uniform_(self.weight, -bound, bound)
        if self.bias is not None:
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x):
        cfg = ops.LayerCfg(k=2, dil=1, mode=ops.EPI_LINEAR, transposed=True)
        v, g = self.wn_params()
        return ops.conv_layer(x, v, g, self.bias, cfg, packed=self.packed())
