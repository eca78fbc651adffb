# coding: utf-8
"""A/B at the north-star shape (Conv1dGLU fwd, B=64 x 256 x 1024, k=3): the ping-pong tap-GEMM that splits while
staging against the persistent planes kernel (tiles 1 / 9 / 2, stagger settings), training-mode epilogue (pre-gate
save) and eval; bit-equality of the two; the DGRAD form; the stand-alone split pass."""
import math, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepvoice3_pytorch_amd import ops, _lib

dev = torch.device("cuda:0")
B, C, T, k = 64, 256, 1024, 3
torch.manual_seed(0)
x = torch.randn(B, C, T, device=dev)
v = torch.randn(2 * C, C, k, device=dev) * math.sqrt(4.0 * 0.95 / (k * C))
g = v.reshape(2 * C, -1).norm(dim=1).view(-1, 1, 1).clone()
bias = torch.zeros(2 * C, device=dev)
lib = _lib.lib()


def timeit(fn, iters=50, settle=30):
    for _ in range(settle):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
  for mode in ("f16x3", "bf16x3", "bf16"):
      ops.set_gemm_precision(mode)
      pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=True)
      f16 = pk.fwd_f16
      for d in (1, 27):
          for train in (False, True):
              bits = rs = None
              dscale = 1.0
              if train:
                  ops.dropout_state.manual_seed(3)
                  bits, rs = ops.dropout_bits(B * C, T, 0.05, dev)
                  dscale = 1 / 0.95
              y0 = torch.empty(B, C, T, device=dev)
              y1 = torch.empty(B, C, T, device=dev)
              ab0 = torch.empty(B, 2 * C, T, device=dev) if train else None
              ab1 = torch.empty(B, 2 * C, T, device=dev) if train else None
              kw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=d, padL=d, mode=ops.EPI_GLU, Cg=C, bias=bias,
                        r=x, residual=1, a_split=pk.fwd_s)

              def old():
                  ops.conv_gemm(x, None, pk.lda, pk.a_half, y=y0, ab=ab0, xmask=bits, xmask_rs=rs or 0, drop_scale=dscale, **kw)
              xp = ops.split_planes(x, bits, rs or 0, dscale, f16=f16)

              def new():
                  ops.conv_gemm(x, None, pk.lda, pk.a_half, y=y1, ab=ab1, x_planes=xp, **kw)
              t_old = timeit(old)
              v_old = lib.dv3_debug_get(10)
              res = []
              for tile, stag in ((1, -1), (1, 0), (9, -1), (2, -1)):
                  lib.dv3_debug_set(4, tile)
                  lib.dv3_debug_set(5, stag)
                  y1.zero_()
                  t_new = timeit(new)
                  diff = float((y1 - y0).abs().max())
                  if train:
                      diff = max(diff, float((ab1 - ab0).abs().max()))
                  res.append("tile%d/stag%d %.1f us (%d) diff %.1e" % (tile, stag, t_new, lib.dv3_debug_get(10), diff))
              lib.dv3_debug_set(4, 0)
              lib.dv3_debug_set(5, -1)
              t_split = timeit(lambda: ops.split_planes(x, bits, rs or 0, dscale, f16=f16))
              print("%s d=%d train=%d | old %.1f us (%d) | %s | split pass %.1f us" % (mode, d, train, t_old, v_old, " | ".join(res), t_split))
      # DGRAD form: gmat (B, 2C, T) -> dx (B, C, T), dropout mask on the output side, residual-gradient addend
      gm = torch.randn(B, 2 * C, T, device=dev)
      dres = torch.randn(B, C, T, device=dev)
      ops.dropout_state.manual_seed(4)
      bits, rs = ops.dropout_bits(B * C, T, 0.05, dev)
      dx0, dx1 = torch.empty(B, C, T, device=dev), torch.empty(B, C, T, device=dev)
      kw = dict(B=B, Cin=2 * C, Tin=T, M=C, Tout=T, J=k, dil=1, padL=1, mode=ops.EPI_DGRAD, r=dres, ymask=bits, ymask_rs=rs,
                drop_scale=1 / 0.95, a_split=pk.bwd_s)
      gp = ops.split_planes(gm, f16=False)
      t_old = timeit(lambda: ops.conv_gemm(gm, None, pk.ldb, 0, y=dx0, **kw))
      res = []
      for tile in (1, 9):
          lib.dv3_debug_set(4, tile)
          t_new = timeit(lambda: ops.conv_gemm(gm, None, pk.ldb, 0, y=dx1, x_planes=gp, **kw))
          res.append("tile%d %.1f us diff %.1e" % (tile, t_new, float((dx1 - dx0).abs().max())))
      lib.dv3_debug_set(4, 0)
      print("%s DGRAD | old %.1f us | %s | split pass %.1f us" % (mode, t_old, " | ".join(res), timeit(lambda: ops.split_planes(gm, f16=False))))


if __name__ == "__main__":
    main()
