# coding: utf-8
"""Round 5: is the 256 x 256 tap-GEMM power-limited or latency-limited?  Same launch, same instruction stream, operands
that toggle fewer multiplier bits: all-zero activations / weights, constant activations.  If the launch time follows the
operand values the limit is the power budget (clock), not the schedule."""
import math, os, sys, time, threading
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepvoice3_pytorch_amd import ops, _lib  # noqa: E402
dev = torch.device("cuda:0")
L = _lib.lib()

def sclk():
    try:
        for line in open("/sys/class/drm/card0/device/pp_dpm_sclk"):
            if "*" in line:
                return line.split(":")[1].strip().split("Mhz")[0]
    except Exception:
        pass
    import glob
    for p in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        for line in open(p):
            if "*" in line:
                return line.split(":")[1].strip()
    return "?"

def timeit(fn, iters=200, settle=50):
    for _ in range(settle):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    samples = []
    stop = [False]
    def sampler():
        while not stop[0]:
            samples.append(sclk()); time.sleep(0.002)
    th = threading.Thread(target=sampler); th.start()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    stop[0] = True; th.join()
    vals = []
    for s in samples:
        try: vals.append(float(str(s).lower().replace("mhz", "")))
        except Exception: pass
    return e0.elapsed_time(e1) * 1e3 / iters, (sum(vals) / len(vals) if vals else float("nan"))

ops.set_gemm_precision("f16x3")
B, C, T, k = 64, 256, 1024, 3
torch.manual_seed(0)
v = torch.randn(2 * C, C, k, device=dev) * math.sqrt(4.0 * 0.95 / (k * C))
g = v.reshape(2 * C, -1).norm(dim=1).view(-1, 1, 1).clone()
bias = torch.zeros(2 * C, device=dev)
pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=False)
pk0 = ops.pack_weights(v * 0 + 1e-30, g, glu_cg=C, need_bwd=False)     # ~zero weights (normalised: still unit rows!)
y = torch.empty(B, C, T, device=dev)
xs = {"randn": torch.randn(B, C, T, device=dev), "zeros": torch.zeros(B, C, T, device=dev),
      "ones": torch.ones(B, C, T, device=dev), "randn*100": torch.randn(B, C, T, device=dev) * 100}
for name, x in xs.items():
    for wname, p in (("randn w", pk),):
        kw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=1, padL=1, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x,
                  residual=1, a_split=p.fwd_s, y=y, tile_hint=30)
        for rnd in range(2):
            t, f = timeit(lambda: ops.conv_gemm(x, None, p.lda, p.a_half, **kw))
            print("x = %-10s %s: %.1f us   avg sclk %.0f MHz" % (name, wname, t, f), flush=True)
# zero weights: overwrite the split image itself
z = pk.fwd_s.clone(); z.zero_()
for name in ("randn", "zeros"):
    x = xs[name]
    kw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=1, padL=1, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x,
              residual=1, a_split=z, y=y, tile_hint=30)
    t, f = timeit(lambda: ops.conv_gemm(x, None, pk.lda, pk.a_half, **kw))
    print("x = %-10s zero weight image: %.1f us   avg sclk %.0f MHz" % (name, t, f), flush=True)
if os.environ.get("DV3_LIBPATH", "").endswith("_exp.so"):
    for name in ("randn", "zeros"):
        x = xs[name]
        kw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=1, padL=1, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x,
                  residual=1, a_split=(pk.fwd_s if name == "randn" else z), y=y, tile_hint=30)
        for abl, an in ((0, "full"), (1, "no MFMAs"), (2, "no staging"), (3, "no tail")):
            L.dv3_debug_set(13, abl)
            t, f = timeit(lambda: ops.conv_gemm(x, None, pk.lda, pk.a_half, **kw))
            print("x = %-6s (w %s) ablation %-12s: %.1f us   avg sclk %.0f MHz" % (name, "randn" if name == "randn" else "zero", an, t, f), flush=True)
        L.dv3_debug_set(13, 0)
