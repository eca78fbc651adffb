# coding: utf-8
"""bf16 storage ("c8") bring-up checks on the GPU: converters, keep-bytes, every layer form forward + backward against
the fp32-storage bf16 path and the exact f32 mode, toy models end to end, timings at the north-star shape."""
import math, os, sys, time, traceback
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepvoice3_pytorch_amd import ops, modules, builder, _lib
from tests.util import rel_err

dev = torch.device("cuda:0")
ok_all = True


def report(name, ok, msg=""):
    global ok_all
    ok_all &= bool(ok)
    print("%-64s %s %s" % (name, "ok  " if ok else "FAIL", msg), flush=True)


def section(f):
    try:
        f()
    except Exception:
        report(f.__name__, False, "exception")
        traceback.print_exc()


def t_converters():
    for (B, C, T) in ((3, 40, 37), (2, 64, 150), (2, 320, 50)):
        x = torch.randn(B, C, T, device=dev)
        x8 = ops.to_c8(x)
        back = ops.from_c8(x8, C)
        report("to_c8/from_c8 roundtrip %s" % ((B, C, T),), torch.equal(back, x.to(torch.bfloat16).float()))
        if C % 32:
            report("  padding channels are zero", float(x8[:, C // 8:].abs().max()) == 0.0 if C % 8 == 0 else True)
        bits, rs = ops.dropout_bits(B * C, T, 0.3, dev)
        k8 = ops.mask_bits_to_c8(bits, rs, B, C, T).cpu().numpy()
        w = bits.cpu().numpy().view(np.uint32).reshape(B * C, rs)
        keep = ((w[:, :, None] >> np.arange(32)[None, None, :]) & 1).reshape(B * C, rs * 32)[:, :T].reshape(B, C, T)
        want = np.zeros((B, ops.c8_groups(C), T), np.uint8)
        for e in range(8):
            sel = keep[:, e::8, :]
            want[:, :sel.shape[1], :] |= (sel << e).astype(np.uint8)
        report("mask_bits_to_c8 %s" % ((B, C, T),), np.array_equal(k8, want))


def run_layer(layer, x, spk, storage, mode, seed=3):
    ops.set_gemm_precision(mode)
    ops.bf16_storage = storage
    for p in layer.parameters():
        p.grad = None
    xin = x.clone().requires_grad_(True)
    ops.dropout_state.manual_seed(seed)
    xi = ops.to_c8(xin) if (storage and mode == "bf16") else xin
    y = layer(xi, spk) if spk is not None else layer(xi)
    y = ops.from_c8(y) if ops.is_c8(y) else y
    w = torch.linspace(-1, 1, y.numel(), device=dev).view_as(y)
    (y * w).sum().backward()
    return y.detach(), xin.grad.detach(), {k: p.grad.detach().clone() for k, p in layer.named_parameters()}


def t_layers():
    cases = []
    for (C, k, d, causal, T, B) in ((64, 3, 2, False, 75, 3), (40, 3, 1, True, 37, 2), (256, 3, 27, False, 150, 2), (128, 1, 1, False, 50, 3)):
        cases.append(("Conv1dGLU C=%d k=%d d=%d causal=%d T=%d" % (C, k, d, causal, T),
                      lambda C=C, k=k, d=d, causal=causal: modules.Conv1dGLU(1, 16, C, C, k, dropout=0.2, dilation=d, causal=causal, residual=True), C, T, B))
        cases.append(("Conv1dGLU(no residual) C=%d k=%d d=%d" % (C, k, d),
                      lambda C=C, k=k, d=d, causal=causal: modules.Conv1dGLU(1, 16, C, C, k, dropout=0.2, dilation=d, causal=causal, residual=False), C, T, B))
        cases.append(("HighwayConv1d C=%d k=%d d=%d causal=%d" % (C, k, d, causal),
                      lambda C=C, k=k, d=d, causal=causal: modules.HighwayConv1d(C, C, k, dilation=d, causal=causal, dropout=0.1), C, T, B))
    for name, mk, C, T, B in cases:
        torch.manual_seed(0)
        layer = mk().to(dev).train()
        with torch.no_grad():
            layer.conv.bias.uniform_(-0.1, 0.1)
        x = torch.randn(B, C, T, device=dev)
        ref = run_layer(layer, x, None, False, "f32")
        b32 = run_layer(layer, x, None, False, "bf16")
        c8 = run_layer(layer, x, None, True, "bf16")
        e_y = rel_err(c8[0].cpu(), ref[0].cpu()); e_yb = rel_err(b32[0].cpu(), ref[0].cpu())
        e_x = rel_err(c8[1].cpu(), ref[1].cpu()); e_xb = rel_err(b32[1].cpu(), ref[1].cpu())
        e_p = max(rel_err(c8[2][k_].cpu(), ref[2][k_].cpu()) for k_ in ref[2])
        e_pb = max(rel_err(b32[2][k_].cpu(), ref[2][k_].cpu()) for k_ in ref[2])
        good = e_y < 3e-2 and e_x < 5e-2 and e_p < 5e-2
        report(name, good, "y %.1e (fp32-storage bf16 %.1e) dx %.1e (%.1e) dparams %.1e (%.1e) wgrad variant %d"
               % (e_y, e_yb, e_x, e_xb, e_p, e_pb, _lib.lib().dv3_debug_get(11)))


def t_plain_layers():
    from deepvoice3_pytorch_amd import conv as _conv
    B, T = 3, 61
    # ReLU: its backward follows the forward's own y > 0 decisions -- compared against torch on those decisions
    torch.manual_seed(4)
    f = modules.Conv1d(64, 128, 1, dropout=0.0).to(dev).train()
    ops.set_gemm_precision("bf16"); ops.bf16_storage = True
    x = torch.randn(B, 64, T, device=dev)
    xin = x.clone().requires_grad_(True)
    y = ops.from_c8(f(ops.to_c8(xin), mode=ops.EPI_RELU, out_c8=True))
    wgt = torch.linspace(-1, 1, y.numel(), device=dev).view_as(y)
    (y * wgt).sum().backward()
    W = f.effective_weight().detach()[:, :, 0]
    dpre = wgt.to(torch.bfloat16).float() * (y.detach() > 0)
    xb = x.to(torch.bfloat16).float()
    e = (rel_err(y.detach().cpu(), torch.relu(torch.einsum("oi,bit->bot", W, xb) + f.bias.detach()[None, :, None]).cpu()),
         rel_err(xin.grad.cpu(), torch.einsum("oi,bot->bit", W, dpre).cpu()),
         rel_err(f.bias.grad.cpu(), dpre.sum((0, 2)).cpu()))
    report("Conv1d 1x1 + ReLU c8->c8 (vs torch on the same decisions)", max(e) < 2e-2, "y %.1e dx %.1e dbias %.1e" % e)
    for (Ci, Co, mode, name) in ((64, 128, ops.EPI_SOFTSIGN, "Conv1d 1x1 + softsign c8->c8"), (128, 64, ops.EPI_LINEAR, "Conv1d 1x1 c8->c8"),
                                 (64, 513, ops.EPI_SIGMOID, "Conv1d 1x1 + sigmoid c8->fp32 (513 ch)")):
        torch.manual_seed(1)
        f = modules.Conv1d(Ci, Co, 1, dropout=0.1).to(dev).train()
        x = torch.randn(B, Ci, T, device=dev)
        outs = {}
        for tag, storage, gm in (("ref", False, "f32"), ("c8", True, "bf16")):
            ops.set_gemm_precision(gm); ops.bf16_storage = storage
            for p in f.parameters():
                p.grad = None
            xin = x.clone().requires_grad_(True)
            c8 = storage and gm == "bf16"
            xi = ops.to_c8(xin) if c8 else xin
            y = f(xi, mode=mode, out_c8=(Co % 8 == 0) if c8 else None)
            y = ops.from_c8(y) if ops.is_c8(y) else y
            (y * torch.linspace(-1, 1, y.numel(), device=dev).view_as(y)).sum().backward()
            outs[tag] = (y.detach(), xin.grad.detach(), {k_: p.grad.detach().clone() for k_, p in f.named_parameters()})
        r, c = outs["ref"], outs["c8"]
        e = (rel_err(c[0].cpu(), r[0].cpu()), rel_err(c[1].cpu(), r[1].cpu()), max(rel_err(c[2][k_].cpu(), r[2][k_].cpu()) for k_ in r[2]))
        report(name, max(e) < 5e-2, "y %.1e dx %.1e dparams %.1e" % e)
    # fp32 input, c8 output with two c8 residuals (the attention out-projection)
    torch.manual_seed(2)
    lin = modules.Linear(64, 128).to(dev).train()
    x = torch.randn(B, 64, T, device=dev); r1 = torch.randn(B, 128, T, device=dev); r2 = torch.randn(B, 128, T, device=dev)
    outs = {}
    for tag, storage, gm in (("ref", False, "f32"), ("c8", True, "bf16")):
        ops.set_gemm_precision(gm); ops.bf16_storage = storage
        for p in lin.parameters():
            p.grad = None
        xin = x.clone().requires_grad_(True); a = r1.clone().requires_grad_(True); b = r2.clone().requires_grad_(True)
        c8 = storage and gm == "bf16"
        y = lin.forward_bct(xin, r=ops.to_c8(a) if c8 else a, r2=ops.to_c8(b) if c8 else b, out_c8=True if c8 else None)
        y = ops.from_c8(y) if ops.is_c8(y) else y
        (y * torch.linspace(-1, 1, y.numel(), device=dev).view_as(y)).sum().backward()
        outs[tag] = (y.detach(), xin.grad.detach(), a.grad.detach(), b.grad.detach(), {k_: p.grad.detach().clone() for k_, p in lin.named_parameters()})
    r, c = outs["ref"], outs["c8"]
    e = [rel_err(c[i].cpu(), r[i].cpu()) for i in range(4)] + [max(rel_err(c[4][k_].cpu(), r[4][k_].cpu()) for k_ in r[4])]
    report("Linear fp32 -> c8 with r, r2", max(e) < 5e-2, "y %.1e dx %.1e dr %.1e dr2 %.1e dparams %.1e" % tuple(e))


def t_models():
    import bench
    from deepvoice3_pytorch_amd import train_step
    for preset in ("deepvoice3_ljspeech", "nyanko_ljspeech", "deepvoice3_vctk"):
        bname, hp, sigma = bench.PRESETS[preset]
        hp = dict(hp)
        res = {}
        for tag, storage, gm in (("f32", False, "f32"), ("bf16", False, "bf16"), ("c8", True, "bf16")):
            ops.set_gemm_precision(gm); ops.bf16_storage = storage
            torch.manual_seed(5)
            model = getattr(builder, bname)(**hp).to(dev)
            rng = np.random.RandomState(3)
            bt = bench.synth_batch(rng, 2, 60, 160, hp, fixed=True)
            spk = torch.from_numpy(rng.randint(0, hp["n_speakers"], 2)) if hp["n_speakers"] > 1 else None
            cfg = train_step.TrainConfig(max_positions=hp["max_positions"], guided_attention_sigma=sigma)
            tr = train_step.Trainer(model, cfg)
            batch = train_step.Batch.from_collate(bt["text"], bt["input_lengths"], bt["mel"], bt["y"], bt["text_positions"],
                                                  bt["frame_positions"], bt["done"], bt["target_lengths"], spk, downsample_step=4, device=dev)
            ops.dropout_state.manual_seed(9)
            tr.arena.grad.zero_()
            scal = tr.forward_backward(batch)
            res[tag] = (float(scal["loss"]), tr.arena.grad.detach().clone())
            tr.close()
        cos = lambda a, b: float((a.double() * b.double()).sum() / (a.double().norm() * b.double().norm()))
        l32, g32 = res["f32"]
        report("train step %s" % preset, abs(res["c8"][0] - l32) < 3e-2 * abs(l32) and cos(res["c8"][1], g32) > 0.99,
               "loss f32 %.5f bf16 %.5f c8 %.5f | grad cos vs f32: bf16 %.5f c8 %.5f" % (l32, res["bf16"][0], res["c8"][0], cos(res["bf16"][1], g32), cos(res["c8"][1], g32)))


def timeit(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1000.0 / n


def t_timing():
    B, C, T, k = 64, 256, 1024, 3
    torch.manual_seed(0)
    layer = modules.Conv1dGLU(1, 16, C, C, k, dropout=0.05, dilation=3, residual=True).to(dev).train()
    x = torch.randn(B, C, T, device=dev)
    ops.set_gemm_precision("bf16")
    for storage in (False, True):
        ops.bf16_storage = storage
        xi = (ops.to_c8(x) if storage else x).detach().requires_grad_(True)
        with torch.no_grad():
            layer.eval(); te = timeit(lambda: layer(xi)); layer.train()
        y = layer(xi)
        gy = torch.ones_like(y)
        def fb():
            y = layer(xi)
            y.backward(gy)
        tf = timeit(lambda: layer(xi))
        tfb = timeit(fb)
        print("north-star Conv1dGLU (B=64, 256 ch, T=1024, k=3, d=3), bf16 GEMM mode, %s storage: eval fwd %.1f us, train fwd %.1f us, fwd+bwd %.1f us (conv variant %d, wgrad variant %d)"
              % ("c8 bf16" if storage else "fp32", te, tf, tfb, _lib.lib().dv3_debug_get(10), _lib.lib().dv3_debug_get(11)), flush=True)


prev_mode, prev_storage = ops.gemm_precision(), ops.bf16_storage
try:
    for f in (t_converters, t_layers, t_plain_layers, t_models, t_timing):
        section(f)
finally:
    ops.set_gemm_precision(prev_mode)
    ops.bf16_storage = prev_storage
print("ALL OK" if ok_all else "SOME FAILED")
