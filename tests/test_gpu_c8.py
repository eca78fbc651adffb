# coding: utf-8
"""-m gpu: the bf16-storage path of the bf16 GEMM mode (BASELINE configs 3/4: "bf16 activations ... fp32 accum").
Activations between the layers of a conv stack are channel-blocked bf16 tensors ("c8", include/dv3hip.h); every
layer form is run forward + backward on c8 tensors and compared with
  * the oracle (oracle/dv3_oracle.py) evaluated in fp32 -- tolerance 2e-2 of the tensor's max: the operands AND the
    stored activations carry 8 significand bits (measured 3.5e-3 .. 5e-3; a wrong channel / frame mapping gives O(1));
  * the exact-fp32 HIP mode (itself held to 1e-4 against the oracle in test_gpu_kernels.py) for the gradients, 5e-2.
Converters and the keep-byte form of the dropout mask are bit-exact checks."""
import numpy as np
import pytest
import torch

from oracle import dv3_oracle as O
from tests.util import rel_err

pytestmark = pytest.mark.gpu
TOL_FWD, TOL_GRAD = 2e-2, 5e-2


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture()
def modes():
    """restore the GEMM mode and the storage switch after the test"""
    from deepvoice3_pytorch_amd import ops
    prev = (ops.gemm_precision(), ops.bf16_storage)
    yield ops
    ops.set_gemm_precision(prev[0])
    ops.bf16_storage = prev[1]


@pytest.mark.parametrize("B,C,T", [(3, 40, 37), (2, 64, 150), (2, 320, 50), (1, 8, 1), (2, 513, 33), (2, 3, 70)])
def test_c8_converters_and_keep_bytes(dev, modes, B, C, T):
    ops = modes
    x = torch.randn(B, C, T, device=dev)
    x8 = ops.to_c8(x)
    assert tuple(x8.shape) == (B, (C + 31) // 32 * 4, T, 8) and x8.dtype == torch.bfloat16
    assert torch.equal(ops.from_c8(x8, C), x.to(torch.bfloat16).float())        # round to nearest even, exact back
    ref = torch.zeros(B, x8.shape[1] * 8, T, device=dev)
    ref[:, :C] = x
    assert torch.equal(x8.float(), ref.view(B, -1, 8, T).permute(0, 1, 3, 2).to(torch.bfloat16).float())
    bits, rs = ops.dropout_bits(B * C, T, 0.3, dev)
    k8 = ops.mask_bits_to_c8(bits, rs, B, C, T).cpu().numpy()
    keep = O.unpack_keep_bits(bits.cpu().numpy().view(np.uint32), B * C, rs, T).reshape(B, C, T).astype(np.uint8)
    want = np.zeros((B, x8.shape[1], T), np.uint8)
    for e in range(8):
        sel = keep[:, e::8, :]
        want[:, :sel.shape[1], :] |= (sel << e).astype(np.uint8)
    assert np.array_equal(k8, want)
    # the one-launch Philox -> keep-bytes form draws the same decisions as dropout_bits for the same seed / site
    ops.dropout_state.manual_seed(21)
    direct = ops.dropout_keep_c8(B, C, T, 0.3, dev)
    ops.dropout_state.manual_seed(21)
    b2, rs2 = ops.dropout_bits(B * C, T, 0.3, dev)
    assert torch.equal(direct, ops.mask_bits_to_c8(b2, rs2, B, C, T))
    # gradients of the converters are the converters
    xin = x.clone().requires_grad_(True)
    w = torch.randn(B, C, T, device=dev)
    (ops.from_c8(ops.to_c8(xin), C) * w).sum().backward()
    assert torch.equal(xin.grad, w.to(torch.bfloat16).float())


def _run(layer, x, storage, mode, ops, seed=3, record=None):
    ops.set_gemm_precision(mode)
    ops.bf16_storage = storage
    for p in layer.parameters():
        p.grad = None
    xin = x.clone().requires_grad_(True)
    ops.dropout_state.manual_seed(seed)
    ops.dropout_state.record = record
    try:
        c8 = storage and mode == "bf16"
        y = layer(ops.to_c8(xin) if c8 else xin)
        assert ops.is_c8(y) == c8
        y = ops.from_c8(y) if c8 else y
        w = torch.linspace(-1, 1, y.numel(), device=x.device).view_as(y)
        (y * w).sum().backward()
    finally:
        ops.dropout_state.record = None
    return y.detach(), xin.grad.detach(), {k: p.grad.detach().clone() for k, p in layer.named_parameters()}


GATED = [("glu", 64, 3, 2, False, 75, 3), ("glu", 40, 3, 1, True, 37, 2), ("glu", 256, 3, 27, False, 150, 2),
         ("glu", 128, 1, 1, False, 50, 3), ("glu_nores", 64, 3, 3, False, 61, 2), ("highway", 64, 3, 2, False, 75, 3),
         ("highway", 40, 3, 1, True, 37, 2), ("highway", 128, 1, 1, False, 50, 3)]


@pytest.mark.parametrize("kind,C,k,d,causal,T,B", GATED)
def test_c8_gated_layers_forward_backward(dev, modes, kind, C, k, d, causal, T, B):
    """Conv1dGLU (modules.py:112-167) / HighwayConv1d (modules.py:170-229), dropout on, c8 in and out"""
    ops = modes
    from deepvoice3_pytorch_amd import modules, _lib
    torch.manual_seed(0)
    if kind == "highway":
        layer = modules.HighwayConv1d(C, C, k, dilation=d, causal=causal, dropout=0.1)
    else:
        layer = modules.Conv1dGLU(1, 16, C, C, k, dropout=0.2, dilation=d, causal=causal, residual=(kind == "glu"))
    layer = layer.to(dev).train()
    layer._dv3_site = "l"
    with torch.no_grad():
        layer.conv.bias.uniform_(-0.1, 0.1)
    x = torch.randn(B, C, T, device=dev)
    rec = {}
    y8, dx8, dp8 = _run(layer, x, True, "bf16", ops, record=rec)
    assert _lib.lib().dv3_debug_get(10) // 1000 == 8            # the planes kernel, single-term bf16
    # the c8 wgrad kernel (+20: two-steps-ahead fetch, +40: staging between the MFMAs; 51xx / 53xx: the transposing-read forms)
    assert _lib.lib().dv3_debug_get(11) in (5001, 5003, 5021, 5023, 5063, 5101, 5103, 5143, 5301, 5303, 5343)
    # forward against the oracle with the recorded keep-bits
    bits, rows, Tm = rec["l"]
    keep = torch.from_numpy(O.unpack_keep_bits(bits.cpu().numpy().view(np.uint32), rows, (Tm + 31) // 32, Tm)).float()
    p = layer.dropout
    sd = {"l.conv." + n: v.detach().cpu() for n, v in layer.conv.state_dict().items()}
    def drop(site, t, p_, layout):
        return t * keep.view(t.shape) / (1 - p_)
    if kind == "highway":
        want = O.highway_conv1d(sd, "l", x.cpu(), k, d, causal, p=p, drop=drop)
    else:
        want = O.conv1d_glu(sd, "l", x.cpu(), k, d, causal, kind == "glu", p=p, drop=drop)
    assert rel_err(y8.cpu(), want) < TOL_FWD
    yr, dxr, dpr = _run(layer, x, False, "f32", ops)       # same seed -> same keep-bits
    assert rel_err(yr.cpu(), want) < 1e-4
    assert rel_err(dx8.cpu(), dxr.cpu()) < TOL_GRAD
    for n in dpr:
        assert rel_err(dp8[n].cpu(), dpr[n].cpu()) < TOL_GRAD, n


def test_c8_plain_layers(dev, modes):
    """1x1 Conv1d / Linear forms around the gated layers: c8 -> c8 (+ ReLU, on the forward's own decisions),
    c8 -> fp32 with a channel count that has no c8 form (513 linear bins), fp32 -> c8 with two c8 residuals (the
    attention out-projection, deepvoice3.py:175 + 348-349)"""
    ops = modes
    from deepvoice3_pytorch_amd import modules
    B, T = 3, 61
    torch.manual_seed(4)
    f = modules.Conv1d(64, 128, 1, dropout=0.0).to(dev).train()
    ops.set_gemm_precision("bf16")
    ops.bf16_storage = True
    x = torch.randn(B, 64, T, device=dev)
    xin = x.clone().requires_grad_(True)
    y = ops.from_c8(f(ops.to_c8(xin), mode=ops.EPI_RELU, out_c8=True))
    wgt = torch.linspace(-1, 1, y.numel(), device=dev).view_as(y)
    (y * wgt).sum().backward()
    W = f.effective_weight().detach()[:, :, 0]
    dpre = wgt.to(torch.bfloat16).float() * (y.detach() > 0)
    xb = x.to(torch.bfloat16).float()
    assert rel_err(y.detach().cpu(), torch.relu(torch.einsum("oi,bit->bot", W, xb) + f.bias.detach()[None, :, None]).cpu()) < TOL_FWD
    assert rel_err(xin.grad.cpu(), torch.einsum("oi,bot->bit", W, dpre).cpu()) < TOL_FWD
    assert rel_err(f.bias.grad.cpu(), dpre.sum((0, 2)).cpu()) < TOL_FWD

    def both(make, call, n_in):
        outs = {}
        for tag, storage, gm in (("ref", False, "f32"), ("c8", True, "bf16")):
            ops.set_gemm_precision(gm)
            ops.bf16_storage = storage
            layer = make()
            ins = [t.clone().requires_grad_(True) for t in n_in]
            ops.dropout_state.manual_seed(5)
            y = call(layer, ins, storage and gm == "bf16")
            y = ops.from_c8(y) if ops.is_c8(y) else y
            (y * torch.linspace(-1, 1, y.numel(), device=dev).view_as(y)).sum().backward()
            outs[tag] = [y.detach()] + [t.grad.detach() for t in ins] + [p.grad.detach().clone() for p in layer.parameters()]
        for a, b in zip(outs["c8"], outs["ref"]):
            assert rel_err(a.cpu(), b.cpu()) < TOL_GRAD

    # (in, out, mode, c8 output): incl. channel counts that are not multiples of 8 on either side (the 513 linear bins)
    for (Ci, Co, mode, o8) in ((64, 128, ops.EPI_SOFTSIGN, True), (128, 64, ops.EPI_LINEAR, True),
                               (64, 513, ops.EPI_SIGMOID, False), (64, 513, ops.EPI_SOFTSIGN, True),
                               (513, 64, ops.EPI_LINEAR, True), (513, 513, ops.EPI_SIGMOID, False)):
        def make(Ci=Ci, Co=Co):
            torch.manual_seed(1)
            return modules.Conv1d(Ci, Co, 1, dropout=0.1).to(dev).train()
        both(make, lambda l, ins, c8, mode=mode, o8=o8: l(ops.to_c8(ins[0]) if c8 else ins[0], mode=mode,
                                                        out_c8=o8 if c8 else None),
             [torch.randn(B, Ci, T, device=dev)])

    def make_lin():
        torch.manual_seed(2)
        return modules.Linear(64, 128).to(dev).train()
    both(make_lin, lambda l, ins, c8: l.forward_bct(ins[0], r=ops.to_c8(ins[1]) if c8 else ins[1],
                                                    r2=ops.to_c8(ins[2]) if c8 else ins[2], out_c8=True if c8 else None),
         [torch.randn(B, 64, T, device=dev), torch.randn(B, 128, T, device=dev), torch.randn(B, 128, T, device=dev)])


@pytest.mark.parametrize("preset", ["deepvoice3_ljspeech", "nyanko_ljspeech", "deepvoice3_vctk"])
def test_c8_train_step_tracks_fp32_storage(dev, modes, preset):
    """one training forward + backward of the three preset models (preset channel counts, short sequences): the c8
    storage mode against the exact-fp32 mode and against the fp32-storage bf16 mode it replaces"""
    ops = modes
    import bench
    from deepvoice3_pytorch_amd import builder, train_step
    bname, hp, sigma = bench.PRESETS[preset]
    hp = dict(hp)
    res = {}
    for tag, storage, gm in (("f32", False, "f32"), ("bf16", False, "bf16"), ("c8", True, "bf16")):
        ops.set_gemm_precision(gm)
        ops.bf16_storage = storage
        torch.manual_seed(5)
        model = getattr(builder, bname)(**hp).to(dev)
        rng = np.random.RandomState(3)
        bt = bench.synth_batch(rng, 2, 60, 160, hp, fixed=True)
        spk = torch.from_numpy(rng.randint(0, hp["n_speakers"], 2)) if hp["n_speakers"] > 1 else None
        tr = train_step.Trainer(model, train_step.TrainConfig(max_positions=hp["max_positions"],
                                                              guided_attention_sigma=sigma))
        batch = train_step.Batch.from_collate(bt["text"], bt["input_lengths"], bt["mel"], bt["y"], bt["text_positions"],
                                              bt["frame_positions"], bt["done"], bt["target_lengths"], spk,
                                              downsample_step=4, device=dev)
        ops.dropout_state.manual_seed(9)
        tr.arena.grad.zero_()
        scal = tr.forward_backward(batch)
        res[tag] = (float(scal["loss"]), tr.arena.grad.detach().clone())
        tr.close()

    def cos(a, b):
        return float((a.double() * b.double()).sum() / (a.double().norm() * b.double().norm()))
    l32, g32 = res["f32"]
    assert abs(res["c8"][0] - l32) < 2e-2 * abs(l32), (res["c8"][0], l32)
    assert cos(res["c8"][1], g32) > 0.995
    assert cos(res["c8"][1], res["bf16"][1]) > 0.995


def _bf16_ulp(t):
    """spacing of bf16 numbers at |t| (8 significand bits): 2^(floor(log2|t|) - 7)"""
    a = t.abs().double().clamp_min(2.0 ** -120)
    return torch.pow(2.0, torch.floor(torch.log2(a)) - 7.0)


@pytest.mark.parametrize("kind,C,k,d,causal,T,B", GATED)
def test_c8_gated_layers_pin_to_the_same_rounding_oracle(dev, modes, kind, C, k, d, causal, T, B):
    """The 2e-2 bound above (against the fp32 oracle) only says "bf16-class".  This pins the c8 forward to the oracle
    evaluated with THE SAME roundings -- input and weight-normed weights rounded to bf16 at the conv
    (O.set_operand_rounding), fp32 accumulate, fp32 tail -- by the storage format's own half ulp:
        |y_hip - y_oracle| <= (1/2 + 1/16) ulp_bf16(y_oracle) + 2e-6 * max|y|
    (the HIP result IS a bf16 number; its fp32 value before the store differs from the oracle's by summation order only),
    and counts where the stored bf16 differs from round_bf16(y_oracle): only values within 2e-6 of a rounding boundary
    may flip, by one ulp.  Eval mode (the dropout scale multiplies the accumulators in the HIP path and the operand in
    the reference: not the same rounding, covered by the tolerance tests above)."""
    ops = modes
    from deepvoice3_pytorch_amd import modules
    torch.manual_seed(0)
    if kind == "highway":
        layer = modules.HighwayConv1d(C, C, k, dilation=d, causal=causal, dropout=0.1)
    else:
        layer = modules.Conv1dGLU(1, 16, C, C, k, dropout=0.2, dilation=d, causal=causal, residual=(kind == "glu"))
    layer = layer.to(dev).eval()
    with torch.no_grad():
        layer.conv.bias.uniform_(-0.1, 0.1)
    x = torch.randn(B, C, T, device=dev).to(torch.bfloat16).float()        # what the c8 tensor holds
    ops.set_gemm_precision("bf16")
    ops.bf16_storage = True
    with torch.no_grad():
        y8 = layer(ops.to_c8(x))
    assert ops.is_c8(y8)
    got = ops.from_c8(y8).cpu().double()
    # the oracle gets the HIP path's OWN bf16 weights, read back from the operand image dv3_weight_norm_split_pack_bf16
    # wrote ([plane hi][tap][k/8][column][8]; the gate rows' columns start at a_half): g * v / ||v|| evaluated on the GPU
    # and on the CPU can land on different sides of a bf16 rounding boundary for a few of the k * C weights of a row, and
    # each such flip is worth ~2e-4 of the output at C = 256 -- the pin is about the GEMM and the tail, not about that
    v_, g_ = layer.conv.wn_params()
    pk = ops.pack_weights(v_.detach(), g_.detach(), glu_cg=C, need_bwd=False, split_only=True)
    kp = (C + 31) // 32 * 32
    img = pk.fwd_s.view(torch.bfloat16)[: k * kp * pk.lda].view(k, kp // 8, pk.lda, 8).float().cpu()
    w_hip = torch.empty(2 * C, C, k)
    for o in range(2 * C):
        col = o if o < C else pk.a_half + (o - C)
        w_hip[o] = img[:, :, col, :].reshape(k, kp)[:, :C].t()
    sd = {"l.conv.weight": w_hip, "l.conv.bias": layer.conv.bias.detach().cpu()}
    O.set_operand_rounding("bf16")
    try:
        if kind == "highway":
            want = O.highway_conv1d(sd, "l", x.cpu(), k, d, causal)
        else:
            want = O.conv1d_glu(sd, "l", x.cpu(), k, d, causal, kind == "glu")
    finally:
        O.set_operand_rounding(None)
    want = want.double()
    ulp = _bf16_ulp(want)
    slack = 2e-6 * float(want.abs().max())
    excess = ((got - want).abs() - (0.5 + 1.0 / 16) * ulp - slack).max()
    assert float(excess) <= 0.0, "c8 output leaves the half-ulp band of the same-rounding oracle by %.3e" % float(excess)
    stored = want.float().to(torch.bfloat16).double()
    flips = got != stored
    frac = float(flips.double().mean())
    assert frac < 2e-2, "too many stored values differ from round_bf16(oracle): %.4f" % frac
    ulp2 = torch.maximum(ulp, _bf16_ulp(got))        # a flip across a power of two moves by the larger binade's ulp
    assert float(((got - stored).abs() - ulp2 * 1.0001 - slack)[flips].max() if flips.any() else -1.0) <= 0.0   # by one ulp


# the 256 x 256 k32 ping-pong c8 kernel (csrc/conv_c8pp.hip) against the 128-row planes kernel it replaces where its
# grid fills the chip: same operand images, same accumulation order, same fused tails -> bit-identical outputs
C8PP = [("glu", 64, 3, 2, False, 75, 3), ("glu", 256, 3, 27, False, 150, 2), ("glu", 128, 3, 1, True, 100, 2),
        ("glu", 96, 3, 9, False, 61, 5), ("glu", 256, 3, 3, False, 800, 4), ("glu_nores", 64, 3, 3, False, 61, 2),
        ("highway", 64, 3, 2, False, 75, 3), ("highway", 128, 1, 1, False, 50, 3), ("highway", 512, 3, 27, True, 150, 2),
        ("glu", 32, 3, 1, False, 33, 7), ("glu", 128, 1, 1, False, 50, 3)]


@pytest.mark.parametrize("kind,C,k,d,causal,T,B", C8PP)
def test_c8pp_kernel_is_bit_identical_to_the_planes_kernel(dev, modes, kind, C, k, d, causal, T, B):
    ops = modes
    from deepvoice3_pytorch_amd import modules, _lib
    L = _lib.lib()
    torch.manual_seed(0)
    if kind == "highway":
        layer = modules.HighwayConv1d(C, C, k, dilation=d, causal=causal, dropout=0.1)
    else:
        layer = modules.Conv1dGLU(1, 16, C, C, k, dropout=0.2, dilation=d, causal=causal, residual=(kind == "glu"))
    layer = layer.to(dev).train()
    with torch.no_grad():
        layer.conv.bias.uniform_(-0.1, 0.1)
    x = torch.randn(B, C, T, device=dev)
    out = {}
    try:
        # planes kernel | 8 waves on 256 x 256 | two 4-wave workgroups per CU on 256 x 128 (round 5: dv3_debug_set(34, 1))
        for tag, thr, nw4, want in (("planes", 0, 0, 8), ("c8pp", 1, 0, 9101), ("c8pp 4-wave", 1, 1, 9111)):
            L.dv3_debug_set(19, thr)
            L.dv3_debug_set(34, nw4)
            out[tag] = _run(layer, x, True, "bf16", ops)
            # the last tap-GEMM of backward is the input gradient: served by the kernel under test
            v = L.dv3_debug_get(10)
            assert (v // 1000 == want) if want < 10 else (v == want), (tag, v)
    finally:
        L.dv3_debug_set(19, 128)
        L.dv3_debug_set(34, 2)
    y0, dx0, dp0 = out["planes"]
    for tag in ("c8pp", "c8pp 4-wave"):
        y1, dx1, dp1 = out[tag]
        assert torch.equal(y0, y1), (tag, float((y0 - y1).abs().max()))
        assert torch.equal(dx0, dx1), (tag, float((dx0 - dx1).abs().max()))
        for n in dp0:
            assert torch.equal(dp0[n], dp1[n]), (tag, n)


@pytest.mark.parametrize("d,causal", [(1, False), (27, True)])
def test_north_star_c8pp_is_the_kernel_the_bench_times(dev, modes, d, causal):
    """BASELINE.json's shape on the bf16 / c8 path: Conv1dGLU at B = 64 x 256 channels x T = 1024, k = 3 -- what
    bench.py's `roofline_bf16_c8` times and what the nyanko / vctk steps dispatch wherever a grid has >= 128 tiles.  With
    the DEFAULT dispatch (nothing forced):
      * eval forward: served by conv_c8pp_kernel (variant 9101), every stored value within (1/2 + 1/16) bf16 ulp of the
        oracle evaluated with the same roundings (bf16 operands from the HIP path's own weight image, fp32 accumulate and
        tail), the boundary flips counted;
      * training forward (keep-bytes, pre-gate save) + backward (input gradient, weight / gain / bias gradients): variant
        9101 again, and bit-identical to the 128-row planes kernel at this very shape (dv3_debug_set(19, 0)), which the
        small-shape tests above hold to the oracle -- both LOAD-phase orders of the kernel (dv3_debug_set(30, v))."""
    ops = modes
    from deepvoice3_pytorch_amd import modules, _lib
    L = _lib.lib()
    B, C, T, k = 64, 256, 1024, 3
    torch.manual_seed(0)
    layer = modules.Conv1dGLU(1, 16, C, C, k, dropout=0.05, dilation=d, causal=causal, residual=True).to(dev)
    with torch.no_grad():
        layer.conv.bias.uniform_(-0.1, 0.1)
    x = torch.randn(B, C, T, device=dev).to(torch.bfloat16).float()
    ops.set_gemm_precision("bf16")
    ops.bf16_storage = True
    layer.eval()
    with torch.no_grad():
        y8 = layer(ops.to_c8(x))
    assert L.dv3_debug_get(10) == 9101, L.dv3_debug_get(10)
    got = ops.from_c8(y8).cpu().double()
    v_, g_ = layer.conv.wn_params()
    pk = ops.pack_weights(v_.detach(), g_.detach(), glu_cg=C, need_bwd=False, split_only=True)
    kp = (C + 31) // 32 * 32
    img = pk.fwd_s.view(torch.bfloat16)[: k * kp * pk.lda].view(k, kp // 8, pk.lda, 8).float().cpu()
    w_hip = torch.empty(2 * C, C, k)
    for o in range(2 * C):
        col = o if o < C else pk.a_half + (o - C)
        w_hip[o] = img[:, :, col, :].reshape(k, kp)[:, :C].t()
    sd = {"l.conv.weight": w_hip, "l.conv.bias": layer.conv.bias.detach().cpu()}
    O.set_operand_rounding("bf16")
    try:
        want = O.conv1d_glu(sd, "l", x.cpu(), k, d, causal, True).double()
    finally:
        O.set_operand_rounding(None)
    ulp = _bf16_ulp(want)
    slack = 2e-6 * float(want.abs().max())
    excess = ((got - want).abs() - (0.5 + 1.0 / 16) * ulp - slack).max()
    assert float(excess) <= 0.0, "variant 9101 leaves the half-ulp band of the same-rounding oracle by %.3e" % float(excess)
    stored = want.float().to(torch.bfloat16).double()
    flips = got != stored
    assert float(flips.double().mean()) < 2e-2
    del want, got, stored, flips, ulp
    # training forward + backward: the benchmarked kernel against the planes kernel at the benchmarked shape
    layer.train()
    out = {}
    try:
        # (34: 0 = 8-wave form only, 1 = the 4-wave form everywhere, 2 = the dispatcher's rule -- what a training step runs:
        #  masked forward on two 4-wave workgroups per CU, this 256-tile input gradient on the 8-wave form)
        for tag, thr, rf, nw4 in (("planes", 0, 1, 0), ("c8pp", 128, 1, 0), ("c8pp staging first", 128, 0, 0),
                                  ("c8pp 4-wave", 128, 1, 1), ("c8pp by the rule", 128, 1, 2)):
            L.dv3_debug_set(19, thr)
            L.dv3_debug_set(30, rf)
            L.dv3_debug_set(34, nw4)
            out[tag] = _run(layer, x, True, "bf16", ops)
            v = L.dv3_debug_get(10)
            assert (v // 1000 == 9) == bool(thr), (tag, v)
            if thr:
                assert v == (9111 if nw4 == 1 else 9101), (tag, v)
    finally:
        L.dv3_debug_set(19, 128)
        L.dv3_debug_set(30, 1)
        L.dv3_debug_set(34, 2)
    y0, dx0, dp0 = out["planes"]
    for tag in ("c8pp", "c8pp staging first", "c8pp 4-wave", "c8pp by the rule"):
        y1, dx1, dp1 = out[tag]
        assert torch.equal(y0, y1), (tag, float((y0 - y1).abs().max()))
        assert torch.equal(dx0, dx1), (tag, float((dx0 - dx1).abs().max()))
        for n in dp0:
            assert torch.equal(dp0[n], dp1[n]), (tag, n)


@pytest.mark.parametrize("d,causal", [(1, False), (27, True)])
def test_north_star_c8_training_pass_against_the_same_rounding_oracle(dev, modes, d, causal):
    """The hop test_north_star_c8pp_is_the_kernel_the_bench_times leaves to the planes kernel: at
    BASELINE's shape (B = 64 x 256 x 1024) the c8 TRAINING forward (keep-bytes, pre-gate save) and every gradient
    DIRECTLY against the oracle evaluated with the same operand rounding (bf16 conv inputs incl. the weight-normed
    weights -- so weight-norm packing is inside the comparison --, fp32 accumulate and tail) and the same dropout
    decisions.  The full tensors are too slow for the host oracle's autograd: the loss weights are zero outside every
    16th batch item, so the HIP path runs the benchmarked shape on the benchmarked kernels (variant 9101 asserted) while
    the oracle differentiates only the sampled items (dx, dW, dg, dbias of a loss that touches nothing else are sums
    over those items only)."""
    ops = modes
    from deepvoice3_pytorch_amd import modules, _lib
    L = _lib.lib()
    B, C, T, k, p = 64, 256, 1024, 3, 0.05
    items = list(range(0, B, 16))
    torch.manual_seed(0)
    layer = modules.Conv1dGLU(1, 16, C, C, k, dropout=p, dilation=d, causal=causal, residual=True).to(dev).train()
    layer._dv3_site = "l"
    with torch.no_grad():
        layer.conv.bias.uniform_(-0.1, 0.1)
    x = torch.randn(B, C, T, device=dev).to(torch.bfloat16).float()
    ops.set_gemm_precision("bf16")
    ops.bf16_storage = True
    for q in layer.parameters():
        q.grad = None
    xin = x.clone().requires_grad_(True)
    w = torch.zeros(B, C, T, device=dev)
    gen = torch.Generator(device="cpu").manual_seed(1)
    w_s = torch.randn(len(items), C, T, generator=gen)
    w[items] = w_s.to(dev)
    rec = {}
    ops.dropout_state.manual_seed(3)
    ops.dropout_state.record = rec
    try:
        y = ops.from_c8(layer(ops.to_c8(xin)))
        v_fwd = L.dv3_debug_get(10)
        (y * w).sum().backward()
        v_bwd = L.dv3_debug_get(10)
    finally:
        ops.dropout_state.record = None
    assert v_fwd in (9101, 9111) and v_bwd == 9101, (v_fwd, v_bwd)     # masked forward: 4-wave form by the rule; dgrad: 9101
    bits, rows, Tm = rec["l"]
    keep = torch.from_numpy(O.unpack_keep_bits(bits.cpu().numpy().view(np.uint32), rows, (Tm + 31) // 32, Tm)).float()
    keep = keep.view(B, C, T)[items]

    def drop(site, t, p_, layout):
        return t * keep / (1 - p_)
    sd = {"l.conv." + n: t.detach().cpu().clone().requires_grad_(True) for n, t in layer.conv.state_dict().items()}
    xs = x[items].cpu().clone().requires_grad_(True)
    prev_threads = torch.get_num_threads()
    torch.set_num_threads(min(32, prev_threads))
    O.set_operand_rounding("bf16")
    try:
        want = O.conv1d_glu(sd, "l", xs, k, d, causal, True, p=p, drop=drop)
        (want * w_s).sum().backward()
    finally:
        O.set_operand_rounding(None)
        torch.set_num_threads(prev_threads)
    # forward: a stored bf16 value against the fp32 tail of the same arithmetic -- half an ulp of the value, i.e. at most
    # 2^-9 of the tensor's range (+ dropout scale placement: the HIP path scales the accumulators, the oracle the operand)
    e_y = rel_err(y.detach()[items].cpu(), want.detach())
    e_dx = rel_err(xin.grad[items].cpu(), xs.grad)
    assert float(xin.grad[[i for i in range(B) if i not in items]].abs().max()) == 0.0
    errs = dict(y=e_y, dx=e_dx)
    for n, q in layer.conv.named_parameters():
        errs[n] = rel_err(q.grad.cpu(), sd["l.conv." + n].grad)
    assert e_y < 6e-3, errs
    assert e_dx < 2e-2, errs
    for n in ("weight_v", "weight_g", "bias"):
        assert errs[n] < 2e-2, errs


def test_c8pp_plain_and_fp32_output_forms(dev, modes):
    """1 x 1 layers through the same kernel: c8 -> c8 with activation / residuals, c8 -> fp32 (513 rows: not a multiple
    of anything), and their input gradients -- bit-identical to the planes kernel"""
    ops = modes
    from deepvoice3_pytorch_amd import modules, _lib
    L = _lib.lib()
    B, T = 3, 130
    ops.set_gemm_precision("bf16")
    ops.bf16_storage = True
    res = {}
    try:
        for tag, thr in (("planes", 0), ("c8pp", 1)):
            L.dv3_debug_set(19, thr)
            outs = []
            for (Ci, Co, mode, o8) in ((64, 128, ops.EPI_RELU, True), (128, 64, ops.EPI_LINEAR, True),
                                       (64, 513, ops.EPI_SIGMOID, False), (513, 64, ops.EPI_LINEAR, True),
                                       (256, 256, ops.EPI_SOFTSIGN, True)):
                torch.manual_seed(1)
                f = modules.Conv1d(Ci, Co, 1, dropout=0.1).to(dev).train()
                xin = torch.randn(B, Ci, T, device=dev).requires_grad_(True)
                ops.dropout_state.manual_seed(5)
                y = f(ops.to_c8(xin), mode=mode, out_c8=o8)
                y = ops.from_c8(y) if ops.is_c8(y) else y
                (y * torch.linspace(-1, 1, y.numel(), device=dev).view_as(y)).sum().backward()
                outs += [y.detach(), xin.grad.detach()] + [p.grad.detach().clone() for p in f.parameters()]
            torch.manual_seed(2)
            lin = modules.Linear(64, 128).to(dev).train()
            ins = [torch.randn(B, 64, T, device=dev).requires_grad_(True), torch.randn(B, 128, T, device=dev).requires_grad_(True),
                   torch.randn(B, 128, T, device=dev).requires_grad_(True)]
            y = ops.from_c8(lin.forward_bct(ops.to_c8(ins[0]), r=ops.to_c8(ins[1]), r2=ops.to_c8(ins[2]), out_c8=True))
            (y * torch.linspace(-1, 1, y.numel(), device=dev).view_as(y)).sum().backward()
            outs += [y.detach()] + [t.grad.detach() for t in ins]
            res[tag] = outs
    finally:
        L.dv3_debug_set(19, 128)
    for i, (a, b) in enumerate(zip(res["planes"], res["c8pp"])):
        assert torch.equal(a, b), (i, float((a - b).abs().max()))


@pytest.mark.parametrize("kind,C,k,d,causal,T,B", [("glu", 64, 3, 2, False, 75, 3), ("glu", 256, 3, 27, False, 150, 2),
                                                   ("highway", 128, 1, 1, False, 50, 3), ("glu", 96, 3, 9, True, 61, 5),
                                                   ("glu", 256, 3, 1, False, 33, 1)])
def test_wgrad_c8_two_steps_ahead_is_bit_identical(dev, modes, kind, C, k, d, causal, T, B):
    """wgrad_c8_kernel<.., PF2> (operands fetched two steps ahead into two register sets) against the one-step form:
    same tile, LDS image and accumulation order -> the same parameter gradients bit for bit (masked and not, K ranges
    of one step included)"""
    ops = modes
    from deepvoice3_pytorch_amd import modules, _lib
    L = _lib.lib()
    torch.manual_seed(0)
    if kind == "highway":
        layer = modules.HighwayConv1d(C, C, k, dilation=d, causal=causal, dropout=0.1)
    else:
        layer = modules.Conv1dGLU(1, 16, C, C, k, dropout=0.2, dilation=d, causal=causal, residual=True)
    layer = layer.to(dev).train()
    x = torch.randn(B, C, T, device=dev)
    out = {}
    try:
        # register-transposing forms: one step ahead | two steps ahead, staging after the MFMAs | ... between them (three
        # taps: variant 5063); transposing-read forms (round 6, dv3_debug_set(52, v)): plain 51xx | staging between the
        # MFMAs 5143 | one barrier per two K steps 53xx | both 5343 (the default)
        for tag, tr, pf2, il in ((0, 0, 0, 0), (1, 0, 1, 0), (2, 0, 1, 1), (3, 1, 1, 1), (4, 2, 1, 1), (5, 3, 1, 1), (6, 4, 1, 1)):
            L.dv3_debug_set(52, tr)
            L.dv3_debug_set(20, pf2)
            L.dv3_debug_set(49, il)
            out[tag] = _run(layer, x, True, "bf16", ops)
            if tr == 0:
                want = 5000 + 20 * pf2 + k + (40 if (il and pf2 and k == 3) else 0)
            else:
                want = 5100 + k + (40 if (tr in (2, 4) and k == 3) else 0) + (200 if tr in (3, 4) else 0)
            assert L.dv3_debug_get(11) == want, (tag, L.dv3_debug_get(11), want)
    finally:
        L.dv3_debug_set(20, 1)
        L.dv3_debug_set(49, 1)
        L.dv3_debug_set(52, 4)
    for tag in range(1, 7):
        for n in out[0][2]:
            assert torch.equal(out[0][2][n], out[tag][2][n]), (tag, n)
