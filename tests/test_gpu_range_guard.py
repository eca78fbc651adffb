# coding: utf-8
"""-m gpu: the range guard of the f16x3 GEMM mode (include/dv3hip.h, dv3_f16_range_events).

The forward operands of the default mode are v * 2^4 (activations) / v * 2^8 (weights) in fp16.  Outside
|x| <= 4094 / |w| <= 255.9 nothing may be saturated silently: the kernels count the operand units
that left the range, values stay usable to twice the range and turn non-finite beyond it, NaN / Inf inputs
propagate as in the fp32 reference (deepvoice3_pytorch/modules.py:145-164 run on such inputs), and the host side
can re-run a computation in the bf16x3 mode (fp32's exponent range) with parity kept."""
import math

import numpy as np
import pytest
import torch

from oracle import dv3_oracle as O
from tests.util import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def f16x3_mode():
    from deepvoice3_pytorch_amd import ops
    prev = ops.set_gemm_precision("f16x3")
    yield
    ops.set_gemm_precision(prev)


def _layer(C, k, seed, g_scale=1.0):
    rng = np.random.RandomState(seed)
    return {"l.conv.weight_v": torch.from_numpy(rng.randn(2 * C, C, k).astype(np.float32) * 0.2),
            "l.conv.weight_g": torch.from_numpy((rng.uniform(0.5, 1.5, (2 * C, 1, 1)) * g_scale).astype(np.float32)),
            "l.conv.bias": torch.from_numpy(rng.uniform(-0.2, 0.2, 2 * C).astype(np.float32))}


def _glu(ops, sd, x, dev, k=3, d=1):
    C = x.shape[1]
    cfg = ops.LayerCfg(k=k, dil=d, causal=False, mode=ops.EPI_GLU, residual=True)
    return ops.conv_layer(x.to(dev), sd["l.conv.weight_v"].to(dev), sd["l.conv.weight_g"].to(dev),
                          sd["l.conv.bias"].to(dev), cfg)


def _lin_layer(C, k, seed, g_scale=1.0):
    rng = np.random.RandomState(seed)
    return (torch.from_numpy(rng.randn(C, C, k).astype(np.float32) * 0.2),
            torch.from_numpy((rng.uniform(0.5, 1.5, (C, 1, 1)) * g_scale).astype(np.float32)),
            torch.from_numpy(rng.uniform(-0.2, 0.2, C).astype(np.float32)))


def _lin(ops, vgb, x, dev, k=3):
    """a plain weight-normed Conv1d (conv.py:7-16 + nn.utils.weight_norm): linear in x, so a GEMM-level error bound
    carries to the output whatever the magnitudes (a gated layer's sigmoid is ill-conditioned at 1e4-size pre-gates)"""
    v, g, b = vgb
    cfg = ops.LayerCfg(k=k, dil=1, causal=False, mode=ops.EPI_LINEAR)
    return ops.conv_layer(x.to(dev), v.to(dev), g.to(dev), b.to(dev), cfg)


def _lin_ref(vgb, x, k=3):
    v, g, b = [t.double() for t in vgb]
    w = g * v / v.reshape(v.shape[0], -1).norm(dim=1).view(-1, 1, 1)
    return torch.nn.functional.conv1d(x.double(), w, b, padding=(k - 1) // 2).float()


def test_in_range_inputs_leave_the_counter_at_zero(dev):
    from deepvoice3_pytorch_amd import ops
    sd = _layer(64, 3, 0)
    x = torch.from_numpy(np.random.RandomState(1).randn(2, 64, 90).astype(np.float32)) * 100.0   # |x| < 4094
    ops.f16_range_events(reset=True)
    y = _glu(ops, sd, x, dev)
    assert ops.f16_range_events() == 0
    assert rel_err(y.cpu(), O.conv1d_glu(sd, "l", x, 3, 1, False, True)) < 5e-6


def test_large_activations_fire_the_counter_and_fall_back_with_parity(dev):
    from deepvoice3_pytorch_amd import ops
    sd = _layer(64, 3, 2)
    x = torch.from_numpy(np.random.RandomState(3).randn(2, 64, 90).astype(np.float32)) * 1e4     # most |x| > 4094
    want = O.conv1d_glu(sd, "l", x, 3, 1, False, True)
    ops.f16_range_events(reset=True)
    y = _glu(ops, sd, x, dev)
    n = ops.f16_range_events()
    assert n > 0, "activations of magnitude 1e4 must be reported"
    # nothing was saturated silently: wherever the f16x3 result is finite it is the right number
    yc = y.cpu()
    fin = torch.isfinite(yc)
    assert float(torch.where(fin, (yc - want).abs(), torch.zeros_like(want)).max()) <= 2e-3 * float(want.abs().max())
    got, fell_back = ops.with_f16_range_fallback(lambda: _glu(ops, sd, x, dev))
    assert fell_back and ops.gemm_precision() == "f16x3"          # the mode is restored afterwards
    assert torch.isfinite(got).all()
    # parity of the fallback on a LINEAR layer of the same size (the GLU's sigmoid turns a 5e-6 pre-gate error into 1e-3
    # of the output where a 1e4-size gate crosses zero: conditioning, not arithmetic)
    vgb = _lin_layer(64, 3, 12)
    ops.f16_range_events(reset=True)
    got, fell_back = ops.with_f16_range_fallback(lambda: _lin(ops, vgb, x, dev))
    assert fell_back and torch.isfinite(got).all()
    assert rel_err(got.cpu(), _lin_ref(vgb, x)) < 5e-5           # the bf16x3 kernel tolerance


def test_twice_the_range_is_still_accurate(dev):
    """|x * 16| in (65504, 2 * 65504): hi saturates, lo carries the rest -- fp16-accurate, counted, finite.  The accuracy
    is checked on a plain (linear) layer, where the operand error bound carries to the output; a gated layer at 1e4-size
    pre-gates turns a 3e-5 error of a gate that happens to sit near zero into several 1e-2 of the output, whatever
    computes it (it is only required to stay finite and to be counted)."""
    from deepvoice3_pytorch_amd import ops
    sd = _layer(32, 3, 4)
    rng = np.random.RandomState(5)
    x = torch.from_numpy((rng.uniform(4200, 8000, (1, 32, 64)) * rng.choice([-1, 1], (1, 32, 64))).astype(np.float32))
    ops.f16_range_events(reset=True)
    y = _glu(ops, sd, x, dev)
    assert ops.f16_range_events(reset=True) > 0
    assert torch.isfinite(y).all()
    vgb = _lin_layer(32, 3, 4)
    yl = _lin(ops, vgb, x, dev)
    assert ops.f16_range_events(reset=True) > 0
    # lo = fp16(a - 65504) has a 32-unit ulp at |a| up to 2 x 65504: 2.5e-4 of the operand
    assert rel_err(yl.cpu(), _lin_ref(vgb, x)) < 1e-3


def test_nan_and_inf_inputs_propagate(dev):
    from deepvoice3_pytorch_amd import ops
    sd = _layer(32, 3, 6)
    x = torch.from_numpy(np.random.RandomState(7).randn(1, 32, 64).astype(np.float32))
    for bad in (float("nan"), float("inf")):
        xb = x.clone()
        xb[0, 5, 20] = bad
        y = _glu(ops, sd, xb, dev).cpu()
        # every output whose receptive field holds the poisoned sample is non-finite (the reference's F.conv1d gives
        # NaN there as well), the rest of the row is untouched
        assert not torch.isfinite(y[0, :, 19:22]).any()
        ref = _glu(ops, sd, x, dev).cpu()
        assert torch.equal(y[0, :, :19], ref[0, :, :19]) and torch.equal(y[0, :, 22:], ref[0, :, 22:])
    ops.f16_range_events(reset=True)


def test_large_weights_fire_the_counter(dev):
    from deepvoice3_pytorch_amd import ops
    sd = _layer(32, 3, 8, g_scale=1000.0)          # |w| up to ~1500 > 255.9
    x = torch.from_numpy(np.random.RandomState(9).randn(1, 32, 40).astype(np.float32))
    ops.f16_range_events(reset=True)
    _glu(ops, sd, x, dev)
    assert ops.f16_range_events(reset=True) > 0
    vgb = _lin_layer(32, 3, 10, g_scale=1000.0)
    got, fell_back = ops.with_f16_range_fallback(lambda: _lin(ops, vgb, x, dev))
    assert fell_back and rel_err(got.cpu(), _lin_ref(vgb, x)) < 5e-5


def test_trainer_reports_and_leaves_the_mode(dev):
    """Trainer.step carries the counter in its scalars; with range_check_every it moves the run to bf16x3"""
    import warnings
    import bench
    from deepvoice3_pytorch_amd import builder, ops, train_step
    hp = dict(n_vocab=30, embed_dim=32, mel_dim=16, linear_dim=17, r=1, downsample_step=4, padding_idx=0, dropout=0.0,
              kernel_size=3, encoder_channels=64, decoder_channels=32, converter_channels=32, use_memory_mask=True,
              force_monotonic_attention=True, use_decoder_state_for_postnet_input=True, key_projection=True,
              value_projection=True, max_positions=128)
    torch.manual_seed(0)
    model = builder.deepvoice3(**hp).to(dev)
    tr = train_step.Trainer(model, train_step.TrainConfig(max_positions=128, range_check_every=1))
    bt = bench.synth_batch(np.random.RandomState(5), 4, 12, 40, hp)
    batch = train_step.Batch.from_collate(bt["text"], bt["input_lengths"], bt["mel"], bt["y"], bt["text_positions"],
                                          bt["frame_positions"], bt["done"], bt["target_lengths"], None,
                                          downsample_step=4, device=dev)
    ops.f16_range_events(reset=True)
    scal = tr.step(batch)
    assert int(scal["f16_range_events"]) == 0 and ops.gemm_precision() == "f16x3"
    with torch.no_grad():       # blow one conv layer's gain far out of the fp16 weight range
        model.seq2seq.encoder.convolutions[2].conv.weight_g.mul_(1e4)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        scal = tr.step(batch)
    # the guard acts BEFORE the update: the step was redone in bf16x3 and only that result was applied
    assert ops.gemm_precision() == "bf16x3" and any("fp16 range" in str(x.message) for x in w)
    assert "f16_range_events" not in scal
    assert math.isfinite(float(scal["grad_norm"])) and math.isfinite(float(scal["loss"]))
    assert bool(torch.isfinite(tr.arena.flat).all()) and bool(torch.isfinite(tr.arena.exp_avg_sq).all())
    assert tr.adam_step == 2 and tr.global_step == 2       # one update per step() call, redo included
    scal = tr.step(batch)       # continues (re-packed images) in the full-range mode
    assert "f16_range_events" not in scal and math.isfinite(float(scal["loss"]))
    tr.close()
