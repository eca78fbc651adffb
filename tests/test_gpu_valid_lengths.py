# coding: utf-8
"""Valid-length steps (ops.ValidLengths, include/dv3hip.h ABI 42): a batch padded BEYOND its own maxima -- to the shape
of a captured step that is replayed for it -- must compute what the reference computes on the batch padded to its own
maxima (train.collate_fn, train.py:293-360; losses train.py:704-740; AttentionLayer deepvoice3.py:159-171; the zero
padding of nn.Conv1d, modules.py:139-143).  Every check here is "padded further + valid lengths == not padded further"
through the same HIP kernels."""
import os
import sys

import numpy as np
import pytest
import torch

from util import rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

HP = dict(n_vocab=30, embed_dim=32, mel_dim=16, linear_dim=17, r=1, downsample_step=4, padding_idx=0, dropout=0.0,
          kernel_size=3, encoder_channels=64, decoder_channels=32, converter_channels=32, use_memory_mask=False,
          force_monotonic_attention=False, use_decoder_state_for_postnet_input=True, key_projection=True,
          value_projection=True, max_positions=256)


def _dev():
    return torch.device("cuda:0")


def _i32(v):
    return torch.tensor([v], dtype=torch.int32, device=_dev())


def test_zero_tail_fp32_and_c8():
    from deepvoice3_pytorch_amd import ops
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(3, 5, 37, generator=g).to(_dev())
    for valid, mult, tail in ((30, 1, 8), (37, 1, 4), (10, 3, 9), (0, 1, 37)):
        y = x.clone()
        ops._zero_tail_raw(y, _i32(valid), mult, tail)
        want = x.clone()
        want[:, :, max(valid * mult, 37 - tail):] = 0
        assert torch.equal(y, want), (valid, mult, tail)
    xc = ops.to_c8(torch.randn(2, 40, 21, generator=g).to(_dev()))
    yc = xc.clone()
    ops._zero_tail_raw(yc, _i32(15), 1, 7)
    want = xc.clone()
    want[:, :, 15:] = 0
    assert torch.equal(yc.view(torch.int16), want.view(torch.int16))
    # the autograd wrapper: same columns of the gradient
    a = torch.randn(2, 4, 16, generator=g).to(_dev()).requires_grad_()
    b = a * 2.0
    out = ops.zero_tail(b, _i32(11), 6)
    assert float(out[:, :, 11:].abs().max()) == 0.0 and torch.equal(out[:, :, :11], (a * 2.0)[:, :, :11])
    out.backward(torch.ones_like(out))
    assert float(a.grad[:, :, 11:].abs().max()) == 0.0 and float((a.grad[:, :, :11] - 2.0).abs().max()) == 0.0


@pytest.mark.parametrize("layout", ["btc", "bct"])
def test_losses_on_the_valid_part(layout):
    from deepvoice3_pytorch_amd import ops
    g = torch.Generator(device="cpu").manual_seed(1)
    B, T, Tv, D, r = 3, 70, 61, 20, 2
    dev = _dev()
    yh = torch.rand(B, T, D, generator=g).to(dev) * 0.98 + 0.01
    if layout == "bct":
        yh = yh.transpose(1, 2).contiguous().transpose(1, 2)
    y = torch.rand(B, T, D, generator=g).to(dev)
    y[:, Tv:] = 0
    lens = torch.tensor([61, 40, 13], dtype=torch.int32, device=dev)
    o_full, g_full = ops.spec_loss_with_grad(yh, y, lens, r, 0.5, 0.1, t_valid=_i32(Tv))
    o_cut, g_cut = ops.spec_loss_with_grad(yh[:, :Tv].contiguous(), y[:, :Tv].contiguous(), lens, r, 0.5, 0.1)
    assert rel_err(o_full.cpu(), o_cut.cpu()) < 1e-6
    assert float(g_full[:, Tv - r:].abs().max()) == 0.0
    assert rel_err(g_full[:, :Tv].cpu(), g_cut.cpu()) < 1e-6
    # done flags
    p = torch.rand(B, T, 1, generator=g).to(dev) * 0.9 + 0.05
    t = (torch.rand(B, T, 1, generator=g) > 0.5).float().to(dev)
    o_full, g_full = ops.bce_loss_with_grad(p, t, t_valid=_i32(Tv))
    o_cut, g_cut = ops.bce_loss_with_grad(p[:, :Tv].contiguous(), t[:, :Tv].contiguous())
    assert rel_err(o_full.cpu(), o_cut.cpu()) < 1e-6
    assert float(g_full[:, Tv:].abs().max()) == 0.0 and rel_err(g_full[:, :Tv].cpu(), g_cut.cpu()) < 1e-6
    # guided attention
    Tk, Tkv = 33, 29
    attn = torch.rand(2, B, T, Tk, generator=g).to(dev)
    il = torch.tensor([29, 20, 7], dtype=torch.int32, device=dev)
    o_full, g_full = ops.guided_attention_loss_with_grad(attn, il, lens, 0.2, _i32(Tv), _i32(Tkv))
    o_cut, g_cut = ops.guided_attention_loss_with_grad(attn[:, :, :Tv, :Tkv].contiguous(), il, lens, 0.2)
    assert rel_err(o_full.cpu(), o_cut.cpu()) < 1e-6
    assert rel_err(g_full[:, :, :Tv, :Tkv].cpu(), g_cut.cpu()) < 1e-6
    assert float(g_full[:, :, Tv:].abs().max()) == 0.0 and float(g_full[:, :, :, Tkv:].abs().max()) == 0.0


def _trainer(builder_name="deepvoice3", hp=HP, seed=0, **cfg):
    from deepvoice3_pytorch_amd import builder, train_step
    torch.manual_seed(seed)
    model = getattr(builder, builder_name)(**hp).to(_dev())
    tc = train_step.TrainConfig(max_positions=hp["max_positions"], outputs_per_step=hp["r"],
                                downsample_step=hp["downsample_step"], **cfg)
    return train_step.Trainer(model, tc)


def _batch(hp=HP, B=4, seed=3, speakers=0):
    import bench
    from deepvoice3_pytorch_amd import train_step
    rng = np.random.RandomState(seed)
    tl = rng.randint(9, 27, B)
    fl = rng.randint(40, 117, B)
    bt = bench.synth_batch(rng, B, 0, 0, hp, lengths=(tl, fl))
    spk = torch.from_numpy(rng.randint(0, speakers, B)).long() if speakers else None
    return train_step.Batch.from_collate(bt["text"], bt["input_lengths"], bt["mel"], bt["y"], bt["text_positions"],
                                         bt["frame_positions"], bt["done"], bt["target_lengths"], spk,
                                         downsample_step=hp["downsample_step"], device=_dev(), r=hp["r"])


def _step_result(tr, batch):
    tr._set_hyper()
    tr._zero_grad()
    scal = tr.forward_backward(batch)
    torch.cuda.synchronize()
    return {k: float(v) for k, v in scal.items()}, tr.arena.grad.detach().cpu().numpy().copy()


@pytest.fixture
def gemm_mode(request):
    from deepvoice3_pytorch_amd import ops
    prev = ops.set_gemm_precision(request.param)
    yield request.param
    ops.set_gemm_precision(prev)


@pytest.mark.parametrize("gemm_mode", ["f16x3", "bf16"], indirect=True)
@pytest.mark.parametrize("which", ["deepvoice3", "deepvoice3_mask", "multispeaker", "nyanko"])
def test_padded_step_equals_the_step_on_the_batch_maxima(which, gemm_mode):
    """losses and every parameter gradient of one training step: the batch padded to a lattice shape with its maxima
    attached against the batch as the reference's collate_fn pads it.  In the bf16 mode (BASELINE configs 3 / 4) the
    activations between the layers are channel-blocked bf16 tensors: the same zero columns, 16-byte units."""
    from deepvoice3_pytorch_amd import data
    hp, name, speakers = dict(HP), "deepvoice3", 0
    if which == "deepvoice3_mask":
        hp["use_memory_mask"] = True
    elif which == "multispeaker":
        name, speakers = "deepvoice3_multispeaker", 5
        hp.update(n_speakers=5, speaker_embed_dim=8)
    elif which == "nyanko":
        name = "nyanko"
        hp = dict(n_vocab=30, embed_dim=32, mel_dim=16, linear_dim=17, r=1, downsample_step=4, padding_idx=0,
                  dropout=0.0, kernel_size=3, encoder_channels=32, decoder_channels=32, converter_channels=64,
                  use_memory_mask=False, force_monotonic_attention=False, use_decoder_state_for_postnet_input=True,
                  max_positions=256)
    tr = _trainer(name, hp)
    b0 = _batch(hp, speakers=speakers)
    Tt, Td = b0.text.shape[1], b0.frame_positions.shape[1]
    t_in, t_dec = data.lattice_shape(Tt + 1, Td + 1, 16, 8)
    b1 = data.pad_to_shape(b0, t_in, t_dec, 15, 7)
    assert b1.text.shape[1] > Tt and b1.frame_positions.shape[1] > Td
    s0, g0 = _step_result(tr, b0)
    s1, g1 = _step_result(tr, b1)
    # (the padded shape changes how the weight-gradient kernels cut the time axis: fp32 sums in another order)
    tol_s, tol_g = (2e-6, 2e-5) if gemm_mode == "f16x3" else (1e-5, 1e-4)
    for k in s0:
        assert abs(s1[k] - s0[k]) <= tol_s * max(1.0, abs(s0[k])), (k, s0[k], s1[k])
    assert rel_err(g1, g0) < tol_g
    # ... and without the maxima the padded batch is another computation (the test would pass vacuously otherwise)
    b1.valid = None
    s2, _ = _step_result(tr, b1)
    assert abs(s2["loss"] - s0["loss"]) > 1e-4 * abs(s0["loss"])


def test_lattice_replay_matches_eager_steps_on_the_unpadded_batches():
    """three optimisation steps over batches of three different maxima, two of which share a lattice shape: the
    LatticeReplay (captured steps, batches padded to the lattice) against eager steps on the batches as collate_fn pads
    them, from the same initial state -- parameters after the steps, and every step's loss terms"""
    import bench
    from deepvoice3_pytorch_amd import data, train_step
    hp = dict(HP)
    items = []
    rng = np.random.RandomState(11)
    for tl, fl in ((20, 100), (25, 90), (21, 97), (30, 60), (13, 50), (29, 59), (22, 101), (24, 88)):
        text = np.concatenate([rng.randint(2, hp["n_vocab"], tl - 1), [1]]).astype(np.int32)
        items.append((text, rng.rand(fl, hp["mel_dim"]).astype(np.float32), rng.rand(fl, hp["linear_dim"]).astype(np.float32)))
    groups = [items[0:2], items[2:4], items[4:6], items[6:8], items[0:2]]
    dev = _dev()
    results = []
    for mode in ("eager", "lattice"):
        tr = _trainer("deepvoice3", hp, seed=1)
        rep = train_step.LatticeReplay(tr) if mode == "lattice" else None
        losses = []
        for g in groups:
            packed = data.pack_batch(g)
            if mode == "lattice":
                b = data.device_collate(packed, dev, 1, 4, lattice=(16, 8))
                scal = rep.step(b)
            else:
                b = data.device_collate(packed, dev, 1, 4)
                scal = tr.step(b)
            torch.cuda.synchronize()
            losses.append({k: float(v) for k, v in scal.items() if k.endswith("loss") or k == "grad_norm"})
        results.append((losses, tr.arena.flat.detach().cpu().numpy().copy()))
        if rep is not None:
            assert rep.stats["captures"] < len(groups) and rep.stats["replays"] == len(groups)
            rep.close()
    (l0, p0), (l1, p1) = results
    for a, b in zip(l0, l1):
        for k in a:
            # (Adam's 1 / sqrt(v) makes the fifth step's losses a few 1e-5 apart where the first step's agree to 1e-6)
            assert abs(a[k] - b[k]) <= 3e-5 * max(1.0, abs(a[k])), (k, a[k], b[k], l0, l1)
    assert rel_err(p1, p0) < 1e-4
    for k in l0[0]:
        assert abs(l0[0][k] - l1[0][k]) <= 2e-6 * max(1.0, abs(l0[0][k])), (k, l0[0][k], l1[0][k])
