# coding: utf-8
"""Data-parallel training step, end to end on the GPU: two ranks (two processes sharing cuda:0, gloo as
the transport -- RCCL refuses two ranks on one device, and the round's test box has one GPU) each run
train_step.Trainer on half of a batch; the result must equal one process running the whole batch.
This drives the real path: HIP kernels, in-place conv-layer gradients, bucket notifications from both
the autograd hooks and ops.grad_ready_hooks, side-stream all-reduce, clip with the 1/world prescale."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HP = dict(n_vocab=30, embed_dim=32, mel_dim=16, linear_dim=17, r=1, downsample_step=4, padding_idx=0, dropout=0.0,
          kernel_size=3, encoder_channels=64, decoder_channels=32, converter_channels=32, use_memory_mask=True,
          force_monotonic_attention=True, use_decoder_state_for_postnet_input=True, key_projection=True,
          value_projection=True, max_positions=128)


def _make_batch(lo, hi):
    sys.path.insert(0, ROOT)
    import bench
    bt = bench.synth_batch(np.random.RandomState(5), 4, 12, 40, HP)      # fixed shapes: equal loss normalisers
    sl = slice(lo, hi)
    return {k: (v[sl] if hasattr(v, "__getitem__") else v) for k, v in bt.items()}


def _run(rank, world, port, q, steps=2):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    from deepvoice3_pytorch_amd import builder, train_step
    dev = torch.device("cuda:0")
    pg = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        pg = dist.group.WORLD
    torch.manual_seed(0)
    model = builder.deepvoice3(**HP).to(dev)
    tr = train_step.Trainer(model, train_step.TrainConfig(max_positions=128), process_group=pg, bucket_mb=0.05)
    per = 4 // world
    bt = _make_batch(rank * per, (rank + 1) * per)
    batch = train_step.Batch.from_collate(bt["text"], bt["input_lengths"], bt["mel"], bt["y"], bt["text_positions"],
                                          bt["frame_positions"], bt["done"], bt["target_lengths"], None,
                                          downsample_step=4, device=dev)
    for _ in range(steps):
        scal = tr.step(batch)
    torch.cuda.synchronize()
    if world > 1:
        assert len(tr.comm.buckets) >= 2
    q.put((rank, tr.arena.flat.cpu().numpy(), float(scal["grad_norm"])))
    if world > 1:
        torch.distributed.destroy_process_group()


def _spawn(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 1000) + world
    procs = [ctx.Process(target=_run, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return sorted(res, key=lambda t: t[0])


def test_two_rank_step_equals_single_process_step():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    try:
        two = _spawn(2)
    except Exception as e:      # gloo without device-tensor support on this build
        pytest.skip("2-process gloo-on-GPU run not possible here: %r" % (e,))
    one = _spawn(1)
    w2a, w2b, w1 = two[0][1], two[1][1], one[0][1]
    rep = float(np.abs(w2a - w2b).max())
    diff = float(np.abs(w2a - w1).max())
    moved = float(np.abs(w1).max())
    msg = "replica diff %.3e, vs single %.3e (max |w| %.3e), grad_norm %r vs %r" % (rep, diff, moved, two[0][2], one[0][2])
    assert rep <= 1e-6 * moved, msg                       # replicas stay identical (gloo may round per rank)
    assert diff < 2e-5 * moved, msg                       # == the single-process step on the whole batch
    assert abs(two[0][2] - one[0][2]) < 1e-4 * abs(one[0][2]), msg
