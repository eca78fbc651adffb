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
    res = []
    for _ in range(240):
        try:
            res.append(q.get(timeout=1.0))
        except Exception:
            if not all(p.is_alive() or p.exitcode == 0 for p in procs):
                break
        if len(res) == world:
            break
    if len(res) != world:
        for p in procs:
            p.kill()
        raise RuntimeError("a rank ended without a result: exit codes %r" % ([p.exitcode for p in procs],))
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


# ----------------------------------------------------------------------------------------------------------------
# The RCCL path itself on the one GPU a test box has: a world-size-1 "nccl" group.  Proves that librccl loads next
# to libdv3hip.so, that the side-stream hand-off in dist.BucketedAllReduce (event -> collective stream ->
# wait_stream in finish()) and the bucket notification order are right under the real backend, and that the segmented
# replay (train_step.GraphedTrainer) issues the same all-reduces from the host between its segment launches.
# ----------------------------------------------------------------------------------------------------------------
def _run_nccl(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from deepvoice3_pytorch_amd import builder, train_step
    from deepvoice3_pytorch_amd import dist as dv3dist
    dev = torch.device("cuda:0")
    bt = _make_batch(0, 4)

    def batch():
        return train_step.Batch.from_collate(bt["text"], bt["input_lengths"], bt["mel"], bt["y"], bt["text_positions"],
                                             bt["frame_positions"], bt["done"], bt["target_lengths"], None,
                                             downsample_step=4, device=dev)

    def weights(pg, graphed, steps=3):
        torch.manual_seed(0)
        model = builder.deepvoice3(**HP).to(dev)
        tr = train_step.Trainer(model, train_step.TrainConfig(max_positions=128), process_group=pg, bucket_mb=0.05)
        info = {}
        if graphed:
            g = train_step.GraphedTrainer(tr, batch(), warmup=1, chunk=2)
            for _ in range(steps - 1):
                scal = g.step()
            info["split"], info["seg_buckets"], info["rest_buckets"] = g.split, g.seg_buckets, g.rest_buckets
            g.close()
        else:
            if tr.comm is not None:
                tr.comm.exposed_events = []
            b = batch()
            for _ in range(steps):
                scal = tr.step(b)
        torch.cuda.synchronize()
        if tr.comm is not None:
            info["buckets"] = len(tr.comm.buckets)
            if not graphed:
                info["exposed_ms"] = tr.comm.exposed_ms()
        w = tr.arena.flat.cpu().numpy().copy()
        gn = float(scal["grad_norm"])
        tr.close()
        return w, gn, info

    plain = weights(None, False)
    pg, rank, world, local_rank = dv3dist.init_from_env(allow_single=True)   # backend "nccl" (= RCCL) on a GPU box
    assert world == 1 and dist.get_backend(pg) == "nccl"
    eager = weights(pg, False)
    graphed = weights(pg, True)
    graphed_plain = weights(None, True)
    maps = open("/proc/self/maps").read()
    q.put(dict(plain=plain, eager=eager, graphed=graphed, graphed_plain=graphed_plain,
               rccl_loaded="librccl" in maps, dv3_loaded="libdv3hip.so" in maps, ranks=dist.get_world_size(pg)))
    dist.destroy_process_group()


def test_world1_nccl_group_step_is_the_plain_step_eager_and_captured():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_run_nccl, args=(29900 + os.getpid() % 500, q))
    p.start()
    res = None
    for _ in range(180):           # poll: a crashed child must fail the test at once, not after a long queue timeout
        try:
            res = q.get(timeout=1.0)
            break
        except Exception:
            if not p.is_alive():
                break
    if res is None:
        p.kill()
        pytest.fail("the RCCL child process ended without a result (exit code %r)" % (p.exitcode,))
    p.join(60)
    assert p.exitcode == 0
    assert res["rccl_loaded"] and res["dv3_loaded"] and res["ranks"] == 1
    w0, g0, _ = res["plain"]
    w1, g1, info1 = res["eager"]
    w2, g2, info2 = res["graphed"]
    w3, g3, _ = res["graphed_plain"]
    assert info1["buckets"] >= 2 and info1["exposed_ms"] is not None and info1["exposed_ms"] >= 0.0
    assert np.array_equal(w0, w1) and g0 == g1, "RCCL world-1 eager step differs from the no-group step"
    assert np.array_equal(w0, w3) and g0 == g3, "whole-step hipGraph differs from the eager step"
    assert np.array_equal(w0, w2) and g0 == g2, "replayed step with the RCCL all-reduces armed differs from the eager step"
    # the segmented replay issues every bucket exactly once, from the host, and not all of them at the end
    assert info2["split"]
    issued = [b for bs in info2["seg_buckets"] for b in bs] + list(info2["rest_buckets"])
    assert sorted(issued) == list(range(info2["buckets"])), (info2["seg_buckets"], info2["rest_buckets"])
    assert any(info2["seg_buckets"][:-1]), "no bucket became complete before the last segment: no overlap with backward"
