# coding: utf-8
"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol
include/dv3hip.h declares, the module tree mirrors the reference's state_dict, the product path
refuses to run without a GPU, the data-parallel gradient exchange works (gloo, world_size 2)."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

from tests.util import MODEL_FIXTURES, load_golden, split_model_fixture, ROOT


def test_library_exports_every_declared_symbol():
    from deepvoice3_pytorch_amd import _lib
    structs, funcs, consts = _lib.parse_header()
    assert len(funcs) >= 25 and "dv3_conv_gemm_f32" in funcs and "dv3_clip_adam_f32" in funcs
    h = ctypes.CDLL(os.path.join(ROOT, "deepvoice3_pytorch_amd", "libdv3hip.so"))
    for name in funcs:
        assert hasattr(h, name), name
    lib = _lib.lib()   # also checks ABI version and sizeof of every descriptor struct
    assert lib.dv3_abi_version() == consts["DV3_ABI_VERSION"]
    for name in structs:
        assert lib.dv3_sizeof(name.encode()) == ctypes.sizeof(_lib.STRUCTS[name])
    assert lib.dv3_sizeof(b"nope") == -1


def test_entry_points_reject_bad_arguments_without_a_gpu():
    """Argument validation happens before any launch, so it is testable on CPU."""
    from deepvoice3_pytorch_amd import _lib
    lib = _lib.lib()
    d = _lib.STRUCTS["dv3_conv_desc"]()
    assert lib.dv3_conv_gemm_f32(ctypes.byref(d), None) == _lib.CONSTS["DV3_EINVAL"]
    assert b"null pointer" in lib.dv3_last_error()
    assert lib.dv3_dropout_bits(None, 0, 0.1, 0, 0, None, None) == _lib.CONSTS["DV3_EINVAL"]


def test_ops_have_no_cpu_fallback():
    from deepvoice3_pytorch_amd import ops, builder
    with pytest.raises(RuntimeError):
        ops.conv_layer(torch.zeros(1, 4, 8), torch.zeros(8, 4, 3), None, None, ops.LayerCfg(k=3))
    m = builder.deepvoice3(n_vocab=10, embed_dim=8, mel_dim=4, linear_dim=5, r=1, downsample_step=4,
                           kernel_size=3, encoder_channels=8, decoder_channels=8, converter_channels=8)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, dtype=torch.long), torch.zeros(1, 2, 4))


@pytest.mark.parametrize("name", MODEL_FIXTURES)
def test_state_dict_matches_reference(name):
    """Same keys and shapes as the reference's state_dict (tests/golden holds the reference's);
    load_state_dict of a reference checkpoint must work, before and after make_generation_fast_."""
    from deepvoice3_pytorch_amd import builder
    fx = load_golden("model_" + name)
    b, hp, sd, _ = split_model_fixture(fx)
    model = getattr(builder, b)(**hp)
    msd = model.state_dict()
    assert set(msd) == set(sd)
    for k in sd:
        assert tuple(msd[k].shape) == tuple(sd[k].shape), k
    model.load_state_dict(sd)
    n_train = sum(p.numel() for p in model.get_trainable_parameters())
    n_all = sum(p.numel() for p in model.parameters())
    dec = model.seq2seq.decoder
    assert n_all - n_train == dec.embed_query_positions.weight.numel() + dec.embed_keys_positions.weight.numel()
    model.make_generation_fast_()
    keys = set(model.state_dict())
    assert not any(k.endswith(("weight_g", "weight_v")) for k in keys)
    # folded weight == g * v / ||v||
    from oracle import dv3_oracle as O
    for k in keys:
        if k.endswith(".weight") and (k[:-7] + ".weight_g") in sd:
            assert torch.allclose(model.state_dict()[k], O.wn_weight(sd, k[:-7]), atol=1e-6), k


def test_builder_errors_match_reference():
    from deepvoice3_pytorch_amd import builder
    with pytest.raises(ValueError):
        builder.nyanko(n_vocab=10, n_speakers=2)
    with pytest.raises(ValueError):
        builder.nyanko(n_vocab=10, r=2)
    with pytest.raises(ValueError):
        builder.deepvoice3(n_vocab=10, r=1, downsample_step=8)       # time_upsampling 8: "Not supported"
    from deepvoice3_pytorch_amd.conv import Conv1d
    c = Conv1d(2, 4, 3).train()
    with pytest.raises(RuntimeError, match="incremental_forward only supports eval mode"):
        c.incremental_forward(torch.zeros(1, 1, 2))


def test_position_table_matches_reference():
    from deepvoice3_pytorch_amd.modules import position_encoding_init
    fx = load_golden("misc")
    assert np.array_equal(position_encoding_init(64, 24, 1.0).numpy(), fx["pe/table_r1.0"])
    assert np.array_equal(position_encoding_init(64, 24, 1.385).numpy(), fx["pe/table_r1.385"])
    assert np.array_equal(position_encoding_init(64, 24, 1.0, sinusoidal=False).numpy(), fx["pe/raw"])


def test_lr_schedule_and_batch_rules():
    from deepvoice3_pytorch_amd import train_step
    fx = load_golden("losses")
    for step in (0, 10, 3999, 4000, 100000):
        assert abs(train_step.noam_learning_rate_decay(5e-4, step) - float(fx["noam/%d" % step])) < 1e-18
    sys.path.insert(0, ROOT)
    import bench
    bt = bench.synth_batch(np.random.RandomState(0), 3, 50, 203, bench.DV3_LJ)
    # train.collate_fn: 203 -> 204 (multiple of 4) + b_pad * downsample_step = 208; Td = 52
    assert bt["mel"].shape == (3, 208, 80) and bt["y"].shape == (3, 208, 513)
    assert bt["frame_positions"].shape == (3, 52) and bt["done"].shape == (3, 52, 1)
    assert float(bt["mel"][:, 0].abs().sum()) == 0.0 and int(bt["text"][0, 49]) == 1
    assert float(bt["done"][0, : 203 // 4 - 1].sum()) == 0.0 and float(bt["done"][0, 203 // 4 - 1:].min()) == 1.0


def test_checkpoint_interop_with_torch_adam_layout(tmp_path):
    """train_step.save/load_checkpoint speak the reference's checkpoint format (train.py:788-867):
    the optimizer entry loads into a torch.optim.Adam over get_trainable_parameters() and back."""
    from deepvoice3_pytorch_amd import builder, train_step
    hp = dict(n_vocab=20, embed_dim=16, mel_dim=8, linear_dim=9, r=1, downsample_step=4, padding_idx=0, dropout=0.0,
              kernel_size=3, encoder_channels=16, decoder_channels=16, converter_channels=16, max_positions=32)
    torch.manual_seed(0)
    model = builder.deepvoice3(**hp)
    tr = train_step.Trainer(model, train_step.TrainConfig(max_positions=32), global_step=41)
    tr.adam_step = 7
    tr.arena.exp_avg.copy_(torch.randn_like(tr.arena.exp_avg))
    tr.arena.exp_avg_sq.copy_(torch.rand_like(tr.arena.exp_avg_sq))
    path = train_step.save_checkpoint(tr, str(tmp_path), global_epoch=3)
    assert os.path.basename(path) == "checkpoint_step000000041.pth"
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert sorted(ck) == ["global_epoch", "global_step", "optimizer", "state_dict"]
    # the reference side: a fresh model + torch Adam over get_trainable_parameters() takes it as is
    torch.manual_seed(1)
    model2 = builder.deepvoice3(**hp)
    model2.load_state_dict(ck["state_dict"])
    opt = torch.optim.Adam(model2.get_trainable_parameters(), lr=5e-4, betas=(0.5, 0.9), eps=1e-6)
    opt.load_state_dict(ck["optimizer"])
    p0 = next(iter(model2.get_trainable_parameters()))
    assert torch.equal(opt.state[p0]["exp_avg"], tr.arena.exp_avg[:p0.numel()].view(p0.shape))
    assert float(opt.state[p0]["step"]) == 7.0
    # ... and a checkpoint written by that optimizer comes back into a fresh trainer
    ref_ck = {"state_dict": model2.state_dict(), "optimizer": opt.state_dict(), "global_step": 1234, "global_epoch": 5}
    torch.manual_seed(2)
    tr3 = train_step.Trainer(builder.deepvoice3(**hp), train_step.TrainConfig(max_positions=32))
    assert train_step.load_checkpoint(ref_ck, tr3) == 5
    assert tr3.global_step == 1234 and tr3.adam_step == 7
    for o, n in zip(tr.arena.offsets, tr.arena.sizes):           # (the 4-element alignment pads are not state)
        assert torch.equal(tr3.arena.exp_avg[o:o + n], tr.arena.exp_avg[o:o + n])
        assert torch.equal(tr3.arena.exp_avg_sq[o:o + n], tr.arena.exp_avg_sq[o:o + n])
        assert torch.equal(tr3.arena.flat[o:o + n], tr.arena.flat[o:o + n])     # parameters live in the arena
    for (k, v), (k2, v2) in zip(tr3.model.state_dict().items(), model.state_dict().items()):
        assert k == k2 and torch.equal(v, v2)


def test_collate_fn_matches_reference():
    """deepvoice3_pytorch_amd.data.collate_fn against the reference's own train.collate_fn
    (tests/golden/collate.npz, generated by oracle/make_golden.py from the unmodified reference):
    ragged items, r in {1, 2, 4}, downsample_step in {1, 4}, with and without speaker ids -- bit exact."""
    from deepvoice3_pytorch_amd import data
    fx = load_golden("collate")
    cases = sorted({k.split("/")[0] for k in fx})
    assert len(cases) == 4
    for c in cases:
        r, ds = [int(v) for v in fx[c + "/r_ds"]]
        items, i = [], 0
        while "%s/item%d/text" % (c, i) in fx:
            it = (fx["%s/item%d/text" % (c, i)], fx["%s/item%d/mel" % (c, i)], fx["%s/item%d/y" % (c, i)])
            if "%s/item%d/spk" % (c, i) in fx:
                it = it + (int(fx["%s/item%d/spk" % (c, i)]),)
            items.append(it)
            i += 1
        x, in_len, mel, y, (tp, fp), done, tgt_len, sid = data.collate_fn(items, outputs_per_step=r, downsample_step=ds)
        got = dict(x=x, input_lengths=in_len, mel=mel, y=y, text_positions=tp, frame_positions=fp, done=done,
                   target_lengths=tgt_len)
        for k, v in got.items():
            want = fx["%s/out/%s" % (c, k)]
            assert v.shape == want.shape, (c, k, v.shape, want.shape)
            assert np.array_equal(v.numpy(), want), (c, k)
        if sid is not None:
            assert np.array_equal(sid.numpy(), fx[c + "/out/speaker_ids"])
        else:
            assert c + "/out/speaker_ids" not in fx


def test_preprocessed_dataset_reads_reference_layout(tmp_path):
    """train.txt + .npy files as preprocess.py writes them (preprocess.py:27-31) -> collate_fn items"""
    from deepvoice3_pytorch_amd import data
    rng = np.random.RandomState(3)
    lines = []
    for i, (n, spk) in enumerate([(12, 0), (20, 1), (9, 0)]):
        np.save(str(tmp_path / ("spec-%05d.npy" % i)), rng.rand(n, 7).astype(np.float32))
        np.save(str(tmp_path / ("mel-%05d.npy" % i)), rng.rand(n, 5).astype(np.float32))
        lines.append("spec-%05d.npy|mel-%05d.npy|%d|text number %d|%d" % (i, i, n, i, spk))
    (tmp_path / "train.txt").write_text("\n".join(lines) + "\n", encoding="utf-8")
    t2s = lambda t: [ord(c) % 20 + 2 for c in t] + [1]
    ds = data.PreprocessedDataset(str(tmp_path), t2s)
    assert len(ds) == 3 and ds.multi_speaker and ds.frame_lengths == [12, 20, 9]
    text, mel, spec, spk = ds[1]
    assert mel.shape == (20, 5) and spec.shape == (20, 7) and spk == 1 and text[-1] == 1
    one = data.PreprocessedDataset(str(tmp_path), t2s, speaker_id=0)      # train.py:113-119
    assert len(one) == 2 and not one.multi_speaker and len(one[0]) == 3
    x, in_len, m, y, (tp, fp), done, tgt, sid = data.collate_fn([ds[i] for i in range(3)])
    assert m.shape == (3, 24, 5) and y.shape == (3, 24, 7) and list(tgt) == [12, 20, 9] and list(sid) == [0, 1, 0]


def _ddp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from deepvoice3_pytorch_amd import dist as dv3dist
    from deepvoice3_pytorch_amd.train_step import FlatArena
    pg, r, w, _ = dv3dist.init_from_env(backend="gloo")
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(n)) for n in (1000, 37, 4096, 5, 20000)]
    arena = FlatArena(params)
    comm = dv3dist.BucketedAllReduce(arena, pg, bucket_mb=0.01)   # ~2.6k floats per bucket -> several buckets
    assert len(comm.buckets) >= 3
    for step in range(2):
        arena.grad.zero_()
        comm.arm()
        # each rank's loss weights its parameters differently; param 3 gets no gradient at all
        loss = sum(((i + 1) * (rank + 1 + step)) * p.sum() for i, p in enumerate(params) if i != 3)
        loss.backward()
        comm.finish()
        for i, p in enumerate(params):
            want = 0.0 if i == 3 else (i + 1) * sum(rr + 1 + step for rr in range(world))
            assert torch.allclose(p.grad, torch.full_like(p.grad, want)), (rank, i)
    # in-place path (ops.ConvLayerFn writes conv-layer gradients straight into p.grad and reports
    # them through ops.grad_ready_hooks): params 0, 2, 4 in place, 1 through autograd, 3 untouched
    from deepvoice3_pytorch_amd import ops
    assert comm._on_inplace_grad in ops.grad_ready_hooks
    arena.grad.zero_()
    comm.arm()
    (2.0 * (rank + 1) * params[1].sum()).backward()
    for i in (4, 0, 2):
        params[i].grad.add_(float((i + 1) * (rank + 1)))
        for hook in ops.grad_ready_hooks:
            hook(params[i])
        # autograd's AccumulateGrad hook fires for such a parameter as well (torch 2.10): a second
        # report must not count twice, or buckets launch before their other gradients exist
        comm._make_hook(i)(params[i])
    assert all(comm.launched[comm.bucket_of[i]] for i in (0, 1, 2, 4) if
               all(j in (0, 1, 2, 4) for j in comm.buckets[comm.bucket_of[i]][2]))
    comm.finish()
    tot = sum(rr + 1 for rr in range(world))
    for i, p in enumerate(params):
        want = 0.0 if i == 3 else (2.0 if i == 1 else float(i + 1)) * tot
        assert torch.allclose(p.grad, torch.full_like(p.grad, want)), (rank, i)
    # the segmented replay's schedule (train_step.GraphedTrainer): while the step is captured the completed buckets are
    # only NOTED, segment by segment; the replay then issues exactly those all-reduces from the host after each segment
    # and the never-completed rest before the optimiser -- every bucket once, same sums as the eager step
    arena.grad.zero_()
    comm.arm()
    ops.SideStream.split_capture = True
    try:
        seg_buckets = []
        for seg in ((4,), (2, 0), (1,)):          # parameter 3 never reports
            for i in seg:
                comm._on_inplace_grad(params[i])
            seg_buckets.append(comm.take_completed())
    finally:
        ops.SideStream.split_capture = False
    comm.disarm()
    seen = [b for bs in seg_buckets for b in bs]
    rest = [b for b in range(len(comm.buckets)) if b not in seen]
    assert len(set(seen)) == len(seen) and rest and not any(comm.launched)
    for step in range(2):                         # two "replays"
        arena.grad.zero_()
        for i, p in enumerate(params):
            if i != 3:
                p.grad.add_(float((i + 1) * (rank + 1 + step)))
        for bs in seg_buckets:
            if bs:
                comm.launch_after(bs, ())
        comm.launch_after(rest, ())
        comm.join()
        for i, p in enumerate(params):
            want = 0.0 if i == 3 else (i + 1) * sum(rr + 1 + step for rr in range(world))
            assert torch.allclose(p.grad, torch.full_like(p.grad, want)), (rank, i, step)
    q.put((rank, float(arena.grad.sum())))
    dist.destroy_process_group()


def test_bucketed_allreduce_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(2))
    assert res[0][1] == res[1][1]      # both ranks hold the same reduced gradient arena


def test_bucketed_allreduce_and_segment_schedule_gloo_world8():
    """the same worker on EIGHT ranks (the node BASELINE.json's metric is defined on): bucketed all-reduce, in-place
    notifications, and the segmented replay's note-then-issue schedule give every rank the 8-rank sums"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_ddp_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(8))
    assert [r for r, _ in res] == list(range(8)) and len(set(v for _, v in res)) == 1


def test_bucket_cutting_isolates_the_shared_table_and_segment_notes():
    """host logic of dist.BucketedAllReduce that needs no process group: an isolated parameter (the speaker table) is a
    bucket of its own and every parameter is in exactly one contiguous bucket; while a step is captured as segment
    graphs a completed bucket is NOTED, not launched (train_step.GraphedTrainer issues it from the host at replay)."""
    from deepvoice3_pytorch_amd import dist as dv3dist, ops
    from deepvoice3_pytorch_amd.train_step import FlatArena
    params = [torch.nn.Parameter(torch.randn(n)) for n in (3000, 40, 5000, 7, 2500, 16)]
    arena = FlatArena(params)
    comm = dv3dist.BucketedAllReduce(arena, None, bucket_mb=0.02, last_bucket_mb=None, isolate=[5, 1])
    try:
        assert [pl for _, _, pl in comm.buckets if 5 in pl] == [[5]] and [pl for _, _, pl in comm.buckets if 1 in pl] == [[1]]
        assert sorted(i for _, _, pl in comm.buckets for i in pl) == list(range(6))
        spans = sorted((lo, hi) for lo, hi, _ in comm.buckets)
        assert spans[0][0] == 0 and spans[-1][1] == arena.total and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        launched = []
        comm._launch = lambda b: launched.append(b)
        comm.arm()
        ops.SideStream.split_capture = True
        try:
            comm._on_inplace_grad(params[5])
            assert comm.take_completed() == [comm.bucket_of[5]] and comm.take_completed() == [] and launched == []
        finally:
            ops.SideStream.split_capture = False
        comm._on_inplace_grad(params[1], None)
        assert launched == [comm.bucket_of[1]]
    finally:
        comm.close()


def _late_group_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    import torch.distributed as dist
    from deepvoice3_pytorch_amd import builder, train_step
    dist.init_process_group("gloo", rank=0, world_size=1)
    hp = dict(n_vocab=20, embed_dim=16, mel_dim=8, linear_dim=17, r=1, downsample_step=4, n_speakers=3, speaker_embed_dim=16,
              padding_idx=0, dropout=0.05, kernel_size=3, encoder_channels=16, decoder_channels=16, converter_channels=16,
              max_positions=64)
    torch.manual_seed(0)
    m1, m2 = builder.deepvoice3_multispeaker(**hp), builder.deepvoice3_multispeaker(**hp)
    m2.load_state_dict(m1.state_dict())
    cfg = train_step.TrainConfig(max_positions=64)
    t1 = train_step.Trainer(m1, cfg, process_group=dist.group.WORLD, bucket_mb=0.01, last_bucket_mb=None)
    t2 = train_step.Trainer(m2, cfg)
    names1 = {id(p): n for n, p in m1.named_parameters()}
    late = [names1[id(p)] for p in t1.late_group]
    n_late = len(late)
    ok = n_late > 0 and all(".speaker_proj." in n for n in late)
    ok = ok and [id(p) for p in t1.arena.params[-n_late:]] == [id(p) for p in t1.late_group]     # a group at the arena's tail
    ok = ok and not any(".speaker_proj." in names1[id(p)] for p in t1.arena.params[:-n_late])
    ok = ok and not t2.late_group and [id(p) for p in t2.arena.params] == [id(p) for p in t2.optimizer_order]
    # ... in buckets of its own: no bucket mixes the group with other parameters
    first = len(t1.arena.params) - n_late
    for lo, hi, plist in t1.comm.buckets:
        ok = ok and (all(i >= first for i in plist) or all(i < first for i in plist))
    # checkpoint interop: the optimizer state is numbered in get_trainable_parameters() order whatever the arena's order
    t1.adam_step = 3
    for k, p in enumerate(t1.optimizer_order):
        o, n = next((o, n) for o, n, q_ in zip(t1.arena.offsets, t1.arena.sizes, t1.arena.params) if q_ is p)
        t1.arena.exp_avg[o:o + n] = float(k + 1)
        t1.arena.exp_avg_sq[o:o + n] = float(2 * k + 1)
    ck = train_step.checkpoint_dict(t1)
    ref = torch.optim.Adam(list(m2.get_trainable_parameters()))
    ok = ok and len(ck["optimizer"]["state"]) == len(ref.param_groups[0]["params"])
    for k, p in enumerate(m1.get_trainable_parameters()):
        st = ck["optimizer"]["state"][k]
        ok = ok and st["exp_avg"].shape == p.shape and float(st["exp_avg"].flatten()[0]) == k + 1
    train_step.load_checkpoint(ck, t2)
    for k, p in enumerate(t2.optimizer_order):
        o, n = next((o, n) for o, n, q_ in zip(t2.arena.offsets, t2.arena.sizes, t2.arena.params) if q_ is p)
        ok = ok and float(t2.arena.exp_avg[o]) == k + 1 and float(t2.arena.exp_avg_sq[o + n - 1]) == 2 * k + 1
    ok = ok and t2.adam_step == 3
    q.put(bool(ok))
    t1.close()
    dist.destroy_process_group()


def test_speaker_projections_form_a_late_group_under_data_parallel():
    """Data parallel + multi-speaker: the Conv1dGLU speaker projections (final only when their block's fused backward
    node has run) sit at the arena's tail in buckets of their own, and the checkpoint keeps torch.optim.Adam's numbering
    (train.py:800-807 interop) although the arena's order differs from get_trainable_parameters()."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_late_group_worker, args=(29700 + os.getpid() % 2000, q))
    p.start()
    p.join(120)
    assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_pack_batch_layout():
    """data.pack_batch: items back to back, no padding; padded_frames restates the frame arithmetic of
    train.collate_fn (train.py:307-316) that collate_fn itself is pinned on"""
    from deepvoice3_pytorch_amd import data
    rng = np.random.RandomState(4)
    items = [(rng.randint(1, 20, L).astype(np.int32), rng.rand(F, 6).astype(np.float32), rng.rand(F, 9).astype(np.float32), k)
             for k, (L, F) in enumerate([(5, 11), (9, 30), (2, 17)])]
    pb = data.pack_batch(items, pin=False)
    assert pb.text.dtype == torch.int64 and pb.text.shape == (16,) and pb.mel.shape == (58, 6) and pb.lin.shape == (58, 9)
    assert np.array_equal(pb.in_len, [5, 9, 2]) and np.array_equal(pb.tgt_len, [11, 30, 17])
    assert np.array_equal(pb.mel[11:41].numpy(), items[1][1]) and np.array_equal(pb.text[5:14].numpy(), items[1][0])
    assert np.array_equal(pb.speaker_ids, [0, 1, 2])
    for r, ds in ((1, 4), (2, 1), (4, 4), (1, 1)):
        T, b_pad = data.padded_frames(pb.tgt_len, r, ds)
        col = data.collate_fn(items, outputs_per_step=r, downsample_step=ds)
        assert col[2].shape[1] == T and b_pad == r
    with pytest.raises(ValueError):
        data.pack_batch([(items[0][0], items[0][1], items[1][2])], pin=False)


def test_restore_parts_and_load_embedding():
    """train.restore_parts / train._load_embedding (train.py:870-897): partial restore by name, entries
    of another shape skipped with the reference's warning, the rest of the model untouched"""
    from deepvoice3_pytorch_amd import builder, train_step
    hp = dict(n_vocab=20, embed_dim=16, mel_dim=8, linear_dim=9, r=1, downsample_step=4, padding_idx=0, dropout=0.0,
              kernel_size=3, encoder_channels=16, decoder_channels=16, converter_channels=16, max_positions=32)
    torch.manual_seed(0)
    src = builder.deepvoice3(**hp)
    torch.manual_seed(1)
    dst = builder.deepvoice3(**dict(hp, n_vocab=24))          # another vocabulary: the embedding table differs
    before = {k: v.clone() for k, v in dst.state_dict().items()}
    ck = {"state_dict": dict(src.state_dict(), **{"not.in.the.model": torch.zeros(3)})}
    with pytest.warns(UserWarning, match="invalid size of weight"):
        restored, skipped = train_step.restore_parts(ck, dst)
    assert skipped == ["seq2seq.encoder.embed_tokens.weight"] and "not.in.the.model" not in restored
    after = dst.state_dict()
    for k in restored:
        assert torch.equal(after[k], ck["state_dict"][k])
    assert torch.equal(after[skipped[0]], before[skipped[0]])
    assert len(restored) == len(before) - 1
    # the embedding alone, same vocabulary
    torch.manual_seed(2)
    dst2 = builder.deepvoice3(**hp)
    train_step.load_embedding(ck, dst2)
    assert torch.equal(dst2.seq2seq.encoder.embed_tokens.weight, src.seq2seq.encoder.embed_tokens.weight)
    with pytest.raises(RuntimeError):
        train_step.load_embedding(ck, dst)
    with pytest.raises(KeyError):
        train_step.load_embedding({"state_dict": {}}, dst2)
    # a seq2seq-only / postnet-only checkpoint into the sub-module (train.py:985-989)
    torch.manual_seed(3)
    dst3 = builder.deepvoice3(**hp)
    train_step.load_submodule_checkpoint({"state_dict": src.seq2seq.state_dict()}, dst3.seq2seq)
    train_step.load_submodule_checkpoint({"state_dict": src.postnet.state_dict()}, dst3.postnet)
    for k, v in src.state_dict().items():
        assert torch.equal(dst3.state_dict()[k], v), k
    with pytest.raises(RuntimeError):
        train_step.load_submodule_checkpoint({"state_dict": src.postnet.state_dict()}, dst3.seq2seq)


def _ragged_pad_rows_numpy(src, row_off, B, T_out, lead=0, t_stride=1):
    """what dv3_ragged_pad_rows_b32 computes (include/dv3hip.h), in numpy: lets the host half of
    data.device_collate run on CPU"""
    one = src.dim() == 1
    out = torch.zeros((B, T_out) if one else (B, T_out, src.shape[1]), dtype=src.dtype)
    ro = row_off.numpy()
    for b in range(B):
        n = int(ro[b + 1] - ro[b])
        t = np.arange(T_out)
        s_ = t * t_stride - lead
        ok = (s_ >= 0) & (s_ < n)
        out[b, torch.from_numpy(t[ok])] = src[torch.from_numpy(ro[b] + s_[ok])]
    return out


@pytest.mark.parametrize("seed", range(12))
def test_device_collate_host_logic_equals_collate_fn(monkeypatch, seed):
    """data.device_collate with the padding kernel emulated in numpy == to_device_batch(collate_fn(...))
    for random ragged batches: lengths of 1, r in {1, 2, 4}, downsample_step in {1, 2, 4}, frame counts
    that are and are not multiples of r * downsample_step, with and without speaker ids.  (The kernel
    itself is pinned against the reference golden on the GPU: tests/test_gpu_model.py.)"""
    from deepvoice3_pytorch_amd import data, ops
    monkeypatch.setattr(ops, "ragged_pad_rows", _ragged_pad_rows_numpy)
    rng = np.random.RandomState(100 + seed)
    r = int(rng.choice([1, 2, 4]))
    ds = int(rng.choice([1, 2, 4]))
    n = int(rng.randint(1, 7))
    multi = bool(rng.randint(2))
    items = []
    for i in range(n):
        L = int(rng.randint(1, 30))
        F = int(rng.randint(1, 70)) if rng.rand() < 0.8 else int(rng.randint(1, 6)) * r * ds
        it = (rng.randint(1, 40, L).astype(np.int32), rng.rand(F, 5).astype(np.float32), rng.rand(F, 7).astype(np.float32))
        items.append(it + (int(rng.randint(3)),) if multi else it)
    got = data.device_collate(data.pack_batch(items, pin=False), "cpu", outputs_per_step=r, downsample_step=ds)
    host = data.to_device_batch(data.collate_fn(items, outputs_per_step=r, downsample_step=ds), "cpu",
                                outputs_per_step=r, downsample_step=ds)
    for name in ("text", "text_positions", "frame_positions", "mel", "y", "done", "input_lengths", "target_lengths",
                 "decoder_lengths", "linear_mask_lengths"):
        g, h = getattr(got, name), getattr(host, name)
        assert g.dtype == h.dtype and g.shape == h.shape and torch.equal(g, h), (name, r, ds)
    assert got.n_frames == host.n_frames
    assert (got.speaker_ids is None) == (host.speaker_ids is None)
    if multi:
        assert torch.equal(got.speaker_ids, host.speaker_ids)


def test_length_bucketed_sampler_shards_are_disjoint_and_cover():
    """data.LengthBucketedSampler (the reference's PartialyRandomizedSimilarTimeLengthSampler, train.py:195-239,
    made rank-aware): the shards of one epoch are disjoint, together they are the epoch's batch list, every rank
    takes the same number of steps, batches hold items of similar length, epochs differ, seeds reproduce."""
    from deepvoice3_pytorch_amd import data
    lengths = np.random.RandomState(0).randint(100, 870, 1003)
    for world in (1, 2, 8):
        per_rank = []
        for r in range(world):
            s = data.LengthBucketedSampler(lengths, batch_size=16, rank=r, world=world, seed=7)
            b = list(s)
            assert len(b) == len(s)
            per_rank.append(b)
        assert len({len(b) for b in per_rank}) == 1
        flat = [i for b in per_rank for x in b for i in x]
        assert len(flat) == len(set(flat))
        full = data.LengthBucketedSampler(lengths, 16, seed=7, drop_last=world > 1).epoch_batches()
        assert world == 1 or all(len(b) == 16 for b in full)      # no short tail batch in a 1/world average
        usable = len(full) - len(full) % world
        assert sorted(flat) == sorted(int(i) for b in full[:usable] for i in b)
        for r in range(world):      # rank r holds batches r, r+world, ...
            assert per_rank[r] == [[int(i) for i in b] for b in full[r:usable:world]]
    # the batches cut off to equalise the step count rotate with the epoch
    s8 = data.LengthBucketedSampler(lengths, 16, rank=0, world=8, seed=7)
    nb = len(s8.epoch_batches())
    assert nb % 8 and len(list(s8)) == nb // 8
    s8.set_epoch(1)
    assert len(list(s8)) == nb // 8
    s = data.LengthBucketedSampler(lengths, 16, seed=7)
    e0 = list(s)
    assert e0 == list(data.LengthBucketedSampler(lengths, 16, seed=7))
    s.set_epoch(1)
    assert list(s) != e0
    # similar lengths inside a batch: items come from one sorted window of batch_group_size (= 32 batches, as
    # in the reference), so the spread inside a batch is bounded by that window, well below a random draw's
    spread = np.mean([np.ptp(lengths[b]) for b in e0 if len(b) == 16])
    assert spread < 0.62 * np.ptp(lengths)
    tight = data.LengthBucketedSampler(lengths, 16, batch_group_size=32, seed=7)
    assert np.mean([np.ptp(lengths[b]) for b in tight if len(b) == 16]) < 0.06 * np.ptp(lengths)
    with pytest.raises(ValueError):
        data.LengthBucketedSampler(lengths, 16, batch_group_size=40)


def _sampler_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deepvoice3_pytorch_amd import data
    lengths = np.random.RandomState(1).randint(50, 500, 333)
    s = data.LengthBucketedSampler(lengths, 8, rank=dist.get_rank(), world=dist.get_world_size(), seed=5)
    mine = torch.zeros(333, dtype=torch.int32)
    steps = 0
    for b in s:
        mine[b] += 1
        steps += 1
    tot = mine.clone()
    dist.all_reduce(tot)
    st = torch.tensor([steps, -steps])
    dist.all_reduce(st, op=dist.ReduceOp.MAX)
    q.put((rank, int(tot.max()), int((tot > 0).sum()), int(st[0]), int(-st[1])))
    dist.destroy_process_group()


def test_sampler_shards_over_gloo_world2():
    """two processes (gloo): no index is drawn by both ranks, and both take the same number of steps"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_sampler_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for _ in range(2):
        rank, mx, covered, smax, smin = q.get(timeout=5)
        assert mx == 1 and smax == smin and covered >= 333 - 2 * 8


def test_prefetcher_yields_the_collated_batches(monkeypatch):
    """data.Prefetcher (producer thread, worker threads filling staging slots, collate) hands out exactly
    to_device_batch(collate_fn(items)) for the sampler's batches, in order, and loops over epochs when asked.
    The padding kernel is emulated in numpy (CPU container); the GPU path: tests/test_gpu_model.py."""
    from deepvoice3_pytorch_amd import data, ops
    monkeypatch.setattr(ops, "ragged_pad_rows", _ragged_pad_rows_numpy)
    rng = np.random.RandomState(3)
    items = []
    for i in range(37):
        L, F = int(rng.randint(2, 30)), int(rng.randint(4, 90))
        items.append((rng.randint(1, 40, L).astype(np.int64), rng.rand(F, 5).astype(np.float32),
                      rng.rand(F, 7).astype(np.float32), int(rng.randint(4))))
    ds = data.ListDataset(items)
    sampler = data.LengthBucketedSampler(ds.frame_lengths, 4, seed=2)
    want = list(sampler)
    pf = data.Prefetcher(ds, sampler, "cpu", outputs_per_step=1, downsample_step=4, depth=2, workers=3)
    got = list(pf)
    pf.close()
    assert len(got) == len(want) == 10
    for idx, g in zip(want, got):
        h = data.to_device_batch(data.collate_fn([items[i] for i in idx], 1, 4), "cpu", 1, 4)
        for name in ("text", "text_positions", "frame_positions", "mel", "y", "done", "input_lengths",
                     "target_lengths", "speaker_ids"):
            assert torch.equal(getattr(g, name), getattr(h, name)), name
    # lattice=(16, 8): the same batches padded to the next lattice point -- zeros beyond what collate_fn pads to, the
    # batch's own maxima attached (device_collate(lattice=) == pad_to_shape(collate(...)))
    pf = data.Prefetcher(ds, sampler, "cpu", outputs_per_step=1, downsample_step=4, depth=2, workers=0, lattice=(16, 8))
    got = list(pf)
    pf.close()
    for idx, g in zip(want, got):
        h = data.to_device_batch(data.collate_fn([items[i] for i in idx], 1, 4), "cpu", 1, 4)
        Tt, Td = h.text.shape[1], h.frame_positions.shape[1]
        t_in, t_dec = data.lattice_shape(Tt, Td, 16, 8)
        hp = data.pad_to_shape(h, t_in, t_dec, 15, 7)
        assert g.text.shape == (len(idx), t_in) and g.y.shape[1] == t_dec * 4
        assert g.valid.tv.tolist() == [Tt, Td, Td, Td * 4] == hp.valid.tv.tolist()
        for name in ("text", "text_positions", "mel", "y", "input_lengths", "target_lengths", "speaker_ids"):
            assert torch.equal(getattr(g, name), getattr(hp, name)), name
        # frames of the batch: the same positions and done flags; the surplus steps: position 0, done 1
        assert torch.equal(g.frame_positions[:, :Td], h.frame_positions) and int(g.frame_positions[:, Td:].abs().sum()) == 0
        assert torch.equal(g.done[:, :Td], h.done) and (Td == t_dec or float(g.done[:, Td:].min()) == 1.0)
    pf = data.Prefetcher(ds, sampler, "cpu", depth=1, workers=0, loop=True)
    it = iter(pf)
    n = sum(1 for _ in zip(range(25), it))      # more than one epoch
    pf.close()
    assert n == 25
    # an item whose mel / linear lengths disagree surfaces in the consumer
    bad = data.ListDataset([(items[0][0], items[0][1], items[1][2])])
    pf = data.Prefetcher(bad, [[0]], "cpu", workers=0)
    with pytest.raises(ValueError):
        next(iter(pf))
    pf.close()


def _resume_fixture():
    import json
    from deepvoice3_pytorch_amd import builder, train_step
    fx = load_golden("checkpoint_resume")
    hpo = json.loads(str(fx["hp_over"]))
    hp = dict(n_vocab=149, embed_dim=hpo["text_embed_dim"], mel_dim=hpo["num_mels"], linear_dim=hpo["fft_size"] // 2 + 1,
              r=1, downsample_step=4, padding_idx=0, dropout=0.0, kernel_size=3, encoder_channels=hpo["encoder_channels"],
              decoder_channels=hpo["decoder_channels"], converter_channels=hpo["converter_channels"],
              use_memory_mask=True, force_monotonic_attention=True, use_decoder_state_for_postnet_input=True,
              max_positions=hpo["max_positions"], key_projection=True, value_projection=True)
    cfg = train_step.TrainConfig(max_positions=hpo["max_positions"])
    path = os.path.join(ROOT, "tests", "golden", str(fx["ckpt_file"]))
    return fx, hp, cfg, path


def test_checkpoint_written_by_the_reference_loads():
    """tests/golden/reference_checkpoint_step*.pth was written by the reference's OWN train.save_checkpoint
    (train.py:788-809; oracle/make_golden.py gen_checkpoint) after one train.train() step: weights, the
    torch.optim.Adam state_dict and the counters must land in the model, the flat moment arenas and the trainer;
    what train_step.save_checkpoint writes back must have the reference's layout again."""
    from deepvoice3_pytorch_amd import builder, train_step
    fx, hp, cfg, path = _resume_fixture()
    ck = torch.load(path, map_location="cpu", weights_only=False)   # as the reference loads it
    model = builder.deepvoice3(**hp)
    trainer = train_step.Trainer(model, cfg)
    epoch = train_step.load_checkpoint(path, trainer)
    assert epoch == int(ck["global_epoch"]) and trainer.global_step == int(ck["global_step"]) == 4000
    assert trainer.adam_step == 1
    for k, v in model.state_dict().items():
        assert torch.equal(v, ck["state_dict"][k]), k
    a = trainer.arena
    for i, (o, n, p) in enumerate(zip(a.offsets, a.sizes, a.params)):
        st = ck["optimizer"]["state"][i]
        assert torch.equal(a.exp_avg[o:o + n].view(p.shape), st["exp_avg"])
        assert torch.equal(a.exp_avg_sq[o:o + n].view(p.shape), st["exp_avg_sq"])
    back = train_step.checkpoint_dict(trainer, global_epoch=epoch)
    assert set(back) == set(ck) and set(back["optimizer"]) == set(ck["optimizer"])
    assert set(back["state_dict"]) == set(ck["state_dict"])
    assert sorted(back["optimizer"]["state"]) == sorted(ck["optimizer"]["state"])
    ref_opt = torch.optim.Adam(builder.deepvoice3(**hp).get_trainable_parameters())
    ref_opt.load_state_dict(back["optimizer"])        # torch accepts what we write


def test_c8_storage_host_logic():
    """bf16 storage bookkeeping that needs no GPU: group counts, which layer forms take the c8 kernels, layout test"""
    import torch
    from deepvoice3_pytorch_amd import ops
    assert [ops.c8_groups(c) for c in (1, 8, 32, 33, 80, 256, 513)] == [4, 4, 4, 8, 12, 32, 68]
    assert ops.is_c8(torch.zeros(2, 4, 5, 8, dtype=torch.bfloat16))
    assert not ops.is_c8(torch.zeros(2, 4, 5, 8)) and not ops.is_c8(torch.zeros(2, 32, 5, dtype=torch.bfloat16))
    assert not ops.is_c8(None)
    v3, v1, v5 = torch.zeros(512, 256, 3), torch.zeros(513, 256), torch.zeros(64, 64, 5)
    glu = ops.LayerCfg(k=3, dil=27, mode=ops.EPI_GLU)
    assert ops._c8_layer_ok(v3, glu, True, 200)
    assert not ops._c8_layer_ok(torch.zeros(40, 256, 3), glu, True, 200)          # gated: Cg = 20 is not a group multiple
    assert ops._c8_layer_ok(v1, ops.LayerCfg(mode=ops.EPI_SIGMOID), False, 200)   # plain layers: any channel count
    assert ops._c8_layer_ok(v1, ops.LayerCfg(mode=ops.EPI_LINEAR), True, 200)
    assert not ops._c8_layer_ok(v5, ops.LayerCfg(k=5), True, 200)                 # 5 taps: fp32 between conversions
    assert not ops._c8_layer_ok(v3, ops.LayerCfg(k=3, dil=33), True, 200)         # halo > 64
    assert not ops._c8_layer_ok(v3, ops.LayerCfg(k=3, t_out=198), True, 200)      # not a same-length layer
    assert ops._c8_layer_ok(v3, ops.LayerCfg(k=3, t_out=200), True, 200)
    assert not ops._c8_layer_ok(torch.zeros(256, 256, 2), ops.LayerCfg(k=2, transposed=True), True, 200)
    prev = ops.gemm_precision()
    try:
        ops.set_gemm_precision("bf16")
        assert ops.storage_c8() == ops.bf16_storage
        ops.set_gemm_precision("f16x3")
        assert not ops.storage_c8()
    finally:
        ops.set_gemm_precision(prev)


def test_archived_bench_line_meets_the_contract():
    """the last bench line kept under profiles/ carries every field the driver contract names (bench.py's output
    format is exercised on the GPU; this guards the archived evidence and the field names)"""
    import glob
    import json
    import os
    from tests.util import ROOT
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_bench_line.json")))
    assert lines
    d = json.loads(open(lines[-1]).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["n_gpus"] == 1
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["dtype"] in ("f16x3", "bf16x3", "f32", "bf16")
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("reference", "port")
    assert abs(d["value"] - d["config"]["global_batch"] * 800 / (d["ms_per_step"] * 1e-3)) < 0.02 * d["value"]
    # round 3: the host's issue time and the launch-mode probe are reported for the headline and for every side config
    flat = ("synth_rtf", "ddp_world1", "dv3lj_b64_ragged_epoch", "ddp_standin", "dv3lj_b64_ragged_epoch_lattice",
            "dv3lj_b16_ragged_epoch_lattice")
    for cfg in [d["config"]] + [c["config"] for k, c in d["configs"].items() if k not in flat]:
        assert cfg["host_enqueue_ms_per_step"] > 0 and "hipgraph" in cfg and "launch_bound" in cfg
        assert cfg["launch_probe"] is None or {"eager_ms_per_step", "hipgraph_ms_per_step"} <= set(cfg["launch_probe"])
    assert "roofline_wgrad" in d and d["roofline_wgrad"]["alg_bytes"] > 2e8       # g + x + dW
    # round 4: the reference's own operating points (preset batch 16, ragged LJSpeech-shaped lengths) and the
    # data-parallel step armed on one GPU (world-size-1 RCCL group, both launch modes) are in the line
    for k in ("dv3lj_b16", "dv3lj_b64_ragged", "dv3lj_b16_ragged"):
        assert d["configs"][k]["config"]["per_gpu_batch"] in (16, 64) and d["configs"][k]["value"] > 0
    assert d["configs"]["dv3lj_b64_ragged"]["config"]["lengths"].startswith("ragged")
    # round 5: an epoch of LJSpeech-shaped lengths cut by the reference's length-bucketed sampler (eager: the shape changes)
    ep = d["configs"].get("dv3lj_b64_ragged_epoch")
    if ep is not None:
        assert ep["value"] > 0 and ep["hipgraph"] is False and ep["real_frames"] > 0 and ep["host_enqueue_ms_per_step"] > 0
    # round 6: the same epoch replayed from captured steps of a lattice of padded shapes (train_step.LatticeReplay)
    for k in ("dv3lj_b64_ragged_epoch_lattice", "dv3lj_b16_ragged_epoch_lattice"):
        lt = d["configs"].get(k)
        if lt is not None and "value" in lt:
            assert lt["value"] > 0 and lt["hipgraph"] is True and lt["host_enqueue_ms_per_step"] > 0
    # round 6: the data-parallel step beside a ring stand-in (dist.RingStandin): step inflation, exposed wait, the bucket
    # schedule and the weak-scaling efficiency they predict, for the three presets and the preset's own batch
    sd = d["configs"].get("ddp_standin")
    if sd is not None:
        for k, e in sd.items():
            if isinstance(e, dict):
                assert e["ranks_modelled"] == 8 and e["channels"] >= 1 and e["busbw_gbps"] > 0
                assert e["ms_per_step"] >= 0.98 * e["no_group_ms_per_step"] and 0.5 < e["predicted_weak_scaling_efficiency"] < 1.05
                assert e["allreduce_exposed_ms"] is not None and len(e["bucket_mb"]) >= 2
                assert abs(e["wire_ms_per_step"] - sum(e["wire_ms_per_bucket"])) < 0.01
    w1 = d["configs"]["ddp_world1"]
    assert w1["backend"] == "nccl" and w1["rccl_ranks"] == 1
    for k, e in w1.items():
        if isinstance(e, dict):
            assert e["gradient_buckets"] >= 2 and e["autograd_hooks_after_first_step"] < 10
            for mode in ("eager", "hipgraph"):
                assert e[mode]["host_enqueue_ms_per_step"] > 0 and e[mode]["ms_per_step"] > 0


def test_bench_probes_the_launch_mode_on_every_world_size(monkeypatch):
    """bench.launch_mode: --graph / --no-graph force a mode; otherwise eager and the segmented replay are probed against
    each other, with more than one rank too (the replay captures nothing of the process group); DV3_BENCH_DDP_GRAPH=0
    keeps more than one rank eager."""
    import argparse
    import bench
    a = argparse.Namespace(no_graph=False, graph=False)
    monkeypatch.delenv("DV3_BENCH_DDP_GRAPH", raising=False)
    assert bench.launch_mode(a, 1) == "auto" and bench.launch_mode(a, 8) == "auto"
    monkeypatch.setenv("DV3_BENCH_DDP_GRAPH", "0")
    assert bench.launch_mode(a, 1) == "auto" and bench.launch_mode(a, 8) is False
    assert bench.launch_mode(argparse.Namespace(no_graph=True, graph=False), 8) is False
    assert bench.launch_mode(argparse.Namespace(no_graph=False, graph=True), 8) is True


@pytest.mark.parametrize("n", [2, 8])
def test_bench_self_launches_its_ranks_dry(n):
    """`python bench.py --gpus N` with no launcher around it must start N ranks itself (the driver's command
    line; N = 8 is the node BASELINE.json's metric is defined on); --dry-launch keeps it to the rendezvous + bucketed
    all-reduce so it runs on a CPU-only box (gloo)."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--dry-launch"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, universal_newlines=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout            # rank 0 alone prints, one line
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["rccl_ranks"] == n and out["config"]["parallelism"] == "dp%d" % n
    assert out["dry_launch"] is True and out["buckets"] >= 2
    if not torch.cuda.is_available():
        assert out["backend"] == "gloo"


def test_step_trace_collects_the_decode_loop_outputs():
    """decode_program.StepTrace (round 6: the module-by-module decode loops write into stacked buffers): growth past
    the initial capacity, the reference's stop rule (deepvoice3.py:469-473), the result shapes"""
    from deepvoice3_pytorch_amd.decode_program import StepTrace
    B, F, Tk, C = 3, 5, 7, 4
    tr = StepTrace(min_steps=2, max_steps=100, teacher_forced=False)
    outs = []
    for t in range(150):
        o = torch.full((B, 1, F), float(t))
        done = torch.full((B, 1, 1), 0.9 if t >= 120 else 0.1)
        tr.push(o, torch.full((B, 1, Tk), t + 0.5), torch.full((B, 1, C), -float(t)), done)
        outs.append(o)
        assert torch.equal(tr.last_output, o)
        if tr.stop(done):
            break
    assert tr.n == 101                                  # max_steps: t > max_decoder_steps ends the loop
    out, ali, dones, st = tr.result()
    assert out.shape == (B, 101, F) and ali.shape == (B, 101, Tk) and st.shape == (B, 101, C) and len(dones) == 101
    assert torch.equal(out, torch.cat(outs, 1)) and float(ali[0, 70, 0]) == 70.5 and float(st[2, 100, 3]) == -100.0
    # every item signals done past min_steps: stop; teacher forcing never stops by itself
    tr = StepTrace(2, 100, False)
    for t in range(5):
        done = torch.full((B, 1, 1), 0.9)
        tr.push(torch.zeros(B, 1, F), torch.zeros(B, 1, Tk), torch.zeros(B, 1, C), done)
        if tr.stop(done):
            break
    assert tr.n == 3
    tf = StepTrace(2, 3, True)
    tf.push(torch.zeros(B, 1, F), torch.zeros(B, 1, Tk), torch.zeros(B, 1, C), torch.ones(B, 1, 1))
    assert not tf.stop(torch.ones(B, 1, 1))


def test_round6_host_switches():
    """TrainConfig accepts the reference's amsgrad=False and refuses True explicitly (train.py:975-979); the audio default
    is the hop-normalised lws window in product and oracle; the ring stand-in's wire time"""
    from deepvoice3_pytorch_amd import train_step, audio, dist
    from oracle import audio_oracle as A
    assert train_step.TrainConfig(amsgrad=False).amsgrad is False
    with pytest.raises(ValueError):
        train_step.TrainConfig(amsgrad=True)
    assert abs(audio.AudioConfig().window_scale - np.sqrt(0.5)) < 1e-12 and audio.AudioConfig(window_scale=1.0).window_scale == 1.0
    assert abs(audio.AudioConfig().window_scale - A.lws_scale()) < 1e-15
    # 2 (n - 1) / n x S / busbw: 25 MiB over an 8-rank ring at 150 GB/s
    ms = dist.RingStandin.ideal_ms(type("S", (), dict(world=8, busbw_gbps=150.0))(), 25 << 20)
    assert abs(ms - 2 * 7 / 8 * (25 << 20) / 150e9 * 1e3) < 1e-9


def test_sub_module_checkpoint_loads_under_a_joint_trainer():
    """the reference loads its `_seq2seq` / `_postnet` checkpoints into model.seq2seq / model.postnet while
    training the whole model (train.py:986-990): load_checkpoint finds the sub-module by the file's keys (or takes
    module=), loads its moments and leaves the others' and the joint step count alone"""
    from deepvoice3_pytorch_amd import builder, train_step
    hp = dict(n_vocab=20, embed_dim=16, mel_dim=8, linear_dim=9, r=1, downsample_step=4, padding_idx=0, dropout=0.0,
              kernel_size=3, encoder_channels=16, decoder_channels=16, converter_channels=16, max_positions=32)
    torch.manual_seed(0)
    src = train_step.Trainer(builder.deepvoice3(**hp), train_step.TrainConfig(max_positions=32), global_step=9,
                             train_seq2seq=False, train_postnet=True)
    src.adam_step = 4
    src.arena.exp_avg.copy_(torch.randn_like(src.arena.exp_avg))
    src.arena.exp_avg_sq.copy_(torch.rand_like(src.arena.exp_avg_sq))
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        path = train_step.save_checkpoint(src, td, global_epoch=1)
        assert path.endswith("_postnet.pth")
        torch.manual_seed(1)
        joint = train_step.Trainer(builder.deepvoice3(**hp), train_step.TrainConfig(max_positions=32))
        joint.adam_step = 11
        before = joint.arena.exp_avg.clone()
        assert train_step.load_checkpoint(path, joint, unsafe=True) == 1
    for (k, v), (k2, v2) in zip(joint.model.postnet.state_dict().items(), src.model.postnet.state_dict().items()):
        assert k == k2 and torch.equal(v, v2)
    assert joint.adam_step == 11 and joint.global_step == 9          # a partial optimizer state keeps the joint count
    slot = {id(p): (o, n) for o, n, p in zip(joint.arena.offsets, joint.arena.sizes, joint.arena.params)}
    post = set(id(p) for p in joint.model.postnet.parameters())
    touched = sum(1 for p in joint.arena.params if id(p) in post and not torch.equal(
        joint.arena.exp_avg[slot[id(p)][0]:slot[id(p)][0] + slot[id(p)][1]], before[slot[id(p)][0]:slot[id(p)][0] + slot[id(p)][1]]))
    untouched = all(torch.equal(joint.arena.exp_avg[slot[id(p)][0]:slot[id(p)][0] + slot[id(p)][1]],
                                before[slot[id(p)][0]:slot[id(p)][0] + slot[id(p)][1]])
                    for p in joint.arena.params if id(p) not in post)
    assert touched > 0 and untouched


def test_mask_plan_state_machine():
    """ops.MaskPlan (host side of dv3_dropout_keep_c8_multi): a plan exists only after two consecutive steps drew the same
    list of (position, kind, shape, p); a step that differs drops it; outside begin_step / end_step nothing is recorded"""
    from deepvoice3_pytorch_amd import ops
    st, plan = ops.dropout_state, ops.MaskPlan()
    st.manual_seed(5)

    def step(shapes):
        plan.rec, plan.s0, plan.ready = [], st.site, {}          # begin_step without the launch (no GPU here)
        for kind, shape in shapes:
            st.next_site()
            assert plan.take(kind, *shape, 0.05) is None          # nothing was pre-drawn
        plan.end_step()
    a = [("both", (2, 64, 50)), ("keep", (2, 64, 50)), ("keep", (2, 128, 25))]
    assert plan.take("keep", 2, 64, 50, 0.05) is None and plan.rec is None
    step(a)
    assert plan.plan is None and len(plan.last) == 3
    step(a)
    assert plan.plan == plan.last and [s[0] for s in plan.plan] == [1, 2, 3]
    step(a[:2])
    assert plan.plan is None
    step(a[:2])
    assert plan.plan is not None and len(plan.plan) == 2
    step([])
    assert plan.plan is None
    # a pre-drawn mask is handed out once, to the call at its position with its signature only
    plan.rec, plan.s0 = [], st.site
    sig = (1, "keep", 2, 64, 50, 0.05)
    plan.ready = {1: (sig, "mask")}
    st.next_site()
    assert plan.take("keep", 2, 64, 51, 0.05) is None
    plan.ready = {2: ((2,) + sig[1:], "mask")}
    st.next_site()
    assert plan.take("keep", 2, 64, 50, 0.05) == "mask" and not plan.ready
    plan.end_step()


def test_valid_lengths_host_logic():
    """ops.ValidLengths / data.lattice_shape / data.pad_to_shape (round 6: batches padded to a lattice of shapes carry
    their own maxima): the host side, on CPU tensors"""
    from deepvoice3_pytorch_amd import data, ops, train_step
    assert data.lattice_shape(33, 17, 16, 8) == (48, 24)
    assert data.lattice_shape(32, 16, 16, 8) == (32, 16)
    v = ops.ValidLengths.make(29, 13, 3, 32, 16, 15, 7, 1, 4, torch.device("cpu"))
    assert v.tv.tolist() == [29, 13, 13, 52] and v.key_valid.tolist() == [29, 29, 29]
    assert abs(float(v.scale[0]) - 29 ** 0.5) < 1e-6
    # the converter's time axes: decoder steps x 1, 2, 4 (each ConvTranspose1d doubles)
    for T, mult in ((16, 1), (32, 2), (64, 4)):
        ptr, tail, m = v.axis_for(T)
        assert int(ptr[0]) == 13 and m == mult and tail == 7 * mult
    with pytest.raises(RuntimeError):
        v.axis_for(48)
    with pytest.raises(ValueError):       # the promised surplus bound must hold
        ops.ValidLengths.make(10, 13, 3, 32, 16, 15, 7, 1, 4, torch.device("cpu"))
    # r = 3: the mel axis is not a power of two times the decoder's
    v3 = ops.ValidLengths.make(29, 13, 3, 32, 16, 15, 7, 3, 2, torch.device("cpu"))
    assert v3.tv.tolist() == [29, 13, 39, 78]
    ptr, tail, m = v3.axis_for(48 * 2)
    assert int(ptr[0]) == 39 and m == 2 and tail == 7 * 3 * 2
    # pad_to_shape keeps the batch and appends zeros (done flags: ones)
    B, Tt, Td, r, ds = 2, 5, 6, 1, 4
    b0 = train_step.Batch(torch.ones(B, Tt, dtype=torch.long), torch.ones(B, Tt, dtype=torch.long),
                          torch.ones(B, Td, dtype=torch.long), torch.ones(B, Td * r, 3), torch.ones(B, Td * r * ds, 4),
                          torch.zeros(B, Td, 1), np.array([5, 3]), np.array([20, 12]), None, r, ds, torch.device("cpu"))
    b1 = data.pad_to_shape(b0, 8, 8)
    assert b1.text.shape == (B, 8) and b1.y.shape == (B, 32, 4) and b1.mel.shape == (B, 8, 3)
    assert int(b1.text[:, 5:].abs().sum()) == 0 and float(b1.y[:, 24:].abs().sum()) == 0
    assert float(b1.done[:, 6:].min()) == 1.0 and float(b1.done[:, :6].max()) == 0.0
    assert b1.valid.tv.tolist() == [5, 6, 6, 24] and (b1.valid.tail_in, b1.valid.tail_dec) == (3, 2)
    assert train_step.LatticeReplay.key_of(b1) == (8, 8, B)
    with pytest.raises(RuntimeError):
        train_step.LatticeReplay.key_of(b0)
    c = train_step.clone_batch(b1)
    assert c.valid.tv is not b1.valid.tv and c.valid.tv.tolist() == b1.valid.tv.tolist() and c.text.data_ptr() != b1.text.data_ptr()


def test_torch_library_operator_surface_is_registered():
    """deepvoice3_pytorch_amd/torch_ops.py: every operator of SURVEY.md 8b's list is a dispatcher entry with a schema,
    shape inference on meta tensors, and NO CPU implementation (the product path has no fallback)"""
    from deepvoice3_pytorch_amd import torch_ops
    for name in torch_ops.OPERATORS:
        op = getattr(torch.ops.dv3hip, name)
        assert op.default._schema.name == "dv3hip::" + name
    x, v = torch.empty(2, 8, 33, device="meta"), torch.empty(16, 8, 3, device="meta")
    y, pre, bits = torch.ops.dv3hip.conv1d_glu_fwd(x, v, None, None, None, 1, False, 0, True, 0.05, 1, 1)
    assert y.shape == (2, 8, 33) and pre.shape == (2, 16, 33) and bits.shape == (2 * 8 * 2,)
    assert torch.ops.dv3hip.conv1d_act_fwd(x, torch.empty(5, 8, 3, device="meta"), None, None, 0, 2, 1).shape == (2, 5, 29)
    assert torch.ops.dv3hip.convtranspose1d_k2s2_fwd(x, torch.empty(8, 6, 2, device="meta"), None, None).shape == (2, 6, 66)
    c, P, Pd, _ = torch.ops.dv3hip.attention_fwd(x, torch.empty(2, 8, 11, device="meta"), torch.empty(2, 11, 8, device="meta"),
                                                 None, 0.0, 0, 0)
    assert c.shape == (2, 8, 33) and P.shape == Pd.shape == (2, 33, 11)
    with pytest.raises(NotImplementedError):
        torch.ops.dv3hip.conv1d_glu_fwd(torch.zeros(2, 8, 33), torch.zeros(16, 8, 3), None, None, None, 1, False, 0, True,
                                        0.0, 1, 1)
    with pytest.raises(NotImplementedError):
        torch.ops.dv3hip.fused_clip_adam(torch.zeros(4), torch.zeros(4), torch.zeros(4), torch.zeros(4), 1e-3, 1, 0.5, 0.9,
                                         1e-6, 0.0, 0.1)
