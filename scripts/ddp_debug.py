import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_ddp as T


def run(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from deepvoice3_pytorch_amd import builder, train_step
    dev = torch.device("cuda:0")
    pg = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        pg = dist.group.WORLD
    torch.manual_seed(0)
    model = builder.deepvoice3(**T.HP).to(dev)
    tr = train_step.Trainer(model, train_step.TrainConfig(max_positions=128), process_group=pg, bucket_mb=0.05)
    per = 4 // world
    bt = T._make_batch(rank * per, (rank + 1) * per)
    batch = train_step.Batch.from_collate(bt["text"], bt["input_lengths"], bt["mel"], bt["y"], bt["text_positions"],
                                          bt["frame_positions"], bt["done"], bt["target_lengths"], None,
                                          downsample_step=4, device=dev)
    tr._set_hyper()
    tr.arena.grad.zero_()
    log = []
    if tr.comm is not None:
        import collections
        from deepvoice3_pytorch_amd import ops
        comm = tr.comm
        orig_launch = comm._launch
        cnt_in, cnt_ag = collections.Counter(), collections.Counter()
        orig_inplace = comm._on_inplace_grad
        def inplace(param):
            cnt_in[comm._index_of.get(id(param))] += 1
            orig_inplace(param)
        ops.grad_ready_hooks[ops.grad_ready_hooks.index(orig_inplace)] = inplace
        orig_make = comm._make_hook
        def launch(b):
            log.append(("launch", b, list(comm.buckets[b][2]), [cnt_in[i] for i in comm.buckets[b][2]]))
            orig_launch(b)
        comm._launch = launch
    scal = tr.forward_backward(batch)
    if tr.comm is not None and rank == 0:
        names_all = [n for n, _ in model.named_parameters()]
        idx_name = {}
        for n, p_ in model.named_parameters():
            if id(p_) in comm._index_of:
                idx_name[comm._index_of[id(p_)]] = n
        print("buckets:", [(b, len(pl)) for b, (_, _, pl) in enumerate(comm.buckets)][:40])
        print("in-place notifications != 1:", [(idx_name.get(i), c) for i, c in cnt_in.items() if c != 1][:20])
        print("n in-place notified params", len(cnt_in), "of", len(comm._index_of))
        print("launch order (bucket, params, in-place counts at launch):", [(b, pl, c) for _, b, pl, c in log if 0 in c][:60])
    if tr.comm is not None:
        tr.comm.finish()
    torch.cuda.synchronize()
    names = [n for n, p in model.named_parameters() if any(p is q_ for q_ in tr.arena.params)]
    q.put((rank, (tr.arena.grad / world).cpu().numpy(), float(scal["loss"]), tr.arena.offsets, tr.arena.sizes,
           [n for n, _ in model.named_parameters()]))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    res = {}
    for world in (2, 1):
        q = ctx.Queue()
        ps = [ctx.Process(target=run, args=(r, world, 29700 + world, q)) for r in range(world)]
        [p.start() for p in ps]
        out = sorted([q.get(timeout=200) for _ in range(world)], key=lambda t: t[0])
        [p.join(30) for p in ps]
        res[world] = out
    g2, g1 = res[2][0][1], res[1][0][1]
    print("loss 2-rank r0 %.6f r1 %.6f | single %.6f" % (res[2][0][2], res[2][1][2], res[1][0][2]))
    print("norm 2-rank %.6f single %.6f" % (np.linalg.norm(g2), np.linalg.norm(g1)))
    offs, sizes, names = res[1][0][3], res[1][0][4], res[1][0][5]
    bad = []
    for o, n, nm in zip(offs, sizes, names):
        a, b = g2[o:o + n], g1[o:o + n]
        d = np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)
        if d > 1e-3:
            bad.append((nm, float(d), float(np.linalg.norm(a)), float(np.linalg.norm(b))))
    print("params off by > 1e-3: %d of %d" % (len(bad), len(offs)))
    for b in bad[:40]:
        print("  %-60s rel %.3f  |g2| %.3e |g1| %.3e" % b)
