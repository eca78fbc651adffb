# coding: utf-8
"""Round 5: the launch census of a REAL training step.  The library records every tap-GEMM descriptor of one eager step
(dv3_debug_set(40, 1)); each is then re-issued stand-alone from a hipGraph of 20 launches (scripts/r5_common.graph_time:
no host in the timing, nothing else on the GPU) on the step's own buffers.  Per launch: shape, the kernel variant that
served it, its stand-alone time, its matrix work at the dense rate, and the bytes it must move -- so the launches that
are far from both are named with their share of the step.

    python scripts/r5_conv_census.py [preset] [gemm] [batch]        (default deepvoice3_ljspeech f16x3 64)
"""
import collections
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch  # noqa: E402
import bench  # noqa: E402
from deepvoice3_pytorch_amd import ops, _lib  # noqa: E402
from r5_common import graph_time  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
preset = sys.argv[1] if len(sys.argv) > 1 else "deepvoice3_ljspeech"
gemm = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64

run = bench.TrainRun(dev, None, 0, 1, preset, gemm, batch, 150, 800, graph=False)
for _ in range(3):
    run.trainer.step(run.batch)
torch.cuda.synchronize()
# the recorded descriptors point into blocks the eager step hands back to torch's caching allocator: they must stay
# mapped (torch.cuda.graph() empties the cache on entry)
torch.cuda.empty_cache = lambda: None
L.dv3_debug_set(40, 1)
run.trainer.step(run.batch)
torch.cuda.synchronize()
L.dv3_debug_set(40, 0)
n = L.dv3_debug_get(40)
Desc = ops._conv_desc
sz = ctypes.sizeof(Desc)
raw = (ctypes.c_char * (n * sz))()
assert L.dv3_debug_read(40, raw, n * sz) == 0
var = (ctypes.c_int * n)()
assert L.dv3_debug_read(41, var, n * 4) == 0
descs = [Desc.from_buffer_copy(bytes(raw[i * sz:(i + 1) * sz])) for i in range(n)]
print("%s %s B=%d: %d tap-GEMM launches in one step" % (preset, gemm, batch, n), flush=True)

MODE = {ops.EPI_LINEAR: "linear", ops.EPI_RELU: "relu", ops.EPI_SIGMOID: "sigmoid", ops.EPI_SOFTSIGN: "softsign",
        ops.EPI_GLU: "glu", ops.EPI_HIGHWAY: "highway", ops.EPI_DGRAD: "dgrad"}


def key(d, v):
    return (MODE.get(d.mode, str(d.mode)), d.B, d.Cin, d.M, d.Tout, d.J, d.dil, 1 if (d.xmask or d.xmask_c8) else 0,
            d.split_terms, d.io_bf16, 1 if d.ab else 0, d.store_mode, v)


groups = collections.OrderedDict()
for i, d in enumerate(descs):
    groups.setdefault(key(d, var[i]), []).append(i)

st = torch.cuda.current_stream()
rows = []
for k, idx in groups.items():
    d = descs[idx[0]]

    def fn(d=d):
        rc = L.dv3_conv_gemm_f32(ctypes.byref(d), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
    us = graph_time(fn, per_graph=20, replays=4)
    terms = 1 if d.split_terms == 1 else 3
    flops = 2.0 * d.M * d.Cin * d.J * d.B * d.Tout * terms
    mfma_us = flops / 2.5e15 * 1e6
    esz = 2 if d.io_bf16 else 4
    gated = k[0] in ("glu", "highway")
    byt = (d.Cin + (d.M // 2 if gated else d.M)) * d.B * d.Tout * esz
    if gated:
        byt += d.Cin * d.B * d.Tout * esz + (d.M * d.B * d.Tout * esz if d.ab else 0)
    hbm_us = byt / 5.0e12 * 1e6
    rows.append((k, len(idx), us, mfma_us, hbm_us))

tot = sum(c * us for (_, c, us, _, _) in rows)
print("sum of the stand-alone times: %.2f ms per step" % (tot / 1e3))
print("%-8s %3s %4s %4s %5s %1s %3s %1s %5s %2s %2s %6s | %3s %8s %8s %7s %7s %6s" % (
    "mode", "B", "Cin", "M", "T", "J", "dil", "m", "split", "io", "ab", "var", "n", "us each", "us total", "mfma us", "hbm us", "x floor"))
for (k, c, us, mf, hb) in sorted(rows, key=lambda r: -r[1] * r[2]):
    floor = max(mf, hb)
    print("%-8s %3d %4d %4d %5d %1d %3d %1d %5d %2d %2d %6d | %3d %8.1f %8.1f %7.1f %7.1f %6.1f" % (
        k[0], k[1], k[2], k[3], k[4], k[5], k[6], k[7], k[8], k[9], k[10], k[12], c, us, c * us, mf, hb, us / max(floor, 1e-3)))
# tile sweep (argv[4] == "sweep"): every split-kernel launch re-issued with each forced tile (tile_hint 21..29: the
# 128 x 128, 128 x 64, 256 x 32, 64 x 32... tiles of conv_gemm_bf16x3.hip's table; 30 = the 256 x 256 k16 ping-pong kernel)
if len(sys.argv) > 4 and sys.argv[4] == "sweep":
    print("---- forced tiles: us per launch (the picker's choice first) ----")
    gain = 0.0
    for (k, c, us, mf, hb) in sorted(rows, key=lambda r: -r[1] * r[2]):
        if k[12] // 1000 not in (3, 5):
            continue
        d = descs[groups[k][0]]
        res = {}
        for hint in (21, 22, 23, 24, 25, 26, 28, 29, 30):
            d.tile_hint = hint

            def fn(d=d):
                return L.dv3_conv_gemm_f32(ctypes.byref(d), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            if fn() != 0:
                continue
            res[hint] = graph_time(fn, per_graph=20, replays=3)
        d.tile_hint = 0
        best = min(res, key=res.get)
        gain += c * max(0.0, us - res[best])
        print("%-7s Cin %4d M %4d T %4d J %d dil %2d m %d var %d x%d: picked %.1f | %s | best %d (%.2f)" % (
            k[0], k[2], k[3], k[4], k[5], k[6], k[7], k[12], c, us, " ".join("%d:%.1f" % (h, t) for h, t in sorted(res.items())),
            best, res[best] / us))
    print("sum over the step of (picked - best forced): %.2f ms" % (gain / 1e3))
# deep-prefetch form of the 128 x 64 tile (argv[4] == "dp"; dv3_debug_set(43, 0 | 2)): every split-kernel launch as picked,
# on the 128 x 64 tile in-phase (hint 22), and on the 128 x 64 tile with the register rings
if len(sys.argv) > 4 and sys.argv[4] == "dp":
    print("---- deep prefetch: us per launch ----")
    tot0 = tot1 = 0.0
    for (k, c, us, mf, hb) in sorted(rows, key=lambda r: -r[1] * r[2]):
        if k[12] // 1000 not in (3, 5) or k[5] not in (1, 3):
            continue
        d = descs[groups[k][0]]

        def fn(d=d):
            return L.dv3_conv_gemm_f32(ctypes.byref(d), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        res = {}
        for (hint, dp) in ((0, 0), (22, 0), (22, 2)):
            d.tile_hint = hint
            L.dv3_debug_set(43, dp)
            if fn() != 0:
                continue
            res[(hint, dp)] = graph_time(fn, per_graph=20, replays=3)
        d.tile_hint = 0
        L.dv3_debug_set(43, 0)
        best = min(res.values())
        tot0 += c * res[(0, 0)]
        tot1 += c * best
        print("%-7s Cin %4d M %4d T %4d J %d dil %2d m %d var %d x%d: picked %.1f | 128x64 %.1f | 128x64 rings %.1f (%.2f of picked)" % (
            k[0], k[2], k[3], k[4], k[5], k[6], k[7], k[12], c, res[(0, 0)], res.get((22, 0), -1), res.get((22, 2), -1),
            res.get((22, 2), 0) / res[(0, 0)]))
    print("sum over the step: picked %.2f ms, best of the three %.2f ms" % (tot0 / 1e3, tot1 / 1e3))
# c8 path (argv[4] == "c8"): every single-term c8 launch as picked, forced onto the 8-wave 256 x 256 form (hint 40) and
# onto two 4-wave workgroups per CU on 256 x 128 tiles (hint 41)
if len(sys.argv) > 4 and sys.argv[4] == "c8":
    print("---- conv_c8pp forms: us per launch ----")
    tot0 = tot1 = 0.0
    for (k, c, us, mf, hb) in sorted(rows, key=lambda r: -r[1] * r[2]):
        if k[12] // 1000 not in (8, 9):
            continue
        d = descs[groups[k][0]]

        def fn(d=d):
            return L.dv3_conv_gemm_f32(ctypes.byref(d), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        res = {}
        for hint, nw4 in ((0, 2), (40, 0), (41, 1)):      # (34: 0 = 8-wave only, 1 = 4-wave always, 2 = by the rule)
            d.tile_hint = 40 if hint else 0
            L.dv3_debug_set(34, nw4)
            if fn() != 0:
                continue
            res[hint] = graph_time(fn, per_graph=20, replays=3)
        d.tile_hint = 0
        L.dv3_debug_set(34, 2)
        best = min(res.values())
        tot0 += c * res[0]
        tot1 += c * best
        print("%-7s Cin %4d M %4d T %4d J %d dil %2d m %d var %d x%d: picked %.1f | 8-wave %.1f | 4-wave x2 %.1f | best/picked %.2f" % (
            k[0], k[2], k[3], k[4], k[5], k[6], k[7], k[12], c, res[0], res.get(40, -1), res.get(41, -1), best / res[0]))
    print("sum over the step: picked %.2f ms, best of the three %.2f ms" % (tot0 / 1e3, tot1 / 1e3))
# k-split form of the 128 x 64 tile (argv[4] == "ks"; dv3_debug_set(44, 0 | 2)): every split-kernel launch as picked, and on
# the 128 x 64 tile with two wave groups per workgroup on halves of the chunk range
if len(sys.argv) > 4 and sys.argv[4] == "ks":
    print("---- k-split: us per launch ----")
    tot0 = tot1 = 0.0
    for (k, c, us, mf, hb) in sorted(rows, key=lambda r: -r[1] * r[2]):
        if k[12] // 1000 not in (3, 5):
            continue
        d = descs[groups[k][0]]

        def fn(d=d):
            return L.dv3_conv_gemm_f32(ctypes.byref(d), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        res = {}
        for (hint, ks) in ((0, 0), (22, 0), (22, 2)):
            d.tile_hint = hint
            L.dv3_debug_set(44, ks)
            if fn() != 0:
                continue
            res[(hint, ks)] = graph_time(fn, per_graph=20, replays=3)
        d.tile_hint = 0
        L.dv3_debug_set(44, 0)
        best = min(res.values())
        tot0 += c * res[(0, 0)]
        tot1 += c * best
        nb2 = -(-k[3] // 128) * -(-(k[1] * k[4]) // 64) if k[0] not in ("glu", "highway") else -(-(k[3] // 2) // 64) * -(-(k[1] * k[4]) // 64)
        print("%-7s Cin %4d M %4d T %4d J %d dil %2d m %d var %d x%d: picked %.1f | 128x64 %.1f | 128x64 k-split %.1f (%.2f of picked; %d tiles, %d k-steps)" % (
            k[0], k[2], k[3], k[4], k[5], k[6], k[7], k[12], c, res[(0, 0)], res.get((22, 0), -1), res.get((22, 2), -1),
            res.get((22, 2), 0) / res[(0, 0)], nb2, -(-k[2] // 32) * k[5]))
    print("sum over the step: picked %.2f ms, best of the three %.2f ms" % (tot0 / 1e3, tot1 / 1e3))
# what a floor-bound small launch would give back: every launch under 60 us brought to 2 x its floor (or 6 us)
small = [(c, us, max(2 * max(mf, hb), 6.0)) for (_, c, us, mf, hb) in rows if us < 60.0]
print("launches under 60 us: %d, %.2f ms of the step; at 2 x their floor (>= 6 us): %.2f ms" % (
    sum(c for c, _, _ in small), sum(c * us for c, us, _ in small) / 1e3, sum(c * min(us, t) for c, us, t in small) / 1e3))
run.close()
