# coding: utf-8
"""Experiment (round 4): does a replayed step keep the two-stream overlap of the eager step when the weight-gradient
branch (ops.SideStream) is captured as its OWN hipGraph and launched on a real second stream?

Round 3 found the whole-step hipGraph 4-7 % slower than eager launches whenever the GPU is the bound
(profiles/r03_side_stream_ab.txt: a captured step runs the same with and without the side stream, i.e. the graph executor
serialises the two branches).  Here the step is cut into three graphs:
    G1  main stream: zero_grad + forward + losses + the input-gradient chain of backward; every fork point RECORDS an
        external event (hipEventRecordWithFlags(.., hipEventRecordExternal)) instead of joining the side stream's capture
    GS  side stream (captured separately, relaxed mode): for every layer WAIT on that external event, wgrad GEMM,
        weight-norm backward; launched on the real side stream right after G1
    G2  main stream, after a normal event wait on the side stream: clip + Adam
Launch order on the host (G1, GS, wait, G2) makes every wait see the record of the SAME step.
Prints ms/step of eager, the single-graph replay and the three-graph replay, and the loss after the same number of steps."""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from deepvoice3_pytorch_amd import ops, train_step  # noqa: E402

dev = torch.device("cuda:0")
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
vp = ctypes.c_void_p
hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(vp), ctypes.c_uint]
hip.hipEventRecordWithFlags.argtypes = [vp, vp, ctypes.c_uint]
hip.hipStreamWaitEvent.argtypes = [vp, vp, ctypes.c_uint]
hip.hipStreamBeginCapture.argtypes = [vp, ctypes.c_int]
hip.hipStreamEndCapture.argtypes = [vp, ctypes.POINTER(vp)]
hip.hipGraphInstantiate.argtypes = [ctypes.POINTER(vp), vp, ctypes.POINTER(vp), ctypes.c_char_p, ctypes.c_size_t]
hip.hipGraphLaunch.argtypes = [vp, vp]
hip.hipGetErrorString.restype = ctypes.c_char_p


hip.hipStreamGetCaptureInfo_v2.argtypes = [vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(vp),
                                           ctypes.POINTER(ctypes.POINTER(vp)), ctypes.POINTER(ctypes.c_size_t)]
hip.hipGraphAddEventRecordNode.argtypes = [ctypes.POINTER(vp), vp, ctypes.POINTER(vp), ctypes.c_size_t, vp]
hip.hipGraphAddEventWaitNode.argtypes = [ctypes.POINTER(vp), vp, ctypes.POINTER(vp), ctypes.c_size_t, vp]
hip.hipStreamUpdateCaptureDependencies.argtypes = [vp, ctypes.POINTER(vp), ctypes.c_size_t, ctypes.c_uint]


def add_event_node(stream_raw, event, record):
    """an event record / wait node at the current frontier of the capture on `stream_raw` (the runtime bundled with this
    torch rejects hipEventRecordWithFlags(.., hipEventRecordExternal) during capture: hipErrorInvalidValue)"""
    status, cid, graph = ctypes.c_int(), ctypes.c_ulonglong(), vp()
    deps, nd = ctypes.POINTER(vp)(), ctypes.c_size_t()
    ck(hip.hipStreamGetCaptureInfo_v2(stream_raw, ctypes.byref(status), ctypes.byref(cid), ctypes.byref(graph),
                                      ctypes.byref(deps), ctypes.byref(nd)), "hipStreamGetCaptureInfo_v2")
    if status.value != 1:
        raise RuntimeError("stream is not capturing (status %d)" % status.value)
    node = vp()
    fn = hip.hipGraphAddEventRecordNode if record else hip.hipGraphAddEventWaitNode
    ck(fn(ctypes.byref(node), graph, deps, nd.value, event), "hipGraphAddEvent%sNode" % ("Record" if record else "Wait"))
    arr = (vp * 1)(node)
    ck(hip.hipStreamUpdateCaptureDependencies(stream_raw, arr, 1, 1), "hipStreamUpdateCaptureDependencies")


def ck(rc, what):
    if rc != 0:
        raise RuntimeError("%s: hip error %d (%s)" % (what, rc, hip.hipGetErrorString(rc).decode()))


def timed(fn, n=12, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, t_host / n * 1e3, out


def probe(preset, gemm, batch):
    r = bench.TrainRun(dev, None, 0, 1, preset, gemm, batch, 150, 800, graph=False)
    t = r.trainer
    ms_e, host_e, scal = timed(lambda: r.step())
    print("%s %s B=%d eager             %.3f ms/step (host loop %.2f)  loss %.4f" % (preset, gemm, batch, ms_e, host_e, float(scal["loss"])), flush=True)
    g = train_step.GraphedTrainer(t, r.batch, warmup=1)
    ms_g, host_g, scal = timed(lambda: g.step())
    print("%s %s B=%d one hipGraph      %.3f ms/step (host loop %.2f)  loss %.4f" % (preset, gemm, batch, ms_g, host_g, float(scal["loss"])), flush=True)
    g.close()
    torch.cuda.synchronize()

    # ---- three graphs ----
    events = []
    for _ in range(160):
        e = vp()
        ck(hip.hipEventCreateWithFlags(ctypes.byref(e), 2), "hipEventCreateWithFlags")      # hipEventDisableTiming
        events.append(e)
    state = dict(n=0, side_graph=None)
    side = t.side_stream
    side_raw = side.cuda_stream
    SS = ops.SideStream
    orig_fork, orig_join = SS.__dict__["fork"], SS.__dict__["join"]

    def ext_fork(cls, *tensors):
        e = events[state["n"]]
        state["n"] += 1
        add_event_node(cls.main.cuda_stream, e, True)
        add_event_node(side_raw, e, False)
        cls.keep.append([None, [tensors]])
        return cls._section

    def ext_join(cls):
        if cls.stream is not None and state["side_graph"] is None:
            gph = vp()
            ck(hip.hipStreamEndCapture(side_raw, ctypes.byref(gph)), "hipStreamEndCapture(side)")
            state["side_graph"] = gph
        cls.keep_alive = cls.keep      # the side graph reads these tensors on every replay: keep them for good
        cls.keep = []
        cls._n = 0

    seed = torch.zeros(1, dtype=torch.int64, device=dev)
    prev_off = ops.dropout_state.dev_offset
    ops.dropout_state.dev_offset = seed
    site0 = ops.dropout_state.site
    SS.fork, SS.join = classmethod(ext_fork), classmethod(ext_join)
    g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g1):
            ck(hip.hipStreamBeginCapture(side_raw, 2), "hipStreamBeginCapture(side, relaxed)")
            t._zero_grad()
            scal = t.forward_backward(r.batch)
        with torch.cuda.graph(g2, pool=g1.pool()):
            t.optimizer_step()
            scal["grad_norm"] = t._scalar(t.norm_out[0:1], 1.0)
            seed.add_(1)
    finally:
        SS.fork, SS.join = orig_fork, orig_join
        ops.dropout_state.site = site0
    side_exec = vp()
    ck(hip.hipGraphInstantiate(ctypes.byref(side_exec), state["side_graph"], None, None, 0), "hipGraphInstantiate(side)")
    print("    captured: %d fork points, side graph instantiated" % state["n"], flush=True)
    evj = torch.cuda.Event()

    def step3():
        t._set_hyper()
        g1.replay()
        ck(hip.hipGraphLaunch(side_exec, side_raw), "hipGraphLaunch(side)")
        evj.record(side)
        torch.cuda.current_stream().wait_event(evj)
        g2.replay()
        ops.bump_param_epoch()
        return scal
    ms_3, host_3, scal = timed(step3)
    print("%s %s B=%d three hipGraphs   %.3f ms/step (host loop %.2f)  loss %.4f  grad_norm %.4f" % (
        preset, gemm, batch, ms_3, host_3, float(scal["loss"]), float(scal["grad_norm"])), flush=True)
    ops.dropout_state.dev_offset = prev_off
    r.close()


cfgs = [("deepvoice3_ljspeech", "f16x3", 64), ("deepvoice3_ljspeech", "f16x3", 16), ("deepvoice3_vctk", "bf16", 64)]
if len(sys.argv) > 1:
    cfgs = [cfgs[int(a)] for a in sys.argv[1:]]
for (preset, gemm, batch) in cfgs:
    try:
        probe(preset, gemm, batch)
    except Exception as e:
        import traceback
        traceback.print_exc()
        print("FAILED %s %s B=%d: %s: %s" % (preset, gemm, batch, type(e).__name__, e), flush=True)
        torch.cuda.synchronize()
