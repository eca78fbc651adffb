# coding: utf-8
"""Host-side operators over the libdv3hip C ABI (include/dv3hip.h).

Every function here enqueues hand-written HIP kernels on torch's *current* stream through
ctypes; torch is used only for device memory (caching allocator), streams and autograd
bookkeeping.  There is no fallback: a non-CUDA tensor or a missing library raises.

Layout convention inside the package: activations are BCT (batch, channel, time), fp32,
contiguous -- the layout of the reference conv stacks (deepvoice3_pytorch/modules.py:139-164).
"""
import contextlib
import ctypes
import functools
import math

import torch

from . import _lib
from ._lib import CONSTS, STRUCTS

EPI_LINEAR = CONSTS["DV3_EPI_LINEAR"]
EPI_RELU = CONSTS["DV3_EPI_RELU"]
EPI_SIGMOID = CONSTS["DV3_EPI_SIGMOID"]
EPI_GLU = CONSTS["DV3_EPI_GLU"]
EPI_HIGHWAY = CONSTS["DV3_EPI_HIGHWAY"]
EPI_DGRAD = CONSTS["DV3_EPI_DGRAD"]
EPI_SOFTSIGN = CONSTS["DV3_EPI_SOFTSIGN"]
STORE_BCT = CONSTS["DV3_STORE_BCT"]
STORE_INTERLEAVE2 = CONSTS["DV3_STORE_INTERLEAVE2"]

_conv_desc = STRUCTS["dv3_conv_desc"]
_wgrad_desc = STRUCTS["dv3_wgrad_desc"]
_planes_desc = STRUCTS["dv3_planes_desc"]
_wn_desc = STRUCTS["dv3_wn_desc"]
_wn_bwd_desc = STRUCTS["dv3_wn_bwd_desc"]
_gate_bwd_desc = STRUCTS["dv3_gate_bwd_desc"]
_softmax_desc = STRUCTS["dv3_softmax_desc"]
_softmax_bwd_desc = STRUCTS["dv3_softmax_bwd_desc"]
_attn_fwd_desc = STRUCTS["dv3_attn_fwd_desc"]
_spec_loss_desc = STRUCTS["dv3_spec_loss_desc"]


# ----------------------------------------------------------------------------------------------
# GEMM arithmetic (include/dv3hip.h, "Split-bf16" / "f16x3"), DV3_GEMM=<mode>:
#   "f16x3"  (default) forward GEMMs on scaled fp16 hi/lo operands (three fp16 MFMAs per product,
#            2^-22-class operands: fp32-class outputs -- what the 1e-4 parity bar needs at the preset
#            model sizes), gradient GEMMs on bf16 hi/lo operands (three bf16 MFMAs; bf16 has fp32's
#            exponent range, so gradients need no scale search); 16/3 x the fp32-MFMA rate;
#   "bf16x3" every GEMM on bf16 hi/lo operands (~5e-6 relative error per GEMM; passes 1e-4 on small
#            models, ~1-3e-4 on the presets: kept for A/B measurements);
#   "f32"    the exact fp32 MFMA chain (v_mfma_f32_32x32x2_f32);
#   "bf16"   operands rounded to bf16 at the matrix-core inputs (hi planes only, one MFMA per
#            product, fp32 accumulate, fp32 master weights) -- BASELINE.json's bf16 configs.
# ----------------------------------------------------------------------------------------------
import os as _os

_GEMM_MODES = ("f16x3", "bf16x3", "f32", "bf16")
_gemm_mode = _os.environ.get("DV3_GEMM", "f16x3")
if _gemm_mode not in _GEMM_MODES:
    raise RuntimeError("DV3_GEMM must be one of %s" % (_GEMM_MODES,))


def set_gemm_precision(mode):
    """one of _GEMM_MODES; returns the previous mode."""
    global _gemm_mode
    if mode not in _GEMM_MODES:
        raise ValueError("gemm precision must be one of %s" % (_GEMM_MODES,))
    prev, _gemm_mode = _gemm_mode, mode
    return prev


def gemm_precision():
    return _gemm_mode


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)
import threading as _threading


class _StreamOverride(_threading.local):
    """raw handle of ops.SideStream's stream while a side-stream section runs ON THIS THREAD (backward runs on
    autograd's thread; another thread launching ops meanwhile keeps its own current stream)"""
    handle = None


_stream_override = _StreamOverride()


def _stream():
    """the current HIP stream of the calling thread's current device as a raw handle.  torch.cuda.current_stream()
    costs ~8 us of Python per call (measured: 0.8 ms per train-step forward); the two C accessors cost ~0.3 us.  The
    device is read on every call (torch.cuda.set_device after the first op, or a second model on another device, must
    not be served the first device's stream)."""
    h = _stream_override.handle
    if h is not None:
        return h
    if _raw_stream is None or _cur_device is None:
        return torch.cuda.current_stream().cuda_stream
    return _raw_stream(_cur_device())


reserved_stream_handles = set()      # raw handles new_stream() must not hand out again (capture streams)
stream_probe_log = []      # one record per concurrent_stream() call: bench.py prints them as `stream_queues`
_hip_rt = None


def new_stream(priority="normal"):
    """a HIP stream as a torch stream object.  priority "low": created with hipStreamCreateWithPriority at the device's
    LEAST priority (torch itself only offers normal and high) and wrapped as an ExternalStream -- for work that should
    yield compute units to the step stream whenever both have workgroups to place (the weight-gradient branch of
    backward: profiles/r05_step_timeline.txt -- the input-gradient chain on the step stream is the critical path, the
    weight-gradient queue is 68-84 % busy).  Never destroyed (a handful per process)."""
    if priority != "low":
        # torch hands its 32 pool streams out round robin: after enough calls (every probe of concurrent_stream takes up
        # to a dozen) the pool comes back to a handle this process has set aside -- the capture stream of
        # train_step._capture_stream: a trainer whose second backward stream IS the capture stream cannot be captured
        # ("operation cannot be performed in the present state")
        for _ in range(40):
            s = torch.cuda.Stream()
            if s.cuda_stream not in reserved_stream_handles:
                return s
        return s
    import ctypes
    global _hip_rt
    if _hip_rt is None:
        _hip_rt = ctypes.CDLL("libamdhip64.so")
    least, greatest = ctypes.c_int(0), ctypes.c_int(0)
    if _hip_rt.hipDeviceGetStreamPriorityRange(ctypes.byref(least), ctypes.byref(greatest)) != 0 or least.value == greatest.value:
        return torch.cuda.Stream()
    h = ctypes.c_void_p()
    if _hip_rt.hipStreamCreateWithPriority(ctypes.byref(h), ctypes.c_uint(1), ctypes.c_int(least.value)) != 0 or not h.value:   # 1 = non-blocking
        return torch.cuda.Stream()
    return torch.cuda.ExternalStream(h.value)


def concurrent_stream(avoid, tries=12, role="side", strict=None, priority="normal"):
    """A torch stream whose work REALLY runs beside the work of every stream in `avoid`.  HIP multiplexes its streams onto a
    few hardware queues (GPU_MAX_HW_QUEUES, default 4): two streams that land on the same queue execute in issue order,
    whatever the events between them say -- the step's second backward stream then adds nothing (measured: the same
    replayed deepvoice3_vctk step at 11.76 or 12.75 ms, by which pool stream a Trainer instance happened to get,
    scripts/r4_probe_order.py).  Probed with pairs of one-thread spin kernels (torch.cuda._sleep): a candidate is taken
    when, against EVERY stream in `avoid`, the best of three pairs takes about as long as one kernel (ratio < 1.5; a
    shared queue gives 2.0).  Every call leaves a record in `stream_probe_log` (role, the ratio of each candidate against
    each avoided stream, whether a concurrent stream was found).  When none of `tries` candidates runs beside all of
    them: a loud warning and the candidate with the smallest worst ratio -- or RuntimeError under strict
    (DV3_STRICT_STREAMS=1).  While a capture is active (no synchronize allowed) or with DV3_STREAM_PROBE=0: a plain
    stream, recorded as unprobed."""
    import warnings
    rec = dict(role=role, avoid=len(avoid), probed=False, found=None, candidates=[], priority=priority)
    stream_probe_log.append(rec)
    if not avoid:
        return new_stream(priority)
    if strict is None:
        strict = _os.environ.get("DV3_STRICT_STREAMS", "") == "1"
    dev = avoid[0].device
    with torch.cuda.device(dev):
        if _os.environ.get("DV3_STREAM_PROBE", "1") in ("0", "") or torch.cuda.is_current_stream_capturing():
            return new_stream(priority)
        cycles = 200000

        def spin_pair(a, s):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record(a)
            with torch.cuda.stream(a):
                torch.cuda._sleep(cycles)
            if s is not None:
                with torch.cuda.stream(s):
                    torch.cuda._sleep(cycles)
                a.wait_stream(s)
            e1.record(a)
            torch.cuda.synchronize()
            return e0.elapsed_time(e1)

        try:
            spin_pair(avoid[0], None)
            one = min(spin_pair(avoid[0], None) for _ in range(3))
        except Exception:      # no spin kernel in this build: take any stream
            return new_stream(priority)
        rec["probed"] = True
        want_shared = _os.environ.get("DV3_SIDE_STREAM_SAME_QUEUE", "") == "1"      # experiment: the opposite choice
        best, best_worst = None, None
        for _ in range(tries):
            s = new_stream(priority)
            ratios = [round(min(spin_pair(a, s) for _ in range(3)) / one, 2) for a in avoid]
            rec["candidates"].append(ratios)
            worst = max(ratios)
            if best is None or worst < best_worst:
                best, best_worst = s, worst
            beside = worst < 1.5
            if beside != want_shared:
                rec["found"] = not want_shared
                return s
        rec["found"] = False
        msg = ("no HIP stream runs beside all %d streams of the step (%s stream; ratios %s): hardware queues are shared, the "
               "overlap this stream is for is lost (GPU_MAX_HW_QUEUES=%s)" % (len(avoid), role, rec["candidates"],
                                                                           _os.environ.get("GPU_MAX_HW_QUEUES", "default 4")))
        if strict:
            raise RuntimeError(msg)
        warnings.warn(msg)
    return best


# ----------------------------------------------------------------------------------------------
# range guard of the f16x3 mode (include/dv3hip.h, dv3_f16_range_events): the forward operands are
# v * 2^4 (activations) / v * 2^8 (weights) in fp16, so |x| > 4094 or |w| > 255.9 leaves the range.
# The kernels never saturate silently (values stay fp16-accurate to twice the range, then turn
# Inf / NaN like a diverged fp32 run) and count every 16-byte operand unit that left the range.
# ----------------------------------------------------------------------------------------------
def f16_range_events_tensor(device, reset=False):
    """-> int32 device tensor [1]: the sticky counter as of this point of the current stream (no host sync)"""
    out = torch.empty(1, dtype=torch.int32, device=device)
    _lib.call("dv3_f16_range_events", out.data_ptr(), int(bool(reset)), _stream())
    return out


def f16_range_events(reset=False, device=None):
    """-> int: operand units that left the fp16 range since the last reset (synchronises with the device)"""
    device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    return int(f16_range_events_tensor(device, reset).item())


def with_f16_range_fallback(fn, *args, **kwargs):
    """Run fn(*args, **kwargs); when the f16x3 kernels report operands outside the fp16 range, run it again in
    the bf16x3 mode (bf16 pairs have fp32's exponent range: ~5e-6 per GEMM instead of ~1e-6, never out of
    range) and return that result.  -> (result, fell_back).  Synchronises once."""
    if _gemm_mode != "f16x3":
        return fn(*args, **kwargs), False
    f16_range_events(reset=True)
    out = fn(*args, **kwargs)
    if f16_range_events(reset=True) == 0:
        return out, False
    prev = set_gemm_precision("bf16x3")
    try:
        return fn(*args, **kwargs), True
    finally:
        set_gemm_precision(prev)


def _ptr(t):
    return t.data_ptr() if t is not None else None


def _chk(t, name, dtype=torch.float32):
    if not t.is_cuda:
        raise RuntimeError("dv3hip op got a non-GPU tensor for %s: the HIP path has no CPU fallback" % name)
    if t.dtype != dtype:
        raise RuntimeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    return t


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _round_up(a, b):
    return (a + b - 1) // b * b


# ----------------------------------------------------------------------------------------------
# dropout keep-bits
# ----------------------------------------------------------------------------------------------
class DropoutState(object):
    """Seed / site bookkeeping for dv3_dropout_bits.  `dev_offset` (uint64 on the device) is
    added to the seed inside the kernel so a replayed hipGraph draws fresh masks each step."""

    def __init__(self):
        self.seed = None
        self.site = 0
        self.dev_offset = None   # optional torch.int64 tensor [1] on the device
        self.record = None       # tests: dict site_name -> (bits tensor, rows, T)

    def manual_seed(self, seed):
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.site = 0

    def next_site(self):
        if self.seed is None:
            self.manual_seed(torch.initial_seed())
        self.site += 1
        return self.site


dropout_state = DropoutState()


def dropout_bits(rows, T, p, device, name=None):
    """-> (bits int32 [rows][ceil(T/32)], row_stride_words). One Philox launch."""
    rs = (T + 31) // 32
    site = dropout_state.next_site()
    hit = mask_plan.take("bits", 1, rows, T, p) if (dropout_state.record is None or name is None) else None
    if hit is not None:
        return hit, rs
    bits = torch.empty(rows * rs, dtype=torch.int32, device=device)
    off = dropout_state.dev_offset
    _lib.call("dv3_dropout_bits", bits.data_ptr(), rows * rs, float(p), dropout_state.seed, site,
              _ptr(off), _stream())
    if dropout_state.record is not None and name is not None:
        dropout_state.record[name] = (bits, rows, T)
    return bits, rs


# ----------------------------------------------------------------------------------------------
# weight norm + packing
# ----------------------------------------------------------------------------------------------
class Packed(object):
    __slots__ = ("fwd", "bwd", "scale", "lda", "a_half", "ldb", "O", "I", "J", "transposed", "glu_cg",
                 "fwd_s", "bwd_s", "fwd_f16", "step_tiles")


SPLIT_BF16, SPLIT_F16 = CONSTS["DV3_SPLIT_DTYPE_BF16"], CONSTS["DV3_SPLIT_DTYPE_F16"]


def split_pack(packed, J, K, ld, dtype=SPLIT_BF16):
    """dv3_split_pack_bf16: fp32 packed image [J][K][ld] -> split image (int16 storage), bf16 hi/lo or
    scaled fp16 hi/lo."""
    kp = _round_up(K, 32)
    out = torch.empty(2 * J * kp * ld, dtype=torch.int16, device=packed.device)
    _lib.call("dv3_split_pack_bf16", packed.data_ptr(), out.data_ptr(), J, K, ld, dtype, _stream())
    out._dv3_f16 = dtype == SPLIT_F16       # conv_gemm reads the operand type off the image
    return out


def pack_weights(v, g, glu_cg=0, transposed=False, need_bwd=True, split_only=False):
    """dv3_weight_norm_pack_f32.  v: (O,I,J) / (O,I) [Conv1d / Linear] or (I,O,J) [ConvTranspose1d].
    split_only (split-bf16 GEMM modes, Conv1d / Linear): one fused launch writes the two split
    images and no fp32 images (pk.fwd / pk.bwd stay None)."""
    _chk(v, "weight_v")
    v = _c(v)
    if v.dim() == 2:
        v = v.unsqueeze(-1)
    pk = Packed()
    pk.fwd_f16 = _gemm_mode == "f16x3"
    fdt = SPLIT_F16 if pk.fwd_f16 else SPLIT_BF16
    if split_only and not transposed and _gemm_mode != "f32":
        O, I, J = v.shape
        pk.O, pk.I, pk.J, pk.transposed, pk.glu_cg = O, I, J, False, glu_cg
        if glu_cg:
            pk.a_half = _round_up(glu_cg, 4)
            pk.lda = 2 * pk.a_half
        else:
            pk.a_half, pk.lda = 0, _round_up(O, 4)
        pk.ldb = _round_up(I, 4)
        dev = v.device
        pk.fwd = pk.bwd = None
        pk.scale = torch.empty(O, dtype=torch.float32, device=dev)
        pk.fwd_s = torch.empty(2 * J * _round_up(I, 32) * pk.lda, dtype=torch.int16, device=dev)
        pk.bwd_s = torch.empty(2 * J * _round_up(O, 32) * pk.ldb, dtype=torch.int16, device=dev) if need_bwd else None
        d = _wn_desc()
        d.v, d.g, d.scale = v.data_ptr(), _ptr(_c(g) if g is not None else None), pk.scale.data_ptr()
        d.lda, d.a_half, d.ldb = pk.lda, pk.a_half, pk.ldb
        d.O, d.I, d.J, d.transposed, d.glu_cg = O, I, J, 0, glu_cg
        d.fwd_dtype = fdt
        pk.fwd_s._dv3_f16 = pk.fwd_f16
        _lib.call("dv3_weight_norm_split_pack_bf16", ctypes.byref(d), pk.fwd_s.data_ptr(), _ptr(pk.bwd_s), _stream())
        return pk
    if transposed:
        I, O, J = v.shape
    else:
        O, I, J = v.shape
    pk.O, pk.I, pk.J, pk.transposed, pk.glu_cg = O, I, J, transposed, glu_cg
    dev = v.device
    if transposed:
        pk.lda, pk.a_half = _round_up(J * O, 4), 0
        pk.fwd = torch.empty(I * pk.lda, dtype=torch.float32, device=dev)
        pk.scale = torch.empty(I, dtype=torch.float32, device=dev)
        pk.ldb = _round_up(I, 4)
        pk.bwd = torch.empty(J * O * pk.ldb, dtype=torch.float32, device=dev) if need_bwd else None
    else:
        if glu_cg:
            pk.a_half = _round_up(glu_cg, 4)
            pk.lda = 2 * pk.a_half
        else:
            pk.a_half, pk.lda = 0, _round_up(O, 4)
        pk.fwd = torch.empty(J * I * pk.lda, dtype=torch.float32, device=dev)
        pk.scale = torch.empty(O, dtype=torch.float32, device=dev)
        pk.ldb = _round_up(I, 4)
        pk.bwd = torch.empty(J * O * pk.ldb, dtype=torch.float32, device=dev) if need_bwd else None
    d = _wn_desc()
    d.v, d.g, d.scale = v.data_ptr(), _ptr(_c(g) if g is not None else None), pk.scale.data_ptr()
    d.fwd_pack, d.lda, d.a_half = pk.fwd.data_ptr(), pk.lda, pk.a_half
    d.bwd_pack, d.ldb = _ptr(pk.bwd), pk.ldb
    d.O, d.I, d.J, d.transposed, d.glu_cg = O, I, J, int(transposed), glu_cg
    _lib.call("dv3_weight_norm_pack_f32", ctypes.byref(d), _stream())
    pk.fwd_s = pk.bwd_s = None
    if _gemm_mode != "f32":
        # operand K/M extents of the two tap-GEMMs: fwd [J'][K=I][lda], bwd [J'][K'][ldb]
        if transposed:
            pk.fwd_s = split_pack(pk.fwd, 1, I, pk.lda, fdt)
            if need_bwd:
                pk.bwd_s = split_pack(pk.bwd, 1, J * O, pk.ldb)
        else:
            pk.fwd_s = split_pack(pk.fwd, J, I, pk.lda, fdt)
            if need_bwd:
                pk.bwd_s = split_pack(pk.bwd, J, O, pk.ldb)
    return pk


class Prepack(object):
    """Every weight-normed Conv1d / Linear layer of a model packed in TWO launches per step
    (dv3_weight_norm_split_pack_multi) instead of two per layer: the scales and both split images of all
    layers live in fixed buffers, the descriptor table in device memory is built once.  `layers`: [(v, g,
    glu_cg)] with v (O,I[,J]) -- parameters whose storage does not move (the trainer's flat arena).
    ConvLayerFn.forward picks the images up through `lookup` while `ops.prepacked` points here."""

    def __init__(self, layers):
        self.mode = _gemm_mode
        if self.mode == "f32":
            raise RuntimeError("Prepack serves the split-operand GEMM modes")
        f16 = self.mode == "f16x3"
        entry_t = STRUCTS["dv3_wn_multi_entry"]
        tab = (entry_t * len(layers))()
        first_row, first_block, rows, blocks, self.by_id, max_j = [], [], 0, 0, {}, 1
        dev = layers[0][0].device
        for n, (v, g, glu_cg) in enumerate(layers):
            v3 = v if v.dim() == 3 else v.unsqueeze(-1)
            if not v3.is_contiguous():
                raise RuntimeError("Prepack needs contiguous weights")
            O, I, J = v3.shape
            pk = Packed()
            pk.O, pk.I, pk.J, pk.transposed, pk.glu_cg, pk.fwd_f16 = O, I, J, False, glu_cg, f16
            if glu_cg:
                pk.a_half = _round_up(glu_cg, 4)
                pk.lda = 2 * pk.a_half
            else:
                pk.a_half, pk.lda = 0, _round_up(O, 4)
            pk.ldb = _round_up(I, 4)
            pk.fwd = pk.bwd = None
            pk.scale = torch.empty(O, dtype=torch.float32, device=dev)
            pk.fwd_s = torch.zeros(2 * J * _round_up(I, 32) * pk.lda, dtype=torch.int16, device=dev)
            pk.bwd_s = torch.zeros(2 * J * _round_up(O, 32) * pk.ldb, dtype=torch.int16, device=dev)
            pk.fwd_s._dv3_f16 = f16
            e = tab[n]
            e.d.v, e.d.g, e.d.scale = v.data_ptr(), _ptr(g), pk.scale.data_ptr()
            e.d.lda, e.d.a_half, e.d.ldb = pk.lda, pk.a_half, pk.ldb
            e.d.O, e.d.I, e.d.J, e.d.transposed, e.d.glu_cg = O, I, J, 0, glu_cg
            e.d.fwd_dtype = SPLIT_F16 if f16 else SPLIT_BF16
            e.fwd_split, e.bwd_split = pk.fwd_s.data_ptr(), pk.bwd_s.data_ptr()
            first_row.append(rows)
            first_block.append(blocks)
            rows += O
            blocks += ((O + 31) // 32) * ((I + 31) // 32)
            max_j = max(max_j, J)
            self.by_id[id(v)] = (pk, v.data_ptr(), glu_cg)
        raw = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8)
        self.table = raw.to(dev)
        self.first_row = torch.tensor(first_row, dtype=torch.int32, device=dev)
        self.first_block = torch.tensor(first_block, dtype=torch.int32, device=dev)
        self.n, self.rows, self.blocks, self.max_j = len(layers), rows, blocks, max_j
        self.keep = layers            # the parameters the table points at stay alive with it

    def run(self):
        _lib.call("dv3_weight_norm_split_pack_multi", self.table.data_ptr(), self.first_row.data_ptr(),
                  self.first_block.data_ptr(), self.n, self.rows, self.blocks, self.max_j, _stream())

    def lookup(self, v, glu_cg):
        hit = self.by_id.get(id(v))
        if hit is None or hit[1] != v.data_ptr() or hit[2] != glu_cg or self.mode != _gemm_mode:
            return None
        return hit[0]


prepacked = None      # set by the trainer for the duration of a training forward
fused_attention = _os.environ.get("DV3_FUSED_ATTN", "1") not in ("0", "")   # one launch for the attention forward


# ----------------------------------------------------------------------------------------------
# operand planes of an activation tensor (include/dv3hip.h, dv3_split_planes_f32)
# ----------------------------------------------------------------------------------------------
# DV3_PLANES=1: every eligible conv layer splits its input once (one HBM pass) and runs the persistent planes
# tap-GEMM; the default (0) keeps the kernels that split while staging.  (Chained layers get their planes from
# the producing epilogue instead -- see ConvLayerFn.)
use_planes = _os.environ.get("DV3_PLANES", "0") not in ("0", "")


def split_planes(x, bits=None, bits_rs=0, scale=1.0, f16=False, B=None, C=None, T=None, x_bs=None, x_rs=None):
    """x fp32 (B, C, T) -> int16 tensor [2][B][C8p][T][8] tagged with its operand type; dropout keep-bits and
    their 1/(1-p) are applied here, once, for the consuming tap-GEMM."""
    if B is None:
        B, C, T = x.shape
    c8p = _round_up(C, 32) // 8
    out = torch.empty(2 * B * c8p * T * 8, dtype=torch.int16, device=x.device)
    d = _planes_desc()
    d.x = x.data_ptr()
    d.x_bs = x_bs if x_bs is not None else x.stride(0)
    d.x_rs = x_rs if x_rs is not None else x.stride(1)
    d.mask, d.mask_rs, d.scale = _ptr(bits), bits_rs, scale
    d.out = out.data_ptr()
    d.B, d.C, d.T, d.dtype = B, C, T, SPLIT_F16 if f16 else SPLIT_BF16
    _lib.call("dv3_split_planes_f32", ctypes.byref(d), _stream())
    out._dv3_f16, out._dv3_c8p = bool(f16), c8p
    return out


def planes_eligible(J, dil, Tin, Tout):
    """the shapes dv3_conv_planes_dispatch takes (the planes carry the dropout mask: no silent fallback)"""
    return _gemm_mode != "f32" and Tin == Tout and (J - 1) * dil <= 64 and J <= 16


# ----------------------------------------------------------------------------------------------
# bf16 activation storage, channel-blocked ("c8", include/dv3hip.h): a (B, C, T) activation as a bf16 tensor
# (B, C8, T, 8), C8 = round_up(C,32)/8.  BASELINE configs 3/4 ("bf16 activations ... fp32 accum"): with
# set_gemm_precision("bf16") the conv stacks keep their activations, saved pre-gates and activation gradients in
# this layout -- every tap-GEMM stages it with plain 16-byte copies and every epilogue writes whole 8-byte halves
# of units.  DV3_BF16_STORAGE=0 keeps fp32 (B, C, T) activations in the bf16 GEMM mode.
# ----------------------------------------------------------------------------------------------
bf16_storage = _os.environ.get("DV3_BF16_STORAGE", "1") not in ("0", "")


def zero_(t):
    """t.zero_() through the library: a fill kernel on the op stream (NOT hipMemsetAsync: its node in a captured step
    replayed with a corrupt fill pattern, csrc/elementwise.hip dv3_memset_b8)"""
    _lib.call("dv3_memset_b8", t.data_ptr(), 0, t.numel() * t.element_size(), _stream())
    return t


def storage_c8():
    """True when the conv stacks should run on channel-blocked bf16 activations"""
    return bf16_storage and _gemm_mode == "bf16"


def c8_groups(C):
    return _round_up(C, 32) // 8


def is_c8(t):
    return t is not None and t.dim() == 4 and t.dtype == torch.bfloat16 and t.shape[-1] == 8


partial_c8_fill = _os.environ.get("DV3_C8_PARTIAL_FILL", "1") not in ("0", "")


def _c8_empty(B, C, T, device):
    """uninitialised c8 tensor; zero-filled when C leaves padding channels / groups (the kernels write whole valid
    groups only; padding must read as zero)"""
    shape = (B, c8_groups(C), T, 8)
    t = torch.empty(shape, dtype=torch.bfloat16, device=device)
    if C % 32:
        # only the groups from the first one with a padding channel on (513 channels: 4 of 68 groups; the whole tensor used to
        # be filled: 56 MB = 9.5 us, eleven times per nyanko step) -- one strided fill kernel
        if partial_c8_fill:
            G, g0 = c8_groups(C), C // 8
            _lib.call("dv3_memset_rows_b8", t.data_ptr() + g0 * T * 16, 0, B, (G - g0) * T * 16, G * T * 16, _stream())
        else:
            zero_(t)
    t._dv3_C = C
    return t


def _c8_C(t, C=None):
    c = getattr(t, "_dv3_C", None)
    if c is None:
        c = C if C is not None else t.shape[1] * 8
    return c


class _ToC8Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(_chk(x, "x"))
        B, C, T = x.shape
        out = _c8_empty(B, C, T, x.device)
        _lib.call("dv3_to_c8_f32", x.data_ptr(), C * T, T, out.data_ptr(), B, C, T, _stream())
        ctx.C = C
        return out

    @staticmethod
    def backward(ctx, g):
        return _from_c8_raw(_c(g), ctx.C)


def _from_c8_raw(xc8, C):
    B, _, T, _ = xc8.shape
    out = torch.empty((B, C, T), dtype=torch.float32, device=xc8.device)
    _lib.call("dv3_from_c8_f32", xc8.data_ptr(), out.data_ptr(), C * T, T, B, C, T, _stream())
    return out


class _FromC8Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xc8, C):
        return _from_c8_raw(_c(xc8), C)

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        B, C, T = g.shape
        out = _c8_empty(B, C, T, g.device)
        _lib.call("dv3_to_c8_f32", g.data_ptr(), C * T, T, out.data_ptr(), B, C, T, _stream())
        return out, None


def to_c8(x):
    """fp32 (B, C, T) -> c8 (differentiable); the tensor remembers its channel count in `_dv3_C`"""
    out = _ToC8Fn.apply(x)
    out._dv3_C = x.shape[1]
    return out


def from_c8(xc8, C=None):
    """c8 -> fp32 (B, C, T) (differentiable)"""
    return _FromC8Fn.apply(xc8, _c8_C(xc8, C))


class MaskPlan(object):
    """A training step's dropout masks in ONE launch (dv3_dropout_keep_c8_multi).  Every Conv1dGLU / HighwayConv1d draws
    its mask when its forward runs (dropout_keep_c8 / dropout_bits_keep below: one ~6 us launch each, 25-35 per step on
    the forward's only queue).  A mask depends on (seed, site number, step counter) only, and the site number is the
    position of the call in the step -- so once two consecutive steps have drawn the same list of (position, kind, shape,
    p), the next step draws the whole list at its start (Trainer.forward_backward -> begin_step) and the layers take
    their masks from it: the same bits the single launches write (tests/test_gpu_model.py).  A call that does not match
    the list falls back to its own launch (same site number), and a step whose list differs from the previous one
    switches the plan off until two steps agree again (ragged eager epochs never plan)."""
    enabled = bool(int(_os.environ.get("DV3_MASK_PLAN", "1")))

    def __init__(self):
        self.plan = None         # tuple of (site offset, kind, B, C, T, p) two consecutive steps agreed on
        self.last = None         # the previous step's list
        self.rec = None          # this step's list (None outside a step)
        self.s0 = 0
        self.ready = {}          # site offset -> (signature, masks)
        self.static = None       # GraphedTrainer's segmented capture: masks in buffers IT fills before each replay
        self.stats = dict(batched_launches=0, planned=0, single=0)

    def begin_step(self, device):
        st = dropout_state
        if st.seed is None:
            st.manual_seed(torch.initial_seed())
        self.rec, self.s0, self.ready = [], st.site, {}
        if not (self.enabled and self.plan) or st.record is not None:
            return
        if self.static is not None:          # drawn by the owner of `static` (its site numbers start at this s0 too)
            self.ready = dict(self.static)
            return
        self.ready, tables = self.build(device, self.s0)
        self.draw(tables, st.dev_offset)

    def build(self, device, s0):
        """mask tensors + descriptor tables of the plan with site numbers from s0 -> ({offset: (sig, masks)}, [tables])"""
        MAX = CONSTS["DV3_DROPOUT_MULTI_MAX"]
        Site = STRUCTS["dv3_dropout_site"]
        ready, tables = {}, []
        for i in range(0, len(self.plan), MAX):
            part = self.plan[i:i + MAX]
            arr = (Site * len(part))()
            for e, sig in zip(arr, part):
                off, kind, B, C, T, p = sig
                keep = torch.empty((B, c8_groups(C), T), dtype=torch.uint8, device=device) if kind != "bits" else None
                bits = torch.empty(B * C * ((T + 31) // 32), dtype=torch.int32, device=device) if kind != "keep" else None
                e.keep, e.bits, e.B, e.C, e.T, e.p, e.site = _ptr(keep), _ptr(bits), B, C, T, p, s0 + off
                ready[off] = (sig, keep if kind == "keep" else bits if kind == "bits" else (bits, (T + 31) // 32, keep))
            tables.append(arr)
        return ready, tables

    def draw(self, tables, dev_offset, stream=None, seed=None):
        for arr in tables:
            _lib.call("dv3_dropout_keep_c8_multi", arr, len(arr), dropout_state.seed if seed is None else seed,
                      _ptr(dev_offset), _stream() if stream is None else stream)
            self.stats["batched_launches"] += 1

    def take(self, kind, B, C, T, p):
        """after dropout_state.next_site(): the planned masks of this call, or None (the caller launches for itself)"""
        if self.rec is None:
            return None
        sig = (dropout_state.site - self.s0, kind, B, C, T, float(p))
        self.rec.append(sig)
        hit = self.ready.pop(sig[0], None)
        if hit is not None and hit[0] == sig:
            self.stats["planned"] += 1
            return hit[1]
        self.stats["single"] += 1
        return None

    def end_step(self):
        rec, self.rec, self.ready = tuple(self.rec or ()), None, {}
        self.plan = rec if (rec and rec == self.last) else None
        self.last = rec


mask_plan = MaskPlan()


def dropout_keep_c8(B, C, T, p, device, name=None):
    """keep-bytes [B][C8][T] of a dropout site over a c8 tensor -- the decisions dropout_bits(B*C, T, ...) draws for
    the same site.  When a test records masks the bits are generated too (and converted), so the record holds the
    layout-independent form the oracle replays."""
    if dropout_state.record is not None and name is not None:
        bits, rs = dropout_bits(B * C, T, p, device, name)
        return mask_bits_to_c8(bits, rs, B, C, T)
    site = dropout_state.next_site()
    hit = mask_plan.take("keep", B, C, T, p)
    if hit is not None:
        return hit
    out = torch.empty((B, c8_groups(C), T), dtype=torch.uint8, device=device)
    _lib.call("dv3_dropout_keep_c8", out.data_ptr(), B, C, T, float(p), dropout_state.seed, site,
              _ptr(dropout_state.dev_offset), _stream())
    return out


def dropout_bits_keep(B, C, T, p, device, name=None):
    """one dropout site in both forms, one launch: -> (keep-bits int32 [B*C][rs], rs, keep-bytes uint8 [B][C8][T])"""
    rs = (T + 31) // 32
    site = dropout_state.next_site()
    hit = mask_plan.take("both", B, C, T, p) if (dropout_state.record is None or name is None) else None
    if hit is not None:
        return hit
    bits = torch.empty(B * C * rs, dtype=torch.int32, device=device)
    keep = torch.empty((B, c8_groups(C), T), dtype=torch.uint8, device=device)
    _lib.call("dv3_dropout_bits_keep", bits.data_ptr(), keep.data_ptr(), B, C, T, float(p), dropout_state.seed, site,
              _ptr(dropout_state.dev_offset), _stream())
    if dropout_state.record is not None and name is not None:
        dropout_state.record[name] = (bits, B * C, T)
    return bits, rs, keep


def pp2_wants_keep_bytes(Cin, J, dil, T, Tout):
    """shapes the 256 x 256 k16 ping-pong tap-GEMM (csrc/conv_gemm_pp2.hip) takes: it stages dropout as keep-bytes"""
    return _gemm_mode in ("f16x3", "bf16x3") and J == 3 and Cin % 32 == 0 and T == Tout and (J - 1) * dil <= 64


def mask_bits_to_c8(bits, bits_rs, B, C, T):
    """dropout keep-bits [B*C][rs] -> keep-bytes [B][C8][T] for the c8 consumers"""
    out = torch.empty((B, c8_groups(C), T), dtype=torch.uint8, device=bits.device)
    _lib.call("dv3_mask_bits_to_c8", bits.data_ptr(), bits_rs, out.data_ptr(), B, C, T, _stream())
    return out


# ----------------------------------------------------------------------------------------------
# raw kernel wrappers
# ----------------------------------------------------------------------------------------------
class GateFuse(object):
    """Round 6: the gate backward of a Conv1dGLU / HighwayConv1d (autograd of modules.py:157-164, 224-226) run by the
    input-gradient launch of the layer that CONSUMES its output (include/dv3hip.h: dv3_conv_desc.pg ...): that launch
    holds dL/d(out) of the producer in registers, so the stand-alone pass over (B, 3C, T) (dv3_gate_bwd_f32: a read of dy
    and of the saved pre-gate pair, a write of the pre-gate gradient, on the queue of backward that has no slack) becomes
    part of its tail.  Request: the producer's saved pair `ab`, its mode / residual flag, its input `x` (highway), `pair`
    = write the pre-gate gradient as pair words (the producer's two gradient GEMMs then stage it without conversion).
    Result (after conv_gemm): dab (B, 2C, T), dres (highway) and the bias partial sums part [2C][n_part]."""
    __slots__ = ("ab", "mode", "residual", "x", "pair", "dab", "dres", "part", "n_part")

    def __init__(self, ab, mode, residual, x=None, pair=False):
        self.ab, self.mode, self.residual, self.x, self.pair = ab, mode, int(residual), x, bool(pair)
        self.dab = self.dres = self.part = None
        self.n_part = 0

    def attach(self, d, B, C, T, device):
        ab = self.ab
        if ab.dtype != torch.float32 or tuple(ab.shape) != (B, 2 * C, T) or not ab.is_contiguous():
            raise RuntimeError("GateFuse: the producer's pre-gate pair must be a contiguous fp32 (B, 2C, T) tensor")
        self.n_part = (B * T + 31) // 32
        self.dab = torch.empty((B, 2 * C, T), dtype=torch.float32, device=device)
        self.part = torch.empty((2 * C, self.n_part), dtype=torch.float32, device=device)
        d.pg, d.dpg, d.pg_part = ab.data_ptr(), self.dab.data_ptr(), self.part.data_ptr()
        d.pg_mode, d.pg_residual, d.pg_pair = self.mode, self.residual, int(self.pair)
        if self.mode == EPI_HIGHWAY:
            x = self.x
            if x is None or x.dtype != torch.float32 or tuple(x.shape) != (B, C, T) or x.stride(2) != 1:
                raise RuntimeError("GateFuse: a highway producer needs its fp32 (B, C, T) input")
            self.dres = torch.empty((B, C, T), dtype=torch.float32, device=device)
            d.pg_x, d.pg_x_bs, d.pg_x_rs, d.dpg_res = x.data_ptr(), x.stride(0), x.stride(1), self.dres.data_ptr()


def conv_gemm(x, a, lda, a_half, *, B, Cin, Tin, M, Tout, J=1, dil=1, padL=0, mode=EPI_LINEAR,
              Cg=0, bias=None, spk=None, spk_strides=(0, 0, 0), r=None, r2=None, residual=0,
              y=None, y_rs=None, ab=None, xmask=None, xmask_rs=0, ymask=None, ymask_rs=0,
              drop_scale=1.0, a_bs=0, store_mode=STORE_BCT, x_bs=None, x_rs=None, tile_hint=0,
              a_split=None, x_planes=None, r_scale=0.0, out_dtype=torch.float32, x_c8=None, out_c8=False,
              xmask_c8=None, ymask_c8=None, gate=None, x_pair=False):
    """dv3_conv_gemm_f32.  x: [B][Cin][Tin] (strides overridable); returns y.  x_planes: the input already
    split into operand planes (split_planes; x may then be None).  bf16 storage: x_c8 = the input as a c8 tensor
    (with its keep-bytes xmask_c8), out_c8 = y / ab written and r / r2 read in c8 (ymask_c8 for DGRAD).
    gate (a GateFuse, DGRAD launches of the split kernels): the tail also runs the gate backward of the layer that
    produced this layer's input and fills gate.dab / .dres / .part; x_pair: x holds pair words (include/dv3hip.h)."""
    gated = mode in (EPI_GLU, EPI_HIGHWAY)
    Cout = Cg if gated else (M // 2 if store_mode == STORE_INTERLEAVE2 else M)
    To = 2 * Tout if store_mode == STORE_INTERLEAVE2 else Tout
    if x_c8 is not None or out_c8:
        return _conv_gemm_c8(x, a, lda, a_half, B=B, Cin=Cin, Tin=Tin, M=M, Tout=Tout, J=J, dil=dil, padL=padL,
                             mode=mode, Cg=Cg, bias=bias, spk=spk, spk_strides=spk_strides, r=r, r2=r2,
                             residual=residual, ab=ab, xmask=xmask, xmask_rs=xmask_rs, ymask=ymask,
                             ymask_rs=ymask_rs, drop_scale=drop_scale, a_split=a_split, r_scale=r_scale, x_c8=x_c8,
                             out_c8=out_c8, xmask_c8=xmask_c8, ymask_c8=ymask_c8, Cout=Cout)
    # bf16 storage (single-term bf16 kernels): x / r / r2 may be bf16 tensors (all three alike), y / ab are written in
    # out_dtype
    in_bf16 = x is not None and x.dtype == torch.bfloat16
    for t_ in (r, r2):
        if t_ is not None and (t_.dtype == torch.bfloat16) != in_bf16:
            raise RuntimeError("conv_gemm: x, r and r2 must share one dtype")
    if y is not None:
        out_dtype = y.dtype
    out_bf16 = out_dtype == torch.bfloat16
    ab_bf16 = ab is not None and ab.dtype == torch.bfloat16
    if ab is not None and out_bf16 and not ab_bf16:
        raise RuntimeError("conv_gemm: a bf16 output needs a bf16 pre-gate save")
    if y is None:
        y = torch.empty((B, Cout, To), dtype=out_dtype, device=(x if x is not None else x_planes).device)
        y_bs, y_rs_ = Cout * To, To
    else:
        y_rs_ = y_rs if y_rs is not None else y.stride(1)
        y_bs = y.stride(0)
    d = _conv_desc()
    d.x = _ptr(x)
    d.a = _ptr(a)
    if x is not None:
        d.x_bs = x_bs if x_bs is not None else x.stride(0)
        d.x_rs = x_rs if x_rs is not None else x.stride(1)
    if x_planes is not None:
        if a_split is None or getattr(a_split, "_dv3_f16", False) != x_planes._dv3_f16:
            raise RuntimeError("x_planes and the split weight image must have the same operand type")
        d.x_planes, d.x_c8p = x_planes.data_ptr(), x_planes._dv3_c8p
    d.a_bs, d.lda, d.a_half = a_bs, lda, a_half
    d.bias = _ptr(bias)
    d.spk = _ptr(spk)
    d.spk_bs, d.spk_rs, d.spk_ts = spk_strides
    if r is not None:
        d.r, d.r_bs, d.r_rs = r.data_ptr(), r.stride(0), r.stride(1)
    if r2 is not None:
        d.r2, d.r2_bs, d.r2_rs = r2.data_ptr(), r2.stride(0), r2.stride(1)
    d.y, d.y_bs, d.y_rs = y.data_ptr(), y_bs, y_rs_
    d.ab = _ptr(ab)
    d.xmask, d.xmask_rs = _ptr(xmask), xmask_rs
    d.xmask_c8 = _ptr(xmask_c8)      # the same keep decisions as bytes [B][C8][T]: the 256 x 256 kernel stages those
    d.ymask, d.ymask_rs = _ptr(ymask), ymask_rs
    d.drop_scale = drop_scale
    d.r_scale = r_scale
    d.io_bf16 = (CONSTS["DV3_IO_IN_BF16"] if in_bf16 else 0) | (CONSTS["DV3_IO_OUT_BF16"] if out_bf16 else 0) | \
        (CONSTS["DV3_IO_AB_BF16"] if (ab_bf16 and not out_bf16) else 0)
    d.B, d.Cin, d.Tin, d.M, d.Cg, d.Tout, d.J, d.dil, d.padL = B, Cin, Tin, M, Cg, Tout, J, dil, padL
    d.mode, d.residual, d.store_mode, d.tile_hint = mode, residual, store_mode, tile_hint
    d.a_split = _ptr(a_split)
    d.split_terms = 0
    if a_split is not None:
        # the image carries its operand type (split_pack / pack_weights tag it): scaled fp16 hi/lo or bf16
        d.split_terms = CONSTS["DV3_SPLIT_F16X3"] if getattr(a_split, "_dv3_f16", False) else \
            (1 if _gemm_mode == "bf16" else 0)
    d.x_pair = int(bool(x_pair))
    if gate is not None:
        if mode != EPI_DGRAD or a_split is None or d.split_terms == 1:
            raise RuntimeError("conv_gemm: the fused gate backward rides on a split-kernel input-gradient launch")
        gate.attach(d, B, M, Tout, y.device)
    if a_split is not None and J == 3 and d.split_terms != 1 and (tile_hint == 0 or streamk == "force") and \
            gate is None and not x_pair:
        ws = _streamk_ws(y.device)
        if ws is not None:
            d.sk_ws, d.sk_ws_bytes = ws
    _lib.call("dv3_conv_gemm_f32", ctypes.byref(d), _stream())
    return y


# stream-K workspaces of the 256 x 256 tap-GEMM kernels (include/dv3hip.h: dv3_conv_desc.sk_ws), one per (device, stream):
# launches that share one must be ordered, which launches on one stream are.  Allocated and zeroed ONCE, eagerly, from the
# ordinary allocator pool -- never inside a graph capture: a buffer taken from a capture's private pool dies with that
# graph while this cache (keyed by the raw stream handle, which torch's stream pool hands out again) would go on serving
# its address: flags that are no longer zero, partial sums read before they are written.  (Round 5: this made the
# nyanko bf16 replay return NaN whenever a deepvoice3 f16x3 replay had run earlier in the process -- ADVICE r4.)  A
# capture whose stream has no workspace yet therefore gets none (tile-per-workgroup form) unless its owner prepared one
# first: train_step.GraphedTrainer calls prepare_streamk_ws(device, capture stream) before it begins to capture.
# DV3_STREAMK=0 turns the form off.
streamk = _os.environ.get("DV3_STREAMK", "1") not in ("0", "")     # "force": also with a forced tile (scripts)
_sk_ws = {}


def _streamk_ws(device):
    if not streamk or _stream_override.handle is not None:
        return None
    key = (device.index, _stream())
    e = _sk_ws.get(key)
    if e is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        n = int(_lib.lib().dv3_conv_streamk_ws_bytes())
        t = torch.zeros((n + 3) // 4, dtype=torch.int32, device=device)
        e = _sk_ws[key] = (t.data_ptr(), n, t)
    return e[0], e[1]


def prepare_streamk_ws(device, stream):
    """create (outside any capture) the stream-K workspace the launches captured on `stream` will use"""
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("prepare_streamk_ws: call before the capture begins")
    with torch.cuda.stream(stream):
        return _streamk_ws(device)


def _conv_gemm_c8(x, a, lda, a_half, *, B, Cin, Tin, M, Tout, J, dil, padL, mode, Cg, bias, spk, spk_strides, r, r2,
                  residual, ab, xmask, xmask_rs, ymask, ymask_rs, drop_scale, a_split, r_scale, x_c8, out_c8, xmask_c8,
                  ymask_c8, Cout):
    """the bf16-storage forms of dv3_conv_gemm_f32 (single-term bf16 kernels): c8 in and / or c8 out"""
    if _gemm_mode != "bf16" or a_split is None:
        raise RuntimeError("c8 activations need the bf16 GEMM mode and a split weight image")
    dev = (x_c8 if x_c8 is not None else x).device
    if out_c8:
        y = _c8_empty(B, Cout, Tout, dev)
        for t_ in (r, r2):
            if t_ is not None and not is_c8(t_):
                raise RuntimeError("conv_gemm: a c8 output reads its residual inputs in c8")
        if ab is not None and not is_c8(ab):
            raise RuntimeError("conv_gemm: a c8 output saves its pre-gate pair in c8")
    else:
        if r is not None or r2 is not None or ab is not None:
            raise RuntimeError("conv_gemm: c8 input with an fp32 output takes no residual / pre-gate save")
        y = torch.empty((B, Cout, Tout), dtype=torch.float32, device=dev)
    d = _conv_desc()
    if x_c8 is not None:
        if x_c8.shape[1] != c8_groups(Cin) or x_c8.shape[2] != Tin or not x_c8.is_contiguous():
            raise RuntimeError("conv_gemm: c8 input shape %s does not match Cin=%d Tin=%d" % (tuple(x_c8.shape), Cin, Tin))
        d.x_planes, d.x_c8p = x_c8.data_ptr(), x_c8.shape[1]
        d.xmask_c8 = _ptr(xmask_c8)
    else:
        d.x, d.x_bs, d.x_rs = x.data_ptr(), x.stride(0), x.stride(1)
        d.xmask, d.xmask_rs = _ptr(xmask), xmask_rs
    d.a, d.a_bs, d.lda, d.a_half = _ptr(a), 0, lda, a_half
    d.bias, d.spk = _ptr(bias), _ptr(spk)
    d.spk_bs, d.spk_rs, d.spk_ts = spk_strides
    d.r, d.r2 = _ptr(r), _ptr(r2)
    d.y, d.ab = y.data_ptr(), _ptr(ab)
    if not out_c8:
        d.y_bs, d.y_rs = Cout * Tout, Tout
        d.ymask, d.ymask_rs = _ptr(ymask), ymask_rs
    else:
        d.ymask_c8 = _ptr(ymask_c8)
    d.drop_scale, d.r_scale = drop_scale, r_scale
    d.io_bf16 = CONSTS["DV3_IO_OUT_C8"] if out_c8 else 0
    d.B, d.Cin, d.Tin, d.M, d.Cg, d.Tout, d.J, d.dil, d.padL = B, Cin, Tin, M, Cg, Tout, J, dil, padL
    d.mode, d.residual, d.store_mode, d.tile_hint = mode, residual, STORE_BCT, 0
    d.a_split, d.split_terms = a_split.data_ptr(), 1
    _lib.call("dv3_conv_gemm_f32", ctypes.byref(d), _stream())
    return y


def pair_words_of(x):
    """fp32 tensor -> the same-shape tensor of PAIR WORDS (include/dv3hip.h: (bf16_rn(v) << 16) | bf16_rn(v - bf16_rn(v)),
    stored in an fp32-typed tensor), computed with torch ops: what dv3_gate_bwd_f32(dab_pair) / the fused tails write.
    For tests and micro-benchmarks that need a pair-word operand without running a gate backward."""
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    w = (hi.view(torch.int16).to(torch.int32) << 16) | (lo.view(torch.int16).to(torch.int32) & 0xffff)
    return w.view(torch.float32)


def wgrad_gemm_c8(g8, x8, *, B, M, Cin, T, J, dil, padL, n_slabs, xmask_c8=None, drop_scale=1.0, rows_of_slabs=False):
    """dv3_wgrad_gemm_f32, c8 form: g8 (B, M/8.., T, 8), x8 (B, Cin/8.., T, 8) -> out [S][J][M][Cin]
    (rows_of_slabs: [J][M][S][Cin], see wgrad_gemm)"""
    d = _wgrad_desc()
    d.g, d.x = g8.data_ptr(), x8.data_ptr()
    d.xmask_c8, d.drop_scale = _ptr(xmask_c8), drop_scale
    if rows_of_slabs:
        out = torch.empty((J, M, n_slabs, Cin), dtype=torch.float32, device=g8.device)
        d.out, d.out_ss, d.ldo = out.data_ptr(), Cin, n_slabs * Cin
    else:
        out = torch.empty((n_slabs, J, M, Cin), dtype=torch.float32, device=g8.device)
        d.out, d.out_ss, d.ldo = out.data_ptr(), J * M * Cin, Cin
    d.B, d.M, d.Cin, d.T, d.Tin, d.J, d.dil, d.padL, d.n_slabs = B, M, Cin, T, T, J, dil, padL, n_slabs
    d.split_bf16, d.k_split, d.c8 = 2, 1, 1
    _lib.call("dv3_wgrad_gemm_f32", ctypes.byref(d), _stream())
    return out


def gate_bwd_c8(dy, ab_or_y, x, *, B, C, T, mode, residual=0, alpha=1.0, want_dres=False, want_dpre=True):
    """dv3_gate_bwd_f32 on c8 tensors -> (dab_or_dpre c8, dres c8, bias_part fp32 [B][rows])"""
    gated = mode in (EPI_GLU, EPI_HIGHWAY)
    dev = dy.device
    rows = 2 * C if gated else C
    dab = _c8_empty(B, rows, T, dev) if (gated or want_dpre) else None
    dres = _c8_empty(B, C, T, dev) if (gated and want_dres) else None
    part = torch.empty((B, rows), dtype=torch.float32, device=dev)
    d = _gate_bwd_desc()
    d.dy, d.ab_or_y, d.x = dy.data_ptr(), _ptr(ab_or_y), _ptr(x)
    d.dab, d.dres, d.bias_part = _ptr(dab), _ptr(dres), part.data_ptr()
    d.alpha = alpha
    d.B, d.C, d.T, d.mode, d.residual, d.c8 = B, C, T, mode, residual, 1
    _lib.call("dv3_gate_bwd_f32", ctypes.byref(d), _stream())
    return dab, dres, part


def wgrad_gemm(g, x, *, B, M, Cin, T, Tin, J=1, dil=1, padL=0, n_slabs=1, xmask=None, xmask_rs=0,
               drop_scale=1.0, out=None, ldo=None, g_bs=None, g_rs=None, x_bs=None, x_rs=None,
               split_bf16=False, k_split=False, rows_of_slabs=False, g_pair=False):
    """dv3_wgrad_gemm_f32 -> out [S][J][M][ldo].  g_pair: g holds pair words (GateFuse(pair=True)).
    rows_of_slabs: the K-split partial sums as [J][M][S][ldo] instead -- the S partial rows of one weight row lie back
    to back (S * ldo contiguous floats), which is how the weight-norm backward reads them (round 3 read S slabs
    1.5 MB apart per element: 36 us per layer at 1.4 TB/s).  Same kernels: only the descriptor's slab stride (ldo) and
    row stride (S * ldo) change."""
    if ldo is None:
        ldo = Cin
    if rows_of_slabs:
        if out is None:
            out = torch.empty((J, M, n_slabs, ldo), dtype=torch.float32, device=g.device)
        out_ss, ldo_k = ldo, n_slabs * ldo
    else:
        if out is None:
            out = torch.empty((n_slabs, J, M, ldo), dtype=torch.float32, device=g.device)
        out_ss, ldo_k = J * M * ldo, ldo
    d = _wgrad_desc()
    d.g, d.g_bs, d.g_rs = g.data_ptr(), (g_bs if g_bs is not None else g.stride(0)), \
        (g_rs if g_rs is not None else g.stride(1))
    d.x, d.x_bs, d.x_rs = x.data_ptr(), (x_bs if x_bs is not None else x.stride(0)), \
        (x_rs if x_rs is not None else x.stride(1))
    d.xmask, d.xmask_rs, d.drop_scale = _ptr(xmask), xmask_rs, drop_scale
    d.out, d.out_ss, d.ldo = out.data_ptr(), out_ss, ldo_k
    d.B, d.M, d.Cin, d.T, d.Tin, d.J, d.dil, d.padL, d.n_slabs = B, M, Cin, T, Tin, J, dil, padL, n_slabs
    d.split_bf16 = (2 if _gemm_mode == "bf16" else 1) if split_bf16 else 0
    d.k_split = int(bool(k_split))
    d.g_pair = int(bool(g_pair))
    _lib.call("dv3_wgrad_gemm_f32", ctypes.byref(d), _stream())
    return out


def gate_bwd(dy, ab_or_y, x, *, B, C, T, mode, residual=0, alpha=1.0, want_dres=False,
             want_dpre=True, pair=False):
    """dv3_gate_bwd_f32 -> (dab_or_dpre, dres, bias_part).  pair (gated modes): dab as pair words (include/dv3hip.h) -- for
    conv_gemm(x_pair=True) / wgrad_gemm(g_pair=True)."""
    gated = mode in (EPI_GLU, EPI_HIGHWAY)
    dev = dy.device
    rows = 2 * C if gated else C
    dab = torch.empty((B, rows, T), dtype=torch.float32, device=dev) if (gated or want_dpre) else None
    dres = torch.empty((B, C, T), dtype=torch.float32, device=dev) if (gated and want_dres) else None
    part = torch.empty((B, rows), dtype=torch.float32, device=dev)
    d = _gate_bwd_desc()
    d.dy, d.ab_or_y, d.x = dy.data_ptr(), _ptr(ab_or_y), _ptr(x)
    d.dab, d.dres, d.bias_part = _ptr(dab), _ptr(dres), part.data_ptr()
    d.alpha = alpha
    d.B, d.C, d.T, d.mode, d.residual = B, C, T, mode, residual
    d.ab_bf16 = int(gated and ab_or_y is not None and ab_or_y.dtype == torch.bfloat16)
    d.dab_pair = int(bool(pair) and gated)
    _lib.call("dv3_gate_bwd_f32", ctypes.byref(d), _stream())
    return dab, dres, part


def weight_norm_bwd(slabs, n_slabs, ldo, v, g, scale, bias_part, n_part, O, I, J, transposed=False,
                    want_bias=True, into=None, rows_of_slabs=False, part_t=False):
    """into = (dv, dg, dbias) gradient buffers to ACCUMULATE into (the parameters' own .grad views in
    the trainer's flat arena) instead of fresh tensors.  rows_of_slabs: the layout wgrad_gemm(rows_of_slabs=True)
    wrote."""
    dev = v.device
    if into is not None:
        dv, dg, dbias = into
    else:
        dv = torch.empty_like(v)
        dg = torch.empty_like(g) if g is not None else None
        dbias = torch.empty(O, dtype=torch.float32, device=dev) if (want_bias and bias_part is not None) else None
    d = _wn_bwd_desc()
    M = J * O if transposed else O
    Jk = 1 if transposed else J
    if rows_of_slabs:
        d.slabs, d.slab_ss, d.ldo, d.n_slabs = slabs.data_ptr(), ldo, n_slabs * ldo, n_slabs
    else:
        d.slabs, d.slab_ss, d.ldo, d.n_slabs = slabs.data_ptr(), Jk * M * ldo, ldo, n_slabs
    d.v, d.g, d.scale = v.data_ptr(), _ptr(g), _ptr(scale)
    d.dv, d.dg = dv.data_ptr(), _ptr(dg)
    d.bias_part, d.n_part, d.dbias = _ptr(bias_part) if dbias is not None else None, n_part, _ptr(dbias)
    d.O, d.I, d.J, d.transposed = O, I, J, int(transposed)
    d.accumulate = int(into is not None)
    d.bias_part_t = int(bool(part_t))          # [O][n_part]: the layout GateFuse.part has
    if into is not None and WnBwdBatch.active:
        WnBwdBatch.add(d, I if transposed else O, (slabs, v, g, scale, bias_part, dv, dg, dbias))
        return dv, dg, dbias
    _lib.call("dv3_weight_norm_bwd_f32", ctypes.byref(d), _stream())
    return dv, dg, dbias


class WnBwdBatch(object):
    """The weight-norm backward of several layers in ONE launch (dv3_weight_norm_bwd_multi).  A single layer's launch
    is a chain of ~24 memory round trips on two workgroups per CU (36 us whatever the layer); a trainer with in-place
    gradients queues the descriptors during backward and flushes every `group` layers (on the stream the layers'
    weight-gradient GEMMs run on, so the order with them is kept) and once at the end.  Off while gradient-ready hooks
    are registered (data parallel: the buckets want each parameter's gradient as early as possible)."""
    active = False
    group = CONSTS["DV3_WN_BWD_MULTI_MAX"]
    _q, _keep, _dv = [], [], set()

    @classmethod
    def add(cls, d, rows, keep):
        if d.dv in cls._dv:             # a parameter used twice (the decoder's last_conv): keep the two updates ordered
            cls.flush()
        c = type(d)()
        ctypes.memmove(ctypes.byref(c), ctypes.byref(d), ctypes.sizeof(d))
        cls._q.append(c)
        cls._keep.append(keep)
        cls._dv.add(d.dv)
        if len(cls._q) >= cls.group:
            cls.flush()

    @classmethod
    def flush(cls):
        """launch what is queued on ops._stream() (the side stream inside a SideStream section: the order with the
        queued layers' weight-gradient GEMMs, which ran there, is the stream order)"""
        if not cls._q:
            return
        q, keep = cls._q, cls._keep
        cls._q, cls._keep, cls._dv = [], [], set()
        arr = (type(q[0]) * len(q))(*q)
        _lib.call("dv3_weight_norm_bwd_multi", ctypes.byref(arr), len(q), _stream())
        if SideStream.stream is not None:
            SideStream.retain(keep)

    @classmethod
    def discard(cls):
        """drop what an aborted backward left queued"""
        cls._q, cls._keep, cls._dv = [], [], set()


def transpose(x, add=None, alpha=1.0):
    """(B,R,C) -> (B,C,R): y = alpha * x^T (+ add)."""
    x = _c(_chk(x, "x"))
    B, R, C = x.shape
    y = torch.empty((B, C, R), dtype=torch.float32, device=x.device)
    _lib.call("dv3_transpose_f32", x.data_ptr(), y.data_ptr(), _ptr(add), B, R, C, float(alpha), _stream())
    return y


def axpby(a, b, alpha):
    a = _c(a)
    out = torch.empty_like(a)
    _lib.call("dv3_axpby_f32", a.data_ptr(), _ptr(_c(b) if b is not None else None), out.data_ptr(),
              a.numel(), float(alpha), _stream())
    return out


# K steps a weight-gradient workgroup gets at least.  0 = the measured rule: 8 for short reductions (B=16: more slabs
# fill the chip), rising with the reduction length to 32 (fp32-storage kernels) / 28 (channel-blocked bf16 kernel) for the
# B=64 ones, where fewer, longer workgroups write fewer partial slabs for the weight-norm backward to read
# (profiles/r06_ksplit_min_steps_ab.txt: -1.3 % / -1.5 % / -0.8 % on the B=64 / B=32 / vctk-bf16 steps; 20 forced at B=16
# costs +1.8 %, 56 at B=64 +7 %; the bf16 kernel's cap was 20 while it transposed in registers -- 32 cost +2 % then -- and
# moved to 28 with the transposing-read kernel, whose K steps are a quarter shorter: 14 costs +1.6 %, 40 +0.3 %).
ksplit_min_steps = int(_os.environ.get("DV3_KSPLIT_MIN_STEPS", "0"))


def _ksplit_count(total_steps, tiles, slots=512, min_steps=None, c8=False):
    if min_steps is None:
        min_steps = ksplit_min_steps or (max(8, min(28, total_steps // 12)) if c8 else max(8, min(32, total_steps // 14)))
    return _ksplit_count_c(total_steps, tiles, slots, min_steps)


@functools.lru_cache(maxsize=None)
def _ksplit_count_c(total_steps, tiles, slots, min_steps):
    """Split-K factor for the bf16x3 wgrad: the grid (tiles x S workgroups) should fill the chip's
    2 x 256 workgroup slots a whole number of times, with at least `min_steps` K steps each."""
    s_max = max(1, total_steps // min_steps)
    if tiles >= slots:
        return 1
    best, best_eff = 1, 0.0
    for s in range(1, min(s_max, slots) + 1):
        blocks = tiles * s
        rounds = -(-blocks // slots)
        eff = blocks / float(rounds * slots)          # occupancy of the last round included
        eff *= 1.0 - 0.008 * s                         # fewer slabs: less partial-sum traffic
        if eff > best_eff:
            best, best_eff = s, eff
    return best


@functools.lru_cache(maxsize=None)
def _slab_count(B, tiles):
    """Split-K factor for wgrad: enough blocks to fill 256 CUs ~2x, at most B."""
    want = max(1, (512 + tiles - 1) // tiles)
    s = min(B, want)
    while B % s:  # equal-sized slabs keep the reduction balanced
        s -= 1
    return max(s, 1)


# ----------------------------------------------------------------------------------------------
# autograd: one conv-like layer (Conv1dGLU / HighwayConv1d / 1x1 Conv1d(+act) / Linear /
# ConvTranspose1d) = pack -> tap-GEMM(+fused tail); backward = gate_bwd -> DGRAD tap-GEMM ->
# wgrad GEMM -> weight-norm backward.
# ----------------------------------------------------------------------------------------------
class LayerCfg(object):
    __slots__ = ("k", "dil", "causal", "mode", "residual", "p", "training", "transposed", "site",
                 "pad_left", "t_out", "out_c8")

    def __init__(self, k=1, dil=1, causal=False, mode=EPI_LINEAR, residual=False, p=0.0,
                 training=False, transposed=False, site=None, pad_left=None, t_out=None, out_c8=None):
        self.k, self.dil, self.causal, self.mode = k, dil, causal, mode
        self.residual, self.p, self.training, self.transposed, self.site = residual, p, training, transposed, site
        self.pad_left, self.t_out = pad_left, t_out   # None: "same" length output (all model layers)
        self.out_c8 = out_c8                           # bf16 storage: None = like the input, True / False forced


def _pad_left(k, dil, causal):
    # modules.py:123-128 (the causal conv pads (k-1)*d on both sides and trims the right)
    return (k - 1) * dil if causal else (k - 1) // 2 * dil


# K-split partial sums of the weight gradient as [J][M][S][Cin] (wgrad_gemm(rows_of_slabs=True)); DV3_SLAB_ROWS=0 = the
# round-3 layout [S][J][M][Cin], for A/B runs
slab_rows_default = _os.environ.get("DV3_SLAB_ROWS", "1") not in ("0", "")

# called as hook(v, g, bias) (entries may be None) once a conv layer's in-place gradients (see ConvLayerFn.backward)
# are final for this step: one call per layer
grad_ready_hooks = []


class SideStream(object):
    """The weight-gradient GEMM + weight-norm backward of a conv layer feed nothing but the parameter gradients, while
    the input-gradient tap-GEMM is the critical path of backward: with a trainer's flat gradient arena (in-place
    gradients) they are issued on a second HIP stream, so the short HBM-bound kernels (weight-norm backward) and the
    partial last rounds of the GEMM grids overlap instead of queueing behind each other.  Inside a captured step the
    fork / join become parallel branches of the hipGraph.  The tensors the side stream reads are kept alive until
    `join()` (the caching allocator would otherwise hand their memory to the next allocation of the main stream)."""
    stream = None          # torch.cuda.Stream while a trainer runs a step, else None (everything on one stream)
    main = None            # the step stream while `stream` is set
    keep = []              # [event | None, [tensors...]] per section, oldest first
    capturing = False      # set by the trainer: no event queries while a hipGraph is being captured
    split_capture = False  # set by GraphedTrainer: the side stream is captured into ITS OWN hipGraphs (include/dv3hip.h:
    split_on_fork = None   # dv3_graph_side_begin / _end); called at every fork point: the segment boundaries
    release_every = 4      # sections between two release points (an event on the side stream + a poll of the oldest)
    _n = 0
    _events = []           # recycled torch.cuda.Event objects

    class _Section(object):
        """`with` body = launches on the side stream.  Only this package's launches are redirected (ops._stream());
        torch's current stream stays the step stream, so tensors allocated inside belong to the step stream's pool
        and are kept alive until the side stream is known to be past them, like the section's inputs (a
        torch.cuda.stream() context costs ~25 us of Python per layer)."""

        def __enter__(self):
            _stream_override.handle = SideStream.stream.cuda_stream

        def __exit__(self, *exc):
            _stream_override.handle = None
            SideStream._release_point()
            return False

    class _MainSection(object):
        """a fork point whose weight gradient stays on the step stream (DV3_SIDE_MAIN: balance of the two queues)"""

        def __enter__(self):
            pass

        def __exit__(self, *exc):
            SideStream._release_point()
            return False

    main_rule, _main_spec = (frozenset(), 0), ""       # (set of fork numbers, every-k) from DV3_SIDE_MAIN
    forks_last = 0         # ... of the step before
    forks = 0              # fork points since the last join (GraphedTrainer sizes its segments from a warm-up step's count)

    @classmethod
    def fork(cls, *tensors):
        # the section's inputs are complete on the step stream: the side stream waits for exactly that point
        cls.forks += 1
        spec = _os.environ.get("DV3_SIDE_MAIN", "")
        if spec != cls._main_spec:
            cls._main_spec, cls.main_rule = spec, _parse_side_main(spec)
        on_main = cls.forks in cls.main_rule[0] or (cls.main_rule[1] and cls.forks % cls.main_rule[1] == 0)
        if cls.split_capture:
            # two separate captures: nothing ties them here; GraphedTrainer closes both every few fork points and the
            # replay orders step-stream segment j before side segment j with an ordinary event
            if cls.split_on_fork is not None:
                cls.split_on_fork()
        else:
            _lib.call("dv3_stream_fork", cls.main.cuda_stream, cls.stream.cuda_stream)
        cls.keep.append([None, [tensors]])
        return cls._main_section if on_main else cls._section

    @classmethod
    def retain(cls, *tensors):
        if cls.keep:
            cls.keep[-1][1].append(tensors)
        else:
            cls.keep.append([None, [tensors]])

    @classmethod
    def _release_point(cls):
        """Every few sections: mark the side stream's position with an event, and drop the operands of the sections an
        earlier, completed event covers -- the memory goes back to the caching allocator layer by layer, as autograd
        frees it on one stream, instead of accumulating until join() (hundreds of MB per layer at B = 64)."""
        cls._n += 1
        if cls.capturing or cls.stream is None or cls._n % cls.release_every:
            return
        done = -1
        for i, (ev, _) in enumerate(cls.keep):
            if ev is not None:
                if not ev.query():
                    break
                done = i
        if done >= 0:
            for ev, _ in cls.keep[:done + 1]:
                if ev is not None:
                    cls._events.append(ev)
            del cls.keep[:done + 1]
        if cls.keep and cls.keep[-1][0] is None:
            ev = cls._events.pop() if cls._events else torch.cuda.Event()
            ev.record(cls.stream)
            cls.keep[-1][0] = ev

    @classmethod
    def join(cls):
        cls.forks_last, cls.forks = cls.forks, 0
        if cls.stream is not None and cls.split_capture:
            # GraphedTrainer ends the last pair of captures itself; the replay joins the streams with an ordinary event.
            # The operands were referenced until here: no later allocation of the step could take their memory.
            for ev, _ in cls.keep:
                if ev is not None:
                    cls._events.append(ev)
            cls.keep = []
            cls._n = 0
            return
        if cls.stream is not None:
            _lib.call("dv3_stream_fork", cls.stream.cuda_stream, cls.main.cuda_stream)
        for ev, _ in cls.keep:
            if ev is not None:
                cls._events.append(ev)
        cls.keep = []
        cls._n = 0


SideStream._section = SideStream._Section()
SideStream._main_section = SideStream._MainSection()


def _parse_side_main(spec):
    """DV3_SIDE_MAIN = "3,7,11" (fork numbers of a step, 1-based) and / or "every:k": those weight gradients are issued on
    the step stream instead of the second one"""
    nums, every = set(), 0
    for tok in spec.replace(" ", "").split(","):
        if tok.startswith("every:"):
            every = int(tok[6:])
        elif tok:
            nums.add(int(tok))
    return nums, every


# Round 6: the gate backward of a gated layer runs in the tail of its consumer's input-gradient launch (GateFuse) when the
# caller marked the layer's output as having no other consumer (`y._dv3_sole = True`: the stack runners of deepvoice3.py /
# nyanko.py do).  DV3_FUSE_GATE=0 restores the stand-alone dv3_gate_bwd_f32 launches (A/B runs).
fuse_gate_bwd = _os.environ.get("DV3_FUSE_GATE", "1") not in ("0", "")
# ... where it pays.  Measured per launch (scripts/r6_gate_fuse_kernels.py, profiles/r06_gate_fuse_kernels.txt): the tail
# costs what the stand-alone kernel costs once that kernel is bandwidth-bound (B = 64: 59 us stand-alone at 5.6 TB/s
# against +73 us of tail at the north-star shape; break-even at 64 x 512 x 150), and half of it where the stand-alone
# launch is latency-bound (B = 16, 256 x 804: 12.3 us against +6.4 us).  Elements (B * C * T) up to which a producer offers
# its gate backward to its consumer (DV3_FUSE_GATE_MAX; 0 = never, a huge value = always):
fuse_gate_max_elems = int(_os.environ.get("DV3_FUSE_GATE_MAX", str(4 << 20)))
# the pre-gate gradient as pair words for the layer's own gradient GEMMs (DV3_PAIR_WORDS=0: fp32, A/B runs)
pair_words = _os.environ.get("DV3_PAIR_WORDS", "1") not in ("0", "")
gate_fuse_stats = {"fused": 0, "standalone": 0}     # gated-layer backwards served either way (tests, bench)


class _GateToken(object):
    """what a gated layer leaves on its output for the consumer's backward (see GateFuse)"""
    __slots__ = ("ab", "mode", "residual", "x", "C", "pair")

    def __init__(self, ab, mode, residual, x, C, pair):
        self.ab, self.mode, self.residual, self.x, self.C, self.pair = ab, mode, residual, x, C, pair


class _GateResult(object):
    """rides on the gradient tensor the consumer's backward returns; the producer's backward takes it only if that very
    tensor -- unmodified -- is what autograd hands it (another consumer of the producer's output would make autograd sum
    the gradients into a different tensor, or bump this one's version)"""
    __slots__ = ("tok", "dab", "dres", "part", "n_part", "pair", "version")

    def __init__(self, tok, gate, version):
        self.tok, self.dab, self.dres, self.part, self.n_part = tok, gate.dab, gate.dres, gate.part, gate.n_part
        self.pair, self.version = gate.pair, version


def mark_sole_consumer(y):
    """the caller's promise that exactly one conv layer consumes y (and nothing else does)"""
    if fuse_gate_bwd and getattr(y, "_dv3_tok", None) is not None:
        y._dv3_sole = True
    return y


class ConvLayerFn(torch.autograd.Function):
    """y = layer(x; v, g, bias[, spk][, r][, r2]).  See LayerCfg.  `packed` may carry a cached
    Packed (eval mode); spk is the additive per-(b,channel[,t]) term on the `a` half (already
    softsign'ed); r / r2 are residual inputs for non-gated modes (each: y = (y + r)*sqrt(.5))."""

    @staticmethod
    def forward(ctx, x, v, g, bias, spk, r, r2, cfg, packed):
        _chk(x, "x")
        x_in = x
        x = _c(x)
        B, Cin, T = x.shape
        mode = cfg.mode
        gated = mode in (EPI_GLU, EPI_HIGHWAY)
        need_grad = any(ctx.needs_input_grad[:7])
        if cfg.transposed:
            I, O, J = v.shape
            M, Cg = J * O, 0
        else:
            O = v.shape[0]
            J = v.shape[2] if v.dim() == 3 else 1
            M, Cg = O, (O // 2 if gated else 0)
        if packed is not None and need_grad and packed.bwd is None:
            packed = None
        # the fused split-only packing when the split-bf16 tap-GEMM is sure to take the shape
        J_ = 1 if cfg.transposed else J
        split_only = (cfg.t_out is None or cfg.t_out == T) and (J_ - 1) * cfg.dil <= 64 and J_ <= 16
        pk = packed
        if pk is None and prepacked is not None and split_only and not cfg.transposed:
            pk = prepacked.lookup(v, Cg)          # packed for the whole model at the top of the step
        if pk is None:
            pk = pack_weights(v, g, glu_cg=Cg, transposed=cfg.transposed, need_bwd=need_grad, split_only=split_only)
        bits, bits_rs, dscale, keep8 = None, 0, 1.0, None
        if cfg.training and cfg.p > 0:
            J_ = 1 if cfg.transposed else (v.shape[2] if v.dim() == 3 else 1)
            if not cfg.transposed and pp2_wants_keep_bytes(Cin, J_, cfg.dil, T, cfg.t_out if cfg.t_out is not None else T):
                bits, bits_rs, keep8 = dropout_bits_keep(B, Cin, T, cfg.p, x.device, cfg.site)
            else:
                bits, bits_rs = dropout_bits(B * Cin, T, cfg.p, x.device, cfg.site)
            dscale = 1.0 / (1.0 - cfg.p)
        padL = _pad_left(J, cfg.dil, cfg.causal) if not cfg.transposed else 0
        if cfg.pad_left is not None:
            padL = cfg.pad_left
        Tout = cfg.t_out if cfg.t_out is not None else T
        if gated and Tout != T:
            raise ValueError("gated layers keep the sequence length")
        # the saved pre-gate pair: bf16 in the bf16 GEMM mode (BASELINE configs 3/4), where it is the largest tensor a
        # training forward writes and the operands are rounded to bf16 anyway; fp32 in the fp32-class modes
        ab_dtype = torch.bfloat16 if (_gemm_mode == "bf16" and pk.fwd_s is not None and split_only) else torch.float32
        ab = torch.empty((B, M, T), dtype=ab_dtype, device=x.device) if (gated and need_grad) else None
        spk_strides = (0, 0, 0)
        if spk is not None:
            spk = _c(spk)
            if spk.dim() == 2:      # (B, Cg): constant over time
                spk_strides = (spk.stride(0), 1, 0)
            else:                   # (B, Cg, T)
                spk_strides = (spk.stride(0), spk.stride(1), 1)
        res_in = x if gated else (_c(r) if r is not None else None)
        r2c = _c(r2) if r2 is not None else None
        xp = None
        if use_planes and pk.fwd_s is not None and planes_eligible(1 if cfg.transposed else J, cfg.dil, T, Tout):
            xp = split_planes(x, bits, bits_rs, dscale, f16=pk.fwd_f16)
        y = conv_gemm(x, pk.fwd, pk.lda, pk.a_half, B=B, Cin=Cin, Tin=T, M=M, Tout=Tout,
                      J=(1 if cfg.transposed else J), dil=cfg.dil, padL=padL, mode=mode, Cg=Cg,
                      bias=bias, spk=spk, spk_strides=spk_strides,
                      r=res_in if (mode == EPI_HIGHWAY or cfg.residual or not gated) else None,
                      r2=r2c, residual=int(cfg.residual), ab=ab, xmask=bits if xp is None else None,
                      xmask_rs=bits_rs if xp is None else 0, x_planes=xp, xmask_c8=keep8 if xp is None else None,
                      drop_scale=dscale, a_split=pk.fwd_s if _gemm_mode != "f32" else None,
                      store_mode=STORE_INTERLEAVE2 if cfg.transposed else STORE_BCT)
        if need_grad:
            # parameters whose .grad lives in the trainer's flat arena take their gradient in place
            # (no AccumulateGrad add kernel per parameter); count the uses so the "gradient final"
            # notification fires once, after the last of them
            leaves = [t for t in (v, g, bias) if t is not None]
            ctx.inplace = bool(leaves) and all(getattr(t, "_dv3_grad_inplace", False) and t.grad is not None and
                                               t.requires_grad for t in leaves)
            ctx.leaves = (v, g, bias) if ctx.inplace else None
            if ctx.inplace:
                v._dv3_pending = getattr(v, "_dv3_pending", 0) + 1
            ctx.cfg, ctx.pk, ctx.dims = cfg, pk, (B, Cin, T, Tout, M, Cg, J, padL)
            ctx.bits, ctx.bits_rs, ctx.dscale = bits, bits_rs, dscale
            ctx.spk_dim = spk.dim() if spk is not None else 0
            ctx.has_r, ctx.has_r2 = (r is not None), (r2 is not None)
            ctx.has_bias = bias is not None
            ctx.save_for_backward(x, v, g, ab if gated else y)
            # round 6 (GateFuse).  As a producer: leave on y what the consumer's input-gradient launch needs to run this
            # layer's gate backward.  As a consumer: remember the producer of x when the caller marked x accordingly.
            ctx.tok = ctx.prod = None
            split_modes = _gemm_mode in ("f16x3", "bf16x3")
            # pair words (include/dv3hip.h): this layer's pre-gate gradient can go to its two gradient GEMMs as the bf16
            # hi / lo pairs they would otherwise build while staging (both on the three-term split kernels, no per-frame
            # speaker-bias gradient reading the tensor as fp32)
            ctx.pair_ok = bool(gated and pair_words and split_modes and not use_planes and pk.bwd_s is not None and
                               split_only and ab.dtype == torch.float32 and (spk is None or spk.dim() == 2) and
                               not (M <= 64 and Cin <= 64))
            if fuse_gate_bwd and split_modes and not use_planes and pk.bwd_s is not None:
                Jd = 1 if cfg.transposed else J
                if gated and ab.dtype == torch.float32 and spk is None and split_only and \
                        B * Cg * T <= fuse_gate_max_elems:
                    ctx.tok = _GateToken(ab, mode, int(cfg.residual), x if mode == EPI_HIGHWAY else None, Cg,
                                         pair=ctx.pair_ok)
                    y._dv3_tok = ctx.tok
                prod = getattr(x_in, "_dv3_tok", None) if (x_in is x and getattr(x_in, "_dv3_sole", False)) else None
                if prod is not None and prod.C == Cin and ctx.needs_input_grad[0] and Tout == T and \
                        (Jd - 1) * cfg.dil <= 64 and Jd <= 16:
                    ctx.prod = prod
        return y

    @staticmethod
    def backward(ctx, dy):
        cfg, pk = ctx.cfg, ctx.pk
        B, Cin, T, Tout, M, Cg, J, padL = ctx.dims
        x, v, g, saved = ctx.saved_tensors
        mode = cfg.mode
        gated = mode in (EPI_GLU, EPI_HIGHWAY)
        dy = _c(dy)
        rs2 = math.sqrt(0.5)
        dr = dr2 = dspk = None
        r_scale = 0.0
        n_part, g_pair, part_t = B, False, False
        if gated:
            # the skip path of a residual GLU passes sqrt(.5) * dy: the DGRAD epilogue reads dy itself
            glu_skip = mode == EPI_GLU and cfg.residual
            fused = getattr(dy, "_dv3_gate", None)
            if fused is not None and ctx.tok is not None and fused.tok is ctx.tok and dy._version == fused.version:
                # the consumer's input-gradient launch already ran this layer's gate backward (GateFuse)
                dab, dres, part = fused.dab, fused.dres, fused.part
                n_part, g_pair, part_t = fused.n_part, fused.pair, True
                dy._dv3_gate = None
                gate_fuse_stats["fused"] += 1
            else:
                g_pair = ctx.pair_ok
                dab, dres, part = gate_bwd(dy, saved, x if mode == EPI_HIGHWAY else None, B=B, C=Cg, T=T,
                                           mode=mode, residual=int(cfg.residual), want_dres=(mode == EPI_HIGHWAY),
                                           pair=g_pair)
                gate_fuse_stats["standalone"] += 1
            if glu_skip:
                dres, r_scale = dy, rs2
            if ctx.spk_dim == 2:
                dspk = part[:, :Cg].contiguous()
            elif ctx.spk_dim == 3:
                dspk = dab[:, :Cg, :]
            gmat, Tg = dab, T
        elif cfg.transposed:
            # dy (B,O,2T): bias sums over the interleaved rows, operand de-interleaved to (B,2O,T)
            O = pk.O
            _, _, part = gate_bwd(dy, None, None, B=B, C=O, T=2 * T, mode=EPI_LINEAR, want_dpre=False)
            gmat = torch.empty((B, 2 * O, T), dtype=torch.float32, device=dy.device)
            _lib.call("dv3_deinterleave2_f32", dy.data_ptr(), gmat.data_ptr(), B, O, T, _stream())
            dres, Tg = None, T
        else:
            # non-gated: y = ((act(pre) + r)*s + r2)*s ; strip the residual scalings first
            alpha = 1.0
            if ctx.has_r2:
                dr2 = axpby(dy, None, rs2)
                alpha *= rs2
            if ctx.has_r:
                dr = axpby(dy, None, alpha * rs2)
                alpha *= rs2
            need_y = mode in (EPI_RELU, EPI_SIGMOID, EPI_SOFTSIGN)
            if need_y and (ctx.has_r or ctx.has_r2):
                raise RuntimeError("activation + fused residual is not used by any layer")
            if mode == EPI_LINEAR and alpha == 1.0:
                _, _, part = gate_bwd(dy, None, None, B=B, C=M, T=Tout, mode=EPI_LINEAR, want_dpre=False)
                gmat = dy
            else:
                gmat, _, part = gate_bwd(dy, saved if need_y else None, None, B=B, C=M, T=Tout, mode=mode,
                                         alpha=alpha)
            dres, Tg = None, Tout
        dx = dv = dg = dbias = None
        Mg = gmat.shape[1]
        if ctx.needs_input_grad[0]:
            # input gradient: transposed, tap-reversed weights; dropout mask on the output side
            Jd = 1 if cfg.transposed else J
            gp = None
            if use_planes and pk.bwd_s is not None and planes_eligible(Jd, cfg.dil, Tg, T):
                gp = split_planes(gmat, f16=False)
            gate = None
            if ctx.prod is not None and gp is None:
                t_ = ctx.prod
                gate = GateFuse(t_.ab, t_.mode, t_.residual, t_.x, pair=t_.pair)
            dx = conv_gemm(gmat, pk.bwd, pk.ldb, 0, B=B, Cin=Mg, Tin=Tg, M=Cin, Tout=T, J=Jd, x_planes=gp,
                           dil=cfg.dil, padL=(Jd - 1) * cfg.dil - padL, mode=EPI_DGRAD, r=dres, r_scale=r_scale,
                           ymask=ctx.bits, ymask_rs=ctx.bits_rs, drop_scale=ctx.dscale,
                           a_split=pk.bwd_s if _gemm_mode != "f32" else None, gate=gate, x_pair=g_pair)
            if gate is not None:
                dx._dv3_gate = _GateResult(ctx.prod, gate, dx._version)
        if ctx.needs_input_grad[1]:
            Jd = 1 if cfg.transposed else J
            tiles = ((Mg + 127) // 128) * ((Cin + 127) // 128) * Jd
            x3 = _gemm_mode != "f32" and not (Mg <= 64 and Cin <= 64)
            if x3 and Jd == 3 and _os.environ.get("DV3_WGRAD_TILE", "0") in ("0", "3"):
                # one 8-wave workgroup per (tile, slab) serves the three taps: one workgroup per CU
                S = _ksplit_count(B * ((Tg + 31) // 32), tiles // 3, slots=256)
            elif x3:   # split-K over contiguous (batch, chunk) ranges: size the grid to 2 workgroups per CU
                S = _ksplit_count(B * ((Tg + 31) // 32), tiles)
            else:
                S = _slab_count(B, tiles)
            v3 = v if v.dim() == 3 else v.unsqueeze(-1)
            slab_rows = slab_rows_default and S > 1
            side = SideStream.fork(gmat, x, ctx.bits, part, dy) if (ctx.inplace and SideStream.stream is not None) \
                else contextlib.nullcontext()
            with side:
                slabs = wgrad_gemm(gmat, x, B=B, M=Mg, Cin=Cin, T=Tg, Tin=T, J=Jd, dil=cfg.dil, padL=padL,
                                   n_slabs=S, xmask=ctx.bits, xmask_rs=ctx.bits_rs, drop_scale=ctx.dscale,
                                   split_bf16=x3, k_split=x3, rows_of_slabs=slab_rows, g_pair=g_pair)
                if ctx.inplace:
                    pv, pg, pb = ctx.leaves
                    weight_norm_bwd(slabs, S, Cin, _c(v3), _c(g) if g is not None else None, pk.scale, part, n_part,
                                    pk.O, pk.I, pk.J, cfg.transposed, want_bias=ctx.has_bias,
                                    into=(pv.grad, pg.grad if pg is not None else None,
                                          pb.grad if pb is not None else None), rows_of_slabs=slab_rows,
                                    part_t=part_t)
                    if SideStream.stream is not None:
                        SideStream.retain(slabs)
                    pv._dv3_pending -= 1
                    if pv._dv3_pending == 0:
                        for hook in grad_ready_hooks:
                            hook(pv, pg, pb)
            if ctx.inplace:
                dv = dg = dbias = None
            else:
                dv, dg, dbias = weight_norm_bwd(slabs, S, Cin, _c(v3), _c(g) if g is not None else None,
                                                pk.scale, part, n_part, pk.O, pk.I, pk.J, cfg.transposed,
                                                want_bias=ctx.has_bias, rows_of_slabs=slab_rows, part_t=part_t)
                dv = dv.view_as(v)
        return dx, dv, dg, dbias, dspk, dr, dr2, None, None


# ----------------------------------------------------------------------------------------------
# per-frame speaker biases of a block of Conv1dGLU layers (include/dv3hip.h: dv3_speaker_bias_fwd_f32 / _bwd_f32)
# ----------------------------------------------------------------------------------------------
fused_speaker_bias = _os.environ.get("DV3_FUSED_SPK", "1") not in ("0", "")
_spk_layer_t, _spk_desc_t = STRUCTS["dv3_spk_layer"], STRUCTS["dv3_spk_desc"]
SPK_MAX_LAYERS = CONSTS["DV3_SPK_MAX_LAYERS"]


class SpeakerBiasBlockFn(torch.autograd.Function):
    """softsign(speaker_proj_l(e)) for the L <= 16 Conv1dGLU layers of a block that share the dropped per-frame speaker
    embedding e (B, E, T) (modules.py:158-162 with deepvoice3.py:78-81): one launch forward, three backward, instead of
    six launch-bound launches per layer.  params = (v_1, g_1, bias_1, ..., v_L, g_L, bias_L); -> L tensors (B, C_l, T).
    Parameter gradients go straight into p.grad when the trainer marked the parameters (`_dv3_grad_inplace`), like
    ConvLayerFn's."""

    @staticmethod
    def forward(ctx, e, holder, *params):
        L = len(params) // 3
        ctx.holder = holder
        B, E, T = e.shape
        if e.stride(2) != 1:
            e = e.contiguous()
        layers = (_spk_layer_t * L)()
        outs = []
        for l in range(L):
            v, g, b = params[3 * l:3 * l + 3]
            C = v.shape[0]
            if not v.is_contiguous() or v.numel() != C * E:
                raise RuntimeError("speaker_bias_block: (C, E) contiguous weights expected")
            o = torch.empty((B, C, T), dtype=torch.float32, device=e.device)
            y = layers[l]
            y.v, y.g, y.bias, y.out, y.C = v.data_ptr(), _ptr(g), _ptr(b), o.data_ptr(), C
            outs.append(o)
        d = _spk_desc_t()
        d.e, d.e_bs, d.e_rs, d.B, d.E, d.T, d.n_layers = e.data_ptr(), e.stride(0), e.stride(1), B, E, T, L
        _lib.call("dv3_speaker_bias_fwd_f32", ctypes.byref(d), layers, _stream())
        ctx.L = L
        ctx.save_for_backward(e, *params, *outs)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        L = ctx.L
        saved = ctx.saved_tensors
        e, params, outs = saved[0], saved[1:1 + 3 * L], saved[1 + 3 * L:]
        B, E, T = e.shape
        leaves = [p for p in params if p is not None]
        inplace = all(getattr(p, "_dv3_grad_inplace", False) and p.grad is not None and p.requires_grad for p in leaves)
        layers = (_spk_layer_t * L)()
        keep, grads = [], []
        for l in range(L):
            v, g, b = params[3 * l:3 * l + 3]
            C = v.shape[0]
            y = layers[l]
            y.v, y.g, y.bias, y.out, y.C = v.data_ptr(), _ptr(g), _ptr(b), outs[l].data_ptr(), C
            c8g = ctx.holder.pop(l, None)      # a c8 layer left its pre-gate gradient here (ConvLayerC8Fn.backward)
            if c8g is not None:
                # the real gradient travelled through the holder; what autograd delivers for this output must be the
                # zero-stride stand-in of ConvLayerC8Fn.backward -- anything else is a SECOND consumer of the bias tensor
                # (a hook, a regulariser) whose gradient would be dropped silently here
                do = douts[l]
                z = _spk_dummy.get(e.device)
                if do is not None and not (z is not None and do.data_ptr() == z.data_ptr() and all(st == 0 for st in do.stride())):
                    raise RuntimeError("speaker_bias_block: the bias of layer %d has a consumer besides its Conv1dGLU layer; "
                                       "its gradient can not be combined with the c8 hand-over (set DV3_FUSED_SPK=0)" % l)
                keep.append(c8g)
                y.dout, y.dout_c8p = c8g.data_ptr(), c8g.shape[1]
            else:
                do = douts[l]
                if do is None:
                    do = torch.zeros((B, C, T), dtype=torch.float32, device=e.device)
                elif do.stride(2) != 1 or do.dtype != torch.float32:
                    do = do.float().contiguous()
                keep.append(do)
                y.dout, y.dout_bs, y.dout_rs = do.data_ptr(), do.stride(0), do.stride(1)
            if inplace:
                gv, gg, gb = v.grad, (g.grad if g is not None else None), (b.grad if b is not None else None)
            else:
                gv = torch.zeros_like(v)
                gg = torch.zeros_like(g) if g is not None else None
                gb = torch.zeros_like(b) if b is not None else None
            y.dv, y.dg, y.dbias = gv.data_ptr(), _ptr(gg), _ptr(gb)
            grads += [gv, gg, gb]
        d = _spk_desc_t()
        d.e, d.e_bs, d.e_rs, d.B, d.E, d.T, d.n_layers = e.data_ptr(), e.stride(0), e.stride(1), B, E, T, L
        n = _lib.lib().dv3_speaker_bias_bwd_scratch_floats(ctypes.byref(d), layers)
        if n <= 0:
            raise RuntimeError("speaker_bias_block: workspace size")
        scratch = torch.empty(n, dtype=torch.float32, device=e.device)
        de = torch.empty((B, E, T), dtype=torch.float32, device=e.device)
        d.de, d.scratch, d.scratch_floats = de.data_ptr(), scratch.data_ptr(), n
        _lib.call("dv3_speaker_bias_bwd_f32", ctypes.byref(d), layers, _stream())
        if inplace:
            for l in range(L):
                for hook in grad_ready_hooks:
                    hook(*params[3 * l:3 * l + 3])
            return (de if ctx.needs_input_grad[0] else None, None) + (None,) * (3 * L)
        return (de if ctx.needs_input_grad[0] else None, None) + tuple(grads)


def speaker_bias_block(e, layers):
    """layers: [(v (C, E), g (C, 1) | None, bias (C) | None)] -> [softsign(speaker_proj_l(e))] (B, C_l, T) fp32"""
    outs = []
    for i in range(0, len(layers), SPK_MAX_LAYERS):
        chunk = layers[i:i + SPK_MAX_LAYERS]
        holder = {}
        got = SpeakerBiasBlockFn.apply(e, holder, *[t for lay in chunk for t in lay])
        for l, o in enumerate(got):
            # a bf16-storage layer that consumes this bias hands its pre-gate gradient (a c8 tensor) over through
            # `holder` instead of converting its first C channels to an fp32 (B, C, T) tensor for autograd
            o._dv3_spk_slot = (holder, l)
        outs += list(got)
    return outs


_spk_dummy = {}


def _spk_dummy_grad(shape, device):
    """a zero-stride fp32 stand-in of the given shape (autograd wants a gradient of the bias's shape; the real one travels
    through SpeakerBiasBlockFn's holder)"""
    z = _spk_dummy.get(device)
    if z is None:
        z = _spk_dummy[device] = torch.zeros(1, dtype=torch.float32, device=device)
    return z.expand(*shape)


class ConvLayerC8Fn(torch.autograd.Function):
    """ConvLayerFn on bf16 channel-blocked activations (bf16 GEMM mode, BASELINE configs 3/4).  x is a c8 tensor
    or an fp32 (B, C, T) one; the output is c8 when cfg.out_c8 (default: when x is), else fp32 (B, C, T).  With a
    c8 output the residual inputs r / r2, the saved pre-gate pair and the incoming gradient are c8 as well.
    Same-length Conv1d / Linear layers with 1 or 3 taps (everything inside the model stacks)."""

    @staticmethod
    def forward(ctx, x, v, g, bias, spk, r, r2, cfg, packed):
        x8 = is_c8(x)
        out8 = x8 if cfg.out_c8 is None else bool(cfg.out_c8)
        if not (x8 or out8) or cfg.transposed:
            raise RuntimeError("ConvLayerC8Fn: needs a c8 side and a plain Conv1d / Linear layer")
        x = _c(x)
        mode = cfg.mode
        gated = mode in (EPI_GLU, EPI_HIGHWAY)
        O = v.shape[0]
        Cin = v.shape[1]
        J = v.shape[2] if v.dim() == 3 else 1
        B, T = x.shape[0], x.shape[2]
        if (x.shape[1] != c8_groups(Cin)) if x8 else (x.shape[1] != Cin):
            raise RuntimeError("conv layer: input %s does not carry %d channels" % (tuple(x.shape), Cin))
        M, Cg = O, (O // 2 if gated else 0)
        if gated and not (x8 and out8):
            raise RuntimeError("gated c8 layers run c8 -> c8 (convert at the stack entry)")
        if J not in (1, 3) or (cfg.t_out is not None and cfg.t_out != T) or (J - 1) * cfg.dil > 64:
            raise RuntimeError("c8 layers: same-length convolutions with 1 or 3 taps")
        need_grad = any(ctx.needs_input_grad[:7])
        pk = packed
        if pk is not None and (pk.fwd_s is None or (need_grad and pk.bwd_s is None)):
            pk = None
        if pk is None and prepacked is not None:
            pk = prepacked.lookup(v, Cg)
        if pk is None:
            pk = pack_weights(v, g, glu_cg=Cg, need_bwd=need_grad, split_only=True)
        bits, bits_rs, dscale, keep8 = None, 0, 1.0, None
        if cfg.training and cfg.p > 0:
            dscale = 1.0 / (1.0 - cfg.p)
            if x8:
                keep8 = dropout_keep_c8(B, Cin, T, cfg.p, x.device, cfg.site)
            else:       # fp32 input: keep-bits for the staging of the fp32 operand
                bits, bits_rs = dropout_bits(B * Cin, T, cfg.p, x.device, cfg.site)
        padL = cfg.pad_left if cfg.pad_left is not None else _pad_left(J, cfg.dil, cfg.causal)
        ab = _c8_empty(B, M, T, x.device) if (gated and need_grad) else None
        spk_strides = (0, 0, 0)
        if spk is not None:
            spk = _c(spk)
            spk_strides = (spk.stride(0), 1, 0) if spk.dim() == 2 else (spk.stride(0), spk.stride(1), 1)
        if gated:
            res_in = x if (mode == EPI_HIGHWAY or cfg.residual) else None
        else:
            res_in = _c(r) if r is not None else None
        r2c = _c(r2) if r2 is not None else None
        y = conv_gemm(None if x8 else x, None, pk.lda, pk.a_half, B=B, Cin=Cin, Tin=T, M=M, Tout=T, J=J, dil=cfg.dil,
                      padL=padL, mode=mode, Cg=Cg, bias=bias, spk=spk, spk_strides=spk_strides, r=res_in, r2=r2c,
                      residual=int(cfg.residual), ab=ab, xmask=None if x8 else bits, xmask_rs=0 if x8 else bits_rs,
                      drop_scale=dscale, a_split=pk.fwd_s, x_c8=x if x8 else None, out_c8=out8, xmask_c8=keep8)
        if need_grad:
            leaves = [t for t in (v, g, bias) if t is not None]
            ctx.inplace = bool(leaves) and all(getattr(t, "_dv3_grad_inplace", False) and t.grad is not None and
                                               t.requires_grad for t in leaves)
            ctx.leaves = (v, g, bias) if ctx.inplace else None
            if ctx.inplace:
                v._dv3_pending = getattr(v, "_dv3_pending", 0) + 1
            ctx.cfg, ctx.pk, ctx.dims = cfg, pk, (B, Cin, T, M, Cg, J, padL)
            ctx.bits, ctx.bits_rs, ctx.dscale, ctx.keep8 = bits, bits_rs, dscale, keep8
            ctx.x8, ctx.out8 = x8, out8
            ctx.spk_dim = spk.dim() if spk is not None else 0
            ctx.spk_slot = getattr(spk, "_dv3_spk_slot", None) if spk is not None else None
            ctx.has_r, ctx.has_r2, ctx.has_bias = (r is not None), (r2 is not None), bias is not None
            ctx.save_for_backward(x, v, g, ab if gated else y)
        return y

    @staticmethod
    def backward(ctx, dy):
        cfg, pk = ctx.cfg, ctx.pk
        B, Cin, T, M, Cg, J, padL = ctx.dims
        x, v, g, saved = ctx.saved_tensors
        mode = cfg.mode
        gated = mode in (EPI_GLU, EPI_HIGHWAY)
        dy = _c(dy)
        rs2 = math.sqrt(0.5)
        dr = dr2 = dspk = None
        dres, r_scale = None, 0.0
        if gated:          # c8 -> c8
            glu_skip = mode == EPI_GLU and cfg.residual
            gmat, dres, part = gate_bwd_c8(dy, saved, x if mode == EPI_HIGHWAY else None, B=B, C=Cg, T=T, mode=mode,
                                           residual=int(cfg.residual), want_dres=(mode == EPI_HIGHWAY))
            if glu_skip:
                dres, r_scale = dy, rs2
            if ctx.spk_dim == 2:
                dspk = part[:, :Cg].contiguous()
            elif ctx.spk_dim == 3 and ctx.spk_slot is not None and fused_speaker_bias:
                holder, slot = ctx.spk_slot          # the block's backward reads the c8 tensor itself
                holder[slot] = gmat
                dspk = _spk_dummy_grad((B, Cg, T), gmat.device)
            elif ctx.spk_dim == 3:      # per-frame speaker bias: the gradient of the `a` half, fp32 (B, Cg, T)
                dspk = torch.empty((B, Cg, T), dtype=torch.float32, device=gmat.device)
                _lib.call("dv3_from_c8_head_f32", gmat.data_ptr(), gmat.shape[1], dspk.data_ptr(), Cg * T, T, B, Cg, T,
                          _stream())
            g8 = gmat
        else:
            alpha = 1.0
            if ctx.has_r2:
                dr2 = dy * rs2
                alpha *= rs2
            if ctx.has_r:
                dr = dy * (alpha * rs2)
                alpha *= rs2
            need_y = mode in (EPI_RELU, EPI_SIGMOID, EPI_SOFTSIGN)
            if need_y and (ctx.has_r or ctx.has_r2):
                raise RuntimeError("activation + fused residual is not used by any layer")
            if ctx.out8:
                plain = mode == EPI_LINEAR and alpha == 1.0
                gm, _, part = gate_bwd_c8(dy, saved if need_y else None, None, B=B, C=M, T=T, mode=mode, alpha=alpha,
                                          want_dpre=not plain)
                g8 = dy if plain else gm
            else:          # fp32 (B, M, T) gradient of a c8 -> fp32 layer: blocked once, for both gradient GEMMs
                if mode == EPI_LINEAR and alpha == 1.0:
                    _, _, part = gate_bwd(dy, None, None, B=B, C=M, T=T, mode=EPI_LINEAR, want_dpre=False)
                    gmat = dy
                else:
                    gmat, _, part = gate_bwd(dy, saved if need_y else None, None, B=B, C=M, T=T, mode=mode, alpha=alpha)
                g8 = _ToC8Fn.apply(gmat)
        dx = dv = dg = dbias = None
        if ctx.needs_input_grad[0]:
            dpad = (J - 1) * cfg.dil - padL
            if ctx.x8:      # dx in c8: keep-bytes of the input dropout on the output side
                dx = conv_gemm(None, None, pk.ldb, 0, B=B, Cin=M, Tin=T, M=Cin, Tout=T, J=J,
                               dil=cfg.dil, padL=dpad, mode=EPI_DGRAD, r=dres, r_scale=r_scale, drop_scale=ctx.dscale,
                               a_split=pk.bwd_s, x_c8=g8, out_c8=True, ymask_c8=ctx.keep8)
            else:           # fp32 input (the attention context): c8 gradient operand, fp32 (B, Cin, T) result
                dx = conv_gemm(None, None, pk.ldb, 0, B=B, Cin=M, Tin=T, M=Cin, Tout=T, J=J, dil=cfg.dil, padL=dpad,
                               mode=EPI_DGRAD, ymask=ctx.bits, ymask_rs=ctx.bits_rs, drop_scale=ctx.dscale,
                               a_split=pk.bwd_s, x_c8=g8, out_c8=False)
        if ctx.needs_input_grad[1]:
            tiles = ((M + 127) // 128) * ((Cin + 127) // 128)
            v3 = v if v.dim() == 3 else v.unsqueeze(-1)
            S = _ksplit_count(B * ((T + 31) // 32), tiles, slots=256, c8=True)
            slab_rows = slab_rows_default and S > 1
            side = SideStream.fork(g8, x, ctx.bits, ctx.keep8, part, dy) if (ctx.inplace and SideStream.stream is not None) \
                else contextlib.nullcontext()
            with side:
                if ctx.x8:
                    x8t, keep8 = x, ctx.keep8
                else:
                    x8t = _ToC8Fn.apply(x)
                    keep8 = mask_bits_to_c8(ctx.bits, ctx.bits_rs, B, Cin, T) if ctx.bits is not None else None
                slabs = wgrad_gemm_c8(g8, x8t, B=B, M=M, Cin=Cin, T=T, J=J, dil=cfg.dil, padL=padL, n_slabs=S,
                                      xmask_c8=keep8, drop_scale=ctx.dscale, rows_of_slabs=slab_rows)
                if ctx.inplace:
                    pv, pg, pb = ctx.leaves
                    weight_norm_bwd(slabs, S, Cin, _c(v3), _c(g) if g is not None else None, pk.scale, part, B,
                                    pk.O, pk.I, pk.J, False, want_bias=ctx.has_bias,
                                    into=(pv.grad, pg.grad if pg is not None else None,
                                          pb.grad if pb is not None else None), rows_of_slabs=slab_rows)
                    if SideStream.stream is not None:
                        SideStream.retain(slabs, x8t, keep8)
                    pv._dv3_pending -= 1
                    if pv._dv3_pending == 0:
                        for hook in grad_ready_hooks:
                            hook(pv, pg, pb)
            if not ctx.inplace:
                dv, dg, dbias = weight_norm_bwd(slabs, S, Cin, _c(v3), _c(g) if g is not None else None,
                                                pk.scale, part, B, pk.O, pk.I, pk.J, False, want_bias=ctx.has_bias,
                                                rows_of_slabs=slab_rows)
                dv = dv.view_as(v)
        return dx, dv, dg, dbias, dspk, dr, dr2, None, None


def _c8_layer_ok(v, cfg, out8, T):
    """the layer forms the c8 kernels serve: same-length Conv1d / Linear with 1 or 3 taps"""
    J = v.shape[2] if v.dim() == 3 else 1
    if cfg.transposed or J not in (1, 3) or (cfg.t_out is not None and cfg.t_out != T) or (J - 1) * cfg.dil > 64:
        return False
    # gated layers: the a / gate halves must start on group boundaries; plain layers take any channel count
    return cfg.mode not in (EPI_GLU, EPI_HIGHWAY) or v.shape[0] % 16 == 0


def conv_layer(x, v, g, bias, cfg, spk=None, r=None, r2=None, packed=None):
    x8, want8 = is_c8(x), getattr(cfg, "out_c8", None)
    if (x8 or want8) and not _c8_layer_ok(v, cfg, x8 if want8 is None else bool(want8), x.shape[2]):
        # a form the c8 kernels do not serve (5- or 7-tap layers of small configurations): this layer runs on
        # fp32 (B, C, T) tensors between two conversions
        y = ConvLayerFn.apply(from_c8(x, v.shape[1]) if x8 else x, v, g, bias, spk,
                              from_c8(r) if is_c8(r) else r, from_c8(r2) if is_c8(r2) else r2, cfg, packed)
        out8 = x8 if want8 is None else bool(want8)
        return to_c8(y) if out8 else y
    if x8 or want8:
        y = ConvLayerC8Fn.apply(x, v, g, bias, spk, r, r2, cfg, packed)
        if is_c8(y):
            gated = cfg.mode in (EPI_GLU, EPI_HIGHWAY)
            y._dv3_C = v.shape[0] // 2 if gated else v.shape[0]
        return y
    return ConvLayerFn.apply(x, v, g, bias, spk, r, r2, cfg, packed)


# ----------------------------------------------------------------------------------------------
# attention core (deepvoice3.py:143-171): S = q^T k -> mask/softmax/dropout -> ctx = v Pd^T sqrt(Tk)
# q (B,E,Tq), k (B,E,Tk), v (B,E,Tk) all BCT; returns ctx (B,E,Tq) BCT and P (B,Tq,Tk).
# ----------------------------------------------------------------------------------------------
def _attn_split(forward=False):
    """the batched attention products (torch.bmm, deepvoice3.py:143,167) follow the GEMM mode; in the
    default mode the FORWARD context product stays on the exact kernel like the scores (fp32-class
    outputs; < 0.3 % of the step's FLOPs), the four gradient products use the bf16 split; the
    incremental decode (Tq == 1) keeps the exact kernel (a 1-row product is not MFMA work)"""
    if forward and _gemm_mode == "f16x3":
        return False
    return _gemm_mode != "f32"


class AttnCoreFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, key_len, last_attended, cfg):
        q, k, v = _c(_chk(q, "q")), _c(_chk(k, "k")), _c(_chk(v, "v"))
        B, E, Tq = q.shape
        Tk = k.shape[2]
        p_drop, training, win_back, win_ahead, site = cfg
        dev = q.device
        bits, bits_rs, dscale = None, 0, 1.0
        if training and p_drop > 0:
            bits, bits_rs = dropout_bits(B * Tq, Tk, p_drop, dev, site)
            dscale = 1.0 / (1.0 - p_drop)
        if fused_attention and last_attended is None and Tk <= 511 and Tq > 1:
            # scores -> mask -> softmax -> dropout -> context in ONE launch (exact fp32 MFMA, csrc/attention.hip)
            vT = transpose(v)                                   # (B, Tk, E): the values in the reference's layout
            ctxv = torch.empty((B, E, Tq), dtype=torch.float32, device=dev)
            P = torch.empty((B, Tq, Tk), dtype=torch.float32, device=dev)
            pd = torch.empty_like(P)
            d = _attn_fwd_desc()
            d.q, d.k, d.vT, d.key_len = q.data_ptr(), k.data_ptr(), vT.data_ptr(), _ptr(key_len)
            d.mask, d.mask_rs, d.drop_scale = _ptr(bits), bits_rs, dscale
            # deepvoice3.py:170-171: s * sqrt(1/s), s = keys of the batch -- a device scalar when the batch is padded beyond
            # its own longest text (ValidLengths; the caller passes that length as every item's key_len)
            vs = valid.scale if valid is not None else None
            d.pd_scale = 1.0 if vs is not None else Tk * math.sqrt(1.0 / Tk)
            d.pd_scale_dev = _ptr(vs)
            d.ctx, d.P, d.pd = ctxv.data_ptr(), P.data_ptr(), pd.data_ptr()
            d.B, d.E, d.Tq, d.Tk = B, E, Tq, Tk
            _lib.call("dv3_attn_fwd_f32", ctypes.byref(d), _stream())
            if any(ctx.needs_input_grad[:3]):
                ctx.save_for_backward(q, k, v, P, pd)
                ctx.bits, ctx.bits_rs, ctx.dscale, ctx.pd_scale = bits, bits_rs, dscale, d.pd_scale
                ctx.scale_dev = vs
            return ctxv, P
        # scores: per-batch operand A = q[b] as [Cin=E][lda=Tq]
        S = conv_gemm(k, q, Tq, 0, B=B, Cin=E, Tin=Tk, M=Tq, Tout=Tk, a_bs=E * Tq)
        pd = torch.empty_like(S)
        d = _softmax_desc()
        d.s, d.pd, d.key_len, d.last_attended = S.data_ptr(), pd.data_ptr(), _ptr(key_len), _ptr(last_attended)
        d.mask, d.mask_rs, d.drop_scale = _ptr(bits), bits_rs, dscale
        vs = valid.scale if valid is not None else None
        d.pd_scale = 1.0 if vs is not None else Tk * math.sqrt(1.0 / Tk)   # deepvoice3.py:170-171: s * sqrt(1/s)
        d.pd_scale_dev = _ptr(vs)
        d.B, d.Tq, d.Tk, d.win_back, d.win_ahead = B, Tq, Tk, win_back, win_ahead
        _lib.call("dv3_attn_softmax_f32", ctypes.byref(d), _stream())
        P = S
        # context: out[b][e][t] = sum_n v[b][e][n] * pd[b][t][n]
        ctxv = wgrad_gemm(v, pd, B=B, M=E, Cin=Tq, T=Tk, Tin=Tk, n_slabs=B,
                          split_bf16=_attn_split(forward=True)).view(B, E, Tq)
        if any(ctx.needs_input_grad[:3]):
            ctx.save_for_backward(q, k, v, P, pd)
            ctx.bits, ctx.bits_rs, ctx.dscale, ctx.pd_scale = bits, bits_rs, dscale, d.pd_scale
            ctx.scale_dev = vs
        return ctxv, P

    @staticmethod
    def backward(ctx, dctx, dP):
        q, k, v, P, pd = ctx.saved_tensors
        B, E, Tq = q.shape
        Tk = k.shape[2]
        dctx = _c(dctx)
        # dv[b][e][n] = sum_t dctx[b][e][t] * pd[b][t][n]  -> needs pd^T (B,Tk,Tq)
        pdT = transpose(pd)
        dv = wgrad_gemm(dctx, pdT, B=B, M=E, Cin=Tk, T=Tq, Tin=Tq, n_slabs=B, split_bf16=_attn_split()).view(B, E, Tk)
        # dpd[b][t][n] = sum_e dctx[b][e][t] * v[b][e][n]   (per-batch operand A = dctx[b])
        dpd = conv_gemm(v, dctx, Tq, 0, B=B, Cin=E, Tin=Tk, M=Tq, Tout=Tk, a_bs=E * Tq)
        dS = torch.empty_like(P)
        d = _softmax_bwd_desc()
        d.p, d.dpd, d.ds = P.data_ptr(), dpd.data_ptr(), dS.data_ptr()
        d.dp_direct = _ptr(_c(dP)) if dP is not None else None
        d.mask, d.mask_rs, d.drop_scale = _ptr(ctx.bits), ctx.bits_rs, ctx.dscale * ctx.pd_scale
        d.scale_dev = _ptr(ctx.scale_dev)
        d.B, d.Tq, d.Tk = B, Tq, Tk
        _lib.call("dv3_attn_softmax_bwd_f32", ctypes.byref(d), _stream())
        # dq[b][e][t] = sum_n k[b][e][n] * dS[b][t][n]
        dq = wgrad_gemm(k, dS, B=B, M=E, Cin=Tq, T=Tk, Tin=Tk, n_slabs=B, split_bf16=_attn_split()).view(B, E, Tq)
        # dk[b][e][n] = sum_t q[b][e][t] * dS[b][t][n]  -> needs dS^T
        dST = transpose(dS)
        dk = wgrad_gemm(q, dST, B=B, M=E, Cin=Tk, T=Tq, Tin=Tq, n_slabs=B, split_bf16=_attn_split()).view(B, E, Tk)
        return dq, dk, dv, None, None, None


def attention_core(q, k, v, key_len=None, last_attended=None, p=0.0, training=False, win_back=1,
                   win_ahead=3, site=None):
    return AttnCoreFn.apply(q, k, v, key_len, last_attended, (p, training, win_back, win_ahead, site))


# ----------------------------------------------------------------------------------------------
# embedding gather (+dropout) straight into BCT, dense deterministic gradient
# ----------------------------------------------------------------------------------------------
class EmbeddingBCTFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, idx, weight, p, training, padding_idx, site):
        _chk(weight, "embedding weight")
        idx = _c(idx.long())
        B, T = idx.shape
        n_vocab, C = weight.shape
        out = torch.empty((B, C, T), dtype=torch.float32, device=weight.device)
        bits, rs, dscale = None, 0, 1.0
        if training and p > 0:
            bits, rs = dropout_bits(B * C, T, p, weight.device, site)
            dscale = 1.0 / (1.0 - p)
        _lib.call("dv3_embedding_bct_f32", idx.data_ptr(), _c(weight).data_ptr(), out.data_ptr(), _ptr(bits),
                  rs, dscale, B, T, C, n_vocab, _stream())
        ctx.save_for_backward(idx)
        ctx.info = (bits, rs, dscale, B, T, C, n_vocab, -1 if padding_idx is None else padding_idx)
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        bits, rs, dscale, B, T, C, n_vocab, pad = ctx.info
        dout = _c(dout)
        dw = torch.empty((n_vocab, C), dtype=torch.float32, device=dout.device)
        _lib.call("dv3_embedding_bct_bwd_f32", idx.data_ptr(), dout.data_ptr(), dw.data_ptr(), _ptr(bits),
                  rs, dscale, B, T, C, n_vocab, pad, _stream())
        return None, dw, None, None, None, None


def embedding_bct(idx, weight, p=0.0, training=False, padding_idx=None, site=None):
    return EmbeddingBCTFn.apply(idx, weight, p, training, padding_idx, site)


class DropoutFn(torch.autograd.Function):
    """Standalone dropout on a (..., T)-last tensor viewed as [rows][T] (shared-mask sites)."""

    @staticmethod
    def forward(ctx, x, p, site):
        x = _c(_chk(x, "x"))
        T = x.shape[-1]
        rows = x.numel() // T
        bits, rs = dropout_bits(rows, T, p, x.device, site)
        scale = 1.0 / (1.0 - p)
        out = torch.empty_like(x)
        _lib.call("dv3_dropout_apply_f32", x.data_ptr(), bits.data_ptr(), rs, scale, out.data_ptr(), rows, T,
                  _stream())
        ctx.info = (bits, rs, scale, rows, T)
        return out

    @staticmethod
    def backward(ctx, dy):
        bits, rs, scale, rows, T = ctx.info
        dy = _c(dy)
        dx = torch.empty_like(dy)
        _lib.call("dv3_dropout_apply_f32", dy.data_ptr(), bits.data_ptr(), rs, scale, dx.data_ptr(), rows, T,
                  _stream())
        return dx, None, None


def dropout(x, p, training, site=None):
    if not training or p <= 0:
        return x
    return DropoutFn.apply(x, p, site)


_const_cache = {}


def _const1(value, device):
    """a cached one-element fp32 device tensor holding `value` (a position rate: torch.full would launch a fill kernel
    per call; during a hipGraph capture nothing is cached -- a tensor made inside a capture lives in the graph's pool)"""
    if torch.cuda.is_current_stream_capturing():
        return torch.full((1,), float(value), dtype=torch.float32, device=device)
    key = (float(value), str(device))
    t = _const_cache.get(key)
    if t is None:
        if len(_const_cache) > 64:
            _const_cache.clear()
        t = _const_cache[key] = torch.full((1,), float(value), dtype=torch.float32, device=device)
    return t


def sincos_pos_bct(pos, table, w=None, base=None, apply_sincos=True):
    """SinusoidalEncoding.forward (modules.py:45-64) -> BCT; w: None | float | tensor[B]; the
    (frozen) table gets no gradient; `base` is added (gradient passes straight through)."""
    pos = _c(pos.long())
    B, T = pos.shape
    n_pos, C = table.shape
    wt, per_batch = None, 0
    if w is not None:
        if torch.is_tensor(w):
            wt = _c(w.detach().float().view(-1))
            per_batch = 1 if wt.numel() > 1 else 0
        else:
            wt = _const1(w, table.device)
    out = torch.empty((B, C, T), dtype=torch.float32, device=table.device)
    _lib.call("dv3_sincos_pos_bct_f32", pos.data_ptr(), _c(table).data_ptr(), _ptr(wt), per_batch,
              _ptr(_c(base.detach()) if base is not None else None), out.data_ptr(), B, T, C, n_pos,
              int(apply_sincos), _stream())
    return out


class _PosEncFn(torch.autograd.Function):
    """out = base + PE(pos; table, w).  Gradient flows to `base` (identity), to the rate `w` when it is a
    tensor (multi-speaker: deepvoice3.py:304-315) and to the table when it is trainable
    (trainable_positional_encodings=True; frozen in every preset)."""

    @staticmethod
    def forward(ctx, base, pos, table, w, apply_sincos):
        out = sincos_pos_bct(pos, table, w, base, apply_sincos)
        ctx.has_base = base is not None
        ctx.table_grad = None
        if table.requires_grad:
            wt, per_batch = None, 0
            if w is not None:
                if torch.is_tensor(w):
                    wt = _c(w.detach().float().view(-1))
                    per_batch = 1 if wt.numel() > 1 else 0
                else:
                    wt = _const1(w, table.device)
            ctx.table_grad = (_c(pos.long()), table.detach(), wt, per_batch, int(apply_sincos))
        if torch.is_tensor(w) and w.requires_grad:
            if not apply_sincos:
                raise RuntimeError("a learnable rate needs the raw-angle table")
            ctx.save_for_backward(pos, table, w)
            ctx.w_shape = w.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        dw = dtable = None
        if ctx.table_grad is not None and ctx.needs_input_grad[2]:
            pos, table, wt, per_batch, apply = ctx.table_grad
            dout_c = _c(dout)
            B, C, T = dout_c.shape
            dtable = torch.empty_like(table)
            _lib.call("dv3_sincos_pos_table_bwd_f32", pos.data_ptr(), _c(table).data_ptr(), _ptr(wt), per_batch,
                      dout_c.data_ptr(), dtable.data_ptr(), B, T, C, table.shape[0], apply, _stream())
        if ctx.needs_input_grad[3]:
            pos, table, w = ctx.saved_tensors
            dout_c = _c(dout)
            B, C, T = dout_c.shape
            wt = _c(w.detach().float().view(-1))
            per_batch = 1 if wt.numel() > 1 else 0
            dwb = torch.empty(B, dtype=torch.float32, device=dout.device)
            # two deterministic stages: enough workgroups for the chip (the one-workgroup-per-item form took 160 us per call
            # in the deepvoice3_vctk step: 64 workgroups of 200 dependent sinf / cosf iterations)
            nch = max(1, min(32, (C * T) // 4096))
            part = torch.empty((B, nch), dtype=torch.float32, device=dout.device)
            _lib.call("dv3_sincos_pos_bwd2_f32", _c(pos.long()).data_ptr(), _c(table).data_ptr(), wt.data_ptr(),
                      per_batch, dout_c.data_ptr(), part.data_ptr(), nch, dwb.data_ptr(), B, T, C, table.shape[0], _stream())
            dw = (dwb if per_batch else dwb.sum(0, keepdim=True)).view(ctx.w_shape)
        return (dout if ctx.has_base else None), None, dtable, dw, None


def add_position_encoding(base, pos, table, w=None, apply_sincos=True):
    return _PosEncFn.apply(base, pos, table, w, apply_sincos)


def position_encoding(pos, table, w=None, apply_sincos=True):
    return _PosEncFn.apply(None, pos, table, w, apply_sincos)


# ----------------------------------------------------------------------------------------------
# valid-length steps: a batch padded beyond its own maxima (to a lattice shape whose captured step is replayed)
# ----------------------------------------------------------------------------------------------
class ValidLengths(object):
    """What a step needs to compute, on a batch padded to (t_in, t_dec) -- text positions / decoder steps --, exactly what
    the reference computes on the same batch padded to its OWN maxima (train.collate_fn, train.py:293-360): those maxima
    as DEVICE scalars (a replayed hipGraph has no host in the loop) plus host-side upper bounds for the surplus.

      tv        int32[4] on the device: {longest text, most decoder steps, x r (mel frames), x r x downsample (linear frames)}
                (tv, scale and key_valid are views of one buffer `buf`)
      scale     float32[1]: s * sqrt(1 / s) for s = longest text -- the context scale of AttentionLayer (deepvoice3.py:170-171)
      key_valid int32[B], every entry the longest text: the softmax's key limit when the model uses no memory mask
      t_in, t_dec / tail_in, tail_dec   (host) the padded sizes and the most surplus columns either axis can have

    Where the surplus would change a result: the non-causal stacks (Encoder, Converter: zero_tail after every layer, and
    on every activation gradient), the attention softmax and its sqrt(keys) scale, and every loss mean.  The causal
    decoder stack needs nothing: its valid frames never read a later one, and the losses feed its surplus frames zeros."""

    def __init__(self, buf, t_in, t_dec, tail_in, tail_dec, r, downsample_step):
        """buf: int32[5 + B] = {tv[4], the bits of the fp32 scale, key_valid[B]} -- ONE buffer, so that a batch's maxima reach
        the device (and a captured step's static copy) in one copy"""
        self.buf = buf
        self.tv, self.scale, self.key_valid = buf[0:4], buf[4:5].view(torch.float32), buf[5:]
        self.t_in, self.t_dec, self.tail_in, self.tail_dec = int(t_in), int(t_dec), int(tail_in), int(tail_dec)
        self.r, self.downsample_step = int(r), int(downsample_step)

    @staticmethod
    def make(max_in, max_dec, B, t_in, t_dec, tail_in, tail_dec, r, downsample_step, device):
        max_in, max_dec = int(max_in), int(max_dec)
        if not (0 < max_in <= t_in and 0 < max_dec <= t_dec and t_in - max_in <= tail_in and t_dec - max_dec <= tail_dec):
            raise ValueError("ValidLengths: maxima (%d, %d) do not fit the padded shape (%d, %d) with tails (%d, %d)" % (
                max_in, max_dec, t_in, t_dec, tail_in, tail_dec))
        device = torch.device(device)
        host = torch.empty(5 + B, dtype=torch.int32)
        if device.type == "cuda":
            host = host.pin_memory()       # an asynchronous copy: a pageable one would make the host wait for the stream
        host[0], host[1], host[2], host[3] = max_in, max_dec, max_dec * r, max_dec * r * downsample_step
        host[4:5].view(torch.float32)[0] = max_in * math.sqrt(1.0 / max_in)
        host[5:] = max_in
        v = ValidLengths(host.to(device, non_blocking=True), t_in, t_dec, tail_in, tail_dec, r, downsample_step)
        v._host = host                     # alive until the copy has run
        return v

    def clone(self):
        return ValidLengths(self.buf.clone(), self.t_in, self.t_dec, self.tail_in, self.tail_dec, self.r, self.downsample_step)

    # (pointer tensor, host upper bound of the surplus) per time axis
    def text(self):
        return self.tv[0:1], self.tail_in

    def dec(self):
        return self.tv[1:2], self.tail_dec

    def mel(self):
        return self.tv[2:3], self.tail_dec * self.r

    def linear(self):
        return self.tv[3:4], self.tail_dec * self.r * self.downsample_step

    def axis_for(self, T):
        """the (pointer, tail, mult) of an activation with T columns in the converter: T is the decoder's axis or the
        mel axis times a power of two (the ConvTranspose1d layers double it)"""
        for ptr, tail, base in ((self.tv[1:2], self.tail_dec, self.t_dec), (self.tv[2:3], self.tail_dec * self.r, self.t_dec * self.r)):
            if T % base == 0 and (T // base) & (T // base - 1) == 0:
                return ptr, tail * (T // base), T // base
        raise RuntimeError("ValidLengths: no time axis of the padded batch has %d columns" % T)


valid = None      # a ValidLengths, set by the trainer for the duration of a step on a batch padded beyond its maxima


def _zero_tail_raw(x, ptr, mult, max_tail):
    """columns >= ptr[0] * mult of x ((B, C, T) fp32 or the channel-blocked bf16 [B][C8][T][8]) to zero, behind
    autograd's back (no version bump: layers save their outputs)"""
    if is_c8(x):
        rows, T, words = x.shape[0] * x.shape[1], x.shape[2], 4
    else:
        _chk(x, "x")
        T, words = x.shape[-1], 1
        rows = x.numel() // T
    if not x.is_contiguous():
        raise RuntimeError("zero_tail: contiguous tensors only")
    _lib.call("dv3_zero_tail_b32", x.data_ptr(), rows, T, words, ptr.data_ptr(), int(mult), int(max_tail), _stream())


class _ZeroTailFn(torch.autograd.Function):
    """forward: the surplus columns of x to zero, in place; backward: the same on the incoming gradient.  The result is a
    view of x -- not x marked dirty -- because the producing layer may have saved x for its own backward (ReLU layers
    do) and the zeroed columns are ones no valid output depends on."""

    @staticmethod
    def forward(ctx, x, ptr, mult, max_tail):
        ctx.args = (ptr, mult, max_tail)
        _zero_tail_raw(x, ptr, mult, max_tail)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, dy):
        ptr, mult, max_tail = ctx.args
        dy = dy if dy.is_contiguous() else dy.contiguous()
        _zero_tail_raw(dy, ptr, mult, max_tail)
        return dy, None, None, None


def zero_tail(x, ptr, max_tail, mult=1):
    """see ValidLengths; ptr: device int32[1] = valid columns at mult 1"""
    if max_tail <= 0:
        return x
    out = _ZeroTailFn.apply(x, ptr, mult, max_tail)
    c = getattr(x, "_dv3_C", None)
    if c is not None:
        out._dv3_C = c          # a channel-blocked tensor remembers its channel count
    return out


# ----------------------------------------------------------------------------------------------
# losses (fused value + gradient; train.py:537-601,704-740)
# ----------------------------------------------------------------------------------------------
def _dense_or_copy(t):
    """Keep a (B,T,D)-shaped tensor whose memory is dense in either BTC or BCT order."""
    B, T, D = t.shape
    st = t.stride()
    if st == (T * D, D, 1) or st == (D * T, 1, T):
        return t
    return t.contiguous()


class SpecLossFn(torch.autograd.Function):
    """-> tensor[4] = {l1_loss, binary_div, (1-w_bd)*l1 + w_bd*bd, mask_sum}; only [2] carries
    gradient.  y_hat, y: logical (B,T,D) (memory BTC or a transposed view of BCT); compares
    y_hat[:, :-r] with y[:, r:]."""

    @staticmethod
    def forward(ctx, y_hat, y, lengths, r, w_masked, w_bd, t_valid=None):
        y_hat, y = _dense_or_copy(_chk(y_hat, "y_hat")), _dense_or_copy(_chk(y, "y"))
        B, T, D = y_hat.shape
        dev = y_hat.device
        out4 = torch.empty(4, dtype=torch.float32, device=dev)
        n_scr = _lib.lib().dv3_spec_loss_scratch_floats(B, T, D)
        scratch = torch.empty(n_scr, dtype=torch.float32, device=dev)
        dyh = torch.empty_strided(y_hat.shape, y_hat.stride(), dtype=torch.float32, device=dev) \
            if ctx.needs_input_grad[0] else None
        d = _spec_loss_desc()
        d.y_hat, d.y, d.lengths = y_hat.data_ptr(), y.data_ptr(), _ptr(lengths)
        d.yh_bs, d.yh_ts, d.yh_ds = y_hat.stride()
        d.y_bs, d.y_ts, d.y_ds = y.stride()
        d.dyh, d.out4, d.scratch = _ptr(dyh), out4.data_ptr(), scratch.data_ptr()
        d.B, d.T, d.D, d.r = B, T, D, r
        d.w_masked, d.w_bd, d.gscale = w_masked, w_bd, 1.0
        d.t_valid = _ptr(t_valid)
        _lib.call("dv3_spec_loss_f32", ctypes.byref(d), _stream())
        ctx.dyh = dyh
        return out4

    @staticmethod
    def backward(ctx, dout):
        g = ctx.dyh * dout[2] if ctx.dyh is not None else None
        return g, None, None, None, None, None, None


def spec_loss(y_hat, y, lengths, r=1, w_masked=0.5, w_bd=0.1, t_valid=None):
    """t_valid: device int32[1] -- only the first t_valid[0] frames take part (ValidLengths)"""
    return SpecLossFn.apply(y_hat, y, lengths, r, w_masked, w_bd, t_valid)


class _NoCtx(object):
    """stand-in for the autograd context when a loss Function's forward is called directly: the fused kernels write
    value AND gradient in one pass, so the trainer takes the gradient tensor and feeds it to autograd.backward itself --
    no select / mul / fill / add nodes between the loss terms and the model outputs (train_step.Trainer)"""
    needs_input_grad = (True,)


def spec_loss_with_grad(y_hat, y, lengths, r=1, w_masked=0.5, w_bd=0.1, t_valid=None):
    """-> (tensor[4] as spec_loss, d total / d y_hat laid out like y_hat); no autograd graph"""
    c = _NoCtx()
    out4 = SpecLossFn.forward(c, y_hat.detach(), y, lengths, r, w_masked, w_bd, t_valid)
    return out4, c.dyh


def guided_attention_loss_with_grad(attn, in_len, out_len, g=0.2, tq_valid=None, tk_valid=None):
    c = _NoCtx()
    out1 = GuidedAttnLossFn.forward(c, attn.detach(), in_len, out_len, g, tq_valid, tk_valid)
    return out1, c.dattn


def bce_loss_with_grad(p, t, t_valid=None):
    c = _NoCtx()
    out1 = BCELossFn.forward(c, p.detach(), t, t_valid)
    return out1, c.dp


def sum_scalars(a, b, c=None, d=None):
    """a[0] + b[0] (+ c[0]) (+ d[0]) -> tensor[1], one tiny launch"""
    out = torch.empty(1, dtype=torch.float32, device=a.device)
    _lib.call("dv3_sum_scalars_f32", a.data_ptr(), b.data_ptr(), _ptr(c), _ptr(d), out.data_ptr(), _stream())
    return out


def scaled_copy(a, alpha=1.0):
    """alpha * a as a new tensor (dv3_axpby_f32): device scalars handed to the caller without a torch kernel"""
    out = torch.empty_like(a)
    _lib.call("dv3_axpby_f32", a.data_ptr(), None, out.data_ptr(), a.numel(), float(alpha), _stream())
    return out


class GuidedAttnLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, attn, in_len, out_len, g, tq_valid=None, tk_valid=None):
        attn = _c(_chk(attn, "attn"))
        L, B, Tq, Tk = attn.shape
        dev = attn.device
        out1 = torch.empty(1, dtype=torch.float32, device=dev)
        scratch = torch.empty(4 * 1024 + 16, dtype=torch.float32, device=dev)
        dattn = torch.empty_like(attn) if ctx.needs_input_grad[0] else None
        if tq_valid is not None:      # the mean over the batch's own (decoder steps, text) maxima (ValidLengths)
            _lib.call("dv3_guided_attn_loss_valid_f32", attn.data_ptr(), in_len.data_ptr(), out_len.data_ptr(),
                      _ptr(dattn), out1.data_ptr(), scratch.data_ptr(), L, B, Tq, Tk, float(g), 1.0,
                      tq_valid.data_ptr(), tk_valid.data_ptr(), _stream())
        else:
            _lib.call("dv3_guided_attn_loss_f32", attn.data_ptr(), in_len.data_ptr(), out_len.data_ptr(),
                      _ptr(dattn), out1.data_ptr(), scratch.data_ptr(), L, B, Tq, Tk, float(g), 1.0, _stream())
        ctx.dattn = dattn
        return out1

    @staticmethod
    def backward(ctx, dout):
        return (ctx.dattn * dout if ctx.dattn is not None else None), None, None, None, None, None


def guided_attention_loss(attn, in_len, out_len, g=0.2, tq_valid=None, tk_valid=None):
    return GuidedAttnLossFn.apply(attn, in_len, out_len, g, tq_valid, tk_valid)


class BCELossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, t, t_valid=None):
        p, t = _c(_chk(p, "p")), _c(_chk(t, "t"))
        dev = p.device
        out1 = torch.empty(1, dtype=torch.float32, device=dev)
        scratch = torch.empty(4 * 1024 + 16, dtype=torch.float32, device=dev)
        dp = torch.empty_like(p) if ctx.needs_input_grad[0] else None
        if t_valid is not None:       # p, t (B, T, 1): the first t_valid[0] steps of every item (ValidLengths)
            if p.dim() != 3 or p.shape[2] != 1:
                raise RuntimeError("bce_loss(t_valid=): (B, T, 1) tensors")
            _lib.call("dv3_bce_loss_valid_f32", p.data_ptr(), t.data_ptr(), _ptr(dp), out1.data_ptr(),
                      scratch.data_ptr(), p.shape[0], p.shape[1], t_valid.data_ptr(), 1.0, _stream())
        else:
            _lib.call("dv3_bce_loss_f32", p.data_ptr(), t.data_ptr(), _ptr(dp), out1.data_ptr(),
                      scratch.data_ptr(), p.numel(), 1.0, _stream())
        ctx.dp = dp
        return out1

    @staticmethod
    def backward(ctx, dout):
        return (ctx.dp * dout if ctx.dp is not None else None), None, None


def bce_loss(p, t, t_valid=None):
    return BCELossFn.apply(p, t, t_valid)


# ----------------------------------------------------------------------------------------------
# optimiser tail on flat arenas
# ----------------------------------------------------------------------------------------------
def grad_sqnorm(flat_grad, partial, out2):
    _lib.call("dv3_grad_sqnorm_f32", flat_grad.data_ptr(), flat_grad.numel(), partial.data_ptr(),
              partial.numel(), out2.data_ptr(), _stream())


# Raw-pointer writes (dv3_clip_adam_f32 on the flat arena, a replayed step graph) do not bump
# tensor._version, so everything that caches a function of the parameters (conv._WNLayer.packed: the
# eval-mode packed weights, hence incremental decode and the decode-step graph) also keys on this
# counter; whoever writes parameters behind autograd's back bumps it.
param_epoch = 0


def bump_param_epoch():
    global param_epoch
    param_epoch += 1


def clip_adam(p, g, m, v, grad_norm, clip, hyper, beta1, beta2, eps, weight_decay=0.0, grad_prescale=1.0):
    bump_param_epoch()
    _lib.call("dv3_clip_adam_f32", p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(),
              _ptr(grad_norm), float(clip), hyper.data_ptr(), float(beta1), float(beta2), float(eps),
              float(weight_decay), float(grad_prescale), _stream())


def conv_step_fits(k, cin):
    """does dv3_conv_step_f32 (the fused decode step, csrc/decode_step.hip) take a k-tap layer with `cin` input
    channels?  Asked of the library itself (dv3_conv_step_lds_bytes), so the Python predicate can not drift from the
    kernel's LDS layout."""
    return _lib.lib().dv3_conv_step_lds_bytes(int(k), int(cin)) <= CONSTS["DV3_CONV_STEP_LDS_MAX"]


def ragged_pad_rows(src, row_off, B, T_out, lead=0, t_stride=1):
    """Items packed back to back as rows (`src` [rows] or [rows][D], f32 or int64; item b = rows
    row_off[b]..row_off[b+1], int32 on the device) -> zero-padded [B][T_out][D] ([B][T_out] for 1-D
    input): out[b][t] = row t*t_stride - lead of item b, or 0 outside it.  The `_pad` / `_pad_2d` loops of
    the reference's collate_fn (train.py:293-360) and its mel time down-sampling (train.py:639-640)."""
    _chk(src, "src", src.dtype)
    _chk(row_off, "row_off", torch.int32)
    if src.dtype not in (torch.float32, torch.int64, torch.int32):
        raise RuntimeError("ragged_pad_rows: f32 / int32 / int64 rows only")
    src = src.contiguous()
    D = 1 if src.dim() == 1 else int(src.shape[1])
    words = D * (src.element_size() // 4)
    out = torch.empty((B, T_out) if src.dim() == 1 else (B, T_out, D), dtype=src.dtype, device=src.device)
    _lib.call("dv3_ragged_pad_rows_b32", src.data_ptr(), row_off.data_ptr(), out.data_ptr(), B, T_out, words,
              lead, t_stride, _stream())
    return out
