# coding: utf-8
"""Reproduce the in-suite gradient mismatch of the nyanko preset train step: the same test body with a device seed
offset (what a GraphedTrainer leaves in ops.dropout_state), across toggles."""
import os, sys, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepvoice3_pytorch_amd import ops, _lib
import tests.test_gpu_preset_scale as TS

dev = torch.device("cuda:0")


def run(label, mode, offset, fused=True, wtile=0, os_env=None):
    prev = ops.set_gemm_precision(mode)
    ops.fused_attention = fused
    _lib.call("dv3_debug_set", 2, wtile)
    if os_env is not None:
        os.environ["DV3_WGRAD_TILE"] = os_env
    ops.dropout_state.dev_offset = None if offset is None else torch.tensor([offset], dtype=torch.int64, device=dev)
    try:
        TS.test_preset_train_step_matches_oracle(dev, "nyanko_ljspeech", mode)
        print("== %-40s PASS" % label, flush=True)
    except AssertionError as e:
        print("== %-40s FAIL %s" % (label, str(e)[:400].replace("\n", " ")), flush=True)
    finally:
        ops.set_gemm_precision(prev)
        ops.fused_attention = True
        _lib.call("dv3_debug_set", 2, 0)
        os.environ.pop("DV3_WGRAD_TILE", None)
        ops.dropout_state.dev_offset = None


run("f16x3 offset None", "f16x3", None)
run("f16x3 offset 0", "f16x3", 0)
run("f16x3 offset 3", "f16x3", 3)
run("f32 offset 3", "f32", 3)
run("f16x3 offset 3 unfused attention", "f16x3", 3, fused=False)
run("f16x3 offset 3 per-tap wgrad", "f16x3", 3, wtile=1, os_env="1")
run("f16x3 offset 5", "f16x3", 5)
