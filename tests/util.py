# coding: utf-8
"""Shared helpers for the tests (fixture loading, tolerances)."""
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
MODEL_FIXTURES = ["dv3_tiny", "dv3_preset_like", "dv3_multispeaker", "nyanko_tiny"]


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def split_model_fixture(fx):
    hp = json.loads(str(fx["hp"]))
    builder = str(fx["builder"])
    sd = {k[3:]: torch.from_numpy(v) for k, v in fx.items() if k.startswith("sd/")}
    inputs = {k[3:]: torch.from_numpy(v) for k, v in fx.items() if k.startswith("in/")}
    return builder, hp, sd, inputs


def rel_err(a, b):
    """max |a-b| / max(|b|) -- the 'rel fp32' measure BASELINE.json's north_star names."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
