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
    """max |a-b| / max(|b|) -- the 'rel fp32' measure BASELINE.json's north_star names.

    TENSOR-MAX-normalised, not element-relative: every element is held to `tol` of the largest
    magnitude of the reference tensor.  For the sigmoid outputs in [0, 1] that north_star's "1e-4 rel" is stated on this
    is the sensible reading; for tensors with a wide dynamic range (attention probabilities of ~1e-3) an element may be
    off by a larger fraction of its own value and still pass -- tests that need an element-wise bound state one."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def synth_state_dict(shapes, stats, seed, keep=None):
    """Deterministic weights for the preset-size goldens: tensor i (keys in sorted order) =
    mean_i + std_i * RandomState(seed + i).standard_normal(shape), float32 -- numpy's legacy generator is
    stable across versions, so oracle/make_golden.py (reference side) and the GPU test (HIP side) rebuild
    identical bits from the few statistics stored in the golden file.  Keys absent from `stats` (the frozen
    position tables, deterministic functions of the hyper-parameters) are taken from `keep`."""
    out = {}
    for i, k in enumerate(sorted(shapes)):
        if k not in stats:
            out[k] = keep[k].detach().clone()
            continue
        mean, std = stats[k]
        rs = np.random.RandomState(seed + i)
        out[k] = torch.from_numpy((rs.standard_normal(tuple(shapes[k])) * std + mean).astype(np.float32))
    return out
