# coding: utf-8
"""torch.ops.dv3hip.* -- the C ABI of include/dv3hip.h registered as PyTorch-ROCm custom operators
(TORCH_LIBRARY(dv3hip), csrc/torch_ops.cpp -> libdv3hip_torch.so).  A thin shim over libdv3hip.so: the header
stays the contract, this is the operator form BASELINE.json's north_star names.

    from deepvoice3_pytorch_amd import torch_ops
    ops = torch_ops.load()                       # == torch.ops.dv3hip
    fs, bs, scale = ops.weight_norm_split_pack(conv.weight_v, conv.weight_g, C, True)
    y = ops.conv1d_glu(x, fs, True, conv.bias, 3, 1, False, True, False)
"""
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.path.join(_HERE, "libdv3hip_torch.so")
_loaded = False


def load():
    """Load libdv3hip_torch.so (once) and return torch.ops.dv3hip; raises when it is missing or stale."""
    global _loaded
    if not _loaded:
        if not os.path.exists(_LIBPATH):
            raise RuntimeError("libdv3hip_torch.so not found at %s -- build it with "
                               "`python -c 'import __graft_entry__ as g; g.build()'`" % _LIBPATH)
        from . import _lib
        _lib.lib()                       # libdv3hip.so first (same HIP runtime as torch, ABI check)
        torch.ops.load_library(_LIBPATH)
        if torch.ops.dv3hip.abi_version() != _lib.CONSTS["DV3_ABI_VERSION"]:
            raise RuntimeError("libdv3hip_torch.so was built against another ABI version: rebuild")
        _loaded = True
    return torch.ops.dv3hip
