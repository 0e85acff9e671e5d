"""Import the *reference* (mandt-lab/PSLD, /root/reference) on CPU inside the build container.

Used ONLY by tools/gen_golden.py to pin the oracle and to
emit golden fixtures.  Nothing here (and nothing from /root/reference) ships to the GPU box;
tests, smoke() and bench.py never import this module.

Recipe: neutralise the nvcc JIT in op/upfirdn2d.py:10 and op/fused_act.py:11,
stub the packages the container lacks (torchvision, pytorch_lightning, torchdiffeq, PIL is
present), then import losses / models / samplers so that the registry fills.
"""
from __future__ import annotations

import os
import sys
import types

REF_ROOT = "/root/reference/main"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    """Returns the reference's ``util`` module (registry) after importing the hot-path plug-ins."""
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("reference not mounted; golden generation only runs in the build container")
    if "util" in sys.modules and getattr(sys.modules["util"], "_PSLD_REF", False):
        return sys.modules["util"]
    sys.dont_write_bytecode = True
    import torch
    import torch.nn as nn
    import torch.utils.cpp_extension as cpp_ext

    cpp_ext.load = lambda *a, **k: None  # op/upfirdn2d.py:10, op/fused_act.py:11

    _stub("torchvision")
    _stub("torchvision.transforms")
    _stub("torchvision.datasets")
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision"].datasets = sys.modules["torchvision.datasets"]

    class _LM(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
            self.global_rank = 0

        def log(self, *a, **k):
            pass

    class _CB:
        def __init__(self, *a, **k):
            pass

    pl = _stub("pytorch_lightning", LightningModule=_LM, Callback=_CB, Trainer=object)
    _stub("pytorch_lightning.utilities")
    _stub("pytorch_lightning.utilities.seed", seed_everything=lambda s, workers=False: torch.manual_seed(s))
    _stub("pytorch_lightning.callbacks", BasePredictionWriter=_CB, Callback=_CB, ModelCheckpoint=_CB)
    pl.callbacks = sys.modules["pytorch_lightning.callbacks"]
    _stub("torchdiffeq", odeint=None)

    sys.path.insert(0, REF_ROOT)
    import util  # noqa: E402

    util._PSLD_REF = True
    import losses  # noqa: F401,E402
    import models  # noqa: F401,E402
    import samplers  # noqa: F401,E402
    return util
