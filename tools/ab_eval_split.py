"""Inference forward at small batches (the reference samples at 16 / GPU: scripts_psld/sota/uncond/cifar10/sample_uncond_psld.sh) with
and without the split-chunk Winograd launches: EM steps per second at B = 4 / 16 / 64 / 128, interleaved in one process.
    python tools/ab_eval_split.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psld_amd  # noqa: E402
from psld_amd import config as C, ops, score_fn as S  # noqa: E402
from psld_amd.registry import get_module  # noqa: E402

psld_amd.import_modules_into_registry()
dev = torch.device("cuda")
cfg = C.c10_sota()
torch.manual_seed(0)
net = get_module("score_fn", "ncsnpp")(cfg).to(dev).eval()
sde = get_module("sde", "psld")(cfg)
sampler = get_module("samplers", "em_sde")(cfg, sde, net)
ts = torch.linspace(0, 0.999, 1000, device=dev, dtype=torch.float64)
real = S._Exec.wino_wanted


def no_split(self, c1, c2, b, h, w, cout):
    return ops.conv3x3_wino_wanted(c1, c2, b, h, w, cout, bool(self.record))


for batch in (4, 16, 64, 128):
    x = sde.prior_sampling((batch, 3, 32, 32), device=dev)
    steps = 30 if batch <= 16 else 10
    res = {}
    for rep in range(2):
        for name, fn in (("unsplit (round 5 rule for the inference forward)", no_split), ("split chunks", real)):
            S._Exec.wino_wanted = fn
            sampler.sample(x, ts[:3], 2, denoise=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sampler.sample(x, ts[: steps + 1], steps, denoise=False)
            torch.cuda.synchronize()
            res.setdefault(name, []).append((time.perf_counter() - t0) / steps)
    S._Exec.wino_wanted = real
    print(f"B={batch:4d}: " + "  |  ".join(f"{k}: {min(v) * 1e3:7.2f} ms/EM step ({batch / min(v):7.1f} evals/s)" for k, v in res.items()), flush=True)
