"""GroupNorm backward at the step's shapes and variants: us per call of psld_gn_bwd_nhwc_f32 (dx + per-image sums; dgamma /
dbeta are a batched reduction elsewhere).
    python tools/bench_gnb.py            (PSLD_GN_BWD_PIPE=0: the one-slab kernel without the early third operand for every shape)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psld_amd import ops  # noqa: E402
from tools.bench_tile import timeit  # noqa: E402

ops.lib()
print(f"{'shape':22s} {'variant':28s} {'us':>8s} {'GB/s (algorithmic)':>20s}")
for B, S, C in ((128, 32, 256), (128, 16, 256), (128, 8, 256), (64, 32, 256), (16, 32, 256), (16, 16, 256)):
    x = torch.randn(B, S, S, C, device="cuda")
    dy = torch.randn_like(x)
    dx = torch.zeros_like(x)
    other = torch.randn_like(x)
    gamma = torch.rand(C, device="cuda") + 0.5
    beta = torch.randn(C, device="cuda") * 0.1
    st = ops.gn_stats(x, gamma, beta)
    sums = torch.empty(B, 2, C, device="cuda")
    for name, kw, nb in (("SiLU", {}, 12), ("SiLU + branch gradient", {"add": other, "add_scale": 0.7}, 16),
                         ("SiLU + dropout 0.15", {"drop_p": 0.15, "seed": 11}, 12),
                         ("SiLU, accumulating", {"accumulate_dx": True}, 16),
                         ("SiLU + branch, accumulating", {"add": other, "add_scale": 0.7, "accumulate_dx": True}, 20),
                         ("no activation", {"act": False}, 12)):
        kw = dict(kw)
        act = kw.pop("act", True)
        t = timeit(lambda: ops.gn_bwd(dy, x, st, gamma, beta, act, dx, sums=sums, **kw), 30)
        print(f"{B}x{S}x{S}x{C:<10d} {name:28s} {t * 1e6:8.1f} {nb * x.numel() / t / 1e9:20.0f}")
