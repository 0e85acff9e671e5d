"""GroupNorm backward: one-slab / resident-slab kernels vs the whole-row team kernel (psld_gn_bwd_team_f32), us per call."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psld_amd import ops  # noqa: E402
from tools.bench_tile import timeit  # noqa: E402

ops.lib()
print(f"{'shape':20s} {'variant':30s} {'slab us':>9s} {'TB/s':>6s} {'team us':>9s} {'TB/s':>6s}")
for B, S, C in ((128, 32, 256), (128, 64, 128), (64, 64, 128), (128, 16, 256), (128, 32, 128), (64, 32, 256), (16, 32, 256), (16, 16, 256)):
    x = torch.randn(B, S, S, C, device="cuda")
    dy = torch.randn_like(x)
    dx = torch.zeros_like(x)
    other = torch.randn_like(x)
    gamma = torch.rand(C, device="cuda") + 0.5
    beta = torch.randn(C, device="cuda") * 0.1
    st = ops.gn_stats(x, gamma, beta)
    k = ops.gn_bwd_team_rows(B, S * S, C)
    sums = torch.empty(B * max(k, 1), 2, C, device="cuda")
    rows = torch.empty(B * max(k, 1), C, device="cuda")
    for name, kw, nb in (("SiLU", {}, 12), ("SiLU + dropout 0.15", {"drop_p": 0.15, "seed": 11}, 12),
                         ("SiLU + branch gradient", {"add": other, "add_scale": 0.7}, 16),
                         ("SiLU, accumulating", {"accumulate_dx": True}, 16),
                         ("SiLU + branch, accumulating", {"add": other, "add_scale": 0.7, "accumulate_dx": True}, 20)):
        t0 = timeit(lambda: ops.gn_bwd(dy, x, st, gamma, beta, True, dx, sums=sums, **kw), 30)
        t1 = timeit(lambda: ops.gn_bwd_team(dy, x, st, gamma, beta, True, dx, sums=sums, colsum_rows=rows, **kw), 30) if k else float("nan")
        by = nb * x.numel()
        print(f"{B}x{S}x{S}x{C:<8d} {name:30s} {t0 * 1e6:9.1f} {by / t0 / 1e12:6.2f} {t1 * 1e6:9.1f} {by / t1 / 1e12:6.2f}")
print("team errors:", ops.gn_team_errors(torch.device("cuda")))
