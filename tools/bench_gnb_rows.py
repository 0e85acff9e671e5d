"""Does the GroupNorm backward stream faster on whole contiguous rows?  The three-pass form (maps above 32x32) reads full NHWC
rows (1 KB per pixel at 256 channels); the one-pass kernels read 128-byte segments (32 channels) of every row.  Same bytes:
32 images of 64x64x256 vs 128 images of 32x32x256.  Run under rocprofv3 --kernel-trace --stats to see the per-kernel times."""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psld_amd import ops

for B, S, C in ((32, 64, 256), (128, 32, 256)):
    x = torch.randn(B, S, S, C, device="cuda")
    dy = torch.randn_like(x)
    dx = torch.zeros_like(x)
    other = torch.randn_like(x)
    gamma = torch.rand(C, device="cuda") + 0.5
    beta = torch.randn(C, device="cuda") * 0.1
    st = ops.gn_stats(x, gamma, beta)
    sums = torch.empty(B, 2, C, device="cuda")
    for kw in ({}, {"add": other, "add_scale": 0.7}, {"add": other, "add_scale": 0.7, "accumulate_dx": True}):
        for _ in range(10):
            ops.gn_bwd(dy, x, st, gamma, beta, True, dx, sums=sums, **kw)
torch.cuda.synchronize()
