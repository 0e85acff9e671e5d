"""Sweep of the split-K factor of the conv weight-gradient kernel (blocks = tiles x taps x nsplit)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psld_amd import ops
from tools.bench_tile import timeit
ops.lib()
B = 128
for (cin, cout, s) in [(256, 256, 32), (512, 256, 32), (256, 256, 16), (256, 256, 8)]:
    x = torch.randn(B, s, s, cin, device="cuda"); dy = torch.randn(B, s, s, cout, device="cuda")
    fl = 2.0 * B * s * s * cout * 9 * cin
    tiles = 2 * (cin // 128) * 9
    out = []
    for ns in (7, 10, 14, 21, 28, 42):
        if B * s * s // ns < 256:
            continue
        slabs = torch.empty(ns, cout, 9, cin, device="cuda")
        t = timeit(lambda: ops.conv2d_wgrad_nhwc(dy, cout, x, 3, 3, 1, 1, s, s, slabs, cin, 0, ns), 10)
        out.append(f"ns={ns:2d} ({tiles*ns:4d} blk) {fl/t/1e12:6.1f} TF")
    print(f"wgrad {cin}->{cout} @{s}: " + " | ".join(out))
