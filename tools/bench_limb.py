"""A/B of the 3x3 limb convolution with fp32 input (split in the kernel, psld_conv3x3_split_f32) against the same
convolution on pre-split limb planes (LDS-DMA staging, psld_conv3x3_limb_f32), interleaved rounds in ONE process
(cdna_hip_programming.md 5.4 rule 24), random data.   python tools/bench_limb.py [--rounds 5] [--iters 10]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psld_amd import ops  # noqa: E402

DEV = "cuda"


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--batch", type=int, default=128)
    args = ap.parse_args()
    B = args.batch
    ops.lib()
    for (cin, cout, s) in [(256, 256, 32), (512, 256, 32), (256, 256, 16), (512, 256, 16), (256, 256, 8), (512, 256, 8)]:
        x = torch.randn(B, s, s, cin, device=DEV)
        w = torch.randn(cout, cin, 3, 3, device=DEV) * 0.05
        wf = ops.conv3x3_frag(w, False)
        bias = torch.randn(cout, device=DEV)
        res = torch.randn(B, s, s, cout, device=DEV)
        epi = ops.epilogue(bias=bias, residual=res, ld_residual=cout, out_scale=0.7)
        y0, y1 = torch.empty(B, s, s, cout, device=DEV), torch.empty(B, s, s, cout, device=DEV)
        xl = ops.f32_to_limb(x)
        fl = 2.0 * B * s * s * cout * 9 * cin
        f0 = lambda: ops.conv3x3_split(x, None, wf, cout, y0, epi)
        f1 = lambda: ops.conv3x3_split(xl, None, wf, cout, y1, epi)
        t0s, t1s = [], []
        for _ in range(args.rounds):
            t0s.append(timeit(f0, args.iters))
            t1s.append(timeit(f1, args.iters))
        same = bool(torch.equal(y0, y1))
        m0, m1 = sorted(t0s)[len(t0s) // 2], sorted(t1s)[len(t1s) // 2]
        print(f"conv fwd {cin}->{cout} @{s} B={B}: fp32-in {fl / m0 / 1e12:6.1f} TF (best {fl / min(t0s) / 1e12:6.1f})   "
              f"limb-in {fl / m1 / 1e12:6.1f} TF (best {fl / min(t1s) / 1e12:6.1f})   x{m0 / m1:.3f}   bitwise {same}")


def wgrad():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--wgrad", action="store_true")
    args = ap.parse_args()
    B = args.batch
    ops.lib()
    for (cin, cout, s) in [(256, 256, 32), (512, 256, 32), (256, 256, 16), (512, 256, 16), (256, 256, 8), (512, 256, 8)]:
        x = torch.randn(B, s, s, cin, device=DEV)
        dy = torch.randn(B, s, s, cout, device=DEV)
        fl = 2.0 * B * s * s * cout * 9 * cin
        kt = B * s * s // 32
        tiles = (cout // 128) * (cin // 64)
        ns = max(1, min(512 // (3 * tiles), kt // 4))
        per = -(-kt // ns)
        ns = -(-kt // per)
        s0, s1 = torch.empty(ns, cout, 9, cin, device=DEV), torch.empty(ns, cout, 9, cin, device=DEV)
        xl = ops.f32_to_limb(x)
        f0 = lambda: ops.conv3x3_wgrad_split(dy, cout, x, s0, cin, 0, ns)
        f1 = lambda: ops.conv3x3_wgrad_split(dy, cout, xl, s1, cin, 0, ns)
        ts = [[], []]
        for _ in range(args.rounds):
            for i, f in enumerate((f0, f1)):
                ts[i].append(timeit(f, args.iters))
        m = [sorted(t)[len(t) // 2] for t in ts]
        print(f"wgrad {cin}->{cout} @{s} B={B} (split {ns}): x fp32 {fl / m[0] / 1e12:6.1f} TF   x limb planes {fl / m[1] / 1e12:6.1f} TF "
              f"x{m[0] / m[1]:.3f}   bitwise {bool(torch.equal(s0, s1))}")


if __name__ == "__main__":
    if "--wgrad" in sys.argv:
        wgrad()
        sys.exit(0)
    main()
