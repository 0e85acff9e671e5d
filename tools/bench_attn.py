"""Fused attention forward (psld_attn_fwd_split_f32) against the three-kernel path (batched limb GEMM, softmax, batched limb
GEMM), interleaved, random data.   python tools/bench_attn.py [--batch 128]"""
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
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    ops.lib()
    b, c = args.batch, 256
    for hw in (256, 64):
        qkv = torch.randn(b, hw, 3 * c, device=DEV)
        q, k, v, ld = qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:], 3 * c
        scale = c ** -0.5
        o0, o1, o2 = (torch.empty(b, hw, c, device=DEV) for _ in range(3))
        p0, p1 = torch.empty(b, hw, hw, device=DEV), torch.empty(b, hw, hw, device=DEV)

        def three():
            if ops.bgemm_split_supported(0, 1, hw, hw, c):
                ops.bgemm_split(0, 1, hw, hw, c, q, ld, hw * ld, k, ld, hw * ld, p0, hw, hw * hw, b, scale)
            else:
                ops.gemm_raw(0, 1, hw, hw, c, q, ld, hw * ld, k, ld, hw * ld, p0, hw, hw * hw, b, ops.epilogue(alpha=scale))
            ops.softmax_rows(p0, p0, b * hw, hw)
            if ops.bgemm_split_supported(0, 0, hw, c, hw):
                ops.bgemm_split(0, 0, hw, c, hw, p0, hw, hw * hw, v, ld, hw * ld, o0, c, hw * c, b)
            else:
                ops.gemm_raw(0, 0, hw, c, hw, p0, hw, hw * hw, v, ld, hw * ld, o0, c, hw * c, b)
        fs = [three, lambda: ops.attn_fwd(q, k, v, ld, b, hw, c, scale, o1, p1), lambda: ops.attn_fwd(q, k, v, ld, b, hw, c, scale, o2, None)]
        ts = [[], [], []]
        for _ in range(args.rounds):
            for i, f in enumerate(fs):
                ts[i].append(timeit(f, args.iters))
        m = [sorted(t)[len(t) // 2] * 1e6 for t in ts]
        fl = 2 * 2.0 * b * hw * hw * c
        err = float((o1 - o0).norm() / o0.norm())
        print(f"attention fwd B={b} HW={hw} C={c}: three kernels {m[0]:7.1f} us   fused + P {m[1]:7.1f} us ({fl / m[1] / 1e6:5.1f} TF)   "
              f"fused, no P {m[2]:7.1f} us ({fl / m[2] / 1e6:5.1f} TF)   rel diff {err:.1e}")


if __name__ == "__main__":
    main()
