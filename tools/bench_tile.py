"""Micro-benchmark of the MFMA tile engine on the north-star layer shapes (run on the GPU box).
    python tools/bench_tile.py [--iters 10]
Prints TFLOP/s per shape for: plain GEMM NT, conv forward, conv data-gradient, conv weight-gradient."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psld_amd import ops  # noqa: E402

DEV = "cuda"


def timeit(fn, iters):
    for _ in range(2):
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
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--quick", action="store_true", help="only 4096^3 GEMM and the 256->256@32 conv (for --pmc runs)")
    args = ap.parse_args()
    B = args.batch
    ops.lib()
    rows = []
    # plain GEMM
    gemm_shapes = [(B * 1024, 256, 2304), (B * 256, 256, 2304), (B * 64, 256, 2304), (4096, 4096, 4096)]
    conv_shapes = [(256, 256, 32), (512, 256, 32), (256, 256, 16), (512, 256, 16), (256, 256, 8), (512, 256, 8)]
    if args.quick:
        gemm_shapes, conv_shapes = [(4096, 4096, 4096)], [(256, 256, 32)]
    for (M, N, K) in gemm_shapes:
        A = torch.randn(M, K, device=DEV)
        Bm = torch.randn(N, K, device=DEV)
        C = torch.empty(M, N, device=DEV)
        t = timeit(lambda: ops.gemm_raw(0, 1, M, N, K, A, K, 0, Bm, K, 0, C, N, 0), args.iters)
        rows.append((f"gemm NT {M}x{N}x{K}", 2.0 * M * N * K / t / 1e12, t * 1e6))
    for (cin, cout, s) in conv_shapes:
        x = torch.randn(B, s, s, cin, device=DEV)
        w = torch.randn(cout, 9, cin, device=DEV) * 0.05
        y = torch.empty(B, s, s, cout, device=DEV)
        bias = torch.zeros(cout, device=DEV)
        epi = ops.epilogue(bias=bias)
        fl = 2.0 * B * s * s * cout * 9 * cin
        t = timeit(lambda: ops.conv2d_nhwc(x, None, w, cout, 3, 3, 1, 1, 1, s, s, y, epi), args.iters)
        rows.append((f"conv fwd {cin}->{cout} @{s}", fl / t / 1e12, t * 1e6))
        if ops.conv3x3_split_supported(cin, 0, B, s, s, cout):
            w_oihw = w.view(cout, 3, 3, cin).permute(0, 3, 1, 2).contiguous()
            wf = ops.conv3x3_frag(w_oihw, False)
            y2 = torch.empty_like(y)
            t = timeit(lambda: ops.conv3x3_split(x, None, wf, cout, y2, epi), args.iters)
            err = ((y2 - y).norm() / y.norm()).item()
            rows.append((f"conv fwd {cin}->{cout} @{s} bf16x6 (rel {err:.1e})", fl / t / 1e12, t * 1e6))
        dy = torch.randn(B, s, s, cout, device=DEV)
        nsplit = max(1, min((768 + 36 * (cin // 256) - 1) // (36 * (cin // 256)), B * s * s // 256))
        slabs = torch.empty(nsplit, cout, 9, cin, device=DEV)
        t = timeit(lambda: ops.conv2d_wgrad_nhwc(dy, cout, x, 3, 3, 1, 1, s, s, slabs, cin, 0, nsplit), args.iters)
        rows.append((f"conv wgrad {cin}->{cout} @{s} (split {nsplit})", fl / t / 1e12, t * 1e6))
        if ops.conv3x3_wgrad_split_supported(cout, cin, B, s, s):
            co_tile = ops.conv3x3_wgrad_split_cout_tile(cout)
            tiles = (cout // co_tile) * (cin // 64)
            kt = B * s * s // 32
            ns2 = max(1, min((768 if co_tile == 64 else 512) // (3 * tiles), kt // 4))
            per = -(-kt // ns2)
            ns2 = -(-kt // per)
            slabs2 = torch.empty(ns2, cout, 9, cin, device=DEV)
            t = timeit(lambda: ops.conv3x3_wgrad_split(dy, cout, x, slabs2, cin, 0, ns2), args.iters)
            ref = slabs.sum(0)
            err = ((slabs2.sum(0) - ref).norm() / ref.norm()).item()
            rows.append((f"conv wgrad {cin}->{cout} @{s} bf16x6 (split {ns2}, rel {err:.1e})", fl / t / 1e12, t * 1e6))
    for name, tf, us in rows:
        print(f"{name:54s} {tf:7.1f} TF  {us:9.1f} us")


if __name__ == "__main__":
    main()
