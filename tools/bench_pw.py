"""Pointwise limb GEMMs (1x1 convolutions: ResBlock shortcuts, attention projections) on the shapes of a B=128 step:
    python tools/bench_pw.py [--batch 128]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psld_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=1, help="ignored (pmc_wait.sh passes it)")
    ap.add_argument("--only", type=int, default=-1, help="index of the one shape to run")
    ap.add_argument("--check", action="store_true", help="also print rel-L2 vs an fp64 GEMM and a CRC of the output bytes (A/B of two kernels: equal CRCs = bitwise equal)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    b = args.batch
    shapes = [("shortcut fwd  512->256 @32 (two sources)", b * 1024, 256, 256, 256, 9),
              ("shortcut dgrad 256->512 @32", b * 1024, 256, 0, 512, 9),
              ("shortcut fwd  512->256 @16 (two sources)", b * 256, 256, 256, 256, 9),
              ("shortcut dgrad 256->512 @16", b * 256, 256, 0, 512, 9),
              ("q|k|v fwd     256->768 @16", b * 256, 256, 0, 768, 17),
              ("q|k|v dgrad   768->256 @16", b * 256, 768, 0, 256, 17),
              ("out proj      256->256 @16", b * 256, 256, 0, 256, 34),
              ("shortcut fwd  512->256 @8 (two sources)", b * 64, 256, 256, 256, 9),
              ("shortcut dgrad 256->512 @8", b * 64, 256, 0, 512, 9)]
    total = 0.0
    if args.only >= 0:
        shapes = shapes[args.only:args.only + 1]
    for name, m, k1, k2, n, per_step in shapes:
        a1 = torch.randn(m, k1, device=dev)
        a2 = torch.randn(m, k2, device=dev) if k2 else None
        w = torch.randn(n, k1 + k2, device=dev) * 0.05
        frag = ops.gemm_frag(w, n, k1 + k2, k1 + k2, 1)
        y = torch.empty(m, n, device=dev)
        bias = torch.randn(n, device=dev)
        epi = ops.epilogue(bias=bias)
        for _ in range(3):
            ops.gemm_split(a1, a2, m, frag, n, y, epi)
        torch.cuda.synchronize()
        chk = ""
        if args.check:
            import zlib
            rows = min(m, 4096)
            af = torch.cat([a1[:rows], a2[:rows]], 1) if a2 is not None else a1[:rows]
            ref = af.double() @ w.double().t() + bias.double()
            rel = float((y[:rows].double() - ref).norm() / ref.norm())
            tail = float((y[m - 128:].double() - ((torch.cat([a1[m - 128:], a2[m - 128:]], 1) if a2 is not None else a1[m - 128:]).double() @ w.double().t() + bias.double())).abs().max())
            chk = f"  rel-L2 {rel:.2e} tail-maxabs {tail:.1e} crc {zlib.crc32(y.cpu().numpy().tobytes()):08x}"
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            ops.gemm_split(a1, a2, m, frag, n, y, epi)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.iters
        flops = 2.0 * m * (k1 + k2) * n
        mb = (m * (k1 + k2) + m * n) * 4 / 1e6
        total += us * per_step
        print(f"{name:44s} M={m:7d}  {us:8.1f} us  {flops / us / 1e6:6.1f} TF  {mb / us:5.2f} TB/s  x{per_step}/step = {us * per_step / 1e3:.2f} ms{chk}")
    print(f"sum over a step: {total / 1e3:.2f} ms")


if __name__ == "__main__":
    main()
