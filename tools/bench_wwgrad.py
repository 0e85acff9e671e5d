"""the Winograd-domain weight gradient (psld_conv3x3_wgrad_wino_f32, wgrad_wino.hip) against the direct
wave-specialised limb kernel (psld_conv3x3_wgrad_split_f32 + the slab reduction), interleaved rounds in ONE process on the
same inputs, B=128.  Per shape: microseconds per call (kernel + its reduction), direct-equivalent TFLOP/s (2 M N 9 C_in), the
ratio, and both results' rel-L2 against an fp64 reference (nine fp64 GEMMs on the device).
    python tools/bench_wwgrad.py [--rounds 5] [--iters 10] [--batch 128] [--no-ref]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psld_amd import ops  # noqa: E402
from psld_amd.score_fn import _pick_nsplit  # noqa: E402

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


def ref64(dy, x):
    """dW[co][ci][ky][kx] in fp64 from NHWC dy / x."""
    b, h, w, co = dy.shape
    ci = x.shape[-1]
    xp = torch.nn.functional.pad(x.double(), (0, 0, 1, 1, 1, 1))
    d = dy.double().reshape(-1, co)
    out = torch.empty(co, ci, 3, 3, dtype=torch.float64, device=dy.device)
    for ky in range(3):
        for kx in range(3):
            out[:, :, ky, kx] = d.t() @ xp[:, ky:ky + h, kx:kx + w, :].reshape(-1, ci)
    return out


def rel(a, b):
    return ((a.double() - b).norm() / b.norm()).item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--no-ref", action="store_true")
    ap.add_argument("--shapes", default="256:0:256:32,256:256:256:32,256:0:256:16,256:256:256:16,128:0:256:32,256:128:256:32,128:0:128:64,128:128:128:64")
    args = ap.parse_args()
    B = args.batch
    ops.lib()
    print(f"B={B}  rounds={args.rounds} x iters={args.iters} (median of rounds); direct = dwgrad_ws_kernel + reduce_slabs, "
          "winograd = wwgrad_ws_kernel + wwgrad_reduce_kernel")
    for spec in args.shapes.split(","):
        c1, c2, cout, s = (int(v) for v in spec.split(":"))
        cin = c1 + c2
        if not ops.conv3x3_wgrad_wino_supported(cout, c1, c2, B, s, s):
            print(f"{c1}+{c2}->{cout} @{s}: not taken by the Winograd-domain kernel")
            continue
        g = torch.Generator(device=DEV).manual_seed(1)
        x1 = torch.randn(B, s, s, c1, device=DEV, generator=g) + 0.25
        x2 = torch.randn(B, s, s, c2, device=DEV, generator=g) + 0.25 if c2 else None
        dy = torch.randn(B, s, s, cout, device=DEV, generator=g)
        fl = 2.0 * B * s * s * cout * 9 * cin
        kt = B * s * s // 32
        ns = _pick_nsplit((cout // 128) * (cin // 64) * 3, kt * 32, min_k=128, resident=256)
        per = -(-kt // ns)
        ns = -(-kt // per)
        n = cout * 9 * cin
        slabs = torch.empty(ns, cout, 9, cin, device=DEV)
        dw0 = torch.empty(cout, cin, 3, 3, device=DEV)
        dw1 = torch.empty(cout, cin, 3, 3, device=DEV)
        nsw, wsb = ops.conv3x3_wgrad_wino_plan(cout, cin, B, s, s)
        wslabs = torch.empty(wsb, device=DEV, dtype=torch.uint8)

        def f0():
            ops.conv3x3_wgrad_split(dy, cout, x1, slabs, cin, 0, ns, x2)
            ops.reduce_slabs(slabs, ns, n, dw0, layout=1, cout=cout, taps=9, cin=cin)

        def f1():
            ops.conv3x3_wgrad_wino(dy, cout, x1, dw1, x2=x2, slabs=wslabs)

        ts = [[], []]
        for _ in range(args.rounds):
            for i, f in enumerate((f0, f1)):
                ts[i].append(timeit(f, args.iters))
        m = [sorted(t)[len(t) // 2] for t in ts]
        line = (f"{c1}+{c2}->{cout} @{s}: direct {m[0] * 1e6:7.1f} us {fl / m[0] / 1e12:6.1f} TF (split {ns}) | winograd {m[1] * 1e6:7.1f} us "
                f"{fl / m[1] / 1e12:6.1f} TF (split {nsw}) | x{m[0] / m[1]:.3f}")
        if not args.no_ref:
            r = ref64(dy, torch.cat([x1, x2], dim=-1) if c2 else x1)
            line += f" | rel-L2 vs fp64: direct {rel(dw0, r):.2e} winograd {rel(dw1, r):.2e}"
        else:
            line += f" | winograd vs direct {rel(dw1, dw0.double()):.2e}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
