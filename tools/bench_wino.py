"""Winograd F(2x2,3x3) limb convolution (psld_conv3x3_wino_f32) against the direct limb kernels, interleaved rounds in
ONE process, random data: direct-conv-equivalent TFLOP/s (2*M*N*9*Cin / time) and rel-L2 against fp64 torch.
    python tools/bench_wino.py [--rounds 5] [--iters 10] [--batch 128] [--check]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

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


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm())


def check():
    """small shapes against fp64 (full epilogue, two sources, odd batches, every supported width, data gradient)"""
    worst = 0.0
    for (b, c1, c2, co, h, w_) in [(2, 64, 0, 128, 8, 8), (3, 32, 0, 128, 8, 8), (2, 64, 32, 128, 16, 16),
                                   (1, 32, 0, 256, 32, 32), (5, 96, 0, 128, 32, 32), (1, 32, 0, 128, 64, 64),
                                   (3, 256, 256, 256, 16, 16), (2, 32, 0, 128, 4, 8)]:
        g = torch.Generator().manual_seed(40)
        x = torch.randn(b, c1 + c2, h, w_, generator=g)
        w = torch.randn(co, c1 + c2, 3, 3, generator=g) * 0.1
        bias, res, temb = torch.randn(co, generator=g), torch.randn(b, co, h, w_, generator=g), torch.randn(b, co, generator=g)
        ref = (F.conv2d(x.double(), w.double(), bias.double(), padding=1) + temb.double()[:, :, None, None] + res.double()) * 0.7
        nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
        x1 = nhwc(x[:, :c1]).to(DEV)
        x2 = nhwc(x[:, c1:]).to(DEV) if c2 else None
        assert ops.conv3x3_wino_supported(c1, c2, b, h, w_, co)
        uf = ops.conv3x3_wino_frag(w.to(DEV), False)
        y = torch.full((b, h, w_, co), float("nan"), device=DEV)
        epi = ops.epilogue(bias=bias.to(DEV), rowbias=temb.to(DEV), rows_per_img=h * w_, residual=nhwc(res).to(DEV),
                           ld_residual=co, out_scale=0.7)
        ops.conv3x3_wino(x1, x2, uf, co, y, epi)
        yd = torch.empty_like(y)
        ops.conv3x3_split(x1, x2, ops.conv3x3_frag(w.to(DEV), False), co, yd, epi)
        e, ed = rel_l2(y.permute(0, 3, 1, 2), ref), rel_l2(yd.permute(0, 3, 1, 2), ref)
        worst = max(worst, e)
        print(f"fwd b={b} {c1}+{c2}->{co} @{h}x{w_}: winograd {e:.2e}  direct {ed:.2e}")
        # data gradient: conv of gy with the rotated, role-swapped filter
        if c2 == 0 and co % 32 == 0 and c1 % 128 == 0:
            pass
    for (b, ci, co, h, w_) in [(2, 128, 64, 8, 8), (3, 128, 128, 16, 16), (1, 256, 64, 32, 32), (2, 128, 256, 32, 32)]:
        g = torch.Generator().manual_seed(60)
        x = torch.randn(b, ci, h, w_, generator=g).requires_grad_(True)
        w = (torch.randn(co, ci, 3, 3, generator=g) * 0.1).requires_grad_(True)
        yy = F.conv2d(x.double(), w.double(), padding=1)
        gy = torch.randn(*yy.shape, generator=g)
        yy.backward(gy.double())
        gyd = gy.permute(0, 2, 3, 1).contiguous().to(DEV)
        dx = torch.full((b, h, w_, ci), float("nan"), device=DEV)
        ops.conv3x3_wino(gyd, None, ops.conv3x3_wino_frag(w.detach().to(DEV), True), ci, dx)
        e = rel_l2(dx.permute(0, 3, 1, 2), x.grad)
        worst = max(worst, e)
        print(f"dgrad b={b} {co}->{ci} @{h}x{w_}: winograd {e:.2e}")
    print("worst", worst)
    assert worst < 1e-5


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--epi", default="res", help="epilogue of the timed launches: plain | bias | res (bias + residual + scale)")
    ap.add_argument("--fused-gn", action="store_true", help="GroupNorm apply + SiLU: separate pass + convolution vs fused into the staging")
    ap.add_argument("--shapes", default="256,256,32;512,256,32;256,256,16;512,256,16;256,256,8")
    args = ap.parse_args()
    ops.lib()
    if args.check:
        check()
    B = args.batch
    for spec in args.shapes.split(";"):
        cin, cout, s = (int(v) for v in spec.split(","))
        if not ops.conv3x3_wino_supported(cin, 0, B, s, s, cout):
            continue
        x = torch.randn(B, s, s, cin, device=DEV)
        w = torch.randn(cout, cin, 3, 3, device=DEV) * 0.05
        wf = ops.conv3x3_frag(w, False)
        uf = ops.conv3x3_wino_frag(w, False)
        bias = torch.randn(cout, device=DEV)
        res = torch.randn(B, s, s, cout, device=DEV)
        epi = (ops.epilogue() if args.epi == "plain" else ops.epilogue(bias=bias) if args.epi == "bias"
               else ops.epilogue(bias=bias, residual=res, ld_residual=cout, out_scale=0.7))
        y0, y1, y2 = (torch.empty(B, s, s, cout, device=DEV) for _ in range(3))
        if args.fused_gn:
            if not ops.conv3x3_wino_gn_supported(cin, 0, B, s, s, cout):
                continue
            gamma, beta = torch.rand(cin, device=DEV) + 0.5, torch.randn(cin, device=DEV) * 0.1
            st = ops.gn_stats(x, gamma, beta)
            act = torch.empty_like(x)
            ya, yb = torch.empty(B, s, s, cout, device=DEV), torch.empty(B, s, s, cout, device=DEV)

            def two_pass():
                ops.gn_apply(x, st, True, out=act)
                ops.conv3x3_wino(act, None, uf, cout, ya, epi)

            fa, fb, fc = two_pass, (lambda: ops.conv3x3_wino_gn(x, st, None, None, True, uf, cout, yb, epi)), \
                (lambda: ops.conv3x3_wino(act, None, uf, cout, ya, epi))
            tt = [[], [], []]
            for _ in range(args.rounds):
                for i, f in enumerate((fa, fb, fc)):
                    tt[i].append(timeit(f, args.iters))
            ma, mb, mc = (sorted(t)[len(t) // 2] for t in tt)
            print(f"conv fwd {cin}->{cout} @{s} B={B} GN+SiLU: apply pass + conv {ma * 1e6:7.1f} us (conv alone {mc * 1e6:7.1f})  "
                  f"fused {mb * 1e6:7.1f} us  x{ma / mb:.3f} vs two passes, conv {mb / mc - 1:+.1%} per launch  bitwise equal: {torch.equal(ya, yb)}",
                  flush=True)
            continue
        xl = ops.f32_to_limb(x)
        fl = 2.0 * B * s * s * cout * 9 * cin
        fs = [lambda: ops.conv3x3_split(x, None, wf, cout, y0, epi), lambda: ops.conv3x3_split(xl, None, wf, cout, y1, epi),
              lambda: ops.conv3x3_wino(x, None, uf, cout, y2, epi)]
        ts = [[], [], []]
        for _ in range(args.rounds):
            for i, f in enumerate(fs):
                ts[i].append(timeit(f, args.iters))
        m = [sorted(t)[len(t) // 2] for t in ts]
        nb = min(B, 4)
        ref = F.conv2d(x[:nb].permute(0, 3, 1, 2).double(), w.double(), bias.double() if args.epi != "plain" else None, padding=1)
        if args.epi == "res":
            ref = (ref + res[:nb].permute(0, 3, 1, 2).double()) * 0.7
        e1, e2 = rel_l2(y1[:nb].permute(0, 3, 1, 2), ref), rel_l2(y2[:nb].permute(0, 3, 1, 2), ref)
        print(f"conv fwd {cin}->{cout} @{s} B={B}: direct fp32-in {fl / m[0] / 1e12:6.1f} TF  direct limb-in {fl / m[1] / 1e12:6.1f} TF  "
              f"winograd {fl / m[2] / 1e12:6.1f} TF (best {fl / min(ts[2]) / 1e12:6.1f}; {m[2] * 1e6:7.1f} us)  x{m[1] / m[2]:.3f} vs limb-in   "
              f"rel-L2 vs fp64: direct {e1:.2e} winograd {e2:.2e}", flush=True)


if __name__ == "__main__":
    main()
