"""Achieved HBM GB/s of the bandwidth-bound kernels at north-star sizes (B=128, C=256, 32x32 unless
noted): algorithmic bytes (each distinct input read once, each output written once) / time.
Peak: 8 TB/s spec (6.3 TB/s measured float4-copy ceiling, MI355X_MICROARCH.md)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psld_amd import ops
from psld_amd._lib import EmCoeffs, SdeParams
from tools.bench_tile import timeit

DEV = "cuda"
ops.lib()
B, S, C = 128, 32, 256
A = 4 * B * S * S * C   # bytes of one activation tensor
rows = []

def rec(name, nbytes, fn, iters=20):
    t = timeit(fn, iters)
    rows.append((name, nbytes / t / 1e9, t * 1e6, nbytes / 1e6))

x = torch.randn(B, S, S, C, device=DEV); y = torch.empty_like(x); dy = torch.randn_like(x); dx = torch.empty_like(x)
gamma = torch.ones(C, device=DEV); beta = torch.zeros(C, device=DEV)
st = ops.gn_stats(x, gamma, beta)
rec("gn_stats (partial+finalize)", A, lambda: ops.gn_stats(x, gamma, beta))
rec("gn_apply + SiLU", 2 * A, lambda: ops.gn_apply(x, st, True, out=y))
rec("gn_apply + SiLU + dropout 0.15", 2 * A, lambda: ops.gn_apply(x, st, True, out=y, drop_p=0.15, seed=7))
dg, db = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
rec("gn_bwd (2 passes over x,dy + dx)", 5 * A, lambda: ops.gn_bwd(dy, x, st, gamma, beta, True, dx, dg, db))
k = np.outer([1, 3, 3, 1], [1, 3, 3, 1]).astype(np.float32); k /= k.sum()
xs = torch.randn(B, 16, 16, C, device=DEV); up = torch.empty(B, 32, 32, C, device=DEV)
rec("FIR up x2 (16->32)", A / 4 + A, lambda: ops.upfirdn2d_raw(xs, k * 4, 2, 1, (2, 1), 1, out=up))
dn = torch.empty(B, 16, 16, C, device=DEV)
rec("FIR down x2 (32->16)", A + A / 4, lambda: ops.upfirdn2d_raw(x, k, 1, 2, (1, 1), 1, out=dn))
# context for the two resamplers: a write-dominated stream (up: 1 byte read per 4 written) and a read-dominated one
rec("  (ceiling) fill: write-only, same bytes as FIR up's output", A, lambda: up.fill_(1.0))
rec("  (ceiling) sum: read-only, same bytes as FIR down's input", A, lambda: torch.sum(x))
cat = torch.empty(B, S, S, 2 * C, device=DEV)
rec("concat copy2d (one half)", 2 * A, lambda: ops.copy2d(x, C, cat, 2 * C, B * S * S, C))
rec("axpby accumulate", 3 * A, lambda: ops.axpby(x, 0.7, None, 0.0, y, accumulate=True))
cs = torch.empty(B, C, device=DEV)
rec("colsum per image", A, lambda: ops.colsum(x, C, B, S * S, C, cs))
p = torch.randn(B, 256, 256, device=DEV); q = torch.empty_like(p)
rec("softmax rows (B x 256 x 256)", 2 * p.numel() * 4, lambda: ops.softmax_rows(p, q, B * 256, 256))
xn = torch.randn(B, 6, S, S, device=DEV)
rec("nchw->nhwc (6 ch)", 2 * xn.numel() * 4, lambda: ops.nchw_to_nhwc(xn))
# SDE kernels, B=512 images of 6x32x32
Bs = 512
prm = SdeParams(); prm.beta_0 = prm.beta_1 = 8.0; prm.nu, prm.gamma = 4.01, 0.01
prm.m_inv = (0.01 - 4.01) ** 2 / 4; prm.numerical_eps = 1e-9; prm.decomp_lower = 1
x0 = torch.rand(Bs, 3, S, S, device=DEV); eps = torch.randn(Bs, 6, S, S, device=DEV)
t = torch.rand(Bs, device=DEV, dtype=torch.float64) * 0.99 + 0.005
flag = torch.zeros(1, dtype=torch.int32, device=DEV)
co = ops.perturb_coeffs(t, prm, 0.0, 0.01, flag)
rec("psld_perturb (x0,eps -> z_t f32), B=512", Bs * 60 * 1024, lambda: ops.perturb(x0, None, eps, co, prm))
pred = torch.randn_like(eps)
rec("sqerr loss + grad, B=512", Bs * (48 + 24) * 1024, lambda: ops.sqerr_loss(eps, pred, True, True))
xs64 = torch.randn(Bs, 6, S, S, device=DEV, dtype=torch.float64); z = torch.randn_like(xs64); xf = torch.empty(Bs, 6, S, S, device=DEV)
kk = EmCoeffs(); kk.beta = 8.0; kk.m_inv, kk.gamma, kk.nu, kk.m = 4.0, 0.01, 4.01, 0.25
kk.c11, kk.c12, kk.c21, kk.c22 = 1.0, -0.1, 0.0, 2.0; kk.dt = 1e-3; kk.score_mode = 0; kk.probability_flow = 0
rec("em_step f64 (+f32 copy), B=512", Bs * (48 + 24 + 48 + 48 + 24) * 1024, lambda: ops.em_step(xs64, pred, z, kk, xf))
n = 97_628_000
pp, g, m, v, e = (torch.randn(n, device=DEV) * 0.01 for _ in range(5))
v.abs_()
norm = torch.zeros(1, dtype=torch.float64, device=DEV)
rec("grad_norm (97.6M)", 4 * n, lambda: ops.grad_norm(g, norm), 10)
rec("clip+Adam (97.6M)", 28 * n, lambda: ops.adam_ema(pp, g, m, v, None, norm, 1.0, 2e-4, 0.9, 0.999, 1e-8, 0.0, 3, 0.9999), 10)
rec("clip+Adam+EMA fused (97.6M)", 36 * n, lambda: ops.adam_ema(pp, g, m, v, e, norm, 1.0, 2e-4, 0.9, 0.999, 1e-8, 0.0, 3, 0.9999), 10)
rec("EMA (97.6M)", 12 * n, lambda: ops.ema(e, pp, 0.9999), 10)
print(f"{'kernel':44s} {'GB/s':>8s} {'% of 8 TB/s':>11s} {'us':>9s} {'MB':>9s}")
for name, gbs, us, mb in rows:
    print(f"{name:44s} {gbs:8.0f} {100 * gbs / 8000:10.1f}% {us:9.1f} {mb:9.1f}")
