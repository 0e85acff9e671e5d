"""Cost of the fused epilogue variants on a short-K pointwise limb GEMM (M=131072, K=256, N=256) and a 3x3 conv."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psld_amd import ops
DEV = "cuda"
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / it
M, K, N = 131072, 256, 256
a = torch.randn(M, K, device=DEV); w = torch.randn(N, K, device=DEV) * 0.05
fr = ops.gemm_frag(w, N, K, K, 1)
y = torch.zeros(M, N, device=DEV); res = torch.randn(M, N, device=DEV); bias = torch.randn(N, device=DEV)
tp = torch.randn(128, N, device=DEV)
fl = 2.0 * M * N * K
for tag, epi in (("none", None), ("bias", ops.epilogue(bias=bias)),
                 ("bias+residual+scale", ops.epilogue(bias=bias, residual=res, ld_residual=N, out_scale=0.7)),
                 ("accumulate", ops.epilogue(alpha=0.7, accumulate=True)),
                 ("bias+rowbias", ops.epilogue(bias=bias, rowbias=tp, rows_per_img=1024))):
    t = timeit(lambda: ops.gemm_split(a, None, M, fr, N, y, epi))
    print(f"gemm {M}x{N}x{K} epilogue {tag:22s} {t*1e6:8.1f} us {fl/t/1e12:6.1f} TF")
x = torch.randn(128, 32, 32, 256, device=DEV); w3 = torch.randn(256, 256, 3, 3, device=DEV) * 0.02
f3 = ops.conv3x3_frag(w3, False); y3 = torch.zeros(128, 32, 32, 256, device=DEV); r3 = torch.randn_like(y3)
fl3 = 2.0 * M * 256 * 2304
for tag, epi in (("none", None), ("bias+residual+scale", ops.epilogue(bias=bias, residual=r3, ld_residual=256, out_scale=0.7)),
                 ("accumulate", ops.epilogue(alpha=0.7, accumulate=True))):
    t = timeit(lambda: ops.conv3x3_split(x, None, f3, 256, y3, epi))
    print(f"conv3x3 256->256 @32 epilogue {tag:22s} {t*1e6:8.1f} us {fl3/t/1e12:6.1f} TF")
