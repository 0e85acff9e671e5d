"""K-sweep of the GEMM NT tile kernel at the conv-like shape (M=131072, N=256) to separate the
K-proportional main loop from the per-launch fixed cost (prologue / epilogue / scheduling)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psld_amd import ops
from tools.bench_tile import timeit
ops.lib()
M, N = 131072, 256
for K in (32, 64, 288, 1152, 2304, 4608):
    A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda"); C = torch.empty(M, N, device="cuda")
    t = timeit(lambda: ops.gemm_raw(0, 1, M, N, K, A, K, 0, B, K, 0, C, N, 0), 10)
    print(f"K={K:5d}  {t*1e6:8.1f} us  {2.0*M*N*K/t/1e12:6.1f} TF")
for (M2, N2) in ((32768, 256), (131072, 128), (131072, 512)):
    K = 2304
    A = torch.randn(M2, K, device="cuda"); B = torch.randn(N2, K, device="cuda"); C = torch.empty(M2, N2, device="cuda")
    t = timeit(lambda: ops.gemm_raw(0, 1, M2, N2, K, A, K, 0, B, K, 0, C, N2, 0), 10)
    print(f"M={M2} N={N2} K={K}  {t*1e6:8.1f} us  {2.0*M2*N2*K/t/1e12:6.1f} TF")
