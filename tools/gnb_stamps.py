"""Diagnostic (libpsld_hip_abl.so): time stamps inside gn_bwd_fused_kernel and its timing-only modes.
    PSLD_HIP_LIB=tools/abl/libpsld_hip_abl.so python tools/gnb_stamps.py [B size C [mode]]
mode: bit 0 delay odd workgroups by (mode >> 8) x 3.4 us, bit 1 no reduction, bit 2 no stores."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psld_amd import _lib, ops  # noqa: E402
from tools.bench_tile import timeit  # noqa: E402

B, S, C = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (128, 32, 256)))
mode = int(sys.argv[4], 0) if len(sys.argv) >= 5 else 0
variant = sys.argv[5] if len(sys.argv) >= 6 else "plain"          # plain | add | acc
ops.lib()
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.psld_abl_set_gnb_debug.argtypes = [ctypes.c_void_p, ctypes.c_int]
x = torch.randn(B, S, S, C, device="cuda")
dy = torch.randn_like(x)
dx = torch.empty_like(x)
gamma = torch.rand(C, device="cuda") + 0.5
beta = torch.randn(C, device="cuda") * 0.1
st = ops.gn_stats(x, gamma, beta)
dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
nwg = B * 32                                  # upper bound on the workgroups of one launch
dbg = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")


other = torch.randn_like(x)
kw = {"add": other, "add_scale": 0.7} if variant == "add" else {"accumulate_dx": True} if variant == "acc" else {}


def run():
    ops.gn_bwd(dy, x, st, gamma, beta, True, dx, dg, db, **kw)


raw.psld_abl_set_gnb_debug(None, mode)
t = timeit(run, 30)
print(f"B={B} {S}x{S} C={C} mode={mode:#x} {variant}: {t * 1e6:.1f} us per call (fused + finalize), {12 * x.numel() / t / 1e9:.0f} GB/s on 12 B/element")
raw.psld_abl_set_gnb_debug(ctypes.c_void_p(dbg.data_ptr()), mode)
run()
torch.cuda.synchronize()
dbg.zero_()
run()
torch.cuda.synchronize()
raw.psld_abl_set_gnb_debug(None, 0)
t = dbg.view(nwg, 8).cpu().double()
t = t[t[:, 0] > 0]
print(f"{t.shape[0]} workgroups stamped")
pipe = bool((t[:, 6] < 1e15).all())        # the resident kernel stamps the shader clock in all eight slots
if pipe:
    seg = ["dy loads issued", "x image landed | x, image asked", "dy (x) landed + pass 1", "next x asked", "thread sums + barrier 1",
           "reduction (2 barriers)", "pass 2 (image) + stores issued"]
    d = t[:, 1:8] - t[:, 0:7]
    for i, n in enumerate(seg):
        col = d[:, i]
        print(f"  {n:28s} median {col.median():8.0f}  p10 {col.quantile(0.1):8.0f}  p90 {col.quantile(0.9):8.0f} shader-clock ticks")
    print(f"  {'second slab, total':28s} median {(t[:, 7] - t[:, 0]).median():8.0f}")
    sys.exit(0)
seg = ["loads issued", "own loads landed + pass 1", "barrier 1", "reduction", "pass 2 + stores issued"]
d = t[:, 1:6] - t[:, 0:5]
for i, n in enumerate(seg):
    col = d[:, i]
    print(f"  {n:28s} median {col.median():8.0f}  p10 {col.quantile(0.1):8.0f}  p90 {col.quantile(0.9):8.0f} shader-clock ticks")
tot = t[:, 5] - t[:, 0]
print(f"  {'workgroup total':28s} median {tot.median():8.0f}")
rt0 = t[:, 6].min()
start = (t[:, 6] - rt0) / 100.0               # s_memrealtime: 100 MHz
end = (t[:, 7] - rt0) / 100.0
order = start.argsort()
print("global timeline (us since the first workgroup started): start / end of every 64th workgroup in start order")
for i in order[::max(1, len(order) // 24)]:
    print(f"  wg {int(i):5d}  {start[i]:7.2f} -> {end[i]:7.2f}  ({end[i] - start[i]:6.2f} us)")
print(f"  last end {end.max():.2f} us; mean workgroup life {(end - start).mean():.2f} us; ticks per us {(tot / ((end - start) + 1e-9)).median():.0f}")
