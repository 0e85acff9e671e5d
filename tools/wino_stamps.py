"""Diagnostic (libpsld_hip_abl.so, PSLD_WINO_ABL=64): s_memtime stamps inside chunk 3 of wino_conv8s_kernel - how long a
wave spends in its transform block, its MFMA block and at the two barriers of a chunk.
    PSLD_HIP_LIB=tools/abl/libpsld_hip_abl.so PSLD_WINO_ABL=64 python tools/wino_stamps.py [cin cout size batch]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psld_amd import _lib, ops  # noqa: E402

cin, cout, s, B = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (256, 256, 32, 128)))
lib = ops.lib()
raw = ctypes.CDLL(_lib.LIB_PATH)
x = torch.randn(B, s, s, cin, device="cuda")
w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
uf = ops.conv3x3_wino_frag(w, False)
res = torch.randn(B, s, s, cout, device="cuda")
epi = ops.epilogue(bias=torch.randn(cout, device="cuda"), residual=res, ld_residual=cout, out_scale=0.7)
y = torch.empty(B, s, s, cout, device="cuda")
nwg = (B * s * s // 128) * (cout // 128)
dbg = torch.zeros(nwg * 8 * 8, dtype=torch.int64, device="cuda")
raw.psld_abl_set_wino_debug.argtypes = [ctypes.c_void_p]
raw.psld_abl_set_wino_debug(ctypes.c_void_p(dbg.data_ptr()))
for _ in range(20):          # clocks settle
    ops.conv3x3_wino(x, None, uf, cout, y, epi)
torch.cuda.synchronize()
dbg.zero_()
ops.conv3x3_wino(x, None, uf, cout, y, epi)
torch.cuda.synchronize()
t = dbg.view(nwg, 8, 8).cpu().double()
d = t[:, :, 1:] - t[:, :, :-1]          # segment lengths in cycles of the 100 MHz-independent shader clock counter
names_lo = ["HP0 transform", "HP0 MFMA", "HP0 store_raw", "HP0 barrier", "HP1 transform", "HP1 MFMA", "HP1 barrier"]
names_hi = ["HP0 MFMA", "HP0 transform", "HP0 store_raw", "HP0 barrier", "HP1 MFMA", "HP1 transform", "HP1 barrier"]
for grp, names, sl in (("waves 0-3 (transform, then MFMA)", names_lo, slice(0, 4)), ("waves 4-7 (MFMA, then transform)", names_hi, slice(4, 8))):
    m = d[:, sl, :].reshape(-1, 7)
    print(grp)
    for i, n in enumerate(names):
        col = m[:, i]
        print(f"  {n:16s} median {col.median():8.0f}  mean {col.mean():8.0f}  p10 {col.quantile(0.1):8.0f}  p90 {col.quantile(0.9):8.0f} cycles")
    tot = (t[:, sl, 7] - t[:, sl, 0]).reshape(-1)
    print(f"  chunk total      median {tot.median():8.0f}  (ideal MFMA time of a chunk per SIMD: 6144 cycles)")
