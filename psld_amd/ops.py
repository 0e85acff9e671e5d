"""Thin Python front-ends of the C ABI: torch tensors in (as device pointers on the current HIP
stream), nothing else.  PyTorch supplies memory and streams only; every computation below runs in
libpsld_hip.so.  All tensors must be contiguous CUDA(ROCm) tensors.
"""
from __future__ import annotations

import ctypes as C
import functools
import os
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import Epilogue, EmCoeffs, SdeParams, SscsCoeffs, check

Tensor = torch.Tensor
INV_SQRT2 = float(1.0 / np.sqrt(2.0))


def lib():
    return _lib.load()


_stream_cache = []   # stack of raw HIP stream handles pinned by stream_scope()


def _stream() -> int:
    if _stream_cache:
        return _stream_cache[-1]
    return torch.cuda.current_stream().cuda_stream


class stream_scope:
    """Pins the raw handle of torch's current stream for the duration of a forward / backward pass so
    that each of the ~1400 launches does not pay a `torch.cuda.current_stream()` lookup.  Nested scopes
    (side streams) push their own handle."""

    def __enter__(self):
        _stream_cache.append(torch.cuda.current_stream().cuda_stream)
        return self

    def __exit__(self, *exc):
        _stream_cache.pop()
        return False


def _p(t: Optional[Tensor]):
    if t is None:
        return None
    return t.data_ptr()


def _chk(t: Tensor, dtype=torch.float32):
    if not t.is_cuda:
        raise RuntimeError("psld_amd kernels need device tensors (no CPU fallback)")
    if t.dtype != dtype or not t.is_contiguous():
        raise RuntimeError(f"expected contiguous {dtype} tensor, got {t.dtype} contiguous={t.is_contiguous()}")
    return t


def epilogue(alpha: float = 1.0, bias: Optional[Tensor] = None, rowbias: Optional[Tensor] = None,
             rows_per_img: int = 1, residual: Optional[Tensor] = None, ld_residual: int = 0,
             residual_stride_batch: int = 0, out_scale: float = 1.0, accumulate: bool = False,
             ld_rowbias: int = 0, gn_part: Optional[Tensor] = None, gn_hw: int = 0) -> Epilogue:
    """``gn_part`` (limb kernels only; see gn_part_buffer): GroupNorm partial sums of the output as a by-product."""
    e = Epilogue()
    e.alpha = alpha
    e.bias = _p(bias)
    e.rowbias = _p(rowbias)
    e.ld_rowbias = (ld_rowbias or rowbias.shape[-1]) if rowbias is not None else 0
    e.rows_per_img = rows_per_img
    e.residual = _p(residual)
    e.ld_residual = ld_residual
    e.residual_stride_batch = residual_stride_batch
    e.out_scale = out_scale
    e.accumulate = 1 if accumulate else 0
    e.gn_part = _p(gn_part)
    e.gn_hw = gn_hw if gn_part is not None else 0
    e.gn_fine = getattr(gn_part, "fine_width", 8) if gn_part is not None else 0
    e._keep = (bias, rowbias, residual, gn_part)  # the struct holds raw pointers: keep the tensors alive
    return e


# ------------------------------------------------------------------------------------------------
# MFMA tile engine
# ------------------------------------------------------------------------------------------------
def gemm_raw(ta: int, tb: int, M: int, N: int, K: int, A: Tensor, lda: int, sa: int, B: Tensor, ldb: int,
             sb: int, Cc: Tensor, ldc: int, sc: int, batch: int = 1, epi: Optional[Epilogue] = None,
             a_off: int = 0, b_off: int = 0, c_off: int = 0):
    """C = epilogue(op(A) op(B)); offsets are in elements."""
    check(lib().psld_gemm_f32(ta, tb, M, N, K, A.data_ptr() + 4 * a_off, lda, sa, B.data_ptr() + 4 * b_off, ldb, sb,
                              Cc.data_ptr() + 4 * c_off, ldc, sc, batch,
                              C.byref(epi) if epi is not None else None, _stream()), "psld_gemm_f32")


def linear(x: Tensor, w: Tensor, bias: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
    """y[M,N] = x[M,K] w[N,K]^T + bias  (nn.Linear / 1x1 conv with OI weights)."""
    M, K = x.shape[0], x.shape[-1]
    N = w.shape[0]
    if out is None:
        out = torch.empty((M, N), device=x.device, dtype=torch.float32)
    gemm_raw(0, 1, M, N, K, x, K, 0, w, K, 0, out, N, 0, 1, epilogue(bias=bias) if bias is not None else None)
    return out


def gemm_tn_splitk(M: int, N: int, K: int, A: Tensor, lda: int, B: Tensor, ldb: int, slabs: Tensor, nsplit: int):
    check(lib().psld_gemm_tn_splitk_f32(M, N, K, A.data_ptr(), lda, B.data_ptr(), ldb, slabs.data_ptr(), nsplit,
                                        _stream()), "psld_gemm_tn_splitk_f32")


def conv2d_nhwc(x1: Tensor, x2: Optional[Tensor], w_ohwi: Tensor, cout: int, kh: int, kw: int, stride: int, pad: int,
                tstride: int, oh: int, ow: int, y: Tensor, epi: Optional[Epilogue] = None, ldy: Optional[int] = None):
    b, ih, iw, c1 = x1.shape
    c2 = x2.shape[-1] if x2 is not None else 0
    ws, wsb = None, 0
    if b * oh * ow <= 16384:     # only small grids ever split K (see psld_conv2d_nhwc_ws_f32)
        wsb = lib().psld_conv2d_workspace_bytes(b, oh, ow, cout)
        ws = workspace(wsb, x1.device).data_ptr()
    check(lib().psld_conv2d_nhwc_ws_f32(x1.data_ptr(), c1, _p(x2), c2, b, ih, iw, w_ohwi.data_ptr(), cout, kh, kw,
                                        stride, pad, tstride, oh, ow, y.data_ptr(), ldy if ldy is not None else cout,
                                        C.byref(epi) if epi is not None else None, ws, wsb, _stream()),
          "psld_conv2d_nhwc_f32")


def set_math_mode(mode: str):
    """'f32' (v_mfma_f32_32x32x2_f32) or 'bf16x6' (three-limb bf16 MFMA, fp32-equivalent); process-wide."""
    check(lib().psld_set_math_mode({"f32": 0, "bf16x6": 1}[mode]), "psld_set_math_mode")


def math_mode() -> str:
    return ("f32", "bf16x6")[lib().psld_get_math_mode()]


@functools.lru_cache(maxsize=None)
def conv3x3_split_supported(c1: int, c2: int, b: int, h: int, w: int, cout: int) -> bool:
    return bool(lib().psld_conv3x3_split_supported(c1, c2, b, h, w, cout))


def conv3x3_frag(w_oihw: Tensor, dgrad: bool, out: Optional[Tensor] = None) -> Tensor:
    """Pre-split 3x3 weights into bf16 limb fragments (uint8 buffer of psld_conv3x3_frag_bytes)."""
    co, ci = w_oihw.shape[0], w_oihw.shape[1]
    if out is None:
        out = torch.empty(lib().psld_conv3x3_frag_bytes(co, ci), dtype=torch.uint8, device=w_oihw.device)
    check(lib().psld_pack_conv3x3_frag(w_oihw.data_ptr(), out.data_ptr(), co, ci, int(dgrad), _stream()),
          "psld_pack_conv3x3_frag")
    return out


def conv3x3_frag_entry(w_oihw: Tensor, dgrad: bool, out: Tensor):
    """Table entry (without the running index) of ``pack_frag_batch`` equivalent to conv3x3_frag(w, dgrad, out)."""
    co, ci = w_oihw.shape[0], w_oihw.shape[1]
    if dgrad:
        return [w_oihw.data_ptr(), out.data_ptr(), ci, co, 9 | (1 << 32), 9, ci * 9]
    return [w_oihw.data_ptr(), out.data_ptr(), co, ci, 9, ci * 9, 9]


def pack_frag_batch(table: Tensor, entries: int, total_items: int):
    """table rows: conv3x3_frag_entry(...) + [first work item]; a tensor has cout*cin/8 work items."""
    check(lib().psld_pack_frag_batch(table.data_ptr(), entries, total_items, _stream()), "psld_pack_frag_batch")


def conv3x3_split(x1: Tensor, x2: Optional[Tensor], wfrag: Tensor, cout: int, y: Tensor,
                  epi: Optional[Epilogue] = None, ldy: Optional[int] = None):
    b, h, w, c1 = x1.shape
    c2 = x2.shape[-1] if x2 is not None else 0
    ws, wsb = None, 0
    if b * h * w <= 32768:
        wsb = lib().psld_conv2d_workspace_bytes(b, h, w, cout)
        ws = workspace(wsb, x1.device).data_ptr()
    if isinstance(x1, LimbPlanes):       # pre-split input(s): the LDS-DMA kernel
        assert x2 is None or isinstance(x2, LimbPlanes)
        check(lib().psld_conv3x3_limb_f32(x1.data_ptr(), c1, x2.data_ptr() if x2 is not None else None, c2, b, h, w,
                                          wfrag.data_ptr(), cout, y.data_ptr(), ldy if ldy is not None else cout,
                                          C.byref(epi) if epi is not None else None, ws, wsb, _stream()),
              "psld_conv3x3_limb_f32")
        return
    check(lib().psld_conv3x3_split_f32(x1.data_ptr(), c1, _p(x2), c2, b, h, w, wfrag.data_ptr(), cout, y.data_ptr(),
                                       ldy if ldy is not None else cout, C.byref(epi) if epi is not None else None,
                                       ws, wsb, _stream()), "psld_conv3x3_split_f32")


@functools.lru_cache(maxsize=None)
def conv3x3_wino_supported(c1: int, c2: int, b: int, h: int, w: int, cout: int) -> bool:
    return bool(lib().psld_conv3x3_wino_supported(c1, c2, b, h, w, cout))


def conv3x3_wino_frag(w_oihw: Tensor, dgrad: bool, out: Optional[Tensor] = None) -> Tensor:
    """3x3 weights -> Winograd-transformed (G g G^T) bf16 limb fragments (uint8 buffer of psld_conv3x3_wino_frag_bytes)."""
    co, ci = w_oihw.shape[0], w_oihw.shape[1]
    if out is None:
        out = torch.empty(lib().psld_conv3x3_wino_frag_bytes(co, ci), dtype=torch.uint8, device=w_oihw.device)
    check(lib().psld_pack_conv3x3_wino(w_oihw.data_ptr(), out.data_ptr(), co, ci, int(dgrad), _stream()),
          "psld_pack_conv3x3_wino")
    return out


def conv3x3_wino_frag_entry(w_oihw: Tensor, dgrad: bool, out: Tensor):
    """Table entry (without the running index) of ``pack_wino_batch`` equivalent to conv3x3_wino_frag(w, dgrad, out)."""
    co, ci = w_oihw.shape[0], w_oihw.shape[1]
    if dgrad:
        return [w_oihw.data_ptr(), out.data_ptr(), ci, co, 1, 9, ci * 9]
    return [w_oihw.data_ptr(), out.data_ptr(), co, ci, 0, ci * 9, 9]


def pack_wino_batch(table: Tensor, entries: int, total_items: int):
    """table rows: conv3x3_wino_frag_entry(...) + [first work item]; a tensor has cout*cin/8 work items."""
    check(lib().psld_pack_wino_batch(table.data_ptr(), entries, total_items, _stream()), "psld_pack_wino_batch")


_WINO_MODE = None     # 0: never, 1: where it pays (default), 2: wherever the kernel takes the shape (tests)


def set_winograd(mode: Optional[int]):
    """Override ``PSLD_WINOGRAD`` for this process (None: back to the environment); see conv3x3_wino_wanted."""
    global _WINO_MODE
    _WINO_MODE = mode
    conv3x3_wino_wanted.cache_clear()


@functools.lru_cache(maxsize=None)
def conv3x3_wino_wanted(c1: int, c2: int, b: int, h: int, w: int, cout: int, split_ok: bool = False) -> bool:
    """Policy (limb-MFMA math mode only - the caller checks that): does a 3x3 stride-1 convolution of this shape run in
    Winograd F(2x2, 3x3) form?  By default (``PSLD_WINOGRAD=1``) when the kernel takes the shape and the launch is at least
    ONE full round of its one-workgroup-per-CU grid - counting (``split_ok``: what the executor passes) the workgroups its
    channel chunks can be split over (conv3x3_wino / conv3x3_wino_gn with allow_split=True).  ``PSLD_WINOGRAD=0`` restores the direct limb kernels everywhere, ``=2`` takes every supported
    shape (parity tests at small batches).  The environment is read once; set_winograd() overrides it.

    History of the threshold: rounds 3-5 asked for 384 workgroups - at 256 (32x32 level, batch 16) the step was 3 % slower than
    with the direct kernels, because a workgroup per CU left no room for the direct weight-gradient kernels that small batches
    overlap on the side stream.  Round 6 moved the weight gradients into the Winograd domain (one workgroup per CU themselves)
    and split small launches over their chunks; re-measured on MI355X (profiles/r06/ab_small_batches_*.txt): every shape in
    Winograd form against the old rule +16 % at B=16, +4 % at B=32, +8 % at B=64, +2 % at B=128 (the 8x8 level)."""
    mode = _WINO_MODE if _WINO_MODE is not None else int(os.environ.get("PSLD_WINOGRAD", "1"))
    if mode == 0 or not conv3x3_wino_supported(c1, c2, b, h, w, cout):
        return False
    if mode == 2:
        return True
    tiles = (b * h * w // 128) * (cout // 128)
    ks = int(lib().psld_conv3x3_wino_ksplit(c1, c2, b, h, w, cout)) if split_ok else 1
    return tiles * ks >= 256


@functools.lru_cache(maxsize=None)
def conv3x3_wino_gn_supported(c1: int, c2: int, b: int, h: int, w: int, cout: int) -> bool:
    return bool(lib().psld_conv3x3_wino_gn_supported(c1, c2, b, h, w, cout))


_fused_gn_override: Optional[int] = None


def set_fused_gn(mode: Optional[int]):
    """Override ``PSLD_FUSED_GN`` for this process (None: back to the environment); see conv3x3_wino_gn_wanted."""
    global _fused_gn_override
    _fused_gn_override = mode
    conv3x3_wino_gn_wanted.cache_clear()


@functools.lru_cache(maxsize=None)
def conv3x3_wino_gn_wanted(c1: int, c2: int, b: int, h: int, w: int, cout: int) -> bool:
    """Should an inference-forward GroupNorm + SiLU + conv3x3 run as ONE launch (psld_conv3x3_wino_gn_f32)?  Only where the
    convolution runs in Winograd form anyway; ``PSLD_FUSED_GN`` = 0: never, 1 (default): maps of at least 32x32, where the
    pair measured 3-5 % shorter than apply pass + convolution at sampling batch sizes (on 16x16 the fused launch's extra
    vector work costs what the apply pass saved: profiles/r04/wino_fused_gn.txt), 2: wherever the kernel takes the shape
    (parity tests)."""
    mode = _fused_gn_override if _fused_gn_override is not None else int(os.environ.get("PSLD_FUSED_GN", "1"))
    if mode == 0 or not conv3x3_wino_wanted(c1, c2, b, h, w, cout) or not conv3x3_wino_gn_supported(c1, c2, b, h, w, cout):
        return False
    return mode == 2 or h * w >= 1024


def conv3x3_wino_gn(x1: Tensor, st1: "GNStats", x2: Optional[Tensor], st2: Optional["GNStats"], act: bool, ufrag: Tensor,
                    cout: int, y: Tensor, epi: Optional[Epilogue] = None, allow_split: bool = False):
    """conv3x3_wino applied to act(GroupNorm(x)) with the apply pass inside the kernel's input staging: x1 / x2 are the
    raw tensors, st1 / st2 their statistics (gn_stats / gn_stats_from_part).  Inference forward only (no dropout, the
    activated tensor is not kept)."""
    b, h, w, c1 = x1.shape
    c2 = x2.shape[-1] if x2 is not None else 0
    wsb = conv3x3_wino_ws_bytes(c1, c2, b, h, w, cout) if allow_split else 0
    if wsb:       # a small grid: channel chunks split like conv3x3_wino(allow_split=True)
        ws = workspace(wsb, x1.device)
        check(lib().psld_conv3x3_wino_gn_ws_f32(x1.data_ptr(), c1, st1.scale.data_ptr(), st1.shift.data_ptr(), _p(x2), c2,
                                                st2.scale.data_ptr() if st2 is not None else None,
                                                st2.shift.data_ptr() if st2 is not None else None, 1 if act else 0, b, h, w,
                                                ufrag.data_ptr(), cout, y.data_ptr(), y.shape[-1],
                                                C.byref(epi) if epi is not None else None, ws.data_ptr(), wsb, _stream()),
              "psld_conv3x3_wino_gn_ws_f32")
        return
    check(lib().psld_conv3x3_wino_gn_f32(x1.data_ptr(), c1, st1.scale.data_ptr(), st1.shift.data_ptr(), _p(x2), c2,
                                         st2.scale.data_ptr() if st2 is not None else None,
                                         st2.shift.data_ptr() if st2 is not None else None, 1 if act else 0, b, h, w,
                                         ufrag.data_ptr(), cout, y.data_ptr(), y.shape[-1],
                                         C.byref(epi) if epi is not None else None, _stream()),
          "psld_conv3x3_wino_gn_f32")


def conv3x3_wino(x1: Tensor, x2: Optional[Tensor], ufrag: Tensor, cout: int, y: Tensor,
                 epi: Optional[Epilogue] = None, ldy: Optional[int] = None, allow_split: bool = False):
    """conv3x3_split in Winograd F(2x2, 3x3) form (fp32 NHWC input(s), fragments of conv3x3_wino_frag).  ``allow_split``: a launch
    whose grid leaves CUs idle may split its channel chunks over workgroups (another summation order than the unsplit launch;
    the executor always asks for it - conv3x3_wino_gn splits the same way, so the fused and unfused inference forwards stay
    bitwise equal)."""
    b, h, w, c1 = x1.shape
    c2 = x2.shape[-1] if x2 is not None else 0
    wsb = conv3x3_wino_ws_bytes(c1, c2, b, h, w, cout) if allow_split else 0
    if wsb:
        # a small grid (the 8x8 level at training batches): channel chunks split over workgroups, one reduction + epilogue pass
        # (which also forms the GroupNorm partial sums an epilogue asks for)
        ws = workspace(wsb, x1.device)
        check(lib().psld_conv3x3_wino_ws_f32(x1.data_ptr(), c1, _p(x2), c2, b, h, w, ufrag.data_ptr(), cout, y.data_ptr(),
                                             ldy if ldy is not None else cout, C.byref(epi) if epi is not None else None,
                                             ws.data_ptr(), wsb, _stream()), "psld_conv3x3_wino_ws_f32")
        return
    check(lib().psld_conv3x3_wino_f32(x1.data_ptr(), c1, _p(x2), c2, b, h, w, ufrag.data_ptr(), cout, y.data_ptr(),
                                      ldy if ldy is not None else cout, C.byref(epi) if epi is not None else None,
                                      _stream()), "psld_conv3x3_wino_f32")


@functools.lru_cache(maxsize=None)
def conv3x3_wino_ws_bytes(c1: int, c2: int, b: int, h: int, w: int, cout: int) -> int:
    """Workspace of the split-chunk route of conv3x3_wino for this shape (0: the launch fills the chip by itself)."""
    return int(lib().psld_conv3x3_wino_ws_bytes(c1, c2, b, h, w, cout))


class LimbPlanes:
    """An NHWC activation stored as bf16 limb planes [rows][c/32][3][32] (include/psld_hip.h): ``t`` is the raw int16
    storage, ``shape`` the logical (b, h, w, c)."""
    __slots__ = ("t", "shape")

    def __init__(self, shape, device=None, t: Optional[Tensor] = None):
        self.shape = tuple(shape)
        b, h, w, c = self.shape
        assert c % 32 == 0, c
        self.t = t if t is not None else torch.empty((b * h * w * c * 3,), device=device, dtype=torch.int16)

    @property
    def device(self):
        return self.t.device

    def data_ptr(self):
        return self.t.data_ptr()

    def record_stream(self, s):
        self.t.record_stream(s)


def f32_to_limb(x: Tensor, out: Optional[LimbPlanes] = None) -> LimbPlanes:
    b, h, w, c = x.shape
    out = out if out is not None else LimbPlanes(x.shape, x.device)
    check(lib().psld_f32_to_limb(_chk(x).data_ptr(), b * h * w, c, out.data_ptr(), _stream()), "psld_f32_to_limb")
    return out


def limb_to_f32(x: LimbPlanes) -> Tensor:
    b, h, w, c = x.shape
    out = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    check(lib().psld_limb_to_f32(x.data_ptr(), b * h * w, c, out.data_ptr(), _stream()), "psld_limb_to_f32")
    return out


@functools.lru_cache(maxsize=None)
def gemm_split_supported(k1: int, k2: int, m: int, n: int) -> bool:
    return bool(lib().psld_gemm_split_supported(k1, k2, m, n))


def gemm_frag_bytes(n: int, k: int) -> int:
    return int(lib().psld_gemm_frag_bytes(n, k))


def gemm_frag(b: Tensor, n: int, k: int, stride_n: int, stride_k: int, out: Optional[Tensor] = None) -> Tensor:
    """Limb fragments of the [n][k] matrix whose element (i, j) sits at b.flatten()[i*stride_n + j*stride_k]."""
    if out is None:
        out = torch.empty(lib().psld_gemm_frag_bytes(n, k), dtype=torch.uint8, device=b.device)
    check(lib().psld_pack_gemm_frag(b.data_ptr(), out.data_ptr(), n, k, stride_n, stride_k, _stream()),
          "psld_pack_gemm_frag")
    return out


def gemm_split(a1: Tensor, a2: Optional[Tensor], m: int, bfrag: Tensor, n: int, y: Tensor,
               epi: Optional[Epilogue] = None, ldy: Optional[int] = None):
    """y[m][n] = epilogue(concat(a1, a2) @ B^T) on the bf16 limb kernels; a1 / a2 are [m][k1] / [m][k2] contiguous."""
    k1 = a1.shape[-1]
    k2 = a2.shape[-1] if a2 is not None else 0
    ws, wsb = None, 0
    if m <= 32768:
        wsb = 8 * m * n * 4
        ws = workspace(wsb, a1.device).data_ptr()
    check(lib().psld_gemm_split_f32(a1.data_ptr(), k1, _p(a2), k2, m, bfrag.data_ptr(), n, y.data_ptr(),
                                    ldy if ldy is not None else n, C.byref(epi) if epi is not None else None,
                                    ws, wsb, _stream()), "psld_gemm_split_f32")


@functools.lru_cache(maxsize=None)
def conv3x3_wgrad_split_supported(cout: int, cin: int, b: int, h: int, w: int) -> bool:
    return bool(lib().psld_conv3x3_wgrad_split_supported(cout, cin, b, h, w))


@functools.lru_cache(maxsize=None)
def gemm_tn_split_supported(m: int, n: int, k: int) -> bool:
    return bool(lib().psld_gemm_tn_split_supported(m, n, k))


def gemm_tn_split(M: int, N: int, K: int, A: Tensor, lda: int, B: Tensor, ldb: int, slabs: Tensor, ldc: int, nsplit: int,
                  B2: Optional[Tensor] = None, ldb2: int = 0, N2: int = 0):
    """slabs[s][M][ldc] = per-K-range partial sums of A^T [B | B2] on the bf16 limb kernel (A: [K][M] rows of lda,
    B: [K][N] rows of ldb, optional B2: [K][N2] rows of ldb2 filling columns N .. N+N2)."""
    check(lib().psld_gemm_tn_split_f32(M, N, K, A.data_ptr(), lda, B.data_ptr(), ldb, _p(B2), ldb2, N2, slabs.data_ptr(),
                                       ldc, nsplit, _stream()), "psld_gemm_tn_split_f32")


@functools.lru_cache(maxsize=None)
def bgemm_split_supported(ta: int, tb: int, m: int, n: int, k: int) -> bool:
    return bool(lib().psld_bgemm_split_supported(ta, tb, m, n, k))


def bgemm_split(ta: int, tb: int, M: int, N: int, K: int, A: Tensor, lda: int, sa: int, B: Tensor, ldb: int, sb: int,
                Cc: Tensor, ldc: int, sc: int, batch: int = 1, alpha: float = 1.0):
    """gemm_raw's operand conventions on the limb kernel, both operands split in the kernel (alpha is the only epilogue)."""
    check(lib().psld_bgemm_split_f32(ta, tb, M, N, K, A.data_ptr(), lda, sa, B.data_ptr(), ldb, sb, Cc.data_ptr(), ldc, sc,
                                     batch, alpha, _stream()), "psld_bgemm_split_f32")


@functools.lru_cache(maxsize=None)
def attn_fwd_supported(hw: int, c: int) -> bool:
    return bool(lib().psld_attn_fwd_split_supported(hw, c))


def attn_fwd(q: Tensor, k: Tensor, v: Tensor, ld: int, batch: int, hw: int, c: int, scale: float, out: Tensor,
             p: Optional[Tensor] = None):
    """out[b][i] = sum_j softmax_j(scale q[b][i].k[b][j]) v[b][j] in one kernel (csrc/attention.hip); q / k / v may be
    column slices of one buffer (common row stride ``ld``).  ``p`` ([batch, hw, hw]): also write the probabilities."""
    check(lib().psld_attn_fwd_split_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), ld, batch, hw, c, float(scale),
                                        out.data_ptr(), c, _p(p), _stream()), "psld_attn_fwd_split_f32")


@functools.lru_cache(maxsize=None)
def conv3x3_wgrad_split_cout_tile(cout: int) -> int:
    return lib().psld_conv3x3_wgrad_split_cout_tile(cout)


def conv3x3_wgrad_split(dy: Tensor, cout: int, x: Tensor, slabs: Tensor, cin_total: int, col0: int, nsplit: int,
                        x2: Optional[Tensor] = None):
    b, h, w, cin = x.shape
    cin2 = x2.shape[-1] if x2 is not None else 0
    if isinstance(x, LimbPlanes):         # x pre-split (GroupNorm's apply pass wrote limb planes): no split for it
        assert x2 is None or isinstance(x2, LimbPlanes)
        check(lib().psld_conv3x3_wgrad_xlimb_f32(dy.data_ptr(), cout, cout, x.data_ptr(), cin,
                                                 x2.data_ptr() if x2 is not None else None, cin2, b, h, w,
                                                 slabs.data_ptr(), cin_total, col0, nsplit, _stream()),
              "psld_conv3x3_wgrad_xlimb_f32")
        return
    check(lib().psld_conv3x3_wgrad_split_f32(dy.data_ptr(), cout, cout, x.data_ptr(), cin, _p(x2), cin2, b, h, w,
                                             slabs.data_ptr(), cin_total, col0, nsplit, _stream()),
          "psld_conv3x3_wgrad_split_f32")


@functools.lru_cache(maxsize=None)
def conv3x3_wgrad_wino_supported(cout: int, cin: int, cin2: int, b: int, h: int, w: int) -> bool:
    return bool(lib().psld_conv3x3_wgrad_wino_supported(cout, cin, cin2, b, h, w))


_WGRAD_WINO_MODE: Optional[int] = None


def set_wgrad_winograd(mode: Optional[int]):
    """Override ``PSLD_WGRAD_WINOGRAD`` for this process (None: back to the environment); see conv3x3_wgrad_wino_wanted."""
    global _WGRAD_WINO_MODE
    _WGRAD_WINO_MODE = mode
    conv3x3_wgrad_wino_wanted.cache_clear()


@functools.lru_cache(maxsize=None)
def conv3x3_wgrad_wino_wanted(cout: int, cin: int, cin2: int, b: int, h: int, w: int) -> bool:
    """Policy (limb-MFMA math mode, fp32 x - the caller checks both): does the weight gradient of a 3x3 stride-1 convolution
    run in the Winograd domain (wgrad_wino.hip)?  ``PSLD_WGRAD_WINOGRAD=1`` (default): when the kernel takes the shape and
    every K split holds at least 8 K tiles (at B=128 that is every layer of the 32x32 / 16x16 levels, and of the 8x8 level now
    that its activations are fp32: forced on there the step gains another 0.6 %) (measured on MI355X, tools/bench_wwgrad.py: x1.40-1.47 on the 32x32 level, x1.23-
    1.37 on the 16x16 level at B=128; short K ranges are prologue + epilogue); ``=0``: the direct limb kernels everywhere;
    ``=2``: every supported shape (parity tests at small batches)."""
    mode = _WGRAD_WINO_MODE if _WGRAD_WINO_MODE is not None else int(os.environ.get("PSLD_WGRAD_WINOGRAD", "1"))
    if mode == 0 or not conv3x3_wgrad_wino_supported(cout, cin, cin2, b, h, w):
        return False
    if mode == 2:
        return True
    ns, _ = conv3x3_wgrad_wino_plan(cout, cin + cin2, b, h, w)
    return (b * h * w // 128) // ns >= 8


@functools.lru_cache(maxsize=None)
def conv3x3_wgrad_wino_plan(cout: int, cin_total: int, b: int, h: int, w: int):
    """(K splits, workspace bytes) of the Winograd-domain weight gradient for this shape."""
    ns = int(lib().psld_conv3x3_wgrad_wino_nsplit(cout, cin_total, b, h, w))
    return ns, int(lib().psld_conv3x3_wgrad_wino_ws_bytes(cout, cin_total, ns))


def conv3x3_wgrad_wino(dy: Tensor, cout: int, x: Tensor, dw: Tensor, x2: Optional[Tensor] = None, accumulate: bool = False,
                       slabs: Optional[Tensor] = None, nsplit: Optional[int] = None, alpha: float = 1.0):
    """dw[cout][cin (+ cin2)][3][3] (OIHW, contiguous) = (or +=) the weight gradient of the 3x3 stride-1 convolution, formed in
    the Winograd F(2x2, 3x3) domain (psld_conv3x3_wgrad_wino_f32: 16 limb products per 2x2 tile instead of 36)."""
    b, h, w, cin = x.shape
    cin2 = x2.shape[-1] if x2 is not None else 0
    ns, wsb = conv3x3_wgrad_wino_plan(cout, cin + cin2, b, h, w)
    if nsplit is not None:
        ns, wsb = nsplit, int(lib().psld_conv3x3_wgrad_wino_ws_bytes(cout, cin + cin2, nsplit))
    if slabs is None:
        slabs = workspace(wsb, x.device)
    assert slabs.numel() * slabs.element_size() >= wsb and dw.is_contiguous() and dw.numel() == cout * (cin + cin2) * 9
    check(lib().psld_conv3x3_wgrad_wino_f32(dy.data_ptr(), cout, cout, x.data_ptr(), cin, _p(x2), cin2, b, h, w,
                                            slabs.data_ptr(), ns, dw.data_ptr(), 1 if accumulate else 0, float(alpha), _stream()),
          "psld_conv3x3_wgrad_wino_f32")


def conv2d_wgrad_nhwc(dy: Tensor, cout: int, x: Tensor, kh: int, kw: int, stride: int, pad: int, oh: int, ow: int,
                      slabs: Tensor, cin_total: int, col0: int, nsplit: int):
    b, ih, iw, cin = x.shape
    check(lib().psld_conv2d_wgrad_nhwc_f32(dy.data_ptr(), cout, cout, x.data_ptr(), cin, b, ih, iw, kh, kw, stride,
                                           pad, oh, ow, slabs.data_ptr(), cin_total, col0, nsplit, _stream()),
          "psld_conv2d_wgrad_nhwc_f32")


def reduce_slabs(slabs: Tensor, nsplit: int, n: int, out: Tensor, layout: int = 0, cout: int = 0, taps: int = 0,
                 cin: int = 0, alpha: float = 1.0):
    check(lib().psld_reduce_slabs_f32(slabs.data_ptr(), nsplit, n, out.data_ptr(), layout, cout, taps, cin, alpha,
                                      _stream()), "psld_reduce_slabs_f32")


def pack_ohwi(w_oihw: Tensor, out: Tensor):
    co, ci = w_oihw.shape[0], w_oihw.shape[1]
    taps = w_oihw.shape[2] * w_oihw.shape[3]
    check(lib().psld_pack_oihw_to_ohwi_f32(w_oihw.data_ptr(), out.data_ptr(), co, ci, taps, _stream()), "pack_ohwi")


def pack_dgrad(w_oihw: Tensor, out: Tensor):
    co, ci = w_oihw.shape[0], w_oihw.shape[1]
    taps = w_oihw.shape[2] * w_oihw.shape[3]
    check(lib().psld_pack_oihw_to_dgrad_f32(w_oihw.data_ptr(), out.data_ptr(), co, ci, taps, _stream()), "pack_dgrad")


def nchw_to_nhwc(x: Tensor) -> Tensor:
    b, c, h, w = x.shape
    y = torch.empty((b, h, w, c), device=x.device, dtype=torch.float32)
    check(lib().psld_nchw_to_nhwc_f32(_chk(x).data_ptr(), y.data_ptr(), b, c, h * w, _stream()), "nchw_to_nhwc")
    return y


def nhwc_to_nchw(x: Tensor) -> Tensor:
    b, h, w, c = x.shape
    y = torch.empty((b, c, h, w), device=x.device, dtype=torch.float32)
    check(lib().psld_nhwc_to_nchw_f32(_chk(x).data_ptr(), y.data_ptr(), b, c, h * w, _stream()), "nhwc_to_nchw")
    return y


# ------------------------------------------------------------------------------------------------
# GroupNorm (+SiLU)
# ------------------------------------------------------------------------------------------------
_ws_cache = {}


def workspace(nbytes: int, device) -> Tensor:
    """Grow-only scratch buffer per (device, stream): kernels using it are stream-ordered."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), _stream())
    t = _ws_cache.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(max(int(nbytes), 1 << 20), device=device, dtype=torch.uint8)
        _ws_cache[key] = t
    return t


class GNStats:
    __slots__ = ("mean", "rstd", "scale", "shift")

    def __init__(self, b, g, c, device):
        self.mean = torch.empty((b, g), device=device, dtype=torch.float32)
        self.rstd = torch.empty((b, g), device=device, dtype=torch.float32)
        self.scale = torch.empty((b, c), device=device, dtype=torch.float32)
        self.shift = torch.empty((b, c), device=device, dtype=torch.float32)


@functools.lru_cache(maxsize=None)
def gn_groups(c: int) -> int:
    return min(c // 4, 32)


def gn_stats(x: Tensor, gamma: Tensor, beta: Tensor, eps: float = 1e-6, groups: Optional[int] = None) -> GNStats:
    """``groups``: override for a tensor that is one source of a concatenation (its share of the groups)."""
    b, h, w, c = x.shape
    g = groups if groups is not None else gn_groups(c)
    st = GNStats(b, g, c, x.device)
    ws = workspace(lib().psld_gn_workspace_bytes(b, h * w, c, g), x.device)
    check(lib().psld_gn_stats_nhwc_f32(x.data_ptr(), b, h * w, c, g, eps, gamma.data_ptr(), beta.data_ptr(),
                                       st.mean.data_ptr(), st.rstd.data_ptr(), st.scale.data_ptr(),
                                       st.shift.data_ptr(), ws.data_ptr(), _stream()), "psld_gn_stats_nhwc_f32")
    return st


def gn_part_width(c: int) -> int:
    """Channels per partial sum a limb kernel's epilogue leaves for a [.., c] output: 8, or 4 when the tensor's own groups
    are narrower (128 channels in 32 groups)."""
    return 8 if (c // gn_groups(c)) % 8 == 0 else 4


@functools.lru_cache(maxsize=None)
def gn_part_supported(b: int, hw: int, c: int) -> bool:
    """Can a limb kernel's epilogue produce the GroupNorm partial sums of its [b, hw, c] output?  (Whole 64-row
    runs per image and groups made of 4- or 8-channel fine groups.  Any grid: a launch too small to fill the chip splits
    its K range and finishes through a reduction + epilogue pass, which since round 6 forms the sums as well -
    conv_reduce_epilogue_gn_kernel; before, every GroupNorm behind such a launch - the whole 8x8 level at B=128 - paid a
    statistics pass over the tensor.)"""
    return hw % 64 == 0 and c % 128 == 0 and (c // gn_groups(c)) % 4 == 0


def gn_part_buffer(b: int, hw: int, c: int, device) -> Tensor:
    """[b][hw/64][c/w][2] float64: sum / sum of squares per 64-row run and w-channel fine group, w = gn_part_width(c)
    (carried as the tensor's ``fine_width`` attribute)."""
    w = gn_part_width(c)
    buf = torch.empty((b, hw // 64, c // w, 2), device=device, dtype=torch.float64)
    buf.fine_width = w
    return buf


def gn_stats_from_part(part: Tensor, shape, gamma: Tensor, beta: Tensor, eps: float = 1e-6,
                       groups: Optional[int] = None) -> GNStats:
    """gn_stats of a tensor of ``shape`` = (b, h, w, c) whose producer left the partial sums in ``part``."""
    b, h, w, c = shape
    g = groups if groups is not None else gn_groups(c)
    st = GNStats(b, g, c, part.device)
    check(lib().psld_gn_stats_from_partials_f32(part.data_ptr(), getattr(part, "fine_width", 8), b, h * w, c, g, eps,
                                                gamma.data_ptr(), beta.data_ptr(), st.mean.data_ptr(), st.rstd.data_ptr(),
                                                st.scale.data_ptr(), st.shift.data_ptr(), _stream()),
          "psld_gn_stats_from_partials_f32")
    return st


def gn_apply(x: Tensor, st: GNStats, act: bool, out: Optional[Tensor] = None, drop_p: float = 0.0,
             seed: int = 0, seed_dev: Optional[Tensor] = None) -> Tensor:
    """``seed_dev``: int64 device scalar added to ``seed`` by the kernel (the per-step part of the dropout seed)."""
    b, h, w, c = x.shape
    if out is None:
        out = torch.empty_like(x)
    check(lib().psld_gn_apply_nhwc_f32(x.data_ptr(), st.scale.data_ptr(), st.shift.data_ptr(), out.data_ptr(), b,
                                       h * w, c, 1 if act else 0, drop_p, seed, _p(seed_dev), _stream()),
          "psld_gn_apply_nhwc_f32")
    return out


def gn_apply_limb(x: Tensor, st: GNStats, act: bool, drop_p: float = 0.0, seed: int = 0,
                  seed_dev: Optional[Tensor] = None) -> LimbPlanes:
    """GroupNorm apply (+SiLU, dropout) writing bf16 limb planes for a 3x3 convolution to stage by LDS-DMA."""
    b, h, w, c = x.shape
    out = LimbPlanes(x.shape, x.device)
    check(lib().psld_gn_apply_limb_nhwc(x.data_ptr(), st.scale.data_ptr(), st.shift.data_ptr(), out.data_ptr(), b,
                                        h * w, c, 1 if act else 0, drop_p, seed, _p(seed_dev), _stream()),
          "psld_gn_apply_limb_nhwc")
    return out


def set_gn_bwd_kernel(kind: str):
    """"auto": the LDS-image kernel where its shape rules hold; "one_slab": the register-resident one-slab kernel for every
    shape (bitwise the same results; for tests and A/B runs)."""
    check(lib().psld_set_gn_bwd_kernel({"auto": 0, "one_slab": 1}[kind]), "psld_set_gn_bwd_kernel")


def get_gn_bwd_kernel() -> str:
    return ("auto", "one_slab")[lib().psld_get_gn_bwd_kernel()]


def gn_bwd(dy: Tensor, x: Tensor, st: GNStats, gamma: Tensor, beta: Tensor, act: bool, dx: Tensor,
           dgamma: Optional[Tensor] = None, dbeta: Optional[Tensor] = None, accumulate_dx: bool = False, drop_p: float = 0.0,
           seed: int = 0, groups: Optional[int] = None, add: Optional[Tensor] = None, add_scale: float = 1.0,
           seed_dev: Optional[Tensor] = None, sums: Optional[Tensor] = None, colsum_img: Optional[Tensor] = None,
           ld_img: int = 0) -> Tensor:
    """Backward of y = dropout(act(GN(x))).  ``add`` (same shape as x): dx additionally receives add_scale * add (gradient of
    a parallel identity branch).  Returns ``sums`` [b][2][c] (per image sum dz, sum dz * xhat): the parameter gradients are
    their sums over the batch - formed here (one more launch) when ``dgamma`` / ``dbeta`` are given, or by the caller for
    many layers at once (param_reduce_batch).  ``colsum_img`` ([b] rows, row stride ``ld_img`` or c; where
    gn_bwd_colsum_supported): the per-image column sums of the dx values this call stores - a bias / time-embedding
    gradient without a pass over dx."""
    b, h, w, c = x.shape
    g = groups if groups is not None else gn_groups(c)
    if sums is None:
        sums = torch.empty((b, 2, c), device=x.device, dtype=torch.float32)
    ws = workspace(lib().psld_gn_workspace_bytes(b, h * w, c, g), x.device)
    check(lib().psld_gn_bwd_nhwc_f32(dy.data_ptr(), x.data_ptr(), st.mean.data_ptr(), st.rstd.data_ptr(),
                                     gamma.data_ptr(), beta.data_ptr(), b, h * w, c, g, 1 if act else 0,
                                     drop_p, seed, _p(seed_dev), dx.data_ptr(), 1 if accumulate_dx else 0,
                                     _p(add), add_scale, sums.data_ptr(), _p(colsum_img),
                                     (ld_img or c) if colsum_img is not None else 0, ws.data_ptr(), _stream()),
          "psld_gn_bwd_nhwc_f32")
    if dgamma is not None or dbeta is not None:
        assert dgamma is not None and dbeta is not None
        param_reduce2(sums, sums.view(-1)[c:], b, 2 * c, c, dbeta, dgamma)
    return sums


@functools.lru_cache(maxsize=None)
def gn_bwd_colsum_supported(b: int, hw: int, c: int, groups: Optional[int] = None) -> bool:
    return bool(lib().psld_gn_bwd_colsum_supported(b, hw, c, groups if groups is not None else gn_groups(c)))


def gn_bwd_team_rows(b: int, hw: int, c: int, groups: Optional[int] = None) -> int:
    """Team size K (= rows per image of the sums) when the whole-row GroupNorm backward takes the shape, else 0 (not
    cached: the answer follows psld_set_gn_bwd_kernel)."""
    return int(lib().psld_gn_bwd_team_rows(b, hw, c, groups if groups is not None else gn_groups(c)))


def gn_bwd_team_wanted(b: int, hw: int, c: int, groups: Optional[int] = None, third: bool = True) -> int:
    """Policy: the team size when the whole-row kernel is the faster one, else 0.  Measured on MI355X (tools/bench_gnb_team.py,
    profiles/r05/ab_gnb_team.txt): with a third operand 102 / 100 / 129 against 117 / 112 / 148 us on 128x32x32x256 (4 sets
    per team); on smaller problems (fewer than two rounds of resident teams, or K = 4) the exchange costs more than the
    access pattern saves.  Maps above 32x32 (CelebA-64's first level) have no one-slab kernel - the alternative there is the
    three-pass form at 20+ bytes per element - so the team kernel also takes the calls without a third operand."""
    if b * hw * c < (1 << 25) or hw < 1024 or (hw == 1024 and not third):
        return 0
    return gn_bwd_team_rows(b, hw, c, groups)


_team_sync = {}      # device index -> zero-initialised slot buffer of the team kernels (one stream at a time uses it)


def gn_team_sync(device) -> Tensor:
    device = torch.device(device)
    key = device.index if device.index is not None else torch.cuda.current_device()
    t = _team_sync.get(key)
    if t is None:
        t = _team_sync[key] = torch.zeros(int(lib().psld_gn_bwd_team_sync_bytes()), device=device, dtype=torch.uint8)
    return t


def check_device_errors(device=None) -> None:
    """Raise if a device-side error word is set (today: a workgroup of the team GroupNorm backward gave up waiting for its
    team and wrote garbage rather than hang the queue).  A host read: call it where the host already waits for the device
    (the training loop's log line, before a checkpoint, after a timed region) - never per launch.  Allocates nothing."""
    if device is None:
        keys = list(_team_sync)
    else:
        device = torch.device(device)
        keys = [device.index if device.index is not None else torch.cuda.current_device()]
    for key in keys:
        t = _team_sync.get(key)
        if t is not None and int(t[:8].view(torch.int64).item()) != 0:
            raise RuntimeError(f"psld_amd: gn_bwd_team_kernel on cuda:{key} timed out waiting for a team member - the "
                               "gradients of this step are invalid (set PSLD_GN_BWD_PIPE=0 to take the one-slab kernels)")


def gn_team_errors(device) -> int:
    """Error word of the team kernels' slot buffer (non-zero: a workgroup gave up waiting for its team); a host read."""
    return int(gn_team_sync(device)[:8].view(torch.int64).item())


def gn_bwd_team(dy: Tensor, x: Tensor, st: GNStats, gamma: Tensor, beta: Tensor, act: bool, dx: Tensor,
                accumulate_dx: bool = False, drop_p: float = 0.0, seed: int = 0, groups: Optional[int] = None,
                add: Optional[Tensor] = None, add_scale: float = 1.0, seed_dev: Optional[Tensor] = None,
                sums: Optional[Tensor] = None, colsum_rows: Optional[Tensor] = None, ld_rows: int = 0) -> Tensor:
    """gn_bwd on whole rows by teams of resident workgroups (psld_gn_bwd_team_f32; shapes: gn_bwd_team_rows > 0).  Returns
    ``sums`` [b * K][2][c]; ``colsum_rows`` [b * K] rows: column sums of the stored dx per team member."""
    b, h, w, c = x.shape
    g = groups if groups is not None else gn_groups(c)
    k = gn_bwd_team_rows(b, h * w, c, g)
    assert k > 0
    if sums is None:
        sums = torch.empty((b * k, 2, c), device=x.device, dtype=torch.float32)
    check(lib().psld_gn_bwd_team_f32(dy.data_ptr(), x.data_ptr(), st.mean.data_ptr(), st.rstd.data_ptr(), gamma.data_ptr(),
                                     beta.data_ptr(), b, h * w, c, g, 1 if act else 0, drop_p, seed, _p(seed_dev), dx.data_ptr(),
                                     1 if accumulate_dx else 0, _p(add), add_scale, sums.data_ptr(), _p(colsum_rows),
                                     (ld_rows or c) if colsum_rows is not None else 0, gn_team_sync(x.device).data_ptr(),
                                     _stream()), "psld_gn_bwd_team_f32")
    return sums


def param_reduce2(src_a: Tensor, src_b: Optional[Tensor], rows: int, ld: int, c: int, dst_a: Tensor,
                  dst_b: Optional[Tensor], alpha: float = 1.0):
    """dst[col] = alpha * sum over rows of src[r * ld + col] for one or two sources of the same shape (one launch)."""
    check(lib().psld_param_reduce2_f32(src_a.data_ptr(), _p(src_b), rows, ld, c, dst_a.data_ptr(), _p(dst_b), alpha,
                                       _stream()), "psld_param_reduce2_f32")


def _f32_bits(v: float) -> int:
    return int(np.float32(v).view(np.uint32))


def param_job(src: Tensor, rows: int, ld: int, c: int, dst1: Tensor, dst2: Optional[Tensor] = None, alpha: float = 1.0,
              src_off: int = 0):
    """Table row (without the running block index) of ``param_reduce_batch``: dst1[col] (and dst2[col]) = alpha * sum over
    ``rows`` rows of src[src_off + r * ld + col], col < c."""
    return (src.data_ptr() + 4 * src_off, rows, ld, c, dst1.data_ptr(), dst2.data_ptr() if dst2 is not None else 0,
            _f32_bits(alpha))


byte_hook = None      # callable(family, algorithmic bytes): set by tools/hbm_in_situ.py for the table-driven launches


def param_reduce_batch(table: Tensor, jobs: int, blocks: int, nbytes: int = 0):
    """table rows: param_job(...) + (first 64-column block,); a job has ceil(c / 64) blocks."""
    if byte_hook is not None:
        byte_hook("param_reduce", nbytes)
    check(lib().psld_param_reduce_batch_f32(table.data_ptr(), jobs, blocks, _stream()), "psld_param_reduce_batch_f32")


@functools.lru_cache(maxsize=None)
def slab_units(n: int, layout: int, taps: int, cin: int) -> int:
    """Work units of a job of ``reduce_slabs_batch`` (0: the job does not qualify for the batched kernel)."""
    return int(lib().psld_reduce_slabs_batch_units(n, layout, taps, cin))


def slab_job(slabs: Tensor, nsplit: int, n: int, out: Tensor, layout: int = 0, taps: int = 1, cin: int = 1,
             alpha: float = 1.0):
    """Table row (without the running unit index and the unit count) of ``reduce_slabs_batch`` equivalent to reduce_slabs(...)."""
    assert slab_units(n, layout, taps, cin) > 0 and slabs.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0
    return (slabs.data_ptr(), nsplit, n, out.data_ptr(), layout, taps, cin, _f32_bits(alpha))


def reduce_slabs_batch(table: Tensor, jobs: int, units: int, nbytes: int = 0):
    """table rows: slab_job(...) + (first unit, units of the job = slab_units(...)).  ``nbytes``: algorithmic bytes of the
    jobs (sum of 4 n (nsplit + 1)), for a measurement hook (tools/hbm_in_situ.py) - the table itself lives on the device."""
    if byte_hook is not None:
        byte_hook("reduce_slabs", nbytes)
    check(lib().psld_reduce_slabs_batch_f32(table.data_ptr(), jobs, units, _stream()), "psld_reduce_slabs_batch_f32")


class Arena:
    """Bump allocator over ONE persistent device buffer: scratch whose ADDRESSES repeat from step to step (the batched
    reduction tables hold raw pointers and are cached by their contents; a hipGraph-captured step must see the addresses
    of its eager warm-up steps).  ``reset()`` rewinds; growing allocates a new buffer (the old one stays alive as long as
    views of it do).  Users order their accesses by stream like any other scratch."""

    def __init__(self, device, nbytes: int = 1 << 24):
        self.device = device
        self.buf = torch.empty(int(nbytes), device=device, dtype=torch.uint8)
        self.off = 0
        self.high = 0
        self.retired = []       # outgrown buffers: alive until the next reset (pending tables hold raw pointers into them)
        self.pinned = False     # a captured hipGraph holds raw pointers into buf: outgrown buffers then stay alive for good

    def reset(self):
        self.high = max(self.high, self.off)
        self.off = 0
        if not self.pinned:
            self.retired = []

    def alloc(self, nbytes: int) -> Tensor:
        nbytes = (int(nbytes) + 255) & ~255
        if self.off + nbytes > self.buf.numel():
            self.high = max(self.high, self.off)
            self.retired.append(self.buf)
            self.buf = torch.empty(max(2 * self.buf.numel(), self.off + nbytes, 2 * self.high), device=self.device, dtype=torch.uint8)
            self.off = 0
        out = self.buf[self.off:self.off + nbytes]
        self.off += nbytes
        return out

    def floats(self, *shape) -> Tensor:
        n = 1
        for d in shape:
            n *= int(d)
        return self.alloc(4 * n).view(torch.float32)[:n].view(*shape)


class TableCache:
    """Device copies of int64 job tables keyed by their contents (a step builds the same tables every time: the upload -
    a synchronous host-to-device copy - happens in the first steps only)."""

    def __init__(self, limit: int = 64):
        self.limit = limit
        self.tabs = {}
        self.pinned = False     # a captured hipGraph holds raw pointers to these tables: no eviction from then on

    def get(self, rows, device) -> Tensor:
        key = hash(tuple(rows))
        ent = self.tabs.get(key)
        if ent is not None and ent[0] == rows:
            return ent[1]
        if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("psld_amd: a job table is missing while a hipGraph is being captured (its upload is a synchronous "
                               "copy); run the eager warm-up steps on the same shapes first")
        if len(self.tabs) >= self.limit and not self.pinned:
            self.tabs.pop(next(iter(self.tabs)))
        t = torch.tensor(rows, dtype=torch.int64, device=device)
        self.tabs[key] = (list(rows), t)
        return t


# ------------------------------------------------------------------------------------------------
# FIR resampling
# ------------------------------------------------------------------------------------------------
def _pad4(pad):
    """(p0, p1) -> same pads on x and y like the reference's public wrapper (op/upfirdn2d.py:145-156);
    (x0, x1, y0, y1) as the pybind op itself takes them (op/upfirdn2d.cpp:12-22)."""
    if len(pad) == 2:
        return (pad[0], pad[1], pad[0], pad[1])
    return tuple(pad)


def upfirdn2d_out_size(in_h, in_w, kh, kw, up, down, pad):
    px0, px1, py0, py1 = _pad4(pad)
    return ((in_h * up + py0 + py1 - kh) // down + 1, (in_w * up + px0 + px1 - kw) // down + 1)


def upfirdn2d_raw(x: Tensor, kernel: np.ndarray, up: int, down: int, pad, layout: int,
                  out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    """layout 0: x is NCHW; 1: NHWC.  ``kernel`` is a host float32 [kh,kw] array."""
    k = np.ascontiguousarray(kernel, dtype=np.float32)
    kh, kw = k.shape
    px0, px1, py0, py1 = _pad4(pad)
    if layout == 0:
        b, c, h, w = x.shape
    else:
        b, h, w, c = x.shape
    oh, ow = upfirdn2d_out_size(h, w, kh, kw, up, down, pad)
    if out is None:
        shape = (b, c, oh, ow) if layout == 0 else (b, oh, ow, c)
        out = torch.empty(shape, device=x.device, dtype=torch.float32)
    check(lib().psld_upfirdn2d_f32(_chk(x).data_ptr(), out.data_ptr(), b, c, h, w, k.ctypes.data, kh, kw, up, up,
                                   down, down, px0, px1, py0, py1, layout, 1 if accumulate else 0,
                                   _stream()), "psld_upfirdn2d_f32")
    return out


def upfirdn2d_bwd_raw(gy: Tensor, kernel: np.ndarray, up: int, down: int, pad, in_hw: Tuple[int, int],
                      layout: int, out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    """Gradient w.r.t. the input: the same op with flipped kernel, up<->down and g_pad
    (op/upfirdn2d.py:31-42,111-116)."""
    k = np.ascontiguousarray(kernel, dtype=np.float32)
    kh, kw = k.shape
    in_h, in_w = in_hw
    px0, px1, py0, py1 = _pad4(pad)
    oh, ow = upfirdn2d_out_size(in_h, in_w, kh, kw, up, down, pad)
    gx0 = kw - px0 - 1
    gy0 = kh - py0 - 1
    gx1 = in_w * up - ow * down + px0 - up + 1
    gy1 = in_h * up - oh * down + py0 - up + 1
    return upfirdn2d_raw(gy, k[::-1, ::-1], down, up, (gx0, gx1, gy0, gy1), layout, out=out, accumulate=accumulate)


# ------------------------------------------------------------------------------------------------
# pointwise
# ------------------------------------------------------------------------------------------------
def axpby(a: Tensor, sa: float, b: Optional[Tensor], sb: float, out: Tensor, accumulate: bool = False):
    check(lib().psld_axpby_f32(a.data_ptr(), sa, _p(b), sb, out.data_ptr(), a.numel(), 1 if accumulate else 0,
                               _stream()), "psld_axpby_f32")
    return out


def silu(x: Tensor) -> Tensor:
    y = torch.empty_like(x)
    check(lib().psld_silu_f32(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "psld_silu_f32")
    return y


def silu_bwd(x: Tensor, dy: Tensor) -> Tensor:
    dx = torch.empty_like(x)
    check(lib().psld_silu_bwd_f32(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), _stream()), "psld_silu_bwd")
    return dx


def colsum(x: Tensor, ld: int, batch: int, hw: int, c: int, out: Tensor, alpha: float = 1.0):
    ws = workspace(lib().psld_colsum_workspace_bytes(batch, hw, c), x.device)
    check(lib().psld_colsum_f32(x.data_ptr(), ld, batch, hw, c, out.data_ptr(), alpha, ws.data_ptr(), _stream()),
          "psld_colsum_f32")
    return out


def bias_grad(x: Tensor, ld: int, batch: int, hw: int, c: int, out: Tensor, alpha: float = 1.0,
              per_image: Optional[Tensor] = None, ld_per_image: int = 0):
    """out[c] = alpha * column sums over (batch, hw); per_image[b*ld_per_image + c] (optional) = unscaled per-image sums."""
    ws = workspace(lib().psld_colsum_workspace_bytes(batch, hw, c), x.device)
    check(lib().psld_bias_grad_f32(x.data_ptr(), ld, batch, hw, c, _p(per_image), ld_per_image, out.data_ptr(), alpha,
                                   ws.data_ptr(), _stream()), "psld_bias_grad_f32")
    return out


def bias_grad_seg(x: Tensor, ld: int, batch: int, hw: int, outs: Sequence[Tensor], seg: int, alpha: float = 1.0):
    """outs[k][seg] = alpha * column sums over (batch, hw) of columns [k*seg, (k+1)*seg) of x, k = 0, 1, 2 (one pass)."""
    ws = workspace(lib().psld_colsum_workspace_bytes(batch, hw, 3 * seg), x.device)
    check(lib().psld_bias_grad_seg_f32(x.data_ptr(), ld, batch, hw, seg, outs[0].data_ptr(), outs[1].data_ptr(),
                                       outs[2].data_ptr(), alpha, ws.data_ptr(), _stream()), "psld_bias_grad_seg_f32")


def copy_batch(table: Tensor, entries: int, total4: int):
    """table rows: [src pointer, dst pointer, float4 count, first float4 index]."""
    check(lib().psld_copy_batch_f32(table.data_ptr(), entries, total4, _stream()), "psld_copy_batch_f32")


def im2col3x3_small(x: Tensor, oh: int, ow: int, stride: int, pad: int, flip: bool = False, ld_out: int = 64) -> Tensor:
    """[B*oh*ow, ld_out] im2col (column = channel*9 + tap, zero-padded) of a few-channel NHWC tensor."""
    b, ih, iw, c = x.shape
    out = torch.empty((b * oh * ow, ld_out), device=x.device, dtype=torch.float32)
    check(lib().psld_im2col3x3_small_f32(_chk(x).data_ptr(), b, ih, iw, c, oh, ow, stride, pad, int(flip), out.data_ptr(),
                                         ld_out, _stream()), "psld_im2col3x3_small_f32")
    return out


def im2col3x3(x: Tensor, stride: int, pad: int, oh: int, ow: int) -> Tensor:
    """[b*oh*ow][9*c] patch matrix of an NHWC tensor, K order (tap, channel)."""
    b, ih, iw, c = x.shape
    cols = torch.empty((b * oh * ow, 9 * c), device=x.device, dtype=torch.float32)
    check(lib().psld_im2col3x3_f32(x.data_ptr(), b, ih, iw, c, oh, ow, stride, pad, cols.data_ptr(), _stream()),
          "psld_im2col3x3_f32")
    return cols


def col2im3x3(dcols: Tensor, shape, stride: int, pad: int, oh: int, ow: int, out: Optional[Tensor] = None) -> Tensor:
    """Adjoint of im2col3x3: ``shape`` = (b, ih, iw, c) of the input tensor."""
    b, ih, iw, c = shape
    if out is None:
        out = torch.empty(tuple(shape), device=dcols.device, dtype=torch.float32)
    check(lib().psld_col2im3x3_f32(dcols.data_ptr(), b, ih, iw, c, oh, ow, stride, pad, out.data_ptr(), _stream()),
          "psld_col2im3x3_f32")
    return out


def conv3x3_fewout_supported(cin: int, cout: int) -> bool:
    return bool(lib().psld_conv3x3_fewout_supported(cin, cout))


def conv3x3_fewout(x: Tensor, w_ohwi: Tensor, bias: Optional[Tensor], cout: int, y: Tensor):
    """3x3 stride-1 pad-1 convolution with 3 / 6 output channels (dot-product kernel; weights packed OHWI)."""
    b, h, w, c = x.shape
    check(lib().psld_conv3x3_fewout_f32(x.data_ptr(), w_ohwi.data_ptr(), _p(bias), y.data_ptr(), b, h, w, c, cout,
                                        _stream()), "psld_conv3x3_fewout_f32")


def scale_copy2d(src: Tensor, ld_src: int, dst: Tensor, ld_dst: int, rows: int, cols: int, alpha: float = 1.0,
                 src_off: int = 0, dst_off: int = 0):
    check(lib().psld_scale_copy2d_f32(src.data_ptr() + 4 * src_off, ld_src, dst.data_ptr() + 4 * dst_off, ld_dst, rows,
                                      cols, alpha, _stream()), "psld_scale_copy2d_f32")


def copy2d(src: Tensor, ld_src: int, dst: Tensor, ld_dst: int, rows: int, cols: int, accumulate: bool = False,
           src_off: int = 0, dst_off: int = 0):
    check(lib().psld_copy2d_f32(src.data_ptr() + 4 * src_off, ld_src, dst.data_ptr() + 4 * dst_off, ld_dst, rows, cols,
                                1 if accumulate else 0, _stream()), "psld_copy2d_f32")


def softmax_rows(x: Tensor, out: Tensor, rows: int, L: int):
    check(lib().psld_softmax_rows_f32(x.data_ptr(), out.data_ptr(), rows, L, _stream()), "psld_softmax_rows_f32")


def softmax_rows_bwd(y: Tensor, dy: Tensor, dx: Tensor, rows: int, L: int):
    check(lib().psld_softmax_rows_bwd_f32(y.data_ptr(), dy.data_ptr(), dx.data_ptr(), rows, L, _stream()),
          "psld_softmax_rows_bwd_f32")


def time_embed(t: Tensor, W: Tensor, use_log: bool) -> Tensor:
    b, e = t.shape[0], W.shape[0]
    out = torch.empty((b, 2 * e), device=t.device, dtype=torch.float32)
    check(lib().psld_time_embed_f32(_chk(t).data_ptr(), W.data_ptr(), out.data_ptr(), b, e, 1 if use_log else 0,
                                    _stream()), "psld_time_embed_f32")
    return out


def fused_bias_act(x: Tensor, bias: Optional[Tensor], act: int = 3, alpha: float = 0.2, scale: float = 2 ** 0.5,
                   refer: Optional[Tensor] = None, grad: int = 0):
    """The reference's second native op with its full argument list (op/fused_bias_act.cpp:11-20; Python wrappers
    op/fused_act.py:20-97): bias indexed along dim 1; ``grad`` = 1 with ``refer`` = the forward output is the first
    derivative applied to the incoming gradient ``x``, ``grad`` = 2 the (zero) second derivative."""
    y = torch.empty_like(x)
    step_b = 1
    for d in x.shape[2:]:
        step_b *= d
    check(lib().psld_fused_bias_act_grad_f32(_chk(x).data_ptr(), _p(bias), _p(refer), y.data_ptr(), x.numel(),
                                             bias.numel() if bias is not None else 1, step_b, act, grad, alpha, scale,
                                             _stream()), "psld_fused_bias_act_grad_f32")
    return y


# ------------------------------------------------------------------------------------------------
# SDE / loss / sampler / optimiser
# ------------------------------------------------------------------------------------------------
COEFF_STRIDE = 12


def perturb_coeffs(t: Tensor, params: SdeParams, xx_0: float, mm_0: float, nan_flag: Tensor) -> Tensor:
    b = t.shape[0]
    out = torch.empty((b, COEFF_STRIDE), device=t.device, dtype=torch.float64)
    check(lib().psld_perturb_coeffs_f64(_chk(t, torch.float64).data_ptr(), b, C.byref(params), float(xx_0),
                                        float(mm_0), out.data_ptr(), nan_flag.data_ptr(), _stream()),
          "psld_perturb_coeffs_f64")
    return out


def perturb(x0: Tensor, m0: Optional[Tensor], eps: Tensor, coeffs: Tensor, params: SdeParams, want_f32=True,
            want_f64=False, want_mu=False):
    b, c, h, w = x0.shape
    z = torch.empty((b, 2 * c, h, w), device=x0.device, dtype=torch.float32) if want_f32 else None
    u = torch.empty((b, 2 * c, h, w), device=x0.device, dtype=torch.float64) if want_f64 else None
    mu = torch.empty((b, 2 * c, h, w), device=x0.device, dtype=torch.float64) if want_mu else None
    check(lib().psld_perturb_f32(_chk(x0).data_ptr(), _p(m0), _chk(eps).data_ptr(), coeffs.data_ptr(),
                                 C.byref(params), b, c, h * w, _p(z), _p(u), _p(mu), _stream()), "psld_perturb_f32")
    return z, u, mu


def sqerr_loss(eps: Tensor, eps_pred: Tensor, reduce_mean: bool, want_grad: bool, grad_scale: float = 1.0):
    n = eps.numel()
    loss = torch.empty((), device=eps.device, dtype=torch.float32)
    grad = torch.empty_like(eps_pred) if want_grad else None
    ws = workspace(lib().psld_reduce_workspace_bytes(n), eps.device)
    check(lib().psld_sqerr_loss_f32(_chk(eps).data_ptr(), _chk(eps_pred).data_ptr(), n, 1 if reduce_mean else 0,
                                    loss.data_ptr(), _p(grad), grad_scale, ws.data_ptr(), _stream()),
          "psld_sqerr_loss_f32")
    return loss, grad


def vp_score_loss(eps: Tensor, eps_pred: Tensor, t: Optional[Tensor], beta0: float, beta1: float, mode: int,
                  reduce_mean: bool, want_grad: bool):
    """ScoreLoss criteria beyond the eps-MSE: mode 1 = L1 (f32 loss), mode 2 = 'nll' weighting (f64 loss)."""
    b = eps.shape[0]
    per = eps.numel() // b
    loss = torch.empty((), device=eps.device, dtype=torch.float64 if mode == 2 else torch.float32)
    grad = torch.empty_like(eps_pred) if want_grad else None
    ws = workspace(lib().psld_reduce_workspace_bytes(eps.numel()), eps.device)
    check(lib().psld_vp_score_loss(_chk(eps).data_ptr(), _chk(eps_pred).data_ptr(),
                                   _chk(t, torch.float64).data_ptr() if t is not None else None, float(beta0), float(beta1),
                                   b, per, mode, 1 if reduce_mean else 0, loss.data_ptr(), _p(grad), 1.0, ws.data_ptr(),
                                   _stream()), "psld_vp_score_loss")
    return loss, grad


def em_step(x: Tensor, eps_pred: Tensor, z: Optional[Tensor], k: EmCoeffs, x_f32: Optional[Tensor]):
    b, c2, h, w = x.shape
    check(lib().psld_em_step_f64(_chk(x, torch.float64).data_ptr(), _chk(eps_pred).data_ptr(), _p(z), C.byref(k), b,
                                 c2 // 2, h * w, _p(x_f32), _stream()), "psld_em_step_f64")


def reverse_sde(x: Tensor, eps_pred: Tensor, k: EmCoeffs):
    b, c2, h, w = x.shape
    f = torch.empty_like(x)
    g = torch.empty_like(x)
    check(lib().psld_reverse_sde_f64(_chk(x, torch.float64).data_ptr(), _chk(eps_pred).data_ptr(), C.byref(k), b,
                                     c2 // 2, h * w, f.data_ptr(), g.data_ptr(), _stream()), "psld_reverse_sde_f64")
    return f, g


def reverse_sde_rows(x: Tensor, eps_pred: Optional[Tensor], t_rev: Tensor, params: SdeParams, xx_0: float, mm_0: float,
                     score_mode: int, probability_flow: bool, nan_flag: Tensor):
    """(f_bar, g_bar) - or (f, g) of the forward SDE when ``eps_pred`` is None - with one time per sample."""
    b, c2, h, w = x.shape
    f = torch.empty_like(x)
    g = torch.empty_like(x)
    check(lib().psld_reverse_sde_rows_f64(_chk(x, torch.float64).data_ptr(), _p(eps_pred),
                                          _chk(t_rev, torch.float64).data_ptr(), C.byref(params), float(xx_0), float(mm_0),
                                          int(score_mode), 1 if probability_flow else 0, b, c2 // 2, h * w,
                                          f.data_ptr(), g.data_ptr(), nan_flag.data_ptr(), _stream()),
          "psld_reverse_sde_rows_f64")
    return f, g


def sscs_analytic(x: Tensor, z: Tensor, k: SscsCoeffs, x_f32: Optional[Tensor]):
    b, c2, h, w = x.shape
    check(lib().psld_sscs_analytic_f64(_chk(x, torch.float64).data_ptr(), _chk(z, torch.float64).data_ptr(),
                                       C.byref(k), b, c2 // 2, h * w, _p(x_f32), _stream()), "psld_sscs_analytic_f64")


def sscs_score_step(x: Tensor, eps_pred: Tensor, k: EmCoeffs):
    b, c2, h, w = x.shape
    check(lib().psld_sscs_score_step_f64(_chk(x, torch.float64).data_ptr(), _chk(eps_pred).data_ptr(), C.byref(k), b,
                                         c2 // 2, h * w, _stream()), "psld_sscs_score_step_f64")


def samples_to_uint8(samples: Tensor, is_augmented: bool = True, denorm: bool = True) -> Tensor:
    """[B,2C,H,W] f64 sampler output -> uint8 [B,H,W,C] ready for the PNG encoder (SimpleImageWriter,
    callbacks.py:88-124): 16x less device->host traffic than copying the f64 state."""
    b, ct, h, w = samples.shape
    c = ct // 2 if is_augmented else ct
    out = torch.empty((b, h, w, c), device=samples.device, dtype=torch.uint8)
    check(lib().psld_samples_to_uint8(_chk(samples, torch.float64).data_ptr(), out.data_ptr(), b, c, ct, h * w,
                                      1 if denorm else 0, _stream()), "psld_samples_to_uint8")
    return out


def uint8_to_images(img: Tensor, norm: bool = True, flip: Optional[Tensor] = None) -> Tensor:
    """uint8 [B,H,W,C] -> f32 [B,C,H,W] in [-1,1] (util.data_scaler + CIFAR10Dataset.__getitem__)."""
    b, h, w, c = img.shape
    out = torch.empty((b, c, h, w), device=img.device, dtype=torch.float32)
    check(lib().psld_uint8_to_images_f32(_chk(img, torch.uint8).data_ptr(), out.data_ptr(), _p(flip), b, c, h, w,
                                         1 if norm else 0, _stream()), "psld_uint8_to_images_f32")
    return out


def _ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr


def lincomb(out: Tensor, base: Optional[Tensor], vs: Sequence[Tensor], coefs: Sequence[float],
            out_f32: Optional[Tensor] = None):
    """out = base + sum_j coefs[j] * vs[j] (float64), optional f32 copy."""
    cf = (C.c_double * len(coefs))(*[float(c) for c in coefs])
    check(lib().psld_lincomb_f64(out.data_ptr(), _p(base), _ptr_array(vs), cf, len(vs), out.numel(), _p(out_f32),
                                 _stream()), "psld_lincomb_f64")
    return out


def scaled_norm_sq(vs: Sequence[Tensor], coefs: Sequence[float], p: Tensor, q: Tensor, atol: float, rtol: float,
                   out: Tensor):
    cf = (C.c_double * len(coefs))(*[float(c) for c in coefs])
    ws = workspace(lib().psld_reduce_workspace_bytes(p.numel()), p.device)
    check(lib().psld_scaled_norm_sq_f64(_ptr_array(vs), cf, len(vs), p.data_ptr(), q.data_ptr(), atol, rtol, p.numel(),
                                        out.data_ptr(), ws.data_ptr(), _stream()), "psld_scaled_norm_sq_f64")
    return out


def vp_perturb(x0: Tensor, eps: Tensor, t: Tensor, beta0: float, beta1: float, want_f32=True, want_f64=False):
    b = x0.shape[0]
    z = torch.empty_like(x0) if want_f32 else None
    u = torch.empty(x0.shape, device=x0.device, dtype=torch.float64) if want_f64 else None
    check(lib().psld_vp_perturb_f32(_chk(x0).data_ptr(), _chk(eps).data_ptr(), _chk(t, torch.float64).data_ptr(),
                                    float(beta0), float(beta1), b, x0.numel() // b, _p(z), _p(u), _stream()),
          "psld_vp_perturb_f32")
    return z, u


def vp_reverse(x: Tensor, eps_pred: Tensor, z: Optional[Tensor], beta: float, std: float, dt: float, pf: bool,
               update: bool, x_f32: Optional[Tensor] = None):
    """update=False -> returns f_bar; update=True -> Euler-Maruyama step of x in place."""
    fbar = None if update else torch.empty_like(x)
    check(lib().psld_vp_reverse_f64(_chk(x, torch.float64).data_ptr(), _chk(eps_pred).data_ptr(), _p(z), float(beta),
                                    float(std), float(dt), 1 if pf else 0, 1 if update else 0, x.numel(), _p(fbar),
                                    _p(x_f32), _stream()), "psld_vp_reverse_f64")
    return fbar


def guide(x: Tensor, grad: Tensor, coef_x: float, coef_m: float, x_f32: Optional[Tensor] = None):
    """x[:, :C] += coef_x * grad[:, :C]; x[:, C:] += coef_m * grad[:, C:] in place on the f64 state (grad f32)."""
    b, c2, h, w = x.shape
    check(lib().psld_guide_f64(_chk(x, torch.float64).data_ptr(), _chk(grad).data_ptr(), float(coef_x), float(coef_m),
                               b, c2 // 2, h * w, _p(x_f32), _stream()), "psld_guide_f64")
    return x


def softmax_xent(logits: Tensor, labels: Tensor, loss_scale: float, grad_scale: float, want_grad: bool = True):
    """(loss 0-d f32, dlogits or None, correct 0-d f32): cross entropy of [rows][n] logits vs int64 labels."""
    rows, n = logits.shape
    # a label outside [0, n) would index past the logits row in the kernel; torch's cross entropy raises for it - this
    # is the device-side equivalent (no host synchronisation: the assert fires when the stream reaches it)
    torch._assert_async(((labels >= 0) & (labels < n)).all(), f"softmax_xent: label outside [0, {n})")
    loss = torch.empty((), device=logits.device, dtype=torch.float32)
    correct = torch.empty((), device=logits.device, dtype=torch.float32)
    grad = torch.empty_like(logits) if want_grad else None
    check(lib().psld_softmax_xent_f32(_chk(logits).data_ptr(), _chk(labels, torch.int64).data_ptr(), rows, n,
                                      float(loss_scale), float(grad_scale), loss.data_ptr(), _p(grad),
                                      correct.data_ptr(), _stream()), "psld_softmax_xent_f32")
    return loss, grad, correct


def mask_combine(x: Tensor, u: Tensor, mask: Tensor, x_f32: Optional[Tensor] = None):
    """x <- x*(1-mask) + u*mask in place on the f64 state [B,2C,H,W]; mask [B,C,H,W] f32 of {0,1}."""
    b, c2, h, w = x.shape
    check(lib().psld_mask_combine_f64(_chk(x, torch.float64).data_ptr(), _chk(u, torch.float64).data_ptr(),
                                      _chk(mask).data_ptr(), b, c2 // 2, h * w, _p(x_f32), _stream()),
          "psld_mask_combine_f64")
    return x


def f64_to_f32(x: Tensor) -> Tensor:
    y = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    check(lib().psld_f64_to_f32(_chk(x, torch.float64).data_ptr(), y.data_ptr(), x.numel(), _stream()), "f64_to_f32")
    return y


def f32_to_f64(x: Tensor) -> Tensor:
    y = torch.empty(x.shape, device=x.device, dtype=torch.float64)
    check(lib().psld_f32_to_f64(_chk(x).data_ptr(), y.data_ptr(), x.numel(), _stream()), "f32_to_f64")
    return y


def grad_norm(g: Tensor, norm_out: Tensor):
    ws = workspace(lib().psld_reduce_workspace_bytes(g.numel()), g.device)
    check(lib().psld_grad_norm_f32(g.data_ptr(), g.numel(), norm_out.data_ptr(), ws.data_ptr(), _stream()),
          "psld_grad_norm_f32")


def adam_ema(p: Tensor, g: Tensor, m: Tensor, v: Tensor, ema: Optional[Tensor], norm: Optional[Tensor],
             max_norm: float, lr: float, beta1: float, beta2: float, eps: float, weight_decay: float, step: int,
             ema_tau: float, write_clipped_grad: bool = False, hyper_dev: Optional[Tensor] = None,
             poison: Optional[Tensor] = None):
    """``hyper_dev``: float32[2] device tensor holding (lr / bias_correction1, 1 / sqrt(bias_correction2)) of THIS step
    (``adam_step_scalars``); then ``lr`` / ``step`` of the call are ignored (captured training step).
    The launch is guarded by the device's error word (``gn_team_sync``): while it is set the step is a no-op and
    ``poison`` (float32 scalar on the device: the loss the caller logs) is overwritten with NaN - no host read."""
    assert poison is None or (poison.dtype == torch.float32 and poison.is_cuda and poison.numel() == 1)
    check(lib().psld_adam_ema_f32(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), _p(ema), p.numel(),
                                  _p(norm), max_norm, lr, beta1, beta2, eps, weight_decay, step, ema_tau,
                                  1 if write_clipped_grad else 0, g.data_ptr() if write_clipped_grad else None,
                                  _p(hyper_dev), gn_team_sync(p.device).data_ptr(), _p(poison), _stream()),
          "psld_adam_ema_f32")


def adam_step_scalars(lr: float, beta1: float, beta2: float, step: int, out: Tensor) -> Tensor:
    """(lr / (1 - beta1^step), 1 / sqrt(1 - beta2^step)) as the launcher of psld_adam_ema_f32 forms them, into the
    2-element float32 HOST tensor ``out``."""
    assert out.dtype == torch.float32 and out.numel() == 2 and not out.is_cuda
    lib().psld_adam_step_scalars(float(lr), float(beta1), float(beta2), int(step), out.data_ptr())
    return out


def adam_step_scalars_dev(lr: float, beta1: float, beta2: float, step: int, out: Tensor) -> Tensor:
    """The same two scalars into the 2-element float32 DEVICE tensor ``out``, by a kernel that takes them by value
    (stream-ordered: no host staging buffer to race with)."""
    assert out.dtype == torch.float32 and out.numel() == 2 and out.is_cuda
    check(lib().psld_adam_step_scalars_dev(float(lr), float(beta1), float(beta2), int(step), out.data_ptr(), _stream()),
          "psld_adam_step_scalars_dev")
    return out


def ema(target: Tensor, src: Tensor, tau: float):
    check(lib().psld_ema_f32(target.data_ptr(), src.data_ptr(), target.numel(), tau, gn_team_sync(target.device).data_ptr(),
                             _stream()), "psld_ema_f32")
