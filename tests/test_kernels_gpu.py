"""GPU parity tests of the individual HIP kernels, called through the C ABI (psld_amd.ops ->
libpsld_hip.so) and checked against the CPU oracle / plain torch fp32 on the same seeded inputs.
Tolerances: fp32 contractions 2e-6..1e-5 rel-L2 (k-ordered fmaf chain vs oneDNN blocking);
f64 SDE math 1e-12; elementwise fp32 1e-6.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import psld_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda"


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.fixture(scope="module")
def ops():
    from psld_amd import ops as _ops
    _ops.lib()
    return _ops


def gen(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


# ---------------------------------------------------------------------------------------------------
# tile engine
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ta,tb", [(0, 1), (0, 0), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (256, 256, 256), (200, 72, 100), (6, 130, 54), (1, 1, 4),
                                   (300, 6, 2304), (129, 257, 36)])
def test_gemm_layouts(ops, ta, tb, M, N, K):
    A = gen(*((K, M) if ta else (M, K)), seed=1)
    B = gen(*((N, K) if tb else (K, N)), seed=2)
    ref = (A.t() if ta else A).double() @ (B.t() if tb else B).double()
    Ad, Bd = A.to(DEV), B.to(DEV)
    Cd = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm_raw(ta, tb, M, N, K, Ad, A.shape[1], 0, Bd, B.shape[1], 0, Cd, N, 0)
    assert rel_l2(Cd, ref) < 2e-6


def test_gemm_asymmetric_identity(ops):
    """A = I with an asymmetric B catches a transposed C write (cdna guide §3)."""
    n = 64
    A = torch.eye(n)
    B = torch.arange(n * n, dtype=torch.float32).reshape(n, n) / 7.0
    Cd = torch.zeros(n, n, device=DEV)
    ops.gemm_raw(0, 0, n, n, n, A.to(DEV), n, 0, B.to(DEV), n, 0, Cd, n, 0)
    assert torch.equal(Cd.cpu(), B)


def test_gemm_batched_epilogue(ops):
    b, M, N, K = 3, 70, 96, 40
    A, B = gen(b, M, K, seed=3), gen(b, N, K, seed=4)
    bias, res = gen(N, seed=5), gen(b, M, N, seed=6)
    rowb = gen(b * 7, N, seed=7)  # rows_per_img = 10 -> M/10 = 7 images per batch entry
    ref = 0.5 * torch.einsum("bmk,bnk->bmn", A.double(), B.double()) + bias.double()
    ref = ref + rowb[:7].double().repeat_interleave(10, dim=0)[None]
    ref = (ref + res.double()) * 0.7
    Cd = torch.ones(b, M, N, device=DEV)
    ref = ref + 1.0
    rb = rowb.to(DEV)
    e = ops.epilogue(alpha=0.5, bias=bias.to(DEV), rowbias=rb, rows_per_img=10, residual=res.to(DEV), ld_residual=N,
                     residual_stride_batch=M * N, out_scale=0.7, accumulate=True)
    ops.gemm_raw(0, 1, M, N, K, A.to(DEV), K, M * K, B.to(DEV), K, N * K, Cd, N, M * N, b, e)
    assert rel_l2(Cd, ref) < 2e-6


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("cfg", [
    dict(b=2, c1=32, c2=0, co=64, s=8, k=3, stride=1, pad=1),
    dict(b=2, c1=64, c2=32, co=64, s=8, k=3, stride=1, pad=1),     # concat of two sources
    dict(b=3, c1=32, c2=0, co=32, s=16, k=1, stride=1, pad=0),     # 1x1
    dict(b=2, c1=6, c2=0, co=32, s=16, k=3, stride=1, pad=1),      # stem (scalar gather path)
    dict(b=2, c1=64, c2=0, co=6, s=16, k=3, stride=1, pad=1),      # head (N=6)
    dict(b=2, c1=32, c2=0, co=32, s=9, k=3, stride=2, pad=0),      # pyramid stride-2 pad-0
    dict(b=1, c1=256, c2=256, co=256, s=8, k=3, stride=1, pad=1),  # north-star K=4608
])
def test_conv_forward(ops, cfg):
    b, c1, c2, co, s, k = cfg["b"], cfg["c1"], cfg["c2"], cfg["co"], cfg["s"], cfg["k"]
    stride, pad = cfg["stride"], cfg["pad"]
    x = gen(b, c1 + c2, s, s, seed=10)
    w = gen(co, c1 + c2, k, k, seed=11, scale=0.1)
    bias = gen(co, seed=12)
    ref = F.conv2d(x.double(), w.double(), bias.double(), stride=stride, padding=pad)
    oh = ref.shape[2]
    x1 = _nhwc(x[:, :c1]).to(DEV)
    x2 = _nhwc(x[:, c1:]).to(DEV) if c2 else None
    wp = torch.empty(co, k * k, c1 + c2, device=DEV)
    ops.pack_ohwi(w.to(DEV), wp)
    assert torch.equal(wp.cpu(), w.permute(0, 2, 3, 1).reshape(co, k * k, c1 + c2))
    y = torch.full((b, oh, oh, co), float("nan"), device=DEV)
    ops.conv2d_nhwc(x1, x2, wp, co, k, k, stride, pad, 1, oh, oh, y, ops.epilogue(bias=bias.to(DEV)))
    assert rel_l2(y.permute(0, 3, 1, 2), ref) < 3e-6


@pytest.mark.parametrize("cfg", [
    dict(b=2, ci=32, co=64, s=8, k=3, stride=1, pad=1),
    dict(b=2, ci=32, co=32, s=9, k=3, stride=2, pad=0),
    dict(b=2, ci=64, co=6, s=8, k=3, stride=1, pad=1),   # head: dy has 6 channels (scalar path)
    dict(b=2, ci=32, co=64, s=8, k=1, stride=1, pad=0),
])
def test_conv_dgrad_and_wgrad(ops, cfg):
    b, ci, co, s, k, stride, pad = (cfg[n] for n in ("b", "ci", "co", "s", "k", "stride", "pad"))
    x = gen(b, ci, s, s, seed=20).requires_grad_(True)
    w = gen(co, ci, k, k, seed=21, scale=0.1).requires_grad_(True)
    y = F.conv2d(x.double(), w.double(), stride=stride, padding=pad)
    gy = gen(*y.shape, seed=22)
    y.backward(gy.double())
    oh = y.shape[2]
    gyd = _nhwc(gy).to(DEV)
    # dgrad = conv of dy with the flipped / transposed filter
    wd = torch.empty(ci, k * k, co, device=DEV)
    ops.pack_dgrad(w.detach().to(DEV), wd)
    dx = torch.full((b, s, s, ci), float("nan"), device=DEV)
    ops.conv2d_nhwc(gyd, None, wd, ci, k, k, 1, k - 1 - pad, stride, s, s, dx)
    assert rel_l2(dx.permute(0, 3, 1, 2), x.grad) < 3e-6
    # wgrad through split-K slabs, reduced straight into OIHW
    nsplit = 3
    slabs = torch.full((nsplit, co, k * k, ci), float("nan"), device=DEV)
    ops.conv2d_wgrad_nhwc(gyd, co, _nhwc(x.detach()).to(DEV), k, k, stride, pad, oh, oh, slabs, ci, 0, nsplit)
    dw = torch.empty(co, ci, k, k, device=DEV)
    ops.reduce_slabs(slabs, nsplit, co * k * k * ci, dw, layout=1, cout=co, taps=k * k, cin=ci)
    assert rel_l2(dw, w.grad) < 3e-6


# ---------------------------------------------------------------------------------------------------
# 3x3 convolutions on bf16 limb MFMA (csrc/conv_split.hip): same fp32 tolerance as the fp32 MFMA kernels
# ---------------------------------------------------------------------------------------------------
def test_limb_fragments_exact(ops):
    """The three bf16 limbs of every weight sum back to the fp32 value bit-exactly (hi + mid + lo == x), and sit
    where the kernel's B-fragment loads expect them."""
    co, ci = 128, 64
    w = gen(co, ci, 3, 3, seed=90, scale=0.3)
    w.view(-1)[:7] = torch.tensor([0.0, 1.0, -1.0, 1e-30, 1e-20, 65504.0, -3.3e38 / 4])
    for dgrad in (False, True):
        if dgrad:
            w = gen(ci, co, 3, 3, seed=91, scale=0.3)    # n_out = cin must be a multiple of 128
        frag = ops.conv3x3_frag(w.to(DEV), dgrad).cpu()
        n_out, k_in = (w.shape[1], w.shape[0]) if dgrad else (w.shape[0], w.shape[1])
        f = frag.view(torch.int16).view(n_out // 128, 2, k_in // 32, 9, 4, 3, 64, 8)      # nt wc chunk tap nb limb lane j
        limbs = (f.to(torch.int32) << 16).view(torch.float32).double().sum(dim=5)          # hi + mid + lo
        # lane -> (k group of 8, column): n = nt*128 + wc*64 + nb*16 + lane%16 ; k = chunk*32 + lane//16*8 + j
        limbs = limbs.view(n_out // 128, 2, k_in // 32, 9, 4, 4, 16, 8)                    # nt wc chunk tap nb kq col j
        rec = limbs.permute(0, 1, 4, 6, 2, 5, 7, 3).reshape(n_out, k_in, 9)               # [n][k][tap]
        wt = w.double().reshape(w.shape[0], w.shape[1], 9)
        ref = wt.flip(2).permute(1, 0, 2) if dgrad else wt
        assert torch.equal(rec, ref.contiguous())


@pytest.mark.parametrize("cfg", [
    dict(b=2, c1=64, c2=0, co=128, h=8, w=8),        # two whole images per 128-pixel tile
    dict(b=3, c1=32, c2=0, co=128, h=8, w=8),        # odd batch: the last tile holds one image only
    dict(b=2, c1=64, c2=32, co=128, h=16, w=16),     # concat of two sources, half-image tiles
    dict(b=1, c1=32, c2=0, co=256, h=32, w=32),      # 4-row tiles, two N tiles
    dict(b=1, c1=32, c2=0, co=128, h=64, w=64),      # CelebA-64 resolution (2-row tiles)
    dict(b=1, c1=256, c2=256, co=256, h=8, w=8),     # north-star K = 4608, split over the channel chunks
    dict(b=2, c1=32, c2=0, co=128, h=4, w=8),        # h*w = 32: four images per tile
    dict(b=12, c1=64, c2=0, co=256, h=32, w=32),     # 192 tiles of 128 rows -> 384 tiles of 64 rows, no K split
])
def test_conv3x3_split_forward(ops, cfg):
    b, c1, c2, co, h, w_ = (cfg[n] for n in ("b", "c1", "c2", "co", "h", "w"))
    assert ops.conv3x3_split_supported(c1, c2, b, h, w_, co)
    x = gen(b, c1 + c2, h, w_, seed=40)
    w = gen(co, c1 + c2, 3, 3, seed=41, scale=0.1)
    bias, res = gen(co, seed=42), gen(b, co, h, w_, seed=43)
    temb = gen(b, co, seed=44)
    ref = (F.conv2d(x.double(), w.double(), bias.double(), padding=1) + temb.double()[:, :, None, None]
           + res.double()) * 0.7
    x1 = _nhwc(x[:, :c1]).to(DEV)
    x2 = _nhwc(x[:, c1:]).to(DEV) if c2 else None
    wf = ops.conv3x3_frag(w.to(DEV), False)
    y = torch.full((b, h, w_, co), float("nan"), device=DEV)
    epi = ops.epilogue(bias=bias.to(DEV), rowbias=temb.to(DEV), rows_per_img=h * w_, residual=_nhwc(res).to(DEV),
                       ld_residual=co, out_scale=0.7)
    ops.conv3x3_split(x1, x2, wf, co, y, epi)
    assert rel_l2(y.permute(0, 3, 1, 2), ref) < 3e-6
    # agrees with the fp32 MFMA kernel to fp32 rounding
    wp = torch.empty(co, 9, c1 + c2, device=DEV)
    ops.pack_ohwi(w.to(DEV), wp)
    y32 = torch.empty_like(y)
    ops.conv2d_nhwc(x1, x2, wp, co, 3, 3, 1, 1, 1, h, w_, y32, epi)
    assert rel_l2(y, y32) < 3e-6


def test_limb_planes_roundtrip_is_exact(ops):
    """fp32 -> bf16 limb planes [rows][c/32][3][32] -> fp32: hi + mid + lo reproduces every value bit for bit, over a
    wide dynamic range, and the plane layout is the documented one."""
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(2, 4, 8, 96, generator=g) * torch.exp(torch.randn(2, 4, 8, 96, generator=g) * 8.0)).to(DEV)
    x[0, 0, 0, :4] = torch.tensor([0.0, -0.0, 1.0, -3.5e-30], device=DEV)
    lp = ops.f32_to_limb(x)
    assert torch.equal(ops.limb_to_f32(lp), x)
    raw = lp.t.view(2 * 4 * 8, 3, 3, 32).cpu()                                     # [row][chunk][limb][32]
    rec = (raw.to(torch.int32) << 16).view(torch.float32).double().sum(dim=2).reshape(2, 4, 8, 96)
    assert torch.equal(rec, x.double().cpu())
    hi = (raw[:, :, 0].to(torch.int32) << 16).view(torch.float32).reshape(2, 4, 8, 96)
    assert torch.equal(hi, x.cpu().bfloat16().float())                              # hi limb = round-to-nearest-even bf16


@pytest.mark.parametrize("cfg", [
    dict(b=2, c1=64, c2=0, co=128, h=8, w=8),
    dict(b=3, c1=32, c2=0, co=128, h=8, w=8),
    dict(b=2, c1=64, c2=32, co=128, h=16, w=16),
    dict(b=1, c1=32, c2=0, co=256, h=32, w=32),
    dict(b=5, c1=96, c2=0, co=128, h=32, w=32),      # 13 row groups, odd chunk count, several tiles per image
    dict(b=1, c1=32, c2=0, co=128, h=64, w=64),      # CelebA-64 resolution: single-image (non double-buffered) variant
    dict(b=1, c1=256, c2=256, co=256, h=8, w=8),     # split over the channel chunks
    dict(b=2, c1=32, c2=0, co=128, h=4, w=8),
    dict(b=12, c1=64, c2=0, co=256, h=32, w=32),     # 64-row tiles without a K split (9 row groups per image)
])
def test_conv3x3_limb_input_is_bitwise_the_split_kernel(ops, cfg):
    """psld_conv3x3_limb_f32 (input as bf16 limb planes, halo tile staged by LDS-DMA into two images) computes the
    same limb products in the same order as psld_conv3x3_split_f32 (fp32 input split in the kernel): bitwise equal
    outputs, with the full epilogue; and both against fp64."""
    b, c1, c2, co, h, w_ = (cfg[n] for n in ("b", "c1", "c2", "co", "h", "w"))
    x = gen(b, c1 + c2, h, w_, seed=40)
    w = gen(co, c1 + c2, 3, 3, seed=41, scale=0.1)
    bias, res = gen(co, seed=42), gen(b, co, h, w_, seed=43)
    temb = gen(b, co, seed=44)
    ref = (F.conv2d(x.double(), w.double(), bias.double(), padding=1) + temb.double()[:, :, None, None]
           + res.double()) * 0.7
    x1 = _nhwc(x[:, :c1]).to(DEV)
    x2 = _nhwc(x[:, c1:]).to(DEV) if c2 else None
    wf = ops.conv3x3_frag(w.to(DEV), False)
    epi = ops.epilogue(bias=bias.to(DEV), rowbias=temb.to(DEV), rows_per_img=h * w_, residual=_nhwc(res).to(DEV),
                       ld_residual=co, out_scale=0.7)
    y = torch.full((b, h, w_, co), float("nan"), device=DEV)
    ops.conv3x3_split(x1, x2, wf, co, y, epi)
    for rep in range(3):        # repeated: a DMA that has not landed when it is read shows up as a run-to-run difference
        yl = torch.full((b, h, w_, co), float("nan"), device=DEV)
        ops.conv3x3_split(ops.f32_to_limb(x1), ops.f32_to_limb(x2) if x2 is not None else None, wf, co, yl, epi)
        assert torch.equal(yl, y), (cfg, rep, rel_l2(yl, y))
    assert rel_l2(yl.permute(0, 3, 1, 2), ref) < 3e-6


def test_conv3x3_split_wide_dynamic_range(ops):
    """Limb products keep fp32 accuracy when operands span many binades (gradients late in training are ~1e-8)."""
    b, c, co, s = 2, 64, 128, 8
    g = torch.Generator().manual_seed(50)
    x = gen(b, c, s, s, seed=51) * torch.exp(torch.randn(b, c, s, s, generator=g) * 6.0) * 1e-6
    w = gen(co, c, 3, 3, seed=52) * torch.exp(torch.randn(co, c, 3, 3, generator=g) * 4.0)
    ref = F.conv2d(x.double(), w.double(), padding=1)
    y = torch.empty(b, s, s, co, device=DEV)
    ops.conv3x3_split(_nhwc(x).to(DEV), None, ops.conv3x3_frag(w.to(DEV), False), co, y)
    ref32 = F.conv2d(x, w, padding=1)                   # plain fp32 on the CPU as the yardstick
    assert rel_l2(y.permute(0, 3, 1, 2), ref) < 2.0 * max(rel_l2(ref32, ref), 2e-7)


# ---------------------------------------------------------------------------------------------------
# Winograd F(2x2, 3x3) limb convolution (conv_wino.hip): the same convolution, re-associated
# ---------------------------------------------------------------------------------------------------
WINO_FWD = [
    dict(b=2, c1=64, c2=0, co=128, h=8, w=8),        # two images per workgroup tile
    dict(b=3, c1=32, c2=0, co=128, h=8, w=8),        # odd batch: the last tile is half empty
    dict(b=2, c1=64, c2=32, co=128, h=16, w=16),     # two sources, odd chunk count
    dict(b=1, c1=32, c2=0, co=256, h=32, w=32),
    dict(b=5, c1=96, c2=0, co=128, h=32, w=32),      # odd batch, odd chunk count
    dict(b=1, c1=32, c2=0, co=128, h=64, w=64),      # CelebA-64 resolution: 4 x 32-pixel blocks
    dict(b=3, c1=256, c2=256, co=256, h=16, w=16),   # the up path's two-source 512 -> 256
    dict(b=2, c1=32, c2=0, co=128, h=4, w=8),        # four images per tile
    dict(b=7, c1=256, c2=0, co=256, h=32, w=32),     # north-star layer shape, odd batch
]


@pytest.mark.parametrize("cfg", [
    dict(b=128, c1=256, c2=0, co=256, h=8, w=8),     # the 8x8 level at the training batch: 128 tiles -> 2 chunk ranges
    dict(b=64, c1=256, c2=256, co=256, h=8, w=8),    # two sources, 64 tiles -> 4 ranges
    dict(b=3, c1=256, c2=256, co=256, h=16, w=16),   # 12 tiles -> 8 ranges of two chunks
    dict(b=7, c1=256, c2=0, co=256, h=32, w=32),     # 112 tiles -> 2 ranges
])
def test_conv3x3_wino_split_chunks(ops, cfg):
    """psld_conv3x3_wino_ws_f32 (round 6): a launch whose grid leaves CUs idle splits its channel chunks over workgroups (plain
    partial outputs + one reduction pass with the whole epilogue) - against fp64 with the full epilogue at the unsplit kernel's
    tolerance, next to the unsplit launch on the same inputs (another summation order: close, not bitwise), with `accumulate`,
    and repeatable bit for bit."""
    b, c1, c2, co, h, w_ = (cfg[n] for n in ("b", "c1", "c2", "co", "h", "w"))
    assert ops.conv3x3_wino_ws_bytes(c1, c2, b, h, w_, co) > 0
    x = gen(b, c1 + c2, h, w_, seed=40)
    w = gen(co, c1 + c2, 3, 3, seed=41, scale=0.1)
    bias, res, temb = gen(co, seed=42), gen(b, co, h, w_, seed=43), gen(b, co, seed=44)
    ref = (F.conv2d(x.double(), w.double(), bias.double(), padding=1) + temb.double()[:, :, None, None] + res.double()) * 0.7
    x1 = _nhwc(x[:, :c1]).to(DEV)
    x2 = _nhwc(x[:, c1:]).to(DEV) if c2 else None
    uf = ops.conv3x3_wino_frag(w.to(DEV), False)
    epi = ops.epilogue(bias=bias.to(DEV), rowbias=temb.to(DEV), rows_per_img=h * w_, residual=_nhwc(res).to(DEV),
                       ld_residual=co, out_scale=0.7)
    y0 = torch.full((b, h, w_, co), float("nan"), device=DEV)
    ops.conv3x3_wino(x1, x2, uf, co, y0, epi)
    y1 = torch.full_like(y0, float("nan"))
    ops.conv3x3_wino(x1, x2, uf, co, y1, epi, allow_split=True)
    e0, e1 = rel_l2(y0.permute(0, 3, 1, 2), ref), rel_l2(y1.permute(0, 3, 1, 2), ref)
    print(f"unsplit {e0:.2e} split {e1:.2e}")
    assert e1 < 3e-6 and e1 < 2.0 * max(e0, 2e-7) and not torch.equal(y0, y1)
    y2 = torch.full_like(y0, float("nan"))
    ops.conv3x3_wino(x1, x2, uf, co, y2, epi, allow_split=True)
    assert torch.equal(y1, y2)
    acc = torch.ones_like(y0)
    ops.conv3x3_wino(x1, x2, uf, co, acc, ops.epilogue(alpha=0.5, accumulate=True), allow_split=True)
    plain = F.conv2d(x.double(), w.double(), padding=1) * 0.5 + 1.0
    assert rel_l2(acc.permute(0, 3, 1, 2), plain) < 3e-6
    if ops.conv3x3_wino_gn_supported(c1, c2, b, h, w_, co):
        # the GroupNorm-fused form splits the same way: bitwise the apply pass + split convolution (the inference forward's pair)
        g1, b1 = (gen(c1, seed=74) * 0.2 + 1.0).to(DEV), (gen(c1, seed=75) * 0.1).to(DEV)
        st1 = ops.gn_stats(x1, g1, b1)
        st2 = a2 = None
        if c2:
            g2, b2 = (gen(c2, seed=76) * 0.2 + 1.0).to(DEV), (gen(c2, seed=77) * 0.1).to(DEV)
            st2 = ops.gn_stats(x2, g2, b2)
            a2 = ops.gn_apply(x2, st2, True)
        ya = torch.full_like(y0, float("nan"))
        ops.conv3x3_wino(ops.gn_apply(x1, st1, True), a2, uf, co, ya, epi, allow_split=True)
        yf = torch.full_like(y0, float("nan"))
        ops.conv3x3_wino_gn(x1, st1, x2, st2, True, uf, co, yf, epi, allow_split=True)
        assert torch.equal(yf, ya)


@pytest.mark.parametrize("cfg", WINO_FWD)
def test_conv3x3_wino_forward(ops, cfg):
    """psld_conv3x3_wino_f32 against fp64 torch with the full epilogue (bias, time-embedding row bias, residual, scale),
    next to the direct limb kernel on the same inputs: within 3e-6 of fp64 and no worse than 2x the direct kernel's
    error (the gate of VERDICT r02 was 1e-5).  Reference: nn.Conv2d 3x3, song_sde/layers.py:103-109.  (The product
    library holds ONE Winograd kernel family; the variants that lost their A/B live in libpsld_hip_abl.so.)"""
    e, ed = _wino_forward_errors(ops, cfg)
    print(f"winograd {e:.2e} direct {ed:.2e}")
    assert e < 3e-6 and e < 2.0 * max(ed, 2e-7)


@pytest.mark.parametrize("cfg", [
    dict(b=2, c1=64, c2=32, co=128, h=16, w=16),     # two sources, each with its own statistics
    dict(b=5, c1=96, c2=0, co=128, h=32, w=32),
    dict(b=1, c1=32, c2=0, co=128, h=64, w=64),
    dict(b=3, c1=256, c2=256, co=256, h=16, w=16),
    dict(b=7, c1=256, c2=0, co=256, h=32, w=32),
])
@pytest.mark.parametrize("act", [True, False])
def test_conv3x3_wino_fused_groupnorm(ops, cfg, act):
    """psld_conv3x3_wino_gn_f32: GroupNorm apply (+SiLU) inside the Winograd kernel's input staging == the apply pass
    followed by psld_conv3x3_wino_f32, bit for bit (same activation arithmetic, zero padding of the ACTIVATED tensor), and
    within the kernel tolerance of fp64 GroupNorm + SiLU + conv.  Reference: GroupNorm_0/1 + act + Conv_0/1,
    layerspp.py:245-263 in eval mode."""
    b, c1, c2, co, h, w_ = (cfg[n] for n in ("b", "c1", "c2", "co", "h", "w"))
    assert ops.conv3x3_wino_gn_supported(c1, c2, b, h, w_, co)
    x = gen(b, c1 + c2, h, w_, seed=70) * 1.5 + 0.3
    w = gen(co, c1 + c2, 3, 3, seed=71, scale=0.1)
    bias, res = gen(co, seed=72), gen(b, co, h, w_, seed=73)
    x1 = _nhwc(x[:, :c1]).to(DEV)
    x2 = _nhwc(x[:, c1:]).to(DEV) if c2 else None
    g1, b1 = (gen(c1, seed=74) * 0.2 + 1.0).to(DEV), (gen(c1, seed=75) * 0.1).to(DEV)
    st1 = ops.gn_stats(x1, g1, b1)
    st2 = None
    if c2:
        g2, b2 = (gen(c2, seed=76) * 0.2 + 1.0).to(DEV), (gen(c2, seed=77) * 0.1).to(DEV)
        st2 = ops.gn_stats(x2, g2, b2)
    uf = ops.conv3x3_wino_frag(w.to(DEV), False)
    epi = ops.epilogue(bias=bias.to(DEV), residual=_nhwc(res).to(DEV), ld_residual=co, out_scale=0.7)
    a1 = ops.gn_apply(x1, st1, act)
    a2 = ops.gn_apply(x2, st2, act) if c2 else None
    y_ref = torch.full((b, h, w_, co), float("nan"), device=DEV)
    ops.conv3x3_wino(a1, a2, uf, co, y_ref, epi)
    y = torch.full((b, h, w_, co), float("nan"), device=DEV)
    ops.conv3x3_wino_gn(x1, st1, x2, st2, act, uf, co, y, epi)
    assert torch.equal(y, y_ref)
    # fp64: GroupNorm per source (its own groups), act, conv
    def gn64(t, gamma, beta):
        c = t.shape[1]
        o = F.group_norm(t.double(), ops.gn_groups(c), gamma.double().cpu(), beta.double().cpu(), eps=1e-6)
        return F.silu(o) if act else o
    parts = [gn64(x[:, :c1], g1, b1)] + ([gn64(x[:, c1:], g2, b2)] if c2 else [])
    ref = (F.conv2d(torch.cat(parts, 1), w.double(), bias.double(), padding=1) + res.double()) * 0.7
    assert rel_l2(y.permute(0, 3, 1, 2), ref) < 5e-6


def _wino_forward_errors(ops, cfg):
    b, c1, c2, co, h, w_ = (cfg[n] for n in ("b", "c1", "c2", "co", "h", "w"))
    assert ops.conv3x3_wino_supported(c1, c2, b, h, w_, co)
    x = gen(b, c1 + c2, h, w_, seed=40)
    w = gen(co, c1 + c2, 3, 3, seed=41, scale=0.1)
    bias, res = gen(co, seed=42), gen(b, co, h, w_, seed=43)
    temb = gen(b, co, seed=44)
    ref = (F.conv2d(x.double(), w.double(), bias.double(), padding=1) + temb.double()[:, :, None, None]
           + res.double()) * 0.7
    x1 = _nhwc(x[:, :c1]).to(DEV)
    x2 = _nhwc(x[:, c1:]).to(DEV) if c2 else None
    epi = ops.epilogue(bias=bias.to(DEV), rowbias=temb.to(DEV), rows_per_img=h * w_, residual=_nhwc(res).to(DEV),
                       ld_residual=co, out_scale=0.7)
    y = torch.full((b, h, w_, co), float("nan"), device=DEV)
    ops.conv3x3_wino(x1, x2, ops.conv3x3_wino_frag(w.to(DEV), False), co, y, epi)
    yd = torch.empty_like(y)
    ops.conv3x3_split(x1, x2, ops.conv3x3_frag(w.to(DEV), False), co, yd, epi)
    return rel_l2(y.permute(0, 3, 1, 2), ref), rel_l2(yd.permute(0, 3, 1, 2), ref)


@pytest.mark.parametrize("cfg", [
    dict(b=2, ci=128, co=64, h=8, w=8),
    dict(b=3, ci=128, co=128, h=16, w=16),
    dict(b=1, ci=256, co=64, h=32, w=32),
    dict(b=2, ci=128, co=256, h=32, w=32),
    dict(b=3, ci=256, co=512, h=16, w=16),
])
def test_conv3x3_wino_dgrad(ops, cfg):
    """Data gradient = the same kernel on the rotated, role-swapped filter (psld_pack_conv3x3_wino(dgrad = 1)), with the
    alpha / accumulate epilogue the backward tape uses."""
    b, ci, co, h, w_ = (cfg[n] for n in ("b", "ci", "co", "h", "w"))
    x = gen(b, ci, h, w_, seed=60).requires_grad_(True)
    w = gen(co, ci, 3, 3, seed=61, scale=0.1).requires_grad_(True)
    y = F.conv2d(x.double(), w.double(), padding=1)
    gy = gen(*y.shape, seed=62)
    y.backward(gy.double())
    gyd = _nhwc(gy).to(DEV)
    assert ops.conv3x3_wino_supported(co, 0, b, h, w_, ci)
    uf = ops.conv3x3_wino_frag(w.detach().to(DEV), True)
    dx = torch.full((b, h, w_, ci), float("nan"), device=DEV)
    ops.conv3x3_wino(gyd, None, uf, ci, dx)
    assert rel_l2(dx.permute(0, 3, 1, 2), x.grad) < 3e-6
    prev = gen(b, h, w_, ci, seed=63).to(DEV)
    acc = prev.clone()
    ops.conv3x3_wino(gyd, None, uf, ci, acc, ops.epilogue(alpha=0.5, accumulate=True))
    assert rel_l2(acc.permute(0, 3, 1, 2), 0.5 * x.grad + prev.permute(0, 3, 1, 2).cpu().double()) < 3e-6


def test_conv3x3_wino_wide_dynamic_range(ops):
    """The transformed operands are sums of four values of possibly very different magnitude: fp32 adds, then the exact
    three-limb split.  Within 4x of plain fp32 direct convolution on operands spanning many binades."""
    b, c, co, s = 2, 64, 128, 8
    g = torch.Generator().manual_seed(50)
    x = gen(b, c, s, s, seed=51) * torch.exp(torch.randn(b, c, s, s, generator=g) * 6.0) * 1e-6
    w = gen(co, c, 3, 3, seed=52) * torch.exp(torch.randn(co, c, 3, 3, generator=g) * 4.0)
    ref = F.conv2d(x.double(), w.double(), padding=1)
    y = torch.empty(b, s, s, co, device=DEV)
    ops.conv3x3_wino(_nhwc(x).to(DEV), None, ops.conv3x3_wino_frag(w.to(DEV), False), co, y)
    ref32 = F.conv2d(x, w, padding=1)
    assert rel_l2(y.permute(0, 3, 1, 2), ref) < 4.0 * max(rel_l2(ref32, ref), 2e-7)


@pytest.mark.parametrize("b,c,co,h", [(24, 128, 256, 32), (96, 128, 256, 16), (6, 32, 128, 64), (48, 128, 128, 32)])
def test_gn_partials_from_wino_epilogue(ops, b, c, co, h):
    """psld_epilogue_t.gn_part from the Winograd kernel's epilogue: GroupNorm statistics of its output, for every group
    size that is a multiple of the fine groups (8 channels; 4 for a 128-channel output, whose own groups are 4 wide:
    gn_fine), against a statistics pass over the written tensor."""
    x = gen(b, h, h, c, seed=70).to(DEV)
    w = gen(co, c, 3, 3, seed=71, scale=0.05).to(DEV)
    bias = gen(co, seed=72).to(DEV)
    gamma, beta = (1 + 0.1 * gen(co, seed=73)).to(DEV), (0.1 * gen(co, seed=74)).to(DEV)
    part = ops.gn_part_buffer(b, h * h, co, DEV)
    part.fill_(float("nan"))
    y = torch.empty(b, h, h, co, device=DEV)
    ops.conv3x3_wino(x, None, ops.conv3x3_wino_frag(w, False), co, y, ops.epilogue(bias=bias, gn_part=part, gn_hw=h * h))
    assert bool(torch.isfinite(part).all())
    assert part.fine_width == (4 if co == 128 else 8)
    for groups in (co // part.fine_width, co // 8, co // 16):
        st = ops.gn_stats_from_part(part, y.shape, gamma, beta, groups=groups)
        ref = ops.gn_stats(y, gamma, beta, groups=groups)
        assert rel_l2(st.mean, ref.mean) < 1e-5 and rel_l2(st.rstd, ref.rstd) < 1e-5


def test_wino_pack_batch_matches_single(ops):
    ws = [gen(128, 64, 3, 3, seed=80).to(DEV), gen(256, 128, 3, 3, seed=81).to(DEV), gen(128, 256, 3, 3, seed=82).to(DEV)]
    singles, outs, rows, total = [], [], [], 0
    for i, w in enumerate(ws):
        dgrad = bool(i & 1)
        singles.append(ops.conv3x3_wino_frag(w, dgrad))
        o = torch.zeros_like(singles[-1])
        outs.append(o)
        rows.append(ops.conv3x3_wino_frag_entry(w, dgrad, o) + [total])
        total += w.shape[0] * w.shape[1] // 8
    ops.pack_wino_batch(torch.tensor(rows, dtype=torch.int64, device=DEV), len(rows), total)
    for a, bb in zip(singles, outs):
        assert torch.equal(a[:-16384], bb[:-16384])       # (the tail is read-ahead padding, never written)


@pytest.mark.parametrize("cfg", [
    dict(b=2, ci=128, co=64, h=8, w=8),
    dict(b=3, ci=128, co=128, h=16, w=16),
    dict(b=1, ci=256, co=64, h=32, w=32),
    dict(b=1, ci=128, co=64, h=64, w=64),
    dict(b=2, ci=128, co=256, h=32, w=32),
])
def test_conv3x3_split_dgrad_and_wgrad(ops, cfg):
    b, ci, co, h, w_ = (cfg[n] for n in ("b", "ci", "co", "h", "w"))
    x = gen(b, ci, h, w_, seed=60).requires_grad_(True)
    w = gen(co, ci, 3, 3, seed=61, scale=0.1).requires_grad_(True)
    y = F.conv2d(x.double(), w.double(), padding=1)
    gy = gen(*y.shape, seed=62)
    y.backward(gy.double())
    gyd = _nhwc(gy).to(DEV)
    assert ops.conv3x3_split_supported(co, 0, b, h, w_, ci)
    dx = torch.full((b, h, w_, ci), float("nan"), device=DEV)
    ops.conv3x3_split(gyd, None, ops.conv3x3_frag(w.detach().to(DEV), True), ci, dx)
    assert rel_l2(dx.permute(0, 3, 1, 2), x.grad) < 3e-6
    # weight gradient: slabs over K ranges (one of them short), into a wider [co][9][cin_total] block at col0
    assert ops.conv3x3_wgrad_split_supported(co, ci, b, h, w_)
    ktiles = b * h * w_ // 32
    nsplit = min(3, ktiles)
    per = -(-ktiles // nsplit)
    nsplit = -(-ktiles // per)
    cin_total, col0 = ci + 64, 64
    slabs = torch.zeros((nsplit, co, 9, cin_total), device=DEV)
    ops.conv3x3_wgrad_split(gyd, co, _nhwc(x.detach()).to(DEV), slabs, cin_total, col0, nsplit)
    dw = slabs.sum(0)[:, :, col0:].reshape(co, 3, 3, ci).permute(0, 3, 1, 2)
    assert rel_l2(dw, w.grad) < 3e-6
    assert torch.count_nonzero(slabs[:, :, :, :col0]) == 0


@pytest.mark.parametrize("cfg", [
    dict(b=2, ci=128, co=64, h=8, w=8),
    dict(b=3, ci=128, co=128, h=16, w=16),
    dict(b=1, ci=256, co=64, h=32, w=32),
    dict(b=1, ci=128, co=64, h=64, w=64),
    dict(b=2, ci=128, co=256, h=32, w=32),
])
def test_conv3x3_wgrad_with_x_as_limb_planes(ops, cfg):
    """psld_conv3x3_wgrad_xlimb_f32: the x operand staged from limb planes without a split - bitwise the result of the
    kernel that splits fp32 x itself (same products, same order), on short last K ranges and column offsets."""
    b, ci, co, h, w_ = (cfg[n] for n in ("b", "ci", "co", "h", "w"))
    x = _nhwc(gen(b, ci, h, w_, seed=60)).to(DEV)
    gy = _nhwc(gen(b, co, h, w_, seed=62)).to(DEV)
    ktiles = b * h * w_ // 32
    nsplit = min(3, ktiles)
    per = -(-ktiles // nsplit)
    nsplit = -(-ktiles // per)
    cin_total, col0 = ci + 64, 64
    ref = torch.zeros((nsplit, co, 9, cin_total), device=DEV)
    ops.conv3x3_wgrad_split(gy, co, x, ref, cin_total, col0, nsplit)
    xl = ops.f32_to_limb(x)
    for rep in range(2):
        got = torch.zeros_like(ref)
        ops.conv3x3_wgrad_split(gy, co, xl, got, cin_total, col0, nsplit)
        assert torch.equal(got, ref), (cfg, rep, rel_l2(got, ref))


@pytest.mark.parametrize("per", [1, 2, 3, 5])
def test_wave_specialised_wgrad_on_short_k_ranges(ops, per):
    """dwgrad_ws_kernel (128-channel tiles, fp32 x: producer waves stage the next K tile while consumer waves multiply) on K
    ranges of 1, 2, 3 and 5 tiles - its prologue, its two-image hand-over and a short last range - against the
    one-role-per-wave kernel on limb-plane x: bit for bit, and against fp64 autograd of nn.Conv2d (layers.py:103-109)."""
    b, ci, co, h, w_ = 3, 128, 256, 16, 16
    xt = gen(b, ci, h, w_, seed=80).requires_grad_(False)
    x = _nhwc(xt).to(DEV)
    gyt = gen(b, co, h, w_, seed=81)
    gy = _nhwc(gyt).to(DEV)
    ktiles = b * h * w_ // 32                                     # 24
    nsplit = -(-ktiles // per)
    got = torch.full((nsplit, co, 9, ci), float("nan"), device=DEV)
    ops.conv3x3_wgrad_split(gy, co, x, got, ci, 0, nsplit)
    ref = torch.full_like(got, float("nan"))
    ops.conv3x3_wgrad_split(gy, co, ops.f32_to_limb(x), ref, ci, 0, nsplit)
    assert torch.equal(got, ref)
    w = torch.zeros(co, ci, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xt.double(), w, padding=1).backward(gyt.double())
    dw = got.sum(0).reshape(co, 3, 3, ci).permute(0, 3, 1, 2)
    assert rel_l2(dw, w.grad) < 3e-6


@pytest.mark.parametrize("cfg", [
    dict(b=2, c1=128, c2=0, co=256, h=32, w=32, nsplit=None),      # one K tile = two tile rows of an image
    dict(b=4, c1=256, c2=0, co=256, h=16, w=16, nsplit=1),         # K tile = half an image; all K tiles in ONE range
    dict(b=8, c1=128, c2=0, co=256, h=8, w=8, nsplit=2),           # K tile = two images
    dict(b=1, c1=128, c2=0, co=256, h=64, w=64, nsplit=4),         # K tile = one tile row
    dict(b=3, c1=256, c2=128, co=256, h=32, w=32, nsplit=3),       # two sources of a concatenation, odd batch
    dict(b=2, c1=128, c2=128, co=512, h=16, w=16, nsplit=None),    # two c_out tiles
    dict(b=1, c1=128, c2=0, co=128, h=64, w=64, nsplit=None),      # 128 output channels (CelebA-64's first level): 128 x 128 tiles
    dict(b=2, c1=128, c2=128, co=384, h=32, w=32, nsplit=2),       # ... three of them, two sources
    dict(b=4, c1=128, c2=256, co=256, h=8, w=8, nsplit=None),      # sources of different widths on 8x8 maps (two K tiles)
    dict(b=6, c1=384, c2=0, co=128, h=16, w=16, nsplit=None),      # three c_in tiles, 12 K tiles in uneven splits
    dict(b=2, c1=128, c2=0, co=640, h=16, w=16, nsplit=None),      # five 128-channel c_out tiles
    dict(b=10, c1=256, c2=0, co=256, h=32, w=32, nsplit=7),        # 80 K tiles in 7 splits: the last one shorter
])
def test_conv3x3_wgrad_winograd_domain(ops, cfg):
    """psld_conv3x3_wgrad_wino_f32 (wgrad_wino.hip): dU = sum over 2x2 tiles of (A dY A^T) (x) (B^T d B) on limb MFMAs, dw =
    G^T dU G - against fp64 autograd of nn.Conv2d (layers.py:103-109) at the forward Winograd kernel's tolerance (3e-6),
    next to the direct limb weight gradient on the same inputs; image borders (zero padding inside the transform), every
    K-tile geometry (two tile rows / half an image / two images / one row per tile), accumulate = 1, repeatability."""
    b, c1, c2, co, h, w_ = (cfg[n] for n in ("b", "c1", "c2", "co", "h", "w"))
    ci = c1 + c2
    assert ops.conv3x3_wgrad_wino_supported(co, c1, c2, b, h, w_)
    xt = gen(b, ci, h, w_, seed=83) + 0.25                 # a mean: the transforms cancel it, the limbs must carry it
    gyt = gen(b, co, h, w_, seed=84)
    w = torch.zeros(co, ci, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xt.double(), w, padding=1).backward(gyt.double())
    x1 = _nhwc(xt[:, :c1]).to(DEV)
    x2 = _nhwc(xt[:, c1:]).to(DEV) if c2 else None
    gy = _nhwc(gyt).to(DEV)
    dw = torch.full((co, ci, 3, 3), float("nan"), device=DEV)
    ops.conv3x3_wgrad_wino(gy, co, x1, dw, x2=x2, nsplit=cfg["nsplit"])
    e = rel_l2(dw, w.grad)
    # the direct limb kernel on the same inputs
    ktiles = b * h * w_ // 32
    slabs = torch.zeros((1, co, 9, ci), device=DEV)
    ops.conv3x3_wgrad_split(gy, co, x1, slabs, ci, 0, 1, x2=x2)
    ed = rel_l2(slabs[0].reshape(co, 3, 3, ci).permute(0, 3, 1, 2), w.grad)
    print(f"winograd-domain wgrad {e:.2e}  direct {ed:.2e}  ({ktiles} pixel K tiles)")
    assert e < 3e-6 and e < 2.0 * max(ed, 3e-7)
    again = torch.full_like(dw, float("nan"))
    ops.conv3x3_wgrad_wino(gy, co, x1, again, x2=x2, nsplit=cfg["nsplit"])
    assert torch.equal(again, dw)
    acc = torch.ones_like(dw)
    ops.conv3x3_wgrad_wino(gy, co, x1, acc, x2=x2, nsplit=cfg["nsplit"], accumulate=True)
    assert torch.equal(acc, dw + 1.0)


def test_conv3x3_wgrad_limb_x_two_sources(ops):
    b, c1, c2, co, s_ = 2, 128, 256, 128, 16
    x = gen(b, c1 + c2, s_, s_, seed=70)
    gyd = _nhwc(gen(b, co, s_, s_, seed=71)).to(DEV)
    x1, x2 = _nhwc(x[:, :c1]).to(DEV), _nhwc(x[:, c1:]).to(DEV)
    ref = torch.zeros((2, co, 9, c1 + c2), device=DEV)
    ops.conv3x3_wgrad_split(gyd, co, x1, ref, c1 + c2, 0, 2, x2)
    got = torch.zeros_like(ref)
    ops.conv3x3_wgrad_split(gyd, co, ops.f32_to_limb(x1), got, c1 + c2, 0, 2, ops.f32_to_limb(x2))
    assert torch.equal(got, ref)


def test_gn_apply_limb_planes_match_the_fp32_pass(ops):
    """psld_gn_apply_limb_nhwc writes exactly the limb decomposition of what psld_gn_apply_nhwc_f32 writes (same
    affine, SiLU and dropout mask)."""
    b, h, c = 3, 8, 256
    x = torch.randn(b, h, h, c, generator=torch.Generator().manual_seed(5)).to(DEV)
    gamma = (1 + 0.2 * torch.randn(c, generator=torch.Generator().manual_seed(6))).to(DEV)
    beta = (0.1 * torch.randn(c, generator=torch.Generator().manual_seed(7))).to(DEV)
    st = ops.gn_stats(x, gamma, beta)
    for act, p in ((True, 0.0), (True, 0.15), (False, 0.0)):
        ref = ops.gn_apply(x, st, act, drop_p=p, seed=1234)
        got = ops.limb_to_f32(ops.gn_apply_limb(x, st, act, drop_p=p, seed=1234))
        assert torch.equal(got, ref), (act, p)


@pytest.mark.parametrize("b,c,ih,stride,pad", [(2, 8, 9, 2, 0), (1, 64, 17, 2, 0), (2, 16, 8, 1, 1)])
def test_im2col_col2im(ops, b, c, ih, stride, pad):
    """Many-channel 3x3 im2col in (tap, channel) order and its adjoint (gather form) against F.unfold / F.fold."""
    oh = (ih + 2 * pad - 3) // stride + 1
    x = gen(b, c, ih, ih, seed=90)
    cols = ops.im2col3x3(_nhwc(x).to(DEV), stride, pad, oh, oh)
    ref = F.unfold(x, 3, padding=pad, stride=stride)                         # [b][c*9][L], row = ch*9 + tap
    ref = ref.view(b, c, 9, oh * oh).permute(0, 3, 2, 1).reshape(b * oh * oh, 9 * c)
    assert torch.equal(cols.cpu(), ref)
    d = gen(b * oh * oh, 9 * c, seed=91)
    dx = ops.col2im3x3(d.to(DEV), (b, ih, ih, c), stride, pad, oh, oh)
    dref = F.fold(d.view(b, oh * oh, 9, c).permute(0, 3, 2, 1).reshape(b, c * 9, oh * oh).double(), (ih, ih), 3,
                  padding=pad, stride=stride)
    assert rel_l2(dx.permute(0, 3, 1, 2), dref) < 1e-6


@pytest.mark.parametrize("b,c,co,h,w_", [(2, 256, 6, 8, 8), (1, 128, 6, 5, 7), (3, 32, 3, 16, 16), (1, 256, 6, 32, 32)])
def test_conv3x3_fewout(ops, b, c, co, h, w_):
    """Head convolution (few output channels) as a dot-product kernel; ragged widths (w % 4 != 0) included."""
    x = gen(b, c, h, w_, seed=95)
    w = gen(co, c, 3, 3, seed=96, scale=0.1)
    bias = gen(co, seed=97)
    ref = F.conv2d(x.double(), w.double(), bias.double(), padding=1)
    assert ops.conv3x3_fewout_supported(c, co)
    wp = torch.empty(co, 3, 3, c, device=DEV)
    ops.pack_ohwi(w.to(DEV), wp)
    y = torch.full((b, h, w_, co), float("nan"), device=DEV)
    ops.conv3x3_fewout(_nhwc(x).to(DEV), wp, bias.to(DEV), co, y)
    assert rel_l2(y.permute(0, 3, 1, 2), ref) < 2e-6
    assert not ops.conv3x3_fewout_supported(512, 6) and not ops.conv3x3_fewout_supported(256, 8)


@pytest.mark.parametrize("b,c,co,h", [(24, 128, 256, 32), (96, 128, 256, 16), (40, 128, 256, 32),      # third: eight-wave pointwise kernel
                                      (48, 128, 128, 32)])                                             # four-channel sums
def test_gn_partials_from_conv_epilogue(ops, b, c, co, h):
    """GroupNorm statistics of a limb convolution's output as a by-product of its epilogue == gn_stats of the output,
    for the consumer's own group size and for the coarser groups of a concatenation source; same for the pointwise form."""
    x = gen(b, c, h, h, seed=110)
    w = gen(co, c, 3, 3, seed=111, scale=0.1)
    bias, res = gen(co, seed=112), gen(b, h, h, co, seed=113)
    gamma, beta = (1 + 0.2 * gen(co, seed=114)).to(DEV), (0.1 * gen(co, seed=115)).to(DEV)
    xd = _nhwc(x).to(DEV)
    assert ops.gn_part_supported(b, h * h, co)
    part = ops.gn_part_buffer(b, h * h, co, DEV)
    part.fill_(float("nan"))
    y = torch.empty(b, h, h, co, device=DEV)
    epi = ops.epilogue(bias=bias.to(DEV), residual=res.to(DEV), ld_residual=co, out_scale=0.7, gn_part=part, gn_hw=h * h)
    ops.conv3x3_split(xd, None, ops.conv3x3_frag(w.to(DEV), False), co, y, epi)
    y_ref = torch.empty_like(y)
    ops.conv3x3_split(xd, None, ops.conv3x3_frag(w.to(DEV), False), co, y_ref,
                      ops.epilogue(bias=bias.to(DEV), residual=res.to(DEV), ld_residual=co, out_scale=0.7))
    assert torch.equal(y, y_ref)
    assert part.fine_width == (4 if co == 128 else 8)            # 128 channels: 32 groups of 4 -> four-channel sums
    for groups in (None, ops.gn_groups(2 * co) // 2):            # own GroupNorm / as one half of a concatenation
        ref = ops.gn_stats(y, gamma, beta, groups=groups)
        got = ops.gn_stats_from_part(part, y.shape, gamma, beta, groups=groups)
        assert (got.mean - ref.mean).abs().max() < 2e-6 * ref.mean.abs().max() + 1e-7
        assert rel_l2(got.rstd, ref.rstd) < 2e-6 and rel_l2(got.scale, ref.scale) < 2e-6
        assert (got.shift - ref.shift).abs().max() < 1e-5
    # pointwise form
    m = b * h * h
    wmat = gen(co, c, seed=116, scale=0.1)
    part.fill_(float("nan"))
    y2 = torch.empty(m, co, device=DEV)
    ops.gemm_split(xd.view(m, c), None, m, ops.gemm_frag(wmat.to(DEV), co, c, c, 1), co, y2,
                   ops.epilogue(bias=bias.to(DEV), gn_part=part, gn_hw=h * h))
    ref = ops.gn_stats(y2.view(b, h, h, co), gamma, beta)
    got = ops.gn_stats_from_part(part, (b, h, h, co), gamma, beta)
    assert rel_l2(got.rstd, ref.rstd) < 2e-6 and rel_l2(got.scale, ref.scale) < 2e-6 and (got.shift - ref.shift).abs().max() < 1e-5


@pytest.mark.parametrize("b,c,co,h", [(16, 256, 256, 8), (128, 256, 256, 8), (6, 128, 256, 16), (8, 128, 128, 8), (3, 512, 256, 8)])
def test_gn_partials_from_split_launches(ops, b, c, co, h):
    """A launch too small to fill the chip splits its K range over workgroups and finishes through the reduction + epilogue
    pass; since round 6 that pass forms the GroupNorm partial sums of what it stores (conv_reduce_epilogue_gn_kernel), so the
    consumer's statistics cost no pass over the tensor: direct limb kernel, Winograd form (allow_split, plain and
    GroupNorm-fused input), pointwise form.  The output is bit for bit the one of the same launch without the sums."""
    x = gen(b, c, h, h, seed=210)
    w = gen(co, c, 3, 3, seed=211, scale=0.1)
    bias, res, temb = gen(co, seed=212), gen(b, h, h, co, seed=213), gen(b, co, seed=214)
    gamma, beta = (1 + 0.2 * gen(co, seed=215)).to(DEV), (0.1 * gen(co, seed=216)).to(DEV)
    xd = _nhwc(x).to(DEV)
    assert ops.gn_part_supported(b, h * h, co)
    part = ops.gn_part_buffer(b, h * h, co, DEV)

    def epis():
        kw = dict(bias=bias.to(DEV), rowbias=temb.to(DEV), rows_per_img=h * h, residual=res.to(DEV), ld_residual=co, out_scale=0.7)
        return ops.epilogue(gn_part=part, gn_hw=h * h, **kw), ops.epilogue(**kw)

    def check_stats(y):
        for groups in (None, ops.gn_groups(2 * co) // 2):
            ref = ops.gn_stats(y, gamma, beta, groups=groups)
            got = ops.gn_stats_from_part(part, y.shape, gamma, beta, groups=groups)
            assert (got.mean - ref.mean).abs().max() < 2e-6 * ref.mean.abs().max() + 1e-7
            assert rel_l2(got.rstd, ref.rstd) < 2e-6 and rel_l2(got.scale, ref.scale) < 2e-6
            assert (got.shift - ref.shift).abs().max() < 1e-5

    # direct limb kernel (splits below 384 tiles when it is handed a workspace: ops.conv3x3_split does)
    e_gn, e_plain = epis()
    y, y_ref = torch.empty(b, h, h, co, device=DEV), torch.empty(b, h, h, co, device=DEV)
    part.fill_(float("nan"))
    wf = ops.conv3x3_frag(w.to(DEV), False)
    ops.conv3x3_split(xd, None, wf, co, y, e_gn)
    ops.conv3x3_split(xd, None, wf, co, y_ref, e_plain)
    assert torch.equal(y, y_ref) and not torch.isnan(part).any()
    check_stats(y)
    # Winograd form with its channel chunks split over workgroups
    if ops.conv3x3_wino_supported(c, 0, b, h, h, co) and ops.conv3x3_wino_ws_bytes(c, 0, b, h, h, co) > 0:
        uf = ops.conv3x3_wino_frag(w.to(DEV), False)
        part.fill_(float("nan"))
        ops.conv3x3_wino(xd, None, uf, co, y, e_gn, allow_split=True)
        ops.conv3x3_wino(xd, None, uf, co, y_ref, e_plain, allow_split=True)
        assert torch.equal(y, y_ref) and not torch.isnan(part).any()
        check_stats(y)
        if ops.conv3x3_wino_gn_supported(c, 0, b, h, h, co):      # GroupNorm + SiLU of the INPUT inside the staging
            gin, bin_ = (1 + 0.2 * gen(c, seed=217)).to(DEV), (0.1 * gen(c, seed=218)).to(DEV)
            st_in = ops.gn_stats(xd, gin, bin_)
            part.fill_(float("nan"))
            ops.conv3x3_wino_gn(xd, st_in, None, None, True, uf, co, y, e_gn, allow_split=True)
            ops.conv3x3_wino_gn(xd, st_in, None, None, True, uf, co, y_ref, e_plain, allow_split=True)
            assert torch.equal(y, y_ref) and not torch.isnan(part).any()
            check_stats(y)
    # pointwise form (splits below 128 tiles of 128 x 256)
    m = b * h * h
    wmat = gen(co, c, seed=219, scale=0.1)
    part.fill_(float("nan"))
    y2, y2_ref = torch.empty(m, co, device=DEV), torch.empty(m, co, device=DEV)
    gf = ops.gemm_frag(wmat.to(DEV), co, c, c, 1)
    ops.gemm_split(xd.view(m, c), None, m, gf, co, y2, ops.epilogue(bias=bias.to(DEV), gn_part=part, gn_hw=h * h))
    ops.gemm_split(xd.view(m, c), None, m, gf, co, y2_ref, ops.epilogue(bias=bias.to(DEV)))
    assert torch.equal(y2, y2_ref) and not torch.isnan(part).any()
    check_stats(y2.view(b, h, h, co))


def test_two_source_weight_gradients(ops):
    """dwgrad / pwgrad reading the input of the convolution from two tensors (unmaterialised concatenation)."""
    b, c1, c2, co, h, w_ = 2, 128, 256, 128, 16, 16
    x = gen(b, c1 + c2, h, w_, seed=80)
    gy = gen(b, co, h, w_, seed=81)
    xh = _nhwc(x)
    x1, x2 = xh[..., :c1].contiguous().to(DEV), xh[..., c1:].contiguous().to(DEV)
    gyd = _nhwc(gy).to(DEV)
    # 3x3
    wt = torch.zeros(co, c1 + c2, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wt, padding=1).backward(gy.double())
    ktiles = b * h * w_ // 32
    nsplit = 4
    slabs = torch.full((nsplit, co, 9, c1 + c2), float("nan"), device=DEV)
    ops.conv3x3_wgrad_split(gyd, co, x1, slabs, c1 + c2, 0, nsplit, x2)
    dw = slabs.sum(0).reshape(co, 3, 3, c1 + c2).permute(0, 3, 1, 2)
    assert rel_l2(dw, wt.grad) < 3e-6
    # 1x1
    m = b * h * w_
    ref = gyd.view(m, co).double().t().cpu() @ xh.reshape(m, c1 + c2).double()
    slabs = torch.full((nsplit, co, c1 + c2), float("nan"), device=DEV)
    ops.gemm_tn_split(co, c1, m, gyd, co, x1, c1, slabs, c1 + c2, nsplit, x2, c2, c2)
    assert rel_l2(slabs.sum(0), ref) < 3e-6


@pytest.mark.parametrize("ta,tb", [(0, 1), (0, 0), (1, 0)])
@pytest.mark.parametrize("m,n,k,batch,pad", [(128, 128, 32, 1, 0), (256, 128, 96, 3, 0), (256, 256, 256, 5, 64)])
def test_bgemm_split(ops, ta, tb, m, n, k, batch, pad):
    """Batched limb GEMM, both operands split in the kernel: NT / NN / TN, strided operands (column slices) and output."""
    A = gen(batch, *((k, m + pad) if ta else (m, k + pad)), seed=77)
    B = gen(batch, *((n, k + pad) if tb else (k, n + pad)), seed=78)
    a_use = A[:, :, pad:] if pad else A
    b_use = B[:, :, pad:] if pad else B
    opa = a_use.double().transpose(1, 2) if ta else a_use.double()
    opb = b_use.double().transpose(1, 2) if tb else b_use.double()
    ref = 0.25 * opa @ opb
    assert ops.bgemm_split_supported(ta, tb, m, n, k)
    Ad, Bd = A.to(DEV), B.to(DEV)
    out = torch.full((batch, m, n + pad), float("nan"), device=DEV)
    lda, ldb = A.shape[2], B.shape[2]
    ops.bgemm_split(ta, tb, m, n, k, Ad.view(-1)[pad:], lda, A.shape[1] * lda, Bd.view(-1)[pad:], ldb, B.shape[1] * ldb,
                    out.view(-1)[pad:], n + pad, m * (n + pad), batch, 0.25)
    assert rel_l2(out[:, :, pad:], ref) < 3e-6
    if pad:
        assert torch.isnan(out[:, :, :pad]).all()
    assert not ops.bgemm_split_supported(1, 1, m, n, k) and not ops.bgemm_split_supported(ta, tb, 64, n, k)


@pytest.mark.parametrize("m,n,k,lda,ldb,ldc,nsplit", [(128, 128, 64, 128, 128, 128, 1), (256, 128, 2048, 256, 384, 128, 3),
                                                       (128, 256, 8192, 128, 256, 320, 7), (768, 256, 1024, 768, 256, 256, 32)])
def test_gemm_tn_split(ops, m, n, k, lda, ldb, ldc, nsplit):
    """Pointwise weight gradient on the limb kernel: slabs sum to A^T B; strided operands and output."""
    a, bm = gen(k, lda, seed=75), gen(k, ldb, seed=76)
    ref = a[:, :m].double().t() @ bm[:, ldb - n:].double()
    assert ops.gemm_tn_split_supported(m, n, k)
    kt = k // 32
    per = -(-kt // nsplit)
    nsplit = -(-kt // per)
    slabs = torch.full((nsplit, m, ldc), float("nan"), device=DEV)
    bd = bm.to(DEV)
    ops.gemm_tn_split(m, n, k, a.to(DEV), lda, bd.view(-1)[ldb - n:], ldb, slabs, ldc, nsplit)
    assert rel_l2(slabs[:, :, :n].sum(0), ref) < 3e-6
    if ldc > n:
        assert torch.isnan(slabs[:, :, n:]).all()           # columns beyond n untouched
    assert not ops.gemm_tn_split_supported(64, 128, 64) and not ops.gemm_tn_split_supported(128, 128, 48)


@pytest.mark.parametrize("b,hw,c,fused_buf", [(3, 256, 256, True), (2, 64, 256, True), (5, 256, 128, False), (1, 64, 128, False)])
def test_fused_attention_forward(ops, b, hw, c, fused_buf):
    """psld_attn_fwd_split_f32: softmax(scale q k^T) v in one kernel against fp64 torch (einsum -> softmax -> einsum of
    AttnBlockpp.forward, layerspp.py:82-86), with q | k | v as column slices of one buffer (row stride 3c) or as three
    tensors, the probabilities written or not, and against the three-kernel path on the same inputs."""
    assert ops.attn_fwd_supported(hw, c)
    g = torch.Generator().manual_seed(90)
    qkv = torch.randn(b, hw, 3 * c, generator=g) * 1.5
    scale = float(c) ** -0.5
    q64, k64, v64 = (t.double() for t in qkv.split(c, dim=-1))
    pref = torch.softmax(torch.einsum("bic,bjc->bij", q64, k64) * scale, dim=-1)
    oref = torch.einsum("bij,bjc->bic", pref, v64)
    dev = qkv.to(DEV)
    if fused_buf:
        q, k, v, ld = dev[..., :c], dev[..., c:2 * c], dev[..., 2 * c:], 3 * c
    else:
        q, k, v = (t.contiguous() for t in dev.split(c, dim=-1))
        ld = c
    out = torch.full((b, hw, c), float("nan"), device=DEV)
    p = torch.full((b, hw, hw), float("nan"), device=DEV)
    ops.attn_fwd(q, k, v, ld, b, hw, c, scale, out, p)
    assert rel_l2(out, oref) < 3e-6 and rel_l2(p, pref) < 3e-6
    out2 = torch.full((b, hw, c), float("nan"), device=DEV)
    ops.attn_fwd(q, k, v, ld, b, hw, c, scale, out2, None)
    assert torch.equal(out, out2)
    # the three-kernel path (batched limb GEMM, softmax rows, batched limb GEMM), where the limb GEMM takes the shape
    if not ops.bgemm_split_supported(0, 1, hw, hw, c):
        return
    p3 = torch.empty((b, hw, hw), device=DEV)
    ops.bgemm_split(0, 1, hw, hw, c, q, ld, hw * ld, k, ld, hw * ld, p3, hw, hw * hw, b, scale)
    ops.softmax_rows(p3, p3, b * hw, hw)
    o3 = torch.empty((b, hw, c), device=DEV)
    ops.bgemm_split(0, 0, hw, c, hw, p3, hw, hw * hw, v, ld, hw * ld, o3, c, hw * c, b)
    assert rel_l2(out, o3) < 3e-6 and rel_l2(p, p3) < 3e-6


@pytest.mark.parametrize("m,n,k1,k2", [(256, 128, 64, 0), (1000, 256, 256, 0), (640, 128, 96, 32), (4096, 768, 256, 0),
                                       (130, 128, 512, 0),
                                       # grids of >= 128 tiles of 128 x 256: the eight-wave kernel (ragged last tile, two
                                       # sources, one stage only, three channel tiles)
                                       (32838, 256, 256, 256), (16500, 512, 256, 0), (11000, 768, 64, 0), (32768, 256, 160, 96),
                                       (16400, 256, 128, 0)])
def test_gemm_split(ops, m, n, k1, k2):
    """Pointwise limb kernels (four waves, 128 x 128 tiles; eight waves, 128 x 256 tiles when those fill the chip):
    y = (concat(a1, a2) @ B^T + bias + residual) * scale, B from an [n][k] or a [k][n] matrix."""
    a = gen(m, k1 + k2, seed=70)
    bmat = gen(n, k1 + k2, seed=71, scale=0.1)
    bias, res = gen(n, seed=72), gen(m, n, seed=73)
    ref = (a.double() @ bmat.double().t() + bias.double() + res.double()) * 0.5
    a1 = a[:, :k1].contiguous().to(DEV)
    a2 = a[:, k1:].contiguous().to(DEV) if k2 else None
    assert ops.gemm_split_supported(k1, k2, m, n)
    epi = ops.epilogue(bias=bias.to(DEV), residual=res.to(DEV), ld_residual=n, out_scale=0.5)
    for transposed in (False, True):
        src = (bmat.t().contiguous() if transposed else bmat).to(DEV)          # [k][n] (NIN.W layout) or [n][k]
        sn, sk = (1, n) if transposed else (k1 + k2, 1)
        frag = ops.gemm_frag(src, n, k1 + k2, sn, sk)
        y = torch.full((m, n), float("nan"), device=DEV)
        ops.gemm_split(a1, a2, m, frag, n, y, epi)
        assert rel_l2(y, ref) < 3e-6
    # strided output (one third of a fused q|k|v buffer)
    wide = torch.zeros(m, 3 * n, device=DEV)
    ops.gemm_split(a1, a2, m, frag, n, wide[:, n:], None, ldy=3 * n)
    assert rel_l2(wide[:, n:2 * n], a.double() @ bmat.double().t()) < 3e-6
    assert torch.count_nonzero(wide[:, :n]) == 0 and torch.count_nonzero(wide[:, 2 * n:]) == 0


def test_pack_frag_batch_matches_single(ops):
    ws = [gen(128, 64, 3, 3, seed=80).to(DEV), gen(256, 32, 3, 3, seed=81).to(DEV), gen(64, 128, 3, 3, seed=82).to(DEV)]
    dg = [False, False, True]
    single = [ops.conv3x3_frag(w, d) for w, d in zip(ws, dg)]
    outs = [torch.zeros_like(t) for t in single]
    rows, total = [], 0
    for w, d, o in zip(ws, dg, outs):
        rows.append(ops.conv3x3_frag_entry(w, d, o) + [total])
        total += w.shape[0] * w.shape[1] // 8
    ops.pack_frag_batch(torch.tensor(rows, dtype=torch.int64, device=DEV), len(rows), total)
    for a, b in zip(single, outs):
        assert torch.equal(a, b)


def test_bias_grad(ops):
    b, hw, c, ld = 5, 64, 128, 384
    x = gen(b, hw, ld, seed=83).to(DEV)
    sl = x[..., 128:256]                                    # column slice of a wider buffer
    per = torch.full((b, c), float("nan"), device=DEV)
    out = torch.full((c,), float("nan"), device=DEV)
    ops.bias_grad(sl, ld, b, hw, c, out, 0.5, per)
    ref = sl.double().sum(dim=1)
    assert rel_l2(per, ref) < 1e-6 and rel_l2(out, 0.5 * ref.sum(dim=0)) < 1e-6
    out2 = torch.empty_like(out)
    ops.bias_grad(sl, ld, b, hw, c, out2, 0.5, None)
    assert torch.equal(out, out2)


def test_conv3x3_split_rejects_unsupported(ops):
    assert not ops.conv3x3_split_supported(6, 0, 2, 32, 32, 128)      # stem: 6 input channels
    assert not ops.conv3x3_split_supported(128, 0, 2, 32, 32, 6)      # head: 6 output channels
    assert not ops.conv3x3_split_supported(64, 0, 2, 12, 12, 128)     # width not a power of two
    assert not ops.conv3x3_wgrad_split_supported(96, 64, 2, 8, 8)
    x = torch.zeros(2, 12, 12, 64, device=DEV)
    with pytest.raises(RuntimeError, match="unsupported shape"):
        ops.conv3x3_split(x, None, torch.zeros(128 * 64 * 54, dtype=torch.uint8, device=DEV), 128,
                          torch.empty(2, 12, 12, 128, device=DEV))


def test_softmax_xent_and_guide(ops):
    rows, n = 37, 10
    z = gen(rows, n, seed=88) * 3
    y = torch.randint(0, n, (rows,), generator=torch.Generator().manual_seed(89))
    loss, grad, correct = ops.softmax_xent(z.to(DEV), y.to(DEV), 1.0 / rows, 1.0 / rows)
    zr = z.double().requires_grad_()
    ref = F.cross_entropy(zr, y)
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-6 and rel_l2(grad, zr.grad) < 1e-6
    assert float(correct) == float((z.argmax(1) == y).sum())
    x, gd = gen(2, 6, 4, 4, seed=90).double(), gen(2, 6, 4, 4, seed=91)
    xd, x32 = x.to(DEV).clone(), torch.empty(2, 6, 4, 4, device=DEV)
    ops.guide(xd, gd.to(DEV), 0.25, -1.5, x32)
    ref = x + torch.cat([0.25 * gd[:, :3].double(), -1.5 * gd[:, 3:].double()], dim=1)
    assert torch.equal(xd.cpu(), ref) and torch.equal(x32.cpu(), ref.float())


def test_mask_combine(ops):
    b, c, h = 3, 3, 8
    x, u = gen(b, 2 * c, h, h, seed=85).double(), gen(b, 2 * c, h, h, seed=86).double()
    mask = (gen(b, c, h, h, seed=87) > 0).float()
    m2 = torch.cat([mask, mask], dim=1).double()
    ref = x * (1 - m2) + u * m2
    xd, x32 = x.to(DEV).clone(), torch.empty(b, 2 * c, h, h, device=DEV)
    ops.mask_combine(xd, u.to(DEV), mask.to(DEV), x32)
    assert torch.equal(xd.cpu(), ref) and torch.equal(x32.cpu(), ref.float())


def test_math_mode_switch(ops):
    mode = ops.math_mode()
    try:
        ops.set_math_mode("f32")
        assert ops.math_mode() == "f32"
        ops.set_math_mode("bf16x6")
        assert ops.math_mode() == "bf16x6"
    finally:
        ops.set_math_mode(mode)



def test_gemm_tn_splitk(ops):
    M, N, K = 64, 96, 1000
    A, B = gen(K, M, seed=30), gen(K, N, seed=31)
    slabs = torch.empty(4, M, N, device=DEV)
    ops.gemm_tn_splitk(M, N, K, A.to(DEV), M, B.to(DEV), N, slabs, 4)
    out = torch.empty(M, N, device=DEV)
    ops.reduce_slabs(slabs, 4, M * N, out)
    assert rel_l2(out, A.double().t() @ B.double()) < 2e-6


# ---------------------------------------------------------------------------------------------------
# GroupNorm + SiLU
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("b,c,s", [(2, 32, 8), (3, 64, 16), (2, 128, 32), (2, 256, 16), (1, 384, 8), (2, 512, 8)])
@pytest.mark.parametrize("act", [True, False])
def test_groupnorm_fwd_bwd(ops, b, c, s, act):
    x = (gen(b, c, s, s, seed=40) * 1.5 + 0.3).requires_grad_(True)
    gamma = (1 + 0.2 * gen(c, seed=41)).requires_grad_(True)
    beta = (0.1 * gen(c, seed=42)).requires_grad_(True)
    g = min(c // 4, 32)
    y = F.group_norm(x.double(), g, gamma.double(), beta.double(), 1e-6)
    if act:
        y = F.silu(y)
    gy = gen(*y.shape, seed=43)
    y.backward(gy.double())
    xd = _nhwc(x.detach()).to(DEV)
    gd, bd = gamma.detach().to(DEV), beta.detach().to(DEV)
    st = ops.gn_stats(xd, gd, bd)
    yd = ops.gn_apply(xd, st, act)
    assert rel_l2(yd.permute(0, 3, 1, 2), y) < 2e-6
    dx = torch.full_like(xd, float("nan"))
    dg, db = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
    ops.gn_bwd(_nhwc(gy).to(DEV), xd, st, gd, bd, act, dx, dg, db)
    assert rel_l2(dx.permute(0, 3, 1, 2), x.grad) < 1e-5
    assert rel_l2(dg, gamma.grad) < 1e-5 and rel_l2(db, beta.grad) < 1e-5
    # identity branch folded in (dx = GN term + 0.5*add), on top of an existing gradient; per-source groups override
    addt = gen(*x.shape, seed=44)
    dx2 = dx.clone()
    ops.gn_bwd(_nhwc(gy).to(DEV), xd, st, gd, bd, act, dx2, dg, db, accumulate_dx=True, add=_nhwc(addt).to(DEV), add_scale=0.5)
    assert rel_l2(dx2.permute(0, 3, 1, 2), 2 * x.grad + 0.5 * addt.double()) < 1e-5
    if g % 2 == 0 and (c // 2) % 4 == 0:
        # GroupNorm of a concatenation = each half normalised over its half of the groups
        h = c // 2
        x1, x2 = xd[..., :h].contiguous(), xd[..., h:].contiguous()
        st1 = ops.gn_stats(x1, gd[:h], bd[:h], groups=g // 2)
        st2 = ops.gn_stats(x2, gd[h:], bd[h:], groups=g // 2)
        ycat = torch.cat([ops.gn_apply(x1, st1, act), ops.gn_apply(x2, st2, act)], dim=-1)
        assert rel_l2(ycat.permute(0, 3, 1, 2), y) < 2e-6


@pytest.mark.parametrize("b,s,c", [(64, 32, 256), (128, 16, 256), (128, 16, 512), (16, 32, 256), (5, 16, 128)])
@pytest.mark.parametrize("variant", ["plain", "branch", "dropout", "accumulate", "branch_accumulate", "no_act"])
def test_groupnorm_backward_lds_image_kernel(ops, b, s, c, variant):
    """gn_bwd_pipe_kernel (resident workgroups, the next slab's x landing in LDS by global_load_lds while the current one is
    reduced and stored) against the register-resident one-slab kernel, selected through psld_set_gn_bwd_kernel: dx,
    dgamma, dbeta (sums over the batch of the per-image sums) bit for bit; the last image against an fp64 reference.  (Small batches and the variants with a third
    operand stay on the one-slab kernel either way: they check its batched loads of that operand.)"""
    x = (gen(b, s, s, c, seed=60) * 1.5 + 0.3).to(DEV)
    dy = gen(b, s, s, c, seed=61).to(DEV)
    gamma, beta = (1 + 0.2 * gen(c, seed=62)).to(DEV), (0.1 * gen(c, seed=63)).to(DEV)
    act = variant != "no_act"
    kw = {}
    if "branch" in variant:
        kw = {"add": gen(b, s, s, c, seed=64).to(DEV), "add_scale": 0.5}
    if "accumulate" in variant:
        kw["accumulate_dx"] = True
    if variant == "dropout":
        kw.update(drop_p=0.15, seed=1234)
    st = ops.gn_stats(x, gamma, beta)
    base = gen(b, s, s, c, seed=65).to(DEV) if "accumulate" in variant else torch.full_like(x, float("nan"))
    out = {}
    initial = ops.get_gn_bwd_kernel()         # "auto" unless the suite runs under PSLD_GN_BWD_PIPE=0
    try:
        for kind in ("auto", "one_slab"):
            ops.set_gn_bwd_kernel(kind)
            dx = base.clone()
            dg, db = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
            ops.gn_bwd(dy, x, st, gamma, beta, act, dx, dg, db, **kw)
            out[kind] = (dx, dg, db)
    finally:
        ops.set_gn_bwd_kernel(initial)
    for got, want in zip(out["auto"], out["one_slab"]):
        assert torch.equal(got, want)
    dx = out["auto"][0]
    # the last image in fp64 (dropout: the mask read off the forward pass with the same seed)
    n = b - 1
    xr = x[n:n + 1].permute(0, 3, 1, 2).double().cpu().requires_grad_(True)
    y = F.group_norm(xr, min(c // 4, 32), gamma.double().cpu(), beta.double().cpu(), 1e-6)
    if act:
        y = F.silu(y)
    if variant == "dropout":
        keep = ops.gn_apply(x, st, act, drop_p=0.15, seed=1234)[n:n + 1] != 0
        assert 0.83 < keep.float().mean().item() < 0.87
        y = y * keep.permute(0, 3, 1, 2).double().cpu() / 0.85
    y.backward(dy[n:n + 1].permute(0, 3, 1, 2).double().cpu())
    want = xr.grad
    if "branch" in variant:
        want = want + 0.5 * kw["add"][n:n + 1].permute(0, 3, 1, 2).double().cpu()
    if "accumulate" in variant:
        want = want + base[n:n + 1].permute(0, 3, 1, 2).double().cpu()
    assert rel_l2(dx[n:n + 1].permute(0, 3, 1, 2), want) < 1e-5


@pytest.mark.parametrize("b,s,c", [(64, 32, 256), (128, 16, 256), (16, 32, 256), (5, 16, 128), (3, 8, 256), (2, 4, 512), (4, 16, 384)])
@pytest.mark.parametrize("variant", ["plain", "dropout", "no_act", "branch", "accumulate", "branch_accumulate"])
def test_groupnorm_backward_column_sums_of_dx(ops, b, s, c, variant):
    """psld_gn_bwd_nhwc_f32 with colsum_img: dx and the per-image sums bit for bit those of the call without it; the
    per-image column sums of the dx values it STORED (written into a wider buffer) against fp64 sums of that dx - on every
    one-pass kernel (1 / 2 / 4 / 16 items per thread, resident workgroups), without a third operand (closed form of the
    channel sums) and with one (identity branch / previous dx: the stored values summed).  The batch total through both
    reducers (psld_param_reduce2_f32, psld_param_reduce_batch_f32): bitwise each other, fp64-close."""
    assert ops.gn_bwd_colsum_supported(b, s * s, c)
    x = (gen(b, s, s, c, seed=70) * 1.5 + 0.3).to(DEV)
    dy = gen(b, s, s, c, seed=71).to(DEV)
    gamma, beta = (1 + 0.2 * gen(c, seed=72)).to(DEV), (0.1 * gen(c, seed=73)).to(DEV)
    act = variant != "no_act"
    kw = {"drop_p": 0.15, "seed": 99} if variant == "dropout" else {}
    if "branch" in variant:
        kw.update(add=gen(b, s, s, c, seed=74).to(DEV), add_scale=0.5)
    if "accumulate" in variant:
        kw["accumulate_dx"] = True
    base = gen(b, s, s, c, seed=75).to(DEV) if "accumulate" in variant else torch.full_like(x, float("nan"))
    st = ops.gn_stats(x, gamma, beta)
    dx0 = base.clone()
    dg0, db0 = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
    sums0 = ops.gn_bwd(dy, x, st, gamma, beta, act, dx0, dg0, db0, **kw)
    dx1 = base.clone()
    wide = torch.full((b, c + 24), float("nan"), device=DEV)
    per_image = wide[:, 8:8 + c]
    sums1 = ops.gn_bwd(dy, x, st, gamma, beta, act, dx1, colsum_img=per_image, ld_img=c + 24, **kw)
    assert torch.equal(dx1, dx0) and torch.equal(sums1, sums0)
    assert torch.isnan(wide[:, :8]).all() and torch.isnan(wide[:, 8 + c:]).all()
    want_img = dx0.double().sum(dim=(1, 2))
    scale = dx0.double().abs().sum(dim=(1, 2)).clamp_min(1e-30)       # a column sum cancels: compare against the sum of magnitudes
    assert ((per_image.double() - want_img).abs() / scale).max().item() < 2e-6
    # dgamma / dbeta / the bias gradient = sums over the batch: two-job launch vs the table-driven one, and fp64
    total = torch.full((c,), float("nan"), device=DEV)
    ops.param_reduce2(per_image, None, b, c + 24, c, total, None, 0.5)
    assert ((total.double() - 0.5 * want_img.sum(0)).abs() / (0.5 * scale.sum(0))).max().item() < 2e-6
    dg1, db1, tot1, tot2 = (torch.full((c,), float("nan"), device=DEV) for _ in range(4))
    jobs = [ops.param_job(sums1, b, 2 * c, c, db1), ops.param_job(sums1, b, 2 * c, c, dg1, src_off=c),
            ops.param_job(per_image, b, c + 24, c, tot1, tot2, 0.5)]
    rows, blocks = [], 0
    for j in jobs:
        rows += list(j) + [blocks]
        blocks += (c + 63) // 64
    ops.param_reduce_batch(torch.tensor(rows, dtype=torch.int64, device=DEV), len(jobs), blocks)
    assert torch.equal(dg1, dg0) and torch.equal(db1, db0) and torch.equal(tot1, total) and torch.equal(tot2, total)
    assert rel_l2(db0, sums0[:, 0].double().sum(0)) < 1e-6 and rel_l2(dg0, sums0[:, 1].double().sum(0)) < 1e-6


@pytest.mark.parametrize("b,s,c", [(128, 32, 256), (128, 16, 256), (16, 32, 256), (5, 16, 128), (3, 32, 128), (7, 16, 512),
                                   (9, 64, 128), (2, 64, 256)])
@pytest.mark.parametrize("variant", ["plain", "branch", "dropout", "accumulate", "branch_accumulate", "no_act"])
def test_groupnorm_backward_on_whole_rows_by_teams(ops, b, s, c, variant):
    """psld_gn_bwd_team_f32 (64 pixels x 128 channels per workgroup - 128 pixels on 64x64 maps, against the three-pass form
    there -, the workgroups of an image exchanging their
    group partial sums through tagged slots) against the one-slab kernel: dx to fp32 rounding (the group terms are formed
    from fp32-rounded member sums), dgamma / dbeta = sums over all batch x K rows, the per-member column sums of the stored
    dx against fp64; three launches in a row (tag counter, slot parity) bitwise equal; no member timed out."""
    initial = ops.get_gn_bwd_kernel()         # "auto" unless the suite runs under PSLD_GN_BWD_PIPE=0
    ops.set_gn_bwd_kernel("auto")
    try:
        _team_kernel_case(ops, b, s, c, variant)
    finally:
        ops.set_gn_bwd_kernel(initial)


def _team_kernel_case(ops, b, s, c, variant):
    k = ops.gn_bwd_team_rows(b, s * s, c)
    px = 128 if s * s > 1024 else 64
    assert k == s * s // px
    x = (gen(b, s, s, c, seed=80) * 1.5 + 0.3).to(DEV)
    dy = gen(b, s, s, c, seed=81).to(DEV)
    gamma, beta = (1 + 0.2 * gen(c, seed=82)).to(DEV), (0.1 * gen(c, seed=83)).to(DEV)
    act = variant != "no_act"
    kw = {}
    if "branch" in variant:
        kw = {"add": gen(b, s, s, c, seed=84).to(DEV), "add_scale": 0.5}
    if "accumulate" in variant:
        kw["accumulate_dx"] = True
    if variant == "dropout":
        kw.update(drop_p=0.15, seed=1234)
    st = ops.gn_stats(x, gamma, beta)
    base = gen(b, s, s, c, seed=85).to(DEV) if "accumulate" in variant else torch.full_like(x, float("nan"))
    dx0 = base.clone()
    dg0, db0 = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
    try:
        ops.set_gn_bwd_kernel("one_slab")
        assert ops.gn_bwd_team_rows(b, s * s, c) == 0            # the selector switches the team form off
        ops.gn_bwd(dy, x, st, gamma, beta, act, dx0, dg0, db0, **kw)
    finally:
        ops.set_gn_bwd_kernel("auto")
    outs = []
    for _ in range(3):
        dx = base.clone()
        rows = torch.full((b * k, c + 8), float("nan"), device=DEV)
        sums = ops.gn_bwd_team(dy, x, st, gamma, beta, act, dx, colsum_rows=rows, ld_rows=c + 8, **kw)
        outs.append((dx, sums.clone(), rows))
    assert ops.gn_team_errors(x.device) == 0
    ops.check_device_errors(x.device)                            # nothing raised: no workgroup gave up
    for dx, sums, rows in outs[1:]:
        assert torch.equal(dx, outs[0][0]) and torch.equal(sums, outs[0][1]) and torch.equal(rows[:, :c], outs[0][2][:, :c])
    dx, sums, rows = outs[0]
    assert torch.isnan(rows[:, c:]).all()
    scale = dx0.abs().max().item()
    assert (dx - dx0).abs().max().item() < 3e-6 * scale and rel_l2(dx, dx0) < 1e-6
    dg, db = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
    ops.param_reduce2(sums, sums.view(-1)[c:], b * k, 2 * c, c, db, dg)
    assert rel_l2(dg, dg0) < 2e-6 and rel_l2(db, db0) < 2e-6
    want = dx.double().view(b, k, px, c).sum(2).view(b * k, c)
    mag = dx.double().abs().view(b, k, px, c).sum(2).view(b * k, c).clamp_min(1e-30)
    assert ((rows[:, :c].double() - want).abs() / mag).max().item() < 2e-6


def test_integration_md_groupnorm_binding(ops):
    """The GroupNorm + SiLU binding INTEGRATION.md shows a maintainer (raw ctypes calls of psld_gn_stats_nhwc_f32,
    psld_gn_apply_nhwc_f32, psld_gn_bwd_nhwc_f32 and psld_param_reduce2_f32 inside a torch.autograd.Function), run as written,
    against autograd of F.silu(F.group_norm(.)) in fp64."""
    from psld_amd import _lib as L
    lib = L.load()

    class GroupNormSiLU(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, gamma, beta, groups):
            b, h, w, c = x.shape
            s = torch.cuda.current_stream().cuda_stream
            mean, rstd = x.new_empty(b, groups), x.new_empty(b, groups)
            scale, shift = x.new_empty(b, c), x.new_empty(b, c)
            ws = x.new_empty(lib.psld_gn_workspace_bytes(b, h * w, c, groups), dtype=torch.uint8)
            assert lib.psld_gn_stats_nhwc_f32(x.data_ptr(), b, h * w, c, groups, 1e-6, gamma.data_ptr(), beta.data_ptr(),
                                              mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                              ws.data_ptr(), s) == 0
            y = torch.empty_like(x)
            assert lib.psld_gn_apply_nhwc_f32(x.data_ptr(), scale.data_ptr(), shift.data_ptr(), y.data_ptr(), b, h * w, c, 1,
                                              0.0, 0, None, s) == 0
            ctx.save_for_backward(x, gamma, beta, mean, rstd)
            ctx.groups = groups
            return y

        @staticmethod
        def backward(ctx, dy):
            x, gamma, beta, mean, rstd = ctx.saved_tensors
            b, h, w, c = x.shape
            s = torch.cuda.current_stream().cuda_stream
            dx, sums = torch.empty_like(x), x.new_empty(b, 2, c)
            dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
            ws = x.new_empty(lib.psld_gn_workspace_bytes(b, h * w, c, ctx.groups), dtype=torch.uint8)
            assert lib.psld_gn_bwd_nhwc_f32(dy.contiguous().data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                            gamma.data_ptr(), beta.data_ptr(), b, h * w, c, ctx.groups, 1, 0.0, 0, None,
                                            dx.data_ptr(), 0, None, 1.0, sums.data_ptr(), None, 0, ws.data_ptr(), s) == 0
            assert lib.psld_param_reduce2_f32(sums.data_ptr(), sums.data_ptr() + 4 * c, b, 2 * c, c, dbeta.data_ptr(),
                                              dgamma.data_ptr(), 1.0, s) == 0
            return dx, dgamma, dbeta, None

    b, hh, c, groups = 6, 16, 128, 32
    x = (gen(b, hh, hh, c, seed=340) * 1.4 + 0.2).to(DEV).requires_grad_(True)
    gamma = (1 + 0.2 * gen(c, seed=341)).to(DEV).requires_grad_(True)
    beta = (0.1 * gen(c, seed=342)).to(DEV).requires_grad_(True)
    gy = gen(b, hh, hh, c, seed=343).to(DEV)
    y = GroupNormSiLU.apply(x, gamma, beta, groups)
    y.backward(gy)
    xr = x.detach().double().cpu().permute(0, 3, 1, 2).requires_grad_(True)
    gr, br = gamma.detach().double().cpu().requires_grad_(True), beta.detach().double().cpu().requires_grad_(True)
    yr = F.silu(F.group_norm(xr, groups, gr, br, 1e-6))
    yr.backward(gy.double().cpu().permute(0, 3, 1, 2))
    assert rel_l2(y.permute(0, 3, 1, 2), yr) < 2e-6
    assert rel_l2(x.grad.permute(0, 3, 1, 2), xr.grad) < 1e-5
    assert rel_l2(gamma.grad, gr.grad) < 1e-5 and rel_l2(beta.grad, br.grad) < 1e-5


def test_split_k_slab_reductions_in_one_launch(ops):
    """psld_reduce_slabs_batch_f32: many weight gradients' split-K slabs reduced by one launch, bit for bit the per-layer
    psld_reduce_slabs_f32 calls - plain and OIHW-scattering layouts (through LDS: whole and partial 256-channel chunks, 1 and 9
    taps), slab counts 1 ... 11."""
    shapes = [(64, 9, 64, 6, 1, 0.7), (128, 1, 128, 3, 0, 1.0), (64, 9, 128, 11, 1, 1.0), (32, 1, 36, 1, 0, 0.5),
              (128, 9, 64, 3, 1, 2.0), (256, 1, 256, 6, 0, 1.0), (16, 9, 512, 5, 1, 1.0), (8, 9, 36, 2, 1, 1.0), (130, 1, 96, 4, 1, 1.0)]
    jobs, want, outs, keep = [], [], [], []
    for i, (co, taps, ci, ns, layout, alpha) in enumerate(shapes):
        n = co * taps * ci
        slabs = gen(ns, n, seed=300 + i).to(DEV)
        ref = torch.full((n,), float("nan"), device=DEV)
        ops.reduce_slabs(slabs, ns, n, ref, layout=layout, cout=co, taps=taps, cin=ci, alpha=alpha)
        out = torch.full((n,), float("nan"), device=DEV)
        jobs.append(ops.slab_job(slabs, ns, n, out, layout, taps, ci, alpha))
        want.append(ref)
        outs.append(out)
        keep.append(slabs)
    # three column blocks of one [c][3c] slab set (layout 2): the q | k | v weight gradients of one GEMM
    cq, nsq = 64, 5
    wide = gen(nsq, cq, 3 * cq, seed=330).to(DEV)
    for kblk in range(3):
        out = torch.full((cq * cq,), float("nan"), device=DEV)
        jobs.append(ops.slab_job(wide.view(-1)[kblk * cq:], nsq, cq * cq, out, 2, cq, 3 * cq, 0.5))
        ref = wide[0, :, kblk * cq:(kblk + 1) * cq].clone()
        for s_ in range(1, nsq):
            ref += wide[s_, :, kblk * cq:(kblk + 1) * cq]
        want.append((ref * 0.5).reshape(-1))
        outs.append(out)
    keep.append(wide)
    rows, units = [], 0
    for j in jobs:
        u = ops.slab_units(j[2], j[4], j[5], j[6])
        assert u > 0
        rows += list(j) + [units, u]
        units += u
    ops.reduce_slabs_batch(torch.tensor(rows, dtype=torch.int64, device=DEV), len(jobs), units)
    for got, ref in zip(outs, want):
        assert torch.equal(got, ref)
    assert rel_l2(want[1], keep[1].double().sum(0)) < 1e-6


def test_bias_gradients_of_three_projections_in_one_pass(ops):
    """psld_bias_grad_seg_f32: the q | k | v bias gradients from one pass over the [rows][3c] gradient buffer against fp64
    column sums and the three psld_bias_grad_f32 calls on its column slices (the pixel lanes per thread differ with the
    width of the pass, so the fp32 partial sums round differently: compared relative to the sum of magnitudes)."""
    b, hw, c = 6, 64, 256
    d = gen(b, hw, 3 * c, seed=310).to(DEV)
    outs = [torch.full((c,), float("nan"), device=DEV) for _ in range(3)]
    ops.bias_grad_seg(d, 3 * c, b, hw, outs, c, 0.5)
    for k in range(3):
        ref = torch.full((c,), float("nan"), device=DEV)
        ops.bias_grad(d[..., k * c:], 3 * c, b, hw, c, ref, 0.5)
        sl = d[..., k * c:(k + 1) * c].double()
        want, scale = 0.5 * sl.sum(dim=(0, 1)), 0.5 * sl.abs().sum(dim=(0, 1))
        assert ((outs[k].double() - want).abs() / scale).max().item() < 1e-6
        assert ((ref.double() - want).abs() / scale).max().item() < 1e-6


# ---------------------------------------------------------------------------------------------------
# FIR resampling (the reference's native op)
# ---------------------------------------------------------------------------------------------------
def test_upfirdn2d_golden(ops, golden):
    g = golden("fir.npz")
    x, k = torch.from_numpy(g["x"]), g["kasym"]
    for i, (up, dn, p0, p1) in enumerate(g["cases"]):
        y = ops.upfirdn2d_raw(x.to(DEV), k, int(up), int(dn), (int(p0), int(p1)), layout=0)
        np.testing.assert_allclose(y.cpu().numpy(), g[f"y_{i}"], rtol=0, atol=3e-6)
    x2 = torch.from_numpy(g["x2"])
    k2 = O.fir_kernel_2d((1, 3, 3, 1))
    up = ops.upfirdn2d_raw(_nhwc(x2).to(DEV), k2 * 4, 2, 1, (2, 1), layout=1)
    np.testing.assert_allclose(up.permute(0, 3, 1, 2).cpu().numpy(), g["up2"], atol=2e-6)
    dn = ops.upfirdn2d_raw(_nhwc(x2).to(DEV), k2, 1, 2, (1, 1), layout=1)
    np.testing.assert_allclose(dn.permute(0, 3, 1, 2).cpu().numpy(), g["down2"], atol=2e-6)


@pytest.mark.parametrize("up,down,pad", [(2, 1, (2, 1)), (1, 2, (1, 1)), (1, 1, (2, 2)), (3, 2, (0, 3))])
def test_upfirdn2d_backward(ops, up, down, pad):
    x = gen(2, 8, 9, 7, seed=50).requires_grad_(True)
    k = torch.tensor([[1.0, 2.0, -1.0], [0.5, 3.0, 0.25], [-2.0, 1.5, 4.0], [0.1, 0.2, 0.3]])
    y = O.upfirdn2d(x, k, up, down, pad)
    gy = gen(*y.shape, seed=51)
    y.backward(gy)
    for layout in (0, 1):
        gyd = (gy if layout == 0 else _nhwc(gy)).to(DEV)
        dx = ops.upfirdn2d_bwd_raw(gyd, k.numpy(), up, down, pad, (9, 7), layout)
        if layout == 1:
            dx = dx.permute(0, 3, 1, 2)
        assert dx.shape == x.shape
        np.testing.assert_allclose(dx.cpu().numpy(), x.grad.numpy(), rtol=0, atol=5e-6)


def test_fused_bias_act(ops):
    x, b = gen(2, 5, 4, 4, seed=52), gen(5, seed=53)
    ref = F.leaky_relu(x + b.view(1, -1, 1, 1), 0.2) * 2 ** 0.5   # op/fused_act.py:86-94
    y = ops.fused_bias_act(x.to(DEV), b.to(DEV))
    np.testing.assert_allclose(y.cpu().numpy(), ref.numpy(), rtol=1e-6, atol=1e-7)
    # gradient modes (op/fused_bias_act_kernel.cu cases 31 / 32, 11 / 12) against autograd of the CPU path: grad = 1 takes the
    # incoming gradient as input and the forward OUTPUT as refer (op/fused_act.py:27-33)
    xr = x.clone().requires_grad_(True)
    out = F.leaky_relu(xr + b.view(1, -1, 1, 1), 0.2) * 2 ** 0.5
    gy = gen(*x.shape, seed=54)
    out.backward(gy)
    g1 = ops.fused_bias_act(gy.to(DEV), None, refer=out.detach().to(DEV), grad=1)
    np.testing.assert_allclose(g1.cpu().numpy(), xr.grad.numpy(), rtol=1e-6, atol=1e-7)
    assert float(ops.fused_bias_act(gy.to(DEV), None, refer=out.detach().to(DEV), grad=2).abs().max()) == 0.0
    lin = ops.fused_bias_act(gy.to(DEV), None, act=1, refer=out.detach().to(DEV), grad=1, scale=0.5)
    np.testing.assert_allclose(lin.cpu().numpy(), (gy * 0.5).numpy(), rtol=1e-6, atol=1e-7)


# ---------------------------------------------------------------------------------------------------
# pointwise / reductions
# ---------------------------------------------------------------------------------------------------
def test_layout_roundtrip(ops):
    x = gen(3, 6, 5, 7, seed=60)
    y = ops.nchw_to_nhwc(x.to(DEV))
    assert torch.equal(y.cpu(), _nhwc(x))
    assert torch.equal(ops.nhwc_to_nchw(y).cpu(), x)
    x = gen(2, 70, 9, 9, seed=61)
    assert torch.equal(ops.nhwc_to_nchw(ops.nchw_to_nhwc(x.to(DEV))).cpu(), x)


def test_pointwise_and_reductions(ops):
    a, b = gen(1003, seed=62), gen(1003, seed=63)
    out = torch.ones(1003, device=DEV)
    ops.axpby(a.to(DEV), 0.5, b.to(DEV), -2.0, out, accumulate=True)
    np.testing.assert_allclose(out.cpu().numpy(), (a * 0.5 + b * -2.0 + 1).numpy(), rtol=1e-6, atol=1e-6)
    x = gen(4096, seed=64) * 3
    np.testing.assert_allclose(ops.silu(x.to(DEV)).cpu().numpy(), F.silu(x).numpy(), rtol=2e-6, atol=1e-7)
    xr = x.clone().requires_grad_(True)
    F.silu(xr).backward(torch.ones_like(xr) * 1.5)
    np.testing.assert_allclose(ops.silu_bwd(x.to(DEV), torch.full((4096,), 1.5, device=DEV)).cpu().numpy(),
                               xr.grad.numpy(), rtol=3e-6, atol=1e-7)
    m = gen(3 * 50, 70, seed=65)
    cs = torch.empty(3, 70, device=DEV)
    ops.colsum(m.to(DEV), 70, 3, 50, 70, cs)
    assert rel_l2(cs, m.double().reshape(3, 50, 70).sum(1)) < 1e-6
    for L in (64, 256, 100, 512, 1024):
        s = gen(37, L, seed=66) * 4
        y = torch.empty(37, L, device=DEV)
        ops.softmax_rows(s.to(DEV), y, 37, L)
        ref = F.softmax(s.double(), dim=-1)
        assert rel_l2(y, ref) < 1e-6
        gy = gen(37, L, seed=67)
        dx = torch.empty(37, L, device=DEV)
        ops.softmax_rows_bwd(y, gy.to(DEV), dx, 37, L)
        refdx = ref * (gy.double() - (ref * gy.double()).sum(-1, keepdim=True))
        assert rel_l2(dx, refdx) < 1e-5


# measured on MI355X (round 2): max abs 4.8e-4, rel-L2 9.4e-5 (1 ulp of logf at ~1e4 rad moves sin/cos by ~1e-3);
# asserted at 2x that
TIME_EMBED_ABS, TIME_EMBED_REL = 1e-3, 2e-4


def test_time_embedding(ops, golden):
    g = golden("layers.npz")
    t = torch.from_numpy(g["gfp.t"])
    from tests.synth import synth_tensor
    W = synth_tensor("W", (128,), torch.Generator().manual_seed(190))
    y = ops.time_embed(t.to(DEV), W.to(DEV), True)
    ref = torch.from_numpy(g["gfp.y"])
    # arguments reach ~1e4 rad: 1 ulp of logf moves sin/cos by ~1e-3 there; typical error is far lower
    e_abs, e_rel = (y.cpu() - ref).abs().max().item(), rel_l2(y, ref)
    print(f"time embedding vs reference: max abs {e_abs:.3e}, rel-L2 {e_rel:.3e}")
    assert e_abs < TIME_EMBED_ABS and e_rel < TIME_EMBED_REL
    tt = torch.tensor([3.0, 999.0])
    freq = torch.exp(torch.arange(16, dtype=torch.float32) * -(math.log(10000) / 15))
    y = ops.time_embed(tt.to(DEV), freq.to(DEV), False)
    assert rel_l2(y, O.positional_embedding(tt, 32)) < 1e-6


# ---------------------------------------------------------------------------------------------------
# SDE kernels vs golden vectors of the reference
# ---------------------------------------------------------------------------------------------------
def _params(ops, nu=4.01, gamma=0.01, lower=True):
    from psld_amd._lib import SdeParams
    p = SdeParams()
    p.beta_0, p.beta_1, p.nu, p.gamma = 8.0, 8.0, nu, gamma
    p.m_inv = (gamma - nu) ** 2 / 4
    p.numerical_eps = 1e-9
    p.decomp_lower = 1 if lower else 0
    return p


def test_perturb_coeffs_golden(ops, golden):
    g = golden("sde_coeffs.npz")
    ts = torch.from_numpy(g["t"]).to(DEV)
    for i, (nu, ga) in enumerate(g["pairs"]):
        for dm in ("lower", "upper"):
            p = _params(ops, float(nu), float(ga), dm == "lower")
            flag = torch.zeros(1, dtype=torch.int32, device=DEV)
            mm0 = 0.04 / p.m_inv
            co = ops.perturb_coeffs(ts, p, 0.0, mm0, flag).cpu().numpy()
            assert flag.item() == 0
            np.testing.assert_allclose(co[:, 8:11].T, g[f"cov_{i}"], rtol=1e-12)
            np.testing.assert_allclose(co[:, 4:8].T, g[f"coeff_{dm}_{i}"], rtol=1e-11, atol=1e-300)


def test_perturb_nan_flag(ops):
    p = _params(ops)
    p.numerical_eps = -1.0   # forces sqrt of a negative at t ~ 0
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.perturb_coeffs(torch.tensor([1e-5], dtype=torch.float64, device=DEV), p, 0.0, 0.01, flag)
    assert flag.item() == 1


def test_perturb_and_em_golden(ops, golden):
    from psld_amd._lib import EmCoeffs
    g = golden("sde_perturb.npz")
    p = _params(ops)
    sde = O.PSLDOracle()
    x0, eps, t = (torch.from_numpy(g[k]).to(DEV) for k in ("x0", "eps", "t"))
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    co = ops.perturb_coeffs(t, p, 0.0, sde.mm_0, flag)
    z, u, mu = ops.perturb(x0, None, eps, co, p, want_f64=True, want_mu=True)
    np.testing.assert_allclose(u.cpu().numpy(), g["u_hsm"], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(mu.cpu().numpy(), g["mu_hsm"], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(z.cpu().numpy(), g["u_hsm"].astype(np.float32), rtol=1.2e-7, atol=1e-9)
    co = ops.perturb_coeffs(torch.from_numpy(g["t_dsm"]).to(DEV), p, 0.0, 0.0, flag)
    _, u, _ = ops.perturb(x0, torch.from_numpy(g["m0"]).to(DEV), eps, co, p, want_f64=True)
    np.testing.assert_allclose(u.cpu().numpy(), g["u_dsm"], rtol=1e-13, atol=1e-15)
    # reverse SDE with the fake score of the golden file, one sample at a time (scalar t per launch)
    uu = torch.from_numpy(g["u"])
    tt = torch.from_numpy(g["t"])
    for pf, tag in ((0, ""), (1, "_pf")):
        for i in range(uu.shape[0]):
            ti = sde.T - tt[i:i + 1]            # psld.py:348: the reverse SDE lives on T - t
            epsp = (0.1 * uu[i:i + 1].float() + ti.float().view(-1, 1, 1, 1))
            c11, c12, c21, c22 = sde.inv_coeff(sde.cov(0.0, sde.mm_0, ti))
            k = EmCoeffs()
            k.beta = float(sde.beta_t(ti)); k.m_inv, k.gamma, k.nu, k.m = sde.m_inv, sde.gamma, sde.nu, sde.m
            k.c11, k.c12, k.c21, k.c22 = (float(c.float()) for c in (c11, c12, c21, c22))
            k.dt = 0.0; k.score_mode = 0; k.probability_flow = pf
            f, gb = ops.reverse_sde(uu[i:i + 1].to(DEV), epsp.to(DEV), k)
            np.testing.assert_allclose(f.cpu().numpy(), g["f_bar" + tag][i:i + 1], rtol=1e-12, atol=1e-14)
            np.testing.assert_allclose(gb.cpu().numpy(), g["g_bar" + tag][i:i + 1], rtol=1e-14)


def test_sqerr_loss(ops):
    a, b = gen(4, 6, 16, 16, seed=70), gen(4, 6, 16, 16, seed=71)
    loss, grad = ops.sqerr_loss(a.to(DEV), b.to(DEV), True, True, grad_scale=1.0)
    br = b.clone().requires_grad_(True)
    ref = ((a - br) ** 2).mean()
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-6 * ref.item()
    assert rel_l2(grad, br.grad) < 1e-6
    loss, _ = ops.sqerr_loss(a.to(DEV), b.to(DEV), False, False)
    assert abs(loss.item() - ((a - b) ** 2).sum().item()) < 1e-6 * loss.item()


def test_adam_clip_ema(ops):
    n = 100003
    p, g = gen(n, seed=80), gen(n, seed=81) * 0.01
    m, v, ema = torch.zeros(n), torch.zeros(n), p.clone()
    pd, gd, md, vd, ed = (t.to(DEV).clone() for t in (p, g, m, v, ema))
    norm = torch.zeros(1, dtype=torch.float64, device=DEV)
    pr, mr, vr, er = p.clone(), m.clone(), v.clone(), ema.clone()
    for step in (1, 2, 3):
        ops.grad_norm(gd, norm)
        ops.adam_ema(pd, gd, md, vd, ed, norm, 1.0, 2e-4, 0.9, 0.999, 1e-8, 0.0, step, 0.9999)
        (gc,), tot = O.clip_grad_norm([g], 1.0)
        assert abs(norm.item() - tot.item()) < 1e-5 * tot.item()
        pr, mr, vr = O.adam_step(pr, gc, mr, vr, step, 2e-4)
        er = O.ema_update(er, pr, 0.9999)
    np.testing.assert_allclose(pd.cpu().numpy(), pr.numpy(), rtol=5e-7, atol=2e-7)
    np.testing.assert_allclose(ed.cpu().numpy(), er.numpy(), rtol=5e-7, atol=2e-7)
    assert rel_l2(md, mr) < 1e-5 and rel_l2(vd, vr) < 1e-5
    t2 = gen(n, seed=82).to(DEV)
    ref = O.ema_update(t2.cpu(), pd.cpu(), 0.99)
    ops.ema(t2, pd, 0.99)
    np.testing.assert_allclose(t2.cpu().numpy(), ref.numpy(), rtol=0, atol=1e-7)


def test_writer_and_loader_edges(ops, golden):
    """bit-exact against the reference's own outputs (tests/golden/edges.npz: PNGs
    written by save_as_images read back; data_scaler tensors) and against the oracle on more inputs."""
    ge = golden("edges.npz")
    np.testing.assert_array_equal(ops.samples_to_uint8(torch.from_numpy(ge["pred"]).to(DEV)).cpu().numpy(), ge["u8"])
    assert torch.equal(ops.uint8_to_images(torch.from_numpy(ge["img"]).to(DEV)).cpu(), torch.from_numpy(ge["tens"]))
    g = torch.Generator().manual_seed(90)
    x = torch.randn(5, 6, 16, 16, generator=g, dtype=torch.float64) * 0.8
    x[0, 0, 0, :4] = torch.tensor([-1.0, 1.0, 0.0, 0.999999])          # clip edges
    u8 = ops.samples_to_uint8(x.to(DEV))
    np.testing.assert_array_equal(u8.cpu().numpy(), O.samples_to_uint8(x))
    raw = ops.samples_to_uint8(x.to(DEV), is_augmented=False, denorm=False)
    np.testing.assert_array_equal(raw.cpu().numpy(), O.samples_to_uint8(x, False, False))
    img = torch.randint(0, 256, (4, 32, 32, 3), generator=g, dtype=torch.uint8)
    t = ops.uint8_to_images(img.to(DEV))
    assert torch.equal(t.cpu(), O.images_to_tensor(img.numpy()))
    flip = torch.tensor([1, 0, 1, 0], dtype=torch.uint8)
    tf = ops.uint8_to_images(img.to(DEV), flip=flip.to(DEV))
    ref = O.images_to_tensor(img.numpy())
    ref[0], ref[2] = ref[0].flip(-1), ref[2].flip(-1)
    assert torch.equal(tf.cpu(), ref)
    assert torch.equal(ops.uint8_to_images(img.to(DEV), norm=False).cpu(), O.images_to_tensor(img.numpy(), False))


def test_a_team_kernel_timeout_is_raised_on_the_host(ops):
    """The team kernel's poll loop is bounded: a workgroup that gives up raises the error word of the slot buffer
    (norm_act.hip).  The word reaches the user: ops.check_device_errors - called by the training loop once per epoch and by
    bench.py after the timed region - raises while it is set."""
    sync = ops.gn_team_sync(DEV)
    torch.cuda.synchronize()
    ops.check_device_errors(DEV)
    sync[:8].view(torch.int64).fill_(1)                          # what the kernel stores on a timeout
    try:
        with pytest.raises(RuntimeError, match="gn_bwd_team_kernel"):
            ops.check_device_errors(DEV)
        with pytest.raises(RuntimeError, match="gn_bwd_team_kernel"):
            ops.check_device_errors()                            # all devices this process drove
    finally:
        sync[:8].view(torch.int64).fill_(0)
    ops.check_device_errors(DEV)
