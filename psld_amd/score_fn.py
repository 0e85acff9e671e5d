"""NCSN++ score network on MI355X: the reference's module surface, a HIP executor underneath.

Surface kept from the reference:
  * ``NCSNpp(config)`` registered as ``score_fn/ncsnpp`` (ncsnpp.py:35-39), ``forward(x, time_cond)``
    with ``x: f32[B,C,H,W]`` (NCHW) and ``time_cond: f32[B]`` -> ``f32[B,out_ch,H,W]`` (ncsnpp.py:287,438);
  * an ``nn.Module`` whose ``state_dict()`` has exactly the reference's keys, shapes and order
    (``all_modules.<i>.<Sub>.<param>``; conv weights OIHW, ``NIN.W`` as [in,out],
    ``GaussianFourierProjection.W`` frozen) so published checkpoints load with ``strict=True``;
  * ``deepcopy`` works (train_sde.py:41), parameters are ordinary ``nn.Parameter``s usable by any
    optimizer / EMA loop, gradients arrive in ``p.grad`` after ``loss.backward()``.

What is different underneath: no eager ATen graph.  ``forward`` runs a fixed program of
libpsld_hip kernels over NHWC activations; with grad enabled the whole network is ONE
``torch.autograd.Function`` whose backward replays a hand-written tape (dgrad / wgrad / GN-bwd
kernels) and writes parameter gradients straight into one flat fp32 buffer (``p.grad`` are views
of it), which is what the fused clip+Adam+EMA kernel and the RCCL bucket reducer consume.

Supported config branches = the ones the north-star configs take: resblock_type
'biggan', progressive 'none', progressive_input 'none' | 'residual', embedding 'fourier' |
'positional', fir True | False, nonlinearity 'swish'.
"""
from __future__ import annotations

import functools
import math
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .registry import register_module

Tensor = torch.Tensor
_ALIGN = 64  # floats: every parameter starts on a 256-byte boundary inside the flat buffers


# ----------------------------------------------------------------------------------------------
# initialisers (song_sde/layers.py:39-76)
# ----------------------------------------------------------------------------------------------
def default_init(shape, scale: float = 1.0) -> Tensor:
    """variance_scaling(scale, 'fan_avg', 'uniform') with in_axis=1, out_axis=0; scale 0 -> 1e-10."""
    scale = 1e-10 if scale == 0 else scale
    rf = float(np.prod(shape)) / shape[1] / shape[0]
    fan_in, fan_out = shape[1] * rf, shape[0] * rf
    variance = scale / ((fan_in + fan_out) / 2)
    return (torch.rand(*shape) * 2.0 - 1.0) * math.sqrt(3 * variance)


# ----------------------------------------------------------------------------------------------
# parameter holders with the reference's attribute names
# ----------------------------------------------------------------------------------------------
class GaussianFourierProjection(nn.Module):
    """layerspp.py:32-41: fixed random frequencies, requires_grad=False."""

    def __init__(self, embedding_size=256, scale=1.0):
        super().__init__()
        self.W = nn.Parameter(torch.randn(embedding_size) * scale, requires_grad=False)


class _Affine(nn.Module):
    """weight / bias holder: nn.Linear ([out,in]), nn.Conv2d (OIHW), nn.GroupNorm ([C])."""

    def __init__(self, weight: Tensor, bias: Tensor):
        super().__init__()
        self.weight = nn.Parameter(weight)
        self.bias = nn.Parameter(bias)


def _linear(in_dim, out_dim):
    return _Affine(default_init((out_dim, in_dim)), torch.zeros(out_dim))  # ncsnpp.py:99-105


def _conv(in_ch, out_ch, k, init_scale=1.0):
    return _Affine(default_init((out_ch, in_ch, k, k), init_scale), torch.zeros(out_ch))  # layers.py:85-109


def _groupnorm(ch):
    return _Affine(torch.ones(ch), torch.zeros(ch))


class NIN(nn.Module):
    """layers.py:531-540: W is [in, out]."""

    def __init__(self, in_dim, num_units, init_scale=0.1):
        super().__init__()
        self.W = nn.Parameter(default_init((in_dim, num_units), init_scale))
        self.b = nn.Parameter(torch.zeros(num_units))


class ResnetBlockBigGANpp(nn.Module):
    """layerspp.py:212-240 (parameters); forward lives in the executor below."""

    def __init__(self, in_ch, out_ch=None, temb_dim=None, up=False, down=False, dropout=0.1, init_scale=0.0):
        super().__init__()
        out_ch = out_ch if out_ch else in_ch
        self.GroupNorm_0 = _groupnorm(in_ch)
        self.Conv_0 = _conv(in_ch, out_ch, 3)
        if temb_dim is not None:
            self.Dense_0 = _linear(temb_dim, out_ch)
        self.GroupNorm_1 = _groupnorm(out_ch)
        self.Dropout_0 = nn.Dropout(dropout)  # parameter-free; the rate is read by the executor
        self.Conv_1 = _conv(out_ch, out_ch, 3, init_scale)
        self.has_shortcut = in_ch != out_ch or up or down
        if self.has_shortcut:
            self.Conv_2 = _conv(in_ch, out_ch, 1)
        self.in_ch, self.out_ch, self.up, self.down = in_ch, out_ch, up, down


class AttnBlockpp(nn.Module):
    """layerspp.py:62-73."""

    def __init__(self, channels, init_scale=0.0):
        super().__init__()
        self.GroupNorm_0 = _groupnorm(channels)
        self.NIN_0 = NIN(channels, channels)
        self.NIN_1 = NIN(channels, channels)
        self.NIN_2 = NIN(channels, channels)
        self.NIN_3 = NIN(channels, channels, init_scale=init_scale)
        self.channels = channels


class Downsample(nn.Module):
    """layerspp.py:129-147 with with_conv=True: fir -> up_or_down_sampling.Conv2d named Conv2d_0,
    else conv3x3(stride 2, pad 0) named Conv_0."""

    def __init__(self, in_ch, out_ch, fir):
        super().__init__()
        if fir:
            self.Conv2d_0 = _conv(in_ch, out_ch, 3)
        else:
            self.Conv_0 = _conv(in_ch, out_ch, 3)
        self.fir, self.in_ch, self.out_ch = fir, in_ch, out_ch

    @property
    def conv(self):
        return self.Conv2d_0 if self.fir else self.Conv_0


# ----------------------------------------------------------------------------------------------
# executor
# ----------------------------------------------------------------------------------------------
class _Node:
    """``gp``: GroupNorm partial sums of ``v`` left behind by the limb kernel that produced it (ops.gn_part_buffer),
    or None: the GroupNorm that reads the node then skips its statistics pass over the tensor.
    ``used``: a consumer has read the node (forward order): the FIRST consumer's backward is the last writer of ``g``.
    ``want_gsum``: the producer has a bias whose gradient is the column sum of ``g``; ``gsum`` [b][c]: those sums per
    image, left by the last writer when it was a one-pass GroupNorm backward (_Exec.gn_backward), else None."""
    __slots__ = ("v", "g", "gp", "used", "want_gsum", "gsum")

    def __init__(self, v, gp=None, want_gsum=False):
        self.v = v
        self.g = None
        self.gp = gp
        self.used = False
        self.want_gsum = want_gsum
        self.gsum = None


class _CatNode:
    """torch.cat([a, b], dim=channels) that is never materialised: the consuming residual block reads both sources
    (two-source convolution kernels, per-source GroupNorm over each source's share of the groups) and writes the
    gradients straight into the sources' buffers."""
    __slots__ = ("a", "b")

    def __init__(self, a: _Node, b: _Node):
        self.a = a
        self.b = b


_OVERLAP_MAX_PIXELS = 65536     # batch x H x W at the input resolution up to which weight gradients go to a side stream
_SLAB_FLUSH_BYTES = 1 << 30     # parked split-K slabs are reduced (one launch) once they pass this many bytes


def _gbuf(node: _Node):
    """Gradient buffer of a node and whether it already holds a partial sum."""
    if node.g is None:
        node.g = torch.empty_like(node.v)
        return node.g, False
    return node.g, True


_RESIDENT_BLOCKS = 512  # 256 CUs x 2 workgroups (64-72 KB LDS each) of the tile kernels


@functools.lru_cache(maxsize=None)
def _pick_nsplit(tiles: int, k: int, min_k: int = 256, resident: int = _RESIDENT_BLOCKS) -> int:
    """Split-K factor of a weight-gradient GEMM: fill whole rounds of resident workgroups (a 1.5-round
    grid wastes a quarter of the machine) while keeping >= ``min_k`` K per split."""
    max_split = max(1, k // min_k)
    best, best_eff = 1, 0.0
    for ns in range(1, min(max_split, 128) + 1):
        blocks = tiles * ns
        rounds = -(-blocks // resident)
        eff = blocks / (rounds * resident)
        if eff > best_eff + 0.02:
            best, best_eff = ns, eff
    return best


def _fir_kernel(k) -> np.ndarray:
    k = np.asarray(k, dtype=np.float32)
    k = np.outer(k, k)
    k /= np.sum(k)
    return k


class _Exec:
    """One forward (and, if ``record``, the tape of its backward) over NHWC tensors."""

    def __init__(self, net: "NCSNpp", record: bool):
        self.net = net
        self.tape = [] if record else None
        self.record = record
        sf = net.sf
        self.s = ops.INV_SQRT2 if sf.skip_rescale else 1.0
        self.drop_p = float(sf.dropout) if net.training else 0.0  # nn.Dropout: active in train mode
        # Dropout masks are derived in the kernels from (seed, element index).  The per-pass part of the seed stays in
        # DEVICE memory (one int64 drawn on the device: no host synchronisation per forward, and a hipGraph-captured
        # training step - which supplies its own static seed word, refreshed before every replay - draws fresh masks);
        # the per-block part is a host constant.
        self.seed_dev = None
        if self.drop_p > 0:
            self.seed_dev = net._dropout_seed_dev if net._dropout_seed_dev is not None else \
                torch.randint(0, 2 ** 62, (1,), device=net._params()[0].device, dtype=torch.int64)
        self.n_drop = 0
        fk = tuple(sf.fir_kernel) if sf.fir else (1, 1)
        self.k_down = _fir_kernel(fk)
        self.k_up = self.k_down * 4.0
        p = self.k_down.shape[0] - 2
        self.pad_up = ((p + 1) // 2 + 1, p // 2)       # up_or_down_sampling.py:222-224
        self.pad_down = ((p + 1) // 2, p // 2)         # :255-257
        self.temb_act: Optional[_Node] = None
        self.watermark = None  # callable(flat_offset) for the DDP reducer
        # weight / bias gradients are off the dependency chain of backward: they run on a side HIP stream so
        # that their MFMA-bound kernels overlap the HBM-bound kernels of the chain (GN backward, reductions)
        # (decided in _run, once the batch is known: net.overlap_wgrad None = automatic)
        self.side = None
        self.side_queue = []        # (fn, tensors) waiting for the next fork
        self.side_group = net.side_group
        self.want_dx = False        # gradient w.r.t. the network input requested (x.requires_grad)
        self.split = ops.math_mode() == "bf16x6"   # 3x3 convs on the bf16 limb kernels (csrc/conv_split.hip)
        import os as _os
        self.limb_planes = _os.environ.get("PSLD_LIMB_PLANES", "1") != "0"    # A/B switch for tools/bench_sample.py
        # forward attention in one kernel (attention.hip) wherever it takes the shape (B=128: 8x8 maps 17 vs 54 us of the
        # three-kernel path, 16x16 maps 62-65 vs 72 us, tools/bench_attn.py); PSLD_FUSED_ATTN=0: the three kernels
        self.fused_attn = _os.environ.get("PSLD_FUSED_ATTN", "1") != "0"
        # bias / time-embedding gradients as column sums a one-pass GroupNorm backward forms of the dx it writes
        # (ops.gn_bwd colsum_img) instead of a column-sum pass over that tensor (PSLD_GN_BWD_COLSUM=0: the passes)
        self.gn_bwd_colsum = _os.environ.get("PSLD_GN_BWD_COLSUM", "1") == "1"
        self.dx_nchw = None
        # Parameter gradients that are reductions over the batch (GroupNorm dgamma / dbeta, bias gradients) or over split-K
        # slabs are not on the dependency chain of backward: their inputs are parked in two persistent arenas and reduced
        # by ONE launch per kind at the end of the pass (net.defer_param_grads; earlier when a gradient bucket is about to
        # be exchanged, or when the parked slabs pass _SLAB_FLUSH_BYTES)
        self.defer = bool(net.defer_param_grads) and record
        self.dense_batched = False
        self.dense_ok = False
        self.dense_pending = []         # (first column of dtp_all, C_out, Dense_0) of blocks whose Dense_0 gradient is parked
        self.pjobs, self.pblocks = [], 0
        self.sjobs, self.sitems, self.sbytes = [], 0, 0

    # -- helpers ------------------------------------------------------------------------------
    def push(self, fn, module=None):
        if self.tape is not None:
            self.tape.append((fn, module))

    def g(self, p: nn.Parameter) -> Tensor:
        return self.net._grad_view(p)

    def on_side(self, fn, *tensors):
        """Run ``fn`` (kernel launches only) on the side stream, ordered after everything queued so far
        on the compute stream.  ``tensors`` are inputs that the compute stream may free afterwards.
        The work is forked in groups of ``side_group`` calls: one event + one stream wait per group instead of per call
        (a cross-stream edge costs ~3.5 us inside a captured graph and ~10 us of host time outside; measured with
        tools/graph_cross.py).  Until its group is launched a call keeps its inputs alive by reference."""
        if self.side is None:
            fn()
            return
        self.side_queue.append((fn, tensors))
        if len(self.side_queue) >= self.side_group:
            self.flush_side()

    def flush_side(self):
        if not self.side_queue:
            return
        queue, self.side_queue = self.side_queue, []
        ev = torch.cuda.Event()
        cur = torch.cuda.current_stream()
        ev.record(cur)
        self.side.wait_event(ev)
        with torch.cuda.stream(self.side), ops.stream_scope():
            for fn, _ in queue:
                fn()
        for _, tensors in queue:
            for t in tensors:
                if t is not None:
                    t.record_stream(self.side)

    def join_side(self):
        if self.side is not None:
            self.flush_side()
            cur = torch.cuda.current_stream()
            cur.wait_stream(self.side)

    # -- deferred reductions ------------------------------------------------------------------------------------
    @staticmethod
    def use(node: _Node) -> bool:
        """Mark ``node`` as read by a consumer; True for the first one (forward order) - its backward runs last among
        the consumers', so whatever it writes into ``node.g`` last is the final gradient."""
        first = not node.used
        node.used = True
        return first

    def defer_param(self, src: Tensor, rows: int, ld: int, c: int, dst1: Tensor, dst2: Optional[Tensor] = None,
                    alpha: float = 1.0, src_off: int = 0):
        """dst1 (and dst2) [c] = alpha * sum over ``rows`` rows of ``src`` (row stride ld): now, or with every other such
        reduction of the pass in one launch (flush_params)."""
        if not self.defer:
            ops.param_reduce2(src if src_off == 0 else src.view(-1)[src_off:], None, rows, ld, c, dst1, None, alpha)
            if dst2 is not None:
                ops.axpby(dst1, 1.0, None, 0.0, dst2)
            return
        self.pjobs.append(ops.param_job(src, rows, ld, c, dst1, dst2, alpha, src_off) + (self.pblocks,))
        self.pblocks += (c + 63) // 64

    def flush_params(self):
        if self.pjobs:
            jobs, blocks = self.pjobs, self.pblocks
            self.pjobs, self.pblocks = [], 0
            rows = [v for job in jobs for v in job]
            table = self.net._tables.get(rows, self.net._params()[0].device)
            ops.param_reduce_batch(table, len(jobs), blocks, sum(4 * (j[1] + 1) * j[3] for j in jobs))

    def slabs_for(self, nbytes: int, device) -> Tensor:
        """Split-K slab storage: the stream's workspace when the reduction follows at once, else a slice of the slab arena
        that stays untouched until flush_slabs."""
        if not self.defer:
            return ops.workspace(nbytes, device)
        return self.net._slab_arena().alloc(nbytes)

    def reduce_slabs(self, slabs: Tensor, nsplit: int, n: int, out: Tensor, layout: int = 0, cout: int = 1, taps: int = 1,
                     cin: int = 1, alpha: float = 1.0, more: bool = False):
        """``more``: further jobs on the SAME slab allocation follow (no flush - which rewinds the arena - in between)."""
        units = ops.slab_units(n, layout, taps, cin) if self.defer else 0
        if units == 0 or out.data_ptr() % 16:
            ops.reduce_slabs(slabs, nsplit, n, out, layout=layout, cout=cout, taps=taps, cin=cin, alpha=alpha)
            return
        self.sjobs.append(ops.slab_job(slabs, nsplit, n, out, layout, taps, cin, alpha) + (self.sitems, units))
        self.sitems += units
        self.sbytes += 4 * n * nsplit
        if self.sbytes >= _SLAB_FLUSH_BYTES and not more:
            self.flush_slabs()

    def flush_slabs(self):
        """One reduction launch for every parked weight gradient - on the stream their producers ran on (callers are
        on_side closures) - after which the arena is reused."""
        if self.sjobs:
            jobs, items = self.sjobs, self.sitems
            self.sjobs, self.sitems, self.sbytes = [], 0, 0
            rows = [v for job in jobs for v in job]
            table = self.net._tables.get(rows, self.net._params()[0].device)
            ops.reduce_slabs_batch(table, len(jobs), items, sum(4 * j[2] * (j[1] + 1) for j in jobs))
            self.net._slab_arena().reset()

    def flush_dense(self):
        """Dense_0 weight gradients of the blocks finished since the last flush: ONE GEMM over their (contiguous) columns of
        dtp_all and one scatter into the flat gradient - the per-bucket form of time_embedding.bwd's batched GEMM."""
        pend, self.dense_pending = self.dense_pending, []
        if not pend:
            return
        net, dtp_all, act = self.net, self.dtp_all, self.temb_act.v
        b, total = dtp_all.shape
        kd = act.shape[1]
        lo, hi = min(o for o, _, _ in pend), max(o + c for o, c, _ in pend)
        if hi - lo == sum(c for _, c, _ in pend) and lo % 4 == 0 and ops.gemm_tn_split_supported(hi - lo, kd, b):
            dwcat = net._persist("dwcat", (total, kd))
            ops.gemm_tn_split(hi - lo, kd, b, dtp_all[:, lo:hi], total, act, kd, dwcat[lo:hi], kd, 1)
            rows, first = [], 0
            for o, c, d0 in pend:
                n4 = c * kd // 4
                rows += [dwcat.data_ptr() + 4 * o * kd, self.g(d0.weight).data_ptr(), n4, first]
                first += n4
            ops.copy_batch(net._tables.get(rows, dwcat.device), len(pend), first)
        else:
            for o, c, d0 in pend:
                if o % 4 == 0 and ops.gemm_tn_split_supported(c, kd, b):          # 16-byte aligned column slice
                    ops.gemm_tn_split(c, kd, b, dtp_all[:, o:o + c], total, act, kd, self.g(d0.weight), kd, 1)
                else:
                    ops.gemm_raw(1, 0, c, kd, b, dtp_all[:, o:o + c], total, 0, act, kd, 0, self.g(d0.weight), kd, 0)

    def flush_deferred(self):
        self.on_side(self.flush_slabs)
        self.flush_params()
        self.flush_dense()

    def finish_backward(self):
        """After the last entry of the backward tape: the parked reductions, then the side stream joins."""
        if self.defer:
            self.flush_deferred()
        self.join_side()

    def gn_backward(self, dy: Tensor, x: Tensor, st, gamma: Tensor, beta: Tensor, dgamma: Tensor, dbeta: Tensor, act: bool,
                    dx: Tensor, accumulate_dx: bool = False, add: Optional[Tensor] = None, add_scale: float = 1.0,
                    drop_p: float = 0.0, seed: int = 0, seed_dev=None, groups: Optional[int] = None,
                    colsum_img: Optional[Tensor] = None, ld_img: int = 0, last_writer_of: Optional[_Node] = None):
        """GroupNorm(+SiLU, dropout) backward; dgamma / dbeta follow from its per-image sums by a deferred reduction.
        ``last_writer_of``: the node whose gradient ``dx`` is, when this call writes it last: if the node's producer wants
        the column sums of that gradient (a bias gradient) and a one-pass kernel takes the shape, they are formed here."""
        b, h, w, c = x.shape
        pa = self.net._param_arena()
        node = last_writer_of
        # large maps with a third operand - and maps above 32x32 in any case - go through the whole-row team kernel unless
        # per-IMAGE column sums are asked for: its sums / column sums come per (image, team member), k rows per image
        k = ops.gn_bwd_team_wanted(b, h * w, c, groups, add is not None or accumulate_dx) if colsum_img is None else 0
        fold = node is not None and node.want_gsum and colsum_img is None and self.gn_bwd_colsum and \
            (k > 0 or ops.gn_bwd_colsum_supported(b, h * w, c, groups))
        rows = b * max(k, 1)
        sums = pa.floats(rows, 2, c)
        if fold:
            colsum_img, ld_img = pa.floats(rows, c), c
        if k:
            ops.gn_bwd_team(dy, x, st, gamma, beta, act, dx, accumulate_dx=accumulate_dx, drop_p=drop_p, seed=seed,
                            groups=groups, add=add, add_scale=add_scale, seed_dev=seed_dev, sums=sums, colsum_rows=colsum_img,
                            ld_rows=ld_img)
        else:
            ops.gn_bwd(dy, x, st, gamma, beta, act, dx, accumulate_dx=accumulate_dx, drop_p=drop_p, seed=seed, groups=groups,
                       add=add, add_scale=add_scale, seed_dev=seed_dev, sums=sums, colsum_img=colsum_img, ld_img=ld_img)
        self.defer_param(sums, rows, 2 * c, c, dbeta)
        self.defer_param(sums, rows, 2 * c, c, dgamma, src_off=c)
        if fold:
            node.gsum = colsum_img

    def bias_from(self, node: _Node, dout: Tensor, bias: nn.Parameter, alpha: float = 1.0,
                  bias2: Optional[nn.Parameter] = None) -> bool:
        """Bias gradient(s) = alpha * column sums of a node's final gradient ``dout``: from the sums its last writer left
        (True: nothing launched now), else by a pass over ``dout`` (False; call it where that pass may run)."""
        gsum, node.gsum = node.gsum, None
        if gsum is not None:
            self.defer_param(gsum, gsum.shape[0], gsum.shape[1], gsum.shape[1], self.g(bias),
                             self.g(bias2) if bias2 is not None else None, alpha)
            return True
        return False

    def wgrad(self, dy: Tensor, x: Tensor, conv: _Affine, k: int, stride: int, pad: int, alpha: float = 1.0,
              x2: Optional[Tensor] = None):
        """dW (OIHW, scaled by alpha) of ``conv`` from the output gradient and the conv's input: split-K partial
        slabs in the stream's workspace, then one deterministic reduction straight into the flat gradient.
        ``x2``: second source when the input is an unmaterialised concatenation (limb kernels only)."""
        b, oh, ow, cout = dy.shape
        c1 = x.shape[-1]
        c2 = x2.shape[-1] if x2 is not None else 0
        cin = c1 + c2
        taps = k * k
        n = cout * taps * cin
        assert x2 is None or self.split, "two-source weight gradients exist on the limb kernels only"
        if self.split and k == 3 and stride == 1 and pad == 1 and not isinstance(x, ops.LimbPlanes) and \
                ops.conv3x3_wgrad_wino_wanted(cout, c1, c2, b, oh, ow):
            # Winograd domain (wgrad_wino.hip): 16 limb products per 2x2 tile instead of 36; its own slabs and reduction
            # (G^T . G over 16 positions), written straight into the flat gradient
            # (the stream's workspace: the reduction follows at once, nothing is parked in the slab arena)
            ops.conv3x3_wgrad_wino(dy, cout, x, self.g(conv.weight), x2=x2, alpha=alpha)
            return
        if self.split and k == 3 and stride == 1 and pad == 1 and \
                ops.conv3x3_wgrad_split_supported(cout, c1, b, oh, ow) and \
                (x2 is None or ops.conv3x3_wgrad_split_supported(cout, c2, b, oh, ow)):
            ktiles = b * oh * ow // 32
            # resident workgroups: 64-channel tiles 3 per CU (46 KB LDS); 128-channel tiles 2 per CU with x as limb planes
            # (dwgrad_kernel<4, true>), ONE 512-thread workgroup per CU for fp32 x (the wave-specialised dwgrad_ws_kernel)
            co_tile = ops.conv3x3_wgrad_split_cout_tile(cout)
            resident = 768 if co_tile == 64 else (512 if isinstance(x, ops.LimbPlanes) else 256)
            nsplit = _pick_nsplit((cout // co_tile) * (cin // 64) * 3, ktiles * 32, min_k=128, resident=resident)
            per = -(-ktiles // nsplit)
            nsplit = -(-ktiles // per)                 # every slab non-empty
            slabs = self.slabs_for(4 * n * nsplit, dy.device)
            ops.conv3x3_wgrad_split(dy, cout, x, slabs, cin, 0, nsplit, x2)
            self.reduce_slabs(slabs, nsplit, n, self.g(conv.weight), layout=1, cout=cout, taps=taps, cin=cin, alpha=alpha)
            return
        if self.split and k == 1 and stride == 1 and pad == 0 and ops.gemm_tn_split_supported(cout, c1, b * oh * ow) and \
                c2 % 128 == 0:
            nsplit = self._tn_split(cout, cin, b * oh * ow)
            slabs = self.slabs_for(4 * n * nsplit, dy.device)
            ops.gemm_tn_split(cout, c1, b * oh * ow, dy, cout, x, c1, slabs, cin, nsplit, x2, c2, c2)
            self.reduce_slabs(slabs, nsplit, n, self.g(conv.weight), alpha=alpha)
            return
        assert x2 is None, "unsupported two-source weight gradient"
        tiles = ((cout + 127) // 128) * ((cin + 127) // 128) * taps
        nsplit = _pick_nsplit(tiles, b * oh * ow)
        slabs = self.slabs_for(4 * n * nsplit, dy.device)
        ops.conv2d_wgrad_nhwc(dy, cout, x, k, k, stride, pad, oh, ow, slabs, cin, 0, nsplit)
        self.reduce_slabs(slabs, nsplit, n, self.g(conv.weight), layout=1, cout=cout, taps=taps, cin=cin, alpha=alpha)

    @staticmethod
    def node_stats(node: _Node, gamma: Tensor, beta: Tensor, groups: Optional[int] = None):
        """GroupNorm statistics of a node: from the producer's partial sums when it left some and the group size is a
        multiple of their 8 channels, else by a pass over the tensor."""
        c = node.v.shape[-1]
        g = groups if groups is not None else ops.gn_groups(c)
        if node.gp is not None and (c // g) % getattr(node.gp, "fine_width", 8) == 0:
            return ops.gn_stats_from_part(node.gp, node.v.shape, gamma, beta, groups=groups)
        return ops.gn_stats(node.v, gamma, beta, groups=groups)

    def part_for(self, b: int, hw: int, c: int, device, limb_kernel: bool):
        """Partial-sum buffer for the epilogue of a limb kernel writing a [b, hw, c] output (None: not applicable)."""
        if self.split and limb_kernel and ops.gn_part_supported(b, hw, c):
            return ops.gn_part_buffer(b, hw, c, device)
        return None

    def bmm(self, ta: int, tb: int, M: int, N: int, K: int, A: Tensor, lda: int, sa: int, B: Tensor, ldb: int, sb: int,
            Cc: Tensor, ldc: int, sc: int, batch: int, alpha: float = 1.0):
        """Batched activation x activation product (attention): limb kernel when the shape allows, fp32 engine otherwise."""
        if self.split and ops.bgemm_split_supported(ta, tb, M, N, K) and A.data_ptr() % 16 == 0 and B.data_ptr() % 16 == 0:
            ops.bgemm_split(ta, tb, M, N, K, A, lda, sa, B, ldb, sb, Cc, ldc, sc, batch, alpha)
        else:
            ops.gemm_raw(ta, tb, M, N, K, A, lda, sa, B, ldb, sb, Cc, ldc, sc, batch,
                         ops.epilogue(alpha=alpha) if alpha != 1.0 else None)

    @staticmethod
    def _tn_split(m: int, n: int, k: int) -> int:
        """K ranges of a pointwise limb weight gradient (128x128 tiles, two workgroups resident per CU)."""
        ktiles = k // 32
        nsplit = _pick_nsplit((m // 128) * (n // 128), k, min_k=128, resident=512)
        per = -(-ktiles // nsplit)
        return -(-ktiles // per)                       # every slab non-empty

    def bias_grad(self, dy: Tensor, out: Tensor, alpha: float = 1.0, per_image=None, ld: Optional[int] = None,
                  ld_per_image: int = 0):
        """``per_image``: True (or a [b, c] tensor) to also get the per-image sums back; ``ld``: row stride of
        ``dy`` when it is a column slice of a wider buffer."""
        b = dy.shape[0]
        c = dy.shape[-1]
        hw = dy.numel() // (b * c)
        ldx = ld if ld is not None else c
        if c % 4 == 0 and c <= 1024 and ldx % 4 == 0 and dy.data_ptr() % 16 == 0:
            if per_image is True:
                per_image = torch.empty((b, c), device=dy.device, dtype=torch.float32)
            ops.bias_grad(dy, ldx, b, hw, c, out, alpha, per_image, ld_per_image)
            return per_image
        assert ld_per_image in (0, c), "strided per-image sums need the vector path (c % 4 == 0)"
        tmp = per_image if isinstance(per_image, Tensor) else torch.empty((b, c), device=dy.device, dtype=torch.float32)
        ops.colsum(dy, ldx, b, hw, c, tmp)
        ops.colsum(tmp, c, 1, b, c, out, alpha)
        return tmp

    def wino_wanted(self, c1: int, c2: int, b: int, h: int, w: int, cout: int) -> bool:
        """ops.conv3x3_wino_wanted counting the launches that fill the chip only with their channel chunks split over workgroups
        (the 8x8 level at B=128, the 16x16 level at B=16)."""
        return ops.conv3x3_wino_wanted(c1, c2, b, h, w, cout, True)

    def conv3(self, x: Tensor, conv: _Affine, out: Tensor, epi, x2: Optional[Tensor] = None):
        """3x3 stride-1 pad-1 convolution of an NHWC tensor (or of the channel concatenation of x and x2)."""
        b, h, w, c = x.shape
        c2 = x2.shape[-1] if x2 is not None else 0
        cout = conv.weight.shape[0]
        if self.split and not isinstance(x, ops.LimbPlanes) and self.wino_wanted(c, c2, b, h, w, cout):
            ops.conv3x3_wino(x, x2, self.net._wfrag(conv, False), cout, out, epi, allow_split=True)   # Winograd F(2x2, 3x3)
        elif self.split and ops.conv3x3_split_supported(c, c2, b, h, w, cout):
            ops.conv3x3_split(x, x2, self.net._frag(conv, False), cout, out, epi)
        else:
            ops.conv2d_nhwc(x, x2, self.net._packed(conv), cout, 3, 3, 1, 1, 1, h, w, out, epi)

    # -- few-channel 3x3 convolutions (6-channel stem / first pyramid level in, 6-channel head out) as K = 64 GEMMs ----
    # K = 9*6 = 54 does not fit the tile engine's 32-channel chunking, so these convolutions used its scalar-gather
    # fallback (1.3 % of a step for 0.1 % of the FLOPs); an explicit im2col of the few-channel tensor (33 MB at
    # B = 128) turns them into fast-path GEMMs.
    def small_in_weights(self, conv: _Affine) -> Tensor:
        """[cout][64]: the OIHW rows (ci*9 + tap, 54 values) zero-padded."""
        co, ci = conv.weight.shape[0], conv.weight.shape[1]

        def build(prev):
            out = prev if prev is not None else torch.zeros((co, 64), device=conv.weight.device, dtype=torch.float32)
            ops.scale_copy2d(conv.weight.detach(), ci * 9, out, 64, co, ci * 9)
            return out
        return self.net._gfrag(conv.weight, "small_in", build)

    def small_in_conv(self, x: Tensor, conv: _Affine, stride: int, pad: int, oh: int, ow: int, out: Tensor, epi):
        """Forward of a 3x3 convolution whose input has <= 7 channels; returns the im2col matrix for the wgrad."""
        cols = ops.im2col3x3_small(x, oh, ow, stride, pad)
        m, cout = cols.shape[0], conv.weight.shape[0]
        ops.gemm_raw(0, 1, m, cout, 64, cols, 64, 0, self.small_in_weights(conv), 64, 0, out, cout, 0, 1, epi)
        return cols

    def small_in_wgrad(self, dy: Tensor, cols: Tensor, conv: _Affine, alpha: float = 1.0):
        cout, k = conv.weight.shape[0], conv.weight.shape[1] * 9
        m = cols.shape[0]
        nsplit = _pick_nsplit(((cout + 127) // 128), m)
        slabs = ops.workspace(4 * cout * 64 * (nsplit + 1), dy.device)
        ops.gemm_tn_splitk(cout, 64, m, dy, cout, cols, 64, slabs, nsplit)
        tmp = slabs.view(torch.float32)[nsplit * cout * 64:(nsplit + 1) * cout * 64]
        ops.reduce_slabs(slabs, nsplit, cout * 64, tmp)
        ops.scale_copy2d(tmp, 64, self.g(conv.weight), k, cout, k, alpha)

    def small_out_backward(self, dy: Tensor, a: Tensor, conv: _Affine, da: Tensor):
        """Data and weight gradient of a 3x3 stride-1 pad-1 convolution with <= 7 OUTPUT channels (the head)."""
        co, ci = conv.weight.shape[0], conv.weight.shape[1]
        b, h, w, _ = dy.shape
        cols = ops.im2col3x3_small(dy, h, w, 1, 1, flip=True)            # [M][64], column = co*9 + tap (mirrored)
        m = cols.shape[0]

        def build(prev):                                                  # [ci][co*9 + tap] = w[co][ci][tap]
            out = prev if prev is not None else torch.zeros((ci, 64), device=conv.weight.device, dtype=torch.float32)
            for o in range(co):
                ops.scale_copy2d(conv.weight.detach(), 9, out, 64, ci, 9, src_off=o * ci * 9, dst_off=o * 9)
            return out
        wd = self.net._gfrag(conv.weight, "small_out", build)
        ops.gemm_raw(0, 1, m, ci, 64, cols, 64, 0, wd, 64, 0, da, ci, 0)

        def side():
            nsplit = _pick_nsplit((ci + 127) // 128, m)
            slabs = ops.workspace(4 * 64 * ci * (nsplit + 1), dy.device)
            ops.gemm_tn_splitk(64, ci, m, cols, 64, a, ci, slabs, nsplit)   # [co*9 + tap][ci]
            tmp = slabs.view(torch.float32)[nsplit * 64 * ci:(nsplit + 1) * 64 * ci]
            ops.reduce_slabs(slabs, nsplit, 64 * ci, tmp)
            ops.reduce_slabs(tmp, 1, co * 9 * ci, self.g(conv.weight), layout=1, cout=co, taps=9, cin=ci)
            self.bias_grad(dy, self.g(conv.bias))
        self.on_side(side, dy, a, cols)

    def dgrad(self, dy: Tensor, conv: _Affine, k: int, stride: int, pad: int, ih: int, iw: int, out: Tensor,
              alpha: float = 1.0, accumulate: bool = False):
        cin = conv.weight.shape[1]
        epi = ops.epilogue(alpha=alpha, accumulate=accumulate) if (alpha != 1.0 or accumulate) else None
        if self.split and k == 3 and stride == 1 and pad == 1 and \
                self.wino_wanted(dy.shape[-1], 0, dy.shape[0], ih, iw, cin):
            ops.conv3x3_wino(dy, None, self.net._wfrag(conv, True), cin, out, epi, allow_split=True)     # Winograd F(2x2, 3x3)
            return
        if self.split and k == 3 and stride == 1 and pad == 1 and \
                ops.conv3x3_split_supported(dy.shape[-1], 0, dy.shape[0], ih, iw, cin):
            ops.conv3x3_split(dy, None, self.net._frag(conv, True), cin, out, epi)
            return
        wd = self.net._packed(conv, dgrad=True)
        ops.conv2d_nhwc(dy, None, wd, cin, k, k, 1, k - 1 - pad, stride, ih, iw, out, epi)

    def resample(self, x: Tensor, up: bool) -> Tensor:
        if up:
            return ops.upfirdn2d_raw(x, self.k_up, 2, 1, self.pad_up, layout=1)
        return ops.upfirdn2d_raw(x, self.k_down, 1, 2, self.pad_down, layout=1)

    def resample_bwd(self, gy: Tensor, up: bool, in_hw, out: Tensor, accumulate: bool):
        if up:
            ops.upfirdn2d_bwd_raw(gy, self.k_up, 2, 1, self.pad_up, in_hw, 1, out=out, accumulate=accumulate)
        else:
            ops.upfirdn2d_bwd_raw(gy, self.k_down, 1, 2, self.pad_down, in_hw, 1, out=out, accumulate=accumulate)

    # -- time embedding (ncsnpp.py:289-313) ------------------------------------------------------
    def time_embedding(self, t: Tensor):
        net = self.net
        mods = net.all_modules
        i = 0
        if net.embedding_type == "fourier":
            emb = ops.time_embed(t, mods[0].W, True)
            i = 1
        else:
            emb = ops.time_embed(t, net._pos_freq(t.device), False)
        if not net.noise_cond:
            self.temb_act = None
            return i
        l1, l2 = mods[i], mods[i + 1]
        t1 = ops.linear(emb, l1.weight, l1.bias)
        s1 = ops.silu(t1)
        temb = ops.linear(s1, l2.weight, l2.bias)
        st = _Node(ops.silu(temb))
        self.temb_act = st
        b = t.shape[0]
        # Dense_0(act(temb)) of EVERY ResBlock in one GEMM against the gathered projection weights
        # (layerspp.py:262-263 runs one small Linear per block); the blocks read column slices of tp_all
        plan = net._temb_plan()
        self.tp_all = self.dtp_all = None
        if plan is not None:
            wcat, bcat, total = plan["wcat"], plan["bcat"], plan["total"]
            self.tp_all = torch.empty((b, total), device=t.device, dtype=torch.float32)
            ops.gemm_raw(0, 1, b, total, wcat.shape[1], st.v, wcat.shape[1], 0, wcat, wcat.shape[1], 0, self.tp_all,
                         total, 0, 1, ops.epilogue(bias=bcat))
            if self.record:
                # persistent (same address every step: the batched reduction tables hold pointers into it); written and
                # read inside ONE backward pass, so forward passes whose backward is still pending can share it
                self.dtp_all = net._persist("dtp_all", (b, total))
        tp_all, dtp_all = self.tp_all, self.dtp_all
        # Dense_0's weight gradients of ALL blocks as one GEMM dtp_all^T act(temb) at the end of the pass (57 eight-workgroup
        # launches of ~10 us otherwise) - unless a bucket reducer needs each block's gradients final at its own watermark
        # With a reducer (and no side stream) the same GEMM runs once per BUCKET, over the columns of the blocks finished since
        # the last one (flush_dense, called with the other parked reductions before a bucket is exchanged).
        self.dense_ok = self.dtp_all is not None and self.defer and self.split and \
            ops.gemm_tn_split_supported(plan["total"], plan["wcat"].shape[1], b)
        self.dense_batched = self.dense_ok and net._reducer is None
        dense_batched = self.dense_batched

        def bwd():
            self.join_side()            # every block wrote its slice of dtp_all / accumulated into st.g
            if dense_batched:
                total, kd = plan["total"], plan["wcat"].shape[1]
                dwcat = net._persist("dwcat", (total, kd))
                ops.gemm_tn_split(total, kd, b, dtp_all, total, st.v, kd, dwcat, kd, 1)
                rows, first = [], 0
                for m_ in plan["blocks"]:
                    o, w_ = plan["offsets"][id(m_)], m_.Dense_0.weight
                    n4 = w_.numel() // 4
                    rows += [dwcat.data_ptr() + 4 * o * kd, self.g(w_).data_ptr(), n4, first]
                    first += n4
                ops.copy_batch(net._tables.get(rows, dwcat.device), len(plan["blocks"]), first)
            if dtp_all is not None:     # d act(temb) = sum over blocks dtp_i W_i = dtp_all Wcat: one GEMM
                wcat = plan["wcat"]
                gb, acc = _gbuf(st)
                total, kd = plan["total"], wcat.shape[1]
                # M = batch is one tile tall and K = sum of the blocks' C_out is long (14592 for C10): cut K into
                # ranges that run as the batches of one launch, then add the partial products in range order
                ks = next((k for k in (256, 128, 64) if total % k == 0), 0)
                if not acc and ks and total // ks >= 8 and (b * kd) % 4 == 0:
                    ns = total // ks
                    slabs = ops.workspace(4 * ns * b * kd, dtp_all.device).view(torch.float32)
                    ops.gemm_raw(0, 0, b, kd, ks, dtp_all, total, ks, wcat, kd, ks * kd, slabs, kd, b * kd, ns)
                    ops.reduce_slabs(slabs, ns, b * kd, gb)
                else:
                    ops.gemm_raw(0, 0, b, kd, total, dtp_all, total, 0, wcat, kd, 0, gb, kd, 0,
                                 epi=ops.epilogue(accumulate=True) if acc else None)
            if st.g is None:
                return
            dtemb = ops.silu_bwd(temb, st.g)
            n2, k2 = l2.weight.shape
            ops.gemm_raw(1, 0, n2, k2, b, dtemb, n2, 0, s1, k2, 0, self.g(l2.weight), k2, 0)
            ops.colsum(dtemb, n2, 1, b, n2, self.g(l2.bias))
            ds1 = torch.empty_like(s1)
            ops.gemm_raw(0, 0, b, k2, n2, dtemb, n2, 0, l2.weight, k2, 0, ds1, k2, 0)
            dt1 = ops.silu_bwd(t1, ds1)
            n1, k1 = l1.weight.shape
            ops.gemm_raw(1, 0, n1, k1, b, dt1, n1, 0, emb, k1, 0, self.g(l1.weight), k1, 0)
            ops.colsum(dt1, n1, 1, b, n1, self.g(l1.bias))

        self.push(bwd, l1)
        return i + 2

    # -- ResnetBlockBigGANpp.forward (layerspp.py:242-274) -------------------------------------------
    def resblock(self, x, mod: ResnetBlockBigGANpp) -> _Node:
        """``x``: a node, or a _CatNode (see concat): then every consumer below reads the two sources side by side."""
        net, s = self.net, self.s
        gn0, gn1 = mod.GroupNorm_0, mod.GroupNorm_1
        xb: Optional[_Node] = None
        if isinstance(x, _CatNode):
            x, xb = x.a, x.b
        first_x = self.use(x)
        first_xb = self.use(xb) if xb is not None else False
        b, h, w, c1 = x.v.shape
        cin = c1 + (xb.v.shape[-1] if xb is not None else 0)
        cout = mod.out_ch
        up, down = mod.up, mod.down
        a0b, st0b, g1, g2 = None, None, None, None
        if xb is None:
            st0 = self.node_stats(x, gn0.weight, gn0.bias)
        else:
            # GroupNorm over the concatenation = each source normalised over its own share of the groups
            cpg = cin // ops.gn_groups(cin)
            g1, g2 = c1 // cpg, (cin - c1) // cpg
            gam, bet = gn0.weight.detach(), gn0.bias.detach()
            st0 = self.node_stats(x, gam[:c1], bet[:c1], groups=g1)
            st0b = self.node_stats(xb, gam[c1:], bet[c1:], groups=g2)
        # The activations go to the 3x3 convolutions as bf16 LIMB PLANES: GroupNorm's apply pass writes them already
        # split (6 B per element instead of 4), the forward convolution stages its halo tile by LDS-DMA with no split in
        # the MFMA kernel and the weight gradient stages its x operand without one (ops.conv3x3_split /
        # conv3x3_wgrad_split on LimbPlanes; both bitwise the fp32-input result).  Blocks that resample between the
        # normalisation and the convolution (up / down) keep fp32.
        ho_, wo_ = (h // 2, w // 2) if down else ((h * 2, w * 2) if up else (h, w))
        c2_ = cin - c1
        # (a convolution that runs in Winograd form transforms fp32 input itself: its producer writes plain fp32)
        lp0 = self.split and self.limb_planes and not (up or down) and not self.wino_wanted(c1, c2_, b, h, w, cout) and \
            ops.conv3x3_split_supported(c1, c2_, b, h, w, cout) and \
            (not self.record or (ops.conv3x3_wgrad_split_supported(cout, c1, b, h, w) and
                                 (c2_ == 0 or ops.conv3x3_wgrad_split_supported(cout, c2_, b, h, w))))
        lp1 = self.split and self.limb_planes and not self.wino_wanted(cout, 0, b, ho_, wo_, cout) and \
            ops.conv3x3_split_supported(cout, 0, b, ho_, wo_, cout) and \
            (not self.record or ops.conv3x3_wgrad_split_supported(cout, cout, b, ho_, wo_))
        apply0 = ops.gn_apply_limb if lp0 else ops.gn_apply
        # Inference forward (no tape): GroupNorm's apply pass + SiLU run inside the Winograd convolution's input staging
        # where that pays (ops.conv3x3_wino_gn_wanted) - the activated tensor is needed nowhere else
        fuse0 = not self.record and not (up or down) and self.split and ops.conv3x3_wino_gn_wanted(c1, c2_, b, h, w, cout)
        fuse1 = not self.record and self.split and self.drop_p == 0 and ops.conv3x3_wino_gn_wanted(cout, 0, b, ho_, wo_, cout)
        a0 = None
        if not fuse0:
            if xb is not None:
                a0b = apply0(xb.v, st0b, True)
            a0 = apply0(x.v, st0, True)
        if fuse0:
            a0r, xr = None, x.v
        elif up or down:
            a0r = self.resample(a0, up)
            xr = self.resample(x.v, up)
            del a0
        else:
            a0r, xr = a0, x.v
        ho, wo = ho_, wo_
        tp, tp_ld, tp_off = None, 0, None
        if self.temb_act is not None:
            tp_off = net._temb_offset(mod) if self.tp_all is not None else None
            if tp_off is not None:
                tp, tp_ld = self.tp_all[:, tp_off:tp_off + cout], self.tp_all.shape[1]
            else:
                tp = ops.linear(self.temb_act.v, mod.Dense_0.weight, mod.Dense_0.bias)
        h1 = torch.empty((b, ho, wo, cout), device=x.v.device, dtype=torch.float32)
        h1p = self.part_for(b, ho * wo, cout, h1.device, ops.conv3x3_split_supported(c1, c2_, b, ho, wo, cout))
        epi0 = ops.epilogue(bias=mod.Conv_0.bias, rowbias=tp, rows_per_img=ho * wo, ld_rowbias=tp_ld, gn_part=h1p, gn_hw=ho * wo)
        if fuse0:
            ops.conv3x3_wino_gn(x.v, st0, xb.v if xb is not None else None, st0b, True, net._wfrag(mod.Conv_0, False), cout,
                                h1, epi0, allow_split=True)
        else:
            self.conv3(a0r, mod.Conv_0, h1, epi0, x2=a0b)
        st1 = self.node_stats(_Node(h1, h1p), gn1.weight, gn1.bias)
        drop_p, seed, seed_dev = 0.0, 0, None
        if self.drop_p > 0:
            drop_p = self.drop_p
            self.n_drop += 1
            seed = (self.n_drop * 0x9E3779B97F4A7C15) & 0x7FFFFFFFFFFFFFFF
            seed_dev = self.seed_dev
        a1 = None if fuse1 else \
            (ops.gn_apply_limb if lp1 else ops.gn_apply)(h1, st1, True, drop_p=drop_p, seed=seed, seed_dev=seed_dev)
        out = torch.empty((b, ho, wo, cout), device=x.v.device, dtype=torch.float32)
        if mod.has_shortcut:
            c2 = mod.Conv_2
            if self.split and ops.gemm_split_supported(c1, cin - c1, b * ho * wo, cout):
                fr = net._pfrag(c2.weight, "fwd", cout, cin, cin, 1)
                ops.gemm_split(xr, xb.v if xb is not None else None, b * ho * wo, fr, cout, out, ops.epilogue(bias=c2.bias))
            else:
                ops.conv2d_nhwc(xr, xb.v if xb is not None else None, c2.weight, cout, 1, 1, 1, 0, 1, ho, wo, out,
                                ops.epilogue(bias=c2.bias))
            res = out
        else:
            res = xr
        outp = self.part_for(b, ho * wo, cout, out.device, ops.conv3x3_split_supported(cout, 0, b, ho, wo, cout))
        epi1 = ops.epilogue(bias=mod.Conv_1.bias, residual=res, ld_residual=cout, out_scale=s, gn_part=outp, gn_hw=ho * wo)
        if fuse1:
            ops.conv3x3_wino_gn(h1, st1, None, None, True, net._wfrag(mod.Conv_1, False), cout, out, epi1, allow_split=True)
        else:
            self.conv3(a1, mod.Conv_1, out, epi1)
        on = _Node(out, outp, want_gsum=True)       # Conv_1.bias (and Conv_2.bias) = s * column sums of its gradient
        if not self.record:
            return on
        temb_act = self.temb_act
        dtp_all = self.dtp_all
        xr_saved = xr if mod.has_shortcut else None
        xb_v = xb.v if xb is not None else None

        def bwd():
            dout = on.g
            on.g = None
            # Conv_1 / Conv_2 bias: s * column sums of dout - left behind by the last writer of dout where that was a
            # one-pass GroupNorm backward, else a pass over dout on the side stream
            have_bias = self.bias_from(on, dout, mod.Conv_1.bias, s, mod.Conv_2.bias if mod.has_shortcut else None)

            # Conv_1 (the 1/sqrt(2) of skip_rescale is folded into alpha); parameter gradients on the side stream
            def side1():
                self.wgrad(dout, a1, mod.Conv_1, 3, 1, 1, alpha=s)
                if not have_bias:
                    self.bias_grad(dout, self.g(mod.Conv_1.bias), alpha=s)
                if mod.has_shortcut:
                    self.wgrad(dout, xr_saved, mod.Conv_2, 1, 1, 0, alpha=s, x2=xb_v)
                    if not have_bias:
                        # Conv_2.bias sees the same output gradient as Conv_1.bias: copy the sum just computed
                        ops.axpby(self.g(mod.Conv_1.bias), 1.0, None, 0.0, self.g(mod.Conv_2.bias))

            self.on_side(side1, dout, a1, xr_saved, xb_v)
            da1 = torch.empty_like(h1)
            self.dgrad(dout, mod.Conv_1, 3, 1, 1, ho, wo, da1, alpha=s)
            dh1 = torch.empty_like(h1)
            # Conv_0's bias gradient and the per-image sums of dh1 (the time-embedding gradient) as a by-product of the
            # GroupNorm backward that writes dh1 (no pass over dh1), where its one-pass kernels take the shape
            csum = self.gn_bwd_colsum and ops.gn_bwd_colsum_supported(b, ho * wo, cout)
            per_img, ldp = None, 0
            if csum:
                if temb_act is not None and tp_off is not None and dtp_all is not None:
                    per_img, ldp = dtp_all[:, tp_off:tp_off + cout], dtp_all.shape[1]
                else:
                    per_img, ldp = net._param_arena().floats(b, cout), cout
            self.gn_backward(da1, h1, st1, gn1.weight, gn1.bias, self.g(gn1.weight), self.g(gn1.bias), True, dh1,
                             drop_p=drop_p, seed=seed, seed_dev=seed_dev, colsum_img=per_img, ld_img=ldp)
            if csum:        # Conv_0.bias = sum over the batch of the per-image sums = Dense_0.bias
                self.defer_param(per_img, b, ldp, cout, self.g(mod.Conv_0.bias),
                                 self.g(mod.Dense_0.bias) if temb_act is not None else None)
            dtp_pre = per_img
            del da1

            # Conv_0 + time-embedding bias
            def side0():
                self.wgrad(dh1, a0r, mod.Conv_0, 3, 1, 1, x2=a0b)
                if temb_act is None:
                    if not csum:
                        self.bias_grad(dh1, self.g(mod.Conv_0.bias))
                    return
                d0 = mod.Dense_0
                kd = d0.weight.shape[1]
                if tp_off is not None and dtp_all is not None:
                    # per-image sums straight into this block's columns of dtp_all; its share of d act(temb) is added
                    # by ONE GEMM over all blocks at the end (time_embedding.bwd)
                    ldt = dtp_all.shape[1]
                    dtp = dtp_all[:, tp_off:tp_off + cout]
                    if not csum:
                        self.bias_grad(dh1, self.g(mod.Conv_0.bias), per_image=dtp, ld_per_image=ldt)
                    if self.dense_batched:
                        pass                # one GEMM over all blocks at the end (time_embedding.bwd)
                    elif self.dense_ok and self.side is None and ops.gemm_tn_split_supported(cout, kd, b):
                        self.dense_pending.append((tp_off, cout, d0))       # one GEMM per gradient bucket (flush_dense)
                    elif self.split and ops.gemm_tn_split_supported(cout, kd, b) and dtp.data_ptr() % 16 == 0:
                        # one "slab" = the gradient itself: K = batch is short enough for a single range
                        ops.gemm_tn_split(cout, kd, b, dtp, ldt, temb_act.v, kd, self.g(d0.weight), kd, 1)
                    else:
                        ops.gemm_raw(1, 0, cout, kd, b, dtp, ldt, 0, temb_act.v, kd, 0, self.g(d0.weight), kd, 0)
                else:
                    dtp = dtp_pre if csum else self.bias_grad(dh1, self.g(mod.Conv_0.bias), per_image=True)
                    ops.gemm_raw(1, 0, cout, kd, b, dtp, cout, 0, temb_act.v, kd, 0, self.g(d0.weight), kd, 0)
                    gb, acc = _gbuf(temb_act)
                    ops.gemm_raw(0, 0, b, kd, cout, dtp, cout, 0, d0.weight, kd, 0, gb, kd, 0,
                                 epi=ops.epilogue(accumulate=True) if acc else None)
                if not csum:
                    # d Dense_0.bias = sum over the batch of dtp = the conv bias gradient just computed
                    ops.axpby(self.g(mod.Conv_0.bias), 1.0, None, 0.0, self.g(d0.bias))

            self.on_side(side0, dh1, a0r, a0b)
            if xb is not None:
                self._resblock_cat_bwd(mod, x, xb, dout, dh1, st0, st0b, g1, g2, first_x, first_xb)
                return
            da0r = torch.empty((b, ho, wo, cin), device=dout.device, dtype=torch.float32)
            self.dgrad(dh1, mod.Conv_0, 3, 1, 1, ho, wo, da0r)
            # (no `del dh1`: side0 above may still be waiting for its fork and looks the name up when it runs)
            xg, acc = _gbuf(x)
            identity = False
            if mod.has_shortcut:
                c2 = mod.Conv_2
                m = b * ho * wo
                def shortcut_dgrad(dst, epi):
                    if self.split and ops.gemm_split_supported(cout, 0, m, cin):
                        fr = net._pfrag(c2.weight, "dgrad", cin, cout, 1, cin)
                        ops.gemm_split(dout, None, m, fr, cin, dst, epi)
                    else:
                        ops.gemm_raw(0, 0, m, cin, cout, dout, cout, 0, c2.weight, cin, 0, dst, cin, 0, epi=epi)

                if up or down:
                    dxr = torch.empty((b, ho, wo, cin), device=dout.device, dtype=torch.float32)
                    shortcut_dgrad(dxr, ops.epilogue(alpha=s))
                    self.resample_bwd(dxr, up, (h, w), xg, acc)
                    del dxr
                else:
                    shortcut_dgrad(xg, ops.epilogue(alpha=s, accumulate=acc))
            else:
                identity = True          # out = (x + h)/sqrt(2): the x branch's gradient s*dout rides on GroupNorm_0's backward
            if up or down:
                da0 = torch.empty((b, h, w, cin), device=dout.device, dtype=torch.float32)
                self.resample_bwd(da0r, up, (h, w), da0, False)
            else:
                da0 = da0r
            # this block read x first (forward order): its GroupNorm_0 backward writes x's gradient last
            self.gn_backward(da0, x.v, st0, gn0.weight, gn0.bias, self.g(gn0.weight), self.g(gn0.bias), True, xg,
                             accumulate_dx=not identity or acc, add=dout if identity else None, add_scale=s,
                             last_writer_of=x if first_x else None)

        self.push(bwd, mod)
        return on

    def _resblock_cat_bwd(self, mod, xa: _Node, xb: _Node, dout: Tensor, dh1: Tensor, sta, stb, g1: int, g2: int,
                          first_a: bool, first_b: bool):
        """Input side of the backward of a residual block fed by an unmaterialised concatenation: the data gradients
        of Conv_0 and of the 1x1 shortcut are computed per source (the fragments of a data gradient are ordered by
        output-channel tile, so each source's share is a contiguous slice) and GroupNorm_0's backward runs per source
        over its groups; everything accumulates straight into the two sources' gradient buffers."""
        net, s = self.net, self.s
        gn0, c2 = mod.GroupNorm_0, mod.Conv_2
        b, h, w, cout = dout.shape
        m = b * h * w
        c1 = xa.v.shape[-1]
        cin = c1 + xb.v.shape[-1]
        wino = self.split and self.wino_wanted(cout, 0, b, h, w, c1) and \
            self.wino_wanted(cout, 0, b, h, w, cin - c1)
        # [cin/128 tiles][...]: data gradient of the 3x3 (Winograd fragments carry 16 KB of read-ahead padding at the end)
        f3 = net._wfrag(mod.Conv_0, True) if wino else net._frag(mod.Conv_0, True)
        f1 = net._pfrag(c2.weight, "dgrad", cin, cout, 1, cin)  # same for the shortcut
        cut3, cut1 = (f3.numel() - (16384 if wino else 0)) * c1 // cin, f1.numel() * c1 // cin
        gam, bet = gn0.weight.detach(), gn0.bias.detach()
        dgam, dbet = self.g(gn0.weight), self.g(gn0.bias)
        for node, lo, hi, fr3, fr1, st, g, first in ((xa, 0, c1, f3[:cut3], f1[:cut1], sta, g1, first_a),
                                                    (xb, c1, cin, f3[cut3:], f1[cut1:], stb, g2, first_b)):
            c = hi - lo
            xg, acc = _gbuf(node)
            ops.gemm_split(dout, None, m, fr1, c, xg, ops.epilogue(alpha=s, accumulate=acc))
            da0 = torch.empty_like(node.v)
            if wino:
                ops.conv3x3_wino(dh1, None, fr3, c, da0, allow_split=True)
            else:
                ops.conv3x3_split(dh1, None, fr3, c, da0)
            self.gn_backward(da0, node.v, st, gam[lo:hi], bet[lo:hi], dgam[lo:hi], dbet[lo:hi], True, xg,
                             accumulate_dx=True, groups=g, last_writer_of=node if first else None)

    # -- AttnBlockpp.forward (layerspp.py:75-91) -----------------------------------------------------
    def attn(self, x: _Node, mod: AttnBlockpp) -> _Node:
        s = self.s
        b, h, w, c = x.v.shape
        hw = h * w
        m = b * hw
        dev = x.v.device
        gn = mod.GroupNorm_0
        first_x = self.use(x)
        st = self.node_stats(x, gn.weight, gn.bias)
        hn = ops.gn_apply(x.v, st, False)
        n0, n1, n2, n3 = mod.NIN_0, mod.NIN_1, mod.NIN_2, mod.NIN_3
        scale = float(int(c) ** (-0.5))
        # limb kernels: q|k|v come from ONE GEMM against the concatenated projections (N = 3c) into one buffer
        fused = self.split and ops.gemm_split_supported(c, 0, m, c)
        net = self.net
        if fused:
            f_qkv, f_qkv_d, b_qkv = net._qkv_frags(mod)
            qkv = torch.empty((b, hw, 3 * c), device=dev, dtype=torch.float32)
            ops.gemm_split(hn, None, m, f_qkv, 3 * c, qkv, ops.epilogue(bias=b_qkv))
            q, k, v = qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:]
            ld = 3 * c
        else:
            qkv = []
            for nin in (n0, n1, n2):
                y = torch.empty((b, hw, c), device=dev, dtype=torch.float32)
                ops.gemm_raw(0, 0, m, c, c, hn, c, 0, nin.W, c, 0, y, c, 0, epi=ops.epilogue(bias=nin.b))
                qkv.append(y)
            q, k, v = qkv
            ld = c
        ho = torch.empty((b, hw, c), device=dev, dtype=torch.float32)
        if self.split and self.fused_attn and ops.attn_fwd_supported(hw, c):
            # QK^T -> softmax -> PV in ONE kernel: the [B, HW, HW] scores never reach HBM; the probabilities are written
            # only when a backward pass will read them
            p = torch.empty((b, hw, hw), device=dev, dtype=torch.float32) if self.record else None
            ops.attn_fwd(q, k, v, ld, b, hw, c, scale, ho, p)
        else:
            p = torch.empty((b, hw, hw), device=dev, dtype=torch.float32)
            self.bmm(0, 1, hw, hw, c, q, ld, hw * ld, k, ld, hw * ld, p, hw, hw * hw, b, scale)
            ops.softmax_rows(p, p, b * hw, hw)
            self.bmm(0, 0, hw, c, hw, p, hw, hw * hw, v, ld, hw * ld, ho, c, hw * c, b)
        out = torch.empty_like(x.v)
        outp = self.part_for(b, hw, c, dev, fused)
        epi_out = ops.epilogue(bias=n3.b, residual=x.v, ld_residual=c, out_scale=s, gn_part=outp, gn_hw=hw)
        if fused:
            f_o = net._pfrag(n3.W, "fwd", c, c, 1, c)
            ops.gemm_split(ho, None, m, f_o, c, out, epi_out)
        else:
            ops.gemm_raw(0, 0, m, c, c, ho, c, 0, n3.W, c, 0, out, c, 0, epi=epi_out)
        on = _Node(out, outp, want_gsum=True)       # NIN_3.b = s * column sums of its gradient
        if not self.record:
            return on

        def nin_wgrad(a_in: Tensor, dy: Tensor, nin: NIN, alpha: float, ldd: int, bias: bool = True):
            # dW[in,out] = a_in^T dy  (K = B*HW -> split-K slabs); dy may be a column slice (row stride ldd)
            if self.split and ops.gemm_tn_split_supported(c, c, m):
                nsplit = self._tn_split(c, c, m)
                slabs = self.slabs_for(4 * c * c * nsplit, dev)
                ops.gemm_tn_split(c, c, m, a_in, c, dy, ldd, slabs, c, nsplit)
            else:
                nsplit = _pick_nsplit(((c + 127) // 128) ** 2, m)
                slabs = self.slabs_for(4 * c * c * nsplit, dev)
                ops.gemm_tn_splitk(c, c, m, a_in, c, dy, ldd, slabs, nsplit)
            self.reduce_slabs(slabs, nsplit, c * c, self.g(nin.W), alpha=alpha)
            if bias:
                self.bias_grad(dy.view(b, hw, 1, c) if ldd == c else dy, self.g(nin.b), alpha=alpha, ld=ldd)

        def bwd():
            dout = on.g
            on.g = None
            have_b3 = self.bias_from(on, dout, n3.b, s)
            self.on_side(lambda: nin_wgrad(ho, dout, n3, s, c, bias=not have_b3), ho, dout)
            dho = torch.empty_like(ho)
            if fused:
                f_od = net._pfrag(n3.W, "dgrad", c, c, c, 1)
                ops.gemm_split(dout, None, m, f_od, c, dho, ops.epilogue(alpha=s))
            else:
                ops.gemm_raw(0, 1, m, c, c, dout, c, 0, n3.W, c, 0, dho, c, 0, epi=ops.epilogue(alpha=s))
            # dP = dho v^T ; dv = P^T dho
            dp = torch.empty_like(p)
            self.bmm(0, 1, hw, hw, c, dho, c, hw * c, v, ld, hw * ld, dp, hw, hw * hw, b)
            if fused:
                dqkv = torch.empty_like(qkv)
                dq, dk, dv = dqkv[..., :c], dqkv[..., c:2 * c], dqkv[..., 2 * c:]
            else:
                dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
            self.bmm(1, 0, hw, c, hw, p, hw, hw * hw, dho, c, hw * c, dv, ld, hw * ld, b)
            ds = dp
            ops.softmax_rows_bwd(p, dp, ds, b * hw, hw)
            self.bmm(0, 0, hw, c, hw, ds, hw, hw * hw, k, ld, hw * ld, dq, ld, hw * ld, b, scale)
            self.bmm(1, 0, hw, c, hw, ds, hw, hw * hw, q, ld, hw * ld, dk, ld, hw * ld, b, scale)
            dhn = torch.empty_like(hn)
            # q / k / v bias gradients: ONE column-sum pass over the [m, 3c] gradient buffer, written to the three parameters
            seg = fused and 3 * c <= 1024
            if seg:
                self.on_side(lambda: ops.bias_grad_seg(dqkv, 3 * c, b, hw, (self.g(n0.b), self.g(n1.b), self.g(n2.b)), c), dqkv)
            # ... and their weight gradients from ONE GEMM hn^T [dq | dk | dv] (N = 3c: hn is staged and split once instead
            # of three times); the batched slab reduction cuts the [c][3c] result into the three parameters
            one_gemm = fused and self.defer and self.split and ops.gemm_tn_split_supported(c, 3 * c, m) and \
                ops.slab_units(c * c, 2, c, 3 * c) > 0

            def qkv_wgrad():
                nsplit = self._tn_split(c, 3 * c, m)
                slabs = self.slabs_for(4 * 3 * c * c * nsplit, dev).view(torch.float32)
                ops.gemm_tn_split(c, 3 * c, m, hn, c, dqkv, 3 * c, slabs, 3 * c, nsplit)
                for i, nin in enumerate((n0, n1, n2)):
                    self.reduce_slabs(slabs[i * c:], nsplit, c * c, self.g(nin.W), layout=2, taps=c, cin=3 * c, more=i < 2)

            if one_gemm:
                self.on_side(qkv_wgrad, hn, dqkv)
                if not seg:
                    for nin, d in ((n0, dq), (n1, dk), (n2, dv)):
                        self.on_side(lambda nin=nin, d=d: self.bias_grad(d, self.g(nin.b), ld=ld), d)
            else:
                for nin, d in ((n0, dq), (n1, dk), (n2, dv)):
                    self.on_side(lambda nin=nin, d=d: nin_wgrad(hn, d, nin, 1.0, ld, bias=not seg), hn, d)
            if fused:
                ops.gemm_split(dqkv, None, m, f_qkv_d, c, dhn)
            else:
                first = True
                for nin, d in ((n0, dq), (n1, dk), (n2, dv)):
                    ops.gemm_raw(0, 1, m, c, c, d, c, 0, nin.W, c, 0, dhn, c, 0,
                                 epi=None if first else ops.epilogue(accumulate=True))
                    first = False
            xg, acc = _gbuf(x)
            self.gn_backward(dhn, x.v, st, gn.weight, gn.bias, self.g(gn.weight), self.g(gn.bias), False, xg,
                             accumulate_dx=acc, add=dout, add_scale=s, last_writer_of=x if first_x else None)

        self.push(bwd, mod)
        return on

    # -- progressive_input == 'residual' (ncsnpp.py:350-357; layerspp.py:149-163) ---------------------
    def pyramid(self, pyr, h: _Node, mod: Downsample, first: bool) -> _Node:
        """pyr: NCHW input tensor (first level) or the previous combined node (NHWC)."""
        s = self.s
        conv = mod.conv
        cout = mod.out_ch
        self.use(h)
        if not first:
            self.use(pyr)
        if mod.fir:
            k = _fir_kernel(self.net.sf.fir_kernel)
            pad = (2, 2)  # up_or_down_sampling.py:173-176: p = (4-2) + (3-1)
            if first:
                xf = ops.nchw_to_nhwc(ops.upfirdn2d_raw(pyr, k, 1, 1, pad, layout=0))
            else:
                xf = ops.upfirdn2d_raw(pyr.v, k, 1, 1, pad, layout=1)
        else:
            raise NotImplementedError("progressive_input='residual' with fir=False is not on the north-star path")
        b, fh, fw, cin = xf.shape
        oh, ow = (fh - 3) // 2 + 1, (fw - 3) // 2 + 1
        out = torch.empty((b, oh, ow, cout), device=xf.device, dtype=torch.float32)
        epi = ops.epilogue(bias=conv.bias, residual=h.v, ld_residual=cout, out_scale=s)
        small = cin * 9 <= 64 and cout % 4 == 0
        cols = None
        net = self.net
        m = b * oh * ow
        # many-channel levels on the limb kernels: explicit im2col (K order = the packed OHWI weights') + pointwise GEMM
        limb = self.split and not small and cin % 4 == 0 and ops.gemm_split_supported(9 * cin, 0, m, cout) and \
            ops.gemm_split_supported(cout, 0, m, 9 * cin)
        if small:
            cols = self.small_in_conv(xf, conv, 2, 0, oh, ow, out, epi)
        elif limb:
            patches = ops.im2col3x3(xf, 2, 0, oh, ow)
            fr = net._gfrag(conv.weight, "s2fwd",
                            lambda prev: ops.gemm_frag(net._packed(conv), cout, 9 * cin, 9 * cin, 1, prev))
            ops.gemm_split(patches, None, m, fr, cout, out, epi)
            del patches
        else:
            ops.conv2d_nhwc(xf, None, self.net._packed(conv), cout, 3, 3, 2, 0, 1, oh, ow, out, epi)
        on = _Node(out, want_gsum=True)             # conv.bias = s * column sums of its gradient
        if not self.record:
            return on

        def bwd():
            dout = on.g
            on.g = None
            have_bias = self.bias_from(on, dout, conv.bias, s)
            hg, acc = _gbuf(h)
            ops.axpby(dout, s, None, 0.0, hg, accumulate=acc)
            def side():
                if small:
                    self.small_in_wgrad(dout, cols, conv, alpha=s)
                else:
                    self.wgrad(dout, xf, conv, 3, 2, 0, alpha=s)
                if not have_bias:
                    self.bias_grad(dout, self.g(conv.bias), alpha=s)

            self.on_side(side, dout, xf)
            if not first:
                dxf = torch.empty_like(xf)
                if limb:
                    frd = net._gfrag(conv.weight, "s2dgrad",
                                     lambda prev: ops.gemm_frag(net._packed(conv), 9 * cin, cout, 1, 9 * cin, prev))
                    dpatches = torch.empty((m, 9 * cin), device=dout.device, dtype=torch.float32)
                    ops.gemm_split(dout, None, m, frd, 9 * cin, dpatches, ops.epilogue(alpha=s))
                    ops.col2im3x3(dpatches, xf.shape, 2, 0, oh, ow, out=dxf)
                    del dpatches
                else:
                    self.dgrad(dout, conv, 3, 2, 0, fh, fw, dxf, alpha=s)
                pg, pacc = _gbuf(pyr)
                ops.upfirdn2d_bwd_raw(dxf, k, 1, 1, pad, (pyr.v.shape[1], pyr.v.shape[2]), 1, out=pg, accumulate=pacc)
            elif self.want_dx:
                # first level reads the network input itself (NCHW): conv dgrad -> NCHW -> FIR backward
                dxf = torch.empty_like(xf)
                self.dgrad(dout, conv, 3, 2, 0, fh, fw, dxf, alpha=s)
                dxf_nchw = ops.nhwc_to_nchw(dxf)
                acc = self.dx_nchw is not None
                if not acc:
                    self.dx_nchw = torch.empty_like(pyr)
                ops.upfirdn2d_bwd_raw(dxf_nchw, k, 1, 1, pad, (pyr.shape[2], pyr.shape[3]), 0, out=self.dx_nchw,
                                      accumulate=acc)

        self.push(bwd, mod)
        return on

    def _cat_ok(self, a: _Node, bnode: _Node, mod) -> bool:
        """Can ``mod`` (a residual block) consume the concatenation of a and b without it being materialised?"""
        if not (self.split and isinstance(mod, ResnetBlockBigGANpp)) or mod.up or mod.down or not mod.has_shortcut:
            return False
        b, h, w, c1 = a.v.shape
        c2 = bnode.v.shape[-1]
        cout, m = mod.out_ch, b * h * w
        cpg = (c1 + c2) // ops.gn_groups(c1 + c2)
        if c1 % 128 or c2 % 128 or c1 % cpg or c2 % cpg or (c1 // 4) > 256 or (c2 // 4) > 256:
            return False
        ok = ops.conv3x3_split_supported(c1, c2, b, h, w, cout) and ops.gemm_split_supported(c1, c2, m, cout)
        if self.record:
            ok = ok and all(ops.conv3x3_split_supported(cout, 0, b, h, w, c) and ops.gemm_split_supported(cout, 0, m, c) and
                            ops.conv3x3_wgrad_split_supported(cout, c, b, h, w) for c in (c1, c2)) and \
                ops.gemm_tn_split_supported(cout, c1, m)
        return ok

    def concat(self, a: _Node, bnode: _Node, consumer=None):
        """torch.cat([h, hs.pop()], dim=1) (ncsnpp.py:374) in NHWC; not materialised when ``consumer`` reads two sources."""
        if consumer is not None and self._cat_ok(a, bnode, consumer):
            return _CatNode(a, bnode)
        self.use(a)
        self.use(bnode)
        b, h, w, c1 = a.v.shape
        c2 = bnode.v.shape[-1]
        rows = b * h * w
        cat = torch.empty((b, h, w, c1 + c2), device=a.v.device, dtype=torch.float32)
        ops.copy2d(a.v, c1, cat, c1 + c2, rows, c1)
        ops.copy2d(bnode.v, c2, cat, c1 + c2, rows, c2, dst_off=c1)
        cn = _Node(cat)
        if self.record:
            def bwd():
                g = cn.g
                cn.g = None
                ga, acc = _gbuf(a)
                ops.copy2d(g, c1 + c2, ga, c1, rows, c1, accumulate=acc)
                gb, acc = _gbuf(bnode)
                ops.copy2d(g, c1 + c2, gb, c2, rows, c2, accumulate=acc, src_off=c1)

            self.push(bwd)
        return cn

    # -- whole network (ncsnpp.py:287-438) --------------------------------------------------------------
    def run(self, x: Tensor, t: Tensor) -> Tensor:
        with ops.stream_scope():
            return self._run(x, t)

    def _run(self, x: Tensor, t: Tensor) -> Tensor:
        net = self.net
        mods = net.all_modules
        mi = self.time_embedding(t)
        pin = net.progressive_input
        x_nhwc = ops.nchw_to_nhwc(x)
        stem = mods[mi]
        mi += 1
        b, hh, ww, _ = x_nhwc.shape
        if self.record:
            # Parameter-gradient kernels on a side stream.  Automatic rule: on while the kernels of the backward chain
            # cannot fill the chip by themselves (32x32 images: B = 16 +9 %, B = 32 +5 %, B = 64 +1.3 %, B = 128 +0.3 % images/s - and per-kernel HIP
            # event timings would be inflated by the concurrent MFMA kernel: off there)
            use = net.overlap_wgrad if net.overlap_wgrad is not None else (b * hh * ww <= _OVERLAP_MAX_PIXELS)
            self.side = net._side_stream() if use else None
        h0 = torch.empty((b, hh, ww, stem.weight.shape[0]), device=x.device, dtype=torch.float32)
        stem_small = x_nhwc.shape[-1] * 9 <= 64 and stem.weight.shape[0] % 4 == 0
        stem_cols = None
        if stem_small:
            stem_cols = self.small_in_conv(x_nhwc, stem, 1, 1, hh, ww, h0, ops.epilogue(bias=stem.bias))
        else:
            ops.conv2d_nhwc(x_nhwc, None, net._packed(stem), stem.weight.shape[0], 3, 3, 1, 1, 1, hh, ww, h0,
                            ops.epilogue(bias=stem.bias))
        n0 = _Node(h0, want_gsum=True)              # stem.bias = column sums of its gradient
        if self.record:
            def stem_bwd():
                g0 = n0.g
                n0.g = None
                have_bias = self.bias_from(n0, g0, stem.bias)

                def side():
                    if stem_small:
                        self.small_in_wgrad(g0, stem_cols, stem)
                    else:
                        self.wgrad(g0, x_nhwc, stem, 3, 1, 1)
                    if not have_bias:
                        self.bias_grad(g0, self.g(stem.bias))

                self.on_side(side, g0, x_nhwc)
                if self.want_dx:
                    dxs = torch.empty_like(x_nhwc)
                    self.dgrad(g0, stem, 3, 1, 1, hh, ww, dxs)
                    dxs = ops.nhwc_to_nchw(dxs)
                    if self.dx_nchw is None:
                        self.dx_nchw = dxs
                    else:
                        ops.axpby(dxs, 1.0, None, 0.0, self.dx_nchw, accumulate=True)

            self.push(stem_bwd, stem)
        hs: List[_Node] = [n0]
        pyr = x
        first_pyr = True
        for lvl in range(net.num_resolutions):
            for _ in range(net.num_res_blocks):
                hnode = self.resblock(hs[-1], mods[mi])
                mi += 1
                if hnode.v.shape[2] in net.attn_resolutions:
                    hnode = self.attn(hnode, mods[mi])
                    mi += 1
                hs.append(hnode)
            if lvl != net.num_resolutions - 1:
                hnode = self.resblock(hs[-1], mods[mi])
                mi += 1
                if pin == "residual":
                    hnode = self.pyramid(pyr, hnode, mods[mi], first_pyr)
                    mi += 1
                    pyr = hnode
                    first_pyr = False
                hs.append(hnode)
        hnode = hs[-1]
        hnode = self.resblock(hnode, mods[mi]); mi += 1
        hnode = self.attn(hnode, mods[mi]); mi += 1
        hnode = self.resblock(hnode, mods[mi]); mi += 1
        if net.is_classifier:
            assert mi + 1 == len(mods)
            return self.clf_head(hnode, mods[mi])
        for lvl in reversed(range(net.num_resolutions)):
            for _ in range(net.num_res_blocks + 1):
                hnode = self.resblock(self.concat(hnode, hs.pop(), mods[mi]), mods[mi])
                mi += 1
            if hnode.v.shape[2] in net.attn_resolutions:
                hnode = self.attn(hnode, mods[mi])
                mi += 1
            if lvl != 0:
                hnode = self.resblock(hnode, mods[mi])
                mi += 1
        assert not hs
        gnf, head = mods[mi], mods[mi + 1]
        assert mi + 2 == len(mods)
        first_last = self.use(hnode)
        stf = self.node_stats(hnode, gnf.weight, gnf.bias)
        af = ops.gn_apply(hnode.v, stf, True)
        oc = head.weight.shape[0]
        y = torch.empty((b, hh, ww, oc), device=x.device, dtype=torch.float32)
        if ops.conv3x3_fewout_supported(af.shape[-1], oc):
            ops.conv3x3_fewout(af, net._packed(head), head.bias, oc, y)
        else:
            ops.conv2d_nhwc(af, None, net._packed(head), oc, 3, 3, 1, 1, 1, hh, ww, y, ops.epilogue(bias=head.bias))
        if self.record:
            last = hnode
            self.head_grad = _Node(y)
            hg = self.head_grad

            def head_bwd():
                dy = hg.g

                daf = torch.empty_like(af)
                if oc * 9 <= 64 and af.shape[-1] % 4 == 0:
                    self.small_out_backward(dy, af, head, daf)
                else:
                    def side():
                        self.wgrad(dy, af, head, 3, 1, 1)
                        self.bias_grad(dy, self.g(head.bias))

                    self.on_side(side, dy, af)
                    self.dgrad(dy, head, 3, 1, 1, hh, ww, daf)
                xg, acc = _gbuf(last)
                self.gn_backward(daf, last.v, stf, gnf.weight, gnf.bias, self.g(gnf.weight), self.g(gnf.bias), True, xg,
                                 accumulate_dx=acc, last_writer_of=last if first_last else None)

            self.push(head_bwd, gnf)
        return ops.nhwc_to_nchw(y)

    # -- NCSNppClassifier head (ncsnpp_clf.py:277-283): flatten in NCHW order + Linear(bias=False) ------------------
    def clf_head(self, hnode: _Node, lin: nn.Linear) -> Tensor:
        self.use(hnode)
        b, h, w, c = hnode.v.shape
        flat = ops.nhwc_to_nchw(hnode.v).view(b, c * h * w)
        n_cls, k = lin.weight.shape
        logits = ops.linear(flat, lin.weight)
        if self.record:
            hg = self.head_grad = _Node(logits)

            def head_bwd():
                dy = hg.g                                                   # [B, n_cls]
                self.on_side(lambda: ops.gemm_raw(1, 0, n_cls, k, b, dy, n_cls, 0, flat, k, 0, self.g(lin.weight), k, 0),
                             dy, flat)
                dflat = torch.empty_like(flat)
                ops.gemm_raw(0, 0, b, k, n_cls, dy, n_cls, 0, lin.weight, k, 0, dflat, k, 0)
                xg, acc = _gbuf(hnode)
                ops.axpby(ops.nchw_to_nhwc(dflat.view(b, c, h, w)), 1.0, None, 0.0, xg, accumulate=acc)

            self.push(head_bwd, lin)
        return logits

    def backward(self, grad_out_nchw: Tensor):
        with ops.stream_scope():
            self._backward(grad_out_nchw)

    def _backward(self, grad_out_nchw: Tensor):
        net = self.net
        g_out = grad_out_nchw.contiguous()
        self.head_grad.g = g_out if net.is_classifier else ops.nchw_to_nhwc(g_out)
        # every pass: gn_backward and the resblock column sums allocate from the arena whether or not the reductions are
        # deferred (with defer off the offset was never rewound and the buffer doubled until OOM); the previous
        # pass's readers of it are stream-ordered before this pass's writers
        net._param_arena().reset()
        red = net._reducer
        for fn, module in reversed(self.tape):
            fn()
            if module is not None and self.watermark is not None:
                off = net._module_offset(module)
                if self.defer and red is not None and red.would_launch(off):
                    self.flush_deferred()       # a bucket is about to be exchanged: its parked reductions first
                self.flush_side()       # the reducer may launch a bucket now: its gradients must at least be enqueued
                self.watermark(off)
        self.finish_backward()
        self.tape = None


class _Pending:
    """Counts forward passes whose backward has not run yet (released on backward or when autograd drops
    the graph) — bookkeeping for diagnostics; several outstanding graphs are fine because a backward that
    finds populated gradients accumulates into them (see ``NCSNpp._begin_backward``)."""

    def __init__(self, net):
        self.net = net
        self.live = True
        net._pending += 1

    def release(self):
        if self.live:
            self.live = False
            self.net._pending -= 1

    def __del__(self):
        self.release()


class _NCSNppFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, t, anchor, net):
        ex = _Exec(net, record=True)
        ex.watermark = net._watermark_hook
        ex.want_dx = bool(x.requires_grad)
        y = ex.run(x, t)
        ctx.ex = ex
        ctx.net = net
        ctx.pending = _Pending(net)
        return y

    @staticmethod
    def backward(ctx, gy):
        ex, net = ctx.ex, ctx.net
        if ex is None:
            raise RuntimeError("psld_amd.NCSNpp: backward through the same forward pass twice is not supported")
        ctx.ex = None
        net._begin_backward(side=ex.side)
        ex.backward(gy)
        net._end_backward()
        ctx.pending.release()
        return ex.dx_nchw, None, None, None


class _NCSNppParamFn(torch.autograd.Function):
    """Same executor, with every trainable parameter an INPUT of the autograd node: backward hands their gradients
    (views of the flat gradient buffer) to autograd, which runs each parameter's AccumulateGrad node - the place where
    torch ``DistributedDataParallel`` (Lightning ``strategy="ddp"``, train_sde.py:114) hangs its reducer hooks.
    ``_NCSNppFn`` assigns ``p.grad`` itself and never reaches those nodes."""

    @staticmethod
    def forward(ctx, x, t, net, *params):
        ex = _Exec(net, record=True)
        ex.want_dx = bool(x.requires_grad)
        y = ex.run(x, t)
        ctx.ex = ex
        ctx.net = net
        ctx.pending = _Pending(net)
        ctx.n_params = len(params)
        return y

    @staticmethod
    def backward(ctx, gy):
        ex, net = ctx.ex, ctx.net
        if ex is None:
            raise RuntimeError("psld_amd.NCSNpp: backward through the same forward pass twice is not supported")
        ctx.ex = None
        net._begin_backward(visible=True, side=ex.side)
        ex.backward(gy)
        grads = net._end_backward_visible()
        ctx.pending.release()
        assert len(grads) == ctx.n_params
        return (ex.dx_nchw, None, None) + grads


@register_module(category="score_fn", name="ncsnpp")
class NCSNpp(nn.Module):
    """NCSN++ (ncsnpp.py:35-285 for the module list; forward in ``_Exec.run``)."""

    is_classifier = False

    def _net_config(self, config):
        return config.model.score_fn

    def __init__(self, config):
        super().__init__()
        self.config = config.model
        sf = self.sf = self._net_config(config)
        if sf.nonlinearity.lower() != "swish":
            raise NotImplementedError("only nonlinearity='swish' is on the north-star path")
        if sf.resblock_type.lower() != "biggan" or sf.progressive.lower() != "none":
            raise NotImplementedError("only resblock_type='biggan', progressive='none' are supported")
        self.nf = nf = sf.nf
        ch_mult = list(sf.ch_mult)
        self.num_res_blocks = nres = sf.num_res_blocks
        self.attn_resolutions = list(sf.attn_resolutions)
        self.num_resolutions = nlev = len(ch_mult)
        self.all_resolutions = [config.data.image_size // (2 ** i) for i in range(nlev)]
        self.noise_cond = sf.noise_cond
        self.skip_rescale = sf.skip_rescale
        self.progressive_input = pin = sf.progressive_input.lower()
        self.embedding_type = emb = sf.embedding_type.lower()
        if pin not in ("none", "residual"):
            raise NotImplementedError("progressive_input must be 'none' or 'residual'")
        if emb not in ("fourier", "positional"):
            raise ValueError(f"embedding type {emb} unknown.")
        init_scale = sf.init_scale
        dropout = sf.dropout
        fir = sf.fir

        modules: List[nn.Module] = []
        if emb == "fourier":
            assert config.training.continuous, "Fourier features are only used for continuous training."
            modules.append(GaussianFourierProjection(embedding_size=nf, scale=sf.fourier_scale))
            embed_dim = 2 * nf
        else:
            embed_dim = nf
        if self.noise_cond:
            modules.append(_linear(embed_dim, nf * 4))
            modules.append(_linear(nf * 4, nf * 4))
        temb_dim = nf * 4 if self.noise_cond else None
        rb = lambda **kw: ResnetBlockBigGANpp(temb_dim=temb_dim, dropout=dropout, init_scale=init_scale, **kw)

        channels = sf.in_ch
        input_pyramid_ch = channels
        modules.append(_conv(channels, nf, 3))
        hs_c = [nf]
        in_ch = nf
        for lvl in range(nlev):
            for _ in range(nres):
                out_ch = nf * ch_mult[lvl]
                modules.append(rb(in_ch=in_ch, out_ch=out_ch))
                in_ch = out_ch
                if self.all_resolutions[lvl] in self.attn_resolutions:
                    modules.append(AttnBlockpp(in_ch, init_scale))
                hs_c.append(in_ch)
            if lvl != nlev - 1:
                modules.append(rb(in_ch=in_ch, down=True))
                if pin == "residual":
                    modules.append(Downsample(input_pyramid_ch, in_ch, fir))
                    input_pyramid_ch = in_ch
                hs_c.append(in_ch)
        in_ch = hs_c[-1]
        modules.append(rb(in_ch=in_ch))
        modules.append(AttnBlockpp(in_ch, init_scale))
        modules.append(rb(in_ch=in_ch))
        if self.is_classifier:
            # ncsnpp_clf.py:196-199: flatten (NCHW order) + bias-free Linear to the class logits
            self.n_cls = sf.n_cls
            modules.append(nn.Linear(in_ch * self.all_resolutions[-1] ** 2, self.n_cls, bias=False))
        else:
            for lvl in reversed(range(nlev)):
                for _ in range(nres + 1):
                    out_ch = nf * ch_mult[lvl]
                    modules.append(rb(in_ch=in_ch + hs_c.pop(), out_ch=out_ch))
                    in_ch = out_ch
                if self.all_resolutions[lvl] in self.attn_resolutions:
                    modules.append(AttnBlockpp(in_ch, init_scale))
                if lvl != 0:
                    modules.append(rb(in_ch=in_ch, up=True))
            assert not hs_c
            modules.append(_groupnorm(in_ch))
            modules.append(_conv(in_ch, sf.out_ch, 3, init_scale))
        self.all_modules = nn.ModuleList(modules)

        # executor state (never part of state_dict)
        self._flat: Optional[Tensor] = None
        self._flat_grad: Optional[Tensor] = None
        self._offsets = None
        self._pack_cache = {}
        self._frag_table = None     # (signature, device table, entries, total work items) of the batched fragment refresh
        self._qkv_bias_table = None
        self._qkv_bias_stamp = {}
        self._wfrag_table = None    # the same for the Winograd fragment sets
        self._temb_plan_cache = None
        self._pack_key = None
        self._epoch = 0
        self._anchor = None
        self._reducer = None
        self._posfreq = None
        self._module_offs = None
        # parameter-gradient kernels on a side stream: None = automatic (small batches, see _Exec._run); True / False or
        # PSLD_OVERLAP_WGRAD=1 / 0 force it
        import os as _os
        _ow = _os.environ.get("PSLD_OVERLAP_WGRAD")
        self.overlap_wgrad = None if _ow is None else _ow == "1"
        self.side_group = 32        # side-stream calls per fork: one event + one stream wait per group (tools/graph_cross.py)
        # dgamma / dbeta / bias gradients / split-K slab reductions of a backward pass in batched launches (_Exec.defer_param,
        # _Exec.reduce_slabs); False: one launch per layer, right where the reference's autograd would compute them
        self.defer_param_grads = True
        self._tables = ops.TableCache()
        self._parena = self._sarena = None
        self._persistent = {}
        # None (auto): parameters become inputs of the autograd node (gradients delivered through AccumulateGrad, so
        # torch DDP / Lightning's ddp strategy can reduce them) when a multi-rank process group exists and no
        # BucketReducer is attached; True / False (or PSLD_AUTOGRAD_PARAMS=1 / 0) force it.
        _ap = _os.environ.get("PSLD_AUTOGRAD_PARAMS")
        self.autograd_params = None if _ap is None else _ap == "1"
        self._plist = None
        self._tlist = None
        self._gviews = None
        self._pending = 0
        self._backward_count = 0    # finished backward passes (FusedAdam.step refuses to re-apply a consumed gradient)
        self._dropout_seed_dev = None   # static int64 device word supplied by a captured training step (wrapper.py)
        self._accumulating = False
        self._grad_stale = False
        self._scratch_grad = None
        self._sviews = None
        self.use_graphs = _os.environ.get("PSLD_GRAPHS", "0") == "1"
        self._graphs = {}
        self._conv_by_weight = {}
        self._side = None

    def pin_scratch(self):
        """A captured hipGraph replays raw pointers into the parameter / slab arenas and the cached job tables: keep every
        buffer they ever pointed to alive (outgrown arena buffers are retained, tables are not evicted)."""
        self._param_arena().pinned = True
        self._slab_arena().pinned = True
        self._tables.pinned = True

    def _param_arena(self) -> "ops.Arena":
        dev = self._params()[0].device
        if self._parena is None or self._parena.device != dev:
            self._parena = ops.Arena(dev, 64 << 20)
        return self._parena

    def _slab_arena(self) -> "ops.Arena":
        dev = self._params()[0].device
        if self._sarena is None or self._sarena.device != dev:
            self._sarena = ops.Arena(dev, _SLAB_FLUSH_BYTES + (256 << 20))
        return self._sarena

    def _persist(self, name: str, shape) -> Tensor:
        """A float32 buffer that keeps its address for this network, name and shape."""
        dev = self._params()[0].device
        key = (name, tuple(shape), dev)
        t = self._persistent.get(key)
        if t is None:
            t = self._persistent[key] = torch.empty(tuple(shape), device=dev, dtype=torch.float32)
        return t

    def _side_stream(self):
        dev = next(self.parameters()).device
        if self._side is None or self._side.device != dev:
            self._side = torch.cuda.Stream(device=dev)
        return self._side

    # ---- flat parameter / gradient storage ----------------------------------------------------------
    def _params(self) -> List[nn.Parameter]:
        """Cached parameter list (the module structure never changes after construction)."""
        if self._plist is None:
            self._plist = list(self.parameters())
        return self._plist

    def _layout(self):
        offs, off = {}, 0
        for p in self._params():
            offs[id(p)] = off
            off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        return offs, off

    def flatten_parameters(self) -> Tensor:
        """Make every parameter a view into one contiguous fp32 buffer (idempotent).  Needed by the
        fused optimiser / EMA / all-reduce; deepcopy() and .to() are followed by a re-flatten."""
        params = self._params()
        flat = self._flat
        if flat is not None:
            # fast validation: every 16th parameter (and the last) still points into the flat buffer
            offs = self._offsets
            base = flat.data_ptr()
            if all(q.data_ptr() == base + 4 * offs[id(q)] for q in params[::16]) and \
                    params[-1].data_ptr() == base + 4 * offs[id(params[-1])]:
                return flat
        dev = params[0].device
        offs, total = self._layout()
        ok = flat is not None and flat.device == dev and flat.numel() == total and all(
            p.data_ptr() == flat.data_ptr() + 4 * offs[id(p)] for p in params)
        if not ok:
            flat = torch.zeros(total, device=dev, dtype=torch.float32)
            for p in params:
                o = offs[id(p)]
                flat[o:o + p.numel()].copy_(p.data.reshape(-1))
                p.data = flat[o:o + p.numel()].view(p.shape)
            self._flat = flat
            self._flat_grad = None
            self._gviews = None
            self._scratch_grad = self._sviews = None
            self._pack_cache.clear()
            self._qkv_bias_table = None
            self._qkv_bias_stamp = {}
            self._epoch += 1
        self._offsets = offs
        self._module_offs = None
        return self._flat

    def flat_grad(self) -> Tensor:
        self.flatten_parameters()
        if self._flat_grad is None or self._flat_grad.device != self._flat.device or \
                self._flat_grad.numel() != self._flat.numel():
            self._flat_grad = torch.zeros_like(self._flat)
            self._gviews = None
            self._gviews = self._views_of(self._flat_grad)
        return self._flat_grad

    def _views_of(self, buf: Tensor):
        return {id(q): buf[self._offsets[id(q)]:self._offsets[id(q)] + q.numel()].view(q.shape) for q in self._params()}

    def _grad_view(self, p: nn.Parameter) -> Tensor:
        """View of the buffer the CURRENT backward writes into (the flat gradient, or the scratch buffer when
        this pass has to be accumulated onto existing gradients)."""
        if self._gviews is None:
            self._gviews = self._views_of(self._flat_grad)
        if self._accumulating:
            if self._sviews is None:
                self._sviews = self._views_of(self._scratch_grad)
            return self._sviews[id(p)]
        return self._gviews[id(p)]

    def _module_offset(self, module: nn.Module) -> int:
        if self._module_offs is None:
            self._module_offs = {}
            for m in self.all_modules:
                ps = list(m.parameters())
                if ps:
                    self._module_offs[id(m)] = min(self._offsets[id(p)] for p in ps)
        return self._module_offs.get(id(module), 0)

    def weights_changed(self):
        """Call after writing parameters through raw pointers (fused optimiser / EMA kernels)."""
        self._epoch += 1

    def _packed(self, conv: _Affine, dgrad: bool = False) -> Tensor:
        """[co][tap][ci] (forward) or [ci][flip tap][co] (data-gradient) copy of an OIHW weight,
        cached until the weights change."""
        w = conv.weight
        key = (id(w), dgrad)
        self._conv_by_weight[id(w)] = conv
        ent = self._pack_cache.get(key)
        stamp = (self._epoch, w._version, w.data_ptr())
        if ent is not None and ent[0] == stamp:
            return ent[1]
        co, ci, kh, kw = w.shape
        out = ent[1] if ent is not None and ent[1].device == w.device else \
            torch.empty((ci, kh * kw, co) if dgrad else (co, kh * kw, ci), device=w.device, dtype=torch.float32)
        (ops.pack_dgrad if dgrad else ops.pack_ohwi)(w.detach(), out)
        self._pack_cache[key] = (stamp, out)
        return out

    def _frag(self, conv: _Affine, dgrad: bool) -> Tensor:
        """bf16 limb fragments of a 3x3 weight (ops.conv3x3_frag), cached until the weights change; once two or
        more exist, a weight update refreshes ALL of them with one batched launch."""
        w = conv.weight
        key = (id(w), dgrad, "frag")
        self._conv_by_weight[id(w)] = conv
        ent = self._pack_cache.get(key)
        stamp = (self._epoch, w._version, w.data_ptr())
        if ent is not None and ent[0] == stamp:
            return ent[1]
        if ent is not None and ent[1].device == w.device and self._refresh_frags():
            ent = self._pack_cache[key]
            if ent[0] == stamp:
                return ent[1]
        out = ops.conv3x3_frag(w.detach(), dgrad, ent[1] if ent is not None and ent[1].device == w.device else None)
        self._pack_cache[key] = (stamp, out)
        self._frag_table = None
        return out

    def _wfrag(self, conv: _Affine, dgrad: bool) -> Tensor:
        """Winograd-transformed bf16 limb fragments of a 3x3 weight (ops.conv3x3_wino_frag), cached and refreshed like
        ``_frag`` (one batched launch for all of them after a weight update)."""
        w = conv.weight
        key = (id(w), dgrad, "wfrag")
        self._conv_by_weight[id(w)] = conv
        ent = self._pack_cache.get(key)
        stamp = (self._epoch, w._version, w.data_ptr())
        if ent is not None and ent[0] == stamp:
            return ent[1]
        if ent is not None and ent[1].device == w.device and self._refresh_wfrags():
            ent = self._pack_cache[key]
            if ent[0] == stamp:
                return ent[1]
        out = ops.conv3x3_wino_frag(w.detach(), dgrad, ent[1] if ent is not None and ent[1].device == w.device else None)
        self._pack_cache[key] = (stamp, out)
        self._wfrag_table = None
        return out

    def _refresh_wfrags(self) -> bool:
        """Re-transform every registered Winograd fragment set into its existing buffer with ONE launch
        (psld_pack_wino_batch).  False when there is nothing to batch."""
        keys = [k for k in self._pack_cache if len(k) == 3 and k[2] == "wfrag"]
        if len(keys) < 2:
            return False
        ws = [self._conv_by_weight[k[0]].weight for k in keys]
        outs = [self._pack_cache[k][1] for k in keys]
        if any(o.device != w.device for o, w in zip(outs, ws)):
            return False
        sig = tuple((k, w.data_ptr(), o.data_ptr()) for k, w, o in zip(keys, ws, outs))
        if self._wfrag_table is None or self._wfrag_table[0] != sig:
            rows, total = [], 0
            for k, w, o in zip(keys, ws, outs):
                rows.append(ops.conv3x3_wino_frag_entry(w.detach(), k[1], o) + [total])
                total += w.shape[0] * w.shape[1] // 8
            self._wfrag_table = (sig, torch.tensor(rows, dtype=torch.int64, device=ws[0].device), len(rows), total)
        _, table, n, total = self._wfrag_table
        ops.pack_wino_batch(table, n, total)
        for k, w, o in zip(keys, ws, outs):
            self._pack_cache[k] = ((self._epoch, w._version, w.data_ptr()),) + tuple(self._pack_cache[k][1:])
        return True

    def _refresh_frags(self) -> bool:
        """Re-split every registered 3x3 / pointwise weight into its existing fragment buffer with ONE launch
        (psld_pack_frag_batch).  False when there is nothing to batch."""
        keys = [k for k in self._pack_cache if len(k) == 3 and k[2] in ("frag", "pfrag")]
        if len(keys) < 2:
            return False
        ws = [self._conv_by_weight[k[0]].weight if k[2] == "frag" else self._pack_cache[k][2] for k in keys]
        outs = [self._pack_cache[k][1] for k in keys]
        if any(o.device != w.device for o, w in zip(outs, ws)):
            return False
        sig = tuple((k, w.data_ptr(), o.data_ptr()) for k, w, o in zip(keys, ws, outs))
        if self._frag_table is None or self._frag_table[0] != sig:
            rows, total = [], 0
            for k, w, o in zip(keys, ws, outs):
                if k[2] == "frag":
                    rows.append(ops.conv3x3_frag_entry(w.detach(), k[1], o) + [total])
                    total += w.shape[0] * w.shape[1] // 8        # work items: one per lane slot
                else:
                    n, kk, sn, sk, c0, ct = self._pack_cache[k][3]
                    rows.append([w.data_ptr(), o.data_ptr(), n, kk | (c0 << 20) | (ct << 40), 1, sn, sk, total])
                    total += n * kk // 8
            self._frag_table = (sig, torch.tensor(rows, dtype=torch.int64, device=ws[0].device), len(rows), total)
        _, table, n, total = self._frag_table
        ops.pack_frag_batch(table, n, total)
        for k, w, o in zip(keys, ws, outs):
            self._pack_cache[k] = ((self._epoch, w._version, w.data_ptr()),) + tuple(self._pack_cache[k][1:])
        return True

    def _pfrag(self, owner: nn.Parameter, tag: str, n: int, k: int, sn: int, sk: int, into: Optional[Tensor] = None,
               chunk0: int = 0, chunks_total: int = 0) -> Tensor:
        """Limb fragments (ops.gemm_frag) of the [n][k] view of ONE parameter (element (i, j) at i*sn + j*sk), cached
        until the weights change and refreshed together with the 3x3 fragments by the batched launch.
        ``into``: the buffer to fill (several parameters that share one fragment set: q | k | v) - then ``chunk0`` /
        ``chunks_total`` place this parameter's K range inside the set's K dimension (psld_pack_frag_batch)."""
        key = (id(owner), tag, "pfrag")
        ent = self._pack_cache.get(key)
        stamp = (self._epoch, owner._version, owner.data_ptr())
        if ent is not None and ent[0] == stamp and (into is None or ent[1].data_ptr() == into.data_ptr()):
            return ent[1]
        if ent is not None and ent[1].device == owner.device and (into is None or ent[1].data_ptr() == into.data_ptr()) and \
                self._refresh_frags():
            ent = self._pack_cache[key]
            if ent[0] == stamp:
                return ent[1]
        if into is None:
            out = ops.gemm_frag(owner.detach(), n, k, sn, sk, ent[1] if ent is not None and ent[1].device == owner.device else None)
        else:       # first use: a one-entry table through the batched entry point (the only one that takes a K placement)
            out = into
            row = [owner.data_ptr(), out.data_ptr(), n, k | (chunk0 << 20) | (chunks_total << 40), 1, sn, sk, 0]
            ops.pack_frag_batch(torch.tensor(row, dtype=torch.int64, device=owner.device), 1, n * k // 8)
        self._pack_cache[key] = (stamp, out, owner, (n, k, sn, sk, chunk0, chunks_total))
        self._frag_table = None
        return out

    def _qkv_frags(self, mod):
        """Fragments of an attention block's q | k | v projections as ONE GEMM operand each way - forward B[n][k] =
        [W_q | W_k | W_v][k][n] (N = 3c), data gradient B[n][k] = [W_q | W_k | W_v][n][k] (K = 3c) - and the concatenated
        bias.  Packed straight from the three parameters into shared buffers by the batched refresh of all fragments (no
        concatenated copy of the weights, no launch of their own after the first step); the biases of ALL attention blocks
        are gathered by one batched copy when they change."""
        n0, n1, n2 = mod.NIN_0, mod.NIN_1, mod.NIN_2
        c = n0.W.shape[0]
        key = (id(n0.W), "qkv", "bufs")
        bufs = self._pack_cache.get(key)
        dev = n0.W.device
        if bufs is None or bufs[0].device != dev:
            fb = ops.gemm_frag_bytes(c, c)
            bufs = (torch.empty(3 * fb, dtype=torch.uint8, device=dev), torch.empty(3 * fb, dtype=torch.uint8, device=dev),
                    torch.empty(3 * c, dtype=torch.float32, device=dev), fb, mod)
            self._pack_cache[key] = bufs
            self._qkv_bias_table = None
        pf, pd, bq, fb = bufs[:4]
        for i, nin in enumerate((n0, n1, n2)):
            # forward: rows n of the set are output channels -> each projection is a contiguous third of the set
            self._pfrag(nin.W, "qkv_f", c, c, 1, c, into=pf[i * fb:(i + 1) * fb])
            # data gradient: the three projections are concatenated along K
            self._pfrag(nin.W, "qkv_d", c, c, c, 1, into=pd, chunk0=i * (c // 32), chunks_total=3 * (c // 32))
        stamp = (self._epoch, n0.b._version, n0.b.data_ptr())
        if self._qkv_bias_stamp.get(id(mod)) != stamp:
            self._refresh_qkv_biases()
        return pf, pd, bq

    def _refresh_qkv_biases(self):
        mods = [m for m in self.all_modules if isinstance(m, AttnBlockpp) and (id(m.NIN_0.W), "qkv", "bufs") in self._pack_cache]
        sig = tuple((m.NIN_0.b.data_ptr(), self._pack_cache[(id(m.NIN_0.W), "qkv", "bufs")][2].data_ptr()) for m in mods)
        if self._qkv_bias_table is None or self._qkv_bias_table[0] != sig:
            rows, first = [], 0
            for m in mods:
                bq = self._pack_cache[(id(m.NIN_0.W), "qkv", "bufs")][2]
                c = m.NIN_0.b.numel()
                for i, nin in enumerate((m.NIN_0, m.NIN_1, m.NIN_2)):
                    rows += [nin.b.data_ptr(), bq.data_ptr() + 4 * i * c, c // 4, first]
                    first += c // 4
            self._qkv_bias_table = (sig, torch.tensor(rows, dtype=torch.int64, device=mods[0].NIN_0.b.device), 3 * len(mods), first)
        _, table, jobs, total = self._qkv_bias_table
        ops.copy_batch(table, jobs, total)
        for m in mods:
            self._qkv_bias_stamp[id(m)] = (self._epoch, m.NIN_0.b._version, m.NIN_0.b.data_ptr())

    def _temb_offset(self, mod) -> Optional[int]:
        plan = self._temb_plan_cache
        return None if plan is None else plan["offsets"].get(id(mod))

    def _temb_plan(self):
        """Gathered time-embedding projections: ``wcat`` [sum C_out][4*nf] and ``bcat`` [sum C_out] hold Dense_0.weight /
        .bias of every ResBlock back to back (refreshed with ONE batched copy when the weights change), ``offsets`` the
        first row of each block.  None when the network is not noise-conditioned."""
        if not self.noise_cond:
            return None
        blocks = [m for m in self.all_modules if isinstance(m, ResnetBlockBigGANpp)]
        dev = blocks[0].Dense_0.weight.device
        plan = self._temb_plan_cache
        stamp = (self._epoch, blocks[0].Dense_0.weight._version, blocks[0].Dense_0.weight.data_ptr(), str(dev))
        if plan is not None and plan["stamp"] == stamp:
            return plan
        if plan is None or plan["wcat"].device != dev:
            total = sum(m.Dense_0.weight.shape[0] for m in blocks)
            kd = blocks[0].Dense_0.weight.shape[1]
            plan = {"wcat": torch.empty((total, kd), device=dev, dtype=torch.float32),
                    "bcat": torch.empty((total,), device=dev, dtype=torch.float32), "total": total, "offsets": {},
                    "table": None, "sig": None, "blocks": blocks}
            off = 0
            for m in blocks:
                plan["offsets"][id(m)] = off
                off += m.Dense_0.weight.shape[0]
        sig = tuple((m.Dense_0.weight.data_ptr(), m.Dense_0.bias.data_ptr()) for m in blocks)
        if plan["table"] is None or plan["sig"] != sig:
            rows, first = [], 0
            for m in blocks:
                w, bias = m.Dense_0.weight, m.Dense_0.bias
                o = plan["offsets"][id(m)]
                for src, dst, n in ((w, plan["wcat"][o], w.numel()), (bias, plan["bcat"][o:], bias.numel())):
                    assert n % 4 == 0 and src.data_ptr() % 16 == 0 and dst.data_ptr() % 16 == 0
                    rows.append([src.data_ptr(), dst.data_ptr(), n // 4, first])
                    first += n // 4
            plan["table"], plan["sig"], plan["total4"] = torch.tensor(rows, dtype=torch.int64, device=dev), sig, first
        ops.copy_batch(plan["table"], plan["table"].shape[0], plan["total4"])
        plan["stamp"] = stamp
        self._temb_plan_cache = plan
        return plan

    def _gfrag(self, owner: nn.Parameter, tag: str, build):
        """Limb fragments derived from ``owner`` (and possibly sibling parameters), cached until the weights change.
        ``build(prev)`` returns the tensor (or tuple of tensors) to keep and refreshes ``prev`` IN PLACE when given
        (captured graphs hold these addresses)."""
        key = (id(owner), tag, "gfrag")
        ent = self._pack_cache.get(key)
        stamp = (self._epoch, owner._version, owner.data_ptr())
        if ent is not None and ent[0] == stamp:
            return ent[1]
        prev = ent[1] if ent is not None and (ent[1][0] if isinstance(ent[1], tuple) else ent[1]).device == owner.device else None
        out = build(prev)
        self._pack_cache[key] = (stamp, out, owner, build)
        return out

    def _pos_freq(self, device):
        if self._posfreq is None or self._posfreq.device != device:
            half = self.nf // 2
            e = math.log(10000) / (half - 1)                                   # layers.py:500-507
            self._posfreq = torch.exp(torch.arange(half, dtype=torch.float32) * -e).to(device)
        return self._posfreq

    # ---- backward bookkeeping ------------------------------------------------------------------------
    def mark_grads_stale(self):
        """The next backward overwrites the gradient buffer (what ``zero_grad`` means for this module)."""
        self._grad_stale = True

    def _trainable(self) -> List[nn.Parameter]:
        if self._tlist is None or len(self._tlist[1]) != sum(p.requires_grad for p in self._params()):
            self._tlist = (None, [p for p in self._params() if p.requires_grad])
        return self._tlist[1]

    def _params_visible(self) -> bool:
        """Should this forward hand the parameters to autograd (see ``autograd_params``)?"""
        if self.autograd_params is not None:
            return self.autograd_params
        import torch.distributed as dist
        return self._reducer is None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def _begin_backward(self, visible: bool = False, side=None):
        """``side``: the stream THIS pass runs its parameter-gradient kernels on (None: everything on one stream)."""
        self.flat_grad()
        # torch semantics: a populated .grad is accumulated into.  Kernels WRITE their results, so in that
        # case this pass goes to a scratch buffer that is added afterwards (one extra 0.4 GB pass).
        probe = next((p for p in self._params() if p.requires_grad), None)
        if visible:
            # autograd adds what backward returns INTO a populated .grad: it must not be handed the very memory
            # .grad aliases, whatever zero_grad() marked
            self._accumulating = probe is not None and probe.grad is not None
        else:
            self._accumulating = probe is not None and (probe.grad is not None) and not self._grad_stale
        self._grad_stale = False
        target = self._flat_grad
        if self._accumulating:
            if self._scratch_grad is None or self._scratch_grad.numel() != self._flat_grad.numel() or \
                    self._scratch_grad.device != self._flat_grad.device:
                self._scratch_grad = torch.zeros_like(self._flat_grad)
                self._sviews = None
            target = self._scratch_grad
        if self._reducer is not None:
            self._reducer.begin(target)
            self._reducer.producer_streams = [side] if side is not None else []

    def _watermark_hook(self, offset: int):
        if self._reducer is not None:
            self._reducer.ready_from(offset)

    def _end_backward(self):
        self._backward_count += 1
        if self._reducer is not None:
            self._reducer.finish()
        if self._accumulating:
            ops.axpby(self._scratch_grad, 1.0, None, 0.0, self._flat_grad, accumulate=True)
            self._accumulating = False
        for p in self._params():
            if not p.requires_grad:
                continue
            gv = self._gviews[id(p)]
            if p.grad is None or p.grad.data_ptr() == gv.data_ptr():
                p.grad = gv
            else:
                p.grad.add_(gv)  # caller kept a foreign .grad tensor: accumulate like autograd would

    def _end_backward_visible(self):
        """Gradients of the trainable parameters, in ``_trainable()`` order, for autograd to accumulate: fresh view
        objects of the buffer this pass wrote (a view nobody else references is adopted by AccumulateGrad as ``.grad``
        without a copy, so ``p.grad`` keeps aliasing the flat gradient buffer the fused optimiser reads)."""
        src = self._scratch_grad if self._accumulating else self._flat_grad
        self._accumulating = False
        self._backward_count += 1
        offs = self._offsets
        return tuple(src[offs[id(p)]:offs[id(p)] + p.numel()].view(p.shape) for p in self._trainable())

    def adopt_foreign_grads(self) -> int:
        """Copy every ``p.grad`` that does NOT alias the flat gradient buffer into its slot (a reducer that swaps
        ``.grad`` for its own bucket views - DDP ``gradient_as_bucket_view=True`` - leaves the reduced values there);
        returns how many were copied.  Called by ``FusedAdam.step``."""
        self.flat_grad()
        if self._gviews is None:
            self._gviews = self._views_of(self._flat_grad)
        tr = self._trainable()
        # probe three parameters first: a reducer that swaps .grad does so for all of them
        if not any(q.grad is not None and q.grad.data_ptr() != self._gviews[id(q)].data_ptr()
                   for q in (tr[0], tr[len(tr) // 2], tr[-1])):
            return 0
        dst, src = [], []
        for p in tr:
            gv = self._gviews[id(p)]
            if p.grad is not None and p.grad.data_ptr() != gv.data_ptr():
                dst.append(gv)
                src.append(p.grad)
        if dst:
            torch._foreach_copy_(dst, src)
        return len(dst)

    def set_reducer(self, reducer):
        """Attach a gradient reducer (psld_amd.ddp.BucketReducer) fed during backward."""
        self._reducer = reducer

    # ---- forward -----------------------------------------------------------------------------------------
    def forward(self, x: Tensor, time_cond: Tensor) -> Tensor:
        if not x.is_cuda:
            raise RuntimeError("psld_amd.NCSNpp runs on MI355X only: the HIP extension has no CPU fallback "
                               "(use oracle/psld_oracle.py as the CPU checker)")
        if x.dtype != torch.float32 or time_cond.dtype != torch.float32:
            raise RuntimeError("NCSNpp expects float32 x and time_cond (ncsnpp.py:287; psld.py:354)")
        ops.lib()
        x = x.contiguous()
        t = time_cond.contiguous()
        self.flatten_parameters()
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self._params())
        if need_grad or (torch.is_grad_enabled() and x.requires_grad):
            self.flat_grad()
            if self._anchor is None or self._anchor.device != x.device:
                self._anchor = torch.zeros(1, device=x.device, requires_grad=True)
            if need_grad and self._params_visible():
                if self._reducer is not None:
                    raise RuntimeError("psld_amd.NCSNpp: autograd_params (gradients through AccumulateGrad, for torch DDP) "
                                       "and a BucketReducer are two gradient exchanges: detach one of them")
                return _NCSNppParamFn.apply(x, t, self, *self._trainable())
            return _NCSNppFn.apply(x, t, self._anchor, self)
        if self.use_graphs and not self.training:
            return self._graph_forward(x, t)
        with torch.no_grad():
            return _Exec(self, record=False).run(x, t)

    # ---- HIP-graph replay of the inference forward (launch-bound regime: small sampling batches) ----------
    def enable_graphs(self, flag: bool = True):
        """Capture the eval-mode forward into a HIP graph per input shape and replay it: one graph launch
        instead of ~600 kernel launches issued from Python (the reference samples at 16 images/GPU,
        where the forward is launch-bound).  Weights may keep changing (EMA): packed copies are refreshed
        in place before a replay."""
        self.use_graphs = bool(flag)
        if not flag:
            self._graphs = {}

    def _graph_forward(self, x: Tensor, t: Tensor) -> Tensor:
        key = (tuple(x.shape), x.device.index)
        ent = self._graphs.get(key)
        if ent is None:
            sx, st = x.clone(), t.clone()
            cur = torch.cuda.current_stream()
            warm = torch.cuda.Stream(device=x.device)
            warm.wait_stream(cur)
            with torch.cuda.stream(warm), torch.no_grad():
                for _ in range(2):   # packs weights, sizes workspaces, one-time kernel attributes
                    _Exec(self, record=False).run(sx, st)
            cur.wait_stream(warm)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph), torch.no_grad():
                sy = _Exec(self, record=False).run(sx, st)
            ent = self._graphs[key] = [graph, sx, st, sy, None]
            self.pin_scratch()       # the graph holds raw pointers into arenas / job tables: no eviction, no freeing from now on
        graph, sx, st, sy, stamp = ent
        now = (self._epoch, self._flat._version)
        if stamp != now:
            self._temb_plan()                                   # gathered Dense_0 weights (refreshed in place)
            for ck, cv in list(self._pack_cache.items()):       # refresh, in the same storage, what the graph reads
                if ck[-1] == "gfrag":
                    self._gfrag(cv[2], ck[1], cv[3])
                elif ck[-1] == "pfrag":
                    if ck[1] in ("fwd", "qkv_f"):
                        n_, k_, sn_, sk_, c0_, ct_ = cv[3]
                        self._pfrag(cv[2], ck[1], n_, k_, sn_, sk_, into=cv[1] if ck[1] == "qkv_f" else None, chunk0=c0_,
                                    chunks_total=ct_)
                elif ck[-1] == "bufs":                          # an attention block's gathered q | k | v bias
                    m_ = cv[4]
                    if self._qkv_bias_stamp.get(id(m_)) != (self._epoch, m_.NIN_0.b._version, m_.NIN_0.b.data_ptr()):
                        self._refresh_qkv_biases()
                elif ck[-1] == "wfrag":                         # Winograd fragments the captured forward reads
                    if not ck[1]:
                        self._wfrag(self._conv_by_weight[ck[0]], False)
                elif ck[-1] == "frag":
                    if not ck[1]:
                        self._frag(self._conv_by_weight[ck[0]], False)
                elif len(ck) == 2 and not ck[1]:
                    self._packed(self._conv_by_weight[ck[0]])
            ent[4] = now
        sx.copy_(x)
        st.copy_(t)
        graph.replay()
        return sy.clone()

    def __deepcopy__(self, memo):
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        skip = {"_tables", "_parena", "_sarena", "_persistent", "_flat", "_flat_grad", "_pack_cache", "_frag_table", "_wfrag_table", "_qkv_bias_table", "_qkv_bias_stamp", "_temb_plan_cache", "_anchor", "_reducer", "_offsets", "_module_offs", "_posfreq",
                "_side", "_plist", "_tlist", "_gviews", "_dropout_seed_dev", "_graphs", "_conv_by_weight", "_scratch_grad", "_sviews"}
        for k, v in self.__dict__.items():
            if k in skip:
                continue
            new.__dict__[k] = copy.deepcopy(v, memo)
        new._flat = new._flat_grad = new._offsets = new._anchor = new._reducer = new._posfreq = new._side = None
        new._tables = ops.TableCache()
        new._parena = new._sarena = None
        new._persistent = {}
        new._module_offs = None
        new._plist = new._tlist = new._gviews = None
        new._graphs = {}
        new._conv_by_weight = {}
        new._pending = 0
        new._backward_count = 0
        new._dropout_seed_dev = None
        new._scratch_grad = new._sviews = None
        new._accumulating = new._grad_stale = False
        new._pack_cache = {}
        new._frag_table = None
        new._wfrag_table = None
        new._qkv_bias_table = None
        new._qkv_bias_stamp = {}
        new._temb_plan_cache = None
        new._pack_key = None
        new._epoch = 0
        # detach copied params from the source's flat buffer (they are re-flattened on first use)
        for p in new.parameters():
            p.data = p.data.clone()
        return new


@register_module(category="clf_fn", name="ncsnpp_clf")
class NCSNppClassifier(NCSNpp):
    """Noise-conditioned classifier for guidance (ncsnpp_clf.py:36-283): the NCSN++ time
    embedding, stem, down path and middle block - the same modules, kernels and executor as ``NCSNpp`` - followed
    by flatten + Linear(bias=False) to ``n_cls`` logits.  Built from the ``clf`` config node (reads
    ``model.clf_fn``); ``clf(x f32[B,in_ch,H,W], t f32[B]) -> f32[B,n_cls]`` with autograd to the parameters
    (training, ``tce_loss``) and to ``x`` (the guidance gradient of ``cc_em_sde``)."""

    is_classifier = True

    def _net_config(self, config):
        return config.model.clf_fn
