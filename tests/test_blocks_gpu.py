"""Per-block parity of the HIP path: every block-level golden of tests/golden/layers.npz (captured from the reference:
ResnetBlockBigGANpp plain / C_in != C_out / concatenated input / down / up - layerspp.py:212-274; AttnBlockpp 16x16 and
8x8 - layerspp.py:62-91; the pyramid Downsample fir+conv - layerspp.py:129-163) is fed through the EXECUTOR's own block
functions (psld_amd/score_fn.py: _Exec.resblock / attn / pyramid), forward and backward (grad_x, grad_temb, every
parameter gradient), so that a regression in one block shows up as that block's test and not as a 75-module network
golden.  The golden blocks are 32-96 channels wide (fp32 MFMA engine); the same harness then runs north-star-width
blocks (128 / 256 / 512 channels: limb kernels, Winograd forward and data gradient forced on) against the live oracle.
"""
import json
import os

import pytest
import torch

from oracle import psld_oracle as O
from tests.synth import synth_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
T = torch.from_numpy


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _layer_meta():
    with open(os.path.join(GOLDEN, "layers_meta.json")) as fh:
        return json.load(fh)


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().to(DEV)


def _nchw(x):
    return x.permute(0, 3, 1, 2)


class Harness:
    """A tiny NCSNpp whose module list gets one extra block: its parameters then live in the network's flat buffers
    like any other module's, and ``_Exec`` runs the block exactly as it runs it inside the network."""

    def __init__(self, module, sd, temb_dim=128):
        import psld_amd
        psld_amd.import_modules_into_registry()
        from psld_amd import config as C, ops
        from psld_amd import score_fn as S
        self.S, self.ops = S, ops
        cfg = C.tiny()
        cfg.model.score_fn.dropout = 0.0
        net = S.NCSNpp(cfg)
        net.all_modules.append(module)
        module.load_state_dict(sd, strict=True)
        self.net = net.to(DEV).eval()
        self.mod = self.net.all_modules[-1]
        self.net.flatten_parameters()
        self.ex = S._Exec(self.net, record=True)
        self.ex.tp_all = self.ex.dtp_all = None
        self.ex.temb_act = None

    def set_temb(self, temb):
        self.temb = temb.to(DEV).contiguous()
        self.ex.temb_act = self.S._Node(self.ops.silu(self.temb))

    def backward(self, out_node, gy_nhwc):
        net, ex = self.net, self.ex
        net._begin_backward()
        out_node.g = gy_nhwc.contiguous()
        with self.ops.stream_scope():
            if ex.defer:
                net._param_arena().reset()
            for fn, _ in reversed(ex.tape):
                fn()
            ex.finish_backward()
        ex.tape = None
        net._end_backward()
        torch.cuda.synchronize()

    def grad(self, name):
        p = dict(self.mod.named_parameters())[name]
        return self.net._grad_view(p)

    def temb_grad(self):
        return self.ops.silu_bwd(self.temb, self.ex.temb_act.g)


def _block_sd(name):
    meta = _layer_meta()[name]
    return synth_state_dict([(k, tuple(s)) for k, s in meta["keys"]], meta["seed"])


@pytest.mark.parametrize("name,kw,two_source", [("res_plain", {}, False), ("res_widen", {}, False), ("res_cat", {}, False),
                                                ("res_cat", {}, True), ("res_down", {"down": True}, False),
                                                ("res_up", {"up": True}, False)])
def test_resblock_golden_on_the_hip_path(golden, name, kw, two_source):
    from psld_amd import score_fn as S
    g = golden("layers.npz")
    sd = _block_sd(name)
    cin, cout = sd["Conv_0.weight"].shape[1], sd["Conv_0.weight"].shape[0]
    h = Harness(S.ResnetBlockBigGANpp(cin, cout, temb_dim=128, dropout=0.0, **kw), sd)
    h.set_temb(T(g[f"{name}.in1"]))
    x = _nhwc(T(g[f"{name}.in0"]))
    if two_source:      # the up path's torch.cat([h, hs.pop()]) (ncsnpp.py:374) as two sources, never materialised when it can be
        a, b = h.S._Node(x[..., :64].contiguous()), h.S._Node(x[..., 64:].contiguous())
        xin = h.ex.concat(a, b, h.mod)
    else:
        xin = a = h.S._Node(x)
    with h.ops.stream_scope():
        out = h.ex.resblock(xin, h.mod)
    assert rel_l2(_nchw(out.v), T(g[f"{name}.y"])) < 1e-5
    h.backward(out, _nhwc(T(g[f"{name}.gy"])))
    gx = torch.cat([a.g, b.g], dim=-1) if two_source else a.g
    assert rel_l2(_nchw(gx), T(g[f"{name}.gin0"])) < 1e-5
    assert rel_l2(h.temb_grad(), T(g[f"{name}.gin1"])) < 1e-5
    for k in sd:
        assert rel_l2(h.grad(k), T(g[f"{name}.gw.{k}"])) < 1e-5, k


@pytest.mark.parametrize("name", ["attn16", "attn8"])
def test_attention_golden_on_the_hip_path(golden, name):
    from psld_amd import score_fn as S
    g = golden("layers.npz")
    sd = _block_sd(name)
    c = sd["NIN_0.W"].shape[0]
    h = Harness(S.AttnBlockpp(c), sd)
    xn = h.S._Node(_nhwc(T(g[f"{name}.in0"])))
    with h.ops.stream_scope():
        out = h.ex.attn(xn, h.mod)
    assert rel_l2(_nchw(out.v), T(g[f"{name}.y"])) < 1e-5
    h.backward(out, _nhwc(T(g[f"{name}.gy"])))
    assert rel_l2(_nchw(xn.g), T(g[f"{name}.gin0"])) < 1e-5
    for k in sd:
        if k == "NIN_1.b":      # the key bias shifts every logit of a row alike: softmax cancels it, the gradient is rounding noise
            assert float(h.grad(k).abs().max()) < 1e-4 * float(h.grad("NIN_0.b").abs().max())
            continue
        assert rel_l2(h.grad(k), T(g[f"{name}.gw.{k}"])) < 1e-5, k


@pytest.mark.parametrize("name", ["pyr_down6", "pyr_down32"])
def test_pyramid_downsample_golden_on_the_hip_path(golden, name):
    """layerspp.Downsample(fir, with_conv) as the executor runs it: fused with the input-pyramid combine
    (pyr + h)/sqrt(2) of ncsnpp.py:350-357 - with h = 0 the output is the golden times 1/sqrt(2)."""
    from psld_amd import score_fn as S
    g = golden("layers.npz")
    sd = _block_sd(name)
    cout, cin = sd["Conv2d_0.weight"].shape[:2]
    h = Harness(S.Downsample(cin, cout, True), sd)
    s = h.ex.s
    x = T(g[f"{name}.in0"]).to(DEV)
    first = cin <= 7
    hz = h.S._Node(torch.zeros((x.shape[0], x.shape[2] // 2, x.shape[3] // 2, cout), device=DEV))
    h.ex.want_dx = True
    pyr_in = x.contiguous() if first else h.S._Node(_nhwc(T(g[f"{name}.in0"])))
    with h.ops.stream_scope():
        out = h.ex.pyramid(pyr_in, hz, h.mod, first)
    assert rel_l2(_nchw(out.v) / s, T(g[f"{name}.y"])) < 1e-5
    h.backward(out, _nhwc(T(g[f"{name}.gy"])) / s)
    gx = h.ex.dx_nchw if first else _nchw(pyr_in.g)
    assert rel_l2(gx, T(g[f"{name}.gin0"])) < 1e-5
    for k in sd:
        assert rel_l2(h.grad(k), T(g[f"{name}.gw.{k}"])) < 1e-5, k
    assert rel_l2(hz.g * (1.0 / s), _nhwc(T(g[f"{name}.gy"])) / s) < 1e-6        # the combine's other branch


# ---- north-star widths against the live oracle (limb kernels; Winograd forced on and off) ----------------------------------
@pytest.mark.parametrize("winograd", [2, 0])
@pytest.mark.parametrize("cin,cout,hw,kw,two", [(256, 256, 32, {}, False), (128, 256, 32, {}, False), (512, 256, 16, {}, True),
                                                (384, 256, 32, {}, True), (256, 256, 32, {"down": True}, False),
                                                (256, 256, 8, {"up": True}, False), (256, 256, 8, {}, False)])
def test_wide_resblock_against_the_oracle(cin, cout, hw, kw, two, winograd):
    from psld_amd import ops, score_fn as S
    ops.set_winograd(winograd)
    ops.set_wgrad_winograd(winograd)         # 2: the weight gradients in the Winograd domain too (wgrad_wino.hip), 0: direct
    try:
        b = 2
        mod = S.ResnetBlockBigGANpp(cin, cout, temb_dim=128, dropout=0.0, **kw)
        keys = [(k, tuple(v.shape)) for k, v in mod.state_dict().items()]
        sd = synth_state_dict(keys, 77)
        gen = torch.Generator().manual_seed(5)
        x = torch.randn(b, cin, hw, hw, generator=gen)
        temb = torch.randn(b, 128, generator=gen)
        h = Harness(mod, sd)
        h.set_temb(temb)
        xd = _nhwc(x)
        if two:
            c1 = 256
            a, bb = h.S._Node(xd[..., :c1].contiguous()), h.S._Node(xd[..., c1:].contiguous())
            xin = h.ex.concat(a, bb, h.mod)
        else:
            xin = a = h.S._Node(xd)
        with h.ops.stream_scope():
            out = h.ex.resblock(xin, h.mod)
        # oracle (fp32 on the CPU: its FIR kernels are float32 like the reference's, up_or_down_sampling.py:182)
        osd = {f"m.{k}": v.clone().requires_grad_(True) for k, v in sd.items()}
        xo, to = x.clone().requires_grad_(True), temb.clone().requires_grad_(True)
        yo = O.resblock_biggan(xo, to, osd, "m", **kw)
        gy = torch.randn(*yo.shape, generator=gen)
        yo.backward(gy)
        assert rel_l2(_nchw(out.v), yo) < 5e-6
        h.backward(out, _nhwc(gy))
        gx = torch.cat([a.g, bb.g], dim=-1) if two else a.g
        assert rel_l2(_nchw(gx), xo.grad) < 1e-5
        assert rel_l2(h.temb_grad(), to.grad) < 1e-5
        for k in sd:
            assert rel_l2(h.grad(k), osd[f"m.{k}"].grad) < 2e-5, k
    finally:
        ops.set_winograd(None)
        ops.set_wgrad_winograd(None)


@pytest.mark.parametrize("c,hw", [(256, 16), (256, 8)])
def test_wide_attention_against_the_oracle(c, hw):
    from psld_amd import score_fn as S
    mod = S.AttnBlockpp(c)
    sd = synth_state_dict([(k, tuple(v.shape)) for k, v in mod.state_dict().items()], 78)
    gen = torch.Generator().manual_seed(6)
    x = torch.randn(2, c, hw, hw, generator=gen)
    h = Harness(mod, sd)
    xn = h.S._Node(_nhwc(x))
    with h.ops.stream_scope():
        out = h.ex.attn(xn, h.mod)
    osd = {f"m.{k}": v.double().requires_grad_(True) for k, v in sd.items()}
    xo = x.double().requires_grad_(True)
    yo = O.attn_block(xo, osd, "m")
    gy = torch.randn(*yo.shape, generator=gen)
    yo.backward(gy.double())
    assert rel_l2(_nchw(out.v), yo) < 5e-6
    h.backward(out, _nhwc(gy))
    assert rel_l2(_nchw(xn.g), xo.grad) < 1e-5
    for k in sd:
        if k == "NIN_1.b":
            assert float(h.grad(k).abs().max()) < 1e-4 * float(h.grad("NIN_0.b").abs().max())
            continue
        assert rel_l2(h.grad(k), osd[f"m.{k}"].grad) < 1e-5, k
