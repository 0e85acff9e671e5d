"""Pins the CPU oracle (oracle/psld_oracle.py) against the golden vectors captured from the
real reference by tools/gen_golden.py.  CPU-only; no reference import at test time."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import psld_oracle as O
from psld_amd import config as C
from tests.conftest import GOLDEN
from tests.synth import synth_state_dict

T = torch.from_numpy


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def make_sde(nu=4.01, gamma=0.01, dm="lower"):
    return O.PSLDOracle(8.0, 8.0, nu, gamma, 0.04, 1e-9, dm)


def test_sde_coefficients(golden):
    g = golden("sde_coeffs.npz")
    ts = T(g["t"])
    for i, (nu, ga) in enumerate(g["pairs"]):
        for dm in ("lower", "upper"):
            sde = make_sde(float(nu), float(ga), dm)
            var = sde.cov(0.0, sde.mm_0, ts)
            np.testing.assert_allclose(torch.stack(var).numpy(), g[f"cov_{i}"], rtol=1e-14, atol=0)
            np.testing.assert_allclose(torch.stack(sde.coeff(var)).numpy(), g[f"coeff_{dm}_{i}"], rtol=1e-13, atol=0)
            np.testing.assert_allclose(torch.stack(sde.inv_coeff(var)).numpy(), g[f"inv_{dm}_{i}"], rtol=1e-13, atol=0)
            vd = sde.cov(0.0, 0.0, ts[1:])
            np.testing.assert_allclose(torch.stack(vd).numpy(), g[f"covdsm_{i}"], rtol=1e-14, atol=0)
    # SURVEY.md §8c spot values for (4.01, 0.01)
    var = make_sde().cov(0.0, make_sde().mm_0, ts)
    np.testing.assert_allclose(var[0].numpy(), [8.0126e-07, 0.228011, 0.986992, 0.999985], rtol=2e-5)


def test_perturb_and_drift(golden):
    g = golden("sde_perturb.npz")
    sde = make_sde()
    x0, eps, t = T(g["x0"]), T(g["eps"]), T(g["t"])
    u, mu, _ = sde.perturb_data(x0, torch.zeros_like(x0), 0, sde.mm_0, t, eps)
    assert u.dtype == torch.float64
    assert torch.equal(u, T(g["u_hsm"])) and torch.equal(mu, T(g["mu_hsm"]))
    u, mu, _ = sde.perturb_data(x0, T(g["m0"]), 0, 0.0, T(g["t_dsm"]), eps)
    assert torch.equal(u, T(g["u_dsm"])) and torch.equal(mu, T(g["mu_dsm"]))
    uu = T(g["u"])
    f, gg = sde.sde(uu, t)
    assert torch.equal(f, T(g["f"])) and torch.equal(gg, T(g["g"]))
    fake = lambda a, b: 0.1 * a + b.view(-1, 1, 1, 1)
    for pf, tag in ((False, ""), (True, "_pf")):
        fb, gb = sde.reverse_sde(uu, t, fake, probability_flow=pf)
        assert fb.dtype == torch.float64
        assert torch.equal(fb, T(g["f_bar" + tag])) and torch.equal(gb, T(g["g_bar" + tag]))
    sc = sde.get_score(T(g["eps_score"]), 0, sde.mm_0, t)
    assert sc.dtype == torch.float32 and torch.equal(sc, T(g["score"]))


def test_nan_guard_raises():
    sde = make_sde()
    with pytest.raises(ValueError, match="Numerical precision error"):
        sde.coeff((torch.tensor([-1.0], dtype=torch.float64), torch.tensor([0.0], dtype=torch.float64),
                   torch.tensor([1.0], dtype=torch.float64)))


def test_fir(golden):
    g = golden("fir.npz")
    x, k = T(g["x"]), T(g["kasym"])
    for i, (up, dn, p0, p1) in enumerate(g["cases"]):
        y = O.upfirdn2d(x, k, int(up), int(dn), (int(p0), int(p1)))
        assert y.shape == g[f"y_{i}"].shape
        np.testing.assert_allclose(y.numpy(), g[f"y_{i}"], rtol=0, atol=2e-6)
    x2 = T(g["x2"])
    np.testing.assert_allclose(O.upsample_2d(x2).numpy(), g["up2"], atol=1e-6)
    np.testing.assert_allclose(O.downsample_2d(x2).numpy(), g["down2"], atol=1e-6)
    np.testing.assert_allclose(O.conv_downsample_2d(x2, T(g["w_cd"])).numpy(), g["convdown2"], atol=1e-5)


def _layer_meta():
    with open(os.path.join(GOLDEN, "layers_meta.json")) as fh:
        return json.load(fh)


def _sd_for(meta, prefix="all_modules.0"):
    sd = synth_state_dict([(k, tuple(s)) for k, s in meta["keys"]], meta["seed"])
    return {f"{prefix}.{k}": v for k, v in sd.items()}


@pytest.mark.parametrize("name,kw", [("res_plain", {}), ("res_widen", {}), ("res_cat", {}),
                                     ("res_down", {"down": True}), ("res_up", {"up": True})])
def test_resblock_fwd_bwd(golden, name, kw):
    g = golden("layers.npz")
    sd = {k: v.requires_grad_(True) for k, v in _sd_for(_layer_meta()[name]).items()}
    x = T(g[f"{name}.in0"]).requires_grad_(True)
    temb = T(g[f"{name}.in1"]).requires_grad_(True)
    y = O.resblock_biggan(x, temb, sd, "all_modules.0", **kw)
    assert rel_l2(y, T(g[f"{name}.y"])) < 1e-6
    y.backward(T(g[f"{name}.gy"]))
    assert rel_l2(x.grad, T(g[f"{name}.gin0"])) < 1e-5
    assert rel_l2(temb.grad, T(g[f"{name}.gin1"])) < 1e-5
    for k, v in sd.items():
        gk = f"{name}.gw.{k[len('all_modules.0.'):]}"
        assert rel_l2(v.grad, T(g[gk])) < 1e-5, k


@pytest.mark.parametrize("name", ["attn16", "attn8"])
def test_attn_fwd_bwd(golden, name):
    g = golden("layers.npz")
    sd = {k: v.requires_grad_(True) for k, v in _sd_for(_layer_meta()[name]).items()}
    x = T(g[f"{name}.in0"]).requires_grad_(True)
    y = O.attn_block(x, sd, "all_modules.0")
    assert rel_l2(y, T(g[f"{name}.y"])) < 1e-6
    y.backward(T(g[f"{name}.gy"]))
    assert rel_l2(x.grad, T(g[f"{name}.gin0"])) < 1e-5
    for k, v in sd.items():
        assert rel_l2(v.grad, T(g[f"{name}.gw.{k[len('all_modules.0.'):]}"])) < 1e-5, k


@pytest.mark.parametrize("name", ["pyr_down6", "pyr_down32"])
def test_pyramid_downsample(golden, name):
    g = golden("layers.npz")
    sd = {k: v.requires_grad_(True) for k, v in _sd_for(_layer_meta()[name]).items()}
    x = T(g[f"{name}.in0"]).requires_grad_(True)
    y = O.pyramid_downsample(x, sd, "all_modules.0")
    assert rel_l2(y, T(g[f"{name}.y"])) < 1e-6
    y.backward(T(g[f"{name}.gy"]))
    assert rel_l2(x.grad, T(g[f"{name}.gin0"])) < 1e-5


def test_gaussian_fourier(golden):
    g = golden("layers.npz")
    W = _sd_for(_layer_meta()["gfp"])["all_modules.0.W"]
    y = O.gaussian_fourier(torch.log(T(g["gfp.t"])), W)
    assert torch.equal(y, T(g["gfp.y"]))


def _net_meta():
    with open(os.path.join(GOLDEN, "net_meta.json")) as fh:
        return json.load(fh)


def _net_cfg(name):
    if name == "tiny":
        return C.tiny()
    if name == "tiny_ablation":
        c = C.tiny()
        c.model.score_fn.embedding_type = "positional"
        c.model.score_fn.fir = False
        c.model.score_fn.progressive_input = "none"
        return c
    if name == "tiny_out3":
        c = C.tiny()
        c.model.score_fn.out_ch = 3
        c.model.sde.gamma, c.model.sde.nu = 0.0, 4.0
        return c
    if name == "c10_sota":
        return C.c10_sota()
    if name == "celeba64":
        return C.celeba64_sota()
    raise KeyError(name)


@pytest.mark.parametrize("name", ["tiny", "tiny_ablation", "tiny_out3", "c10_sota", "celeba64"])
def test_full_network_forward(golden, name):
    meta = _net_meta()[name]
    g = golden(f"net_{name}.npz")
    sd = synth_state_dict([(k, tuple(s)) for k, s in meta["keys"]], meta["seed"])
    with torch.no_grad():
        y = O.ncsnpp_forward(sd, _net_cfg(name), T(g["x"]), T(g["t"]))
    assert rel_l2(y, T(g["y"])) < 2e-6


def test_state_dict_census():
    m = _net_meta()
    assert m["c10_sota"]["n_keys"] == 749 and m["c10_sota"]["n_params"] == 97627910
    assert m["celeba64"]["n_params"] == 62769286
    assert O.count_resblocks(C.c10_sota()) == 57


def _tiny_sd():
    meta = _net_meta()["tiny"]
    return synth_state_dict([(k, tuple(s)) for k, s in meta["keys"]], meta["seed"])


def test_hsm_loss_and_grads(golden):
    g = golden("loss_tiny.npz")
    cfg = C.tiny()
    sde = O.PSLDOracle.from_config(cfg)
    sd = {k: v.requires_grad_(not k == "all_modules.0.W") for k, v in _tiny_sd().items()}
    x0, eps, t = T(g["x0"]), T(g["eps"]), T(g["t"])
    loss = O.psld_score_loss(sde, x0, t, lambda z, tt: O.ncsnpp_forward(sd, cfg, z, tt), eps)
    assert abs(loss.item() - float(g["loss"])) < 1e-6 * abs(float(g["loss"]))
    loss.backward()
    norms = dict(zip(g["grad_norm_keys"].tolist(), g["grad_norms"].tolist()))
    for k, v in sd.items():
        if v.grad is not None:
            assert abs(v.grad.norm().item() - norms[k]) <= 2e-5 * norms[k] + 1e-9, k
    for k in g.files:
        if k.startswith("g:"):
            assert rel_l2(sd[k[2:]].grad, T(g[k])) < 1e-5, k
    # DSM value: m_0 = sqrt(mm_0) * randn (losses.py:96)
    cfg.training.mode = "dsm"
    with torch.no_grad():
        ld = O.psld_score_loss(sde, x0, t.clamp(min=1e-3), lambda z, tt: O.ncsnpp_forward(sd, cfg, z, tt), eps,
                               mode="dsm", m_0=np.sqrt(sde.mm_0) * T(g["dsm_m0_unit"]))
    assert abs(ld.item() - float(g["loss_dsm"])) < 1e-6 * abs(float(g["loss_dsm"]))


def test_train_steps(golden):
    g = golden("train_tiny.npz")
    cfg = C.tiny()
    sde = O.PSLDOracle.from_config(cfg)
    sd = _tiny_sd()
    sd0 = {k: v.clone() for k, v in sd.items()}
    ema = {k: v.clone() for k, v in sd.items()}
    state = {}
    for step in range(3):
        loss, gnorm, _ = O.train_step(sde, sd, cfg, T(g[f"x0_{step}"]), T(g[f"t_{step}"]), T(g[f"eps_{step}"]),
                                      state, step + 1, ema_sd=ema)
        assert abs(loss.item() - g["losses"][step]) < 2e-5 * abs(g["losses"][step])
        assert abs(gnorm.item() - g["grad_norms"][step]) < 1e-4 * g["grad_norms"][step]
    keys = g["keys"].tolist()
    for k, dn, en in zip(keys, g["param_delta_norms"], g["ema_delta_norms"]):
        d = (sd[k] - sd0[k]).double().norm().item()
        assert abs(d - dn) <= 5e-3 * dn + 1e-9, (k, d, dn)
        e = (ema[k] - sd0[k]).double().norm().item()
        assert abs(e - en) <= 5e-2 * en + 2e-8, (k, e, en)  # EMA delta ~ (1-tau)*delta: fp32 rounding noise
    for k in g.files:
        if k.startswith("p:"):
            np.testing.assert_allclose(sd[k[2:]].numpy(), g[k], rtol=0, atol=2e-6)
        if k.startswith("e:"):
            np.testing.assert_allclose(ema[k[2:]].numpy(), g[k], rtol=0, atol=1e-7)


@pytest.mark.parametrize("tag", ["3_uniform", "3_quadratic", "10_uniform", "10_quadratic"])
def test_em_sampler(golden, tag):
    g = golden("em_tiny.npz")
    cfg = C.tiny()
    sde = O.PSLDOracle.from_config(cfg)
    sd = _tiny_sd()
    seen = []

    def score_fn(u, tt):
        assert u.dtype == torch.float32 and tt.dtype == torch.float32
        seen.append(tt[0].clone())
        return O.ncsnpp_forward(sd, cfg, u, tt)

    n_disc, stride = tag.split("_")
    ts, n = O.sampling_times(sde.T, cfg.evaluation.eval_eps, int(n_disc), True, stride)
    np.testing.assert_array_equal(ts.numpy(), g[f"ts_{tag}"])
    x = O.em_sample(sde, score_fn, T(g[f"batch_{tag}"]), ts, n, True, cfg.evaluation.eval_eps,
                    noise=list(T(g[f"noise_{tag}"])))
    assert x.dtype == torch.float64
    assert rel_l2(x, T(g[f"x_{tag}"])) < 1e-6
    np.testing.assert_array_equal(torch.stack(seen).numpy(), g[f"seen_t_{tag}"])


@pytest.mark.parametrize("stride", ["uniform", "quadratic"])
def test_em_sampler_c10_sota(golden, stride):
    """BASELINE configs[4] on its own network: the oracle's EM loop over the C10-SOTA NCSN++ against the
    reference's EulerMaruyamaSampler (tools/gen_golden.py em_c10_section), 3 predictor steps + denoise."""
    g = golden("em_c10_sota.npz")
    cfg = C.c10_sota()
    sde = O.PSLDOracle.from_config(cfg)
    meta = _net_meta()["c10_sota"]
    sd = synth_state_dict([(k, tuple(s)) for k, s in meta["keys"]], meta["seed"])
    seen = []

    def score_fn(u, tt):
        assert u.dtype == torch.float32 and tt.dtype == torch.float32
        seen.append(tt[0].clone())
        return O.ncsnpp_forward(sd, cfg, u, tt)

    ts, n = O.sampling_times(sde.T, cfg.evaluation.eval_eps, 4, True, stride)
    np.testing.assert_array_equal(ts.numpy(), g[f"ts_{stride}"])
    x = O.em_sample(sde, score_fn, T(g[f"batch_{stride}"]), ts, n, True, cfg.evaluation.eval_eps,
                    noise=list(T(g[f"noise_{stride}"])))
    assert x.dtype == torch.float64
    assert rel_l2(x, T(g[f"x_{stride}"])) < 2e-6
    np.testing.assert_array_equal(torch.stack(seen).numpy(), g[f"seen_t_{stride}"])


@pytest.mark.parametrize("tag", ["xm_3", "xm_6", "m_3", "m_6"])
def test_sscs_sampler(golden, tag):
    """symmetric-splitting sampler, default (score_xm) and gamma=0 (score_m, out_ch=3)."""
    g = golden("sscs_tiny.npz")
    name = "tiny" if tag.startswith("xm") else "tiny_out3"
    cfg = _net_cfg(name)
    meta = _net_meta()[name]
    sd = synth_state_dict([(k, tuple(s)) for k, s in meta["keys"]], meta["seed"])
    sde = O.PSLDOracle.from_config(cfg)
    n = int(tag.split("_")[1]) - 1
    x = O.sscs_sample(sde, lambda u, t: O.ncsnpp_forward(sd, cfg, u, t), T(g[f"batch_{tag}"]), T(g[f"ts_{tag}"]), n,
                      True, cfg.evaluation.eval_eps, noise=list(T(g[f"noise_{tag}"])))
    assert x.dtype == torch.float64
    assert rel_l2(x, T(g[f"x_{tag}"])) < 1e-12


@pytest.mark.parametrize("tag", ["hsm_3", "hsm_6", "dsm_3", "dsm_6"])
def test_inpainting_sampler(golden, tag):
    """ES3EulerMaruyamaInpainter (EM step, re-perturbed known pixels, mask combine) with every
    random draw of the reference replayed in call order."""
    g = golden("inpaint_tiny.npz")
    cfg = _net_cfg("tiny")
    meta = _net_meta()["tiny"]
    sd = synth_state_dict([(k, tuple(s)) for k, s in meta["keys"]], meta["seed"])
    sde = O.PSLDOracle.from_config(cfg)
    mode, n_disc = tag.split("_")
    draws = iter([T(g[f"draw_{tag}_{i}"]) for i in range(int(g[f"ndraws_{tag}"]))])

    def draw(shape, dtype):
        d = next(draws)
        assert tuple(d.shape) == tuple(shape)
        return d.to(dtype)

    x = O.inpaint_sample(sde, lambda u, t: O.ncsnpp_forward(sd, cfg, u, t), T(g[f"x0_{tag}"]), T(g[f"mask_{tag}"]),
                         T(g[f"ts_{tag}"]), int(n_disc) - 1, True, cfg.evaluation.eval_eps, training_mode=mode, draw=draw)
    assert x.dtype == torch.float64 and next(draws, None) is None
    assert rel_l2(x, T(g[f"x_{tag}"])) < 1e-6
    known = T(g[f"mask_{tag}"]).bool()
    # where the mask is 1 the x half of the result is the perturbation MEAN of the known image at t = eps
    assert rel_l2(x[:, :3][known], (T(g[f"x0_{tag}"]).double() * 1.0)[known]) < 0.05


def _clf_sd():
    with open(os.path.join(GOLDEN, "clf_meta.json")) as fh:
        meta = json.load(fh)
    return synth_state_dict([(k, tuple(sh)) for k, sh in meta["keys"]], meta["seed"]), meta


def test_classifier_logits_and_input_gradient(golden):
    """NCSNppClassifier (ncsnpp_clf.py) - logits and d log p(y|x) / dx, the guidance signal."""
    g = golden("clf_tiny.npz")
    sd, _ = _clf_sd()
    cfg = C.tiny_clf()
    x = T(g["x"]).requires_grad_()
    logits = O.ncsnpp_clf_forward(sd, cfg, x, T(g["t"]))
    assert rel_l2(logits, T(g["logits"])) < 1e-5
    sel = torch.log_softmax(logits, dim=-1)[range(4), T(g["y"])]
    assert rel_l2(torch.autograd.grad(sel.sum(), x)[0], T(g["dsel_dx"])) < 1e-5


def test_classifier_tce_loss_and_grads(golden):
    g = golden("clf_tiny.npz")
    sd, _ = _clf_sd()
    sd = {k: v.requires_grad_() for k, v in sd.items()}
    cfg, sde = C.tiny_clf(), O.PSLDOracle.from_config(C.tiny())
    loss, acc = O.tce_loss(sde, T(g["x0"]), T(g["y"]), T(g["t_loss"]), lambda u, t: O.ncsnpp_clf_forward(sd, cfg, u, t),
                           m0_draw=T(g["m0_draw"]), eps=T(g["eps"]))
    assert abs(loss.item() - float(g["loss"])) < 1e-6 * abs(float(g["loss"])) and float(acc) == float(g["acc"])
    loss.backward()
    norms = dict(zip(g["grad_norm_keys"].tolist(), g["grad_norms"].tolist()))
    for k, v in sd.items():
        if v.grad is not None:
            assert abs(v.grad.norm().item() - norms[k]) <= 2e-5 * norms[k] + 1e-9, k
    for k in g.files:
        if k.startswith("g:"):
            assert rel_l2(sd[k[2:]].grad, T(g[k])) < 1e-5, k


@pytest.mark.parametrize("tag", ["cc3", "cc5"])
def test_class_conditional_em_sampler(golden, tag):
    """ClassCondEulerMaruyamaSampler (samplers/sde.py:62-114): scalar label and per-sample label tensor."""
    g = golden("clf_tiny.npz")
    csd, meta = _clf_sd()
    ccfg, dcfg = C.tiny_clf(), C.tiny()
    nm = _net_meta()["tiny"]
    sd = synth_state_dict([(k, tuple(s)) for k, s in nm["keys"]], nm["seed"])
    sde = O.PSLDOracle.from_config(dcfg)
    lab = T(g[f"label_{tag}"])
    lab = int(lab) if lab.dim() == 0 else lab
    n = int(tag[2:]) - 1
    x = O.cc_em_sample(sde, lambda u, t: O.ncsnpp_forward(sd, dcfg, u, t), lambda u, t: O.ncsnpp_clf_forward(csd, ccfg, u, t),
                       T(g[f"batch_{tag}"]), T(g[f"ts_{tag}"]), n, lab, meta["clf_temp"], True, dcfg.evaluation.eval_eps,
                       noise=list(T(g[f"noise_{tag}"])))
    assert x.dtype == torch.float64
    assert rel_l2(x, T(g[f"x_{tag}"])) < 1e-6


def test_writer_loader_edges(golden):
    """vectors produced by the reference's save_as_images (PNG read back) and data_scaler."""
    g = golden("edges.npz")
    np.testing.assert_array_equal(O.samples_to_uint8(T(g["pred"])), g["u8"])
    assert torch.equal(O.images_to_tensor(g["img"]), T(g["tens"]))
    assert torch.equal(O.images_to_tensor(g["img"], norm=False), T(g["tens01"]))


def test_bbode_oracle_runs_and_is_consistent():
    """SURVEY 8(f) rank 2 (parity UNPINNED for the torchdiffeq bridge, see oracle header): the restated
    sampler integrates the probability-flow ODE; tighter tolerances change the result by O(tolerance)."""
    cfg = C.tiny()
    meta = _net_meta()["tiny"]
    sd = synth_state_dict([(k, tuple(s)) for k, s in meta["keys"]], meta["seed"])
    sde = O.PSLDOracle.from_config(cfg)
    g = torch.Generator().manual_seed(3)
    batch = torch.cat([torch.randn(1, 3, 16, 16, generator=g), torch.randn(1, 3, 16, 16, generator=g) * np.sqrt(sde.m)], 1)
    fn = lambda u, t: O.ncsnpp_forward(sd, cfg, u, t)
    x1, n1 = O.bbode_sample(sde, fn, batch, 1e-2, 1e-2)
    x2, n2 = O.bbode_sample(sde, fn, batch, 1e-3, 1e-3)
    assert x1.dtype == torch.float64 and n2 >= n1 > 8
    assert rel_l2(x1, x2) < 5e-2


def _vp_setup():
    with open(os.path.join(GOLDEN, "vpsde_meta.json")) as fh:
        meta = json.load(fh)
    return C.tiny_vpsde(), synth_state_dict([(k, tuple(s)) for k, s in meta["keys"]], meta["seed"])


def test_vpsde_baseline(golden):
    """VP-SDE perturbation, eps-MSE loss + gradients and EM sampling vs the reference."""
    g = golden("vpsde_tiny.npz")
    cfg, sd = _vp_setup()
    sde = O.VPSDEOracle(cfg.model.sde.beta_min, cfg.model.sde.beta_max)
    x0, eps, t = T(g["x0"]), T(g["eps"]), T(g["t"])
    xt = sde.perturb_data(x0, t, eps)
    assert xt.dtype == torch.float64 and torch.equal(xt, T(g["x_t"]))
    p = {k: v.requires_grad_(True) for k, v in sd.items()}
    loss = O.score_loss(sde, x0, t, lambda u, tt: O.ncsnpp_forward(p, cfg, u, tt), eps)
    assert abs(loss.item() - float(g["loss"])) < 1e-6 * float(g["loss"])
    loss.backward()
    for k in g.files:
        if k.startswith("g:"):
            assert rel_l2(p[k[2:]].grad, T(g[k])) < 1e-5, k
    with torch.no_grad():
        x = O.em_sample(sde, lambda u, tt: O.ncsnpp_forward(sd, cfg, u, tt), T(g["batch"]), T(g["ts"]), 4, True,
                        cfg.evaluation.eval_eps, noise=list(T(g["noise"])))
    assert rel_l2(x, T(g["x_em"])) < 1e-6


@pytest.mark.parametrize("tag", ["nll_l2_mean", "nll_l2_sum", "fid_l1_mean", "fid_l1_sum"])
def test_vpsde_score_loss_weightings(golden, tag):
    """ScoreLoss 'nll' weighting (losses.py:55-63) and the L1 criterion (:38-39) vs the reference, with gradients."""
    g = golden("vploss_tiny.npz")
    cfg, sd = _vp_setup()
    sde = O.VPSDEOracle(cfg.model.sde.beta_min, cfg.model.sde.beta_max)
    weighting, l_type, red = tag.split("_")
    p = {k: v.requires_grad_(True) for k, v in sd.items()}
    loss = O.score_loss(sde, T(g["x0"]), T(g["t"]), lambda u, tt: O.ncsnpp_forward(p, cfg, u, tt), T(g["eps"]),
                        reduce_mean=red == "mean", weighting=weighting, l_type=l_type)
    ref = float(g["loss_" + tag])
    assert loss.dtype == T(g["loss_" + tag]).dtype
    assert abs(loss.item() - ref) < 2e-6 * abs(ref)
    loss.backward()
    for k in g.files:
        if k.startswith(f"g_{tag}:"):
            assert rel_l2(p[k.split(":", 1)[1]].grad, T(g[k])) < 2e-5, k


def test_predict_x_from_eps(golden):
    """psld.py:289-328 against the reference (three times, f32 state)."""
    g = golden("predict_x.npz")
    sde = make_sde()
    for i, tv in enumerate(g["t"]):
        x, m = sde.predict_x_from_eps(T(g["z"]), T(g["eps"]), torch.tensor(tv, dtype=torch.float64))
        assert x.dtype == torch.float32
        assert torch.equal(x, T(g[f"x_{i}"])) and torch.equal(m, T(g[f"m_{i}"]))
