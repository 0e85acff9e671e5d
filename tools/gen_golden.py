"""Generate golden vectors from the REAL reference (CPU path), build container only.

    python tools/gen_golden.py            # writes tests/golden/*.npz, *.json

The fixtures are data (inputs + the reference's outputs).  Weights are not stored: they are
regenerated from a seed by tests/synth.py.  Re-running is deterministic on this image.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

from ref_shim import import_reference  # noqa: E402
from psld_amd import config as C  # noqa: E402
from tests.synth import synth_state_dict, synth_inputs  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)


def npy(t):
    return t.detach().cpu().numpy()


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez(path, **{k: (npy(v) if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print(f"  wrote {name}: {os.path.getsize(path)/1024:.1f} KiB")


def keys_shapes(module):
    return [(k, tuple(v.shape)) for k, v in module.state_dict().items()]


def load_synth(module, seed):
    ks = keys_shapes(module)
    sd = synth_state_dict(ks, seed)
    module.load_state_dict(sd, strict=True)
    return ks


def inpaint_section(get, C):
    """M. ES3EulerMaruyamaInpainter (8(f) rank 4): EM step + perturb the known image to the current time + mask
    combine, samplers/sde.py:117-224.  Every random draw (torch.randn of prior_sampling, torch.randn_like of the
    predictor and of _perturb) is replaced by a recorded tensor, in call order."""
    print("inpainting sampler (tiny)")
    PSLD, NCSNpp, IP = get("sde", "psld"), get("score_fn", "ncsnpp"), get("samplers", "ip_em_sde")
    out = {}
    for mode in ("hsm", "dsm"):
        cfg = C.tiny()
        cfg.training.mode = mode
        sde = PSLD(cfg)
        net = NCSNpp(cfg)
        load_synth(net, 1000)
        net.eval()
        sampler = IP(cfg, sde, net)
        for n_disc in (3, 6):
            g = torch.Generator().manual_seed(120 + n_disc)
            x0 = torch.rand(2, 3, 16, 16, generator=g) * 2 - 1
            mask = (torch.rand(2, 3, 16, 16, generator=g) > 0.4).type(torch.long)
            n = n_disc - 1
            tsx = torch.linspace(0, sde.T - cfg.evaluation.eval_eps, n + 1, dtype=torch.float64)
            draws = []
            o_like, o_randn = torch.randn_like, torch.randn

            def draw(shape, dtype):
                d = o_randn(*shape, generator=g, dtype=torch.float64)
                draws.append(d)
                return d.to(dtype)

            torch.randn_like = lambda x_, **kw: draw(tuple(x_.shape), x_.dtype)
            torch.randn = lambda *shape, **kw: draw(tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else shape, torch.float32)
            try:
                xf = sampler.sample((x0, mask), tsx, n, denoise=True, eps=cfg.evaluation.eval_eps)
            finally:
                torch.randn_like, torch.randn = o_like, o_randn
            tag = f"{mode}_{n_disc}"
            out[f"x0_{tag}"], out[f"mask_{tag}"], out[f"ts_{tag}"], out[f"x_{tag}"] = x0, mask, tsx, xf
            out[f"ndraws_{tag}"] = np.array(len(draws))
            for i, d in enumerate(draws):
                out[f"draw_{tag}_{i}"] = d
            assert xf.dtype == torch.float64 and len(draws) == 4 + 3 * (n + 1)
    save("inpaint_tiny.npz", **out)


def clf_section(get, C):
    """N. classifier guidance (8(f) rank 4): NCSNppClassifier logits and input gradient, PSLDTimeCELoss with
    gradients, ClassCondEulerMaruyamaSampler with replayed noise."""
    print("classifier guidance (tiny)")
    PSLD, NCSNpp = get("sde", "psld"), get("score_fn", "ncsnpp")
    Clf, TCE, CC = get("clf_fn", "ncsnpp_clf"), get("losses", "tce_loss"), get("samplers", "cc_em_sde")
    dcfg, ccfg = C.tiny(), C.tiny_clf()
    root = C.with_clf(dcfg, ccfg)
    root.clf.evaluation.clf_temp = 2.5
    sde = PSLD(dcfg)
    clf = Clf(ccfg)
    cks = load_synth(clf, 5000)
    g = torch.Generator().manual_seed(130)
    out = {}
    # forward + gradient of the selected log-probability w.r.t. the input (what the sampler uses)
    clf.eval()
    x = torch.randn(4, 6, 16, 16, generator=g)
    t = torch.rand(4, generator=g) * 0.9 + 0.05
    y = torch.tensor([3, 0, 9, 3])
    xin = x.clone().requires_grad_()
    logits = clf(xin, t)
    sel = torch.log_softmax(logits, dim=-1)[range(4), y]
    out.update(x=x, t=t, y=y, logits=logits.detach(), dsel_dx=torch.autograd.grad(sel.sum(), xin)[0])
    # training loss (HSM) with parameter gradients
    clf.train()
    x0 = torch.rand(4, 3, 16, 16, generator=g) * 2 - 1
    tt = torch.rand(4, generator=g, dtype=torch.float64) * (1 - 1e-5) + 1e-5
    m0d, epsd = torch.randn(4, 3, 16, 16, generator=g), torch.randn(4, 6, 16, 16, generator=g)
    it = iter([m0d, epsd])
    orig = torch.randn_like
    torch.randn_like = lambda x_, **kw: next(it).to(x_.dtype)
    try:
        loss, acc = TCE(root, sde)(x0, y, tt, clf)
    finally:
        torch.randn_like = orig
    loss.backward()
    out.update(x0=x0, t_loss=tt, m0_draw=m0d, eps=epsd, loss=loss.detach(), acc=torch.as_tensor(acc))
    gn = {k: p.grad.norm().item() for k, p in clf.named_parameters() if p.grad is not None}
    out["grad_norm_keys"], out["grad_norms"] = np.array(list(gn.keys())), np.array(list(gn.values()))
    names = [k for k, _ in clf.named_parameters()]
    for k in (names[0], names[3], names[-1]):
        out["g:" + k] = dict(clf.named_parameters())[k].grad
    # class-conditional EM sampling
    clf.eval()
    net = NCSNpp(dcfg)
    load_synth(net, 1000)
    net.eval()
    for lab, n_disc in ((7, 3), (torch.tensor([1, 4]), 5)):
        root.clf.evaluation.label_to_sample = lab
        sampler = CC(root, sde, net, clf)
        batch = torch.cat([torch.randn(2, 3, 16, 16, generator=g),
                           torch.randn(2, 3, 16, 16, generator=g) * np.sqrt(sde.m)], dim=1)
        n = n_disc - 1
        tsx = torch.linspace(0, sde.T - dcfg.evaluation.eval_eps, n + 1, dtype=torch.float64)
        noises = [torch.randn(2, 6, 16, 16, generator=g, dtype=torch.float64) for _ in range(n + 1)]
        it = iter(noises)
        torch.randn_like = lambda x_, **kw: next(it).to(x_.dtype)
        try:
            xf = sampler.sample(batch, tsx, n, denoise=True, eps=dcfg.evaluation.eval_eps)
        finally:
            torch.randn_like = orig
        tag = f"cc{n_disc}"
        out[f"batch_{tag}"], out[f"noise_{tag}"], out[f"ts_{tag}"], out[f"x_{tag}"] = batch, torch.stack(noises), tsx, xf
        out[f"label_{tag}"] = torch.as_tensor(lab)
        assert xf.dtype == torch.float64 and next(it, None) is None
    save("clf_tiny.npz", **out)
    with open(os.path.join(OUT, "clf_meta.json"), "w") as fh:
        json.dump({"seed": 5000, "keys": [[k, list(s)] for k, s in cks], "clf_temp": 2.5}, fh)


def em_c10_section(get, C):
    """configs[4] on its own network: the reference's EulerMaruyamaSampler over the C10-SOTA NCSN++ (seed-2000 synthetic
    weights), B=2, n_discrete_steps=4 (3 predictor steps + the denoising step), noise replayed."""
    print("EM sampler (C10-SOTA)")
    cfg = C.c10_sota()
    sde = get("sde", "psld")(cfg)
    net = get("score_fn", "ncsnpp")(cfg)
    load_synth(net, 2000)
    net.eval()
    seen_t = []

    def score_fn(u, tt):
        seen_t.append(tt.detach().clone())
        return net(u, tt)

    sampler = get("samplers", "em_sde")(cfg, sde, score_fn)
    out = {}
    for stride in ("uniform", "quadratic"):
        g = torch.Generator().manual_seed(4242)
        batch = torch.cat([torch.randn(2, 3, 32, 32, generator=g),
                           torch.randn(2, 3, 32, 32, generator=g) * np.sqrt(sde.m)], dim=1)
        n = 3
        t_final = sde.T - cfg.evaluation.eval_eps
        tsx = torch.linspace(0, t_final, n + 1, dtype=torch.float64)
        if stride == "quadratic":
            tsx = t_final * torch.flip(1 - (tsx / t_final) ** 2.0, dims=[0])
        noises = [torch.randn(2, 6, 32, 32, generator=g, dtype=torch.float64) for _ in range(n)]
        it = iter(noises)
        orig = torch.randn_like
        torch.randn_like = lambda x_, **kw: next(it).to(x_.dtype)
        seen_t.clear()
        try:
            xf = sampler.sample(batch, tsx, n, denoise=True, eps=cfg.evaluation.eval_eps)
        finally:
            torch.randn_like = orig
        assert xf.dtype == torch.float64
        out[f"batch_{stride}"] = batch
        out[f"noise_{stride}"] = torch.stack(noises)
        out[f"x_{stride}"] = xf
        out[f"ts_{stride}"] = tsx
        out[f"seen_t_{stride}"] = torch.stack([s[0] for s in seen_t])
    save("em_c10_sota.npz", **out)


def vp_loss_section(get, C):
    """ScoreLoss beyond the eps-MSE: weighting='nll' (g(t)^2-weighted score error, losses.py:55-63) and the L1 criterion
    of weighting='fid' (losses.py:38-39), mean and sum reductions, with parameter gradients."""
    print("VP-SDE ScoreLoss: nll weighting / l1")
    vcfg = C.tiny_vpsde()
    vsde = get("sde", "vpsde")(vcfg)
    net = get("score_fn", "ncsnpp")(vcfg)
    load_synth(net, 4000)
    net.train()
    g = torch.Generator().manual_seed(93)
    x0 = torch.rand(4, 3, 16, 16, generator=g) * 2 - 1
    eps = torch.randn(4, 3, 16, 16, generator=g)
    t = torch.rand(4, generator=g, dtype=torch.float64) * (1 - 1e-3) + 1e-3
    out = {"x0": x0, "eps": eps, "t": t}
    names = [k for k, _ in net.named_parameters()]
    for weighting, l_type in (("nll", "l2"), ("fid", "l1")):
        for red in (True, False):
            cfg = C.tiny_vpsde()
            cfg.training.loss.weighting, cfg.training.loss.l_type, cfg.training.loss.reduce_mean = weighting, l_type, red
            for p_ in net.parameters():
                p_.grad = None
            loss = get("losses", "score_loss")(cfg, vsde)(x0, t, net, eps=eps)
            loss.backward()
            tag = f"{weighting}_{l_type}_{'mean' if red else 'sum'}"
            out["loss_" + tag] = loss.detach()
            out["gnorm_" + tag] = torch.stack([p_.grad.double().norm() for p_ in net.parameters() if p_.grad is not None]).norm()
            for k in (names[0], names[5], names[-1]):
                out[f"g_{tag}:{k}"] = dict(net.named_parameters())[k].grad.clone()
    save("vploss_tiny.npz", **out)


def predict_x_section(get, C):
    """PSLD.predict_x_from_eps (psld.py:289-328): API surface off the hot path, one time per call."""
    print("predict_x_from_eps")
    sde = get("sde", "psld")(C.c10_sota())
    g = torch.Generator().manual_seed(77)
    z = torch.randn(2, 6, 8, 8, generator=g)
    eps = torch.randn(2, 6, 8, 8, generator=g)
    out = {"z": z, "eps": eps, "t": np.array([0.05, 0.37, 0.9])}
    for i, tv in enumerate(out["t"]):
        x, m = sde.predict_x_from_eps(z, eps, torch.tensor(tv, dtype=torch.float64))
        out[f"x_{i}"], out[f"m_{i}"] = x, m
    save("predict_x.npz", **out)


def main():
    util = import_reference()
    if "--only" in sys.argv:
        which = sys.argv[sys.argv.index("--only") + 1]
        {"inpaint": inpaint_section, "clf": clf_section, "em_c10": em_c10_section, "vp_loss": vp_loss_section, "predict_x": predict_x_section}[which](util.get_module, C)
        return
    get = util.get_module
    PSLD = get("sde", "psld")
    NCSNpp = get("score_fn", "ncsnpp")
    Loss = get("losses", "psld_score_loss")
    EM = get("samplers", "em_sde")
    from models.score_fn.song_sde import layerspp, up_or_down_sampling as uds
    from models.score_fn.song_sde.op import upfirdn2d as ref_upfirdn2d
    import callbacks as ref_callbacks

    # ------------------------------------------------------------------ A. SDE scalars
    print("SDE coefficients")
    ts = torch.tensor([1e-5, 0.1, 0.5, 1.0], dtype=torch.float64)
    pairs = [(4.01, 0.01), (4.02, 0.02), (4.005, 0.005), (4.0, 0.0), (1.0, 2.0)]
    out = {"t": ts, "pairs": np.array(pairs)}
    for i, (nu, ga) in enumerate(pairs):
        for dm in ("lower", "upper"):
            cfg = C.c10_sota()
            cfg.model.sde.nu, cfg.model.sde.gamma, cfg.model.sde.decomp_mode = nu, ga, dm
            sde = PSLD(cfg)
            var = sde._cov(0.0, sde.mm_0, ts)
            out[f"cov_{i}"] = torch.stack(var)
            out[f"coeff_{dm}_{i}"] = torch.stack([c * torch.ones_like(ts) for c in sde.get_coeff(var)])
            out[f"inv_{dm}_{i}"] = torch.stack([c * torch.ones_like(ts) for c in sde.get_inv_coeff(var)])
            var_d = sde._cov(0.0, 0.0, ts[1:])  # DSM: mm_0 = 0 (t=1e-5 is numerically singular there)
            out[f"covdsm_{i}"] = torch.stack(var_d)
    save("sde_coeffs.npz", **out)

    # ------------------------------------------------------------------ B/C. perturb, drift
    print("perturb / reverse-sde")
    cfg = C.c10_sota()
    sde = PSLD(cfg)
    x0, eps, t = synth_inputs(4, 3, 8, seed=11)
    t[0], t[1] = 1e-5, 1.0
    u_hsm, mu_hsm, _ = sde.perturb_data(x0, torch.zeros_like(x0), 0, sde.mm_0, t, eps=eps)
    g = torch.Generator().manual_seed(12)
    m0 = np.sqrt(sde.mm_0) * torch.randn(x0.shape, generator=g)
    t_d = t.clone()
    t_d[0] = 1e-3
    u_dsm, mu_dsm, _ = sde.perturb_data(x0, m0, 0, 0.0, t_d, eps=eps)
    u = torch.randn(4, 6, 8, 8, generator=g, dtype=torch.float64)

    def fake_score(uu, tt):
        assert uu.dtype == torch.float32 and tt.dtype == torch.float32
        return 0.1 * uu + tt.view(-1, 1, 1, 1)

    f, gg = sde.sde(u, t)
    fb, gb = sde.reverse_sde(u, t, fake_score, probability_flow=False)
    fbp, gbp = sde.reverse_sde(u, t, fake_score, probability_flow=True)
    epsn = torch.randn(4, 6, 8, 8, generator=g)
    score = sde.get_score(epsn, 0, sde.mm_0, t)
    save("sde_perturb.npz", x0=x0, eps=eps, t=t, u_hsm=u_hsm, mu_hsm=mu_hsm, m0=m0, t_dsm=t_d,
         u_dsm=u_dsm, mu_dsm=mu_dsm, u=u, f=f, g=gg, f_bar=fb, g_bar=gb, f_bar_pf=fbp, g_bar_pf=gbp,
         eps_score=epsn, score=score)

    # ------------------------------------------------------------------ D. FIR ops
    print("upfirdn2d")
    g = torch.Generator().manual_seed(21)
    x = torch.randn(2, 5, 9, 7, generator=g)
    kasym = torch.tensor([[1.0, 2.0, -1.0], [0.5, 3.0, 0.25], [-2.0, 1.5, 4.0], [0.1, 0.2, 0.3]])
    out = {"x": x, "kasym": kasym}
    cases = [(1, 1, (0, 0)), (2, 1, (2, 1)), (1, 2, (1, 1)), (2, 2, (1, 2)), (1, 1, (2, 2)), (3, 2, (0, 3)),
             (1, 1, (-1, 2))]
    for i, (up, dn, pad) in enumerate(cases):
        out[f"y_{i}"] = ref_upfirdn2d(x, kasym, up=up, down=dn, pad=pad)
    out["cases"] = np.array([[u_, d_, p_[0], p_[1]] for u_, d_, p_ in cases])
    x2 = torch.randn(2, 8, 8, 8, generator=g)
    out["x2"] = x2
    out["up2"] = uds.upsample_2d(x2, (1, 3, 3, 1), factor=2)
    out["down2"] = uds.downsample_2d(x2, (1, 3, 3, 1), factor=2)
    w = torch.randn(6, 8, 3, 3, generator=g)
    out["w_cd"] = w
    out["convdown2"] = uds.conv_downsample_2d(x2, w, k=(1, 3, 3, 1))
    save("fir.npz", **out)

    # ------------------------------------------------------------------ D. layers (fwd + bwd)
    print("layers")
    act = nn.SiLU()
    out = {}
    meta = {}

    def run_block(name, mod, inputs, seed):
        ks = load_synth(mod, seed)
        meta[name] = {"seed": seed, "keys": [[k, list(s)] for k, s in ks]}
        ins = [i.clone().requires_grad_(True) for i in inputs]
        y = mod(*ins)
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(seed + 1))
        y.backward(gy)
        for j, i in enumerate(ins):
            out[f"{name}.in{j}"] = i.detach()
            out[f"{name}.gin{j}"] = i.grad
        out[f"{name}.y"] = y.detach()
        out[f"{name}.gy"] = gy
        for k, p in mod.named_parameters():
            if p.grad is not None:
                out[f"{name}.gw.{k}"] = p.grad

    g = torch.Generator().manual_seed(31)
    temb = torch.randn(2, 128, generator=g)
    rb = lambda **kw: layerspp.ResnetBlockBigGANpp(act=act, temb_dim=128, dropout=0.0, fir=True,
                                                   fir_kernel=(1, 3, 3, 1), init_scale=0.0,
                                                   skip_rescale=True, **kw)
    run_block("res_plain", rb(in_ch=32, out_ch=32), [torch.randn(2, 32, 8, 8, generator=g), temb], 100)
    run_block("res_widen", rb(in_ch=32, out_ch=64), [torch.randn(2, 32, 8, 8, generator=g), temb], 110)
    run_block("res_cat", rb(in_ch=96, out_ch=64), [torch.randn(2, 96, 8, 8, generator=g), temb], 120)
    run_block("res_down", rb(in_ch=32, down=True), [torch.randn(2, 32, 8, 8, generator=g), temb], 130)
    run_block("res_up", rb(in_ch=32, up=True), [torch.randn(2, 32, 8, 8, generator=g), temb], 140)
    run_block("attn16", layerspp.AttnBlockpp(channels=32, skip_rescale=True, init_scale=0.0),
              [torch.randn(2, 32, 16, 16, generator=g)], 150)
    run_block("attn8", layerspp.AttnBlockpp(channels=64, skip_rescale=True, init_scale=0.0),
              [torch.randn(2, 64, 8, 8, generator=g)], 160)
    run_block("pyr_down6", layerspp.Downsample(in_ch=6, out_ch=32, with_conv=True, fir=True,
                                               fir_kernel=(1, 3, 3, 1)),
              [torch.randn(2, 6, 16, 16, generator=g)], 170)
    run_block("pyr_down32", layerspp.Downsample(in_ch=32, out_ch=32, with_conv=True, fir=True,
                                                fir_kernel=(1, 3, 3, 1)),
              [torch.randn(2, 32, 8, 8, generator=g)], 180)
    gfp = layerspp.GaussianFourierProjection(embedding_size=128, scale=16)
    tt = torch.tensor([1e-5, 1e-3, 0.3, 0.99999, 1.0])
    ks = load_synth(gfp, 190)
    meta["gfp"] = {"seed": 190, "keys": [[k, list(s)] for k, s in ks]}
    out["gfp.t"] = tt
    out["gfp.y"] = gfp(torch.log(tt))
    save("layers.npz", **out)
    with open(os.path.join(OUT, "layers_meta.json"), "w") as fh:
        json.dump(meta, fh)

    # ------------------------------------------------------------------ E/F. full nets
    print("full networks")
    net_meta = {}

    def run_net(name, cfg, batch, seed, with_grad=False):
        net = NCSNpp(cfg)
        ks = load_synth(net, seed)
        net.eval()
        size = cfg.data.image_size
        g = torch.Generator().manual_seed(seed + 7)
        x = torch.randn(batch, cfg.model.score_fn.in_ch, size, size, generator=g)
        t = torch.rand(batch, generator=g) * 0.98 + 0.01
        with torch.no_grad():
            y = net(x, t)
        net_meta[name] = {"seed": seed, "n_keys": len(ks), "n_params": int(sum(np.prod(s) for _, s in ks)),
                          "keys": [[k, list(s)] for k, s in ks]}
        save(f"net_{name}.npz", x=x, t=t, y=y)
        print(f"    {name}: |y| rms {y.pow(2).mean().sqrt().item():.4f}, {net_meta[name]['n_params']} params")
        return net

    tiny = C.tiny()
    run_net("tiny", tiny, 2, 1000)
    tiny_abl = C.tiny()
    tiny_abl.model.score_fn.embedding_type = "positional"
    tiny_abl.model.score_fn.fir = False
    tiny_abl.model.score_fn.progressive_input = "none"
    run_net("tiny_ablation", tiny_abl, 2, 1010)
    tiny3 = C.tiny()
    tiny3.model.score_fn.out_ch = 3
    tiny3.model.sde.gamma, tiny3.model.sde.nu = 0.0, 4.0
    run_net("tiny_out3", tiny3, 2, 1020)
    run_net("c10_sota", C.c10_sota(), 2, 2000)
    run_net("celeba64", C.celeba64_sota(), 1, 3000)
    with open(os.path.join(OUT, "net_meta.json"), "w") as fh:
        json.dump(net_meta, fh)

    # ------------------------------------------------------------------ G. loss + grads (tiny)
    print("loss / grads / train steps (tiny)")
    cfg = C.tiny()
    sde = PSLD(cfg)
    net = NCSNpp(cfg)
    load_synth(net, 1000)
    net.train()
    crit = Loss(cfg, sde)
    x0, eps, t = synth_inputs(4, 3, 16, seed=41)
    loss = crit(x0, t, net, eps=eps)
    loss.backward()
    gn = {k: p.grad.norm().item() for k, p in net.named_parameters() if p.grad is not None}
    sel = ["all_modules.3.weight", "all_modules.3.bias", "all_modules.4.Conv_0.weight",
           "all_modules.4.Dense_0.weight", "all_modules.4.GroupNorm_1.weight", "all_modules.1.weight",
           "all_modules.2.bias"]
    names = [k for k, _ in net.named_parameters()]
    attn_keys = [k for k in names if "NIN_1.W" in k][:1] + [k for k in names if "NIN_3.b" in k][:1]
    down_keys = [k for k in names if "Conv2d_0.weight" in k][:1]
    last = [names[-2], names[-1]]
    sel = sel + attn_keys + down_keys + last
    pd = dict(net.named_parameters())
    out = {"x0": x0, "eps": eps, "t": t, "loss": loss.detach(),
           "grad_norm_keys": np.array(list(gn.keys())), "grad_norms": np.array(list(gn.values())),
           "total_norm": torch.linalg.vector_norm(torch.stack([p.grad.norm() for p in net.parameters() if p.grad is not None]))}
    for k in sel:
        out["g:" + k] = pd[k].grad
    # dsm variant of the loss value
    cfg_d = C.tiny()
    cfg_d.training.mode = "dsm"
    torch.manual_seed(5)
    m0_draw = torch.randn_like(x0)  # what losses.py:96 will draw first
    torch.manual_seed(5)
    with torch.no_grad():
        out["loss_dsm"] = Loss(cfg_d, sde)(x0, t.clamp(min=1e-3), net, eps=eps)
    out["dsm_m0_unit"] = m0_draw
    save("loss_tiny.npz", **out)

    # ------------------------------------------------------------------ I. two train steps
    import copy
    net = NCSNpp(cfg)
    load_synth(net, 1000)
    net.train()
    ema = copy.deepcopy(net)
    for p in ema.parameters():
        p.requires_grad = False
    oc = cfg.training.optimizer
    opt = torch.optim.Adam(net.parameters(), lr=oc.lr, betas=(oc.beta_1, oc.beta_2), eps=oc.eps,
                           weight_decay=oc.weight_decay)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: min(s / oc.warmup, 1.0))
    emacb = ref_callbacks.EMAWeightUpdate(tau=cfg.training.ema_decay)
    losses, gnorms = [], []
    out = {}
    for step in range(3):
        x0, eps, t = synth_inputs(4, 3, 16, seed=50 + step)
        out[f"x0_{step}"], out[f"eps_{step}"], out[f"t_{step}"] = x0, eps, t
        loss = crit(x0, t, net, eps=eps)
        opt.zero_grad()
        loss.backward()
        gnorms.append(torch.nn.utils.clip_grad_norm_(net.parameters(), oc.grad_clip).item())
        opt.step()
        sched.step()
        emacb.update_weights(net, ema)
        losses.append(loss.item())
    out["losses"] = np.array(losses)
    out["grad_norms"] = np.array(gnorms)
    out["keys"] = np.array(names)
    out["param_norms"] = np.array([pd_.detach().double().norm().item() for pd_ in net.parameters()])
    sd0 = synth_state_dict(keys_shapes(net), 1000)
    out["param_delta_norms"] = np.array([(p.detach() - sd0[k]).double().norm().item() for k, p in net.named_parameters()])
    out["ema_delta_norms"] = np.array([(p.detach() - sd0[k]).double().norm().item() for k, p in ema.named_parameters()])
    for k in sel[:4] + last:
        out["p:" + k] = dict(net.named_parameters())[k].detach()
        out["e:" + k] = dict(ema.named_parameters())[k].detach()
    save("train_tiny.npz", **out)

    # ------------------------------------------------------------------ H. EM sampler (tiny)
    print("EM sampler (tiny)")
    net = NCSNpp(cfg)
    load_synth(net, 1000)
    net.eval()
    seen_t = []

    def score_fn(u, tt):
        seen_t.append(tt.detach().clone())
        return net(u, tt)

    sampler = EM(cfg, sde, score_fn)
    out = {}
    for n_disc in (3, 10):
        for stride in ("uniform", "quadratic"):
            g = torch.Generator().manual_seed(60 + n_disc)
            batch = torch.cat([torch.randn(2, 3, 16, 16, generator=g),
                               torch.randn(2, 3, 16, 16, generator=g) * np.sqrt(sde.m)], dim=1)
            n = n_disc - 1
            t_final = sde.T - cfg.evaluation.eval_eps
            tsx = torch.linspace(0, t_final, n + 1, dtype=torch.float64)
            if stride == "quadratic":
                tsx = t_final * torch.flip(1 - (tsx / t_final) ** 2.0, dims=[0])
            noises = [torch.randn(2, 6, 16, 16, generator=g, dtype=torch.float64) for _ in range(n)]
            it = iter(noises)
            orig = torch.randn_like
            torch.randn_like = lambda x_, **kw: next(it).to(x_.dtype)
            seen_t.clear()
            try:
                xf = sampler.sample(batch, tsx, n, denoise=True, eps=cfg.evaluation.eval_eps)
            finally:
                torch.randn_like = orig
            tag = f"{n_disc}_{stride}"
            out[f"batch_{tag}"] = batch
            out[f"noise_{tag}"] = torch.stack(noises)
            out[f"x_{tag}"] = xf
            out[f"ts_{tag}"] = tsx
            out[f"seen_t_{tag}"] = torch.stack([s[0] for s in seen_t])
            assert xf.dtype == torch.float64
    save("em_tiny.npz", **out)

    # ------------------------------------------------------------------ J. SSCS sampler (8(f) rank 1)
    print("SSCS sampler (tiny)")
    SSCS = get("samplers", "sscs_sde")
    out = {}
    for cname, ccfg, seed in (("xm", C.tiny(), 1000), ("m", tiny3, 1020)):
        sde_c = PSLD(ccfg)
        net = NCSNpp(ccfg)
        load_synth(net, seed)
        net.eval()
        sampler = SSCS(ccfg, sde_c, net)
        for n_disc in (3, 6):
            g = torch.Generator().manual_seed(70 + n_disc)
            batch = torch.cat([torch.randn(2, 3, 16, 16, generator=g),
                               torch.randn(2, 3, 16, 16, generator=g) * np.sqrt(sde_c.m)], dim=1)
            n = n_disc - 1
            tsx = torch.linspace(0, sde_c.T - ccfg.evaluation.eval_eps, n + 1, dtype=torch.float64)
            noises = [torch.randn(2, 6, 16, 16, generator=g, dtype=torch.float64) for _ in range(2 * n + 1)]
            it = iter(noises)
            orig = torch.randn_like
            torch.randn_like = lambda x_, **kw: next(it).to(x_.dtype)
            try:
                xf = sampler.sample(batch, tsx, n, denoise=True, eps=ccfg.evaluation.eval_eps)
            finally:
                torch.randn_like = orig
            tag = f"{cname}_{n_disc}"
            out[f"batch_{tag}"], out[f"noise_{tag}"], out[f"x_{tag}"], out[f"ts_{tag}"] = batch, torch.stack(noises), xf, tsx
            assert xf.dtype == torch.float64
    save("sscs_tiny.npz", **out)

    # ------------------------------------------------------------------ K. writer / loader edges (8(f) rank 3)
    print("writer / loader edges")
    import tempfile
    from PIL import Image
    g = torch.Generator().manual_seed(90)
    pred = torch.randn(5, 6, 16, 16, generator=g, dtype=torch.float64) * 0.8
    pred[0, 0, 0, :4] = torch.tensor([-1.0, 1.0, 0.0, 0.999999])
    with tempfile.TemporaryDirectory() as td:
        samples, _ = torch.chunk(pred.cpu(), 2, dim=1)                     # callbacks.py:103-107
        util.save_as_images(samples, file_name=os.path.join(td, "o"), denorm=True)
        u8 = np.stack([np.asarray(Image.open(os.path.join(td, f"o_{i}.png"))) for i in range(5)])
    img = torch.randint(0, 256, (4, 32, 32, 3), generator=g, dtype=torch.uint8).numpy()
    if not hasattr(np, "float"):
        np.float = float                                                    # util.py:27 predates numpy 1.24
    tens = torch.stack([torch.tensor(util.data_scaler(im, norm=True)).permute(2, 0, 1).float() for im in img])
    tens01 = torch.stack([torch.tensor(util.data_scaler(im, norm=False)).permute(2, 0, 1).float() for im in img])
    save("edges.npz", pred=pred, u8=u8, img=img, tens=tens, tens01=tens01)

    # ------------------------------------------------------------------ L. VP-SDE baseline (8(f) rank 4)
    print("VP-SDE (tiny)")
    vcfg = C.tiny_vpsde()
    VP = get("sde", "vpsde")
    vsde = VP(vcfg)
    net = NCSNpp(vcfg)
    vks = load_synth(net, 4000)
    net.train()
    g = torch.Generator().manual_seed(91)
    x0 = torch.rand(4, 3, 16, 16, generator=g) * 2 - 1
    eps = torch.randn(4, 3, 16, 16, generator=g)
    t = torch.rand(4, generator=g, dtype=torch.float64) * (1 - 1e-5) + 1e-5
    out = {"x0": x0, "eps": eps, "t": t, "x_t": vsde.perturb_data(x0, t, noise=eps)}
    vloss = get("losses", "score_loss")(vcfg, vsde)(x0, t, net, eps=eps)
    vloss.backward()
    out["loss"] = vloss.detach()
    gn = {k: p.grad.norm().item() for k, p in net.named_parameters() if p.grad is not None}
    out["grad_norm_keys"], out["grad_norms"] = np.array(list(gn.keys())), np.array(list(gn.values()))
    names = [k for k, _ in net.named_parameters()]
    for k in (names[0], names[2], names[5], names[-1]):
        out["g:" + k] = dict(net.named_parameters())[k].grad
    net.eval()
    sampler = EM(vcfg, vsde, net)
    batch = torch.randn(2, 3, 16, 16, generator=g)
    n = 4
    tsx = torch.linspace(0, vsde.T - vcfg.evaluation.eval_eps, n + 1, dtype=torch.float64)
    noises = [torch.randn(2, 3, 16, 16, generator=g, dtype=torch.float64) for _ in range(n)]
    it = iter(noises)
    orig = torch.randn_like
    torch.randn_like = lambda x_, **kw: next(it).to(x_.dtype)
    try:
        xf = sampler.sample(batch, tsx, n, denoise=True, eps=vcfg.evaluation.eval_eps)
    finally:
        torch.randn_like = orig
    out.update(batch=batch, noise=torch.stack(noises), ts=tsx, x_em=xf)
    save("vpsde_tiny.npz", **out)
    with open(os.path.join(OUT, "vpsde_meta.json"), "w") as fh:
        json.dump({"seed": 4000, "keys": [[k, list(s)] for k, s in vks]}, fh)
    inpaint_section(get, C)
    clf_section(get, C)
    em_c10_section(get, C)
    vp_loss_section(get, C)
    predict_x_section(get, C)
    print("done")


if __name__ == "__main__":
    main()
