"""BASELINE.json's full-size configurations under -m gpu (VERDICT r01 "next" #1).

* configs[4] (EM sampler) on its OWN network: C10-SOTA NCSN++ against the vector the reference's
  EulerMaruyamaSampler produced (tests/golden/em_c10_sota.npz), then at the real per-GPU batch of 512 through
  properties that do not need the CPU: batch independence (rows [0:2] of the B=512 run equal the B=2 run to fp32
  rounding: M = 524 288 rows, every `row * stride` product in the kernels has to be 64-bit clean), finiteness and
  bitwise repeatability.
* configs[1] / configs[3] (train step at the real batch): a train-mode forward + backward at B=128 (C10-SOTA) and
  B=64 (CelebA-64).  Samples are independent (GroupNorm is per sample, no batch statistics), so the output rows equal
  the small-batch run / the reference golden, and the flat gradient of a sum-type loss equals the sum of the
  gradients of the batch's slices, each computed by a separate small-batch pass.

Tolerances: outputs 2e-5 (reference golden) / 5e-6 (batch independence, same kernels different tilings); gradient
sums 1e-5 rel-L2 (fp32 accumulation order over 131 072 pixels vs 64 partial sums).
"""
import numpy as np
import pytest
import torch

from tests.test_model_gpu import DEV, T, _build, rel_l2

pytestmark = pytest.mark.gpu


def _sampler(cfg, net, noise):
    from psld_amd.registry import get_module
    sde = get_module("sde", "psld")(cfg)
    seen = []

    def score_fn(u, tt):
        assert u.dtype == torch.float32 and tt.dtype == torch.float32
        seen.append(tt[0].item())
        return net(u, tt)

    sampler = get_module("samplers", "em_sde")(cfg, sde, score_fn)
    sampler.noise_fn = lambda i, x: noise[i]
    return sde, sampler, seen


@pytest.mark.parametrize("stride", ["uniform", "quadratic"])
def test_em_sampler_on_c10_sota_matches_reference(golden, stride):
    """configs[4], its own network, B=2: 3 predictor steps + the denoising step, float64 noise replayed, against
    the REFERENCE's output (main/samplers/sde.py:38-58 through main/models/wrapper.py:101-122)."""
    from psld_amd.registry import get_module
    net, cfg, _ = _build("c10_sota")
    g = golden("em_c10_sota.npz")
    noise = T(g[f"noise_{stride}"]).to(DEV)
    sde, sampler, seen = _sampler(cfg, net, noise)
    cfg.evaluation.n_discrete_steps = 4
    cfg.evaluation.stride_type = stride
    wr = get_module("pl_modules", "sde_wrapper")(cfg, sde, net, ema_score_fn=net, sampler_cls=None)
    ts = wr.sampling_times(DEV)
    np.testing.assert_allclose(ts.cpu().numpy(), g[f"ts_{stride}"], rtol=1e-15, atol=1e-16)
    x = sampler.sample(T(g[f"batch_{stride}"]).to(DEV), ts, wr.n_discrete_steps, denoise=True, eps=cfg.evaluation.eval_eps)
    assert x.dtype == torch.float64
    err = rel_l2(x, T(g[f"x_{stride}"]))
    print(f"EM on C10-SOTA ({stride}): rel-L2 vs reference = {err:.3e}")
    assert err < 1e-5
    np.testing.assert_allclose(np.array(seen, dtype=np.float32), g[f"seen_t_{stride}"], rtol=1.5e-7, atol=0)


def test_em_sampler_batch_512_is_batch_independent_and_repeatable(golden):
    """configs[4] at its real per-GPU batch: 2 predictor steps + denoise at B=512 on the C10-SOTA net.  Rows [0:2]
    (fed the golden batch and noise) must equal the B=2 run, rows [510:512] (a copy of the same two samples at the far
    end of the batch) must equal them too, everything is finite, and a second run is bitwise identical."""
    net, cfg, _ = _build("c10_sota")
    g = golden("em_c10_sota.npz")
    B = 512
    gen = torch.Generator(device=DEV).manual_seed(99)
    b2 = T(g["batch_uniform"]).to(DEV)
    n2 = T(g["noise_uniform"]).to(DEV)[:2]                       # [2 steps][2][6][32][32]
    batch = torch.randn(B, 6, 32, 32, device=DEV, generator=gen)
    batch[:, 3:] *= 0.5                                         # sqrt(m) = 0.5: momentum prior scale
    noise = torch.randn(2, B, 6, 32, 32, device=DEV, generator=gen, dtype=torch.float64)
    for lo in (0, B - 2):
        batch[lo:lo + 2] = b2
        noise[:, lo:lo + 2] = n2
    ts = T(g["ts_uniform"]).to(DEV)[:3]
    _, s2, _ = _sampler(cfg, net, n2)
    x2 = s2.sample(b2, ts, 2, denoise=True, eps=cfg.evaluation.eval_eps)
    _, s512, _ = _sampler(cfg, net, noise)
    xa = s512.sample(batch, ts, 2, denoise=True, eps=cfg.evaluation.eval_eps)
    xb = s512.sample(batch, ts, 2, denoise=True, eps=cfg.evaluation.eval_eps)
    assert xa.shape == (B, 6, 32, 32) and xa.dtype == torch.float64
    assert bool(torch.isfinite(xa).all())
    assert torch.equal(xa, xb), "B=512 sampling is not bitwise repeatable"
    e_lo, e_hi = rel_l2(xa[:2], x2), rel_l2(xa[B - 2:], x2)
    print(f"B=512 vs B=2: rows[0:2] {e_lo:.3e}, rows[510:512] {e_hi:.3e}")
    assert e_lo < 5e-6 and e_hi < 5e-6
    # the rows in between are different samples: they must differ from the planted ones
    assert rel_l2(xa[2:4], x2) > 0.1


@pytest.mark.parametrize("name,batch", [("c10_sota", 128), ("celeba64", 128)])
def test_train_forward_backward_at_full_batch_equals_its_slices(golden, name, batch):
    """configs[1] (B=128, C10-SOTA) and configs[3]'s per-GPU batch (CelebA-64, B=128): train mode, dropout 0.
    At these batch sizes the 32x32 / 16x16 convolutions run in Winograd form while the two-sample slices run the direct
    kernels, so (1)-(3) also pin the two forms against each other on full-size layers.  (1) output
    rows of the planted golden samples equal the reference golden; (2) the flat parameter gradient of
    sum(y * w) equals the sum of the gradients of the B/2 two-sample slices, each from its own pass; (3) the HSM loss
    at full batch equals the mean of the slice losses and its gradient the mean of theirs (perturb + loss kernels)."""
    from psld_amd.registry import get_module
    net, cfg, _ = _build(name, train=True)
    cfg.model.score_fn.dropout = 0.0
    net.sf.dropout = 0.0
    size = cfg.data.image_size
    g = golden(f"net_{name}.npz")
    gx, gt, gy = T(g["x"]).to(DEV), T(g["t"]).to(DEV), T(g["y"])
    nb = gx.shape[0]
    gen = torch.Generator(device=DEV).manual_seed(7)
    x = torch.randn(batch, 6, size, size, device=DEV, generator=gen)
    t = torch.rand(batch, device=DEV, generator=gen) * 0.98 + 0.01
    w = torch.randn(batch, 6, size, size, device=DEV, generator=gen)
    x[:nb], t[:nb] = gx, gt
    x[batch - nb:], t[batch - nb:] = gx, gt

    def run(xs, ts, ws):
        for p in net.parameters():
            p.grad = None
        net.mark_grads_stale()
        y = net(xs.contiguous(), ts.contiguous())
        (y * ws).sum().backward()
        return y.detach(), net.flat_grad().double().clone()

    y, gfull = run(x, t, w)
    assert bool(torch.isfinite(y).all()) and bool(torch.isfinite(gfull).all())
    e0, e1 = rel_l2(y[:nb], gy), rel_l2(y[batch - nb:], gy)
    print(f"{name} B={batch}: output rows vs reference golden {e0:.3e} (front) {e1:.3e} (back)")
    assert e0 < 2e-5 and e1 < 2e-5
    gsum = torch.zeros_like(gfull)
    worst_y = 0.0
    step = 2
    for lo in range(0, batch, step):
        ys, gs = run(x[lo:lo + step], t[lo:lo + step], w[lo:lo + step])
        gsum += gs
        worst_y = max(worst_y, rel_l2(y[lo:lo + step], ys))
    eg = ((gfull - gsum).norm() / gsum.norm()).item()
    print(f"{name} B={batch}: worst slice output {worst_y:.3e}; flat gradient vs sum of {batch // step} slices {eg:.3e}")
    assert worst_y < 5e-6
    assert eg < 1e-5

    # HSM criterion at the full batch: loss = mean over the batch -> mean of slice losses, gradient = mean of theirs
    sde = get_module("sde", "psld")(cfg)
    crit = get_module("losses", "psld_score_loss")(cfg, sde)
    x0 = torch.rand(batch, 3, size, size, device=DEV, generator=gen) * 2 - 1
    eps = torch.randn(batch, 6, size, size, device=DEV, generator=gen)
    tt = (torch.rand(batch, device=DEV, generator=gen, dtype=torch.float64) * (1 - 1e-5) + 1e-5)

    def run_loss(lo, hi):
        for p in net.parameters():
            p.grad = None
        net.mark_grads_stale()
        loss = crit(x0[lo:hi].contiguous(), tt[lo:hi].contiguous(), net, eps=eps[lo:hi].contiguous())
        loss.backward()
        return loss.item(), net.flat_grad().double().clone()

    lfull, gl = run_loss(0, batch)
    parts = 8                                              # 8 slices of B/8: bounds the time of this second sweep
    per = batch // parts
    lsum, gsum = 0.0, torch.zeros_like(gl)
    for i in range(parts):
        li, gi = run_loss(i * per, (i + 1) * per)
        lsum += li / parts
        gsum += gi / parts
    el = abs(lfull - lsum) / abs(lsum)
    eg2 = ((gl - gsum).norm() / gsum.norm()).item()
    print(f"{name} B={batch}: HSM loss {lfull:.6f} vs mean of slices {lsum:.6f} ({el:.2e}); gradient {eg2:.3e}")
    assert np.isfinite(lfull) and el < 2e-6 and eg2 < 1e-5


def test_headline_configuration_with_dropout_is_repeatable_and_linear_in_the_batch():
    """The exact bench configuration (configs[1]: C10-SOTA, B=128, dropout 0.15, default Winograd policy: the 32x32 / 16x16
    convolutions in Winograd form, GroupNorm partial sums from their epilogues, dropout inside GroupNorm_1's apply pass).
    The dropout mask is a function of (seed, element index), so with the same seed word (a) two passes are bitwise equal -
    output and all 97.6 M gradients - and (b) the images do not couple: the gradient of sum(y * w) equals the sum of the
    gradients of the two half-batch objectives (w zeroed on the other half), each from its own full-batch pass with the
    SAME masks.  (The same network at dropout 0 is pinned against the direct kernels and the reference golden above.)"""
    net, cfg, _ = _build("c10_sota", train=True)
    assert abs(net.sf.dropout - 0.15) < 1e-12 and cfg.model.score_fn.dropout == 0.15
    batch = 128
    gen = torch.Generator(device=DEV).manual_seed(11)
    x = torch.randn(batch, 6, 32, 32, device=DEV, generator=gen)
    t = torch.rand(batch, device=DEV, generator=gen) * 0.98 + 0.01
    w = torch.randn(batch, 6, 32, 32, device=DEV, generator=gen)
    seed_word = torch.tensor([0x1234567], dtype=torch.int64, device=DEV)

    def run(ws):
        for p in net.parameters():
            p.grad = None
        net.mark_grads_stale()
        net._dropout_seed_dev = seed_word            # the per-step seed word a captured training step supplies
        try:
            y = net(x, t)
            (y * ws).sum().backward()
        finally:
            net._dropout_seed_dev = None
        return y.detach().clone(), net.flat_grad().clone()

    y1, g1 = run(w)
    y2, g2 = run(w)
    assert bool(torch.isfinite(y1).all()) and bool(torch.isfinite(g1).all())
    assert torch.equal(y1, y2) and torch.equal(g1, g2)
    # with a gradient reducer attached the parked reductions - and the batched Dense_0 weight gradients - run once per
    # BUCKET instead of once per pass (a world of one: the buckets are formed, nothing is exchanged): bitwise the same
    from psld_amd.ddp import BucketReducer
    red = BucketReducer()
    net.set_reducer(red)
    try:
        y3, g3 = run(w)
    finally:
        net.set_reducer(None)
    assert len(red.launched) >= 6 and red.launched[-1][0] == 0
    assert torch.equal(y1, y3) and torch.equal(g1, g3)
    # dropout really is on: the eval forward differs
    net.eval()
    with torch.no_grad():
        ye = net(x, t)
    net.train()
    assert rel_l2(y1, ye) > 1e-2
    wa, wb = w.clone(), w.clone()
    wa[batch // 2:] = 0
    wb[:batch // 2] = 0
    ya, ga = run(wa)
    yb, gb = run(wb)
    assert torch.equal(ya, y1) and torch.equal(yb, y1)
    e = ((g1.double() - (ga.double() + gb.double())).norm() / g1.double().norm()).item()
    print(f"B=128 dropout 0.15: gradient vs sum of the two half-batch objectives {e:.3e}")
    assert e < 5e-6


def test_graph_captured_training_step_on_c10_sota_is_bitwise_the_eager_step():
    """configs[1]'s network at the reference's per-GPU batch of 16 (train_uncond_psld.sh:25-30): the hipGraph-captured
    step (SDEWrapper.enable_graphs: ~1400 launches incl. the weight-gradient side stream and the batched parameter-
    gradient reductions, whose pointer tables must be the ones of the eager warm-up steps) against the eager step on a
    twin, dropout 0.15 and EMA on: same losses, parameters, Adam state and EMA bitwise after 5 steps (2 eager, the
    capture + 2 replays)."""
    import copy
    import psld_amd
    from psld_amd import config as C
    from psld_amd.optim import EMAWeightUpdate
    from psld_amd.registry import get_module
    psld_amd.import_modules_into_registry()
    cfg = C.c10_sota()
    cfg.training.batch_size = 16
    torch.manual_seed(7)
    net_a = get_module("score_fn", "ncsnpp")(cfg).to(DEV).train()
    net_b = copy.deepcopy(net_a)
    sde = get_module("sde", "psld")(cfg)
    data = [torch.rand(16, 3, 32, 32, device=DEV, generator=torch.Generator(device=DEV).manual_seed(i)) * 2 - 1 for i in range(5)]
    runs = []
    for net, graphs in ((net_a, False), (net_b, True)):
        ema = copy.deepcopy(net)
        for p in ema.parameters():
            p.requires_grad = False
        crit = get_module("losses", "psld_score_loss")(cfg, sde)
        wr = get_module("pl_modules", "sde_wrapper")(cfg, sde, net, ema_score_fn=ema, criterion=crit)
        if graphs:
            wr.enable_graphs(True, warmup_steps=2)
        cb = EMAWeightUpdate(cfg.training.ema_decay)
        torch.manual_seed(21)
        losses = []
        for i in range(5):
            losses.append(wr.training_step(data[i], i).item())
            cb.on_train_batch_end(None, wr)
        opt = wr.optimizers()
        runs.append((losses, net.flatten_parameters().clone(), opt._m.clone(), opt._v.clone(), ema.flatten_parameters().clone()))
        if graphs:
            assert "graph" in next(iter(wr._graph_steps.values()))
    (la, pa, ma, va, ea), (lb, pb, mb, vb, eb) = runs
    assert la == lb
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb) and torch.equal(ea, eb)


def _cu_hog():
    """tests/helpers/libcuhog.so (built by __graft_entry__.build()): a kernel that holds `blocks` workgroups resident."""
    import ctypes
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "libcuhog.so")
    if not os.path.exists(path):
        import __graft_entry__
        __graft_entry__.build()
    hog = ctypes.CDLL(path)
    hog.cu_hog.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                           ctypes.c_void_p]
    hog.cu_hog.restype = ctypes.c_int
    return hog


def test_team_groupnorm_backward_beside_a_resident_foreign_kernel():
    """VERDICT r05 next #3b - the N = 8 hazard rehearsed on one GPU.  gn_bwd_team_kernel sizes its grid to what the device
    holds when it has the device to itself; RCCL's channel kernels will be resident beside it.  Here 32 workgroups of 512
    threads and 64 KB of LDS (RCCL's largest kernel shape) sit on a second stream for the whole of a C10-SOTA B=128 forward
    + backward (24 team launches): no team member may time out (error word 0) and the flat gradient must be bitwise the one
    of the undisturbed pass - a member that read a slot too early, or gave up, would change it."""
    from psld_amd import ops
    net, cfg, _ = _build("c10_sota", train=True)
    cfg.model.score_fn.dropout = 0.0
    net.sf.dropout = 0.0
    B = 128
    gen = torch.Generator(device=DEV).manual_seed(17)
    x = torch.randn(B, 6, 32, 32, device=DEV, generator=gen)
    t = torch.rand(B, device=DEV, generator=gen) * 0.98 + 0.01
    w = torch.randn(B, 6, 32, 32, device=DEV, generator=gen)
    assert ops.gn_bwd_team_wanted(B, 32 * 32, 256, None, True) > 0, "the team kernel does not take this shape any more"

    def run():
        net.mark_grads_stale()
        y = net(x, t)
        (y * w).sum().backward()
        return net.flat_grad()

    run()                                   # sizes workspaces, arenas and weight caches
    torch.cuda.synchronize()
    ref = run().clone()
    torch.cuda.synchronize()
    assert ops.gn_team_errors(DEV) == 0
    hog = _cu_hog()
    side = torch.cuda.Stream(device=DEV)
    scratch = torch.zeros(1 << 22, device=DEV)
    for rep in range(3):
        done_hog, done_pass = torch.cuda.Event(), torch.cuda.Event()
        torch.cuda.synchronize()
        rc = hog.cu_hog(scratch.data_ptr(), scratch.numel(), 32, 512, 65536, 400e3, side.cuda_stream)    # resident for 400 ms
        assert rc == 0, rc
        done_hog.record(side)
        got = run()
        done_pass.record(torch.cuda.current_stream(DEV))
        done_pass.synchronize()
        still_there = not done_hog.query()
        torch.cuda.synchronize()
        assert still_there, "the foreign kernel left before the pass ended: nothing was rehearsed"
        assert ops.gn_team_errors(DEV) == 0, "a team member timed out beside the resident kernel"
        assert torch.equal(got, ref), f"repetition {rep}: the gradient changed beside a resident foreign kernel"
    ops.check_device_errors(DEV)
