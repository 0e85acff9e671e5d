"""Euler-Maruyama reverse-SDE sampler with the reference's interface
(main/samplers/base.py:4-31, main/samplers/sde.py:9-58).

``EulerMaruyamaSampler(config, sde, score_fn, corrector_fn=None).sample(batch, ts,
n_discrete_steps, denoise=True, eps=1e-3)`` -> float64 ``[B, 2C, H, W]``.  Per step: ONE network
call + ONE fused kernel (drift, score transform, Euler update, noise, f32 copy for the next call);
the reference issues ~30 eager ops per step (psld.py:330-364, sde.py:16-26).  Noise comes from
``torch.randn_like`` on the float64 state like the reference, so a seeded run draws the same
stream the reference would draw on this device.
"""
from __future__ import annotations

import abc

import numpy as np
import torch

from . import ops
from .registry import register_module


class Sampler(abc.ABC):
    def __init__(self, config, sde, score_fn, corrector_fn=None):
        super().__init__()
        self.config = config
        self.sde = sde
        self.score_fn = score_fn
        self.corrector_fn = corrector_fn

    @property
    def n_steps(self):
        return self.config.evaluation.n_discrete_steps

    @abc.abstractmethod
    def predictor_update_fn(self):
        raise NotImplementedError

    def corrector_update_fn(self, x, t, dt):
        if self.corrector_fn is not None:
            return self.corrector_fn(x, t, dt)
        return x, x   # base.py:27-28: identity

    @abc.abstractmethod
    def sample(self):
        raise NotImplementedError


@register_module(category="samplers", name="em_sde")
class EulerMaruyamaSampler(Sampler):
    def __init__(self, config, sde, score_fn, corrector_fn=None):
        super().__init__(config, sde, score_fn, corrector_fn=corrector_fn)
        self.noise_fn = None  # test hook: callable(step, x) -> float64 noise replacing torch.randn_like

    def _step(self, x64, x32, t: float, dt: float, noise):
        """One predictor update in place on x64 (sde.py:16-26); returns nothing."""
        sde = self.sde
        t_rev = sde.T - t                                            # psld.py:348
        t32 = torch.full((x64.shape[0],), float(np.float32(t_rev)), device=x64.device, dtype=torch.float32)
        eps_pred = self.score_fn(x32, t32)                           # psld.py:354
        sde.em_update(x64, eps_pred.contiguous(), noise, t_rev, dt, x32)   # PSLD or VP-SDE fused update

    def predictor_update_fn(self, x, t, dt):
        """Reference-shaped entry: returns (x, x_mean) for a float64/float32 state ``x``."""
        tt = float(t) if not torch.is_tensor(t) else float(t.item())
        dd = float(dt) if not torch.is_tensor(dt) else float(dt.reshape(-1)[0].item())
        x64 = ops.f32_to_f64(x.contiguous()) if x.dtype == torch.float32 else x.contiguous().clone()
        x32 = ops.f64_to_f32(x64)
        mean = x64.clone()
        self._step(mean, x32.clone(), tt, dd, None)
        self._step(x64, x32, tt, dd, torch.randn_like(x64))
        return x64, mean

    def denoising_fn(self, x, t, dt):
        tt = float(t) if not torch.is_tensor(t) else float(t.item())
        dd = float(dt) if not torch.is_tensor(dt) else float(dt.reshape(-1)[0].item())
        x64 = ops.f32_to_f64(x.contiguous()) if x.dtype == torch.float32 else x.contiguous().clone()
        self._step(x64, ops.f64_to_f32(x64), tt, dd, None)
        return x64

    def sample(self, batch, ts, n_discrete_steps, denoise=True, eps=1e-3):
        if not batch.is_cuda:
            raise RuntimeError("psld_amd sampler needs device tensors (no CPU fallback)")
        self.nfe = n_discrete_steps
        tl = ts.detach().to(torch.float64).cpu().tolist()           # one host read per sample() call
        x32 = batch.to(torch.float32).contiguous().clone()
        x64 = ops.f32_to_f64(x32) if batch.dtype != torch.float64 else batch.contiguous().clone()
        with torch.no_grad():
            for i in range(n_discrete_steps):
                dt = tl[i + 1] - tl[i]                               # sde.py:45
                z = self.noise_fn(i, x64) if self.noise_fn is not None else torch.randn_like(x64)
                self._step(x64, x32, tl[i], dt, z.contiguous())
                if self.corrector_fn is not None:                    # sde.py:49-50
                    x64, _ = self.corrector_update_fn(x64, ts[i], dt)
                    x64 = x64.contiguous()
                    x32 = ops.f64_to_f32(x64)
            if denoise:
                # sde.py:52-57: torch.tensor(T - eps) and torch.tensor(eps) are float32 0-d tensors
                t_d = float(np.float32(self.sde.T - eps))
                dt_d = float(np.float32(eps))
                self._step(x64, x32, t_d, dt_d, None)
        return x64


@register_module(category="samplers", name="cc_em_sde")
class ClassCondEulerMaruyamaSampler(EulerMaruyamaSampler):
    """Class-conditional EM sampling with classifier guidance (samplers/sde.py:62-114):
    ``f <- f + g^2 * clf_temp * grad_x log p(y | x_t)`` before the Euler step.  ``config`` is the root node with
    ``.diffusion`` and ``.clf``; ``clf_fn`` is an ``NCSNppClassifier``.  Per step: score-network forward, classifier
    forward + backward to its input (same executor and kernels as the score network), the fused EM kernel and one
    guidance kernel.  Like the reference, the classifier is evaluated at the SAMPLER time t (as float32), not T - t."""

    def __init__(self, config, sde, score_fn, clf_fn, corrector_fn=None):
        super().__init__(config, sde, score_fn, corrector_fn=corrector_fn)
        self.clf_fn = clf_fn
        self.y = self.config.clf.evaluation.label_to_sample
        self.clf_temp = self.config.clf.evaluation.clf_temp

    def _labels(self, b, device):
        y = self.y
        if torch.is_tensor(y) and y.dim() > 0:
            return y.to(device=device, dtype=torch.int64).contiguous()
        return torch.full((b,), int(y), device=device, dtype=torch.int64)

    def _guidance(self, x32, t: float):
        """d/dx sum_r log softmax(clf(x, t))_r[y_r] (sde.py:84-92), float32, without the temperature."""
        x_in = x32.detach().clone().requires_grad_()
        t32 = torch.full((x32.shape[0],), float(np.float32(t)), device=x32.device, dtype=torch.float32)
        with torch.enable_grad():
            logits = self.clf_fn(x_in, t32)
            # d log_softmax(z)[y] / dz = onehot(y) - softmax(z)
            _, dlogits, _ = ops.softmax_xent(logits.detach().contiguous(), self._labels(x32.shape[0], x32.device), 1.0, -1.0)
            (grad,) = torch.autograd.grad(logits, x_in, grad_outputs=dlogits)
        return grad.contiguous()

    def _step(self, x64, x32, t: float, dt: float, noise):
        grad = self._guidance(x32, t)                                  # at the state BEFORE the update
        with torch.no_grad():
            super()._step(x64, x32, t, dt, noise)
        sde = self.sde
        beta = float(sde.beta_t(sde.T - t))
        gx2 = float(np.sqrt(beta * sde.gamma)) ** 2                    # psld.py:339-340, squared at sde.py:94
        gm2 = float(np.sqrt(beta * sde.m * sde.nu)) ** 2
        ops.guide(x64, grad, gx2 * self.clf_temp * dt, gm2 * self.clf_temp * dt, x32)

    def sample(self, batch, ts, n_discrete_steps, denoise=True, eps=1e-3):
        if not batch.is_cuda:
            raise RuntimeError("psld_amd sampler needs device tensors (no CPU fallback)")
        self.nfe = n_discrete_steps
        tl = ts.detach().to(torch.float64).cpu().tolist()
        x32 = batch.to(torch.float32).contiguous().clone()
        x64 = ops.f32_to_f64(x32) if batch.dtype != torch.float64 else batch.contiguous().clone()
        for i in range(n_discrete_steps):
            z = self.noise_fn(i, x64) if self.noise_fn is not None else torch.randn_like(x64)
            self._step(x64, x32, tl[i], tl[i + 1] - tl[i], z.contiguous())
            if self.corrector_fn is not None:
                x64, _ = self.corrector_update_fn(x64, ts[i], tl[i + 1] - tl[i])
                x64 = x64.contiguous()
                x32 = ops.f64_to_f32(x64)
        if denoise:
            # sde.py:107-113: the predictor draws its noise here too; the result is x_mean
            if self.noise_fn is not None:
                self.noise_fn(n_discrete_steps, x64)
            else:
                torch.randn_like(x64)
            self._step(x64, x32, float(np.float32(self.sde.T - eps)), float(np.float32(eps)), None)
        return x64


@register_module(category="samplers", name="ip_em_sde")
class ES3EulerMaruyamaInpainter(EulerMaruyamaSampler):
    """Inpainting with the EM sampler (samplers/sde.py:117-224): after every predictor update the
    known image is perturbed to the current reverse time and written over the masked-in pixels of both state halves.
    ``sample((x_0, mask), ts, n)``: ``x_0`` f32 [B,C,H,W], ``mask`` [B,C,H,W] of {0,1} (1 = known pixel).  Per step:
    one network call, the fused EM kernel, the fused perturbation kernel and one combine kernel.  Random draws are
    made in the reference's order (prior x, prior m; then per update: predictor noise, momentum draw - discarded in
    HSM mode like the reference does -, perturbation noise), so a seeded run consumes the stream the same way."""

    def __init__(self, config, sde, score_fn, corrector_fn=None):
        super().__init__(config, sde, score_fn, corrector_fn=corrector_fn)
        self.draw_fn = None   # test hook: callable(shape, dtype, device) -> tensor replacing torch.randn

    def _draw(self, shape, dtype, device):
        if self.draw_fn is not None:
            return self.draw_fn(tuple(shape), dtype, device)
        return torch.randn(*shape, dtype=dtype, device=device)

    def _perturb(self, x_0, t_rev: float):
        """sde.py:127-142: (u_t, mu_t) f64 of the known image at forward time ``t_rev``."""
        sde = self.sde
        m_0 = float(np.sqrt(sde.mm_0)) * self._draw(x_0.shape, x_0.dtype, x_0.device)
        mm_0 = 0.0
        if self.config.training.mode == "hsm":
            m_0, mm_0 = None, sde.mm_0                                # zeros: the kernel takes NULL
        eps = self._draw((x_0.shape[0], 2 * x_0.shape[1], *x_0.shape[2:]), x_0.dtype, x_0.device)
        t = torch.full((x_0.shape[0],), t_rev, device=x_0.device, dtype=torch.float64)
        u, mu, _ = sde.perturb_data(x_0, m_0, 0, mm_0, t, eps=eps)
        return u, mu

    def sample(self, batch, ts, n_discrete_steps, denoise=True, eps=1e-3):
        x_0, mask = batch
        if not x_0.is_cuda:
            raise RuntimeError("psld_amd sampler needs device tensors (no CPU fallback)")
        self.nfe = n_discrete_steps
        sde, dev = self.sde, x_0.device
        x_0 = x_0.to(torch.float32).contiguous()
        maskf = mask.to(device=dev, dtype=torch.float32).contiguous()
        tl = ts.detach().to(torch.float64).cpu().tolist()
        px = self._draw(x_0.shape, torch.float32, dev)               # psld.py:366-370
        pm = self._draw(x_0.shape, torch.float32, dev) * float(np.sqrt(sde.m))
        x32 = torch.cat([px, pm], dim=1).contiguous()
        x64 = ops.f32_to_f64(x32)
        u_k, _ = self._perturb(x_0, float(sde.T))                    # sde.py:195-203
        ops.mask_combine(x64, u_k, maskf, x32)
        with torch.no_grad():
            for i in range(n_discrete_steps):
                dt = tl[i + 1] - tl[i]
                z = self._draw(x64.shape, torch.float64, dev)
                self._step(x64, x32, tl[i], dt, z.contiguous())
                u_k, _ = self._perturb(x_0, sde.T - tl[i])          # sde.py:166-169
                ops.mask_combine(x64, u_k, maskf, x32)
            if denoise:
                # sde.py:213-221: torch.tensor(T - eps) / torch.tensor(eps) are float32 0-d tensors; the predictor's
                # noise is drawn and discarded, the result is the combined x_mean
                t_d = float(np.float32(sde.T - eps))
                self._draw(x64.shape, torch.float64, dev)
                self._step(x64, x32, t_d, float(np.float32(eps)), None)
                _, mu_k = self._perturb(x_0, float(np.float32(sde.T) - np.float32(t_d)))
                ops.mask_combine(x64, mu_k, maskf, None)
        return x64


@register_module(category="samplers", name="sscs_sde")
class SSCSSampler(Sampler):
    """Symmetric-splitting CLD sampler (samplers/sde.py:227-370): per step two analytic Ornstein-
    Uhlenbeck half steps (closed-form mean matrix + Cholesky noise, one fused kernel each) around
    one Euler score step (one network call + one fused kernel).  Same surface and noise-draw order
    as the reference (two ``randn_like`` per step, one unused draw in the denoising step)."""

    def __init__(self, config, sde, score_fn, corrector_fn=None):
        super().__init__(config, sde, score_fn, corrector_fn=corrector_fn)
        self.noise_fn = None   # test hook: callable(draw_index, x) -> float64 noise

    # host-side scalars (python doubles) ----------------------------------------------------------
    def _analytic_coeffs(self, t: float, dt: float):
        import math
        from ._lib import SscsCoeffs
        sde = self.sde
        db = sde.b_t(sde.T - (t + dt)) - sde.b_t(sde.T - t)              # sde.py:241
        nu, ga, m, mi = sde.nu, sde.gamma, sde.m, sde.m_inv
        sf = math.exp(((nu + ga) / 4) * db)                               # :245-246
        a1, a2, c1, c2 = (nu - ga) / 4, -((ga - nu) ** 2) / 8, 0.5, (ga - nu) / 4
        k = SscsCoeffs()
        k.a_xx, k.a_xm = (1 - a1 * db) * sf, (-a2 * db) * sf              # :255
        k.a_mx, k.a_mm = (-c1 * db) * sf, (1 - c2 * db) * sf              # :257
        db2 = db * db
        lam = (nu + ga) / 2
        sc, isc = math.exp(lam * db), math.exp(-lam * db)                 # :271-272
        xx = ((-mi / 2) * db2 - ((ga - nu) / 2) * db + (isc - 1)) * sc + sde.eps
        xm = (((ga - nu) / 4) * db2) * sc
        mm = ((-1 / 2) * db2 - (m * (nu - ga) / 2) * db + m * (isc - 1)) * sc + sde.eps
        try:                                                              # psld.py:154-186
            if sde.decomp_mode == "lower":
                l11 = math.sqrt(xx)
                l21 = xm / l11
                k.c11, k.c12, k.c21, k.c22 = l11, 0.0, l21, math.sqrt(mm - l21 ** 2)
            else:
                u22 = math.sqrt(mm)
                u12 = xm / u22
                k.c11, k.c12, k.c21, k.c22 = math.sqrt(xx - u12 ** 2), u12, 0.0, u22
        except ValueError:
            raise ValueError("Numerical precision error.")
        return k

    def predictor_update_fn(self, u, t, dt):
        raise NotImplementedError("use sample(); the fused kernels update the state in place")

    def sample(self, batch, ts, n_discrete_steps, denoise=True, eps=1e-3):
        if not batch.is_cuda:
            raise RuntimeError("psld_amd sampler needs device tensors (no CPU fallback)")
        sde = self.sde
        self.nfe = n_discrete_steps
        tl = ts.detach().to(torch.float64).cpu().tolist()
        x32 = batch.to(torch.float32).contiguous().clone()
        x64 = ops.f32_to_f64(x32) if batch.dtype != torch.float64 else batch.contiguous().clone()
        draws = [0]

        def noise():
            i = draws[0]
            draws[0] += 1
            z = self.noise_fn(i, x64) if self.noise_fn is not None else torch.randn_like(x64)
            return z.contiguous()

        with torch.no_grad():
            for i in range(n_discrete_steps):
                t, dt = tl[i], tl[i + 1] - tl[i]
                ka = self._analytic_coeffs(t, dt / 2)
                ops.sscs_analytic(x64, noise(), ka, x32)                  # sde.py:333
                t_rev = sde.T - t
                t32 = torch.full((x64.shape[0],), float(np.float32(t_rev)), device=x64.device, dtype=torch.float32)
                eps_pred = self.score_fn(x32, t32)
                ops.sscs_score_step(x64, eps_pred.contiguous(), sde.em_coeffs(t_rev, dt))   # :334
                ops.sscs_analytic(x64, noise(), ka, x32)                  # :335
                if self.corrector_fn is not None:
                    x64, _ = self.corrector_update_fn(x64, ts[i], dt)
                    x64 = x64.contiguous()
                    x32 = ops.f64_to_f32(x64)
            if denoise:                                                   # sde.py:363-368, :340-351
                t_d = float(np.float32(sde.T - eps))
                dt_d = float(np.float32(eps))
                t_rev = sde.T - t_d
                t32 = torch.full((x64.shape[0],), float(np.float32(t_rev)), device=x64.device, dtype=torch.float32)
                eps_pred = self.score_fn(x32, t32)
                noise()                                                    # drawn and discarded by the reference
                ops.em_step(x64, eps_pred.contiguous(), None, sde.em_coeffs(t_rev, dt_d), x32)
        return x64


@register_module(category="samplers", name="bb_ode")
class BBODESampler(Sampler):
    """Black-box probability-flow ODE sampler (samplers/ode.py:9-76) with the adaptive RK45 on the device.

    The reference hands the ODE to ``torchdiffeq.odeint(method="scipy_solver")``, i.e. scipy's
    ``solve_ivp(RK45)`` on the host: every function evaluation copies the state device -> numpy ->
    device.  Here the Dormand-Prince stages, the error norm and the step update are fused kernels on
    float64 device buffers; only the scalar error norm comes back per step.  The step controller follows
    scipy's RK45 (safety 0.9, factors [0.2, 10], `select_initial_step`), and like torchdiffeq's bridge
    the RHS sees the state and the time rounded to float32."""

    C = (0.0, 1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0)
    A = ((), (1 / 5,), (3 / 40, 9 / 40), (44 / 45, -56 / 15, 32 / 9),
         (19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729),
         (9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656))
    B = (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84)
    E = (-71 / 57600, 0.0, 71 / 16695, -71 / 1920, 17253 / 339200, -22 / 525, 1 / 40)

    def __init__(self, config, sde, score_fn, corrector_fn=None):
        super().__init__(config, sde, score_fn, corrector_fn=corrector_fn)
        self.nfe = 0
        self.rtol = config.evaluation.sampler.rtol
        self.atol = config.evaluation.sampler.atol
        self.solver_opts = {"solver": config.evaluation.sampler.solver}
        if self.solver_opts["solver"] != "RK45":
            raise NotImplementedError("device-side BB-ODE implements scipy's RK45 (the solver every script uses)")
        self._counter = 0

    @property
    def n_steps(self):
        return self.nfe

    @property
    def mean_nfe(self):
        if self._counter != 0:
            return self.nfe / self._counter
        raise ValueError("Run .sample() to compute mean_nfe")

    def predictor_update_fn(self, x, t, dt):
        pass

    def _rhs(self, t: float, y64, y32, shape):
        """f(t, y) = reverse_sde(..., probability_flow=True)[0] on float32-rounded (t, y) (ode.py:41-45)."""
        self.nfe += 1
        t32 = float(np.float32(t))
        f, _ = self.sde.reverse_sde(y32.view(shape), t32, self.score_fn, probability_flow=True)
        return f.view(-1)

    def sample(self, batch, ts=None, n_discrete_steps=None, denoise=True, eps=1e-3):
        if not batch.is_cuda:
            raise RuntimeError("psld_amd sampler needs device tensors (no CPU fallback)")
        import math
        sde, rtol, atol = self.sde, float(self.rtol), float(self.atol)
        self._counter += 1
        shape = batch.shape
        dev = batch.device
        y32 = batch.to(torch.float32).contiguous().clone().view(-1)
        y = ops.f32_to_f64(y32)
        n = y.numel()
        t, t_bound = 0.0, sde.T - eps
        acc = torch.zeros(1, dtype=torch.float64, device=dev)

        def norm(vs, cs, p, q):
            ops.scaled_norm_sq(vs, cs, p, q, atol, rtol, acc)
            return math.sqrt(float(acc.item()) / n)

        with torch.no_grad():
            f = self._rhs(t, y, y32, shape)
            # scipy select_initial_step (order 4)
            d0 = norm([y], [1.0], y, y)
            d1 = norm([f], [1.0], y, y)
            h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
            h0 = min(h0, t_bound - t)
            y1 = torch.empty_like(y)
            y1_32 = torch.empty_like(y32)
            ops.lincomb(y1, y, [f], [h0], y1_32)
            f1 = self._rhs(t + h0, y1, y1_32, shape)
            d2 = norm([f1, f], [1.0, -1.0], y, y) / h0
            h1 = max(1e-6, h0 * 1e-3) if (d1 <= 1e-15 and d2 <= 1e-15) else (0.01 / max(d1, d2)) ** (1 / 5)
            h_abs = min(100 * h0, h1, t_bound - t)
            K = [f] + [None] * 6
            ys, ys32 = torch.empty_like(y), torch.empty_like(y32)
            y_new, y_new32 = torch.empty_like(y), torch.empty_like(y32)
            while t < t_bound:
                min_step = 10 * abs(np.nextafter(t, np.inf) - t)
                h_abs = max(h_abs, min_step)
                rejected = False
                while True:
                    if h_abs < min_step:
                        raise RuntimeError("BB-ODE: required step size is less than spacing between numbers")
                    t_new = t + h_abs
                    if t_new - t_bound > 0:
                        t_new = t_bound
                    h = t_new - t
                    h_abs = abs(h)
                    for s in range(1, 6):                                     # scipy rk_step
                        ops.lincomb(ys, y, K[:s], [a * h for a in self.A[s]], ys32)
                        K[s] = self._rhs(t + self.C[s] * h, ys, ys32, shape)
                    ops.lincomb(y_new, y, K[:6], [b * h for b in self.B], y_new32)
                    K[6] = self._rhs(t + h, y_new, y_new32, shape)
                    err = norm(K, [e * h for e in self.E], y, y_new)
                    if err < 1:
                        factor = 10.0 if err == 0 else min(10.0, 0.9 * err ** -0.2)
                        if rejected:
                            factor = min(1.0, factor)
                        h_abs *= factor
                        break
                    h_abs *= max(0.2, 0.9 * err ** -0.2)
                    rejected = True
                t = t_new
                y, y_new = y_new, y
                y32, y_new32 = y_new32, y32
                K[0] = K[6]
            # torchdiffeq casts the solution back to y0's dtype (float32); ode.py:64
            x32 = y32.view(shape)
            if denoise:                                                       # ode.py:66-75
                f, _ = sde.reverse_sde(x32, sde.T - eps, self.score_fn, probability_flow=True)
                self.nfe += 1
                x = ops.f32_to_f64(x32.contiguous())
                ops.lincomb(x.view(-1), x.view(-1), [f.view(-1)], [eps])
                return x
            return x32
