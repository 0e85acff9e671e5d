"""PSLD SDE with the reference's interface (main/models/sde/psld.py:12-377, base.py:5-64),
computed by libpsld_hip kernels on the device.

Same constructor (``PSLD(config)``), same attributes (``beta_0, beta_1, nu, gamma, m_inv, m,
kappa, mm_0, eps, decomp_mode, T, mode, type``) and methods (``perturb_data, sde, reverse_sde,
get_score, prior_sampling, cond_marginal_prob, _mean, _cov, get_coeff, get_inv_coeff``), same
dtypes at the boundary (f64 ``u_t`` / ``f_bar``; f32 score), NaN coefficients raise
``ValueError("Numerical precision error.")`` like psld.py:166-171.
"""
from __future__ import annotations

import math
from typing import Callable

import numpy as np
import torch

from . import ops
from ._lib import EmCoeffs, SdeParams
from .registry import register_module

Tensor = torch.Tensor


@register_module(category="sde", name="psld")
class PSLD:
    def __init__(self, config):
        s = config.model.sde
        self.N = s.n_timesteps
        self.beta_0 = s.beta_min
        self.beta_1 = s.beta_max
        self.nu = s.nu
        self.gamma = s.gamma
        assert self.nu != 0 or self.gamma != 0
        self.m_inv = (self.gamma - self.nu) ** 2 / 4
        self.m = 1 / self.m_inv
        self.kappa = s.kappa
        self.mm_0 = self.kappa * self.m
        self.eps = s.numerical_eps
        self.decomp_mode = s.decomp_mode
        assert self.decomp_mode in ["lower", "upper"]
        self.check_nan = True  # host sync per call, like the reference's torch.sum(torch.isnan(..)) > 0
        p = SdeParams()
        p.beta_0, p.beta_1, p.nu, p.gamma = float(self.beta_0), float(self.beta_1), float(self.nu), float(self.gamma)
        p.m_inv, p.numerical_eps = float(self.m_inv), float(self.eps)
        p.decomp_lower = 1 if self.decomp_mode == "lower" else 0
        self._params = p

    def __repr__(self):
        return (f"Initialized SDE with m_inv:{self.m_inv}, gamma: {self.gamma}, nu: {self.nu}, "
                f"Decomp mode: {self.decomp_mode}")

    # psld.py:38-60
    def beta_t(self, t):
        return self.beta_0 + t * (self.beta_1 - self.beta_0)

    def b_t(self, t):
        return self.beta_0 * t + 0.5 * (t ** 2) * (self.beta_1 - self.beta_0)

    @property
    def T(self):
        return 1.0

    @property
    def mode(self):
        if self.gamma == 0:
            return "score_m"
        elif self.nu == 0:
            return "score_x"
        return "score_xm"

    @property
    def type(self):
        return f"psld-{self.mode}"

    # ---- per-sample scalars on the device ---------------------------------------------------------
    def prefetch_coeffs(self, t: Tensor, xx_0, mm_0) -> None:
        """Compute (and NaN-check) the coefficient table of ``t`` now, on the CURRENT stream, and keep it for the next
        ``_coeff_table`` call with this very tensor and the same initial variances.  The training step calls this on
        a stream of its own that does not wait for the compute stream: the host read of the NaN flag then waits for
        two tiny kernels instead of for the previous step's whole queue."""
        self._prefetched = None
        tb = self._coeff_table(t, xx_0, mm_0)
        self._prefetched = (t, float(xx_0), float(mm_0), tb)

    def _coeff_table(self, t: Tensor, xx_0, mm_0) -> Tensor:
        if not t.is_cuda:
            raise RuntimeError("psld_amd.PSLD needs device tensors (no CPU fallback)")
        pf = getattr(self, "_prefetched", None)
        if pf is not None:
            self._prefetched = None
            if pf[0] is t and pf[1] == float(xx_0) and pf[2] == float(mm_0):
                pf[3].record_stream(torch.cuda.current_stream(t.device))
                return pf[3]
        t = t.to(torch.float64).contiguous()
        flag = torch.zeros(1, dtype=torch.int32, device=t.device)
        table = ops.perturb_coeffs(t, self._params, float(xx_0), float(mm_0), flag)
        if self.check_nan and int(flag.item()) != 0:
            raise ValueError("Numerical precision error.")
        return table

    def _cov(self, xx_0, mm_0, t):
        tb = self._coeff_table(t, xx_0, mm_0)
        return tb[:, 8].contiguous(), tb[:, 9].contiguous(), tb[:, 10].contiguous()

    def get_coeff(self, var):
        xx, xm, mm = var
        if self.decomp_mode == "lower":
            l11 = torch.sqrt(xx)
            l21 = xm / l11
            out = (l11, torch.zeros_like(xx), l21, torch.sqrt(mm - l21 ** 2.0))
        else:
            u22 = torch.sqrt(mm)
            u12 = xm / u22
            out = (torch.sqrt(xx - u12 ** 2.0), u12, torch.zeros_like(mm), u22)
        if any(bool(torch.isnan(c).any()) for c in out):
            raise ValueError("Numerical precision error.")
        return out

    def get_inv_coeff(self, var):
        xx, xm, mm = var
        det = xx * mm - xm ** 2
        if self.decomp_mode == "lower":
            out = (torch.sqrt(1 / xx), -xm / (torch.sqrt(xx) * torch.sqrt(det)), torch.zeros_like(xx),
                   torch.sqrt(xx / det))
        else:
            out = (torch.sqrt(mm / det), torch.zeros_like(mm), -xm / (torch.sqrt(mm) * torch.sqrt(det)),
                   torch.sqrt(1 / mm))
        if any(bool(torch.isnan(c).any()) for c in out):
            raise ValueError("Numerical precision error.")
        return out

    # ---- host-side scalar restatement (sampler: one t per step, no device round trip) ---------------
    def _cov_host(self, xx_0: float, mm_0: float, t: float):
        """psld.py:86-152 in python floats (IEEE double)."""
        nu, ga, mi, m = self.nu, self.gamma, self.m_inv, self.m
        lam = (nu + ga) / 2
        b = self.b_t(t)
        b2 = b ** 2
        sc, isc = math.exp(-lam * b), math.exp(lam * b)
        xx = (mi / 4 * b2 * xx_0 + mi ** 2 / 4 * b2 * mm_0 + (nu - ga) / 2 * b * xx_0 + (-mi / 2) * b2
              + (ga - nu) / 2 * b + (isc - 1) + xx_0) * sc
        xm = ((ga - nu) / 8 * b2 * xx_0 + mi * (ga - nu) / 8 * b2 * mm_0 + (-1 / 2) * b * xx_0 + mi / 2 * b * mm_0
              + (nu - ga) / 4 * b2) * sc
        mm = (1 / 4 * b2 * xx_0 + mi / 4 * b2 * mm_0 + (ga - nu) / 2 * b * mm_0 + (-1 / 2) * b2
              + m * (nu - ga) / 2 * b + m * (isc - 1) + mm_0) * sc
        return xx + self.eps, xm, mm + self.eps

    def _inv_coeff_host(self, var):
        """psld.py:188-220 in python floats; NaN -> ValueError."""
        xx, xm, mm = var
        det = xx * mm - xm ** 2
        try:
            if self.decomp_mode == "lower":
                out = (math.sqrt(1 / xx), -xm / (math.sqrt(xx) * math.sqrt(det)), 0.0, math.sqrt(xx / det))
            else:
                out = (math.sqrt(mm / det), 0.0, -xm / (math.sqrt(mm) * math.sqrt(det)), math.sqrt(1 / mm))
        except (ValueError, ZeroDivisionError):
            raise ValueError("Numerical precision error.")
        if any(math.isnan(c) for c in out):
            raise ValueError("Numerical precision error.")
        return out

    def em_coeffs(self, t_rev: float, dt: float, probability_flow: bool = False) -> EmCoeffs:
        """Scalars of one reverse-SDE evaluation at (already reversed) time ``t_rev = T - t``."""
        c11, c12, c21, c22 = self._inv_coeff_host(self._cov_host(0.0, self.mm_0, t_rev))
        k = EmCoeffs()
        k.beta = float(self.beta_t(t_rev))
        k.m_inv, k.gamma, k.nu, k.m = float(self.m_inv), float(self.gamma), float(self.nu), float(self.m)
        # psld.py:253-258: coefficients are cast to float32 before they touch eps
        k.c11, k.c12, k.c21, k.c22 = (float(np.float32(c)) for c in (c11, c12, c21, c22))
        k.dt = float(dt)
        if self.decomp_mode == "lower" and self.mode == "score_m":
            k.score_mode = 1
        elif self.decomp_mode == "upper" and self.mode == "score_x":
            k.score_mode = 2
        else:
            k.score_mode = 0
        k.probability_flow = 1 if probability_flow else 0
        return k

    # ---- perturbation kernel (psld.py:62-84, 222-228, 262-287) ------------------------------------------
    def _mean(self, x_0, m_0, t):
        return self.perturb_data(x_0, m_0, 0.0, 0.0, t, eps=torch.zeros(
            x_0.shape[0], 2 * x_0.shape[1], *x_0.shape[2:], device=x_0.device))[1]

    def cond_marginal_prob(self, x_0, m_0, xx_0, mm_0, t):
        tb = self._coeff_table(t, xx_0, mm_0)
        eps0 = torch.zeros(x_0.shape[0], 2 * x_0.shape[1], *x_0.shape[2:], device=x_0.device)
        _, _, mu = ops.perturb(x_0.contiguous(), None if m_0 is None else m_0.contiguous(), eps0, tb, self._params,
                               want_f32=False, want_mu=True)
        return mu, (tb[:, 8].contiguous(), tb[:, 9].contiguous(), tb[:, 10].contiguous())

    def perturb_data(self, x_0, m_0, xx_0, mm_0, t, eps=None, want_f32=False):
        """Returns (u_t f64, mu_t f64, (xx_t, xm_t, mm_t)) like the reference; with ``want_f32`` the
        f32 cast of u_t (losses.py:114) comes back as a 4th element straight from the kernel."""
        drew = eps is None
        if drew:
            eps = torch.randn(x_0.shape[0], 2 * x_0.shape[1], *x_0.shape[2:], device=x_0.device)
        tb = self._coeff_table(t, xx_0, mm_0)
        z, u, mu = ops.perturb(x_0.contiguous(), None if m_0 is None else m_0.contiguous(), eps.contiguous(), tb,
                               self._params, want_f32=want_f32, want_f64=True, want_mu=True)
        var = (tb[:, 8].contiguous(), tb[:, 9].contiguous(), tb[:, 10].contiguous())
        if want_f32:
            return u, mu, var, z
        return u, mu, var

    def predict_x_from_eps(self, z_t, eps, t):
        """psld.py:289-328: the (x_0, m_0) pair implied by a noisy state and an epsilon.  Like the reference this takes
        ONE time for the whole batch (it builds a 2x2 matrix from ``b_t``); ``t`` may be a float or a one-element
        tensor.  Off the training / sampling path (nothing in the reference calls it): composed from the coefficient
        kernel and a handful of elementwise torch ops on the device tensors."""
        if torch.is_tensor(t):
            tf = t.reshape(-1)
            assert tf.numel() == 1 or bool((tf == tf[0]).all()), "predict_x_from_eps takes ONE time for the whole batch"
            tt = float(tf[0])
        else:
            tt = float(t)
        tv = torch.full((1,), tt, dtype=torch.float64, device=z_t.device)
        var = self._cov(0.0, self.mm_0, tv)
        l11, l12, l21, l22 = (c.reshape(()) for c in self.get_coeff(var))
        eps_x, eps_m = torch.chunk(eps, 2, dim=1)
        z_x, z_m = torch.chunk(z_t, 2, dim=1)
        mu_x = z_x - (l11 * eps_x + l12 * eps_m)
        mu_m = z_m - (l21 * eps_x + l22 * eps_m)
        b = self.b_t(tt)
        sf = math.exp((self.nu + self.gamma) / 4 * b)
        a1, a2 = (self.nu - self.gamma) / 4, (self.gamma - self.nu) ** 2 / 8
        c1, c2 = -0.5, (self.gamma - self.nu) / 4
        cm = torch.tensor([[a1 * b + 1, a2 * b], [c1 * b, c2 * b + 1]], device=z_t.device, dtype=torch.float64)  # t is f64 there
        ci = torch.linalg.inv(cm) * sf
        return ci[0, 0] * mu_x + ci[0, 1] * mu_m, ci[1, 0] * mu_x + ci[1, 1] * mu_m

    def perturb_f32(self, x_0, m_0, xx_0, mm_0, t, eps) -> Tensor:
        """Training fast path: only the f32 state that feeds the network."""
        tb = self._coeff_table(t, xx_0, mm_0)
        z, _, _ = ops.perturb(x_0.contiguous(), None if m_0 is None else m_0.contiguous(), eps.contiguous(), tb,
                              self._params, want_f32=True)
        return z

    # ---- score / drift (psld.py:230-260, 330-364) ----------------------------------------------------------
    def _score_mode(self) -> int:
        if self.decomp_mode == "lower" and self.mode == "score_m":
            return 1
        if self.decomp_mode == "upper" and self.mode == "score_x":
            return 2
        return 0

    def _rows(self, u_t, t):
        """Device tensors for the per-sample-time kernels: f64 state, f64 t[B] (a 0-d / 1-element t is broadcast)."""
        if not u_t.is_cuda:
            raise RuntimeError("psld_amd.PSLD needs device tensors (no CPU fallback)")
        b = u_t.shape[0]
        tt = t.detach().to(device=u_t.device, dtype=torch.float64).reshape(-1)
        if tt.numel() == 1 and b != 1:
            tt = tt.expand(b)
        assert tt.numel() == b, (tuple(t.shape), b)
        return u_t.to(torch.float64).contiguous(), tt.contiguous()

    def get_score(self, eps, xx_0, mm_0, t):
        """score = -L_t^{-T} eps, f32 (psld.py:230-260) — host composition over tiny tensors."""
        var = self._cov(xx_0, mm_0, t)
        c11, c12, c21, c22 = self.get_inv_coeff(var)
        f32 = torch.float32
        r = lambda c, like: c.view(-1, *([1] * (like.dim() - 1))).type(f32)
        if self.decomp_mode == "lower" and self.mode == "score_m":
            return torch.cat([torch.zeros_like(eps), -r(c22, eps) * eps], dim=1)
        if self.decomp_mode == "upper" and self.mode == "score_x":
            return torch.cat([-r(c11, eps) * eps, torch.zeros_like(eps)], dim=1)
        ex, em = torch.chunk(eps, 2, dim=1)
        return torch.cat([-r(c11, ex) * ex - r(c12, em) * em, -r(c21, ex) * ex - r(c22, em) * em], dim=1)

    def sde(self, u_t, t):
        """(f, g) of psld.py:330-343.  ``t``: a float, or a tensor with one time per sample (stays on the device)."""
        if torch.is_tensor(t):
            u64, tt = self._rows(u_t, t)
            flag = torch.zeros(1, dtype=torch.int32, device=u64.device)
            out = ops.reverse_sde_rows(u64, None, tt, self._params, 0.0, self.mm_0, self._score_mode(), False, flag)
            if self.check_nan and int(flag.item()) != 0:
                raise ValueError("Numerical precision error.")
            return out
        tt = float(t)
        k = self.em_coeffs(tt, 0.0)
        k.c11 = k.c12 = k.c21 = k.c22 = 0.0   # score := 0 -> f_bar = -f
        zeros = torch.zeros(u_t.shape[0], u_t.shape[1] if k.score_mode == 0 else u_t.shape[1] // 2, *u_t.shape[2:],
                            device=u_t.device)
        fb, g = ops.reverse_sde(u_t.to(torch.float64).contiguous(), zeros, k)
        return -fb, g

    def reverse_sde(self, u_t, t, score_fn: Callable, probability_flow=False):
        """(f_bar, g_bar) of psld.py:345-364.  ``t``: a float (one host-computed coefficient set, the samplers' path),
        or a tensor with one time per sample: then nothing is read back from ``t`` - the network gets
        ``(T - t).float()`` and the kernel derives each sample's coefficients from its own time."""
        if torch.is_tensor(t):
            u64, tt = self._rows(u_t, t)
            t_rev = self.T - tt                                          # psld.py:348
            u32 = ops.f64_to_f32(u64) if u_t.dtype != torch.float32 else u_t.contiguous()
            eps_pred = score_fn(u32, ops.f64_to_f32(t_rev))              # psld.py:354
            flag = torch.zeros(1, dtype=torch.int32, device=u64.device)
            out = ops.reverse_sde_rows(u64, eps_pred.contiguous(), t_rev, self._params, 0.0, self.mm_0,
                                       self._score_mode(), probability_flow, flag)
            if self.check_nan and int(flag.item()) != 0:
                raise ValueError("Numerical precision error.")
            return out
        tt = self.T - float(t)
        u64 = u_t.to(torch.float64).contiguous()
        u32 = ops.f64_to_f32(u64) if u_t.dtype != torch.float32 else u_t.contiguous()
        t32 = torch.full((u_t.shape[0],), float(np.float32(tt)), device=u_t.device, dtype=torch.float32)
        eps_pred = score_fn(u32, t32)
        k = self.em_coeffs(tt, 0.0, probability_flow)
        return ops.reverse_sde(u64, eps_pred.contiguous(), k)

    def em_update(self, x64, eps_pred, noise, t_rev: float, dt: float, x32):
        """One Euler-Maruyama predictor update in place on the float64 state (samplers/sde.py:16-26)."""
        ops.em_step(x64, eps_pred, noise, self.em_coeffs(t_rev, dt), x32)

    def prior_sampling(self, shape, device=None):
        """psld.py:366-370 (drawn on the CPU by the reference; ``device`` draws it in place)."""
        p_x = torch.randn(*shape, device=device)
        p_m = torch.randn(*shape, device=device) * np.sqrt(self.m)
        return torch.cat([p_x, p_m], dim=1)

    def prior_logp(self, z):
        pass

    def likelihood_weighting(self, t):
        beta_t = self.beta_t(t)
        return beta_t * self.gamma, beta_t * self.m * self.nu
