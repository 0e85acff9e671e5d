"""VP-SDE baseline with the reference's interface (main/models/sde/vpsde.py:9-99), SURVEY.md 8(f) rank 4.
Non-augmented 3-channel state; the same NCSN++ (in_ch = out_ch = 3) predicts eps."""
from __future__ import annotations

import math

import numpy as np
import torch

from . import ops
from .registry import register_module


@register_module(category="sde", name="vpsde")
class VPSDE:
    def __init__(self, config):
        self.N = config.model.sde.n_timesteps
        self.beta_0 = config.model.sde.beta_min
        self.beta_1 = config.model.sde.beta_max

    def beta_t(self, t):
        return self.beta_0 + t * (self.beta_1 - self.beta_0)

    @property
    def T(self):
        return 1.0

    @property
    def type(self):
        return "vpsde"

    def _lmc(self, t):
        return -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0          # vpsde.py:74-76

    def _std(self, t):
        if torch.is_tensor(t):
            return torch.sqrt(1.0 - torch.exp(2.0 * self._lmc(t)))
        return math.sqrt(1.0 - math.exp(2.0 * self._lmc(t)))

    def get_score(self, eps, t):
        return -eps / self._std(t).view(-1, *([1] * (eps.dim() - 1)))                        # vpsde.py:26-27

    def perturb_data(self, x_0, t, noise=None):
        """vpsde.py:29-37 -> float64 x_t like the reference (f64 t promotes)."""
        if noise is None:
            noise = torch.randn_like(x_0)
        _, u = ops.vp_perturb(x_0.contiguous(), noise.contiguous(), t.to(torch.float64).contiguous(), self.beta_0,
                              self.beta_1, want_f32=False, want_f64=True)
        return u

    def perturb_f32(self, x_0, t, noise):
        z, _ = ops.vp_perturb(x_0.contiguous(), noise.contiguous(), t.to(torch.float64).contiguous(), self.beta_0,
                              self.beta_1, want_f32=True)
        return z

    def cond_marginal_prob(self, x_0, t):
        lmc = self._lmc(t)
        mean = torch.exp(lmc[:, None, None, None]) * x_0
        return mean, torch.sqrt(1.0 - torch.exp(2.0 * lmc)).view(-1, 1, 1, 1)

    @staticmethod
    def _uniform_time(t) -> float:
        if torch.is_tensor(t):
            vals = t.detach().reshape(-1).to(torch.float64).tolist()
            if any(v != vals[0] for v in vals):
                raise NotImplementedError("per-sample times in sde()/reverse_sde(): call once per distinct t")
            return float(vals[0])
        return float(t)

    def sde(self, x_t, t):
        beta = self.beta_t(self._uniform_time(t))
        return -0.5 * beta * x_t, math.sqrt(beta)

    def reverse_sde(self, x_t, t, score_fn, probability_flow=False):
        tt = self.T - self._uniform_time(t)                                                   # vpsde.py:50
        x64 = x_t.to(torch.float64).contiguous()
        x32 = ops.f64_to_f32(x64) if x_t.dtype != torch.float32 else x_t.contiguous()
        t32 = torch.full((x_t.shape[0],), float(np.float32(tt)), device=x_t.device, dtype=torch.float32)
        eps_pred = score_fn(x32, t32)
        beta = float(self.beta_t(tt))
        f = ops.vp_reverse(x64, eps_pred.contiguous(), None, beta, self._std(tt), 0.0, probability_flow, update=False)
        g = torch.zeros_like(f) if probability_flow else torch.full_like(f, math.sqrt(beta))
        return f, g

    def em_update(self, x64, eps_pred, noise, t_rev: float, dt: float, x32):
        """One Euler-Maruyama predictor update in place (samplers/sde.py:16-26)."""
        ops.vp_reverse(x64, eps_pred, noise, float(self.beta_t(t_rev)), self._std(t_rev), dt, False, update=True,
                       x_f32=x32)

    def prior_sampling(self, shape, device=None):
        return torch.randn(*shape, device=device)

    def prior_logp(self, z):
        n = np.prod(z.shape[1:])
        return -n / 2.0 * np.log(2 * np.pi) - torch.sum(z ** 2, dim=(1, 2, 3)) / 2.0

    def likelihood_weighting(self, t):
        return self.beta_t(t)
