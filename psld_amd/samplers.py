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
        k = sde.em_coeffs(t_rev, dt)
        ops.em_step(x64, eps_pred.contiguous(), noise, k, x32)

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
