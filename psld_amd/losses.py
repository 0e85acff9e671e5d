"""HSM / DSM score-matching loss with the reference's interface (main/losses.py:69-130).

``PSLDScoreLoss(config, sde).forward(x_0, t, score_fn, eps=None) -> scalar`` with an autograd graph
through ``score_fn``.  Perturbation, the f64->f32 casts and the squared-error reduction (+ its
gradient) are libpsld_hip kernels; ``torch.randn`` draws eps exactly as the reference does.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .registry import get_module, register_module


class _SqErr(torch.autograd.Function):
    """loss = mean|sum (target - pred)^2; the kernel also emits d loss / d pred."""

    @staticmethod
    def forward(ctx, pred, target, reduce_mean):
        loss, grad = ops.sqerr_loss(target, pred.contiguous(), reduce_mean, want_grad=True)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None   # g is the 0-d upstream gradient (1.0 for loss.backward())


@register_module(category="losses", name="psld_score_loss")
class PSLDScoreLoss(nn.Module):
    def __init__(self, config, sde):
        super().__init__()
        assert config.training.loss.weighting in ["fid"]
        assert config.training.mode in ["hsm", "dsm"]
        assert isinstance(sde, get_module("sde", "psld"))
        self.sde = sde
        self.l_type = config.training.loss.l_type
        self.weighting = config.training.loss.weighting
        self.mode = config.training.mode
        self.decomp_mode = config.model.sde.decomp_mode
        self.reduce_strategy = "mean" if config.training.loss.reduce_mean else "sum"

    def prefetch(self, t):
        """The perturbation coefficients ``forward`` will ask for (SDEWrapper computes and checks them early)."""
        self.sde.prefetch_coeffs(t, 0, self.sde.mm_0 if self.mode == "hsm" else 0.0)

    def forward(self, x_0, t, score_fn, eps=None, m_draw=None):
        """``eps`` / ``m_draw``: the two normal draws of losses.py:96,108 when the caller has made them already (tests;
        the captured training step, which keeps every RNG call outside its hipGraph)."""
        sde = self.sde
        # losses.py:96-102: DSM samples the momentum, HSM marginalises it; the momentum is drawn in BOTH modes (HSM
        # discards it), so a seeded run consumes the device RNG stream exactly like the reference
        if m_draw is None:
            m_draw = torch.randn_like(x_0)
        if self.mode == "hsm":
            m_0, mm_0 = None, sde.mm_0
        else:
            m_0, mm_0 = np.sqrt(sde.mm_0) * m_draw, 0.0
        if eps is None:
            eps = torch.randn(x_0.shape[0], 2 * x_0.shape[1], *x_0.shape[2:], device=x_0.device)
        assert eps.shape[1] == 2 * x_0.shape[1]
        eps = eps.contiguous()
        z_t = sde.perturb_f32(x_0, m_0, 0, mm_0, t, eps)                # losses.py:113-114
        t32 = ops.f64_to_f32(t.contiguous()) if t.dtype == torch.float64 else t.float()
        eps_pred = score_fn(z_t, t32)                                   # losses.py:115
        if sde.mode == "score_m" and self.decomp_mode == "lower":      # losses.py:118-127
            target = torch.chunk(eps, 2, dim=1)[1].contiguous()
        elif sde.mode == "score_x" and self.decomp_mode == "upper":
            target = torch.chunk(eps, 2, dim=1)[0].contiguous()
        else:
            target = eps
        assert eps_pred.shape == target.shape
        return _SqErr.apply(eps_pred, target, self.reduce_strategy == "mean")


class _VpScoreLoss(torch.autograd.Function):
    """L1 criterion (mode 1) / g(t)^2-weighted score error (mode 2); the kernel also emits d loss / d pred."""

    @staticmethod
    def forward(ctx, pred, target, t, beta0, beta1, mode, reduce_mean):
        loss, grad = ops.vp_score_loss(target, pred.contiguous(), t, beta0, beta1, mode, reduce_mean, want_grad=True)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g.to(grad.dtype), None, None, None, None, None, None


@register_module(category="losses", name="score_loss")
class ScoreLoss(nn.Module):
    """Loss for non-augmented score models (VP-SDE), main/losses.py:21-65: the eps-prediction criterion (MSE, or L1
    for ``l_type='l1'``) for weighting 'fid', the g(t)^2-weighted score error for weighting 'nll'."""

    def __init__(self, config, sde):
        super().__init__()
        assert config.training.loss.weighting in ["nll", "fid"]
        self.sde = sde
        self.l_type = config.training.loss.l_type
        self.weighting = config.training.loss.weighting
        if self.weighting == "nll" and self.l_type != "l2":
            raise ValueError("l_type can only be `l2` when using nll weighting")       # losses.py:33-35
        self.reduce_strategy = "mean" if config.training.loss.reduce_mean else "sum"

    def forward(self, x_0, t, score_fn, eps=None):
        if eps is None:
            eps = torch.randn_like(x_0)
        assert eps.shape == x_0.shape
        eps = eps.contiguous()
        x_t = self.sde.perturb_f32(x_0, t, eps)                          # losses.py:48-49
        t32 = ops.f64_to_f32(t.contiguous()) if t.dtype == torch.float64 else t.float()
        eps_pred = score_fn(x_t, t32)
        mean = self.reduce_strategy == "mean"
        if self.weighting == "nll":                                      # losses.py:55-63 (f64 loss: t is f64)
            return _VpScoreLoss.apply(eps_pred, eps, t.to(torch.float64).contiguous(), self.sde.beta_0, self.sde.beta_1,
                                      2, mean)
        if self.l_type != "l2":                                          # losses.py:38-39: nn.L1Loss
            return _VpScoreLoss.apply(eps_pred, eps, None, 0.0, 0.0, 1, mean)
        return _SqErr.apply(eps_pred, eps, mean)


class _SoftmaxXent(torch.autograd.Function):
    """loss = mean|sum cross entropy; the kernel also emits d loss / d logits and the top-1 hit count."""

    @staticmethod
    def forward(ctx, logits, labels, reduce_mean):
        scale = 1.0 / logits.shape[0] if reduce_mean else 1.0
        loss, grad, correct = ops.softmax_xent(logits.contiguous(), labels.contiguous(), scale, scale)
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(correct)
        return loss, correct

    @staticmethod
    def backward(ctx, g, _g_correct):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None


@register_module(category="losses", name="tce_loss")
class PSLDTimeCELoss(nn.Module):
    """Loss of the noise-conditioned classifier used for guidance (main/losses.py:132-178).
    ``forward(x_0, y, t, clf_fn) -> (loss, top-1 accuracy as a fraction in [0, 1] like util.compute_top_k)``; ``config`` is the root node holding
    ``.diffusion`` and ``.clf``.  Both ``randn_like`` draws of the reference are made, in its order (the momentum
    draw is discarded in HSM mode, exactly like there)."""

    def __init__(self, config, sde):
        super().__init__()
        assert config.diffusion.training.mode in ["hsm", "dsm"]
        assert isinstance(sde, get_module("sde", "psld"))
        self.sde = sde
        self.l_type = config.clf.training.loss.l_type
        self.mode = config.diffusion.training.mode
        self.reduce_strategy = "mean" if config.diffusion.training.loss.reduce_mean else "sum"

    def forward(self, x_0, y, t, clf_fn, m0_draw=None, eps=None):
        sde = self.sde
        if m0_draw is None:
            m0_draw = torch.randn_like(x_0)                              # losses.py:152 (drawn in both modes)
        if self.mode == "hsm":
            m_0, mm_0 = None, sde.mm_0
        else:
            m_0, mm_0 = np.sqrt(sde.mm_0) * m0_draw, 0.0
        if eps is None:
            eps = torch.randn(x_0.shape[0], 2 * x_0.shape[1], *x_0.shape[2:], device=x_0.device)   # :164
        u_t = sde.perturb_f32(x_0, m_0, 0, mm_0, t, eps.contiguous())   # losses.py:167 + the .type(float32) of :170
        t32 = ops.f64_to_f32(t.contiguous()) if t.dtype == torch.float64 else t.float()   # layers.py: timesteps.float()
        y_pred = clf_fn(u_t, t32)
        loss, correct = _SoftmaxXent.apply(y_pred, y, self.reduce_strategy == "mean")
        return loss, correct * (1.0 / y_pred.shape[0])                  # losses.py:11-15 compute_top_k(k=1): a fraction
