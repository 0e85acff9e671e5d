"""Per-step parameter maintenance as single launches over the network's flat buffers.

``FusedAdam`` is a ``torch.optim.Optimizer`` (so ``LambdaLR`` and checkpointing of
``param_groups`` work as in wrapper.py:128-155) whose ``step()`` is: one grad-norm reduction, one
fused clip+Adam(+EMA) kernel over all parameters — replacing ``clip_grad_norm_`` + ``Adam.step``
(wrapper.py:82-86) and, optionally, ``EMAWeightUpdate`` (callbacks.py:57-64).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, module, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_clip=0.0,
                 ema_module=None, ema_decay=0.9999):
        params = [p for p in module.parameters()]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.module = module
        self.grad_clip = float(grad_clip)
        self.ema_module = ema_module
        self.ema_decay = float(ema_decay)
        self._step = 0
        self._m = self._v = self._norm = None
        self._backward_seen = -1     # module._backward_count at the last zero_grad()/step(): step() needs a newer one
        self.poison = None           # float32 device scalar (the step's loss): turned NaN by the kernel when it refuses the step

    def _state_buffers(self):
        flat = self.module.flatten_parameters()
        if self._m is None or self._m.numel() != flat.numel() or self._m.device != flat.device:
            self._m = torch.zeros_like(flat)
            self._v = torch.zeros_like(flat)
            self._norm = torch.zeros(1, dtype=torch.float64, device=flat.device)
            if flat.is_cuda:
                ops.gn_team_sync(flat.device)      # the error word the kernels are guarded by: exists before any capture
        return flat

    @property
    def grad_norm(self) -> Optional[torch.Tensor]:
        """Device tensor holding the pre-clip global L2 norm of the last step (no host sync)."""
        return self._norm

    def zero_grad(self, set_to_none: bool = True):
        # No memset of 390 MB: the network's next backward overwrites the gradient buffer.
        self.module.mark_grads_stale()
        if not set_to_none:
            self.module.flat_grad().zero_()
        elif self.module._params_visible():
            # gradients arrive through autograd's AccumulateGrad (torch DDP): a populated .grad would be added to
            for p in self.module._trainable():
                p.grad = None
        self._backward_seen = self.module._backward_count

    def _launch(self, hyper_dev=None, poison=None):
        """The kernels of one step (norm, clip + Adam (+ EMA)) against the current ``self._step`` / lr; with
        ``hyper_dev`` the two step-dependent scalars are read from device memory instead (captured training step).
        A device error word raised during this step's backward (``ops.gn_team_sync``) turns the launch into a no-op for
        p / m / v / ema and ``poison`` (the loss scalar) into NaN."""
        group = self.param_groups[0]
        flat = self._state_buffers()
        grad = self.module.flat_grad()
        if self.grad_clip > 0:
            ops.grad_norm(grad, self._norm)
        ema_flat = None
        if self.ema_module is not None:
            ema_flat = self.ema_module.flatten_parameters()
        # torch.optim.Adam skips parameters without a gradient; the kernel sweeps the whole flat buffer, where frozen
        # parameters (GaussianFourierProjection.W) see g = 0: a no-op unless weight decay is on -> keep them aside
        frozen = [p for p in self.module._params() if not p.requires_grad] if group["weight_decay"] != 0 else []
        kept = [p.detach().clone() for p in frozen]
        ops.adam_ema(flat, grad, self._m, self._v, ema_flat, self._norm if self.grad_clip > 0 else None,
                     self.grad_clip, group["lr"], group["betas"][0], group["betas"][1], group["eps"],
                     group["weight_decay"], max(1, self._step), self.ema_decay, hyper_dev=hyper_dev,
                     poison=poison if poison is not None else self.poison)
        for p, k in zip(frozen, kept):
            p.detach().copy_(k)

    def _after_step(self):
        self.module.weights_changed()
        self.module.mark_grads_stale()      # consumed: a following backward starts a fresh gradient
        self._backward_seen = self.module._backward_count
        if self.ema_module is not None:
            self.ema_module.weights_changed()

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        self._state_buffers()
        self.module.flat_grad()
        if self.module._backward_count == self._backward_seen:
            raise RuntimeError("FusedAdam.step(): no backward pass has run since the last zero_grad()/step(); the gradient "
                               "buffer holds the previous step's (consumed) gradient")
        self.module.adopt_foreign_grads()     # a foreign reducer may have replaced .grad (DDP bucket views)
        self._step += 1
        self._launch()
        self.poison = None
        self._after_step()

    def state_dict(self):
        sd = super().state_dict()
        sd["fused"] = {"step": self._step, "m": self._m, "v": self._v}
        return sd

    def load_state_dict(self, sd):
        fused = sd.pop("fused", None)
        super().load_state_dict(sd)
        if fused is not None:
            self._step = fused["step"]
            self._state_buffers()
            if fused["m"] is not None:
                self._m.copy_(fused["m"])
                self._v.copy_(fused["v"])


class EMAWeightUpdate:
    """callbacks.py:17-64: ``targ = targ*tau + src*(1-tau)`` over every parameter after each batch —
    one kernel over the flat buffers instead of a 749-tensor Python loop."""

    def __init__(self, tau: float = 0.9999):
        self.tau = tau

    def on_train_batch_end(self, trainer, pl_module, outputs=None, batch=None, batch_idx=0, *a, **k):
        self.update_weights(pl_module.score_fn, pl_module.ema_score_fn)

    def update_weights(self, online_net, target_net) -> None:
        with torch.no_grad():
            ops.ema(target_net.flatten_parameters(), online_net.flatten_parameters(), self.tau)
        target_net.weights_changed()
