"""``SDEWrapper``: the reference's Lightning module surface (main/models/wrapper.py:12-155).

Same constructor signature, same ``training_step`` / ``predict_step`` / ``configure_optimizers``
semantics.  With pytorch_lightning installed it IS a ``pl.LightningModule`` (train_sde.py:57-60,
eval/sample.py:62-69 work unchanged); on boxes without Lightning (this image) a minimal base class
supplies ``optimizers()/lr_schedulers()/manual_backward()/log()`` and ``psld_amd.cli`` drives it.
"""
from __future__ import annotations

import contextlib

import torch
import torch.nn as nn

from . import ops
from .optim import FusedAdam
from .registry import register_module

try:  # pragma: no cover - Lightning is not in the offline image
    import pytorch_lightning as pl
    _Base = pl.LightningModule
    HAVE_LIGHTNING = True
except Exception:  # noqa: BLE001
    HAVE_LIGHTNING = False

    class _Base(nn.Module):
        """The slice of LightningModule that wrapper.py uses (manual optimisation)."""

        def __init__(self):
            super().__init__()
            self.global_rank = 0
            self._optim = None
            self._sched = None
            self.logged = {}

        @property
        def device(self):
            return next(self.parameters()).device

        def _ensure_optim(self):
            if self._optim is None:
                cfg = self.configure_optimizers()
                self._optim = cfg["optimizer"]
                self._sched = cfg["lr_scheduler"]["scheduler"]

        def optimizers(self):
            self._ensure_optim()
            return self._optim

        def lr_schedulers(self):
            self._ensure_optim()
            return self._sched

        def manual_backward(self, loss):
            loss.backward()

        def log(self, name, value, **kw):
            self.logged[name] = value


@register_module(category="pl_modules", name="sde_wrapper")
class SDEWrapper(_Base):
    def __init__(self, config, sde, score_fn, ema_score_fn=None, criterion=None, sampler_cls=None,
                 corrector_fn=None):
        super().__init__()
        self.config = config
        self.score_fn = score_fn
        self.ema_score_fn = ema_score_fn
        self.sde = sde
        self.criterion = criterion
        self.train_eps = self.config.training.train_eps
        self.sampler = None
        if sampler_cls is not None:
            self.sampler = sampler_cls(
                self.config, self.sde,
                self.ema_score_fn if config.evaluation.sample_from == "target" else self.score_fn,
                corrector_fn=corrector_fn)
        self.eval_eps = self.config.evaluation.eval_eps
        self.denoise = self.config.evaluation.denoise
        n = self.config.evaluation.n_discrete_steps
        self.n_discrete_steps = n - 1 if self.denoise else n           # wrapper.py:52-54
        self.val_eps = self.config.evaluation.eval_eps
        self.stride_type = self.config.evaluation.stride_type
        self.automatic_optimization = False
        self.fuse_ema = False   # set True to fold the EMA update into the optimiser kernel

    def forward(self):
        pass

    # ---- hipGraph-captured training step (launch-bound regime: small per-GPU batches) ---------------------------------
    def enable_graphs(self, flag: bool = True, warmup_steps: int = 2):
        """Capture the whole training step - perturb, forward, loss, backward tape, norm + clip + Adam - into a HIP
        graph per batch shape and replay it: one graph launch instead of ~2700 kernel launches issued from Python (the
        reference's per-GPU batch of 16, scripts_psld/.../train_uncond_psld.sh:25-30; measured in profiles/: worth a few
        percent on its own there - the eager step with the weight-gradient side stream is the faster mode at B = 16, the
        graph pays below B ~ 8 where the host is the bound).
        Everything the reference draws per step (t, the discarded momentum draw, eps) and the dropout seed is drawn
        OUTSIDE the graph into static device buffers, in the reference's order, so a seeded run consumes the RNG stream
        exactly like the eager step; the step-dependent Adam scalars and the LR-schedule value travel through a 2-float
        device buffer written before each replay.  The first ``warmup_steps`` steps of a shape run eagerly (they size
        workspaces and weight caches).  The NaN check of the perturbation coefficients (a host read) stays outside the
        graph, on the early stream of ``_draw_times``.  Not captured: gradient exchange (``set_reducer``), foreign
        optimizers."""
        self._graphs_on = bool(flag)
        self._graph_warmup = int(warmup_steps)
        if not flag:
            self._graph_steps = {}

    def _draw_times(self, b: int, dev):
        """``t_ ~ U[0, 1)`` and ``t`` (wrapper.py:72-73) plus the NaN check of the perturbation coefficients
        (psld.py:166-171, a host read) on a stream of their own that does NOT wait for the compute stream: none of it
        depends on the previous step, and on the compute stream the flag read would make the host wait for the whole
        queue of the previous step at every step boundary (4.4 ms of a 37 ms step at batch 16).  The random draw is
        the first of the step on the host, so the device RNG stream is consumed in the reference's order."""
        main = torch.cuda.current_stream(dev)
        early = getattr(self, "_early", None)
        if early is None or early.device != dev:
            early = self._early = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(early):
            t_ = torch.rand(b, device=dev, dtype=torch.float64)
            t = t_ * (self.sde.T - self.train_eps) + self.train_eps
            prefetch = getattr(self.criterion, "prefetch", None)
            if prefetch is not None and getattr(self.sde, "check_nan", False):
                prefetch(t)
        main.wait_stream(early)
        t_.record_stream(main)
        t.record_stream(main)
        return t_, t

    def _graph_ok(self, batch) -> bool:
        optim = self.optimizers()
        return (getattr(self, "_graphs_on", False) and torch.is_tensor(batch) and batch.is_cuda
                and isinstance(optim, FusedAdam) and getattr(self.score_fn, "_reducer", None) is None
                and not self.score_fn._params_visible() and optim.ema_module is None
                and optim.param_groups[0]["weight_decay"] == 0)

    def _graphed_step(self, batch):
        net, optim, sde = self.score_fn, self.optimizers(), self.sde
        if not hasattr(self, "_graph_steps"):
            self._graph_steps = {}
        key = (tuple(batch.shape), batch.device.index, bool(net.training))
        ent = self._graph_steps.get(key)
        if ent is None:
            ent = self._graph_steps[key] = {"seen": 0}
        if ent["seen"] < self._graph_warmup:
            ent["seen"] += 1
            return None                                   # eager warm-up step
        dev = batch.device
        b, c, h, w = batch.shape
        if "graph" not in ent:
            ent["x0"] = torch.empty_like(batch)
            ent["t_"] = torch.empty(b, device=dev, dtype=torch.float64)
            ent["m_draw"] = torch.empty_like(batch)
            ent["eps"] = torch.empty((b, 2 * c, h, w), device=dev, dtype=torch.float32)
            ent["seed"] = torch.zeros(1, device=dev, dtype=torch.int64)
            ent["hyper"] = torch.zeros(2, device=dev, dtype=torch.float32)
        # the step's random draws, in the eager step's order (wrapper.py:72, losses.py:96,108, then the dropout seed)
        t_new, _ = self._draw_times(b, dev)       # includes the NaN check the captured step cannot make
        ent["t_"].copy_(t_new)
        torch.randn(batch.shape, device=dev, out=ent["m_draw"])
        torch.randn(ent["eps"].shape, device=dev, out=ent["eps"])
        if net.training and float(net.sf.dropout) > 0:
            ent["seed"].random_(0, 2 ** 62)
        ent["x0"].copy_(batch)
        group = optim.param_groups[0]
        # the two step-dependent Adam scalars: written by a kernel that takes them BY VALUE, in stream order (a pinned
        # host buffer rewritten per step would be read by its asynchronous copies only when they execute - with graphs
        # the host runs several steps ahead)
        ops.adam_step_scalars_dev(group["lr"], group["betas"][0], group["betas"][1], optim._step + 1, ent["hyper"])
        if "graph" not in ent:
            optim._state_buffers()
            net.flat_grad()
            check_nan, sde.check_nan = getattr(sde, "check_nan", False), False
            graph = torch.cuda.CUDAGraph()
            # the dropout seed word is baked into the graph as a pointer: the attribute is set for the capture only
            # (an eager step that found it set would skip its own seed draw and repeat the static word's mask)
            net._dropout_seed_dev = ent["seed"]
            try:
                with torch.cuda.graph(graph):
                    t = ent["t_"] * (sde.T - self.train_eps) + self.train_eps
                    loss = self.criterion(ent["x0"], t, net, eps=ent["eps"], m_draw=ent["m_draw"])
                    net.mark_grads_stale()
                    loss.backward()
                    with torch.no_grad():
                        optim._launch(hyper_dev=ent["hyper"], poison=loss.detach())
            finally:
                sde.check_nan = check_nan
                net._dropout_seed_dev = None
            ent["graph"], ent["loss"] = graph, loss.detach()
            # the graph holds raw pointers into the network's arenas and job tables: from now on an arena
            # that has to grow keeps its old buffer alive and the table cache does not evict
            net.pin_scratch()
        optim._step += 1                    # after a successful capture: a failed one leaves the step count alone
        ent["graph"].replay()
        optim._after_step()
        optim._opt_called = True          # what LambdaLR's step-order check looks at (optimizer.step() ran)
        self.lr_schedulers().step()
        return ent["loss"]

    def training_step(self, batch, batch_idx):
        if self._graph_ok(batch):
            loss = self._graphed_step(batch)
            if loss is not None:
                self.log("loss", loss, prog_bar=True)
                return loss
        optim = self.optimizers()
        lr_sched = self.lr_schedulers()
        x_0 = batch
        if x_0.is_cuda:
            _, t = self._draw_times(x_0.shape[0], x_0.device)                   # wrapper.py:72-73
        else:
            t_ = torch.rand(x_0.shape[0], device=x_0.device, dtype=torch.float64)
            t = t_ * (self.sde.T - self.train_eps) + self.train_eps
        loss = self.criterion(x_0, t, self.score_fn)
        optim.zero_grad()
        self.manual_backward(loss)
        gc = self.config.training.optimizer.grad_clip
        fused = isinstance(getattr(optim, "optimizer", optim), FusedAdam)
        if gc != 0 and not fused:
            torch.nn.utils.clip_grad_norm_(self.score_fn.parameters(), gc)   # wrapper.py:82-85
        if fused and loss.is_cuda and loss.dtype == torch.float32:
            getattr(optim, "optimizer", optim).poison = loss.detach()   # a refused step (device error word) logs NaN
        optim.step()                  # FusedAdam: norm + clip + Adam (+EMA) in two launches
        lr_sched.step()
        self.log("loss", loss, prog_bar=True)
        return loss

    def on_predict_start(self):
        seed = self.config.evaluation.seed
        torch.manual_seed(seed + self.global_rank)                        # wrapper.py:93-99

    def sampling_times(self, device):
        t_final = self.sde.T - self.eval_eps
        ts = torch.linspace(0, t_final, self.n_discrete_steps + 1, device=device, dtype=torch.float64)
        if self.stride_type == "quadratic":
            ts = t_final * torch.flip(1 - (ts / t_final) ** 2.0, dims=[0])
        return ts

    def predict_step(self, batch, batch_idx, dataloader_idx=None):
        first = batch[0] if isinstance(batch, (tuple, list)) else batch   # (x_0, mask) for the inpainter
        ts = self.sampling_times(first.device)                           # wrapper.py:101-114
        return self.sampler.sample(batch, ts, self.n_discrete_steps, denoise=self.denoise, eps=self.eval_eps)

    def configure_optimizers(self):
        oc = self.config.training.optimizer
        if oc.name != "Adam":
            raise NotImplementedError(f"Optimizer {oc.name} not supported yet!")
        optimizer = FusedAdam(self.score_fn, lr=oc.lr, betas=(oc.beta_1, oc.beta_2), eps=oc.eps,
                              weight_decay=oc.weight_decay, grad_clip=oc.grad_clip,
                              ema_module=self.ema_score_fn if self.fuse_ema else None,
                              ema_decay=self.config.training.ema_decay)
        if oc.warmup == 0:
            lr_lambda = lambda step: 1.0
        else:
            lr_lambda = lambda step: min(step / oc.warmup, 1.0)          # wrapper.py:143-147
        scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda)
        return {"optimizer": optimizer,
                "lr_scheduler": {"scheduler": scheduler, "interval": "step", "strict": False}}


@register_module(category="pl_modules", name="tclf_wrapper")
class TClfWrapper(_Base):
    """The reference's classifier-guidance Lightning module (main/models/clf_wrapper.py:11-135):
    ``training_step((x_0, y))`` trains the noise-conditioned classifier with ``tce_loss``; ``predict_step`` draws
    class-conditional samples with ``cc_em_sde``.  ``config`` is the root node with ``.diffusion`` and ``.clf``.
    The reference leaves the optimisation to Lightning's automatic mode (zero_grad, backward, Adam step, LambdaLR
    step per batch, no gradient clipping); the same sequence is issued here explicitly."""

    def __init__(self, config, sde, clf_fn, score_fn=None, criterion=None, sampler_cls=None, corrector_fn=None):
        super().__init__()
        self.config = config
        self.sde = sde
        self.clf_fn = clf_fn
        self.criterion = criterion
        self.train_eps = self.config.diffusion.training.train_eps
        self.score_fn = score_fn
        self.sampler = None
        if sampler_cls is not None:
            self.sampler = sampler_cls(self.config, self.sde, self.score_fn, self.clf_fn, corrector_fn=corrector_fn)
        ev = self.config.diffusion.evaluation
        self.eval_eps = ev.eval_eps
        self.denoise = ev.denoise
        self.n_discrete_steps = ev.n_discrete_steps - 1 if self.denoise else ev.n_discrete_steps
        self.val_eps = ev.eval_eps
        self.stride_type = ev.stride_type
        self.automatic_optimization = False

    def forward(self):
        pass

    def training_step(self, batch, batch_idx):
        optim = self.optimizers()
        lr_sched = self.lr_schedulers()
        x_0, y = batch
        t_ = torch.rand(x_0.shape[0], device=x_0.device, dtype=torch.float64)   # clf_wrapper.py:63-64
        t = t_ * (self.sde.T - self.train_eps) + self.train_eps
        loss, acc = self.criterion(x_0, y, t, self.clf_fn)
        optim.zero_grad()
        self.manual_backward(loss)
        optim.step()
        lr_sched.step()
        self.log("loss", loss, prog_bar=True)
        self.log("Top1-Acc", acc, prog_bar=True)
        return loss

    def on_predict_start(self):
        torch.manual_seed(self.config.clf.evaluation.seed + self.global_rank)   # clf_wrapper.py:73-79

    def sampling_times(self, device):
        t_final = self.sde.T - self.eval_eps
        ts = torch.linspace(0, t_final, self.n_discrete_steps + 1, device=device, dtype=torch.float64)
        if self.stride_type == "quadratic":
            ts = t_final * torch.flip(1 - (ts / t_final) ** 2.0, dims=[0])
        return ts

    def predict_step(self, batch, batch_idx, dataloader_idx=None):
        return self.sampler.sample(batch, self.sampling_times(batch.device), self.n_discrete_steps,
                                   denoise=self.denoise, eps=self.eval_eps)

    def configure_optimizers(self):
        oc = self.config.clf.training.optimizer
        if oc.name != "Adam":
            raise NotImplementedError(f"Optimizer {oc.name} not supported yet!")
        optimizer = FusedAdam(self.clf_fn, lr=oc.lr, betas=(oc.beta_1, oc.beta_2), eps=oc.eps,
                              weight_decay=oc.weight_decay, grad_clip=0.0)
        lr_lambda = (lambda step: 1.0) if oc.warmup == 0 else (lambda step: min(step / oc.warmup, 1.0))
        scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda)
        return {"optimizer": optimizer,
                "lr_scheduler": {"scheduler": scheduler, "interval": "step", "strict": False}}
