"""Stand-alone drivers mirroring the reference's two entry points without Hydra / Lightning:

    python -m psld_amd.cli train  --config c10_sota [key=value ...]        (main/train_sde.py:21-120)
    python -m psld_amd.cli sample --config c10_sota [key=value ...]        (main/eval/sample.py:28-109)
    python -m psld_amd.cli inpaint --config c10_sota --data x.npy --mask m.npy [key=value ...]
                                                                           (main/eval/inpaint.py:29-135)
    python -m psld_amd.cli train_clf --config c10_sota --clf-config clf_c10 --data x.npy --labels y.npy [...]
                                                                           (main/train_clf.py:24-110)
    python -m psld_amd.cli cc_sample --config c10_sota --clf-config clf_c10 [dataset.clf.evaluation.label_to_sample=9 ...]
                                                                           (main/eval/class_cond_sample.py:29-138)

For the two classifier-guidance commands ``dataset.clf.<key>=value`` / ``clf.<key>=value`` go to the ``clf`` node.

``key=value`` are Hydra-style dotted overrides on the ``dataset.diffusion`` node; the prefix
``dataset.diffusion.`` is accepted and stripped, so the override lists of ``scripts_psld/**.sh`` can be
pasted unchanged (``+dataset=...`` group selectors are ignored: ``--config`` picks the preset).
Multi-GPU: launch with ``torchrun --nproc-per-node N`` — one process per GPU, RCCL gradient all-reduce
for training (``psld_amd.ddp.BucketReducer``), collective-free sharding for sampling.

Checkpoints use Lightning's layout (``{"state_dict": {"score_fn.<key>": ..., "ema_score_fn.<key>": ...},
"global_step": ..., "epoch": ...}``, train_sde.py:67-73 / eval/sample.py:62-69), so ``.ckpt`` files written
by the reference load here and vice versa.
"""
from __future__ import annotations

import argparse
import ast
import copy
import os
import sys
import time

import numpy as np
import torch


def parse_overrides(cfg, items):
    for it in items:
        if "=" not in it:
            continue
        k, v = it.split("=", 1)
        k = k.lstrip("+")
        if k == "dataset":          # Hydra group selector (+dataset=cifar10/cifar10_psld): --config picks the preset
            continue
        if k.startswith("dataset.diffusion."):
            k = k[len("dataset.diffusion."):]
        v = v.strip()
        vs = v.strip("\\'\"")
        try:
            val = ast.literal_eval(vs)
        except (ValueError, SyntaxError):
            val = {"true": True, "false": False}.get(vs.lower(), vs)
        cfg.override(k, val)
    return cfg


def build(cfg, device):
    import psld_amd
    from psld_amd.registry import get_module
    psld_amd.import_modules_into_registry()
    score_fn = get_module("score_fn", cfg.model.score_fn.name)(cfg).to(device)
    ema = copy.deepcopy(score_fn)
    for p in ema.parameters():
        p.requires_grad = False
    sde = get_module("sde", cfg.model.sde.name)(cfg)
    return score_fn, ema, sde


def save_checkpoint(path, wrapper, optim, step, epoch, sched=None):
    sd = {}
    for k, v in wrapper.score_fn.state_dict().items():
        sd["score_fn." + k] = v.detach().cpu()
    for k, v in wrapper.ema_score_fn.state_dict().items():
        sd["ema_score_fn." + k] = v.detach().cpu()
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save({"state_dict": sd, "global_step": step, "epoch": epoch,
                "optimizer_states": [optim.state_dict()] if optim is not None else [],
                "lr_schedulers": [sched.state_dict()] if sched is not None else []}, path)   # Lightning's key


def adopt_torch_adam_state(optim, state_dict) -> bool:
    """Moments of a ``torch.optim.Adam`` state_dict (what a reference / Lightning checkpoint carries: per-parameter
    ``exp_avg`` / ``exp_avg_sq`` / ``step`` keyed by parameter index) -> FusedAdam's flat m / v buffers.  True when
    every trainable parameter was found with a matching shape."""
    state = state_dict.get("state", {})
    params = [p for p in optim.module.parameters()]
    if not state or any(not isinstance(v, dict) or "exp_avg" not in v for v in state.values()):
        return False
    optim._state_buffers()
    offs = optim.module._offsets
    steps = set()
    for i, p in enumerate(params):
        ent = state.get(i, state.get(str(i)))
        if ent is None:
            if p.requires_grad:
                return False
            continue
        if tuple(ent["exp_avg"].shape) != tuple(p.shape):
            return False
    for i, p in enumerate(params):
        ent = state.get(i, state.get(str(i)))
        if ent is None:
            continue
        o = offs[id(p)]
        optim._m[o:o + p.numel()].copy_(ent["exp_avg"].reshape(-1))
        optim._v[o:o + p.numel()].copy_(ent["exp_avg_sq"].reshape(-1))
        steps.add(int(ent["step"]) if "step" in ent else -1)
    if len(steps) == 1 and next(iter(steps)) >= 0:
        optim._step = next(iter(steps))
    return True


def load_checkpoint(path, score_fn, ema, optim=None, sched=None):
    """Returns (global_step, epoch).  The optimiser state is restored from a FusedAdam state (own checkpoints) or
    converted from a torch.optim.Adam state (reference checkpoints); if neither works the moments restart from zero
    TOGETHER with the bias-correction step, and that is said loudly.  The LR scheduler state is restored when the
    checkpoint has one (Lightning's ``lr_schedulers``), else re-derived from the global step."""
    import warnings
    ck = torch.load(path, map_location="cpu", weights_only=False)
    sd = ck["state_dict"] if "state_dict" in ck else ck
    s1 = {k[len("score_fn."):]: v for k, v in sd.items() if k.startswith("score_fn.")}
    s2 = {k[len("ema_score_fn."):]: v for k, v in sd.items() if k.startswith("ema_score_fn.")}
    score_fn.load_state_dict(s1, strict=True)
    ema.load_state_dict(s2 if s2 else s1, strict=True)
    step = ck.get("global_step", 0)
    if optim is not None:
        states = ck.get("optimizer_states") or []
        restored = False
        if states:
            osd = states[0]
            if "fused" in osd:
                optim.load_state_dict(dict(osd))
                restored = True
            else:
                restored = adopt_torch_adam_state(optim, osd)
                if restored and optim._step == 0:
                    optim._step = step
        if not restored:
            warnings.warn(f"{path}: no usable optimizer state - Adam moments AND the bias-correction step restart from "
                          "zero (weights, EMA and the LR schedule position are restored)", RuntimeWarning, stacklevel=2)
            optim._step = 0
            if optim._m is not None:
                optim._m.zero_(), optim._v.zero_()
    if sched is not None:
        scheds = ck.get("lr_schedulers") or []
        if scheds:
            sched.load_state_dict(scheds[0])
        else:
            sched.last_epoch = step
            sched._step_count = step + 1
        for group, lr in zip(sched.optimizer.param_groups, [base * lmbda(sched.last_epoch) for lmbda, base in
                                                            zip(sched.lr_lambdas, sched.base_lrs)]):
            group["lr"] = lr
        sched._last_lr = [g["lr"] for g in sched.optimizer.param_groups]
    return step, ck.get("epoch", 0)


def _dataset(cfg, args, device, rank):
    """uint8 [N,H,W,3] array (``--data file.npy``) or synthetic CIFAR-shaped images; decoded on the device."""
    size = cfg.data.image_size
    if args.data and args.data != "synthetic":
        arr = np.load(args.data, mmap_mode="r")
        assert arr.dtype == np.uint8 and arr.shape[1:] == (size, size, 3), arr.shape
        return torch.from_numpy(np.ascontiguousarray(arr)).to(device)
    g = torch.Generator().manual_seed(1234)          # ONE dataset, the same on every rank; ranks take shards of it
    return torch.randint(0, 256, (args.synthetic_size, size, size, 3), generator=g, dtype=torch.uint8).to(device)


def epoch_indices(n_items: int, batch_size: int, seed: int, epoch: int, rank: int, world: int) -> torch.Tensor:
    """Indices this rank trains on in ``epoch``, in order: what Lightning's strategy="ddp" gives the reference
    (train_sde.py:100-114) - a ``DistributedSampler(shuffle=True)`` over the dataset (one permutation per epoch from
    ``seed + epoch``, identical on every rank, padded to a multiple of ``world`` by wrapping around, rank r takes
    every world-th element from r) under a ``DataLoader(drop_last=True)``.  Every rank gets the same number of
    batches, so the per-step collectives cannot deadlock; an epoch is ONE pass over the data in total."""
    g = torch.Generator().manual_seed(seed + epoch)
    perm = torch.randperm(n_items, generator=g)
    total = -(-n_items // world) * world
    if total > n_items:             # DistributedSampler: repeat the list as often as the padding needs (world > n_items)
        pad = total - n_items
        perm = torch.cat([perm, perm.repeat(-(-pad // n_items))[:pad]])
    mine = perm[rank:total:world]
    return mine[:mine.numel() // batch_size * batch_size]


def train(args, overrides):
    from psld_amd import config as C, ops
    from psld_amd.ddp import BucketReducer, init_distributed
    from psld_amd.optim import EMAWeightUpdate
    from psld_amd.registry import get_module
    import torch.distributed as dist
    rank, local, world = init_distributed()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cfg = parse_overrides(getattr(C, args.config)(), overrides)
    torch.manual_seed(cfg.training.seed)                                 # train_sde.py:29
    score_fn, ema, sde = build(cfg, dev)
    score_fn.train()
    crit = get_module("losses", cfg.training.loss.name)(cfg, sde)
    wrapper = get_module("pl_modules", cfg.model.pl_module)(cfg, sde, score_fn, ema_score_fn=ema, criterion=crit)
    ema_cb = EMAWeightUpdate(cfg.training.ema_decay) if cfg.training.use_ema else None
    if world > 1:
        score_fn.set_reducer(BucketReducer())
    optim = wrapper.optimizers()
    sched = wrapper.lr_schedulers()
    step, epoch0 = 0, 0
    if cfg.training.restore_path:
        score_fn.flatten_parameters()
        step, epoch0 = load_checkpoint(cfg.training.restore_path, score_fn, ema, optim, sched)
        score_fn.to(dev), ema.to(dev)
    data = _dataset(cfg, args, dev, rank)
    bs = min(cfg.training.batch_size, data.shape[0])                      # train_sde.py:101-102
    gen = torch.Generator(device=dev).manual_seed(cfg.training.seed + rank)   # horizontal flips: per-rank stream
    ckdir = os.path.join(cfg.training.results_dir or "psld_results", "checkpoints")
    t0 = time.perf_counter()
    for epoch in range(epoch0, cfg.training.epochs):
        perm = epoch_indices(data.shape[0], bs, cfg.training.seed, epoch, rank, world).to(dev)
        n = perm.numel()
        for i in range(0, n, bs):
            idx = perm[i:i + bs]
            flip = (torch.rand(bs, device=dev, generator=gen) < 0.5).to(torch.uint8) if cfg.data.hflip else None
            x0 = ops.uint8_to_images(data[idx].contiguous(), norm=cfg.data.norm, flip=flip)
            loss = wrapper.training_step(x0, step)
            if ema_cb is not None:
                ema_cb.on_train_batch_end(None, wrapper)
            step += 1
            if rank == 0 and step % max(1, cfg.training.log_step * args.log_every) == 0:
                dt = time.perf_counter() - t0
                print(f"epoch {epoch} step {step} loss {loss.item():.4f} ({step * bs * world / dt:.1f} img/s)", flush=True)
            if args.max_steps and step >= args.max_steps:
                break
        ops.check_device_errors(dev)        # every rank, once per epoch, before anything is written
        if rank == 0 and ((epoch + 1) % cfg.training.chkpt_interval == 0 or (args.max_steps and step >= args.max_steps)):
            name = f"{cfg.model.sde.name}-{cfg.training.chkpt_prefix}-epoch={epoch:02d}-loss={loss.item():.4f}.ckpt"
            save_checkpoint(os.path.join(ckdir, name), wrapper, optim, step, epoch + 1, sched)
            save_checkpoint(os.path.join(ckdir, "last.ckpt"), wrapper, optim, step, epoch + 1, sched)
        if args.max_steps and step >= args.max_steps:
            break
    if world > 1:
        dist.barrier(device_ids=[local])
        dist.destroy_process_group()


def sample(args, overrides):
    from psld_amd import config as C, ops
    from psld_amd.ddp import init_distributed, shard_range
    from psld_amd.registry import get_module
    import torch.distributed as dist
    rank, local, world = init_distributed()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cfg = parse_overrides(getattr(C, args.config)(), overrides)
    ev = cfg.evaluation
    score_fn, ema, sde = build(cfg, dev)
    if ev.chkpt_path:
        load_checkpoint(ev.chkpt_path, score_fn, ema)
        score_fn.to(dev), ema.to(dev)
    score_fn.eval(), ema.eval()
    sampler_cls = get_module("samplers", ev.sampler.name)
    wrapper = get_module("pl_modules", cfg.model.pl_module)(cfg, sde, score_fn, ema_score_fn=ema,
                                                            sampler_cls=sampler_cls)
    wrapper.global_rank = rank
    wrapper.on_predict_start()                                            # seed + rank (wrapper.py:93-99)
    lo, hi = shard_range(ev.n_samples, rank, world)                       # eval/sample.py:108-109 shards the latents
    base = os.path.join(ev.save_path or "psld_samples", str(ev.path_prefix)) if ev.path_prefix else (ev.save_path or "psld_samples")
    out_dir = os.path.join(base, "images")
    os.makedirs(out_dir, exist_ok=True)
    size, ch = cfg.data.image_size, cfg.data.num_channels
    t0 = time.perf_counter()
    for bi, start in enumerate(range(lo, hi, ev.batch_size)):
        b = min(ev.batch_size, hi - start)
        batch = sde.prior_sampling((b, ch, size, size), device=dev)       # latent.py:10-19 (drawn per batch, on device)
        x = wrapper.predict_step(batch, bi)
        if x.dtype != torch.float64:
            x = x.double()
        u8 = ops.samples_to_uint8(x.contiguous(), is_augmented=cfg.model.sde.is_augmented, denorm=cfg.data.norm).cpu().numpy()
        stem = os.path.join(out_dir, f"output_{ev.sample_prefix}_{rank}_{bi}")   # callbacks.py:120-122
        if ev.save_mode == "image":
            try:
                from PIL import Image
                for i, im in enumerate(u8):
                    Image.fromarray(im).save(stem + "_%d.png" % i, "png")
            except ImportError:
                np.save(stem + ".npy", u8)
        else:
            np.save(stem + ".npy", u8)
        if rank == 0:
            done = start + b - lo
            print(f"rank 0: {done}/{hi - lo} samples, {done / (time.perf_counter() - t0):.2f} img/s", flush=True)
    if world > 1:
        dist.barrier(device_ids=[local])
        dist.destroy_process_group()


def _split_overrides(items):
    """(diffusion overrides, clf overrides): ``dataset.clf.x=..`` / ``clf.x=..`` belong to the clf node."""
    dif, clf = [], []
    for it in items:
        k = it.split("=", 1)[0].lstrip("+")
        if k.startswith("dataset.clf.") or k.startswith("clf."):
            clf.append(it.lstrip("+").replace("dataset.clf.", "", 1).replace("clf.", "", 1) if "=" in it else it)
        else:
            dif.append(it)
    return dif, clf


def _root_config(args, overrides):
    from psld_amd import config as C
    dif, clf = _split_overrides(overrides)
    dcfg = parse_overrides(getattr(C, args.config)(), dif)
    ccfg = parse_overrides(getattr(C, args.clf_config)(), clf)
    ccfg.data.image_size = dcfg.data.image_size
    return C.with_clf(dcfg, ccfg)


def train_clf(args, overrides):
    """main/train_clf.py: train the noise-conditioned classifier (``ncsnpp_clf`` + ``tce_loss``) on (image, label)
    pairs; checkpoints in Lightning layout under ``clf.training.results_dir/checkpoints`` (keys ``clf_fn.<param>``)."""
    from psld_amd import ops
    from psld_amd.ddp import BucketReducer, init_distributed
    from psld_amd.registry import get_module
    import psld_amd
    import torch.distributed as dist
    rank, local, world = init_distributed()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    root = _root_config(args, overrides)
    cc = root.clf
    torch.manual_seed(cc.training.seed)
    psld_amd.import_modules_into_registry()
    clf = get_module("clf_fn", cc.model.clf_fn.name)(cc).to(dev).train()
    sde = get_module("sde", root.diffusion.model.sde.name)(root.diffusion)
    crit = get_module("losses", cc.training.loss.name)(root, sde)
    wrapper = get_module("pl_modules", cc.model.pl_module)(root, sde, clf, score_fn=None, criterion=crit)
    if world > 1:
        clf.set_reducer(BucketReducer())
    data = _dataset(root.diffusion, args, dev, rank)
    if args.labels and args.labels != "synthetic":
        labels = torch.from_numpy(np.load(args.labels).astype(np.int64)).to(dev)
    else:
        labels = torch.randint(0, cc.model.clf_fn.n_cls, (data.shape[0],),
                               generator=torch.Generator().manual_seed(4321)).to(dev)     # one label set for all ranks
    assert labels.shape[0] == data.shape[0]
    bs = min(cc.training.batch_size, data.shape[0])
    gen = torch.Generator(device=dev).manual_seed(cc.training.seed + rank)
    ckdir = os.path.join(cc.training.results_dir or "psld_clf_results", "checkpoints")
    step = 0
    for epoch in range(cc.training.epochs):
        perm = epoch_indices(data.shape[0], bs, cc.training.seed, epoch, rank, world).to(dev)   # DistributedSampler shard
        n = perm.numel()
        for i in range(0, n, bs):
            idx = perm[i:i + bs]
            flip = (torch.rand(bs, device=dev, generator=gen) < 0.5).to(torch.uint8) if cc.data.hflip else None
            x0 = ops.uint8_to_images(data[idx].contiguous(), norm=cc.data.norm, flip=flip)
            loss = wrapper.training_step((x0, labels[idx].contiguous()), step)
            step += 1
            if rank == 0 and step % max(1, cc.training.log_step * args.log_every) == 0:
                print(f"epoch {epoch} step {step} loss {loss.item():.4f} top1 {100 * float(wrapper.logged['Top1-Acc']):.1f}%", flush=True)
            if args.max_steps and step >= args.max_steps:
                break
        done = args.max_steps and step >= args.max_steps
        if rank == 0 and ((epoch + 1) % cc.training.chkpt_interval == 0 or done):
            sd = {"clf_fn." + k: v.detach().cpu() for k, v in clf.state_dict().items()}
            os.makedirs(ckdir, exist_ok=True)
            name = f"{cc.model.clf_fn.name}-{root.diffusion.model.sde.name}-{cc.training.chkpt_prefix}-epoch={epoch:02d}-loss={loss.item():.4f}.ckpt"
            for fn in (name, "last.ckpt"):
                torch.save({"state_dict": sd, "global_step": step, "epoch": epoch + 1}, os.path.join(ckdir, fn))
        if done:
            break
    if world > 1:
        dist.barrier(device_ids=[local])
        dist.destroy_process_group()


def cc_sample(args, overrides):
    """main/eval/class_cond_sample.py: class-conditional samples with classifier guidance (``cc_em_sde``)."""
    from psld_amd import ops
    from psld_amd.ddp import init_distributed, shard_range
    from psld_amd.registry import get_module
    import torch.distributed as dist
    rank, local, world = init_distributed()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    root = _root_config(args, overrides)
    cfg, cc = root.diffusion, root.clf
    ev = cfg.evaluation
    score_fn, ema, sde = build(cfg, dev)
    if ev.chkpt_path:
        load_checkpoint(ev.chkpt_path, score_fn, ema)
        score_fn.to(dev), ema.to(dev)
    net = ema if ev.sample_from == "target" else score_fn
    net.eval()
    clf = get_module("clf_fn", cc.model.clf_fn.name)(cc)
    if cc.evaluation.chkpt_path:
        ck = torch.load(cc.evaluation.chkpt_path, map_location="cpu", weights_only=False)
        sd = ck["state_dict"] if "state_dict" in ck else ck
        clf.load_state_dict({k[len("clf_fn."):]: v for k, v in sd.items() if k.startswith("clf_fn.")}, strict=True)
    clf = clf.to(dev).eval()
    sampler_cls = get_module("samplers", "cc_em_sde" if ev.sampler.name == "em_sde" else ev.sampler.name)
    wrapper = get_module("pl_modules", cc.model.pl_module)(root, sde, clf, score_fn=net, sampler_cls=sampler_cls)
    wrapper.global_rank = rank
    wrapper.on_predict_start()
    lo, hi = shard_range(ev.n_samples, rank, world)
    base = os.path.join(ev.save_path or "psld_cc_samples", str(ev.path_prefix)) if ev.path_prefix else (ev.save_path or "psld_cc_samples")
    out_dir = os.path.join(base, "images")
    os.makedirs(out_dir, exist_ok=True)
    size, ch = cfg.data.image_size, cfg.data.num_channels
    for bi, start in enumerate(range(lo, hi, ev.batch_size)):
        b = min(ev.batch_size, hi - start)
        x = wrapper.predict_step(sde.prior_sampling((b, ch, size, size), device=dev), bi)
        u8 = ops.samples_to_uint8(x.contiguous(), is_augmented=cfg.model.sde.is_augmented, denorm=cfg.data.norm)
        _save_u8(os.path.join(out_dir, f"output_{ev.sample_prefix}_{rank}_{bi}"), u8.cpu().numpy(), ev.save_mode)
        if rank == 0:
            print(f"rank 0: {start + b - lo}/{hi - lo} samples of class {cc.evaluation.label_to_sample}", flush=True)
    if world > 1:
        dist.barrier(device_ids=[local])
        dist.destroy_process_group()


def _save_u8(stem, u8, save_mode):
    if save_mode == "image":
        try:
            from PIL import Image
            for i, im in enumerate(u8):
                Image.fromarray(im).save(stem + "_%d.png" % i, "png")
            return
        except ImportError:
            pass
    np.save(stem + ".npy", u8)


def inpaint(args, overrides):
    """main/eval/inpaint.py: images of ``--data`` with the pixels where ``--mask`` is 0 re-synthesised by the
    ``ip_em_sde`` sampler.  Masks: uint8/bool [N,H,W,3] (1 = keep; the reference derives them from MNIST digits,
    datasets/inpaint.py:34-41) or, with ``--mask synthetic``, a centred square hole.  Writes images/, corrupt/ and
    batch/ like InpaintingImageWriter(save_batch=True) (callbacks.py:155-215)."""
    from psld_amd import config as C, ops
    from psld_amd.ddp import init_distributed, shard_range
    from psld_amd.registry import get_module
    import torch.distributed as dist
    rank, local, world = init_distributed()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cfg = parse_overrides(getattr(C, args.config)(), overrides)
    ev = cfg.evaluation
    score_fn, ema, sde = build(cfg, dev)
    if ev.chkpt_path:
        load_checkpoint(ev.chkpt_path, score_fn, ema)
        score_fn.to(dev), ema.to(dev)
    score_fn.eval(), ema.eval()
    wrapper = get_module("pl_modules", cfg.model.pl_module)(cfg, sde, score_fn, ema_score_fn=ema,
                                                            sampler_cls=get_module("samplers", "ip_em_sde"))
    wrapper.global_rank = rank
    wrapper.on_predict_start()
    data = _dataset(cfg, args, dev, 0)
    size = cfg.data.image_size
    n = min(ev.n_samples, data.shape[0])                                  # datasets/inpaint.py:43-44
    if args.mask and args.mask != "synthetic":
        marr = np.load(args.mask, mmap_mode="r")
        assert marr.shape[1:] == (size, size, 3), marr.shape
        masks = torch.from_numpy(np.ascontiguousarray(marr[:n])).to(dev).ne(0)
    else:
        masks = torch.ones((n, size, size, 3), dtype=torch.bool, device=dev)
        masks[:, size // 4: 3 * size // 4, size // 4: 3 * size // 4] = False
    lo, hi = shard_range(n, rank, world)
    base = os.path.join(ev.save_path or "psld_inpaint", str(ev.path_prefix)) if ev.path_prefix else (ev.save_path or "psld_inpaint")
    dirs = {k: os.path.join(base, k) for k in ("images", "corrupt", "batch")}
    for d in dirs.values():
        os.makedirs(d, exist_ok=True)
    for bi, start in enumerate(range(lo, hi, ev.batch_size)):
        stop = min(start + ev.batch_size, hi)
        x0 = ops.uint8_to_images(data[start:stop].contiguous(), norm=cfg.data.norm)
        mask = masks[start:stop].permute(0, 3, 1, 2).contiguous().to(torch.long)
        x = wrapper.predict_step((x0, mask), bi)
        name = f"output_{ev.sample_prefix}_{rank}_{bi}"
        u8 = ops.samples_to_uint8(x.contiguous(), is_augmented=cfg.model.sde.is_augmented, denorm=cfg.data.norm)
        _save_u8(os.path.join(dirs["images"], name), u8.cpu().numpy(), ev.save_mode)
        img01 = (x0 * 0.5 + 0.5) if cfg.data.norm else x0                   # callbacks.py:198
        to_u8 = lambda t: (t.clamp(0, 1) * 255).round().to(torch.uint8).permute(0, 2, 3, 1).cpu().numpy()
        _save_u8(os.path.join(dirs["corrupt"], name), to_u8(img01 * mask), ev.save_mode)
        _save_u8(os.path.join(dirs["batch"], name), to_u8(img01), ev.save_mode)
        if rank == 0:
            print(f"rank 0: inpainted {stop - lo}/{hi - lo}", flush=True)
    if world > 1:
        dist.barrier(device_ids=[local])
        dist.destroy_process_group()


def main(argv=None):
    ap = argparse.ArgumentParser(prog="psld_amd.cli")
    sub = ap.add_subparsers(dest="cmd", required=True)
    for name in ("train", "sample", "inpaint", "train_clf", "cc_sample"):
        p = sub.add_parser(name)
        p.add_argument("--config", default="c10_sota", choices=["c10_sota", "celeba64_sota", "yaml_default", "tiny"])
        p.add_argument("--data", default="synthetic", help="uint8 [N,H,W,3] .npy file or 'synthetic'")
        p.add_argument("--synthetic-size", type=int, default=2048)
        p.add_argument("--max-steps", type=int, default=0)
        p.add_argument("--log-every", type=int, default=10)
        p.add_argument("--mask", default="synthetic", help="inpaint: uint8 [N,H,W,3] .npy (1 = keep) or 'synthetic'")
        p.add_argument("--clf-config", default="clf_c10", choices=["clf_c10", "clf_default", "tiny_clf"])
        p.add_argument("--labels", default="synthetic", help="train_clf: int [N] .npy or 'synthetic'")
    args, overrides = ap.parse_known_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("psld_amd needs an MI355X: there is no CPU fallback")
    {"train": train, "sample": sample, "inpaint": inpaint, "train_clf": train_clf,
     "cc_sample": cc_sample}[args.cmd](args, overrides)


if __name__ == "__main__":
    main(sys.argv[1:])
