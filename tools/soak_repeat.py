"""Race hunt for the two-stream training step: the same seeded 150-step run (batch 16, dropout on, weight gradients on
the side stream, NaN check on the early stream) executed twice must produce bit-identical losses and parameters - every
kernel is deterministic, so any difference is a missing stream dependency.  Also reports memory growth.
    python tools/soak_repeat.py [--steps 150] [--batch 16]"""
import argparse
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psld_amd  # noqa: E402
from psld_amd import config as C  # noqa: E402
from psld_amd.optim import EMAWeightUpdate  # noqa: E402
from psld_amd.registry import get_module  # noqa: E402


def run(steps, batch, seed):
    dev = torch.device("cuda")
    cfg = C.c10_sota()
    cfg.training.batch_size = batch
    torch.manual_seed(seed)
    net = get_module("score_fn", "ncsnpp")(cfg).to(dev).train()
    ema = copy.deepcopy(net)
    sde = get_module("sde", "psld")(cfg)
    crit = get_module("losses", "psld_score_loss")(cfg, sde)
    wr = get_module("pl_modules", "sde_wrapper")(cfg, sde, net, ema_score_fn=ema, criterion=crit)
    cb = EMAWeightUpdate(cfg.training.ema_decay)
    g = torch.Generator(device=dev).manual_seed(seed + 1)
    losses = []
    for i in range(steps):
        x = torch.rand(batch, 3, 32, 32, device=dev, generator=g) * 2 - 1
        losses.append(wr.training_step(x, i).detach().clone())
        cb.on_train_batch_end(None, wr)
    torch.cuda.synchronize()
    return torch.stack(losses).cpu(), net.flatten_parameters().detach().cpu().clone(), \
        ema.flatten_parameters().detach().cpu().clone(), torch.cuda.max_memory_allocated() >> 20


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--batch", type=int, default=16)
    args = ap.parse_args()
    psld_amd.import_modules_into_registry()
    a = run(args.steps, args.batch, 7)
    b = run(args.steps, args.batch, 7)
    same = torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    print(f"batch {args.batch}, {args.steps} steps twice: losses / parameters / EMA bitwise equal: {same}; "
          f"loss {float(a[0][0]):.4f} -> {float(a[0][-1]):.4f}, finite {bool(torch.isfinite(a[0]).all())}; peak {a[3]} / {b[3]} MiB")
    if not same:
        first = int((a[0] != b[0]).nonzero()[0]) if (a[0] != b[0]).any() else -1
        print("first differing step:", first)
        sys.exit(1)


if __name__ == "__main__":
    main()
