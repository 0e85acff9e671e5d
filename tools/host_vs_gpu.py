"""Is a small-batch training step bound by the host (Python + launches) or by the GPU?  Times N steps twice: until the
host has ENQUEUED them (no synchronisation) and until the GPU has finished them.
    python tools/host_vs_gpu.py [--batch 16] [--steps 30]"""
import argparse
import copy
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psld_amd  # noqa: E402
from psld_amd import config as C, ops  # noqa: E402
from psld_amd.optim import EMAWeightUpdate  # noqa: E402
from psld_amd.registry import get_module  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--graphs", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    psld_amd.import_modules_into_registry()
    ops.lib()
    cfg = C.c10_sota()
    cfg.training.batch_size = args.batch
    torch.manual_seed(0)
    net = get_module("score_fn", "ncsnpp")(cfg).to(dev).train()
    ema = copy.deepcopy(net)
    sde = get_module("sde", "psld")(cfg)
    crit = get_module("losses", "psld_score_loss")(cfg, sde)
    wrapper = get_module("pl_modules", "sde_wrapper")(cfg, sde, net, ema_score_fn=ema, criterion=crit)
    cb = EMAWeightUpdate(cfg.training.ema_decay)
    if args.graphs:
        wrapper.enable_graphs(True)
    x = torch.rand(args.batch, 3, 32, 32, device=dev) * 2 - 1
    for i in range(5):
        wrapper.training_step(x, i)
        cb.on_train_batch_end(None, wrapper)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        wrapper.training_step(x, i)
        cb.on_train_batch_end(None, wrapper)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    # the host's own cost of issuing ONE step (queue empty: no back-pressure from the launch queue, which holds fewer
    # packets than 30 steps and makes the back-to-back figure above track the GPU time whenever the GPU is the bound)
    t_one = 0.0
    for i in range(10):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        wrapper.training_step(x, i)
        cb.on_train_batch_end(None, wrapper)
        t_one += time.perf_counter() - t1
    torch.cuda.synchronize()
    print(f"batch {args.batch}: host cost of one step issued into an empty queue {t_one / 10 * 1e3:.2f} ms")
    print(f"batch {args.batch}{' graph' if args.graphs else ''}: host enqueue {t_host / args.steps * 1e3:.2f} ms/step, finished {t_all / args.steps * 1e3:.2f} ms/step "
          f"-> {'host' if t_host > 0.9 * t_all else 'GPU'}-bound")


if __name__ == "__main__":
    main()
