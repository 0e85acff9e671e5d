"""Where the host time of a small-batch training step goes (cProfile over N steps, top entries by own time).
    python tools/host_profile.py [--batch 16] [--steps 10]"""
import argparse
import cProfile
import copy
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psld_amd  # noqa: E402
from psld_amd import config as C, ops  # noqa: E402
from psld_amd.optim import EMAWeightUpdate  # noqa: E402
from psld_amd.registry import get_module  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--top", type=int, default=45)
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
    x = torch.rand(args.batch, 3, 32, 32, device=dev) * 2 - 1
    for i in range(3):
        wrapper.training_step(x, i)
        cb.on_train_batch_end(None, wrapper)
    torch.cuda.synchronize()
    # the backward tape runs on the autograd engine's thread: give it a profiler of its own
    from psld_amd import score_fn as SF
    pr_b = cProfile.Profile()
    orig = SF._Exec._backward

    def profiled_backward(self, g):
        pr_b.enable()
        try:
            return orig(self, g)
        finally:
            pr_b.disable()
    SF._Exec._backward = profiled_backward
    pr = cProfile.Profile()
    pr.enable()
    for i in range(args.steps):
        wrapper.training_step(x, i)
        cb.on_train_batch_end(None, wrapper)
    pr.disable()
    torch.cuda.synchronize()
    SF._Exec._backward = orig
    st = pstats.Stats(pr)
    st.add(pr_b)
    st.sort_stats("tottime")
    total = sum(v[2] for v in st.stats.values())
    print(f"host time under the profiler: {total / args.steps * 1e3:.1f} ms/step")
    rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:args.top]
    for (fn, line, name), (cc, nc, tt, ct, _) in rows:
        print(f"{tt / args.steps * 1e3:7.2f} ms  {nc // args.steps:6d} calls  {os.path.basename(fn)}:{line} {name}")


if __name__ == "__main__":
    main()
