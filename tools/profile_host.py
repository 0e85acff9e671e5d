"""cProfile of the host side of one training step at a tiny batch (GPU time negligible)."""
import copy, cProfile, os, pstats, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psld_amd
from psld_amd import config as C
from psld_amd.optim import EMAWeightUpdate
from psld_amd.registry import get_module
psld_amd.import_modules_into_registry()
dev = torch.device("cuda")
cfg = C.c10_sota()
net = get_module("score_fn", "ncsnpp")(cfg).to(dev).train()
ema = copy.deepcopy(net)
sde = get_module("sde", "psld")(cfg)
crit = get_module("losses", "psld_score_loss")(cfg, sde)
wr = get_module("pl_modules", "sde_wrapper")(cfg, sde, net, ema_score_fn=ema, criterion=crit)
cb = EMAWeightUpdate(cfg.training.ema_decay)
x = torch.rand(2, 3, 32, 32, device=dev) * 2 - 1
for i in range(3):
    wr.training_step(x, i); cb.on_train_batch_end(None, wr)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for i in range(5):
    wr.training_step(x, i); cb.on_train_batch_end(None, wr)
torch.cuda.synchronize()
print("ms/step at B=2:", (time.perf_counter() - t0) / 5 * 1e3)
pr = cProfile.Profile()
pr.enable()
for i in range(3):
    wr.training_step(x, i); cb.on_train_batch_end(None, wr)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
