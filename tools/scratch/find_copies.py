import copy, os, sys, traceback, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import psld_amd
from psld_amd import config as C, ops
from psld_amd.optim import EMAWeightUpdate
from psld_amd.registry import get_module
dev = torch.device("cuda", 0)
psld_amd.import_modules_into_registry(); ops.lib()
cfg = C.c10_sota(); cfg.training.batch_size = 16
net = get_module("score_fn", "ncsnpp")(cfg).to(dev).train()
ema = copy.deepcopy(net)
sde = get_module("sde", "psld")(cfg)
crit = get_module("losses", "psld_score_loss")(cfg, sde)
w = get_module("pl_modules", "sde_wrapper")(cfg, sde, net, ema_score_fn=ema, criterion=crit)
cb = EMAWeightUpdate(cfg.training.ema_decay)
x = torch.rand(16, 3, 32, 32, device=dev) * 2 - 1
for i in range(3):
    w.training_step(x, i); cb.on_train_batch_end(None, w)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    w.training_step(x, 3); cb.on_train_batch_end(None, w)
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if "copy_" in e.name or "Memcpy" in e.name or e.name in ("aten::cat", "aten::clone", "aten::contiguous"):
        st = [s for s in (e.stack or []) if "psld_amd" in s or "bench" in s][:2]
        cnt[(e.name, tuple(st))] += 1
for k, v in cnt.most_common(25):
    print(v, k)
