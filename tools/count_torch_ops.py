"""Counts the ATen ops PyTorch itself executes during one training step (plumbing audit):
everything else is libpsld_hip kernels."""
import copy, os, sys
import torch
import torch.utils._python_dispatch as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psld_amd
from psld_amd import config as C
from psld_amd.optim import EMAWeightUpdate
from psld_amd.registry import get_module

class Counter(pd.TorchDispatchMode):
    def __init__(self):
        super().__init__(); self.c = {}
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        self.c[str(func)] = self.c.get(str(func), 0) + 1
        return func(*args, **(kwargs or {}))

psld_amd.import_modules_into_registry()
dev = torch.device("cuda")
cfg = C.c10_sota() if len(sys.argv) < 2 else C.tiny()
net = get_module("score_fn", "ncsnpp")(cfg).to(dev).train()
ema = copy.deepcopy(net)
sde = get_module("sde", "psld")(cfg)
crit = get_module("losses", "psld_score_loss")(cfg, sde)
wr = get_module("pl_modules", "sde_wrapper")(cfg, sde, net, ema_score_fn=ema, criterion=crit)
cb = EMAWeightUpdate(cfg.training.ema_decay)
x = torch.rand(8, 3, cfg.data.image_size, cfg.data.image_size, device=dev) * 2 - 1
wr.training_step(x, 0); cb.on_train_batch_end(None, wr)
with Counter() as c:
    wr.training_step(x, 1); cb.on_train_batch_end(None, wr)
for k, v in sorted(c.c.items(), key=lambda kv: -kv[1]):
    print(f"{v:6d}  {k}")
