import sys, time, copy
sys.path.insert(0, '.')
import torch, psld_amd
from psld_amd import config as C
from psld_amd.optim import FusedAdam, EMAWeightUpdate
from psld_amd.registry import get_module
psld_amd.import_modules_into_registry()
dev = torch.device("cuda")
cfg = C.c10_sota()
torch.manual_seed(0)
net = get_module("score_fn", "ncsnpp")(cfg).to(dev).train()
ema = copy.deepcopy(net)
sde = get_module("sde", "psld")(cfg)
crit = get_module("losses", "psld_score_loss")(cfg, sde)
wr = get_module("pl_modules", "sde_wrapper")(cfg, sde, net, ema_score_fn=ema, criterion=crit)
cb = EMAWeightUpdate(cfg.training.ema_decay)
losses = []
mem = []
t0 = time.time()
for i in range(200):
    x = torch.rand(32, 3, 32, 32, device=dev) * 2 - 1
    loss = wr.training_step(x, i)
    cb.on_train_batch_end(None, wr)
    if i % 20 == 0:
        losses.append(float(loss)); mem.append(torch.cuda.memory_allocated() >> 20)
torch.cuda.synchronize()
print("steps/s", 200 / (time.time() - t0))
print("loss", [round(l, 4) for l in losses])
print("MiB", mem, "peak", torch.cuda.max_memory_allocated() >> 20)
ema.eval()
with torch.no_grad():
    y = ema(torch.randn(8, 6, 32, 32, device=dev), torch.rand(8, device=dev))
print("ema out finite", bool(torch.isfinite(y).all()))
