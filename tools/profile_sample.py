"""Forward-only (EM sampling) kernel mix at B=512: run under `rocprofv3 --kernel-trace --stats`."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psld_amd
from psld_amd import config as C
from psld_amd.registry import get_module
psld_amd.import_modules_into_registry()
dev = torch.device("cuda")
cfg = C.c10_sota()
torch.manual_seed(0)
net = get_module("score_fn", "ncsnpp")(cfg).to(dev).eval()
sde = get_module("sde", "psld")(cfg)
sampler = get_module("samplers", "em_sde")(cfg, sde, net)
ts = torch.linspace(0, 0.999, 1000, device=dev, dtype=torch.float64)
batch, steps = 512, 6
x = sde.prior_sampling((batch, 3, 32, 32), device=dev)
sampler.sample(x, ts[:3], 2, denoise=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
sampler.sample(x, ts[: steps + 1], steps, denoise=False)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"B={batch} {dt*1e3:.2f} ms/EM step {batch/dt:.1f} evals/s")
