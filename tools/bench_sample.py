"""EM sampling throughput (network evaluations/s) eager vs HIP-graph replay of the forward."""
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
for batch in ((512,) if os.environ.get("ONLY512") else (16, 64, 512)):
    x = sde.prior_sampling((batch, 3, 32, 32), device=dev)
    for graphs in (False, True):
        net.enable_graphs(graphs)
        steps = 20 if batch <= 64 else 4
        sampler.sample(x, ts[:3], 2, denoise=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sampler.sample(x, ts[: steps + 1], steps, denoise=False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print(f"B={batch:4d} graphs={int(graphs)}  {dt*1e3:8.2f} ms/EM step  {batch/dt:8.1f} evals/s  {76.46e9*batch/dt/1e12:6.1f} TF")
