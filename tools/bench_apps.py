"""Per-step timings of the SURVEY 8(f) samplers on the north-star networks (C10-SOTA score net, clf_c10 classifier):
EM, SSCS, inpainting (ip_em_sde) and classifier guidance (cc_em_sde).  python tools/bench_apps.py [--batch 64]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psld_amd
from psld_amd import config as C
from psld_amd.registry import get_module

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--steps", type=int, default=8)
args = ap.parse_args()
psld_amd.import_modules_into_registry()
dev = torch.device("cuda")
torch.manual_seed(0)
dcfg, ccfg = C.c10_sota(), C.clf_c10()
root = C.with_clf(dcfg, ccfg)
root.clf.evaluation.label_to_sample, root.clf.evaluation.clf_temp = 3, 1.0
net = get_module("score_fn", "ncsnpp")(dcfg).to(dev).eval()
for p in net.parameters():
    p.requires_grad = False
clf = get_module("clf_fn", "ncsnpp_clf")(ccfg).to(dev).eval()
sde = get_module("sde", "psld")(dcfg)
B, n = args.batch, args.steps
ts = torch.linspace(0, 0.999, 1000, device=dev, dtype=torch.float64)[: n + 1]


def timed(name, fn):
    fn(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(n)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{name:34s} B={B:4d}  {dt * 1e3:8.2f} ms/step  {B / dt:8.1f} samples-steps/s")


x = sde.prior_sampling((B, 3, 32, 32), device=dev)
em = get_module("samplers", "em_sde")(dcfg, sde, net)
timed("em_sde", lambda k: em.sample(x, ts[: k + 1], k, denoise=False))
ss = get_module("samplers", "sscs_sde")(dcfg, sde, net)
timed("sscs_sde", lambda k: ss.sample(x, ts[: k + 1], k, denoise=False))
ip = get_module("samplers", "ip_em_sde")(dcfg, sde, net)
x0 = torch.rand(B, 3, 32, 32, device=dev) * 2 - 1
mask = (torch.rand(B, 3, 32, 32, device=dev) > 0.3).long()
timed("ip_em_sde (inpainting)", lambda k: ip.sample((x0, mask), ts[: k + 1], k, denoise=False))
cc = get_module("samplers", "cc_em_sde")(root, sde, net, clf)
timed("cc_em_sde (classifier guidance)", lambda k: cc.sample(x, ts[: k + 1], k, denoise=False))
with torch.no_grad():
    t32 = torch.full((B,), 0.5, device=dev)
    x32 = x.float()
    timed("classifier forward only", lambda k: [clf(x32, t32) for _ in range(k)])
