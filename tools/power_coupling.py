"""establish or retire "power coupling" - does time taken out of (or added to) the bandwidth-bound kernels
come back as clock under the matrix kernels?  The C10-SOTA B=128 training step in three conditions, interleaved in ONE
process on one box, socket power and sclk sampled from sysfs at ~50 Hz by a second process:

  default   the product step
  pipe0     PSLD_GN_BWD_PIPE=0's kernels (psld_set_gn_bwd_kernel(ONE_SLAB)): the GroupNorm backward ~0.8 ms slower per step
  pad       an idle kernel of ~10 us (one workgroup asleep) after every GroupNorm apply pass: ~+1 ms of idle chip per step

per condition: ms/step, average power / sclk over the timed window, and the mean duration of wino_conv8s_kernel launches
(HIP events around every launch, a second pass).  If the MFMA kernels get FASTER when the bandwidth-bound part gets slower
or idler, the claim stands.
    python tools/power_coupling.py [--steps 40] [--reps 2]"""
import argparse
import ctypes
import glob
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _sensors(pci=None):
    """hwmon files of the GPU at PCI address ``pci`` ("0000:bb:dd.f"; None: the first card that has them)."""
    out = {}
    cards = sorted(glob.glob("/sys/class/drm/card*/device"))
    if pci:
        cards = [c for c in cards if os.path.basename(os.path.realpath(c)).lower() == pci.lower()] or cards
    for hw in [h for c in cards for h in glob.glob(os.path.join(c, "hwmon", "hwmon*"))]:
        for name in ("power1_average", "power1_input"):
            p = os.path.join(hw, name)
            if os.path.exists(p) and "power" not in out:
                out["power"] = p
        p = os.path.join(hw, "freq1_input")
        if os.path.exists(p) and "sclk" not in out:
            out["sclk"] = p
    return out


def sampler(stop, q, period, pci):
    s = _sensors(pci)
    rows = []
    while not stop.is_set():
        t = time.monotonic()
        row = [t]
        for k in ("power", "sclk"):
            try:
                with open(s[k]) as fh:
                    row.append(float(fh.read().strip()))
            except Exception:  # noqa: BLE001
                row.append(float("nan"))
        rows.append(row)
        time.sleep(period)
    q.put((s, rows))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--pad-us", type=float, default=10.0)
    args = ap.parse_args()
    import copy
    import torch
    pr = torch.cuda.get_device_properties(0)        # (no HIP context yet: the sampler is spawned, not forked)
    pci = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id) if hasattr(pr, "pci_bus_id") else None
    ctx = mp.get_context("spawn")
    stop, q = ctx.Event(), ctx.Queue()
    proc = ctx.Process(target=sampler, args=(stop, q, 0.02, pci))
    proc.start()

    import psld_amd
    from psld_amd import config as C, ops
    from psld_amd.optim import EMAWeightUpdate
    from psld_amd.registry import get_module
    import bench
    psld_amd.import_modules_into_registry()
    dev = torch.device("cuda", 0)
    cfg = C.c10_sota()
    cfg.training.batch_size = 128
    torch.manual_seed(0)
    net = get_module("score_fn", "ncsnpp")(cfg).to(dev).train()
    ema = copy.deepcopy(net)
    for p in ema.parameters():
        p.requires_grad = False
    sde = get_module("sde", "psld")(cfg)
    crit = get_module("losses", "psld_score_loss")(cfg, sde)
    wr = get_module("pl_modules", "sde_wrapper")(cfg, sde, net, ema_score_fn=ema, criterion=crit)
    cb = EMAWeightUpdate(cfg.training.ema_decay)
    g = torch.Generator(device=dev).manual_seed(0)
    data = [torch.rand(128, 3, 32, 32, device=dev, generator=g) * 2 - 1 for _ in range(4)]
    hog = ctypes.CDLL(os.path.join(ROOT, "tests", "helpers", "libcuhog.so"))
    hog.cu_hog.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
    hog.cu_hog.restype = ctypes.c_int
    scratch = torch.zeros(1024, device=dev)
    real_apply = ops.gn_apply
    pads = {"n": 0}

    def padded_apply(*a, **k):
        r = real_apply(*a, **k)
        hog.cu_hog(scratch.data_ptr(), 0, 1, 64, 0, args.pad_us, torch.cuda.current_stream().cuda_stream)
        pads["n"] += 1
        return r

    def set_mode(m):
        ops.gn_apply = padded_apply if m == "pad" else real_apply
        ops.set_gn_bwd_kernel("one_slab" if m == "pipe0" else "auto")

    def run(n, first):
        torch.cuda.synchronize()
        t0 = time.monotonic()
        for i in range(n):
            wr.training_step(data[(first + i) % 4], first + i)
            cb.on_train_batch_end(None, wr)
        torch.cuda.synchronize()
        return t0, time.monotonic()

    probe = bench.ConvProbe(ops)
    probe.install()
    run(5, 0)
    windows = []
    step = 5
    for rep in range(args.reps):
        for mode in ("default", "pipe0", "pad"):
            set_mode(mode)
            run(3, step)
            step += 3
            pads["n"] = 0
            t0, t1 = run(args.steps, step)
            step += args.steps
            n_pads = pads["n"]
            probe.reset()
            probe.enabled = True
            run(10, step)
            step += 10
            probe.enabled = False
            w = probe.summary("wino")
            windows.append({"mode": mode, "rep": rep, "t0": t0, "t1": t1, "ms_per_step": 1e3 * (t1 - t0) / args.steps,
                            "wino_avg_us": w["avg_us"], "wino_launches": w["launches"], "pads_per_step": n_pads / args.steps})
    set_mode("default")
    stop.set()
    sensors, rows = q.get(timeout=30)
    proc.join(timeout=10)
    print("sensors:", json.dumps(sensors), f"({len(rows)} samples)")
    for w in windows:
        sel = [r for r in rows if w["t0"] + 0.3 <= r[0] <= w["t1"] - 0.1]
        pw = [r[1] for r in sel if r[1] == r[1]]
        ck = [r[2] for r in sel if r[2] == r[2]]
        w["samples"] = len(sel)
        w["power_w"] = (sum(pw) / len(pw) / 1e6) if pw else None
        w["sclk_mhz"] = (sum(ck) / len(ck) / 1e6) if ck else None
        print(f"{w['mode']:8s} rep {w['rep']}: {w['ms_per_step']:7.2f} ms/step  power {w['power_w'] if w['power_w'] is None else round(w['power_w'], 1)} W  "
              f"sclk {w['sclk_mhz'] if w['sclk_mhz'] is None else round(w['sclk_mhz'])} MHz  wino_conv8s mean {w['wino_avg_us']:6.1f} us "
              f"({w['wino_launches']} launches)  idle kernels/step {w['pads_per_step']:.0f}  [{w['samples']} samples]")
    print(json.dumps({"windows": [{k: v for k, v in w.items() if k not in ("t0", "t1")} for w in windows]}))


if __name__ == "__main__":
    main()
