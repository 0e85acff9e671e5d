#!/usr/bin/env python
"""Headline benchmark: HSM training images/sec of the CIFAR-10 PSLD NCSN++ (C10-SOTA) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 128]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` with N > 1 and no launcher in the environment (WORLD_SIZE unset) starts the N ranks itself: the parent
spawns `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process before it has touched the
GPU and exits with the child's code.  A world size that differs from --gpus is an error, never a silent 1-rank run.

One step = the reference's full training step (main/models/wrapper.py:64-91 + callbacks.py:42-64):
t ~ U[eps,1] -> PSLD perturb -> NCSN++ forward -> HSM loss -> backward (-> RCCL bucketed all-reduce
when N>1) -> global-norm clip -> Adam -> LambdaLR -> EMA, dropout 0.15 on, fp32, per-GPU batch 128,
synthetic CIFAR-shaped data resident in HBM.  Prints ONE JSON line (rank 0).

roofline: the dominant kernel is dconv_kernel (csrc/conv_split.hip): the direct 3x3 convolution (forward and
data gradient) on bf16 limb MFMA ("bf16x6": fp32 operands split exactly into three bf16 limbs, six
v_mfma_f32_16x16x32_bf16 products per fp32 product, fp32 accumulate).  Every launch in the timed region is
bracketed by HIP events on its own stream; achieved = sum(2*M*N*K) / sum(duration) in fp32-equivalent
(algorithmic) TFLOP/s, peak = dense bf16 MFMA peak / 6 limb products.  cpu_baseline: the CPU oracle (a port of the reference's
CPU path, pinned to its golden vectors) timed on this host for one B=16 train step (N=1 only).
"""
from __future__ import annotations

import argparse
import copy
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 matrix peak
LIMB_PRODUCTS = 6               # bf16 MFMA products per fp32 product in PSLD_MATH_BF16X6
SUSTAINED_BF16_MFMA_TFLOPS = 1900.0   # measured: tools/mfma_peak.hip, 16x16x32 bf16 MFMA on random operands (power-limited)


class ConvProbe:
    """HIP-event timing of every launch of the 3x3 limb-MFMA convolution (ops.conv3x3_split: dconv_kernel, forward
    and data gradient) and, separately, of the fp32-MFMA tile-engine convolutions (ops.conv2d_nhwc)."""

    def __init__(self, ops):
        self.ops = ops
        self.orig_split = ops.conv3x3_split
        self.orig_tile = ops.conv2d_nhwc
        self.orig_wino = ops.conv3x3_wino
        self.orig_wgrad = ops.conv3x3_wgrad_split
        self.orig_wgrad_wino = ops.conv3x3_wgrad_wino
        self.records = {"split": [], "tile": [], "wino": [], "wgrad": [], "wgrad_wino": []}
        self.enabled = False

    def install(self):
        probe = self

        def timed(kind, flops, fn):
            if not probe.enabled:
                return fn()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            fn()
            e.record()
            probe.records[kind].append((s, e, flops))

        def split(x1, x2, wfrag, cout, y, epi=None, ldy=None):
            cin = x1.shape[-1] + (x2.shape[-1] if x2 is not None else 0)
            flops = 2.0 * x1.shape[0] * x1.shape[1] * x1.shape[2] * cout * 9 * cin
            timed("split", flops, lambda: probe.orig_split(x1, x2, wfrag, cout, y, epi, ldy))

        def wino(x1, x2, ufrag, cout, y, epi=None, ldy=None, **kw):
            cin = x1.shape[-1] + (x2.shape[-1] if x2 is not None else 0)
            flops = 2.0 * x1.shape[0] * x1.shape[1] * x1.shape[2] * cout * 9 * cin      # direct-convolution (algorithmic) count
            timed("wino", flops, lambda: probe.orig_wino(x1, x2, ufrag, cout, y, epi, ldy, **kw))

        def tile(x1, x2, w, cout, kh, kw, stride, pad, tstride, oh, ow, y, epi=None, ldy=None):
            cin = x1.shape[-1] + (x2.shape[-1] if x2 is not None else 0)
            flops = 2.0 * x1.shape[0] * oh * ow * cout * kh * kw * cin
            if tstride > 1:
                flops /= tstride * tstride      # structural zeros of the strided data-gradient
            timed("tile", flops, lambda: probe.orig_tile(x1, x2, w, cout, kh, kw, stride, pad, tstride, oh, ow, y,
                                                         epi, ldy))

        def wgrad(dy, cout, x, slabs, cin_total, col0, nsplit, x2=None):
            cin = x.shape[-1] + (x2.shape[-1] if x2 is not None else 0)
            flops = 2.0 * x.shape[0] * x.shape[1] * x.shape[2] * cout * 9 * cin
            timed("wgrad", flops, lambda: probe.orig_wgrad(dy, cout, x, slabs, cin_total, col0, nsplit, x2))

        def wgrad_wino(dy, cout, x, dw, x2=None, **kw):
            cin = x.shape[-1] + (x2.shape[-1] if x2 is not None else 0)
            flops = 2.0 * x.shape[0] * x.shape[1] * x.shape[2] * cout * 9 * cin      # direct-convolution (algorithmic) count
            timed("wgrad_wino", flops, lambda: probe.orig_wgrad_wino(dy, cout, x, dw, x2=x2, **kw))

        self.ops.conv3x3_split = split
        self.ops.conv3x3_wino = wino
        self.ops.conv2d_nhwc = tile
        self.ops.conv3x3_wgrad_split = wgrad
        self.ops.conv3x3_wgrad_wino = wgrad_wino

    def reset(self):
        self.records = {k: [] for k in self.records}

    def summary(self, kind):
        recs = self.records[kind] if isinstance(kind, str) else [r for k in kind for r in self.records[k]]
        if not recs:
            return None
        ms = sum(s.elapsed_time(e) for s, e, _ in recs)
        fl = sum(f for _, _, f in recs)
        return {"launches": len(recs), "total_ms": ms, "total_flop": fl,
                "avg_us": 1e3 * ms / len(recs), "tflops": fl / (ms * 1e-3) / 1e12}


def physical_cores():
    """One logical CPU per physical core among the CPUs this process may run on (SMT siblings dropped).  Topology from
    /proc/cpuinfo (physical id, core id); when that is not conclusive (containers mask it) every allowed CPU is kept."""
    allowed = sorted(os.sched_getaffinity(0))
    ids = {}
    try:
        cpu = None
        phys = core = None
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.startswith("processor"):
                    cpu, phys, core = int(ln.split(":")[1]), None, None
                elif ln.startswith("physical id"):
                    phys = int(ln.split(":")[1])
                elif ln.startswith("core id"):
                    core = int(ln.split(":")[1])
                if cpu is not None and phys is not None and core is not None:
                    ids[cpu] = (phys, core)
                    cpu = None
    except (OSError, ValueError):
        ids = {}
    seen, keep = set(), []
    for c in allowed:
        key = ids.get(c, ("cpu", c))
        if key not in seen:
            seen.add(key)
            keep.append(c)
    if len(keep) * 4 < len(allowed):      # fewer than a quarter survive: the topology is not believable
        keep = allowed
    # a container's CPU quota (cgroup v2 cpu.max / v1 cfs_quota) bounds what threads can actually run: more threads than
    # that only spin against the throttle (a 256-CPU box with a 16-CPU quota ran 128 threads 20x slower than 16)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, per = fh.read().split()[:2]
            if q != "max":
                quota = int(q) / int(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, per = int(fq.read()), int(fp.read())
                if q > 0:
                    quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None and quota >= 1:
        keep = keep[:max(1, int(quota))]
    return keep


def cpu_baseline(cfg, batch=16, timed=3, budget_s=150.0):
    """Oracle (kind 'port') on the host cores, configs[0] exactly (B=16, fp32, one process): threads pinned one per
    physical core, 1 warm-up + `timed` full HSM train steps (median) and the eval-mode forward."""
    from oracle import psld_oracle as O
    from psld_amd.score_fn import NCSNpp
    from tests.synth import synth_inputs
    cores = physical_cores()
    os.sched_setaffinity(0, cores)
    os.environ["OMP_NUM_THREADS"] = str(len(cores))
    torch.set_num_threads(len(cores))
    torch.manual_seed(0)
    net = NCSNpp(cfg)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    sde = O.PSLDOracle.from_config(cfg)
    x0, eps, t = synth_inputs(batch, 3, cfg.data.image_size, seed=0)
    g = torch.Generator().manual_seed(1)
    p = cfg.model.score_fn.dropout
    masks = None
    if p > 0:
        # one pre-scaled Bernoulli mask per ResBlock (shape of its GroupNorm_1 output)
        shapes = []
        hook_sd = sd
        with torch.no_grad():
            rec = []
            orig = O.resblock_biggan

            def spy(x, temb, sdd, pfx, **kw):
                out = orig(x, temb, sdd, pfx, **{k: v for k, v in kw.items() if k != "dropout_mask"})
                rec.append(out.shape)
                return out

            O.resblock_biggan = spy
            try:
                O.ncsnpp_forward(hook_sd, cfg, torch.zeros(1, 6, cfg.data.image_size, cfg.data.image_size),
                                 torch.ones(1) * 0.5)
            finally:
                O.resblock_biggan = orig
            shapes = rec
        masks = [(torch.rand(batch, *s[1:], generator=g) >= p).float() / (1 - p) for s in shapes]
    ema_sd = {k: v.clone() for k, v in sd.items()}
    opt_state = {}
    times = []
    t_begin = time.perf_counter()
    for i in range(1 + timed):                 # step 0 = warm-up (allocator, thread pool, oneDNN primitive caches)
        t0 = time.perf_counter()
        O.train_step(sde, sd, cfg, x0, t, eps, opt_state, i + 1, ema_sd=ema_sd, dropout_masks=masks)
        times.append(time.perf_counter() - t0)
        print(f"cpu_baseline: step {i} {times[-1]:.1f} s on {len(cores)} threads", file=sys.stderr, flush=True)
        if i >= 1 and time.perf_counter() - t_begin > budget_s:      # bounded sample: at least one timed step
            break
    steps = sorted(times[1:])
    dt = steps[len(steps) // 2]
    z = torch.randn(batch, 6, cfg.data.image_size, cfg.data.image_size)
    tt = torch.rand(batch) * 0.98 + 0.01
    fwd = []
    with torch.no_grad():
        for i in range(3):
            t0 = time.perf_counter()
            O.ncsnpp_forward(sd, cfg, z, tt)
            fwd.append(time.perf_counter() - t0)
    fdt = sorted(fwd[1:])[0]
    model = ""
    try:
        with open("/proc/cpuinfo") as fh:
            model = next((ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name")), "")
    except OSError:
        pass
    return {"value": batch / dt, "unit": "images/s", "cores": len(cores), "kind": "port",
            "eval_forward_images_per_s": batch / fdt,
            "train_step_s": [round(x, 2) for x in times], "host": model, "torch": torch.__version__,
            "sample": f"{len(times) - 1} HSM train steps after 1 warm-up (median; fwd+bwd+clip+Adam+EMA), B={batch}, C10-SOTA "
                      f"NCSN++, fp32, {dt:.1f} s/step on {len(cores)} threads pinned to physical cores; eval forward "
                      f"{fdt:.2f} s (best of 2 after 1 warm-up)"}


def sampling_run(cfg, ema_net, sde, dev, batch, n_discrete_steps):
    """Second half of BASELINE.json's metric, measured: ONE full per-GPU batch of configs[4] end to end —
    prior sampling -> (n_discrete_steps - 1) EM predictor steps + the denoising step on the EMA network at B=512
    (main/eval/sample.py:59-109 -> wrapper.py:101-122 -> samplers/sde.py:38-58) -> uint8 conversion of the position
    half (callbacks.py:103-107, util.py:147-158) -> host.  50 000 samples on 8 GPUs = ceil(50000 / (8*512)) = 13 such
    batches per GPU (the last one partial), no collective."""
    from psld_amd import ops
    from psld_amd.registry import get_module
    ema_net.eval()
    cfg.evaluation.n_discrete_steps = n_discrete_steps
    wrapper = get_module("pl_modules", "sde_wrapper")(cfg, sde, ema_net, ema_score_fn=ema_net,
                                                      sampler_cls=get_module("samplers", "em_sde"))
    with torch.no_grad():
        warm = sde.prior_sampling((batch, 3, 32, 32), device=dev)
        wrapper.sampler.sample(warm, wrapper.sampling_times(dev)[:3], 2, denoise=True, eps=cfg.evaluation.eval_eps)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x = sde.prior_sampling((batch, 3, 32, 32), device=dev)
        xs = wrapper.predict_step(x, 0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        u8 = ops.samples_to_uint8(xs, is_augmented=True)
        host = u8.cpu()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    ema_net.train()
    assert host.shape[0] == batch and host.dtype == torch.uint8
    batch_s = t2 - t0
    evals = n_discrete_steps                      # n-1 predictor steps + 1 denoising step, one network call each
    batches = -(-50000 // (8 * batch))
    full = n_discrete_steps == 1000
    return {"batch_per_gpu": batch, "n_discrete_steps": n_discrete_steps, "network_evals": evals,
            "measured_batch_s": batch_s, "sampler_s": t1 - t0, "uint8_and_d2h_s": t2 - t1,
            "finite": bool(torch.isfinite(xs).all()),
            "network_evals_per_s": batch * evals / (t1 - t0), "ms_per_em_step": 1e3 * (t1 - t0) / evals,
            "fwd_tflops": 76.46e9 * batch * evals / (t1 - t0) / 1e12,
            "per_gpu_50k_samples_8gpu_s": batches * batch_s * (1000.0 / n_discrete_steps),
            "note": ("ONE full batch measured end to end on 1 GPU (prior -> 999 EM steps + denoise -> uint8 -> host); "
                     if full else
                     f"SHORTENED run ({n_discrete_steps} of 1000 discretisation steps) scaled by 1000/{n_discrete_steps}; ") +
                    f"per_gpu_50k = {batches} batches/GPU x measured_batch_s: each of 8 GPUs samples its own shard, no "
                    "collective (SURVEY 8(e)); the last batch of the real run is partial (50000 - 12*4096 = 848 samples "
                    "over 8 GPUs)"}


def sampling_shard(rank: int, world: int, seed: int, n_samples: int = 50000, batch: int = 512):
    """What rank ``rank`` of ``world`` samples in the 50 000-sample run (main/eval/sample.py:100-109: Trainer.predict
    shards the latent dataset over the ranks; wrapper.py:93-99: RNG seed = evaluation.seed + global_rank): its
    contiguous shard, the number of per-GPU batches it takes and its seed.  No collective on the data path."""
    from psld_amd.ddp import shard_range
    lo, hi = shard_range(n_samples, rank, world)
    return {"rank": rank, "seed": seed + rank, "lo": lo, "hi": hi, "batches": -(-(hi - lo) // batch)}


def reduce_sampling(res, rank: int, world: int, on_dev: bool, dev, shard):
    """All ranks sampled one batch of their own concurrently; rank 0 reports the SLOWEST rank (max over ranks, like the
    training timing) and the figure for the whole 50 000-sample run: batches-per-rank x that time."""
    import torch.distributed as dist
    where = dev if on_dev else "cpu"
    mx = torch.tensor([res["measured_batch_s"], res["sampler_s"]], device=where, dtype=torch.float64)
    mn = mx.clone()
    ok = torch.tensor([1.0 if res["finite"] else 0.0], device=where, dtype=torch.float64)
    seeds = torch.zeros(world, device=where, dtype=torch.float64)
    seeds[rank] = shard["seed"]
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    dist.all_reduce(mn, op=dist.ReduceOp.MIN)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    dist.all_reduce(seeds, op=dist.ReduceOp.SUM)
    out = dict(res)
    scale = float(mx[0]) / res["measured_batch_s"]
    out.update({"n_gpus": world, "measured_batch_s": float(mx[0]), "sampler_s": float(mx[1]),
                "measured_batch_s_min_over_ranks": float(mn[0]), "finite": bool(ok.item() == 1.0),
                "rank_seeds": [int(v) for v in seeds.tolist()], "batches_per_rank": shard["batches"],
                "network_evals_per_s": world * res["network_evals_per_s"] / scale,
                "fwd_tflops": res["fwd_tflops"] / scale})
    steps_scale = 1000.0 / res["n_discrete_steps"]
    out["wallclock_50k_samples_s"] = shard["batches"] * float(mx[0]) * steps_scale
    how = f"batches per rank ({shard['batches']}) x that"
    if res.get("partial_batch"):       # the shard's last batch is partial and was measured too: max over ranks of it
        pm = torch.tensor([res["partial_batch"]["measured_batch_s"]], device=where, dtype=torch.float64)
        dist.all_reduce(pm, op=dist.ReduceOp.MAX)
        out["partial_batch"] = dict(res["partial_batch"], measured_batch_s=float(pm[0]))
        out["wallclock_50k_samples_s"] = ((shard["batches"] - 1) * float(mx[0]) + float(pm[0])) * steps_scale
        how = (f"{shard['batches'] - 1} full batches x that + the measured partial batch of "
               f"{res['partial_batch']['batch']} samples (max over ranks)")
    out.pop("per_gpu_50k_samples_8gpu_s", None)
    out["note"] = (f"every one of the {world} ranks sampled ONE batch of its own shard concurrently (seed + rank, no "
                   f"collective); measured_batch_s = max over ranks; wallclock_50k_samples_s = {how}, scaled to 1000 "
                   "discretisation steps if the run was shortened")
    return out


def _guarded_record(name):
    """A committed profile record (the newest of profiles/r06 ... r03/<name>) that carries the hash of the kernel
    sources it was measured on; a record measured on other sources is refused."""
    import hashlib
    rec, path = None, None
    for rnd in ("r06", "r05", "r04", "r03"):
        path = os.path.join(ROOT, "profiles", rnd, name)
        try:
            with open(path) as fh:
                rec = json.load(fh)
            break
        except Exception:  # noqa: BLE001
            continue
    if rec is None:
        return None, f"no record (profiles/r06/{name})"
    h = hashlib.sha256()
    for f in rec.get("sources", []):
        try:
            with open(os.path.join(ROOT, f), "rb") as fh:
                h.update(fh.read())
        except OSError:
            return None, f"{path} names a missing source file {f}"
    if h.hexdigest() != rec.get("sources_sha256"):
        return None, f"{os.path.relpath(path, ROOT)} is stale: the kernel sources changed since it was measured"
    rec["_path"] = os.path.relpath(path, ROOT)
    return rec, None


def pmc_traffic_record():
    """HBM bytes per launch of the dominant convolution from the committed PMC passes (pmc_traffic.json, written by
    tools/pmc_traffic.py from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs)."""
    return _guarded_record("pmc_traffic.json")


def hbm_in_situ_record():
    """Byte-weighted HBM rate of the bandwidth-bound kernels INSIDE the B=128 training step (hbm_in_situ.json, written by
    tools/hbm_in_situ.py: algorithmic bytes from the launch arguments / rocprofv3 kernel durations of the same process)."""
    return _guarded_record("hbm_in_situ.json")


def hbm_in_situ_forward_record():
    """The same for the EVAL FORWARD at B=128 alone - the figure north_star names ("the U-Net forward at batch
    128x6x32x32"): hbm_in_situ_forward.json, written by `tools/hbm_in_situ.py run-forward` / `join`."""
    return _guarded_record("hbm_in_situ_forward.json")


# algorithmic work of one eval forward per image (measured with torch.utils.flop_counter on the reference) and the
# fused plan's compulsory activation bytes per image where SURVEY states them (C10-SOTA only); weights are read once per batch
FORWARD_WORK = {"c10_sota": {"gflop": 76.46, "act_bytes": 173.2e6, "weight_bytes": 390.5e6},
                "celeba64_sota": {"gflop": 84.17, "act_bytes": None, "weight_bytes": 251.1e6}}


def forward_block(net, dev, batch, size, steps=5, config="c10_sota"):
    """SURVEY 8(d) 'Which roofline': the end-to-end eval forward at the training batch on BOTH axes - algorithmic FLOPs
    against the 416.7 TFLOP/s ceiling of the limb algorithm and the fused plan's compulsory bytes (176.3 MB per image at
    B=128) against 8 TB/s - timed with HIP events on the launch stream."""
    was_training = net.training
    net.eval()
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(batch, 6, size, size, device=dev, generator=g)
    t = torch.rand(batch, device=dev, generator=g) * 0.98 + 0.01
    with torch.no_grad():
        for _ in range(2):
            net(x, t)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            net(x, t)
        e1.record()
        torch.cuda.synchronize()
    net.train(was_training)
    dt = e0.elapsed_time(e1) * 1e-3 / steps
    work = FORWARD_WORK[config]
    fl = work["gflop"] * 1e9 * batch
    by = (work["act_bytes"] + work["weight_bytes"] / batch) * batch if work["act_bytes"] is not None else None
    return {"config": config, "batch": batch, "ms": dt * 1e3, "images_per_s": batch / dt, "tflops": fl / dt / 1e12,
            "frac_of_limb_mfma_ceiling": fl / dt / 1e12 / (PEAK_BF16_MFMA_TFLOPS / LIMB_PRODUCTS),
            "compulsory_bytes": by, "gb_per_s": by / dt / 1e9 if by else None, "frac_of_hbm_8tbs": by / dt / 8.0e12 if by else None,
            "note": f"eval forward of {config}, HIP events; {work['gflop']} GFLOP per image (SURVEY 8(d))" +
                    (f" and {work['act_bytes'] / 1e6:.1f} MB + weights / B of compulsory traffic per image: the forward is compute-bound "
                     "(434 FLOP/B), the HBM line is met per bandwidth-bound kernel (roofline.hbm_bound_*)" if by else
                     "; SURVEY states no fused-plan byte count for this configuration: FLOP axis only")}


def forward_hbm_live(net, dev, batch, size, steps=3):
    """The forward's bandwidth-bound line measured IN THIS RUN (the committed record is builder-side):
    every bandwidth-bound entry point of `steps` eval forwards is bracketed by HIP events on the launch stream and sized
    from its arguments with tools/hbm_in_situ.py's byte formulas (distinct inputs read once + outputs written once).  An
    event pair around a ~5 us launch also times the gap to its neighbours, so this reads a little LOWER than the rocprofv3
    record (kernel durations only) - it is a floor, measured by the driver's own process."""
    from psld_amd import _lib
    from tools.hbm_in_situ import BYTES
    real = _lib.load_real()
    recs = []

    class _Timed:
        def __getattr__(self, name):
            fn = getattr(real, name)
            ent = BYTES.get(name)
            if ent is None:
                return fn
            fam, calc = ent

            def timed(*args):
                vals = [int(v.value if hasattr(v, "value") and v.value is not None else 0) if hasattr(v, "value") else (v if v is not None else 0)
                        for v in args]
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                r = fn(*args)
                e.record()
                try:
                    recs.append((fam, int(calc(vals)), s, e))
                except Exception:  # noqa: BLE001
                    pass
                return r
            return timed

    was_training = net.training
    net.eval()
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(batch, 6, size, size, device=dev, generator=g)
    t = torch.rand(batch, device=dev, generator=g) * 0.98 + 0.01
    try:
        with torch.no_grad():
            net(x, t)
            torch.cuda.synchronize()
            _lib.set_proxy(_Timed())
            for _ in range(steps):
                net(x, t)
            torch.cuda.synchronize()
    finally:
        _lib.set_proxy(None)
        net.train(was_training)
    fam_b, fam_t, fam_n = {}, {}, {}
    for fam, b, s, e in recs:
        fam_b[fam] = fam_b.get(fam, 0) + b
        fam_t[fam] = fam_t.get(fam, 0.0) + s.elapsed_time(e) * 1e-3
        fam_n[fam] = fam_n.get(fam, 0) + 1
    tb, tt = sum(fam_b.values()), sum(fam_t.values())
    if tt <= 0:
        return None
    return {"hbm_bound_aggregate_frac_live": tb / tt / 8.0e12, "gb_per_s": tb / tt / 1e9, "bytes_per_forward": tb / steps,
            "ms_per_forward": 1e3 * tt / steps, "launches_per_forward": len(recs) / steps,
            "families": {f: {"launches": fam_n[f] / steps, "mb": fam_b[f] / steps / 1e6, "ms": 1e3 * fam_t[f] / steps,
                             "frac_of_8tbs": fam_b[f] / fam_t[f] / 8.0e12} for f in sorted(fam_b, key=lambda k: -fam_t[k])},
            "note": "HIP events around every bandwidth-bound entry point of the eval forward in THIS run (algorithmic bytes / event time, "
                    "launch gaps included: a floor under the rocprofv3-based hbm_bound_aggregate_frac)"}


def config_block(name, dev, batch, probe, steps=10, warmup=5):
    """BASELINE configs[3] in the driver's own run: the full HSM train step of another configuration
    (CelebA-64: 6x64x64, ch_mult [1,2,2,2], 4 blocks per level) at the per-GPU batch, `warmup` + `steps` steps timed like the
    headline pass (wall clock between device fences, nothing inside), then the same steps with the 3x3 limb convolutions
    bracketed by HIP events for its own roofline fraction."""
    from psld_amd import config as C, ops
    from psld_amd.optim import EMAWeightUpdate
    from psld_amd.registry import get_module
    cfg = getattr(C, name)()
    cfg.training.batch_size = batch
    size = cfg.data.image_size
    torch.manual_seed(cfg.training.seed)
    net = get_module("score_fn", "ncsnpp")(cfg).to(dev).train()
    ema = copy.deepcopy(net)
    for p in ema.parameters():
        p.requires_grad = False
    sde = get_module("sde", "psld")(cfg)
    sde.check_nan = True
    crit = get_module("losses", "psld_score_loss")(cfg, sde)
    wrapper = get_module("pl_modules", "sde_wrapper")(cfg, sde, net, ema_score_fn=ema, criterion=crit)
    ema_cb = EMAWeightUpdate(cfg.training.ema_decay)
    g = torch.Generator(device=dev).manual_seed(0)
    data = [torch.rand(batch, 3, size, size, device=dev, generator=g) * 2 - 1 for _ in range(2)]

    def run(n, first):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        last = None
        for i in range(n):
            last = wrapper.training_step(data[(first + i) % len(data)], first + i)
            ema_cb.on_train_batch_end(None, wrapper)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, last

    run(warmup, 0)
    dt, last = run(steps, warmup)
    out = {"config": name, "workload": f"{name} NCSN++ full HSM train step (perturb+fwd+loss+bwd+clip+Adam+EMA), 6x{size}x{size}",
           "per_gpu_batch": batch, "steps": steps, "warmup": warmup, "images_per_s": batch * steps / dt,
           "ms_per_step": 1e3 * dt / steps, "final_loss": float(last.item()),
           "whole_step_tflops": 3 * FORWARD_WORK[name]["gflop"] * 1e9 * batch * steps / dt / 1e12}
    if probe is not None:
        probe.reset()
        probe.enabled = True
        dtp, _ = run(steps, warmup + steps)
        probe.enabled = False
        ps = probe.summary(("split", "wino"))
        if ps is not None:
            peak = PEAK_BF16_MFMA_TFLOPS / LIMB_PRODUCTS
            out["ms_per_step_probed_pass"] = 1e3 * dtp / steps
            out["roofline"] = {"bound": "mfma", "kernel": "3x3 limb-MFMA convolutions, forward + data gradient (as the headline's)",
                               "achieved": ps["tflops"], "peak": peak, "unit": "TFLOP/s", "frac": ps["tflops"] / peak,
                               "launches": ps["launches"], "avg_launch_us": ps["avg_us"],
                               "share_of_step": ps["total_ms"] / (1e3 * dtp)}
            out["weight_gradient"] = wgrad_summary(probe, dtp)
    ops.check_device_errors(dev)
    del wrapper, net, ema
    torch.cuda.empty_cache()
    return out


def wgrad_summary(probe, dt_pass):
    """The 3x3 weight gradients of the probed pass: Winograd-domain launches (kernel + its reduction) and direct limb
    launches (kernel only; their slabs are reduced in batched launches), direct-equivalent TFLOP/s against the limb ceiling."""
    peak = PEAK_BF16_MFMA_TFLOPS / LIMB_PRODUCTS
    out = {}
    for kind, label in (("wgrad_wino", "winograd_domain"), ("wgrad", "direct")):
        r = probe.summary(kind)
        if r is not None:
            out[label] = {"tflops_direct_equivalent": r["tflops"], "frac_of_limb_ceiling": r["tflops"] / peak,
                          "launches": r["launches"], "avg_launch_us": r["avg_us"], "share_of_step": r["total_ms"] / (1e3 * dt_pass)}
    return out or None


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probe", action="store_true")
    ap.add_argument("--no-forward", action="store_true", help="skip the eval-forward block (kernel-stats profiles of the training steps only)")
    ap.add_argument("--no-config-block", action="store_true", help="skip the CelebA-64 (BASELINE configs[3]) train-step block of the default run")
    ap.add_argument("--bucket-mb", type=int, default=64)
    ap.add_argument("--config", default="c10_sota", choices=["c10_sota", "celeba64_sota"],
                    help="c10_sota = BASELINE.json configs[0..2] (headline); celeba64_sota = configs[3] (extra data point)")
    ap.add_argument("--sample-batch", type=int, default=512, help="per-GPU batch of the EM sampling run (0 = skip)")
    ap.add_argument("--sample-steps", type=int, default=1000,
                    help="n_discrete_steps of the sampling run (1000 = configs[4], ~3 min at B=512; fewer = scaled estimate)")
    ap.add_argument("--graphs", action="store_true",
                    help="hipGraph-captured training step (SDEWrapper.enable_graphs): for the launch-bound small-batch regime")
    ap.add_argument("--launch-check", action="store_true",
                    help="form the process group, all-reduce ones, print the JSON line and exit (no model work; runs on "
                         "CPU with gloo: how the self-launch path is tested without a GPU)")
    ap.add_argument("--per-layer-reductions", action="store_true",
                    help="A/B: dgamma / dbeta / bias / split-K slab reductions launched per layer (NCSNpp.defer_param_grads = "
                         "False) instead of one batched launch per pass; recorded in the JSON line")
    ap.add_argument("--no-partial-sample", action="store_true",
                    help="skip the second sampling run (the last, partial batch of a GPU's shard of the 50 000 samples)")
    ap.add_argument("--launch-timeout", type=float, default=float(os.environ.get("PSLD_LAUNCH_TIMEOUT_S", "1500")),
                    help="self-launched --gpus N > 1: seconds after which the parent kills the job and exits 124")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--test-hang-rank", type=int, default=-1, help=argparse.SUPPRESS)   # tests: this rank never joins
    return ap.parse_args()


def self_launch(args) -> int:
    """--gpus N > 1 without a launcher: start N ranks as a child `torch.distributed.run` and return its exit code.
    The parent has made no HIP call (importing torch does not initialise the GPU).  The child runs in its own session; if
    it has not finished after --launch-timeout seconds the parent kills exactly that process group (agent and ranks) and
    exits 124 - a watchdog only ever kills the child and exits non-zero, it never re-executes anything."""
    import signal
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    child = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        return child.wait(timeout=args.launch_timeout)
    except subprocess.TimeoutExpired:
        print(f"bench.py: the {args.gpus}-rank job (pid {child.pid}) did not finish within --launch-timeout "
              f"{args.launch_timeout:.0f} s; killing its process group", file=sys.stderr, flush=True)
    except KeyboardInterrupt:
        pass
    for sig in (signal.SIGTERM, signal.SIGKILL):
        try:
            os.killpg(child.pid, sig)          # start_new_session: pgid == the child's pid, nobody else is in it
        except ProcessLookupError:
            break
        try:
            child.wait(timeout=10)
            break
        except subprocess.TimeoutExpired:
            continue
    return 124


def psld_env():
    """PSLD_* variables of this process (the kernel / policy switches): recorded in the JSON line so that a non-default run
    cannot pass for the default one."""
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("PSLD_")}


def refuse_ablations():
    """Timing-only ablation modes compute wrong results by construction.  They exist only in libpsld_hip_abl.so (make -C tools/abl),
    which tools/ab_*.sh load through PSLD_HIP_LIB; a benchmark line is never produced with them."""
    bad = [k for k in os.environ if k.startswith("PSLD_") and k.endswith("_ABL")]
    if "abl" in os.path.basename(os.environ.get("PSLD_HIP_LIB", "")):
        bad.append("PSLD_HIP_LIB=" + os.environ["PSLD_HIP_LIB"])
    if bad:
        print("bench.py: refusing to run with ablation switches / the ablation library: " + ", ".join(sorted(bad)), file=sys.stderr)
        return True
    return False


def main():
    args = parse_args()
    if refuse_ablations():
        return 2
    if args.cpu_baseline_only:
        from psld_amd import config as C
        print(json.dumps(cpu_baseline(C.c10_sota())), flush=True)
        return 0
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)

    import psld_amd
    from psld_amd import config as C, ops
    from psld_amd.ddp import BucketReducer, init_distributed
    from psld_amd.optim import EMAWeightUpdate
    from psld_amd.registry import get_module
    import torch.distributed as dist

    force_pg = os.environ.get("PSLD_FORCE_PG", "0") == "1"   # 1-GPU rehearsal of the RCCL path
    # PSLD_DIST_BACKEND=gloo + PSLD_SHARE_GPU=1: rehearsal of the N>1 code path with several ranks on ONE
    # GPU (RCCL needs one device per rank; gloo stages through the host)
    backend = os.environ.get("PSLD_DIST_BACKEND") or None
    share_gpu = os.environ.get("PSLD_SHARE_GPU", "0") == "1"
    if args.launch_check and backend is None and torch.cuda.device_count() < max(1, args.gpus):
        backend = "gloo"
    if args.test_hang_rank >= 0 and int(os.environ.get("RANK", "0")) == args.test_hang_rank:
        time.sleep(3600)
    from psld_amd.ddp import RendezvousTimeout
    try:
        rank, local, world = init_distributed(backend=backend, force=force_pg)
    except RendezvousTimeout as e:
        print(f"bench.py: {e}", file=sys.stderr, flush=True)
        return 4
    if share_gpu:
        local = 0
    if world != max(1, args.gpus):
        print(f"bench.py: --gpus {args.gpus} but the launcher provides WORLD_SIZE={world}; refusing to report a "
              f"{world}-rank run as a {args.gpus}-GPU number", file=sys.stderr)
        return 2
    dist_on = world > 1 or force_pg
    backend_name = dist.get_backend() if dist_on else None

    def barrier():
        if dist_on:
            dist.barrier(device_ids=[local]) if backend_name == "nccl" else dist.barrier()

    ones_ok = None
    if dist_on:        # the group really spans `world` ranks: an all-reduce of ones must give world
        on_dev = backend_name == "nccl"
        if on_dev:
            torch.cuda.set_device(local)
        one = torch.ones(1, device=torch.device("cuda", local) if on_dev else "cpu")
        dist.all_reduce(one)
        ones_ok = float(one.item()) == float(world)
        # RCCL writes its version banner through C stdio when the first communicator comes up; push it out NOW so that the
        # JSON line below stays the last line of stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
        if not ones_ok:
            print(f"bench.py: all-reduce of ones gave {one.item()} on a world of {world}", file=sys.stderr)
            return 3
    if args.launch_check:
        barrier()
        samp = None
        if dist_on:     # the sampling half's rank rule, on fake timings: shard, seed + rank, max over ranks
            sh = sampling_shard(rank, world, 0, batch=max(1, args.sample_batch))
            fake = {"measured_batch_s": 1.0 + rank, "sampler_s": 0.5 + rank, "finite": True, "n_discrete_steps": 1000,
                    "network_evals_per_s": 1.0, "fwd_tflops": 1.0}
            own = -(-50000 // world)
            if own % max(1, args.sample_batch):          # the shard's last batch is partial: it is measured too
                fake["partial_batch"] = {"batch": own % max(1, args.sample_batch), "measured_batch_s": 0.25 + rank,
                                         "finite": True, "ms_per_em_step": 0.25 + rank}
            samp = reduce_sampling(fake, rank, world, backend_name == "nccl", torch.device("cuda", local) if backend_name == "nccl" else None, sh)
            lohi = torch.zeros(2 * world, dtype=torch.float64, device=torch.device("cuda", local) if backend_name == "nccl" else "cpu")
            lohi[2 * rank], lohi[2 * rank + 1] = sh["lo"], sh["hi"]
            dist.all_reduce(lohi)
            samp["shards"] = [[int(lohi[2 * r]), int(lohi[2 * r + 1])] for r in range(world)]
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "backend": backend_name,
                              "allreduce_ones_ok": ones_ok, "parallelism": f"dp{world}", "sampling_check": samp,
                              "self_launched": os.environ.get("TORCHELASTIC_RUN_ID") is not None}), flush=True)
        if dist_on:
            dist.destroy_process_group()
        return 0

    # the CPU baseline runs FIRST, in a child process with its own thread pool pinned to physical cores, before this
    # process has touched the GPU; nothing else is running on the box while it is timed
    cpu_base = None
    if world == 1 and not args.no_cpu_baseline and args.config == "c10_sota":
        import subprocess
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"],
                               capture_output=True, text=True, timeout=900)
            cpu_base = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else \
                {"error": (r.stderr or r.stdout)[-400:]}
        except Exception as e:  # noqa: BLE001
            cpu_base = {"error": repr(e)}

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    psld_amd.import_modules_into_registry()
    ops.lib()

    cfg = getattr(C, args.config)()
    cfg.training.batch_size = args.batch
    size = cfg.data.image_size
    torch.manual_seed(cfg.training.seed)                      # same seed on every rank (train_sde.py:29)
    net = get_module("score_fn", "ncsnpp")(cfg).to(dev).train()
    if args.per_layer_reductions:
        net.defer_param_grads = False
    ema = copy.deepcopy(net)
    for p in ema.parameters():
        p.requires_grad = False
    sde = get_module("sde", "psld")(cfg)
    sde.check_nan = True
    crit = get_module("losses", "psld_score_loss")(cfg, sde)
    wrapper = get_module("pl_modules", "sde_wrapper")(cfg, sde, net, ema_score_fn=ema, criterion=crit)
    ema_cb = EMAWeightUpdate(cfg.training.ema_decay)
    if args.graphs:
        wrapper.enable_graphs(True)
        args.no_probe = True          # per-launch HIP-event brackets cannot sit inside a captured graph
        args.warmup = max(args.warmup, 3)   # two eager steps, then the capture
    reducer = None
    if dist_on:
        reducer = BucketReducer(bucket_bytes=args.bucket_mb << 20, force_collective=force_pg, profile=True)
        net.set_reducer(reducer)
    g = torch.Generator(device=dev).manual_seed(rank)         # per-rank data
    data = [torch.rand(args.batch, 3, size, size, device=dev, generator=g) * 2 - 1 for _ in range(4)]

    def step(i):
        loss = wrapper.training_step(data[i % len(data)], i)
        ema_cb.on_train_batch_end(None, wrapper)
        return loss

    probe = ConvProbe(ops)
    if not args.no_probe:
        probe.install()
    for i in range(args.warmup):
        step(i)

    def fence():
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()

    def timed_pass(first_step):
        """K steps bracketed by barrier + device synchronize on both sides; wall clock, max over ranks."""
        fence()
        t0 = time.perf_counter()
        last_ = None
        for i in range(args.steps):
            last_ = step(first_step + i)
        fence()
        dt_ = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt_], device=dev if backend_name == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_ = float(tt.item())
        return dt_, last_

    fence()
    if reducer is not None:
        reducer.stats()                                       # drop the warm-up steps' events
    # pass 1 = the headline: the product's step, no per-launch HIP-event brackets inside the timed region
    dt, last = timed_pass(args.warmup)
    # pass 2 = the same K steps with every 3x3 limb convolution bracketed by HIP events -> `roofline`
    dt_probed = None
    if not args.no_probe:
        probe.enabled = True
        dt_probed, _ = timed_pass(args.warmup + args.steps)
        probe.enabled = False
    loss_val = float(last.item())
    # the headline pass's probe summaries now: the configuration block below reuses the probe
    ps = probe.summary(("split", "wino"))      # every 3x3 limb convolution, forward + data gradient
    pw, pd, pt = probe.summary("wino"), probe.summary("split"), probe.summary("tile")
    wg = wgrad_summary(probe, dt_probed) if dt_probed is not None else None
    ops.check_device_errors(dev)      # a device-side timeout (team GroupNorm backward) voids the run: raise on the rank that saw it
    # replicas stay identical: same seed, same averaged gradient -> same parameters on every rank
    in_sync = None
    if world > 1:
        chk = net.flatten_parameters().double().sum().reshape(1)
        chk = chk if backend_name == "nccl" else chk.cpu()
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        in_sync = bool(lo.item() == hi.item())

    sampling = None
    if args.sample_batch > 0 and args.config == "c10_sota":
        # second half of the metric: EVERY rank samples one batch of its own shard at the same time (eval/sample.py:100-109)
        # and then the LAST, partial batch of that shard (50 000 / 8 = 6250 = 12 x 512 + 106 per GPU on 8 GPUs)
        shard = sampling_shard(rank, world, int(cfg.evaluation.seed), batch=args.sample_batch)
        if world > 1:
            torch.manual_seed(shard["seed"])                  # wrapper.py:93-99: seed + global_rank
        n_steps = max(3, args.sample_steps)
        sampling = sampling_run(cfg, ema, sde, dev, args.sample_batch, n_steps)
        own = -(-50000 // (world if world > 1 else 8))        # the same on every rank (shard_range: ceil(n / world) per rank)
        partial_n = own % args.sample_batch
        if partial_n and not args.no_partial_sample:
            part = sampling_run(cfg, ema, sde, dev, partial_n, n_steps)
            sampling["partial_batch"] = {"batch": partial_n, "measured_batch_s": part["measured_batch_s"],
                                         "finite": part["finite"], "ms_per_em_step": part["ms_per_em_step"]}
            full_batches = own // args.sample_batch
            sampling["per_gpu_50k_samples_8gpu_s"] = (full_batches * sampling["measured_batch_s"] +
                                                      part["measured_batch_s"]) * (1000.0 / n_steps)
            sampling["note"] = sampling["note"].split("per_gpu_50k")[0] + \
                (f"per_gpu_50k = {full_batches} full batches x measured_batch_s + ONE measured partial batch of {partial_n} "
                 f"samples ({own} samples per GPU on 8 GPUs): each GPU samples its own shard, no collective (SURVEY 8(e))")
        if world > 1:
            sampling = reduce_sampling(sampling, rank, world, backend_name == "nccl", dev, shard)
        fence()
    fwd_blk = None
    if rank == 0 and not args.launch_check and not args.no_forward and torch.cuda.is_available():
        fwd_blk = forward_block(net, dev, args.batch, size, config=args.config)
        try:
            fwd_blk["hbm_live"] = forward_hbm_live(net, dev, args.batch, size)
        except Exception as e:  # noqa: BLE001
            fwd_blk["hbm_live"] = {"error": repr(e)}
    fence()
    # BASELINE configs[3] (CelebA-64) at its per-GPU batch, in the default single-GPU run only
    other_blk = None
    if world == 1 and not args.no_config_block and args.config == "c10_sota" and args.batch == 128:
        other_blk = config_block("celeba64_sota", dev, 128, None if args.no_probe else probe)
    fence()

    if rank == 0:
        total_imgs = world * args.batch * args.steps
        out = {
            "metric": "train images/sec, CIFAR-10 PSLD (6ch 32x32) NCSN++ HSM step",
            "value": total_imgs / dt, "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "ms_per_step_probed_pass": (1e3 * dt_probed / args.steps) if dt_probed is not None else None,
            "timing_note": ("value / ms_per_step: K steps of the product path, no instrumentation inside the timed region; "
                            "ms_per_step_probed_pass: the SAME K steps run a second time with every 3x3 limb convolution "
                            "bracketed by two hipEventRecord calls (what `roofline` is computed from)"),
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "math": ("f32 in / f32 accumulate; 3x3 convolutions, 1x1 / NIN projections, attention products and their gradients as exact "
                     "3-limb bf16 splits, 6 bf16 MFMA products per fp32 product (dropped terms < 2^-23 of the product); the 3x3 "
                     "convolutions of the 32x32 / 16x16 levels (forward, data gradient) in Winograd F(2x2,3x3) form on the same "
                     "limb arithmetic (fp32 transforms; PSLD_WINOGRAD=0: direct); everything else fp32 MFMA / fp32 VALU"
                     if ops.math_mode() == "bf16x6" else "f32 MFMA (v_mfma_f32_32x32x2_f32)"),
            "config": {"workload": "C10-SOTA NCSN++ (nf=128, ch_mult=[2,2,2], nres=8, attn@16, fir, fourier, "
                                   "dropout 0.15) full HSM train step: perturb+fwd+loss+bwd+clip+Adam+EMA",
                       "per_gpu_batch": args.batch, "global_batch": world * args.batch,
                       "parallelism": f"dp{world}", "image": "6x32x32"},
            "images_per_sec_per_gpu": total_imgs / dt / world,
            "final_loss": loss_val,
            "captured_step": bool(args.graphs and getattr(wrapper, "_graph_steps", None) and
                                  any("graph" in e for e in wrapper._graph_steps.values())),
        }
        if dist_on:
            st = reducer.stats()
            out["distributed"] = {"backend": ("rccl" if backend_name == "nccl" else backend_name), "world": world,
                                  "allreduce_ones_ok": ones_ok, "replicas_in_sync": in_sync,
                                  "gradient_bytes_per_step": int(net.flat_grad().numel()) * 4,
                                  "shared_gpu_rehearsal": share_gpu}
            out["overlap"] = st if st is not None else {"note": "no collective ran"}
        pmc, pmc_err = pmc_traffic_record()
        hbm, hbm_err = hbm_in_situ_record()
        if hbm is not None and (args.batch != hbm.get("batch", 128) or args.config != "c10_sota"):
            hbm, hbm_err = None, f"the in-situ record was measured at C10-SOTA B={hbm.get('batch', 128)}, this run is {args.config} B={args.batch}"
        if ps is not None:
            peak = PEAK_BF16_MFMA_TFLOPS / LIMB_PRODUCTS
            issued = ((pw["total_flop"] / 2.25 if pw else 0.0) + (pd["total_flop"] if pd else 0.0)) * LIMB_PRODUCTS / (ps["total_ms"] * 1e-3) / 1e12
            out["roofline"] = {
                "bound": "mfma",
                "kernel": "3x3 limb-MFMA convolutions, forward + data gradient: wino_conv8s_kernel (Winograd F(2x2,3x3), "
                          "32x32 and 16x16 levels) + dconv_kernel / dconv_lp_kernel (direct, 8x8 level), bf16x6",
                "achieved": ps["tflops"], "peak": peak, "unit": "TFLOP/s", "frac": ps["tflops"] / peak,
                "peak_note": f"hardware utilisation first: frac_of_dense_bf16_peak = {issued / PEAK_BF16_MFMA_TFLOPS:.3f} (the bf16 MFMA "
                             "FLOPs the matrix pipe really issues / the 2500 TFLOP/s dense peak).  frac is algorithmic: "
                             "achieved = ALGORITHMIC (direct-convolution) 2*M*N*9*Cin of every launch / its HIP-event duration; "
                             "peak = 2500 TFLOP/s dense bf16 MFMA / 6 limb products per fp32 product = the fp32-equivalent "
                             "ceiling of a DIRECT limb convolution.  The Winograd launches issue 2.25x fewer MFMAs than "
                             "that count (mfma_issued_tflops is what the matrix pipe really runs); a register-only loop "
                             "with random operand data sustains 1.90 PFLOP/s (tools/mfma_peak.hip).",
                "mfma_issued_tflops": issued,
                "frac_of_sustained_mfma": issued / SUSTAINED_BF16_MFMA_TFLOPS,
                "frac_of_f32_mfma_peak": ps["tflops"] / PEAK_F32_MFMA_TFLOPS,
                "winograd": ({"tflops_direct_equivalent": pw["tflops"], "launches": pw["launches"], "avg_launch_us": pw["avg_us"],
                              "share_of_step": pw["total_ms"] / (1e3 * (dt_probed or dt))} if pw else None),
                "direct": ({"tflops": pd["tflops"], "launches": pd["launches"], "avg_launch_us": pd["avg_us"],
                            "share_of_step": pd["total_ms"] / (1e3 * (dt_probed or dt))} if pd else None),
                "frac_of_dense_bf16_peak": issued / PEAK_BF16_MFMA_TFLOPS,
                "traffic": pmc.get("traffic_bytes") if pmc else None,
                "traffic_note": pmc.get("note") if pmc else pmc_err,
                "hbm_bound_aggregate_frac": hbm.get("hbm_bound_aggregate_frac") if hbm else None,
                "hbm_bound_under_0.6": hbm.get("under_0.6") if hbm else None,
                "hbm_bound_note": (f"{hbm['_path']}: {hbm['aggregate_gb_per_s']:.0f} GB/s byte-weighted over the bandwidth-bound kernels of "
                                   f"the B=128 step ({hbm['bytes_per_step'] / 1e9:.1f} GB algorithmic in {hbm['ms_per_step']:.1f} ms per step)")
                if hbm else hbm_err,
                "launches": ps["launches"], "avg_launch_us": ps["avg_us"],
                "share_of_step": ps["total_ms"] / (1e3 * (dt_probed or dt)),
                "weight_gradient": wg}
        elif pt is not None:        # PSLD_MATH=f32: the fp32 MFMA tile engine carries the convolutions
            out["roofline"] = {"bound": "mfma", "kernel": "tile_kernel_fast<IM2COL,KC> (fp32 MFMA convolutions)",
                               "achieved": pt["tflops"], "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": pt["tflops"] / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                               "launches": pt["launches"], "avg_launch_us": pt["avg_us"],
                               "share_of_step": pt["total_ms"] / (1e3 * (dt_probed or dt))}
        else:
            out["roofline"] = None
        if ps is not None and pt is not None:
            out["other_convs_f32_mfma"] = {"tflops": pt["tflops"], "launches": pt["launches"],
                                           "share_of_step": pt["total_ms"] / (1e3 * (dt_probed or dt))}
        fwd_gflop = {"c10_sota": 76.46, "celeba64_sota": 84.17}[args.config]   # measured fwd GFLOP/img
        step_flops = 3 * fwd_gflop * 1e9 * args.batch           # train step = 3 x forward
        out["whole_step_tflops_per_gpu"] = step_flops * args.steps / dt / 1e12
        if args.config != "c10_sota":
            out["metric"] = out["metric"].replace("CIFAR-10 PSLD (6ch 32x32)", f"{args.config} (6ch {size}x{size})")
            out["config"]["workload"] = f"{args.config} NCSN++ full HSM train step"
            out["config"]["image"] = f"6x{size}x{size}"
        if fwd_blk is not None:
            fh_, fh_err = hbm_in_situ_forward_record()
            if fh_ is not None and (args.batch != fh_.get("batch", 128) or args.config != "c10_sota"):
                fh_, fh_err = None, f"the forward in-situ record was measured at C10-SOTA B={fh_.get('batch', 128)}, this run is {args.config} B={args.batch}"
            fwd_blk["hbm_bound_aggregate_frac"] = fh_.get("hbm_bound_aggregate_frac") if fh_ else None
            fwd_blk["hbm_bound_under_0.6"] = fh_.get("under_0.6") if fh_ else None
            fwd_blk["hbm_bound_note"] = (f"{fh_['_path']}: {fh_['aggregate_gb_per_s']:.0f} GB/s byte-weighted over the bandwidth-bound "
                                         f"kernels of the B={fh_.get('batch', 128)} eval forward ({fh_['bytes_per_step'] / 1e9:.2f} GB "
                                         f"algorithmic in {fh_['ms_per_step']:.2f} ms per forward)") if fh_ else fh_err
            out["forward"] = fwd_blk
        if other_blk is not None:
            out["celeba64"] = other_blk
        if sampling is not None:
            out["sampling"] = sampling
        out["cpu_baseline"] = cpu_base
        out["env"] = psld_env()
        out["options"] = {"per_layer_reductions": bool(args.per_layer_reductions), "graphs": bool(args.graphs)}
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if dist_on:
        barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
