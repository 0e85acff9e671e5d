"""In-situ HBM rate of the bandwidth-bound kernels INSIDE the training step (SURVEY 8(d) "Which
roofline"): algorithmic bytes of every bandwidth-bound entry point, logged from its arguments by a stand-in for the loaded
library, joined with rocprofv3's per-kernel durations of the same process.

    # on the GPU box
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/hbm_prof -- python3 tools/hbm_in_situ.py run gpurun_out/r04/hbm_bytes.json
    python3 tools/hbm_in_situ.py join gpurun_out/r04/hbm_prof gpurun_out/r04/hbm_bytes.json profiles/r04/hbm_in_situ
    # the eval forward alone (north_star's figure): `run-forward` instead of `run`, joined into .../hbm_in_situ_forward

Algorithmic bytes = every distinct input read once + every output written once (fp32 unless noted), from the launch
arguments: GroupNorm statistics 4 B/element (nothing when the producer's epilogue left partial sums), apply 8 (10 into limb
planes), backward 12 (+4 with an identity-branch gradient, +4 when accumulating), FIR in + out, bias-gradient column sums 4,
split-K slab reductions (nsplit + 1) x 4, Adam 36 B/parameter with the EMA, EMA 12, gradient norm 4, axpby / copies /
softmax / SiLU by their operands, perturbation 60 B/pixel, loss 8 (+4 with the gradient)."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PEAK = 8.0e12
SOURCES = ["psld_amd/csrc/norm_act.hip", "psld_amd/csrc/resample.hip", "psld_amd/csrc/pointwise.hip", "psld_amd/csrc/optim.hip",
           "psld_amd/csrc/sde.hip", "psld_amd/csrc/wgrad_wino.hip", "psld_amd/csrc/common.h", "psld_amd/score_fn.py"]

# entry point -> (family, bytes(args))
def _b(fam, fn):
    return fam, fn

BYTES = {
    "psld_gn_stats_nhwc_f32": _b("gn_stats", lambda a: 4 * a[1] * a[2] * a[3]),
    "psld_gn_stats_from_partials_f32": _b("gn_stats", lambda a: a[2] * max(1, a[3] // 64) * (a[4] // max(1, a[1])) * 16),
    "psld_gn_apply_nhwc_f32": _b("gn_apply", lambda a: 8 * a[4] * a[5] * a[6]),
    "psld_gn_apply_limb_nhwc": _b("gn_apply_limb", lambda a: 10 * a[4] * a[5] * a[6]),
    "psld_gn_bwd_team_f32": _b("gn_bwd", lambda a: (12 + (4 if a[15] else 0) + (4 if a[16] else 0)) * a[6] * a[7] * a[8]),
    "psld_gn_bwd_nhwc_f32": _b("gn_bwd", lambda a: (12 + (4 if a[15] else 0) + (4 if a[16] else 0)) * a[6] * a[7] * a[8]),
    "psld_param_reduce2_f32": _b("param_reduce", lambda a: 4 * (a[2] + 1) * a[4] * (2 if a[1] else 1)),
    "psld_bias_grad_seg_f32": _b("bias_grad", lambda a: 4 * a[2] * a[3] * 3 * a[4]),
    "psld_upfirdn2d_f32": _b("fir", lambda a: 4 * a[2] * a[3] * (a[4] * a[5] + _fir_out(a) * (2 if a[18] else 1))),
    "psld_axpby_f32": _b("axpby", lambda a: 4 * a[5] * (2 + (1 if a[2] else 0) + (1 if a[6] else 0))),
    "psld_silu_f32": _b("silu", lambda a: 8 * a[2]),
    "psld_silu_bwd_f32": _b("silu", lambda a: 12 * a[3]),
    "psld_colsum_f32": _b("bias_grad", lambda a: 4 * a[2] * a[3] * a[4]),
    "psld_bias_grad_f32": _b("bias_grad", lambda a: 4 * a[2] * a[3] * a[4]),
    "psld_reduce_slabs_f32": _b("reduce_slabs", lambda a: 4 * a[2] * (a[1] + 1)),
    # Winograd-domain weight gradient: only its reduction kernel is bandwidth-bound (16 positions x nsplit slabs in, OIHW out)
    "psld_conv3x3_wgrad_wino_f32": _b("reduce_slabs", lambda a: 4 * a[2] * (a[4] + a[6]) * (16 * a[11] + 9)),
    "psld_copy2d_f32": _b("copy2d", lambda a: (8 + (4 if a[6] else 0)) * a[4] * a[5]),
    "psld_scale_copy2d_f32": _b("copy2d", lambda a: 8 * a[4] * a[5]),
    "psld_softmax_rows_f32": _b("softmax", lambda a: 8 * a[2] * a[3]),
    "psld_softmax_rows_bwd_f32": _b("softmax", lambda a: 12 * a[3] * a[4]),
    "psld_nchw_to_nhwc_f32": _b("layout", lambda a: 8 * a[2] * a[3] * a[4]),
    "psld_nhwc_to_nchw_f32": _b("layout", lambda a: 8 * a[2] * a[3] * a[4]),
    "psld_perturb_f32": _b("perturb", lambda a: 4 * a[5] * a[7] * (a[6] + 4 * a[6]) + (16 * a[5] * a[7] * 2 * a[6] if a[9] else 0)),
    "psld_sqerr_loss_f32": _b("sqerr", lambda a: (8 + (4 if a[5] else 0)) * a[2]),
    "psld_adam_ema_f32": _b("adam", lambda a: (28 + (8 if a[4] else 0)) * a[5]),
    "psld_ema_f32": _b("ema", lambda a: 12 * a[2]),
    "psld_grad_norm_f32": _b("grad_norm", lambda a: 4 * a[1]),
    "psld_f32_to_limb": _b("limb_convert", lambda a: 10 * a[1] * a[2]),
}

# kernel-name substring -> family (first match wins)
KERNELS = [
    ("gn_apply_limb_kernel", "gn_apply_limb"), ("gn_apply_kernel", "gn_apply"), ("gn_partial_kernel", "gn_stats"),
    ("gn_finalize_kernel", "gn_stats"), ("gn_bwd_", "gn_bwd"), ("upfirdn", "fir"), ("axpby_kernel", "axpby"),
    ("silu_", "silu"), ("colsum", "bias_grad"), ("bias_grad", "bias_grad"), ("reduce_slabs", "reduce_slabs"), ("wwgrad_reduce", "reduce_slabs"), ("param_reduce", "param_reduce"),
    ("scale_copy2d", "copy2d"), ("copy2d_kernel", "copy2d"), ("softmax_rows", "softmax"), ("nchw_to_nhwc", "layout"),
    ("nhwc_to_nchw", "layout"), ("perturb_kernel", "perturb"), ("perturb_coeffs", "perturb"), ("sqerr", "sqerr"),
    ("adam_ema_kernel", "adam"), ("ema_kernel", "ema"), ("sumsq", "grad_norm"), ("f32_to_limb", "limb_convert"),
]


def _fir_out(a):
    kh, kw, ux, uy, dx, dy, px0, px1, py0, py1 = a[7:17]
    oh = (a[4] * uy + py0 + py1 - kh) // dy + 1
    ow = (a[5] * ux + px0 + px1 - kw) // dx + 1
    return oh * ow


class _Log:
    """Stand-in for the loaded library: forwards every call, adds the algorithmic bytes of the bandwidth-bound ones."""

    class _T:
        _keep = []

    def __init__(self, real):
        self._real, self._tape = real, self._T()
        self.bytes, self.calls = collections.Counter(), collections.Counter()

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        ent = BYTES.get(name)
        if ent is None:
            return fn
        fam, calc = ent

        def logged(*args):
            vals = [int(v.value if hasattr(v, "value") and v.value is not None else 0) if hasattr(v, "value") else (v if v is not None else 0)
                    for v in args]
            try:
                self.bytes[fam] += int(calc(vals))
                self.calls[fam] += 1
            except Exception as e:  # noqa: BLE001
                print("hbm_in_situ: cannot size", name, e, file=sys.stderr)
            return fn(*args)
        return logged


def _hook_tables(log):
    """The table-driven launches (psld_param_reduce_batch_f32, psld_reduce_slabs_batch_f32) carry their jobs in device memory:
    the executor reports their algorithmic bytes through ops.byte_hook."""
    from psld_amd import ops

    def hook(fam, nbytes):
        log.bytes[fam] += int(nbytes)
        log.calls[fam] += 1
    ops.byte_hook = hook


def run(out_path, steps=8, warmup=2):
    import copy
    import torch
    import psld_amd
    from psld_amd import _lib, config as C
    from psld_amd.optim import EMAWeightUpdate
    from psld_amd.registry import get_module
    log = _Log(_lib.load_real())
    _lib.set_proxy(log)                                  # from here on every launch of the process is sized
    _hook_tables(log)
    psld_amd.import_modules_into_registry()
    cfg = C.c10_sota()
    dev = torch.device("cuda")
    torch.manual_seed(0)
    net = get_module("score_fn", "ncsnpp")(cfg).to(dev).train()
    ema = copy.deepcopy(net)
    for p in ema.parameters():
        p.requires_grad = False
    sde = get_module("sde", "psld")(cfg)
    crit = get_module("losses", "psld_score_loss")(cfg, sde)
    wr = get_module("pl_modules", "sde_wrapper")(cfg, sde, net, ema_score_fn=ema, criterion=crit)
    cb = EMAWeightUpdate(cfg.training.ema_decay)
    x0 = torch.rand(128, 3, 32, 32, device=dev) * 2 - 1
    for i in range(warmup + steps):
        wr.training_step(x0, i)
        cb.on_train_batch_end(None, wr)
    torch.cuda.synchronize()
    _lib.set_proxy(None)
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    with open(out_path, "w") as fh:
        json.dump({"steps": warmup + steps, "batch": 128, "config": "c10_sota", "bytes": dict(log.bytes), "calls": dict(log.calls)}, fh, indent=1)
    print("wrote", out_path)


def run_forward(out_path, steps=20, warmup=3, batch=128):
    """The eval forward alone at the training batch (north_star: ">= 70 % of the per-GPU HBM roofline on the U-Net
    forward at batch 128x6x32x32"): the EMA network's inference path, i.e. GroupNorm + SiLU fused into the Winograd
    staging where the executor does that (those bytes then ride on an MFMA-bound kernel and are not counted here)."""
    import torch
    import psld_amd
    from psld_amd import _lib, config as C
    from psld_amd.registry import get_module
    log = _Log(_lib.load_real())
    _lib.set_proxy(log)
    _hook_tables(log)
    psld_amd.import_modules_into_registry()
    cfg = C.c10_sota()
    dev = torch.device("cuda")
    torch.manual_seed(0)
    net = get_module("score_fn", "ncsnpp")(cfg).to(dev).eval()
    x = torch.randn(batch, 6, 32, 32, device=dev)
    t = torch.rand(batch, device=dev) * 0.98 + 0.01
    with torch.no_grad():
        for _ in range(warmup + steps):
            net(x, t)
    torch.cuda.synchronize()
    _lib.set_proxy(None)
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    with open(out_path, "w") as fh:
        json.dump({"steps": warmup + steps, "batch": batch, "config": "c10_sota", "phase": "eval forward",
                   "bytes": dict(log.bytes), "calls": dict(log.calls)}, fh, indent=1)
    print("wrote", out_path)


def join(prof_dir, bytes_path, out_prefix):
    rec = json.load(open(bytes_path))
    files = glob.glob(os.path.join(prof_dir, "**", "*kernel_stats.csv"), recursive=True)
    assert files, f"no *kernel_stats.csv under {prof_dir}"
    t_ns, n_k = collections.Counter(), collections.Counter()
    for f in files:
        for r in csv.DictReader(open(f, newline="")):
            fam = next((fm for pat, fm in KERNELS if pat in r["Name"]), None)
            if fam:
                t_ns[fam] += float(r["TotalDurationNs"])
                n_k[fam] += int(r["Calls"])
    steps = rec["steps"]
    rows, tb, tt = [], 0.0, 0.0
    for fam, b in sorted(rec["bytes"].items(), key=lambda kv: -t_ns.get(kv[0], 0)):
        t = t_ns.get(fam, 0.0) * 1e-9
        if t <= 0:
            continue
        rate = b / t
        rows.append({"family": fam, "calls": rec["calls"][fam], "kernel_launches": n_k[fam], "bytes_per_step": b / steps,
                     "ms_per_step": t * 1e3 / steps, "gb_per_s": rate / 1e9, "frac_of_8tbs": rate / PEAK})
        tb += b
        tt += t
    h = hashlib.sha256()
    for f in SOURCES:
        h.update(open(os.path.join(ROOT, f), "rb").read())
    phase = rec.get("phase", "training step")
    what = "eval forwards" if phase == "eval forward" else "training steps"
    out = {"source": "tools/hbm_in_situ.py: algorithmic bytes from the launch arguments of the bandwidth-bound entry points of "
                     f"{steps} C10-SOTA B={rec.get('batch', 128)} {what} (model set-up included) / rocprofv3 --kernel-trace --stats durations of their "
                     "kernels in the same process",
           "batch": rec.get("batch", 128), "phase": phase,
           "aggregate_gb_per_s": tb / tt / 1e9, "hbm_bound_aggregate_frac": tb / tt / PEAK, "bytes_per_step": tb / steps,
           "ms_per_step": tt * 1e3 / steps, "under_0.6": [r["family"] for r in rows if r["frac_of_8tbs"] < 0.6],
           "families": rows, "sources": SOURCES, "sources_sha256": h.hexdigest()}
    json.dump(out, open(out_prefix + ".json", "w"), indent=1)
    with open(out_prefix + ".md", "w") as fh:
        fh.write(f"| kernel family (in the B={rec.get('batch', 128)} {phase}) | launches / step | MB / step (algorithmic) | ms / step | GB/s | of 8 TB/s |\n|---|---|---|---|---|---|\n")
        for r in rows:
            fh.write(f"| {r['family']} | {r['kernel_launches'] / steps:.0f} | {r['bytes_per_step'] / 1e6:.1f} | {r['ms_per_step']:.3f} | "
                     f"{r['gb_per_s']:.0f} | {r['frac_of_8tbs']:.3f} |\n")
        fh.write(f"| **byte-weighted aggregate** | | {tb / steps / 1e6:.1f} | {tt * 1e3 / steps:.3f} | {tb / tt / 1e9:.0f} | **{tb / tt / PEAK:.3f}** |\n")
    print(open(out_prefix + ".md").read())


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    elif sys.argv[1] == "run-forward":
        run_forward(sys.argv[2])
    else:
        join(sys.argv[2], sys.argv[3], sys.argv[4])
