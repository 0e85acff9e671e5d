"""CPU tests of the host-side logic that needs no GPU: config overrides in the reference's Hydra syntax,
registry semantics, module construction / state_dict compatibility, split-K heuristic."""
import json
import os

import pytest
import torch

from psld_amd import config as C
from tests.conftest import GOLDEN


def test_hydra_style_overrides_from_the_reference_scripts():
    from psld_amd.cli import parse_overrides
    # argument strings as they appear in scripts_psld/sota/uncond/cifar10/train_uncond_psld.sh:2-21
    args = ["+dataset=cifar10/cifar10_psld", "dataset.diffusion.data.root=\\'/data/\\'",
            "dataset.diffusion.data.norm=True", "dataset.diffusion.model.score_fn.ch_mult=[2,2,2]",
            "dataset.diffusion.model.score_fn.num_res_blocks=8", "dataset.diffusion.model.score_fn.dropout=0.15",
            "dataset.diffusion.model.score_fn.progressive_input='residual'", "dataset.diffusion.model.score_fn.fir=True",
            "dataset.diffusion.model.score_fn.embedding_type='fourier'", "dataset.diffusion.model.sde.nu=4.01",
            "dataset.diffusion.training.devices=8", "dataset.diffusion.training.fp16=False",
            "+dataset.diffusion.evaluation.sampler.rtol=1e-4"]
    cfg = parse_overrides(C.yaml_default(), args)
    sf = cfg.model.score_fn
    assert sf.ch_mult == [2, 2, 2] and sf.num_res_blocks == 8 and sf.dropout == 0.15
    assert sf.progressive_input == "residual" and sf.fir is True and sf.embedding_type == "fourier"
    assert cfg.data.root == "/data/" and cfg.model.sde.nu == 4.01 and cfg.training.devices == 8
    assert cfg.training.fp16 is False and cfg.evaluation.sampler.rtol == 1e-4
    # the result equals the SOTA preset on every network / SDE key
    ref = C.c10_sota()
    assert dict(cfg.model.score_fn) == dict(ref.model.score_fn) and dict(cfg.model.sde) == dict(ref.model.sde)


def test_registry_semantics():
    import psld_amd
    from psld_amd.registry import _MODULES, get_module, register_module
    psld_amd.import_modules_into_registry()
    for cat, name in (("score_fn", "ncsnpp"), ("sde", "psld"), ("losses", "psld_score_loss"),
                      ("samplers", "em_sde"), ("samplers", "sscs_sde"), ("samplers", "bb_ode"),
                      ("pl_modules", "sde_wrapper")):
        assert get_module(cat, name) is _MODULES[cat][name]
    with pytest.raises(ValueError, match="No module named"):
        get_module("sde", "nope")

    class Fake:
        pass

    class FakeUtil:
        _MODULES = {"sde": {"psld": Fake}}

    psld_amd.install_into(FakeUtil)
    assert FakeUtil._MODULES["sde"]["psld"] is get_module("sde", "psld")
    assert FakeUtil._MODULES["score_fn"]["ncsnpp"] is get_module("score_fn", "ncsnpp")
    with pytest.raises(ValueError, match="Already registered"):
        register_module(category="sde", name="psld")(Fake)


@pytest.mark.parametrize("name,preset", [("tiny", C.tiny), ("c10_sota", C.c10_sota), ("celeba64", C.celeba64_sota)])
def test_state_dict_matches_reference_census(name, preset):
    """Keys, shapes AND order of state_dict() equal the reference's (captured in tests/golden/net_meta.json);
    init follows the reference's scheme (zero biases, unit GroupNorm, 1e-10-scaled output layers)."""
    from psld_amd.score_fn import NCSNpp
    with open(os.path.join(GOLDEN, "net_meta.json")) as fh:
        meta = json.load(fh)[name]
    net = NCSNpp(preset())
    sd = net.state_dict()
    assert [(k, list(v.shape)) for k, v in sd.items()] == [(k, s) for k, s in meta["keys"]]
    assert sum(v.numel() for v in sd.values()) == meta["n_params"]
    assert not net.all_modules[0].W.requires_grad
    last = net.all_modules[-1]
    assert float(last.weight.abs().max()) < 1e-4 and float(last.bias.abs().max()) == 0.0     # init_scale=0 -> 1e-10
    gn = net.all_modules[-2]
    assert torch.all(gn.weight == 1) and torch.all(gn.bias == 0)
    fan = net.all_modules[3].weight            # stem conv: fan_avg uniform bound sqrt(3/((fan_in+fan_out)/2))
    co, ci, kh, kw = fan.shape
    bound = (3.0 / ((ci * kh * kw + co * kh * kw) / 2)) ** 0.5
    assert float(fan.abs().max()) <= bound + 1e-6 and float(fan.abs().max()) > 0.8 * bound


def test_unsupported_config_branches_fail_loudly():
    from psld_amd.score_fn import NCSNpp
    for key, val in (("resblock_type", "ddpm"), ("progressive", "output_skip"), ("nonlinearity", "elu"),
                     ("progressive_input", "input_skip")):
        cfg = C.tiny()
        cfg.model.score_fn[key] = val
        with pytest.raises(NotImplementedError):
            NCSNpp(cfg)


def test_split_k_fills_whole_rounds():
    from psld_amd.score_fn import _pick_nsplit
    for tiles, k in ((36, 131072), (72, 131072), (36, 8192), (4, 131072), (9, 32768)):
        ns = _pick_nsplit(tiles, k)
        blocks = tiles * ns
        rounds = -(-blocks // 512)
        assert blocks / (rounds * 512) > 0.9 and k // ns >= 256
    assert _pick_nsplit(36, 300) == 1


def test_cpu_tensor_is_rejected_without_touching_the_gpu():
    from psld_amd.score_fn import NCSNpp
    net = NCSNpp(C.tiny())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(1, 6, 16, 16), torch.ones(1))


def test_epoch_indices_match_torch_distributed_sampler():
    """The CLI's per-rank epoch shard is what Lightning strategy="ddp" gives the reference (train_sde.py:100-114):
    torch's DistributedSampler(shuffle=True, seed, set_epoch) under DataLoader(drop_last=True)."""
    from torch.utils.data import DistributedSampler
    from psld_amd.cli import epoch_indices
    n, bs, seed = 1003, 16, 7
    ds = list(range(n))
    for world in (1, 2, 8):
        seen = []
        for rank in range(world):
            smp = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True, seed=seed)
            for epoch in (0, 3):
                smp.set_epoch(epoch)
                ref = torch.tensor(list(iter(smp)))
                ref = ref[:ref.numel() // bs * bs]
                mine = epoch_indices(n, bs, seed, epoch, rank, world)
                assert torch.equal(mine, ref), (world, rank, epoch)
            seen.append(epoch_indices(n, bs, seed, 0, rank, world))
        # one epoch = one pass: ranks see disjoint indices (up to the wrap-around padding) and equal step counts
        assert len({s.numel() for s in seen}) == 1
        allidx = torch.cat(seen)
        assert allidx.unique().numel() >= allidx.numel() - world


def test_checkpoint_resume_restores_scheduler_and_adopts_torch_adam_state(tmp_path, recwarn):
    """Resume: the LambdaLR position is restored (not re-warmed from 0) and a torch.optim.Adam state -
    what a reference checkpoint carries - is converted into FusedAdam's flat moments; with no usable state the
    bias-correction step restarts together with the moments, loudly."""
    import copy
    from psld_amd import cli
    from psld_amd.optim import FusedAdam
    from psld_amd.score_fn import NCSNpp

    cfg = C.tiny(image_size=8, nf=16, ch_mult=(1,), num_res_blocks=1, attn_resolutions=(8,))
    net = NCSNpp(cfg)
    ema = copy.deepcopy(net)

    class W:
        score_fn, ema_score_fn = net, ema

    params = [p for p in net.parameters()]
    # a reference-style checkpoint: torch.optim.Adam state after 7 steps
    adam = torch.optim.Adam([p for p in params if p.requires_grad], lr=1e-3)
    for p in params:
        if p.requires_grad:
            p.grad = torch.randn_like(p)
    for _ in range(7):
        adam.step()
    ref_sd = adam.state_dict()
    # the reference's optimizer indexes only trainable parameters in module order; FusedAdam indexes all parameters
    idx = [i for i, p in enumerate(params) if p.requires_grad]
    state = {idx[k]: v for k, v in ref_sd["state"].items()}
    path = str(tmp_path / "ref.ckpt")
    sd = {"score_fn." + k: v for k, v in net.state_dict().items()}
    torch.save({"state_dict": sd, "global_step": 7, "epoch": 1,
                "optimizer_states": [{"state": state, "param_groups": ref_sd["param_groups"]}]}, path)
    opt = FusedAdam(net, lr=2e-4)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: min(s / 5000, 1.0))
    step, epoch = cli.load_checkpoint(path, net, ema, opt, sched)
    assert (step, epoch) == (7, 1) and opt._step == 7
    p0 = next(p for p in params if p.requires_grad)
    o = net._offsets[id(p0)]
    assert torch.equal(opt._m[o:o + p0.numel()].view(p0.shape), adam.state[p0]["exp_avg"])
    assert torch.equal(opt._v[o:o + p0.numel()].view(p0.shape), adam.state[p0]["exp_avg_sq"])
    assert sched.last_epoch == 7 and abs(opt.param_groups[0]["lr"] - 2e-4 * 7 / 5000) < 1e-12
    # own checkpoint: round trip incl. the scheduler state
    for _ in range(3):
        sched.step()
    own = str(tmp_path / "own.ckpt")
    cli.save_checkpoint(own, W, opt, 10, 2, sched)
    opt2 = FusedAdam(net, lr=2e-4)
    sched2 = torch.optim.lr_scheduler.LambdaLR(opt2, lambda s: min(s / 5000, 1.0))
    assert cli.load_checkpoint(own, net, ema, opt2, sched2) == (10, 2)
    assert opt2._step == 7 and torch.equal(opt2._m, opt._m) and sched2.last_epoch == 10
    assert abs(opt2.param_groups[0]["lr"] - 2e-4 * 10 / 5000) < 1e-12
    # no optimizer state at all: loud, and the step restarts with the moments
    bare = str(tmp_path / "bare.ckpt")
    torch.save({"state_dict": sd, "global_step": 9, "epoch": 1}, bare)
    opt3 = FusedAdam(net, lr=2e-4)
    opt3._step = 5
    with pytest.warns(RuntimeWarning, match="restart from"):
        cli.load_checkpoint(bare, net, ema, opt3, None)
    assert opt3._step == 0


def test_arena_and_table_cache_keep_addresses_stable():
    """Host logic of the batched parameter-gradient reductions (psld_amd/ops.py): the bump allocator hands out the SAME
    addresses after every reset (device tables hold raw pointers and are cached by content), keeps an outgrown buffer alive
    until the next reset (parked jobs point into it), and the table cache uploads a table once per distinct content."""
    from psld_amd import ops
    dev = torch.device("cpu")
    a = ops.Arena(dev, 4096)
    first = [a.alloc(1000).data_ptr(), a.floats(10, 3).data_ptr()]
    assert first[1] - first[0] == 1024 and (first[0] - a.buf.data_ptr()) % 256 == 0      # 256-byte granules of the buffer
    a.reset()
    assert [a.alloc(1000).data_ptr(), a.floats(10, 3).data_ptr()] == first
    old = a.buf
    big = a.alloc(8192)                                                     # outgrows the 4 KB buffer
    assert a.buf is not old and a.retired and a.retired[0] is old and big.numel() == 8192
    a.reset()
    assert not a.retired and a.buf.numel() >= 8192
    p1 = a.alloc(1000).data_ptr()
    a.reset()
    assert a.alloc(1000).data_ptr() == p1                                   # stable again after the growth
    t = ops.TableCache(limit=2)
    rows = [1, 2, 3, 4]
    t1 = t.get(rows, dev)
    assert t.get(list(rows), dev) is t1 and t1.dtype == torch.int64 and t1.tolist() == rows
    t2 = t.get([5, 6], dev)
    t.get([7, 8], dev)                                                      # evicts the oldest entry
    assert len(t.tabs) == 2 and t.get([5, 6], dev) is t2 and t.get(rows, dev) is not t1
    # once a hipGraph holds raw pointers into them (NCSNpp.pin_scratch): outgrown arena buffers stay alive, tables are not evicted
    a.pinned = True
    old = a.buf
    a.alloc(4 * a.buf.numel())
    a.reset()
    assert a.retired and a.retired[0] is old
    t.pinned = True
    kept = [t.get([5, 6], dev), t.get(rows, dev)]
    n0 = len(t.tabs)
    t.get([9, 9], dev)
    t.get([10, 10], dev)
    assert len(t.tabs) == n0 + 2 > t.limit and t.get([5, 6], dev) is kept[0] and t.get(rows, dev) is kept[1]
    # job rows: pointer arithmetic and the float bit pattern of alpha
    src = torch.zeros(8, 6)
    dst = torch.zeros(6)
    job = ops.param_job(src, 8, 6, 4, dst, None, 0.5, src_off=2)
    assert job[0] == src.data_ptr() + 8 and job[1:4] == (8, 6, 4) and job[4] == dst.data_ptr() and job[5] == 0
    import struct
    assert job[6] == struct.unpack("<I", struct.pack("<f", 0.5))[0]


def test_bucket_reducer_announces_a_launch_before_it_happens():
    """BucketReducer.would_launch(offset) is what lets the network reduce its parked parameter gradients BEFORE a bucket is
    exchanged: it must say yes exactly when ready_from(offset) launches something."""
    from psld_amd.ddp import BucketReducer
    red = BucketReducer(bucket_bytes=4 * 256)
    red.begin(torch.zeros(1000))
    launched = 0
    for off in (990, 900, 736, 700, 480, 300, 100, 32, 0):
        will = red.would_launch(off)
        red.ready_from(off)
        assert will == (len(red.launched) > launched), off
        launched = len(red.launched)
    assert launched == 6 and not red.would_launch(0)


def test_device_error_check_is_a_no_op_before_any_team_launch():
    from psld_amd import ops
    ops.check_device_errors()               # no slot buffer was ever allocated here: nothing to read, nothing raised


def test_bucket_timing_sources_and_missing_timing():
    """BucketReducer.stats() takes a collective's length from a pair of events of its own (side-stream form) or from the
    process group's Work (compute-stream form); a group without timing gives comm = None, never a made-up number."""
    from psld_amd.ddp import BucketReducer, _bucket_ms

    class Ev:
        def __init__(self, t):
            self.t = t

        def elapsed_time(self, other):
            return other.t - self.t

    class Work:
        def __init__(self, ms):
            self.ms = ms

        def _get_duration(self):
            if self.ms is None:
                raise RuntimeError("timing not enabled")
            return self.ms

    assert _bucket_ms((Ev(1.0), Ev(3.5))) == 2.5 and _bucket_ms(Work(0.75)) == 0.75
    assert _bucket_ms(Work(None)) != _bucket_ms(Work(None))          # NaN
    red = BucketReducer(profile=True)
    red._prof.append(([Work(0.5), Work(1.5)], (Ev(10.0), Ev(10.25))))
    red._prof.append(([(Ev(0.0), Ev(1.0)), Work(1.0)], (Ev(20.0), Ev(20.75))))
    st = red.stats()
    assert st["steps"] == 2 and st["buckets_per_step"] == 2.0
    assert abs(st["comm_ms_per_step"] - 2.0) < 1e-12 and abs(st["exposed_ms_per_step"] - 0.5) < 1e-12
    assert abs(st["hidden_ms_per_step"] - 1.5) < 1e-12
    red._prof.append(([Work(None)], (Ev(0.0), Ev(0.1))))
    st = red.stats()
    assert st["comm_ms_per_step"] is None and st["hidden_ms_per_step"] is None and abs(st["exposed_ms_per_step"] - 0.1) < 1e-12
