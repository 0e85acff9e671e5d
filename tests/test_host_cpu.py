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
