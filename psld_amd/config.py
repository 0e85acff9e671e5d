"""Attribute-style config node accepted by every plug-in constructor.

The reference passes the Hydra/OmegaConf node ``config.dataset.diffusion`` to every
class and only ever uses attribute access on it (SURVEY.md Appendix C;
reference main/train_sde.py:25, main/eval/sample.py:31).  OmegaConf is not
available on the GPU box, so this module provides a tiny attr-dict with the same
access pattern plus the two north-star presets.  An OmegaConf ``DictConfig`` works
unchanged wherever a ``Config`` is accepted.

Key names and default values follow main/configs/dataset/cifar10/cifar10_psld.yaml:1-99;
``c10_sota`` applies the overrides of
scripts_psld/sota/uncond/cifar10/train_uncond_psld.sh:6-21 and ``celeba64_sota``
those of scripts_psld/sota/uncond/celeba64/train_uncond_psld.sh:7-20.
"""
from __future__ import annotations

import copy


class Config(dict):
    """dict with recursive attribute access (``cfg.model.score_fn.nf``)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        for k, v in list(self.items()):
            if isinstance(v, dict) and not isinstance(v, Config):
                self[k] = Config(v)

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e

    def __setattr__(self, name, value):
        if isinstance(value, dict) and not isinstance(value, Config):
            value = Config(value)
        self[name] = value

    def __deepcopy__(self, memo):
        return Config(copy.deepcopy(dict(self), memo))

    def override(self, dotted: str, value):
        """``cfg.override("model.score_fn.nf", 64)`` — Hydra-style dotted override."""
        node = self
        parts = dotted.split(".")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = Config(value) if isinstance(value, dict) else value
        return self


def _yaml_defaults() -> dict:
    return {
        "data": {
            "root": "", "name": "cifar10", "image_size": 32, "hflip": True,
            "num_channels": 3, "norm": True, "return_target": False,
        },
        "model": {
            "pl_module": "sde_wrapper",
            "score_fn": {
                "name": "ncsnpp", "in_ch": 6, "out_ch": 6, "nonlinearity": "swish",
                "nf": 128, "ch_mult": [1, 2, 2, 2], "num_res_blocks": 4,
                "attn_resolutions": [16], "dropout": 0.1, "resamp_with_conv": True,
                "noise_cond": True, "fir": False, "fir_kernel": [1, 3, 3, 1],
                "skip_rescale": True, "resblock_type": "biggan", "progressive": "none",
                "progressive_input": "none", "progressive_combine": "sum",
                "embedding_type": "positional", "init_scale": 0.0, "fourier_scale": 16,
            },
            "sde": {
                "name": "psld", "beta_min": 8.0, "beta_max": 8.0, "nu": 4.01,
                "gamma": 0.01, "kappa": 0.04, "decomp_mode": "lower",
                "numerical_eps": 1e-9, "n_timesteps": 1000, "is_augmented": True,
            },
        },
        "training": {
            "seed": 0, "continuous": True, "mode": "hsm",
            "loss": {"name": "psld_score_loss", "l_type": "l2", "reduce_mean": True,
                     "weighting": "fid"},
            "optimizer": {"name": "Adam", "lr": 2e-4, "beta_1": 0.9, "beta_2": 0.999,
                          "weight_decay": 0, "eps": 1e-8, "warmup": 5000, "grad_clip": 1.0},
            "train_eps": 1e-5, "fp16": False, "use_ema": True, "ema_decay": 0.9999,
            "batch_size": 32, "epochs": 5000, "log_step": 1, "accelerator": "gpu",
            "devices": [0], "chkpt_interval": 1, "restore_path": "", "results_dir": "",
            "workers": 1, "chkpt_prefix": "",
        },
        "evaluation": {
            "sampler": {"name": "em_sde"}, "seed": 0, "chkpt_path": "", "save_path": "",
            "n_discrete_steps": 1000, "denoise": True, "eval_eps": 1e-3,
            "stride_type": "uniform", "use_pflow": False, "sample_from": "target",
            "accelerator": "gpu", "devices": [0], "n_samples": 50000, "workers": 2,
            "batch_size": 64, "save_mode": "image", "sample_prefix": "gpu",
            "path_prefix": "",
        },
    }


def yaml_default() -> Config:
    """The shipped YAML defaults (ch_mult=[1,2,2,2], nres=4, no fir, positional)."""
    return Config(_yaml_defaults())


def c10_sota() -> Config:
    """CIFAR-10 SOTA net: BASELINE.json configs[0..2,4] (SURVEY.md §8 'C10-SOTA')."""
    c = yaml_default()
    sf = c.model.score_fn
    sf.ch_mult = [2, 2, 2]
    sf.num_res_blocks = 8
    sf.dropout = 0.15
    sf.progressive_input = "residual"
    sf.fir = True
    sf.embedding_type = "fourier"
    c.training.batch_size = 16
    return c


def celeba64_sota() -> Config:
    """CelebA-64 net: BASELINE.json configs[3]."""
    c = yaml_default()
    c.data.name = "celeba"
    c.data.image_size = 64
    sf = c.model.score_fn
    sf.ch_mult = [1, 2, 2, 2]
    sf.num_res_blocks = 4
    sf.dropout = 0.1
    sf.progressive_input = "residual"
    sf.fir = True
    sf.embedding_type = "fourier"
    c.model.sde.nu = 4.005
    c.model.sde.gamma = 0.005
    c.training.batch_size = 16
    return c


def tiny(image_size: int = 16, nf: int = 32, ch_mult=(1, 2), num_res_blocks: int = 1,
         attn_resolutions=(8,)) -> Config:
    """Reduced net with every block kind of C10-SOTA (used by parity tests/smoke)."""
    c = c10_sota()
    c.data.image_size = image_size
    sf = c.model.score_fn
    sf.nf = nf
    sf.ch_mult = list(ch_mult)
    sf.num_res_blocks = num_res_blocks
    sf.attn_resolutions = list(attn_resolutions)
    sf.dropout = 0.0
    return c


def tiny_vpsde() -> Config:
    """Reduced VP-SDE baseline net (keys of main/configs/dataset/cifar10/cifar10_vpsde.yaml:10-40)."""
    c = tiny()
    sf = c.model.score_fn
    sf.in_ch = sf.out_ch = 3
    sf.fir = False
    sf.progressive_input = "none"
    sf.embedding_type = "positional"
    c.model.sde = Config({"name": "vpsde", "beta_min": 0.1, "beta_max": 20.0, "n_timesteps": 1000,
                          "is_augmented": False})
    c.training.loss.name = "score_loss"
    return c


# ---- classifier guidance: the ``clf`` node of cifar10_psld.yaml:101-170 -------------------------
def _clf_yaml_defaults() -> dict:
    return {
        "data": {"root": "", "name": "cifar10", "image_size": 32, "hflip": True, "num_channels": 3, "norm": True,
                 "return_target": True},
        "model": {
            "pl_module": "tclf_wrapper",
            "clf_fn": {
                "name": "ncsnpp_clf", "in_ch": 6, "nonlinearity": "swish", "nf": 128, "ch_mult": [1, 2, 2, 2],
                "num_res_blocks": 2, "attn_resolutions": [16], "dropout": 0.1, "resamp_with_conv": True,
                "noise_cond": True, "fir": False, "fir_kernel": [1, 3, 3, 1], "skip_rescale": True,
                "resblock_type": "biggan", "progressive": "none", "progressive_input": "none",
                "progressive_combine": "sum", "embedding_type": "positional", "init_scale": 0.0,
                "fourier_scale": 16, "n_cls": 10,
            },
        },
        "training": {
            "seed": 0, "continuous": True,
            "loss": {"name": "tce_loss", "l_type": "l2", "reduce_mean": True},
            "optimizer": {"name": "Adam", "lr": 2e-4, "beta_1": 0.9, "beta_2": 0.999, "weight_decay": 0,
                          "eps": 1e-8, "warmup": 5000},
            "fp16": False, "batch_size": 32, "epochs": 500, "log_step": 1, "accelerator": "gpu", "devices": [0],
            "chkpt_interval": 1, "restore_path": "", "results_dir": "", "workers": 1, "chkpt_prefix": "",
        },
        "evaluation": {"seed": 0, "chkpt_path": "", "accelerator": "gpu", "devices": [0], "workers": 1,
                       "batch_size": 64, "clf_temp": 1.0, "label_to_sample": 0},
    }


def clf_default() -> Config:
    """The shipped ``clf`` YAML node (n_cls is ``???`` there; 10 = CIFAR-10)."""
    return Config(_clf_yaml_defaults())


def clf_c10() -> Config:
    """Classifier of scripts_psld/ablations/cond/cifar10/{train,sample}_tclf_psld.sh."""
    c = clf_default()
    cf = c.model.clf_fn
    cf.ch_mult = [1, 2, 3, 4]
    cf.num_res_blocks = 4
    cf.attn_resolutions = [16, 8]
    cf.n_cls = 10
    return c


def tiny_clf(image_size: int = 16, nf: int = 32, ch_mult=(1, 2), num_res_blocks: int = 1,
             attn_resolutions=(8,), n_cls: int = 10) -> Config:
    """Reduced classifier with every block kind of the shipped one (parity tests)."""
    c = clf_default()
    c.data.image_size = image_size
    cf = c.model.clf_fn
    cf.nf, cf.ch_mult, cf.num_res_blocks = nf, list(ch_mult), num_res_blocks
    cf.attn_resolutions, cf.n_cls, cf.dropout = list(attn_resolutions), n_cls, 0.0
    return c


def with_clf(diffusion: Config, clf: Config) -> Config:
    """Root node of the classifier-guidance apps (``config.dataset`` in main/train_clf.py:27 and
    main/eval/class_cond_sample.py:33): ``.diffusion`` and ``.clf`` side by side."""
    return Config({"diffusion": diffusion, "clf": clf})
