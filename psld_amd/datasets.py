"""``SDELatentDataset`` with the reference's interface (main/datasets/latent.py:6-28): samples of the
forward SDE's equilibrium distribution, the input of the samplers.  The reference pre-draws all
``n_samples`` latents on the CPU at construction (1.2 GB for 50 000 CIFAR latents); ``device=`` draws
them where they are consumed, and ``get_batch`` draws a fresh batch without materialising the set."""
from __future__ import annotations

from .registry import register_module


@register_module(category="datasets", name="latent")
class SDELatentDataset:
    def __init__(self, sde, config, device=None, lazy: bool = False):
        self.sde = sde
        self.device = device
        self.num_samples = config.evaluation.n_samples
        self.shape = [self.num_samples, config.data.num_channels, config.data.image_size, config.data.image_size]
        self.samples = None if lazy else self.sde.prior_sampling(self.shape, device=device)

    def get_batch(self, shape):
        return self.sde.prior_sampling(shape, device=self.device)

    def __getitem__(self, idx):
        if self.samples is None:
            return self.get_batch([1] + self.shape[1:])[0]
        return self.samples[idx]

    def __len__(self):
        return self.num_samples
