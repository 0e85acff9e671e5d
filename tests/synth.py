"""Deterministic synthetic weights / inputs for parity tests (test infrastructure).

Golden fixtures store only inputs and expected outputs; weights are regenerated from a seed
with this recipe (on the GPU box too — same image, same torch CPU generator), so that a
97.6 M-parameter state_dict never has to be committed.  The recipe also re-randomises the
reference's zero-initialised layers (init_scale=0 -> 1e-10: layers.py:75, layerspp.py:72,233,
ncsnpp.py:281-283), without which rel-L2 on the network output is meaningless.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, Tuple

import torch


def synth_tensor(key: str, shape: Tuple[int, ...], g: torch.Generator) -> torch.Tensor:
    leaf = key.rsplit(".", 1)[-1]
    if len(shape) == 4:                      # conv OIHW
        fan_in = shape[1] * shape[2] * shape[3]
        return (torch.rand(shape, generator=g) * 2 - 1) * math.sqrt(3.0 / fan_in)
    if len(shape) == 2:
        fan_in = shape[0] if leaf == "W" else shape[1]   # NIN.W is [in,out]; Linear is [out,in]
        return (torch.rand(shape, generator=g) * 2 - 1) * math.sqrt(3.0 / fan_in)
    if len(shape) == 1:
        if leaf == "W":                      # GaussianFourierProjection.W
            return torch.randn(shape, generator=g) * 16.0
        if leaf == "weight":                 # GroupNorm gamma
            return 1.0 + 0.2 * torch.randn(shape, generator=g)
        return 0.05 * torch.randn(shape, generator=g)    # biases / beta
    raise ValueError((key, shape))


def synth_state_dict(keys_shapes: Iterable[Tuple[str, Tuple[int, ...]]], seed: int) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    return {k: synth_tensor(k, tuple(s), g) for k, s in keys_shapes}


def synth_inputs(batch: int, ch: int, size: int, seed: int):
    """x0-like in [-1,1] (util.py:25-30 range), eps ~ N(0,1), t ~ U[1e-5,1] f64."""
    g = torch.Generator().manual_seed(seed)
    x0 = torch.rand(batch, ch, size, size, generator=g) * 2 - 1
    eps = torch.randn(batch, 2 * ch, size, size, generator=g)
    t = torch.rand(batch, generator=g, dtype=torch.float64) * (1.0 - 1e-5) + 1e-5
    return x0, eps, t
