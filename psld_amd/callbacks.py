"""Callbacks with the reference's names (main/callbacks.py): ``EMAWeightUpdate`` (one fused kernel over the
flat parameter buffers, :17-64) and ``SimpleImageWriter`` (:67-124) whose float64 -> uint8 conversion runs
on the device (16x less device->host traffic); PNG encoding stays on the host."""
from __future__ import annotations

import os

import numpy as np
import torch

from . import ops
from .optim import EMAWeightUpdate  # noqa: F401  (re-export under the reference's module name)


class SimpleImageWriter:
    def __init__(self, output_dir, write_interval="batch", sample_prefix="", path_prefix="", save_mode="image",
                 is_norm=True, is_augmented=True):
        self.output_dir = output_dir
        self.interval = write_interval
        self.sample_prefix = sample_prefix
        self.path_prefix = path_prefix
        self.save_mode = save_mode
        self.is_norm = is_norm
        self.is_augmented = is_augmented

    def write_on_batch_end(self, trainer, pl_module, prediction, batch_indices, batch, batch_idx, dataloader_idx=0):
        rank = getattr(pl_module, "global_rank", 0)
        base = os.path.join(self.output_dir, str(self.path_prefix)) if self.path_prefix != "" else self.output_dir
        img_dir = os.path.join(base, "images")
        os.makedirs(img_dir, exist_ok=True)
        stem = os.path.join(img_dir, f"output_{self.sample_prefix }_{rank}_{batch_idx}")     # callbacks.py:120-122
        pred = prediction if prediction.dtype == torch.float64 else prediction.double()
        if self.save_mode == "image":
            u8 = ops.samples_to_uint8(pred.contiguous(), self.is_augmented, self.is_norm).cpu().numpy()
            from PIL import Image
            for i, im in enumerate(u8):
                Image.fromarray(im).save(stem + "_%d.png" % i, "png")                        # util.py:147-158
        else:
            # save_as_np (util.py:161-169): per-sample min/max normalised float arrays, HWC
            x = pred.cpu()
            if self.is_augmented:
                x, _ = torch.chunk(x, 2, dim=1)
            x = x.clone()
            if self.is_norm:
                b, c, h, w = x.shape
                for ch in range(3):
                    v = x[:, ch].reshape(b, -1)
                    v -= v.min(1, keepdim=True)[0]
                    v /= (v.max(1, keepdim=True)[0] - v.min(1, keepdim=True)[0])
                    x[:, ch] = v.reshape(b, h, w)
            for i, out in enumerate(x.permute(0, 2, 3, 1).contiguous().numpy()):
                np.save(stem + "_%d.npy" % i, out)
