"""SHA-1 of the outputs of psld_conv3x3_wino_f32 over shapes x epilogue forms (seeded inputs): two runs under different
PSLD_WINO_PERSIST settings must print the same lines - the CU-resident kernel keeps wino_conv8s_kernel's arithmetic order.
    python tools/wino_digest.py [--small]"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psld_amd import ops  # noqa: E402

DEV = "cuda"


def digest(t):
    return hashlib.sha1(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:16]


def main():
    small = "--small" in sys.argv
    if small:     # for PSLD_WINO_PERSIST=<few workgroups>
        shapes = [(8, 64, 0, 128, 32, 32), (16, 32, 32, 256, 16, 16), (33, 64, 0, 128, 8, 8), (2, 32, 0, 128, 64, 64),
                  (17, 32, 0, 128, 4, 8)]
    else:
        shapes = [(128, 256, 0, 256, 32, 32), (128, 256, 256, 256, 16, 16), (511, 64, 0, 256, 8, 8), (16, 64, 0, 128, 64, 64),
                  (65, 128, 0, 256, 32, 32)]
    for (b, c1, c2, co, h, w) in shapes:
        g = torch.Generator().manual_seed(b * 7 + c1)
        x1 = torch.randn(b, h, w, c1, generator=g).to(DEV)
        x2 = torch.randn(b, h, w, c2, generator=g).to(DEV) if c2 else None
        wt = (torch.randn(co, c1 + c2, 3, 3, generator=g) * 0.1).to(DEV)
        uf = ops.conv3x3_wino_frag(wt, False)
        bias = torch.randn(co, generator=g).to(DEV)
        res = torch.randn(b, h, w, co, generator=g).to(DEV)
        temb = torch.randn(b, co, generator=g).to(DEV)
        prev = torch.randn(b, h, w, co, generator=g).to(DEV)
        forms = {
            "plain": dict(),
            "bias": dict(bias=bias),
            "full": dict(bias=bias, rowbias=temb, rows_per_img=h * w, residual=res, ld_residual=co, out_scale=0.7),
            "acc": dict(accumulate=True, out_scale=0.5),
        }
        for name, kw in forms.items():
            y = prev.clone() if name == "acc" else torch.full((b, h, w, co), float("nan"), device=DEV)
            ops.conv3x3_wino(x1, x2, uf, co, y, ops.epilogue(**kw))
            torch.cuda.synchronize()
            assert torch.isfinite(y).all(), (b, c1, c2, co, h, w, name)
            print(f"{b}x{c1}+{c2}->{co}@{h}x{w} {name}: {digest(y)}")
        if (h * w) % 64 == 0:
            gp = torch.zeros(b * (h * w // 64) * (co // 8) * 2, dtype=torch.float64, device=DEV)
            y = torch.empty(b, h, w, co, device=DEV)
            ops.conv3x3_wino(x1, x2, uf, co, y, ops.epilogue(bias=bias, gn_part=gp, gn_hw=h * w))
            torch.cuda.synchronize()
            print(f"{b}x{c1}+{c2}->{co}@{h}x{w} gn: {digest(y)} {digest(gp)}")


if __name__ == "__main__":
    main()
