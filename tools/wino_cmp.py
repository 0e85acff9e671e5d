"""Save / compare full outputs of psld_conv3x3_wino_f32 across processes (different PSLD_WINO_PERSIST settings).
    python tools/wino_cmp.py save /tmp/a.pt ; python tools/wino_cmp.py cmp /tmp/a.pt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psld_amd import ops  # noqa: E402

DEV = "cuda"


def run():
    outs = {}
    for (b, c1, co, h, w) in [(128, 256, 256, 32, 32), (65, 128, 256, 32, 32), (128, 64, 128, 16, 16)]:
        g = torch.Generator().manual_seed(b * 7 + c1)
        x1 = torch.randn(b, h, w, c1, generator=g).to(DEV)
        wt = (torch.randn(co, c1, 3, 3, generator=g) * 0.1).to(DEV)
        uf = ops.conv3x3_wino_frag(wt, False)
        for rep in range(2):
            y = torch.full((b, h, w, co), float("nan"), device=DEV)
            ops.conv3x3_wino(x1, None, uf, co, y, ops.epilogue())
            torch.cuda.synchronize()
            outs[(b, c1, co, h, w, rep)] = y.cpu()
        outs[(b, c1, co, h, w, "x")] = x1.cpu()
    return outs


def main():
    mode, path = sys.argv[1], sys.argv[2]
    outs = run()
    for k in list(outs):
        if k[-1] == 1:
            a, b_ = outs[k[:-1] + (0,)], outs[k]
            print(k[:-1], "repeatable in-process:", torch.equal(a, b_))
    if mode == "save":
        torch.save(outs, path)
        return
    ref = torch.load(path)
    for k, v in outs.items():
        r = ref[k]
        if torch.equal(r, v):
            print(k, "EQUAL")
            continue
        d = (r - v).abs()
        nz = d > 0
        print(k, "DIFFER: n =", int(nz.sum()), "of", d.numel(), "max", float(d.max()), "nan", int(torch.isnan(v).sum()))
        if k[-1] != "x":
            idx = nz.nonzero()
            for dim, name in enumerate(["img", "y", "x", "ch"]):
                vals, cnt = idx[:, dim].unique(return_counts=True)
                print("   ", name, "distinct", len(vals), "first", vals[:12].tolist(), "counts", cnt[:12].tolist())


if __name__ == "__main__":
    main()
