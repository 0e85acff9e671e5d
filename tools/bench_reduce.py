"""Split-K slab reductions: per-layer psld_reduce_slabs_f32 launches vs ONE psld_reduce_slabs_batch_f32 launch.
    python tools/bench_reduce.py [jobs]"""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psld_amd import ops

dev = torch.device("cuda")
njobs = int(sys.argv[1]) if len(sys.argv) > 1 else 24
shapes = [(256, 9, 256, 11), (256, 9, 512, 6), (256, 1, 512, 8), (256, 1, 256, 16)]
jobs, rows, units, byts = [], [], 0, 0
keep = []
for i in range(njobs):
    co, taps, ci, ns = shapes[i % len(shapes)]
    n = co * taps * ci
    slabs = torch.randn(ns, n, device=dev)
    out = torch.empty(n, device=dev)
    layout = 1 if taps == 9 else 0
    j = ops.slab_job(slabs, ns, n, out, layout, taps, ci, 1.0)
    u = ops.slab_units(n, layout, taps, ci)
    rows += list(j) + [units, u]
    units += u
    byts += 4 * n * (ns + 1)
    jobs.append((slabs, ns, n, out, layout, co, taps, ci))
    keep.append(slabs)
table = torch.tensor(rows, dtype=torch.int64, device=dev)


def timeit(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e-3


def per_layer():
    for slabs, ns, n, out, layout, co, taps, ci in jobs:
        ops.reduce_slabs(slabs, ns, n, out, layout=layout, cout=co, taps=taps, cin=ci)


t1 = timeit(per_layer)
t2 = timeit(lambda: ops.reduce_slabs_batch(table, njobs, units))
print(f"{njobs} jobs, {byts / 1e9:.2f} GB: per layer {t1 * 1e6:.0f} us = {byts / t1 / 1e12:.2f} TB/s | one launch {t2 * 1e6:.0f} us = {byts / t2 / 1e12:.2f} TB/s")
