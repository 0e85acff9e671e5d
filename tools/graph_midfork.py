"""Captured graph: main chain of N small kernels; every G links a side chain of G kernels forks off (one cross edge per
group), joined at the end.  Does replay overlap the side groups with the main chain?"""
import sys
import time
import torch

dev = "cuda"
x0 = torch.randn(1, 512, 512, device=dev)
w = torch.randn(1, 512, 512, device=dev) * 0.01
N = 240


def tape(side, group):
    cur = torch.cuda.current_stream()
    x = x0
    outs, queue = [], []
    def flush():
        if not queue:
            return
        ev = torch.cuda.Event(); ev.record(cur); side.wait_event(ev)
        with torch.cuda.stream(side):
            for q in queue:
                outs.append(torch.bmm(q, w))
        queue.clear()
    for i in range(N):
        x = torch.bmm(x, w)
        if side is None:
            outs.append(torch.bmm(x, w))
        else:
            queue.append(x)
            if len(queue) >= group:
                flush()
    if side is not None:
        flush()
        cur.wait_stream(side)
    return x, outs


side = torch.cuda.Stream()
tape(None, 1)
torch.cuda.synchronize()
for group in (0, 1, 8, 40, 120, 240):
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        keep = tape(side if group else None, group)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    print(f"graph  group={group:4d}", round((time.perf_counter() - t0) / 5 * 1e3, 3), "ms")
for group in (0, 1, 8, 40, 240):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        tape(side if group else None, group)
    torch.cuda.synchronize()
    print(f"eager  group={group:4d}", round((time.perf_counter() - t0) / 5 * 1e3, 3), "ms")
