"""Captured graph with a main chain and a side chain that forks off EVERY main link (the shape of the training tape
with weight gradients on a side stream): does the replay overlap the two chains?"""
import time
import torch

dev = "cuda"
xa = [torch.randn(1, 512, 512, device=dev) for _ in range(2)]
w = torch.randn(1, 512, 512, device=dev) * 0.01
N = 200


def tape(side):
    cur = torch.cuda.current_stream()
    x = xa[0]
    outs = []
    for _ in range(N):
        x = torch.bmm(x, w)                      # main chain link
        if side is None:
            outs.append(torch.bmm(x, w))         # "weight gradient" of the link, same stream
        else:
            ev = torch.cuda.Event()
            ev.record(cur)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                outs.append(torch.bmm(x, w))
    if side is not None:
        cur.wait_stream(side)
    return x, outs


side = torch.cuda.Stream()
tape(None)
torch.cuda.synchronize()
for use_side in (False, True, False, True):
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        keep = tape(side if use_side else None)
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    print("graph, side stream" if use_side else "graph, one stream ", round((time.perf_counter() - t0) / 5 * 1e3, 3), "ms")
for use_side in (False, True):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        tape(side if use_side else None)
    torch.cuda.synchronize()
    print("eager, side stream" if use_side else "eager, one stream ", round((time.perf_counter() - t0) / 5 * 1e3, 3), "ms")
