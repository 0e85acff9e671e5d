"""Do the parallel branches of a captured HIP graph run concurrently?  Two forked streams, each a chain of
small-grid kernels; compare replay time with the same work captured on one stream."""
import time
import torch

dev = "cuda"
a = torch.randn(64, 256, 256, device=dev)      # bmm on one 256x256 pair per chain link: ~1 workgroup-scale kernels
b = torch.randn(64, 256, 256, device=dev)
xa = [torch.randn(1, 512, 512, device=dev) for _ in range(2)]
w = torch.randn(1, 512, 512, device=dev) * 0.01
N = 200


def chain(x):
    for _ in range(N):
        x = torch.bmm(x, w)
    return x


def capture(two_streams):
    g = torch.cuda.CUDAGraph()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        if two_streams:
            s2.wait_stream(cur)
            with torch.cuda.stream(s2):
                y2 = chain(xa[1])
            y1 = chain(xa[0])
            cur.wait_stream(s2)
        else:
            y1 = chain(xa[0])
            y2 = chain(xa[1])
    return g, (y1, y2)


for _ in range(2):
    chain(xa[0])
torch.cuda.synchronize()
for two in (False, True, False, True):
    g, keep = capture(two)
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    print("graph, two streams" if two else "graph, one stream ", (time.perf_counter() - t0) / 5 * 1e3, "ms")
# eager with two streams
s2 = torch.cuda.Stream()
for two in (False, True):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        if two:
            s2.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s2):
                chain(xa[1])
            chain(xa[0])
            torch.cuda.current_stream().wait_stream(s2)
        else:
            chain(xa[0]); chain(xa[1])
    torch.cuda.synchronize()
    print("eager, two streams" if two else "eager, one stream ", (time.perf_counter() - t0) / 5 * 1e3, "ms")
