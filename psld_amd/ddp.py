"""Data-parallel gradient exchange: bucketed all-reduce over RCCL/xGMI overlapped with backward.

Replaces Lightning's ``strategy="ddp"`` (main/train_sde.py:114 -> torch DDP -> NCCL).  One process
per GPU; parameters, Adam state and the EMA copy are replicated; the only exchange per step is the
mean of the flat fp32 gradient buffer (390.5 MB for the CIFAR-10 net).

The network's backward produces parameter gradients from the END of the flat buffer towards the
front (reverse module order) and reports a watermark; every bucket that lies entirely above the
watermark is all-reduced immediately on a side HIP stream while the remaining layers' dgrad/wgrad
kernels keep the compute stream busy.  ``finish()`` flushes the rest and makes the compute stream
wait.  xGMI is point-to-point, so buckets are large (default 64 MiB: ~6 collectives per step).

The same class runs on CPU tensors with the gloo backend (no streams) — that is how the N>1 path is
tested in this container (tests/test_ddp_cpu.py).
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None, force: bool = False) -> Tuple[int, int, int]:
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun); returns (rank, local_rank, world).
    ``force`` creates the process group even for world size 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" is RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


class BucketReducer:
    def __init__(self, process_group=None, bucket_bytes: int = 64 << 20, average: bool = True,
                 force_collective: bool = False):
        self.pg = process_group
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.average = average
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # force_collective: issue the collectives even for a single-rank group (exercises the RCCL +
        # side-stream path on a 1-GPU box)
        self.force = force_collective and dist.is_initialized()
        self._flat = None
        self._bounds: List[Tuple[int, int]] = []
        self._next = -1
        self._side = None
        self._works = []
        self.launched: List[Tuple[int, int]] = []   # (lo, hi) in launch order (inspected by tests)
        self.producer_streams = []   # extra streams that write gradients (the network's wgrad stream)

    def begin(self, flat_grad: torch.Tensor):
        self._flat = flat_grad
        n = flat_grad.numel()
        if not self._bounds or self._bounds[-1][1] != n:
            self._bounds = [(lo, min(n, lo + self.bucket_elems)) for lo in range(0, n, self.bucket_elems)]
        self._next = len(self._bounds) - 1
        self._works = []
        self.launched = []
        if flat_grad.is_cuda and self._side is None:
            self._side = torch.cuda.Stream(device=flat_grad.device)

    def _launch(self, lo: int, hi: int):
        self.launched.append((lo, hi))
        if self.world == 1 and not self.force:
            return
        view = self._flat[lo:hi]
        if self._flat.is_cuda and dist.get_backend(self.pg) == "nccl":
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self._side.wait_event(ev)
            for ps in self.producer_streams:
                pe = torch.cuda.Event()
                pe.record(ps)
                self._side.wait_event(pe)
            with torch.cuda.stream(self._side):
                w = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                if self.average:
                    w.wait()             # stream-level wait on the side stream only
                    view.mul_(1.0 / self.world)
            self._works.append(w)
        else:
            if self._flat.is_cuda:
                torch.cuda.current_stream().synchronize()   # gloo reads the buffer from the host side
                for ps in self.producer_streams:
                    ps.synchronize()
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg)
            if self.average:
                view.mul_(1.0 / self.world)

    def ready_from(self, offset: int):
        """All gradients at flat offsets >= ``offset`` are final."""
        while self._next >= 0 and self._bounds[self._next][0] >= offset:
            lo, hi = self._bounds[self._next]
            self._launch(lo, hi)
            self._next -= 1

    def finish(self):
        while self._next >= 0:
            lo, hi = self._bounds[self._next]
            self._launch(lo, hi)
            self._next -= 1
        if self._flat is not None and self._flat.is_cuda and (self.world > 1 or self.force) and self._works:
            for w in self._works:
                w.wait()
            torch.cuda.current_stream().wait_stream(self._side)
        self._works = []


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard of ``n_items`` independent samples for this rank (sampling path:
    no collective, eval/sample.py:108-109 shards the latent dataset the same way)."""
    per = (n_items + world - 1) // world
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)
