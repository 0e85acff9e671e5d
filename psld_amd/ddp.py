"""Data-parallel gradient exchange: bucketed all-reduce over RCCL/xGMI overlapped with backward.

Replaces Lightning's ``strategy="ddp"`` (main/train_sde.py:114 -> torch DDP -> NCCL).  One process
per GPU; parameters, Adam state and the EMA copy are replicated; the only exchange per step is the
mean of the flat fp32 gradient buffer (390.5 MB for the CIFAR-10 net).

The network's backward produces parameter gradients from the END of the flat buffer towards the
front (reverse module order) and reports a watermark; every bucket that lies entirely above the
watermark is all-reduced immediately, off the compute stream, while the remaining layers' dgrad/wgrad
kernels keep that stream busy: issued from the compute stream itself (the process group runs it on its own
stream, which waits for that point) when every gradient is produced there, or behind a side stream of the
reducer's that waits for both producers when the pass runs its weight gradients on a second stream.
``finish()`` flushes the rest and makes the compute stream wait.  xGMI is point-to-point, so buckets are
large (default 64 MiB, geometric from the front: 8 collectives per step for the CIFAR-10 net).

The same class runs on CPU tensors with the gloo backend (no streams) — that is how the N>1 path is
tested in this container (tests/test_ddp_cpu.py).
"""
from __future__ import annotations

import collections
import datetime

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


class RendezvousTimeout(RuntimeError):
    """Not every rank reached the rendezvous in time; ``missing`` lists the ranks that did not (as far as this rank can
    tell: rank 0 hosts the store and sees everyone, another rank may only know that rank 0 never answered)."""

    def __init__(self, msg: str, missing: List[int]):
        super().__init__(msg)
        self.missing = missing


def _rendezvous(rank: int, world: int, timeout_s: float):
    """The env:// store of this job (torchrun's agent store when there is one, else a TCPStore hosted by rank 0) with
    every rank checked in: each rank writes ``psld/arrived/<rank>`` and waits for the others.  A rank that never shows
    up is NAMED after ``timeout_s`` instead of leaving the job to hang in the first collective."""
    import datetime
    import time
    timeout = datetime.timedelta(seconds=timeout_s)
    # torchrun's agent hosts the store itself (TORCHELASTIC_USE_AGENT_STORE): then every rank is a client
    hosted_by_agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "False") == "True"
    try:
        store = dist.TCPStore(os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]), world_size=world,
                              is_master=(rank == 0 and not hosted_by_agent), timeout=timeout, wait_for_workers=False)
    except Exception as e:  # noqa: BLE001 - a client that cannot reach rank 0's store
        raise RendezvousTimeout(f"rank {rank}: no rendezvous store at {os.environ.get('MASTER_ADDR')}:"
                                f"{os.environ.get('MASTER_PORT')} within {timeout_s:.0f} s (rank 0 never arrived?): {e!r}",
                                [0]) from e
    store.set_timeout(timeout)
    # the agent store of a restarted torchrun job outlives its workers: keys carry the restart count
    gen = os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
    key = lambda r: f"psld/arrived/{gen}/{r}"  # noqa: E731
    store.set(key(rank), "1")
    deadline = time.monotonic() + timeout_s
    missing = [r for r in range(world) if r != rank]
    while missing:
        missing = [r for r in missing if not store.check([key(r)])]
        if not missing:
            break
        if time.monotonic() > deadline:
            raise RendezvousTimeout(f"rank {rank}: rank(s) {missing} of {world} did not reach the rendezvous within "
                                    f"{timeout_s:.0f} s", missing)
        time.sleep(0.05)
    return store, timeout


def init_distributed(backend: Optional[str] = None, force: bool = False,
                     timeout_s: Optional[float] = None, pg_timeout_s: Optional[float] = None) -> Tuple[int, int, int]:
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun); returns (rank, local_rank, world).
    ``force`` creates the process group even for world size 1.  ``timeout_s`` (default PSLD_DIST_TIMEOUT_S or 180)
    bounds the RENDEZVOUS - a rank that never joins raises ``RendezvousTimeout`` naming it.  ``pg_timeout_s`` (default
    PSLD_PG_TIMEOUT_S or 600, never below the rendezvous timeout) is the process group's watchdog for every later
    collective: its own, more generous value (rank-0-only work such as a checkpoint on a slow filesystem, or
    uneven sampling shards in front of a barrier, must not trip a 3-minute watchdog)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if timeout_s is None:
        timeout_s = float(os.environ.get("PSLD_DIST_TIMEOUT_S", "180"))
    if pg_timeout_s is None:
        pg_timeout_s = float(os.environ.get("PSLD_PG_TIMEOUT_S", "600"))
    pg_timeout_s = max(pg_timeout_s, timeout_s)
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            if world == 1:          # single-rank rehearsal: any free port (a fixed one may still be held by a previous run)
                import socket
                with socket.socket() as s:
                    s.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(s.getsockname()[1])
            else:
                os.environ["MASTER_PORT"] = "29500"
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" is RCCL on ROCm
        store, timeout = _rendezvous(rank, world, timeout_s)       # before any HIP call: a missing rank costs no GPU state
        if backend == "nccl":
            torch.cuda.set_device(local)
            # Work._get_duration(): BucketReducer(profile=True) reads the time each collective occupied the process
            # group's stream from it (start / end events the group records itself)
            os.environ.setdefault("TORCH_NCCL_ENABLE_TIMING", "1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=pg_timeout_s),
                                store=dist.PrefixStore("psld/pg", store))
    return rank, local, world


class BucketReducer:
    def __init__(self, process_group=None, bucket_bytes: int = 64 << 20, average: bool = True,
                 force_collective: bool = False, profile: bool = False, first_bucket_bytes: Optional[int] = None):
        self.pg = process_group
        # profile: bracket every collective with HIP events on the side stream and the join in finish() with events
        # on the compute stream -> stats(): how much of the exchange ran hidden under backward (bench.py "overlap")
        self.profile = profile
        self._prof = collections.deque(maxlen=4096)   # (bounded: stats() drains it) per finished backward: ([(start, end) per bucket], (join0, join1)) or host seconds
        self._cur = None
        self.bucket_elems = max(1, bucket_bytes // 4)
        # Geometric sizes from the FRONT of the flat buffer: gradients are produced from its end towards its front, so the
        # front bucket (stem-side parameters) is launched last and is the one collective nothing can hide - keep it small
        # (default 1/8 of bucket_bytes: 8 MiB) and double up to bucket_bytes towards the end, where the collectives run
        # under the rest of backward.
        self.first_elems = max(1, (first_bucket_bytes if first_bucket_bytes is not None else bucket_bytes // 8) // 4)
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
            self._bounds = self.make_bounds(n, self.first_elems, self.bucket_elems)
        self._next = len(self._bounds) - 1
        self._works = []
        self.launched = []
        self._cur = [] if self.profile else None
        if flat_grad.is_cuda and self._side is None:
            self._side = torch.cuda.Stream(device=flat_grad.device)

    @staticmethod
    def make_bounds(n: int, first: int, full: int) -> List[Tuple[int, int]]:
        """[lo, hi) of the buckets, front to back: sizes first, 2 first, 4 first, ... capped at ``full``; a remainder
        shorter than half a bucket joins its neighbour."""
        bounds, lo, size = [], 0, min(first, full)
        while lo < n:
            hi = min(n, lo + size)
            if n - hi < size // 2:
                hi = n
            bounds.append((lo, hi))
            lo, size = hi, min(full, size * 2)
        return bounds

    def _launch(self, lo: int, hi: int):
        self.launched.append((lo, hi))
        if self.world == 1 and not self.force:
            return
        view = self._flat[lo:hi]
        if self._flat.is_cuda and dist.get_backend(self.pg) == "nccl" and not self.producer_streams:
            # Every gradient of the bucket was enqueued on the CURRENT stream: issue the collective from it.  The process
            # group makes its own stream wait for this point and runs the collective there, beside the rest of backward;
            # finish() joins.  One foreign queue with a pending wait instead of two (a side stream of ours in front of the
            # group's): measured on MI355X, every queue that holds an unsatisfied wait while the compute stream
            # dispatches costs the step ~0.5-1 ms (profiles/r05/rccl_occupancy.txt).
            op = dist.ReduceOp.AVG if self.average else dist.ReduceOp.SUM
            w = dist.all_reduce(view, op=op, group=self.pg, async_op=True)
            self._works.append(w)
            if self._cur is not None:
                self._cur.append(w)
        elif self._flat.is_cuda and dist.get_backend(self.pg) == "nccl":
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self._side.wait_event(ev)
            for ps in self.producer_streams:
                pe = torch.cuda.Event()
                pe.record(ps)
                self._side.wait_event(pe)
            with torch.cuda.stream(self._side):
                if self._cur is not None:
                    s0 = torch.cuda.Event(enable_timing=True)
                    s0.record(self._side)
                # RCCL averages inside the collective (ReduceOp.AVG): no separate scaling pass over the bucket
                op = dist.ReduceOp.AVG if self.average else dist.ReduceOp.SUM
                w = dist.all_reduce(view, op=op, group=self.pg, async_op=True)
                w.wait()                 # stream-level: orders the side stream after the collective, no host wait
                if self._cur is not None:
                    s1 = torch.cuda.Event(enable_timing=True)
                    s1.record(self._side)
                    self._cur.append((s0, s1))
            self._works.append(w)
        else:
            import time as _time
            if self._flat.is_cuda:
                torch.cuda.current_stream().synchronize()   # gloo reads the buffer from the host side
                for ps in self.producer_streams:
                    ps.synchronize()
            t0 = _time.perf_counter()
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg)   # gloo has no AVG
            if self.average:
                view.mul_(1.0 / self.world)
            if self._cur is not None:
                if self._flat.is_cuda:
                    torch.cuda.current_stream().synchronize()
                self._cur.append(_time.perf_counter() - t0)

    def would_launch(self, offset: int) -> bool:
        """Would ``ready_from(offset)`` exchange a bucket now?  (The network reduces its parked parameter gradients first.)"""
        return self._next >= 0 and self._bounds[self._next][0] >= offset

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
            cur = torch.cuda.current_stream()
            j0 = j1 = None
            if self._cur is not None:
                j0 = torch.cuda.Event(enable_timing=True)
                j0.record(cur)
            if self.producer_streams:
                cur.wait_stream(self._side)
            else:
                for w in self._works:
                    w.wait()             # stream-level: the compute stream waits for the collective's end event
            if self._cur is not None:
                j1 = torch.cuda.Event(enable_timing=True)
                j1.record(cur)
                self._prof.append((self._cur, (j0, j1)))
        elif self._cur:
            self._prof.append((self._cur, None))
        self._cur = None
        self._works = []

    def stats(self, reset: bool = True):
        """Exchange timing over the backward passes since the last reset (``profile=True``): per step the number of
        buckets, the time the collectives occupied their stream (``comm_ms``: the reducer's side stream, or the process
        group's own when the exchange is issued from the compute stream), the time the compute stream
        stood waiting for them at the join (``exposed_ms``) and the difference (``hidden_ms``: ran under backward).
        Host-synchronous backends (gloo) expose everything.  Call after a device synchronize."""
        steps = len(self._prof)
        if steps == 0:
            return None
        buckets = comm = exposed = 0.0
        for evs, join in self._prof:
            buckets += len(evs)
            if join is None:
                c = 1e3 * sum(evs)
                comm += c
                exposed += c
            else:
                comm += sum(_bucket_ms(e) for e in evs)
                exposed += join[0].elapsed_time(join[1])
        if reset:
            self._prof.clear()
        if comm != comm:            # a collective whose length could not be read (no timing in the process group)
            return {"steps": steps, "buckets_per_step": buckets / steps, "bucket_mb": self.bucket_elems * 4 / 2 ** 20,
                    "comm_ms_per_step": None, "exposed_ms_per_step": exposed / steps, "hidden_ms_per_step": None}
        return {"steps": steps, "buckets_per_step": buckets / steps, "bucket_mb": self.bucket_elems * 4 / 2 ** 20,
                "comm_ms_per_step": comm / steps, "exposed_ms_per_step": exposed / steps,
                "hidden_ms_per_step": max(0.0, comm - exposed) / steps}


def _bucket_ms(entry) -> float:
    """Time one collective occupied its stream: a pair of HIP events of ours (side-stream form) or the process group's own
    start / end events (Work._get_duration, needs TORCH_NCCL_ENABLE_TIMING=1 - init_distributed sets it)."""
    if isinstance(entry, tuple):
        return entry[0].elapsed_time(entry[1])
    try:
        return float(entry._get_duration())
    except Exception:       # noqa: BLE001 - timing not enabled in this process group: the exchange ran, its length is unknown
        return float("nan")


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard of ``n_items`` independent samples for this rank (sampling path:
    no collective, eval/sample.py:108-109 shards the latent dataset the same way)."""
    per = (n_items + world - 1) // world
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)
