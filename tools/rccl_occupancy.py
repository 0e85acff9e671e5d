"""What will RCCL's channel workgroups cost the B=128 step?  An EMULATION on one GPU (VERDICT r04 #2c).

The 8-GPU run is the driver's.  What can be measured on a 1-GPU box is the part of the cost that does not depend on the
wire: a collective is a kernel of `blocks` workgroups that sits on CUs for bytes x 2(N-1)/N / bus-bandwidth, can only start
where a CU drains, and keeps this step's one-workgroup-per-CU kernels off those CUs while it runs.  This tool runs bench.py's
own training loop with the real BucketReducer (1-rank RCCL group: PSLD_FORCE_PG=1; same buckets, same events, same join) and
replaces every bucket's all-reduce by `cu_hog` (tests/helpers/cu_hog.hip) on a stream of its own that waits for the caller's - what
ProcessGroupNCCL does -: `blocks` x `threads` threads, `lds` bytes of LDS, spinning for the time the bucket would be on an
8-rank ring at `--busbw` GB/s while walking the bucket's bytes.  --real leaves the 1-rank RCCL collectives in place;
--side-stream forces the reducer's side-stream form (its own stream in front of the group's); --schedule-only keeps only the
per-bucket flush schedule; --variant / --side-priority bisect what a foreign queue with a pending wait costs.

    python tools/rccl_occupancy.py --blocks 32 --threads 256 --lds 32768 --busbw 250 [bench args...]

prints bench.py's JSON line (value, overlap.comm_ms_per_step / exposed_ms_per_step) with "emulation" added.  blocks = 0: the
reducer's own overhead (events, stream waits, 1-rank RCCL calls left as they are).
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_hog() -> str:
    out = os.path.join(os.environ.get("TMPDIR", "/tmp"), "libcuhog.so")
    if not os.path.exists(out):
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", out,
                        os.path.join(ROOT, "tests", "helpers", "cu_hog.hip")], check=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=32)
    ap.add_argument("--threads", type=int, default=256)
    ap.add_argument("--lds", type=int, default=32768)
    ap.add_argument("--busbw", type=float, default=250.0, help="GB/s bus bandwidth of the emulated 8-rank all-reduce")
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--touch", type=int, default=1, help="1: the hog walks the bucket's bytes; 0: spins only")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--host-sync", action="store_true", help="the host waits for the bucket's event before it issues the exchange")
    ap.add_argument("--real", action="store_true", help="leave the (1-rank) RCCL collectives in place")
    ap.add_argument("--side-stream", action="store_true", help="force the reducer's side-stream form")
    ap.add_argument("--side-priority", type=int, default=None)
    ap.add_argument("--no-profile", action="store_true", help="BucketReducer(profile=False): no timing events around the buckets")
    ap.add_argument("--schedule-only", action="store_true",
                    help="the reducer's flush schedule without its events, stream waits and collectives (what the per-bucket "
                         "reductions alone cost)")
    args, rest = ap.parse_known_args()
    os.environ["PSLD_FORCE_PG"] = "1"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    import torch                             # first: the hog library must bind to the HIP runtime torch brings
    hog = ctypes.CDLL(build_hog())
    hog.cu_hog.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                           ctypes.c_void_p]
    hog.cu_hog.restype = ctypes.c_int

    import torch.distributed as dist
    real = dist.all_reduce
    launched = {"n": 0, "usec": 0.0}

    class _Work:
        """What ProcessGroupNCCL hands back: the collective runs on the group's OWN stream, which waited for the caller's
        stream; wait() makes the caller's stream wait for its end."""

        def __init__(self, s0, s1):
            self.s0, self.s1 = s0, s1

        def wait(self):
            torch.cuda.current_stream().wait_event(self.s1)
            return True

        def _get_duration(self):
            return self.s0.elapsed_time(self.s1)

    pg_stream = {}

    def fake(tensor, op=None, group=None, async_op=False):
        if not async_op or not tensor.is_cuda or tensor.numel() < 1024:
            return real(tensor, **({} if op is None else {"op": op}), group=group, async_op=async_op)
        usec = tensor.numel() * 4 * 2.0 * (args.ranks - 1) / args.ranks / (args.busbw * 1e9) * 1e6
        launched["n"] += 1
        launched["usec"] += usec
        hs = pg_stream.setdefault("s", torch.cuda.Stream(device=tensor.device))
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        if args.host_sync:
            ev.synchronize()        # the host waits for the device to get here: the stream wait below is satisfied at once
        hs.wait_event(ev)
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record(hs)
        if args.blocks > 0:
            rc = hog.cu_hog(tensor.data_ptr(), tensor.numel() if args.touch else 0, args.blocks, args.threads, args.lds, usec,
                            hs.cuda_stream)
            assert rc == 0, rc
        s1.record(hs)
        return _Work(s0, s1)

    if not args.real:
        dist.all_reduce = fake
    if args.side_stream:
        # round 5's first design: the reducer's own side stream in front of the process group's (two foreign queues)
        from psld_amd.ddp import BucketReducer as _D
        _begin0 = _D.begin

        def begin_side(self, flat_grad):
            _begin0(self, flat_grad)
            type(self).producer_streams = property(lambda self_: [torch.cuda.current_stream()], lambda self_, v: None)
        _D.begin = begin_side
    if args.side_priority is not None:
        from psld_amd.ddp import BucketReducer as _P
        _begin = _P.begin

        def begin(self, flat_grad):
            if flat_grad.is_cuda and self._side is None:
                self._side = torch.cuda.Stream(device=flat_grad.device, priority=args.side_priority)
            _begin(self, flat_grad)
        _P.begin = begin
    if args.variant:
        # which part of BucketReducer._launch costs the time: 1 = the event record on the compute stream only, 2 = + the side
        # stream waiting for it, 3 = + the compute stream joining the side stream at the end of backward
        from psld_amd.ddp import BucketReducer as _B

        def launch(self, lo, hi):
            self.launched.append((lo, hi))
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            if args.variant >= 2:
                self._side.wait_event(ev)
            if args.variant >= 3:
                self._works.append(None)
        _B._launch = launch
    if args.schedule_only:
        from psld_amd.ddp import BucketReducer
        BucketReducer._launch = lambda self, lo, hi: self.launched.append((lo, hi))
    import bench
    sys.argv = ["bench.py", "--sample-batch", "0", "--no-cpu-baseline", "--no-forward", "--no-probe"] + rest
    import io
    from contextlib import redirect_stdout
    buf = io.StringIO()
    with redirect_stdout(buf):
        rc = bench.main()
    line = next((ln for ln in reversed(buf.getvalue().strip().splitlines()) if ln.startswith("{")), None)
    if rc or line is None:
        print(buf.getvalue())
        return rc or 1
    out = json.loads(line)
    steps = out["steps"] + out["warmup"]
    keep = {k: out.get(k) for k in ("value", "ms_per_step", "overlap", "n_gpus")}
    keep["emulation"] = {"blocks": args.blocks, "threads": args.threads, "lds_bytes": args.lds, "busbw_GBps": args.busbw,
                         "ranks": args.ranks, "touch_bytes": bool(args.touch),
                         "collectives_per_step": launched["n"] / max(1, steps),
                         "emulated_wire_ms_per_step": launched["usec"] / 1e3 / max(1, steps)}
    print(json.dumps(keep), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
