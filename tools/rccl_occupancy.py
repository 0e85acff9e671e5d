"""What will RCCL's channel workgroups cost the B=128 step?  An EMULATION on one GPU (VERDICT r04 #2c).

The 8-GPU run is the driver's.  What can be measured on a 1-GPU box is the part of the cost that does not depend on the
wire: a collective is a kernel of `blocks` workgroups that sits on CUs for bytes x 2(N-1)/N / bus-bandwidth, can only start
where a CU drains, and keeps this step's one-workgroup-per-CU kernels off those CUs while it runs.  This tool runs bench.py's
own training loop with the real BucketReducer (1-rank RCCL group: PSLD_FORCE_PG=1; same buckets, same side stream, same
events, same join) and replaces every bucket's all-reduce by `cu_hog` (tools/cu_hog.hip): `blocks` x `threads` threads, `lds`
bytes of LDS, spinning for the time the bucket would be on an 8-rank ring at `--busbw` GB/s while walking the bucket's bytes.

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
                        os.path.join(ROOT, "tools", "cu_hog.hip")], check=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=32)
    ap.add_argument("--threads", type=int, default=256)
    ap.add_argument("--lds", type=int, default=32768)
    ap.add_argument("--busbw", type=float, default=250.0, help="GB/s bus bandwidth of the emulated 8-rank all-reduce")
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--touch", type=int, default=1, help="1: the hog walks the bucket's bytes; 0: spins only")
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

    class _Done:
        def wait(self):
            return True

    def fake(tensor, op=None, group=None, async_op=False):
        if not async_op or not tensor.is_cuda or tensor.numel() < 1024:
            return real(tensor, **({} if op is None else {"op": op}), group=group, async_op=async_op)
        usec = tensor.numel() * 4 * 2.0 * (args.ranks - 1) / args.ranks / (args.busbw * 1e9) * 1e6
        launched["n"] += 1
        launched["usec"] += usec
        if args.blocks > 0:
            rc = hog.cu_hog(tensor.data_ptr(), tensor.numel() if args.touch else 0, args.blocks, args.threads, args.lds, usec,
                            torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
        return _Done()

    dist.all_reduce = fake
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
