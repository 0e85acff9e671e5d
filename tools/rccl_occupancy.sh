#!/bin/bash
# Emulated cost of RCCL's channel workgroups on the B=128 step (tools/rccl_occupancy.py).  One GPU, same box, interleaved.
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r05
mkdir -p $OUT
export TMPDIR=/tmp
A="--steps 10 --warmup 3"
{
for r in 1 2; do
  echo "no process group (the N = 1 step)"; python3 bench.py $A --sample-batch 0 --no-cpu-baseline --no-forward --no-probe 2> /dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(json.dumps({'value': d['value'], 'ms_per_step': d['ms_per_step']}))"
  echo "reducer, 1-rank RCCL group, collectives replaced by nothing (blocks 0)"; python3 tools/rccl_occupancy.py --blocks 0 $A 2> /dev/null | tail -1
  echo "16 x 256 threads, 32 KB LDS, 250 GB/s"; python3 tools/rccl_occupancy.py --blocks 16 $A 2> /dev/null | tail -1
  echo "32 x 256 threads, 32 KB LDS, 250 GB/s"; python3 tools/rccl_occupancy.py --blocks 32 $A 2> /dev/null | tail -1
  echo "64 x 256 threads, 32 KB LDS, 250 GB/s"; python3 tools/rccl_occupancy.py --blocks 64 $A 2> /dev/null | tail -1
  echo "32 x 512 threads, 64 KB LDS, 150 GB/s"; python3 tools/rccl_occupancy.py --blocks 32 --threads 512 --lds 65536 --busbw 150 $A 2> /dev/null | tail -1
done
} | tee $OUT/rccl_occupancy.txt
