#!/bin/bash
# End-of-round refresh on the final build: full -m gpu suite, HBM-traffic counters (pmc_traffic.json carries the SHA-256 of the
# kernel sources and bench.py refuses a stale one), the default bench line, small-batch lines.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03
mkdir -p $OUT
cd $ROOT
python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1
grep -E "passed|failed" $OUT/gpu_tests.log | tail -1
bash tools/pmc_r03.sh
mkdir -p profiles/r03 && cp $OUT/pmc_traffic.json profiles/r03/pmc_traffic.json
cd $ROOT
python3 bench.py > $OUT/bench_default_run.json 2> $OUT/bench_default_run.err
python3 bench.py --batch 16 --steps 30 --warmup 5 --sample-batch 0 --no-cpu-baseline --no-probe > $OUT/bench_b16_eager.json 2>/dev/null
python3 bench.py --batch 16 --steps 30 --warmup 5 --sample-batch 0 --no-cpu-baseline --no-probe --tape > $OUT/bench_b16_tape.json 2>/dev/null
python3 bench.py --batch 64 --steps 20 --warmup 5 --sample-batch 0 --no-cpu-baseline --no-probe > $OUT/bench_b64_eager.json 2>/dev/null
tail -1 $OUT/bench_default_run.json | cut -c1-300
