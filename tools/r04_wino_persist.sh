#!/bin/bash
# r04: CU-resident Winograd kernel vs one workgroup per item: bitwise digests, fp64 check, micro-benchmark
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04_persist; mkdir -p $O
PSLD_WINO_PERSIST=0 timeout 600 python3 tools/wino_digest.py > $O/digest_p0.txt 2>&1
PSLD_WINO_PERSIST=1 timeout 600 python3 tools/wino_digest.py > $O/digest_p1.txt 2>&1
PSLD_WINO_PERSIST=0 timeout 600 python3 tools/wino_digest.py --small > $O/digest_s0.txt 2>&1
PSLD_WINO_PERSIST=48 timeout 600 python3 tools/wino_digest.py --small > $O/digest_s48.txt 2>&1
echo "== digest diff (full)"; diff $O/digest_p0.txt $O/digest_p1.txt && echo SAME
echo "== digest diff (small, 48 workgroups)"; diff $O/digest_s0.txt $O/digest_s48.txt && echo SAME
tail -3 $O/digest_p1.txt
PSLD_WINO_PERSIST=1 timeout 900 python3 tools/bench_wino.py --check --rounds 1 --iters 2 --shapes "256,256,32" 2>&1 | tail -4
for r in 1 2; do
  for p in 0 1; do
    echo "== PSLD_WINO_PERSIST=$p"
    PSLD_WINO_PERSIST=$p timeout 900 python3 tools/bench_wino.py --rounds 5 2>&1 | grep "conv fwd" | sed -e 's/direct fp32-in.*winograd/winograd/' -e 's/rel-L2.*//'
  done
done | tee $O/bench_ab.txt
echo "== B=512"
for p in 0 1; do PSLD_WINO_PERSIST=$p timeout 900 python3 tools/bench_wino.py --rounds 3 --batch 512 2>&1 | grep "conv fwd" | sed -e 's/direct fp32-in.*winograd/winograd/' -e 's/rel-L2.*//'; done | tee $O/bench_ab_b512.txt
