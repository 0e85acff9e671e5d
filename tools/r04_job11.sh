cd ${GRAFT_REPO_ROOT:-/root/repo}
export PSLD_HIP_LIB=$PWD/psld_amd/libpsld_hip_abl.so
python3 tools/wino_digest.py > /tmp/d0.txt; PSLD_WINO_Q=1 python3 tools/wino_digest.py > /tmp/d1.txt; diff /tmp/d0.txt /tmp/d1.txt > /tmp/dd.txt && echo DIGESTS_SAME || head -5 /tmp/dd.txt
python3 tools/wino_digest.py --small > /tmp/s0.txt; PSLD_WINO_Q=1 python3 tools/wino_digest.py --small > /tmp/s1.txt; diff /tmp/s0.txt /tmp/s1.txt > /tmp/sd.txt && echo SMALL_DIGESTS_SAME || head -5 /tmp/sd.txt
PSLD_WINO_Q=1 python3 tools/bench_wino.py --check --rounds 1 --iters 1 --shapes "256,256,8" 2>&1 | tail -3
S="256,256,32;512,256,32;256,256,16;512,256,16;256,256,8"
run() { echo "== $*"; env "$@" python3 tools/bench_wino.py --rounds 5 --shapes "$S" $EXTRA 2>&1 | grep "conv fwd" | sed -e 's/direct fp32-in.*limb-in *[0-9.]* TF//' | cut -c1-110; }
for r in 1 2; do
run PSLD_WINO_Q=0
run PSLD_WINO_Q=1
done
run PSLD_WINO_Q=1 PSLD_WINO_ABL=1
run PSLD_WINO_Q=1 PSLD_WINO_ABL=2
run PSLD_WINO_Q=1 PSLD_WINO_ABL=3
EXTRA="--batch 512"
run PSLD_WINO_Q=0
run PSLD_WINO_Q=1
