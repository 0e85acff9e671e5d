cd ${GRAFT_REPO_ROOT:-/root/repo}
PSLD_WINO_PERSIST=0 python3 tools/wino_cmp.py save /tmp/p0.pt | tail -3
echo "== persist 1 vs saved"; PSLD_WINO_PERSIST=1 python3 tools/wino_cmp.py cmp /tmp/p0.pt | grep -E "EQUAL|DIFFER"
echo "== persist 1 LA 3 vs saved"; PSLD_WINO_LA=3 PSLD_WINO_PERSIST=1 python3 tools/wino_cmp.py cmp /tmp/p0.pt | grep -E "EQUAL|DIFFER"
PSLD_WINO_PERSIST=0 python3 tools/wino_digest.py > /tmp/d0.txt; PSLD_WINO_PERSIST=1 python3 tools/wino_digest.py > /tmp/d1.txt; diff /tmp/d0.txt /tmp/d1.txt && echo DIGESTS_SAME
S="256,256,32;512,256,32;256,256,16;512,256,16"
run() { echo "== $*"; env "$@" python3 tools/bench_wino.py --rounds 5 --shapes "$S" 2>&1 | grep "conv fwd" | sed -e 's/direct fp32-in.*limb-in *[0-9.]* TF//' | cut -c1-110; }
for r in 1 2; do
run PSLD_WINO_PERSIST=0
run PSLD_WINO_PERSIST=1
run PSLD_WINO_PERSIST=1 PSLD_WINO_LA=3
done
run PSLD_WINO_PERSIST=1 PSLD_WINO_ABL=16
run PSLD_WINO_PERSIST=1 PSLD_WINO_ABL=32
run PSLD_WINO_PERSIST=1 PSLD_WINO_ABL=48
run PSLD_WINO_PERSIST=1 PSLD_WINO_ABL=1
run PSLD_WINO_PERSIST=1 PSLD_WINO_ABL=2
run PSLD_WINO_PERSIST=0 PSLD_WINO_ABL=1
run PSLD_WINO_PERSIST=0 PSLD_WINO_ABL=2
run PSLD_WINO_PERSIST=0 PSLD_WINO_ABL=3
