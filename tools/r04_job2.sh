cd ${GRAFT_REPO_ROOT:-/root/repo}
PSLD_WINO_PERSIST=0 python3 tools/wino_cmp.py save /tmp/p0.pt
echo "== persist 0 again vs saved"; PSLD_WINO_PERSIST=0 python3 tools/wino_cmp.py cmp /tmp/p0.pt
echo "== persist 1 vs saved"; PSLD_WINO_PERSIST=1 python3 tools/wino_cmp.py cmp /tmp/p0.pt
echo "== persist 64 vs saved"; PSLD_WINO_PERSIST=64 python3 tools/wino_cmp.py cmp /tmp/p0.pt
for r in 1 2; do for p in 0 1; do echo "== PSLD_WINO_PERSIST=$p"; PSLD_WINO_PERSIST=$p python3 tools/bench_wino.py --rounds 5 2>&1 | grep "conv fwd" | cut -c1-175; done; done
echo "== B=512"; for p in 0 1; do PSLD_WINO_PERSIST=$p python3 tools/bench_wino.py --rounds 3 --batch 512 2>&1 | grep "conv fwd" | cut -c1-175; done
