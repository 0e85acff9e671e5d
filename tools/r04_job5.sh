cd ${GRAFT_REPO_ROOT:-/root/repo}
python3 -m pytest tests/test_kernels_gpu.py -q -x -k "wino" 2>&1 | tail -5
python3 tools/bench_wino.py --fused-gn --rounds 5 --batch 512 --shapes "256,256,32;512,256,32;256,256,16" 2>&1 | grep "conv fwd"
python3 tools/bench_wino.py --fused-gn --rounds 5 --batch 128 --shapes "256,256,32;256,256,16" 2>&1 | grep "conv fwd"
