cd ${GRAFT_REPO_ROOT:-/root/repo}
python3 -m pytest tests/test_kernels_gpu.py -q -x -k "wino" 2>&1 | tail -2
python3 -m pytest tests/test_model_gpu.py -q -x -k "fused_into or hip_graph_forward or with_winograd" 2>&1 | tail -3
for r in 1 2; do for m in 0 1 2; do echo "PSLD_FUSED_GN=$m"; ONLY512=1 PSLD_FUSED_GN=$m python3 tools/bench_sample.py 2>&1 | grep "graphs=0"; done; done
