# A/B of the single-pass GroupNorm backward (PSLD_GN_BWD_FUSED)
run() { python3 bench.py "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value'],1), round(d['ms_per_step'],2))"; }
python3 tools/bench_hbm.py 2>&1 | grep -i "gn_bwd"
PSLD_GN_BWD_FUSED=0 python3 tools/bench_hbm.py 2>&1 | grep -i "gn_bwd"
for m in 1 0 1 0; do echo "B128 fused=$m"; PSLD_GN_BWD_FUSED=$m run --steps 10 --warmup 3 --sample-batch 0 --no-cpu-baseline --no-probe; done
for m in 1 0 1 0; do echo "B16 fused=$m"; PSLD_GN_BWD_FUSED=$m run --batch 16 --steps 40 --warmup 5 --sample-batch 0 --no-cpu-baseline --no-probe; done
