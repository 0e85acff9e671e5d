# A/B of the 64-row tile variant of the 3x3 limb kernels (PSLD_DCONV_MT64)
run() { python3 bench.py "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value'],1), round(d['ms_per_step'],2))"; }
for m in 1 0; do echo "MT64=$m micro B=16"; PSLD_DCONV_MT64=$m python3 tools/bench_limb.py --batch 16 --rounds 3 2>&1 | grep conv; done
for m in 1 0; do echo "MT64=$m micro B=128"; PSLD_DCONV_MT64=$m python3 tools/bench_limb.py --rounds 2 2>&1 | grep "@8"; done
for m in 1 0 1 0; do echo "B16 MT64=$m"; PSLD_DCONV_MT64=$m run --batch 16 --steps 40 --warmup 5 --sample-batch 0 --no-cpu-baseline --no-probe; done
for m in 1 0 1 0; do echo "B128 MT64=$m"; PSLD_DCONV_MT64=$m run --steps 10 --warmup 3 --sample-batch 0 --no-cpu-baseline --no-probe; done
