run() { python3 bench.py "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value'],1), round(d['ms_per_step'],2))"; }
B16="--batch 16 --steps 40 --warmup 5 --sample-batch 0 --no-cpu-baseline --no-probe"
for m in 1 0 1 0; do echo "B16 graphs MT64=$m"; PSLD_DCONV_MT64=$m run $B16 --graphs; done
for m in 1 0 1 0; do echo "B16 eager no-overlap MT64=$m"; PSLD_OVERLAP_WGRAD=0 PSLD_DCONV_MT64=$m run $B16; done
for m in 1 0; do echo "B32 eager MT64=$m"; PSLD_DCONV_MT64=$m run --batch 32 --steps 30 --warmup 5 --sample-batch 0 --no-cpu-baseline --no-probe; done
