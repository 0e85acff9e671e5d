run() { python3 bench.py "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value'],1), round(d['ms_per_step'],2), d.get('captured_step'))"; }
B16="--batch 16 --steps 40 --warmup 5 --sample-batch 0 --no-cpu-baseline --no-probe"
for g in 8 64; do echo "B16 eager group=$g"; PSLD_SIDE_GROUP=$g run $B16; done
echo "B16 eager overlap=0"; PSLD_OVERLAP_WGRAD=0 run $B16
echo "B16 graphs"; run $B16 --graphs
echo "B16 graphs overlap=0"; PSLD_OVERLAP_WGRAD=0 run $B16 --graphs
echo "B32 eager"; run --batch 32 --steps 30 --warmup 5 --sample-batch 0 --no-cpu-baseline --no-probe
echo "B32 eager overlap=0"; PSLD_OVERLAP_WGRAD=0 run --batch 32 --steps 30 --warmup 5 --sample-batch 0 --no-cpu-baseline --no-probe
echo "B64 eager overlap=1"; PSLD_OVERLAP_WGRAD=1 run --batch 64 --steps 20 --warmup 5 --sample-batch 0 --no-cpu-baseline --no-probe
echo "B64 eager overlap=0"; PSLD_OVERLAP_WGRAD=0 run --batch 64 --steps 20 --warmup 5 --sample-batch 0 --no-cpu-baseline --no-probe
echo "B128"; run --steps 10 --warmup 3 --sample-batch 0 --no-cpu-baseline --no-probe
