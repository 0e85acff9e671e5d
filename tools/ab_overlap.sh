# A/B of the weight-gradient side stream (PSLD_OVERLAP_WGRAD) and its fork granularity (PSLD_SIDE_GROUP)
run() { python3 bench.py "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value'],1), round(d['ms_per_step'],2), d.get('captured_step'))"; }
B16="--batch 16 --steps 30 --warmup 5 --sample-batch 0 --no-cpu-baseline --no-probe"
echo "B16 eager overlap=0"; PSLD_OVERLAP_WGRAD=0 run $B16
for g in 1 4 8 16 64; do echo "B16 eager group=$g"; PSLD_SIDE_GROUP=$g run $B16; done
echo "B16 graphs overlap=0"; PSLD_OVERLAP_WGRAD=0 run $B16 --graphs
for g in 1 8 16 64 1000; do echo "B16 graphs group=$g"; PSLD_SIDE_GROUP=$g run $B16 --graphs; done
for o in 0 1; do echo "B128 overlap=$o"; PSLD_OVERLAP_WGRAD=$o run --steps 10 --warmup 3 --sample-batch 0 --no-cpu-baseline --no-probe; done
