for w in 3 2; do echo "PSLD_WINO_W4=$w"; PSLD_WINO_W4=$w python tools/bench_wino.py --rounds 5 --shapes "256,256,32;512,256,32;256,256,16;512,256,16" 2>&1 | grep "conv fwd" | cut -c1-150; done
