for a in 0 4 8; do echo "W8S ABL=$a"; PSLD_WINO_ABL=$a python tools/bench_wino.py --rounds 5 --shapes "256,256,32;512,256,16" 2>&1 | grep "conv fwd" | cut -c1-150; done
