for a in 0 1 2 3; do echo "W8S ABL=$a"; PSLD_WINO_ABL=$a python tools/bench_wino.py --rounds 3 --shapes "256,256,32;512,256,16" 2>&1 | grep "conv fwd"; done
