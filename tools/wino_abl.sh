for n in 1 0; do echo "PSLD_WINO_NLOOP=$n"; PSLD_WINO_NLOOP=$n python tools/bench_wino.py --rounds 5 --shapes "256,256,32;512,256,32;256,256,16;512,256,16" 2>&1 | grep "conv fwd" | cut -c1-150; done
