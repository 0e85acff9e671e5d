#!/bin/bash
# NOTE (round 5): kept as the record of how profiles of that round were produced; switches it names that lost their A/B
# (PSLD_FUSED_ATTN=3, PSLD_DWGRAD_WS outside the ablation library, --tape, ...) were removed in round 5 - see git history.
# Round-4 same-box A/Bs (run on the GPU box through gpurun; results in profiles/r04/).  The kernel variants and timing-only
# ablations live in libpsld_hip_abl.so (make -C psld_amd/csrc abl); bench.py refuses that library by name, so the two
# step-level A/Bs of product-equivalent kernels load a copy under a neutral name.
#   bash tools/ab_r04.sh wino | dwgrad | fusedgn | stamps | l2 | gnb | gnprev | colsum | attn
cd ${GRAFT_REPO_ROOT:-/root/repo}
ABL=$PWD/psld_amd/libpsld_hip_abl.so
S="256,256,32;512,256,32;256,256,16;512,256,16"
wino() { echo "== $*"; env PSLD_HIP_LIB=$ABL "$@" python3 tools/bench_wino.py --rounds 5 --shapes "$S" 2>&1 | grep "conv fwd" | sed -e 's/direct fp32-in.*limb-in *[0-9.]* TF//' | cut -c1-110; }
step() { env "$@" python3 bench.py --steps 10 --warmup 3 --sample-batch 0 --no-cpu-baseline --no-probe --no-forward 2> /dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('  %.1f img/s %.2f ms/step' % (d['value'], d['ms_per_step']))"; }
case "${1:-wino}" in
wino)       # product kernel | round-4 body with CU-resident workgroups (+ its ablations) | positions split over the waves | variants
  PSLD_HIP_LIB=$ABL python3 tools/wino_digest.py > /tmp/d0.txt
  for v in PSLD_WINO_PERSIST=1 PSLD_WINO_Q=1 PSLD_WINO_ERAW=1 PSLD_WINO_LA=3; do env PSLD_HIP_LIB=$ABL $v python3 tools/wino_digest.py > /tmp/d1.txt; diff -q /tmp/d0.txt /tmp/d1.txt > /dev/null && echo "$v: digests equal"; done
  for r in 1 2; do
    wino PSLD_X=0; wino PSLD_WINO_PERSIST=1; wino PSLD_WINO_PERSIST=1 PSLD_WINO_LA=3; wino PSLD_WINO_Q=1; wino PSLD_WINO_ERAW=1; wino PSLD_WINO_LA=3
  done
  for a in 16 32 48 1 2; do wino PSLD_WINO_PERSIST=1 PSLD_WINO_ABL=$a; done
  for a in 1 2 3 128 256 4; do wino PSLD_WINO_ABL=$a; done
  for a in 1 2 3; do wino PSLD_WINO_Q=1 PSLD_WINO_ABL=$a; done ;;
dwgrad)     # round 3's weight gradient vs the wave-specialised one: micro-benchmark and full step
  for r in 1 2; do for w in 0 1; do echo "== PSLD_DWGRAD_WS=$w"; PSLD_HIP_LIB=$ABL PSLD_DWGRAD_WS=$w python3 tools/bench_limb.py --wgrad --rounds 5 --iters 5 2>&1 | grep wgrad | cut -c1-140; done; done
  cp $ABL /tmp/libpsld_hip_prev.so
  for r in 1 2 3; do echo "round-3 dwgrad"; step PSLD_HIP_LIB=/tmp/libpsld_hip_prev.so PSLD_DWGRAD_WS=0; echo "wave-specialised dwgrad"; step PSLD_X=1; done ;;
fusedgn)    # GroupNorm apply + SiLU: separate pass + convolution vs fused into the Winograd staging; EM step
  python3 tools/bench_wino.py --fused-gn --rounds 5 --batch 512 --shapes "256,256,32;512,256,32;256,256,16" 2>&1 | grep "conv fwd"
  python3 tools/bench_wino.py --fused-gn --rounds 5 --batch 128 --shapes "256,256,32;256,256,16" 2>&1 | grep "conv fwd"
  for r in 1 2; do for m in 0 1 2; do echo "PSLD_FUSED_GN=$m"; ONLY512=1 PSLD_FUSED_GN=$m python3 tools/bench_sample.py 2>&1 | grep "graphs=0"; done; done ;;
stamps)     # s_memtime timeline of the product Winograd kernel and of its ablated builds
  for a in 64 65 66 67; do echo "=== PSLD_WINO_ABL=$a"; PSLD_HIP_LIB=$ABL PSLD_WINO_ABL=$a python3 tools/wino_stamps.py 256 256 32 128; done ;;
gnb)        # GroupNorm backward: time stamps of the one-slab kernel and its timing-only modes, the resident kernel, per-variant
            # launch times of both, and the full step with either (PSLD_GN_BWD_PIPE is a product switch: psld_set_gn_bwd_kernel)
  for m in 0 2 4 6 0x218; do PSLD_GN_BWD_PIPE=0 PSLD_HIP_LIB=$ABL python3 tools/gnb_stamps.py 128 32 256 $m 2>&1 | grep -E "^B=|median|last end"; done
  PSLD_HIP_LIB=$ABL python3 tools/gnb_stamps.py 128 32 256 0 2>&1 | grep -E "^B=|median"
  PSLD_HIP_LIB=$ABL python3 tools/gnb_stamps.py 128 16 256 0 2>&1 | grep -E "^B=|median"
  python3 tools/bench_gnb.py > /tmp/gnb_a.txt 2>&1; PSLD_GN_BWD_PIPE=0 python3 tools/bench_gnb.py > /tmp/gnb_b.txt 2>&1
  echo "(left: default; right: PSLD_GN_BWD_PIPE=0, the one-slab kernel for every shape)"; paste /tmp/gnb_a.txt /tmp/gnb_b.txt | cut -c1-82,140-170
  for r in 1 2 3; do echo "one-slab kernel everywhere"; step PSLD_GN_BWD_PIPE=0; echo "resident kernel where it applies"; step PSLD_X=1; done
  for r in 1 2; do echo "B=16 one-slab"; env PSLD_GN_BWD_PIPE=0 python3 bench.py --batch 16 --steps 40 --warmup 5 --sample-batch 0 --no-cpu-baseline --no-probe --no-forward 2>/dev/null | tail -1 | cut -c1-90; echo "B=16 default"; python3 bench.py --batch 16 --steps 40 --warmup 5 --sample-batch 0 --no-cpu-baseline --no-probe --no-forward 2>/dev/null | tail -1 | cut -c1-90; done ;;
gnprev)     # full step: this build against the same build with the GroupNorm kernels of commit d5c7866.  The library is built in
            # the development container (needs the git history), in psld_amd/csrc:
            #   git show d5c7866:psld_amd/csrc/norm_act.hip > prev.hip; append "extern \"C\" int psld_set_gn_bwd_kernel(int) { return 0; }
            #   extern \"C\" int psld_get_gn_bwd_kernel(void) { return 0; }"; hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -I../../include -I. -c prev.hip
            #   hipcc -shared -fPIC --offload-arch=gfx950 <the other objects> prev.o -o ../libpsld_hip_prevgn.so
  for r in 1 2 3 4; do echo "round-4 committed GroupNorm kernels"; step PSLD_HIP_LIB=$PWD/psld_amd/libpsld_hip_prevgn.so; echo "this build"; step PSLD_X=1; done ;;
colsum)     # Conv_0's bias / time-embedding gradient: column-sum pass over dh1 vs by-product of the GroupNorm backward
  for r in 1 2 3 4; do echo "column-sum pass over dh1 (PSLD_GN_BWD_COLSUM=0)"; step PSLD_GN_BWD_COLSUM=0; echo "by-product of the GroupNorm backward (default)"; step PSLD_X=1; done ;;
attn)       # fused attention forward: kernel level, then step and EM step with the 16x16 maps on the three-kernel path (=3) or fused
  python3 tools/bench_attn.py; python3 tools/bench_attn.py --batch 512
  PSLD_HIP_LIB=$ABL PSLD_ATTN_QW4=1 python3 tools/bench_attn.py | grep "HW=256"
  for r in 1 2 3; do echo "fused on 8x8 only (PSLD_FUSED_ATTN=3)"; step PSLD_FUSED_ATTN=3; echo "fused on 16x16 too (default)"; step PSLD_X=1; done
  for r in 1 2; do for m in 3 1; do echo "EM step, PSLD_FUSED_ATTN=$m"; ONLY512=1 PSLD_FUSED_ATTN=$m python3 tools/bench_sample.py 2>&1 | grep "graphs=0"; done; done ;;
l2)         # fragment-stream micro-benchmark (hipcc -O3 --offload-arch=gfx950 tools/l2_stream.hip -o tools/l2_stream)
  tools/l2_stream ;;
esac
