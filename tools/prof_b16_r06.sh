#!/bin/bash
# rocprofv3 per-kernel totals of the B=16 training step (run through gpurun): bash tools/prof_b16_r06.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b16 -o b16 -- python3 $R/bench.py --batch 16 --steps 20 --warmup 5 --sample-batch 0 --no-cpu-baseline --no-forward --no-config-block --no-probe > $R/gpurun_out/s2_b16_prof.json 2> $R/gpurun_out/s2_b16_prof.err
find $R/gpurun_out/prof_b16 -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/s2_b16_kernel_stats.csv \;
rm -rf $R/gpurun_out/prof_b16
cd $R && python3 tools/kstats.py gpurun_out/s2_b16_kernel_stats.csv 25 | head -45
