#!/bin/bash
# Copy the judged artefacts of a profile / bench run from gpurun_out/r02 (scratch) into profiles/r02 (tracked)
cd "$(dirname "$0")/.."
O=gpurun_out/r02; P=profiles/r02
for f in bench_default_run bench_b16_eager bench_b16_eager_no_side_stream bench_b16_graph bench_b32_eager bench_celeba64 bench_rccl_1rank rehearsal_gloo_2rank bench_train_b128_profiled_run; do
  [ -f $O/$f.json ] && grep '^{' $O/$f.json | tail -1 > $P/$f.json
done
for f in bench_train_b128_kernel_stats.csv tile_kernels.txt limb_planes_ab.txt wgrad_xlimb_ab.txt limb_planes_b16.txt hbm_kernels.txt sampling.txt host_vs_gpu.txt graph_fork_cost.txt; do
  [ -f $O/$f ] && cp $O/$f $P/
done
sed -i '/amdgpu.ids/d' $P/*.txt
