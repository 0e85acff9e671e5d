#!/bin/bash
# kernel TRACE (per-launch durations and grids) of a few B=128 training steps: bash tools/trace_step_r05.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-trace}
mkdir -p $ROOT/gpurun_out/r05
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/r05/trace_$TAG -o step -- python3 $ROOT/bench.py --steps 2 --warmup 2 --sample-batch 0 --no-cpu-baseline --no-forward --no-probe > /dev/null 2>&1
python3 - <<PY
import csv, glob, re, collections
f = glob.glob("$ROOT/gpurun_out/r05/trace_$TAG/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
# last step only: take the last quarter of the launches
rows = rows[len(rows) * 3 // 4:]
agg = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    m = re.search(r"(dwgrad_ws_kernel|dwgrad_kernel<[^>]*>|wino_conv8s_kernel<[^>]*>|pwgrad_kernel|pw8_kernel<0>|dconv_\w+<[^>]*>)", n)
    if m:
        key = (m.group(1), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]))
        agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items()):
    print(f"{k[0]:40s} grid {k[1]:6d} x {k[2]:3d}  n={len(v):3d}  avg {sum(v)/len(v):8.1f} us  total {sum(v)/1e3:7.2f} ms  [{min(v):.0f} .. {max(v):.0f}]")
PY
rm -rf $ROOT/gpurun_out/r05/trace_$TAG
