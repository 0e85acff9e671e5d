#!/bin/bash
# which kernels surround the __amd_rocclr_copyBuffer launches of a training step?  (kernel trace of tools/count_torch_ops.py)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_cb -o tr -- python3 $ROOT/tools/count_torch_ops.py > /dev/null 2> $OUT/trace_cb.err
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/trace_cb/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:50] for r in rows]
prev, nxt = collections.Counter(), collections.Counter()
n = 0
for i, nm in enumerate(names):
    if "copyBuffer" in nm:
        n += 1
        prev[names[i - 1] if i else ""] += 1
        nxt[names[i + 1] if i + 1 < len(names) else ""] += 1
print("copyBuffer launches:", n, "of", len(names))
print("preceded by:", prev.most_common(8))
print("followed by:", nxt.most_common(8))
PY
rm -rf $OUT/trace_cb
