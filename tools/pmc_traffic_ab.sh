#!/bin/bash
# HBM traffic of the Winograd launch under both tile orders (PSLD_WINO_NMAJOR=1 default | 0): two counter-only passes each
set -u
export PSLD_HIP_LIB=${GRAFT_REPO_ROOT:-/root/repo}/tools/abl/libpsld_hip_abl.so   # PSLD_WINO_NMAJOR exists only in the ablation library (make -C tools/abl)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for nm in 1 0; do
  export PSLD_WINO_NMAJOR=$nm
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch$nm -- python3 $ROOT/tools/bench_wino.py --rounds 1 --iters 3 --shapes "256,256,32;512,256,32" > /dev/null 2> $OUT/pmc_fetch.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write$nm -- python3 $ROOT/tools/bench_wino.py --rounds 1 --iters 3 --shapes "256,256,32;512,256,32" > /dev/null 2> $OUT/pmc_write.err
  python3 - <<PY
import csv, glob
for tag, d in (("FETCH_SIZE", "$OUT/pmc_fetch$nm"), ("WRITE_SIZE", "$OUT/pmc_write$nm")):
    rows = [r for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f))]
    w = [float(r["Counter_Value"]) for r in rows if "wino_conv8s" in r["Kernel_Name"] and r["Counter_Name"] == tag]
    half = len(w) // 2
    for name, vals in (("256->256@32", w[:half]), ("512->256@32", w[half:])):
        mb = sum(vals) / len(vals) * 1024 * (2 if tag == "FETCH_SIZE" else 1) / 1e6
        print(f"NMAJOR=$nm {name} {tag}: {mb:.1f} MB per launch ({len(vals)} launches)")
PY
  rm -rf $OUT/pmc_fetch$nm $OUT/pmc_write$nm
done
unset PSLD_WINO_NMAJOR
for nm in 1 0; do echo "NMAJOR=$nm"; PSLD_WINO_NMAJOR=$nm python3 $ROOT/tools/bench_wino.py --rounds 5 --shapes "256,256,32;512,256,32;256,256,16;512,256,16" | grep "conv fwd" | cut -c1-140; done
