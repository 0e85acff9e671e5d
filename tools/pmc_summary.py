"""Per-kernel averages of a rocprofv3 --pmc pass (csv), as profiles/r01/pmc_tile_kernels.md reads them: effective clock =
GRBM_GUI_ACTIVE / 8 XCDs / duration; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (duration x clock x 1024 SIMDs); VALU
instructions; LDS conflict cycles / LDS active cycles.   python tools/pmc_summary.py <dir>"""
import re
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                m = re.search(r"(\w+(?:<[^>]*>)?)\(", row["Kernel_Name"].replace("(anonymous namespace)::", ""))
                name = m.group(1) if m else row["Kernel_Name"][:60]
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
                if row.get("Start_Timestamp") and row.get("End_Timestamp"):
                    dur[(name, row.get("Dispatch_Id"))] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    dsum = defaultdict(list)
    for (name, _), v in dur.items():
        dsum[name].append(v)
    print("| kernel | launches | avg us | eff. clock GHz | MFMA busy | VALU insts/launch | LDS conflict share |")
    print("|---|---|---|---|---|---|---|")
    for name, c in sorted(acc.items(), key=lambda kv: -sum(dsum.get(kv[0], [0]))):
        n = len(next(iter(c.values())))
        avg = lambda k: (sum(c[k]) / len(c[k])) if k in c and c[k] else float("nan")
        us = (sum(dsum[name]) / len(dsum[name]) / 1e3) if dsum.get(name) else float("nan")
        if not (us > 20):
            continue
        clock = avg("GRBM_GUI_ACTIVE") / 8 / (us * 1e3) if us == us else float("nan")
        busy = avg("SQ_VALU_MFMA_BUSY_CYCLES") / (us * 1e3 * clock * 1024) if clock == clock and clock > 0 else float("nan")
        ldsc = avg("SQ_LDS_BANK_CONFLICT") / avg("SQ_LDS_IDX_ACTIVE") if avg("SQ_LDS_IDX_ACTIVE") else float("nan")
        print(f"| {name} | {n} | {us:.1f} | {clock:.2f} | {busy:.3f} | {avg('SQ_INSTS_VALU'):.3g} | {ldsc:.3f} |")
    print("\nraw per-kernel averages (all counters):")
    for name, c in acc.items():
        if dsum.get(name) and sum(dsum[name]) / len(dsum[name]) > 20e3:
            print(name, {k: round(sum(v) / len(v), 1) for k, v in c.items()})


if __name__ == "__main__":
    main()
