"""Per-dispatch wait-state view of tools/pmc_wait.sh's two passes.  MFMA busy is quoted twice: against the dispatch
duration (GRBM_GUI_ACTIVE, which under counter collection includes ~100 us of start/stop overhead per dispatch) and
against the wave residency (SQ_WAVE_CYCLES, quad-cycles, 2 waves per SIMD): the second is the fraction of the time
waves are on the SIMD that its matrix pipe is busy.   python tools/pmc_wait_summary.py gpurun_out/r02 [min_us] [name]"""
import csv
import glob
import sys
from collections import defaultdict


def load(d):
    f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[-1]
    disp = defaultdict(dict)
    for r in csv.DictReader(open(f)):
        k = int(r["Dispatch_Id"])
        disp[k]["name"] = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:44]
        disp[k][r["Counter_Name"]] = float(r["Counter_Value"])
        disp[k]["dur"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        disp[k]["vgpr"] = int(r.get("VGPR_Count", 0) or 0)
    return disp


def main():
    root = sys.argv[1]
    min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
    pat = sys.argv[3] if len(sys.argv) > 3 else ""
    a, b = load(root + "/pmc_wait"), load(root + "/pmc_wait2")
    print("| kernel | us (profiled) | clock GHz | MFMA busy / dispatch | MFMA busy / residency | parked | issue stall | issuing | LDS | VALU | VALU insts M | LDS insts M |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for k in sorted(a):
        d = a[k]
        if d["dur"] < min_us * 1e3 or pat not in d["name"]:
            continue
        wc = d["SQ_WAVE_CYCLES"]
        e = b.get(k, {})
        gui = d["GRBM_GUI_ACTIVE"] / 8
        print(f"| {d['name']} | {d['dur'] / 1e3:.0f} | {gui / d['dur']:.2f} | {d['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui * 1024):.3f} | "
              f"{d['SQ_VALU_MFMA_BUSY_CYCLES'] / (wc * 4 / 2):.3f} | {d['SQ_WAIT_ANY'] / wc:.3f} | {d['SQ_WAIT_INST_ANY'] / wc:.3f} | "
              f"{d['SQ_ACTIVE_INST_ANY'] / wc:.3f} | {e.get('SQ_ACTIVE_INST_LDS', 0) / wc:.3f} | {e.get('SQ_ACTIVE_INST_VALU', 0) / wc:.3f} | "
              f"{e.get('SQ_INSTS_VALU', 0) / 1e6:.1f} | {e.get('SQ_INSTS_LDS', 0) / 1e6:.1f} |")


if __name__ == "__main__":
    main()
