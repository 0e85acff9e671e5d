"""Per-step summary of a rocprofv3 kernel_stats.csv: python tools/kstats.py file.csv steps [pattern ...]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
pats = sys.argv[3:]
tot = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e6
calls = sum(int(r["Calls"]) for r in rows) / steps
mfma = ("wino_conv", "dwgrad", "wwgrad_ws", "pw8_kernel", "pwgrad", "dconv", "bgemm", "attn_fwd", "tile_kernel")
m = sum(float(r["TotalDurationNs"]) for r in rows if any(k in r["Name"] for k in mfma)) / steps / 1e6
print(f"{tot:.2f} ms/step of kernels, {calls:.0f} launches/step; MFMA kernels {m:.2f} ms, the rest {tot - m:.2f} ms")
for p in pats:
    t = sum(float(r["TotalDurationNs"]) for r in rows if p in r["Name"]) / steps / 1e6
    c = sum(int(r["Calls"]) for r in rows if p in r["Name"]) / steps
    print(f"  {p:28s} {c:8.1f} launches {t:8.3f} ms")
if not pats:
    for r in rows[:50]:
        n = r["Name"].replace("(anonymous namespace)::", "")[:72]
        print(f"  {n:72s} {int(r['Calls']) / steps:8.1f} {float(r['TotalDurationNs']) / steps / 1e6:8.3f} ms {float(r['AverageNs']) / 1e3:8.1f} us")
