#!/usr/bin/env python3
"""Print the top kernels of a rocprofv3 --kernel-trace --stats run (CSV output).  usage: kstats.py DIR [top] [steps]"""
import csv
import glob
import sys

d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 0
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
if not f:
    sys.exit(f"no *kernel_stats.csv under {d}")
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
print(f"{f[0]}: total {tot / 1e6:.3f} ms in {calls} launches" + (f"; per step {tot / 1e3 / steps:.1f} us, {calls / steps:.1f} launches" if steps else ""))
for r in rows[:top]:
    per = f"{float(r['TotalDurationNs']) / 1e3 / steps:9.1f}" if steps else ""
    print(f"{r['Name'][:90]:90s} {int(r['Calls']):6d} {float(r['TotalDurationNs']) / 1e3:10.1f} us  avg {float(r['AverageNs']) / 1e3:8.2f} us {per}")
