#!/usr/bin/env python3
"""Compare two rocprofv3 *_kernel_stats.csv files kernel by kernel (average ns, calls):  python tools/ks_compare.py A.csv B.csv"""
import csv
import re
import sys


def load(path):
    out = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            name = re.sub(r"\(.*", "", row["Name"].replace("lgn::", "").replace("void ", ""))
            out[name] = (int(row["Calls"]), float(row["AverageNs"]), float(row["TotalDurationNs"]))
    return out


def main():
    a, b = load(sys.argv[1]), load(sys.argv[2])
    print(f"{'kernel':58s} {'calls':>5s} {'A us':>8s} {'B us':>8s} {'B-A total us/step':>18s}")
    steps = float(sys.argv[3]) if len(sys.argv) > 3 else 56.0
    tot = 0.0
    for k in sorted(set(a) | set(b), key=lambda k: -(b.get(k, a.get(k))[2])):
        ca, ua, ta = a.get(k, (0, 0.0, 0.0))
        cb, ub, tb = b.get(k, (0, 0.0, 0.0))
        d = (tb - ta) / steps / 1e3
        tot += d
        if max(ta, tb) / steps > 500:
            print(f"{k[:58]:58s} {cb:5d} {ua / 1e3:8.2f} {ub / 1e3:8.2f} {d:18.2f}")
    print(f"sum of differences: {tot:.1f} us per step")


if __name__ == "__main__":
    main()
