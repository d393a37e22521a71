#!/usr/bin/env python
"""Per-kernel averages of rocprofv3 --pmc passes (csv output).
    python tools/pmc_summary.py gpurun_out/pmc_r01b > profiles/r01_b_sponza1080p_pmc.txt
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(n):
    n = n.replace("void ", "").replace("atn::", "")
    return n.split("(")[0]


def main(d):
    acc = defaultdict(lambda: defaultdict(list))
    for f in sorted(glob.glob(os.path.join(d, "pass*", "*counter_collection.csv"))):
        for row in csv.DictReader(open(f)):
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("# per-launch averages over all dispatches of each kernel; source: %s" % d)
    for k in sorted(acc):
        if k.startswith("__amd"):
            continue
        print("\n[%s]" % k)
        for c in sorted(acc[k]):
            v = acc[k][c]
            print("  %-36s n=%4d  avg=%16.1f  sum=%18.1f" % (c, len(v), sum(v) / len(v), sum(v)))


if __name__ == "__main__":
    main(sys.argv[1])
