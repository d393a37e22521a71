#!/usr/bin/env python
"""Static instruction budget of k_shade's ingredients (tools/shade_budget.hip): VALU / SALU / memory instructions of every probe
kernel in the gfx950 ISA minus the empty probe, and of the product's k_shade_wn<false, 0, 4> as a whole.  CPU only (hipcc -S).
    python tools/shade_budget.py [--json out.json]"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aten_amd import build  # noqa: E402


def counts(asm, name):
    on, v, s, m, tr = False, 0, 0, 0, 0
    for l in asm:
        if l.startswith(name + ":"):
            on = True
            continue
        if on:
            t = l.strip()
            if not t or t[0] in ";.":
                continue
            op = t.split()[0]
            if op.startswith("v_"):
                v += 1
                if re.match(r"v_(rcp|rsq|sqrt|sin|cos|exp|log)_", op):
                    tr += 1
            elif op.startswith("s_"):
                s += 1
            elif op.split("_")[0] in ("global", "flat", "buffer", "ds", "scratch"):
                m += 1
            if op == "s_endpgm":
                break
    return v, s, m, tr


def main():
    flags = [f for f in build.HIP_FLAGS if f not in ("-shared", "-fPIC")] + ["-fno-slp-vectorize", '-DATN_BUILD_ID="x"', "-I", os.path.join(ROOT, "include")]
    out_s = "/tmp/shade_budget.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["--cuda-device-only", "-S", "-o", out_s, os.path.join(ROOT, "tools", "shade_budget.hip")])
    asm = open(out_s).read().split("\n")
    names = [m.group(1) for l in asm for m in [re.match(r"^(probe_\w+):", l)] if m]
    base = counts(asm, "probe_empty")
    rows = {}
    print("%-26s %6s %6s %5s %6s" % ("ingredient", "VALU", "SALU", "mem", "trans"))
    for n in names:
        c = counts(asm, n)
        d = tuple(max(0, x - y) for x, y in zip(c, base))
        rows[n[6:]] = {"valu": d[0], "salu": d[1], "mem": d[2], "trans": d[3]}
        print("%-26s %6d %6d %5d %6d" % (n[6:], *d))
    if "--json" in sys.argv:
        json.dump(rows, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
