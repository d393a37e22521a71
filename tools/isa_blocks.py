#!/usr/bin/env python
"""Per-basic-block VALU/SALU/MEM instruction counts of one kernel in a gfx950 .s file.
usage: tools/isa_blocks.py <file.s> <mangled-kernel-name-prefix>"""
import re
import sys


def main(path, prefix):
    txt = open(path).read().split("\n")
    out, on = [], False
    for l in txt:
        if re.match(r"^%s.*:" % re.escape(prefix), l):
            on = True
        if on:
            out.append(l)
            if "s_endpgm" in l:
                break
    cur = ["entry", 0, 0, 0, []]
    blocks = []
    for l in out:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur)
            cur = [m.group(1), 0, 0, 0, []]
            continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        op = t.split()[0]
        if op.startswith("v_"):
            cur[1] += 1
        elif op.startswith("s_"):
            cur[2] += 1
        elif op.split("_")[0] in ("global", "flat", "buffer", "ds", "scratch"):
            cur[3] += 1
            cur[4].append(op)
        if op.startswith("s_cbranch") or op.startswith("s_branch"):
            cur[4].append(t.replace("s_cbranch_", "cb_").replace("s_branch", "br"))
    blocks.append(cur)
    for b in blocks:
        print("%-12s VALU %3d SALU %3d MEM %2d  %s" % (b[0], b[1], b[2], b[3], " ".join(b[4])))
    print("total VALU %d SALU %d MEM %d" % (sum(b[1] for b in blocks), sum(b[2] for b in blocks), sum(b[3] for b in blocks)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
