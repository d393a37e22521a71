#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite output) into a text table.

    python tools/rocprof_summary.py gpurun_out/prof_r01/sponza_results.db > profiles/r01_sponza_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = list(cur.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
        "max(vgpr_count), max(sgpr_count), max(scratch_size), max(lds_size), max(grid_x), max(workgroup_x) "
        "from kernels group by name order by 3 desc"))
    total = sum(r[2] for r in rows) or 1
    print("# rocprofv3 --kernel-trace --stats summary of %s" % path)
    print("%-72s %7s %12s %11s %11s %11s %6s %5s %5s %8s %9s %5s" % (
        "kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct", "vgpr", "sgpr", "scratch", "grid", "wg"))
    for r in rows:
        print("%-72s %7d %12.3f %11.2f %11.2f %11.2f %6.2f %5d %5d %8d %9d %5d" % (
            r[0][:72], r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total,
            r[6] or 0, r[7] or 0, r[8] or 0, r[10] or 0, r[11] or 0))


if __name__ == "__main__":
    main(sys.argv[1])
