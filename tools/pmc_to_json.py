#!/usr/bin/env python
"""Per-kernel, per-launch averages of rocprofv3 --pmc passes (tools/pmc_collect.sh) as JSON, with the derived
fractions bench.py reports in its `roofline` object.

    python tools/pmc_to_json.py gpurun_out/pmc_r02b "sponza_lod 1920x1080 1spp 5-bounce" > profiles/r02_b_counters_sponza1080p.json

Formulas (MI355X: 256 CUs, 1024 SIMDs, 8 XCDs; MI355X_MICROARCH.md for the peaks and the FETCH_SIZE correction):
  cycles            = GRBM_GUI_ACTIVE / 8                      (the counter sums the 8 XCDs)
  valu_busy         = SQ_ACTIVE_INST_VALU / 256 / cycles        (rocprofiler-sdk's own VALUBusy: quad-cycles per CU)
  lane_utilisation  = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64)
  hbm_bytes         = FETCH_SIZE[KB] * 1024 * 2 + WRITE_SIZE[KB] * 1024       (gfx950: FETCH_SIZE tallies 128-B requests at 64 B)
  l2_bytes_max      = TCC_REQ * 128                             (upper bound: every L2 request a full 128-B line)
  l2_hit_rate       = TCC_HIT / (TCC_HIT + TCC_MISS)
  l1_hit_rate       = 1 - TCP_TCC_READ_REQ / TCP_TOTAL_ACCESSES
  l1_stall          = TCP_PENDING_STALL_CYCLES / 256 / cycles   (share of the launch a CU's L1 sits on pending misses)
  tcp_lane_accesses_per_cu_cycle  = TCP_TOTAL_ACCESSES / 256 / cycles        (one per lane and load; ceiling = 64 B/clk data path)
  tcp_cache_accesses_per_cu_cycle = TCP_TOTAL_CACHE_ACCESSES / 256 / cycles  (one per distinct 64-B chunk a quad of lanes touches: tag rate)
  tcp_active        = TCP_GATE_EN2 / 256 / cycles
The ceilings of the last three are MEASURED: profiles/r03_calibration.json (tools/valu_calib.hip).
`kernel_sources_sha16` = aten_amd.build.kernel_sources_sha16() of the tree the passes ran on, `build_id` = atn_build_id() of the
library they ran (ATEN_AMD_LIB included): sources hash + extra compile flags; bench.py refuses a mismatch of either.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def short(n):
    n = n.replace("void ", "").replace("atn::", "")
    return n.split("(")[0]


def main(d, workload):
    acc = defaultdict(lambda: defaultdict(list))
    for f in sorted(glob.glob(os.path.join(d, "pass*", "*counter_collection.csv"))):
        for row in csv.DictReader(open(f)):
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    from aten_amd.build import kernel_sources_sha16, loaded_build_id
    out = {"workload": workload, "kernel_sources_sha16": kernel_sources_sha16(), "build_id": loaded_build_id(), "source": "rocprofv3 --pmc, one pass per counter set (tools/pmc_collect.sh); per-launch averages over all dispatches of a kernel",
           "kernels": {}}
    for k in sorted(acc):
        if k.startswith("__amd") or not k.startswith("k_"):
            continue
        c = {n: sum(v) / len(v) for n, v in acc[k].items()}
        e = {"launches_sampled": max(len(v) for v in acc[k].values()), "counters": {n: round(x, 1) for n, x in sorted(c.items())}}
        g = lambda n: c.get(n)
        if g("GRBM_GUI_ACTIVE"):
            cyc = g("GRBM_GUI_ACTIVE") / 8.0
            e["cycles"] = round(cyc)
            if g("SQ_ACTIVE_INST_VALU") is not None:
                e["valu_busy"] = round(g("SQ_ACTIVE_INST_VALU") / 256.0 / cyc, 4)
            if g("SQ_INSTS_VALU") is not None:
                e["valu_insts_per_simd_cycle"] = round(g("SQ_INSTS_VALU") / 1024.0 / cyc, 4)
            if g("TCP_TOTAL_ACCESSES_sum") is not None:
                e["tcp_lane_accesses_per_cu_cycle"] = round(g("TCP_TOTAL_ACCESSES_sum") / 256.0 / cyc, 4)
            if g("TCP_TOTAL_CACHE_ACCESSES_sum") is not None:
                e["tcp_cache_accesses_per_cu_cycle"] = round(g("TCP_TOTAL_CACHE_ACCESSES_sum") / 256.0 / cyc, 4)
            if g("TCP_GATE_EN2_sum") is not None:
                e["tcp_active"] = round(g("TCP_GATE_EN2_sum") / 256.0 / cyc, 4)
            if g("TCP_PENDING_STALL_CYCLES_sum") is not None:
                e["l1_stall"] = round(g("TCP_PENDING_STALL_CYCLES_sum") / 256.0 / cyc, 4)
        if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
            e["lane_utilisation"] = round(g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64.0), 4)
        if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
            e["hbm_bytes"] = round(g("FETCH_SIZE") * 1024 * 2 + g("WRITE_SIZE") * 1024)
            e["hbm_read_bytes"] = round(g("FETCH_SIZE") * 1024 * 2)
            # FETCH_SIZE = 64 B per fabric read request.  A streaming read makes 128-byte requests (hence the guide's x2),
            # a random 16-byte lookup ONE 64-byte request (profiles/r03_calibration.json: k_cal_hbm_gather 63.8 B and 1.00
            # request per lookup, k_cal_hbm_stream 8.0 B per 16 B): for a kernel that mixes both the truth lies between
            e["hbm_bytes_lower"] = round(g("FETCH_SIZE") * 1024 + g("WRITE_SIZE") * 1024)
            e["hbm_write_bytes"] = round(g("WRITE_SIZE") * 1024)
        if g("TCC_REQ_sum") is not None:
            e["l2_bytes_max"] = round(g("TCC_REQ_sum") * 128)
        if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and g("TCC_HIT_sum") + g("TCC_MISS_sum") > 0:
            e["l2_hit_rate"] = round(g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")), 4)
        if g("TCP_TOTAL_ACCESSES_sum"):
            e["l1_hit_rate"] = round(1.0 - (g("TCP_TCC_READ_REQ_sum") or 0.0) / g("TCP_TOTAL_ACCESSES_sum"), 4)
        out["kernels"][k] = e
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
