#!/usr/bin/env python
"""Joins tools/valu_calib.hip's own timings (plain.jsonl) with the per-kernel PMC averages of the rocprofv3 passes
(tools/calib_collect.sh) into the calibration record bench.py reads: what the counters behind the roofline object read
when a kernel SATURATES the resource they are meant to measure on gfx950.

    python tools/calib_to_json.py gpurun_out/r03_calib > profiles/r03_calibration.json

cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs).  Ceilings:
  valu_busy_fma / valu_busy_walkmix = SQ_ACTIVE_INST_VALU / 256 / cycles at saturation (the reading rocprofiler's gfx94x
      VALUBusy formula gives for a kernel that does nothing but issue VALU from 8 waves per SIMD);
  valu_insts_per_simd_cycle_*       = SQ_INSTS_VALU / 1024 SIMDs / cycles (wave-instructions per SIMD per cycle);
  tcp_accesses_per_cu_cycle_*       = TCP_TOTAL_ACCESSES / 256 / cycles at saturation, random and row-shaped 16-B gathers;
  chase_*                           = dependent 2 x 16-B hops per CU per microsecond at 1 / 5 / 8 waves per SIMD.
"""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict


def short(n):
    return n.replace("void ", "").split("(")[0]


def main(d):
    plain = [json.loads(l) for l in open(os.path.join(d, "plain.jsonl")) if l.startswith("{")]
    acc = defaultdict(lambda: defaultdict(list))
    order = defaultdict(list)
    for f in sorted(glob.glob(os.path.join(d, "pass*", "*counter_collection.csv"))):
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            if not k.startswith("k_cal"):
                continue
            acc[(k, row["Dispatch_Id"], os.path.basename(os.path.dirname(f)))][row["Counter_Name"]].append(float(row["Counter_Value"]))
    # a kernel is launched 1 warm-up + 5 timed times per variant, variants in program order: group dispatches of a
    # kernel name into runs of 6 and average each run (the variants of k_cal_l1_gather / the tables of k_cal_l1_chase)
    per_pass = defaultdict(lambda: defaultdict(list))
    for (k, disp, ps), c in acc.items():
        per_pass[(k, ps)][int(disp)].append(c)
    variants = defaultdict(lambda: defaultdict(dict))       # kernel -> variant index -> counter -> mean
    for (k, ps), dd in per_pass.items():
        ids = sorted(dd)
        for vi in range(len(ids) // 6):
            run = ids[vi * 6 + 1: vi * 6 + 6]
            sums = defaultdict(list)
            for i in run:
                for c in dd[i]:
                    for n, v in c.items():
                        sums[n].append(sum(v))
            for n, v in sums.items():
                variants[k][vi][n] = sum(v) / len(v)
    kernels = []
    seen = defaultdict(int)
    for p in plain:
        k = p["kernel"]
        vi = seen[k]; seen[k] += 1
        c = variants.get(k, {}).get(vi, {})
        e = dict(p)
        e["counters"] = {n: round(v, 1) for n, v in sorted(c.items())}
        if c.get("GRBM_GUI_ACTIVE"):
            cyc = c["GRBM_GUI_ACTIVE"] / 8.0
            e["cycles"] = round(cyc)
            e["clock_GHz_profiled"] = round(cyc / (p["ms"] * 1e6), 3)
            if "SQ_ACTIVE_INST_VALU" in c:
                e["valu_busy"] = round(c["SQ_ACTIVE_INST_VALU"] / 256.0 / cyc, 4)
            if "SQ_INSTS_VALU" in c:
                e["valu_insts_per_simd_cycle"] = round(c["SQ_INSTS_VALU"] / 1024.0 / cyc, 4)
            if "SQ_ACTIVE_INST_ANY" in c:
                e["any_busy"] = round(c["SQ_ACTIVE_INST_ANY"] / 256.0 / cyc, 4)
            if "TCP_TOTAL_ACCESSES_sum" in c:
                e["tcp_accesses_per_cu_cycle"] = round(c["TCP_TOTAL_ACCESSES_sum"] / 256.0 / cyc, 4)
                if "lane_loads_16B" in p:
                    e["tcp_accesses_per_lane_load"] = round(c["TCP_TOTAL_ACCESSES_sum"] / p["lane_loads_16B"], 4)
            if "TCP_TOTAL_CACHE_ACCESSES_sum" in c:
                e["tcp_cache_accesses_per_cu_cycle"] = round(c["TCP_TOTAL_CACHE_ACCESSES_sum"] / 256.0 / cyc, 4)
            if "SQ_INSTS_VMEM_RD" in c:
                e["vmem_rd_insts_per_cu_cycle"] = round(c["SQ_INSTS_VMEM_RD"] / 256.0 / cyc, 5)
        if "FETCH_SIZE" in c and "lookups_16B" in p:
            e["fetch_size_bytes_per_lookup_raw"] = round(c["FETCH_SIZE"] * 1024.0 / p["lookups_16B"], 2)
            if "TCC_EA0_RDREQ_sum" in c:
                e["tcc_ea_rdreq_per_lookup"] = round(c["TCC_EA0_RDREQ_sum"] / p["lookups_16B"], 4)
                e["tcc_ea_rdreq_32B_per_lookup"] = round(c.get("TCC_EA0_RDREQ_32B_sum", 0.0) / p["lookups_16B"], 4)
                if "TCC_BUBBLE_sum" in c:
                    e["tcc_bubble_per_lookup"] = round(c["TCC_BUBBLE_sum"] / p["lookups_16B"], 4)
        if "FETCH_SIZE" in c and "hops" in p and "64MB" in str(p.get("table", "")):
            hops_total = p["lane_hops_per_us"] * p["ms"] * 1e3
            e["fetch_size_bytes_per_hop_raw"] = round(c["FETCH_SIZE"] * 1024.0 / hops_total, 2)
        kernels.append(e)

    def find(k, **kw):
        for e in kernels:
            if e["kernel"] == k and all(str(kw[a]) in str(e.get(a, "")) for a in kw):
                return e
        return {}
    ceil = {
        "valu_busy_fma": find("k_cal_valu_fma").get("valu_busy"),
        "valu_insts_per_simd_cycle_fma": find("k_cal_valu_fma").get("valu_insts_per_simd_cycle"),
        "valu_busy_fma_sgpr": find("k_cal_valu_fma_sgpr").get("valu_busy"),
        "valu_insts_per_simd_cycle_fma_sgpr": find("k_cal_valu_fma_sgpr").get("valu_insts_per_simd_cycle"),
        "valu_busy_walkmix": find("k_cal_valu_walkmix").get("valu_busy"),
        "valu_insts_per_simd_cycle_walkmix": find("k_cal_valu_walkmix").get("valu_insts_per_simd_cycle"),
        "tcp_accesses_per_cu_cycle_random": find("k_cal_l1_gather", variant="random (one").get("tcp_accesses_per_cu_cycle"),
        "tcp_accesses_per_cu_cycle_rows": find("k_cal_l1_gather", variant="rows").get("tcp_accesses_per_cu_cycle"),
        "tcp_cache_accesses_per_cu_cycle_random": find("k_cal_l1_gather", variant="random (one").get("tcp_cache_accesses_per_cu_cycle"),
        "tcp_cache_accesses_per_cu_cycle_rows": find("k_cal_l1_gather", variant="rows").get("tcp_cache_accesses_per_cu_cycle"),
        "tcp_accesses_per_cu_cycle_pairs": find("k_cal_l1_gather", variant="pairs").get("tcp_accesses_per_cu_cycle"),
        "tcp_cache_accesses_per_cu_cycle_pairs": find("k_cal_l1_gather", variant="pairs").get("tcp_cache_accesses_per_cu_cycle"),
        "tcp_accesses_per_lane_load_random": find("k_cal_l1_gather", variant="random (one").get("tcp_accesses_per_lane_load"),
        "lane_loads_per_cu_per_us_random": find("k_cal_l1_gather", variant="random (one").get("lane_loads_per_cu_per_us"),
        "lane_loads_per_cu_per_us_rows": find("k_cal_l1_gather", variant="rows").get("lane_loads_per_cu_per_us"),
    }
    ceil["fetch_size_raw_bytes_per_random_16B_lookup"] = find("k_cal_hbm_gather").get("fetch_size_bytes_per_lookup_raw")
    ceil["tcc_ea_rdreq_per_random_16B_lookup"] = find("k_cal_hbm_gather").get("tcc_ea_rdreq_per_lookup")
    ceil["tcc_ea_rdreq_32B_per_random_16B_lookup"] = find("k_cal_hbm_gather").get("tcc_ea_rdreq_32B_per_lookup")
    st = find("k_cal_hbm_stream")
    ceil["stream_fetch_size_raw_bytes_per_16B"] = st.get("fetch_size_bytes_per_lookup_raw")
    ceil["stream_tcc_ea_rdreq_per_16B"] = st.get("tcc_ea_rdreq_per_lookup")
    ceil["stream_tcc_bubble_per_16B"] = st.get("tcc_bubble_per_lookup")
    ceil["random_tcc_bubble_per_lookup"] = find("k_cal_hbm_gather").get("tcc_bubble_per_lookup")
    ceil["stream_GBps"] = st.get("useful_GBps")
    ceil["hbm_random_16B_lookups_per_us"] = find("k_cal_hbm_gather").get("lookups_per_us")
    for w in (1, 5, 8):
        for tab, key in (("16KB", "l1"), ("1MB", "l2"), ("64MB", "mall")):
            e = find("k_cal_l1_chase<%d>" % w, table=tab)
            ceil["chase_%s_w%d_lane_hops_per_cu_per_us" % (key, w)] = e.get("lane_hops_per_cu_per_us")
            ceil["chase_%s_w%d_ns_per_hop" % (key, w)] = e.get("ns_per_hop")
    try:
        head = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=os.path.dirname(os.path.abspath(__file__)),
                                       stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        head = None
    # the record names the micro-benchmark source it was measured with (content hash: there is no .git on the GPU box);
    # bench.py refuses ceilings whose source is not the tools/valu_calib.hip in the tree
    import hashlib
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "valu_calib.hip")
    json.dump({"source": "tools/valu_calib.hip under rocprofv3 --pmc (tools/calib_collect.sh), MI355X gfx950", "git_head": head,
               "calib_source_sha16": hashlib.sha256(open(src, "rb").read()).hexdigest()[:16],
               "ceilings": ceil, "kernels": kernels}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1])
