#!/usr/bin/env python
"""profiles/<tag>_matrix.json + profiles/<tag>_matrix.md from the bench lines tools/results_matrix.sh wrote (columns of BASELINE.md
section 4).    usage: tools/results_matrix.py <dir with the cells' json> <tag>"""
import glob
import json
import os
import sys


def main(d, tag):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rows = []
    order = {"cornell": 0, "sponza": 1, "atrium": 2}
    for f in sorted(glob.glob(os.path.join(d, "*.json"))):
        name = os.path.basename(f)[:-5]
        try:
            b = json.loads(open(f).read().strip().splitlines()[-1])
        except Exception:
            continue
        rf = b.get("roofline") or {}
        cpu = b.get("cpu_baseline") or {}
        fr = rf.get("fractions") or {}
        rows.append({
            "cell": name, "workload": b["config"]["workload"], "width": b["config"]["width"], "height": b["config"]["height"], "spp": b["config"]["spp"],
            "sample_loop": ("one sample" if b["config"]["spp"] == 1 else ("all samples traced" if "all samples" in b["config"]["workload"] or name.endswith("_all") else "break on terminate (pathtracing.cpp:350-352)")),
            "ms_per_frame_throughput": b["ms_per_step"], "frames_in_flight": b["config"]["frames_in_flight"], "ms_per_frame_latency": b["ms_per_frame_latency"],
            "Msamples_per_s": b["value"], "Mray_segments_per_s": b.get("Mray_segments_per_s"), "ray_segments_per_frame": b.get("ray_segments_per_frame"),
            "dominant_kernel": rf.get("kernel"), "bound": rf.get("bound"), "roofline_frac": rf.get("frac"), "fractions": fr,
            "achieved": rf.get("achieved"), "peak": rf.get("peak"), "unit": rf.get("unit"),
            "hbm_GBps_dominant_kernel": (round(rf["traffic"] / (rf["roofline_launch_ms"] * 1e-3) / 1e9, 1) if rf.get("traffic") and rf.get("roofline_launch_ms") else None),
            "traffic_bytes_per_launch": rf.get("traffic"),
            "valu_busy": fr.get("valu"), "lane_utilisation": (rf.get("pmc") or {}).get("lane_utilisation"),
            "cpu_baseline_Msamples_per_s": cpu.get("value"), "cpu_cores": cpu.get("cores"), "cpu_sample": cpu.get("sample"),
            "regeneration": b["config"].get("regeneration"), "film_sha256": b.get("film_sha256"),
            "parity_reference": "tests/test_gpu_parity.py, test_gpu_config4.py, test_gpu_regen.py (films vs the CPU oracle at oracle-sized frames; byte-equal across schedules)",
        })
    rows.sort(key=lambda r: (r["height"], order.get(r["cell"].split("_")[0], 9), r["spp"], r["sample_loop"]))
    json.dump({"tag": tag, "rows": rows}, open(os.path.join(root, "profiles", "%s_matrix.json" % tag), "w"), indent=1)
    with open(os.path.join(root, "profiles", "%s_matrix.md" % tag), "w") as f:
        f.write("| scene | size | spp / sample loop | ms/frame (throughput, N in flight) | ms/frame (latency) | Msamples/s | Mray-segments/s | dominant kernel: bound, fraction | HBM GB/s (that kernel) | VALU issue / ceiling | CPU baseline Msamples/s (cores) | regenerated burst ms/frame |\n")
        f.write("|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for r in rows:
            rg = r["regeneration"] or {}
            f.write("| %s | %dx%d | %d, %s | %.3f (%d) | %.3f | %.1f | %s | %s: %s, %s | %s | %s | %s (%s) | %s |\n" % (
                r["cell"].split("_")[0], r["width"], r["height"], r["spp"], r["sample_loop"].split(" (")[0], r["ms_per_frame_throughput"], r["frames_in_flight"],
                r["ms_per_frame_latency"], r["Msamples_per_s"], r["Mray_segments_per_s"], r["dominant_kernel"], r["bound"], r["roofline_frac"],
                r["hbm_GBps_dominant_kernel"], r["valu_busy"], r["cpu_baseline_Msamples_per_s"], r["cpu_cores"], rg.get("ms_per_frame")))
    print(open(os.path.join(root, "profiles", "%s_matrix.md" % tag)).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
