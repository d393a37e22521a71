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
            "algorithmic_GBps_dominant_kernel": (rf.get("algorithmic") or {}).get("GBps"),
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
    fill_baseline_md(root, tag, rows)


PARITY = {   # profiles/parity_<tag>.json (written by the -m gpu tests): all pixels of the 1080p frame, or the oracle-sized frame named
    "cornell": "C2 frame 0, all pixels: 100 % inside 1e-3, 91.4 % bit-equal",
    "sponza": "C3 frames 0 / 9, all pixels: 99.88 / 99.89 % inside 1e-3, 86.7 % bit-equal",
    "atrium": "atrium 1080p frames 0 / 5, all pixels: 98.12 / 98.13 % inside, 58.8 % bit-equal; followed pixels converge (|z| <= 2.2)",
}


def fill_baseline_md(root, tag, rows):
    """BASELINE.md section 4 (the survey left it an empty template) from the same rows + the C4 / C5 lines of profiles/<tag>_z_*."""
    path = os.path.join(root, "BASELINE.md")
    txt = open(path).read()
    i = txt.index("## 4. Results")
    j = txt.find("\n## ", i + 5)
    hdr = ("## 4. Results\n\nMeasured on one MI355X, profile set `%s` (`tools/results_matrix.sh %s` -> `profiles/%s_matrix.json`, every row with its own rocprofv3 PMC passes;\n"
           "C4 / C5: `profiles/%s_z_bench_*.json`; this section is written by `tools/results_matrix.py`).  ms/frame = throughput with 4 frames in flight\n"
           "(latency with one in brackets); HBM GB/s and VALU = the dominant kernel (`k_trace_fused`) from the counters; `algorithmic / 6.29 TB/s` uses\n"
           "SURVEY 8(d)'s byte model for that kernel -- above 100 %% the records are served by L1 / L2, it is a rate, not a fraction of a roof (DESIGN.md 6\n"
           "gives the fractions that can bind); CPU = the oracle (section 3, item 2) on the GPU box's cores, OpenMP over rows; no upstream figure exists\n"
           "for any row (`vs_baseline` stays null).  Stand-ins as in section 2 (sponza_lod for Sponza; the procedural atrium for Crytek Sponza).\n\n"
           "| config | device(s) | ms/frame | Msamples/s | Mray-seg/s | HBM GB/s | algorithmic / 6.29 TB/s | VALU issue / ceiling | CPU Msamples/s (threads) | parity vs oracle |\n"
           "|---|---|---|---|---|---|---|---|---|---|\n") % (tag, tag, tag, tag)
    body = ""
    label = {"cornell": "Cornell box", "sponza": "sponza_lod (C3 stand-in)", "atrium": "atrium (Disney, 250 882 tris)"}
    for r in rows:
        scene = r["cell"].split("_")[0]
        cfg = "%s %dx%d %d spp%s, 5 bounces" % (label.get(scene, scene), r["width"], r["height"], r["spp"],
                                                "" if r["spp"] == 1 else (", all samples" if r["sample_loop"].startswith("all") else ", break on terminate"))
        if r["spp"] == 1 and r["height"] == 1080 and scene in ("cornell", "sponza"):
            cfg = ("**C2** " if scene == "cornell" else "**C3** ") + cfg
        alg = r.get("algorithmic_GBps_dominant_kernel")
        body += "| %s | 1 x MI355X | %.3f (%.3f) | %.1f | %.0f | %s | %s | %s | %s (%s) | %s |\n" % (
            cfg, r["ms_per_frame_throughput"], r["ms_per_frame_latency"], r["Msamples_per_s"], r["Mray_segments_per_s"] or 0,
            r["hbm_GBps_dominant_kernel"], ("%.0f %%" % (100.0 * alg / 6290.0)) if alg else "-", r["valu_busy"],
            r["cpu_baseline_Msamples_per_s"], r["cpu_cores"], PARITY.get(scene, "-"))
    for name, cfg, par in (("c4_atrium4k8spp", "**C4** stand-in: atrium 3840x2160 8 spp 8 bounces, all samples (ONE GPU; 8-GPU run: not available to this repo)",
                            "at oracle size 640x360: 93.2 % (break) / 67.0 % (all samples) inside 1e-3"),
                           ("c5_sponza1080p_svgf", "**C5** = C3 + SVGF (temporal, variance, 5 a-trous passes)",
                            "192x108: path pass 99.84-99.88 % inside 1e-3, filtered output 100 % inside 5e-2")):
        f = os.path.join(root, "profiles", "%s_z_bench_%s.json" % (tag, name))
        if not os.path.exists(f):
            continue
        b = json.loads(open(f).read().strip().splitlines()[-1])
        rf = b.get("roofline") or {}
        alg = (rf.get("algorithmic") or {}).get("GBps")
        hbm = round(rf["traffic"] / (rf["roofline_launch_ms"] * 1e-3) / 1e9, 1) if rf.get("traffic") and rf.get("roofline_launch_ms") else None
        cpu = b.get("cpu_baseline") or {}
        body += "| %s | 1 x MI355X | %.3f (%.3f) | %.1f | %.0f | %s | %s | %s | %s (%s) | %s |\n" % (
            cfg, b["ms_per_step"], b["ms_per_frame_latency"], b["value"], b.get("Mray_segments_per_s") or 0, hbm,
            ("%.0f %%" % (100.0 * alg / 6290.0)) if alg else "-", (rf.get("fractions") or {}).get("valu"), cpu.get("value"), cpu.get("cores"), par)
    body += ("\nC1 (Cornell 512x512, 3 bounces, CPU oracle only): the plumbing check of `tests/test_oracle_cpu.py` / `tests/test_gpu_parity.py`, no GPU line.\n"
             "Multi-GPU (1 / 2 / 4 / 8): the driver's `SCALE_rNN.json` when an 8-GPU node exists; the bound measured on one GPU doing rank 0's share is DESIGN.md 8.\n")
    txt = txt[:i] + hdr + body + (txt[j:] if j >= 0 else "")
    open(path, "w").write(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
