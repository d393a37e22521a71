#!/usr/bin/env python
"""DESIGN.md from docs/design_parts/*.md; every measured number in section 0's table, section 6's result sentence and the matrix /
regeneration tables is read from profiles/r06_z_bench_*.json and profiles/r06_matrix.json.  --out FILE writes elsewhere
(tests/test_docs_cpu.py holds DESIGN.md against it)."""
import os
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
parts = ["00_state.md", "01_path.md", "03_oracle_parity.md", "05_layout.md", "06_measurement.md", "07_sections.md", "08_mgpu.md"]
txt = "".join(open(os.path.join(here, p)).read() for p in parts)

import json


def bench(name):
    return json.load(open(os.path.join(root, "profiles", "r06_z_bench_%s.json" % name)))


def trace_ms(b):
    k = b["kernel_ms_per_frame_isolated"]
    return k.get("trace_fused", 0.0) + k.get("trace_closest", 0.0)


def numbers_row(label, workload, b, r05, filters=False):
    # r05 = (ms, Mrays, latency, trace, shade) from profiles/r05_z_* for the brackets
    rg = (b["config"].get("regeneration") or {}).get("ms_per_frame")
    dec = 1 if b["ms_per_step"] >= 100 else 2
    f = lambda v: "%.*f" % (dec, v)
    sh = b["kernel_ms_per_frame_isolated"]["shade"]
    shade = "%s (%s)" % (f(sh), r05[4]) + (", filters %.2f" % sum(v for k, v in b["kernel_ms_per_frame_isolated"].items() if k.startswith("svgf_")) if filters else "")
    return "| %s | %s | **%s** (%s) | **%.0f** (%s) | %s (%s) | %s (%s) | %s | %s |\n" % (
        label, workload, f(b["ms_per_step"]), r05[0], b["value"], r05[1], f(b["ms_per_frame_latency"]), r05[2], f(trace_ms(b)), r05[3], shade,
        f(rg) if rg else "—")


d = bench("default")
c3, atr, c2, c4, c5 = bench("c3_sponza1080p"), bench("atrium1080p"), bench("c2_cornell1080p"), bench("c4_atrium4k8spp"), bench("c5_sponza1080p_svgf")
numbers = ("| config | workload | ms / frame | Mrays/s | latency (one frame in flight) | trace, all launches (isolated) | shade (isolated) | regenerated burst, K = 8 |\n"
           "|---|---|---|---|---|---|---|---|\n"
           + numbers_row("C3 (headline)", "sponza_lod 1080p 1 spp 5-bounce GGX + IBL, reference-built `sponza_lod.sbvh`", c3, ("3.20", "648", "4.12", "2.96", "1.12"))
           + "| C3, own tree | the same frames through the tree `atns_build_blas_opt` builds (§7d) | **%.2f** (3.02) | **%.0f** (686) | | | | |\n" % (d["own_tree"]["ms_per_step"], d["own_tree"]["value"])
           + "| C3, reference tree re-arranged | `sponza_lod.sbvh` through `atns_optimize_nodes` | %.2f (3.09) | %.0f (672) | | | | |\n" % (
               d["reference_tree_optimized"]["ms_per_step"], d["reference_tree_optimized"]["value"])
           + numbers_row("companion", "atrium, 250 882 triangles, Disney + textures + IBL + lamp, own tree", atr, ("4.75", "436", "6.54", "4.97", "1.53"))
           + numbers_row("C2", "Cornell box 1080p 1 spp 5-bounce NEE", c2, ("1.03", "2000", "1.13", "0.61", "0.59"))
           + numbers_row("C4 stand-in", "atrium 4K 8 spp 8-bounce, all samples traced", c4, ("177.1", "375", "205", "149.6", "53.8"))
           + numbers_row("C5", "C3 + SVGF passes", c5, ("4.29", "483", "4.89", "2.98", "1.15"), filters=True))


def fr(b):
    return b["roofline"]["fractions"]


r3, ra, r2, r4 = c3["roofline"], atr["roofline"], c2["roofline"], c4["roofline"]
roofline = ("`k_trace_fused<true,false,false>` on the headline — average launch %.3f ms isolated (%.3f in the serialised PMC passes; the "
            "kernel-trace average, %.2f ms, is wall time under four overlapping frames) — **bound `l1` %.3f**: %.3f of %.2f TCP lane slots per CU-clock, "
            "tag lookups %.3f, TCP active %.1f %%; `l2` %.3f, `valu` %.3f (%.2f of the walk-mix ceiling), `hbm` **%.3f** (`traffic` %.0f MB per launch against "
            "%.2f GB of SURVEY §8(d) algorithmic bytes: %.1f TB/s, a rate); lane utilisation %.3f, L1 / L2 hit %.0f / %.0f %%; `useful` %.0f node visits "
            "per CU per µs = %.3f of the L1-resident chase.  " % (
                r3["roofline_launch_ms"], r3["pmc"]["avg_launch_ms_profiled"], r3["avg_launch_ms"], r3["frac"], r3["achieved"], r3["peak"],
                r3["fraction_detail"]["l1_tag_lookups"], 100 * r3["fraction_detail"]["tcp_active"], fr(c3)["l2"], fr(c3)["valu"],
                r3["fraction_detail"]["valu_vs_packed_mix_ceiling"], fr(c3)["hbm"], r3["traffic"] / 1e6, r3["algorithmic"]["bytes_per_launch"] / 1e9,
                r3["algorithmic"]["GBps"] / 1e3, r3["pmc"]["lane_utilisation"], 100 * r3["pmc"]["l1_hit_rate"], 100 * r3["pmc"]["l2_hit_rate"],
                r3["useful"]["node_visits_per_cu_per_us"], r3["useful"]["node_visits_per_cu_per_us"] / r3["useful"]["l1_resident_chase_ceiling"])
            + "Atrium: `l1` %.3f, `l2` %.3f, `hbm` %.3f, `valu` %.3f, lane utilisation %.3f, L2 hit %.0f %%, useful %.3f.  " % (
                fr(atr)["l1"], fr(atr)["l2"], fr(atr)["hbm"], fr(atr)["valu"], ra["pmc"]["lane_utilisation"], 100 * ra["pmc"]["l2_hit_rate"],
                ra["useful"]["node_visits_per_cu_per_us"] / ra["useful"]["l1_resident_chase_ceiling"])
            + "Cornell (plain walk over the LDS copy): **`valu` %.3f** of the `v_fma` ceiling = %.2f × the ceiling of its own instruction mix, everything else ≤ %.2f.  " % (
                fr(c2)["valu"], r2["fraction_detail"]["valu_vs_packed_mix_ceiling"], max(v for k, v in fr(c2).items() if k != "valu"))
            + "C4 (atrium 4K): `l2` %.3f, `l1` %.3f, `hbm` %.3f, `valu` %.3f, TCP active %.0f %%.  " % (
                fr(c4)["l2"], fr(c4)["l1"], fr(c4)["hbm"], fr(c4)["valu"], 100 * r4["fraction_detail"]["tcp_active"])
            + "`k_shade`: %.3f ms per launch, %.0f MB of HBM-side traffic per launch = %.2f of the peak by the ×2 rule, %.2f × its compulsory bytes (lower bound %.2f ×); "
              "atrium %.2f × (%.2f ×), C4 %.2f × (%.2f ×) — and VALU-issue-bound all the same (§0, §7f)." % (
                r3["shade"]["avg_launch_ms"], r3["shade"]["traffic"] / 1e6, r3["shade"]["frac"], r3["shade"]["traffic_over_compulsory"],
                r3["shade"]["traffic_over_compulsory_lower_bound"], ra["shade"]["traffic_over_compulsory"], ra["shade"]["traffic_over_compulsory_lower_bound"],
                r4["shade"]["traffic_over_compulsory"], r4["shade"]["traffic_over_compulsory_lower_bound"]))
txt = txt.replace("@@C4_MS@@", "%.1f" % c4["ms_per_step"])
matrix_path = os.path.join(root, "profiles", "r06_matrix.json")
if os.path.exists(matrix_path):
    # (condensed: profiles/r06_matrix.md has the Mray-segments/s, HBM GB/s and VALU columns too)
    rows = json.load(open(matrix_path))["rows"]
    matrix = ("| scene | size | spp, sample loop | ms / frame (4 in flight) | latency | Msamples/s | `k_trace_fused`: bound, fraction | HBM GB/s | CPU Msamples/s (16 threads) | regenerated bursts |\n"
              "|---|---|---|---|---|---|---|---|---|---|\n")
    for r in rows:
        rg = r["regeneration"] or {}
        loop = {"one sample": "1", "all samples traced": "8, all", "break on terminate (pathtracing.cpp:350-352)": "8, break"}.get(r["sample_loop"], r["sample_loop"])
        matrix += "| %s | %s | %s | %.2f | %.2f | %.0f | %s %.2f | %.0f | %.1f | %.2f |\n" % (
            r["cell"].split("_")[0].replace("sponza", "sponza_lod"), "1080p" if r["height"] == 1080 else "4K", loop, r["ms_per_frame_throughput"], r["ms_per_frame_latency"],
            r["Msamples_per_s"], r["bound"], r["roofline_frac"] or 0.0, r["hbm_GBps_dominant_kernel"] or 0.0, r["cpu_baseline_Msamples_per_s"] or 0.0, rg.get("ms_per_frame") or 0.0)
    cell = {r["cell"]: r for r in rows}

    def pair(name):
        r = cell[name]
        rg = r["regeneration"]
        sv, pv = r["ms_per_frame_throughput"], rg["ms_per_frame"]
        f = (lambda v: "%.2f" % v) if sv < 20 else (lambda v: "%.1f" % v)
        return ("**%s** \\| %s" if sv <= pv else "%s \\| **%s**") % (f(sv), f(pv))

    order = ["sponza_1080p", "atrium_1080p", "cornell_1080p", "sponza_4k", "atrium_4k", "cornell_4k"]
    regen_ms = ("| 8 spp, ms per frame | sponza_lod 1080p | atrium 1080p | Cornell 1080p | sponza_lod 4K | atrium 4K | Cornell 4K |\n|---|---|---|---|---|---|---|\n"
                "| break on terminate (the CPU renderer's loop, `pathtracing.cpp:350-352`) | " + " | ".join(pair(c + "_8spp_brk") for c in order) + " |\n"
                "| all samples traced | " + " | ".join(pair(c + "_8spp_all") for c in order) + " |\n")

    def pct(name):
        return "%+.0f %%" % (100.0 * (cell[name]["regeneration"]["speedup"] - 1.0))

    def arrow(name):
        r = cell[name]
        return "%.2f → %.2f" % (r["ms_per_frame_throughput"], r["regeneration"]["ms_per_frame"])
    alls = [100.0 * (cell[c + "_8spp_all"]["regeneration"]["speedup"] - 1.0) for c in order]
    regen_break = ("sponza_lod 1080p 8 spp %s ms per frame, atrium %s (4K: %s / %s; Cornell %s; all samples traced %.0f … %.0f %%)" % (
        arrow("sponza_1080p_8spp_brk"), arrow("atrium_1080p_8spp_brk"), pct("sponza_4k_8spp_brk"), pct("atrium_4k_8spp_brk"),
        pct("cornell_1080p_8spp_brk"), max(alls), min(alls))).replace("-", "−")
    regen_gain = "%s / %s" % (pct("sponza_1080p_8spp_brk"), pct("atrium_1080p_8spp_brk"))
    bad = [r["cell"] for r in rows if r["regeneration"] and not r["regeneration"]["film_equals_serial"]]
    if bad:
        print("WARNING: film_equals_serial false in", bad)
else:
    matrix = regen_ms = "(profiles/r06_matrix.json: not collected yet)\n"
    regen_break = regen_gain = "(not collected)"
txt = txt.replace("@@REGEN_MS_TABLE@@", regen_ms).replace("@@REGEN_BREAK_SENTENCE@@", regen_break).replace("@@REGEN_GAIN@@", regen_gain)
txt = txt.replace("@@NUMBERS_TABLE@@", numbers).replace("@@ROOFLINE_SENTENCE@@", roofline).replace("@@MATRIX_TABLE@@", matrix)
import sys
out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else os.path.join(root, "DESIGN.md")
open(out, "w").write(txt)
print(len(txt.encode()), "bytes")
