#!/usr/bin/env python
"""DESIGN.md from docs/design_parts/*.md (+ the matrix table from profiles/r06_matrix.md when it exists)."""
import os
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
parts = ["00_state.md", "01_path.md", "03_oracle_parity.md", "05_layout.md", "06_measurement.md", "07_sections.md", "08_mgpu.md"]
txt = "".join(open(os.path.join(here, p)).read() for p in parts)

numbers = """| config | workload | ms / frame | Mrays/s | latency (one frame in flight) | trace, all launches (isolated) | shade (isolated) | regenerated burst, K = 8 |
|---|---|---|---|---|---|---|---|
| C3 (headline) | sponza_lod 1080p 1 spp 5-bounce GGX + IBL, reference-built `sponza_lod.sbvh` | **3.15** (3.20) | **658** (648) | 4.06 (4.12) | 2.94 (2.96) | 1.08 (1.12) | 3.60 |
| C3, own tree | the same frames through the tree `atns_build_blas_opt` builds (§7d) | **2.95** (3.02) | **703** (686) | | | | |
| C3, reference tree re-arranged | `sponza_lod.sbvh` through `atns_optimize_nodes` | 3.00 (3.09) | 692 (672) | | | | |
| companion | atrium, 250 882 triangles, Disney + textures + IBL + lamp, own tree | **4.70** (4.75) | **441** (436) | 6.40 (6.54) | 4.89 (4.97) | 1.47 (1.53) | 5.18 |
| C2 | Cornell box 1080p 1 spp 5-bounce NEE | **1.03** (1.03) | **2020** (2000) | 1.11 (1.13) | 0.61 (0.61) | 0.57 (0.59) | 1.27 |
| C4 stand-in | atrium 4K 8 spp 8-bounce, all samples traced | **172.5** (177.1) | **385** (375) | 196.6 (205) | 143.6 (149.6) | 50.9 (53.8) | 207.4 |
| C5 | C3 + SVGF passes | **4.26** (4.29) | **487** (483) | 4.83 (4.89) | 2.95 (2.98) | 1.10 (1.15), filters 0.74 | — |
"""
roofline = """`k_trace_fused<true,false,false>` on the headline — average launch 0.554 ms isolated (0.562 in the serialised
PMC passes; rocprofv3's kernel-trace average, 1.24 ms, is wall time under four overlapping frames) — **bound `l1` 0.419**: 1.606 of 3.83 TCP lane
slots per CU-clock, tag lookups 0.417, TCP active 87.5 %; `l2` 0.331, `valu` 0.326 (0.53 of the walk-mix ceiling), `hbm` **0.063** (`traffic` 278 MB per
launch against 8.18 GB of SURVEY §8(d) algorithmic bytes: 14.8 TB/s, a rate); lane utilisation 0.457, L1 / L2 hit 91 / 96 %; `useful` 930 node visits per CU
per µs = 0.464 of the L1-resident chase.  Atrium: `l1` 0.381, `l2` 0.344, `hbm` 0.287, `valu` 0.307, lane utilisation 0.378, L2 hit 80 %, useful 0.356.
Cornell (plain walk over the LDS copy): **`valu` 0.693** of the `v_fma` ceiling = 1.12 × the ceiling of its own instruction mix, everything else ≤ 0.16.
C4 (atrium 4K): `l2` 0.485, `l1` 0.455, `hbm` 0.369, `valu` 0.374, TCP active 94 %.  `k_shade`: 0.216 ms per launch, 936 MB of HBM-side traffic per launch = 0.54
of the peak by the ×2 rule, 2.55 × its compulsory bytes (lower bound 1.57 ×); atrium 3.26 × (1.97 ×), C4 3.54 × (2.12 ×) — and VALU-issue-bound all the same (§0, §7f)."""
matrix_path = os.path.join(root, "profiles", "r06_matrix.md")
matrix = open(matrix_path).read() if os.path.exists(matrix_path) else "(profiles/r06_matrix.md: not collected yet)\n"
txt = txt.replace("@@NUMBERS_TABLE@@", numbers).replace("@@ROOFLINE_SENTENCE@@", roofline).replace("@@MATRIX_TABLE@@", matrix)
open(os.path.join(root, "DESIGN.md"), "w").write(txt)
print(len(txt.encode()), "bytes")
