#!/bin/bash
# k_shade traffic by ingredient: HBM-side bytes per launch (FETCH_SIZE / WRITE_SIZE passes) for the sponza bench with
# textures and / or the environment map removed.  usage: tools/traffic_exp.sh <outdir>
set -u
OUT=$1
cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
for EX in "" "notex" "noibl" "notex,noibl"; do
    NAME=${EX:-full}; NAME=${NAME//,/_}
    if [ -n "$EX" ]; then A="--experiment $EX"; else A=""; fi
    bash tools/pmc_collect.sh "$OUT/$NAME" "6 7" --config c3 $A > "$OUT/$NAME.log" 2>&1
    python tools/pmc_to_json.py "$OUT/$NAME" "$NAME" > "$OUT/counters_$NAME.json"
    python - <<PY
import json
d = json.load(open("$OUT/counters_$NAME.json"))
for k, v in d["kernels"].items():
    if "shade" in k or "fused" in k:
        print("$NAME", k[:28], "read MB", round(v["hbm_read_bytes"] / 1e6, 1), "write MB", round(v["hbm_write_bytes"] / 1e6, 1), "launches", v["launches_sampled"])
PY
done
