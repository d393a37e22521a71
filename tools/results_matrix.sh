#!/bin/bash
# The results table BASELINE.json's north_star asks for, in one command on the GPU box: {Cornell box, sponza_lod on the reference-built
# tree, atrium} x {1080p, 4K} x {1 spp, 8 spp with the CPU renderer's break-on-terminate sample loop, 8 spp all samples traced}, 5 bounces.
# Per cell: the PMC passes (tools/pmc_collect.sh -> profiles/<tag>_counters_<cell>.json) and then the bench line, which reads them:
# ms / frame as throughput and as latency, Msamples/s, Mray-segments/s, the dominant kernel with its roofline fractions, HBM GB/s, VALU
# busy, the CPU baseline on this box (the benchmarked frame, or a stated reduced frame).  tools/results_matrix.py turns the lines into
# profiles/<tag>_matrix.json and a markdown table.
# usage: tools/results_matrix.sh <tag e.g. r06> [cells-regex]
set -u
TAG=${1:?tag}; ONLY=${2:-.}
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${TAG}_matrix
mkdir -p "$OUT" profiles
cell() {    # name  workload-tag  bench-args...
  local NAME=$1 WL=$2; shift 2
  echo "$NAME" | grep -Eq "$ONLY" || return 0
  bash tools/pmc_collect.sh "$OUT/$NAME/pmc" "1 2 3 4 5 6 7" "$@" --no-companion --no-own-tree > "$OUT/$NAME.pmc.log" 2>&1
  python tools/pmc_to_json.py "$OUT/$NAME/pmc" "$WL" > "profiles/${TAG}_counters_matrix_${NAME}.json"
  cp "profiles/${TAG}_counters_matrix_${NAME}.json" "$OUT/"      # (gpurun brings back gpurun_out/ only: copy from there into profiles/)
  rm -rf "$OUT/$NAME/pmc"
  [ "${MATRIX_PMC_ONLY:-0}" = "1" ] && return 0
  timeout 900 python bench.py "$@" --no-companion --no-own-tree --cpu-budget 8 --repeats 3 --min-timed-seconds 1 > "$OUT/$NAME.json" 2> "$OUT/$NAME.err"
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/$NAME.json").read().strip().splitlines()[-1])
    print("$NAME", d["value"], d["unit"], d["ms_per_step"], "ms; latency", d["ms_per_frame_latency"], "roofline", {k: d["roofline"].get(k) for k in ("bound", "frac", "kernel")})
except Exception as e:
    print("$NAME FAILED", e); print(open("$OUT/$NAME.err").read()[-600:])
PY
}
for RES in "1920 1080 1080p" "3840 2160 4k"; do
  set -- $RES; W=$1; H=$2; R=$3
  for SC in "cornell cornell" "sponza sponza_lod" "atrium atrium"; do
    set -- $SC; SCENE=$1; ST=$2
    cell ${SCENE}_${R}_1spp      "$ST ${W}x${H} 1spp 5-bounce"        --scene $SCENE --width $W --height $H --spp 1 --depth 5
    cell ${SCENE}_${R}_8spp_brk  "$ST ${W}x${H} 8spp 5-bounce break"  --scene $SCENE --width $W --height $H --spp 8 --depth 5 --steps 24
    cell ${SCENE}_${R}_8spp_all  "$ST ${W}x${H} 8spp 5-bounce"        --scene $SCENE --width $W --height $H --spp 8 --depth 5 --steps 24 --all-samples
  done
done
[ "${MATRIX_PMC_ONLY:-0}" = "1" ] || python tools/results_matrix.py "$OUT" "$TAG"
