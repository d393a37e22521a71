#!/bin/bash
# One round's measurement set for a bench configuration, on the GPU box:
#   1. PMC passes (tools/pmc_collect.sh) -> profiles/<tag>_counters_<name>.json (tools/pmc_to_json.py)
#   2. rocprofv3 --kernel-trace --stats of the same command -> <tag>_<name>_kernel_stats.txt
#   3. the bench line itself (which reads the counters file written in step 1)
# usage: tools/profile_round.sh <tag e.g. r02_a> <name e.g. c3_sponza1080p> "<workload tag>" [bench args...]
set -u
TAG=$1; NAME=$2; WORKLOAD=$3; shift 3
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG/$NAME
mkdir -p "$OUT" profiles
bash tools/pmc_collect.sh "$OUT/pmc" "1 2 3 4 5 6 7" "$@" > "$OUT/pmc.log" 2>&1
python tools/pmc_to_json.py "$OUT/pmc" "$WORKLOAD" > "profiles/${TAG}_counters_${NAME}.json"
cp "profiles/${TAG}_counters_${NAME}.json" "$OUT/"
python tools/pmc_summary.py "$OUT/pmc" > "$OUT/${TAG}_${NAME}_pmc.txt"
# the raw per-dispatch CSVs (tens of MB for the 4K config) have been reduced to the two files above: gpurun_out/ travels back only below 64 MiB
find "$OUT/pmc" -mindepth 1 -maxdepth 1 -type d -name 'pass*' -exec rm -rf {} +
(cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o trace -- python bench.py --steps 30 --warmup 3 --no-cpu-baseline "$@" > "$OUT/prof.log" 2>&1)
DB=$(find "$OUT/prof" -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py "$DB" > "$OUT/${TAG}_${NAME}_kernel_stats.txt"
rm -rf "$OUT/prof"
timeout 900 python bench.py "$@" > "$OUT/${TAG}_bench_${NAME}.json" 2> "$OUT/bench.err"
tail -c 300 "$OUT/bench.err"
python - <<PY
import json
d = json.load(open("$OUT/${TAG}_bench_${NAME}.json"))
print("$NAME", d["value"], d["unit"], d["ms_per_step"], "ms;", json.dumps(d["roofline"])[:900])
print("cpu", d["cpu_baseline"])
PY
