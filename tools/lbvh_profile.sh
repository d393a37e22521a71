#!/bin/bash
# LBVH rebuild measurement set, on the GPU box: tools/lbvh_bench.py lines + rocprofv3 --kernel-trace --stats per size.
# usage: tools/lbvh_profile.sh <tag e.g. r02_d>
set -u
TAG=$1
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
timeout 600 python tools/lbvh_bench.py 12852 100000 1000000 4000000 > "$OUT/${TAG}_lbvh_rebuild.jsonl" 2> "$OUT/lbvh.err"
cat "$OUT/${TAG}_lbvh_rebuild.jsonl"
for N in 12852 1000000; do
    (cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$N" -o trace -- python tools/lbvh_bench.py $N > "$OUT/prof_$N.log" 2>&1)
    DB=$(find "$OUT/prof_$N" -name "*.db" | head -1)
    [ -n "$DB" ] && python tools/rocprof_summary.py "$DB" > "$OUT/${TAG}_lbvh_${N}_kernel_stats.txt"
    rm -rf "$OUT/prof_$N"
done
