#!/bin/bash
# Runs tools/valu_calib.hip on the GPU box: once plain (wall times, HIP events), then under `rocprofv3 --pmc`, one pass
# per counter set (counters only -- no trace domains beside --pmc), and joins both into profiles/<tag>_calibration.json.
# usage: tools/calib_collect.sh <tag e.g. r03>
set -u
TAG=${1:-r03}
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${TAG}_calib
mkdir -p "$OUT" profiles
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o "$OUT/valu_calib" tools/valu_calib.hip || exit 1
"$OUT/valu_calib" > "$OUT/plain.jsonl" 2> "$OUT/plain.err"
cat "$OUT/plain.jsonl"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
declare -A SETS
SETS[1]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY"
SETS[2]="SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
SETS[3]="TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
SETS[4]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"
SETS[5]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
SETS[6]="FETCH_SIZE GRBM_GUI_ACTIVE"
SETS[7]="TCC_BUBBLE_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_sum"
for i in 1 2 3 4 5 6 7; do
  timeout 300 rocprofv3 --pmc ${SETS[$i]} --output-format csv -d "$OUT/pass$i" -o pmc -- "$OUT/valu_calib" > "$OUT/pass$i.log" 2>&1
  echo "pass $i rc=$? : ${SETS[$i]}"
done
python tools/calib_to_json.py "$OUT" > "profiles/${TAG}_calibration.json"
cp "profiles/${TAG}_calibration.json" "$OUT/"
python - <<PY
import json
d = json.load(open("profiles/${TAG}_calibration.json"))
print(json.dumps(d["ceilings"], indent=1))
PY
