#!/bin/bash
# PMC counters of the serial loop's kernels next to the regenerated pool's (tools/regen_pmc_driver.py), separate passes.
# usage: tools/regen_pmc.sh <outdir>
set -u
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
declare -A SETS
SETS[1]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY"
SETS[2]="SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
SETS[3]="TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
SETS[5]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
for i in 1 2 3 5; do
  PMC=${SETS[$i]}
  timeout 300 rocprofv3 --pmc $PMC --output-format csv -d "$OUT/pass$i" -o pmc -- python tools/regen_pmc_driver.py both 8 > "$OUT/pass$i.log" 2>&1
  echo "pass $i rc=$? : $PMC"
done
python tools/pmc_summary.py "$OUT" > "$OUT/summary.txt"
