#!/bin/bash
# On the GPU box: bench every aten_amd/_variants/libaten_amd_<name>.so named on the command line (built by
# tools/build_variants.sh).  usage: [SCENES="sponza atrium"] [PMC=1] tools/bench_variants.sh name ...
cd "$GRAFT_REPO_ROOT"
SCENES=${SCENES:-"sponza atrium"}
for name in "$@"; do
  so=$PWD/aten_amd/_variants/libaten_amd_$name.so
  [ -f "$so" ] || { echo "$name: no library"; continue; }
  for scene in $SCENES; do
    ATEN_AMD_LIB=$so timeout 300 python bench.py --scene $scene --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame_isolated']
print('%-24s %-8s %8.3f ms/frame  isolated: fused %.3f shade %.3f' % ('$name','$scene',d['ms_per_step'],k['trace_fused'],k['shade']))"
    if [ "${PMC:-0}" = "1" ]; then
      for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE"; do
        tag=$(echo $set | cut -c1-6 | tr -d ' ')
        (cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && rm -rf /tmp/pmc_${name}_$tag && ATEN_AMD_LIB=$so timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_${name}_$tag -o pmc -- python bench.py --scene $scene --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1)
      done
      python - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pmc_${name}_*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    if 'trace_fused<true' in k:
        c=acc[k]; m=lambda n: sum(c[n])/max(len(c[n]),1)/1e6
        cyc=m('GRBM_GUI_ACTIVE')/8
        print('    %-30s cycles %.3fM VALU %.1fM SALU %.1fM VMEM %.2fM | wave-cycles: wait %.0f%% stall %.0f%% issue %.0f%% | TCP lane %.0fM (%.2f/CU/clk) cache %.0fM (%.2f/CU/clk) L2 req %.1fM l1stall %.2f lanes %.3f' % (
            k[-30:], cyc, m('SQ_INSTS_VALU'), m('SQ_INSTS_SALU'), m('SQ_INSTS_VMEM_RD'),
            100*m('SQ_WAIT_ANY')/m('SQ_WAVE_CYCLES'), 100*m('SQ_WAIT_INST_ANY')/m('SQ_WAVE_CYCLES'), 100*m('SQ_ACTIVE_INST_ANY')/m('SQ_WAVE_CYCLES'),
            m('TCP_TOTAL_ACCESSES_sum'), m('TCP_TOTAL_ACCESSES_sum')/256/cyc, m('TCP_TOTAL_CACHE_ACCESSES_sum'), m('TCP_TOTAL_CACHE_ACCESSES_sum')/256/cyc,
            m('TCP_TCC_READ_REQ_sum'), m('TCP_PENDING_STALL_CYCLES_sum')/256/cyc, m('SQ_THREAD_CYCLES_VALU')/max(m('SQ_ACTIVE_INST_VALU')*64, 1e-9)))
PY
    fi
  done
done
