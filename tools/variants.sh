#!/bin/bash
# Build kernel variants (extra -D flags) on the GPU box, bench each, and (PMC=1) count wave-level
# instructions of the trace kernel.   usage: [PMC=1] [SCENES="sponza cornell"] tools/variants.sh "name:flags" ...
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/variants
SCENES=${SCENES:-"sponza cornell"}
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  so=/tmp/libaten_amd_$name.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -I ${SRC:-.}/include $flags -o $so ${SRC:-.}/aten_amd/csrc/aten_amd.hip 2> gpurun_out/variants/$name.build.log || { echo "$name: BUILD FAILED"; tail -5 gpurun_out/variants/$name.build.log; continue; }
  for scene in $SCENES; do
    ATEN_AMD_LIB=$so timeout 300 python bench.py --scene $scene --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']; print('%-28s %-8s %8.3f ms  fused %.3f shade %.3f gen %.3f' % ('$name','$scene',d['ms_per_step'],k['trace_fused'],k['shade'],k['gen_path']))"
    if [ "${PMC:-0}" = "1" ]; then
      (cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && rm -rf /tmp/pmc_$name && ATEN_AMD_LIB=$so timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/pmc_$name -o pmc -- python bench.py --scene $scene --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1)
      python - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pmc_$name/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
for k in acc:
    if 'trace_fused' in k or 'k_shade' in k:
        c=acc[k]; m=lambda n: sum(c[n])/max(len(c[n]),1)/1e6
        print('    %-40s VALU %.1fM SALU %.1fM VMEM %.2fM waves %d wavecyc %.0fM wait %.0fM valu_active %.0fM' % (k[-40:], m('SQ_INSTS_VALU'), m('SQ_INSTS_SALU'), m('SQ_INSTS_VMEM_RD'), m('SQ_WAVES')*1e6, m('SQ_WAVE_CYCLES'), m('SQ_WAIT_ANY'), m('SQ_ACTIVE_INST_VALU')))
PY
    fi
  done
done
