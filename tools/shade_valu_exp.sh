#!/bin/bash
# k_shade's instruction counts per launch with ingredients removed (bench.py --experiment notex / noibl; sponza_lod).  Counters only.
cd "$GRAFT_REPO_ROOT"
for ex in "" notex noibl "notex,noibl"; do
  OUT=/tmp/shade_pmc; rm -rf $OUT
  (cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU_TRANS_F32 SQ_WAVES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-companion ${ex:+--experiment $ex} > /tmp/shade_pmc.log 2>&1)
  python - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    if 'k_shade' not in k: continue
    c=acc[k]; m=lambda n: sum(c[n])/max(len(c[n]),1)
    print('%-14s %-34s n=%3d  VALU %7.2fM (per wave %6.0f) SALU %6.2fM VMEM %5.2fM TRANS %5.2fM  lanes %.3f  cycles %.3fM' % (
        '${ex:-full}', k[-34:], len(c['SQ_INSTS_VALU']), m('SQ_INSTS_VALU')/1e6, m('SQ_INSTS_VALU')/max(m('SQ_WAVES'),1), m('SQ_INSTS_SALU')/1e6, m('SQ_INSTS_VMEM_RD')/1e6,
        m('SQ_INSTS_VALU_TRANS_F32')/1e6, m('SQ_THREAD_CYCLES_VALU')/max(m('SQ_ACTIVE_INST_VALU')*64,1e-9), m('GRBM_GUI_ACTIVE')/8e6))
PY
done
