#!/bin/bash
# Instruction-cache counters per kernel (counters only: rocprofv3 --pmc, dispatches serialised).  usage: tools/icache_exp.sh [bench args]
cd "$GRAFT_REPO_ROOT"
OUT=/tmp/icache_pmc; rm -rf $OUT
(cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && timeout 600 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-companion "$@" > /tmp/icache.log 2>&1)
python - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    c=acc[k]; m=lambda n: sum(c[n])/max(len(c[n]),1)
    if m('SQC_ICACHE_REQ') < 1e4: continue
    print('%-44s n=%3d  icache req %8.2fM hit %.4f miss %8.3fM dup %8.3fM  ifetch %8.2fM  wait_inst/wave_cycles %.3f  cycles %.3fM' % (
        k[-44:], len(c['SQC_ICACHE_REQ']), m('SQC_ICACHE_REQ')/1e6, m('SQC_ICACHE_HITS')/max(m('SQC_ICACHE_REQ'),1), m('SQC_ICACHE_MISSES')/1e6,
        m('SQC_ICACHE_MISSES_DUPLICATE')/1e6, m('SQ_IFETCH')/1e6, m('SQ_WAIT_INST_ANY')/max(m('SQ_WAVE_CYCLES'),1), m('GRBM_GUI_ACTIVE')/8e6))
PY
tail -3 /tmp/icache.log
