#!/bin/bash
# On the GPU box: ray cells off / on (ATEN_AMD_CELLS, read when the context is created) on the workloads whose tree does not fit
# an XCD's L2 -- frame time, isolated kernel times, and the L2 / fabric counters of the fused trace launch.
# usage: tools/cells_exp.sh [scene ...]    (default: atrium)
cd "$GRAFT_REPO_ROOT"
SCENES=${*:-atrium}
for scene in $SCENES; do
  for cells in 0 1; do
    ATEN_AMD_CELLS=$cells timeout 600 python bench.py --scene $scene --steps 30 --warmup 5 --no-cpu-baseline --no-companion 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_frame_isolated']
print('cells=$cells %-8s %8.3f ms/frame (spread %.3f) latency %.3f  isolated: fused %.3f closest %.3f shade %.3f' % ('$scene',d['ms_per_step'],d['spread'],d['ms_per_frame_latency'],k['trace_fused'],k.get('trace_closest',0),k['shade']))"
    out=/tmp/pmc_cells_${scene}_$cells
    rm -rf $out
    ATEN_AMD_CELLS=$cells tools/pmc_collect.sh $out "3 5 6" --scene $scene --no-companion > /dev/null 2>&1
    python - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$out/pass*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'].split('(')[0].replace('void ','').replace('atn::','')][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    if k.startswith('k_trace_fused') or k.startswith('k_shade'):
        c=acc[k]; m=lambda n: (sum(c[n])/max(len(c[n]),1)) if n in c else float('nan')
        hit=m('TCC_HIT_sum'); miss=m('TCC_MISS_sum')
        print('    %-44s n=%3d  TCC req %7.2fM hit %.3f  EA0_RDREQ %7.3fM  TCP->TCC rd %7.2fM  FETCH_SIZE %8.0f KB  l1 stall cyc %7.1fM' % (
            k[:44], len(c.get('TCC_REQ_sum',[])), m('TCC_REQ_sum')/1e6, hit/max(hit+miss,1), m('TCC_EA0_RDREQ_sum')/1e6, m('TCP_TCC_READ_REQ_sum')/1e6, m('FETCH_SIZE'), m('TCP_PENDING_STALL_CYCLES_sum')/1e6))
PY
  done
done
