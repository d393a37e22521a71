#!/bin/bash
# A/B of variant libraries on the two refill-walk scenes only (sponza_lod, atrium).  usage: [STEPS=100] tools/variants_sa.sh name ...
cd "$GRAFT_REPO_ROOT"
STEPS=${STEPS:-100}
for name in "$@"; do
  if [ "$name" = product ]; then unset ATEN_AMD_LIB; else export ATEN_AMD_LIB=$PWD/aten_amd/_variants/libaten_amd_$name.so; [ -f "$ATEN_AMD_LIB" ] || { echo "$name: no library"; continue; }; fi
  for cfg in "--scene sponza --no-companion" "--scene atrium"; do
    timeout 300 python bench.py $cfg --steps $STEPS --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame_isolated']
print('%-12s %-32s %8.3f ms/frame (spread %.3f) latency %.3f  isolated: %s' % ('$name','$cfg',d['ms_per_step'],d.get('spread',0),d.get('ms_per_frame_latency',0),' '.join('%s %.3f'%(n,v) for n,v in k.items() if n in ('trace_closest','shade','trace_fused'))))"
  done
done
