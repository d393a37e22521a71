#!/bin/bash
# Host-side knobs (environment variables read at context creation) on the product library: shade chunk items, persistent trace grid,
# plain-walk block size.  usage: tools/env_sweep.sh
cd "$GRAFT_REPO_ROOT"
run() {  # label, env...
  label=$1; shift
  for cfg in "--scene sponza --no-companion" "--scene atrium" "--config c2"; do
    env "$@" timeout 300 python bench.py $cfg --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame_isolated']
print('%-28s %-32s %8.3f ms/frame (spread %.3f) latency %.3f  isolated: %s' % ('$label','$cfg',d['ms_per_step'],d.get('spread',0),d.get('ms_per_frame_latency',0),' '.join('%s %.3f'%(n,v) for n,v in k.items() if n in ('trace_closest','shade','trace_fused'))))"
  done
}
run default ATN_NOTHING=1
run shade_items=2 ATEN_AMD_SHADE_ITEMS=2
run shade_items=3 ATEN_AMD_SHADE_ITEMS=3
run trace_blocks=1024 ATEN_AMD_TRACE_BLOCKS=1024
run trace_blocks=1280 ATEN_AMD_TRACE_BLOCKS=1280
run trace_blocks=1408 ATEN_AMD_TRACE_BLOCKS=1408
run simple_block=128 ATEN_AMD_SIMPLE_BLOCK=128
run simple_block=256 ATEN_AMD_SIMPLE_BLOCK=256
run default ATN_NOTHING=1
