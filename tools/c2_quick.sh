#!/bin/bash
# Cornell-only A/B of variant libraries and ATEN_AMD_SIMPLE_BLOCK.  usage: tools/c2_quick.sh name ...
cd "$GRAFT_REPO_ROOT"
for name in "$@"; do
  if [ "$name" = product ]; then unset ATEN_AMD_LIB; else export ATEN_AMD_LIB=$PWD/aten_amd/_variants/libaten_amd_$name.so; fi
  for sb in 64 128 256; do
    ATEN_AMD_SIMPLE_BLOCK=$sb timeout 300 python bench.py --config c2 --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame_isolated']
print('%-10s block=%-4s %8.3f ms/frame (spread %.3f) latency %.3f  isolated: shade %.3f trace_fused %.3f' % ('$name','$sb',d['ms_per_step'],d.get('spread',0),d.get('ms_per_frame_latency',0),k['shade'],k['trace_fused']))"
  done
done
