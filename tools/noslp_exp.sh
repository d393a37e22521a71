#!/bin/bash
# r04: product build vs -fno-slp-vectorize (no packed-FP32 pairs), on the GPU box.  usage: tools/noslp_exp.sh <variant-name>
cd "$GRAFT_REPO_ROOT"
V=${1:-noslp}
so=$PWD/aten_amd/_variants/libaten_amd_$V.so
line() {  # name, env-prefix..., bench args
python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame_isolated']
print('%-10s %-22s %8.3f ms/frame (spread %.3f) latency %.3f  isolated: %s' % ('$1','$2',d['ms_per_step'],d.get('spread',0),d.get('ms_per_frame_latency',0),' '.join('%s %.3f'%(n,v) for n,v in k.items())))"
}
for lib in product $V; do
  if [ $lib = product ]; then unset ATEN_AMD_LIB; else export ATEN_AMD_LIB=$so; fi
  timeout 300 python bench.py --scene sponza --steps 50 --warmup 5 --no-cpu-baseline --no-companion 2>/dev/null | line $lib sponza
  timeout 300 python bench.py --scene atrium --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | line $lib atrium
  timeout 300 python bench.py --config c2 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | line $lib c2
  timeout 300 python bench.py --config c5 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | line $lib c5
  ATEN_AMD_SVGF_ATROUS4=0 timeout 300 python bench.py --config c5 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | line $lib c5_atrous1
done
export ATEN_AMD_LIB=$so
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
