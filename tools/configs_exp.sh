#!/bin/bash
# All bench configurations with the library in the tree (or ATEN_AMD_LIB), one line each.  usage: tools/configs_exp.sh [tag] [steps]
cd "$GRAFT_REPO_ROOT"
TAG=${1:-product}; STEPS=${2:-50}
line() {
python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame_isolated']
print('%-10s %-12s %8.3f ms/frame (spread %.3f) latency %.3f  %.1f Mrays/s  isolated: %s' % ('$TAG','$1',d['ms_per_step'],d.get('spread',0),d.get('ms_per_frame_latency',0),d['value'],' '.join('%s %.3f'%(n,v) for n,v in k.items())))"
}
timeout 300 python bench.py --scene sponza --steps $STEPS --warmup 5 --no-cpu-baseline --no-companion 2>/dev/null | line sponza
timeout 300 python bench.py --scene atrium --steps $STEPS --warmup 5 --no-cpu-baseline 2>/dev/null | line atrium
timeout 300 python bench.py --config c2 --steps $STEPS --warmup 5 --no-cpu-baseline 2>/dev/null | line c2
timeout 300 python bench.py --config c5 --steps $STEPS --warmup 5 --no-cpu-baseline 2>/dev/null | line c5
timeout 600 python bench.py --config c4 --no-cpu-baseline 2>/dev/null | line c4
