#!/bin/bash
# Shard curves (tools/shard_curve.py: rank 0's share of an N-way split on one GPU) over the walk flavour and k_shade's waves per SIMD.
cd "$GRAFT_REPO_ROOT"
for scene in sponza atrium cornell; do
  for fif in 3 4; do
    for env in "" "ATEN_AMD_TRACE=s" "ATEN_AMD_TRACE=r" "ATEN_AMD_SHADE_WAVES=4" "ATEN_AMD_SHADE_WAVES=5"; do
      printf "%-8s fif=%d %-24s " $scene $fif "${env:-default}"
      env $env timeout 300 python tools/shard_curve.py --scene $scene --steps 100 --frames-in-flight $fif 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); m=d['ms_per_frame_rank0_of_N']
print('  '.join('N=%s %.3f' % (k, v) for k, v in m.items()))"
    done
  done
done
