#!/bin/bash
# Shard curves (tools/shard_curve.py: rank 0's share of an N-way split on one GPU) over the walk flavour and k_shade's waves per SIMD.
# usage: [SCENES="sponza atrium cornell"] [FIFS="3 4"] [ENVS="default ATEN_AMD_TRACE=s ..."] tools/shard_matrix.sh
cd "$GRAFT_REPO_ROOT"
SCENES=${SCENES:-"sponza atrium cornell"}; FIFS=${FIFS:-"3 4"}
ENVS=${ENVS:-"default ATEN_AMD_TRACE=s ATEN_AMD_TRACE=r ATEN_AMD_SHADE_WAVES=4 ATEN_AMD_SHADE_WAVES=5"}
for scene in $SCENES; do
  for fif in $FIFS; do
    for env in $ENVS; do
      printf "%-8s fif=%d %-24s " $scene $fif "$env"
      e=$env; [ "$e" = default ] && e="ATN_NOTHING=1"
      env $e timeout 300 python tools/shard_curve.py --scene $scene --steps 100 --frames-in-flight $fif 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); m=d['ms_per_frame_rank0_of_N']
print('  '.join('N=%s %.3f' % (k, v) for k, v in m.items()))"
    done
  done
done
