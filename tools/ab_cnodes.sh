# experiment driver (GPU box): ambiguous-step statistics of the compressed inner records + PMC A/B
cd "$GRAFT_REPO_ROOT"
for name in cn_stat1 cn_stat2; do
  for scene in sponza atrium; do
    ATEN_AMD_LIB=$PWD/aten_amd/_variants/libaten_amd_$name.so timeout 300 python bench.py --scene $scene --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$name $scene', d['work_per_frame'])"
  done
done
ATEN_AMD_CNODES=0 PMC=1 bash tools/bench_variants.sh cn
ATEN_AMD_CNODES=1 PMC=1 bash tools/bench_variants.sh cn
