cd "$GRAFT_REPO_ROOT"
for p in 1 0; do
ATEN_AMD_PROBE_STREAMS=$p python bench.py --config c2 --no-cpu-baseline --steps 200 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('probe=$p c2', d['value'], d['ms_per_step'], d['ms_per_frame_latency'])"
ATEN_AMD_PROBE_STREAMS=$p python tools/shard_curve.py --scene cornell --steps 100 --frames-in-flight 1 2>/dev/null | tail -1
done
time python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
