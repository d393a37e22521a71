cd "$GRAFT_REPO_ROOT"
run() { lbl=$1; fif=$2; shift; shift
env "$@" python bench.py --scene atrium --no-cpu-baseline --no-companion --steps 100 --frames-in-flight $fif 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$lbl fif=$fif', d['value'], d['ms_per_step'], d['kernel_ms_per_frame_isolated']['trace_fused'], d['kernel_ms_per_frame_isolated']['shade'])"
}
run base 3 A=1
run base 2 A=1
run tb1536 3 ATEN_AMD_TRACE_BLOCKS=1536
run tb3072 3 ATEN_AMD_TRACE_BLOCKS=3072
run items2 3 ATEN_AMD_SHADE_ITEMS=2
run items3 3 ATEN_AMD_SHADE_ITEMS=3
run firstrefill 3 ATEN_AMD_FIRST_SIMPLE=0
run batches2 3 ATEN_AMD_BATCHES=2 ATEN_AMD_MIN_BATCH=100000
