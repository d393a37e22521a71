for e in "X=1" "ATEN_AMD_TRACE=s" "ATEN_AMD_TRACE_BLOCKS=1024" "ATEN_AMD_TRACE_BLOCKS=1536" "ATEN_AMD_BATCHES=2" "ATEN_AMD_SHADE_ITEMS=2"; do
  env $e python bench.py --steps 30 --warmup 5 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame_isolated']; print('%-32s %8.3f ms  iso: fused %.3f shade %.3f' % ('$e',d['ms_per_step'],k['trace_fused'],k['shade']))"
done
