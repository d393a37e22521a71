for e in "X=1" "ATEN_AMD_BATCHES=2" "ATEN_AMD_BATCHES=3" "ATEN_AMD_TRACE_BLOCKS=1280" "ATEN_AMD_TRACE_BLOCKS=1536" "ATEN_AMD_TRACE_BLOCKS=2560" "ATEN_AMD_TRACE=s"; do
  env $e python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']; print('%-32s %8.3f ms  fused %.3f shade %.3f' % ('$e',d['ms_per_step'],k['trace_fused'],k['shade']))"
done
