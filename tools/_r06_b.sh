mkdir -p gpurun_out/r06_l
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06_l/gpu_tests.log
cat gpurun_out/r06_l/gpu_tests.log
timeout 900 python bench.py --steps 40 --warmup 5 --no-own-tree --no-cpu-baseline > gpurun_out/r06_l/bench_default.json 2> gpurun_out/r06_l/bench_default.err
tail -3 gpurun_out/r06_l/bench_default.err
python -c "
import json; d=json.loads(open('gpurun_out/r06_l/bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['ms_per_frame_latency'], d['kernel_ms_per_frame_isolated'], d['film_sha256']); print(d['config'].get('companion'))"
