mkdir -p gpurun_out/r06_b
timeout 1500 python -m pytest tests/test_gpu_regen.py -x -q 2>&1 | tail -15 > gpurun_out/r06_b/regen_tests.log
cat gpurun_out/r06_b/regen_tests.log
for sh in 1 8; do
  timeout 300 python tools/regen_diag.py --shard $sh --burst 8 >> gpurun_out/r06_b/diag.jsonl 2>&1
done
timeout 300 python tools/regen_diag.py --scene cornell --shard 1 --burst 8 >> gpurun_out/r06_b/diag.jsonl 2>&1
timeout 600 python tools/regen_diag.py --scene atrium --width 3840 --height 2160 --spp 8 --depth 8 --burst 1 >> gpurun_out/r06_b/diag.jsonl 2>&1
timeout 600 python tools/regen_diag.py --scene atrium --width 3840 --height 2160 --spp 8 --depth 8 --burst 1 --all-samples >> gpurun_out/r06_b/diag.jsonl 2>&1
cat gpurun_out/r06_b/diag.jsonl
