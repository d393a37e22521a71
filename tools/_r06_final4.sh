timeout 1200 python -m pytest tests/test_gpu_anyhit_twin.py tests/test_gpu_lbvh.py tests/test_gpu_cxx_app.py tests/test_gpu_mgpu.py -m gpu -x -q 2>&1 | tail -8
timeout 900 python tools/rebuilt_layout_bound.py --out gpurun_out/r06_rebuilt_layout_bound.jsonl > gpurun_out/r06_rebuilt_layout_bound.log 2>&1
tail -3 gpurun_out/r06_rebuilt_layout_bound.log
