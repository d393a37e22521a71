bash tools/results_matrix.sh r06 '4k' > gpurun_out/r06_matrix_4k.log 2>&1
tail -12 gpurun_out/r06_matrix_4k.log
timeout 600 python tools/rebuilt_layout_bound.py --out gpurun_out/r06_rebuilt_layout_bound.jsonl > gpurun_out/r06_rebuilt_layout_bound.log 2>&1
tail -5 gpurun_out/r06_rebuilt_layout_bound.log
