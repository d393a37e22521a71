bash tools/results_matrix.sh r06 '1080p' > gpurun_out/r06_matrix_1080p.log 2>&1
tail -40 gpurun_out/r06_matrix_1080p.log
