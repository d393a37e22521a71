bash tools/results_matrix.sh r06 '4k' > gpurun_out/r06_matrix_4k.log 2>&1
tail -12 gpurun_out/r06_matrix_4k.log
