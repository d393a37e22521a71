mkdir -p gpurun_out/r06_z
rm -f gpurun_out/parity/parity_report.jsonl
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r06_z/gpu_tests.log
cat gpurun_out/r06_z/gpu_tests.log
cp gpurun_out/parity/parity_report.jsonl gpurun_out/r06_z/parity_report.jsonl
bash tools/profile_all.sh r06_z > gpurun_out/r06_z/profile_all.log 2>&1
tail -30 gpurun_out/r06_z/profile_all.log
