mkdir -p gpurun_out/r05_dir
timeout 900 python -m pytest tests/test_gpu_direction_lists.py -x -q > gpurun_out/r05_dir/pytest.log 2>&1; tail -15 gpurun_out/r05_dir/pytest.log
for ax in 0 1 5 7; do
  timeout 300 python bench.py --scene sponza_own_tree --direction-axes $ax --no-cpu-baseline --no-companion --no-own-tree > gpurun_out/r05_dir/own_$ax.json 2> gpurun_out/r05_dir/own_$ax.err; tail -1 gpurun_out/r05_dir/own_$ax.err
  timeout 300 python bench.py --scene atrium --direction-axes $ax --no-cpu-baseline --no-companion --no-own-tree > gpurun_out/r05_dir/atrium_$ax.json 2> gpurun_out/r05_dir/atrium_$ax.err; tail -1 gpurun_out/r05_dir/atrium_$ax.err
done
