MATRIX_PMC_ONLY=1 bash tools/results_matrix.sh r06 '1080p' > gpurun_out/r06_matrix_1080p_pmc.log 2>&1
bash tools/results_matrix.sh r06 '4k' > gpurun_out/r06_matrix_4k.log 2>&1
tail -40 gpurun_out/r06_matrix_4k.log
strip() { python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l); continue
    d['regen'].pop('closest_per_stage'); d['regen'].pop('shadow_per_stage'); print(json.dumps(d))
"; }
mkdir -p gpurun_out/r06_c4
timeout 600 python tools/regen_diag.py --scene atrium --width 3840 --height 2160 --spp 8 --depth 8 --burst 1 | strip >> gpurun_out/r06_c4/diag.jsonl 2>&1
timeout 600 python tools/regen_diag.py --scene atrium --width 3840 --height 2160 --spp 8 --depth 8 --burst 1 --all-samples | strip >> gpurun_out/r06_c4/diag.jsonl 2>&1
cat gpurun_out/r06_c4/diag.jsonl
