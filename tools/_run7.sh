mkdir -p gpurun_out/r04g
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "tiled or ensemble or eval or model or predict" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -s -k "10k_trees" 2>&1 | tail -6
timeout 900 python bench.py --workload infer --docs 20000000 --cpu-rounds 0 > gpurun_out/r04g/bench_infer_20M.json 2> gpurun_out/r04g/bench_infer_20M.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04g/bench_infer_20M.json').read().strip().splitlines()[-1])
print('infer 20M docs:', d['value'], d['ms_per_step'], d['config']['node_visits_per_s'])
PY
