mkdir -p gpurun_out/r04f
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "sparse or compact or mixed" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "c3" 2>&1 | tail -5
B="timeout 300 python bench.py --no-pmc --cpu-rounds 0 --c1-trees 0 --ns-rounds 0 --shape c3"
RLHIP_CROWS=0 $B > gpurun_out/r04f/bench_c3_rows.json 2>/dev/null; python tools/bench_line.py c3_rows < gpurun_out/r04f/bench_c3_rows.json
$B > gpurun_out/r04f/bench_c3_crows.json 2>/dev/null; python tools/bench_line.py c3_crows < gpurun_out/r04f/bench_c3_crows.json
RLHIP_CROWS=0 $B > gpurun_out/r04f/bench_c3_rows2.json 2>/dev/null; python tools/bench_line.py c3_rows < gpurun_out/r04f/bench_c3_rows2.json
$B > gpurun_out/r04f/bench_c3_crows2.json 2>/dev/null; python tools/bench_line.py c3_crows < gpurun_out/r04f/bench_c3_crows2.json
