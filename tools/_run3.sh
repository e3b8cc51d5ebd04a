mkdir -p gpurun_out/r04d
B="timeout 300 python bench.py --no-pmc --cpu-rounds 0 --c1-trees 0 --node-rounds 0"
for v in 16 8 4; do RLHIP_SUB_CHILD=$v $B > gpurun_out/r04d/bench_sub$v.json 2>/dev/null; python tools/bench_line.py sub$v < gpurun_out/r04d/bench_sub$v.json; done
for v in 512 2048; do RLHIP_HIST_GRID=$v $B > gpurun_out/r04d/bench_grid$v.json 2>/dev/null; python tools/bench_line.py grid$v < gpurun_out/r04d/bench_grid$v.json; done
RLHIP_HIST_NT=1024 $B > gpurun_out/r04d/bench_nt1024.json 2>/dev/null; python tools/bench_line.py nt1024 < gpurun_out/r04d/bench_nt1024.json
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
