mkdir -p gpurun_out/r04g
timeout 1200 python bench.py --workload infer > gpurun_out/r04g/bench_infer_c4_100M.json 2> gpurun_out/r04g/bench_infer_c4_100M.err
tail -c 400 gpurun_out/r04g/bench_infer_c4_100M.err
timeout 1200 tools/gpu_pmc_infer.sh r04_infer 4000000
