mkdir -p gpurun_out/r04h
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tie_break.py -x -q > gpurun_out/r04h/tests.log 2>&1; tail -3 gpurun_out/r04h/tests.log
B="timeout 300 python bench.py --no-pmc --cpu-rounds 0 --c1-trees 0 --node-rounds 0 --ns-rounds 0"
for rep in 1 2; do
$B > gpurun_out/r04h/b.json 2>/dev/null; python tools/bench_line.py c2_lds < gpurun_out/r04h/b.json
RLHIP_LIB=$PWD/ranklib_amd/lib/variants/nofb.so $B > gpurun_out/r04h/b.json 2>/dev/null; python tools/bench_line.py c2_before < gpurun_out/r04h/b.json
done
for sh in c3 c1 c0; do
$B --shape $sh > gpurun_out/r04h/b.json 2>/dev/null; python tools/bench_line.py ${sh}_lds < gpurun_out/r04h/b.json
RLHIP_LIB=$PWD/ranklib_amd/lib/variants/nofb.so $B --shape $sh > gpurun_out/r04h/b.json 2>/dev/null; python tools/bench_line.py ${sh}_before < gpurun_out/r04h/b.json
done
RLHIP_LIB=$PWD/ranklib_amd/lib/variants/clk.so timeout 300 python tools/phase_clocks.py c2 300 > gpurun_out/r04h/phase_clocks_late.txt 2>&1
RLHIP_LIB=$PWD/ranklib_amd/lib/variants/clk.so timeout 300 python tools/step_trace.py c2 300 > gpurun_out/r04h/trace_c2_t300.txt 2>&1
