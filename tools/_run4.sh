mkdir -p gpurun_out/r04e
timeout 600 python bench.py > gpurun_out/r04e/bench_default.json 2> gpurun_out/r04e/bench_default.err
tail -c 600 gpurun_out/r04e/bench_default.err
B="timeout 300 python bench.py --no-pmc --cpu-rounds 0 --c1-trees 0 --node-rounds 0 --ns-rounds 0"
for rep in 1 2; do
$B > gpurun_out/r04e/bench_fw5_$rep.json 2>/dev/null; python tools/bench_line.py fw5 < gpurun_out/r04e/bench_fw5_$rep.json
RLHIP_LIB=$PWD/ranklib_amd/lib/variants/fw4.so $B > gpurun_out/r04e/bench_fw4_$rep.json 2>/dev/null; python tools/bench_line.py fw4 < gpurun_out/r04e/bench_fw4_$rep.json
done
for sh in c1 c3; do
$B --shape $sh > gpurun_out/r04e/bench_${sh}_fw5.json 2>/dev/null; python tools/bench_line.py ${sh}_fw5 < gpurun_out/r04e/bench_${sh}_fw5.json
RLHIP_LIB=$PWD/ranklib_amd/lib/variants/fw4.so $B --shape $sh > gpurun_out/r04e/bench_${sh}_fw4.json 2>/dev/null; python tools/bench_line.py ${sh}_fw4 < gpurun_out/r04e/bench_${sh}_fw4.json
done
