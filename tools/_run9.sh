mkdir -p gpurun_out/r04h
B="timeout 300 python bench.py --no-pmc --cpu-rounds 0 --c1-trees 0 --node-rounds 0 --ns-rounds 0"
for rep in 1 2; do
$B > gpurun_out/r04h/b.json 2>/dev/null; python tools/bench_line.py default_fw4 < gpurun_out/r04h/b.json
for v in nofb fw3 fw5; do RLHIP_LIB=$PWD/ranklib_amd/lib/variants/$v.so $B > gpurun_out/r04h/b.json 2>/dev/null; python tools/bench_line.py $v < gpurun_out/r04h/b.json; done
done
