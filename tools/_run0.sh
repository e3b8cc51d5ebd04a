set -x
mkdir -p gpurun_out/r04a
python bench.py --no-pmc --cpu-rounds 0 --c1-trees 0 > gpurun_out/r04a/bench_c2.json 2> gpurun_out/r04a/bench_c2.err
tail -c 3000 gpurun_out/r04a/bench_c2.json
RLHIP_LIB=$PWD/ranklib_amd/lib/variants/clk.so python tools/phase_clocks.py c2 25 > gpurun_out/r04a/phase_clocks_early.txt 2>&1
RLHIP_LIB=$PWD/ranklib_amd/lib/variants/clk.so python tools/phase_clocks.py c2 300 > gpurun_out/r04a/phase_clocks_late.txt 2>&1
tools/gpu_profile.sh r04a_c2 --steps 20 --warmup 5 --plain
tools/gpu_profile.sh r04a_c2_late --steps 20 --warmup 300 --plain
