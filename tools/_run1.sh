set -x
mkdir -p gpurun_out/r04b
B="timeout 300 python bench.py --no-pmc --cpu-rounds 0 --c1-trees 0 --node-rounds 0"
for nt in 512 1024; do RLHIP_HIST_NT=$nt $B > gpurun_out/r04b/bench_nt$nt.json 2>/dev/null; python tools/bench_line.py nt$nt < gpurun_out/r04b/bench_nt$nt.json; done
RLHIP_LIB=$PWD/ranklib_amd/lib/variants/fw4.so RLHIP_HIST_NT=1024 $B > gpurun_out/r04b/bench_fw4_nt1024.json 2>/dev/null; python tools/bench_line.py fw4 < gpurun_out/r04b/bench_fw4_nt1024.json
RLHIP_HIST_NT=256 timeout 300 tools/gpu_profile.sh r04b_c2 --steps 20 --warmup 5 --plain
RLHIP_HIST_NT=256 timeout 300 tools/gpu_profile.sh r04b_c2_late --steps 20 --warmup 300 --plain
RLHIP_HIST_NT=1024 timeout 300 tools/gpu_profile.sh r04b_c2_late_nt1024 --steps 20 --warmup 300 --plain
