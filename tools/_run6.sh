timeout 300 tools/gpu_profile.sh r04f_c3 --shape c3 --steps 20 --warmup 5 --plain
RLHIP_LIB=$PWD/ranklib_amd/lib/variants/clk.so timeout 300 python tools/step_trace.py c3 20 > gpurun_out/r04f/trace_c3_t20.txt 2>&1
