mkdir -p gpurun_out/r04c
for v in clk_fw4 clk_fw2; do
RLHIP_LIB=$PWD/ranklib_amd/lib/variants/$v.so timeout 300 python tools/phase_clocks.py c2 300 > gpurun_out/r04c/phase_clocks_late_$v.txt 2>&1
done
