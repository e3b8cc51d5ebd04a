#!/bin/bash
# usage (on the GPU box, via gpurun): tools/ab_libs.sh <out.txt> <rounds> <shape> lib1.so lib2.so ...   ("-" = the in-tree library)
# Same-box A/B of builds of librlhip.so (RLHIP_LIB): every library runs the plain timed region three times, interleaved.
out=$1; rounds=$2; shape=$3; shift 3
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset RLHIP_LIB; else export RLHIP_LIB=$R/$lib; fi
    python $R/bench.py --plain --shape $shape --steps $rounds --warmup 5 2>/dev/null | grep '"metric"' | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-28s rep $rep  %8.2f rounds/s  %7.4f ms' % ('$lib', d['value'], d['ms_per_step']))"
  done
done | tee $out
