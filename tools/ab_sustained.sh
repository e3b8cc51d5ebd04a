#!/bin/bash
# usage (on the GPU box, via gpurun): tools/ab_sustained.sh <out.txt> <rounds> <shape> lib1.so lib2.so ...   ("-" = the in-tree library; VAR=value entries set an environment variable instead)
# Same-box A/B of the LATE rounds: tools/long_run.py prints rounds/s per 100 rounds (late trees are chain-like: 18-26 growth steps instead of 10).
out=$1; rounds=$2; shape=$3; shift 3
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for lib in "$@"; do
    unset RLHIP_LIB; envs=""
    if [[ "$lib" == *=* ]]; then envs="$lib"; elif [ "$lib" != "-" ]; then export RLHIP_LIB=$R/$lib; fi
    env $envs python $R/tools/long_run.py $shape $rounds 2>/dev/null | grep '^round' | awk -v l="$lib" -v r=$rep '{printf "%-34s rep %s  round %5s  %8s rounds/s\n", l, r, $2, $3}'
  done
done | tee $out
