#!/bin/bash
# same-box A/B of round 5's lambda builds (on the GPU box, via gpurun): tools/ab_r05i.sh <out> <rounds> <shape> <reps> label:lib[:VAR=val] ...
#   lib = "-" (in-tree) or a name under ranklib_amd/lib/variants/
out=$1; rounds=${2:-40}; shape=${3:-c2}; reps=${4:-2}; shift 4
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in $(seq 1 $reps); do
  for spec in "$@"; do
    IFS=: read -r label lib kv <<< "$spec"
    envs=()
    [ "$lib" != "-" ] && envs+=("RLHIP_LIB=$R/ranklib_amd/lib/variants/$lib.so")
    [ -n "$kv" ] && envs+=("$kv")
    env "${envs[@]}" python $R/bench.py --plain --shape $shape --steps $rounds --warmup 5 2>/dev/null | grep '"metric"' | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-6s %-14s rep $rep %8.2f rounds/s  %7.4f ms' % ('$shape', '$label', d['value'], d['ms_per_step']))"
  done
done | tee -a $out
