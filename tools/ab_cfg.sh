#!/bin/bash
# usage (on the GPU box, via gpurun): tools/ab_cfg.sh <out.txt> <rounds> <shape> "VAR=v,VAR2=w" "VAR=x" ...   ("-" = no variables)
# Same-box A/B of run-time knob SETS: every configuration runs the plain timed region three times, interleaved.
out=$1; rounds=$2; shape=$3; shift 3
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
  for cfg in "$@"; do
    envs=""; [ "$cfg" != "-" ] && envs=$(echo "$cfg" | tr ',' ' ')
    env $envs python $R/bench.py --plain --shape $shape --steps $rounds --warmup 5 2>/dev/null | grep '"metric"' | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-5s %-52s rep $rep  %8.2f rounds/s  %7.4f ms' % ('$shape', '$cfg', d['value'], d['ms_per_step']))"
  done
done | tee -a $out
