#!/bin/bash
# usage (GPU box): tools/sweep_env.sh <VAR> "<v1 v2 ..>" [bench args]   -- rounds/s and per-kernel ms of bench.py for values of one tuning knob
var=$1; vals=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in $vals; do
  env $var=$v python $R/bench.py --no-pmc --cpu-rounds 0 --sustain 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d.get('kernel_ms_per_round',{})
print('$var=$v', 'rounds/s %.1f' % d['value'], 'ms: root %.3f lambda %.3f nodes %s' % (k.get('hist_root',0), k.get('lambda',0), k.get('hist_nodes')))
"
done
