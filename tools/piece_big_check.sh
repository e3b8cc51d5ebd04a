#!/bin/bash
# usage (GPU box): tools/piece_big_check.sh [docs] [features] [kind] [rounds] [ranks]
# k ranks as processes on the one GPU (host-callback transport over gloo, tests/dist_worker.py) at a size the -m gpu suite does not reach, against one GPU:
# trees, scores, metrics byte-identical; prints the distributed float chains' repair statistics (natural window misses at piece boundaries) and the exchange counters.
cd ${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-1200000}; F=${2:-136}; K=${3:-mslr}; R=${4:-12}; W=${5:-3}
python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port 29777 tests/dist_worker.py /tmp/pb.npz $N $F $K 3 31 $R LAMBDAMART NDCG 10 > /tmp/pb.log 2>&1 || tail -20 /tmp/pb.log
N=$N F=$F K=$K R=$R W=$W python - <<PY
import numpy as np, sys, os
sys.path.insert(0, "tests")
import torch; torch.cuda.init()
N, F, K, R, W = int(os.environ["N"]), int(os.environ["F"]), os.environ["K"], int(os.environ["R"]), int(os.environ["W"])
z = np.load("/tmp/pb.npz")
print("%d ranks, %d x %d (%s), %d rounds: piece stats (repair rounds, pieces re-evaluated) %s, exchange counters %s" % (W, N, F, K, R, z["piece_stats"], z["dist_stats"]))
from test_gpu_dist import single, same
ref = single(N, F, K, 3, 31, R)
trees = [{k: z["t%d_%s" % (i, k)] for k in ("feature", "threshold", "left", "right", "output", "deviance", "count")} for i in range(R)]
same(ref, (trees, [float(v) for v in z["mets"]], z["scores"], float(z["final"])))
print("identical to one GPU")
PY
