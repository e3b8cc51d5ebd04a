cd $GRAFT_REPO_ROOT
python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29777 tests/dist_worker.py /tmp/pb.npz 1200000 136 mslr 3 31 12 LAMBDAMART NDCG 10 > /tmp/pb.log 2>&1; tail -3 /tmp/pb.log
python - <<PY
import numpy as np, sys
sys.path.insert(0, "tests")
import torch; torch.cuda.init()
z = np.load("/tmp/pb.npz")
print("3 ranks, 1.2 M x 136, 12 rounds: piece stats (repair rounds, pieces re-evaluated)", z["piece_stats"], "dist stats", z["dist_stats"])
from test_gpu_dist import single, same
ref = single(1200000, 136, "mslr", 3, 31, 12)
trees = [{k: z["t%d_%s" % (i, k)] for k in ("feature", "threshold", "left", "right", "output", "deviance", "count")} for i in range(12)]
same(ref, (trees, [float(v) for v in z["mets"]], z["scores"], float(z["final"])))
print("identical to one GPU")
PY
