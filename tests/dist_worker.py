"""Worker of tests/test_gpu_dist.py: one rank of a k-rank sharded training run on ONE GPU.
Launched with torch.distributed.run; transport = host callbacks over gloo."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist
    from ranklib_amd import _native as N
    from ranklib_amd import dist as D
    from ranklib_amd import synth

    out_path, n_docs, n_feat, kind, seed, leaves, rounds = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
    ranker, metric, k = (sys.argv[8], sys.argv[9], int(sys.argv[10])) if len(sys.argv) > 10 else ("LAMBDAMART", "NDCG", 10)
    # "valid": sharded validation set + early stopping; "rccl": RCCL transport; "noa2a": a transport without an all-to-all (emulated with all-gathers);
    # "tcm1": -tc -1 (every distinct value a threshold: the continuous columns get tables of more than 4095 entries, split into virtual features);
    # "leafm1": -leaf -1 with min leaf support 40; "qrel": external judgments (a third of the lists get an ideal DCG of their own times 1.5)
    opts = sys.argv[11].split(",") if len(sys.argv) > 11 else []
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    X, lab, qoff = synth.make_dataset(n_docs, n_feat, kind, seed_offset=seed)
    if "dupcols" in opts:    # the second half of the columns repeats the first: ties over several features that share one cut
        X = X.copy(); h = n_feat // 2; X[:, h:2 * h] = 2.0 * X[:, :h] + 1.0
    Xs, ls, qs = D.shard(X, lab, qoff, rank, world)
    tr = D.TorchHostTransport()
    estop = 1 if "valid" in opts else 100
    # "perdev": one rank per GPU (a multi-GPU box: the real RCCL data path over xGMI); otherwise every rank shares device 0
    device = int(os.environ.get("LOCAL_RANK", "0")) if "perdev" in opts else 0
    if "perdev" in opts:
        torch.cuda.set_device(device)
    g = N.Trainer(n_trees=rounds, n_leaves=-1 if "leafm1" in opts else leaves, ranker=ranker, metric=metric, metric_k=k, early_stop_rounds=estop,
                  min_leaf_support=40 if "leafm1" in opts else 1, device=device, n_threshold=-1 if "tcm1" in opts else 256)
    g.set_train(Xs, ls, qs)
    if "qrel" in opts:       # -qrel: every rank passes the judgments of ITS lists (tests/test_gpu_dist.py external_judgments is the same rule)
        qb, qe = D.partition_queries(qoff, world)[rank]
        qi = np.arange(qb, qe)
        g.set_external_judgments(False, ideal_dcg=np.where(qi % 3 == 0, 10.0 + (qi % 7), np.nan), rel_doc_count=(qi % 4).astype(np.int32))
    if "valid" in opts:
        Xv, lv, qv = synth.make_dataset(n_docs // 3, n_feat, kind, seed_offset=seed + 77)
        lv = lv[::-1].copy()                  # labels unrelated to the features: the validation metric wanders and the early stop fires
        g.set_validation(*D.shard(Xv, lv, qv, rank, world))
    if "rccl" in opts:
        # RCCL transport with every rank on the SAME device (the only GPU of the test box).  RCCL may refuse that ("Duplicate GPU
        # detected"): the test then only proves that ncclCommInitRank was reached with N ranks and failed cleanly.
        box = [g.dist_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        try:
            g.dist_init(box[0], rank, world)
        except N.RankLibError as ex:
            if rank == 0:
                np.savez(out_path, refused=str(ex))
            dist.barrier()
            dist.destroy_process_group()
            return
    else:
        g.dist_init_callback(rank, world, tr.allreduce, tr.allgather, None if "noa2a" in opts else tr.alltoallv)
    g.init()
    trees, mets, vmets = [], [], []
    for _ in range(rounds):
        t, tm, vm, stop = g.boost_round()
        trees.append(t.trimmed()); mets.append(float(tm)); vmets.append(float(vm) if vm is not None else 0.0)
        if stop:
            break
    final, vfinal = g.finish()
    sc = g.array("SCORE")
    parts = [None] * world
    dist.all_gather_object(parts, sc)
    stats = g.array("CHAIN_STATS")
    if rank == 0:
        np.savez(out_path, scores=np.concatenate(parts), mets=np.array(mets), vmets=np.array(vmets), final=final, vfinal=(vfinal or 0.0), kept=g.num_trees(),
                 dist_stats=g.dist_stats(), stats=stats, tie_stats=g.array("TIE_STATS"), piece_stats=g.array("PIECE_STATS"),
                 **{"t%d_%s" % (i, k): v for i, t in enumerate(trees) for k, v in t.items()})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
