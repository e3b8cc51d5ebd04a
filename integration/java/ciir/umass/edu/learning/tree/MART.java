package ciir.umass.edu.learning.tree;

import java.util.List;

import ciir.umass.edu.learning.RankList;
import ciir.umass.edu.learning.Ranker;
import ciir.umass.edu.metric.MetricScorer;

/**
 * Drop-in for ciir.umass.edu.learning.tree.MART (-ranker 0) on top of the rlhip LambdaMART drop-in: the reference's MART
 * only overrides computePseudoResponses and updateTreeOutput (learning/tree/MART.java:47-65); here both live behind
 * rl_params.ranker = RL_RANKER_MART in librlhip.so.  Written against RankLib 2.10.x, not compiled in the build image
 * (no JDK); the same switch is exercised through the ctypes mirror (ranklib_amd.learning.MART).
 */
public class MART extends LambdaMART {
    public MART() {
    }

    public MART(final List<RankList> samples, final int[] features, final MetricScorer scorer) {
        super(samples, features, scorer);
    }

    @Override
    public Ranker createNew() {
        return new MART();
    }

    @Override
    public String name() {
        return "MART";
    }

    @Override
    protected int rankerId() {
        return 0;
    }
}
