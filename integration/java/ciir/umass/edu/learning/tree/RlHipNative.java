package ciir.umass.edu.learning.tree;

import java.nio.FloatBuffer;

/** static native methods bound by integration/jni/RlHipNative.c to librlhip.so (include/rlhip.h). */
final class RlHipNative {
    static { System.loadLibrary("rlhipjni"); }
    private RlHipNative() {}
    /** metric: 0 NDCG, 1 DCG, 2 MAP, 3 ERR (RL_METRIC_*); ranker: 6 LambdaMART, 0 MART (RL_RANKER_*); flags: RL_FLAG_* (16 = RL_FLAG_JAVA_ORDER) */
    static native long create(int nTrees, int nLeaves, int nThreshold, int minLeafSupport, int stopEarly, float lr, int metric, int k,
            int ranker, int device, float featureSamplingRate, long seed, int flags);
    static native void destroy(long h);
    /** ERRScorer.MAX (-gmax): process-wide, before create (rlhip.h rl_set_err_max) */
    static native int setErrMax(double maxGain);
    /** nodes of the largest possible tree, after init: the array length boostRound needs (rlhip.h rl_tree_capacity) */
    static native int treeCapacity(long h);
    static native int setData(long h, boolean validation, FloatBuffer X, long nDocs, int nFeatures, float[] labels, int[] qoff,
            int[] featureIds, int[] qkey);
    /** a block of rows after setData(.., X = null, ..): X holds nDocs * nFeatures floats from position 0 (rlhip.h rl_set_rows) */
    static native int setRows(long h, boolean validation, long firstDoc, long nDocs, FloatBuffer X);
    /** -qrel: per ranked list, its qid's idealGains entry (NaN = none) / relDocCount (0 = qid not in the file); either may be null (rlhip.h rl_set_external_judgments) */
    static native int setExternalJudgments(long h, boolean validation, double[] idealDcg, int[] relDocCount);
    static native int init(long h);
    static native int boostRound(long h, int[] feature, float[] threshold, int[] left, int[] right, float[] output, float[] metrics);
    static native double[] finish(long h);
    static native int numTrees(long h);
}
