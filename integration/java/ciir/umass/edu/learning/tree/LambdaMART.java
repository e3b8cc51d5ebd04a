package ciir.umass.edu.learning.tree;

import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.nio.FloatBuffer;
import java.util.HashMap;
import java.util.List;
import java.util.Map;
import java.util.logging.Logger;

import ciir.umass.edu.learning.DataPoint;
import ciir.umass.edu.learning.RankList;
import ciir.umass.edu.learning.Ranker;
import ciir.umass.edu.metric.APScorer;
import ciir.umass.edu.metric.ERRScorer;
import ciir.umass.edu.metric.MetricScorer;
import ciir.umass.edu.metric.NDCGScorer;
import ciir.umass.edu.parsing.ModelLineProducer;
import ciir.umass.edu.utilities.MergeSorter;
import ciir.umass.edu.utilities.RankLibError;
import ciir.umass.edu.utilities.SimpleMath;

/**
 * Drop-in for RankLib's ciir.umass.edu.learning.tree.LambdaMART (same FQCN, placed ahead of RankLib's jar on the
 * classpath or compiled into it): RankerFactory's prototype table then hands THIS class to "-ranker 6".
 * The boosting loop runs on an MI355X through librlhip.so; the model is rebuilt from RankLib's own public
 * Split / RegressionTree / Ensemble classes, so eval(), model(), save() and loadFromString() behave exactly as
 * before and model files stay interchangeable in both directions.
 *
 * NOT COMPILED IN THIS REPOSITORY (no JDK in the build image).  Written against RankLib 2.10.x sources.
 */
public class LambdaMART extends Ranker {
    private static final Logger logger = Logger.getLogger(LambdaMART.class.getName());

    // same names, types and defaults as the reference (learning/tree/LambdaMART.java:37-42)
    public static int nTrees = 1000;
    public static float learningRate = 0.1F;
    public static int nThreshold = 256;
    public static int nRoundToStopEarly = 100;
    public static int nTreeLeaves = 10;
    public static int minLeafSupport = 1;
    public static int device = 0;
    /** RL_FLAG_JAVA_ORDER (rlhip.h): split gains from the f64 histogram the reference itself would hold, so that stored (feature,
     *  threshold) pairs equal the reference's even where exact arithmetic ties.  Several times slower; one GPU only. */
    public static boolean javaOrder = Boolean.getBoolean("rlhip.javaOrder");
    /** Feature sampling (FeatureHistogram.samplingRate < 1, set by RFRanker.init) draws through a seeded hash instead of the
     *  reference's unseeded java.util.Random: every trainer takes the next seed of this sequence (rlhip.h rl_params.seed). */
    public static long seed = System.nanoTime();

    protected Ensemble ensemble = null;
    protected double[] impacts = null;          // read by RFRanker (learning/tree/RFRanker.java:88-91); never written, as in the reference
    private long handle = 0;

    public LambdaMART() {}

    public LambdaMART(final List<RankList> samples, final int[] features, final MetricScorer scorer) {
        super(samples, features, scorer);
    }

    /** rows per rl_set_rows block: the direct buffer stays far below Integer.MAX_VALUE bytes whatever the data set's size */
    private static final int UPLOAD_BLOCK_BYTES = 64 << 20;

    private void upload(final List<RankList> lists, final boolean validation, final Map<String, Integer> qids) {
        long n = 0;
        for (final RankList rl : lists) n += rl.size();
        if (n > Integer.MAX_VALUE - 4096) throw RankLibError.create("rlhip: more than 2^31 documents in one data set");
        final float[] labels = new float[(int) n];
        final int[] qoff = new int[lists.size() + 1];
        final int[] qkey = new int[lists.size()];
        int k = 0;
        for (int q = 0; q < lists.size(); q++) {
            final RankList rl = lists.get(q);
            Integer key = qids.get(rl.getID());                  // equal qid strings share the idealGains cache entry
            if (key == null) { key = qids.size(); qids.put(rl.getID(), key); }
            qkey[q] = key;
            for (int j = 0; j < rl.size(); j++, k++) labels[k] = rl.get(j).getLabel();
            qoff[q + 1] = k;
        }
        // labels / offsets first (X = null), then the rows in blocks through ONE reusable direct buffer (rlhip.h rl_set_rows):
        // MSLR-WEB30K's 3.77 M x 136 floats are 2.05 GB, more than a ByteBuffer can hold
        RlHipNative.setData(handle, validation, null, n, features.length, labels, qoff, validation ? null : features, qkey);
        final int rowsPerBlock = Math.max(1, UPLOAD_BLOCK_BYTES / (4 * features.length));
        final FloatBuffer X = ByteBuffer.allocateDirect(rowsPerBlock * features.length * 4).order(ByteOrder.nativeOrder()).asFloatBuffer();
        long first = 0;
        int inBlock = 0;
        for (final RankList rl : lists) {
            for (int j = 0; j < rl.size(); j++) {
                final DataPoint dp = rl.get(j);
                for (final int fid : features) X.put(dp.getFeatureValue(fid));     // NaN/missing -> 0 here, as in the reference
                if (++inBlock == rowsPerBlock) { RlHipNative.setRows(handle, validation, first, inBlock, X); first += inBlock; inBlock = 0; X.clear(); }
            }
        }
        if (inBlock > 0) RlHipNative.setRows(handle, validation, first, inBlock, X);
    }

    /** What a RankLib scorer holds per qid for -qrel (NDCGScorer.idealGains, APScorer.relDocCount: metric/NDCGScorer.java:32, metric/APScorer.java:33).
     *  Preferred: the public getter INTEGRATION.md adds to the two scorers (getIdealGains() / getRelDocCount(), two lines each) -- no reflection on
     *  private state, works in a sealed module.  Without the patch the protected field is read reflectively; when neither is reachable training stops
     *  with a RankLibError that names the flag, never with silently different lambdas. */
    private static Object scorerField(final Object scorer, final Class<?> owner, final String name) {
        final String getter = "get" + Character.toUpperCase(name.charAt(0)) + name.substring(1);
        try {
            return owner.getMethod(getter).invoke(scorer);
        } catch (final NoSuchMethodException e) {
            // unpatched RankLib: fall through to the field
        } catch (final ReflectiveOperationException | RuntimeException e) {
            throw RankLibError.create("rlhip: " + owner.getSimpleName() + "." + getter + "() failed (needed for -qrel)", e);
        }
        try {
            final java.lang.reflect.Field f = owner.getDeclaredField(name);
            f.setAccessible(true);
            return f.get(scorer);
        } catch (final ReflectiveOperationException | RuntimeException e) {
            throw RankLibError.create("rlhip: cannot read " + owner.getSimpleName() + "." + name + " (needed for -qrel): add the getter " + getter
                    + "() from INTEGRATION.md to " + owner.getSimpleName() + ", or train without -qrel", e);
        }
    }

    /** -qrel (eval/Evaluator.java:243-244, :580-591): whatever the scorer holds per qid when training starts is what the reference's
     *  swapChange / score would find in it -- NDCG: idealGains entries (the external file's, and entries an earlier fold cached,
     *  metric/NDCGScorer.java:114-122,134-143); MAP: relDocCount of the judgment file, 0 for a qid that is not in it (metric/APScorer.java:86-94,124-143). */
    @SuppressWarnings("unchecked")
    private void forwardJudgments(final List<RankList> lists, final boolean validation) {
        if (scorer instanceof NDCGScorer) {
            final Map<String, Double> ig = (Map<String, Double>) scorerField(scorer, NDCGScorer.class, "idealGains");
            if (ig != null && !ig.isEmpty()) {
                final double[] ideal = new double[lists.size()];
                for (int q = 0; q < ideal.length; q++) { final Double d = ig.get(lists.get(q).getID()); ideal[q] = d == null ? Double.NaN : d; }
                RlHipNative.setExternalJudgments(handle, validation, ideal, null);
            }
        } else if (scorer instanceof APScorer) {
            final Map<String, Integer> rdc = (Map<String, Integer>) scorerField(scorer, APScorer.class, "relDocCount");
            if (rdc != null) {
                final int[] cnt = new int[lists.size()];
                for (int q = 0; q < cnt.length; q++) { final Integer it = rdc.get(lists.get(q).getID()); cnt[q] = it == null ? 0 : it; }
                RlHipNative.setExternalJudgments(handle, validation, null, cnt);
            }
        }
    }

    @Override
    public void init() {
        logger.info(() -> "Initializing... ");
        final String mname = scorer.name().split("@")[0];                  // "NDCG@10" -> "NDCG", "MAP" -> "MAP"
        final int metric = "NDCG".equals(mname) ? 0 : "DCG".equals(mname) ? 1 : "MAP".equals(mname) ? 2 : "ERR".equals(mname) ? 3 : -1;
        if (metric < 0) throw RankLibError.create("rlhip: the train metric must be NDCG, DCG, MAP or ERR (got " + scorer.name() + ")");
        RlHipNative.setErrMax(ERRScorer.MAX);                                // -gmax (eval/Evaluator.java:241-242): the trainer created next uses it
        handle = RlHipNative.create(nTrees, nTreeLeaves, nThreshold, minLeafSupport, nRoundToStopEarly, learningRate, metric, scorer.getK(),
                rankerId(), device, FeatureHistogram.samplingRate, seed++, javaOrder ? 16 : 0);
        impacts = new double[features.length];
        try {
            final Map<String, Integer> qids = new HashMap<>();
            upload(samples, false, qids);
            forwardJudgments(samples, false);
            if (validationSamples != null) { upload(validationSamples, true, qids); forwardJudgments(validationSamples, true); }
            RlHipNative.init(handle);
        } catch (final RuntimeException e) {      // never leak the device memory behind a failed init
            RlHipNative.destroy(handle);
            handle = 0;
            throw e;
        }
    }

    @Override
    public void learn() {
        ensemble = new Ensemble();
        logger.info(() -> "Training starts...");
        if (validationSamples != null) printLogLn(new int[] { 7, 9, 9 }, new String[] { "#iter", scorer.name() + "-T", scorer.name() + "-V" });
        else printLogLn(new int[] { 7, 9 }, new String[] { "#iter", scorer.name() + "-T" });
        final int cap = RlHipNative.treeCapacity(handle);          // 2 * leaves - 1, at least 3 (the root always splits once, RegressionTree.java:62-67); -leaf -1: 2 * (N / mls) - 1
        final int[] feature = new int[cap], left = new int[cap], right = new int[cap];
        final float[] threshold = new float[cap], output = new float[cap], metrics = new float[2];
        try {
            for (int m = 0; m < nTrees; m++) {
                printLog(new int[] { 7 }, new String[] { Integer.toString(m + 1) });
                final int r = RlHipNative.boostRound(handle, feature, threshold, left, right, output, metrics);
                ensemble.add(new RegressionTree(build(0, feature, threshold, left, right, output)), learningRate);
                printLog(new int[] { 9 }, new String[] { Double.toString(SimpleMath.round(metrics[0], 4)) });
                if (validationSamples != null) printLog(new int[] { 9 }, new String[] { Double.toString(SimpleMath.round(metrics[1], 4)) });
                flushLog();
                if (r < 0) break;                                       // early stop (learning/tree/LambdaMART.java:248)
            }
            final double[] fin = RlHipNative.finish(handle);             // rollback + scorer.score(rank(samples))
            while (ensemble.treeCount() > RlHipNative.numTrees(handle)) ensemble.remove(ensemble.treeCount() - 1);
            scoreOnTrainingData = fin[0];
            logger.info(() -> "Finished sucessfully.");
            logger.info(() -> scorer.name() + " on training data: " + SimpleMath.round(scoreOnTrainingData, 4));
            if (validationSamples != null) {
                bestScoreOnValidationData = fin[1];
                logger.info(() -> scorer.name() + " on validation data: " + SimpleMath.round(bestScoreOnValidationData, 4));
            }
            // the block every -ranker 6 log ends with (learning/tree/LambdaMART.java:267-271); impacts[] is all zeros in the reference too
            // (nothing ever adds to it), so the stable descending sort leaves the features in their own order
            logger.info(() -> "-- FEATURE IMPACTS");
            final int[] ftrsSorted = MergeSorter.sort(this.impacts, false);
            for (final int ftr : ftrsSorted) logger.info(() -> " Feature " + features[ftr] + " reduced error " + impacts[ftr]);
        } finally {
            RlHipNative.destroy(handle);
            handle = 0;
        }
    }

    private static Split build(final int n, final int[] f, final float[] t, final int[] l, final int[] r, final float[] o) {
        if (f[n] == -1) { final Split s = new Split(); s.setOutput(o[n]); return s; }      // Split.java:40-48,80-82
        final Split s = new Split(f[n], t[n], 0);
        s.setLeft(build(l[n], f, t, l, r, o));
        s.setRight(build(r[n], f, t, l, r, o));
        return s;
    }

    @Override public double eval(final DataPoint dp) { return ensemble.eval(dp); }
    @Override public Ranker createNew() { return new LambdaMART(); }
    @Override public String toString() { return ensemble.toString(); }
    @Override public String name() { return "LambdaMART"; }

    /** RL_RANKER_*: the MART drop-in (same package) overrides this with 0 */
    protected int rankerId() { return 6; }
    public Ensemble getEnsemble() { return ensemble; }

    @Override
    public String model() {                                            // identical text to the reference (:290-301)
        final StringBuilder output = new StringBuilder();
        output.append("## " + name() + "\n");
        output.append("## No. of trees = " + nTrees + "\n");
        output.append("## No. of leaves = " + nTreeLeaves + "\n");
        output.append("## No. of threshold candidates = " + nThreshold + "\n");
        output.append("## Learning rate = " + learningRate + "\n");
        output.append("## Stop early = " + nRoundToStopEarly + "\n");
        output.append("\n");
        output.append(toString());
        return output.toString();
    }

    @Override
    public void loadFromString(final String fullText) {
        final ModelLineProducer lineByLine = new ModelLineProducer();
        lineByLine.parse(fullText, (model, endEns) -> {});
        ensemble = new Ensemble(lineByLine.getModel().toString());
        features = ensemble.getFeatures();
    }

    @Override
    public void printParameters() {
        logger.info(() -> "No. of trees: " + nTrees);
        logger.info(() -> "No. of leaves: " + nTreeLeaves);
        logger.info(() -> "No. of threshold candidates: " + nThreshold);
        logger.info(() -> "Min leaf support: " + minLeafSupport);
        logger.info(() -> "Learning rate: " + learningRate);
        logger.info(() -> "Stop early: " + nRoundToStopEarly + " rounds without performance gain on validation data");
    }
}
