package ciir.umass.edu.learning.tree;

import java.io.File;
import java.io.PrintWriter;
import java.nio.file.Files;
import java.util.HashMap;
import java.util.List;
import java.util.Map;

import ciir.umass.edu.features.FeatureManager;
import ciir.umass.edu.learning.RankList;
import ciir.umass.edu.metric.MetricScorer;
import ciir.umass.edu.metric.MetricScorerFactory;
import ciir.umass.edu.utilities.MyThreadPool;

/**
 * ORACLE PINNING KIT -- runs RankLib's OWN LambdaMART / MART on the committed fixture inputs and writes what the reference
 * actually computes, bit for bit, so that tests/golden/ can hold reference-OBSERVED vectors instead of outputs of this
 * repository's restatement (VERDICT r01 "parity unpinned").
 *
 * Needs a JDK and RankLib's classes (the unmodified reference, NOT the drop-in of this directory) on the classpath:
 *
 *   javac -cp RankLib.jar -d out integration/java/ciir/umass/edu/learning/tree/GoldenDump.java
 *   java  -cp RankLib.jar:out ciir.umass.edu.learning.tree.GoldenDump tests/golden/letor tests/golden/java
 *   python -m pytest tests/test_java_golden.py            # activates once tests/golden/java/*.txt exist
 *
 * NOT COMPILED IN THIS REPOSITORY (no JDK in the build image).  It only uses members that are public or protected in RankLib
 * 2.10.x: LambdaMART's protected fields (thresholds, hist, modelScores, pseudoResponses, weights) and protected hooks
 * (computePseudoResponses, updateTreeOutput, computeModelScoreOnTraining / OnValidation), FeatureHistogram's public arrays.
 *
 * Output, one record per line, floats / doubles as hexadecimal IEEE bit patterns:
 *   thr f b..            thresholds[f]                               (LambdaMART.java:108-150)
 *   bins f v..           hist.sampleToThresholdMap[f]                (FeatureHistogram.java:88-107)
 *   scores m b..         modelScores before round m
 *   lambda m b.. / weight m b..   pseudoResponses / weights of round m    (:361-396)
 *   roottot m s q        hist.sumResponse, hist.sqSumResponse after update  (FeatureHistogram.java:133-137)
 *   rootsum m f b..      hist.sum[f] (cumulative) of the root            (:126-146)
 *   tree m / .. / endtree    rt.toString("") after updateTreeOutput      (Split.java:132-155)
 *   leaves m b..         leaf outputs in Split.leaves() order, float bits of (float) getOutput()
 *   tmetric m b / vmetric m b     the float computeModelScoreOnTraining / OnValidation returned in round m
 *   final train b valid b trees n     after learn(): scoreOnTrainingData, bestScoreOnValidationData (double bits), trees kept
 */
public final class GoldenDump {
    private GoldenDump() {}

    static final class Out {
        final PrintWriter w; int round = -1;
        Out(final File f) throws Exception { w = new PrintWriter(f, "UTF-8"); }
        void doubles(final String tag, final double[] a) {
            final StringBuilder b = new StringBuilder(tag);
            for (final double v : a) b.append(' ').append(Long.toHexString(Double.doubleToLongBits(v)));
            w.println(b);
        }
        void floats(final String tag, final float[] a) {
            final StringBuilder b = new StringBuilder(tag);
            for (final float v : a) b.append(' ').append(Integer.toHexString(Float.floatToIntBits(v)));
            w.println(b);
        }
        void ints(final String tag, final int[] a) {
            final StringBuilder b = new StringBuilder(tag);
            for (final int v : a) b.append(' ').append(v);
            w.println(b);
        }
    }

    /** the hooks shared by the two subclasses below */
    interface Probe {
        double[] scores(); double[] lambdas(); double[] weightsOf(); FeatureHistogram root(); float[][] thr();
    }

    static void beforeRound(final Out o, final Probe p) { o.round++; o.doubles("scores " + o.round, p.scores()); }
    static void afterLambdas(final Out o, final Probe p) {
        o.doubles("lambda " + o.round, p.lambdas());
        o.doubles("weight " + o.round, p.weightsOf());
    }
    static void atTreeOutput(final Out o, final Probe p, final RegressionTree rt) {
        final FeatureHistogram h = p.root();          // the root's arrays are never reused by its children (FeatureHistogram.java:345-346)
        o.doubles("roottot " + o.round, new double[] { h.sumResponse, h.sqSumResponse });
        for (int f = 0; f < h.sum.length; f++) o.doubles("rootsum " + o.round + " " + f, h.sum[f]);
        o.w.println("tree " + o.round);
        o.w.print(rt.toString(""));
        o.w.println("endtree");
        final List<Split> leaves = rt.leaves();
        final float[] out = new float[leaves.size()];
        for (int i = 0; i < out.length; i++) out[i] = (float) leaves.get(i).getOutput();
        o.floats("leaves " + o.round, out);
    }
    static void init(final Out o, final Probe p) {
        final float[][] t = p.thr();
        for (int f = 0; f < t.length; f++) o.floats("thr " + f, t[f]);
        final int[][] m = p.root().sampleToThresholdMap;
        for (int f = 0; f < m.length; f++) o.ints("bins " + f, m[f]);
    }

    static final class DumpLambdaMART extends LambdaMART implements Probe {
        final Out o;
        DumpLambdaMART(final List<RankList> s, final int[] f, final MetricScorer sc, final Out o) { super(s, f, sc); this.o = o; }
        @Override public void init() { super.init(); GoldenDump.init(o, this); }
        @Override protected void computePseudoResponses() { beforeRound(o, this); super.computePseudoResponses(); afterLambdas(o, this); }
        @Override protected void updateTreeOutput(final RegressionTree rt) { super.updateTreeOutput(rt); atTreeOutput(o, this, rt); }
        @Override protected float computeModelScoreOnTraining() { final float v = super.computeModelScoreOnTraining(); o.floats("tmetric " + o.round, new float[] { v }); return v; }
        @Override protected float computeModelScoreOnValidation() { final float v = super.computeModelScoreOnValidation(); o.floats("vmetric " + o.round, new float[] { v }); return v; }
        public double[] scores() { return modelScores; } public double[] lambdas() { return pseudoResponses; } public double[] weightsOf() { return weights; }
        public FeatureHistogram root() { return hist; } public float[][] thr() { return thresholds; }
    }

    static final class DumpMART extends MART implements Probe {
        final Out o;
        DumpMART(final List<RankList> s, final int[] f, final MetricScorer sc, final Out o) { super(s, f, sc); this.o = o; }
        @Override public void init() { super.init(); GoldenDump.init(o, this); }
        @Override protected void computePseudoResponses() { beforeRound(o, this); super.computePseudoResponses(); afterLambdas(o, this); }
        @Override protected void updateTreeOutput(final RegressionTree rt) { super.updateTreeOutput(rt); atTreeOutput(o, this, rt); }
        @Override protected float computeModelScoreOnTraining() { final float v = super.computeModelScoreOnTraining(); o.floats("tmetric " + o.round, new float[] { v }); return v; }
        @Override protected float computeModelScoreOnValidation() { final float v = super.computeModelScoreOnValidation(); o.floats("vmetric " + o.round, new float[] { v }); return v; }
        public double[] scores() { return modelScores; } public double[] lambdas() { return pseudoResponses; } public double[] weightsOf() { return weights; }
        public FeatureHistogram root() { return hist; } public float[][] thr() { return thresholds; }
    }

    public static void main(final String[] args) throws Exception {
        final File in = new File(args[0]), outDir = new File(args[1]);
        outDir.mkdirs();
        MyThreadPool.init(args.length > 2 ? Integer.parseInt(args[2]) : Runtime.getRuntime().availableProcessors());
        for (final File pf : in.listFiles((d, n) -> n.endsWith(".params.txt"))) {
            final String name = pf.getName().replace(".params.txt", "");
            final Map<String, String> p = new HashMap<>();
            for (final String line : Files.readAllLines(pf.toPath())) { final int i = line.indexOf('='); if (i > 0) p.put(line.substring(0, i), line.substring(i + 1).trim()); }
            LambdaMART.nTrees = Integer.parseInt(p.get("n_trees"));
            LambdaMART.nTreeLeaves = Integer.parseInt(p.get("n_leaves"));
            LambdaMART.nThreshold = Integer.parseInt(p.get("n_threshold"));
            LambdaMART.minLeafSupport = Integer.parseInt(p.get("mls"));
            LambdaMART.learningRate = Float.parseFloat(p.get("lr"));
            LambdaMART.nRoundToStopEarly = p.containsKey("early_stop") ? Integer.parseInt(p.get("early_stop")) : 100;
            final String metric = p.containsKey("metric") ? p.get("metric") : "NDCG";
            final int k = Integer.parseInt(p.get("k"));
            final MetricScorer scorer = new MetricScorerFactory().createScorer("MAP".equals(metric) ? "MAP" : metric + "@" + k);
            final List<RankList> train = FeatureManager.readInput(new File(in, name + ".train.txt").getPath());
            final int[] features = FeatureManager.getFeatureFromSampleVector(train);
            final Out o = new Out(new File(outDir, name + ".txt"));
            final LambdaMART r = "MART".equals(p.get("ranker")) ? new DumpMART(train, features, scorer, o) : new DumpLambdaMART(train, features, scorer, o);
            final File vf = new File(in, name + ".valid.txt");
            if (vf.exists()) r.setValidationSet(FeatureManager.readInput(vf.getPath()));
            r.init();
            r.learn();
            o.doubles("scores_final", ((Probe) r).scores());
            o.w.println("final train " + Long.toHexString(Double.doubleToLongBits(r.getScoreOnTrainingData())) + " valid "
                    + Long.toHexString(Double.doubleToLongBits(vf.exists() ? r.getScoreOnValidationData() : 0.0)) + " trees " + r.getEnsemble().treeCount());
            o.w.println("model");
            o.w.print(r.model());
            o.w.println("endmodel");
            o.w.close();
            System.out.println(name + ": " + r.getEnsemble().treeCount() + " trees kept");
        }
        MyThreadPool.getInstance().shutdown();
    }
}
