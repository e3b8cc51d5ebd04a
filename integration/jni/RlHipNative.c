/*
 * RlHipNative.c -- JNI shim between RankLib (Java) and librlhip.so (include/rlhip.h).
 *
 * NOT COMPILED IN THIS REPOSITORY'S CI: the build image has no JDK (no jni.h).  A RankLib maintainer builds it
 * with:   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include \
 *             RlHipNative.c -L../../ranklib_amd/lib -lrlhip -o librlhipjni.so
 * All logic lives behind the C ABI; this file only moves arrays and rethrows errors as
 * ciir.umass.edu.utilities.RankLibError (utilities/RankLibError.java:32-34).
 */
#include <jni.h>
#include <stdlib.h>
#include "rlhip.h"

static void rl_throw(JNIEnv *env)
{
    jclass cls = (*env)->FindClass(env, "ciir/umass/edu/utilities/RankLibError");
    jmethodID create = (*env)->GetStaticMethodID(env, cls, "create", "(Ljava/lang/String;)Lciir/umass/edu/utilities/RankLibError;");
    jstring msg = (*env)->NewStringUTF(env, rl_last_error());
    jthrowable ex = (jthrowable)(*env)->CallStaticObjectMethod(env, cls, create, msg);
    (*env)->Throw(env, ex);
}
#define CHECK(rc) do { if ((rc) != RL_OK) { rl_throw(env); return 0; } } while (0)

/* long create(int nTrees, int nLeaves, int nThreshold, int minLeafSupport, int stopEarly, float lr, int metric, int k,
 *             int ranker, int device, float featureSamplingRate, long seed, int flags)
 * metric: RL_METRIC_*, ranker: RL_RANKER_* (include/rlhip.h); featureSamplingRate = FeatureHistogram.samplingRate (set by RFRanker.init) */
JNIEXPORT jlong JNICALL Java_ciir_umass_edu_learning_tree_RlHipNative_create(JNIEnv *env, jclass c, jint nTrees, jint nLeaves,
        jint nThreshold, jint mls, jint stopEarly, jfloat lr, jint metric, jint k, jint ranker, jint device,
        jfloat featureSamplingRate, jlong seed, jint flags)
{
    rl_params p; rl_trainer *t = NULL;
    rl_params_default(&p);
    p.n_trees = nTrees; p.n_leaves = nLeaves; p.n_threshold = nThreshold; p.min_leaf_support = mls;
    p.early_stop_rounds = stopEarly; p.learning_rate = lr; p.metric = metric; p.metric_k = k; p.ranker = ranker; p.device = device;
    p.feature_sampling_rate = featureSamplingRate; p.seed = (uint64_t)seed; p.flags = flags;
    CHECK(rl_create(&p, &t));
    return (jlong)(intptr_t)t;
}

/* ERRScorer.MAX (-gmax): a process-wide static on both sides; call before create() */
JNIEXPORT jint JNICALL Java_ciir_umass_edu_learning_tree_RlHipNative_setErrMax(JNIEnv *env, jclass c, jdouble maxGain)
{ CHECK(rl_set_err_max(maxGain)); return 0; }

/* nodes of the largest possible tree (after init): the array length boostRound needs, also for -leaf -1 */
JNIEXPORT jint JNICALL Java_ciir_umass_edu_learning_tree_RlHipNative_treeCapacity(JNIEnv *env, jclass c, jlong h)
{ int32_t n = 0; CHECK(rl_tree_capacity((rl_trainer *)(intptr_t)h, &n)); return n; }

JNIEXPORT void JNICALL Java_ciir_umass_edu_learning_tree_RlHipNative_destroy(JNIEnv *env, jclass c, jlong h)
{ rl_destroy((rl_trainer *)(intptr_t)h); }

/* X: direct FloatBuffer [nDocs*nFeatures] filled through DataPoint.getFeatureValue(features[f]) */
JNIEXPORT jint JNICALL Java_ciir_umass_edu_learning_tree_RlHipNative_setData(JNIEnv *env, jclass c, jlong h, jboolean validation,
        jobject X, jlong nDocs, jint nFeatures, jfloatArray labels, jintArray qoff, jintArray featureIds, jintArray qkey)
{
    float *x = X ? (float *)(*env)->GetDirectBufferAddress(env, X) : NULL;      /* null: the rows follow through setRows */
    jint nq = (*env)->GetArrayLength(env, qoff) - 1;
    jfloat *lab = (*env)->GetFloatArrayElements(env, labels, NULL);
    jint *qo = (*env)->GetIntArrayElements(env, qoff, NULL);
    jint *fid = featureIds ? (*env)->GetIntArrayElements(env, featureIds, NULL) : NULL;
    jint *qk = qkey ? (*env)->GetIntArrayElements(env, qkey, NULL) : NULL;
    int rc = validation ? rl_set_validation((rl_trainer *)(intptr_t)h, x, nDocs, lab, (const int32_t *)qo, nq, (const int32_t *)qk)
                        : rl_set_train((rl_trainer *)(intptr_t)h, x, nDocs, nFeatures, lab, (const int32_t *)qo, nq,
                                       (const int32_t *)fid, (const int32_t *)qk);
    (*env)->ReleaseFloatArrayElements(env, labels, lab, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, qoff, qo, JNI_ABORT);
    if (fid) (*env)->ReleaseIntArrayElements(env, featureIds, fid, JNI_ABORT);
    if (qk) (*env)->ReleaseIntArrayElements(env, qkey, qk, JNI_ABORT);
    CHECK(rc);
    return 0;
}

/* rows [firstDoc, firstDoc + nDocs) of the data set declared by setData(.., X = null, ..) */
JNIEXPORT jint JNICALL Java_ciir_umass_edu_learning_tree_RlHipNative_setRows(JNIEnv *env, jclass c, jlong h, jboolean validation,
        jlong firstDoc, jlong nDocs, jobject X)
{
    CHECK(rl_set_rows((rl_trainer *)(intptr_t)h, validation ? 1 : 0, firstDoc, nDocs, (const float *)(*env)->GetDirectBufferAddress(env, X)));
    return 0;
}

/* -qrel: what a scorer that loaded an external judgment file holds, resolved per ranked list by the caller */
JNIEXPORT jint JNICALL Java_ciir_umass_edu_learning_tree_RlHipNative_setExternalJudgments(JNIEnv *env, jclass c, jlong h, jboolean validation,
        jdoubleArray idealDcg, jintArray relDocCount)
{
    jdouble *idl = idealDcg ? (*env)->GetDoubleArrayElements(env, idealDcg, NULL) : NULL;
    jint *rdc = relDocCount ? (*env)->GetIntArrayElements(env, relDocCount, NULL) : NULL;
    const int rc = rl_set_external_judgments((rl_trainer *)(intptr_t)h, validation ? 1 : 0, idl, (const int32_t *)rdc);
    if (idl) (*env)->ReleaseDoubleArrayElements(env, idealDcg, idl, JNI_ABORT);
    if (rdc) (*env)->ReleaseIntArrayElements(env, relDocCount, rdc, JNI_ABORT);
    CHECK(rc);
    return 0;
}

JNIEXPORT jint JNICALL Java_ciir_umass_edu_learning_tree_RlHipNative_init(JNIEnv *env, jclass c, jlong h)
{ CHECK(rl_init((rl_trainer *)(intptr_t)h)); return 0; }

/* one round; the flat tree is returned through the caller's arrays (capacity 2*nLeaves-1);
 * metrics[0] = train metric, metrics[1] = validation metric; returns (stop ? -n_nodes : n_nodes) */
JNIEXPORT jint JNICALL Java_ciir_umass_edu_learning_tree_RlHipNative_boostRound(JNIEnv *env, jclass c, jlong h, jintArray feature,
        jfloatArray threshold, jintArray left, jintArray right, jfloatArray output, jfloatArray metrics)
{
    rl_tree t; float m[2] = {0, 0}; int32_t stop = 0;
    t.cap = (*env)->GetArrayLength(env, feature);
    t.feature = (int32_t *)(*env)->GetIntArrayElements(env, feature, NULL);
    t.threshold = (*env)->GetFloatArrayElements(env, threshold, NULL);
    t.left = (int32_t *)(*env)->GetIntArrayElements(env, left, NULL);
    t.right = (int32_t *)(*env)->GetIntArrayElements(env, right, NULL);
    t.output = (*env)->GetFloatArrayElements(env, output, NULL);
    t.deviance = NULL; t.count = NULL;
    int rc = rl_boost_round((rl_trainer *)(intptr_t)h, &t, &m[0], &m[1], &stop);
    (*env)->ReleaseIntArrayElements(env, feature, (jint *)t.feature, 0);
    (*env)->ReleaseFloatArrayElements(env, threshold, t.threshold, 0);
    (*env)->ReleaseIntArrayElements(env, left, (jint *)t.left, 0);
    (*env)->ReleaseIntArrayElements(env, right, (jint *)t.right, 0);
    (*env)->ReleaseFloatArrayElements(env, output, t.output, 0);
    (*env)->SetFloatArrayRegion(env, metrics, 0, 2, m);
    CHECK(rc);
    return stop ? -t.n_nodes : t.n_nodes;
}

/* returns {train score, validation score}; trees beyond the best validation round are dropped */
JNIEXPORT jdoubleArray JNICALL Java_ciir_umass_edu_learning_tree_RlHipNative_finish(JNIEnv *env, jclass c, jlong h)
{
    double v[2] = {0, 0};
    CHECK(rl_finish((rl_trainer *)(intptr_t)h, &v[0], &v[1]));
    jdoubleArray out = (*env)->NewDoubleArray(env, 2);
    (*env)->SetDoubleArrayRegion(env, out, 0, 2, v);
    return out;
}

JNIEXPORT jint JNICALL Java_ciir_umass_edu_learning_tree_RlHipNative_numTrees(JNIEnv *env, jclass c, jlong h)
{ int32_t n = 0; CHECK(rl_num_trees((rl_trainer *)(intptr_t)h, &n)); return n; }
