/* MINIMAL stand-in for <jni.h>, used ONLY by tests/test_abi.py to type-check integration/jni/RlHipNative.c in an image without a JDK.
 * It declares, with the signatures of the Java Native Interface Specification (chapter 4, "JNI Functions"), exactly the entries of
 * JNINativeInterface_ the shim calls -- nothing links against it, no JVM is involved, and a real build uses the JDK's own header. */
#ifndef RLHIP_TEST_JNI_STANDIN_H
#define RLHIP_TEST_JNI_STANDIN_H
#include <stdarg.h>
#include <stdint.h>

typedef int32_t jint;
typedef int64_t jlong;
typedef float jfloat;
typedef double jdouble;
typedef uint8_t jboolean;
typedef jint jsize;
struct _jobject;
typedef struct _jobject *jobject;
typedef jobject jclass, jstring, jarray, jthrowable, jintArray, jfloatArray, jdoubleArray;
struct _jmethodID;
typedef struct _jmethodID *jmethodID;

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_FALSE 0
#define JNI_TRUE 1
#define JNI_ABORT 2

struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;
struct JNINativeInterface_ {
    jclass (*FindClass)(JNIEnv *env, const char *name);
    jint (*Throw)(JNIEnv *env, jthrowable obj);
    jmethodID (*GetStaticMethodID)(JNIEnv *env, jclass clazz, const char *name, const char *sig);
    jobject (*CallStaticObjectMethod)(JNIEnv *env, jclass clazz, jmethodID methodID, ...);
    jstring (*NewStringUTF)(JNIEnv *env, const char *utf);
    jsize (*GetArrayLength)(JNIEnv *env, jarray array);
    jdoubleArray (*NewDoubleArray)(JNIEnv *env, jsize len);
    jint *(*GetIntArrayElements)(JNIEnv *env, jintArray array, jboolean *isCopy);
    jfloat *(*GetFloatArrayElements)(JNIEnv *env, jfloatArray array, jboolean *isCopy);
    jdouble *(*GetDoubleArrayElements)(JNIEnv *env, jdoubleArray array, jboolean *isCopy);
    void (*ReleaseDoubleArrayElements)(JNIEnv *env, jdoubleArray array, jdouble *elems, jint mode);
    void (*ReleaseIntArrayElements)(JNIEnv *env, jintArray array, jint *elems, jint mode);
    void (*ReleaseFloatArrayElements)(JNIEnv *env, jfloatArray array, jfloat *elems, jint mode);
    void (*SetFloatArrayRegion)(JNIEnv *env, jfloatArray array, jsize start, jsize len, const jfloat *buf);
    void (*SetDoubleArrayRegion)(JNIEnv *env, jdoubleArray array, jsize start, jsize len, const jdouble *buf);
    void *(*GetDirectBufferAddress)(JNIEnv *env, jobject buf);
};
#endif
