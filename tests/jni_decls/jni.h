/*
 * PROTOTYPE-ONLY declarations of the handful of JNI names jni/jsdr_jni.c uses, so that tests/test_jni_sources.py
 * can type-check the shim with `gcc -fsyntax-only` in a container without a JDK.  This is NOT a JNI implementation
 * and nothing links against it; the real build (jni/Makefile) uses $JAVA_HOME/include/jni.h.  Signatures follow the
 * JNI specification (Java SE, chapter 4 "JNI functions").
 */
#ifndef JSDR_TEST_JNI_DECLS_H
#define JSDR_TEST_JNI_DECLS_H
#include <stdint.h>

typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef uint8_t jboolean;
typedef float jfloat;
typedef double jdouble;
typedef jint jsize;
typedef struct _jobject *jobject;
typedef jobject jclass;
typedef jobject jarray;
typedef jarray jbyteArray;
typedef jarray jintArray;
typedef jarray jfloatArray;
typedef jarray jdoubleArray;
typedef jarray jlongArray;
typedef jarray jobjectArray;
typedef jobject jstring;

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2

struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;
struct JNINativeInterface_ {
    jclass (*FindClass)(JNIEnv *env, const char *name);
    jint (*ThrowNew)(JNIEnv *env, jclass clazz, const char *msg);
    jsize (*GetArrayLength)(JNIEnv *env, jarray array);
    void *(*GetPrimitiveArrayCritical)(JNIEnv *env, jarray array, jboolean *isCopy);
    void (*ReleasePrimitiveArrayCritical)(JNIEnv *env, jarray array, void *carray, jint mode);
    void (*GetByteArrayRegion)(JNIEnv *env, jbyteArray array, jsize start, jsize len, jbyte *buf);
    void (*GetFloatArrayRegion)(JNIEnv *env, jfloatArray array, jsize start, jsize len, jfloat *buf);
    void (*SetIntArrayRegion)(JNIEnv *env, jintArray array, jsize start, jsize len, const jint *buf);
    void (*SetByteArrayRegion)(JNIEnv *env, jbyteArray array, jsize start, jsize len, const jbyte *buf);
    void (*SetFloatArrayRegion)(JNIEnv *env, jfloatArray array, jsize start, jsize len, const jfloat *buf);
    void (*SetDoubleArrayRegion)(JNIEnv *env, jdoubleArray array, jsize start, jsize len, const jdouble *buf);
    void (*SetLongArrayRegion)(JNIEnv *env, jlongArray array, jsize start, jsize len, const jlong *buf);
    jobject (*GetObjectArrayElement)(JNIEnv *env, jobjectArray array, jsize index);
    const char *(*GetStringUTFChars)(JNIEnv *env, jstring str, jboolean *isCopy);
    void (*ReleaseStringUTFChars)(JNIEnv *env, jstring str, const char *chars);
    void (*DeleteLocalRef)(JNIEnv *env, jobject obj);
};
#endif
