/*
 * jsdr_jni.c -- JNI shim between the Hip* plugin classes (java/com/ashbysoft/java_sdr) and libjsdr_hip.so.
 *
 * One function per native method of HipNative.java; each forwards to the C ABI of include/jsdr_hip.h and turns a
 * JSDR_ERR into an IllegalStateException carrying jsdr_last_error() -- an unchecked exception escaping a handler
 * ends the audio thread with a status message, exactly what the reference's loop does with any handler failure
 * (JavaAudio.java:321-328).  The audio thread re-uses its float[] / byte[] every iteration (JavaAudio.java:220-224):
 * the arrays are pinned with GetPrimitiveArrayCritical only for the duration of the call, inputs released with
 * JNI_ABORT (nothing to copy back), and every jsdr_*_receive_* has copied what it needs before it returns.
 *
 * Build where a JDK exists (this repo's build container has none; tests/test_jni_sources.py checks the file against
 * HipNative.java and jsdr_hip.h, and compiles it against prototype-only declarations):
 *     make -C jni JAVA_HOME=/path/to/jdk
 */
#include <jni.h>
#include <stdint.h>
#include "jsdr_hip.h"

#define FFT(h) ((jsdr_fft *)(intptr_t)(h))
#define BPSK(h) ((jsdr_bpsk *)(intptr_t)(h))
#define DEMOD(h) ((jsdr_demod *)(intptr_t)(h))

static void fail(JNIEnv *e)
{
    jclass c = (*e)->FindClass(e, "java/lang/IllegalStateException");
    if (c) (*e)->ThrowNew(e, c, jsdr_last_error());
}

/* ------------------------------------------------------------------ fft.java */
JNIEXPORT jlong JNICALL Java_com_ashbysoft_java_1sdr_HipNative_fftCreate(JNIEnv *e, jclass c, jint n, jint rate)
{
    jsdr_fft *h = 0;
    if (jsdr_fft_create(&h, n, rate) != JSDR_OK) {
        fail(e);
        return 0;
    }
    return (jlong)(intptr_t)h;
}

JNIEXPORT void JNICALL Java_com_ashbysoft_java_1sdr_HipNative_fftDestroy(JNIEnv *e, jclass c, jlong h)
{
    if (jsdr_fft_destroy(FFT(h)) != JSDR_OK) fail(e);
}

JNIEXPORT void JNICALL Java_com_ashbysoft_java_1sdr_HipNative_fftReceive(JNIEnv *e, jclass c, jlong h, jfloatArray buf,
                                                                         jfloatArray psd)
{
    jfloat *in = (*e)->GetPrimitiveArrayCritical(e, buf, 0);
    jfloat *out = (*e)->GetPrimitiveArrayCritical(e, psd, 0);
    int rc = (in && out) ? jsdr_fft_receive_f32(FFT(h), in, out) : JSDR_ERR;
    if (out) (*e)->ReleasePrimitiveArrayCritical(e, psd, out, 0);
    if (in) (*e)->ReleasePrimitiveArrayCritical(e, buf, in, JNI_ABORT);
    if (rc != JSDR_OK) fail(e);
}

JNIEXPORT void JNICALL Java_com_ashbysoft_java_1sdr_HipNative_fftReceiveRaw(JNIEnv *e, jclass c, jlong h, jbyteArray raw,
                                                                            jint ic, jint qc, jfloatArray psd)
{
    jbyte *in = (*e)->GetPrimitiveArrayCritical(e, raw, 0);
    jfloat *out = (*e)->GetPrimitiveArrayCritical(e, psd, 0);
    int rc = (in && out) ? jsdr_fft_receive_i16(FFT(h), (const int16_t *)in, ic, qc, out) : JSDR_ERR;
    if (out) (*e)->ReleasePrimitiveArrayCritical(e, psd, out, 0);
    if (in) (*e)->ReleasePrimitiveArrayCritical(e, raw, in, JNI_ABORT);
    if (rc != JSDR_OK) fail(e);
}

/* ------------------------------------------------------------------ FUNcubeBPSKDemod.java */
JNIEXPORT jlong JNICALL Java_com_ashbysoft_java_1sdr_HipNative_bpskCreate(JNIEnv *e, jclass c, jint rate, jint samples,
                                                                          jint tuning, jint doFFT, jint doUp)
{
    jsdr_bpsk *h = 0;
    if (jsdr_bpsk_create(&h, rate, samples, tuning, doFFT, doUp, 1, samples) != JSDR_OK) {
        fail(e);
        return 0;
    }
    return (jlong)(intptr_t)h;
}

JNIEXPORT void JNICALL Java_com_ashbysoft_java_1sdr_HipNative_bpskDestroy(JNIEnv *e, jclass c, jlong h)
{
    if (jsdr_bpsk_destroy(BPSK(h)) != JSDR_OK) fail(e);
}

JNIEXPORT void JNICALL Java_com_ashbysoft_java_1sdr_HipNative_bpskReceive(JNIEnv *e, jclass c, jlong h, jfloatArray buf)
{
    jfloat *in = (*e)->GetPrimitiveArrayCritical(e, buf, 0);
    int rc = in ? jsdr_bpsk_receive_f32(BPSK(h), in) : JSDR_ERR;
    if (in) (*e)->ReleasePrimitiveArrayCritical(e, buf, in, JNI_ABORT);
    if (rc != JSDR_OK) fail(e);
}

JNIEXPORT void JNICALL Java_com_ashbysoft_java_1sdr_HipNative_bpskReceiveRaw(JNIEnv *e, jclass c, jlong h, jbyteArray raw,
                                                                             jint ic, jint qc)
{
    jbyte *in = (*e)->GetPrimitiveArrayCritical(e, raw, 0);
    int rc = in ? jsdr_bpsk_receive_i16(BPSK(h), (const int16_t *)in, ic, qc) : JSDR_ERR;
    if (in) (*e)->ReleasePrimitiveArrayCritical(e, raw, in, JNI_ABORT);
    if (rc != JSDR_OK) fail(e);
}

JNIEXPORT void JNICALL Java_com_ashbysoft_java_1sdr_HipNative_bpskCounters(JNIEnv *e, jclass c, jlong h, jintArray out10)
{
    int32_t v[JSDR_BPSK_NCOUNTERS];
    if ((*e)->GetArrayLength(e, out10) < JSDR_BPSK_NCOUNTERS || jsdr_bpsk_get_counters(BPSK(h), 0, v) != JSDR_OK) {
        fail(e);
        return;
    }
    (*e)->SetIntArrayRegion(e, out10, 0, JSDR_BPSK_NCOUNTERS, (const jint *)v);
}

JNIEXPORT void JNICALL Java_com_ashbysoft_java_1sdr_HipNative_bpskDecoded(JNIEnv *e, jclass c, jlong h, jbyteArray out256)
{
    uint8_t v[256];
    if ((*e)->GetArrayLength(e, out256) < 256 || jsdr_bpsk_get_decoded(BPSK(h), 0, v) != JSDR_OK) {
        fail(e);
        return;
    }
    (*e)->SetByteArrayRegion(e, out256, 0, 256, (const jbyte *)v);
}

JNIEXPORT jint JNICALL Java_com_ashbysoft_java_1sdr_HipNative_bpskBits(JNIEnv *e, jclass c, jlong h, jbyteArray out)
{
    int n = 0;
    jsize cap = (*e)->GetArrayLength(e, out);
    jbyte *o = (*e)->GetPrimitiveArrayCritical(e, out, 0);
    int rc = o ? jsdr_bpsk_get_bits(BPSK(h), 0, (int8_t *)o, (int)cap, &n) : JSDR_ERR;
    if (o) (*e)->ReleasePrimitiveArrayCritical(e, out, o, 0);
    if (rc != JSDR_OK) {
        fail(e);
        return -1;
    }
    return n;
}

JNIEXPORT void JNICALL Java_com_ashbysoft_java_1sdr_HipNative_bpskState(JNIEnv *e, jclass c, jlong h, jdoubleArray out18)
{
    double v[18];
    if ((*e)->GetArrayLength(e, out18) < 18 || jsdr_bpsk_get_state(BPSK(h), 0, v) != JSDR_OK) {
        fail(e);
        return;
    }
    (*e)->SetDoubleArrayRegion(e, out18, 0, 18, v);
}

/* ------------------------------------------------------------------ FECDecoder.java */
JNIEXPORT jint JNICALL Java_com_ashbysoft_java_1sdr_HipNative_fecDecode(JNIEnv *e, jclass c, jbyteArray raw5200,
                                                                        jbyteArray out256)
{
    int rc = -1, st = JSDR_ERR;
    if ((*e)->GetArrayLength(e, raw5200) >= 5200 && (*e)->GetArrayLength(e, out256) >= 256) {
        jbyte *r = (*e)->GetPrimitiveArrayCritical(e, raw5200, 0);
        jbyte *o = (*e)->GetPrimitiveArrayCritical(e, out256, 0);
        if (r && o) st = jsdr_fec_decode((const uint8_t *)r, (uint8_t *)o, &rc);
        if (o) (*e)->ReleasePrimitiveArrayCritical(e, out256, o, 0);
        if (r) (*e)->ReleasePrimitiveArrayCritical(e, raw5200, r, JNI_ABORT);
    }
    if (st != JSDR_OK) {
        fail(e);
        return -1;
    }
    return rc; /* -1 or the channel error count, FECDecoder.java:851 */
}

/* ------------------------------------------------------------------ demod.java */
JNIEXPORT jlong JNICALL Java_com_ashbysoft_java_1sdr_HipNative_demodCreate(JNIEnv *e, jclass c, jint rate, jint n)
{
    jsdr_demod *h = 0;
    if (jsdr_demod_create(&h, rate, n, 1, n) != JSDR_OK) {
        fail(e);
        return 0;
    }
    return (jlong)(intptr_t)h;
}

JNIEXPORT void JNICALL Java_com_ashbysoft_java_1sdr_HipNative_demodDestroy(JNIEnv *e, jclass c, jlong h)
{
    if (jsdr_demod_destroy(DEMOD(h)) != JSDR_OK) fail(e);
}

JNIEXPORT void JNICALL Java_com_ashbysoft_java_1sdr_HipNative_demodConfigure(JNIEnv *e, jclass c, jlong h, jint mode,
                                                                             jboolean fir, jboolean down, jboolean agc)
{
    if (jsdr_demod_configure(DEMOD(h), mode, fir ? 1 : 0, down ? 1 : 0, agc ? 1 : 0) != JSDR_OK) fail(e);
}

JNIEXPORT void JNICALL Java_com_ashbysoft_java_1sdr_HipNative_demodWeights(JNIEnv *e, jclass c, jlong h, jint flo, jint fhi)
{
    if (jsdr_demod_weights(DEMOD(h), flo, fhi, 0, 0) != JSDR_OK) fail(e);
}

JNIEXPORT void JNICALL Java_com_ashbysoft_java_1sdr_HipNative_demodReceive(JNIEnv *e, jclass c, jlong h, jfloatArray buf,
                                                                           jbyteArray bbf)
{
    jfloat *in = (*e)->GetPrimitiveArrayCritical(e, buf, 0);
    jbyte *out = (*e)->GetPrimitiveArrayCritical(e, bbf, 0);
    /* bbf is little-endian (L,R) int16 pairs (demod.java:239-240,473-478): the device layout on a little-endian host */
    int rc = (in && out) ? jsdr_demod_receive_f32(DEMOD(h), in, (int16_t *)out) : JSDR_ERR;
    if (out) (*e)->ReleasePrimitiveArrayCritical(e, bbf, out, 0);
    if (in) (*e)->ReleasePrimitiveArrayCritical(e, buf, in, JNI_ABORT);
    if (rc != JSDR_OK) fail(e);
}

JNIEXPORT void JNICALL Java_com_ashbysoft_java_1sdr_HipNative_demodFrameStats(JNIEnv *e, jclass c, jlong h, jfloatArray out2)
{
    float v[2];
    if ((*e)->GetArrayLength(e, out2) < 2 || jsdr_demod_frame_stats(DEMOD(h), 0, &v[0], &v[1]) != JSDR_OK) {
        fail(e);
        return;
    }
    (*e)->SetFloatArrayRegion(e, out2, 0, 2, v);
}
