/*
 * jsdr_jni.c -- JNI shim between the Hip* plugin classes (java/com/ashbysoft/java_sdr) and libjsdr_hip.so.
 *
 * One function per native method of HipNative.java; each forwards to the C ABI of include/jsdr_hip.h and turns a
 * JSDR_ERR into an IllegalStateException carrying jsdr_last_error() -- an unchecked exception escaping a handler
 * ends the audio thread with a status message, exactly what the reference's loop does with any handler failure
 * (JavaAudio.java:321-328).
 *
 * Buffers.  The audio thread re-uses its float[] / byte[] every iteration (JavaAudio.java:220-224), and around an
 * "audio-change" a handler can be handed a buffer of the OLD geometry.  So every receive
 *   1. checks the Java array's length against the frame size the handle was created with (IllegalArgumentException
 *      with its own text otherwise -- never a read or write past the array);
 *   2. copies the frame with Get<Type>ArrayRegion into a staging buffer that belongs to the handle, calls the C ABI on
 *      that, and copies results back with Set<Type>ArrayRegion.  No GetPrimitiveArrayCritical region surrounds a
 *      device call: the JNI specification forbids blocking inside one (the GC is locked out for its duration), and a
 *      receive() is a host->device copy, a chain of kernels and a wait.
 * The jlong handle the Java side holds is a pointer to the context struct below (0 = no handle; every entry point
 * rejects it).
 *
 * Build where a JDK exists (this repo's build container has none; tests/test_jni_sources.py checks the file against
 * HipNative.java and jsdr_hip.h, and compiles it against prototype-only declarations):
 *     make -C jni JAVA_HOME=/path/to/jdk
 */
#include <jni.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "jsdr_hip.h"

typedef struct {
    jsdr_fft *h;
    int n;          /* samples per frame */
    float *in;      /* [2n] or, reinterpreted, 4n raw bytes */
    float *psd;     /* [n+2] */
} fft_ctx;
typedef struct {
    jsdr_bpsk *h;
    int n;
    float *in;      /* [2n] / 4n raw bytes */
} bpsk_ctx;
typedef struct {
    jsdr_demod *h;
    int n;
    float *in;      /* [2n] */
    int16_t *out;   /* [2n] (L,R) pairs */
} demod_ctx;
typedef struct {
    jsdr_phase *h;
    int n;
    float *in;      /* [2n] */
    int32_t *pix;   /* [n+1] */
    float *avgi, *avgq;
} phase_ctx;

typedef struct {
    jsdr_group *g;
    int ndev, per_dev, total;
    int64_t max_batch, slot_bytes;
    int16_t **raw;  /* [ndev] device buffers [per_dev][2 * max_batch], each on its own device */
} group_ctx;

#define CTX(type, h) ((type *)(intptr_t)(h))

static void throw_new(JNIEnv *e, const char *cls, const char *msg)
{
    jclass c = (*e)->FindClass(e, cls);
    if (c) (*e)->ThrowNew(e, c, msg);
}

/* a C ABI call failed: its own message */
static void fail(JNIEnv *e) { throw_new(e, "java/lang/IllegalStateException", jsdr_last_error()); }

static void fail_msg(JNIEnv *e, const char *msg) { throw_new(e, "java/lang/IllegalStateException", msg); }

/* argument check of the shim itself: its own text, never a stale jsdr_last_error() */
static int bad_length(JNIEnv *e, const char *what, jsize have, jsize want)
{
    if (have == want) return 0;
    char msg[160];
    snprintf(msg, sizeof(msg), "%s: array of %d elements, the handle's frame needs %d", what, (int)have, (int)want);
    throw_new(e, "java/lang/IllegalArgumentException", msg);
    return 1;
}

static int short_array(JNIEnv *e, const char *what, jsize have, jsize want)
{
    if (have >= want) return 0;
    char msg[160];
    snprintf(msg, sizeof(msg), "%s: array of %d elements, at least %d needed", what, (int)have, (int)want);
    throw_new(e, "java/lang/IllegalArgumentException", msg);
    return 1;
}

static int null_handle(JNIEnv *e, const void *ctx, const char *what)
{
    if (ctx) return 0;
    char msg[96];
    snprintf(msg, sizeof(msg), "%s: no native handle (closed, or its creation failed)", what);
    throw_new(e, "java/lang/IllegalStateException", msg);
    return 1;
}

/* ------------------------------------------------------------------ fft.java */
JNIEXPORT jlong JNICALL Java_com_ashbysoft_java_1sdr_HipNative_fftCreate(JNIEnv *e, jclass c, jint n, jint rate)
{
    fft_ctx *x = n > 0 ? calloc(1, sizeof(*x)) : 0;
    if (x) {
        x->n = n;
        x->in = malloc(sizeof(float) * 2 * (size_t)n);
        x->psd = malloc(sizeof(float) * ((size_t)n + 2));
    }
    if (!x || !x->in || !x->psd) {
        if (x) { free(x->in); free(x->psd); free(x); }
        fail_msg(e, "fftCreate: bad frame size or out of memory");
        return 0;
    }
    if (jsdr_fft_create(&x->h, n, rate) != JSDR_OK) {
        free(x->in); free(x->psd); free(x);
        fail(e);
        return 0;
    }
    return (jlong)(intptr_t)x;
}

JNIEXPORT void JNICALL Java_com_ashbysoft_java_1sdr_HipNative_fftDestroy(JNIEnv *e, jclass c, jlong h)
{
    fft_ctx *x = CTX(fft_ctx, h);
    if (!x) return;
    int rc = jsdr_fft_destroy(x->h);
    free(x->in); free(x->psd); free(x);
    if (rc != JSDR_OK) fail(e);
}

JNIEXPORT void JNICALL Java_com_ashbysoft_java_1sdr_HipNative_fftReceive(JNIEnv *e, jclass c, jlong h, jfloatArray buf,
                                                                         jfloatArray psd)
{
    fft_ctx *x = CTX(fft_ctx, h);
    if (null_handle(e, x, "fftReceive")) return;
    if (bad_length(e, "fftReceive: buf", (*e)->GetArrayLength(e, buf), 2 * x->n)) return;
    if (bad_length(e, "fftReceive: psd", (*e)->GetArrayLength(e, psd), x->n + 2)) return;
    (*e)->GetFloatArrayRegion(e, buf, 0, 2 * x->n, x->in);
    if (jsdr_fft_receive_f32(x->h, x->in, x->psd) != JSDR_OK) {
        fail(e);
        return;
    }
    (*e)->SetFloatArrayRegion(e, psd, 0, x->n + 2, x->psd);
}

JNIEXPORT void JNICALL Java_com_ashbysoft_java_1sdr_HipNative_fftReceiveRaw(JNIEnv *e, jclass c, jlong h, jbyteArray raw,
                                                                            jint ic, jint qc, jfloatArray psd)
{
    fft_ctx *x = CTX(fft_ctx, h);
    if (null_handle(e, x, "fftReceiveRaw")) return;
    if (bad_length(e, "fftReceiveRaw: raw", (*e)->GetArrayLength(e, raw), 4 * x->n)) return;
    if (bad_length(e, "fftReceiveRaw: psd", (*e)->GetArrayLength(e, psd), x->n + 2)) return;
    (*e)->GetByteArrayRegion(e, raw, 0, 4 * x->n, (jbyte *)x->in);  /* 4n bytes fit the 2n-float staging buffer */
    if (jsdr_fft_receive_i16(x->h, (const int16_t *)x->in, ic, qc, x->psd) != JSDR_OK) {
        fail(e);
        return;
    }
    (*e)->SetFloatArrayRegion(e, psd, 0, x->n + 2, x->psd);
}

/* ------------------------------------------------------------------ FUNcubeBPSKDemod.java */
JNIEXPORT jlong JNICALL Java_com_ashbysoft_java_1sdr_HipNative_bpskCreate(JNIEnv *e, jclass c, jint rate, jint samples,
                                                                          jint tuning, jint doFFT, jint doUp)
{
    bpsk_ctx *x = samples > 0 ? calloc(1, sizeof(*x)) : 0;
    if (x) {
        x->n = samples;
        x->in = malloc(sizeof(float) * 2 * (size_t)samples);
    }
    if (!x || !x->in) {
        free(x);
        fail_msg(e, "bpskCreate: bad frame size or out of memory");
        return 0;
    }
    if (jsdr_bpsk_create(&x->h, rate, samples, tuning, doFFT, doUp, 1, samples) != JSDR_OK) {
        free(x->in); free(x);
        fail(e);
        return 0;
    }
    return (jlong)(intptr_t)x;
}

JNIEXPORT void JNICALL Java_com_ashbysoft_java_1sdr_HipNative_bpskDestroy(JNIEnv *e, jclass c, jlong h)
{
    bpsk_ctx *x = CTX(bpsk_ctx, h);
    if (!x) return;
    int rc = jsdr_bpsk_destroy(x->h);
    free(x->in); free(x);
    if (rc != JSDR_OK) fail(e);
}

JNIEXPORT void JNICALL Java_com_ashbysoft_java_1sdr_HipNative_bpskReceive(JNIEnv *e, jclass c, jlong h, jfloatArray buf)
{
    bpsk_ctx *x = CTX(bpsk_ctx, h);
    if (null_handle(e, x, "bpskReceive")) return;
    if (bad_length(e, "bpskReceive: buf", (*e)->GetArrayLength(e, buf), 2 * x->n)) return;
    (*e)->GetFloatArrayRegion(e, buf, 0, 2 * x->n, x->in);
    if (jsdr_bpsk_receive_f32(x->h, x->in) != JSDR_OK) fail(e);
}

JNIEXPORT void JNICALL Java_com_ashbysoft_java_1sdr_HipNative_bpskReceiveRaw(JNIEnv *e, jclass c, jlong h, jbyteArray raw,
                                                                             jint ic, jint qc)
{
    bpsk_ctx *x = CTX(bpsk_ctx, h);
    if (null_handle(e, x, "bpskReceiveRaw")) return;
    if (bad_length(e, "bpskReceiveRaw: raw", (*e)->GetArrayLength(e, raw), 4 * x->n)) return;
    (*e)->GetByteArrayRegion(e, raw, 0, 4 * x->n, (jbyte *)x->in);
    if (jsdr_bpsk_receive_i16(x->h, (const int16_t *)x->in, ic, qc) != JSDR_OK) fail(e);
}

/* The results of the last completed receive(), for ANY thread (the Swing thread paints them while the audio thread is
 * inside the next receive, FUNcubeBPSKDemod.java:220-228,331-337): jsdr_bpsk_snapshot_read takes no lock and makes no
 * device call.  counters11 = the ten counters + the number of bits sliced in that frame; returns the number of frames
 * received so far (0: nothing yet, the arrays are left untouched). */
JNIEXPORT jlong JNICALL Java_com_ashbysoft_java_1sdr_HipNative_bpskSnapshot(JNIEnv *e, jclass c, jlong h,
                                                                            jintArray counters11, jdoubleArray state18,
                                                                            jbyteArray decoded256, jbyteArray bits512)
{
    bpsk_ctx *x = CTX(bpsk_ctx, h);
    if (null_handle(e, x, "bpskSnapshot")) return 0;
    if (short_array(e, "bpskSnapshot: counters", (*e)->GetArrayLength(e, counters11), JSDR_BPSK_NCOUNTERS + 1) ||
        short_array(e, "bpskSnapshot: state", (*e)->GetArrayLength(e, state18), 18) ||
        short_array(e, "bpskSnapshot: decoded", (*e)->GetArrayLength(e, decoded256), 256) ||
        short_array(e, "bpskSnapshot: bits", (*e)->GetArrayLength(e, bits512), 512))
        return 0;
    jsdr_bpsk_snapshot sn;
    if (jsdr_bpsk_snapshot_read(x->h, &sn) != JSDR_OK) {
        fail(e);
        return 0;
    }
    if (sn.frames == 0) return 0;
    jint cn[JSDR_BPSK_NCOUNTERS + 1];
    for (int i = 0; i < JSDR_BPSK_NCOUNTERS; i++) cn[i] = sn.counters[i];
    cn[JSDR_BPSK_NCOUNTERS] = sn.nbits;
    (*e)->SetIntArrayRegion(e, counters11, 0, JSDR_BPSK_NCOUNTERS + 1, cn);
    (*e)->SetDoubleArrayRegion(e, state18, 0, 18, sn.state);
    (*e)->SetByteArrayRegion(e, decoded256, 0, 256, (const jbyte *)sn.decoded);
    (*e)->SetByteArrayRegion(e, bits512, 0, 512, (const jbyte *)sn.bits);
    return (jlong)sn.frames;
}

/* every bit of the last receive() (the snapshot keeps the first 512); a device call: audio thread only */
JNIEXPORT jint JNICALL Java_com_ashbysoft_java_1sdr_HipNative_bpskBits(JNIEnv *e, jclass c, jlong h, jbyteArray out)
{
    bpsk_ctx *x = CTX(bpsk_ctx, h);
    if (null_handle(e, x, "bpskBits")) return -1;
    int n = 0;
    jsize cap = (*e)->GetArrayLength(e, out);
    int8_t *tmp = malloc(cap > 0 ? (size_t)cap : 1);
    if (!tmp) {
        fail_msg(e, "bpskBits: out of memory");
        return -1;
    }
    int rc = jsdr_bpsk_get_bits(x->h, 0, tmp, (int)cap, &n);
    if (rc == JSDR_OK) (*e)->SetByteArrayRegion(e, out, 0, n < cap ? n : cap, (const jbyte *)tmp);
    free(tmp);
    if (rc != JSDR_OK) {
        fail(e);
        return -1;
    }
    return n;
}

/* ------------------------------------------------------------------ FECDecoder.java */
JNIEXPORT jint JNICALL Java_com_ashbysoft_java_1sdr_HipNative_fecDecode(JNIEnv *e, jclass c, jbyteArray raw5200,
                                                                        jbyteArray out256)
{
    if (short_array(e, "fecDecode: raw", (*e)->GetArrayLength(e, raw5200), 5200) ||
        short_array(e, "fecDecode: out", (*e)->GetArrayLength(e, out256), 256))
        return -1;
    uint8_t raw[5200], out[256];
    int rc = -1;
    (*e)->GetByteArrayRegion(e, raw5200, 0, 5200, (jbyte *)raw);
    (*e)->GetByteArrayRegion(e, out256, 0, 256, (jbyte *)out);  /* out[] keeps its content when the decode fails (:780) */
    if (jsdr_fec_decode(raw, out, &rc) != JSDR_OK) {
        fail(e);
        return -1;
    }
    (*e)->SetByteArrayRegion(e, out256, 0, 256, (const jbyte *)out);
    return rc; /* -1 or the channel error count, FECDecoder.java:851 */
}

/* ------------------------------------------------------------------ demod.java */
JNIEXPORT jlong JNICALL Java_com_ashbysoft_java_1sdr_HipNative_demodCreate(JNIEnv *e, jclass c, jint rate, jint n)
{
    demod_ctx *x = n > 0 ? calloc(1, sizeof(*x)) : 0;
    if (x) {
        x->n = n;
        x->in = malloc(sizeof(float) * 2 * (size_t)n);
        x->out = malloc(sizeof(int16_t) * 2 * (size_t)n);
    }
    if (!x || !x->in || !x->out) {
        if (x) { free(x->in); free(x->out); free(x); }
        fail_msg(e, "demodCreate: bad frame size or out of memory");
        return 0;
    }
    if (jsdr_demod_create(&x->h, rate, n, 1, n) != JSDR_OK) {
        free(x->in); free(x->out); free(x);
        fail(e);
        return 0;
    }
    return (jlong)(intptr_t)x;
}

JNIEXPORT void JNICALL Java_com_ashbysoft_java_1sdr_HipNative_demodDestroy(JNIEnv *e, jclass c, jlong h)
{
    demod_ctx *x = CTX(demod_ctx, h);
    if (!x) return;
    int rc = jsdr_demod_destroy(x->h);
    free(x->in); free(x->out); free(x);
    if (rc != JSDR_OK) fail(e);
}

JNIEXPORT void JNICALL Java_com_ashbysoft_java_1sdr_HipNative_demodConfigure(JNIEnv *e, jclass c, jlong h, jint mode,
                                                                             jboolean fir, jboolean down, jboolean agc)
{
    demod_ctx *x = CTX(demod_ctx, h);
    if (null_handle(e, x, "demodConfigure")) return;
    if (jsdr_demod_configure(x->h, mode, fir ? 1 : 0, down ? 1 : 0, agc ? 1 : 0) != JSDR_OK) fail(e);
}

JNIEXPORT void JNICALL Java_com_ashbysoft_java_1sdr_HipNative_demodWeights(JNIEnv *e, jclass c, jlong h, jint flo, jint fhi)
{
    demod_ctx *x = CTX(demod_ctx, h);
    if (null_handle(e, x, "demodWeights")) return;
    if (jsdr_demod_weights(x->h, flo, fhi, 0, 0) != JSDR_OK) fail(e);
}

JNIEXPORT void JNICALL Java_com_ashbysoft_java_1sdr_HipNative_demodReceive(JNIEnv *e, jclass c, jlong h, jfloatArray buf,
                                                                           jbyteArray bbf)
{
    demod_ctx *x = CTX(demod_ctx, h);
    if (null_handle(e, x, "demodReceive")) return;
    if (bad_length(e, "demodReceive: buf", (*e)->GetArrayLength(e, buf), 2 * x->n)) return;
    if (bad_length(e, "demodReceive: bbf", (*e)->GetArrayLength(e, bbf), 4 * x->n)) return;
    (*e)->GetFloatArrayRegion(e, buf, 0, 2 * x->n, x->in);
    if (jsdr_demod_receive_f32(x->h, x->in, x->out) != JSDR_OK) {
        fail(e);
        return;
    }
    /* bbf is little-endian (L,R) int16 pairs (demod.java:239-240,473-478): the device layout on a little-endian host */
    (*e)->SetByteArrayRegion(e, bbf, 0, 4 * x->n, (const jbyte *)x->out);
}

JNIEXPORT void JNICALL Java_com_ashbysoft_java_1sdr_HipNative_demodFrameStats(JNIEnv *e, jclass c, jlong h, jfloatArray out2)
{
    demod_ctx *x = CTX(demod_ctx, h);
    if (null_handle(e, x, "demodFrameStats")) return;
    float v[2];
    if (short_array(e, "demodFrameStats: out", (*e)->GetArrayLength(e, out2), 2)) return;
    if (jsdr_demod_frame_stats(x->h, 0, &v[0], &v[1]) != JSDR_OK) {
        fail(e);
        return;
    }
    (*e)->SetFloatArrayRegion(e, out2, 0, 2, v);
}

/* ------------------------------------------------------------------ phase.java */
JNIEXPORT jlong JNICALL Java_com_ashbysoft_java_1sdr_HipNative_phaseCreate(JNIEnv *e, jclass c, jint n)
{
    phase_ctx *x = n > 0 ? calloc(1, sizeof(*x)) : 0;
    if (x) {
        x->n = n;
        x->in = malloc(sizeof(float) * 2 * (size_t)n);
        x->pix = malloc(sizeof(int32_t) * ((size_t)n + 1));
        x->avgi = malloc(sizeof(float) * ((size_t)n + 1));
        x->avgq = malloc(sizeof(float) * ((size_t)n + 1));
    }
    if (!x || !x->in || !x->pix || !x->avgi || !x->avgq || jsdr_phase_create(&x->h, n) != JSDR_OK) {
        int abi = x && x->in && x->pix && x->avgi && x->avgq;
        if (x) { free(x->in); free(x->pix); free(x->avgi); free(x->avgq); free(x); }
        if (abi) fail(e); else fail_msg(e, "phaseCreate: bad frame size or out of memory");
        return 0;
    }
    return (jlong)(intptr_t)x;
}

JNIEXPORT void JNICALL Java_com_ashbysoft_java_1sdr_HipNative_phaseDestroy(JNIEnv *e, jclass c, jlong h)
{
    phase_ctx *x = CTX(phase_ctx, h);
    if (!x) return;
    int rc = jsdr_phase_destroy(x->h);
    free(x->in); free(x->pix); free(x->avgi); free(x->avgq); free(x);
    if (rc != JSDR_OK) fail(e);
}

/* phase.receive (phase.java:123-128): the frame goes to the device, where max|x| (:75-80) is taken at once */
JNIEXPORT void JNICALL Java_com_ashbysoft_java_1sdr_HipNative_phaseReceive(JNIEnv *e, jclass c, jlong h, jfloatArray buf)
{
    phase_ctx *x = CTX(phase_ctx, h);
    if (null_handle(e, x, "phaseReceive")) return;
    if (bad_length(e, "phaseReceive: buf", (*e)->GetArrayLength(e, buf), 2 * x->n)) return;
    (*e)->GetFloatArrayRegion(e, buf, 0, 2 * x->n, x->in);
    if (jsdr_phase_receive_f32(x->h, x->in) != JSDR_OK) fail(e);
}

JNIEXPORT jfloat JNICALL Java_com_ashbysoft_java_1sdr_HipNative_phaseMaxabs(JNIEnv *e, jclass c, jlong h)
{
    phase_ctx *x = CTX(phase_ctx, h);
    float m = 0;
    if (null_handle(e, x, "phaseMaxabs")) return 0;
    if (jsdr_phase_get_max(x->h, &m) != JSDR_OK) fail(e);
    return m;
}

/* the per-pixel-column means of I and Q for a panel bx pixels wide (phase.java:93-116); returns the column count */
JNIEXPORT jint JNICALL Java_com_ashbysoft_java_1sdr_HipNative_phaseColumns(JNIEnv *e, jclass c, jlong h, jint bx,
                                                                           jintArray pix, jfloatArray avgi, jfloatArray avgq)
{
    phase_ctx *x = CTX(phase_ctx, h);
    if (null_handle(e, x, "phaseColumns")) return -1;
    int ncol = 0;
    if (jsdr_phase_get_columns(x->h, bx, x->pix, x->avgi, x->avgq, x->n + 1, &ncol) != JSDR_OK) {
        fail(e);
        return -1;
    }
    if (short_array(e, "phaseColumns: pix", (*e)->GetArrayLength(e, pix), ncol) ||
        short_array(e, "phaseColumns: avgi", (*e)->GetArrayLength(e, avgi), ncol) ||
        short_array(e, "phaseColumns: avgq", (*e)->GetArrayLength(e, avgq), ncol))
        return -1;
    (*e)->SetIntArrayRegion(e, pix, 0, ncol, (const jint *)x->pix);
    (*e)->SetFloatArrayRegion(e, avgi, 0, ncol, x->avgi);
    (*e)->SetFloatArrayRegion(e, avgq, 0, ncol, x->avgq);
    return ncol;
}

/* ------------------------------------------------------------------ one JVM, several GPUs (jsdr.java:479-483; jsdr_group_*) */
static void group_free(group_ctx *x)
{
    if (!x) return;
    if (x->g) {
        /* handles do not remember their device: the JVM thread that called must come back on the device it was on, or its
         * single-GPU plugin handles (created on device 0) would launch on the last device of the loop */
        int prev = -1;
        (void)jsdr_get_device(&prev);
        for (int d = 0; x->raw && d < x->ndev; d++) {
            int dev = d;
            if (jsdr_group_device(x->g, d, &dev, NULL, NULL) == JSDR_OK && jsdr_set_device(dev) == JSDR_OK && x->raw[d]) jsdr_free(x->raw[d]);
        }
        jsdr_group_destroy(x->g);
        if (prev >= 0) (void)jsdr_set_device(prev);
    }
    free(x->raw);
    free(x);
}

JNIEXPORT jlong JNICALL Java_com_ashbysoft_java_1sdr_HipNative_groupCreate(JNIEnv *e, jclass c, jint ndev, jint rate, jint samples,
                                                                          jint tuning, jint doFFT, jint doUp, jint totalStreams,
                                                                          jlong maxBatch, jint flags)
{
    if (ndev < 1 || totalStreams < ndev || maxBatch < samples) {
        throw_new(e, "java/lang/IllegalArgumentException", "groupCreate: ndev >= 1, totalStreams >= ndev, maxBatch >= samples");
        return 0;
    }
    group_ctx *x = (group_ctx *)calloc(1, sizeof(*x));
    if (!x) {
        fail_msg(e, "groupCreate: out of memory");
        return 0;
    }
    x->ndev = ndev;
    x->total = totalStreams;
    x->per_dev = totalStreams / ndev;
    x->max_batch = maxBatch;
    if (jsdr_group_create(&x->g, ndev, NULL, rate, samples, tuning, doFFT, doUp, totalStreams, maxBatch, flags) != JSDR_OK ||
        jsdr_group_info(x->g, NULL, NULL, &x->slot_bytes, NULL) != JSDR_OK) {
        fail(e);
        x->g = NULL;
        group_free(x);
        return 0;
    }
    x->raw = (int16_t **)calloc((size_t)ndev, sizeof(*x->raw));
    int prev = -1;
    (void)jsdr_get_device(&prev);
    for (int d = 0; x->raw && d < ndev; d++) {
        int dev = d;
        if (jsdr_group_device(x->g, d, &dev, NULL, NULL) != JSDR_OK || jsdr_set_device(dev) != JSDR_OK ||
            jsdr_malloc((void **)&x->raw[d], (size_t)x->per_dev * (size_t)maxBatch * 4) != JSDR_OK) {
            fail(e);
            if (prev >= 0) (void)jsdr_set_device(prev);
            group_free(x);
            return 0;
        }
    }
    if (prev >= 0) (void)jsdr_set_device(prev);  /* the calling JVM thread stays on the device it came with */
    if (!x->raw) {
        fail_msg(e, "groupCreate: out of memory");
        group_free(x);
        return 0;
    }
    return (jlong)(intptr_t)x;
}

JNIEXPORT void JNICALL Java_com_ashbysoft_java_1sdr_HipNative_groupDestroy(JNIEnv *e, jclass c, jlong h)
{
    group_free(CTX(group_ctx, h));
}

JNIEXPORT void JNICALL Java_com_ashbysoft_java_1sdr_HipNative_groupInfo(JNIEnv *e, jclass c, jlong h, jlongArray info7)
{
    group_ctx *x = CTX(group_ctx, h);
    if (null_handle(e, x, "groupInfo") || short_array(e, "groupInfo", (*e)->GetArrayLength(e, info7), 7)) return;
    int nd = 0, per = 0, ver = 0, slot_bits = 0, nfec_max = 0;
    int64_t sb = 0, bits_off = 0, fec_off = 0;
    jsdr_bpsk *dem = NULL;
    if (jsdr_group_info(x->g, &nd, &per, &sb, &ver) != JSDR_OK || jsdr_group_device(x->g, 0, NULL, &dem, NULL) != JSDR_OK ||
        jsdr_bpsk_slot_info(dem, &sb, &bits_off, &fec_off, &slot_bits, &nfec_max) != JSDR_OK) {
        fail(e);
        return;
    }
    const jlong v[7] = {nd, per, sb, ver, bits_off, fec_off, nfec_max};
    (*e)->SetLongArrayRegion(e, info7, 0, 7, v);
}

JNIEXPORT jlong JNICALL Java_com_ashbysoft_java_1sdr_HipNative_groupLoadRecordings(JNIEnv *e, jclass c, jlong h, jobjectArray paths,
                                                                                  jint channels, jint rate, jlong firstFrame,
                                                                                  jlong nframes)
{
    group_ctx *x = CTX(group_ctx, h);
    if (null_handle(e, x, "groupLoadRecordings") || bad_length(e, "groupLoadRecordings: paths", (*e)->GetArrayLength(e, paths), x->total))
        return -1;
    if (nframes < 0 || nframes > x->max_batch) {
        throw_new(e, "java/lang/IllegalArgumentException", "groupLoadRecordings: more frames than the group's maxBatch");
        return -1;
    }
    const char **cp = (const char **)calloc((size_t)x->per_dev, sizeof(*cp));
    jstring *js = (jstring *)calloc((size_t)x->per_dev, sizeof(*js));
    int64_t *got = (int64_t *)calloc((size_t)x->per_dev, sizeof(*got));
    jlong sum = -1;
    if (cp && js && got) {
        int prev = -1;
        (void)jsdr_get_device(&prev);
        sum = 0;
        for (int d = 0; d < x->ndev && sum >= 0; d++) {
            int dev = d, ok = 1;
            for (int s = 0; s < x->per_dev; s++) {  /* this device's shard of the global stream ids: contiguous */
                js[s] = (jstring)(*e)->GetObjectArrayElement(e, paths, d * x->per_dev + s);
                cp[s] = js[s] ? (*e)->GetStringUTFChars(e, js[s], NULL) : NULL;
                if (!cp[s]) ok = 0;
            }
            if (ok && (jsdr_group_device(x->g, d, &dev, NULL, NULL) != JSDR_OK || jsdr_set_device(dev) != JSDR_OK ||
                       jsdr_recordings_load(cp, x->per_dev, channels, rate, firstFrame, nframes, x->raw[d], 2 * x->max_batch, got, NULL) != JSDR_OK))
                ok = -1;
            for (int s = 0; s < x->per_dev; s++) {
                if (cp[s]) (*e)->ReleaseStringUTFChars(e, js[s], cp[s]);
                if (js[s]) (*e)->DeleteLocalRef(e, js[s]);
                if (ok > 0) sum += got[s];
            }
            if (ok <= 0) {
                if (ok < 0) fail(e);
                else throw_new(e, "java/lang/IllegalArgumentException", "groupLoadRecordings: null path");
                sum = -1;
            }
        }
        if (prev >= 0) (void)jsdr_set_device(prev);
    } else {
        fail_msg(e, "groupLoadRecordings: out of memory");
    }
    free(cp);
    free(js);
    free(got);
    return sum;
}

JNIEXPORT void JNICALL Java_com_ashbysoft_java_1sdr_HipNative_groupBatch(JNIEnv *e, jclass c, jlong h, jlong nsamples, jint ic, jint qc)
{
    group_ctx *x = CTX(group_ctx, h);
    if (null_handle(e, x, "groupBatch")) return;
    if (nsamples < 0 || nsamples > x->max_batch) {
        throw_new(e, "java/lang/IllegalArgumentException", "groupBatch: more samples than the group's maxBatch");
        return;
    }
    if (jsdr_group_batch_i16(x->g, (const int16_t *const *)x->raw, 2 * x->max_batch, nsamples, ic, qc, NULL) != JSDR_OK) fail(e);
}

JNIEXPORT void JNICALL Java_com_ashbysoft_java_1sdr_HipNative_groupSync(JNIEnv *e, jclass c, jlong h)
{
    group_ctx *x = CTX(group_ctx, h);
    if (null_handle(e, x, "groupSync")) return;
    if (jsdr_group_sync(x->g) != JSDR_OK) fail(e);
}

JNIEXPORT void JNICALL Java_com_ashbysoft_java_1sdr_HipNative_groupReadSlot(JNIEnv *e, jclass c, jlong h, jint stream, jbyteArray slot)
{
    group_ctx *x = CTX(group_ctx, h);
    if (null_handle(e, x, "groupReadSlot") || short_array(e, "groupReadSlot", (*e)->GetArrayLength(e, slot), (jsize)x->slot_bytes)) return;
    uint8_t *buf = (uint8_t *)malloc((size_t)x->slot_bytes);
    if (!buf) {
        fail_msg(e, "groupReadSlot: out of memory");
        return;
    }
    if (jsdr_group_read_slot(x->g, 0, stream, buf) != JSDR_OK)
        fail(e);
    else
        (*e)->SetByteArrayRegion(e, slot, 0, (jsize)x->slot_bytes, (const jbyte *)buf);
    free(buf);
}
