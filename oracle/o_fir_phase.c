/*
 * o_fir_phase.c -- oracle: fir.java arithmetic and phase.java reductions.
 * TEST INFRASTRUCTURE (see jsdr_oracle.h).  Build with -ffp-contract=off.
 */
#include "jsdr_oracle.h"
#include <math.h>
#include <string.h>

#define JO_PI 3.14159265358979323846
/* Math.sin/cos on a double argument: evaluated in long double and rounded once, so the value does not depend on
 * how a compiler lowers sin()/cos() (hipcc's host compiler rewrites pairs into sincos / vector calls) */
static double jsin(double x) { return (double)sinl((long double)x); }
static double jcos(double x) { return (double)cosl((long double)x); }

/* Java (int) of a double: truncate toward zero, saturate, NaN -> 0 */
static int java_d2i(double v)
{
    if (v != v) return 0;
    if (v >= 2147483647.0) return 2147483647;
    if (v <= -2147483648.0) return (-2147483647 - 1);
    return (int)v;
}

/* fir.java:30-32 */
void jo_fir_init(jo_fir_t *f)
{
    memset(f, 0, sizeof(*f));
    f->fof = 20;
}

/* fir.java:169-195 weights(f1,f2): 21-tap windowed-sinc band-pass (Hamming), or the
 * all-pass unit impulse when both arguments are Integer.MIN_VALUE; clears delay line. */
void jo_fir_weights(jo_fir_t *f, int f1, int f2, float sample_rate)
{
    const int len = 21;
    if (f1 == (-2147483647 - 1) && f2 == (-2147483647 - 1)) {
        for (int i = 0; i < len; i++) f->wfir[i] = 0;
        f->wfir[(len - 1) / 2] = 1;
    } else {
        double df1 = (double)f1 / sample_rate; /* float promoted to double */
        double df2 = (double)f2 / sample_rate;
        int ord = len - 1;
        for (int n = 0; n < len; n++) {
            if (n == ord / 2) {
                f->wfir[n] = 2 * (df2 - df1);
            } else {
                f->wfir[n] = (jsin(2 * JO_PI * df2 * (n - ord / 2)) / (JO_PI * (n - ord / 2)))
                           - (jsin(2 * JO_PI * df1 * (n - ord / 2)) / (JO_PI * (n - ord / 2)));
            }
            f->wfir[n] = f->wfir[n] * (0.54 - 0.46 * jcos(2 * JO_PI * n / ord));
        }
    }
    for (int i = 0; i < len; i++) f->fir[i] = 0;
    f->fof = len - 1;
}

/* fir.java:198-211 filter(in): ring write, newest-first MAC in double, (int) truncation */
int jo_fir_filter(jo_fir_t *f, int in)
{
    const int len = 21;
    f->fir[f->fof] = in;
    double o = 0;
    for (int i = 0; i < len; i++) {
        int ti = (f->fof + i) % len;
        o = o + f->fir[ti] * f->wfir[i];
    }
    f->fof = f->fof - 1;
    if (f->fof < 0) f->fof = len - 1;
    return java_d2i(o);
}

/* fir.java:221-228 complex_gen: NCO sample, counter wraps at (int)rate */
void jo_fir_complex_gen(int sig[2], int wav[2], float sample_rate)
{
    double w = (2 * JO_PI * wav[0] * wav[1]) / sample_rate;
    sig[0] = java_d2i(jcos(w) * 4096);
    sig[1] = java_d2i(jsin(w) * 4096);
    wav[1] += 1;
    if (wav[1] >= (int)sample_rate) wav[1] = 0;
}

/* fir.java:214-218 complex_mod: int32 complex multiply (wrapping) */
void jo_fir_complex_mod(const int s1[2], const int s2[2], int out[2])
{
    uint32_t a = (uint32_t)s1[0], b = (uint32_t)s1[1], c = (uint32_t)s2[0], d = (uint32_t)s2[1];
    out[0] = (int32_t)(a * c - b * d);
    out[1] = (int32_t)(a * d + b * c);
}

/* phase.java:75-80 */
float jo_phase_maxabs(const float *dpy, int len)
{
    float max = -1;
    for (int s = 0; s < len; s++) {
        float a = (float)fabs((double)dpy[s]);
        if (max < a) max = a;
    }
    return max;
}

/* phase.java:81-116: per-pixel-column running means of I and Q.  `pos += step` is a
 * float accumulation; a column closes when (int)pos exceeds the last pixel.          */
int jo_phase_columns(const float *dpy, int len, int bx, int *pix_out, float *avgi_out, float *avgq_out)
{
    float step = (float)(bx * 2) / (float)len;
    float pos = 0;
    int lpix = 0;
    float avgi = 0, avgq = 0;
    int acnt = 0;
    int ncol = 0;
    for (int s = 0; s < len; s += 2) {
        avgi += dpy[s];
        avgq += dpy[s + 1];
        acnt += 1;
        pos += step;
        int pix = (int)pos;
        if (pix > lpix) {
            avgi = avgi / acnt;
            avgq = avgq / acnt;
            pix_out[ncol] = pix;
            avgi_out[ncol] = avgi;
            avgq_out[ncol] = avgq;
            ncol++;
            lpix = pix;
            acnt = 0;
            avgi = avgq = 0;
        }
    }
    return ncol;
}

/* FUNcubeBPSKDemod.java:466-492 (RxDownSample) as a stand-alone operator, generalised over tap count and decimation --
 * BASELINE config 3's "FIR + decimate over an IQ batch".  The reference's own state machine, not a closed form: a ring of
 * `ntaps` (I,Q) doubles written at dsPos, dsPos walking DOWN and wrapping; every `decim`-th sample the sum over ring
 * slots n = 0..ntaps-1 of dsBuf[(n+dsPos)%ntaps]*taps[n] (newest sample first, product and sum rounded separately),
 * times `scale` (HOWARD_FUDGE_FACTOR at :487).  The samples are JavaAudio's floats (JavaAudio.java:276-293) widened to
 * double, as RxMixTuner hands them over when the tuner is off (:395-396); the ring starts cleared (:467).
 * Returns the number of (fi,fq) pairs written to out (interleaved doubles). */
int64_t jo_fir_decimate(const int16_t *raw, int64_t nsamples, const double *taps, int ntaps, int decim, double scale,
                        double *out)
{
    double ds_i[128], ds_q[128];
    if (ntaps < 1 || ntaps > 128 || decim < 1) return -1;
    for (int n = 0; n < ntaps; n++) ds_i[n] = ds_q[n] = 0.0;
    const float fmax = (float)32767;
    int dsPos = ntaps - 1, dsCnt = 0;
    int64_t no = 0;
    for (int64_t t = 0; t < nsamples; t++) {
        ds_i[dsPos] = (double)((float)raw[2 * t] / fmax);
        ds_q[dsPos] = (double)((float)raw[2 * t + 1] / fmax);
        if (++dsCnt >= decim) {
            double fi = 0.0, fq = 0.0;
            for (int n = 0; n < ntaps; n++) {
                int dsi = (n + dsPos) % ntaps;
                fi += ds_i[dsi] * taps[n];
                fq += ds_q[dsi] * taps[n];
            }
            dsCnt = 0;
            out[2 * no] = fi * scale;
            out[2 * no + 1] = fq * scale;
            no++;
        }
        dsPos--;
        if (dsPos < 0) dsPos = ntaps - 1;
    }
    return no;
}
