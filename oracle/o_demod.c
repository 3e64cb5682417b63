/*
 * o_demod.c -- oracle: the demod.java AM/FM chain, demod.java:341-483.
 * TEST INFRASTRUCTURE (see jsdr_oracle.h).  Build with -ffp-contract=off: every float operation rounds to
 * float separately, as Java's do (x86-64 evaluates float expressions in float).
 */
#include "jsdr_oracle.h"
#include <math.h>
#include <string.h>

#define JO_PI 3.14159265358979323846
/* Math.sin/cos/sqrt on a double argument: long double, rounded once (see o_fir_phase.c) */
static double jsin(double x) { return (double)sinl((long double)x); }
static double jcos(double x) { return (double)cosl((long double)x); }

static int java_f2i(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}
/* Math.max(float,float): NaN if either is NaN */
static float jmaxf(float a, float b)
{
    if (a != a || b != b) return NAN;
    return a > b ? a : b;
}

/* demod.java:75-80,219-241: fir = new float[42], wfir = new float[21] (zeros), car/phi/li/lq = 0; the
 * constructor reads mode / fir-enable / agc-enable and the filter points from the config */
void jo_demod_init(jo_demod_t *d, int rate)
{
    memset(d, 0, sizeof(*d));
    d->rate = rate;
    d->flo = (-2147483647 - 1);
    d->fhi = 2147483647;
    d->fof = 0; /* Java field default; weights() sets len-2 */
}

/* demod.java:341-375 weights() */
void jo_demod_weights(jo_demod_t *d)
{
    const int len = 21;
    if ((-2147483647 - 1) == d->flo) {
        for (int i = 0; i < len; i++) d->wfir[i] = 0;
        d->wfir[(len - 1) / 2] = 1;
    } else {
        float rate = (float)d->rate;
        float nlo = (float)d->flo / rate;
        float nhi = (float)d->fhi / rate;
        int ord = len - 1;
        for (int n = 0; n < len; n++) {
            if (n == ord / 2) {
                d->wfir[n] = 2.0f * (nhi - nlo);
            } else {
                d->wfir[n] = (float)((jsin(2 * JO_PI * nhi * (double)(n - ord / 2)) / (JO_PI * (double)(n - ord / 2)))
                                   - (jsin(2 * JO_PI * nlo * (double)(n - ord / 2)) / (JO_PI * (double)(n - ord / 2))));
            }
            d->wfir[n] *= (float)(0.54 - 0.46 * jcos(2 * JO_PI * (double)n / (double)ord));
        }
        d->phi = (float)(2 * JO_PI * nlo);
        d->car = 0.0f;
    }
    for (int i = 0; i < 42; i++) d->fir[i] = 0.0f;
    d->fof = 42 - 2;
}

/* demod.java:300-312 filterMove(lo, hi): the only caller of weights() */
int jo_demod_filter_move(jo_demod_t *d, int lo, int hi)
{
    /* Java int arithmetic wraps */
    lo = (int)((unsigned)lo + (unsigned)d->flo);
    hi = (int)((unsigned)hi + (unsigned)d->fhi);
    if (lo < hi && lo > (-d->rate / 2) && hi < d->rate / 2) {
        d->flo = lo;
        d->fhi = hi;
        jo_demod_weights(d);
        return 1;
    }
    return 0;
}

/* demod.java:378-396 filter() */
static int demod_filter(const float in[2], float out[2], float *buf, const float *w, int o)
{
    buf[o] = in[0];
    buf[o + 1] = in[1];
    float oi = 0, oq = 0;
    for (int i = 0; i < 42; i += 2) {
        int ti = (o + i) % 42;
        oi = oi + buf[ti] * w[i / 2];
        oq = oq + buf[ti + 1] * w[i / 2];
    }
    out[0] = oi;
    out[1] = oq;
    o = o - 2;
    if (o < 0) o = 42 - 2;
    return o;
}

/* demod.java:398-483 receive(buf): len floats in (I,Q interleaved), len int16 out (L,R per sample, :473-478) */
void jo_demod_receive(jo_demod_t *d, const float *buf, int len, int16_t *out)
{
    float *sam = d->sam;
    d->max = 0;
    d->avg = 0;
    float fmgain = (float)d->rate / (JO_MODE_NFM == d->mode ? 5000.0f : 75000.0f);
    for (int s = 0; s < len; s += 2) {
        sam[s] = buf[s];
        sam[s + 1] = buf[s + 1];
        if (d->dofir) {
            float fs[2] = {sam[s], sam[s + 1]};
            float os[2] = {0, 0};
            d->fof = demod_filter(fs, os, d->fir, d->wfir, d->fof);
            sam[s] = os[0];
            sam[s + 1] = os[1];
        }
        if (d->dodwn) {
            float ci = (float)jcos((double)d->car);
            float cq = (float)jsin((double)d->car);
            d->car -= d->phi;
            if (d->car < 0.0f) d->car += (float)(2 * JO_PI);
            float fs[2] = {sam[s], sam[s + 1]};
            sam[s] = (fs[0] * ci - fs[1] * cq);
            sam[s + 1] = (fs[0] * cq + fs[1] * ci);
        }
        if (JO_MODE_OFF == d->mode) {
            sam[s] = sam[s + 1] = 0;
        } else if (JO_MODE_RAW == d->mode) {
            ;
        } else if (JO_MODE_AM == d->mode) {
            sam[s] = (float)sqrt((double)(sam[s] * sam[s] + sam[s + 1] * sam[s + 1]));
            d->avg = ((float)(s / 2) * d->avg + sam[s]) / (float)(s / 2 + 1);
        } else if (JO_MODE_NFM == d->mode || JO_MODE_WFM == d->mode) {
            float v = ((d->li * sam[s + 1]) - (d->lq * sam[s])) * fmgain;
            d->li = sam[s];
            d->lq = sam[s + 1];
            sam[s] = v;
        }
        d->max = jmaxf(d->max, (float)fabs((double)sam[s]));
    }
    if (JO_MODE_AM == d->mode) d->max -= d->avg;
    for (int s = 0; s < len; s += 2) {
        sam[s] = (JO_MODE_AM == d->mode ? sam[s] - d->avg : sam[s]) * (d->doagc ? 1.0f / d->max : 1.0f);
        int16_t v = (int16_t)java_f2i(sam[s] * (float)32767);
        out[s] = v;
        out[s + 1] = v;
    }
}
