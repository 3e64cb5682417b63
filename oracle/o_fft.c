/*
 * o_fft.c -- oracle: sample conversion, FFT stand-ins, fft.java receive().
 * TEST INFRASTRUCTURE (see jsdr_oracle.h).  Build with -ffp-contract=off.
 */
#include "jsdr_oracle.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

#define JO_PI 3.14159265358979323846 /* == java.lang.Math.PI */

/* JavaAudio.java:276-293.  `short s = cnv.getShort(); s += (short)m_ic;` wraps mod 2^16,
 * then `(float)s/(float)Short.MAX_VALUE` is a float division.  Mono => Q = 0.        */
void jo_convert_i16(const int16_t *raw, int nframes, int chns, int ic, int qc, float *out)
{
    const float fmax = (float)32767;
    for (int f = 0; f < nframes; f++) {
        int16_t s = raw[f * chns];
        s = (int16_t)(s + (int16_t)ic);
        out[2 * f] = (float)s / fmax;
        if (chns > 1) {
            s = raw[f * chns + 1];
            s = (int16_t)(s + (int16_t)qc);
            out[2 * f + 1] = (float)s / fmax;
        } else {
            out[2 * f + 1] = 0;
        }
    }
}

static unsigned bitrev(unsigned x, int bits)
{
    unsigned r = 0;
    for (int i = 0; i < bits; i++) {
        r = (r << 1) | (x & 1u);
        x >>= 1;
    }
    return r;
}

static int ilog2(int n)
{
    int b = 0;
    while ((1 << b) < n) b++;
    return b;
}

void jo_fft_twiddles_f64(double *w, int n)
{
    /* evaluated in long double and rounded once: the result does not depend on whether a compiler turns
     * sin()/cos() pairs into sincos() or a vector-libm call (hipcc's host compiler does, and that differs
     * from scalar glibc in the last bit for 1 of the 2048 entries at n=4096) */
    for (int k = 0; k < n / 2; k++) {
        long double ang = 2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)n;
        w[2 * k] = (double)cosl(ang);
        w[2 * k + 1] = (double)(-sinl(ang));
    }
    /* exact values on the axes so that trivial twiddles stay trivial */
    w[0] = 1.0;
    w[1] = -0.0;
    if (n >= 4) {
        w[2 * (n / 4)] = 0.0;
        w[2 * (n / 4) + 1] = -1.0;
    }
}

/* Stand-in for JTransforms DoubleFFT_1D.complexForward / complexInverse(a,true)
 * (call sites FUNcubeBPSKDemod.java:422-423,459).  Radix-2 decimation in time:
 * bit-reversal permutation, then log2(n) stages; butterfly
 *     t = w*b  (tr = wr*br - wi*bi ; ti = wr*bi + wi*br, every product and sum rounded)
 *     a' = a + t ; b' = a - t
 * The HIP double FFT (csrc/bpsk_fft.hip) performs the same butterflies on the same
 * twiddle table, so both are bit-identical.  PARITY UNPINNED vs JTransforms.          */
/* ---- non power-of-two frames (n = 2^a 3^b 5^c, e.g. the reference's default 9600 = blen/size):
 * Stockham autosort passes, radices 4,4,..,(2),3..,5..,7.. in this order (frames above 9600 samples: one radix-2 pass
 * first, see jo_fft_mixed_radices); pass with radix r and P = product of the
 * earlier radices takes butterfly b (k = b mod P) from in[b + j*n/r], j < r, multiplies input j >= 1 by the
 * table entry T[k*j] = exp(-2 pi i k j/(P r)) (cosl/sinl rounded once; exact on the axes), applies the fixed-order
 * r-point DFT below and stores output q at out[(b-k)*r + k + q*P].  The inverse transform is
 * conj o forward o conj (conjugation is exact), scaled by 1/n afterwards.  The HIP kernel (csrc/bpsk_fftm.hip)
 * performs the same operations on the same tables.  PARITY UNPINNED vs JTransforms.                            */
typedef struct { double x, y; } cd_t;
static cd_t cd(double x, double y) { cd_t r = {x, y}; return r; }
static cd_t cadd(cd_t a, cd_t b) { return cd(a.x + b.x, a.y + b.y); }
static cd_t csub(cd_t a, cd_t b) { return cd(a.x - b.x, a.y - b.y); }
static cd_t cmul(cd_t u, cd_t w) { return cd(u.x * w.x - u.y * w.y, u.x * w.y + u.y * w.x); }

static void dft_r(cd_t *v, int r)
{
    if (r == 2) {
        cd_t a = cadd(v[0], v[1]), b = csub(v[0], v[1]);
        v[0] = a;
        v[1] = b;
    } else if (r == 4) {
        cd_t a = cadd(v[0], v[2]), b = csub(v[0], v[2]), c = cadd(v[1], v[3]), d = csub(v[1], v[3]);
        v[0] = cadd(a, c);
        v[2] = csub(a, c);
        v[1] = cd(b.x + d.y, b.y - d.x); /* b - i d */
        v[3] = cd(b.x - d.y, b.y + d.x); /* b + i d */
    } else if (r == 3) {
        const double S = 0.86602540378443864676; /* sin(2 pi/3) */
        cd_t t1 = cadd(v[1], v[2]);
        cd_t t2 = cd(v[0].x - 0.5 * t1.x, v[0].y - 0.5 * t1.y);
        cd_t d = csub(v[1], v[2]);
        cd_t t3 = cd(S * d.x, S * d.y);
        v[0] = cadd(v[0], t1);
        v[1] = cd(t2.x + t3.y, t2.y - t3.x);
        v[2] = cd(t2.x - t3.y, t2.y + t3.x);
    } else if (r == 7) {
        /* 7 (round 4: 44.1 kHz sound cards give n = rate/10 = 4410 = 2 3^2 5 7^2, the reference's own sine4410.wav):
         * the radix-5 scheme one size up.  a_k = v[k] + v[7-k], b_k = v[k] - v[7-k];
         *   out[j], out[7-j] = m_j -/+ i n_j,  m_j = ((x0 + C(j) a1) + C(2j) a2) + C(3j) a3,  n_j = (S(j) b1 + S(2j) b2) + S(3j) b3
         * with C(k) = cos(2 pi k/7), S(k) = sin(2 pi k/7), indices mod 7 (C(7-k) = C(k), S(7-k) = -S(k)); sums left to right */
        const double C1 = 0.62348980185873353053, C2 = -0.22252093395631440429, C3 = -0.90096886790241912624;
        const double S1 = 0.78183148246802980871, S2 = 0.97492791218182360702, S3 = 0.43388373911755812048;
        cd_t a1 = cadd(v[1], v[6]), a2 = cadd(v[2], v[5]), a3 = cadd(v[3], v[4]);
        cd_t b1 = csub(v[1], v[6]), b2 = csub(v[2], v[5]), b3 = csub(v[3], v[4]);
        cd_t x0 = v[0];
        cd_t m1 = cd(((x0.x + C1 * a1.x) + C2 * a2.x) + C3 * a3.x, ((x0.y + C1 * a1.y) + C2 * a2.y) + C3 * a3.y);
        cd_t m2 = cd(((x0.x + C2 * a1.x) + C3 * a2.x) + C1 * a3.x, ((x0.y + C2 * a1.y) + C3 * a2.y) + C1 * a3.y);
        cd_t m3 = cd(((x0.x + C3 * a1.x) + C1 * a2.x) + C2 * a3.x, ((x0.y + C3 * a1.y) + C1 * a2.y) + C2 * a3.y);
        cd_t n1 = cd((S1 * b1.x + S2 * b2.x) + S3 * b3.x, (S1 * b1.y + S2 * b2.y) + S3 * b3.y);
        cd_t n2 = cd((S2 * b1.x - S3 * b2.x) - S1 * b3.x, (S2 * b1.y - S3 * b2.y) - S1 * b3.y);
        cd_t n3 = cd((S3 * b1.x - S1 * b2.x) + S2 * b3.x, (S3 * b1.y - S1 * b2.y) + S2 * b3.y);
        v[0] = cd(((x0.x + a1.x) + a2.x) + a3.x, ((x0.y + a1.y) + a2.y) + a3.y);
        v[1] = cd(m1.x + n1.y, m1.y - n1.x);
        v[6] = cd(m1.x - n1.y, m1.y + n1.x);
        v[2] = cd(m2.x + n2.y, m2.y - n2.x);
        v[5] = cd(m2.x - n2.y, m2.y + n2.x);
        v[3] = cd(m3.x + n3.y, m3.y - n3.x);
        v[4] = cd(m3.x - n3.y, m3.y + n3.x);
    } else { /* 5 */
        const double C1 = 0.30901699437494742410, C2 = -0.80901699437494742410; /* cos(2 pi/5), cos(4 pi/5) */
        const double S1 = 0.95105651629515357212, S2 = 0.58778525229247312917; /* sin(2 pi/5), sin(4 pi/5) */
        cd_t a1 = cadd(v[1], v[4]), a2 = cadd(v[2], v[3]), b1 = csub(v[1], v[4]), b2 = csub(v[2], v[3]);
        cd_t x0 = v[0];
        cd_t m1 = cd((x0.x + C1 * a1.x) + C2 * a2.x, (x0.y + C1 * a1.y) + C2 * a2.y);
        cd_t m2 = cd((x0.x + C2 * a1.x) + C1 * a2.x, (x0.y + C2 * a1.y) + C1 * a2.y);
        cd_t n1 = cd(S1 * b1.x + S2 * b2.x, S1 * b1.y + S2 * b2.y);
        cd_t n2 = cd(S2 * b1.x - S1 * b2.x, S2 * b1.y - S1 * b2.y);
        v[0] = cd((x0.x + a1.x) + a2.x, (x0.y + a1.y) + a2.y);
        v[1] = cd(m1.x + n1.y, m1.y - n1.x);
        v[4] = cd(m1.x - n1.y, m1.y + n1.x);
        v[2] = cd(m2.x + n2.y, m2.y - n2.x);
        v[3] = cd(m2.x - n2.y, m2.y + n2.x);
    }
}

/* radix list for n = 2^a 3^b 5^c 7^d; returns the count (0: unsupported) */
int jo_fft_mixed_radices(int n, int *rad)
{
    int c = 0;
    if (n < 2) return 0;
    /* frames above 9600 samples (19200 = the FCD Pro+ default at 192 kHz, JavaAudio.java:58-59) start with ONE radix-2
     * pass: it splits the transform into two independent 9600-point halves (even and odd output bins), each of which
     * fits a workgroup's LDS on the GPU; the order is this definition's to choose (JTransforms' is unknowable here) */
    if (n > 9600 && n % 2 == 0) { rad[c++] = 2; n /= 2; }
    while (n % 4 == 0) { rad[c++] = 4; n /= 4; }
    if (n % 2 == 0) { rad[c++] = 2; n /= 2; }
    while (n % 3 == 0) { rad[c++] = 3; n /= 3; }
    while (n % 5 == 0) { rad[c++] = 5; n /= 5; }
    while (n % 7 == 0) { rad[c++] = 7; n /= 7; }
    /* round 5: any other prime factor, ascending (11.025 kHz cards give n = 1102 = 2 19 29; audio-rate is a free integer,
     * JavaAudio.java:49,58-59) -- a pass of that radix with the r-point DFT written out as its definition (below) */
    for (int p = 11; n > 1 && c < 30; p += 2) {
        if (p * p > n) p = n;
        while (n % p == 0 && c < 30) { rad[c++] = p; n /= p; }
    }
    return n == 1 ? c : 0;
}

/* T[m] = exp(-2 pi i m/len), m < len, long double + one rounding; exact on the axes */
void jo_fft_mixed_table(double *t, int len)
{
    for (int m = 0; m < len; m++) {
        long double ang = 2.0L * 3.14159265358979323846264338327950288L * (long double)m / (long double)len;
        t[2 * m] = (double)cosl(ang);
        t[2 * m + 1] = (double)(-sinl(ang));
    }
    t[0] = 1.0;
    t[1] = -0.0;
    if (len % 4 == 0) {
        t[2 * (len / 4)] = 0.0;
        t[2 * (len / 4) + 1] = -1.0;
        t[2 * (3 * len / 4)] = -0.0;
        t[2 * (3 * len / 4) + 1] = 1.0;
    }
    if (len % 2 == 0) {
        t[2 * (len / 2)] = -1.0;
        t[2 * (len / 2) + 1] = -0.0;
    }
}

static void fft_f64_mixed_forward(cd_t *a, int n)
{
    int rad[32];
    int np = jo_fft_mixed_radices(n, rad);
    cd_t *b = (cd_t *)malloc(sizeof(cd_t) * (size_t)n);
    cd_t *in = a, *out = b;
    int P = 1;
    for (int p = 0; p < np; p++) {
        const int r = rad[p], nb = n / r, len = P * r;
        double *t = (double *)malloc(sizeof(double) * 2 * (size_t)len);
        jo_fft_mixed_table(t, len);
        if (r > 7) {
            /* a prime radix above 7: the r-point DFT as its definition.  Inputs twiddled as in every pass, then
             *   out_q = v_0 + v_1 W[q mod r] + v_2 W[2q mod r] + ... + v_{r-1} W[(r-1)q mod r],  W[m] = exp(-2 pi i m/r)
             * (jo_fft_mixed_table(r): long double, one rounding), every product a full complex multiply (even by W[0]),
             * summed left to right.  O(r) per output: a correctness path for the sizes consumer cards produce. */
            double *wr = (double *)malloc(sizeof(double) * 2 * (size_t)r);
            cd_t *v = (cd_t *)malloc(sizeof(cd_t) * (size_t)r);
            jo_fft_mixed_table(wr, r);
            for (int bf = 0; bf < nb; bf++) {
                const int k = bf % P;
                for (int j = 0; j < r; j++) {
                    v[j] = in[bf + j * nb];
                    if (j >= 1 && P > 1) v[j] = cmul(v[j], cd(t[2 * (k * j)], t[2 * (k * j) + 1]));
                }
                for (int q = 0; q < r; q++) {
                    cd_t acc = v[0];
                    for (int j = 1; j < r; j++) {
                        const int m = (int)(((long long)j * q) % r);
                        acc = cadd(acc, cmul(v[j], cd(wr[2 * m], wr[2 * m + 1])));
                    }
                    out[(bf - k) * r + k + q * P] = acc;
                }
            }
            free(v);
            free(wr);
        } else
        for (int bf = 0; bf < nb; bf++) {
            const int k = bf % P;
            cd_t v[7];
            for (int j = 0; j < r; j++) {
                v[j] = in[bf + j * nb];
                if (j >= 1 && P > 1) v[j] = cmul(v[j], cd(t[2 * (k * j)], t[2 * (k * j) + 1]));
            }
            dft_r(v, r);
            for (int q = 0; q < r; q++) out[(bf - k) * r + k + q * P] = v[q];
        }
        free(t);
        P *= r;
        cd_t *tmp = in;
        in = out;
        out = tmp;
    }
    if (in != a) memcpy(a, in, sizeof(cd_t) * (size_t)n);
    free(b);
}

static void fft_f64_mixed(double *a, int n, int inverse, int scale)
{
    cd_t *c = (cd_t *)a;
    if (inverse)
        for (int i = 0; i < n; i++) c[i].y = -c[i].y;
    fft_f64_mixed_forward(c, n);
    if (inverse) {
        for (int i = 0; i < n; i++) c[i].y = -c[i].y;
        if (scale) {
            double norm = 1.0 / (double)n;
            for (int i = 0; i < 2 * n; i++) a[i] *= norm;
        }
    }
}

void jo_fft_f64(double *a, int n, int inverse, int scale)
{
    if (n & (n - 1)) {
        fft_f64_mixed(a, n, inverse, scale);
        return;
    }
    int bits = ilog2(n);
    double *w = (double *)malloc(sizeof(double) * (size_t)n);
    jo_fft_twiddles_f64(w, n);
    for (int i = 0; i < n; i++) {
        int j = (int)bitrev((unsigned)i, bits);
        if (j > i) {
            double tr = a[2 * i], ti = a[2 * i + 1];
            a[2 * i] = a[2 * j];
            a[2 * i + 1] = a[2 * j + 1];
            a[2 * j] = tr;
            a[2 * j + 1] = ti;
        }
    }
    for (int half = 1; half < n; half <<= 1) {
        int step = n / (2 * half);
        for (int base = 0; base < n; base += 2 * half) {
            for (int j = 0; j < half; j++) {
                double wr = w[2 * (j * step)];
                double wi = w[2 * (j * step) + 1];
                if (inverse) wi = -wi;
                int ia = base + j, ib = base + j + half;
                double br = a[2 * ib], bi = a[2 * ib + 1];
                double p1 = wr * br, p2 = wi * bi, p3 = wr * bi, p4 = wi * br;
                double tr = p1 - p2;
                double ti = p3 + p4;
                double ar = a[2 * ia], ai = a[2 * ia + 1];
                a[2 * ia] = ar + tr;
                a[2 * ia + 1] = ai + ti;
                a[2 * ib] = ar - tr;
                a[2 * ib + 1] = ai - ti;
            }
        }
    }
    if (inverse && scale) {
        double norm = 1.0 / (double)n;
        for (int i = 0; i < 2 * n; i++) a[i] *= norm;
    }
    free(w);
}

/* Stand-in for JTransforms FloatFFT_1D.complexForward (call site fft.java:194-195).
 * Same radix-2 network in float; twiddles rounded from double.  PARITY UNPINNED.    */
void jo_fft_f32(float *a, int n)
{
    int bits = ilog2(n);
    float *w = (float *)malloc(sizeof(float) * (size_t)n);
    for (int k = 0; k < n / 2; k++) {
        double ang = 2.0 * JO_PI * (double)k / (double)n;
        w[2 * k] = (float)cos(ang);
        w[2 * k + 1] = (float)(-sin(ang));
    }
    for (int i = 0; i < n; i++) {
        int j = (int)bitrev((unsigned)i, bits);
        if (j > i) {
            float tr = a[2 * i], ti = a[2 * i + 1];
            a[2 * i] = a[2 * j];
            a[2 * i + 1] = a[2 * j + 1];
            a[2 * j] = tr;
            a[2 * j + 1] = ti;
        }
    }
    for (int half = 1; half < n; half <<= 1) {
        int step = n / (2 * half);
        for (int base = 0; base < n; base += 2 * half) {
            for (int j = 0; j < half; j++) {
                float wr = w[2 * (j * step)], wi = w[2 * (j * step) + 1];
                int ia = base + j, ib = base + j + half;
                float br = a[2 * ib], bi = a[2 * ib + 1];
                float tr = wr * br - wi * bi;
                float ti = wr * bi + wi * br;
                float ar = a[2 * ia], ai = a[2 * ia + 1];
                a[2 * ia] = ar + tr;
                a[2 * ia + 1] = ai + ti;
                a[2 * ib] = ar - tr;
                a[2 * ib + 1] = ai - ti;
            }
        }
    }
    free(w);
}

/* exact reference DFT, X[k] = sum x[t] e^{-2 pi i k t / n}, accumulated in long double */
void jo_dft_exact(const float *in, int n, double *out)
{
    long double *c = (long double *)malloc(sizeof(long double) * (size_t)n);
    long double *s = (long double *)malloc(sizeof(long double) * (size_t)n);
    for (int k = 0; k < n; k++) {
        long double ang = 2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)n;
        c[k] = cosl(ang);
        s[k] = sinl(ang);
    }
    for (int k = 0; k < n; k++) {
        long double re = 0, im = 0;
        for (int t = 0; t < n; t++) {
            int idx = (int)(((long long)k * t) % n);
            long double xr = in[2 * t], xi = in[2 * t + 1];
            /* (xr + i xi)(c - i s) */
            re += xr * c[idx] + xi * s[idx];
            im += xi * c[idx] - xr * s[idx];
        }
        out[2 * k] = (double)re;
        out[2 * k + 1] = (double)im;
    }
    free(c);
    free(s);
}

/* fft.java:196-224 -- PSD in dBFS, first strictly-greater maximum, bin -> Hz in Java int
 * arithmetic (wrapping multiply, truncating divide).                                  */
void jo_fft_psd_from_spectrum(const float *dat, int n, int rate, float *psd)
{
    int datlen = 2 * n;
    float cf = 2.0f / (float)(datlen / 2);
    cf = cf * cf;
    float m = -FLT_MAX;
    int p = -1;
    for (int s = 0; s < datlen - 1; s += 2) {
        float pw = ((dat[s] * dat[s]) + (dat[s + 1] * dat[s + 1])) * cf;
        psd[s / 2] = 10.0f * (float)log10((double)pw);
        if (m < psd[s / 2]) {
            m = psd[s / 2];
            p = s;
        }
    }
    if (p < datlen / 2) {
        p = (int32_t)((uint32_t)p * (uint32_t)rate) / datlen;
    } else {
        p -= datlen;
        p = (int32_t)((uint32_t)p * (uint32_t)rate) / datlen;
    }
    psd[n] = (float)p;
    psd[n + 1] = m;
}

/* fft.java:190-228 receive(): copy, complex forward FFT, PSD rule.                 */
void jo_fft_receive(const float *buf, int n, int rate, float *psd)
{
    float *dat = (float *)malloc(sizeof(float) * 2 * (size_t)n);
    memcpy(dat, buf, sizeof(float) * 2 * (size_t)n);
    if ((n & (n - 1)) == 0) {
        jo_fft_f32(dat, n);
    } else {
        /* non power-of-two frame (the reference's default n=9600): JTransforms uses a mixed-radix plan; the
         * stand-in is the exact DFT rounded to float (parity for this path is the 1e-5 tolerance anyway) */
        double *x = (double *)malloc(sizeof(double) * 2 * (size_t)n);
        jo_dft_exact(buf, n, x);
        for (int i = 0; i < 2 * n; i++) dat[i] = (float)x[i];
        free(x);
    }
    jo_fft_psd_from_spectrum(dat, n, rate, psd);
    free(dat);
}

void jo_bench_fft(const int16_t *raw, int64_t nframes, int n, int rate, float *psd_last)
{
    float *buf = (float *)malloc(sizeof(float) * 2 * (size_t)n);
    float *psd = (float *)malloc(sizeof(float) * ((size_t)n + 2));
    for (int64_t f = 0; f < nframes; f++) {
        jo_convert_i16(raw + f * 2 * n, n, 2, 0, 0, buf);
        jo_fft_receive(buf, n, rate, psd);
    }
    if (psd_last) memcpy(psd_last, psd, sizeof(float) * ((size_t)n + 2));
    free(buf);
    free(psd);
}
