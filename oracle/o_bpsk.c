/*
 * o_bpsk.c -- oracle: FUNcubeBPSKDemod.java receive() chain, one stream per object.
 * TEST INFRASTRUCTURE (see jsdr_oracle.h).  Build with -ffp-contract=off.
 *
 * Follows FUNcubeBPSKDemod.java:26-96 (constants), :159-162 (sin/cos tables),
 * :192-209 (setup), :357-595 (receive .. RxDemodulate).  Debug rings that only feed the
 * painter (tuned/downSmpl/demodIQ/demodBits/demodRe) are replaced by optional traces.
 */
#include "jsdr_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define JO_PI 3.14159265358979323846

enum {
    DS_N = 27, DM_N = 65, SYNC_N = 65, FEC_BITS = 5200, FEC_BLOCK = 256,
    DOWN_RATE = 9600, BIT_RATE = 1200, SPB = DOWN_RATE / BIT_RATE, SINCOS = 256
};

/* FUNcubeBPSKDemod.java:27-55 -- taps are float literals widened to double; the filter
 * is symmetric, first 14 listed.                                                       */
static const float ds_half[14] = {
    -6.103515625000e-004F, -1.220703125000e-004F, +2.380371093750e-003F, +6.164550781250e-003F,
    +7.324218750000e-003F, +7.629394531250e-004F, -1.464843750000e-002F, -3.112792968750e-002F,
    -3.225708007813e-002F, -1.617431640625e-003F, +6.463623046875e-002F, +1.502380371094e-001F,
    +2.231445312500e-001F, +2.518310546875e-001F
};
/* FUNcubeBPSKDemod.java:58-77 -- 65-tap matched filter, symmetric, first 33 listed; the
 * reference stores it twice back to back (130 entries) to avoid a modulo.              */
static const float dm_half[33] = {
    -0.0101130691F, -0.0086975143F, -0.0038246093F, +0.0033563764F, +0.0107237026F, +0.0157790936F,
    +0.0164594107F, +0.0119213911F, +0.0030315224F, -0.0076488191F, -0.0164594107F, -0.0197184277F,
    -0.0150109226F, -0.0023082460F, +0.0154712381F, +0.0327423589F, +0.0424493086F, +0.0379940454F,
    +0.0154712381F, -0.0243701991F, -0.0750320094F, -0.1244834076F, -0.1568500423F, -0.1553748911F,
    -0.1061032953F, -0.0015013786F, +0.1568500423F, +0.3572048240F, +0.5786381191F, +0.7940228249F,
    +0.9744923010F, +1.0945250059F, +1.1366117829F
};

static double dsFilter[DS_N];
static double dmFilter[2 * DM_N];
static int8_t SYNC_VECTOR[SYNC_N];
static int consts_ready = 0;

static const double VCO_PHASE_INC = 2.0 * JO_PI * 1200.0 / (double)DOWN_RATE;
static const double BIT_SMOOTH1 = 1.0 / 200.0;
static const double BIT_SMOOTH2 = 1.0 / 800.0;
static const double BIT_PHASE_INC = 1.0 / (double)DOWN_RATE;
static const double BIT_TIME = 1.0 / (double)BIT_RATE;
/* :399-402 -- computed in float, widened */
static const double CFREQ_INV_AVERAGE_FACTOR = 1.0F - (2.0F / (1 + 1));
static const double CFREQ_AVERAGE_FACTOR = 2.0F / (1 + 1);
static const double PSD_INV_AVERAGE_FACTOR = 1.0F - (2.0F / (10 + 1));
static const double PSD_AVERAGE_FACTOR = 2.0F / (10 + 1);
static const int dmHalfTable[8] = {4, 5, 6, 7, 0, 1, 2, 3};

static void build_consts(void)
{
    if (consts_ready) return;
    for (int i = 0; i < 14; i++) {
        dsFilter[i] = (double)ds_half[i];
        dsFilter[DS_N - 1 - i] = (double)ds_half[i];
    }
    for (int i = 0; i < 33; i++) {
        dmFilter[i] = (double)dm_half[i];
        dmFilter[DM_N - 1 - i] = (double)dm_half[i];
    }
    for (int i = 0; i < DM_N; i++) dmFilter[DM_N + i] = dmFilter[i];
    /* SYNC_VECTOR (:79-81) == the encoder's sync LFSR (FECDecoder.java:600-605), 1->+1, 0->-1 */
    int sr = 0x7f;
    for (int i = 0; i < SYNC_N; i++) {
        SYNC_VECTOR[i] = (sr & 64) ? 1 : -1;
        int v = sr & 0x48;
        v ^= v >> 4;
        v ^= v >> 2;
        v ^= v >> 1;
        sr = ((sr << 1) | (v & 1)) & 0xffff;
    }
    consts_ready = 1;
}

int jo_bpsk_table(int which, double *out, int cap)
{
    build_consts();
    int n = 0;
    if (which == 0) {
        n = DS_N;
        for (int i = 0; i < n && i < cap; i++) out[i] = dsFilter[i];
    } else if (which == 1) {
        n = 2 * DM_N;
        for (int i = 0; i < n && i < cap; i++) out[i] = dmFilter[i];
    } else if (which == 2) {
        n = SYNC_N;
        for (int i = 0; i < n && i < cap; i++) out[i] = (double)SYNC_VECTOR[i];
    } else
        return -1;
    return n < cap ? n : cap;
}

void jo_bpsk_sincos(double sin_tab[256], double cos_tab[256])
{
    /* Math.sin/cos(n*2.0*Math.PI/SINCOS_SIZE) (:160-161): the double argument is formed exactly as in Java,
     * the function is evaluated in long double and rounded once (= the correctly rounded value; Java's own
     * Math.sin/cos are only specified to 1 ulp, SURVEY.md 7 hard part 7) */
    for (int n = 0; n < SINCOS; n++) {
        double arg = n * 2.0 * JO_PI / SINCOS;
        sin_tab[n] = (double)sinl((long double)arg);
        cos_tab[n] = (double)cosl((long double)arg);
    }
}

typedef struct {
    int32_t rc;
    int64_t bitidx;
    uint8_t data[FEC_BLOCK];
} fec_log_t;

struct jo_bpsk {
    int rate, samples;
    int doFFT, doUp, decodeOK;
    uint8_t decoded[FEC_BLOCK];
    double tuning, tuPhaseInc;
    int cntRaw, cntDS, cntBit, cntFEC, cntDec, dmErrBits;
    double energy1, energy2;
    double sinTab[SINCOS], cosTab[SINCOS];
    /* tuner */
    double tuPhase;
    /* fft acquire */
    double avePeakPower, aveCentreBin;
    int centreBin;
    /* down sampler */
    double dsBuf[DS_N][2];
    int dsPos, dsCnt;
    double HOWARD_FUDGE_FACTOR;
    /* demodulator */
    double vcoPhase;
    double dmBuf[DM_N][2];
    int dmPos;
    double dmEnergy[SPB + 2];
    int dmBitPos, dmPeakPos, dmNewPeak, dmCorr, dmMaxCorr;
    double dmEnergyOut;
    double dmBitPhase;
    double dmLastIQ[2];
    int8_t dmFECCorr[FEC_BITS];
    uint8_t dmFECBits[FEC_BITS];
    /* logs */
    int8_t *bits;
    int64_t nbits, capbits;
    fec_log_t *fec;
    int nfec, capfec;
    double *trace, *trace_ds;
    int64_t ntrace, ntrace_ds, captrace;
    /* test instruments of doBufferFFT (never part of the arithmetic unless switched on): a per-frame record of the two
     * data-dependent decisions and their margins, and a perturbation of the forward spectrum of the size of an FFT's own
     * rounding error -- "how often would another correct double FFT have decided differently?" */
    double *probe;          /* JO_BPSK_PROBE_N doubles per frame */
    int64_t nprobe, capprobe;
    double perturb_scale;   /* 0: off; else every bin's re, im += scale * u * log2(n) * ||x||_2 * U(-1,1) */
    uint64_t perturb_state;
    /* round 5: the same question one stage further -- the inverse transform's output feeds RxDownSample and, through the
     * filters, the three decisions of RxDemodulate (energy2 > 100, di < 0, the 8-way dmNewPeak argmax, :544-545,:586-592).
     * inv_err[0..1]: a bound on what two correct double FFTs can differ by in one real sample of this / the previous frame
     * (inverse rounding + the forward transform's error in the 204 gathered bins), amplified through both FIRs for (fi,fq);
     * dec[]: smallest margin of each decision in units of the error that bound allows in the compared quantity, and
     * counts; errE: the bound carried through the energy IIRs.  With perturb_scale != 0 the inverse output is perturbed too. */
    double inv_err[2];
    double dec_margin[3];   /* di, energy2 vs 100, argmax gap */
    int64_t dec_count[3];
    double errE[SPB + 2];
    uint64_t dec_hash[2];   /* rolling hashes of the energy2 > 100 outcomes and of the dmNewPeak sequence */
    /* round 6: a plain log of the quantities RxDemodulate decides on (both modes; instruments only): per detector instant
     * {di, energy2}, per bit clock {dmEnergy[dmNewPeak] - runner-up, dmNewPeak} -- for the table-perturbation measurement */
    double *dlog_det, *dlog_peak;
    int64_t ndet, npeak, capdlog;
};

jo_bpsk_t *jo_bpsk_new(int rate, int blen, int size, int tuning, int do_fft, int do_up)
{
    build_consts();
    jo_bpsk_t *d = (jo_bpsk_t *)calloc(1, sizeof(*d));
    jo_bpsk_sincos(d->sinTab, d->cosTab);
    d->rate = rate;
    d->samples = blen / size;                              /* :194 */
    d->tuning = (double)tuning;                            /* :195 */
    d->tuPhaseInc = 2.0 * JO_PI * d->tuning / (double)rate; /* :196 */
    d->doFFT = do_fft != 0;
    d->doUp = do_up != 0;
    d->dsPos = DS_N - 1;                                   /* :468 */
    d->dsCnt = 0;
    d->HOWARD_FUDGE_FACTOR = 0.9 * 32768.0;                /* :469 */
    d->dmPos = DM_N - 1;                                   /* :496 */
    d->dmEnergyOut = 1.0;                                  /* :499 */
    return d;
}

void jo_bpsk_free(jo_bpsk_t *d)
{
    if (!d) return;
    free(d->bits);
    free(d->fec);
    free(d->trace);
    free(d->trace_ds);
    free(d->probe);
    free(d->dlog_det);
    free(d->dlog_peak);
    free(d);
}

static void log_bit(jo_bpsk_t *d, int8_t b)
{
    if (d->nbits == d->capbits) {
        d->capbits = d->capbits ? d->capbits * 2 : 4096;
        d->bits = (int8_t *)realloc(d->bits, (size_t)d->capbits);
    }
    d->bits[d->nbits++] = b;
}

static void log_fec(jo_bpsk_t *d, int rc)
{
    if (d->nfec == d->capfec) {
        d->capfec = d->capfec ? d->capfec * 2 : 16;
        d->fec = (fec_log_t *)realloc(d->fec, sizeof(fec_log_t) * (size_t)d->capfec);
    }
    d->fec[d->nfec].rc = rc;
    d->fec[d->nfec].bitidx = d->nbits; /* number of bits sliced so far, incl. the trigger bit */
    memcpy(d->fec[d->nfec].data, d->decoded, FEC_BLOCK);
    d->nfec++;
}

/* :505-595 */
static void RxDemodulate(jo_bpsk_t *d, double i, double q)
{
    if (d->trace_ds && d->ntrace_ds < d->captrace) {
        d->trace_ds[2 * d->ntrace_ds] = i;
        d->trace_ds[2 * d->ntrace_ds + 1] = q;
        d->ntrace_ds++;
    }
    /* advance phase of VCO, wrap at 2*Pi (:511-513) */
    d->vcoPhase += VCO_PHASE_INC;
    if (d->vcoPhase > 2.0 * JO_PI) d->vcoPhase -= 2.0 * JO_PI;
    /* quadrature demodulate to base band, store in FIR ring (:515-516) */
    int vk = (int)(d->vcoPhase * (double)SINCOS / (2.0 * JO_PI)) % SINCOS;
    d->dmBuf[d->dmPos][0] = i * d->cosTab[vk];
    d->dmBuf[d->dmPos][1] = q * d->sinTab[vk];
    /* matched FIR in ring-slot order with rotated taps (:518-523) */
    double fi = 0.0, fq = 0.0;
    for (int n = 0; n < DM_N; n++) {
        int dmi = (DM_N - d->dmPos + n);
        fi += d->dmBuf[n][0] * dmFilter[dmi];
        fq += d->dmBuf[n][1] * dmFilter[dmi];
    }
    d->dmPos--;
    if (d->dmPos < 0) d->dmPos = DM_N - 1;

    if (d->trace && d->ntrace < d->captrace) {
        d->trace[2 * d->ntrace] = fi;
        d->trace[2 * d->ntrace + 1] = fq;
        d->ntrace++;
    }

    /* smoothed bit energy (:534-535) */
    d->energy1 = fi * fi + fq * fq;
    double ey = 0.0; /* (instruments only) bound on |delta fi|, |delta fq| */
    if (d->probe) {
        double ds1 = 0.0, dm1 = 0.0;
        for (int n = 0; n < DS_N; n++) ds1 += fabs(dsFilter[n]);
        for (int n = 0; n < DM_N; n++) dm1 += fabs(dmFilter[n]);
        ey = (d->inv_err[0] > d->inv_err[1] ? d->inv_err[0] : d->inv_err[1]) * ds1 * d->HOWARD_FUDGE_FACTOR * dm1;
        d->errE[d->dmBitPos] = d->errE[d->dmBitPos] * (1.0 - BIT_SMOOTH1) + (2.0 * (fabs(fi) + fabs(fq)) * ey + 2.0 * ey * ey) * BIT_SMOOTH1;
    }
    d->dmEnergy[d->dmBitPos] = (d->dmEnergy[d->dmBitPos] * (1.0 - BIT_SMOOTH1)) + (d->energy1 * BIT_SMOOTH1);
    /* at peak bit energy? decode (:537-575) */
    if (d->dmBitPos == d->dmPeakPos) {
        d->dmEnergyOut = (d->dmEnergyOut * (1.0 - BIT_SMOOTH2)) + (d->energy1 * BIT_SMOOTH2);
        double di = -(d->dmLastIQ[0] * fi + d->dmLastIQ[1] * fq);
        double dq = d->dmLastIQ[0] * fq - d->dmLastIQ[1] * fi;
        const double err_d = ey * (fabs(d->dmLastIQ[0]) + fabs(d->dmLastIQ[1]) + fabs(fi) + fabs(fq)) + 2.0 * ey * ey; /* |delta di|, |delta dq| */
        d->dmLastIQ[0] = fi;
        d->dmLastIQ[1] = fq;
        d->energy2 = sqrt(di * di + dq * dq);
        if (d->dlog_det && d->ndet < d->capdlog) {
            d->dlog_det[2 * d->ndet] = di;
            d->dlog_det[2 * d->ndet + 1] = d->energy2;
            d->ndet++;
        }
        if (d->probe) {
            d->dec_hash[0] = (d->dec_hash[0] * 0x100000001b3ull) ^ (uint64_t)(d->energy2 > 100.0);
            if (err_d > 0.0) {
                const double m2 = fabs(d->energy2 - 100.0) / (1.4142135623730951 * err_d);
                if (d->dec_count[1] == 0 || m2 < d->dec_margin[1]) d->dec_margin[1] = m2;
                d->dec_count[1]++;
                if (d->energy2 > 100.0) {
                    const double m1 = fabs(di) / err_d;
                    if (d->dec_count[0] == 0 || m1 < d->dec_margin[0]) d->dec_margin[0] = m1;
                    d->dec_count[0]++;
                }
            }
        }
        if (d->energy2 > 100.0) {
            int bit = di < 0.0;
            memmove(d->dmFECCorr, d->dmFECCorr + 1, FEC_BITS - 1);
            d->dmFECCorr[FEC_BITS - 1] = (int8_t)(bit ? 1 : -1);
            log_bit(d, (int8_t)(bit ? 1 : -1));
            d->dmCorr = 0;
            for (int n = 0; n < SYNC_N; n++) d->dmCorr += d->dmFECCorr[n * 80] * SYNC_VECTOR[n];
            if (d->dmCorr >= 45) {
                for (int n = 0; n < FEC_BITS; n++) d->dmFECBits[n] = (uint8_t)(d->dmFECCorr[n] == 1 ? 0xc0 : 0x40);
                d->dmErrBits = jo_fec_decode(d->dmFECBits, d->decoded);
                d->cntFEC++;
                d->dmMaxCorr = 0;
                d->decodeOK = d->dmErrBits < 0 ? 0 : 1;
                d->cntDec += (d->decodeOK ? 1 : 0);
                log_fec(d, d->dmErrBits);
            }
            if (d->dmCorr > d->dmMaxCorr) d->dmMaxCorr = d->dmCorr;
            d->cntBit++;
        }
    }
    /* half-way into next bit? reset peak energy point (:577-579) */
    if (d->dmBitPos == dmHalfTable[d->dmPeakPos]) d->dmPeakPos = d->dmNewPeak;
    d->dmBitPos = (d->dmBitPos + 1) % SPB;
    /* advance phase of bit position (:581-593) */
    d->dmBitPhase += BIT_PHASE_INC;
    if (d->dmBitPhase >= BIT_TIME) {
        d->dmBitPhase -= BIT_TIME;
        d->dmBitPos = 0;
        double eMax = -1.0e10F;
        for (int n = 0; n < SPB; n++) {
            if (d->dmEnergy[n] > eMax) {
                d->dmNewPeak = n;
                eMax = d->dmEnergy[n];
            }
        }
        if (d->dlog_peak && d->npeak < d->capdlog) {
            int second = -1;
            for (int n = 0; n < SPB; n++)
                if (n != d->dmNewPeak && (second < 0 || d->dmEnergy[n] > d->dmEnergy[second])) second = n;
            d->dlog_peak[2 * d->npeak] = d->dmEnergy[d->dmNewPeak] - d->dmEnergy[second];
            d->dlog_peak[2 * d->npeak + 1] = (double)d->dmNewPeak;
            d->npeak++;
        }
        if (d->probe) {
            d->dec_hash[1] = (d->dec_hash[1] * 0x100000001b3ull) ^ (uint64_t)d->dmNewPeak;
            int second = -1;
            for (int n = 0; n < SPB; n++)
                if (n != d->dmNewPeak && (second < 0 || d->dmEnergy[n] > d->dmEnergy[second])) second = n;
            const double err = d->errE[d->dmNewPeak] + d->errE[second];
            if (err > 0.0) {
                const double m3 = (d->dmEnergy[d->dmNewPeak] - d->dmEnergy[second]) / err;
                if (d->dec_count[2] == 0 || m3 < d->dec_margin[2]) d->dec_margin[2] = m3;
                d->dec_count[2]++;
            }
        }
    }
    d->cntDS++;
}

/* :470-492 */
static void RxDownSample(jo_bpsk_t *d, double i, double q)
{
    d->dsBuf[d->dsPos][0] = i;
    d->dsBuf[d->dsPos][1] = q;
    if (++d->dsCnt >= d->rate / DOWN_RATE) {
        double fi = 0.0, fq = 0.0;
        for (int n = 0; n < DS_N; n++) {
            int dsi = (n + d->dsPos) % DS_N;
            fi += d->dsBuf[dsi][0] * dsFilter[n];
            fq += d->dsBuf[dsi][1] * dsFilter[n];
        }
        d->dsCnt = 0;
        RxDemodulate(d, fi * d->HOWARD_FUDGE_FACTOR, fq * d->HOWARD_FUDGE_FACTOR);
    }
    d->dsPos--;
    if (d->dsPos < 0) d->dsPos = DS_N - 1;
    d->cntRaw++;
}

/* :382-397 */
static void RxMixTuner(jo_bpsk_t *d, double i, double q)
{
    d->tuPhase += d->tuPhaseInc;
    if (d->tuPhase > 2.0 * JO_PI) d->tuPhase -= 2.0 * JO_PI;
    if (d->tuPhase > 0.0) {
        int k = (int)(d->tuPhase * (double)SINCOS / (2.0 * JO_PI)) % SINCOS;
        double mi = i * d->cosTab[k];
        double mq = q * d->sinTab[k];
        RxDownSample(d, mi, mq);
    } else {
        RxDownSample(d, i, q);
    }
}

/* :366-379 */
static void doBufferTune(jo_bpsk_t *d, const float *buf)
{
    for (int n = 0; n < d->samples; n++) {
        double i = (double)buf[n * 2];
        double q = (double)buf[n * 2 + 1];
        RxMixTuner(d, i, q);
    }
}

/* :406-464 */
static void doBufferFFT(jo_bpsk_t *d, const float *buf)
{
    int samples = d->samples;
    double *fftFwd = (double *)malloc(sizeof(double) * 2 * (size_t)samples);
    double *fftRev = (double *)calloc(2 * (size_t)samples, sizeof(double));
    double *psd = (double *)calloc((size_t)samples, sizeof(double));
    double *avePsd = (double *)calloc((size_t)samples, sizeof(double));
    for (int n = 0; n < samples; n++) {
        fftFwd[2 * n] = (double)buf[n * 2];
        fftFwd[2 * n + 1] = (double)buf[n * 2 + 1];
    }
    double xnorm = 0.0;  /* (instruments only) ||x||_2 of the frame */
    if (d->probe || d->perturb_scale != 0.0) {
        for (int n = 0; n < 2 * samples; n++) xnorm += fftFwd[n] * fftFwd[n];
        xnorm = sqrt(xnorm);
    }
    jo_fft_f64(fftFwd, samples, 0, 0);
    if (d->perturb_scale != 0.0) {
        /* a forward-error bound of a floating-point FFT: |X'_k - X_k| <= c u log2(n) ||x||_2 (Higham, ASNA 24.2); here
         * every component moves by up to scale x that, uniformly distributed (splitmix64) */
        const double bound = d->perturb_scale * 1.1102230246251565e-16 * log2((double)samples) * xnorm;
        for (int n = 0; n < 2 * samples; n++) {
            uint64_t z = (d->perturb_state += 0x9e3779b97f4a7c15ull);
            z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
            z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
            z ^= z >> 31;
            fftFwd[n] += bound * (((double)(z >> 11) * (1.0 / 9007199254740992.0)) * 2.0 - 1.0);
        }
    }
    for (int i = 0; i < samples / 2; i++)
        psd[i] = sqrt(fftFwd[2 * i] * fftFwd[2 * i] + fftFwd[2 * i + 1] * fftFwd[2 * i + 1]);
    double maxBin = 0.0;
    int binPos = -1;
    int beg = d->doUp ? samples / 4 : 0;
    int end = d->doUp ? samples / 2 : samples / 4;
    avePsd[0] = 0;
    for (int i = beg + 75; i < end - 75; i++) {
        avePsd[i] = 0;
        for (int j = i - 50; j < i + 50; j++) avePsd[i] += psd[j];
        if (maxBin < avePsd[i]) {
            maxBin = avePsd[i];
            binPos = i;
        }
    }
    if (d->centreBin < 0) d->centreBin = 0;
    if (d->centreBin > end - 1) d->centreBin = end - 1;
    d->avePeakPower = (PSD_AVERAGE_FACTOR * avePsd[d->centreBin]) + (PSD_INV_AVERAGE_FACTOR * d->avePeakPower);
    const int took = maxBin > (d->avePeakPower / 4) * 5 && binPos > 0;
    if (took) {
        d->aveCentreBin = (CFREQ_AVERAGE_FACTOR * (float)binPos) + (CFREQ_INV_AVERAGE_FACTOR * d->aveCentreBin);
        d->centreBin = (int)(d->aveCentreBin + 1.0F);
    }
    if (d->centreBin < 102) d->centreBin = 102;
    if (d->probe && d->nprobe < d->capprobe) {
        /* the runner-up of the first-maximum search: the largest sum at any OTHER position */
        double second = -1.0;
        for (int i = beg + 75; i < end - 75; i++)
            if (i != binPos && avePsd[i] > second) second = avePsd[i];
        double *r = d->probe + JO_BPSK_PROBE_N * d->nprobe++;
        r[0] = (double)binPos;
        r[1] = maxBin;
        r[2] = second;
        r[3] = (d->avePeakPower / 4) * 5;  /* the threshold maxBin was compared with (:447) */
        r[4] = (double)took;
        r[5] = (double)d->centreBin;
        r[6] = xnorm;
        r[7] = (double)samples;
    }
    memcpy(fftRev, fftFwd + 2 * (d->centreBin - 102), sizeof(double) * 2 * 204);
    double ynorm = 0.0;
    if (d->probe || d->perturb_scale != 0.0) {
        for (int n = 0; n < 2 * 204; n++) ynorm += fftRev[n] * fftRev[n];
        ynorm = sqrt(ynorm);
        /* one real sample of the scaled inverse: its own rounding, u log2(n) ||Y||_2 / n, plus what the forward transform's
         * error in the 204 gathered bins (each component within u log2(n) ||x||_2) adds: (1/n) sum |dY_k| */
        d->inv_err[1] = d->inv_err[0];
        d->inv_err[0] = 1.1102230246251565e-16 * log2((double)samples) * (ynorm + 204.0 * 1.4142135623730951 * xnorm) / (double)samples;
    }
    jo_fft_f64(fftRev, samples, 1, 1);
    if (d->perturb_scale != 0.0) {
        const double bound = d->perturb_scale * 1.1102230246251565e-16 * log2((double)samples) * ynorm / (double)samples;
        for (int n = 0; n < samples; n++) {
            uint64_t z = (d->perturb_state += 0x9e3779b97f4a7c15ull);
            z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
            z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
            z ^= z >> 31;
            fftRev[2 * n] += bound * (((double)(z >> 11) * (1.0 / 9007199254740992.0)) * 2.0 - 1.0);
        }
    }
    for (int i = 0; i < samples; i++) RxDownSample(d, fftRev[2 * i], fftRev[2 * i]); /* Q dropped (:462) */
    free(fftFwd);
    free(fftRev);
    free(psd);
    free(avePsd);
}

void jo_bpsk_receive(jo_bpsk_t *d, const float *buf) /* :357-364 */
{
    if (d->doFFT)
        doBufferFFT(d, buf);
    else
        doBufferTune(d, buf);
}

void jo_bpsk_receive_i16(jo_bpsk_t *d, const int16_t *raw, int nframes_total, int ic, int qc)
{
    float *buf = (float *)malloc(sizeof(float) * 2 * (size_t)d->samples);
    int nb = nframes_total / d->samples;
    for (int b = 0; b < nb; b++) {
        jo_convert_i16(raw + (size_t)b * 2 * d->samples, d->samples, 2, ic, qc, buf);
        jo_bpsk_receive(d, buf);
    }
    free(buf);
}

void jo_bpsk_counters(const jo_bpsk_t *d, int32_t out[10])
{
    out[0] = d->cntRaw; out[1] = d->cntDS; out[2] = d->cntBit; out[3] = d->cntFEC;
    out[4] = d->cntDec; out[5] = d->dmErrBits; out[6] = d->dmCorr; out[7] = d->dmMaxCorr;
    out[8] = d->decodeOK; out[9] = d->centreBin;
}

int64_t jo_bpsk_bits(const jo_bpsk_t *d, int8_t *out, int64_t cap)
{
    int64_t n = d->nbits < cap ? d->nbits : cap;
    if (out && n > 0) memcpy(out, d->bits, (size_t)n);
    return d->nbits;
}

int jo_bpsk_fec_count(const jo_bpsk_t *d) { return d->nfec; }

int jo_bpsk_fec_get(const jo_bpsk_t *d, int idx, int32_t *rc, int64_t *bitidx, uint8_t out[256])
{
    if (idx < 0 || idx >= d->nfec) return -1;
    *rc = d->fec[idx].rc;
    *bitidx = d->fec[idx].bitidx;
    memcpy(out, d->fec[idx].data, FEC_BLOCK);
    return 0;
}

void jo_bpsk_decoded(const jo_bpsk_t *d, uint8_t out[256]) { memcpy(out, d->decoded, FEC_BLOCK); }

void jo_bpsk_trace_enable(jo_bpsk_t *d, int64_t cap_pairs)
{
    free(d->trace);
    free(d->trace_ds);
    d->trace = (double *)malloc(sizeof(double) * 2 * (size_t)cap_pairs);
    d->trace_ds = (double *)malloc(sizeof(double) * 2 * (size_t)cap_pairs);
    d->captrace = cap_pairs;
    d->ntrace = d->ntrace_ds = 0;
}

/* test instruments of doBufferFFT (see struct jo_bpsk) */
void jo_bpsk_fft_probe_enable(jo_bpsk_t *d, int64_t cap_frames)
{
    free(d->probe);
    d->probe = (double *)calloc((size_t)cap_frames * JO_BPSK_PROBE_N, sizeof(double));
    d->capprobe = cap_frames;
    d->nprobe = 0;
}

int64_t jo_bpsk_fft_probe(const jo_bpsk_t *d, double *out, int64_t cap_frames)
{
    int64_t n = d->nprobe < cap_frames ? d->nprobe : cap_frames;
    if (out && n > 0) memcpy(out, d->probe, sizeof(double) * JO_BPSK_PROBE_N * (size_t)n);
    return d->nprobe;
}

/* out[0..2]: smallest margins (di, energy2 vs 100, argmax gap) in units of the error two correct double FFTs allow in that
 * quantity; out[3..5]: decisions counted; out[6..7]: hashes of the threshold outcomes and of the dmNewPeak sequence */
void jo_bpsk_decision_margins(const jo_bpsk_t *d, double *out)
{
    for (int i = 0; i < 3; i++) {
        out[i] = d->dec_margin[i];
        out[3 + i] = (double)d->dec_count[i];
    }
    out[6] = (double)(d->dec_hash[0] >> 11);
    out[7] = (double)(d->dec_hash[1] >> 11);
}

/* round 6 instruments: replace the sin / cos tables (to perturb their entries by an ulp), log what RxDemodulate decides on */
void jo_bpsk_set_sincos(jo_bpsk_t *d, const double sin_tab[256], const double cos_tab[256])
{
    memcpy(d->sinTab, sin_tab, sizeof(d->sinTab));
    memcpy(d->cosTab, cos_tab, sizeof(d->cosTab));
}

void jo_bpsk_declog_enable(jo_bpsk_t *d, int64_t cap)
{
    free(d->dlog_det);
    free(d->dlog_peak);
    d->dlog_det = (double *)calloc((size_t)cap * 2, sizeof(double));
    d->dlog_peak = (double *)calloc((size_t)cap * 2, sizeof(double));
    d->capdlog = cap;
    d->ndet = d->npeak = 0;
}

int64_t jo_bpsk_declog(const jo_bpsk_t *d, int which, double *out, int64_t cap)
{
    const int64_t n = which == 0 ? d->ndet : d->npeak;
    const double *src = which == 0 ? d->dlog_det : d->dlog_peak;
    const int64_t m = n < cap ? n : cap;
    if (out && src && m > 0) memcpy(out, src, sizeof(double) * 2 * (size_t)m);
    return n;
}

void jo_bpsk_fft_perturb(jo_bpsk_t *d, double scale, uint64_t seed)
{
    d->perturb_scale = scale;
    d->perturb_state = seed;
}

int64_t jo_bpsk_trace(const jo_bpsk_t *d, double *out, int64_t cap_pairs)
{
    int64_t n = d->ntrace < cap_pairs ? d->ntrace : cap_pairs;
    if (out && n > 0) memcpy(out, d->trace, sizeof(double) * 2 * (size_t)n);
    return d->ntrace;
}

int64_t jo_bpsk_trace_ds(const jo_bpsk_t *d, double *out, int64_t cap_pairs)
{
    int64_t n = d->ntrace_ds < cap_pairs ? d->ntrace_ds : cap_pairs;
    if (out && n > 0) memcpy(out, d->trace_ds, sizeof(double) * 2 * (size_t)n);
    return d->ntrace_ds;
}

void jo_bpsk_state(const jo_bpsk_t *d, double out[18])
{
    out[0] = d->tuPhase; out[1] = d->vcoPhase; out[2] = d->dmBitPhase; out[3] = d->dmEnergyOut;
    out[4] = d->energy1; out[5] = d->energy2; out[6] = d->avePeakPower; out[7] = d->aveCentreBin;
    for (int i = 0; i < 8; i++) out[8 + i] = d->dmEnergy[i];
    out[16] = d->dmLastIQ[0]; out[17] = d->dmLastIQ[1];
}

void jo_bpsk_istate(const jo_bpsk_t *d, int32_t out[6])
{
    out[0] = d->dsPos; out[1] = d->dsCnt; out[2] = d->dmPos; out[3] = d->dmBitPos;
    out[4] = d->dmPeakPos; out[5] = d->dmNewPeak;
}
