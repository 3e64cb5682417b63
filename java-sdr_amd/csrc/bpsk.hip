// bpsk.hip -- FUNcubeBPSKDemod.receive() chain for batches of independent streams, exact-order FP64.
//
// Reference: FUNcubeBPSKDemod.java:357-595 (receive -> doBufferTune -> RxMixTuner -> RxDownSample ->
// RxDemodulate), constants :26-96, tables :159-162.  This translation unit is compiled with
// -ffp-contract=off: every double product and sum is rounded separately, in the reference's order, so
// the slicer bits are bit-identical to the Java arithmetic by construction (tests compare with the
// oracle's restatement).
//
// Pipeline per batch of L input samples x S streams (all on one HIP stream):
//   host        : input-independent schedules (tuner / VCO table indices per sample) in exact double,
//                 cached while the phase state repeats (it is an exact 8-cycle at 12 kHz / 96 kHz)
//   k_front*    : int16 -> float -> double, tuner mix, 27-tap low-pass at the decimated instants
//                 (newest-first order, :479-483), x HOWARD_FUDGE_FACTOR, VCO mix  -> dm[s][64+j]
//                 (k_fm: fused with the matched filter, the default for int16 input with a periodic tuner schedule;
//                 k_front_reg: register-staged lane windows; k_front: generic, float input; bpsk_fft.hip /
//                 bpsk_fftm.hip: FFT-acquire mode)
//   k_matched   : 65-tap matched filter in RING-SLOT order with rotated taps (:519-523) -> y[s][j]=(fi,fq)
//   k_tail      : bit-energy IIRs, peak tracking, differential slicer (:534-593) -> bits
//   k_sync      : 65-symbol sync correlation at stride 80 over the 5200-bit window (:556-560)
//   k_sync_fin  : trigger ordering, dmCorr / dmMaxCorr bookkeeping (:567-572)
//   k_fec_bpsk  : FECDecode of every triggered window (fec.hip)
// DESIGN.md has the lane/LDS mapping of each kernel and its roofline.
#include "bpsk_fec.h"
#include "bpsk_fft.h"
#include <math.h>
#include <atomic>
#include <thread>
#include <stddef.h>
#include <stdlib.h>
#include <type_traits>
#include <vector>

namespace jsdr {

enum { DS_N = 27, DM_N = 65, HIST_BITS = 5200, MIN_TRIG = 8, SYNC_N = 65 };
// slack (elements) in front of and behind the (fi,fq) buffers: k_tail8 prefetches whole chunks without range checks
enum { Y_PAD = 1024 };



// FUNcubeBPSKDemod.java:27-55, float literals widened (symmetric, first 14)
static const float h_ds_half[14] = {-6.103515625000e-004F, -1.220703125000e-004F, +2.380371093750e-003F,
                                    +6.164550781250e-003F, +7.324218750000e-003F, +7.629394531250e-004F,
                                    -1.464843750000e-002F, -3.112792968750e-002F, -3.225708007813e-002F,
                                    -1.617431640625e-003F, +6.463623046875e-002F, +1.502380371094e-001F,
                                    +2.231445312500e-001F, +2.518310546875e-001F};
// FUNcubeBPSKDemod.java:58-77 (symmetric, first 33)
static const float h_dm_half[33] = {
    -0.0101130691F, -0.0086975143F, -0.0038246093F, +0.0033563764F, +0.0107237026F, +0.0157790936F, +0.0164594107F,
    +0.0119213911F, +0.0030315224F, -0.0076488191F, -0.0164594107F, -0.0197184277F, -0.0150109226F, -0.0023082460F,
    +0.0154712381F, +0.0327423589F, +0.0424493086F, +0.0379940454F, +0.0154712381F, -0.0243701991F, -0.0750320094F,
    -0.1244834076F, -0.1568500423F, -0.1553748911F, -0.1061032953F, -0.0015013786F, +0.1568500423F, +0.3572048240F,
    +0.5786381191F, +0.7940228249F, +0.9744923010F, +1.0945250059F, +1.1366117829F};

struct BpskConst {
    double ds_taps[32];   // [27] used
    double dm_taps[96];   // [65] used, zero beyond (edge steps of the register-blocked loops read past 64)
    signed char sync[72]; // [65] used, +1/-1
};
__constant__ BpskConst c_bpsk;

#define JSDR_WAVE_SYNC()                                      \
    do {                                                      \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                      \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)

// per-stream demodulator state that depends on the data (FUNcubeBPSKDemod.java:497-503)
struct TailState {
    double dmEnergy[8];
    double dmEnergyOut;
    double lastI, lastQ;
    double energy1, energy2;
    int peakPos, newPeak;
    int dmCorr, dmMaxCorr;
    int cntBit, cntFEC, cntDec, dmErrBits, decodeOK;
    int nbits_prev;  // bits sliced in the previous call (locates the 5200-bit history in the other bitlog)
    int overflow;    // sticky: more sync hits in one call than the handle's capacity (trig_cap), or more bits than max_bits
    int uncertified; // fast variant, sticky: a slicer decision fell inside the error margin and could not be redone exactly
    // fast variant only (k_tail<CERT>)
    double emax;        // running maximum of fi*fi+fq*fq over the life of the stream: scales the error bounds
    long long last_g;   // 9600 Hz index of the sample of the last decision (whose (fi,fq) are lastI, lastQ); -1: none yet
    long long redone;   // decisions recomputed in exact order because they fell inside the margin
};

// ------------------------------------------------------------------------------------------- k_front
// int16 -> float -> double, tuner mix, 27-tap low-pass at the decimated instants, x HOWARD, VCO mix.
//
// One wave = one tile of 64*R consecutive outputs of one stream (R = RD/D outputs per lane).
//   stage : the wave copies the tile's raw IQ dwords (4 B/sample, fully coalesced, loads issued in batches
//           of 8) and tuner indices (1 B/sample) into LDS -- 5 B per sample instead of the 16 B of a mixed
//           double2.  Lane l's span starts at sample RD*l; one pad dword (four pad bytes for the indices)
//           per span makes the lane stride odd in dwords: the per-lane reads below are conflict-free.
//   walk  : every lane walks ITS OWN samples from newest to oldest in blocks of D (one output period).
//           A sample at offset t of block b belongs to the windows of outputs b, b-1, .. with ages
//           26-t, 26-t-D, .. (compile-time constants), so the 27-tap sums of NACC = 26/D+1 outputs are in
//           flight at once in rotating register accumulators.  Walking backwards in time makes every
//           output accumulate ages 0,1,..,26: the reference's newest-first order (:479-483).  Each sample
//           is converted and mixed once per lane, in registers; no LDS traffic per multiply-add.
// Samples before the start of the call come from the raw history kept by k_hist_in.
template <int D, int RD>
struct FrontGeom {
    static_assert(RD % D == 0, "a lane span must hold whole outputs");
    static constexpr int R = RD / D;                 // outputs per lane
    static constexpr int NACC = 26 / D + 1;          // outputs whose windows contain one sample
    static_assert(R >= NACC, "span too short for the rotating accumulators");
    static constexpr int NT = 64 * RD - D + 27;      // samples per wave tile
    static constexpr int RSTR = RD + 1;              // elements between lane spans (raw)
    static constexpr int KSTR = RD + 4;              // bytes between lane spans (tuner indices)
    static constexpr int RAW_EL = 64 * RSTR + 32;    // raw elements per wave
    static constexpr int K_BYTES = 64 * KSTR + 64;
};

struct FrontArgs {
    const int *raw;            // int16 pairs as dwords, [S][stride] (null when rawf is used)
    const float2 *rawf;        // alternative input: the float frame of IAudioHandler.receive, [S][stride]
    long long stride_pairs;
    long long nsamples;        // L
    int ic, qc;
    int mix;                   // 0: tuPhase never exceeds 0 (tuning <= 0): samples pass unmixed (:388,:395)
    const unsigned char *ktu;  // [26 + L] tuner table index per sample, 26 history entries first
    const unsigned char *kvco; // [nds]
    const double *sincos;      // cos[256], sin[256]
    const int2 *hist;          // [S][32]: the 26 inputs before this call: DC-corrected int16 pair (.x) or float2 bits
    double2 *dm;               // [S][dm_stride]: 64 history + nds VCO-mixed samples
    long long dm_stride;
    double2 *ds_dbg;           // optional [S][nds] down-sampler outputs (after HOWARD), may be null
    long long nds;
    int first_out;             // input index whose arrival completes output 0 (= D-1-dsCnt0)
    const double2 *tcs;        // k_front_reg<PER>: unwrapped periodic tuner table (see FmArgs), else null
    int tper;
};

template <bool F32IN>
struct FrontElem {
    using type = int;
};
template <>
struct FrontElem<true> {
    using type = float2;
};

// one block of D samples (newest first): outputs b-JLO .. b-JHI take part
template <int D, int RD, bool F32IN, bool MIX, int JLO, int JHI, int NACC>
__device__ __forceinline__ void front_block(int b, const typename FrontElem<F32IN>::type *xl, const unsigned char *kl,
                                            const double *sc, double (&ai)[NACC], double (&aq)[NACC])
{
#pragma unroll
    for (int t = D - 1; t >= 0; t--) {
        if (26 - t - D * JLO >= 0) {  // the sample lies in at least one participating window (compile time)
            const int m = D * b + t;
            const int wrap = (m >= RD) ? 1 : 0;
            double di, dq;
            if constexpr (F32IN) {
                const float2 f = xl[m + wrap];
                di = (double)f.x;  // (double)buf[n*2]  :372
                dq = (double)f.y;
            } else {
                const int w = xl[m + wrap];
                di = (double)i16_to_float_java((int)(short)(w & 0xffff));
                dq = (double)i16_to_float_java(w >> 16);
            }
            if constexpr (MIX) {  // :388-390 component-wise, not a complex multiply (a template flag, not a
                                  // branch: the block stays one basic block and its LDS reads pipeline)
                const int k = kl[m + 4 * wrap];
                di = di * sc[k];
                dq = dq * sc[256 + k];
            }
#pragma unroll
            for (int j = JLO; j <= JHI; j++) {
                if (26 - t - D * j >= 0) {  // age of this sample in the window of output b-j
                    const double tp = ds_tap(26 - t - D * j);
                    ai[j] += di * tp;
                    aq[j] += dq * tp;
                }
            }
        }
    }
}

template <int D, int RD, bool F32IN, bool MIX>
__global__ __launch_bounds__(128, 4) void k_front(FrontArgs a)
{
    using G = FrontGeom<D, RD>;
    using Elem = typename FrontElem<F32IN>::type;
    constexpr int R = G::R, NACC = G::NACC;
    extern __shared__ __align__(16) unsigned char smem[];
    double *sc = reinterpret_cast<double *>(smem);  // [512]
    for (int i = threadIdx.x; i < 512; i += blockDim.x) sc[i] = a.sincos[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    constexpr size_t WAVE_BYTES = (size_t)G::RAW_EL * sizeof(Elem) + G::K_BYTES;
    unsigned char *wbase = smem + 512 * sizeof(double) + wave * WAVE_BYTES;
    Elem *rawL = reinterpret_cast<Elem *>(wbase);
    unsigned char *kL = wbase + (size_t)G::RAW_EL * sizeof(Elem);
    const int s = blockIdx.y;
    const long long ntiles = (a.nds + 64 * R - 1) / (64 * R);
    const double HOWARD = 0.9 * 32768.0;  // :469
    const int *raw = a.raw + (long long)s * a.stride_pairs;
    const float2 *rawf = a.rawf + (long long)s * a.stride_pairs;
    const int2 *hist = a.hist + (long long)s * 32;
    const long long L = a.nsamples;
    for (long long tile = (long long)blockIdx.x * nwave + wave; tile < ntiles; tile += (long long)gridDim.x * nwave) {
        const long long j0 = tile * 64 * R;
        const long long lo = (long long)a.first_out + (long long)D * j0 - 26;  // input index of tile sample 0
        // ---- stage raw samples + tuner indices: 8 independent loads in flight per lane
        constexpr int NIT = (G::NT + 63) / 64;
#pragma unroll 1
        for (int it0 = 0; it0 < NIT; it0 += 8) {
            Elem w[8];
            unsigned char kk[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int e = (it0 + u) * 64 + lane;
                const long long n = lo + e;
                const bool inr = (n >= 0) && (n < L);
                const long long idx = inr ? n : 0;
                if constexpr (F32IN) w[u] = rawf[idx]; else w[u] = raw[idx];
                kk[u] = a.ktu[26 + idx];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int e = (it0 + u) * 64 + lane;
                const long long n = lo + e;
                const bool inr = (n >= 0) && (n < L);
                if (e < G::NT) {
                    Elem v = w[u];
                    if constexpr (F32IN) {
                        if (!inr) v = make_float2(0.f, 0.f);
                    } else {
                        int si = java_short_add((int)(short)(v & 0xffff), a.ic);
                        int sq = java_short_add(v >> 16, a.qc);
                        v = inr ? ((si & 0xffff) | (sq << 16)) : 0;
                    }
                    const int span = e / RD;
                    rawL[e + span] = v;
                    kL[e + 4 * span] = inr ? kk[u] : (unsigned char)0;
                }
            }
        }
        if (lo < 0) {  // first tile of the call: the 26 inputs before it come from the history
            const int e = lane;
            const long long n = lo + e;
            if (n < 0) {
                const int2 h = hist[26 + n];
                const int span = e / RD;
                if constexpr (F32IN) rawL[e + span] = make_float2(__int_as_float(h.x), __int_as_float(h.y));
                else rawL[e + span] = h.x;
                kL[e + 4 * span] = a.ktu[26 + n];
            }
        }
        JSDR_WAVE_SYNC();
        // ---- walk this lane's span from newest to oldest, one output period per block
        const Elem *xl = rawL + G::RSTR * lane;
        const unsigned char *kl = kL + G::KSTR * lane;
        double ai[NACC], aq[NACC];
#pragma unroll
        for (int j = 0; j < NACC; j++) {
            ai[j] = 0.0;
            aq[j] = 0.0;
        }
        const long long jl = j0 + (long long)R * lane;
        auto finish = [&](int b) {  // output b is complete: x HOWARD (:486), VCO mix (:515-516), rotate
            const long long j = jl + b;
            if (j < a.nds) {
                const double oi = ai[0] * HOWARD, oq = aq[0] * HOWARD;
                if (a.ds_dbg) a.ds_dbg[(long long)s * a.nds + j] = make_double2(oi, oq);
                const int kv = a.kvco[j];
                a.dm[(long long)s * a.dm_stride + 64 + j] = make_double2(oi * sc[kv], oq * sc[256 + kv]);
            }
#pragma unroll
            for (int j2 = 0; j2 + 1 < NACC; j2++) {
                ai[j2] = ai[j2 + 1];
                aq[j2] = aq[j2 + 1];
            }
            ai[NACC - 1] = 0.0;
            aq[NACC - 1] = 0.0;
        };
        auto rotate_only = [&]() {
#pragma unroll
            for (int j2 = 0; j2 + 1 < NACC; j2++) {
                ai[j2] = ai[j2 + 1];
                aq[j2] = aq[j2 + 1];
            }
            ai[NACC - 1] = 0.0;
            aq[NACC - 1] = 0.0;
        };
        // top blocks b = R-1+k (k = NACC-1 .. 1): only outputs <= R-1 exist, i.e. j >= k
        if constexpr (NACC >= 7) { front_block<D, RD, F32IN, MIX, 6, NACC - 1, NACC>(R + 5, xl, kl, sc, ai, aq); rotate_only(); }
        if constexpr (NACC >= 6) { front_block<D, RD, F32IN, MIX, 5, NACC - 1, NACC>(R + 4, xl, kl, sc, ai, aq); rotate_only(); }
        if constexpr (NACC >= 5) { front_block<D, RD, F32IN, MIX, 4, NACC - 1, NACC>(R + 3, xl, kl, sc, ai, aq); rotate_only(); }
        if constexpr (NACC >= 4) { front_block<D, RD, F32IN, MIX, 3, NACC - 1, NACC>(R + 2, xl, kl, sc, ai, aq); rotate_only(); }
        if constexpr (NACC >= 3) { front_block<D, RD, F32IN, MIX, 2, NACC - 1, NACC>(R + 1, xl, kl, sc, ai, aq); rotate_only(); }
        if constexpr (NACC >= 2) { front_block<D, RD, F32IN, MIX, 1, NACC - 1, NACC>(R + 0, xl, kl, sc, ai, aq); rotate_only(); }
        // main blocks: every window exists
#pragma unroll 1
        for (int b = R - 1; b >= NACC - 1; b--) {
            front_block<D, RD, F32IN, MIX, 0, NACC - 1, NACC>(b, xl, kl, sc, ai, aq);
            finish(b);
        }
        // bottom blocks b = NACC-2 .. 0: outputs below 0 belong to the previous lane
        if constexpr (NACC >= 7) { front_block<D, RD, F32IN, MIX, 0, 5, NACC>(5, xl, kl, sc, ai, aq); finish(5); }
        if constexpr (NACC >= 6) { front_block<D, RD, F32IN, MIX, 0, 4, NACC>(4, xl, kl, sc, ai, aq); finish(4); }
        if constexpr (NACC >= 5) { front_block<D, RD, F32IN, MIX, 0, 3, NACC>(3, xl, kl, sc, ai, aq); finish(3); }
        if constexpr (NACC >= 4) { front_block<D, RD, F32IN, MIX, 0, 2, NACC>(2, xl, kl, sc, ai, aq); finish(2); }
        if constexpr (NACC >= 3) { front_block<D, RD, F32IN, MIX, 0, 1, NACC>(1, xl, kl, sc, ai, aq); finish(1); }
        if constexpr (NACC >= 2) { front_block<D, RD, F32IN, MIX, 0, 0, NACC>(0, xl, kl, sc, ai, aq); finish(0); }
        if constexpr (NACC == 1) { /* D >= 27: every block is a main block */ }
        JSDR_WAVE_SYNC();
    }
}

// ------------------------------------------------------------------------------------------- k_front_any
// The front end for ANY decimation rate / 9600 (the reference's audio-rate is a free integer, JavaAudio.java:49,59: 32 kHz
// gives 3, 22.05 kHz 2, 64 kHz 6 ...): one thread per output, its 27-sample window walked newest -> oldest (:479-483) with the
// decimation a run-time value.  Same conversions, same operand order as k_front; no staging, no register blocking -- the
// rates java-sdr ships defaults for (44.1 / 48 / 96 / 192 kHz: decimation 4, 5, 10, 20) have the specialised kernels.
template <bool F32IN>
__global__ __launch_bounds__(256) void k_front_any(FrontArgs a, int D)
{
    __shared__ double sc[512];
    for (int i = threadIdx.x; i < 512; i += blockDim.x) sc[i] = a.sincos[i];
    __syncthreads();
    const int s = blockIdx.y;
    const double HOWARD = 0.9 * 32768.0;  // :469
    const int *raw = a.raw + (long long)s * a.stride_pairs;
    const float2 *rawf = a.rawf + (long long)s * a.stride_pairs;
    const int2 *hist = a.hist + (long long)s * 32;
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < a.nds; j += (long long)gridDim.x * blockDim.x) {
        const long long n_new = (long long)a.first_out + (long long)D * j;  // the input whose arrival completes output j
        double fi = 0.0, fq = 0.0;
        for (int age = 0; age < DS_N; age++) {
            const long long n = n_new - age;  // >= -26: before the call, the history k_hist_in kept (zeros at the stream's start)
            double di, dq;
            if constexpr (F32IN) {
                float2 f;
                if (n >= 0) {
                    f = rawf[n];
                } else {
                    const int2 h = hist[26 + n];
                    f = make_float2(__int_as_float(h.x), __int_as_float(h.y));
                }
                di = (double)f.x;  // (double)buf[n*2]  :372
                dq = (double)f.y;
            } else {
                int w;
                if (n >= 0) {
                    w = raw[n];
                    const int si = java_short_add((int)(short)(w & 0xffff), a.ic);
                    const int sq = java_short_add(w >> 16, a.qc);
                    w = (si & 0xffff) | (sq << 16);
                } else {
                    w = hist[26 + n].x;  // kept DC-corrected
                }
                di = (double)i16_to_float_java((int)(short)(w & 0xffff));
                dq = (double)i16_to_float_java(w >> 16);
            }
            if (a.mix) {  // :388-390 component-wise, not a complex multiply
                const int k = a.ktu[26 + n];
                di = di * sc[k];
                dq = dq * sc[256 + k];
            }
            const double tp = c_bpsk.ds_taps[age];
            fi += di * tp;
            fq += dq * tp;
        }
        const double oi = fi * HOWARD, oq = fq * HOWARD;  // :486
        if (a.ds_dbg) a.ds_dbg[(long long)s * a.nds + j] = make_double2(oi, oq);
        const int kv = a.kvco[j];
        a.dm[(long long)s * a.dm_stride + 64 + j] = make_double2(oi * sc[kv], oq * sc[256 + kv]);  // :515-516
    }
}

// lane-span geometry of the register-staged front end: a lane owns RD samples = R outputs, its window NS samples
template <int D, int RD>
struct FrontDmaGeom {
    static_assert(RD % D == 0 && RD % 4 == 0, "lane span: whole outputs, whole quads");
    static constexpr int R = RD / D;
    static constexpr int NS = RD - D + 27;
    static constexpr int NSQ = (NS + 3) / 4;
    static constexpr int NT = 64 * RD - D + 27;
};

// ------------------------------------------------------------------------------------------- k_front_reg
// The int16 fast path without an LDS image: lane l of a tile reads ITS OWN window of NS = RD-D+27 samples
// (RD*l .. RD*l+NS-1) straight into registers with 16-byte loads, newest quad first, and walks it while the older
// quads are still in flight (vmcnt counts them down in issue order).  The overlap of neighbouring windows (26
// samples) is served by L1/L2, not by HBM.  Against an LDS image of the tile (round 1's k_front_dma): no LDS but the 4 KB sin/cos table, so occupancy
// is set by registers alone and does not collapse when the side stream's kernels hold LDS on the same CU --
// the LDS-image kernel was latency bound and its time went with 1/occupancy.  Same arithmetic, same order.
// PER: the tuner index is periodic in the sample number with a period that divides the lane span RD (verified by
// the host over every sample of the call): the (cos, sin) pair of window sample m sits at the compile-time offset m
// from a wave-uniform base of an unwrapped table -- scalar loads and SGPR operands instead of the 1 B/sample index
// stream, the per-sample index arithmetic and the LDS lookups.  FAST: fused multiply-add per tap (the
// margin-certified variant).  The samples are converted two at a time on the packed FP32 pipe (fm_convert).
template <int D, int RD, bool MIX, bool DC, bool PER = false, bool FAST = false>
__global__ __launch_bounds__(256) void k_front_reg(FrontArgs a)
{
    using G = FrontDmaGeom<D, RD>;
    constexpr int R = G::R;
    __shared__ double sc[512];
    for (int i = threadIdx.x; i < 512; i += blockDim.x) sc[i] = a.sincos[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const int s = blockIdx.y;
    const long long ntiles = (a.nds + 64 * R - 1) / (64 * R);
    const double HOWARD = 0.9 * 32768.0;  // :469
    const int *raw = a.raw + (long long)s * a.stride_pairs;
    const int2 *hist = a.hist + (long long)s * 32;
    const int Lm1 = (int)(a.nsamples - 1);
    for (long long tile = (long long)blockIdx.x * nwave + wave; tile < ntiles; tile += (long long)gridDim.x * nwave) {
        const long long j0 = tile * 64 * R;
        const int n0 = a.first_out + (int)(D * j0) - 26 + RD * lane;  // input index of this lane's sample 0
        typedef const __attribute__((address_space(4))) double *const_tab_t;
        const_tab_t tb = nullptr;
        if constexpr (PER && MIX) {  // entry of this wave's sample 0 (RD*lane is a multiple of the period)
            const long long v = (long long)a.first_out + (long long)D * j0;
            const int e0 = __builtin_amdgcn_readfirstlane((int)(((v % a.tper) + a.tper) % a.tper));
            tb = (const_tab_t)(a.tcs + e0);  // tb[2m] = cos, tb[2m+1] = sin
        }
        // ---- this lane's window, newest quad first
        int4 W[G::NSQ];
        unsigned K[G::NSQ];
        const bool inside = n0 >= 0 && n0 + 4 * G::NSQ - 1 <= Lm1;
        if (inside) {
#pragma unroll
            for (int q = G::NSQ - 1; q >= 0; q--) {
                W[q] = *reinterpret_cast<const int4 *>(raw + n0 + 4 * q);  // 4-byte aligned 16-byte load
                if constexpr (!PER) K[q] = *reinterpret_cast<const unsigned *>(a.ktu + 26 + n0 + 4 * q);
            }
        } else {  // first / last window of the call: history before sample 0, clamp beyond the last one
#pragma unroll
            for (int q = G::NSQ - 1; q >= 0; q--) {
                int w[4];
                unsigned k4 = 0;
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int n = n0 + 4 * q + t;
                    if (n < 0) {
                        int hw = (n >= -26) ? hist[26 + n].x : 0;
                        if constexpr (DC) {  // stored corrected: undo this call's correction, the walk re-applies it
                            const int si = (int)(short)((hw & 0xffff) - a.ic);
                            const int sq = (int)(short)((hw >> 16) - a.qc);
                            hw = (si & 0xffff) | (sq << 16);
                        }
                        w[t] = hw;
                    } else {
                        w[t] = raw[n > Lm1 ? Lm1 : n];
                    }
                    const int nk = n < -26 ? -26 : (n > Lm1 ? Lm1 : n);
                    if constexpr (!PER) k4 |= (unsigned)a.ktu[26 + nk] << (8 * t);
                }
                W[q] = make_int4(w[0], w[1], w[2], w[3]);
                K[q] = k4;
            }
        }
        // ---- walk from newest to oldest; sin/cos entries one quad ahead (LDS latency)
        double ai[R], aq[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            ai[r] = 0.0;
            aq[r] = 0.0;
        }
        double CS[G::NSQ][8];
        auto load_sc = [&](int q) {
            if constexpr (MIX && !PER) {
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int k = (K[q] >> (8 * t)) & 0xff;
                    CS[q][2 * t] = sc[k];
                    CS[q][2 * t + 1] = sc[256 + k];
                }
            }
        };
        load_sc(G::NSQ - 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = G::NSQ - 1; q >= 0; q--) {
            if (q - 1 >= 0) load_sc(q - 1);
            const int4 w4 = W[q];
#pragma unroll
            for (int t = 3; t >= 0; t--) {
                const int m = 4 * q + t;
                if (m < G::NS) {
                    const int w = (t == 0) ? w4.x : (t == 1) ? w4.y : (t == 2) ? w4.z : w4.w;
                    double di, dq;
                    fm_convert(w, a.ic, a.qc, DC, di, dq);
                    if constexpr (MIX) {  // :388-390 component-wise, not a complex multiply
                        if constexpr (PER) {
                            di = di * tb[2 * m];
                            dq = dq * tb[2 * m + 1];
                        } else {
                            di = di * CS[q][2 * t];
                            dq = dq * CS[q][2 * t + 1];
                        }
                    }
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        if (m >= D * r && m <= D * r + 26) {  // sample m has age D*r+26-m in the window of output r
                            const double tp = ds_tap(D * r + 26 - m);
                            if constexpr (FAST) {
                                ai[r] = __builtin_fma(di, tp, ai[r]);
                                aq[r] = __builtin_fma(dq, tp, aq[r]);
                            } else {
                                ai[r] += di * tp;
                                aq[r] += dq * tp;
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < R; r++) asm volatile("" : "+v"(ai[r]), "+v"(aq[r])::"memory");  // sums are due here
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- x HOWARD_FUDGE_FACTOR (:486), VCO mix (:515-516)
        const long long jl = j0 + (long long)R * lane;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const long long j = jl + r;
            if (j < a.nds) {
                const double oi = ai[r] * HOWARD, oq = aq[r] * HOWARD;
                if (a.ds_dbg) a.ds_dbg[(long long)s * a.nds + j] = make_double2(oi, oq);
                const int kv = a.kvco[j];
                a.dm[(long long)s * a.dm_stride + 64 + j] = make_double2(oi * sc[kv], oq * sc[256 + kv]);
            }
        }
    }
}

// keep the 26 most recent inputs (DC-corrected int16 pair, or the float pair) for the next call
struct HistArgs {
    const int *raw;
    const float2 *rawf;
    long long stride_pairs, nsamples;
    int ic, qc;
    const int2 *hist_old;
    int2 *hist_new;
    int nstreams;
};
__global__ void k_hist_in(HistArgs a)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int s = t >> 5, i = t & 31;
    if (s >= a.nstreams || i >= 26) return;
    long long n = a.nsamples - 26 + i;
    int2 v;
    if (n < 0) {
        v = a.hist_old[(long long)s * 32 + (26 + n)];
    } else if (a.rawf) {
        float2 f = a.rawf[(long long)s * a.stride_pairs + n];
        v = make_int2(__float_as_int(f.x), __float_as_int(f.y));
    } else {
        int w = a.raw[(long long)s * a.stride_pairs + n];
        int si = java_short_add((int)(short)(w & 0xffff), a.ic);
        int sq = java_short_add(w >> 16, a.qc);
        v = make_int2((si & 0xffff) | (sq << 16), 0);
    }
    a.hist_new[(long long)s * 32 + i] = v;
}

// The 26-sample input history is kept in the form of the input that produced it (DC-corrected int16 pair in .x, or the
// float pair's bits).  A handle fed through the other form next (receive(float[]) after int16 batches or the other way
// round) gets it converted: int16 -> float is JavaAudio's rule; float -> int16 exists exactly when the float is some
// (float)s/32767f (what IAudioHandler delivers, JavaAudio.java:281-288) -- anything else is reported (bad[0] != 0).
__global__ void k_hist_convert(int2 *hist, int nstreams, int to_float, int *bad)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int s = t >> 5, i = t & 31;
    if (s >= nstreams || i >= 26) return;
    int2 v = hist[(long long)s * 32 + i];
    if (to_float) {
        const float fi = i16_to_float_java((int)(short)(v.x & 0xffff)), fq = i16_to_float_java(v.x >> 16);
        v = make_int2(__float_as_int(fi), __float_as_int(fq));
    } else {
        const float fi = __int_as_float(v.x), fq = __int_as_float(v.y);
        const int si = (int)rintf(fi * 32767.0f), sq = (int)rintf(fq * 32767.0f);
        const bool ok = si >= -32768 && si <= 32767 && sq >= -32768 && sq <= 32767 && i16_to_float_java(si) == fi &&
                        i16_to_float_java(sq) == fq;
        if (!ok) atomicOr(bad, 1);
        v = make_int2((si & 0xffff) | (sq << 16), 0);
    }
    hist[(long long)s * 32 + i] = v;
}

// ------------------------------------------------------------------------------------------- k_matched
// 65-tap matched filter, summed in ring-slot order n=0..64 with tap 65-dmPos+n (:519-523).  In time
// terms (g = global index of a 9600 Hz sample, slot(g) = (64-g) mod 65): the window [g-64, g] holds one
// sample s0 with slot 0; the reference adds s0, s0-1, .., g-64 (ages g-s0 .. 64) and then g, g-1, .., s0+1
// (ages 0 .. g-s0-1).  All outputs g = s0+u, u = 0..64, share s0.  Lane l of a workgroup owns the block
// s0 = G + 65 l; wave w computes R consecutive u for all 64 blocks with R accumulators per rail held in
// registers: at every step all lanes need the SAME taps (scalar loads) and one double2 from LDS
// (stride 65 elements -> conflict-free), which is then used 4R times.  Edge steps where an output has
// run out of taps are peeled at compile time, so exactly 65 products are summed per output, in order.
struct MatchedArgs {
    const double2 *dm;   // [S][dm_stride], index 64 + (g - g_first)
    long long dm_stride;
    double2 *y;          // [S][y_stride]
    long long y_stride;
    long long nds;
    long long g_first;   // global 9600 Hz index of dm[64] (= samples demodulated before this call)
    long long tile0;     // global index of the first block of tile 0 (== 64 mod 65, <= g_first)
};

// FAST: one fused multiply-add per tap instead of the reference's separately rounded product and sum (the
// margin-certified variant); the exact-order form is the default everywhere.
template <int R, bool FAST = false>
__device__ __forceinline__ void matched_block(const double2 *xl /* &X[s0] of this lane */, int u0, double (&ai)[R],
                                              double (&aq)[R])
{
    const double *f = c_bpsk.dm_taps;
#pragma unroll
    for (int r = 0; r < R; r++) {
        ai[r] = 0.0;
        aq[r] = 0.0;
    }
    // phase 1: s = s0 - i, ages u+i.  main part: every output still has a tap
    const int n1 = 66 - u0 - R;
    int i = 0;
#ifndef JSDR_MATCHED_NO_CHUNKS
    // eight steps at a time: their 8+R-1 taps come in with one pair of scalar loads and their eight samples with eight
    // LDS reads in flight together -- as a plain loop the compiler reloads all R taps every other step and waits for each
    // load on the spot (lgkmcnt counts scalar loads and LDS reads together).  Same products, same order.
    for (; i + 8 <= n1; i += 8) {
        double tw[8 + R - 1];
#pragma unroll
        for (int k = 0; k < 8 + R - 1; k++) tw[k] = f[u0 + i + k];
        double2 v8[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v8[k] = xl[-(i + k)];
#pragma unroll
        for (int k = 0; k < 8; k++) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                if constexpr (FAST) {
                    ai[r] = __builtin_fma(v8[k].x, tw[k + r], ai[r]);
                    aq[r] = __builtin_fma(v8[k].y, tw[k + r], aq[r]);
                } else {
                    ai[r] += v8[k].x * tw[k + r];
                    aq[r] += v8[k].y * tw[k + r];
                }
            }
        }
    }
#endif
    for (; i < n1; i++) {
        double2 v = xl[-i];
#pragma unroll
        for (int r = 0; r < R; r++) {
            double t = f[u0 + i + r];
            if constexpr (FAST) {
                ai[r] = __builtin_fma(v.x, t, ai[r]);
                aq[r] = __builtin_fma(v.y, t, aq[r]);
            } else {
                ai[r] += v.x * t;
                aq[r] += v.y * t;
            }
        }
    }
    // phase 1 tail: outputs drop out from the top (age would exceed 64)
#pragma unroll
    for (int q = 0; q < R - 1; q++) {
        const int i2 = n1 + q;
        double2 v = xl[-i2];
#pragma unroll
        for (int r = 0; r < R - 1 - q; r++) {
            double t = f[u0 + i2 + r];
            if constexpr (FAST) {
                ai[r] = __builtin_fma(v.x, t, ai[r]);
                aq[r] = __builtin_fma(v.y, t, aq[r]);
            } else {
                ai[r] += v.x * t;
                aq[r] += v.y * t;
            }
        }
    }
    // phase 2 head: s = s0 + u0 + R-1-q, only outputs with u >= s-s0 take part, age = u-(s-s0)
#pragma unroll
    for (int q = 0; q < R - 1; q++) {
        double2 v = xl[u0 + R - 1 - q];
#pragma unroll
        for (int r = R - 1 - q; r < R; r++) {
            double t = f[r - (R - 1 - q)];
            if constexpr (FAST) {
                ai[r] = __builtin_fma(v.x, t, ai[r]);
                aq[r] = __builtin_fma(v.y, t, aq[r]);
            } else {
                ai[r] += v.x * t;
                aq[r] += v.y * t;
            }
        }
    }
    // phase 2 main: s = s0 + u0 - m, m = 0..u0-1, age = r + m
    int m = 0;
#ifndef JSDR_MATCHED_NO_CHUNKS
    for (; m + 8 <= u0; m += 8) {
        double tw[8 + R - 1];
#pragma unroll
        for (int k = 0; k < 8 + R - 1; k++) tw[k] = f[m + k];
        double2 v8[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v8[k] = xl[u0 - (m + k)];
#pragma unroll
        for (int k = 0; k < 8; k++) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                if constexpr (FAST) {
                    ai[r] = __builtin_fma(v8[k].x, tw[k + r], ai[r]);
                    aq[r] = __builtin_fma(v8[k].y, tw[k + r], aq[r]);
                } else {
                    ai[r] += v8[k].x * tw[k + r];
                    aq[r] += v8[k].y * tw[k + r];
                }
            }
        }
    }
#endif
    for (; m < u0; m++) {
        double2 v = xl[u0 - m];
#pragma unroll
        for (int r = 0; r < R; r++) {
            double t = f[r + m];
            if constexpr (FAST) {
                ai[r] = __builtin_fma(v.x, t, ai[r]);
                aq[r] = __builtin_fma(v.y, t, aq[r]);
            } else {
                ai[r] += v.x * t;
                aq[r] += v.y * t;
            }
        }
    }
}

template <bool FAST>
__global__ __launch_bounds__(512) void k_matched(MatchedArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double2 *X = reinterpret_cast<double2 *>(smem);  // [64 + 4160]: X[i] = sample (G - 64 + i)
    const int s = blockIdx.y;
    const long long G = a.tile0 + 4160LL * blockIdx.x;
    const double2 *dm = a.dm + (long long)s * a.dm_stride;
    {
        // all nine loads of a thread are in flight before the first LDS store: as a plain loop the compiler issues
        // one load, waits for it, stores, and pays the full memory latency nine times per workgroup
        constexpr int NLD = (64 + 4160 + 511) / 512;
        double2 v[NLD];
#pragma unroll
        for (int q = 0; q < NLD; q++) {
            const int i = threadIdx.x + 512 * q;
            const long long rel = (G - 64 + i) - a.g_first;  // index relative to the first new sample
            const bool in = i < 64 + 4160 && rel >= -64 && rel < a.nds;
            // clamped address, masked value: no branch between the loads
            const double2 x = dm[64 + (in ? rel : 0)];
            v[q] = in ? x : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int q = 0; q < NLD; q++) {
            const int i = threadIdx.x + 512 * q;
            if (i < 64 + 4160) X[i] = v[q];
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const double2 *xl = X + 64 + 65 * lane;  // &X[s0]
    const long long s0 = G + 65LL * lane;
    double2 *y = a.y + (long long)s * a.y_stride;
    if (wave == 0) {
        double ai[9], aq[9];
        matched_block<9, FAST>(xl, 0, ai, aq);
#pragma unroll
        for (int r = 0; r < 9; r++) {
            long long rel = s0 + r - a.g_first;
            if (rel >= 0 && rel < a.nds) y[rel] = make_double2(ai[r], aq[r]);
        }
    } else {
        const int u0 = 9 + 8 * (wave - 1);
        double ai[8], aq[8];
        matched_block<8, FAST>(xl, u0, ai, aq);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            long long rel = s0 + u0 + r - a.g_first;
            if (rel >= 0 && rel < a.nds) y[rel] = make_double2(ai[r], aq[r]);
        }
    }
}

// keep the last 64 VCO-mixed samples as the next call's history (one wave per stream; loads before stores)
__global__ __launch_bounds__(64) void k_dm_history(double2 *dm, long long dm_stride, long long nds, int nstreams)
{
    int s = blockIdx.x;
    if (s >= nstreams) return;
    double2 *p = dm + (long long)s * dm_stride;
    double2 v = p[nds + threadIdx.x];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    p[threadIdx.x] = v;
}

// ------------------------------------------------------------------------------------------- k_fm
// Front end and matched filter in ONE kernel: int16 IQ -> tuner mix -> 27-tap /D low-pass -> x HOWARD -> VCO mix
// (k_front_reg's arithmetic) -> 65-tap matched filter in ring-slot order (k_matched's) -> y = (fi,fq).  The VCO-mixed
// 9600 Hz samples of a tile never leave the CU: the front half writes them into the LDS image the matched filter
// reads (16 B per 9600 Hz sample that used to go to HBM and come back: 3.4 GB per 1024 x 2^20 batch), and the
// 1 B/sample tuner-index stream is gone as well -- the tuner table index is periodic in the sample number (an
// exact 8-cycle at 12 kHz / 96 kHz; the host verifies the period over every sample of the call), so the (cos, sin)
// pair of window sample m sits at a COMPILE-TIME offset from a tile-uniform base in an unwrapped table: scalar
// loads, SGPR operands, no per-sample index arithmetic and no LDS lookups.
//
//   tile   : NB = 62 blocks of 65 outputs (k_matched's lane = block mapping) + 64 samples of halo = 4094 VCO-mixed
//            samples = 65.5 KB of LDS; 512 threads; two workgroups per CU.
//   front  : 1024 jobs of R = 4 outputs (two rounds of 512 threads, 99.9 % of the lanes busy); a lane reads ITS
//            OWN 57-sample window with 4-byte aligned 16-byte loads, newest quad first, converts I and Q of a
//            sample together (packed FP32: the same IEEE operations as the scalar form, two per instruction) and
//            walks newest -> oldest with the R accumulator pairs in registers (:479-483).  The halo is recomputed
//            (1.6 %); before the call's first sample it comes from the 64 samples the previous call saved.  Windows
//            that reach into the previous call's 26 samples or past the last sample are read from the stream's edge
//            images (k_fm_edges) with the same loads, so every job -- edge or not -- takes the same arithmetic in the
//            same pass; the VCO table indices come in with the window and their table reads are issued under the last
//            quad's arithmetic.
//   matched: as k_matched, from the LDS image.
// (Compile-time timing probes, never defined in the product build (java-sdr_amd/build.py): each of them only REMOVES
// work -- a half of the kernel, a load, a store, a barrier -- and none changes an address that is still accessed; the
// one probe that did (JSDR_X_COAL, round 2) faulted and was deleted.  A probe that needs "wrong data" keeps the
// kernel's own bounds.)
// Where the time goes (2048 streams x 2^20 samples, alone, tools/build_define.sh with -DJSDR_X_NOFRONT / _NOMATCHED /
// _CLK): front half 2.15 ms + matched half 2.12 ms = the kernel's 4.3-4.4 ms; the matched half issues FP64 at ~95 % of
// the chip's measured rate, the front half at ~80 % (1.94 ms with its window loads replaced by constants: the loads'
// latency costs a tenth of it); coalesced window addresses or a register-resident tuner table change nothing.
// Every floating-point operation and its order are those of k_front_reg + k_matched (FAST = false), so (fi,fq)
// stay bit-identical to the reference.  Used when the input is int16, the tuner schedule is periodic with a period
// that divides the lane span (or the tuner is off, tuning <= 0); everything else takes the three-kernel path.
#ifndef JSDR_FM_NB
#define JSDR_FM_NB 62
#endif
enum { FM_NB = JSDR_FM_NB, FM_NT = 64 + 65 * FM_NB, FM_THREADS = 512, FM_TABLE_SLACK = 128, FM_EDGE = 128 };

#ifdef JSDR_X_CLK  // timing experiment: s_memtime ticks (10 ns) per phase, summed over every wave of the launch
__device__ unsigned long long g_fm_clk[256][64];  // [blockIdx & 255][phase]: same-address atomics serialise
#define FM_CLK(i)                                                                          \
    do {                                                                                   \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                      \
        clk_acc_[i] += now_ - clk_last_;                                                   \
        clk_last_ = now_;                                                                  \
    } while (0)
#else
#define FM_CLK(i) do {} while (0)
#endif
struct FmArgs {
    const int *raw;             // int16 pairs as dwords, [S][stride]
    long long stride_pairs;
    int nsamples;               // L
    int ic, qc;
    const int *edges;           // [S][4 * FM_EDGE]: k_fm_edges' images of the stream around sample 0 and around the last sample
    const double2 *tcs;         // unwrapped tuner table: entry e = (cos, sin) for samples n with (n + 26) mod P == e mod P
    int tper;                   // P (1 when the tuner is off)
    const unsigned char *kvco;  // [nds] VCO table index per decimated sample
    const double *sincos;       // cos[256], sin[256]
    const double2 *dmh_old;     // [S][64] the 64 VCO-mixed samples before this call
    double2 *dmh_new;           // [S][64] the last 64 of this call
    double2 *y;                 // [S][y_stride]
    long long y_stride;
    int nds;
    long long g_first;          // global 9600 Hz index of this call's output 0
    long long tile0;            // global index of tile 0's first block (== 64 mod 65, <= g_first)
    int first_out;              // input index whose arrival completes output 0
    int *amax;                  // FAST: [S] running maximum of |int16 sample| per stream, float bits (never reset)
    int ntiles, nstreams;       // work items = ntiles x nstreams, stream-major; the grid strides over them
    int grid_limit;             // > 0: at most that many workgroups, striding over the work items (jsdr_bpsk_set_cu_share)
};

// The stream's edge images for k_fm: E[0 .. 2*FM_EDGE) = samples -FM_EDGE .. FM_EDGE-1, E[2*FM_EDGE .. 4*FM_EDGE) = samples
// L-FM_EDGE .. L+FM_EDGE-1 as raw int16 pairs: the previous call's 26 samples before sample 0 (kept DC-corrected: the
// correction is taken off again, k_fm's conversion re-applies it -- 16-bit wrap-around both ways, so exactly the stored
// value), zero before them and beyond the last sample.
struct EdgeArgs {
    const int *raw;
    long long stride_pairs;
    int nsamples, ic, qc, dc;
    const int2 *hist;
    int *edges;
    int nstreams;
};
__global__ void k_fm_edges(EdgeArgs a)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int s = t / (4 * FM_EDGE), i = t % (4 * FM_EDGE);
    if (s >= a.nstreams) return;
    const int n = i < 2 * FM_EDGE ? i - FM_EDGE : a.nsamples - FM_EDGE + (i - 2 * FM_EDGE);
    int w = 0;
    if (n >= 0 && n < a.nsamples) {
        w = a.raw[(long long)s * a.stride_pairs + n];
    } else if (n < 0 && n >= -26) {
        w = a.hist[(long long)s * 32 + 26 + n].x;
        if (a.dc) {
            const int si = (int)(short)((w & 0xffff) - a.ic);
            const int sq = (int)(short)((w >> 16) - a.qc);
            w = (si & 0xffff) | (sq << 16);
        }
    }
    a.edges[(long long)s * (4 * FM_EDGE) + i] = w;
}

// k_fm_edges and k_hist_in in ONE launch (the k_fm path): both only read the call's input and the previous call's history,
// and write disjoint buffers (the edge images, the next call's history) -- one dependent launch less per call.
// ... and, in the receive() form, the schedule's tables: they travel behind the frame in ONE host->device copy and are put
// where the kernels read them here (two byte ranges at most: the VCO indices and the unwrapped tuner table).
struct ScatterArgs {
    const unsigned char *src[2];
    unsigned char *dst[2];
    int bytes[2];
};
__global__ void k_fm_prep(EdgeArgs e, HistArgs hi, ScatterArgs sc)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int nedge = e.nstreams * 4 * FM_EDGE;
    {
        const int v = t - nedge - 32 * e.nstreams;
        if (v >= 0) {
            if (v < sc.bytes[0]) sc.dst[0][v] = sc.src[0][v];
            else if (v - sc.bytes[0] < sc.bytes[1]) sc.dst[1][v - sc.bytes[0]] = sc.src[1][v - sc.bytes[0]];
            return;
        }
    }
    if (t < nedge) {
        const int s = t / (4 * FM_EDGE), i = t % (4 * FM_EDGE);
        const int n = i < 2 * FM_EDGE ? i - FM_EDGE : e.nsamples - FM_EDGE + (i - 2 * FM_EDGE);
        int w = 0;
        if (n >= 0 && n < e.nsamples) {
            w = e.raw[(long long)s * e.stride_pairs + n];
        } else if (n < 0 && n >= -26) {
            w = e.hist[(long long)s * 32 + 26 + n].x;
            if (e.dc) {
                const int si = (int)(short)((w & 0xffff) - e.ic);
                const int sq = (int)(short)((w >> 16) - e.qc);
                w = (si & 0xffff) | (sq << 16);
            }
        }
        e.edges[(long long)s * (4 * FM_EDGE) + i] = w;
        return;
    }
    // ---- k_hist_in's part (int16 input only on this path)
    const int u = t - nedge;
    const int s = u >> 5, i = u & 31;
    if (s >= hi.nstreams || i >= 26) return;
    const long long n = hi.nsamples - 26 + i;
    int2 v;
    if (n < 0) {
        v = hi.hist_old[(long long)s * 32 + (26 + n)];
    } else {
        const int w = hi.raw[(long long)s * hi.stride_pairs + n];
        const int si = java_short_add((int)(short)(w & 0xffff), hi.ic);
        const int sq = java_short_add(w >> 16, hi.qc);
        v = make_int2((si & 0xffff) | (sq << 16), 0);
    }
    hi.hist_new[(long long)s * 32 + i] = v;
}

// SMALL: the instantiation for SHORT calls (at most FM_THREADS outputs: the receive() form) -- the matched half as one
// output per thread; a kernel of its own so that its loop does not sit in the batch kernel's register allocation (as a
// run-time branch it cost the batch kernel its fourth wave per SIMD: 123 -> 131 VGPRs, 16.8 -> 20.1 ms at 8192 streams)
template <int D, int R, bool MIX, bool DC, bool FAST, bool SMALL = false>
#ifndef JSDR_FM_MINWAVES
#define JSDR_FM_MINWAVES 2
#endif
__global__ __launch_bounds__(FM_THREADS, JSDR_FM_MINWAVES) void k_fm(FmArgs a)
{
    constexpr int RD = D * R, NS = RD - D + 27, NSQ = (NS + 3) / 4;
    constexpr int JOBS = (FM_NT + R - 1) / R, ROUNDS = (JOBS + FM_THREADS - 1) / FM_THREADS;
    extern __shared__ __align__(16) unsigned char smem[];
    double2 *X = reinterpret_cast<double2 *>(smem);                    // [FM_NT]: X[t] = sample G - 64 + t
    double *sc = reinterpret_cast<double *>(smem + FM_NT * sizeof(double2));  // [512]
    for (int i = threadIdx.x; i < 512; i += FM_THREADS) sc[i] = a.sincos[i];
#ifdef JSDR_X_CLK
    unsigned long long clk_acc_[6] = {0, 0, 0, 0, 0, 0}, clk_last_ = 0;
#endif
    const long long nwork = (long long)a.ntiles * a.nstreams;
#pragma unroll 1
    for (long long work = blockIdx.x; work < nwork; work += gridDim.x) {
    const int s = (int)(work / a.ntiles);
    const long long G = a.tile0 + (long long)(65 * FM_NB) * (work % a.ntiles);
    const int jrel0 = (int)(G - 64 - a.g_first);  // call-relative output index of X[0] (negative in the first tile)
#ifdef JSDR_X_FM_SMALLSET  // timing probe (round 5): every stream reads one of 16 streams' samples -- 64 MB, served by the memory-side cache
    const int *raw = a.raw + (long long)(s & 15) * a.stride_pairs;
#else
    const int *raw = a.raw + (long long)s * a.stride_pairs;
#endif
    const int *edges = a.edges + (long long)s * (4 * FM_EDGE);
    const double2 *dmh_old = a.dmh_old + (long long)s * 64;
    const int Lm1 = a.nsamples - 1, nds = a.nds, P = a.tper;
    const double HOWARD = 0.9 * 32768.0;  // :469
    // tile-uniform base into the unwrapped tuner table: window sample m of ANY job of this tile uses entry e0 + m
    // (the lane span RD is a multiple of the period)
    int e0 = 0;
    if constexpr (MIX) {
        const long long v = (long long)a.first_out + (long long)D * jrel0;
        e0 = (int)(((v % P) + P) % P);
    }
    // constant address space: the table is read-only for the launch, and only loads the compiler may assume invariant
    // become scalar loads (a plain global pointer in a kernel that also stores gives 57 vector loads per job)
    typedef const __attribute__((address_space(4))) double *const_tab_t;
    const_tab_t tb = (const_tab_t)(a.tcs + e0);  // tb[2m] = cos, tb[2m+1] = sin
    float amx = 0.0f;  // FAST: largest |sample| this lane converts
    __syncthreads();  // sin/cos table
#ifdef JSDR_X_CLK
    clk_last_ = __builtin_amdgcn_s_memtime();
#endif
    // ================================================================================ front half
#ifdef JSDR_X_NOFRONT
    if (a.nds < 0)
#endif
#pragma unroll 1
    for (int round = 0; round < ROUNDS; round++) {
        const int job = threadIdx.x + FM_THREADS * round;
        const int t0 = R * job;
        if (t0 >= FM_NT) break;
        const int j0 = jrel0 + t0;                   // first output of the job, call relative
        const int n0 = a.first_out + D * j0 - 26;    // its window's first sample
        // `none`: no output of the job is filtered here (the halo before output 0, slots past the call's last output).
        // Every other job takes the same arithmetic in the same pass.  A window that reaches back into the previous
        // call's 26 samples, or past the last sample, is read from the stream's EDGE IMAGE instead of the input -- 256
        // samples around sample 0 (history, then input) and 256 around the last one (zero beyond), laid out by
        // k_fm_edges before this kernel -- with the same fifteen loads; outputs that are not this call's (a job that
        // straddles output 0 or the last output) are replaced at the store.  (These jobs used to run a
        // one-output-at-a-time loop of dependent loads after the others had finished: one of them held its workgroup
        // for longer than a whole regular tile takes.  The tile's last job, which owns fewer than R image slots when R
        // does not divide FM_NT, drops the surplus at the store -- on the old edge path it cost every tile ~7 us.)
        const bool none = j0 + R <= 0 || j0 >= nds || nds <= 0;
        const bool regular = j0 >= 0 && j0 + R <= nds;
        if (!none) {
            int4 W[NSQ];
            const int *wp = raw + n0;
            if (n0 < 0) wp = edges + (n0 + FM_EDGE);
            else if (n0 + 4 * NSQ - 1 > Lm1) wp = edges + 2 * FM_EDGE + (n0 - (a.nsamples - FM_EDGE));
            // the VCO table indices of the job's outputs come in WITH the window: left where they are used, after the
            // last quad, the load was issued there and waited for on the spot -- a full memory latency at the end of
            // every round (nothing may cross the quads' scheduling barriers, so the source order decides)
            unsigned kv4[(R + 3) / 4];  // R byte indices, four to a register
#pragma unroll
            for (int k = 0; k < (R + 3) / 4; k++) kv4[k] = 0;
            if (regular) {
#pragma unroll
                for (int r = 0; r < R; r++) kv4[r / 4] |= (unsigned)a.kvco[j0 + r] << (8 * (r % 4));
            } else {
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int j = j0 + r;
                    kv4[r / 4] |= (unsigned)a.kvco[j < 0 ? 0 : (j >= nds ? nds - 1 : j)] << (8 * (r % 4));
                }
            }
#pragma unroll
            for (int q = NSQ - 1; q >= 0; q--) W[q] = *reinterpret_cast<const int4 *>(wp + 4 * q);
#ifndef JSDR_FM_NOPIN
            __builtin_amdgcn_sched_barrier(0);
#endif
            double ai[R], aq[R];
            double vc[R], vs[R];  // the outputs' VCO (cos, sin): LDS reads issued under the last quad's arithmetic
#pragma unroll
            for (int r = 0; r < R; r++) {
                ai[r] = 0.0;
                aq[r] = 0.0;
            }
#pragma unroll
            for (int q = NSQ - 1; q >= 0; q--) {
                const int4 w4 = W[q];
                if (q == 0) {
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const int kv = (kv4[r / 4] >> (8 * (r % 4))) & 0xff;
                        vc[r] = sc[kv];
                        vs[r] = sc[256 + kv];
                    }
                }
#pragma unroll
                for (int t = 3; t >= 0; t--) {
                    const int m = 4 * q + t;
                    if (m < NS) {
                        const int w = (t == 0) ? w4.x : (t == 1) ? w4.y : (t == 2) ? w4.z : w4.w;
                        double di, dq;
                        fm_convert(w, a.ic, a.qc, DC, di, dq, FAST ? &amx : nullptr);
                        if constexpr (MIX) {  // :388-390 component-wise, not a complex multiply
                            di = di * tb[2 * m];
                            dq = dq * tb[2 * m + 1];
                        }
#pragma unroll
                        for (int r = 0; r < R; r++) {
                            if (m >= D * r && m <= D * r + 26) {  // age D*r+26-m in the window of output r
                                const double tp = ds_tap(D * r + 26 - m);
                                if constexpr (FAST) {
                                    ai[r] = __builtin_fma(di, tp, ai[r]);
                                    aq[r] = __builtin_fma(dq, tp, aq[r]);
                                } else {
                                    ai[r] += di * tp;
                                    aq[r] += dq * tp;
                                }
                            }
                        }
                    }
                }
#ifndef JSDR_FM_NOPIN
#pragma unroll
                for (int r = 0; r < R; r++) asm volatile("" : "+v"(ai[r]), "+v"(aq[r])::"memory");  // sums are due here
                if constexpr (FAST) asm volatile("" : "+v"(amx));  // (or the maxima sink to the end of the round with all 114 floats alive)
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
            if (regular) {
#pragma unroll
                for (int r = 0; r < R; r++) {  // x HOWARD_FUDGE_FACTOR (:486), VCO mix (:515-516)
                    const double oi = ai[r] * HOWARD, oq = aq[r] * HOWARD;
                    if (FM_NT % R == 0 || t0 + r < FM_NT) X[t0 + r] = make_double2(oi * vc[r], oq * vs[r]);
                }
            } else {  // a job that straddles output 0 or the call's last output: one or two per stream and call
                double oi[R], oq[R];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    oi[r] = ai[r] * HOWARD * vc[r];
                    oq[r] = aq[r] * HOWARD * vs[r];
                }
#pragma unroll 1
                for (int r = 0; r < R; r++) {
                    const int t = t0 + r, j = j0 + r;
                    double2 val = make_double2(0.0, 0.0);
                    if (j >= 0 && j < nds) {
                        // (run-time r: a select chain, not an indexed register file)
                        val.x = r == 0 ? oi[0] : r == 1 ? oi[1] : r == 2 ? oi[2] : r == 3 ? oi[3] : oi[R - 1];
                        val.y = r == 0 ? oq[0] : r == 1 ? oq[1] : r == 2 ? oq[2] : r == 3 ? oq[3] : oq[R - 1];
                    } else if (j >= -64 && j < 0) {
                        val = dmh_old[64 + j];
                    }
                    if (t < FM_NT) X[t] = val;
                }
            }
        } else {
#pragma unroll 1
            for (int r = 0; r < R; r++) {
                const int t = t0 + r, j = j0 + r;
                const double2 val = (j >= -64 && j < 0) ? dmh_old[64 + j] : make_double2(0.0, 0.0);
                if (t < FM_NT) X[t] = val;
            }
        }
    }
    if constexpr (FAST) {  // non-negative floats order like their bit patterns
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) amx = fmaxf(amx, __shfl_xor(amx, off, 64));
        if ((threadIdx.x & 63) == 0 && amx > 0.0f) atomicMax(a.amax + s, __float_as_int(amx));
    }
    FM_CLK(1);  // front half
    __syncthreads();
    FM_CLK(3);  // barrier after the front half
    // ---- the call's last 64 VCO-mixed samples are the next call's halo; every sample is owned by one tile
    if (jrel0 + FM_NT > nds - 64) {  // uniform
        double2 *dmh_new = a.dmh_new + (long long)s * 64;
        for (int t = 64 + threadIdx.x; t < FM_NT; t += FM_THREADS) {
            const int j = jrel0 + t;
            if (j >= nds - 64 && j < nds && j >= 0) dmh_new[j - (nds - 64)] = X[t];
        }
        if (work % a.ntiles == 0 && nds < 64 && (int)threadIdx.x < 64 - nds) dmh_new[threadIdx.x] = dmh_old[threadIdx.x + nds];
    }
    // ================================================================================ matched filter
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int blk = lane < FM_NB ? lane : FM_NB - 1;  // lanes 62, 63 shadow block 61 and store nothing
    const double2 *xl = X + 64 + 65 * blk;            // &X[s0]
    const int rel0 = jrel0 + 64 + 65 * blk;           // call-relative index of s0
    double2 *y = a.y + (long long)s * a.y_stride;
    if constexpr (SMALL) {
        // A SHORT call (the receive() form: 205 outputs of a 2048-sample frame): one output per thread, the reference's
        // ring-slot order (:519-523) as a per-thread loop over the image -- (s0, s0-1, .., g-64) then (g, .., s0+1) with
        // s0 = g - u the sample in ring slot 0 -- instead of the lane-per-block mapping, whose eight waves each walk all
        // 65 taps for 62 blocks of which a short call fills four (19 us of a 70 us receive).  Same operands, same order:
        // the same doubles (this is tail_exact_sample's second half).
        const int rel = (int)threadIdx.x, t = rel - jrel0;
        if (rel < nds && t >= 64 && t < FM_NT) {
            const long long g = a.g_first + rel;
            const int u = (int)(((g - 64) % 65 + 65) % 65);
            const double2 *xg = X + t;  // &X[g]
            const double *f = c_bpsk.dm_taps;
            double yi = 0.0, yq = 0.0;
            for (int i = 0; i <= 64 - u; i++) {  // s0, s0-1, .., g-64: ages u .. 64
                const double2 x = xg[-(u + i)];
                const double tp = f[u + i];
                if constexpr (FAST) {
                    yi = __builtin_fma(x.x, tp, yi);
                    yq = __builtin_fma(x.y, tp, yq);
                } else {
                    yi += x.x * tp;
                    yq += x.y * tp;
                }
            }
            for (int m = 0; m < u; m++) {        // g, g-1, .., s0+1: ages 0 .. u-1
                const double2 x = xg[-m];
                const double tp = f[m];
                if constexpr (FAST) {
                    yi = __builtin_fma(x.x, tp, yi);
                    yq = __builtin_fma(x.y, tp, yq);
                } else {
                    yi += x.x * tp;
                    yq += x.y * tp;
                }
            }
            y[rel] = make_double2(yi, yq);
        }
    } else {
#ifdef JSDR_X_NOMATCHED
    if (a.nds < 0)
#endif
#ifndef JSDR_FM_DIRECT_STORE  // (the old lane-strided stores: 17.08 vs 16.86 ms at 8192 streams, one session)
    // The tile's 65 * FM_NB outputs leave through the image, which is dead once every wave has walked its blocks: a lane's
    // outputs go to LDS at its block's stride (65 slots: conflict-free), and the workgroup then stores the tile's outputs
    // -- contiguous in y -- as fully coalesced 16-byte accesses, instead of one 16-byte store per lane at a 1040-byte stride
    {
        double ai[9], aq[9];
        int u0 = 0, nout = 9;
        if (wave == 0) {
            matched_block<9, FAST>(xl, 0, ai, aq);
        } else {
            u0 = 9 + 8 * (wave - 1);
            nout = 8;
            double bi[8], bq[8];
            matched_block<8, FAST>(xl, u0, bi, bq);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                ai[r] = bi[r];
                aq[r] = bq[r];
            }
        }
        __syncthreads();  // every wave has finished reading the image
        if (lane < FM_NB) {
#pragma unroll
            for (int r = 0; r < 9; r++)
                if (r < nout) X[65 * blk + u0 + r] = make_double2(ai[r], aq[r]);
        }
        __syncthreads();
        const int relb = jrel0 + 64;  // call-relative index of the tile's first output
        for (int o = (int)threadIdx.x; o < 65 * FM_NB; o += FM_THREADS) {
            const int rel = relb + o;
            if (rel >= 0 && rel < nds) y[rel] = X[o];
        }
    }
#else
    if (wave == 0) {
        double ai[9], aq[9];
        matched_block<9, FAST>(xl, 0, ai, aq);
#pragma unroll
        for (int r = 0; r < 9; r++) {
            const int rel = rel0 + r;
            if (lane < FM_NB && rel >= 0 && rel < nds) y[rel] = make_double2(ai[r], aq[r]);
        }
    } else {
        const int u0 = 9 + 8 * (wave - 1);
        double ai[8], aq[8];
        matched_block<8, FAST>(xl, u0, ai, aq);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int rel = rel0 + u0 + r;
            if (lane < FM_NB && rel >= 0 && rel < nds) y[rel] = make_double2(ai[r], aq[r]);
        }
    }
#endif
    }
    FM_CLK(4);  // matched half
    __syncthreads();  // the next work item reuses the image
    FM_CLK(5);  // barrier after the matched half
    }
#ifdef JSDR_X_CLK
    if ((threadIdx.x & 63) == 0)
        for (int i = 0; i < 6; i++) atomicAdd(&g_fm_clk[blockIdx.x & 255][(threadIdx.x >> 6) * 8 + i], clk_acc_[i]);
#endif
}

// ------------------------------------------------------------------------------------------- k_tail
// One wave per stream, the 9600 Hz tail (:534-593) in chunks of 64 bit periods (512 samples).
// The bit clock is input independent and exactly periodic (bitPos == g mod 8, a new peak is measured
// after every g == 7 mod 8; verified on the host at create time).  The only truly serial arithmetic is
// the nine first-order IIRs (eight dmEnergy channels :535, dmEnergyOut :538); everything else is taken
// out of their loop:
//   parallel : coalesced load of the chunk (prefetched one chunk ahead), energy1 = fi*fi+fq*fq     (:534)
//   serial   : 64 steps; lanes 0..7 advance dmEnergy[lane], lane 8 advances dmEnergyOut SPECULATING that
//              the peak position stays where it is (decision point = bitPos v in every period)
//   parallel : lane = period: first-maximum argmax of the eight energies after that period (:586-592)
//   check    : if every new peak equals v the speculation held (the steady state of a locked demodulator):
//              decision points are (period, v).  Otherwise (rare: acquisition, fades) the chunk is replayed
//              by the scalar state machine of :537,:577-579 and dmEnergyOut is recomputed from its saved value
//   parallel : lane = decision: differential detector, sqrt, threshold, bit (:539-545), ordered compaction
struct TailArgs {
    const double2 *y;
    long long y_stride;
    long long nds;
    long long g_first;
    TailState *st;
    signed char *bitlog_new;        // [S][stride]: 5200 history + new bits
    const signed char *bitlog_old;
    long long bitlog_stride;
    int *nbits;                     // [S] bits sliced in this call
    int max_bits;
    int nstreams;
    // ---- fast variant (k_tail<true>): what it takes to bound the error of (fi,fq) and to redo a sample in exact order
    double ey;                      // bound on |fi' - fi|, |fq' - fq| of the FMA-contracted front end + matched filter at FULL SCALE
    const int *amax;                // [S] largest |int16 sample| the fast kernels have converted so far (float bits): the bound scales with it
    double margin_scale;            // safety factor on the detector margins (>= 1; tests raise it to force the exact path)
    double argmax_scale;            // ... on the argmax margin (tests raise it to provoke an uncertifiable decision)
    const int *raw;                 // the call's input, as FmArgs
    long long stride_pairs;
    int ic, qc, decim, first_out, mix, tper;
    const double2 *tcs;
    const unsigned char *kvco;
    const double *sincos;
};

// (fi,fq) of the 9600 Hz sample g in EXACT order, by the whole wave, from the call's raw input: the 65 VCO-mixed samples
// g-64..g (27-tap low-pass each, :479-483, one per lane) through LDS, then the 65 products in ring-slot order (:519-523).
// The cold path of the fast variant: only samples whose 65-sample window lies inside this call (j0 >= 7, checked by the
// caller, keeps every input index >= 0).
__device__ __forceinline__ double2 tail_exact_sample(const TailArgs &a, int s, long long g, double2 *dmL, int lane)
{
    const int *raw = a.raw + (long long)s * a.stride_pairs;
    const double HOWARD = 0.9 * 32768.0;
    const bool dc = (a.ic != 0) || (a.qc != 0);
    for (int i = lane; i < 65; i += 64) {
        const long long j = g - 64 + i - a.g_first;  // call-relative index of the decimated sample
        double fi = 0.0, fq = 0.0;
        for (int age = 0; age < 27; age++) {
            const long long n = (long long)a.first_out + (long long)a.decim * j - age;
            double di, dq;
            fm_convert(raw[n], a.ic, a.qc, dc, di, dq);
            if (a.mix) {
                const double2 cs = a.tcs[(int)((n + 26) % a.tper)];
                di = di * cs.x;
                dq = dq * cs.y;
            }
            const double tp = c_bpsk.ds_taps[age];
            fi += di * tp;
            fq += dq * tp;
        }
        const double oi = fi * HOWARD, oq = fq * HOWARD;
        const int kv = a.kvco[j];
        dmL[i] = make_double2(oi * a.sincos[kv], oq * a.sincos[256 + kv]);
    }
    JSDR_WAVE_SYNC();
    const int u = (int)(((g - 64) % 65 + 65) % 65);  // g = s0 + u, s0 the sample in ring slot 0
    const double *f = c_bpsk.dm_taps;
    double yi = 0.0, yq = 0.0;
    for (int i = 0; i <= 64 - u; i++) {  // s0, s0-1, .., g-64: ages u .. 64
        const double2 x = dmL[64 - u - i];
        yi += x.x * f[u + i];
        yq += x.y * f[u + i];
    }
    for (int m = 0; m < u; m++) {        // g, g-1, .., s0+1: ages 0 .. u-1
        const double2 x = dmL[64 - m];
        yi += x.x * f[m];
        yq += x.y * f[m];
    }
    JSDR_WAVE_SYNC();
    return make_double2(yi, yq);
}

#ifdef JSDR_X_T8CLK  // timing experiment: s_memtime ticks per phase, summed over every wave of every launch
__device__ unsigned long long g_t8_clk[8];
#define T8_CLK(i)                                                     \
    do {                                                              \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        t8acc_[i] += now_ - t8last_;                                  \
        t8last_ = now_;                                               \
    } while (0)
#else
#define T8_CLK(i) do {} while (0)
#endif
// CERT = the fast variant's tail: the same arithmetic on (fi,fq) that carry a bounded error |d| <= ey, plus, for every
// data-dependent decision, a margin that covers the worst case of that error (DESIGN.md "fast variant"):
//   argmax of the eight smoothed energies (:586-592): certified when the winner leads by more than twice the bound on
//     an energy's error; otherwise the stream is marked uncertified (the IIR state cannot be redone locally)
//   energy2 > 100 (:544) and di < 0 (:545): when inside the margin, the two (fi,fq) samples of the detector are
//     recomputed from the raw input in exact order and the decision is taken on those
template <bool CERT>
__global__ __launch_bounds__(64) void k_tail(TailArgs a)
{
    // (fi,fq) of the chunk's 64 samples at the bit position the peak tracker holds when the chunk starts -- the
    // differential detector's inputs while the demodulator is locked; a decision anywhere else (acquisition, a moving
    // peak) reads its sample from y.  The whole chunk (8 KB) used to sit here: with 10 KB instead of 17 KB a CU holds
    // sixteen of these one-wave workgroups instead of nine, and the kernel is occupancy x latency bound.
    __shared__ double2 ydL[64];        // [period]
    // The exact kernel serves handles of fewer than 2048 streams since round 4 (k_tail8 takes the others): a wave per SIMD at
    // most, so LDS no longer decides how many of these workgroups a CU holds -- the whole chunk's (fi,fq) stay here for the
    // decisions that fall on another bit position (acquisition, fades, the FFT-acquire mode's frame seams) instead of being
    // read again from y (two dependent L2 round trips per chunk: with the serial state machine below, 27 us a chunk unlocked
    // against 3 us locked).  The fast variant's tail (CERT) runs at 8192 streams and keeps the small footprint.
    constexpr bool ALLY = !CERT;
    __shared__ double2 yAll[ALLY ? 512 : 1];
    // The chains' work area, in place and ROW PER CHAIN: before the chain, row c < 8 holds the inputs of dmEnergy[c] --
    // energy1 x 1/200 of the sample at bit position c of every period (the products are taken lane-parallel when the chunk
    // is staged, not by the nine chain lanes one at a time) -- and row 8 energy1 x 1/800 of the sample at position v, for
    // dmEnergyOut; afterwards dmEnergy[c] after every period (a chain has the inputs of its next eight links in registers
    // before it overwrites them).  A chain lane reads and writes ITS row two periods per LDS instruction: the kernel is
    // bound by the number of LDS instructions its twenty waves per CU issue (SQ counters: half of a wave's life spent
    // waiting on lgkmcnt), and one 8-byte access per link in each direction was most of them.  6 KB a workgroup.
    __shared__ __align__(16) double eT[9][72];  // [chain][period]; 72: rows 16-byte aligned, two-way conflicts at worst
    __shared__ unsigned char maskL[64];
    __shared__ short declist[136];
    __shared__ double2 dmL[CERT ? 66 : 1];
    const int lane = threadIdx.x;
    const int s = blockIdx.x;
    if (s >= a.nstreams) return;
#ifdef JSDR_X_T8CLK
    unsigned long long t8acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t8last_ = __builtin_amdgcn_s_memtime();
#endif
    TailState *sp = &a.st[s];
    double emax = CERT ? sp->emax : 0.0;
    long long last_g = CERT ? sp->last_g : -1;
    int uncert = 0;
    long long redone = 0;
    const double2 *y = a.y + (long long)s * a.y_stride;
    signed char *blog = a.bitlog_new + (long long)s * a.bitlog_stride;
    const int nbits_prev = sp->nbits_prev;
    const int cntBit0 = sp->cntBit;
    // carry the 5200-bit shift register (dmFECCorr, :503) over from the previous call's log
    {
        // (the source is byte aligned only; 82 byte loads per lane, in two batches kept in flight together -- a
        // load / wait / store loop would pay the memory latency 82 times before the first chunk starts)
        const signed char *old = a.bitlog_old + (long long)s * a.bitlog_stride + nbits_prev;
        constexpr int NQ = (HIST_BITS + 63) / 64, HALF = (NQ + 1) / 2;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            signed char t[HALF];
#pragma unroll
            for (int q = 0; q < HALF; q++) {
                const int i = lane + 64 * (h * HALF + q);
                t[q] = old[i < HIST_BITS ? i : HIST_BITS - 1];
            }
#pragma unroll
            for (int q = 0; q < HALF; q++) {
                const int i = lane + 64 * (h * HALF + q);
                if (i < HIST_BITS) blog[i] = t[q];
            }
        }
    }
    const double K1 = 1.0 - 1.0 / 200.0, S1 = 1.0 / 200.0;  // BIT_SMOOTH1 (:89)
    const double K2 = 1.0 - 1.0 / 800.0, S2 = 1.0 / 800.0;  // BIT_SMOOTH2 (:90)
    // lanes 0..7 carry dmEnergy[lane], lane 8 carries dmEnergyOut
    double e = (lane < 8) ? sp->dmEnergy[lane] : sp->dmEnergyOut;
    const double Kc = (lane < 8) ? K1 : K2;
    int peakPos = __builtin_amdgcn_readfirstlane(sp->peakPos);
    int newPeak = __builtin_amdgcn_readfirstlane(sp->newPeak);
    double lastI = sp->lastI, lastQ = sp->lastQ, energy1 = sp->energy1, energy2 = sp->energy2;
    int nbits = 0;
    const long long g_first = a.g_first, g_end = a.g_first + a.nds;
    const long long M_first = g_first >> 3, M_last = (g_end - 1) >> 3;

    double2 pre[8];
    auto fetch = [&](long long MB) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            long long g = 8 * MB + k * 64 + lane;
            pre[k] = (g >= g_first && g < g_end) ? y[g - g_first] : make_double2(0.0, 0.0);
        }
    };
    if (a.nds > 0) fetch(M_first);

    for (long long MB = M_first; MB <= M_last && a.nds > 0; MB += 64) {
        // ---------------- stage the prefetched chunk, start fetching the next one
        const int v = peakPos;
#pragma unroll
        for (int k = 0; k < 8; k++) {  // sample k*64 + lane = period k*8 + lane/8, position lane%8
            const double en = pre[k].x * pre[k].x + pre[k].y * pre[k].y;  // :534
            if constexpr (ALLY) yAll[k * 64 + lane] = pre[k];
            eT[lane & 7][k * 8 + (lane >> 3)] = en * S1;
            if ((lane & 7) == v) {
                ydL[k * 8 + (lane >> 3)] = pre[k];
                eT[8][k * 8 + (lane >> 3)] = en * S2;
            }
        }
        JSDR_WAVE_SYNC();
        const double2 *ychunk = y + (8 * MB - g_first);  // sample i of the chunk (only in-range samples are ever decisions)
        auto ysample = [&](int pos) {
            if constexpr (ALLY) return yAll[pos];
            else return (pos & 7) == v ? ydL[pos >> 3] : ychunk[pos];
        };
        double m_en = 0.0, m_d = 0.0, m_e2 = 0.0;  // this chunk's margins
        if constexpr (CERT) {
            double em = 0.0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const double en = pre[k].x * pre[k].x + pre[k].y * pre[k].y;
                em = en > em ? en : em;
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const double o = __shfl_xor(em, off, 64);
                em = o > em ? o : em;
            }
            emax = em > emax ? em : emax;
            // |fi|,|fq| <= sqrt(emax); u = 2^-53.  energy1: 2 sqrt2 sqrt(emax) ey + 7 u emax; through the IIR (gain 1, 3
            // roundings a link, 1/(1-K) = 200 links deep): + 1200 u emax.  di, dq: 4 sqrt(emax) ey + 7 u emax.
            // (all linear in the input: the full-scale bound ey shrinks with the stream's largest sample so far)
            const double sq = sqrt(emax), U = 1.1102230246251565e-16;
            const double eys = a.ey * ((double)__int_as_float(a.amax[s]) * (1.0000001 / 32767.0));
            m_en = a.argmax_scale * 2.0 * (2.83 * sq * eys + 2.0 * eys * eys + 1207.0 * U * emax);
            m_d = a.margin_scale * (4.1 * sq * eys + 2.0 * eys * eys + 7.0 * U * emax);
            m_e2 = 1.5 * m_d + (m_d > 0.0 ? 4.0e-14 : 0.0);
        }
        if (MB + 64 <= M_last) fetch(MB + 64);
        const int nper = (int)((M_last - MB + 1) < 64 ? (M_last - MB + 1) : 64);
        const bool interior = (8 * MB >= g_first) && (8 * (MB + 64) <= g_end);  // every sample of all 64 periods is in range (uniform)
        T8_CLK(0);  // staging
        // ---------------- serial IIRs, speculating that the peak position stays at v
        const bool spec = (newPeak == peakPos);
        const double e_in = e;
        {
            const int idx = (lane < 8) ? lane : v;
            const int col = (lane < 8) ? lane : 8;
            const long long glane = 8 * MB + idx;
            const bool lane_iir = lane < 8, lane_out = (lane == 8) && spec;
            const bool lane_on = lane_iir || lane_out;
            // the 64 energies this lane will fold in, fetched up front: the serial chain below then touches
            // registers only (an LDS read per step would put ~100 cycles of latency on every link of the chain)
#ifndef JSDR_TAIL_XS
#define JSDR_TAIL_XS 8
#endif
            // (a few at a time: the preload is 16 VGPRs instead of 128 -- at 8192 streams the tail is occupancy x latency
            //  bound, and what counts is how many of these one-wave workgroups a CU holds)
            constexpr int XS = JSDR_TAIL_XS;
            const int pfirst = (glane < g_first) ? 1 : 0;  // period 0 of the chunk lacks this lane's sample
#pragma unroll
            for (int p0 = 0; p0 < 64; p0 += XS) {
                double xs[XS];
#pragma unroll
                for (int p = 0; p < XS; p += 2) {  // (the x S products were taken at staging)
                    const double2 x2 = *reinterpret_cast<const double2 *>(&eT[col][p0 + p]);
                    xs[p] = x2.x;
                    xs[p + 1] = x2.y;
                }
                // (pinned: left to itself the compiler sinks the reads back into the chain, one LDS wait per two steps)
#pragma unroll
                for (int p = 0; p < XS; p++) asm volatile("" : "+v"(xs[p]));
                // only the first and the last period of a call can be partial; everywhere else, with the peak position
                // settled (the locked demodulator), a link of the chain is one multiply and one add.  The energies go to
                // LDS as they fall out (stores are not on the chain; a register copy of all 64 would double the kernel's
                // footprint beside the kernels it overlaps with).
                if (interior && spec) {
                    // lanes 0..8 only, under ONE exec mask for the whole chain (a mask per store costs an exec write and
                    // its hazard on every link); lane 8 (dmEnergyOut) stores into the rows' pad column, which nobody reads
                    if (lane <= 8) {
#pragma unroll
                        for (int p = 0; p < XS; p += 2) {
                            const double e0 = (e * Kc) + xs[p];  // :535 / :538
                            e = (e0 * Kc) + xs[p + 1];
                            *reinterpret_cast<double2 *>(&eT[lane][p0 + p]) = make_double2(e0, e);
                        }
                    }
                } else {
#pragma unroll
                    for (int p = 0; p < XS; p++) {
                        const double ne = (e * Kc) + xs[p];
                        const bool ok = lane_on && (p0 + p >= pfirst) && (glane + 8 * (p0 + p) < g_end);
                        if (ok) e = ne;
                        if (lane_iir) eT[lane][p0 + p] = e;
                    }
                }
            }
        }
        JSDR_WAVE_SYNC();
        T8_CLK(1);  // chains
        // ---------------- new peak after every period whose last sample (bitPos 7) is in range (:582-593)
        int np = -1;
        if (lane < nper) {
            const long long g7 = 8 * (MB + lane) + 7;
            if (g7 >= g_first && g7 < g_end) {
                double bv = eT[0][lane], sv = -1.0e300;
                np = 0;
#pragma unroll
                for (int c = 1; c < 8; c++) {
                    double ov = eT[c][lane];
                    if (ov > bv) {  // strict: the first maximum wins
                        sv = bv;
                        bv = ov;
                        np = c;
                    } else if (CERT && ov > sv) {
                        sv = ov;
                    }
                }
                if constexpr (CERT) {
                    // (m_en == 0: nothing but zeros has gone through the fast kernels, both variants hold the same numbers)
                    if (!(bv - sv > m_en) && m_en > 0.0) uncert = 1;  // the order of the two largest is not certain
                }
            }
        }
        T8_CLK(2);  // argmax
        bool replay = false;
        const bool bad = (np >= 0) && (np != v);
        const bool held = spec && (__ballot(bad) == 0ull);
        // A locked stream away from the call's edges: every period has its one decision at bit position v, in period
        // order -- the decision list is known without the mask / prefix-sum / list round trips through LDS.
        const bool fastd = held && interior;
        int nd = 64;
        if (!fastd) {
            if (held) {
                if (lane < nper) {
                    const long long g = 8 * (MB + lane) + v;
                    maskL[lane] = (g >= g_first && g < g_end) ? (unsigned char)(1 << v) : (unsigned char)0;
                }
                // peakPos stays v; every measured peak was v, so newPeak stays v as well
            } else {
                // The peak moved (acquisition, fades, frame seams of the FFT-acquire mode).  The peakPos/newPeak
                // machine of :537,:577-579,:592 is integer only -- its inputs are the per-period argmax values np,
                // which do not depend on dmEnergyOut -- so it runs as scalar code over the periods, one mask per
                // period; dmEnergyOut is then redone from its saved value along the decision list further down.
                // Walking the bit positions cfirst..clast of a period: a decision where c == peakPos (:537); at c ==
                // (peakPos+4)&7 peakPos = newPeak (dmHalfTable, :500,:577-578), after which a second decision can fall at the NEW
                // peakPos if that position is still to come.  In closed form (k_tail8's; branch-free, scalar).
                // Away from the call's edges every period is whole and measured, and then the machine is not serial at all:
                // whenever peakPos != newPeak the switch position (peakPos+4)&7 lies inside the period, so peakPos leaves every
                // period equal to the newPeak it entered with, which is the peak measured one period earlier --
                //     newPeak(p) = np[p-1],  peakPos(p) = np[p-2]   (the chunk's first two periods take the carried state)
                // and each lane writes down its own period's decisions (the scalar walk over 64 periods was most of the 27 us an
                // unlocked chunk took, against 3 us for a locked one).
                int mymask_r = 0;
                if (interior) {
                    const int up1 = __shfl_up(np, 1, 64), up2 = __shfl_up(np, 2, 64);
                    const int nwv = lane == 0 ? newPeak : up1;
                    const int pkv = lane == 0 ? peakPos : (lane == 1 ? newPeak : up2);
                    const int h = (pkv + 4) & 7;
                    const bool diff = pkv != nwv;
                    const bool d1 = !(diff && h < pkv);
                    const bool d2 = diff && nwv > h;
                    mymask_r = (d1 ? 1 << pkv : 0) | (d2 ? 1 << nwv : 0);
                    peakPos = __builtin_amdgcn_readlane(np, 62);
                    newPeak = __builtin_amdgcn_readlane(np, 63);
                } else
                for (int p = 0; p < nper; p++) {
                    const long long gbase = 8 * (MB + p);
                    const int cfirst = (gbase < g_first) ? (int)(g_first - gbase) : 0;
                    const int clast = (gbase + 7 >= g_end) ? (int)(g_end - 1 - gbase) : 7;
                    const int h = (peakPos + 4) & 7;
                    const bool pin = peakPos >= cfirst && peakPos <= clast;
                    const bool hin = h >= cfirst && h <= clast && peakPos != newPeak;
                    const bool d1 = pin && !(hin && h < peakPos);  // the old peakPos decides unless the switch came first
                    const bool d2 = hin && newPeak > h && newPeak <= clast;
                    const int mask = (d1 ? 1 << peakPos : 0) | (d2 ? 1 << newPeak : 0);
                    peakPos = hin ? newPeak : peakPos;
                    if (lane == p) mymask_r = mask;
                    if (clast == 7) newPeak = __builtin_amdgcn_readlane(np, p);
                }
                if (lane < 64) maskL[lane] = (unsigned char)mymask_r;
                replay = true;
            }
            JSDR_WAVE_SYNC();
            // ---------------- decision list of the chunk, in time order
            const int mymask = (lane < nper) ? (int)maskL[lane] : 0;
            const int mycnt = __popc(mymask);
            // inclusive prefix sum over lanes: a period holds at most two decisions, so two ballots count them (six dependent
            // cross-lane adds before)
            const unsigned long long b1 = __ballot(mycnt >= 1), b2 = __ballot(mycnt >= 2);
            const unsigned long long upto = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
            const int pre_sum = __popcll(b1 & upto) + __popcll(b2 & upto);
            nd = __popcll(b1) + __popcll(b2);
            {
                int pos = pre_sum - mycnt;
                int m = mymask;
                while (m) {
                    int c = __ffs(m) - 1;
                    m &= m - 1;
                    declist[pos++] = (short)(lane * 8 + c);
                }
            }
            JSDR_WAVE_SYNC();
            if (replay) {
                // dmEnergyOut (:538) over the decisions in time order: the products x*S2 lane-parallel, the chain
                // e = e*K2 + (x*S2) on broadcast values, in the reference's operation order
                double xa = 0.0, xb = 0.0;
                auto en_of = [&](int pos) {  // energy1 of the chunk's sample pos, as :534 forms it
                    const double2 q = ysample(pos);
                    return q.x * q.x + q.y * q.y;
                };
                if (lane < nd) xa = en_of(declist[lane]) * S2;
                if (lane + 64 < nd) xb = en_of(declist[lane + 64]) * S2;
                double eo = __shfl(e_in, 8, 64);
                // (lane indices as constants, sixteen links per uniform test: as a counted loop every link paid the hazards
                //  of a lane select in an SGPR and a branch)
                auto chain = [&](double xs, int cnt) {
#pragma unroll 1
                    for (int d0 = 0; d0 < 64; d0 += 16) {
                        if (d0 >= cnt) break;
#pragma unroll
                        for (int u = 0; u < 16; u++) {
                            const int lo = __builtin_amdgcn_readlane(__double2loint(xs), u);
                            const int hi = __builtin_amdgcn_readlane(__double2hiint(xs), u);
                            const double ne = (eo * K2) + __hiloint2double(hi, lo);
                            eo = (d0 + u < cnt) ? ne : eo;
                        }
                        // the next sixteen inputs move down to lanes 0..15
                        xs = __hiloint2double(__shfl_down(__double2hiint(xs), 16, 64), __shfl_down(__double2loint(xs), 16, 64));
                    }
                };
                chain(xa, nd < 64 ? nd : 64);
                if (nd > 64) chain(xb, nd - 64);
                if (lane == 8) e = eo;
            }
        }
        T8_CLK(3);  // machine, list, replay
        // ---------------- parallel: differential detector per decision (:539-545)
        for (int d0 = 0; d0 < nd; d0 += 64) {
            const int d = d0 + lane;
            const bool have = d < nd;
            double2 cur = make_double2(0.0, 0.0), prv = make_double2(lastI, lastQ);
            if (have) {
                cur = fastd ? ydL[d] : ysample((int)declist[d]);
                if (d > 0) prv = fastd ? ydL[d - 1] : ysample((int)declist[d - 1]);
            }
            double di = -((prv.x * cur.x) + (prv.y * cur.y));
            double dq = (prv.x * cur.y) - (prv.y * cur.x);
            double e2 = sqrt((di * di) + (dq * dq));
            if constexpr (CERT) {
                const bool unsure = have && (fabs(e2 - 100.0) <= m_e2 || (e2 > 100.0 && fabs(di) <= m_d));
                unsigned long long um = __ballot(unsure);
                if (um) {  // cold: redo those decisions on (fi,fq) recomputed from the raw input in exact order
                    const int ci = have ? (fastd ? 8 * d + v : (int)declist[d]) : 0;
                    const int pi = (have && d > 0) ? (fastd ? 8 * (d - 1) + v : (int)declist[d - 1]) : -1;
                    while (um) {
                        const int L = __ffsll((long long)um) - 1;
                        um &= um - 1ull;
                        const long long gc = 8 * MB + __shfl(ci, L, 64);
                        const int pl = __shfl(pi, L, 64);
                        const long long gp = pl >= 0 ? 8 * MB + pl : last_g;
                        // both windows (65 VCO-mixed samples of 27 inputs each) must lie inside this call
                        const bool can = gp >= 0 && gp - 64 - a.g_first >= 7 && gc - 64 - a.g_first >= 7 && gc < a.g_first + a.nds &&
                                         (a.mix == 0 || a.tper > 0);
                        if (!can) {
                            // (the first samples of a call reach back into the previous one) not recomputable: the decision
                            // stands if it clears the margin proper -- margin_scale only widens what is sent to the redo
                            const double e2L = __shfl(e2, L, 64), diL = __shfl(di, L, 64);
                            if (fabs(e2L - 100.0) <= m_e2 / a.margin_scale || (e2L > 100.0 && fabs(diL) <= m_d / a.margin_scale)) uncert = 1;
                            continue;
                        }
                        // (one inlined instance in a two-trip loop: a call would put the kernel on the function-call ABI --
                        //  256 VGPRs, a stack, one wave per SIMD)
                        double2 ec = make_double2(0.0, 0.0), ep = make_double2(0.0, 0.0);
#pragma unroll 1
                        for (int which = 0; which < 2; which++) {
                            const double2 r = tail_exact_sample(a, s, which ? gp : gc, dmL, lane);
                            if (which) ep = r; else ec = r;
                        }
                        if (lane == L) {
                            di = -((ep.x * ec.x) + (ep.y * ec.y));
                            dq = (ep.x * ec.y) - (ep.y * ec.x);
                            e2 = sqrt((di * di) + (dq * dq));
                        }
                        redone++;
                    }
                }
            }
            const bool valid = have && (e2 > 100.0);
            const unsigned long long vm = __ballot(valid);
            if (valid) {
                int rank = __popcll(vm & ((1ull << lane) - 1ull));
                int pos = nbits + rank;
                if (pos < a.max_bits) blog[HIST_BITS + pos] = (di < 0.0) ? (signed char)1 : (signed char)-1;
            }
            nbits += __popcll(vm);
            // the last decision of the chunk defines dmLastIQ / energy2 for what follows
            const int lastd = nd - 1 - d0;
            if (lastd >= 0 && lastd < 64) {
                energy2 = __shfl(e2, lastd, 64);
                lastI = __shfl(cur.x, lastd, 64);
                lastQ = __shfl(cur.y, lastd, 64);
                if constexpr (CERT) {
                    const int li = have ? (fastd ? 8 * d + v : (int)declist[d]) : 0;
                    last_g = 8 * MB + __shfl(li, lastd, 64);
                }
            }
        }
        JSDR_WAVE_SYNC();
    }
#ifdef JSDR_X_T8CLK
    T8_CLK(4);  // detector (of the last chunk; the others' land on "staging")
    if (lane == 0)
        for (int i = 0; i < 6; i++) atomicAdd(&g_t8_clk[i], t8acc_[i]);
#endif
    // energy1 = that of the last sample processed (:534)
    if (a.nds > 0) {
        const double2 q = y[a.nds - 1];
        energy1 = q.x * q.x + q.y * q.y;
    }
    // ---------------- write back
    const bool any_uncert = CERT && (__ballot(uncert != 0) != 0ull);
    const double e8 = __shfl(e, 8, 64);
    if (lane < 8) sp->dmEnergy[lane] = e;
    if (lane == 0) {
        sp->dmEnergyOut = e8;
        sp->lastI = lastI;
        sp->lastQ = lastQ;
        sp->energy1 = energy1;
        sp->energy2 = energy2;
        sp->peakPos = peakPos;
        sp->newPeak = newPeak;
        const int nb = nbits < a.max_bits ? nbits : a.max_bits;
        sp->cntBit = cntBit0 + nb;
        sp->nbits_prev = nb;
        if (nbits > a.max_bits) sp->overflow = 1;
        a.nbits[s] = nb;
        if constexpr (CERT) {
            sp->emax = emax;
            sp->last_g = last_g;
            sp->redone += redone;
            if (any_uncert) sp->uncertified = 1;
        }
    }
}

// ------------------------------------------------------------------------------------------- k_tail8
// The exact-order tail (:534-593) with EIGHT STREAMS PER WAVE: lane = 8 x (stream of the wave) + bit position.  k_tail
// (one wave per stream) keeps nine of its 64 lanes busy while the nine IIR chains run and moves every energy through
// LDS twice to get it to a chain lane and back; here the eight dmEnergy chains of eight streams ARE the 64 lanes, and a
// lane reads the samples of its own bit position straight from y (the eight lanes of a stream read 128 contiguous
// bytes per period) -- no transposition on the way in.  Per chunk of CH bit periods:
//   A  every lane: energy1 (:534), its dmEnergy link (:535) and -- speculating that its position is the peak -- a
//      dmEnergyOut link (:538); the chain values go to an LDS image for the argmax, the (fi,fq) of the lanes that sit on
//      the peak position to a small LDS list; the next chunk's samples are requested as this chunk's are consumed
//   B  lane = (stream, period): first-maximum argmax of the eight energies after the period (:586-592)
//   fast path (every stream of the wave locked: peakPos == newPeak == every measured peak): the decisions are (period,
//      peakPos); lane = (stream, decision): differential detector, threshold, bit (:539-545), ordered compaction
//   general path (acquisition, fades, the frame seams of the FFT-acquire mode): the peakPos / newPeak machine (:537,
//      :577-579,:592) in closed form per period (at most two decisions fall into one period), lane-replicated per stream;
//      the decision samples re-read from y (L2), dmEnergyOut re-run over them in time order, the detector lane-parallel
//      over the 2 CH decision slots -- no per-stream scalar loop, the eight streams of the wave go through it together
// energy2 = sqrt(di^2+dq^2) > 100 (:543-544) is decided as di^2+dq^2 > 10000: sqrt is correctly rounded and monotone,
// sqrt(10000) = 100 exactly and sqrt(nextafter(10000)) = 100 + 9.1e-15 rounds to the double above 100
// (tests/test_host_logic.py checks the neighbourhood); the square root itself is taken once, for the state the call leaves.
template <int CH, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_tail8(TailArgs a)
{
#ifdef JSDR_X_T8CLK
    unsigned long long t8acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t8last_ = __builtin_amdgcn_s_memtime();
#endif
    static_assert(CH == 16, "two decision slots per period in one 64-bit mask, one nibble per period in another");
    constexpr int ROW = 82;  // doubles per period of the energy image: 8 streams x 10 (8 used) + 2 -> conflict-free b128 reads
    __shared__ __align__(16) double EoL_[WPB][CH * ROW];
    __shared__ __align__(16) double2 FQL_[WPB][8][CH];
    __shared__ __align__(16) unsigned char NPL_[WPB][8][CH];
    // (WPB waves per workgroup, each on its own: a workgroup of four puts one wave on every SIMD of a CU, so that the four
    //  take the registers ONE workgroup of the kernel they run beside leaves free -- four one-wave workgroups land on four CUs)
    const int wv = WPB > 1 ? (int)(threadIdx.x >> 6) : 0, wblk = (int)blockIdx.x * WPB + wv;
    double *EoL = EoL_[wv];
    double2 (*FQL)[CH] = FQL_[wv];
    unsigned char (*NPL)[CH] = NPL_[wv];
    double *X2L = EoL;                                               // general path, once the argmax has read the image:
    double2 *CURL = reinterpret_cast<double2 *>(EoL + 8 * 2 * CH);   // [8][2 CH] each
    static_assert(8 * 2 * CH * 3 <= CH * ROW, "work areas fit the dead energy image");
    const int lane = threadIdx.x & 63, s8 = lane >> 3, c = lane & 7;
    const int S = a.nstreams;
    const int sraw = wblk * 8 + s8;
    const bool live = sraw < S;
    const int s = live ? sraw : S - 1;  // (surplus lanes of the last wave shadow its last stream and store nothing)
    TailState *sp = &a.st[s];
    const int nds = (int)a.nds;
    const long long g_first = a.g_first;
    // ---- carry the 5200-bit shift register (dmFECCorr, :503) over from the previous call's log: the whole wave per stream,
    // dwords (the rows are 16-byte aligned, the source starts at any byte)
    for (int t = 0; t < 8; t++) {
        const int st = wblk * 8 + t;
        if (st >= S) break;
        const int nprev = __builtin_amdgcn_readfirstlane(a.st[st].nbits_prev);
        const signed char *old = a.bitlog_old + (long long)st * a.bitlog_stride + nprev;
        const int sh = (int)(reinterpret_cast<unsigned long long>(old) & 3ull);
        const unsigned *ow = reinterpret_cast<const unsigned *>(old - sh);
        unsigned *nw32 = reinterpret_cast<unsigned *>(a.bitlog_new + (long long)st * a.bitlog_stride);
        constexpr int NW = HIST_BITS / 4, NIT = (NW + 63) / 64;
        unsigned lo[NIT], hi[NIT];
#pragma unroll
        for (int q = 0; q < NIT; q++) {
            const int i = lane + 64 * q;
            const int ic = i < NW ? i : NW - 1;
            lo[q] = ow[ic];
            hi[q] = ow[ic + 1];
        }
#pragma unroll
        for (int q = 0; q < NIT; q++) {
            const int i = lane + 64 * q;
            if (i < NW) nw32[i] = __builtin_amdgcn_alignbyte(hi[q], lo[q], (unsigned)sh);
        }
    }
    const double K1 = 1.0 - 1.0 / 200.0, S1 = 1.0 / 200.0;  // BIT_SMOOTH1 (:89)
    const double K2 = 1.0 - 1.0 / 800.0, S2 = 1.0 / 800.0;  // BIT_SMOOTH2 (:90)
    double e = sp->dmEnergy[c];
    double eo = sp->dmEnergyOut;
    int pk = sp->peakPos, nw = sp->newPeak;
    double lastI = sp->lastI, lastQ = sp->lastQ;
    const int cntBit0 = sp->cntBit;
    int nbits = 0;
    int ord_last = -1;    // this lane's latest decision (ordinal within the call) and its di^2 + dq^2: the stream's last one
    double x_last = 0.0;  // gives energy2 (:543)
    const long long M_first = g_first >> 3, M_last = (g_first + nds - 1) >> 3;
    const double2 *ys = a.y + (long long)s * a.y_stride;
    signed char *blog = a.bitlog_new + (long long)s * a.bitlog_stride;
    int rel0 = (int)(8 * M_first - g_first);  // call-relative index of the chunk's sample 0 (-7 .. 0 for the first chunk)
    // (reads before sample 0 and past the last one stay inside the buffers' slack, Y_PAD; what they return is never used)
    // The call's first period may start and its last may end anywhere (range masks, uniform over the streams).  One code path:
    // with the masks under a branch -- or two sample buffers picked by a branch -- the compiler reconciles the register
    // assignment of the requests in flight where the paths meet, with copies behind an s_waitcnt vmcnt(0): the whole memory
    // latency per chunk (3.0 ms at 8192 streams where the reads alone take 2.2).
    // (the state is due HERE, before the first requests go out: a loop-carried value that is still on its way at the loop's
    //  entry makes the compiler wait for it -- and for everything requested before it -- in EVERY iteration)
    asm volatile("" : "+v"(e), "+v"(eo), "+v"(lastI), "+v"(lastQ), "+v"(pk), "+v"(nw));
    double2 F[CH];
    if (nds > 0) {
        const double2 *p0 = ys + (rel0 + c);
#pragma unroll
        for (int p = 0; p < CH; p++) F[p] = p0[8 * p];
    }
    T8_CLK(0);  // shift register, state, first requests
    for (long long MB = M_first; MB <= M_last && nds > 0; MB += CH, rel0 += 8 * CH) {
        const int v = pk;
        const bool mine = c == v;
        double eo_l = eo;
        const double2 *pn = ys + (rel0 + 8 * CH + c);
        // ---------------- A: the chains; a sample's register is asked for the next chunk's as soon as it has been read
#pragma unroll
        for (int p = 0; p < CH; p++) {
            const double2 f = F[p];
            const double en = (f.x * f.x) + (f.y * f.y);  // :534
            if (mine) FQL[s8][p] = f;
            double x1 = en * S1, x2 = en * S2;
            asm volatile("" : "+v"(x1), "+v"(x2));  // (due HERE: nothing of the sample may be needed below the request)
            // (the sample's last use lies ABOVE the request that overwrites it: scheduled the other way round -- the scheduler's
            //  preference -- the new sample needs registers of its own and comes home through copies behind an s_waitcnt
            //  vmcnt(0) at the loop's back edge: the whole memory latency per chunk)
            __builtin_amdgcn_sched_barrier(0);
#ifndef JSDR_X_T8_NOLOAD  // (timing probe: the first chunk's samples over and over -- the kernel without its HBM reads)
            F[p] = pn[8 * p];
#endif
            __builtin_amdgcn_sched_barrier(0);
            const double ne = (e * K1) + x1;       // :535
            const double no = (eo_l * K2) + x2;    // :538, were this lane's position the peak
            const bool inr = (unsigned)(rel0 + 8 * p + c) < (unsigned)nds;
            e = inr ? ne : e;
            eo_l = inr ? no : eo_l;
            EoL[p * ROW + s8 * 10 + c] = e;
        }
        JSDR_WAVE_SYNC();
        T8_CLK(1);  // A
#ifdef JSDR_X_T8_LOADONLY  // (timing probe: the reads and the chains only)
        if (nds > 0) continue;
#endif
        // ---------------- B: new peak after every period whose last sample is in range (:582-593); lane = (stream, period)
        bool fail = pk != nw;
#pragma unroll
        for (int i = 0; i < CH / 8; i++) {
            const int p = 8 * i + c;
            const double2 *row = reinterpret_cast<const double2 *>(&EoL[p * ROW + s8 * 10]);
            const double2 q0 = row[0], q1 = row[1], q2 = row[2], q3 = row[3];
            double bv = q0.x;
            int np = 0;
            if (q0.y > bv) { bv = q0.y; np = 1; }  // strict: the first maximum wins
            if (q1.x > bv) { bv = q1.x; np = 2; }
            if (q1.y > bv) { bv = q1.y; np = 3; }
            if (q2.x > bv) { bv = q2.x; np = 4; }
            if (q2.y > bv) { bv = q2.y; np = 5; }
            if (q3.x > bv) { bv = q3.x; np = 6; }
            if (q3.y > bv) { bv = q3.y; np = 7; }
            const bool meas = rel0 + 8 * p + 7 < nds;
            np = meas ? np : 8;
            NPL[s8][p] = (unsigned char)np;
            fail = fail || (meas && np != v);
        }
        const bool general = __ballot(fail) != 0ull;
        JSDR_WAVE_SYNC();
        T8_CLK(2);  // B
        const int ord0 = 2 * (int)(MB - M_first) * 1;  // ordinal of the chunk's slot 0 (two slots per period)
        if (!general) {
            // ---------------- locked: one decision per period, at bit position v
            eo = __shfl(eo_l, 8 * s8 + v, 64);
#pragma unroll
            for (int i = 0; i < CH / 8; i++) {
                const int p = 8 * i + c;
                const int drel = rel0 + 8 * p + v;
                const bool inr = drel >= 0 && drel < nds;
                const double2 cur = FQL[s8][p];
                const double2 pv = FQL[s8][p > 0 ? p - 1 : 0];
                const bool from_state = p == 0 || drel < 8;  // the decision before this one fell into an earlier chunk or call
                const double pI = from_state ? lastI : pv.x, pQ = from_state ? lastQ : pv.y;
                const double di = -((pI * cur.x) + (pQ * cur.y));  // :539
                const double dq = (pI * cur.y) - (pQ * cur.x);     // :540
                const double x = (di * di) + (dq * dq);
                const bool valid = inr && x > 10000.0;             // energy2 > 100 (:544)
                const unsigned vm = (unsigned)(__ballot(valid) >> (8 * s8)) & 0xffu;
                if (valid && live) {
                    const int pos = nbits + __popc(vm & ((1u << c) - 1u));
                    if (pos < a.max_bits) blog[HIST_BITS + pos] = (di < 0.0) ? (signed char)1 : (signed char)-1;  // :545
                }
                nbits += __popc(vm);
                if (inr) {
                    ord_last = ord0 + 2 * p;
                    x_last = x;
                }
            }
            {   // dmLastIQ (:541-542) = the chunk's last decision sample
                int plast = (nds - 1 - rel0 - v) >> 3;  // (arithmetic shift: floor)
                plast = plast < CH - 1 ? plast : CH - 1;
                const int pfirst = rel0 + v >= 0 ? 0 : 1;
                const double2 l = FQL[s8][plast > 0 ? plast : 0];
                if (plast >= pfirst) {
                    lastI = l.x;
                    lastQ = l.y;
                }
            }
            T8_CLK(3);  // locked
        } else {
            // ---------------- general: the peakPos / newPeak machine per period, in closed form.  Walking the bit positions
            // cf..cl of a period (:537,:577-579): a decision where c == peakPos; at c == (peakPos+4)&7 peakPos = newPeak, after
            // which a second decision can fall at the NEW peakPos if that position is still to come.
            unsigned long long smask = 0ull;  // bit 2p+k: decision slot k of period p is taken
            unsigned long long pos1 = 0ull, pos2 = 0ull;  // nibble p: the bit position of slot 0 / slot 1
            {
                const uint4 n4 = *reinterpret_cast<const uint4 *>(&NPL[s8][0]);
#pragma unroll
                for (int p = 0; p < CH; p++) {
                    const unsigned w = p < 4 ? n4.x : (p < 8 ? n4.y : (p < 12 ? n4.z : n4.w));
                    const int np = (int)((w >> (8 * (p & 3))) & 0xffu);
                    int cf = -(rel0 + 8 * p), cl = nds - 1 - (rel0 + 8 * p);
                    cf = cf < 0 ? 0 : cf;    // first / last bit position of the period that belongs to the call
                    cl = cl > 7 ? 7 : cl;    // (cl < cf: none of it does)
                    const int h = (pk + 4) & 7;
                    const bool pin = pk >= cf && pk <= cl;
                    const bool hin = h >= cf && h <= cl && pk != nw;
                    // the old peakPos decides unless the switch came first (h < peakPos and h inside the period)
                    const bool d1 = pin && !(hin && h < pk);
                    const bool d2 = hin && nw > h && nw <= cl;
                    if (d1) {
                        smask |= 1ull << (2 * p);
                        pos1 |= (unsigned long long)pk << (4 * p);
                    }
                    if (d2) {
                        smask |= 1ull << (2 * p + 1);
                        pos2 |= (unsigned long long)nw << (4 * p);
                    }
                    pk = hin ? nw : pk;
                    nw = np < 8 ? np : nw;  // :592
                }
            }
            // the decision samples, lane = (stream, slot); dmEnergyOut's inputs and the samples themselves go to LDS
            double2 cur[2 * CH / 8];
            bool has[2 * CH / 8];
#pragma unroll
            for (int r = 0; r < 2 * CH / 8; r++) {
                const int slot = 8 * r + c, p = slot >> 1;
                has[r] = ((smask >> slot) & 1ull) != 0ull;
                const int cpos = (int)((((slot & 1) ? pos2 : pos1) >> (4 * p)) & 15ull);
                const int rel = has[r] ? rel0 + 8 * p + cpos : 0;
                cur[r] = ys[rel];
            }
#pragma unroll
            for (int r = 0; r < 2 * CH / 8; r++) {
                const int slot = 8 * r + c;
                const double en = (cur[r].x * cur[r].x) + (cur[r].y * cur[r].y);
                X2L[s8 * 2 * CH + slot] = en * S2;
                CURL[s8 * 2 * CH + slot] = cur[r];
            }
            JSDR_WAVE_SYNC();
            // dmEnergyOut (:538) over the decisions in time order (lane-replicated per stream)
            {
                const double2 *x2 = reinterpret_cast<const double2 *>(&X2L[s8 * 2 * CH]);
#pragma unroll
                for (int q = 0; q < CH; q++) {
                    const double2 xx = x2[q];
                    const double n0 = (eo * K2) + xx.x;
                    eo = ((smask >> (2 * q)) & 1ull) ? n0 : eo;
                    const double n1 = (eo * K2) + xx.y;
                    eo = ((smask >> (2 * q + 1)) & 1ull) ? n1 : eo;
                }
            }
            // detector per slot; the previous decision is the nearest taken slot below (or the state)
#pragma unroll
            for (int r = 0; r < 2 * CH / 8; r++) {
                const int slot = 8 * r + c;
                const unsigned long long below = smask & ((1ull << slot) - 1ull);
                const int prev = below ? 63 - __clzll((long long)below) : 0;
                const double2 pv = CURL[s8 * 2 * CH + prev];
                const double pI = below ? pv.x : lastI, pQ = below ? pv.y : lastQ;
                const double di = -((pI * cur[r].x) + (pQ * cur[r].y));
                const double dq = (pI * cur[r].y) - (pQ * cur[r].x);
                const double x = (di * di) + (dq * dq);
                const bool valid = has[r] && x > 10000.0;
                const unsigned vm = (unsigned)(__ballot(valid) >> (8 * s8)) & 0xffu;
                if (valid && live) {
                    const int pos = nbits + __popc(vm & ((1u << c) - 1u));
                    if (pos < a.max_bits) blog[HIST_BITS + pos] = (di < 0.0) ? (signed char)1 : (signed char)-1;
                }
                nbits += __popc(vm);
                if (has[r]) {
                    ord_last = ord0 + slot;
                    x_last = x;
                }
            }
            if (smask) {
                const double2 l = CURL[s8 * 2 * CH + (63 - __clzll((long long)smask))];
                lastI = l.x;
                lastQ = l.y;
            }
            T8_CLK(4);  // general
        }
        JSDR_WAVE_SYNC();
    }
    // ---------------- what the call leaves
    double energy1 = sp->energy1, energy2 = sp->energy2;
    if (nds > 0) {
        const double2 q = ys[nds - 1];
        energy1 = (q.x * q.x) + (q.y * q.y);  // that of the last sample processed (:534)
    }
    {   // energy2 (:543) of the stream's last decision: the lane of the stream that holds the highest ordinal
        int o = ord_last;
        double x = x_last;
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) {
            const int oo = __shfl_xor(o, off, 64);
            const double xo = __shfl_xor(x, off, 64);
            if (oo > o) {
                o = oo;
                x = xo;
            }
        }
        if (o >= 0) energy2 = sqrt(x);
    }
    if (live) {
        sp->dmEnergy[c] = e;
        if (c == 0) {
            sp->dmEnergyOut = eo;
            sp->lastI = lastI;
            sp->lastQ = lastQ;
            sp->energy1 = energy1;
            sp->energy2 = energy2;
            sp->peakPos = pk;
            sp->newPeak = nw;
            const int nb = nbits < a.max_bits ? nbits : a.max_bits;
            sp->cntBit = cntBit0 + nb;
            sp->nbits_prev = nb;
            if (nbits > a.max_bits) sp->overflow = 1;
            a.nbits[s] = nb;
        }
    }
#ifdef JSDR_X_T8CLK
    T8_CLK(5);  // write-back
    if (lane == 0)
        for (int i = 0; i < 6; i++) atomicAdd(&g_t8_clk[i], t8acc_[i]);
#endif
}

// ------------------------------------------------------------------------------------------- k_sync
// sync-vector correlation for every new bit (:556-559): window = the 5200 most recent bits, 65 taps at
// stride 80.  corr[b] kept (int8) for the dmMaxCorr bookkeeping; hits (>=45, :560) go to a per-stream list.
struct SyncArgs {
    const signed char *bitlog;
    long long bitlog_stride;
    const int *nbits;
    signed char *corr;      // [S][max_bits]
    int max_bits;
};
__global__ void k_sync(SyncArgs a)
{
    const int s = blockIdx.y;
    const int nb = a.nbits[s];
    const signed char *bl = a.bitlog + (long long)s * a.bitlog_stride;
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += gridDim.x * blockDim.x) {
        const signed char *w = bl + b + 1;  // dmFECCorr after shifting bit b in
        int c = 0;
#pragma unroll 5
        for (int n = 0; n < SYNC_N; n++) c += (int)w[n * 80] * (int)c_bpsk.sync[n];
        a.corr[(long long)s * a.max_bits + b] = (signed char)c;
    }
}

// k_sync_t: the same correlations from a TRANSPOSED image of the stream's bit log in LDS.  Window p of output b is
// W[b + 80 n], W[p] = bitlog[1 + p]: with T[p mod 80][p div 80] = W[p] the 65 bytes of an output are CONTIGUOUS in row
// b mod 80 from column b div 80 -- eighteen aligned dword reads, a byte alignment and seventeen v_dot4_i32_i8 against
// the packed sync vector instead of 65 strided byte loads and 65 multiply-adds (integer arithmetic: same sums).  One
// workgroup per stream; the row stride is 4 * odd so that the rows of 32 consecutive outputs fall on 32 different banks.
// Its first wave then orders the hits and leaves dmCorr / dmMaxCorr (sync_fin_wave below; a kernel of its own until
// round 3: one dependent launch less per call): the workgroup's correlations are read back from global memory behind a
// device-scope release / acquire pair around the barrier.
struct SyncFinArgs {
    int *trig_count, *trig_bits;
    int trig_cap;
    TailState *st;
    int fuse;  // 1: this kernel orders the hits itself (short calls: one dependent launch less); 0: k_sync_fin follows --
               // for a long call the scan is a chain of L2 round trips that one wave walks while the workgroup's LDS
               // image and its other three waves' registers stay allocated (measured at 8192 x 2^20: the step +4 ms)
};
__device__ __forceinline__ void sync_fin_wave(int s, int lane, int nb, const signed char *c, int *trig_count, int *trig_bits,
                                              int trig_cap, TailState *st);
__global__ __launch_bounds__(256) void k_sync_t(SyncArgs a, int row_stride, SyncFinArgs f)
{
    extern __shared__ __align__(16) unsigned char smem[];
    signed char *T = reinterpret_cast<signed char *>(smem);  // [80][row_stride]
    const int s = blockIdx.x;
    const int nb = a.nbits[s];
    if (nb <= 0) {  // no new bit: no hit, dmCorr / dmMaxCorr stay (what sync_fin_wave does with nb = 0)
        if (f.fuse && threadIdx.x == 0) f.trig_count[s] = 0;
        return;
    }
    const int np = HIST_BITS - 1 + nb;  // W[0 .. np), W = the stream's log from byte 1: the last window ends at W[nb-1 + 80*64]
    {
        // the log comes in as 16-byte loads, eight per thread in flight (the rows are 16-byte aligned; W[p] is byte p + 1 of
        // the row): as a byte-at-a-time loop every one of its 120 iterations waited for its own load
        const uint4 *row16 = reinterpret_cast<const uint4 *>(a.bitlog + (long long)s * a.bitlog_stride);
        const int n16 = (np + 1 + 15) / 16;
        for (int i0 = 0; i0 < n16; i0 += 256 * 8) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = i0 + 256 * u + (int)threadIdx.x;
                v[u] = row16[i < n16 ? i : n16 - 1];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = i0 + 256 * u + (int)threadIdx.x;
                if (i < n16) {
                    const int p0 = 16 * i - 1;  // W index of the quad's first byte
                    int r = (p0 + 80) % 80, q = (p0 + 80) / 80 - 1;
                    const unsigned w4[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                    for (int b = 0; b < 16; b++) {
                        const int p = p0 + b;
                        if (p >= 0 && p < np) T[r * row_stride + q] = (signed char)((w4[b >> 2] >> (8 * (b & 3))) & 0xffu);
                        r += 1;
                        if (r >= 80) {
                            r = 0;
                            q += 1;
                        }
                    }
                }
            }
        }
    }
    // packed sync vector: bytes 4i .. 4i+3 (zero beyond 64)
    int S4[17];
#pragma unroll
    for (int i = 0; i < 17; i++) {
        int v = 0;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (4 * i + k < SYNC_N) v |= ((int)c_bpsk.sync[4 * i + k] & 0xff) << (8 * k);
        S4[i] = v;
    }
    __syncthreads();
    {
        int b = threadIdx.x, r = b % 80, q0 = b / 80;
        for (; b < nb; b += 256) {
            const int *row = reinterpret_cast<const int *>(T + r * row_stride + (q0 & ~3));
            const int sh = q0 & 3;
            int d[18];
#pragma unroll
            for (int i = 0; i < 18; i++) d[i] = row[i];
            int c = 0;
#pragma unroll
            for (int i = 0; i < 17; i++) {
                const int w = (int)__builtin_amdgcn_alignbyte((unsigned)d[i + 1], (unsigned)d[i], (unsigned)sh);
                c = __builtin_amdgcn_sdot4(w, S4[i], c, false);
            }
            a.corr[(long long)s * a.max_bits + b] = (signed char)c;
            r += 256 % 80;
            q0 += 256 / 80;
            if (r >= 80) {
                r -= 80;
                q0 += 1;
            }
        }
    }
    if (!f.fuse) return;  // (uniform)
    __threadfence();  // release: every wave's correlations are visible device-wide ...
    __syncthreads();
    if (threadIdx.x < 64) {
        __threadfence();  // ... acquire: and read from there, not from a stale L1 line
        sync_fin_wave(s, (int)threadIdx.x, nb, a.corr + (long long)s * a.max_bits, f.trig_count, f.trig_bits, f.trig_cap, f.st);
    }
}

// the hits (correlation >= 45, :560) in bit order -- one wave per stream walks the correlations 64 at a time, a ballot
// and a prefix count give every hit its slot: deterministic, and when a call holds more hits than the handle has room
// for it is the FIRST trig_cap that are kept (the stream is flagged; the getters then fail instead of returning a
// truncated result).  Then dmCorr / dmMaxCorr exactly as the serial loop leaves them (:556-572): after a hit dmMaxCorr
// restarts from 0 (:567) and immediately takes that bit's correlation (:571-572).
__device__ __forceinline__ void sync_fin_wave(int s, int lane, int nb, const signed char *c, int *trig_count, int *trig_bits,
                                              int trig_cap, TailState *st)
{
    int nt = 0, last_hit = -1;
    for (int b0 = 0; b0 < nb; b0 += 64) {
        const int b = b0 + lane;
        const bool hit = b < nb && (int)c[b] >= 45;
        const unsigned long long m = __ballot(hit);
        if (hit) {
            const int slot = nt + __popcll(m & ((1ull << lane) - 1ull));
            if (slot < trig_cap) trig_bits[s * trig_cap + slot] = b;
        }
        if (m) last_hit = b0 + 63 - __clzll(m);
        nt += __popcll(m);
    }
    int from = 0, best = st[s].dmMaxCorr;
    if (nt > 0) {
        from = last_hit;
        best = 0;
    }
    for (int b = from + lane; b < nb; b += 64) best = best > (int)c[b] ? best : (int)c[b];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        int o = __shfl_xor(best, off, 64);
        best = best > o ? best : o;
    }
    if (lane == 0) {
        if (nb > 0) st[s].dmCorr = c[nb - 1];
        st[s].dmMaxCorr = best;
        st[s].cntFEC += nt;  // the reference counts every hit (:566), decoded or not
        if (nt > trig_cap) {
            st[s].overflow = 1;
            nt = trig_cap;
        }
        trig_count[s] = nt;
    }
}

// after the strided k_sync (the fallback for calls whose bit log does not fit a workgroup's LDS); k_sync_t does this itself
__global__ __launch_bounds__(64) void k_sync_fin(const int *nbits, const signed char *corr, int max_bits, int *trig_count,
                                                 int *trig_bits, int trig_cap, TailState *st, int nstreams)
{
    const int s = blockIdx.x, lane = threadIdx.x;
    if (s >= nstreams) return;
    sync_fin_wave(s, lane, nbits[s], corr + (long long)s * max_bits, trig_count, trig_bits, trig_cap, st);
}

}  // namespace jsdr

using namespace jsdr;

// =============================================================================================== host
// the input-independent schedule of one call: tuner / VCO table indices per sample (FUNcubeBPSKDemod.java:384-390, :511-516)
struct Schedule {
    bool valid = false;
    double tu0 = 0, vco0 = 0;  // phase state at the start of the call ...
    int ds0 = 0;
    long long L = -1;
    bool first = false;        // ... which is the first of the stream (history samples are zeros)
    unsigned char khist0[26] = {0};
    double tu1 = 0, vco1 = 0;  // state at its end
    int ds1 = 0, mix = 1, tper = 0;
    long long nds = 0;
    std::vector<unsigned char> ktu, kvco;
    std::vector<double2> tcs;
};

struct SnapPack {
    TailState t;
    int last[2];
    int cdec, nbits, centreBin, pad;
    double avePeakPower, aveCentreBin;
    unsigned char decoded[256];
    signed char bits[512];
};

struct SideJob;
struct jsdr_bpsk {
    int rate = 0, nsf = 0, tuning = 0, do_fft = 0, do_up = 0, nstreams = 0, decim = 0;
    long long max_batch = 0, max_ds = 0;
    int max_bits = 0;
    int trig_cap = MIN_TRIG;  // FECDecode calls (sync hits) one stream can log per call; sized from max_bits at create
    // input-independent scheduler state, exact doubles (FUNcubeBPSKDemod.java:381,:494,:501,:468)
    double tuPhase = 0.0, tuPhaseInc = 0.0, vcoPhase = 0.0;
    int dsCnt = 0;
    long long n_in = 0, n_ds = 0;  // samples consumed / demodulated since creation (cntRaw, cntDS)
    // one-entry schedule cache
    bool cache_valid = false;
    double c_tu0 = 0, c_vco0 = 0, c_tu1 = 0, c_vco1 = 0;
    int c_ds0 = 0, c_ds1 = 0;
    long long c_L = -1, c_nds = 0;
    std::vector<unsigned char> h_ktu;  // [26 history + L]
    unsigned char h_khist[26] = {0};   // tuner indices of the 26 samples before the next call
    unsigned char c_khist[26] = {0};
    int mix = 1, c_mix = 1;
    int c_kshift = -1;
    std::vector<unsigned char> h_kvco;
    // device
    DevBuf<double> sincos;
    DevBuf<unsigned char> ktu;
    DevBuf<unsigned char> kvco;
    DevBuf<int2> hist_in[2];
    int hist_cur = 0;
    bool hist_is_float = false;    // form of the samples in hist_in[hist_cur] (the input form of the call that wrote them)
    DevBuf<int> hist_bad;          // k_hist_convert's "not an int16 sample" flag
    DevBuf<int> amax;              // fast variant: [S] running maximum of |int16 sample| (float bits)
    DevBuf<SnapPack> snap_dev;     // receive(): the packed results of the call, fetched in one copy (k_snapshot_pack)
    DevBuf<int> fm_edges;          // k_fm: [S][4 * FM_EDGE] the stream around sample 0 and around the last sample (k_fm_edges)
    DevBuf<double2> dm, y[2];  // y is double-buffered: the tail of call k overlaps the front end of call k+1
    int y_cur = 0;
    // fused front end + matched filter (k_fm): the 64-sample halo lives in its own double buffer, the tuner table is
    // an unwrapped periodic (cos, sin) table
    DevBuf<double2> dmh[2], tcs;
    int dmh_cur = 0;
    int tab_cur = 0;               // which half of kvco / tcs holds the current schedule's tables
    bool halo_in_dmh = false;      // where the last call left the 64 VCO-mixed history samples (dm[s][0..63] or dmh)
    bool use_fm = true;            // JSDR_FM=0: always the three-kernel path
    long long last_fm_items = 0, last_fm_grid = 0;  // jsdr_bpsk_last_launch
    int share_wgs_per_cu = 0;      // jsdr_bpsk_set_cu_share: workgroups per CU k_fm is held to (0: one per tile, all the chip takes)
    int num_cu = 256;
    int variant = 0;               // 0 exact-order FP64, 1 fast (FMA-contracted FP64, margin-certified decisions)
    double fast_ey = 0.0;          // bound on the error of (fi,fq) in the fast variant (set at create from the taps)
    double margin_scale = 1.0;     // JSDR_FAST_MARGIN_SCALE: widens the detector margins (tests force the exact redo path with it)
    double argmax_scale = 1.0;     // JSDR_FAST_ARGMAX_SCALE: widens the argmax margin (tests provoke an uncertifiable stream)
    const char *front_name = "k_front";  // the front-end kernel the last call launched
    const char *tail_name = "k_tail";    // ... the tail kernel (k_tail / k_tail8) ...
    const char *fec_name = "k_fec_bpsk"; // ... and the FEC form (one wave per block, or the batch form's kernels)
    int c_tper = 0;                // period of the cached tuner schedule (0: not periodic with a period <= 256)
    bool ktu_uploaded = false;     // the device copy of the per-sample tuner index table matches the cached schedule
    std::vector<double2> h_tcs;
    Schedule prefetch;             // the next call's schedule, stepped on `worker` while the GPU runs the current call
    std::thread worker;
    bool prefetch_on = true;       // JSDR_SCHED_PREFETCH=0: always on the calling thread
    long long sched_sync = 0, sched_prefetched = 0;  // schedules computed on the calling thread / taken from the worker
    hipStream_t tail_stream = nullptr;   // non-blocking side stream for the latency-bound 9600 Hz tail + FEC
    hipEvent_t ev_matched = nullptr;     // caller stream -> tail stream: (fi,fq) of this call are complete
    hipEvent_t ev_tail_done[2] = {nullptr, nullptr};  // tail stream -> caller stream: y[i] may be overwritten
    bool tail_pending[2] = {false, false};
    // results of the last receive_*() of a 1-stream handle, double-buffered for a reader on another thread (the Swing
    // EDT paints while the audio thread receives): the writer fills the idle copy, then publishes it
    jsdr_bpsk_snapshot snap[2];
    std::atomic<unsigned> snap_seq[2];   // odd while that copy is being written
    std::atomic<int> snap_cur{-1};       // -1: nothing received yet
    long long snap_count = 0;
    hipEvent_t ev_pack_done = nullptr;   // pack stream -> tail stream: the result arrays of the previous call have been read
    bool pack_pending = false;
    bool overlap = true;
    long long dm_stride = 0, y_stride = 0;
    DevBuf<TailState> tail;
    DevBuf<signed char> bitlog[2];
    int bitlog_cur = 0;
    long long bitlog_stride = 0;
    DevBuf<int> nbits, trig_count, trig_bits, fec_rc, fec_last, cnt_dec;
    DevBuf<signed char> corr;
    DevBuf<unsigned char> fec_data, decoded;
    DevBuf<unsigned long long> fec_scratch;  // Viterbi decision words of every (stream, hit) block
    DevBuf<unsigned char> fec_vit;           // batch form (k_vitq): the Viterbi output bytes of every (stream, hit) block
    DevBuf<int> fec_work;                    // ... its work list, [0] = count
    DevBuf<int> fec_done;                    // [S] k_fec_bpsk's per-stream count of finished blocks (zero between launches)
    // receive_*() of a 1-stream handle: every host<->device copy of the call goes through ONE pinned arena (the frame in,
    // the schedule's tables when they change, the packed results out).  A copy from / to pageable memory is staged by the
    // runtime and costs a multiple of the transfer; the arena is reused every call, which is safe because receive()
    // synchronises before it returns.  Batch calls (asynchronous, caller-owned streams) keep the pageable path.
    unsigned char *pin = nullptr;
    size_t pin_bytes = 0, pin_off = 0;
    bool pin_call = false;
    bool vco_cs_in_blob = false;  // FFT-acquire mode: the current schedule's VCO factors sit behind the frame in stage_raw
    size_t vco_cs_blob_off = 0;
    size_t rx_frame_bytes = 0;  // receive(): the frame sits at the arena's head and has not been sent yet (bpsk_run sends it,
                                // with the schedule's tables behind it in the SAME copy when they changed)
    bool snap_fused = false;  // the last call's k_fec_bpsk packed the snapshot itself (receive() of a 1-stream handle)
    DevBuf<int> stage_raw;  // one frame for receive_*()
    DevBuf<FftFrontState> fft_state;  // FFT-acquire mode only
    DevBuf<double2> fft_tw;
    DevBuf<double> ds_taps_dev;
    DevBuf<double2> vco_cs;       // FFT-acquire mode: (cos, sin) of the VCO table entry of every decimated sample of the call
    std::vector<double> h_sincos;
    std::vector<double2> h_vco_cs;
    DevBuf<long long> phase_clk;  // JSDR_FFT_PHASECLK=1: k_front_fft's per-phase cycle counts, printed at destroy
    int logn = 0;
    bool fft_mixed = false;  // FFT-acquire frame is not a power of two (bpsk_fftm.hip)
    bool fft_2x = false;     // ... and is 2 m with m an LDS-sized frame (n = 19200): two m-point halves per transform
    AcqgPlan gen_plan;       // gen_plan.on: none of the LDS front ends takes the frame (or its decimation): bpsk_acqg.hip, always in three phases
    int fm_np = 0, fm_rad[12] = {0}, fm_off[12] = {0}, fm_off1[12] = {0};
    DevBuf<double2> fft2x_ek;  // per-stream scratch of the 2 m front end
    DevBuf<double> fft2x_r0;
    // round 6: the three-phase front end (bpsk_acq.hip) for calls of two or more frames per stream -- what its phases hand each
    // other, for nstreams x acq_chunk frames; allocated at the first such call
    DevBuf<unsigned char> acq_scratch;
    int acq_chunk = 0;
    // round 6, fast variant: the streams jsdr_bpsk_recover_uncertified() has replayed live on in an EXACT shadow handle (lock-step
    // with this one from then on); their getters and their packed slots come from it
    jsdr_bpsk *shadow = nullptr;
    std::vector<int> shadow_ids;   // ascending stream ids, index = the shadow's stream
    std::vector<int> shadow_map;   // [nstreams] index into the shadow, or -1
    DevBuf<int16_t> shadow_in;     // [U][2 max_batch] the shadow's rows of a call's input
    DevBuf<unsigned char> shadow_slots;
    long long batch_calls = 0;     // jsdr_bpsk_batch_i16 calls since creation (a recovery must be given all of them)
    long long recovered_events = 0;
    int acq_mode = -1;  // JSDR_ACQ3 (tests, A/B): 0 never, 1 whenever the frame size allows (also for one frame a call); -1: from two frames a call
    int num_cu_known = 0;
    long long last_nds = 0;
    int last_y = 0;
    hipStream_t last_stream = 0;
    // optional per-kernel HIP-event timing (bench.py's roofline leg)
    bool prof_on = false;
    struct ProfRec {
        int kernel;
        hipEvent_t a, b;
    };
    std::vector<ProfRec> prof_recs;
    std::vector<hipEvent_t> prof_pool;
};

enum { PK_FRONT = 0, PK_HIST, PK_MATCHED, PK_DMHIST, PK_TAIL, PK_SYNC, PK_SYNCFIN, PK_FEC, PK_FM, PK_SYNCT, PK_PREP,
       PK_ACQ_FWD, PK_ACQ_SCAN, PK_ACQ_INV, PK_ACQ_EDGES, PK_COUNT };
static const char *const kProfNames[PK_COUNT] = {"k_front", "k_hist_in", "k_matched", "k_dm_history", "k_tail", "k_sync",
                                                 "k_sync_fin", "k_fec_bpsk", "k_fm", "k_sync_t", "k_fm_prep",
                                                 "k_acq_fwd", "k_acq_scan", "k_acq_inv", "k_acq_edges"};

static hipEvent_t prof_event(jsdr_bpsk *h)
{
    if (!h->prof_pool.empty()) {
        hipEvent_t e = h->prof_pool.back();
        h->prof_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
struct ProfScope {
    jsdr_bpsk *h;
    hipStream_t st;
    hipEvent_t a = nullptr, b = nullptr;
    int k;
    ProfScope(jsdr_bpsk *h_, int k_, hipStream_t st_) : h(h_), st(st_), k(k_)
    {
        if (h->prof_on) {
            a = prof_event(h);
            b = prof_event(h);
            (void)hipEventRecord(a, st);
        }
    }
    ~ProfScope()
    {
        if (h->prof_on && a && b) {
            (void)hipEventRecord(b, st);
            h->prof_recs.push_back({k, a, b});
        }
    }
};

// the three-phase front end's launches (bpsk_acq.hip) under the same timing scopes
struct AcqProfCtx {
    jsdr_bpsk *h;
    hipEvent_t a[4];
};
static void acq_prof_mark(void *ctx, int phase, bool begin, hipStream_t st)
{
    AcqProfCtx *c = static_cast<AcqProfCtx *>(ctx);
    if (!c->h->prof_on) return;
    if (begin) {
        c->a[phase] = prof_event(c->h);
        (void)hipEventRecord(c->a[phase], st);
    } else {
        hipEvent_t b = prof_event(c->h);
        (void)hipEventRecord(b, st);
        c->h->prof_recs.push_back({PK_ACQ_FWD + phase, c->a[phase], b});
    }
}

static const double JPI = 3.14159265358979323846;

// FUNcubeBPSKDemod.java:384-390 / :511-516 -- advance the phase accumulators exactly as the reference does and record
// the table index each sample will use.  Input independent: a function of the phase state at the start of the call,
// the call's length and the configuration.  The recurrences round state-dependently (tuPhase += inc; wrap at 2 pi;
// truncate tuPhase*256/(2 pi)), so they are stepped one sample at a time in double, mul THEN div -- on the host: a
// single GPU lane needs ~35 cycles per link of this dependent FP64 chain (17 ms per 2^20 samples, against ~4 ms on a
// host core), and the links cannot be spread over lanes.  What keeps it off the critical path instead: the state
// repeats exactly for periodic configurations (cache hit, nothing computed), and for the others the schedule of the
// NEXT call is computed on a worker thread while the GPU works on this one (compute_schedule + the prefetch below).
static void compute_schedule(Schedule &sc, bool do_fft, double tuPhaseInc, int decim, const double *sincos)
{
    const double two_pi = 2.0 * JPI;
    const double vinc = 2.0 * JPI * 1200.0 / (double)9600;  // VCO_PHASE_INC (:88)
    const long long L = sc.L;
    sc.ktu.resize((size_t)L + 26);
    memcpy(sc.ktu.data(), sc.khist0, 26);
    sc.kvco.clear();
    sc.kvco.reserve((size_t)(L / decim + 2));
    double tu = sc.tu0, vco = sc.vco0;
    int cnt = sc.ds0;
    long long nmix = 0;
    unsigned char *kt = sc.ktu.data() + 26;
    for (long long n = 0; n < L; n++) {
        int k = 0;
        if (!do_fft) {  // doBufferFFT never runs the tuner (:406-464)
            tu += tuPhaseInc;
            if (tu > two_pi) tu -= two_pi;
            if (tu > 0.0) {  // :388
                k = (int)(tu * (double)256 / two_pi) % 256;
                nmix++;
            }
        }
        kt[n] = (unsigned char)k;
        if (++cnt >= decim) {
            cnt = 0;
            vco += vinc;
            if (vco > two_pi) vco -= two_pi;
            sc.kvco.push_back((unsigned char)((int)(vco * (double)256 / two_pi) % 256));
        }
    }
    // tuPhase > 0 holds for every sample (tuning > 0) or for none (tuning <= 0): one flag per call
    sc.mix = (nmix == L) ? 1 : (nmix == 0 ? 0 : -1);
    sc.tu1 = tu;
    sc.vco1 = vco;
    sc.ds1 = cnt;
    sc.nds = (long long)sc.kvco.size();
    // Is the tuner index periodic in the sample number?  (An exact 8-cycle at 12 kHz / 96 kHz.)  Candidate from the
    // head of the table, then verified over EVERY sample of the call, history included -- at the start of a stream
    // the 26 history samples are zeros, whose table entry does not matter.
    sc.tper = 0;
    if (!do_fft && sc.mix == 1) {
        const unsigned char *k = sc.ktu.data() + (sc.first ? 26 : 0);
        const long long len = L + (sc.first ? 0 : 26);
        const long long head = len < 1024 ? len : 1024;
        for (int p = 1; p <= 256 && p < len; p++) {
            if (memcmp(k, k + p, (size_t)(head - p)) != 0) continue;
            if (memcmp(k, k + p, (size_t)(len - p)) == 0) {
                sc.tper = p;
                // unwrapped table: entry e <-> samples n with (n + 26) mod p == e mod p
                const long long off = (sc.first ? 26 : 0);  // k[i] is the index of sample n = i + off - 26
                sc.tcs.resize((size_t)p + FM_TABLE_SLACK);
                for (int e = 0; e < p + FM_TABLE_SLACK; e++) {
                    const int i = (int)(((e - off) % p + p) % p);  // smallest i >= 0 with (i + off) mod p == e mod p
                    const int kk = k[i];
                    sc.tcs[(size_t)e] = make_double2(sincos[kk], sincos[256 + kk]);
                }
            }
            break;
        }
    }
    sc.valid = true;
}

static bool schedule_matches(const Schedule &sc, double tu, double vco, int ds, const unsigned char *khist, long long L, bool first)
{
    return sc.valid && sc.L == L && sc.tu0 == tu && sc.vco0 == vco && sc.ds0 == ds && sc.first == first &&
           memcmp(sc.khist0, khist, 26) == 0;
}

static void schedule_key(Schedule &sc, double tu, double vco, int ds, const unsigned char *khist, long long L, bool first)
{
    sc.valid = false;
    sc.tu0 = tu;
    sc.vco0 = vco;
    sc.ds0 = ds;
    sc.L = L;
    sc.first = first;
    memcpy(sc.khist0, khist, 26);
}

// Returns the number of decimated outputs of a call of L samples and leaves its schedule in the handle.
static long long build_schedule(jsdr_bpsk *h, long long L)
{
    if (h->cache_valid && h->c_L == L && h->c_tu0 == h->tuPhase && h->c_vco0 == h->vcoPhase && h->c_ds0 == h->dsCnt &&
        memcmp(h->c_khist, h->h_khist, 26) == 0) {
        h->tuPhase = h->c_tu1;
        h->vcoPhase = h->c_vco1;
        h->dsCnt = h->c_ds1;
        h->mix = h->c_mix;
        memcpy(h->h_khist, h->h_ktu.data() + L, 26);
        return h->c_nds;
    }
    if (h->worker.joinable()) h->worker.join();
    Schedule &pf = h->prefetch;
    const bool first = h->n_in == 0;
    if (!schedule_matches(pf, h->tuPhase, h->vcoPhase, h->dsCnt, h->h_khist, L, first)) {
        schedule_key(pf, h->tuPhase, h->vcoPhase, h->dsCnt, h->h_khist, L, first);
        compute_schedule(pf, h->do_fft != 0, h->tuPhaseInc, h->decim, h->h_sincos.data());
        h->sched_sync++;
    } else {
        h->sched_prefetched++;
    }
    // adopt it (the vectors change hands: the outgoing ones become the worker's scratch)
    h->c_tu0 = pf.tu0;
    h->c_vco0 = pf.vco0;
    h->c_ds0 = pf.ds0;
    memcpy(h->c_khist, pf.khist0, 26);
    h->h_ktu.swap(pf.ktu);
    h->h_kvco.swap(pf.kvco);
    if (pf.tper > 0) h->h_tcs.swap(pf.tcs);
    h->c_tper = pf.tper;
    h->mix = h->c_mix = pf.mix;
    h->tuPhase = h->c_tu1 = pf.tu1;
    h->vcoPhase = h->c_vco1 = pf.vco1;
    h->dsCnt = h->c_ds1 = pf.ds1;
    h->c_L = L;
    h->c_nds = pf.nds;
    memcpy(h->h_khist, h->h_ktu.data() + L, 26);
    h->cache_valid = false;  // device copy refreshed by the caller
    pf.valid = false;
    // the next call, assuming the same length: nothing to do if the state has come back to where this call started (the
    // cache will hit), otherwise step it on the worker thread while the GPU runs this call
    const bool comes_back = h->c_tu1 == h->c_tu0 && h->c_vco1 == h->c_vco0 && h->c_ds1 == h->c_ds0 &&
                            memcmp(h->h_khist, h->c_khist, 26) == 0;
    if (!comes_back && h->prefetch_on && L >= 65536) {  // (a short call's schedule costs less than starting a thread)
        schedule_key(pf, h->tuPhase, h->vcoPhase, h->dsCnt, h->h_khist, L, false);
        const bool do_fft = h->do_fft != 0;
        const double inc = h->tuPhaseInc;
        const int decim = h->decim;
        const double *sincos = h->h_sincos.data();
        Schedule *dst = &pf;
        h->worker = std::thread([dst, do_fft, inc, decim, sincos] { compute_schedule(*dst, do_fft, inc, decim, sincos); });
    }
    return h->c_nds;
}

// the bit clock (:581-584) must be the regular one the tail kernel assumes
static bool bit_clock_is_regular()
{
    const double inc = 1.0 / (double)9600, bt = 1.0 / (double)1200;
    double ph = 0.0;
    int pos = 0;
    for (int t = 0; t < 8 * 64; t++) {
        int expect = t & 7;
        if (pos != expect) return false;
        pos = (pos + 1) % 8;
        ph += inc;
        bool roll = false;
        if (ph >= bt) {
            ph -= bt;
            pos = 0;
            roll = true;
        }
        if (roll != (expect == 7)) return false;
        if (roll && ph != 0.0) return false;
    }
    return true;
}

template <int D, int RD, bool F32IN, bool MIX>
static void launch_front_t(const FrontArgs &fa, int nstreams, long long nds, hipStream_t st)
{
    using G = FrontGeom<D, RD>;
    constexpr int WAVES = 2;
    constexpr size_t elem = F32IN ? sizeof(float2) : sizeof(int);
    const size_t lds = 512 * sizeof(double) + WAVES * ((size_t)G::RAW_EL * elem + G::K_BYTES);
    long long ntiles = (nds + 64 * G::R - 1) / (64 * G::R);
    long long gx = (ntiles + WAVES - 1) / WAVES;
    if (gx > 2048) gx = 2048;
    if (gx < 1) gx = 1;
    (void)ensure_dynamic_lds(reinterpret_cast<const void *>(k_front<D, RD, F32IN, MIX>), lds);
    hipLaunchKernelGGL((k_front<D, RD, F32IN, MIX>), dim3((unsigned)gx, (unsigned)nstreams), dim3(64 * WAVES), lds, st,
                       fa);
}

static bool front_reg_enabled()
{
    static const bool reg = [] {
        const char *e = knob("JSDR_FRONT_REG");  // JSDR_FRONT_REG=0: the generic kernel instead
        return !e || atoi(e) != 0;
    }();
    return reg;
}

// can the register-staged kernel take this call?  (int16 input, 32-bit sample indices, at least one full window)
static bool front_reg_applies(const FrontArgs &fa)
{
    return front_reg_enabled() && !fa.rawf && fa.nsamples <= 0x3fffffffLL && fa.nsamples >= 64;
}

template <int D, int RD, bool MIX, bool DC, bool PER, bool FAST>
static void launch_front_reg_k(const FrontArgs &fa, int nstreams, long long nds, hipStream_t st)
{
    using G = FrontDmaGeom<D, RD>;
    // block size and tiles per wave make no difference between 1..4 waves and 1..10 tiles (swept; within 5 %)
    constexpr int WAVES = 4;
    const long long ntiles = (nds + 64 * G::R - 1) / (64 * G::R);
    long long gx = (ntiles + WAVES * 5 - 1) / (WAVES * 5);  // five tiles per wave
    if (gx > 2048) gx = 2048;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL((k_front_reg<D, RD, MIX, DC, PER, FAST>), dim3((unsigned)gx, (unsigned)nstreams), dim3(64 * WAVES), 0, st, fa);
}

// the register-staged fast path (int16 input) at every rate: 44.1 / 48 / 96 / 192 kHz
template <int D, int RD>
static bool launch_front_reg(const FrontArgs &fa, int nstreams, long long nds, bool fast, hipStream_t st)
{
    if (!front_reg_applies(fa)) return false;
    const bool dc = (fa.ic != 0) || (fa.qc != 0);
    const bool per = fa.mix && fa.tcs != nullptr;
#define JSDR_REG(MIX, DC, PER, FAST) launch_front_reg_k<D, RD, MIX, DC, PER, FAST>(fa, nstreams, nds, st)
    if (!fa.mix) {
        if (fast) { if (dc) JSDR_REG(false, true, false, true); else JSDR_REG(false, false, false, true); }
        else { if (dc) JSDR_REG(false, true, false, false); else JSDR_REG(false, false, false, false); }
    } else if (per) {
        if (fast) { if (dc) JSDR_REG(true, true, true, true); else JSDR_REG(true, false, true, true); }
        else { if (dc) JSDR_REG(true, true, true, false); else JSDR_REG(true, false, true, false); }
    } else {
        if (fast) { if (dc) JSDR_REG(true, true, false, true); else JSDR_REG(true, false, false, true); }
        else { if (dc) JSDR_REG(true, true, false, false); else JSDR_REG(true, false, false, false); }
    }
#undef JSDR_REG
    return true;
}

template <int D, int RD>
static void launch_front(const FrontArgs &fa, int nstreams, long long nds, hipStream_t st)
{
    if (fa.rawf) {
        if (fa.mix) launch_front_t<D, RD, true, true>(fa, nstreams, nds, st);
        else launch_front_t<D, RD, true, false>(fa, nstreams, nds, st);
    } else {
        if (fa.mix) launch_front_t<D, RD, false, true>(fa, nstreams, nds, st);
        else launch_front_t<D, RD, false, false>(fa, nstreams, nds, st);
    }
}


static thread_local long long g_fm_last_items = 0, g_fm_last_grid = 0;  // what the last k_fm launch of this thread covered (jsdr_bpsk_last_launch)

template <int D, int R>
static int launch_fm_t(const FmArgs &a_in, bool mix, bool dc, bool fast, int nstreams, hipStream_t st)
{
    const size_t lds = (size_t)FM_NT * sizeof(double2) + 512 * sizeof(double);
    const long long span = 65LL * FM_NB;
    const long long ntiles = (a_in.g_first + a_in.nds - a_in.tile0 + span - 1) / span;
    FmArgs a = a_in;
    a.ntiles = (int)ntiles;
    a.nstreams = nstreams;
    static const long long grid_cap = [] {
        const char *e = knob("JSDR_FM_GRID");  // tuning knob: total workgroups (default: one per tile)
        return e ? atoll(e) : 0LL;
    }();
    long long gx = ntiles * nstreams;
    if (grid_cap > 0 && gx > grid_cap) gx = grid_cap;
    if (a.grid_limit > 0 && gx > a.grid_limit) gx = a.grid_limit;
    g_fm_last_items = ntiles * nstreams;
    g_fm_last_grid = gx;
    const dim3 grid((unsigned)gx), block(FM_THREADS);
#define JSDR_FM_LAUNCH(MIX, DC, FAST)                                                                           \
    do {                                                                                                        \
        JSDR_LDS_ATTR((k_fm<D, R, MIX, DC, FAST>), lds);                                                        \
        hipLaunchKernelGGL((k_fm<D, R, MIX, DC, FAST>), grid, block, lds, st, a);                               \
    } while (0)
#define JSDR_FM_LAUNCH_SMALL(MIX, DC)                                                                           \
    do {                                                                                                        \
        JSDR_LDS_ATTR((k_fm<D, R, MIX, DC, false, true>), lds);                                                 \
        hipLaunchKernelGGL((k_fm<D, R, MIX, DC, false, true>), grid, block, lds, st, a);                        \
    } while (0)
    if (!fast && a.nds <= FM_THREADS && ntiles == 1) {  // a short call (receive()): the one-output-per-thread matched half
        if (mix) { if (dc) JSDR_FM_LAUNCH_SMALL(true, true); else JSDR_FM_LAUNCH_SMALL(true, false); }
        else { if (dc) JSDR_FM_LAUNCH_SMALL(false, true); else JSDR_FM_LAUNCH_SMALL(false, false); }
    } else if (fast) {
        if (mix) { if (dc) JSDR_FM_LAUNCH(true, true, true); else JSDR_FM_LAUNCH(true, false, true); }
        else { if (dc) JSDR_FM_LAUNCH(false, true, true); else JSDR_FM_LAUNCH(false, false, true); }
    } else {
        if (mix) { if (dc) JSDR_FM_LAUNCH(true, true, false); else JSDR_FM_LAUNCH(true, false, false); }
        else { if (dc) JSDR_FM_LAUNCH(false, true, false); else JSDR_FM_LAUNCH(false, false, false); }
    }
#undef JSDR_FM_LAUNCH
#undef JSDR_FM_LAUNCH_SMALL
    JSDR_LAUNCH_CHECK();
    return JSDR_OK;
}

static int launch_fm(const FmArgs &a, int decim, bool mix, bool dc, bool fast, int nstreams, hipStream_t st)
{
    switch (decim) {
        case 4: return launch_fm_t<4, 5>(a, mix, dc, fast, nstreams, st);
        case 5: return launch_fm_t<5, 4>(a, mix, dc, fast, nstreams, st);
        case 10: return launch_fm_t<10, 4>(a, mix, dc, fast, nstreams, st);
        case 20: return launch_fm_t<20, 4>(a, mix, dc, fast, nstreams, st);
    }
    set_error("bpsk: unsupported decimation %d", decim);
    return JSDR_ERR;
}

static int sync_last(jsdr_bpsk *h);
static int publish_snapshot(jsdr_bpsk *h);

// host -> device copy of a call's input or tables: through the pinned arena inside receive_*(), pageable otherwise
static int h2d_call(jsdr_bpsk *h, void *dst_dev, const void *src_host, size_t bytes, hipStream_t st)
{
    if (h->pin_call && h->pin) {
        const size_t off = (h->pin_off + 63) & ~(size_t)63;
        if (off + bytes <= h->pin_bytes) {
            memcpy(h->pin + off, src_host, bytes);
            h->pin_off = off + bytes;
            JSDR_HIP_TRY(hipMemcpyAsync(dst_dev, h->pin + off, bytes, hipMemcpyHostToDevice, st));
            return JSDR_OK;
        }
    }
    JSDR_HIP_TRY(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, st));
    return JSDR_OK;
}

// The side section of a call: the 9600 Hz tail, the sync correlation and the FEC of every hit, on the handle's side stream
// (the caller's when there is none), after the front end of THAT call.  Everything it needs from the call is in the job.
struct SideJob {
    bool valid = false;
    int yb = 0;
    long long nds = 0, g_first = 0;
    int first_out = 0, ic = 0, qc = 0;
    const int *raw = nullptr;
    long long stride_pairs = 0;
    unsigned char *kvco_p = nullptr;
    double2 *tcs_p = nullptr;
    hipStream_t st = nullptr;
};

static int run_side(jsdr_bpsk *h, const SideJob &j)
{
    const int S = h->nstreams;
    const int yb = j.yb;
    hipStream_t ts = h->overlap ? h->tail_stream : j.st;
    if (h->overlap) JSDR_HIP_TRY(hipStreamWaitEvent(ts, h->ev_matched, 0));
    if (h->pack_pending) {  // a pack of the previous call's results may still be reading what the tail section rewrites
        JSDR_HIP_TRY(hipStreamWaitEvent(ts, h->ev_pack_done, 0));
        h->pack_pending = false;
    }
    {
        TailArgs ta;
        ta.y = h->y[yb].p + Y_PAD;
        ta.y_stride = h->y_stride;
        ta.nds = j.nds;
        ta.g_first = j.g_first;
        ta.st = h->tail.p;
        ta.bitlog_new = h->bitlog[h->bitlog_cur ^ 1].p;
        ta.bitlog_old = h->bitlog[h->bitlog_cur].p;
        ta.bitlog_stride = h->bitlog_stride;
        ta.nbits = h->nbits.p;
        ta.max_bits = h->max_bits;
        ta.nstreams = S;
        ta.ey = h->fast_ey;
        ta.amax = h->amax.p;
        ta.margin_scale = h->margin_scale;
        ta.argmax_scale = h->argmax_scale;
        ta.raw = j.raw;
        ta.stride_pairs = j.stride_pairs;
        ta.ic = j.ic;
        ta.qc = j.qc;
        ta.decim = h->decim;
        ta.first_out = j.first_out;
        ta.mix = h->mix;
        ta.tper = (h->mix == 1) ? h->c_tper : 0;
        ta.tcs = j.tcs_p;
        ta.kvco = j.kvco_p;
        ta.sincos = h->sincos.p;
        ProfScope ps(h, PK_TAIL, ts);
        static const bool use_tail8 = [] {
            const char *e = knob("JSDR_TAIL8");  // JSDR_TAIL8=0: the one-wave-per-stream tail (A/B timing)
            return !e || atoi(e) != 0;
        }();
        static const bool force_tail8 = [] {
            const char *e = knob("JSDR_TAIL8");  // JSDR_TAIL8=2: k_tail8 whatever the number of streams (the tests' small handles)
            return e && atoi(e) == 2;
        }();
        h->tail_name = (h->variant == 0 || h->do_fft || !j.raw) && use_tail8 && (S >= 2048 || force_tail8) ? "k_tail8" : "k_tail";
        if (h->variant != 0 && !h->do_fft && j.raw)
            hipLaunchKernelGGL(k_tail<true>, dim3((unsigned)S), dim3(64), 0, ts, ta);
        else if (use_tail8 && (S >= 2048 || force_tail8))
            // (below ~2000 streams there are too few waves of eight streams to fill the chip and a wave's own latency per chunk
            //  decides: 1024 streams, FFT-acquire lines of round 4: locked 0.64 ms (k_tail) against 1.6, unlocked 5.6 against 3.5)
        {
            static const int wpb = [] {
                const char *e = knob("JSDR_TAIL8_WPB");  // JSDR_TAIL8_WPB=1: one-wave workgroups (A/B timing)
                return e ? atoi(e) : 4;
            }();
            const unsigned waves = (unsigned)((S + 7) / 8);
            if (wpb == 1)
                hipLaunchKernelGGL((k_tail8<16, 1>), dim3(waves), dim3(64), 0, ts, ta);
            else
                hipLaunchKernelGGL((k_tail8<16, 4>), dim3((waves + 3) / 4), dim3(256), 0, ts, ta);
        }
        else
            hipLaunchKernelGGL(k_tail<false>, dim3((unsigned)S), dim3(64), 0, ts, ta);
        JSDR_LAUNCH_CHECK();
        h->bitlog_cur ^= 1;
    }
    {
        SyncArgs sa;
        sa.bitlog = h->bitlog[h->bitlog_cur].p;
        sa.bitlog_stride = h->bitlog_stride;
        sa.nbits = h->nbits.p;
        sa.corr = h->corr.p;
        sa.max_bits = h->max_bits;
        int gx = (h->max_bits + 255) / 256;
        long long maxnew = j.nds / 4 + 8;
        if ((long long)gx * 256 > maxnew + 255) gx = (int)((maxnew + 255) / 256);
        if (gx < 1) gx = 1;
        {
            // transposed-image kernel when the stream's log fits a workgroup's LDS (always, up to ~8M samples a call)
            int cols = (HIST_BITS + h->max_bits + 79) / 80 + 72;  // + the 18 dwords an output reads past its first column
            int rs = (cols + 3) & ~3;
            if (((rs / 4) & 1) == 0) rs += 4;  // 4 * odd
            const size_t lds = (size_t)80 * rs + 16;
            static const bool use_t = [] {
                const char *e = knob("JSDR_SYNC_T");  // JSDR_SYNC_T=0: the strided kernel
                return !e || atoi(e) != 0;
            }();
            ProfScope ps(h, (use_t && lds <= 150 * 1024) ? PK_SYNCT : PK_SYNC, ts);  // timed under the name rocprof shows
            if (use_t && lds <= 150 * 1024) {
                JSDR_LDS_ATTR(k_sync_t, lds);
                SyncFinArgs sf;
                sf.trig_count = h->trig_count.p;
                sf.trig_bits = h->trig_bits.p;
                sf.trig_cap = h->trig_cap;
                sf.st = h->tail.p;
                // (the hand-over inside the workgroup is a device-scope release / acquire pair: an L2 write-back per
                //  workgroup on this multi-XCD part -- nothing for one stream, a tax beside the PSD kernel for thousands)
                sf.fuse = (S == 1 && j.nds <= 16384) ? 1 : 0;  // at most ~2000 new bits: the scan is a few dozen iterations
                hipLaunchKernelGGL(k_sync_t, dim3((unsigned)S), dim3(256), lds, ts, sa, rs, sf);
                JSDR_LAUNCH_CHECK();
                if (!sf.fuse) {
                    ProfScope ps2(h, PK_SYNCFIN, ts);
                    hipLaunchKernelGGL(k_sync_fin, dim3((unsigned)S), dim3(64), 0, ts, h->nbits.p, h->corr.p, h->max_bits,
                                       h->trig_count.p, h->trig_bits.p, h->trig_cap, h->tail.p, S);
                    JSDR_LAUNCH_CHECK();
                }
            } else {
                hipLaunchKernelGGL(k_sync, dim3((unsigned)gx, (unsigned)S), dim3(256), 0, ts, sa);
                JSDR_LAUNCH_CHECK();
                ProfScope ps2(h, PK_SYNCFIN, ts);
                hipLaunchKernelGGL(k_sync_fin, dim3((unsigned)S), dim3(64), 0, ts, h->nbits.p, h->corr.p, h->max_bits,
                                   h->trig_count.p, h->trig_bits.p, h->trig_cap, h->tail.p, S);
                JSDR_LAUNCH_CHECK();
            }
        }
        BpskFecArgs fa2;
        fa2.bitlog = h->bitlog[h->bitlog_cur].p;
        fa2.bitlog_stride = h->bitlog_stride;
        fa2.trig_count = h->trig_count.p;
        fa2.trig_bits = h->trig_bits.p;
        fa2.max_trig = h->trig_cap;
        fa2.decoded = h->decoded.p;
        fa2.fec_rc = h->fec_rc.p;
        fa2.fec_data = h->fec_data.p;
        fa2.last = h->fec_last.p;
        fa2.cnt_dec = h->cnt_dec.p;
        fa2.nstreams = S;
        fa2.dec_scratch = h->fec_scratch.p;
        fa2.done = h->fec_done.p;
        fa2.fuse = (S == 1) ? 1 : 0;
        fa2.ncopy = 0;
        fa2.vit = h->fec_vit.p;  // (null below VIT64_MIN_STREAMS: one wave per block)
        fa2.work_count = h->fec_work.p;
        fa2.work_list = h->fec_work.p ? h->fec_work.p + 1 : nullptr;
        h->snap_fused = false;
        if (S == 1 && h->pin_call) {  // receive(): the snapshot is packed by the block that completes the FEC work
            SnapPack *sp = h->snap_dev.p;
            auto add = [&](const void *src, void *dst, size_t bytes) {
                fa2.csrc[fa2.ncopy] = static_cast<const unsigned char *>(src);
                fa2.cdst[fa2.ncopy] = static_cast<unsigned char *>(dst);
                fa2.cbytes[fa2.ncopy] = (int)bytes;
                fa2.ncopy++;
            };
            add(h->tail.p, &sp->t, sizeof(TailState));
            add(h->fec_last.p, sp->last, 2 * sizeof(int));
            add(h->cnt_dec.p, &sp->cdec, sizeof(int));
            add(h->nbits.p, &sp->nbits, sizeof(int));
            if (h->do_fft) {
                add(&h->fft_state.p->centreBin, &sp->centreBin, sizeof(int));
                add(&h->fft_state.p->avePeakPower, &sp->avePeakPower, 2 * sizeof(double));  // avePeakPower, aveCentreBin
            }
            add(h->decoded.p, sp->decoded, 256);
            // the call's bits: what the log holds behind the history (the host clears the snapshot's bytes beyond nbits)
            const long long have = h->bitlog_stride - HIST_BITS;
            add(h->bitlog[h->bitlog_cur].p + HIST_BITS, sp->bits, (size_t)(have < 512 ? have : 512));
            h->snap_fused = true;
        }
        ProfScope ps(h, PK_FEC, ts);
        h->fec_name = (fa2.vit && !fa2.fuse) ? "k_fec_bits+k_vitq+k_fec_rs" : "k_fec_bpsk";
        if (launch_fec_bpsk(fa2, ts) != JSDR_OK) return JSDR_ERR;
    }
    if (h->overlap) {
        JSDR_HIP_TRY(hipEventRecord(h->ev_tail_done[yb], ts));
        h->tail_pending[yb] = true;
        h->y_cur ^= 1;
    }
    return JSDR_OK;
}

static int bpsk_run(jsdr_bpsk *h, const int16_t *raw_dev, const float *rawf_dev, long long stride_i16, long long L,
                    int ic, int qc, hipStream_t st)
{
    JSDR_REQUIRE(h, "bpsk: null handle");
    JSDR_REQUIRE(raw_dev || rawf_dev, "bpsk: null input");
    JSDR_REQUIRE(L > 0 && L <= h->max_batch, "bpsk: nsamples=%lld outside (0, max_batch_samples=%lld]", L, h->max_batch);
    JSDR_REQUIRE((stride_i16 & 1) == 0 && (h->nstreams == 1 || stride_i16 >= 2 * L),
                 "bpsk: stream stride %lld too small for %lld samples", stride_i16, L);
    JSDR_REQUIRE(!h->do_fft || (L % h->nsf) == 0, "bpsk: FFT-acquire mode needs whole frames (%lld %% %d != 0)", L, h->nsf);
    JSDR_REQUIRE(h->variant == 0 || raw_dev, "bpsk: the fast variant takes int16 input (its certification pass re-reads the raw samples)");
    const int first_out = h->decim - 1 - h->dsCnt;
    const long long g_first = h->n_ds;
    const long long nds = build_schedule(h, L);
    JSDR_REQUIRE(nds <= h->max_ds, "bpsk: internal: %lld decimated samples exceed capacity %lld", nds, h->max_ds);
    JSDR_REQUIRE(h->do_fft || h->mix >= 0, "bpsk: tuner phase changes sign inside a call (unsupported)");
    if (!h->do_fft) {
        const bool want_float = rawf_dev != nullptr;
        if (h->n_in > 0 && want_float != h->hist_is_float) {  // the previous call came through the other input form
            JSDR_HIP_TRY(hipMemsetAsync(h->hist_bad.p, 0, sizeof(int), st));
            hipLaunchKernelGGL(k_hist_convert, dim3((unsigned)((h->nstreams * 32 + 255) / 256)), dim3(256), 0, st,
                               h->hist_in[h->hist_cur].p, h->nstreams, want_float ? 1 : 0, h->hist_bad.p);
            JSDR_LAUNCH_CHECK();
            if (!want_float) {
                int bad = 0;
                JSDR_HIP_TRY(hipMemcpyAsync(&bad, h->hist_bad.p, sizeof(int), hipMemcpyDeviceToHost, st));
                JSDR_HIP_TRY(hipStreamSynchronize(st));
                JSDR_REQUIRE(!bad, "bpsk: int16 input after float frames whose samples are not (float)s/32767f values: the input "
                             "history cannot be carried over");
            }
        }
        h->hist_is_float = want_float;
    }
    // fused path (k_fm): int16 input, a tuner schedule that is periodic with a period dividing the lane span (or
    // no tuner at all), 32-bit sample indices
    const bool std_decim = h->decim == 4 || h->decim == 5 || h->decim == 10 || h->decim == 20;  // the specialised front ends
    const int fm_rd = h->decim == 4 ? 20 : h->decim * 4;  // D * R of the k_fm instantiation
    const bool per_ok = !h->do_fft && h->mix == 1 && h->c_tper > 0 && fm_rd % h->c_tper == 0;
    const bool fm_ok = h->use_fm && std_decim && !h->do_fft && raw_dev && !rawf_dev && nds > 0 && L <= 0x3fffffffLL &&
                       (h->mix == 0 || per_ok);
    const int kshift = 0;
    const bool fresh = !h->cache_valid;
    if (fresh) h->ktu_uploaded = false;
    if (fresh) {
        // The VCO / tuner tables are double-buffered: the fast variant's tail (side stream) may still re-read those of
        // the previous call.  The half written now was last used two schedules ago; the tail that read it is the one
        // that also frees y[y_cur], so waiting for that one (not for the previous call's) keeps the overlap.
        h->tab_cur ^= 1;
        if (h->variant != 0 && h->overlap && h->tail_pending[h->y_cur]) {
            JSDR_HIP_TRY(hipStreamWaitEvent(st, h->ev_tail_done[h->y_cur], 0));
            h->tail_pending[h->y_cur] = false;
        }
    }
    unsigned char *kvco_p = h->kvco.p + (size_t)h->tab_cur * (size_t)h->max_ds;
    double2 *tcs_p = h->tcs.p + (size_t)h->tab_cur * (256 + FM_TABLE_SLACK);
    // the 1 B/sample index table is only read by the kernels without the periodic table (k_front, k_front_reg<PER = false>)
    const bool reg_will_run = front_reg_enabled() && std_decim && raw_dev && !rawf_dev && L <= 0x3fffffffLL && L >= 64;
    const bool need_ktu = !h->do_fft && !fm_ok && !(per_ok && reg_will_run) && h->mix != 0;
    if (need_ktu && (!h->ktu_uploaded || kshift != h->c_kshift)) {
        h->c_kshift = kshift;
        if (h2d_call(h, h->ktu.p + kshift, h->h_ktu.data(), (size_t)L + 26, st) != JSDR_OK) return JSDR_ERR;
        h->ktu_uploaded = true;
    }
    ScatterArgs sc;
    memset(&sc, 0, sizeof(sc));
    bool tables_sent = false;
    if (h->rx_frame_bytes) {
        // receive(): the frame is at the arena's head.  When the call takes k_fm and its tables changed, they ride behind
        // the frame in the same copy and k_fm_prep scatters them (three copies were ~10 us each of a 70 us call)
        size_t total = h->rx_frame_bytes;
        const size_t o1 = (total + 63) & ~(size_t)63;
        const size_t tcs_bytes = h->c_tper > 0 ? sizeof(double2) * h->h_tcs.size() : 0;
        const size_t o2 = (o1 + (size_t)nds + 63) & ~(size_t)63;
        if (fm_ok && fresh && nds > 0 && o2 + tcs_bytes <= h->pin_bytes && o2 + tcs_bytes <= h->stage_raw.n * sizeof(int)) {
            unsigned char *dev = reinterpret_cast<unsigned char *>(h->stage_raw.p);
            memcpy(h->pin + o1, h->h_kvco.data(), (size_t)nds);
            sc.src[0] = dev + o1;
            sc.dst[0] = kvco_p;
            sc.bytes[0] = (int)nds;
            if (tcs_bytes) {
                memcpy(h->pin + o2, h->h_tcs.data(), tcs_bytes);
                sc.src[1] = dev + o2;
                sc.dst[1] = reinterpret_cast<unsigned char *>(tcs_p);
                sc.bytes[1] = (int)tcs_bytes;
            }
            total = o2 + tcs_bytes;
            h->pin_off = total;
            tables_sent = true;
        }
        else if (h->do_fft && fresh && nds > 0) {
            // FFT-acquire mode: the VCO factors of the call's outputs (the only table its front end reads) behind the frame
            // (at a FIXED offset behind the largest frame form -- a float frame: parked behind the int16 frame, the factors of
            //  a cached schedule were overwritten by the next float frame that failed the short-grid check, and read as they were)
            const size_t ov = (sizeof(float) * 2 * (size_t)h->nsf + 63) & ~(size_t)63, vb = sizeof(double2) * (size_t)nds;
            if (ov + vb <= h->pin_bytes && ov + vb <= h->stage_raw.n * sizeof(int)) {
                double2 *dst = reinterpret_cast<double2 *>(h->pin + ov);
                for (long long j = 0; j < nds; j++)
                    dst[j] = make_double2(h->h_sincos[h->h_kvco[(size_t)j]], h->h_sincos[256 + h->h_kvco[(size_t)j]]);
                total = ov + vb;
                h->pin_off = total;
                h->vco_cs_in_blob = true;
                h->vco_cs_blob_off = ov;
                tables_sent = true;
            }
        }
        JSDR_HIP_TRY(hipMemcpyAsync(h->stage_raw.p, h->pin, total, hipMemcpyHostToDevice, st));
        h->rx_frame_bytes = 0;
    }
    if (fresh && tables_sent) h->cache_valid = true;
    if (fresh && !tables_sent) {
        if (nds > 0)
            if (h2d_call(h, kvco_p, h->h_kvco.data(), (size_t)nds, st) != JSDR_OK) return JSDR_ERR;
        if (nds > 0 && h->do_fft) {
            h->vco_cs_in_blob = false;
            h->h_vco_cs.resize((size_t)nds);
            for (long long j = 0; j < nds; j++)
                h->h_vco_cs[(size_t)j] = make_double2(h->h_sincos[h->h_kvco[(size_t)j]], h->h_sincos[256 + h->h_kvco[(size_t)j]]);
            if (h2d_call(h, h->vco_cs.p, h->h_vco_cs.data(), sizeof(double2) * (size_t)nds, st) != JSDR_OK) return JSDR_ERR;
        }
        if (h->c_tper > 0)
            if (h2d_call(h, tcs_p, h->h_tcs.data(), sizeof(double2) * h->h_tcs.size(), st) != JSDR_OK) return JSDR_ERR;
        // the host vectors must stay untouched until the copies ran; pageable memcpyAsync stages
        // synchronously, so they are safe to reuse on return
        h->cache_valid = true;
    }
    // the 64-sample halo of VCO-mixed samples lives where the previous call's path left it
    if (nds > 0 && !h->do_fft && fm_ok != h->halo_in_dmh) {
        if (fm_ok)
            JSDR_HIP_TRY(hipMemcpy2DAsync(h->dmh[h->dmh_cur].p, 64 * sizeof(double2), h->dm.p, (size_t)h->dm_stride * sizeof(double2),
                                          64 * sizeof(double2), (size_t)h->nstreams, hipMemcpyDeviceToDevice, st));
        else
            JSDR_HIP_TRY(hipMemcpy2DAsync(h->dm.p, (size_t)h->dm_stride * sizeof(double2), h->dmh[h->dmh_cur].p, 64 * sizeof(double2),
                                          64 * sizeof(double2), (size_t)h->nstreams, hipMemcpyDeviceToDevice, st));
        h->halo_in_dmh = fm_ok;
    }
    const int S = h->nstreams;
    FrontArgs fa;
    fa.raw = reinterpret_cast<const int *>(raw_dev);
    fa.rawf = reinterpret_cast<const float2 *>(rawf_dev);
    fa.stride_pairs = stride_i16 / 2;
    fa.nsamples = L;
    fa.ic = ic;
    fa.qc = qc;
    fa.mix = h->mix;
    fa.ktu = h->ktu.p + kshift;
    fa.kvco = kvco_p;
    fa.sincos = h->sincos.p;
    fa.hist = h->hist_in[h->hist_cur].p;
    fa.dm = h->dm.p;
    fa.dm_stride = h->dm_stride;
    fa.ds_dbg = nullptr;
    fa.nds = nds;
    fa.first_out = first_out;
    fa.tcs = per_ok ? tcs_p : nullptr;
    fa.tper = h->c_tper;
    bool hist_done = false;  // the next call's input history has been written (k_fm_prep does it in the k_fm path)
    if (h->do_fft) {
        FftFrontArgs xa;
        xa.raw = fa.raw;
        xa.rawf = fa.rawf;
        xa.stride_pairs = fa.stride_pairs;
        xa.nframes = (int)(L / h->nsf);
        xa.n = h->nsf;
        xa.logn = h->logn;
        xa.ic = ic;
        xa.qc = qc;
        xa.do_up = h->do_up;
        xa.decim = h->decim;
        xa.first_out = first_out;
        xa.vco_cs = h->vco_cs_in_blob ? reinterpret_cast<const double2 *>(reinterpret_cast<const unsigned char *>(h->stage_raw.p) + h->vco_cs_blob_off)
                                      : h->vco_cs.p;
        xa.tw = h->fft_tw.p;
        xa.st = h->fft_state.p;
        xa.dm = h->dm.p;
        xa.dm_stride = h->dm_stride;
        xa.nds = nds;
        xa.ds_taps = h->ds_taps_dev.p;
        xa.phase_clk = h->phase_clk.p;
        // round 6: frames of 2^k samples, two or more per stream in the call: three phases over FRAMES (bpsk_acq.hip); a call of
        // one frame per stream (a live receive()) keeps the fused kernel -- one launch instead of four
        // The 2^k frames' three-phase kernels are the faster ones per frame as well (n = 2048, 1024 x 2^20: 8.65 against 9.1 ms);
        // the default mixed-radix frames' are the fused kernel's passes cut in two and cost 10-14 % more per frame at a full grid
        // (9600: 12.6 against 11.5 ms; the spectrum rows' round trip, a ticket a frame) -- they are taken where frames fill the
        // chip better than streams: ceil(S F / W) * 1.15 < ceil(S / W) * F, W = the workgroups the chip holds (one a CU).
        bool three = !h->fft_2x && acq3_supported(h->nsf) && h->acq_mode != 0 && (xa.nframes >= 2 || h->acq_mode == 1) &&
                     L < 0x7fffffffLL && (long long)S * ((xa.nframes + 1) / 2) < 0x7fffffffLL;  // (its kernels' 32-bit frame arithmetic)
        if (h->gen_plan.on) three = true;  // (no fused kernel exists for these frames)
        if (three && h->fft_mixed && h->acq_mode < 0) {
            if (h->num_cu_known == 0) {
                int dev = 0, cus = 0;
                JSDR_HIP_TRY(hipGetDevice(&dev));
                JSDR_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
                h->num_cu_known = cus > 0 ? cus : 256;
            }
            const long long W = h->num_cu_known, F = xa.nframes;
            // (4800 / 4410: the fused kernel takes two frames at once and is 1.38 / 1.27 x the three-phase kernels' pace at a full grid)
            const double pace = (h->nsf == 4800 && F >= 2) ? 1.4 : (h->nsf == 4410 && F >= 2) ? 1.3 : 1.15;
            three = (double)(((long long)S * F + W - 1) / W) * pace < (double)((((long long)S + W - 1) / W) * F);
        }
        if (three) {
            const size_t per = acq3_frame_bytes(h->nsf, h->do_up) + 64 + (h->gen_plan.on ? acqg_image_bytes(h->nsf) : 0);
            if (!h->acq_scratch.p) {
                // sized for the largest call the handle takes, capped (JSDR_ACQ_SCRATCH_MB, default 6 GiB): longer calls go in
                // several launches of acq_chunk frames per stream
                long long cap_mb = 6144;
                if (const char *e = knob("JSDR_ACQ_SCRATCH_MB")) cap_mb = atoll(e) > 0 ? atoll(e) : cap_mb;
                long long fmax = h->max_batch / h->nsf;
                if (fmax < 1) fmax = 1;
                long long chunk = (cap_mb << 20) / (long long)(per * (size_t)S);
                if (chunk < 1) chunk = 1;
                if (chunk > fmax) chunk = fmax;
                if (const char *e = knob("JSDR_ACQ_CHUNK")) chunk = atoll(e) > 0 && atoll(e) < chunk ? atoll(e) : chunk;
                if (h->acq_scratch.alloc(per * (size_t)S * (size_t)chunk + 512) != JSDR_OK) return JSDR_ERR;
                h->acq_chunk = (int)chunk;
                int dev = 0, cus = 0;
                JSDR_HIP_TRY(hipGetDevice(&dev));
                JSDR_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
                h->num_cu = cus > 0 ? cus : 256;
            }
            h->front_name = h->gen_plan.on ? "k_acqg_pass" : h->fft_mixed ? "k_acqm_fwd" : "k_acq_fwd";
            AcqmPlan plan;
            plan.np = h->fm_np;
            plan.rad = h->fm_rad;
            plan.tw_off = h->fm_off;
            plan.wr_off = h->fm_off1;
            AcqProfCtx pc{h, {nullptr, nullptr, nullptr, nullptr}};
            AcqProf prof;
            prof.ctx = &pc;
            prof.mark = acq_prof_mark;
            if (launch_acq3(xa, S, h->acq_scratch.p, h->acq_scratch.n, h->acq_chunk, h->num_cu, st, prof, plan, &h->gen_plan) != JSDR_OK) return JSDR_ERR;
        } else {
        ProfScope ps(h, PK_FRONT, st);
        h->front_name = h->fft_2x ? "k_front_fft2x" : (h->fft_mixed ? (fftm_pairs(xa.n, xa.nframes) ? "k_front_fftm2" : "k_front_fftm") : "k_front_fft");
        const int frc = h->fft_2x ? launch_front_fft2x(xa, h->fm_np, h->fm_rad, h->fm_off, h->fm_off1, h->fft2x_ek.p, h->fft2x_r0.p, S, st)
                                  : (h->fft_mixed ? launch_front_fftm(xa, h->fm_np, h->fm_rad, h->fm_off, h->fm_off1, h->fft2x_ek.p, S, st) : launch_front_fft(xa, S, st));
        if (frc != JSDR_OK) return JSDR_ERR;
        }
    } else if (nds > 0 && fm_ok) {
        // wait for the tail that last read y[y_cur] (two calls ago) before the fused kernel overwrites it
        if (h->overlap && h->tail_pending[h->y_cur]) {
            JSDR_HIP_TRY(hipStreamWaitEvent(st, h->ev_tail_done[h->y_cur], 0));
            h->tail_pending[h->y_cur] = false;
        }
        FmArgs ma;
        ma.raw = fa.raw;
        ma.stride_pairs = fa.stride_pairs;
        ma.nsamples = (int)L;
        ma.ic = ic;
        ma.qc = qc;
        ma.edges = h->fm_edges.p;
        ma.tcs = tcs_p;
        ma.tper = h->mix ? h->c_tper : 1;
        ma.kvco = kvco_p;
        ma.sincos = h->sincos.p;
        ma.dmh_old = h->dmh[h->dmh_cur].p;
        ma.dmh_new = h->dmh[h->dmh_cur ^ 1].p;
        ma.y = h->y[h->y_cur].p + Y_PAD;
        ma.y_stride = h->y_stride;
        ma.nds = (int)nds;
        ma.g_first = g_first;
        ma.tile0 = g_first - (((g_first - 64) % 65 + 65) % 65);
        ma.first_out = first_out;
        ma.amax = h->amax.p;
        ma.grid_limit = h->share_wgs_per_cu > 0 ? h->share_wgs_per_cu * h->num_cu : 0;
        {
            EdgeArgs ea;
            ea.raw = fa.raw;
            ea.stride_pairs = fa.stride_pairs;
            ea.nsamples = (int)L;
            ea.ic = ic;
            ea.qc = qc;
            ea.dc = (ic != 0) || (qc != 0);
            ea.hist = fa.hist;
            ea.edges = h->fm_edges.p;
            ea.nstreams = S;
            HistArgs ha;  // the next call's 26-sample input history, in the same launch (k_hist_in's work)
            ha.raw = fa.raw;
            ha.rawf = nullptr;
            ha.stride_pairs = fa.stride_pairs;
            ha.nsamples = L;
            ha.ic = ic;
            ha.qc = qc;
            ha.hist_old = h->hist_in[h->hist_cur].p;
            ha.hist_new = h->hist_in[h->hist_cur ^ 1].p;
            ha.nstreams = S;
            ProfScope psh(h, PK_PREP, st);
            hipLaunchKernelGGL(k_fm_prep, dim3((unsigned)(((long long)S * (4 * FM_EDGE + 32) + sc.bytes[0] + sc.bytes[1] + 255) / 256)),
                               dim3(256), 0, st, ea, ha, sc);
            JSDR_LAUNCH_CHECK();
            hist_done = true;
        }
        ProfScope ps(h, PK_FM, st);
        h->front_name = "k_fm";
        if (launch_fm(ma, h->decim, h->mix != 0, (ic != 0) || (qc != 0), h->variant != 0, S, st) != JSDR_OK) return JSDR_ERR;
        h->last_fm_items = g_fm_last_items;
        h->last_fm_grid = g_fm_last_grid;
        h->dmh_cur ^= 1;
    } else if (nds > 0) {
        ProfScope ps(h, PK_FRONT, st);
        const bool fast = false;  // (a fast handle whose call cannot take k_fm runs it in exact order: the amplitude the
                                  //  fast variant's bound scales with is tracked by k_fm only)
        h->front_name = "k_front";
        switch (h->decim) {
            case 4:
                if (launch_front_reg<4, 20>(fa, S, nds, fast, st)) h->front_name = "k_front_reg";
                else launch_front<4, 40>(fa, S, nds, st);
                break;
            case 5:
                if (launch_front_reg<5, 20>(fa, S, nds, fast, st)) h->front_name = "k_front_reg";
                else launch_front<5, 40>(fa, S, nds, st);
                break;
            case 10:
                if (launch_front_reg<10, 40>(fa, S, nds, fast, st)) h->front_name = "k_front_reg";
                else launch_front<10, 40>(fa, S, nds, st);
                break;
            case 20:
                if (launch_front_reg<20, 80>(fa, S, nds, fast, st)) h->front_name = "k_front_reg";
                else launch_front<20, 80>(fa, S, nds, st);
                break;
            default: {  // any other rate
                h->front_name = "k_front_any";
                long long gx = (nds + 255) / 256;
                if (gx > 4096) gx = 4096;
                if (fa.rawf)
                    hipLaunchKernelGGL(k_front_any<true>, dim3((unsigned)gx, (unsigned)S), dim3(256), 0, st, fa, h->decim);
                else
                    hipLaunchKernelGGL(k_front_any<false>, dim3((unsigned)gx, (unsigned)S), dim3(256), 0, st, fa, h->decim);
            }
        }
        JSDR_LAUNCH_CHECK();
    }
    if (!h->do_fft && hist_done) h->hist_cur ^= 1;
    if (!h->do_fft && !hist_done) {
        HistArgs ha;
        ha.raw = fa.raw;
        ha.rawf = fa.rawf;
        ha.stride_pairs = fa.stride_pairs;
        ha.nsamples = L;
        ha.ic = ic;
        ha.qc = qc;
        ha.hist_old = h->hist_in[h->hist_cur].p;
        ha.hist_new = h->hist_in[h->hist_cur ^ 1].p;
        ha.nstreams = S;
        ProfScope ps(h, PK_HIST, st);
        hipLaunchKernelGGL(k_hist_in, dim3((unsigned)((S * 32 + 255) / 256)), dim3(256), 0, st, ha);
        JSDR_LAUNCH_CHECK();
        h->hist_cur ^= 1;
    }
    const int yb = h->y_cur;
    hipStream_t ts = h->overlap ? h->tail_stream : st;
    if (h->overlap && h->tail_pending[yb]) {
        // the tail that last read y[yb] (two calls ago) must be done before the matched filter overwrites it
        JSDR_HIP_TRY(hipStreamWaitEvent(st, h->ev_tail_done[yb], 0));
        h->tail_pending[yb] = false;
    }
    if (nds > 0 && !(fm_ok && !h->do_fft)) {
        MatchedArgs ma;
        ma.dm = h->dm.p;
        ma.dm_stride = h->dm_stride;
        ma.y = h->y[yb].p + Y_PAD;
        ma.y_stride = h->y_stride;
        ma.nds = nds;
        ma.g_first = g_first;
        // first block boundary (slot 0 <=> g == 64 mod 65) at or before g_first; may be "negative" for g < 64
        long long b0 = g_first - (((g_first - 64) % 65 + 65) % 65);
        ma.tile0 = b0;
        long long ntiles = (g_first + nds - b0 + 4159) / 4160;
        const size_t lds = (64 + 4160) * sizeof(double2);
        JSDR_LDS_ATTR(k_matched<false>, lds);
        JSDR_LDS_ATTR(k_matched<true>, lds);
        {
            ProfScope ps(h, PK_MATCHED, st);
            hipLaunchKernelGGL(k_matched<false>, dim3((unsigned)ntiles, (unsigned)S), dim3(512), lds, st, ma);
        }
        JSDR_LAUNCH_CHECK();
        ProfScope ps2(h, PK_DMHIST, st);
        hipLaunchKernelGGL(k_dm_history, dim3((unsigned)S), dim3(64), 0, st, h->dm.p, h->dm_stride, nds, S);
        JSDR_LAUNCH_CHECK();
    }
    {
        SideJob job;
        job.valid = true;
        job.yb = yb;
        job.nds = nds;
        job.g_first = g_first;
        job.first_out = first_out;
        job.ic = ic;
        job.qc = qc;
        job.raw = fa.raw;
        job.stride_pairs = fa.stride_pairs;
        job.kvco_p = kvco_p;
        job.tcs_p = tcs_p;
        job.st = st;
        if (h->overlap) JSDR_HIP_TRY(hipEventRecord(h->ev_matched, st));
        // (the side section goes out now.  In the pipeline it then waits behind the NEXT call's PSD kernel anyway -- k_fft holds
        //  every register of every SIMD, a tail wave finds no room until it ends -- and runs beside the first 5 ms of that call's
        //  k_fm: profiles/r04_b_timeline.txt.  Holding it back on the host until the next call's start measured slower.)
        if (run_side(h, job) != JSDR_OK) return JSDR_ERR;
    }
    h->last_y = yb;
    h->n_in += L;
    h->n_ds += nds;
    h->last_nds = nds;
    h->last_stream = st;
    return JSDR_OK;
}

extern "C" {

int jsdr_bpsk_create(jsdr_bpsk **out, int rate, int nsamples_per_frame, int tuning_hz, int do_fft, int do_up,
                     int nstreams, int64_t max_batch_samples)
{
    JSDR_REQUIRE(out, "jsdr_bpsk_create: null handle pointer");
    *out = nullptr;
    JSDR_REQUIRE(rate >= 1, "jsdr_bpsk_create: rate %d", rate);
    // adsc.rate/DOWN_SAMPLE_RATE (:476), int division.  Below 9600 Hz (an 8 kHz card) the quotient is 0 and `++dsCnt >= 0`
    // holds for every sample: RxDownSample filters at every input, which is what a decimation of 1 does
    const int decim = rate / 9600 > 0 ? rate / 9600 : 1;
    // any rate the reference would take (:476: adsc.rate / DOWN_SAMPLE_RATE, whatever it is); 4, 5, 10, 20 -- the rates
    // java-sdr has defaults for -- take the specialised front ends, everything else the one-thread-per-output kernel.
    // FFT-acquire mode: the power-of-two and the 2 m front ends size their per-thread output lists for a decimation of at
    // least 4; the mixed-radix one (any other frame up to 9600 samples) loops and takes any.
    // Round 6: whatever those refuse -- and any other frame the oracle defines -- goes through the any-frame passes (bpsk_acqg.hip).
    JSDR_REQUIRE(nsamples_per_frame > 0 && nstreams > 0 && nstreams <= 65535, "jsdr_bpsk_create: bad geometry");
    if (max_batch_samples < nsamples_per_frame) max_batch_samples = nsamples_per_frame;
    const bool fft_pow2 = nsamples_per_frame >= 1024 && nsamples_per_frame <= 8192 &&
                          (nsamples_per_frame & (nsamples_per_frame - 1)) == 0;
    const bool fft_lds = do_fft && (fftm_supported(nsamples_per_frame) || (decim >= 4 && (fft_pow2 || fft2x_supported(nsamples_per_frame))));
    bool fft_gen = do_fft && !fft_lds;
    if (const char *e = knob("JSDR_ACQG")) fft_gen = do_fft && atoi(e) != 0;  // (tests: the any-frame passes for a frame the LDS kernels take)
    JSDR_REQUIRE(!fft_gen || acqg_supported(nsamples_per_frame),
                 "jsdr_bpsk_create: FFT-acquire mode needs a frame of 416 .. 4194304 samples whose prime factors r above 7 keep n r within "
                 "2^31 (got %d): below 416 the 204 gathered bins (FUNcubeBPSKDemod.java:458) do not end inside the frame",
                 nsamples_per_frame);
    JSDR_REQUIRE(bit_clock_is_regular(), "jsdr_bpsk_create: bit clock schedule is not the regular 8-cycle");
    if (fec_prepare() != JSDR_OK) return JSDR_ERR;
    jsdr_bpsk *h = new jsdr_bpsk();
    h->snap_seq[0].store(0);
    h->snap_seq[1].store(0);
    memset(h->snap, 0, sizeof(h->snap));
    h->rate = rate;
    h->nsf = nsamples_per_frame;
    h->tuning = tuning_hz;
    h->do_fft = do_fft;
    h->do_up = do_up;
    h->nstreams = nstreams;
    h->decim = decim;
    h->max_batch = max_batch_samples;
    h->max_ds = max_batch_samples / decim + 2;
    h->max_bits = (int)(h->max_ds / 4 + 16);
    // a legitimate frame yields one hit per 5200 bits; leave room for false alarms (FUNcubeBPSKDemod.java:560 has no limit,
    // so a call with more hits than this flags the stream instead of returning a truncated log)
    h->trig_cap = h->max_bits / 2600 + 4;
    if (h->trig_cap < MIN_TRIG) h->trig_cap = MIN_TRIG;
    h->tuPhaseInc = 2.0 * JPI * (double)tuning_hz / (double)rate;  // :196
    while ((1 << h->logn) < nsamples_per_frame) h->logn++;
    if (do_fft) h->max_batch = (h->max_batch / nsamples_per_frame) * nsamples_per_frame;
    // one stream (the receive() drop-in): nothing of another stream to run beside the tail, and the hop to the side
    // stream costs a cross-stream event per call
    if (nstreams == 1) h->overlap = false;
    // FFT-acquire mode: the front ends hold a CU's whole LDS (one workgroup of a mixed-radix frame, four of a power-of-two
    // frame), so the side stream's kernels cannot run beside them -- they starve and delay it (measured, one session each,
    // tools/ab_overlap_acq.sh / ab_env.sh: a step 14.0 vs 14.9 ms at n = 9600, 17.5 vs 18.1 at 4800, no difference at 19200;
    // round 4: 11.4 vs 12.15 at n = 2048 -- k_sync_t takes 5.1 ms beside k_front_fft against 0.11 alone -- 11.6 vs 12.5 at 4096)
    if (do_fft) h->overlap = false;
    // round 6: beside the three-phase front end's kernels of the 1024- and 2048-sample frames (128-thread workgroups, two waves a
    // SIMD) the side section does overlap usefully -- the tail / sync / FEC of call k under the forward kernel of call k + 1: a step
    // 9.97 against 11.25 ms at n = 2048 (1024 x 2^20), 10.50 / 12.28 at 1024, 1.16 / 1.61 at 64 streams; at n = 4096 it loses again
    // (12.29 against 10.87) -- so a handle of such frames that takes calls of several frames keeps its side stream
    if (do_fft && nstreams > 1 && (nsamples_per_frame == 1024 || nsamples_per_frame == 2048) && max_batch_samples >= 2LL * nsamples_per_frame)
        h->overlap = true;
    if (const char *e = knob("JSDR_NO_OVERLAP")) h->overlap = atoi(e) == 0;
    if (const char *e = knob("JSDR_FM")) h->use_fm = atoi(e) != 0;
    if (const char *e = knob("JSDR_SCHED_PREFETCH")) h->prefetch_on = atoi(e) != 0;
    const size_t S = (size_t)nstreams;
    // FEC of a batch handle: the lane-per-block Viterbi (fec.hip, k_vitq) once there are enough blocks to fill waves of 64;
    // below that (and for the 1-stream receive() form, whose latency counts) one wave per block
    bool vitq = nstreams >= 256;
    if (const char *e = knob("JSDR_VITQ")) vitq = atoi(e) != 0 && nstreams > 1;
    h->dm_stride = 64 + h->max_ds + 64;
    h->y_stride = h->max_ds;
    h->bitlog_stride = (HIST_BITS + h->max_bits + 64 + 15) & ~15LL;  // (rows 16-byte aligned: k_tail8 carries the register over in dwords)
    bool ok = h->sincos.alloc(512) == JSDR_OK && h->ktu.alloc((size_t)h->max_batch + 8192) == JSDR_OK &&
              h->kvco.alloc(2 * (size_t)h->max_ds) == JSDR_OK && h->hist_in[0].alloc(S * 32) == JSDR_OK &&
              h->hist_in[1].alloc(S * 32) == JSDR_OK && h->dm.alloc(S * (size_t)h->dm_stride) == JSDR_OK &&
              h->y[0].alloc(S * (size_t)h->y_stride + 2 * Y_PAD) == JSDR_OK && h->y[1].alloc(S * (size_t)h->y_stride + 2 * Y_PAD) == JSDR_OK && h->tail.alloc(S) == JSDR_OK &&
              h->bitlog[0].alloc(S * (size_t)h->bitlog_stride) == JSDR_OK &&
              h->bitlog[1].alloc(S * (size_t)h->bitlog_stride) == JSDR_OK && h->nbits.alloc(S) == JSDR_OK &&
              h->trig_count.alloc(S) == JSDR_OK && h->trig_bits.alloc(S * h->trig_cap) == JSDR_OK &&
              h->fec_scratch.alloc(vitq ? (size_t)fec_vitq_scratch_words(nstreams, h->trig_cap) : S * h->trig_cap * (size_t)fec_dec_scratch_words()) == JSDR_OK && h->fec_done.alloc(S) == JSDR_OK &&
              (!vitq || (h->fec_vit.alloc(S * h->trig_cap * 320) == JSDR_OK && h->fec_work.alloc(S * h->trig_cap + 1) == JSDR_OK)) &&
              h->fec_rc.alloc(S * h->trig_cap) == JSDR_OK && h->fec_last.alloc(S * 2) == JSDR_OK &&
              h->cnt_dec.alloc(S) == JSDR_OK && h->corr.alloc(S * (size_t)h->max_bits) == JSDR_OK &&
              h->fec_data.alloc(S * h->trig_cap * 256) == JSDR_OK && h->decoded.alloc(S * 256) == JSDR_OK &&
              // (one frame; a 1-stream handle's receive() sends the schedule's tables behind it in the same copy)
              h->stage_raw.alloc((size_t)nsamples_per_frame * 2 + (nstreams == 1 ? ((do_fft ? sizeof(double2) : 1) * (size_t)h->max_ds + sizeof(double2) * (256 + FM_TABLE_SLACK) + 256) / 4 : 0)) == JSDR_OK &&
              h->ds_taps_dev.alloc(32) == JSDR_OK &&
              h->dmh[0].alloc(S * 64) == JSDR_OK && h->dmh[1].alloc(S * 64) == JSDR_OK && h->hist_bad.alloc(1) == JSDR_OK && h->amax.alloc(S) == JSDR_OK && h->fm_edges.alloc(S * 4 * FM_EDGE) == JSDR_OK && h->snap_dev.alloc(1) == JSDR_OK && h->tcs.alloc(2 * (256 + FM_TABLE_SLACK)) == JSDR_OK &&
              (!do_fft || (h->fft_state.alloc(S) == JSDR_OK && h->fft_tw.alloc(fft_gen ? 3 * (size_t)nsamples_per_frame + 64 : fft_pow2 ? (size_t)nsamples_per_frame : (size_t)65536) == JSDR_OK &&
                            h->vco_cs.alloc((size_t)h->max_ds) == JSDR_OK)) &&
              (!(do_fft && fft2x_supported(nsamples_per_frame)) ||
               fft_gen ||
               (h->fft2x_ek.alloc(S * fft2x_scratch_ek(nsamples_per_frame)) == JSDR_OK &&
                h->fft2x_r0.alloc(S * fft2x_scratch_r0(nsamples_per_frame)) == JSDR_OK)) &&
              // (a mixed-radix frame with a prime factor above 7: the out-of-place pass's scratch, in the 2 m front end's slot)
              (!(do_fft && !fft_gen && !fft_pow2 && fftm_supported(nsamples_per_frame) && fftm_scratch(nsamples_per_frame) > 0) ||
               h->fft2x_ek.alloc(S * fftm_scratch(nsamples_per_frame)) == JSDR_OK);
    if (ok && nstreams == 1) {
        // [frame | ktu | kvco | vco_cs | tcs] + alignment slack, then the SnapPack slot (not handed out by h2d_call)
        const size_t need = sizeof(float) * 2 * (size_t)nsamples_per_frame + ((size_t)h->max_batch + 26) + (size_t)h->max_ds +
                            (do_fft ? sizeof(double2) * (size_t)h->max_ds : 0) + sizeof(double2) * (256 + FM_TABLE_SLACK) + 8 * 64;
        const size_t arena = (need + 63) & ~(size_t)63;
        void *pp = nullptr;
        if (arena <= ((size_t)8 << 20) && hipHostMalloc(&pp, arena + sizeof(SnapPack), hipHostMallocDefault) == hipSuccess) {
            h->pin = static_cast<unsigned char *>(pp);
            h->pin_bytes = arena;
        } else {
            (void)hipGetLastError();  // no pinned memory: the pageable path serves
        }
    }
    if (!ok) {
        jsdr_bpsk_destroy(h);
        return JSDR_ERR;
    }
    // tables (:159-162): Math.sin/cos are allowed 1 ulp; the host libm stands in (DESIGN.md "tables")
    std::vector<double> sc(512);
    for (int n = 0; n < 256; n++) {  // double argument as in Java, correctly rounded function value
        double arg = n * 2.0 * JPI / 256;
        sc[n] = (double)cosl((long double)arg);
        sc[256 + n] = (double)sinl((long double)arg);
    }
    BpskConst bc;
    memset(&bc, 0, sizeof(bc));
    for (int i = 0; i < 14; i++) {
        bc.ds_taps[i] = (double)h_ds_half[i];
        bc.ds_taps[26 - i] = (double)h_ds_half[i];
    }
    for (int i = 0; i < 33; i++) {
        bc.dm_taps[i] = (double)h_dm_half[i];
        bc.dm_taps[64 - i] = (double)h_dm_half[i];
    }
    int sr = 0x7f;  // sync LFSR == SYNC_VECTOR (:79-81; FECDecoder.java:600-605)
    for (int i = 0; i < SYNC_N; i++) {
        bc.sync[i] = (sr & 64) ? 1 : -1;
        int v = sr & 0x48;
        v ^= v >> 4;
        v ^= v >> 2;
        v ^= v >> 1;
        sr = ((sr << 1) | (v & 1)) & 0xffff;
    }
    if (do_fft) {
        std::vector<double2> tw;
        h->fft_mixed = !fft_pow2;
        h->fft_2x = !fft_pow2 && fft2x_supported(nsamples_per_frame);
        if (fft_gen) {
            h->fft_mixed = h->fft_2x = false;
            acqg_twiddles(tw, nsamples_per_frame, &h->gen_plan);
        } else if (fft_pow2)
            fft_twiddles_f64(tw, nsamples_per_frame);
        else if (h->fft_2x)
            fft2x_twiddles(tw, nsamples_per_frame, &h->fm_np, h->fm_rad, h->fm_off, h->fm_off1);
        else
            fftm_twiddles(tw, nsamples_per_frame, &h->fm_np, h->fm_rad, h->fm_off, h->fm_off1);  // (fm_off1: the prime radices' r-point tables)
        if (tw.size() > h->fft_tw.n || hipMemcpy(h->fft_tw.p, tw.data(), sizeof(double2) * tw.size(), hipMemcpyHostToDevice) != hipSuccess ||
            h->fft_state.zero() != JSDR_OK) {
            set_error("jsdr_bpsk_create: FFT-mode initialisation failed");
            jsdr_bpsk_destroy(h);
            return JSDR_ERR;
        }
        if (const char *e = knob("JSDR_ACQ3")) h->acq_mode = atoi(e) != 0 ? 1 : 0;
        if (const char *e = knob("JSDR_FFT_PHASECLK"))
            if (atoi(e) != 0 && (h->phase_clk.alloc(16 + 2 * 4096) != JSDR_OK || h->phase_clk.zero() != JSDR_OK)) h->phase_clk.release();
    }
    if (hipMemcpy(h->ds_taps_dev.p, bc.ds_taps, sizeof(double) * 27, hipMemcpyHostToDevice) != hipSuccess) {
        set_error("jsdr_bpsk_create: tap upload failed");
        jsdr_bpsk_destroy(h);
        return JSDR_ERR;
    }
    std::vector<TailState> ts(S);
    memset(ts.data(), 0, sizeof(TailState) * S);
    for (size_t i = 0; i < S; i++) {
        ts[i].dmEnergyOut = 1.0;  // :499
        ts[i].last_g = -1;
    }
    {
        // Worst-case error of (fi,fq) when both FIR stages use fused multiply-adds instead of the reference's separately
        // rounded products and sums (u = 2^-53; |x| <= 32768/32767, |cos|,|sin| <= 1):
        //   27-tap stage : |s' - s| <= (gamma_28 + gamma_27) T1,            T1 = sum |x_k| |t_k| <= 1.00004 sum|dsFilter|
        //   x HOWARD, VCO: |dm' - dm| <= 56 u T1 HOWARD + 4 u Dmax,         Dmax = HOWARD T1 bounds |dm|
        //   65-tap stage : |y' - y| <= (gamma_66 + gamma_65) F1 Dmax + F1 |dm' - dm|,   F1 = sum|dmFilter|
        double t1 = 0.0, f1 = 0.0;
        for (int i = 0; i < 27; i++) t1 += fabs(bc.ds_taps[i]);
        for (int i = 0; i < 65; i++) f1 += fabs(bc.dm_taps[i]);
        t1 *= 1.00004;
        const double U = 1.1102230246251565e-16, HOWARD = 0.9 * 32768.0, dmax = HOWARD * t1;
        h->fast_ey = 1.01 * U * f1 * (131.0 * dmax + 56.0 * t1 * HOWARD + 4.0 * dmax);
        if (const char *e = knob("JSDR_FAST_MARGIN_SCALE")) {
            const double v = atof(e);
            if (v >= 1.0) h->margin_scale = v;
        }
        if (const char *e = knob("JSDR_FAST_ARGMAX_SCALE")) {
            const double v = atof(e);
            if (v >= 1.0) h->argmax_scale = v;
        }
    }
    h->h_sincos = sc;
    bool up = hipMemcpy(h->sincos.p, sc.data(), sizeof(double) * 512, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpyToSymbol(HIP_SYMBOL(c_bpsk), &bc, sizeof(bc)) == hipSuccess &&
              hipMemcpy(h->tail.p, ts.data(), sizeof(TailState) * S, hipMemcpyHostToDevice) == hipSuccess &&
              h->hist_in[0].zero() == JSDR_OK && h->hist_in[1].zero() == JSDR_OK && h->dm.zero() == JSDR_OK &&
              h->dmh[0].zero() == JSDR_OK && h->dmh[1].zero() == JSDR_OK && h->tcs.zero() == JSDR_OK && h->amax.zero() == JSDR_OK &&
              h->bitlog[0].zero() == JSDR_OK && h->bitlog[1].zero() == JSDR_OK && h->decoded.zero() == JSDR_OK &&
              h->nbits.zero() == JSDR_OK && h->trig_count.zero() == JSDR_OK && h->fec_last.zero() == JSDR_OK && h->fec_done.zero() == JSDR_OK &&
              h->cnt_dec.zero() == JSDR_OK && h->y[0].zero() == JSDR_OK && h->y[1].zero() == JSDR_OK &&
              hipStreamCreateWithFlags(&h->tail_stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&h->ev_matched, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&h->ev_tail_done[0], hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&h->ev_tail_done[1], hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&h->ev_pack_done, hipEventDisableTiming) == hipSuccess;
    if (!up || hipDeviceSynchronize() != hipSuccess) {
        set_error("jsdr_bpsk_create: device initialisation failed");
        jsdr_bpsk_destroy(h);
        return JSDR_ERR;
    }
    *out = h;
    return JSDR_OK;
}

int jsdr_bpsk_destroy(jsdr_bpsk *h)
{
    if (!h) return JSDR_OK;
    if (h->worker.joinable()) h->worker.join();
    h->sincos.release();
    h->ktu.release();
    h->kvco.release();
    h->hist_in[0].release();
    h->hist_in[1].release();
    h->dm.release();
    h->hist_bad.release();
    h->amax.release();
    h->fm_edges.release();
    h->snap_dev.release();
    h->dmh[0].release();
    h->dmh[1].release();
    h->tcs.release();
    h->y[0].release();
    h->y[1].release();
    if (h->tail_stream) {
        (void)hipStreamSynchronize(h->tail_stream);
        (void)hipStreamDestroy(h->tail_stream);
    }
    if (h->ev_matched) (void)hipEventDestroy(h->ev_matched);
    if (h->ev_pack_done) (void)hipEventDestroy(h->ev_pack_done);
    for (int i = 0; i < 2; i++)
        if (h->ev_tail_done[i]) (void)hipEventDestroy(h->ev_tail_done[i]);
    h->tail.release();
    h->bitlog[0].release();
    h->bitlog[1].release();
    h->nbits.release();
    h->trig_count.release();
    h->trig_bits.release();
    h->fec_rc.release();
    h->fec_last.release();
    h->cnt_dec.release();
    h->corr.release();
    h->fec_data.release();
    h->fec_scratch.release();
    h->fec_vit.release();
    h->fec_work.release();
    h->fec_done.release();
    if (h->pin) (void)hipHostFree(h->pin);
    h->pin = nullptr;
    h->decoded.release();
    h->stage_raw.release();
    h->fft_state.release();
    h->fft_tw.release();
    h->fft2x_ek.release();
    h->fft2x_r0.release();
    h->acq_scratch.release();
    if (h->shadow) (void)jsdr_bpsk_destroy(h->shadow);
    h->shadow = nullptr;
    h->shadow_in.release();
    h->shadow_slots.release();
    h->vco_cs.release();
#ifdef JSDR_X_T8CLK
    {
        unsigned long long cc[8] = {0};
        if (hipMemcpyFromSymbol(cc, HIP_SYMBOL(g_t8_clk), sizeof(cc)) == hipSuccess) {
            static const char *nm[6] = {"prologue", "A chains", "B argmax", "locked", "general", "write-back"};
            unsigned long long tot = 0;
            for (int i = 0; i < 6; i++) tot += cc[i];
            for (int i = 0; i < 6; i++) fprintf(stderr, "k_tail8 clk %-12s %14llu ticks %5.1f%%\n", nm[i], cc[i], 100.0 * cc[i] / (tot ? tot : 1));
            memset(cc, 0, sizeof(cc));
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_t8_clk), cc, sizeof(cc));
        }
    }
#endif
#ifdef JSDR_X_CLK
    {
        static unsigned long long cc[256][64];
        if (hipMemcpyFromSymbol(cc, HIP_SYMBOL(g_fm_clk), sizeof(cc)) == hipSuccess) {
            static const char *nm[6] = {"-", "front half", "-", "barrier front", "matched", "barrier matched"};
            for (int i = 0; i < 6; i++) {
                if (i == 0 || i == 2) continue;
                fprintf(stderr, "k_fm clk %-16s", nm[i]);
                for (int w = 0; w < 8; w++) {
                    unsigned long long c = 0, tot = 0;
                    for (int b = 0; b < 256; b++) {
                        c += cc[b][w * 8 + i];
                        for (int k = 0; k < 6; k++) tot += cc[b][w * 8 + k];
                    }
                    fprintf(stderr, " w%d %5.1f%%", w, 100.0 * c / (tot ? tot : 1));
                }
                fprintf(stderr, "\n");
            }
            memset(cc, 0, sizeof(cc));
            hipMemcpyToSymbol(HIP_SYMBOL(g_fm_clk), cc, sizeof(cc));
        }
    }
#endif
    if (h->phase_clk.p) {
        static const char *const names_p2[8] = {"load+scatter", "forward FFT", "|X|", "boxcar+argmax", "centre-bin rule",
                                                "gather/zero", "inverse FFT", "scale+RxDownSample"};
        static const char *const names_mx[8] = {"load", "forward FFT", "centre-bin rule", "gather/zero/place", "inverse FFT",
                                                "RxDownSample", "|X|", "boxcar+argmax"};
        const char *const *names = h->fft_mixed ? names_mx : names_p2;
        long long c[16] = {0};
        if (hipDeviceSynchronize() == hipSuccess &&
            hipMemcpy(c, h->phase_clk.p, sizeof(c), hipMemcpyDeviceToHost) == hipSuccess) {
            if (h->acq_chunk > 0) {  // the three-phase front end ran: the last launch's workgroup 0 of k_acq_fwd / k_acq_inv
                static const char *const nf[6] = {"convert+pass 1", "pass 2", "last pass: loads", "last pass+|X|+spec", "boxcar", "argmax+peak"};
                static const char *const ni[5] = {"gather", "passes 1+2 fused", "last pass", "compact store", "edges+RxDownSample"};
                long long tf = 0, ti = 0;
                for (int k = 0; k < 6; k++) tf += c[k];
                for (int k = 0; k < 5; k++) ti += c[8 + k];
                for (int k = 0; k < 6; k++)
                    fprintf(stderr, "[jsdr] k_acq_fwd phase %-20s %12lld ticks  %5.1f %%\n", nf[k], c[k], tf ? 100.0 * (double)c[k] / (double)tf : 0.0);
                for (int k = 0; k < 5; k++)
                    fprintf(stderr, "[jsdr] k_acq_inv phase %-20s %12lld ticks  %5.1f %%\n", ni[k], c[8 + k], ti ? 100.0 * (double)c[8 + k] / (double)ti : 0.0);
                memset(c, 0, sizeof(c));
                // every k_acq_fwd workgroup's first and last tick (100 MHz): how many ran from the start, how far apart they ended
                std::vector<long long> w(2 * 4096);
                if (hipMemcpy(w.data(), h->phase_clk.p + 16, sizeof(long long) * w.size(), hipMemcpyDeviceToHost) == hipSuccess) {
                    long long t0 = 0, e0 = 0, e1 = 0;
                    int n = 0, late = 0;
                    for (int i = 0; i < 4096; i++)
                        if (w[2 * i + 1]) {
                            if (!n || w[2 * i] < t0) t0 = w[2 * i];
                            if (!n || w[2 * i + 1] < e0) e0 = w[2 * i + 1];
                            if (!n || w[2 * i + 1] > e1) e1 = w[2 * i + 1];
                            n++;
                        }
                    for (int i = 0; i < 4096; i++)
                        if (w[2 * i + 1] && w[2 * i] - t0 > (e1 - t0) / 10) late++;
                    fprintf(stderr, "[jsdr] k_acq_fwd %d workgroups: %d started late (> 10 %% into the launch); first end %.3f ms, last end %.3f ms after the first start\n",
                            n, late, (double)(e0 - t0) / 1e5, (double)(e1 - t0) / 1e5);
                }
            }
            long long tot = 0;
            for (int k = 0; k < 8; k++) tot += c[k];
            for (int k = 0; k < 8; k++)
                fprintf(stderr, "[jsdr] k_front_fft phase %-20s %12lld ticks  %5.1f %%\n", names[k], c[k],
                        tot ? 100.0 * (double)c[k] / (double)tot : 0.0);
        }
        h->phase_clk.release();
    }
    h->ds_taps_dev.release();
    for (auto &r : h->prof_recs) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    for (auto e : h->prof_pool) (void)hipEventDestroy(e);
    delete h;
    return JSDR_OK;
}

// the shadow's rows of a call's input, gathered into its own stream-major buffer
static int shadow_stage(jsdr_bpsk *h, const int16_t *raw_dev, int64_t stride, int64_t nsamples, hipStream_t st)
{
    for (size_t i = 0; i < h->shadow_ids.size(); i++)
        JSDR_HIP_TRY(hipMemcpyAsync(h->shadow_in.p + i * (size_t)(2 * h->max_batch), raw_dev + (int64_t)h->shadow_ids[i] * stride,
                                    (size_t)nsamples * 4, hipMemcpyDeviceToDevice, st));
    return JSDR_OK;
}

int jsdr_bpsk_batch_i16(jsdr_bpsk *h, const int16_t *raw_dev, int64_t stream_stride_i16, int64_t nsamples, int ic,
                        int qc, void *stream)
{
    if (bpsk_run(h, raw_dev, nullptr, stream_stride_i16, nsamples, ic, qc, as_stream(stream)) != JSDR_OK) return JSDR_ERR;
    h->batch_calls++;
    if (h->shadow) {  // the recovered streams, in exact order, in lock-step
        if (shadow_stage(h, raw_dev, stream_stride_i16, nsamples, as_stream(stream)) != JSDR_OK) return JSDR_ERR;
        if (bpsk_run(h->shadow, h->shadow_in.p, nullptr, 2 * h->max_batch, nsamples, ic, qc, as_stream(stream)) != JSDR_OK) return JSDR_ERR;
    }
    return JSDR_OK;
}

// Fast variant: the streams the calls so far left uncertified are REPLAYED from the handle's first call on an internal exact
// handle (every uncertified stream at once, in lock-step), which serves them from then on: their getters and their packed slots
// come from it, the sticky flag no longer withholds them, and every later batch call advances it beside the fast kernels.
// raw_dev_calls[k] / nsamples_calls[k]: what jsdr_bpsk_batch_i16 was given at call k -- ALL calls since creation, with the
// buffers still holding those samples (a batch over recordings has them; a live source does not and uses the exact variant).
// Why from the first call: the decision that could not be certified depends on the exact energy history, and after a fast call
// no exact state exists to restart from.  Cost: one small exact handle's call per call replayed (latency bound,
// about a millisecond each), whatever the number of streams recovered.
int jsdr_bpsk_recover_uncertified(jsdr_bpsk *h, const int16_t *const *raw_dev_calls, const int64_t *nsamples_calls, int ncalls,
                                  int64_t stream_stride_i16, int ic, int qc, int *recovered, void *stream)
{
    JSDR_REQUIRE(h && (ncalls == 0 || (raw_dev_calls && nsamples_calls)), "jsdr_bpsk_recover_uncertified: null argument");
    if (recovered) *recovered = 0;
    if (h->variant == 0) return JSDR_OK;
    JSDR_REQUIRE((long long)ncalls == h->batch_calls, "jsdr_bpsk_recover_uncertified: %d calls given, the handle has seen %lld (all of "
                 "them are replayed)", ncalls, h->batch_calls);
    long long total = 0;
    for (int k = 0; k < ncalls; k++) {
        JSDR_REQUIRE(raw_dev_calls[k] && nsamples_calls[k] >= 0 && nsamples_calls[k] <= h->max_batch, "jsdr_bpsk_recover_uncertified: call %d", k);
        total += nsamples_calls[k];
    }
    JSDR_REQUIRE(total == h->n_in, "jsdr_bpsk_recover_uncertified: the calls given hold %lld samples per stream, the handle has taken %lld",
                 total, h->n_in);
    if (sync_last(h) != JSDR_OK) return JSDR_ERR;
    std::vector<TailState> ts((size_t)h->nstreams);
    JSDR_HIP_TRY(hipMemcpy(ts.data(), h->tail.p, sizeof(TailState) * ts.size(), hipMemcpyDeviceToHost));
    std::vector<int> ids;
    int fresh = 0;
    for (int s = 0; s < h->nstreams; s++)
        if (ts[(size_t)s].uncertified) {
            ids.push_back(s);
            if (h->shadow_map.empty() || h->shadow_map[(size_t)s] < 0) fresh++;
        }
    if (fresh == 0) return JSDR_OK;  // nothing new to recover (the shadow, if any, is up to date)
    jsdr_bpsk *sh = nullptr;
    if (jsdr_bpsk_create(&sh, h->rate, h->nsf, h->tuning, 0, h->do_up, (int)ids.size(), h->max_batch) != JSDR_OK) return JSDR_ERR;
    if (h->shadow) (void)jsdr_bpsk_destroy(h->shadow);
    h->shadow = nullptr;
    h->shadow_ids = ids;
    h->shadow_map.assign((size_t)h->nstreams, -1);
    for (size_t i = 0; i < ids.size(); i++) h->shadow_map[(size_t)ids[i]] = (int)i;
    int64_t sb = 0;
    (void)jsdr_bpsk_slot_info(h, &sb, nullptr, nullptr, nullptr, nullptr);
    if (h->shadow_in.alloc(ids.size() * (size_t)(2 * h->max_batch)) != JSDR_OK || h->shadow_slots.alloc(ids.size() * (size_t)sb) != JSDR_OK) {
        (void)jsdr_bpsk_destroy(sh);
        h->shadow_ids.clear();
        h->shadow_map.clear();
        return JSDR_ERR;
    }
    h->shadow = sh;
    for (int k = 0; k < ncalls; k++) {
        if (shadow_stage(h, raw_dev_calls[k], stream_stride_i16, nsamples_calls[k], as_stream(stream)) != JSDR_OK ||
            bpsk_run(sh, h->shadow_in.p, nullptr, 2 * h->max_batch, nsamples_calls[k], ic, qc, as_stream(stream)) != JSDR_OK)
            return JSDR_ERR;
    }
    h->recovered_events++;
    if (recovered) *recovered = (int)ids.size();
    return JSDR_OK;
}

int jsdr_bpsk_set_cu_share(jsdr_bpsk *h, int wgs_per_cu)
{
    JSDR_REQUIRE(h && wgs_per_cu >= 0 && wgs_per_cu <= 16, "jsdr_bpsk_set_cu_share: bad argument");
    if (wgs_per_cu > 0) {
        int dev = 0, cus = 0;
        JSDR_HIP_TRY(hipGetDevice(&dev));
        JSDR_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        h->num_cu = cus > 0 ? cus : 256;
    }
    h->share_wgs_per_cu = wgs_per_cu;
    return JSDR_OK;
}

int jsdr_bpsk_pair_shares(jsdr_bpsk *h, int *fft_wgs_per_cu, int *bpsk_wgs_per_cu)
{
    JSDR_REQUIRE(h && fft_wgs_per_cu && bpsk_wgs_per_cu, "jsdr_bpsk_pair_shares: null argument");
    // measured (DESIGN.md "the step", profiles/r04_experiments.md, r05_j_bench.json): the 2 + 1 split pays for the exact variant's
    // tune-mode kernel at 96 kHz / 2048-sample frames from 8192 streams per device (33.5 against 35.1 ms); at 4096 / 2048 / 1024
    // streams it is slower than one after the other (19.8 / 10.2 / 5.9 against 17.9 / 9.5 / 5.1), and so it is with the fast variant
    const bool pays = !h->do_fft && h->rate == 96000 && h->nsf == 2048 && h->variant == 0 && h->nstreams >= 8192;
    *fft_wgs_per_cu = pays ? 2 : 0;
    *bpsk_wgs_per_cu = pays ? 1 : 0;
    return JSDR_OK;
}

int jsdr_bpsk_last_launch(jsdr_bpsk *h, int64_t *work_items, int64_t *workgroups)
{
    JSDR_REQUIRE(h && work_items && workgroups, "jsdr_bpsk_last_launch: null argument");
    *work_items = h->last_fm_items;
    *workgroups = h->last_fm_grid;
    return JSDR_OK;
}

int jsdr_bpsk_receive_i16(jsdr_bpsk *h, const int16_t *raw_host, int ic, int qc)
{
    JSDR_REQUIRE(h && raw_host, "jsdr_bpsk_receive_i16: null argument");
    JSDR_REQUIRE(h->nstreams == 1, "jsdr_bpsk_receive_i16: handle has %d streams; receive() is the 1-stream form",
                 h->nstreams);
    h->pin_call = true;
    h->pin_off = 0;
    int rc = JSDR_OK;
    const size_t fb = sizeof(int16_t) * 2 * (size_t)h->nsf;
    if (h->pin && fb <= h->pin_bytes) {  // the frame waits at the arena's head: bpsk_run sends it (with the tables, if new)
        memcpy(h->pin, raw_host, fb);
        h->pin_off = fb;
        h->rx_frame_bytes = fb;
    } else {
        rc = h2d_call(h, h->stage_raw.p, raw_host, fb, 0);
    }
    if (rc == JSDR_OK)
        rc = bpsk_run(h, reinterpret_cast<const int16_t *>(h->stage_raw.p), nullptr, 2LL * h->nsf, h->nsf, ic, qc, 0);
    h->rx_frame_bytes = 0;
    if (rc == JSDR_OK) rc = publish_snapshot(h);  // synchronises: the arena is free again
    if (rc != JSDR_OK) (void)hipDeviceSynchronize();  // (a failed call, wherever it failed: nothing may still be reading the arena)
    h->pin_call = false;
    return rc;
}

int jsdr_bpsk_receive_f32(jsdr_bpsk *h, const float *iq_host)
{
    JSDR_REQUIRE(h && iq_host, "jsdr_bpsk_receive_f32: null argument");
    JSDR_REQUIRE(h->nstreams == 1, "jsdr_bpsk_receive_f32: handle has %d streams; receive() is the 1-stream form",
                 h->nstreams);
    h->pin_call = true;
    h->pin_off = 0;
    // What IAudioHandler delivers are (float)s / 32767f values of DC-corrected shorts (JavaAudio.java:281-288).  When every
    // sample of the frame IS such a value -- checked here, on the host, with the same float division -- the frame goes
    // through the int16 kernels (k_fm: two launches instead of four, half the bytes up): same doubles, the two input
    // forms are bit-identical by construction (test_bpsk_one_stream_fed_through_both_input_forms_alternately).  Any other
    // float takes the float kernels as before.
    const size_t nfl = 2 * (size_t)h->nsf;
    static const bool route = [] {
        const char *e = knob("JSDR_F32_AS_I16");  // JSDR_F32_AS_I16=0: float frames always take the float kernels (tests)
        return !e || atoi(e) != 0;
    }();
    bool as_i16 = route && h->variant == 0 && h->pin && (h->n_in == 0 || !h->hist_is_float) && nfl * sizeof(int16_t) <= h->pin_bytes;
    if (as_i16) {
        int16_t *q = reinterpret_cast<int16_t *>(h->pin);  // the arena's head: the frame's slot
        unsigned bad = 0;
        for (size_t i = 0; i < nfl; i++) {
            const float f = iq_host[i];
            float v = f * 32767.0f;
            if (!(v == v)) {  // NaN: not the conversion of any short (and the casts below are undefined for it)
                bad = 1;
                v = 0.0f;
            }
            v = v > 32767.0f ? 32767.0f : (v < -32768.0f ? -32768.0f : v);
            const int sv = (int)__builtin_rintf(v);
            const float back = (float)sv / 32767.0f;
            unsigned bf, bb;  // compared as bit patterns: -0.0f is NOT the conversion of any short, and stays a float
            memcpy(&bf, &f, 4);
            memcpy(&bb, &back, 4);
            bad |= bf ^ bb;
            q[i] = (int16_t)sv;
        }
        as_i16 = bad == 0;
    }
    int rc;
    if (as_i16) {
        h->pin_off = nfl * sizeof(int16_t);
        h->rx_frame_bytes = nfl * sizeof(int16_t);  // sent by bpsk_run
        rc = bpsk_run(h, reinterpret_cast<const int16_t *>(h->stage_raw.p), nullptr, 2LL * h->nsf, h->nsf, 0, 0, 0);
        h->rx_frame_bytes = 0;
    } else {
        rc = h2d_call(h, h->stage_raw.p, iq_host, sizeof(float) * 2 * (size_t)h->nsf, 0);
        if (rc == JSDR_OK) rc = bpsk_run(h, nullptr, reinterpret_cast<const float *>(h->stage_raw.p), 2LL * h->nsf, h->nsf, 0, 0, 0);
    }
    if (rc == JSDR_OK) rc = publish_snapshot(h);
    if (rc != JSDR_OK) (void)hipDeviceSynchronize();
    h->pin_call = false;
    return rc;
}

// ---- the 1-stream receive() forms publish their results for a concurrent reader (jsdr_bpsk_snapshot_read).  What the
// four getters behind the snapshot fetch with eleven small blocking copies (each a round trip to the device: ~130 us of
// a 290 us receive) is packed by ONE tiny kernel at the end of the side stream's work and comes back in ONE copy.
__global__ void k_snapshot_pack(SnapPack *out, const TailState *st, const int *fec_last, const int *cnt_dec, const int *nbits,
                                const FftFrontState *fs, const unsigned char *decoded, const signed char *bits_new)
{
    const int i = threadIdx.x;
    if (i == 0) {
        out->t = st[0];
        out->last[0] = fec_last[0];
        out->last[1] = fec_last[1];
        out->cdec = cnt_dec[0];
        out->nbits = nbits[0];
        out->centreBin = fs ? fs[0].centreBin : 0;
        out->pad = 0;
        out->avePeakPower = fs ? fs[0].avePeakPower : 0.0;
        out->aveCentreBin = fs ? fs[0].aveCentreBin : 0.0;
    }
    if (i < 256) out->decoded[i] = decoded[i];
    const int nb = nbits[0];
    for (int k = i; k < 512; k += blockDim.x) out->bits[k] = k < nb ? bits_new[k] : (signed char)0;
}

static void counters_from(const jsdr_bpsk *h, const TailState &t, const int last[2], int cdec, int centreBin, int32_t *out)
{
    out[0] = (int32_t)h->n_in;
    out[1] = (int32_t)h->n_ds;
    out[2] = t.cntBit;
    out[3] = t.cntFEC;
    out[4] = cdec;
    out[5] = last[0];
    out[6] = t.dmCorr;
    out[7] = t.dmMaxCorr;
    out[8] = last[1];
    out[9] = h->do_fft ? centreBin : 0;
}

static void state_from(const jsdr_bpsk *h, const TailState &t, double avePeakPower, double aveCentreBin, double *out)
{
    out[0] = h->tuPhase;
    out[1] = h->vcoPhase;
    // dmBitPhase (:501,:581-584): k steps of +1/9600 from 0.0 within the current bit, replayed exactly
    double ph = 0.0;
    for (int i = 0; i < (int)(h->n_ds & 7); i++) ph += 1.0 / (double)9600;
    out[2] = ph;
    out[3] = t.dmEnergyOut;
    out[4] = t.energy1;
    out[5] = t.energy2;
    out[6] = h->do_fft ? avePeakPower : 0.0;
    out[7] = h->do_fft ? aveCentreBin : 0.0;
    for (int i = 0; i < 8; i++) out[8 + i] = t.dmEnergy[i];
    out[16] = t.lastI;
    out[17] = t.lastQ;
}

static int publish_snapshot(jsdr_bpsk *h)
{
    // pack on the stream the call's last kernels ran on, fetch once
    hipStream_t ts = (h->overlap && h->tail_stream) ? h->tail_stream : h->last_stream;
    if (!h->snap_fused) {
        hipLaunchKernelGGL(k_snapshot_pack, dim3(1), dim3(256), 0, ts, h->snap_dev.p, h->tail.p, h->fec_last.p, h->cnt_dec.p,
                           h->nbits.p, h->do_fft ? h->fft_state.p : (const FftFrontState *)nullptr, h->decoded.p,
                           h->bitlog[h->bitlog_cur].p + HIST_BITS);
        JSDR_LAUNCH_CHECK();
    }
    SnapPack pk_stack;
    // (the arena's last slot: reserved at create, never handed out by h2d_call)
    SnapPack *pkp = h->pin ? reinterpret_cast<SnapPack *>(h->pin + h->pin_bytes) : &pk_stack;
    JSDR_HIP_TRY(hipMemcpyAsync(pkp, h->snap_dev.p, sizeof(SnapPack), hipMemcpyDeviceToHost, ts));
    if (sync_last(h) != JSDR_OK) return JSDR_ERR;
    JSDR_HIP_TRY(hipStreamSynchronize(ts));
    if (h->snap_fused) {  // what k_snapshot_pack does on the device: no FFT state in tune mode, no bytes beyond the call's bits
        if (!h->do_fft) {
            pkp->centreBin = 0;
            pkp->avePeakPower = 0.0;
            pkp->aveCentreBin = 0.0;
        }
        for (int k = pkp->nbits < 0 ? 0 : pkp->nbits; k < 512; k++) pkp->bits[k] = 0;
    }
    const SnapPack &pk = *pkp;
    // what the getters refuse, the snapshot refuses (check_overflow)
    JSDR_REQUIRE(!pk.t.overflow, "receive: stream 0 exceeded its per-call capacity (%d bits / %d FEC calls per call of at most %lld samples)",
                 h->max_bits, h->trig_cap, h->max_batch);
    JSDR_REQUIRE(h->variant == 0 || !pk.t.uncertified, "receive: stream 0: the fast variant could not certify a slicer decision (inside "
                 "its error margin and not recomputable in exact order); run this stream with JSDR_VARIANT_EXACT");
    const int cur = h->snap_cur.load(std::memory_order_relaxed);
    const int w = cur == 0 ? 1 : 0;  // the copy no reader is directed to
    h->snap_seq[w].fetch_add(1, std::memory_order_acq_rel);  // odd: writing
    jsdr_bpsk_snapshot &sn = h->snap[w];
    counters_from(h, pk.t, pk.last, pk.cdec, pk.centreBin, sn.counters);
    state_from(h, pk.t, pk.avePeakPower, pk.aveCentreBin, sn.state);
    memcpy(sn.decoded, pk.decoded, 256);
    static_assert(sizeof(sn.bits) == sizeof(pk.bits), "snapshot bit capacity");
    memcpy(sn.bits, pk.bits, sizeof(sn.bits));
    sn.nbits = pk.nbits;
    sn.frames = ++h->snap_count;
    h->snap_seq[w].fetch_add(1, std::memory_order_release);  // even: complete
    h->snap_cur.store(w, std::memory_order_release);
    return JSDR_OK;
}

static int sync_last(jsdr_bpsk *h)
{
    JSDR_HIP_TRY(hipStreamSynchronize(h->last_stream));
    if (h->tail_stream) JSDR_HIP_TRY(hipStreamSynchronize(h->tail_stream));
    if (h->shadow) return sync_last(h->shadow);
    return JSDR_OK;
}

// a stream jsdr_bpsk_recover_uncertified() has replayed is served by the exact shadow handle
#define JSDR_SHADOWED(h, stream) ((h)->shadow && (stream) >= 0 && (stream) < (h)->nstreams && (h)->shadow_map[(size_t)(stream)] >= 0)

// a stream that overflowed its per-call capacity has an incomplete result log: every getter says so
static int check_overflow(jsdr_bpsk *h, int stream, const char *who)
{
    int ov = 0;
    JSDR_HIP_TRY(hipMemcpy(&ov, reinterpret_cast<const char *>(h->tail.p + stream) + offsetof(TailState, overflow), sizeof(int),
                           hipMemcpyDeviceToHost));
    JSDR_REQUIRE(!ov, "%s: stream %d exceeded its per-call capacity (%d bits / %d FEC calls per call of at most %lld samples)", who,
                 stream, h->max_bits, h->trig_cap, h->max_batch);
    if (h->variant != 0) {
        int un = 0;
        JSDR_HIP_TRY(hipMemcpy(&un, reinterpret_cast<const char *>(h->tail.p + stream) + offsetof(TailState, uncertified), sizeof(int),
                               hipMemcpyDeviceToHost));
        JSDR_REQUIRE(!un, "%s: stream %d: the fast variant could not certify a slicer decision (inside its error margin and not "
                     "recomputable in exact order); run this stream with JSDR_VARIANT_EXACT", who, stream);
    }
    return JSDR_OK;
}

int jsdr_bpsk_get_counters(jsdr_bpsk *h, int stream, int32_t out[JSDR_BPSK_NCOUNTERS])
{
    JSDR_REQUIRE(h && out, "jsdr_bpsk_get_counters: null argument");
    JSDR_REQUIRE(stream >= 0 && stream < h->nstreams, "jsdr_bpsk_get_counters: stream %d out of range", stream);
    if (JSDR_SHADOWED(h, stream)) return jsdr_bpsk_get_counters(h->shadow, h->shadow_map[(size_t)stream], out);
    if (sync_last(h) != JSDR_OK) return JSDR_ERR;
    TailState t;
    int last[2], cdec;
    JSDR_HIP_TRY(hipMemcpy(&t, h->tail.p + stream, sizeof(t), hipMemcpyDeviceToHost));
    JSDR_HIP_TRY(hipMemcpy(last, h->fec_last.p + 2 * stream, sizeof(last), hipMemcpyDeviceToHost));
    JSDR_HIP_TRY(hipMemcpy(&cdec, h->cnt_dec.p + stream, sizeof(int), hipMemcpyDeviceToHost));
    if (check_overflow(h, stream, "jsdr_bpsk_get_counters") != JSDR_OK) return JSDR_ERR;
    int centreBin = 0;
    if (h->do_fft) {
        FftFrontState fs;
        JSDR_HIP_TRY(hipMemcpy(&fs, h->fft_state.p + stream, sizeof(fs), hipMemcpyDeviceToHost));
        centreBin = fs.centreBin;
    }
    counters_from(h, t, last, cdec, centreBin, out);
    return JSDR_OK;
}

int jsdr_bpsk_get_bits(jsdr_bpsk *h, int stream, int8_t *bits_host, int cap, int *nbits)
{
    JSDR_REQUIRE(h && nbits, "jsdr_bpsk_get_bits: null argument");
    JSDR_REQUIRE(stream >= 0 && stream < h->nstreams, "jsdr_bpsk_get_bits: stream %d out of range", stream);
    if (JSDR_SHADOWED(h, stream)) return jsdr_bpsk_get_bits(h->shadow, h->shadow_map[(size_t)stream], bits_host, cap, nbits);
    if (sync_last(h) != JSDR_OK || check_overflow(h, stream, "jsdr_bpsk_get_bits") != JSDR_OK) return JSDR_ERR;
    int nb = 0;
    JSDR_HIP_TRY(hipMemcpy(&nb, h->nbits.p + stream, sizeof(int), hipMemcpyDeviceToHost));
    *nbits = nb;
    int n = nb < cap ? nb : cap;
    if (n > 0 && bits_host)
        JSDR_HIP_TRY(hipMemcpy(bits_host, h->bitlog[h->bitlog_cur].p + (size_t)stream * h->bitlog_stride + HIST_BITS,
                               (size_t)n, hipMemcpyDeviceToHost));
    return JSDR_OK;
}

int jsdr_bpsk_get_fec_count(jsdr_bpsk *h, int stream, int *count)
{
    JSDR_REQUIRE(h && count, "jsdr_bpsk_get_fec_count: null argument");
    JSDR_REQUIRE(stream >= 0 && stream < h->nstreams, "jsdr_bpsk_get_fec_count: stream %d out of range", stream);
    if (JSDR_SHADOWED(h, stream)) return jsdr_bpsk_get_fec_count(h->shadow, h->shadow_map[(size_t)stream], count);
    if (sync_last(h) != JSDR_OK || check_overflow(h, stream, "jsdr_bpsk_get_fec_count") != JSDR_OK) return JSDR_ERR;
    JSDR_HIP_TRY(hipMemcpy(count, h->trig_count.p + stream, sizeof(int), hipMemcpyDeviceToHost));
    return JSDR_OK;
}

int jsdr_bpsk_get_fec(jsdr_bpsk *h, int stream, int idx, int32_t *rc, int32_t *bit_index, uint8_t out_host[256])
{
    JSDR_REQUIRE(h && rc && bit_index && out_host, "jsdr_bpsk_get_fec: null argument");
    JSDR_REQUIRE(stream >= 0 && stream < h->nstreams, "jsdr_bpsk_get_fec: stream %d out of range", stream);
    if (JSDR_SHADOWED(h, stream)) return jsdr_bpsk_get_fec(h->shadow, h->shadow_map[(size_t)stream], idx, rc, bit_index, out_host);
    int cnt = 0;
    if (jsdr_bpsk_get_fec_count(h, stream, &cnt) != JSDR_OK) return JSDR_ERR;
    JSDR_REQUIRE(idx >= 0 && idx < cnt, "jsdr_bpsk_get_fec: index %d outside the %d calls of the last batch", idx, cnt);
    int b = 0;
    JSDR_HIP_TRY(hipMemcpy(rc, h->fec_rc.p + stream * h->trig_cap + idx, sizeof(int), hipMemcpyDeviceToHost));
    JSDR_HIP_TRY(hipMemcpy(&b, h->trig_bits.p + stream * h->trig_cap + idx, sizeof(int), hipMemcpyDeviceToHost));
    *bit_index = b + 1;
    JSDR_HIP_TRY(hipMemcpy(out_host, h->fec_data.p + ((size_t)stream * h->trig_cap + idx) * 256, 256, hipMemcpyDeviceToHost));
    return JSDR_OK;
}

int jsdr_bpsk_get_decoded(jsdr_bpsk *h, int stream, uint8_t out_host[256])
{
    JSDR_REQUIRE(h && out_host, "jsdr_bpsk_get_decoded: null argument");
    JSDR_REQUIRE(stream >= 0 && stream < h->nstreams, "jsdr_bpsk_get_decoded: stream %d out of range", stream);
    if (JSDR_SHADOWED(h, stream)) return jsdr_bpsk_get_decoded(h->shadow, h->shadow_map[(size_t)stream], out_host);
    if (sync_last(h) != JSDR_OK || check_overflow(h, stream, "jsdr_bpsk_get_decoded") != JSDR_OK) return JSDR_ERR;
    JSDR_HIP_TRY(hipMemcpy(out_host, h->decoded.p + (size_t)stream * 256, 256, hipMemcpyDeviceToHost));
    return JSDR_OK;
}

int jsdr_bpsk_get_trace(jsdr_bpsk *h, int stream, double *out_host, int64_t cap_pairs, int64_t *npairs)
{
    JSDR_REQUIRE(h && npairs, "jsdr_bpsk_get_trace: null argument");
    JSDR_REQUIRE(stream >= 0 && stream < h->nstreams, "jsdr_bpsk_get_trace: stream %d out of range", stream);
    if (JSDR_SHADOWED(h, stream)) return jsdr_bpsk_get_trace(h->shadow, h->shadow_map[(size_t)stream], out_host, cap_pairs, npairs);
    if (sync_last(h) != JSDR_OK) return JSDR_ERR;
    *npairs = h->last_nds;
    long long n = h->last_nds < cap_pairs ? h->last_nds : cap_pairs;
    if (n > 0 && out_host)
        JSDR_HIP_TRY(hipMemcpy(out_host, h->y[h->last_y].p + Y_PAD + (size_t)stream * h->y_stride, sizeof(double2) * (size_t)n,
                               hipMemcpyDeviceToHost));
    return JSDR_OK;
}

int jsdr_bpsk_get_state(jsdr_bpsk *h, int stream, double out[18])
{
    JSDR_REQUIRE(h && out, "jsdr_bpsk_get_state: null argument");
    JSDR_REQUIRE(stream >= 0 && stream < h->nstreams, "jsdr_bpsk_get_state: stream %d out of range", stream);
    if (JSDR_SHADOWED(h, stream)) return jsdr_bpsk_get_state(h->shadow, h->shadow_map[(size_t)stream], out);
    if (sync_last(h) != JSDR_OK) return JSDR_ERR;
    TailState t;
    JSDR_HIP_TRY(hipMemcpy(&t, h->tail.p + stream, sizeof(t), hipMemcpyDeviceToHost));
    double app = 0.0, acb = 0.0;
    if (h->do_fft) {
        FftFrontState fs;
        JSDR_HIP_TRY(hipMemcpy(&fs, h->fft_state.p + stream, sizeof(fs), hipMemcpyDeviceToHost));
        app = fs.avePeakPower;
        acb = fs.aveCentreBin;
    }
    state_from(h, t, app, acb, out);
    return JSDR_OK;
}

int jsdr_bpsk_sync(jsdr_bpsk *h)
{
    JSDR_REQUIRE(h, "jsdr_bpsk_sync: null handle");
    return sync_last(h);
}

// any thread, no HIP call, no lock: the results of the last completed receive_*() of a 1-stream handle
int jsdr_bpsk_snapshot_read(jsdr_bpsk *h, jsdr_bpsk_snapshot *out)
{
    JSDR_REQUIRE(h && out, "jsdr_bpsk_snapshot_read: null argument");
    for (int attempt = 0; attempt < 1000; attempt++) {
        const int cur = h->snap_cur.load(std::memory_order_acquire);
        JSDR_REQUIRE(cur >= 0, "jsdr_bpsk_snapshot_read: nothing received yet");
        const unsigned s0 = h->snap_seq[cur].load(std::memory_order_acquire);
        if (s0 & 1u) continue;  // being rewritten: the writer has lapped us, take the newer copy
        memcpy(out, &h->snap[cur], sizeof(*out));
        std::atomic_thread_fence(std::memory_order_acquire);
        if (h->snap_seq[cur].load(std::memory_order_relaxed) == s0) return JSDR_OK;
    }
    set_error("jsdr_bpsk_snapshot_read: could not get a stable copy");
    return JSDR_ERR;
}

int jsdr_bpsk_profile_enable(jsdr_bpsk *h, int on)
{
    JSDR_REQUIRE(h, "jsdr_bpsk_profile_enable: null handle");
    h->prof_on = on != 0;
    return JSDR_OK;
}

// the demodulator's constant tables as this library holds them (host side, no device needed): 0 = dsFilter[27]
// (FUNcubeBPSKDemod.java:27-55), 1 = dmFilter[65] (:58-77, one of the two identical copies), 2 = SYNC_VECTOR[65] (:79-81)
int jsdr_bpsk_table(int which, double *out, int cap)
{
    JSDR_REQUIRE(out, "jsdr_bpsk_table: null argument");
    const int n = which == 0 ? 27 : ((which == 1 || which == 2) ? 65 : 0);
    JSDR_REQUIRE(n > 0 && cap >= n, "jsdr_bpsk_table: table %d needs room for %d values", which, n);
    if (which == 0) {
        for (int i = 0; i < 14; i++) out[i] = out[26 - i] = (double)h_ds_half[i];
    } else if (which == 1) {
        for (int i = 0; i < 33; i++) out[i] = out[64 - i] = (double)h_dm_half[i];
    } else {
        int sr = 0x7f;  // the sync LFSR (FECDecoder.java:600-605) == SYNC_VECTOR
        for (int i = 0; i < 65; i++) {
            out[i] = (sr & 64) ? 1.0 : -1.0;
            int v = sr & 0x48;
            v ^= v >> 4;
            v ^= v >> 2;
            v ^= v >> 1;
            sr = ((sr << 1) | (v & 1)) & 0xffff;
        }
    }
    return JSDR_OK;
}

int jsdr_bpsk_profile_count(void) { return PK_COUNT; }

const char *jsdr_bpsk_front_kernel(jsdr_bpsk *h) { return h ? h->front_name : ""; }

const char *jsdr_bpsk_tail_kernel(jsdr_bpsk *h) { return h ? h->tail_name : ""; }
const char *jsdr_bpsk_fec_kernel(jsdr_bpsk *h) { return h ? h->fec_name : ""; }

int jsdr_bpsk_side_stream(jsdr_bpsk *h, int *on)
{
    JSDR_REQUIRE(h && on, "jsdr_bpsk_side_stream: null argument");
    *on = h->overlap ? 1 : 0;
    return JSDR_OK;
}

// fast variant: decisions redone in exact order (all streams, since creation), streams that ended up uncertified, the
// error bound of (fi,fq) the margins are built on
int jsdr_bpsk_cert_stats(jsdr_bpsk *h, int64_t *redone, int64_t *uncertified_streams, double *ey)
{
    JSDR_REQUIRE(h, "jsdr_bpsk_cert_stats: null handle");
    if (sync_last(h) != JSDR_OK) return JSDR_ERR;
    std::vector<TailState> ts((size_t)h->nstreams);
    JSDR_HIP_TRY(hipMemcpy(ts.data(), h->tail.p, sizeof(TailState) * ts.size(), hipMemcpyDeviceToHost));
    long long r = 0, u = 0;
    for (size_t s = 0; s < ts.size(); s++) {
        r += ts[s].redone;
        u += (ts[s].uncertified && !JSDR_SHADOWED(h, (int)s)) ? 1 : 0;  // (a recovered stream is served in exact order: not counted)
    }
    if (redone) *redone = r;
    if (uncertified_streams) *uncertified_streams = u;
    if (ey) *ey = h->fast_ey * h->margin_scale;
    return JSDR_OK;
}

// fast variant: WHICH streams are uncertified (ascending ids, at most cap of them; *count = how many there are in all).
// The flag is set by the call in which the uncertifiable decision fell and stays set: the results of that call and of
// every later one are withheld for that stream (its getters fail).  What a caller does with the list: run those streams
// -- from the first sample of the flagged call on an exact handle that has seen the same earlier input, or from the start
// of the stream -- with JSDR_VARIANT_EXACT, which decides the same near-tie by the reference's own arithmetic.
int jsdr_bpsk_uncertified_streams(jsdr_bpsk *h, int32_t *ids, int cap, int *count)
{
    JSDR_REQUIRE(h && count, "jsdr_bpsk_uncertified_streams: null argument");
    JSDR_REQUIRE(cap == 0 || ids, "jsdr_bpsk_uncertified_streams: null id buffer");
    *count = 0;
    if (h->variant == 0) return JSDR_OK;  // the exact variant certifies nothing and withholds nothing
    if (sync_last(h) != JSDR_OK) return JSDR_ERR;
    std::vector<TailState> ts((size_t)h->nstreams);
    JSDR_HIP_TRY(hipMemcpy(ts.data(), h->tail.p, sizeof(TailState) * ts.size(), hipMemcpyDeviceToHost));
    int n = 0;
    for (int s = 0; s < h->nstreams; s++)
        if (ts[(size_t)s].uncertified && !JSDR_SHADOWED(h, s)) {
            if (n < cap) ids[n] = s;
            n++;
        }
    *count = n;
    return JSDR_OK;
}

int jsdr_bpsk_stream_recovered(jsdr_bpsk *h, int stream, int *recovered)
{
    JSDR_REQUIRE(h && recovered && stream >= 0 && stream < h->nstreams, "jsdr_bpsk_stream_recovered: bad argument");
    *recovered = JSDR_SHADOWED(h, stream) ? 1 : 0;
    return JSDR_OK;
}

int jsdr_bpsk_schedule_stats(jsdr_bpsk *h, int64_t *computed_inline, int64_t *prefetched)
{
    JSDR_REQUIRE(h, "jsdr_bpsk_schedule_stats: null handle");
    if (computed_inline) *computed_inline = h->sched_sync;
    if (prefetched) *prefetched = h->sched_prefetched;
    return JSDR_OK;
}

int jsdr_bpsk_set_variant(jsdr_bpsk *h, int variant)
{
    JSDR_REQUIRE(h, "jsdr_bpsk_set_variant: null handle");
    JSDR_REQUIRE(variant == JSDR_VARIANT_EXACT || variant == JSDR_VARIANT_FAST, "jsdr_bpsk_set_variant: unknown variant %d", variant);
    JSDR_REQUIRE(h->n_in == 0, "jsdr_bpsk_set_variant: the variant is fixed once samples have been received");
    JSDR_REQUIRE(variant == JSDR_VARIANT_EXACT || !h->do_fft, "jsdr_bpsk_set_variant: the fast variant covers the tune mode only");
    h->variant = variant;
    return JSDR_OK;
}

const char *jsdr_bpsk_profile_name(int k) { return (k >= 0 && k < PK_COUNT) ? kProfNames[k] : ""; }

int jsdr_bpsk_profile_read(jsdr_bpsk *h, double *ms_total, int *launches)
{
    JSDR_REQUIRE(h && ms_total && launches, "jsdr_bpsk_profile_read: null argument");
    for (int k = 0; k < PK_COUNT; k++) {
        ms_total[k] = 0.0;
        launches[k] = 0;
    }
    for (auto &r : h->prof_recs) {
        float ms = 0.f;
        JSDR_HIP_TRY(hipEventSynchronize(r.b));
        JSDR_HIP_TRY(hipEventElapsedTime(&ms, r.a, r.b));
        ms_total[r.kernel] += (double)ms;
        launches[r.kernel] += 1;
        h->prof_pool.push_back(r.a);
        h->prof_pool.push_back(r.b);
    }
    h->prof_recs.clear();
    return JSDR_OK;
}

int jsdr_bpsk_slot_info(jsdr_bpsk *h, int64_t *slot_bytes, int64_t *bits_offset, int64_t *fec_offset, int *slot_bits,
                        int *nfec_max)
{
    JSDR_REQUIRE(h, "jsdr_bpsk_slot_info: null handle");
    int64_t bits = (h->max_bits + 15) & ~15;
    if (bits_offset) *bits_offset = 64;
    if (fec_offset) *fec_offset = 64 + bits;
    if (slot_bits) *slot_bits = (int)bits;
    if (nfec_max) *nfec_max = h->trig_cap;
    if (slot_bytes) *slot_bytes = 64 + bits + (int64_t)h->trig_cap * 264;
    return JSDR_OK;
}

}  // extern "C"

namespace jsdr {
// slot = int32 header[16] | int8 bits[slot_bits] | trig_cap x {int32 rc, int32 bit_index, uint8 data[256]}
// header: nbits, nfec, cntRaw, cntDS, cntBit, cntFEC, cntDec, dmErrBits, dmCorr, dmMaxCorr, decodeOK, overflow (a stream
// that overflowed its per-call bit / FEC capacity: its slot is incomplete), uncertified (fast variant: a decision of this
// stream could not be certified), 0...
__global__ void k_pack_slots(unsigned char *slots, long long slot_bytes, int slot_bits, int trig_cap, const TailState *st,
                             const int *nbits, const signed char *bitlog, long long bitlog_stride, const int *trig_count,
                             const int *trig_bits, const int *fec_rc, const unsigned char *fec_data, const int *fec_last,
                             const int *cnt_dec, int n_in, int n_ds)
{
    const int s = blockIdx.x;
    unsigned char *slot = slots + (long long)s * slot_bytes;
    int *hdr = reinterpret_cast<int *>(slot);
    const int nb = nbits[s], nt = trig_count[s];
    if (threadIdx.x < 16) {
        int v = 0;
        switch (threadIdx.x) {
            case 0: v = nb; break;
            case 1: v = nt; break;
            case 2: v = n_in; break;
            case 3: v = n_ds; break;
            case 4: v = st[s].cntBit; break;
            case 5: v = st[s].cntFEC; break;
            case 6: v = cnt_dec[s]; break;
            case 7: v = fec_last[2 * s]; break;
            case 8: v = st[s].dmCorr; break;
            case 9: v = st[s].dmMaxCorr; break;
            case 10: v = fec_last[2 * s + 1]; break;
            case 11: v = st[s].overflow; break;
            case 12: v = st[s].uncertified; break;
            default: v = 0;
        }
        hdr[threadIdx.x] = v;
    }
    const signed char *bl = bitlog + (long long)s * bitlog_stride + HIST_BITS;
    for (int i = threadIdx.x; i < slot_bits; i += blockDim.x) slot[64 + i] = (i < nb) ? (unsigned char)bl[i] : 0;
    unsigned char *f = slot + 64 + slot_bits;
    for (int t = 0; t < trig_cap; t++) {
        int *fh = reinterpret_cast<int *>(f + t * 264);
        if (threadIdx.x == 0) {
            fh[0] = (t < nt) ? fec_rc[s * trig_cap + t] : 0;
            fh[1] = (t < nt) ? trig_bits[s * trig_cap + t] + 1 : 0;
        }
        for (int i = threadIdx.x; i < 256; i += blockDim.x)
            f[t * 264 + 8 + i] = (t < nt) ? fec_data[((long long)s * trig_cap + t) * 256 + i] : 0;
    }
}
}  // namespace jsdr

extern "C" int jsdr_bpsk_pack_slots(jsdr_bpsk *h, uint8_t *slots_dev, void *stream)
{
    JSDR_REQUIRE(h && slots_dev, "jsdr_bpsk_pack_slots: null argument");
    int64_t slot_bytes = 0;
    int slot_bits = 0;
    jsdr_bpsk_slot_info(h, &slot_bytes, nullptr, nullptr, &slot_bits, nullptr);
    if (h->overlap && h->tail_pending[h->last_y]) {  // results of the last call come from the side stream
        JSDR_HIP_TRY(hipStreamWaitEvent(as_stream(stream), h->ev_tail_done[h->last_y], 0));
    } else if (!h->overlap && as_stream(stream) != h->last_stream && h->ev_matched) {
        // no side stream (1-stream handles, JSDR_NO_OVERLAP): the results come from the stream of the last call; a pack on
        // ANOTHER stream waits for that one (the event is free in this mode)
        JSDR_HIP_TRY(hipEventRecord(h->ev_matched, h->last_stream));
        JSDR_HIP_TRY(hipStreamWaitEvent(as_stream(stream), h->ev_matched, 0));
    }
    hipLaunchKernelGGL(k_pack_slots, dim3((unsigned)h->nstreams), dim3(256), 0, as_stream(stream), slots_dev,
                       (long long)slot_bytes, slot_bits, h->trig_cap, h->tail.p, h->nbits.p, h->bitlog[h->bitlog_cur].p,
                       h->bitlog_stride, h->trig_count.p, h->trig_bits.p, h->fec_rc.p, h->fec_data.p, h->fec_last.p,
                       h->cnt_dec.p, (int)h->n_in, (int)h->n_ds);
    JSDR_LAUNCH_CHECK();
    // the per-call result arrays are single-buffered: the next call's tail / sync / FEC (side stream) must not
    // overwrite them while this kernel is still reading
    JSDR_HIP_TRY(hipEventRecord(h->ev_pack_done, as_stream(stream)));
    h->pack_pending = true;
    if (h->shadow) {
        // the recovered streams' slots are the exact shadow's (same slot layout: same max_batch), laid over the fast handle's
        if (jsdr_bpsk_pack_slots(h->shadow, h->shadow_slots.p, stream) != JSDR_OK) return JSDR_ERR;
        for (size_t i = 0; i < h->shadow_ids.size(); i++)
            JSDR_HIP_TRY(hipMemcpyAsync(slots_dev + (size_t)h->shadow_ids[i] * (size_t)slot_bytes, h->shadow_slots.p + i * (size_t)slot_bytes,
                                        (size_t)slot_bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    }
    return JSDR_OK;
}
