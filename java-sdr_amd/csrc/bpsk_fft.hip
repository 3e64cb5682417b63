// bpsk_fft.hip -- FUNcubeBPSKDemod FFT-acquire front end, doBufferFFT (FUNcubeBPSKDemod.java:406-464):
//   forward FFT of the frame (JTransforms DoubleFFT_1D.complexForward, :422-423)
//   |X| over the lower half (:425-427), 100-wide boxcar + first-maximum search over a quarter band (:433-443)
//   peak-power IIR, threshold, centre-bin update and clamp (:444-453)
//   204 bins around the centre moved to bin 0 (:458), scaled inverse FFT (:459)
//   RxDownSample(re, re) for every sample (:461-463): 27-tap low-pass at the decimated instants (:470-492)
// and then the same VCO mix as the tune-mode front end, so that k_matched / k_tail / k_sync / FEC run unchanged.
//
// Compiled with -ffp-contract=off.  The FFT is the SAME radix-2 decimation-in-time network as the oracle's
// jo_fft_f64 on the SAME twiddle table (uploaded by the host), butterfly for butterfly:
//     t = w*b (tr = wr*br - wi*bi, ti = wr*bi + wi*br), a' = a + t, b' = a - t
// so spectra, centre bins and everything downstream are bit-identical to the oracle.  JTransforms' own
// rounding is unknowable (source absent): parity with the Java library itself is unpinned.
//
// MI355X mapping: one 256-thread workgroup per stream, persistent over the frames of the call (the centre-bin
// state is sequential from frame to frame); four workgroups share a CU at N=2048 (38.9 KB of LDS and <=128 VGPRs
// each), i.e. the 1024 streams of the benchmark are resident at once.  The kernel is LDS-bandwidth bound, so the
// structure is about LDS bytes:
//   * the frame is a padded double2 image in LDS (slot e + e/8: conflict-free for all pass shapes);
//   * THREE radix-2 stages per LDS round trip with the 8 points of a group in registers (4 passes, not 11);
//   * the first pass of the forward FFT takes its inputs straight from the global-load registers (prefetched
//     one frame ahead), the first pass of the inverse FFT straight from the spectrum (204 bins + zeros): no
//     bit-reversal scatter, no zero fill;
//   * the last inverse pass scales and stores only the real parts (all RxDownSample reads);
//   * twiddles: stages with wing <= 32 from a 1 KB LDS table, the first pass from scalar loads (uniform),
//     the wide stages from the per-stage contiguous global table through L1 (coalesced 16 B per lane);
//   * |X| and the boxcar sums reuse the dead upper part of the image.
#include "bpsk_fft.h"
#include <math.h>

namespace jsdr {

// LDS image of the frame: element e lives at slot e + (e >> 3).
__device__ __forceinline__ int xpad(int e) { return e + (e >> 3); }

template <int BITS>
__device__ __forceinline__ int brevn(int x)
{
    return (int)(__brev((unsigned)x) >> (32 - BITS));
}

constexpr int kLdsTwiddleWing = 32;  // stages with half <= 32 read LDS (63 entries)

// twiddle jj of the stage with wing distance HALF: Ts[HALF-1+jj] = W_n^(jj * n/(2*HALF))
template <int HALF, bool UNIFORM>
__device__ __forceinline__ double2 tw_get(const double2 *TsL, const double2 *__restrict__ tsg, int jj)
{
    if (!UNIFORM && HALF <= kLdsTwiddleWing) return TsL[HALF - 1 + jj];
    return tsg[(unsigned)(HALF - 1 + jj)];  // 32-bit offset on the scalar base
}

// G consecutive stages of the radix-2 DIT network on the 2^G values v[m] = x[base + j + HALF0*m].  Every
// butterfly is the oracle's:  t = w*b (tr = wr*br - wi*bi, ti = wr*bi + wi*br), a' = a + t, b' = a - t;
// only the grouping differs from jo_fft_f64, not a single operation.
template <int G, int HALF0, bool INVERSE, bool UNIFORM>
__device__ __forceinline__ void dit_stages(double2 (&v)[1 << G], int j, const double2 *TsL, const double2 *__restrict__ tsg)
{
    constexpr int M = 1 << G;
#pragma unroll
    for (int t = 0; t < G; t++) {
        double2 w[1 << (G - 1)];
#pragma unroll
        for (int u = 0; u < (1 << t); u++) {
            // stage wing HALF0 << t
            if (t == 0) w[u] = tw_get<HALF0, UNIFORM>(TsL, tsg, j + HALF0 * u);
            if (t == 1) w[u] = tw_get<HALF0 * 2, UNIFORM>(TsL, tsg, j + HALF0 * u);
            if (t == 2) w[u] = tw_get<HALF0 * 4, UNIFORM>(TsL, tsg, j + HALF0 * u);
        }
#pragma unroll
        for (int m = 0; m < M; m++) {
            if ((m >> t) & 1) continue;  // m is the upper wing's index
            const double2 wv = w[m & ((1 << t) - 1)];
            const double wr = wv.x;
            const double wi = INVERSE ? -wv.y : wv.y;
            const double2 bq = v[m + (1 << t)];
            const double p1 = wr * bq.x, p2 = wi * bq.y, p3 = wr * bq.y, p4 = wi * bq.x;
            const double tr = p1 - p2;
            const double ti = p3 + p4;
            const double2 aq = v[m];
            v[m] = make_double2(aq.x + tr, aq.y + ti);
            v[m + (1 << t)] = make_double2(aq.x - tr, aq.y - ti);
        }
    }
}

// The first three stages (wings 1, 2, 4) of the FORWARD transform on a group of eight converted int16 samples, with the
// multiplications by the trivial twiddles 1 = (1, -0) and -i = (0, -1) of the table not performed:
//     w = 1 :  t = b                     w = -i :  t = (b.y, -b.x)
// instead of tr = wr*br - wi*bi, ti = wr*bi + wi*br.  Same results to the last bit, signs of zeros included, because no
// value in this network is ever -0.0: a converted int16 sample is never -0.0 ((float)0/32767f = +0.0f), and a sum or a
// difference is -0.0 only if an operand already is.  With that, for w = 1 the products wi*bi = -0*bi and wi*br are zeros
// that leave br and bi unchanged; for w = -i, tr = (+-0) + bi = bi exactly (bi = +0 gives +0 either way), and ti = (+-0) - br
// differs from -br at most in the sign of a zero -- which a + t and a - t (a never -0.0) absorb.  Twelve of the twenty
// butterflies of a group are trivial: 60 of its 200 operations.  (Float input may hold -0.0f and takes dit_stages.)
__device__ __forceinline__ void dit_first3_i16(double2 (&v)[8], const double2 *__restrict__ tsg)
{
    auto bf1 = [](double2 &a, double2 &b) {  // w = 1
        const double2 x = a, y = b;
        a = make_double2(x.x + y.x, x.y + y.y);
        b = make_double2(x.x - y.x, x.y - y.y);
    };
    auto bfi = [](double2 &a, double2 &b) {  // w = -i: t = (b.y, -b.x)
        const double2 x = a, y = b;
        a = make_double2(x.x + y.y, x.y - y.x);
        b = make_double2(x.x - y.y, x.y + y.x);
    };
    auto bfw = [](double2 &a, double2 &b, const double2 w) {  // the oracle's butterfly
        const double2 x = a, y = b;
        const double p1 = w.x * y.x, p2 = w.y * y.y, p3 = w.x * y.y, p4 = w.y * y.x;
        const double tr = p1 - p2, ti = p3 + p4;
        a = make_double2(x.x + tr, x.y + ti);
        b = make_double2(x.x - tr, x.y - ti);
    };
    // wing 1: twiddle Ts[0] = 1
    bf1(v[0], v[1]);
    bf1(v[2], v[3]);
    bf1(v[4], v[5]);
    bf1(v[6], v[7]);
    // wing 2: Ts[1] = 1, Ts[2] = -i
    bf1(v[0], v[2]);
    bfi(v[1], v[3]);
    bf1(v[4], v[6]);
    bfi(v[5], v[7]);
    // wing 4: Ts[3] = 1, Ts[4] = W8, Ts[5] = -i, Ts[6] = W8^3
    bf1(v[0], v[4]);
    bfw(v[1], v[5], tsg[4]);
    bfi(v[2], v[6]);
    bfw(v[3], v[7], tsg[6]);
}

// one LDS round trip: G stages starting at wing HALF0 over the whole frame.  REAL_OUT: scale and store only
// the real part (last pass of the inverse transform; :462 reads nothing else).
// A pass whose groups span at most 512 elements (HALF0 << G <= 512) is WAVE-LOCAL when the pass before it was one too (or
// was the first pass): thread q's group lies in elements [512 (q >> 6), 512 (q >> 6) + 512), the block its own wave wrote
// -- no other wave's stores are read, so the wave's own LDS order is all the synchronisation there is to do.
template <int G, int HALF0, bool INVERSE, int LOGN, bool REAL_OUT>
__device__ __forceinline__ void dit_pass(double2 *X, const double2 *TsL, const double2 *__restrict__ tsg, int tid, double norm)
{
    constexpr int M = 1 << G;
    constexpr int GROUPS = 1 << (LOGN - G);
    constexpr bool WAVE_LOCAL = (HALF0 << G) <= 512 && GROUPS >= 256;
#ifdef JSDR_X_FFT_WGBAR  // (probe: workgroup barriers in front of every pass, as before round 4)
    __syncthreads();
#else
    if constexpr (WAVE_LOCAL) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
#endif
#pragma unroll
    for (int q0 = 0; q0 < GROUPS; q0 += 256) {
        int q = q0 + tid;
        if (GROUPS < 256 && q >= GROUPS) continue;
        // opaque per frame: keeps the per-pass address arithmetic (64-bit twiddle addresses!) from being hoisted
        // out of the frame loop into registers the allocator then spills
        asm volatile("" : "+v"(q));
        const int j = q & (HALF0 - 1);
        const int base = (q - j) << G;
        static_assert(HALF0 % 8 == 0, "wing distances of the LDS passes are multiples of 8");
        double2 *x0 = X + xpad(base + j);  // element + HALF0*m sits (HALF0 + HALF0/8)*m slots further: immediate offsets
        constexpr int STEP = HALF0 + HALF0 / 8;
        double2 v[M];
#pragma unroll
        for (int m = 0; m < M; m++) v[m] = x0[STEP * m];
        dit_stages<G, HALF0, INVERSE, false>(v, j, TsL, tsg);
#pragma unroll
        for (int m = 0; m < M; m++) {
            if (REAL_OUT)
                x0[STEP * m].x = v[m].x * norm;
            else
                x0[STEP * m] = v[m];
        }
    }
}

// The LAST pass of the FORWARD transform at LOGN = 11 (wings H = 512 and 2H = 1024; group j holds x[j + H m], m = 0..3, and
// ends as bins j, j + H, j + 2H, j + 3H) restricted to the bins below need_end that anybody reads (|X| over the searched
// quarter band, the 204 bins around the centre: need_end = end + 102 = 614 / 1126).  A bin nobody reads is neither computed
// nor stored; the ones that are come from the same operations in the same order as dit_stages<2, H> performs them.  Lower
// half band: 24 instead of 40 operations and one store instead of four for 410 of the 512 groups.
template <int LOGN>
__device__ __forceinline__ void dit_last2_band(double2 *X, const double2 *__restrict__ tsg, int tid, int need_end)
{
    constexpr int H = 1 << (LOGN - 2);
    constexpr int STEP = H + H / 8;
    static_assert(H > kLdsTwiddleWing && H % 256 == 0, "wide stages: twiddles from the global table");
    auto bf = [](const double2 aq, const double2 bq, const double2 wv, bool want_b, double2 &ao, double2 &bo) {
        const double wr = wv.x, wi = wv.y;
        const double p1 = wr * bq.x, p2 = wi * bq.y, p3 = wr * bq.y, p4 = wi * bq.x;
        const double tr = p1 - p2;
        const double ti = p3 + p4;
        ao = make_double2(aq.x + tr, aq.y + ti);
        if (want_b) bo = make_double2(aq.x - tr, aq.y - ti);
    };
    __syncthreads();
#pragma unroll
    for (int it = 0; it < H / 256; it++) {
        int j = it * 256 + tid;
        asm volatile("" : "+v"(j));
        const bool n1 = j + H < need_end, n2 = j + 2 * H < need_end, n3 = j + 3 * H < need_end;
        double2 *x0 = X + xpad(j);
        const double2 v0 = x0[0], v1 = x0[STEP], v2 = x0[2 * STEP], v3 = x0[3 * STEP];
        const double2 w1 = tsg[(unsigned)(H - 1 + j)];
        double2 a01, b01 = v0, a23, b23 = v2;
        bf(v0, v1, w1, n1 || n3, a01, b01);  // wing H
        bf(v2, v3, w1, n1 || n3, a23, b23);
        double2 o0, o2 = a01, o1 = b01, o3 = b01;
        bf(a01, a23, tsg[(unsigned)(2 * H - 1 + j)], n2, o0, o2);  // wing 2H
        x0[0] = o0;
        if (n2) x0[2 * STEP] = o2;
        if (n1 || n3) {
            bf(b01, b23, tsg[(unsigned)(2 * H - 1 + j + H)], n3, o1, o3);
            x0[STEP] = o1;
            if (n3) x0[3 * STEP] = o3;
        }
    }
}

// The LAST pass of the inverse transform, leaving the scaled real parts (all RxDownSample reads, :462) as a COMPACT array of
// doubles over the dead image -- sample t at double slot FF_RB0 + t, the previous frame's last 26 samples (hist) in front of
// them -- so that a RxDownSample window, history included, is one contiguous run of 27 doubles: 14 conflict-free 16-byte
// reads instead of 27 eight-byte reads at a 160-byte lane stride.  Every butterfly of the pass is in registers before the
// first store (the compact slots overlap other butterflies' inputs).
constexpr int FF_RB0 = 32;
template <int G, int HALF0, int LOGN>
__device__ __forceinline__ void dit_pass_real_compact(double2 *X, const double2 *TsL, const double2 *__restrict__ tsg, int tid,
                                                      double norm, const double *hist)
{
    constexpr int M = 1 << G;
    constexpr int GROUPS = 1 << (LOGN - G);
    constexpr int NIT = (GROUPS + 255) / 256;
    constexpr int STEP = HALF0 + HALF0 / 8;
    static_assert(GROUPS % 256 == 0, "every thread owns NIT whole groups");
    double o[NIT][M];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        int q = it * 256 + tid;
        asm volatile("" : "+v"(q));
        const int j = q & (HALF0 - 1);
        const int base = (q - j) << G;
        const double2 *x0 = X + xpad(base + j);
        double2 v[M];
#pragma unroll
        for (int m = 0; m < M; m++) v[m] = x0[STEP * m];
        dit_stages<G, HALF0, true, false>(v, j, TsL, tsg);
#pragma unroll
        for (int m = 0; m < M; m++) o[it][m] = v[m].x * norm;
    }
    __syncthreads();
    double *Rb = reinterpret_cast<double *>(X);
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const int q = it * 256 + tid;
        const int j = q & (HALF0 - 1);
        const int base = (q - j) << G;
#pragma unroll
        for (int m = 0; m < M; m++) Rb[FF_RB0 + base + j + HALF0 * m] = o[it][m];
    }
    if (tid < 26) Rb[FF_RB0 - 26 + tid] = hist[tid];
}

// the passes after the first: wings 8,64,512 (and what is left)
// COMPACT (inverse only): the last pass is dit_pass_real_compact
// need_end > 0 (forward transform, LOGN = 11): only the bins below need_end are read afterwards
template <bool INVERSE, int LOGN, bool SKIP8, bool COMPACT = false>
__device__ __forceinline__ void fft_rest(double2 *X, const double2 *TsL, const double2 *__restrict__ tsg, int tid, double norm,
                                         const double *hist = nullptr, int need_end = 0)
{
    static_assert(LOGN >= 10 && LOGN <= 13, "frame sizes 1024..8192");
    if (!SKIP8) dit_pass<3, 8, INVERSE, LOGN, false>(X, TsL, tsg, tid, norm);
    if constexpr (LOGN == 13) {
        dit_pass<3, 64, INVERSE, LOGN, false>(X, TsL, tsg, tid, norm);
        dit_pass<3, 512, INVERSE, LOGN, false>(X, TsL, tsg, tid, norm);
        if constexpr (COMPACT)
            dit_pass_real_compact<1, 4096, LOGN>(X, TsL, tsg, tid, norm, hist);
        else
            dit_pass<1, 4096, INVERSE, LOGN, INVERSE>(X, TsL, tsg, tid, norm);
    } else if constexpr (LOGN == 12) {
        dit_pass<3, 64, INVERSE, LOGN, false>(X, TsL, tsg, tid, norm);
        if constexpr (COMPACT)
            dit_pass_real_compact<3, 512, LOGN>(X, TsL, tsg, tid, norm, hist);
        else
            dit_pass<3, 512, INVERSE, LOGN, INVERSE>(X, TsL, tsg, tid, norm);
    } else if constexpr (LOGN == 11) {
        dit_pass<3, 64, INVERSE, LOGN, false>(X, TsL, tsg, tid, norm);
        if constexpr (COMPACT)
            dit_pass_real_compact<2, 512, LOGN>(X, TsL, tsg, tid, norm, hist);
        else if constexpr (!INVERSE) {
            if (need_end > 0)
                dit_last2_band<LOGN>(X, tsg, tid, need_end);
            else
                dit_pass<2, 512, false, LOGN, false>(X, TsL, tsg, tid, norm);
        } else
            dit_pass<2, 512, INVERSE, LOGN, INVERSE>(X, TsL, tsg, tid, norm);
    } else {
        dit_pass<3, 64, INVERSE, LOGN, false>(X, TsL, tsg, tid, norm);
        if constexpr (COMPACT)
            dit_pass_real_compact<1, 512, LOGN>(X, TsL, tsg, tid, norm, hist);
        else
            dit_pass<1, 512, INVERSE, LOGN, INVERSE>(X, TsL, tsg, tid, norm);
    }
    __syncthreads();
}

#ifndef JSDR_FF_MINWG11
#define JSDR_FF_MINWG11 4  // (probe builds: 3 = 170 VGPRs and three workgroups a CU at n = 2048)
#endif
template <int LOGN, bool F32IN>
__global__ __launch_bounds__(256, (LOGN <= 11 ? JSDR_FF_MINWG11 : (LOGN == 12 ? 2 : 1))) void k_front_fft(FftFrontArgs a)
{
    constexpr int N = 1 << LOGN;
    constexpr int XSLOTS = N + (N >> 3);
    constexpr int G1 = (N / 8 + 255) / 256;        // first-pass groups per thread
    constexpr int KBP = ((N / 4 - 150) / 2 + 1 + 255) / 256;  // boxcar output pairs per thread
    constexpr int JB = (N / 4 + 1 + 255) / 256;    // RxDownSample outputs per thread and frame (decimation >= 4)
    extern __shared__ __align__(16) unsigned char smem[];
    double2 *X = reinterpret_cast<double2 *>(smem);       // [XSLOTS], padded image
    double2 *TsL = X + XSLOTS;                            // [64] twiddles of the narrow stages
    double *hist = reinterpret_cast<double *>(TsL + 64);  // [32]
    double *taps = hist + 32;                             // [32]
    double *redv = taps + 32;                             // [4] per-wave best value
    int *redi = reinterpret_cast<int *>(redv + 4);        // [4] per-wave best index
    // |X| and the boxcar sums: above redi for small frames, inside the (dead) upper part of X for N >= 2048
    double *P = (N >= 2048) ? reinterpret_cast<double *>(X + xpad(N / 2 + 104))  // bins up to N/2+101 are gathered
                            : reinterpret_cast<double *>(redi + 8);
    double *A = P + N / 2;
    const double2 *__restrict__ tsg = a.tw;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = blockIdx.x;
    if (tid < 2 * kLdsTwiddleWing - 1) TsL[tid] = tsg[tid];
    FftFrontState *sp = &a.st[s];
    if (tid < 26) hist[tid] = sp->hist[tid];
    (void)taps;  // (the taps are compile-time constants now: ds_tap(); the slot stays for the layout)
    double avePeakPower = sp->avePeakPower, aveCentreBin = sp->aveCentreBin;
    int centreBin = sp->centreBin;
    // :399-402 -- float expressions widened to double
    const double CFREQ_INV = (double)(1.0F - (2.0F / (1 + 1))), CFREQ_AVG = (double)(2.0F / (1 + 1));
    const double PSD_INV = (double)(1.0F - (2.0F / (10 + 1))), PSD_AVG = (double)(2.0F / (10 + 1));
    const double HOWARD = 0.9 * 32768.0;
    const int beg = a.do_up ? N / 4 : 0;
    const int end = a.do_up ? N / 2 : N / 4;
    const int D = a.decim;
    const double norm = 1.0 / (double)N;
    const int *__restrict__ raw = a.raw + (long long)s * a.stride_pairs;
    const float2 *__restrict__ rawf = a.rawf + (long long)s * a.stride_pairs;
    double2 *dm = a.dm + (long long)s * a.dm_stride;
    // diagnostics (JSDR_FFT_PHASECLK=1): thread 0 of stream 0 accumulates the clock ticks of every phase in LDS
    long long *clk = reinterpret_cast<long long *>(redi + 8 + (N >= 2048 ? 0 : 2 * N));
    long long tprev = 0;
    const bool timing = a.phase_clk != nullptr && s == 0 && tid == 0;
    if (timing)
        for (int k = 0; k < 8; k++) clk[k] = 0;
#define PHASE(k)                                     \
    if (timing) {                                    \
        const long long now_ = (long long)clock64(); \
        clk[k] += now_ - tprev;                      \
        tprev = now_;                                \
    }
    // First-pass group q = tid + 256*c owns slots 8q..8q+7, i.e. the frame elements brev(8q+m) =
    // brev3(m)*N/8 + brev(q): eight loads N/8 apart.  Frame 0 now; every later frame is fetched while its
    // predecessor is processed.
    int bq[G1];
    int pre[G1][8];
    float2 pref[G1][8];
#pragma unroll
    for (int c = 0; c < G1; c++) {
        const int q = tid + 256 * c;
        bq[c] = brevn<LOGN - 3>(q & (N / 8 - 1));
        if (N / 8 >= 256 || q < N / 8) {
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const int t = brevn<3>(m) * (N / 8) + bq[c];
                if (F32IN)
                    pref[c][m] = rawf[t];
                else
                    pre[c][m] = raw[t];
            }
        }
    }
    __syncthreads();
    if (timing) tprev = (long long)clock64();

    for (int f = 0; f < a.nframes; f++) {
        const long long t0 = (long long)f * N;  // call-relative index of the frame's first sample
        // VCO factors of this frame's RxDownSample outputs: in flight during the transforms
        long long jlo = (t0 - a.first_out + D - 1) / D;
        if (t0 <= a.first_out) jlo = 0;
        double2 cs0 = make_double2(0.0, 0.0);
        {
            const long long j = jlo + tid;
            const long long te = (long long)a.first_out + (long long)D * j;
            if (te < t0 + N && j < a.nds) cs0 = a.vco_cs[j];
        }
        // ---- forward FFT (:416-423): first pass from the load registers
#pragma unroll
        for (int c = 0; c < G1; c++) {
            const int q = tid + 256 * c;
            if (N / 8 >= 256 || q < N / 8) {
                double2 v[8];
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    if (F32IN) {
                        v[m] = make_double2((double)pref[c][m].x, (double)pref[c][m].y);
                    } else {
                        const int w = pre[c][m];
                        v[m] = make_double2((double)i16_to_float_java(java_short_add((int)(short)(w & 0xffff), a.ic)),
                                            (double)i16_to_float_java(java_short_add(w >> 16, a.qc)));
                    }
                }
                if (f + 1 < a.nframes) {
#pragma unroll
                    for (int m = 0; m < 8; m++) {
                        const long long t = t0 + N + brevn<3>(m) * (N / 8) + bq[c];
                        if (F32IN)
                            pref[c][m] = rawf[t];
                        else
                            pre[c][m] = raw[t];
                    }
                }
                if (F32IN)
                    dit_stages<3, 1, false, true>(v, 0, TsL, tsg);
                else
                    dit_first3_i16(v, tsg);
#pragma unroll
                for (int m = 0; m < 8; m++) X[xpad(8 * q + m)] = v[m];
            }
        }
        PHASE(0)
#ifdef JSDR_X_FFT_NOBAND  // (probe: the forward transform's last pass in full)
        fft_rest<false, LOGN, false>(X, TsL, tsg, tid, norm);
#else
        fft_rest<false, LOGN, false>(X, TsL, tsg, tid, norm, nullptr, LOGN == 11 ? end + 102 : 0);  // bins < end + 102 are read
#endif
        PHASE(1)
        // ---- |X| (:425-427), for the bins the boxcar reads: [beg+24, end-24) -- a quarter band, not the N/2 bins the
        // reference fills (the double-precision root is ~28 instructions a bin)
        constexpr int NBAND = N / 4 - 48;
#pragma unroll
        for (int i0 = 0; i0 < NBAND; i0 += 256) {
            const int r = i0 + tid;
            if (NBAND % 256 == 0 || r < NBAND) {
                const int i = beg + 24 + r;
                const double2 v = X[xpad(i)];
                P[i] = sqrt(v.x * v.x + v.y * v.y);
            }
        }
        __syncthreads();
        PHASE(2)
        // ---- 100-wide boxcar, summed j ascending for every i (:433-437); first maximum (:439-442).  A thread owns
        // the outputs i (even) and i+1: their windows P[i-50..i+49] and P[i-49..i+50] come out of the same 51
        // aligned 16-byte reads, each summed in its own ascending chain.
        double bestv = 0.0;  // maxBin starts at 0.0, binPos at -1
        int besti = -1;
#pragma unroll
        for (int b = 0; b < KBP; b++) {
            const int i = beg + 74 + 2 * (tid + 256 * b);  // beg is a multiple of 4: i - 50 is even
            if (i < end - 75) {
                const double2 *w = reinterpret_cast<const double2 *>(P + i - 50);
                double a0, a1;
                boxcar_pair(w, a0, a1);
                asm volatile("" : "+v"(a0), "+v"(a1));  // due here: sunk into the conditional uses below, the sums drag all 51 reads along
                if (i >= beg + 75) {
                    A[i] = a0;
                    if (bestv < a0) {  // i ascends within a thread: strict '<' keeps the first maximum
                        bestv = a0;
                        besti = i;
                    }
                }
                if (i + 1 < end - 75) {
                    A[i + 1] = a1;
                    if (bestv < a1) {
                        bestv = a1;
                        besti = i + 1;
                    }
                }
            }
        }
        wave_first_max(bestv, besti);  // (DPP: bpsk_fft.h)
        if (lane == 0) {
            redv[wave] = bestv;
            redi[wave] = besti;
        }
        __syncthreads();
        PHASE(3)
        // ---- centre-bin rule (:444-453), evaluated by every thread on the same values
        {
            double maxBin = 0.0;
            int binPos = -1;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const double ov = redv[w];
                const int oi = redi[w];
                if (oi >= 0 && (ov > maxBin || (ov == maxBin && (binPos < 0 || oi < binPos)))) {
                    maxBin = ov;
                    binPos = oi;
                }
            }
            if (centreBin < 0) centreBin = 0;
            if (centreBin > end - 1) centreBin = end - 1;
            // aveTemp is cleared per frame (:431) and only [beg+75, end-75) is filled
            const double atc = (centreBin >= beg + 75 && centreBin < end - 75) ? A[centreBin] : 0.0;
            avePeakPower = (PSD_AVG * atc) + (PSD_INV * avePeakPower);
            if (maxBin > (avePeakPower / 4) * 5 && binPos > 0) {
                aveCentreBin = (CFREQ_AVG * (double)(float)binPos) + (CFREQ_INV * aveCentreBin);
                centreBin = (int)(aveCentreBin + (double)1.0F);
            }
            if (centreBin < 102) centreBin = 102;
        }
        PHASE(4)
        // ---- inverse FFT of the 204 bins around the centre moved to bin 0 of a zeroed array (:458-459)
        if constexpr (LOGN >= 11) {
            // Only input 0 of every first-pass group can be non-zero (bins >= N/8 > 204 are zero), and a butterfly
            // whose second operand is +0 returns its first operand twice -- unless that is -0, where IEEE gives
            // (-0)+(+0) = +0: such a value takes the full first pass.  So the first pass is a broadcast, and the
            // wing-8 pass reads its eight inputs straight from the spectrum.
            constexpr int G2 = (N / 8) / 256;  // wing-8 groups per thread
            double2 vin[G2][8];
#pragma unroll
            for (int c = 0; c < G2; c++) {
                const int q = tid + 256 * c;
                const int rb = brevn<LOGN - 6>(q >> 3);
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    vin[c][m] = make_double2(0.0, 0.0);
                    if ((brevn<3>(m) << (LOGN - 6)) < 204) {  // compile time
                        const int k = (brevn<3>(m) << (LOGN - 6)) | rb;
                        if (k < 204) vin[c][m] = X[xpad(centreBin - 102 + k)];
                    }
                }
            }
            __syncthreads();  // every spectrum read is done before the image is overwritten
#pragma unroll
            for (int c = 0; c < G2; c++) {
                int q = tid + 256 * c;
                asm volatile("" : "+v"(q));
                const int j = q & 7;
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    const double2 v0 = vin[c][m];
                    const long long NEGZ = (long long)0x8000000000000000ull;
                    if (__double_as_longlong(v0.x) == NEGZ || __double_as_longlong(v0.y) == NEGZ) {
                        double2 t[8];
                        t[0] = v0;
#pragma unroll
                        for (int i = 1; i < 8; i++) t[i] = make_double2(0.0, 0.0);
                        dit_stages<3, 1, true, true>(t, 0, TsL, tsg);
                        double2 r = t[0];
#pragma unroll
                        for (int i = 1; i < 8; i++)
                            if (j == i) r = t[i];
                        vin[c][m] = r;
                    }
                }
                dit_stages<3, 8, true, false>(vin[c], j, TsL, tsg);
                double2 *x0 = X + xpad(((q - j) << 3) + j);
#pragma unroll
                for (int m = 0; m < 8; m++) x0[9 * m] = vin[c][m];
            }
            PHASE(5)
            fft_rest<true, LOGN, true, true>(X, TsL, tsg, tid, norm, hist);  // leaves re/N compact (:462)
        } else {
            // N = 1024: two inputs of a first-pass group can be non-zero; the first pass runs in full on registers
            double2 vin[G1][8];
#pragma unroll
            for (int c = 0; c < G1; c++) {
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    vin[c][m] = make_double2(0.0, 0.0);
                    if (brevn<3>(m) * (N / 8) < 204) {  // compile time: the only slots a bin below 204 can land in
                        const int k = brevn<3>(m) * (N / 8) + bq[c];
                        if (k < 204) vin[c][m] = X[xpad(centreBin - 102 + k)];
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < G1; c++) {
                const int q = tid + 256 * c;
                if (N / 8 >= 256 || q < N / 8) {
                    dit_stages<3, 1, true, true>(vin[c], 0, TsL, tsg);
#pragma unroll
                    for (int m = 0; m < 8; m++) X[xpad(8 * q + m)] = vin[c][m];
                }
            }
            PHASE(5)
            fft_rest<true, LOGN, false, true>(X, TsL, tsg, tid, norm, hist);  // leaves re/N compact (:462)
        }
        PHASE(6)
        // ---- RxDownSample(re, re) (:461-463, :470-492): outputs whose window ends inside this frame, from the compact samples
        // (sample t at double slot FF_RB0 + t, the previous frame's last 26 in front: every window is one contiguous run)
        {
            const double *Rb = reinterpret_cast<const double *>(smem);
            const bool even_d = (D & 1) == 0;  // then every window of the call ends on the same parity (N is even)
            const int par = (int)((a.first_out - t0) & 1);
#pragma unroll
            for (int b = 0; b < JB; b++) {
                const long long j = jlo + tid + 256 * b;
                const long long te = (long long)a.first_out + (long long)D * j;  // window end, call-relative
                if (te < t0 + N && j < a.nds) {
                    const int e = (int)(te - t0);  // 0..N-1 within the frame
                    double fi = 0.0;
                    if (even_d) {
                        // the 27 samples e-26 .. e as 14 aligned 16-byte reads: d[i] = slot ((e + 6) & ~1) + i
                        const double2 *w2 = reinterpret_cast<const double2 *>(Rb + ((e + FF_RB0 - 26) & ~1));
                        double d[28];
#pragma unroll
                        for (int i = 0; i < 14; i++) {
                            const double2 t = w2[i];
                            d[2 * i] = t.x;
                            d[2 * i + 1] = t.y;
                        }
                        if (par) {
#pragma unroll
                            for (int k = 0; k < 27; k++) fi += d[27 - k] * ds_tap(k);  // newest first (:479-483)
                        } else {
#pragma unroll
                            for (int k = 0; k < 27; k++) fi += d[26 - k] * ds_tap(k);
                        }
                    } else {
                        const double *w = Rb + (FF_RB0 + e);
#pragma unroll
                        for (int k = 0; k < 27; k++) fi += w[-k] * ds_tap(k);
                    }
                    const double o = fi * HOWARD;  // fi == fq: both rails get the same samples
                    const double2 cs = (b == 0) ? cs0 : a.vco_cs[j];
                    dm[64 + j] = make_double2(o * cs.x, o * cs.y);  // :515-516
                }
            }
            if (tid < 26) hist[tid] = Rb[FF_RB0 + N - 26 + tid];  // (nobody reads hist[] before the next frame's last pass)
            __syncthreads();  // every window is read before the next frame's first pass overwrites the image
        }
        PHASE(7)
    }
#undef PHASE
    __syncthreads();
    if (tid < 26) sp->hist[tid] = hist[tid];
    if (tid == 0) {
        sp->avePeakPower = avePeakPower;
        sp->aveCentreBin = aveCentreBin;
        sp->centreBin = centreBin;
        if (timing)
            for (int k = 0; k < 8; k++) a.phase_clk[k] = clk[k];
    }
}

template <int LOGN, bool F32IN>
static int launch_front_fft_t(const FftFrontArgs &a, int nstreams, hipStream_t st)
{
    constexpr int N = 1 << LOGN;
    constexpr size_t lds = sizeof(double2) * ((size_t)N + (N >> 3) + 64) + sizeof(double) * (32 + 32 + 4) + 32 + 64 +
                           (N >= 2048 ? 0 : sizeof(double) * (size_t)N);
    JSDR_LDS_ATTR((k_front_fft<LOGN, F32IN>), lds);
    hipLaunchKernelGGL((k_front_fft<LOGN, F32IN>), dim3((unsigned)nstreams), dim3(256), lds, st, a);
    JSDR_LAUNCH_CHECK();
    return JSDR_OK;
}

int launch_front_fft(const FftFrontArgs &a, int nstreams, hipStream_t st)
{
    const bool f32 = a.rawf != nullptr;
    switch (a.logn) {
        case 10: return f32 ? launch_front_fft_t<10, true>(a, nstreams, st) : launch_front_fft_t<10, false>(a, nstreams, st);
        case 11: return f32 ? launch_front_fft_t<11, true>(a, nstreams, st) : launch_front_fft_t<11, false>(a, nstreams, st);
        case 12: return f32 ? launch_front_fft_t<12, true>(a, nstreams, st) : launch_front_fft_t<12, false>(a, nstreams, st);
        case 13: return f32 ? launch_front_fft_t<13, true>(a, nstreams, st) : launch_front_fft_t<13, false>(a, nstreams, st);
        default: JSDR_REQUIRE(false, "bpsk: FFT-acquire frame of 2^%d samples is not supported (1024..8192)", a.logn);
    }
}

// per-stage twiddles Ts[half-1+j] = W_n^(j*n/(2*half)), j < half, half = 1,2,..,n/2 (n-1 entries), every value taken
// from the SAME table as oracle/o_fft.c jo_fft_twiddles_f64 builds (long double + one rounding, exact on the axes)
void fft_twiddles_f64(std::vector<double2> &w, int n)
{
    std::vector<double2> base((size_t)n / 2);
    for (int k = 0; k < n / 2; k++) {
        long double ang = 2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)n;
        base[k] = make_double2((double)cosl(ang), (double)(-sinl(ang)));
    }
    base[0] = make_double2(1.0, -0.0);
    if (n >= 4) base[n / 4] = make_double2(0.0, -1.0);
    w.assign((size_t)n, make_double2(0.0, 0.0));
    for (int half = 1; half < n; half <<= 1) {
        const int step = n / (2 * half);
        for (int j = 0; j < half; j++) w[(size_t)half - 1 + j] = base[(size_t)j * step];
    }
}

}  // namespace jsdr
