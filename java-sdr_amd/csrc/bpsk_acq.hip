// bpsk_acq.hip -- FUNcubeBPSKDemod FFT-acquire front end, doBufferFFT (FUNcubeBPSKDemod.java:406-464), in THREE PHASES
// (round 6).  What the reference's frame loop carries from frame to frame is three numbers (avePeakPower, aveCentreBin,
// centreBin, :403-405, updated at :444-453) and the down-sampler's 26-sample history (:466-492).  Everything else is a
// pure function of the frame:
//   phase A  k_acq_fwd   one (stream, frame) per workgroup pass, any number in flight: forward transform (:416-423), |X| on the
//                        searched band (:425-427), the 100-wide boxcar and its first maximum (:433-443).  Leaves per frame:
//                        the spectrum bins a gather can reach, the boxcar sums of the band, (maxBin, binPos).
//   phase B  k_acq_scan  one wave per stream, sequential over the call's frames: :444-453 exactly (clamps, float-valued
//                        factors, avePsd[centreBin] read from phase A's band -- zero outside the loop's range, as Java's
//                        `new double[]` leaves it).  Leaves centreBin per frame and the state.
//   phase C  k_acq_inv   one (stream, frame) per workgroup pass: the 204 bins around the frame's centre to bin 0 (:458),
//                        inverse transform (:459), RxDownSample(re, re) (:461-463, :470-492) for every window that lies
//                        inside the frame, the VCO mix (:515-516).  Leaves the frame's first and last 26 real samples.
//   edges    k_acq_edges the few windows per frame that reach back into the previous frame (or, for the call's first
//                        frame, into the stream's history), and the history the call leaves.
// The grid is FRAMES, not streams: occupancy no longer depends on the number of streams (one recorded stream of 512 frames
// fills the chip), the kernels of the two transforms are separate and simpler than the fused front end (k_front_fft sat at
// its 128-VGPR cap and spilled), and a thread holds 16 points (two waves per 2048-sample frame, four radix-2 stages per LDS
// round trip instead of three).
//
// Compiled with -ffp-contract=off.  The transform is the oracle's radix-2 decimation-in-time network (oracle/o_fft.c
// jo_fft_f64) on the same twiddle table, butterfly for butterfly:
//     t = w*b (tr = wr*br - wi*bi, ti = wr*bi + wi*br), a' = a + t, b' = a - t
// only the grouping of the stages into passes differs -- spectra, centre bins and everything downstream are bit-identical
// to the oracle and to k_front_fft (which keeps serving calls of ONE frame per stream: a live receive()).
#include "bpsk_fft.h"
#include <math.h>

namespace jsdr {

// ---- LDS image of a frame: element e at 16-byte slot (e & ~15) | ((e & 15) ^ key(e >> 4)), key(r) = (r ^ r>>4 ^ r>>8) & 15.
// Every pass reads / writes elements e0 | (H m), m = 0..2^G-1, lanes over e0: within a row of 16 the XOR is a permutation
// (consecutive lanes -> distinct bank groups), and rows that differ in ANY nibble get different keys, so the first pass --
// a lane owns a whole row, eight lanes of a store are eight rows apart in bits 4.. after the bit reversal of the coalesced
// loads -- stores conflict-free as well.  No padding: four 2048-sample frames are 128 KB of a CU's 160.
__host__ __device__ constexpr int acq_key(int row) { return (row ^ (row >> 4) ^ (row >> 8)) & 15; }
__device__ __forceinline__ int acq_slot(int e) { return (e & ~15) | ((e & 15) ^ acq_key(e >> 4)); }
// element e0 | HM, HM a multiple of 16 whose bits are clear in e0 (every pass: e0 = base + j, HM = H m): the key splits into
// key(e0 >> 4) ^ key(HM >> 4), and with m an unrolled loop index the second half folds to a constant
__device__ __forceinline__ int acq_slot_hm(int e0, int HM)
{
    return ((e0 & ~15) + HM) | (((e0 & 15) ^ acq_key(e0 >> 4)) ^ acq_key(HM >> 4));
}

template <int BITS>
__device__ __forceinline__ int acq_brev(int x)
{
    return BITS == 0 ? 0 : (int)(__brev((unsigned)x) >> (32 - (BITS ? BITS : 1)));
}

constexpr int ACQ_TWL = 256;  // stages with wing <= 128 read their twiddles from an LDS copy of tw[0..254]

// twiddle jj of the stage with wing HALF: tw[HALF - 1 + jj] = W_n^(jj n / (2 HALF))
template <int HALF, bool UNIFORM>
__device__ __forceinline__ double2 acq_tw(const double2 *TsL, const double2 *__restrict__ tsg, int jj)
{
    if (!UNIFORM && 2 * HALF <= ACQ_TWL) return TsL[HALF - 1 + jj];
    if (UNIFORM) {  // the same entry in every lane: a scalar load
        typedef const __attribute__((address_space(4))) double *ctab_t;
        ctab_t t = (ctab_t)tsg;
        return make_double2(t[2 * (HALF - 1 + jj)], t[2 * (HALF - 1 + jj) + 1]);
    }
    return tsg[(unsigned)(HALF - 1 + jj)];
}

// G consecutive stages of the network on the 2^G values v[m] = x[base + j + HALF0 m]: the oracle's butterfly, unchanged
template <int G, int HALF0, bool INVERSE, bool UNIFORM>
__device__ __forceinline__ void acq_stages(double2 (&v)[1 << G], int j, const double2 *TsL, const double2 *__restrict__ tsg)
{
    constexpr int M = 1 << G;
#pragma unroll
    for (int t = 0; t < G; t++) {
#ifdef JSDR_ACQ_STAGE_FENCE
        if (!UNIFORM) __builtin_amdgcn_sched_barrier(0);  // a stage's twiddles are requested when the stage before is done, not all up front
#endif
        double2 w[M / 2];
#pragma unroll
        for (int u = 0; u < (1 << t); u++) {
            if (t == 0) w[u] = acq_tw<HALF0, UNIFORM>(TsL, tsg, j + HALF0 * u);
            if (t == 1) w[u] = acq_tw<HALF0 * 2, UNIFORM>(TsL, tsg, j + HALF0 * u);
            if (t == 2) w[u] = acq_tw<HALF0 * 4, UNIFORM>(TsL, tsg, j + HALF0 * u);
            if (t == 3) w[u] = acq_tw<HALF0 * 8, UNIFORM>(TsL, tsg, j + HALF0 * u);
        }
#pragma unroll
        for (int m = 0; m < M; m++) {
            if ((m >> t) & 1) continue;
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

// The first FOUR stages (wings 1, 2, 4, 8) of the FORWARD transform on sixteen converted int16 samples, the multiplications by
// the table's trivial twiddles 1 = (1, -0) and -i = (0, -1) not performed (w = 1: t = b; w = -i: t = (b.y, -b.x)).  Same
// results to the last bit, signs of zeros included, because no value in this part of the network is ever -0.0: a converted
// int16 sample is never -0.0, and a sum or a difference is -0.0 only if an operand already is (bpsk_fft.hip dit_first3_i16
// has the argument in full).  Float input may hold -0.0f and takes acq_stages.
__device__ __forceinline__ void acq_first4_i16(double2 (&v)[16], const double2 (&w)[8])  // w = tw[4], tw[6], tw[8..10], tw[12..14]
{
    auto bf1 = [](double2 &a, double2 &b) {
        const double2 x = a, y = b;
        a = make_double2(x.x + y.x, x.y + y.y);
        b = make_double2(x.x - y.x, x.y - y.y);
    };
    auto bfi = [](double2 &a, double2 &b) {
        const double2 x = a, y = b;
        a = make_double2(x.x + y.y, x.y - y.x);
        b = make_double2(x.x - y.y, x.y + y.x);
    };
    auto bfw = [](double2 &a, double2 &b, const double2 w) {
        const double2 x = a, y = b;
        const double p1 = w.x * y.x, p2 = w.y * y.y, p3 = w.x * y.y, p4 = w.y * y.x;
        const double tr = p1 - p2, ti = p3 + p4;
        a = make_double2(x.x + tr, x.y + ti);
        b = make_double2(x.x - tr, x.y - ti);
    };
#pragma unroll
    for (int h = 0; h < 16; h += 8) {
        // wing 1: tw[0] = 1
        bf1(v[h + 0], v[h + 1]);
        bf1(v[h + 2], v[h + 3]);
        bf1(v[h + 4], v[h + 5]);
        bf1(v[h + 6], v[h + 7]);
        // wing 2: tw[1] = 1, tw[2] = -i
        bf1(v[h + 0], v[h + 2]);
        bfi(v[h + 1], v[h + 3]);
        bf1(v[h + 4], v[h + 6]);
        bfi(v[h + 5], v[h + 7]);
        // wing 4: tw[3] = 1, tw[4] = W8, tw[5] = -i, tw[6] = W8^3
        bf1(v[h + 0], v[h + 4]);
        bfw(v[h + 1], v[h + 5], w[0]);
        bfi(v[h + 2], v[h + 6]);
        bfw(v[h + 3], v[h + 7], w[1]);
    }
    // wing 8: tw[7 + j] = W16^j; j = 0: 1, j = 4: -i
    bf1(v[0], v[8]);
    bfw(v[1], v[9], w[2]);
    bfw(v[2], v[10], w[3]);
    bfw(v[3], v[11], w[4]);
    bfi(v[4], v[12]);
    bfw(v[5], v[13], w[5]);
    bfw(v[6], v[14], w[6]);
    bfw(v[7], v[15], w[7]);
}

// The workgroup barrier of these kernels orders LDS traffic ONLY.  __syncthreads() is a workgroup-scope release / acquire
// fence pair around s_barrier, and the release half waits for EVERY outstanding memory operation of the wave (s_waitcnt
// vmcnt(0)): the next frame's samples requested a moment ago, the spectrum rows and boxcar sums just stored -- a round trip to
// HBM at every barrier, with two waves a SIMD to cover it.  Nothing a thread of these kernels writes to global memory is read
// by another thread of the same launch, so the barrier waits for the wave's LDS operations and nothing else.
template <int T>
__device__ __forceinline__ void acq_barrier()
{
    if constexpr (T <= 64) {  // one wave per frame: its own LDS order is all there is to keep
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    } else {
#ifdef JSDR_X_ACQ_NOBAR  // (timing probe only, wrong data: what lock-step at the barriers costs)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
    }
}

// one LDS round trip in the middle of a transform: G stages starting at wing HALF0 over the whole frame, in place (a thread
// writes the slots it read)
template <int G, int HALF0, bool INVERSE, int LOGN>
__device__ __forceinline__ void acq_mid_pass(double2 *X, const double2 *TsL, const double2 *__restrict__ tsg, int tid)
{
    constexpr int N = 1 << LOGN, T = N / 16, M = 1 << G, NG = 16 >> G;
    static_assert(HALF0 % 16 == 0, "the passes behind the first move whole rows");
#pragma unroll
    for (int it = 0; it < NG; it++) {
        const int q = tid + T * it;
        const int j = q & (HALF0 - 1);
        const int e0 = ((q - j) << G) + j;
        double2 v[M];
#pragma unroll
        for (int m = 0; m < M; m++) v[m] = X[acq_slot_hm(e0, HALF0 * m)];
        acq_stages<G, HALF0, INVERSE, false>(v, j, TsL, tsg);
#pragma unroll
        for (int m = 0; m < M; m++) X[acq_slot_hm(e0, HALF0 * m)] = v[m];
    }
}

// the grouping of a transform's stages into passes: 4 in registers first, then G2 (4), then what is left
template <int LOGN>
struct AcqPlan {
    static_assert(LOGN >= 10 && LOGN <= 13, "frames of 1024 .. 8192 samples");
    static constexpr int G3 = LOGN == 13 ? 3 : LOGN - 8;  // third pass
    static constexpr int G4 = LOGN == 13 ? 2 : 0;         // fourth pass (8192 samples only)
    static constexpr int GL = G4 ? G4 : G3;               // the last pass's stages
};

// ---- twiddles a thread keeps in registers for the whole launch: the G stages of a pass that starts at wing HALF0, for group
// position j -- stage t, entry u at w[(1 << t) - 1 + u] = tw[(HALF0 << t) - 1 + j + HALF0 u].  The frame loop then holds no
// global load but the samples' own: on this part VMEM operations return in order, so a wait for a twiddle requested after the
// next frame's samples is a wait for those samples (a round trip to HBM at two waves a SIMD).
template <int G, int HALF0>
__device__ __forceinline__ void acq_load_tw(double2 (&w)[(1 << G) - 1], int j, const double2 *tsg_)
{
    // through a pointer the compiler cannot see through: loads it can prove invariant are sunk to their first use, below the
    // barrier's memory clobber and below the next frame's samples -- the very order these requests are here to avoid
    // (as an integer, and back into the GLOBAL address space: a laundered generic pointer gives flat loads, which count as LDS
    //  operations too and make every wait a wait for everything)
    typedef double d2v_ __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(1))) const d2v_ gbl_d2v_;
    unsigned long long ta = (unsigned long long)tsg_;
    asm volatile("" : "+s"(ta));
    gbl_d2v_ *tsg = (gbl_d2v_ *)ta;
#pragma unroll
    for (int t = 0; t < G; t++)
#pragma unroll
        for (int u = 0; u < (1 << t); u++) {
            const d2v_ x = tsg[(unsigned)((HALF0 << t) - 1 + j + HALF0 * u)];
            w[(1 << t) - 1 + u] = make_double2(x.x, x.y);
        }
}
// stages [T0, T1) of such a pass on v[m] = x[base + j + HALF0 m], twiddles from the thread's registers
template <int G, int T0, int T1, bool INVERSE>
__device__ __forceinline__ void acq_stages_w(double2 (&v)[1 << G], const double2 (&w)[(1 << G) - 1])
{
    constexpr int M = 1 << G;
#pragma unroll
    for (int t = T0; t < T1; t++) {
#pragma unroll
        for (int m = 0; m < M; m++) {
            if ((m >> t) & 1) continue;
            const double2 wv = w[(1 << t) - 1 + (m & ((1 << t) - 1))];
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

// the largest of the wave's non-negative doubles, in every lane: row shifts and row broadcasts on the two halves (DPP moves
// cost an instruction each; the shuffle form goes through the LDS crossbar, ~100 cycles a step with nothing to cover it)
__device__ __forceinline__ double acq_wave_max(double v)
{
#define ACQ_DPP_MAX(ctrl, rmask)                                                                                  \
    {                                                                                                             \
        const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), ctrl, rmask, 0xf, false); \
        const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), ctrl, rmask, 0xf, false); \
        v = fmax(v, __hiloint2double(hi, lo));                                                                    \
    }
    ACQ_DPP_MAX(0x111, 0xf)  // row_shr:1
    ACQ_DPP_MAX(0x112, 0xf)  // row_shr:2
    ACQ_DPP_MAX(0x114, 0xf)  // row_shr:4
    ACQ_DPP_MAX(0x118, 0xf)  // row_shr:8 -- lane 15 of every row holds its row's maximum
    ACQ_DPP_MAX(0x142, 0xa)  // row_bcast:15 into rows 1 and 3
    ACQ_DPP_MAX(0x143, 0xc)  // row_bcast:31 into rows 2 and 3 -- lane 63 holds the wave's
#undef ACQ_DPP_MAX
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

typedef const __attribute__((address_space(4))) double *acq_ctab_t;  // constant address space: uniform entries become scalar loads
__device__ __forceinline__ double2 acq_tw_s(const double2 *tsg, int i)
{
    acq_ctab_t t = (acq_ctab_t)tsg;
    return make_double2(t[2 * i], t[2 * i + 1]);
}

// ============================================================================================================= phase A
template <int LOGN, bool F32IN>
__global__ __launch_bounds__((1 << LOGN) / 16, 2) void k_acq_fwd(AcqArgs a)
{
    constexpr int N = 1 << LOGN, T = N / 16;
    using Plan = AcqPlan<LOGN>;
    constexpr int GL = Plan::GL, ML = 1 << GL, NGL = 16 >> GL, HL = N >> GL;  // the last pass: wings HL .. N/2
    constexpr int NA = N / 4 - 150;                                           // boxcar outputs (:433)
    constexpr int RB = (NA + T - 1) / T + (((NA + T - 1) / T) % 2 == 0 ? 1 : 0);  // per thread, odd: 24 / 40-byte lane strides hit distinct banks
    extern __shared__ __align__(16) unsigned char smem[];
    double2 *X = reinterpret_cast<double2 *>(smem);            // [N] the image
    double2 *TsL = X + N;                                      // [ACQ_TWL]
    double *redv = reinterpret_cast<double *>(TsL + ACQ_TWL);  // [8] per-wave best value
    int *redi = reinterpret_cast<int *>(redv + 8);             // [8] per-wave best index
    double *P = reinterpret_cast<double *>(smem);              // |X| over [beg + 24, end - 24): over the image, dead by then
    long long *clkL = reinterpret_cast<long long *>(redi + 8);  // [8]
    const bool timing = a.clk != nullptr && blockIdx.x == 0 && threadIdx.x == 0;
    long long tprev = 0;
    if (timing)
        for (int k = 0; k < 8; k++) clkL[k] = 0;
    const double2 *__restrict__ tsg = a.tw;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long wg_t0 = a.clk != nullptr ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
    for (int i = tid; i < ACQ_TWL - 1; i += T) TsL[i] = tsg[i];
    const int beg = a.do_up ? N / 4 : 0;
    const int end = a.do_up ? N / 2 : N / 4;
    const int pbase = beg + 24;
    const long long nfr = (long long)a.S * a.F;
    // the first pass's non-trivial twiddles tw[4], tw[6] (wing 4) and tw[8..10], tw[12..14] (wing 8): uniform, scalar registers
    double2 w1[8];
    {
        constexpr int idx[8] = {4, 6, 8, 9, 10, 12, 13, 14};
#pragma unroll
        for (int i = 0; i < 8; i++) w1[i] = acq_tw_s(tsg, idx[i]);
    }
    // first-pass group q = brev(tid): its sixteen inputs are the frame's elements brev4(m) N/16 + tid -- coalesced loads
    int pre[16];
    float2 pref[16];
    auto fetch = [&](int s, int f) {
        const long long off = (long long)s * a.stride_pairs + (long long)(a.f0 + f) * N + tid;
#pragma unroll
        for (int m = 0; m < 16; m++) {
            if (F32IN)
                pref[m] = a.rawf[off + acq_brev<4>(m) * (N / 16)];
            else
                pre[m] = a.raw[off + acq_brev<4>(m) * (N / 16)];
        }
    };
    // Frames are handed out in RUNS of a.run consecutive frames, a ticket each from one counter: with an equal share per workgroup
    // (all 1024 resident from the start, two waves a SIMD) the workgroups of one launch ended between 3.3 and 5.0 ms -- a wave's
    // pace depends on whom it shares its SIMD with -- and the launch took as long as the slowest.  The next run's ticket is taken in
    // the current run's first frame (thread 0, handed round through LDS), so the request for the next frame's samples never waits
    // for it.
    // A run lies inside ONE stream (a.rps runs a stream): (stream, frame) come from a ticket by one 32-bit division a run, and no
    // 64-bit division is left in the frame loop (two of them a frame -- frame id / F, window index / D -- were a tenth of k_acq_inv).
    const int K = a.run;
    const int nruns = a.S * a.rps;
    (void)nfr;
    int *tkL = reinterpret_cast<int *>(clkL + 8);  // [2]
    if (tid == 0) tkL[0] = (int)atomicAdd(a.tickets + 0, 1u);
    acq_barrier<T>();
    int r_next = tkL[0];
    int s = 0, f = 0, fe = 0;       // this frame: stream, frame of the launch, end of the run
    int sn = 0, fn = 0, fen = 0;    // the frame this workgroup takes next
    bool have = r_next < nruns;
    bool first = true;              // the first frame of a run: take the next run's ticket
    auto run_of = [&](int r, int &s_, int &f_, int &fe_) {
        s_ = (int)((unsigned)r / (unsigned)a.rps);
        f_ = (r - s_ * a.rps) * K;
        fe_ = f_ + K < a.F ? f_ + K : a.F;
    };
    if (have) {
        run_of(r_next, s, f, fe);
        fetch(s, f);
    }
    if (timing) tprev = (long long)clock64();
    while (have) {
        const long long g = (long long)s * a.F + f;  // the frame's row in what the phases hand each other
        unsigned tk = 0;
        if (first && tid == 0) tk = atomicAdd(a.tickets + 0, 1u);
        // opaque per frame: nothing derived from the thread index is loop invariant, or LLVM hoists every address of every pass out
        // of the frame loop and spills them (the same trap as in k_front_fft)
        int tf = tid;
        asm volatile("" : "+v"(tf));
        // ---- forward transform (:416-423): the first four stages from the load registers
        {
            double2 v[16];
#pragma unroll
            for (int m = 0; m < 16; m++) {
                if (F32IN) {
                    v[m] = make_double2((double)pref[m].x, (double)pref[m].y);
                } else {
                    double di, dq;
                    fm_convert(pre[m], a.ic, a.qc, true, di, dq);  // both rails on the packed FP32 pipe (common.h)
                    v[m] = make_double2(di, dq);
                }
            }
            if (F32IN)
                acq_stages<4, 1, false, true>(v, 0, TsL, tsg);
            else
                acq_first4_i16(v, w1);
            const int q1f = acq_brev<LOGN - 4>(tf);
            const int key = acq_key(q1f);
#pragma unroll
            for (int m = 0; m < 16; m++) X[16 * q1f + (m ^ key)] = v[m];
        }
        acq_barrier<T>();
        ACQ_PHASE(0)
        // The last pass's twiddles (wings >= 256: no room in LDS) are requested HERE, a whole pass ahead of their use and ahead of
        // the next frame's samples: VMEM operations return in order on this part, so a wait for a twiddle requested after the
        // samples would be a wait for the samples (a round trip to HBM at two waves a SIMD).
        double2 twl[NGL][ML - 1];
#pragma unroll
        for (int it = 0; it < NGL; it++) acq_load_tw<GL, HL>(twl[it], tf + T * it, tsg);
        __builtin_amdgcn_sched_barrier(0);
        acq_mid_pass<4, 16, false, LOGN>(X, TsL, tsg, tf);
        {
            // (pinned HERE by an opaque zero: left alone the compiler moves this store up beside the atomic -- same condition -- and
            //  waits for the ticket's round trip at the top of the frame; the wave's partner then waits for it at the next barrier)
            int zlate = 0;
            asm volatile("" : "+v"(zlate));
            if (first && tid == 0) tkL[1] = (int)tk + zlate;
        }
        acq_barrier<T>();
        ACQ_PHASE(1)
        r_next = tkL[1];
        // the frame this workgroup takes next (or this one again)
        bool more = true;
        if (f + 1 < fe) {
            sn = s;
            fn = f + 1;
            fen = fe;
        } else if (r_next < nruns) {
            run_of(r_next, sn, fn, fen);
        } else {
            sn = s;
            fn = f;
            fen = fe;
            more = false;
        }
        if constexpr (Plan::G4 != 0) {
            acq_mid_pass<Plan::G3, 256, false, LOGN>(X, TsL, tsg, tf);
            acq_barrier<T>();
        }
        // ---- last pass: wings HL .. N/2, group j holds x[j + HL m] and ends as the bins j + HL m.  Only the bins somebody reads
        // are formed in its last stage: the gather's reach (acq_spec_index) and |X| over [beg + 24, end - 24) (inside it).
        // The next frame's samples are requested behind the frame's last request for a twiddle; from here to the next frame's first
        // pass no wait for a global load is left.
        double2 o[NGL][ML];
#pragma unroll
        for (int it = 0; it < NGL; it++) {
            const int j = tf + T * it;
#pragma unroll
            for (int m = 0; m < ML; m++) o[it][m] = X[acq_slot_hm(j, HL * m)];
        }
        // (unconditional: under a branch the wait counts of everything behind it are the minimum over both paths, i.e. a wait for
        //  a twiddle becomes a wait for these samples again; the workgroup's last frame requests itself once more)
        fetch(sn, fn);
        __builtin_amdgcn_sched_barrier(0);
        acq_barrier<T>();  // the image is dead: |X| goes over it
        ACQ_PHASE(2)
        double2 *specg = a.spec + g * a.nsb;
#pragma unroll
        for (int it = 0; it < NGL; it++) {
            const int j = tf + T * it;
            acq_stages_w<GL, 0, GL - 1, false>(o[it], twl[it]);  // all but the last stage in full
            // last stage, wing N/2: bins ba = j + HL m and bb = ba + N/2
#pragma unroll
            for (int m = 0; m < ML / 2; m++) {
                const int ba = j + HL * m, bb = ba + N / 2;
                const int ia = acq_spec_index(ba, N, a.do_up), ib = acq_spec_index(bb, N, a.do_up);
                if (ia < 0 && ib < 0) continue;
                const double2 wv = twl[it][ML / 2 - 1 + m];
                const double2 aq = o[it][m], bq = o[it][m + ML / 2];
                const double p1 = wv.x * bq.x, p2 = wv.y * bq.y, p3 = wv.x * bq.y, p4 = wv.y * bq.x;
                const double tr = p1 - p2;
                const double ti = p3 + p4;
                if (ia >= 0) {
                    const double2 r = make_double2(aq.x + tr, aq.y + ti);
#ifndef JSDR_X_ACQ_NOSPEC  // (timing probe only: the spectrum rows not stored)
                    specg[ia] = r;
#endif
                    if (ba >= pbase && ba < end - 24) P[ba - pbase] = sqrt(r.x * r.x + r.y * r.y);  // :425-427
                }
                if (ib >= 0) {
                    const double2 r = make_double2(aq.x - tr, aq.y - ti);
#ifndef JSDR_X_ACQ_NOSPEC
                    specg[ib] = r;
#endif
                    if (bb >= pbase && bb < end - 24) P[bb - pbase] = sqrt(r.x * r.x + r.y * r.y);
                }
            }
        }
        acq_barrier<T>();
        ACQ_PHASE(3)
        // ---- 100-wide boxcar, summed j ascending for every i (:433-437); first maximum (:439-442).  A thread owns RB consecutive
        // outputs i0 .. i0 + RB - 1: their windows P[i - 50 .. i + 49] are one run of 99 + RB values, each output its own
        // ascending chain.
        double bestv = 0.0;  // maxBin starts at 0.0, binPos at -1
        int besti = -1;
        {
            const int i0 = beg + 75 + RB * tf;
#ifdef JSDR_X_ACQ_NOBOX  // (timing probe only: no boxcar)
            if (false) {
#else
            if (i0 < end - 75) {
#endif
                const double *w = P + (i0 - 50 - pbase);
                double acc[RB];
                constexpr int NV = 99 + RB, CH = 8, NCH = (NV + CH - 1) / CH;
                double cur[CH], nxt[CH];
#pragma unroll
                for (int u = 0; u < CH; u++) cur[u] = w[u];
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    if (c + 1 < NCH) {
#pragma unroll
                        for (int u = 0; u < CH; u++)
                            if ((c + 1) * CH + u < NV) nxt[u] = w[(c + 1) * CH + u];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < CH; u++) {
                        const int k = c * CH + u;
                        if (k < NV) {
#pragma unroll
                            for (int r = 0; r < RB; r++) {
                                if (k == r) acc[r] = cur[u];  // 0.0 + x == x for the |X| values (never -0.0)
                                if (k > r && k < r + 100) acc[r] += cur[u];
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < CH; u++) cur[u] = nxt[u];
                }
                double *ab = a.aband + g * a.na + (i0 - (beg + 75));
#pragma unroll
                for (int r = 0; r < RB; r++) {
                    if (i0 + r < end - 75) {
                        ab[r] = acc[r];
                        if (bestv < acc[r]) {  // i ascends within a thread: strict '<' keeps the first maximum
                            bestv = acc[r];
                            besti = i0 + r;
                        }
                    }
                }
            }
        }
        ACQ_PHASE(4)
        // the wave's first maximum: the largest value (sums of |X| are never negative), and of the lanes that hold it the
        // lowest -- outputs ascend with the lane, and a lane kept the first of its own
        {
            const double mv = acq_wave_max(bestv);
            int mi = -1;
            if (mv > 0.0) {
                const unsigned long long bal = __ballot(bestv == mv && besti >= 0);
                mi = __builtin_amdgcn_readlane(besti, (int)__builtin_ctzll(bal));
            }
            bestv = mv;
            besti = mi;
        }
        if constexpr (T > 64) {
            if (lane == 0) {
                redv[wave] = bestv;
                redi[wave] = besti;
            }
            acq_barrier<T>();
            if (tid == 0) {
                double mv = 0.0;
                int mi = -1;
#pragma unroll
                for (int w = 0; w < T / 64; w++) {
                    const double ov = redv[w];
                    const int oi = redi[w];
                    if (oi >= 0 && (ov > mv || (ov == mv && (mi < 0 || oi < mi)))) {
                        mv = ov;
                        mi = oi;
                    }
                }
                AcqPeak pk;
                pk.maxBin = mv;
                pk.binPos = mi;
                pk.pad = 0;
                a.peak[g] = pk;
            }
        } else {
            if (tid == 0) {
                AcqPeak pk;
                pk.maxBin = besti >= 0 ? bestv : 0.0;
                pk.binPos = besti;
                pk.pad = 0;
                a.peak[g] = pk;
            }
            acq_barrier<T>();  // P is read before the next frame's first pass stores over it
        }
        ACQ_PHASE(5)
        first = fn != f + 1 || sn != s;
        have = more;
        s = sn;
        f = fn;
        fe = fen;
    }
    if (timing)
        for (int k = 0; k < 8; k++) a.clk[k] = clkL[k];
    if (a.clk != nullptr && threadIdx.x == 0 && blockIdx.x < 4096) {  // every workgroup's first and last tick (residency / balance)
        a.clk[16 + 2 * blockIdx.x] = wg_t0;
        a.clk[16 + 2 * blockIdx.x + 1] = (long long)__builtin_amdgcn_s_memrealtime();
    }
}

// ============================================================================================================= phase B
// One wave per stream.  Lane l holds frame f0' + l of a block of 64: avePsd[centreBin] read under the centre bin the block
// STARTS with (64 loads in flight instead of one dependent load per frame).  The rule runs over the block in order, every lane
// on the same values; when a frame moves the centre bin, the frames behind it are re-read under the new one.  In the steady
// state (a locked carrier's centre bin moves rarely) a block is one round trip to memory.
__global__ __launch_bounds__(64) void k_acq_scan(AcqArgs a)
{
    const int s = blockIdx.x, lane = threadIdx.x;
    const int n = a.n;
    const int beg = a.do_up ? n / 4 : 0;
    const int end = a.do_up ? n / 2 : n / 4;
    FftFrontState *sp = &a.st[s];
    double avePeakPower = sp->avePeakPower, aveCentreBin = sp->aveCentreBin;
    int centreBin = sp->centreBin;
    // :399-402 -- float expressions widened to double
    const double CFREQ_INV = (double)(1.0F - (2.0F / (1 + 1))), CFREQ_AVG = (double)(2.0F / (1 + 1));
    const double PSD_INV = (double)(1.0F - (2.0F / (10 + 1))), PSD_AVG = (double)(2.0F / (10 + 1));
    const long long g0 = (long long)s * a.F;
    int f = 0;
    while (f < a.F) {
        // :444-445 for the block's first frame; the lanes behind it assume the centre bin stays
        int cb = centreBin;
        if (cb < 0) cb = 0;
        if (cb > end - 1) cb = end - 1;
        const int fl = f + lane;
        double atc = 0.0, mb = 0.0;
        int bp = -1;
        if (fl < a.F) {
            const AcqPeak pk = a.peak[g0 + fl];
            mb = pk.maxBin;
            bp = pk.binPos;
            // avePsd is cleared per frame (:431) and only [beg + 75, end - 75) is filled
            if (cb >= beg + 75 && cb < end - 75) atc = a.aband[(g0 + fl) * a.na + (cb - (beg + 75))];
        }
        const int cnt = (a.F - f) < 64 ? (a.F - f) : 64;
        int done = 0;
        for (int l = 0; l < cnt; l++) {
            // (lane l's values to every lane by v_readlane -- the loop index is uniform; __shfl goes through the LDS crossbar: five
            //  dependent ~100-cycle round trips a frame were most of this kernel)
            const double atc_l = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(atc), l), __builtin_amdgcn_readlane(__double2loint(atc), l));
            const double mb_l = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(mb), l), __builtin_amdgcn_readlane(__double2loint(mb), l));
            const int bp_l = __builtin_amdgcn_readlane(bp, l);
            if (centreBin < 0) centreBin = 0;
            if (centreBin > end - 1) centreBin = end - 1;
            avePeakPower = (PSD_AVG * atc_l) + (PSD_INV * avePeakPower);
            if (mb_l > (avePeakPower / 4) * 5 && bp_l > 0) {
                aveCentreBin = (CFREQ_AVG * (double)(float)bp_l) + (CFREQ_INV * aveCentreBin);
                centreBin = (int)(aveCentreBin + (double)1.0F);
            }
            if (centreBin < 102) centreBin = 102;
            if (lane == 0) a.cbin[g0 + f + l] = centreBin;
            done = l + 1;
            // the next frame reads avePsd under THIS frame's centre bin (after its own clamps): if that is not what the block
            // was loaded under, reload from the next frame on
            int cn = centreBin;
            if (cn > end - 1) cn = end - 1;
            if (cn != cb) break;
        }
        f += done;
    }
    if (lane == 0) {
        sp->avePeakPower = avePeakPower;
        sp->aveCentreBin = aveCentreBin;
        sp->centreBin = centreBin;
    }
}

// ============================================================================================================= phase C
constexpr int ACQ_RB0 = 32;  // the compact real samples start at double slot ACQ_RB0 (as k_front_fft's FF_RB0)

template <int LOGN>
__global__ __launch_bounds__((1 << LOGN) / 16, 2) void k_acq_inv(AcqArgs a)
{
    constexpr int N = 1 << LOGN, T = N / 16;
    using Plan = AcqPlan<LOGN>;
    constexpr int GL = Plan::GL, ML = 1 << GL, NGL = 16 >> GL, HL = N >> GL;
    constexpr int JB = (N / 4 + 1 + T - 1) / T;  // RxDownSample outputs per thread and frame (decimation >= 4)
    constexpr int NB = (204 + T - 1) / T;        // gathered bins a thread brings in
    extern __shared__ __align__(16) unsigned char smem[];
    double2 *X = reinterpret_cast<double2 *>(smem);  // [N] the image; afterwards the compact real samples
    double2 *TsL = X + N;                            // [ACQ_TWL]
    double2 *B = TsL + ACQ_TWL;                      // [204] the gathered bins (everything behind them is zero, :414-415)
    long long *clkL = reinterpret_cast<long long *>(B + 204);  // [8]
    int *negzL = reinterpret_cast<int *>(clkL + 8);            // [2] a gathered bin of the frame holds a -0.0 (by frame parity)
    const bool timing = a.clk != nullptr && blockIdx.x == 0 && threadIdx.x == 0;
    long long tprev = 0;
    if (timing) {
        for (int k = 0; k < 8; k++) clkL[k] = 0;
        tprev = (long long)clock64();
    }
    const double2 *__restrict__ tsg = a.tw;
    const int tid = threadIdx.x;
    for (int i = tid; i < ACQ_TWL - 1; i += T) TsL[i] = tsg[i];
    if (tid < 2) negzL[tid] = 0;
    const int D = a.decim;
    const double norm = 1.0 / (double)N;
    const double HOWARD = 0.9 * 32768.0;
    const long long nfr = (long long)a.S * a.F;
    const int lo1 = a.do_up ? N / 4 - 26 : 0;
    // in registers for the whole launch: the wing-8 twiddle of this thread's slot of a first-pass group, negated for the slots
    // 8 .. 15 (v0 - w v8 = v0 + (-w) v8, product for product)
    double wr8, wi8;
    {
        const double2 wq = tsg[7 + (tid & 7)];
        wr8 = (tid & 8) ? -wq.x : wq.x;
        wi8 = (tid & 8) ? wq.y : -wq.y;  // (the inverse conjugates: wi = -wq.y)
    }
    // where the frame's 204 bins sit in phase A's row: rows hold bins [0, n/4 + 28) (lower band) or [0, 204) + [n/4 - 26, n/2 + 28)
    // (upper band): every centre bin the rule can produce (102, or binPos + 1 with binPos in [beg + 75, end - 75)) gathers inside them
    auto row_off = [&](int c) {
        int off = c - 102;
        if (a.do_up) off = (c == 102) ? 0 : 204 + (c - 102 - lo1);
        if (off < 0) off = 0;
        if (off + 204 > a.nsb) off = a.nsb - 204;
        return off;
    };
    // one frame ahead: the centre bin, then the bins
    double2 bins[NB];
    int c_next = 0;
    auto fetch_bins = [&](long long g, int c) {
        const double2 *src = a.spec + g * a.nsb + row_off(c);
#pragma unroll
        for (int i = 0; i < NB; i++) {
            const int k = tid + T * i;
            bins[i] = k < 204 ? src[k] : make_double2(0.0, 0.0);
        }
    };
    // frames in runs of a.run, a ticket each (see k_acq_fwd); the next run's ticket is known from the run's first frame on, the next
    // frame's centre bin is requested at the top of a frame (a run has at least two frames)
    const int K = a.run;
    const int nruns = a.S * a.rps;
    (void)nfr;
    int *tkL = negzL + 2;  // [2]
    if (tid == 0) tkL[0] = (int)atomicAdd(a.tickets + 1, 1u);
    acq_barrier<T>();
    int r_next = tkL[0];
    int s = 0, f = 0, fe = 0, sn = 0, fn = 0, fen = 0;
    bool first = true;
    auto run_of = [&](int r, int &s_, int &f_, int &fe_) {
        s_ = (int)((unsigned)r / (unsigned)a.rps);
        f_ = (r - s_ * a.rps) * K;
        fe_ = f_ + K < a.F ? f_ + K : a.F;
    };
    bool have = r_next < nruns;
    // a frame's bins reach LDS during the frame BEFORE it (requested before that frame's last pass, stored behind its compact samples:
    // the image of bins is dead once the first passes have read it), so that a frame starts without a wait for memory and without
    // a barrier of its own; the launch's first frame is the exception
    auto store_bins = [&](int tf_, int slot) {
        bool nz = false;
        const long long NEGZ = (long long)0x8000000000000000ull;
#pragma unroll
        for (int i = 0; i < NB; i++) {
            const int k = tf_ + T * i;
            if (k < 204) {
                B[k] = bins[i];
                nz = nz || __double_as_longlong(bins[i].x) == NEGZ || __double_as_longlong(bins[i].y) == NEGZ;
            }
        }
        if (nz) negzL[slot] = 1;
    };
    if (have) {
        run_of(r_next, s, f, fe);
        const long long g0_ = (long long)s * a.F + f;
        fetch_bins(g0_, a.cbin[g0_]);
        store_bins(tid, 0);
    }
    acq_barrier<T>();
    r_next = nruns;  // (unknown until the run's first frame has fetched it)
    int par = 0;
    for (; have; par ^= 1) {
        unsigned tk = 0;
        if (first && tid == 0) tk = atomicAdd(a.tickets + 1, 1u);
        int tf = tid;  // opaque per frame (see k_acq_fwd)
        asm volatile("" : "+v"(tf));
        const long long g = (long long)s * a.F + f;
        // (a launch's samples number less than 2^31 -- launch_acq3 sees to it: 32-bit window arithmetic, one 32-bit division a frame)
        const int t0 = (a.f0 + f) * N;  // call-relative index of the frame's first sample
        // ---- the 204 bins around the centre at bin 0 of a zeroed array (:458): in LDS since the frame before
        // the VCO factors of this frame's outputs and the next frame's centre bin: in flight during the transform
        const int fo = (int)a.first_out;
        const int jlo = t0 <= fo ? 0 : (int)((unsigned)(t0 - fo + D - 1) / (unsigned)D);
        const int nds = (int)a.nds;
        double2 cs[JB];
#pragma unroll
        for (int b = 0; b < JB; b++) {
            const int j = jlo + tf + T * b;
            const int te = fo + D * j;
            cs[b] = (te < t0 + N && j < nds) ? a.vco_cs[j] : make_double2(0.0, 0.0);
        }
        // (unconditional requests: see k_acq_fwd; in a run's first frame the next frame is the run's second -- a run has two frames
        //  or is the stream's last -- so the next run's ticket, taken in the first frame, is known whenever it is needed)
        bool more = true;
        if (f + 1 < fe) {
            sn = s;
            fn = f + 1;
            fen = fe;
        } else if (!first && r_next < nruns) {
            run_of(r_next, sn, fn, fen);
        } else {
            sn = s;
            fn = f;
            fen = fe;
            more = false;
        }
        const long long gn = (long long)sn * a.F + fn;
        // (a SCALAR load: the value is uniform, and as a vector load the compiler makes it so on the spot -- global_load, s_waitcnt
        //  vmcnt(0), v_readfirstlane: a round trip to memory, and the previous frame's stores, at the top of every frame)
        c_next = ((const __attribute__((address_space(4))) int *)a.cbin)[gn];
        ACQ_PHASE(0)
        // ---- inverse transform (:459).  Slot 16 q + m of the bit-reversed array holds input brev4(m) N/16 + brev(q): below 204 only
        // for m = 0 (N >= 2048; and m = 8 at N = 2048), so the first three stages of a group are broadcasts of its slots 0 and 8
        // -- a butterfly whose second operand is +0 returns its first operand twice, unless that holds a -0.0 (IEEE: (-0) + (+0) =
        // +0): a frame with such a bin takes the four stages in full -- and stage four is out[j'] = v0 + w v8, out[j' + 8] = v0 - w v8.
        // The second pass (wings 16 .. 128) needs slot j of sixteen groups: it forms them itself from the bins, no LDS round trip.
        if constexpr (LOGN >= 11) {
            const int j = tf & 15, hi = tf >> 4;
            double2 x[16];
            if (negzL[par] == 0) {
#pragma unroll
                for (int m = 0; m < 16; m++) {
                    const int k0 = acq_brev<LOGN - 4>(16 * hi + m);
                    const double2 v0 = k0 < 204 ? B[k0] : make_double2(0.0, 0.0);
                    x[m] = v0;
                    // (a structurally zero v8 leaves v0 as it is: v0 + (+-0) = v0 for every v0 that is not -0.0)
                    if (LOGN == 11 && acq_brev<4>(m) * 8 < 76) {  // compile time: 128 + k0 < 204 is possible
                        if (N / 16 + k0 < 204) {
                            const double2 v8 = B[N / 16 + k0];
                            const double p1 = wr8 * v8.x, p2 = wi8 * v8.y, p3 = wr8 * v8.y, p4 = wi8 * v8.x;
                            const double tr = p1 - p2;
                            const double ti = p3 + p4;
                            x[m] = make_double2(v0.x + tr, v0.y + ti);
                        }
                    }
                }
            } else {
                // a bin with a -0.0 component: every group in full (one body, not sixteen)
#pragma unroll 1
                for (int m = 0; m < 16; m++) {
                    const int k0 = acq_brev<LOGN - 4>(16 * hi + m);
                    double2 t[16];
#pragma unroll
                    for (int i = 0; i < 16; i++) t[i] = make_double2(0.0, 0.0);
                    if (k0 < 204) t[0] = B[k0];
                    if (N / 16 + k0 < 204) t[8] = B[N / 16 + k0];
                    acq_stages<4, 1, true, true>(t, 0, TsL, tsg);
                    double2 r = t[0];
#pragma unroll
                    for (int i = 1; i < 16; i++)
                        if (j == i) r = t[i];
#pragma unroll
                    for (int i = 0; i < 16; i++)
                        if (i == m) x[i] = r;
                }
            }
            acq_stages<4, 16, true, false>(x, j, TsL, tsg);
            const int e0 = (hi << 8) + j;
#pragma unroll
            for (int m = 0; m < 16; m++) X[acq_slot_hm(e0, 16 * m)] = x[m];
        } else {
            // N = 1024: four inputs of a first-pass group can be non-zero; the first pass runs in full on registers
            const int q = tf;  // (natural order: the bins come from LDS)
            double2 v[16];
#pragma unroll
            for (int m = 0; m < 16; m++) {
                const int k = acq_brev<4>(m) * (N / 16) + acq_brev<LOGN - 4>(q);
                v[m] = k < 204 ? B[k] : make_double2(0.0, 0.0);
            }
            acq_stages<4, 1, true, true>(v, 0, TsL, tsg);
            const int key = acq_key(q);
#pragma unroll
            for (int m = 0; m < 16; m++) X[16 * q + (m ^ key)] = v[m];
            acq_barrier<T>();
            acq_mid_pass<4, 16, true, LOGN>(X, TsL, tsg, tf);
        }
        {
            // (pinned HERE by an opaque zero: left alone the compiler moves this store up beside the atomic -- same condition -- and
            //  waits for the ticket's round trip at the top of the frame; the wave's partner then waits for it at the next barrier)
            int zlate = 0;
            asm volatile("" : "+v"(zlate));
            if (first && tid == 0) tkL[1] = (int)tk + zlate;
        }
        acq_barrier<T>();
        ACQ_PHASE(1)
        if (tf == 0) negzL[par] = 0;  // (read by everybody before the barrier above; set again two frames on at the earliest)
        if constexpr (Plan::G4 != 0) {
            acq_mid_pass<Plan::G3, 256, true, LOGN>(X, TsL, tsg, tf);
            acq_barrier<T>();
        }
        // the last pass's twiddles first, the next frame's bins (its centre bin arrived during the passes above) behind them: in-order
        // returns, see k_acq_fwd (a pass earlier they cost the fused pass its registers)
        double2 twl[NGL][ML - 1];
#pragma unroll
        for (int it = 0; it < NGL; it++) acq_load_tw<GL, HL>(twl[it], tf + T * it, tsg);
        __builtin_amdgcn_sched_barrier(0);
        fetch_bins(gn, c_next);
        __builtin_amdgcn_sched_barrier(0);
        // ---- last pass: RxDownSample reads nothing but re / n (:461-463) -- the imaginary halves of the last stage have no reader
        double o[NGL][ML];
#pragma unroll
        for (int it = 0; it < NGL; it++) {
            const int j = tf + T * it;
            double2 v[ML];
#pragma unroll
            for (int m = 0; m < ML; m++) v[m] = X[acq_slot_hm(j, HL * m)];
            acq_stages_w<GL, 0, GL, true>(v, twl[it]);
#pragma unroll
            for (int m = 0; m < ML; m++) o[it][m] = v[m].x * norm;
        }
        acq_barrier<T>();  // every butterfly of the pass is in registers: the compact samples go over the image
        ACQ_PHASE(2)
        double *Rb = reinterpret_cast<double *>(smem);
#pragma unroll
        for (int it = 0; it < NGL; it++) {
            const int j = tf + T * it;
#pragma unroll
            for (int m = 0; m < ML; m++) Rb[ACQ_RB0 + j + HL * m] = o[it][m];
        }
        store_bins(tf, par ^ 1);  // the NEXT frame's bins (requested before the last pass; nothing is stored to memory in between)
        acq_barrier<T>();
        ACQ_PHASE(3)
        // ---- the frame's first and last 26 samples for the windows that cross into / out of it (k_acq_edges)
        if (tf < 26) {
            double *eg = a.edges + g * 52;
            eg[tf] = Rb[ACQ_RB0 + tf];
            eg[26 + tf] = Rb[ACQ_RB0 + N - 26 + tf];
        }
        // ---- RxDownSample(re, re) (:461-463, :470-492) for the outputs whose 27-sample window lies inside this frame
        {
            const bool even_d = (D & 1) == 0;  // then every window of the call ends on the same parity (N is even)
            const int wpar = (int)((a.first_out - t0) & 1);
#pragma unroll
            for (int b = 0; b < JB; b++) {
                const int j = jlo + tf + T * b;
                const int te = fo + D * j;  // window end, call-relative
                if (te < t0 + N && j < nds) {
                    const int e = te - t0;
                    if (e >= 26) {
                        double fi = 0.0;
                        if (even_d) {
                            // the 27 samples e-26 .. e as 14 aligned 16-byte reads
                            const double2 *w2 = reinterpret_cast<const double2 *>(Rb + ((e + ACQ_RB0 - 26) & ~1));
                            double d[28];
#pragma unroll
                            for (int i = 0; i < 14; i++) {
                                const double2 t = w2[i];
                                d[2 * i] = t.x;
                                d[2 * i + 1] = t.y;
                            }
                            if (wpar) {
#pragma unroll
                                for (int k = 0; k < 27; k++) fi += d[27 - k] * ds_tap(k);  // newest first (:479-483)
                            } else {
#pragma unroll
                                for (int k = 0; k < 27; k++) fi += d[26 - k] * ds_tap(k);
                            }
                        } else {
                            const double *w = Rb + (ACQ_RB0 + e);
#pragma unroll
                            for (int k = 0; k < 27; k++) fi += w[-k] * ds_tap(k);
                        }
                        const double ov = fi * HOWARD;  // fi == fq: both rails get the same samples
                        a.dm[(long long)s * a.dm_stride + 64 + j] = make_double2(ov * cs[b].x, ov * cs[b].y);  // :515-516
                    }
                }
            }
        }
        acq_barrier<T>();  // every window is read before the next frame's passes store over the image
        ACQ_PHASE(4)
        r_next = tkL[1];
        if (!more && first && f + 1 >= fe && r_next < nruns) {
            // a run of ONE frame (a stream's last, F not a multiple of the run): its successor was not known at its top -- it is now
            run_of(r_next, sn, fn, fen);
            more = true;
            const long long gq = (long long)sn * a.F + fn;
            fetch_bins(gq, a.cbin[gq]);  // (the bins requested for "this frame again" are replaced; stored before the next frame starts)
            store_bins(tf, par ^ 1);
            acq_barrier<T>();
        }
        first = fn != f + 1 || sn != s;
        have = more;
        s = sn;
        f = fn;
        fe = fen;
    }
    if (timing)
        for (int k = 0; k < 8; k++) a.clk[8 + k] = clkL[k];
}

// ============================================================================================================= edges
// One workgroup per stream.  The windows of a frame that begin in the frame before it: at most ceil(26 / D) per frame.  The
// call's first frame reaches into the stream's history (FftFrontState::hist: the last 26 samples of the previous call, zeros
// before the first), which this kernel also brings up to date -- after every thread has read it.
__global__ __launch_bounds__(256) void k_acq_edges(AcqArgs a)
{
    const int s = blockIdx.x, tid = threadIdx.x;
    const int n = a.n, D = a.decim;
    const int EO = 26 / D + 2;   // upper bound of the windows per frame that end within its first 26 samples
    const int FB = 256 / EO;     // frames a block takes: one thread per (frame, window)
    const double HOWARD = 0.9 * 32768.0;
    FftFrontState *sp = &a.st[s];
    // rows[i] = the first / last 26 samples of frame fb - 1 + i, i = 0 .. FB: read once, coalesced (a window walks 27 of them, and from
    // global memory that was 27 dependent loads a thread); for the call's first frame the row in front is the stream's history
    __shared__ double rows[(256 / 2 + 1) * 52];
    const int fb = blockIdx.y * FB;
    const long long g0 = (long long)s * a.F;
    const int nrows = (fb + FB < a.F ? FB : a.F - fb) + 1;
    for (int i = tid; i < nrows * 52; i += 256) {
        const int row = i / 52, c = i - row * 52;
        const int f = fb - 1 + row;
        rows[i] = f >= 0 ? a.edges[(g0 + f) * 52 + c] : (c >= 26 ? sp->hist[c - 26] : 0.0);
    }
    __syncthreads();
    {
        const int fi = tid / EO, r = tid - fi * EO;
        const int f = fb + fi;
        if (fi < FB && f < a.F) {
            const int fo = (int)a.first_out;
            const int t0 = (a.f0 + f) * n;
            const int jlo = t0 <= fo ? 0 : (int)((unsigned)(t0 - fo + D - 1) / (unsigned)D);
            const int j = jlo + r;
            const int te = fo + D * j;
            const int e = te - t0;
            if (te < t0 + n && j < (int)a.nds && e >= 0 && e < 26) {
                // sample e - k of the frame, k = 0 .. 26, newest first (:479-483): from the frame's own first 26 samples, or from the 26
                // before it (the previous frame's last ones; the stream's history for the call's first frame)
                const double *head = rows + (fi + 1) * 52;
                const double *prev = rows + fi * 52 + 26;
                double fiv = 0.0;
#pragma unroll
                for (int k = 0; k < 27; k++) {
                    const int i = e - k;
                    const double x = i >= 0 ? head[i] : prev[26 + i];
                    fiv += x * ds_tap(k);
                }
                const double ov = fiv * HOWARD;
                const double2 cs = a.vco_cs[j];
                a.dm[(long long)s * a.dm_stride + 64 + j] = make_double2(ov * cs.x, ov * cs.y);
            }
        }
    }
    // the history the call leaves: by the stream's first block, which is the only one that read the old one (before the barrier above)
    if (blockIdx.y == 0 && tid < 26) sp->hist[tid] = a.edges[(g0 + a.F - 1) * 52 + 26 + tid];
}


// ============================================================================================================= host
extern int g_acq_last_grid[4];
// bytes of scratch one frame needs between the phases (spec row, boxcar band, peak, centre bin, edges), 16-byte aligned parts
static void acq3_layout(int n, int do_up, int *nsb, int *na)
{
    // (Java's integer n / 4 and n / 2, as the reference computes its band, :429-430)
    const int beg = do_up ? n / 4 : 0, end = do_up ? n / 2 : n / 4;
    *nsb = do_up ? 204 + (n / 2 + 28 - (n / 4 - 26)) : n / 4 + 28;
    if (*nsb < 204) *nsb = 204;  // (frames below 704 samples, any-frame path: the gather at the clamp value 102 takes bins [0, 204))
    *na = ((end - beg - 150) + 1) & ~1;
    if (*na < 2) *na = 2;
}

bool acq3_supported(int n)
{
    return n == 1024 || n == 2048 || n == 4096 || n == 8192 || acqm_supported(n);
}

size_t acq3_frame_bytes(int n, int do_up)
{
    int nsb, na;
    acq3_layout(n, do_up, &nsb, &na);
    return sizeof(double2) * (size_t)nsb + sizeof(double) * (size_t)na + sizeof(AcqPeak) + 16 + sizeof(double) * 52;
}

static int acq_edge_blocks(const AcqArgs &a)
{
    const int fb = 256 / (26 / a.decim + 2);  // frames a block of k_acq_edges takes
    return (a.F + fb - 1) / fb;
}

template <int LOGN>
static int launch_acq3_t(AcqArgs &a, bool f32, int num_cu, hipStream_t st, const AcqProf &prof)
{
    constexpr int N = 1 << LOGN, T = N / 16;
    constexpr size_t lds_fwd = sizeof(double2) * ((size_t)N + ACQ_TWL) + 8 * sizeof(double) + 8 * sizeof(int) + 8 * sizeof(long long) + 16;
    constexpr size_t lds_inv = sizeof(double2) * ((size_t)N + ACQ_TWL + 204) + 8 * sizeof(long long) + 32;
    const long long nfr = (long long)a.S * a.F;
    int per_cu_f = 1, per_cu_i = 1;
    if (f32) {
        JSDR_LDS_ATTR((k_acq_fwd<LOGN, true>), lds_fwd);
        JSDR_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_f, k_acq_fwd<LOGN, true>, T, lds_fwd));
    } else {
        JSDR_LDS_ATTR((k_acq_fwd<LOGN, false>), lds_fwd);
        JSDR_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_f, k_acq_fwd<LOGN, false>, T, lds_fwd));
    }
    JSDR_LDS_ATTR((k_acq_inv<LOGN>), lds_inv);
    JSDR_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_i, k_acq_inv<LOGN>, T, lds_inv));
    if (per_cu_f < 1) per_cu_f = 1;
    if (per_cu_i < 1) per_cu_i = 1;
#ifdef JSDR_X_ACQ_WGS  // (timing probe: fewer workgroups a CU)
    if (per_cu_f > JSDR_X_ACQ_WGS) per_cu_f = JSDR_X_ACQ_WGS;
    if (per_cu_i > JSDR_X_ACQ_WGS) per_cu_i = JSDR_X_ACQ_WGS;
#endif
    long long gf = (long long)per_cu_f * num_cu, gi = (long long)per_cu_i * num_cu;
    if (gf > nfr) gf = nfr;
    if (gi > nfr) gi = nfr;
    g_acq_last_grid[0] = (int)gf;
    g_acq_last_grid[1] = (int)gi;
    g_acq_last_grid[2] = per_cu_f;
    g_acq_last_grid[3] = per_cu_i;
    auto mark = [&](int phase, bool begin) {
        if (prof.mark) prof.mark(prof.ctx, phase, begin, st);
    };
    JSDR_HIP_TRY(hipMemsetAsync(a.tickets, 0, 2 * sizeof(unsigned), st));  // the run counters of the two frame-parallel kernels
    a.rps = (a.F + a.run - 1) / a.run;
    JSDR_REQUIRE((long long)a.S * a.rps < 0x7fffffffLL && (long long)(a.f0 + a.F) * a.n < 0x7fffffffLL,
                 "bpsk: a three-phase FFT-acquire launch of %d streams x %d frames is beyond its 32-bit frame arithmetic", a.S, a.F);
    mark(0, true);
    if (f32)
        hipLaunchKernelGGL((k_acq_fwd<LOGN, true>), dim3((unsigned)gf), dim3(T), lds_fwd, st, a);
    else
        hipLaunchKernelGGL((k_acq_fwd<LOGN, false>), dim3((unsigned)gf), dim3(T), lds_fwd, st, a);
    mark(0, false);
    JSDR_LAUNCH_CHECK();
    mark(1, true);
    hipLaunchKernelGGL(k_acq_scan, dim3((unsigned)a.S), dim3(64), 0, st, a);
    mark(1, false);
    JSDR_LAUNCH_CHECK();
    mark(2, true);
    hipLaunchKernelGGL((k_acq_inv<LOGN>), dim3((unsigned)gi), dim3(T), lds_inv, st, a);
    mark(2, false);
    JSDR_LAUNCH_CHECK();
    mark(3, true);
    hipLaunchKernelGGL(k_acq_edges, dim3((unsigned)a.S, (unsigned)acq_edge_blocks(a)), dim3(256), 0, st, a);
    mark(3, false);
    JSDR_LAUNCH_CHECK();
    return JSDR_OK;
}

int g_acq_last_grid[4] = {0, 0, 0, 0};

// The call's frames in launches of at most chunk_frames per stream (the scratch holds S * chunk_frames frames)
// the default mixed-radix frames: k_acqm_fwd / k_acqm_inv (bpsk_fftm.hip) around the same scan and edges
static int launch_acq3_m(AcqArgs &a, const FftFrontArgs &fa, const AcqmPlan &plan, int num_cu, hipStream_t st, const AcqProf &prof)
{
    auto mark = [&](int phase, bool begin) {
        if (prof.mark) prof.mark(prof.ctx, phase, begin, st);
    };
    JSDR_HIP_TRY(hipMemsetAsync(a.tickets, 0, 2 * sizeof(unsigned), st));
    mark(0, true);
    if (launch_acqm(a, fa, plan.np, plan.rad, plan.tw_off, plan.wr_off, num_cu, 0, st) != JSDR_OK) return JSDR_ERR;
    mark(0, false);
    mark(1, true);
    hipLaunchKernelGGL(k_acq_scan, dim3((unsigned)a.S), dim3(64), 0, st, a);
    mark(1, false);
    JSDR_LAUNCH_CHECK();
    mark(2, true);
    if (launch_acqm(a, fa, plan.np, plan.rad, plan.tw_off, plan.wr_off, num_cu, 1, st) != JSDR_OK) return JSDR_ERR;
    mark(2, false);
    mark(3, true);
    hipLaunchKernelGGL(k_acq_edges, dim3((unsigned)a.S, (unsigned)acq_edge_blocks(a)), dim3(256), 0, st, a);
    mark(3, false);
    JSDR_LAUNCH_CHECK();
    return JSDR_OK;
}

// any other frame: the passes of bpsk_acqg.hip around the same scan and edges
static int launch_acq3_g(AcqArgs &a, const AcqgPlan &gen, double2 *img, hipStream_t st, const AcqProf &prof)
{
    auto mark = [&](int phase, bool begin) {
        if (prof.mark) prof.mark(prof.ctx, phase, begin, st);
    };
    JSDR_REQUIRE((long long)(a.f0 + a.F) * a.n < 0x7fffffffLL, "bpsk: an FFT-acquire call of %d frames of %d samples is beyond k_acq_edges' 32-bit sample index",
                 a.f0 + a.F, a.n);
    mark(0, true);
    if (launch_acqg(a, gen, img, 0, st) != JSDR_OK) return JSDR_ERR;
    mark(0, false);
    mark(1, true);
    hipLaunchKernelGGL(k_acq_scan, dim3((unsigned)a.S), dim3(64), 0, st, a);
    mark(1, false);
    JSDR_LAUNCH_CHECK();
    mark(2, true);
    if (launch_acqg(a, gen, img, 1, st) != JSDR_OK) return JSDR_ERR;
    mark(2, false);
    mark(3, true);
    hipLaunchKernelGGL(k_acq_edges, dim3((unsigned)a.S, (unsigned)acq_edge_blocks(a)), dim3(256), 0, st, a);
    mark(3, false);
    JSDR_LAUNCH_CHECK();
    return JSDR_OK;
}

int launch_acq3(const FftFrontArgs &fa, int nstreams, unsigned char *scratch, size_t scratch_bytes, int chunk_frames, int num_cu,
                hipStream_t st, const AcqProf &prof, const AcqmPlan &plan, const AcqgPlan *gen)
{
    const bool generic = gen != nullptr && gen->on;
    JSDR_REQUIRE(generic || acq3_supported(fa.n), "bpsk: the three-phase FFT-acquire front end takes frames of 1024 .. 8192 samples (2^k) or 9600 / 4800 / 4410, not %d", fa.n);
    int nsb, na;
    acq3_layout(fa.n, fa.do_up, &nsb, &na);
    if (chunk_frames < 1) chunk_frames = 1;
    const size_t nf = (size_t)nstreams * (size_t)chunk_frames;
    JSDR_REQUIRE(nf * (acq3_frame_bytes(fa.n, fa.do_up) + (generic ? acqg_image_bytes(fa.n) : 0)) + 512 <= scratch_bytes,
                 "bpsk: FFT-acquire scratch too small (%zu frames)", nf);
    AcqArgs a;
    a.raw = fa.raw;
    a.rawf = fa.rawf;
    a.stride_pairs = fa.stride_pairs;
    a.ic = fa.ic;
    a.qc = fa.qc;
    a.S = nstreams;
    a.n = fa.n;
    a.do_up = fa.do_up;
    a.decim = fa.decim;
    a.first_out = fa.first_out;
    a.nds = fa.nds;
    a.vco_cs = fa.vco_cs;
    a.tw = fa.tw;
    a.st = fa.st;
    a.dm = fa.dm;
    a.dm_stride = fa.dm_stride;
    a.nsb = nsb;
    a.na = na;
    unsigned char *p = scratch;
    a.spec = reinterpret_cast<double2 *>(p);
    p += sizeof(double2) * nf * (size_t)nsb;
    a.aband = reinterpret_cast<double *>(p);
    p += sizeof(double) * nf * (size_t)na;
    a.peak = reinterpret_cast<AcqPeak *>(p);
    p += sizeof(AcqPeak) * nf;
    a.edges = reinterpret_cast<double *>(p);
    p += sizeof(double) * 52 * nf;
    a.cbin = reinterpret_cast<int *>(p);
    p += ((sizeof(int) * nf + 63) & ~(size_t)63);
    a.tickets = reinterpret_cast<unsigned *>(p);
    p += 64;
    double2 *img = reinterpret_cast<double2 *>(scratch + (((size_t)(p - scratch) + 255) & ~(size_t)255));  // (any-frame path only)
    a.run = 4;
    if (const char *e = knob("JSDR_ACQ_RUN")) a.run = atoi(e) >= 2 ? atoi(e) : 4;
    a.nwg = 0;
    a.clk = fa.phase_clk;
    for (int f0 = 0; f0 < fa.nframes; f0 += chunk_frames) {
        a.f0 = f0;
        a.F = fa.nframes - f0 < chunk_frames ? fa.nframes - f0 : chunk_frames;
        int rc;
        if (generic) {
            rc = launch_acq3_g(a, *gen, img, st, prof);
            if (rc != JSDR_OK) return rc;
            continue;
        }
        if (acqm_supported(fa.n)) {
            rc = launch_acq3_m(a, fa, plan, num_cu, st, prof);
            if (rc != JSDR_OK) return rc;
            continue;
        }
        switch (fa.logn) {
            case 10: rc = launch_acq3_t<10>(a, fa.rawf != nullptr, num_cu, st, prof); break;
            case 11: rc = launch_acq3_t<11>(a, fa.rawf != nullptr, num_cu, st, prof); break;
            case 12: rc = launch_acq3_t<12>(a, fa.rawf != nullptr, num_cu, st, prof); break;
            default: rc = launch_acq3_t<13>(a, fa.rawf != nullptr, num_cu, st, prof); break;
        }
        if (rc != JSDR_OK) return rc;
    }
    return JSDR_OK;
}

}  // namespace jsdr
