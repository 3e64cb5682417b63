// bpsk_fftm.hip -- FUNcubeBPSKDemod FFT-acquire front end (doBufferFFT, FUNcubeBPSKDemod.java:406-464) for frames
// that are NOT a power of two: n = 2^a 3^b 5^c up to 9600, i.e. the reference's own default frame
// (blen = rate*size/10 => n = 9600 at 96 kHz, 4800 at 48 kHz; JavaAudio.java:58-59).
//
// Compiled with -ffp-contract=off.  The transform is the oracle's mixed-radix definition (oracle/o_fft.c,
// fft_f64_mixed): Stockham autosort passes with radices 4,..,(2),3..,5.. in this order, the same per-pass twiddle
// tables (long double + one rounding, exact on the axes), the same fixed-order 2/3/4/5-point butterflies, the
// inverse as conj o forward o conj -- so centre bins, traces, bits and FEC bytes are bit-identical to the
// oracle.  JTransforms' own rounding is unknowable (source absent): parity with the Java library itself is unpinned.
//
// MI355X mapping: one 512-thread workgroup per stream, persistent over the frames of the call (the centre-bin
// state is sequential).  The frame lives in LDS as double2[n] (153.6 KB at n = 9600: one workgroup per CU); a
// pass loads every butterfly into registers (<= 24 double2 per thread; 1024 threads would cap a thread at 128
// VGPRs and spill), barrier, stores to the autosort positions -- in place, no second image.  |X| and the boxcar sums of the searched quarter band reuse the dead
// upper part of the image.  The rest (boxcar, first maximum, centre-bin rule, 204 bins to bin 0, RxDownSample,
// VCO) is k_front_fft's, with run-time sizes.
#include "bpsk_fft.h"
#include "bpsk_radix.h"
#include <math.h>

namespace jsdr {

// timing probe only (never in the product build): the passes without their workgroup barriers -- wrong data, unchanged
// addresses -- to see what lock-step at the barriers costs
#ifdef JSDR_X_NOBAR
#define FM_PASS_SYNC() __builtin_amdgcn_wave_barrier()
#else
#define FM_PASS_SYNC() __syncthreads()
#endif

#ifndef JSDR_FM_T
#define JSDR_FM_T 768
#endif
constexpr int FM_T = JSDR_FM_T;  // (probe builds: -DJSDR_FM_T=1024)
constexpr int FM_NMAX = 9600;
constexpr int FM_MAXPASS = 12;

struct FftmArgs {
    FftFrontArgs f;  // n, raw, state, dm, ... (logn unused; f.tw = concatenated per-pass tables)
    int np;
    int rad[FM_MAXPASS];
    int tw_off[FM_MAXPASS];  // offset of pass p's table T[m] = exp(-2 pi i m/(P r)), m < P r
    unsigned pmagic[FM_MAXPASS];  // b / P == (b * pmagic) >> 32 for b < 2^16 (P = product of the earlier radices)
    int lds_tw;              // the tables of the first passes, lds_tw entries in all, are copied to LDS
    // round 5: a prime radix above 7 (fm_pass_generic): wr_off[p] = offset of W[m] = exp(-2 pi i m/r), m < r; the pass is
    // out of place through a per-stream scratch of n elements in global memory
    int wr_off[FM_MAXPASS];
    double2 *gscratch;
    long long gscratch_stride;
};

// Typed views of the LDS image and of the twiddle tables.  The passes are separate (noinline) functions: through a plain
// `double2 *` parameter the compiler has to treat every access as a FLAT one -- a 64-bit address computation per element,
// a null check per pointer conversion, flat loads that tie the LDS and the vector-memory counters together.  With the
// address space in the type an access is a ds_read / ds_write on a 32-bit offset with an immediate, or a global_load.
typedef double d2v __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) d2v lds_d2v;
typedef __attribute__((address_space(3))) double lds_f64;
typedef __attribute__((address_space(1))) const d2v gbl_d2v;
typedef __attribute__((address_space(1))) const int gbl_i32;
typedef __attribute__((address_space(1))) const f2v gbl_f2v;

struct LdsRef {
    lds_d2v *q;
    __device__ __forceinline__ operator double2() const
    {
        const d2v v = *q;
        return make_double2(v.x, v.y);
    }
    __device__ __forceinline__ void operator=(const double2 &v) const
    {
        d2v t;
        t.x = v.x;
        t.y = v.y;
        *q = t;
    }
    __device__ __forceinline__ void put_x(double x) const { *reinterpret_cast<lds_f64 *>(q) = x; }
};
struct LdsArr {
    lds_d2v *p;
    __device__ __forceinline__ LdsRef operator[](int i) const { return LdsRef{p + i}; }
    __device__ __forceinline__ LdsArr operator+(int i) const { return LdsArr{p + i}; }
};
struct GblArr {
    gbl_d2v *p;
    __device__ __forceinline__ double2 operator[](int i) const
    {
        const d2v v = p[i];
        return make_double2(v.x, v.y);
    }
    __device__ __forceinline__ GblArr operator+(int i) const { return GblArr{p + i}; }
};
// (the low half of a flat address inside the LDS aperture is the LDS offset)
__device__ __forceinline__ LdsArr lds_arr(const void *generic) { return LdsArr{(lds_d2v *)(unsigned)(unsigned long long)generic}; }
__device__ __forceinline__ GblArr gbl_arr(const double2 *g) { return GblArr{(gbl_d2v *)(unsigned long long)g}; }

// one Stockham pass, in place: every butterfly of the pass is in registers before the first store
// NN / PP: frame size and stride as compile-time constants for the two default frames (9600, 4800): every LDS offset
// becomes an immediate, `b mod P` a mask or a constant multiply-high, and the butterfly count per thread is the
// frame's, not the largest frame's.  0 = run-time values (any other supported n).
template <int R, int NN = 0, int PP = 0, class TW = const double2 *, int NIMG = 1>
__device__ __attribute__((noinline)) void fm_pass(LdsArr X0, TW tw, int n_rt, int P_rt, unsigned pmagic, int tid)
{
    static_assert(NIMG == 1 || NN != 0, "several images: compile-time frame size");
    constexpr int ITERS = ((NIMG * ((NN ? NN : FM_NMAX) / R)) + FM_T - 1) / FM_T;
    const int n = NN ? NN : n_rt;
    const int P = PP ? PP : P_rt;
    const int nb = n / R;
    double2 v[ITERS][R];
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int bb = it * FM_T + tid;
        const int img = (NIMG > 1 && bb >= nb) ? 1 : 0;
        const int b = bb - img * nb;
        const LdsArr X = X0 + img * NN;
        if (bb < NIMG * nb) {
            const int k = PP ? (b % PP) : (P == 1) ? 0 : b - (int)__umulhi((unsigned)b, pmagic) * P;  // b % P without the division sequence
#pragma unroll
            for (int j = 0; j < R; j++) {
                v[it][j] = X[b + j * nb];
                if (j >= 1 && P > 1) v[it][j] = cdmul(v[it][j], tw[k * j]);
            }
            dft_r<R>(v[it]);
        }
    }
    FM_PASS_SYNC();
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int bb = it * FM_T + tid;
        const int img = (NIMG > 1 && bb >= nb) ? 1 : 0;
        const int b = bb - img * nb;
        const LdsArr X = X0 + img * NN;
        if (bb < NIMG * nb) {
            const int k = PP ? (b % PP) : (P == 1) ? 0 : b - (int)__umulhi((unsigned)b, pmagic) * P;  // (2^32/1 does not fit the magic)
            const int j0 = (b - k) * R + k;
#pragma unroll
            for (int q = 0; q < R; q++) X[j0 + q * P] = v[it][q];
        }
    }
    FM_PASS_SYNC();
}

// The LAST pass of the inverse transform: RxDownSample reads nothing but re/n (FUNcubeBPSKDemod.java:461-463), so only the
// real part of every output is formed (the imaginary halves of the 5-point butterfly -- a third of its operations --
// have no reader and are not computed) and it is stored already scaled: X[i].x = re * (1/n), once per sample instead of
// once per tap that reads it.  Same operands, same operations, same order for everything that IS computed.
// TT: the two-dimensional tables of an odd half (fm_pass_t): tw[(j-1) P + k] instead of tw[k j]
// COMPACT: the scaled real parts leave as a plain array of doubles over the (dead) image -- sample t at double slot
// FM_RB0 + t, the previous frame's last 26 samples (hist) copied in front of them at FM_RB0 - 26 .. FM_RB0 - 1 -- so that a
// RxDownSample window, history included, is ONE contiguous run of 27 doubles: 14 conflict-free 16-byte reads per output
// instead of 27 eight-byte reads at a 160-byte lane stride (16 lanes on the same banks).
constexpr int FM_RB0 = 32;
// COMPACT == 2: the samples go straight to an array in global memory (the even samples of a 2 m frame, k_front_fft2x)
// NIMG (round 6, here and in the passes below): the pass over NIMG images at once -- frames f and f + 1 of a stream side by side in
// LDS, image i at X + i NN -- so that a frame of 4800 samples fills the 768 threads' work items as one of 9600 does (a pass pair of
// ONE 4800-sample frame has 300 items)
template <int NN, int PP, bool TT = false, int COMPACT = 0, class TW = const double2 *, int NIMG = 1>
__device__ __attribute__((noinline)) void fm_pass5_real(LdsArr X, TW tw, double norm, int tid, const double *hist_ = nullptr,
                                                         double *gout = nullptr)
{
    constexpr int R = 5, nb = NN / R, ITERS = (NIMG * nb + FM_T - 1) / FM_T;
    double o[ITERS][R];
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int bb = it * FM_T + tid;
        if (bb < NIMG * nb) {
            const int img = (NIMG > 1 && bb >= nb) ? 1 : 0;
            const int b = bb - img * nb;
            const LdsArr Xi = X + img * NN;
            const int k = b % PP;
            double2 v[R];
#pragma unroll
            for (int j = 0; j < R; j++) {
                v[j] = Xi[b + j * nb];
                if (j >= 1) v[j] = cdmul(v[j], TT ? tw[(j - 1) * PP + k] : tw[k * j]);
            }
            dft_r<R>(v);
#pragma unroll
            for (int q = 0; q < R; q++) o[it][q] = v[q].x * norm;
        }
    }
    FM_PASS_SYNC();
    lds_f64 *Rb0 = reinterpret_cast<lds_f64 *>(X.p);
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int bb = it * FM_T + tid;
        if (bb < NIMG * nb) {
            const int img = (NIMG > 1 && bb >= nb) ? 1 : 0;
            const int b = bb - img * nb;
            const LdsArr Xi = X + img * NN;
            lds_f64 *Rbi = Rb0 + img * (2 * NN);
            const int k = b % PP;
            const int j0 = (b - k) * R + k;
#pragma unroll
            for (int q = 0; q < R; q++) {
                if (COMPACT == 2)
                    gout[j0 + q * PP] = o[it][q];
                else if (COMPACT == 1)
                    Rbi[FM_RB0 + j0 + q * PP] = o[it][q];
                else
                    Xi[j0 + q * PP].put_x(o[it][q]);
            }
        }
    }
    lds_f64 *Rb = Rb0;
    if (COMPACT == 1 && hist_ != nullptr) {
        const lds_f64 *hist = (const lds_f64 *)(unsigned)(unsigned long long)hist_;
        if (tid < 26) Rb[FM_RB0 - 26 + tid] = hist[tid];
    }
    if (COMPACT == 2) __threadfence_block();
    FM_PASS_SYNC();
}

// The LAST pass of the forward transform: of the n bins only those the front end can read are formed -- |X| over
// [beg+24, end-24) (:425-427, the boxcar's reach) and the 204 bins around a centre bin (:458), which the rule and its
// clamps keep in [102, end-1] (:444-453): everything below need_end = end + 102, end = n/4 (lower band) or n/2 (upper).  A
// butterfly whose only needed output is q = 0 forms (x0 + a1) + a2 alone (32 of its 72 operations); all others run in
// full and store what is read.
// need_end: outputs [0, need_end) of THIS transform are read (a half transform of a 2 m frame holds every second bin)
template <int NN, int PP, bool TT = false, class TW = const double2 *, int NIMG = 1>
__device__ __attribute__((noinline)) void fm_pass5_band(LdsArr X, TW tw, int need_end, int tid)
{
    constexpr int R = 5, nb = NN / R, ITERS = (NIMG * nb + FM_T - 1) / FM_T;
    static_assert(PP == nb, "the last pass: one butterfly per k");
    double2 v[ITERS][R];
    unsigned need[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int bb = it * FM_T + tid;
        const int img = (NIMG > 1 && bb >= nb) ? 1 : 0;
        const int b = bb - img * nb;  // == k
        const LdsArr Xi = X + img * NN;
        need[it] = 0u;
        if (bb < NIMG * nb) {
#pragma unroll
            for (int q = 0; q < R; q++) {
                const int bin = b + q * PP;
                if (bin < need_end) need[it] |= 1u << q;
            }
#pragma unroll
            for (int j = 0; j < R; j++) {
                v[it][j] = Xi[b + j * nb];
                if (j >= 1) v[it][j] = cdmul(v[it][j], TT ? tw[(j - 1) * PP + b] : tw[b * j]);
            }
            if (need[it] & ~1u) {
                dft_r<R>(v[it]);
            } else {
                // output 0 alone, as dft_r<5> forms it: (x0 + a1) + a2 with a1 = v1 + v4, a2 = v2 + v3
                const double2 a1 = cdadd(v[it][1], v[it][4]), a2 = cdadd(v[it][2], v[it][3]);
                v[it][0] = make_double2((v[it][0].x + a1.x) + a2.x, (v[it][0].y + a1.y) + a2.y);
            }
        }
    }
    FM_PASS_SYNC();
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int bb = it * FM_T + tid;
        if (bb < NIMG * nb) {
            const int img = (NIMG > 1 && bb >= nb) ? 1 : 0;
            const int b = bb - img * nb;
            const LdsArr Xi = X + img * NN;
#pragma unroll
            for (int q = 0; q < R; q++)
                if (need[it] & (1u << q)) Xi[b + q * PP] = v[it][q];
        }
    }
    FM_PASS_SYNC();
}

// The first THREE passes (4, 4, 4) of the inverse transform of the default frames, straight from the gathered bins.
// After them the image holds C = n/64 blocks of 64: block m = the 64-point transform of z[m + C i], i < 64, of which only
// z[m] (i = 0), z[m + C] (i = 1, if m + C < 204) and z[m + 2C] (i = 2, if m + 2C < 204) are not the zeroed array's.  An
// input that is alone in its butterfly passes through it unchanged (a + 0 = a; 0 * w = 0), so passes 1 and 2 only copy,
// and in pass 3 butterfly k < 16 of block m takes v0 = z[m], v1 = z[m + C] T64[k], v2 = z[m + 2C] T64[2k], v3 = 0:
//     a = v0 + v2, b = v0 - v2, c = d = v1:   out[k] = a + c, out[k+32] = a - c, out[k+16] = b - i d, out[k+48] = b + i d
// -- the operations the three general passes perform on these operands, minus the ones whose second operand is a zero of
// the zeroed array.  (Those return their first operand; only the SIGN of an exact zero can differ from the general
// passes' -- a sum of zeros -- and a zero's sign reaches nothing RxDownSample computes: its accumulators start at +0.0 and
// x + (+-0) = x.)  Replaces fm_first_from_bins + fm_pass2<4,4>: one LDS round trip and two thirds of a transform's
// radix-4 arithmetic less per frame.  z = conj of the spectrum bin (the inverse is conj o forward o conj).
// src / CONJ: where the 204 values z[0..203] come from -- the spectrum image itself (conjugated on the way, CONJ) or a
// small array of already conjugated values beside the image.  tw1[k * s1], tw2[k * s2]: the pass-3 twiddles of inputs
// 1 and 2 -- T64[k], T64[2k] for a transform with the ordinary tables, U[k], U[16 + k] for an odd half's (fm_pass_t).
// src_d1 (NIMG == 2): where the second image's 204 values start, relative to src0
template <int NN, bool CONJ, class TW1, class TW2, int NIMG = 1>
__device__ __attribute__((noinline)) void fm_inv_blocks(LdsArr X, LdsArr src0, TW1 tw1, int s1, TW2 tw2, int s2, int tid, int src_d1 = 0)
{
    constexpr int C = NN / 64, NITEM = C * 16, ITERS = (NIMG * NITEM + FM_T - 1) / FM_T;
    static_assert(3 * C > 204, "at most three of a block's 64 inputs come from the 204 bins");
    double2 z0[ITERS], z1[ITERS], z2[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int idd = it * FM_T + tid;
        const int img = (NIMG > 1 && idd >= NITEM) ? 1 : 0;
        const int id = idd - img * NITEM;
        const LdsArr src = src0 + img * src_d1;
        const int m = id >> 4;
        z0[it] = z1[it] = z2[it] = make_double2(0.0, 0.0);
        if (idd < NIMG * NITEM) {
            const double2 a = src[m];
            z0[it] = make_double2(a.x, CONJ ? -a.y : a.y);
            if (m + C < 204) {
                const double2 b = src[m + C];
                z1[it] = make_double2(b.x, CONJ ? -b.y : b.y);
            }
            if (m + 2 * C < 204) {
                const double2 c = src[m + 2 * C];
                z2[it] = make_double2(c.x, CONJ ? -c.y : c.y);
            }
        }
    }
    FM_PASS_SYNC();  // every bin is in registers before the image is overwritten
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int idd = it * FM_T + tid;
        if (idd < NIMG * NITEM) {
            const int img = (NIMG > 1 && idd >= NITEM) ? 1 : 0;
            const int id = idd - img * NITEM;
            const int m = id >> 4, k = id & 15;
            double2 a = z0[it], b = z0[it];
            if (m + 2 * C < 204) {
                const double2 v2 = cdmul(z2[it], tw2[k * s2]);
                a = cdadd(z0[it], v2);
                b = cdsub(z0[it], v2);
            }
            const LdsArr o = X + (img * NN + 64 * m + k);
            if (m + C < 204) {
                const double2 v1 = cdmul(z1[it], tw1[k * s1]);
                o[0] = cdadd(a, v1);
                o[32] = cdsub(a, v1);
                o[16] = make_double2(b.x + v1.y, b.y - v1.x);
                o[48] = make_double2(b.x - v1.y, b.y + v1.x);
            } else {
                o[0] = a;
                o[32] = a;
                o[16] = b;
                o[48] = b;
            }
        }
    }
    FM_PASS_SYNC();
}

// n = 9600: the first FOUR passes (4, 4, 4, 2) of the inverse transform straight from the gathered bins.  With C = 150 a
// block of 64 (see fm_inv_blocks) has at most TWO live inputs -- block m holds E_m[k + 16 c] = z[m] (+, -i, -, +i) v1,
// v1 = z[m + 150] T64[k] for m < 54, and plainly z[m] in every slot for 54 <= m < 150.  Pass 4 (radix 2, stride 64) pairs
// block m < 75 with block m + 75 (always of the plain kind):
//     out[128 m + kk] = E_m[kk] + w,  out[128 m + kk + 64] = E_m[kk] - w,   w = z[m + 75] T128[kk],  kk < 64
// -- the general passes' operations on these operands, minus those whose second operand is a zero of the zeroed array (as
// in fm_inv_blocks).  An item (m, k < 16) forms the eight outputs kk = k + 16 c, c < 4: 1200 items, one LDS round trip for
// four passes; what follows is fm_pass2<3, 5> and the real-parts-only last pass.
template <bool CONJ, class TW64, class TW128>
__device__ __attribute__((noinline)) void fm_inv_blocks128_9600(LdsArr X, LdsArr src, TW64 t64, TW128 t128, int tid)
{
    constexpr int NITEM = 75 * 16, ITERS = (NITEM + FM_T - 1) / FM_T;
    double2 z0[ITERS], zc[ITERS], z1[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int id = it * FM_T + tid;
        const int m = id >> 4;
        z0[it] = zc[it] = z1[it] = make_double2(0.0, 0.0);
        if (id < NITEM) {
            const double2 a = src[m], c = src[m + 75];
            z0[it] = make_double2(a.x, CONJ ? -a.y : a.y);
            zc[it] = make_double2(c.x, CONJ ? -c.y : c.y);
            if (m < 54) {
                const double2 b = src[m + 150];
                z1[it] = make_double2(b.x, CONJ ? -b.y : b.y);
            }
        }
    }
    FM_PASS_SYNC();  // every bin is in registers before the image is overwritten
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int id = it * FM_T + tid;
        if (id < NITEM) {
            const int m = id >> 4, k = id & 15;
            double2 e[4] = {z0[it], z0[it], z0[it], z0[it]};
            if (m < 54) {
                const double2 v1 = cdmul(z1[it], t64[k]);
                e[0] = cdadd(z0[it], v1);
                e[2] = cdsub(z0[it], v1);
                e[1] = make_double2(z0[it].x + v1.y, z0[it].y - v1.x);
                e[3] = make_double2(z0[it].x - v1.y, z0[it].y + v1.x);
            }
            const LdsArr o = X + (128 * m + k);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const double2 w = cdmul(zc[it], t128[k + 16 * c]);
                o[16 * c] = cdadd(e[c], w);
                o[16 * c + 64] = cdsub(e[c], w);
            }
        }
    }
    FM_PASS_SYNC();
}

// TWO consecutive Stockham passes (radix R1 at stride P, then R2 at stride P*R1) in one LDS round trip.  The R1*R2
// points of a group are closed under both passes: group g = m0*P + k1 (k1 < P) takes the R2 first-pass butterflies
// b1 = g + j2*(n/(R1 R2)) -- their outputs q1 feed the R1 second-pass butterflies b2 = m0*P*R1 + (k1 + q1*P), which
// write z[m0*P*R1*R2 + k1 + q1*P + q2*P*R1].  Same operands, same tables, same operation order as fm_pass<R1>
// followed by fm_pass<R2>: only the intermediate image stays in registers (an LDS store costs 13 cycles per wave
// instruction, and the image is written once per pass).
// the image between the fused first two passes (fm_first2_from_raw: a thread stores 16 consecutive slots) and the pass
// pair that reads it: slot s lives at s ^ ((s >> 4) & 7) -- the stores of eight neighbouring lanes then fall into eight
// different 16-byte bank groups, and so do the reads of eight consecutive slots (an aligned run of 8 is permuted in place)
__device__ __forceinline__ int fm_swz(int s) { return s ^ ((s >> 4) & 7); }

// SWZ_IN: the image read is the swizzled one
template <int R1, int R2, int NN = 0, int PP = 0, bool SWZ_IN = false, class TW1 = const double2 *, class TW2 = const double2 *, int NIMG = 1>
__device__ __attribute__((noinline)) void fm_pass2(LdsArr X0, TW1 tw1, TW2 tw2, int n_rt, int P_rt, unsigned pmagic, int tid)
{
    constexpr int RR = R1 * R2;
    static_assert(NIMG == 1 || NN != 0, "several images: compile-time frame size");
    constexpr int ITERS = ((NIMG * ((NN ? NN : FM_NMAX) / RR)) + FM_T - 1) / FM_T;
    const int n = NN ? NN : n_rt;
    const int P = PP ? PP : P_rt;
    const int ng = n / RR, nb1 = n / R1;
    double2 v[ITERS][R2][R1];
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int gg = it * FM_T + tid;
        const int img = (NIMG > 1 && gg >= ng) ? 1 : 0;
        const int g = gg - img * ng;
        const LdsArr X = X0 + img * NN;
        if (gg < NIMG * ng) {
            const int k1 = PP ? (g % PP) : (P == 1) ? 0 : g - (int)__umulhi((unsigned)g, pmagic) * P;
#pragma unroll
            for (int j2 = 0; j2 < R2; j2++) {
                const int b1 = g + j2 * ng;
#pragma unroll
                for (int j1 = 0; j1 < R1; j1++) {
                    v[it][j2][j1] = X[SWZ_IN ? fm_swz(b1 + j1 * nb1) : b1 + j1 * nb1];
                    if (j1 >= 1 && P > 1) v[it][j2][j1] = cdmul(v[it][j2][j1], tw1[k1 * j1]);
                }
                dft_r<R1>(v[it][j2]);
            }
            __builtin_amdgcn_sched_barrier(0);  // the second stage's twiddles are fetched after the first stage's are dead
#pragma unroll
            for (int q1 = 0; q1 < R1; q1++) {
                const int k2 = k1 + q1 * P;
                double2 w[R2];
#pragma unroll
                for (int j2 = 0; j2 < R2; j2++) {
                    w[j2] = v[it][j2][q1];
                    if (j2 >= 1) w[j2] = cdmul(w[j2], tw2[k2 * j2]);
                }
                dft_r<R2>(w);
#pragma unroll
                for (int q2 = 0; q2 < R2; q2++) v[it][q2][q1] = w[q2];
            }
        }
    }
    FM_PASS_SYNC();
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int gg = it * FM_T + tid;
        const int img = (NIMG > 1 && gg >= ng) ? 1 : 0;
        const int g = gg - img * ng;
        const LdsArr X = X0 + img * NN;
        if (gg < NIMG * ng) {
            const int k1 = PP ? (g % PP) : (P == 1) ? 0 : g - (int)__umulhi((unsigned)g, pmagic) * P;
            const LdsArr z = X + ((g - k1) * RR + k1);
#pragma unroll
            for (int q2 = 0; q2 < R2; q2++)
#pragma unroll
                for (int q1 = 0; q1 < R1; q1++) z[q1 * P + q2 * P * R1] = v[it][q2][q1];
        }
    }
    FM_PASS_SYNC();
}

// The first pass stores X[4b + q]: lane stride 4 slots, so the 8 lanes of a 16-byte-store group share two bank groups
// (4-way conflict, 32 LDS cycles per wave store instead of 8).  Rotating which output a lane stores in which
// instruction -- output (i + (lane >> 1)) & 3 in instruction i -- puts the 8 lanes on 8 different bank groups; the
// rotation is two conditional-move stages over the four values.
__device__ __forceinline__ void fm_store4_rotated(LdsArr X, int b, const double2 (&v)[4], int tid)
{
    const int r = (tid >> 1) & 3;
    double2 w[4], t[4], u[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        w[i] = v[i];
        // fresh values: LLVM otherwise turns "select of array elements" into a dynamically indexed private array (scratch)
        asm volatile("" : "+v"(w[i].x), "+v"(w[i].y));
    }
    // (component by component on values: `c ? a[i] : a[j]` on the structs is an lvalue select, i.e. an address select)
    const bool c1 = (r & 1) != 0, c2 = (r & 2) != 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const double ax = w[(i + 1) & 3].x, ay = w[(i + 1) & 3].y, bx = w[i].x, by = w[i].y;
        t[i] = make_double2(c1 ? ax : bx, c1 ? ay : by);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const double ax = t[(i + 2) & 3].x, ay = t[(i + 2) & 3].y, bx = t[i].x, by = t[i].y;
        u[i] = make_double2(c2 ? ax : bx, c2 ? ay : by);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) X[4 * b + ((i + r) & 3)] = u[i];
}

// The first pass (radix 4, stride 1, no twiddles) of the default frames, fused with what produces its input: the
// forward transform's straight from the frame's samples in global memory (x[b + j n/4], converted as :416-421), the
// inverse's from the 204 gathered bins (only input 0 of the butterflies b < 204 is not the zeroed array's (0, -0)).
// Same butterflies on the same operands as fm_pass<4>; what goes away is one LDS write of the whole image, one read
// of it and two barriers per transform.
template <int NN, bool F32IN, int NIMG = 1>
__device__ __attribute__((noinline)) void fm_first_from_raw(LdsArr X0, const int *raw_, const float2 *rawf_, int ic, int qc, int tid)
{
    constexpr int nb = NN / 4, ITERS = (NIMG * nb + FM_T - 1) / FM_T;
    gbl_i32 *raw = (gbl_i32 *)(unsigned long long)raw_;
    gbl_f2v *rawf = (gbl_f2v *)(unsigned long long)rawf_;
    int w[ITERS][4];
    f2v wf[ITERS][4];
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        int bb = it * FM_T + tid;
        bb = bb < NIMG * nb ? bb : NIMG * nb - 1;
        const int img = (NIMG > 1 && bb >= nb) ? 1 : 0;
        const int b = bb - img * nb + img * NN;  // (the second image's frame follows the first in memory)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (F32IN)
                wf[it][j] = rawf[b + j * nb];
            else
                w[it][j] = raw[b + j * nb];
        }
    }
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int bb = it * FM_T + tid;
        const int img = (NIMG > 1 && bb >= nb) ? 1 : 0;
        const int b = bb - img * nb;
        const LdsArr X = X0 + img * NN;
        if (bb < NIMG * nb) {
            double2 v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (F32IN)
                    v[j] = make_double2((double)wf[it][j].x, (double)wf[it][j].y);
                else
                    v[j] = make_double2((double)i16_to_float_java(java_short_add((int)(short)(w[it][j] & 0xffff), ic)),
                                        (double)i16_to_float_java(java_short_add(w[it][j] >> 16, qc)));
            }
            dft_r<4>(v);
            fm_store4_rotated(X, b, v, tid);
        }
    }
    __syncthreads();
}

// The first TWO passes (4, 4; strides 1 and 4) of the forward transform straight from the frame's samples: group g < n/16
// holds x[g + j2 n/16 + j1 n/4] -- four first-pass butterflies (no twiddles at stride 1), their outputs q1 through the
// four second-pass butterflies (T16[q1 j2]) -- and stores its 16 results at slots 16 g + q1 + 4 q2 of the SWIZZLED image
// (fm_swz).  The operations of fm_first_from_raw followed by fm_pass<4, n, 4>, on the same operands in the same order;
// what goes away is one LDS write and one read of the whole image, two barriers, and the first pass's rotated stores.
template <int NN, bool F32IN, class TW2>
__device__ __attribute__((noinline)) void fm_first2_from_raw(LdsArr X, const int *raw_, const float2 *rawf_, int ic, int qc, TW2 tw2,
                                                              int tid)
{
    constexpr int ng = NN / 16, nb1 = NN / 4;
    static_assert(ng <= FM_T, "one group per thread");
    gbl_i32 *raw = (gbl_i32 *)(unsigned long long)raw_;
    gbl_f2v *rawf = (gbl_f2v *)(unsigned long long)rawf_;
    const int g = tid < ng ? tid : ng - 1;
    int w[4][4];
    f2v wf[4][4];
#pragma unroll
    for (int j2 = 0; j2 < 4; j2++)
#pragma unroll
        for (int j1 = 0; j1 < 4; j1++) {
            if (F32IN)
                wf[j2][j1] = rawf[g + j2 * ng + j1 * nb1];
            else
                w[j2][j1] = raw[g + j2 * ng + j1 * nb1];
        }
    const bool dc = (ic != 0) || (qc != 0);
    if (tid < ng) {
        double2 v[4][4];
#pragma unroll
        for (int j2 = 0; j2 < 4; j2++) {
#pragma unroll
            for (int j1 = 0; j1 < 4; j1++) {
                if (F32IN)
                    v[j2][j1] = make_double2((double)wf[j2][j1].x, (double)wf[j2][j1].y);
                else  // (the pair at once, the correction skipped when there is none: common.h)
                    fm_convert(w[j2][j1], ic, qc, dc, v[j2][j1].x, v[j2][j1].y);
            }
            dft_r<4>(v[j2]);
        }
#pragma unroll
        for (int q1 = 0; q1 < 4; q1++) {
            double2 u[4];
#pragma unroll
            for (int j2 = 0; j2 < 4; j2++) {
                u[j2] = v[j2][q1];
                if (j2 >= 1) u[j2] = cdmul(u[j2], tw2[q1 * j2]);
            }
            dft_r<4>(u);
#pragma unroll
            for (int q2 = 0; q2 < 4; q2++) X[fm_swz(16 * tid + q1 + 4 * q2)] = u[q2];
        }
    }
    __syncthreads();
}

// The same in two steps (round 6, n = 9600): the frame's samples are REQUESTED a phase early -- at the top of the previous frame's
// RxDownSample, whose loop runs for longer than the round trip to memory takes and waits for nothing of its own -- and the two passes
// find them in registers.  With one workgroup a CU nothing else covers that latency: it was paid in full at the top of every frame.
struct FmRaw16 {
    int w[16];
    f2v wf[16];
};
template <int NN, bool F32IN>
__device__ __forceinline__ void fm_first2_request(FmRaw16 &r, const int *raw_, const float2 *rawf_, int tid)
{
    constexpr int ng = NN / 16, nb1 = NN / 4;
    gbl_i32 *raw = (gbl_i32 *)(unsigned long long)raw_;
    gbl_f2v *rawf = (gbl_f2v *)(unsigned long long)rawf_;
    const int g = tid < ng ? tid : ng - 1;
#pragma unroll
    for (int j2 = 0; j2 < 4; j2++)
#pragma unroll
        for (int j1 = 0; j1 < 4; j1++) {
            if (F32IN)
                r.wf[4 * j2 + j1] = rawf[g + j2 * ng + j1 * nb1];
            else
                r.w[4 * j2 + j1] = raw[g + j2 * ng + j1 * nb1];
        }
}
template <int NN, bool F32IN, class TW2>
__device__ __forceinline__ void fm_first2_from_regs(LdsArr X, const FmRaw16 &r, int ic, int qc, TW2 tw2, int tid)
{
    constexpr int ng = NN / 16;
    const bool dc = (ic != 0) || (qc != 0);
    if (tid < ng) {
        double2 v[4][4];
#pragma unroll
        for (int j2 = 0; j2 < 4; j2++) {
#pragma unroll
            for (int j1 = 0; j1 < 4; j1++) {
                if (F32IN)
                    v[j2][j1] = make_double2((double)r.wf[4 * j2 + j1].x, (double)r.wf[4 * j2 + j1].y);
                else
                    fm_convert(r.w[4 * j2 + j1], ic, qc, dc, v[j2][j1].x, v[j2][j1].y);
            }
            dft_r<4>(v[j2]);
        }
#pragma unroll
        for (int q1 = 0; q1 < 4; q1++) {
            double2 u[4];
#pragma unroll
            for (int j2 = 0; j2 < 4; j2++) {
                u[j2] = v[j2][q1];
                if (j2 >= 1) u[j2] = cdmul(u[j2], tw2[q1 * j2]);
            }
            dft_r<4>(u);
#pragma unroll
            for (int q2 = 0; q2 < 4; q2++) X[fm_swz(16 * tid + q1 + 4 * q2)] = u[q2];
        }
    }
    __syncthreads();
}

// fm_first_from_raw<NN, F32IN, 2> in the same two steps (the paired kernel): d1 = where the second image's frame starts relative to
// the first's (NN, or 0 when the call's last frame stands alone: the second image is then a copy, computed and dropped)
template <int NN, bool F32IN>
__device__ __forceinline__ void fm_first_request2(FmRaw16 &r, const int *raw_, const float2 *rawf_, int d1, int tid)
{
    constexpr int nb = NN / 4, ITERS = (2 * nb + FM_T - 1) / FM_T;
    static_assert(ITERS * 4 <= 16, "sixteen samples a thread");
    gbl_i32 *raw = (gbl_i32 *)(unsigned long long)raw_;
    gbl_f2v *rawf = (gbl_f2v *)(unsigned long long)rawf_;
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        int bb = it * FM_T + tid;
        bb = bb < 2 * nb ? bb : 2 * nb - 1;
        const int img = bb >= nb ? 1 : 0;
        const int b = bb - img * nb + img * d1;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (F32IN)
                r.wf[4 * it + j] = rawf[b + j * nb];
            else
                r.w[4 * it + j] = raw[b + j * nb];
        }
    }
}
template <int NN, bool F32IN>
__device__ __forceinline__ void fm_first_from_regs2(LdsArr X0, const FmRaw16 &r, int ic, int qc, int tid)
{
    constexpr int nb = NN / 4, ITERS = (2 * nb + FM_T - 1) / FM_T;
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int bb = it * FM_T + tid;
        const int img = bb >= nb ? 1 : 0;
        const int b = bb - img * nb;
        const LdsArr X = X0 + img * NN;
        if (bb < 2 * nb) {
            double2 v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (F32IN)
                    v[j] = make_double2((double)r.wf[4 * it + j].x, (double)r.wf[4 * it + j].y);
                else
                    v[j] = make_double2((double)i16_to_float_java(java_short_add((int)(short)(r.w[4 * it + j] & 0xffff), ic)),
                                        (double)i16_to_float_java(java_short_add(r.w[4 * it + j] >> 16, qc)));
            }
            dft_r<4>(v);
            fm_store4_rotated(X, b, v, tid);
        }
    }
    __syncthreads();
}

// The same for one half of a frame of 2 NN samples (k_front_fft2x): the frame's radix-2 first pass, X_c[h] = x[h] +/- x[h + NN]
// (dft_r<2>), feeds the half's first two passes directly -- 32 samples per thread in flight at once (as a load / convert /
// store loop the half paid the HBM latency thirteen times).  ODD: the half of the odd bins, whose passes multiply EVERY
// input j >= 1 by the two-dimensional tables U_p[(j-1) P' + k'] (fm_pass_t) -- tw0 = U_0 (stride 1), tw1 = U_1 (stride 4);
// the even half has no stride-1 twiddles and tw1 = T16.
template <int NN, bool F32IN, bool ODD>
__device__ __attribute__((noinline)) void fm_first2x_from_raw(LdsArr X, const int *raw_, const float2 *rawf_, int ic, int qc, GblArr tw0,
                                                               GblArr tw1, int tid)
{
    constexpr int ng = NN / 16, nb1 = NN / 4;
    static_assert(ng <= FM_T, "one group per thread");
    gbl_i32 *raw = (gbl_i32 *)(unsigned long long)raw_;
    gbl_f2v *rawf = (gbl_f2v *)(unsigned long long)rawf_;
    const int g = tid < ng ? tid : ng - 1;
    const bool dc = (ic != 0) || (qc != 0);
    double2 v[4][4];
#pragma unroll
    for (int j2 = 0; j2 < 4; j2++) {
        int w0[4], w1[4];
        f2v f0[4], f1[4];
#pragma unroll
        for (int j1 = 0; j1 < 4; j1++) {
            const int h = g + j2 * ng + j1 * nb1;
            if (F32IN) {
                f0[j1] = rawf[h];
                f1[j1] = rawf[h + NN];
            } else {
                // (the frame's SECOND read -- the odd half's -- is its last use: marked so, like the dm stores, to leave the L2 to the
                //  per-stream scratch of the even halves.  Counter traffic 27.7 -> 26.8 GB a step, the kernel 15.58 -> 15.4 ms: the 32
                //  workgroups of an XCD still cycle 7 MB of frame + scratch through a 4 MB L2)
                if (ODD) {
                    w0[j1] = __builtin_nontemporal_load(&raw[h]);
                    w1[j1] = __builtin_nontemporal_load(&raw[h + NN]);
                } else {
                    w0[j1] = raw[h];
                    w1[j1] = raw[h + NN];
                }
            }
        }
#pragma unroll
        for (int j1 = 0; j1 < 4; j1++) {
            double2 a, b;
            if (F32IN) {
                a = make_double2((double)f0[j1].x, (double)f0[j1].y);
                b = make_double2((double)f1[j1].x, (double)f1[j1].y);
            } else {
                fm_convert(w0[j1], ic, qc, dc, a.x, a.y);
                fm_convert(w1[j1], ic, qc, dc, b.x, b.y);
            }
            v[j2][j1] = ODD ? cdsub(a, b) : cdadd(a, b);
        }
    }
    if (tid < ng) {
#pragma unroll
        for (int j2 = 0; j2 < 4; j2++) {
            if (ODD) {
#pragma unroll
                for (int j1 = 1; j1 < 4; j1++) v[j2][j1] = cdmul(v[j2][j1], tw0[j1 - 1]);
            }
            dft_r<4>(v[j2]);
        }
#pragma unroll
        for (int q1 = 0; q1 < 4; q1++) {
            double2 u[4];
#pragma unroll
            for (int j2 = 0; j2 < 4; j2++) {
                u[j2] = v[j2][q1];
                if (j2 >= 1) u[j2] = cdmul(u[j2], ODD ? tw1[(j2 - 1) * 4 + q1] : tw1[q1 * j2]);
            }
            dft_r<4>(u);
#pragma unroll
            for (int q2 = 0; q2 < 4; q2++) X[fm_swz(16 * tid + q1 + 4 * q2)] = u[q2];
        }
    }
    __syncthreads();
}

template <int NN>
__device__ __attribute__((noinline)) void fm_first_from_bins(LdsArr X, double2 in0, int tid)
{
    constexpr int nb = NN / 4, ITERS = (nb + FM_T - 1) / FM_T;
    static_assert(FM_T >= 204, "butterfly b < 204 belongs to thread b");
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int b = it * FM_T + tid;
        if (b < nb) {
            const double2 Z = make_double2(0.0, -0.0);  // conj of the zeroed array
            double2 v[4] = {it == 0 ? in0 : Z, Z, Z, Z};
            dft_r<4>(v);
            fm_store4_rotated(X, b, v, tid);
        }
    }
    __syncthreads();
}

// A pass whose radix is a prime above 7 (round 5; the oracle's generic branch in fft_f64_mixed_forward): one thread per OUTPUT,
//   out[(b - k) r + k + q P] = v_0 + v_1 W[q mod r] + ... + v_{r-1} W[(r-1) q mod r],   v_j = X[b + j nb] (x T[k j] for j >= 1, P > 1)
// every product a full complex multiply, summed left to right -- the same operations in the same order as the oracle, the
// twiddled inputs re-formed for every output (identical values).  Out of place: outputs go to the stream's scratch in
// global memory, then back into the image.  O(n r) multiplications a pass: a correctness path (11.025 kHz: n = 1102 = 2 19 29).
__device__ __attribute__((noinline)) void fm_pass_generic(LdsArr X, const double2 *tw, const double2 *wr, double2 *G, int n, int P, int r,
                                                          int tid)
{
    const int nb = n / r;
    for (int o = tid; o < n; o += FM_T) {
        const int q = o / nb, b = o - q * nb;  // outputs of one q are contiguous over b: neighbouring threads read neighbouring inputs
        const int k = b % P;
        double2 acc = X[b];
        int m = 0;  // (j q) mod r
        for (int j = 1; j < r; j++) {
            m += q;
            if (m >= r) m -= r;
            double2 v = X[b + j * nb];
            if (P > 1) v = cdmul(v, tw[k * j]);
            acc = cdadd(acc, cdmul(v, wr[m]));
        }
        G[(b - k) * r + k + q * P] = acc;
    }
    __syncthreads();
    for (int i = tid; i < n; i += FM_T) X[i] = G[i];
    __syncthreads();
}

// ---- round 5: the 4410-sample frame of a 44.1 kHz card (2 . 3 . 3 . 5 . 7 . 7, the oracle's order) with compile-time passes.
// fm_passr_real / fm_passr_band: fm_pass5_real / fm_pass5_band for any radix (the last pass of 4410 is a radix-7 one).
template <int R, int NN, int PP, int COMPACT, class TW, int NIMG = 1>
__device__ __attribute__((noinline)) void fm_passr_real(LdsArr X, TW tw, double norm, int tid, const double *hist_)
{
    constexpr int nb = NN / R, ITERS = (NIMG * nb + FM_T - 1) / FM_T;
    static_assert(COMPACT == 1, "compact real samples");
    double o[ITERS][R];
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int bb = it * FM_T + tid;
        if (bb < NIMG * nb) {
            const int img = (NIMG > 1 && bb >= nb) ? 1 : 0;
            const int b = bb - img * nb;
            const LdsArr Xi = X + img * NN;
            const int k = b % PP;
            double2 v[R];
#pragma unroll
            for (int j = 0; j < R; j++) {
                v[j] = Xi[b + j * nb];
                if (j >= 1) v[j] = cdmul(v[j], tw[k * j]);
            }
            dft_r<R>(v);  // (only the real parts are read: the imaginary halves have no reader and are not formed)
#pragma unroll
            for (int q = 0; q < R; q++) o[it][q] = v[q].x * norm;
        }
    }
    FM_PASS_SYNC();
    lds_f64 *Rb = reinterpret_cast<lds_f64 *>(X.p);
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int bb = it * FM_T + tid;
        if (bb < NIMG * nb) {
            const int img = (NIMG > 1 && bb >= nb) ? 1 : 0;
            const int b = bb - img * nb;
            const int k = b % PP;
            const int j0 = (b - k) * R + k;
#pragma unroll
            for (int q = 0; q < R; q++) Rb[img * (2 * NN) + FM_RB0 + j0 + q * PP] = o[it][q];
        }
    }
    if (hist_ != nullptr) {
        const lds_f64 *hist = (const lds_f64 *)(unsigned)(unsigned long long)hist_;
        if (tid < 26) Rb[FM_RB0 - 26 + tid] = hist[tid];
    }
    FM_PASS_SYNC();
}

template <int R, int NN, int PP, class TW, int NIMG = 1>
__device__ __attribute__((noinline)) void fm_passr_band(LdsArr X, TW tw, int need_end, int tid)
{
    constexpr int nb = NN / R, ITERS = (NIMG * nb + FM_T - 1) / FM_T;
    static_assert(PP == nb, "the last pass: one butterfly per k");
    double2 v[ITERS][R];
    unsigned need[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int bb = it * FM_T + tid;
        const int img = (NIMG > 1 && bb >= nb) ? 1 : 0;
        const int b = bb - img * nb;  // == k
        const LdsArr Xi = X + img * NN;
        need[it] = 0u;
        if (bb < NIMG * nb) {
#pragma unroll
            for (int q = 0; q < R; q++)
                if (b + q * PP < need_end) need[it] |= 1u << q;
            if (need[it]) {  // (a butterfly none of whose outputs is read is not formed: its inputs are dead after this pass)
#pragma unroll
                for (int j = 0; j < R; j++) {
                    v[it][j] = Xi[b + j * nb];
                    if (j >= 1) v[it][j] = cdmul(v[it][j], tw[b * j]);
                }
                dft_r<R>(v[it]);
            }
        }
    }
    FM_PASS_SYNC();
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int bb = it * FM_T + tid;
        if (bb < NIMG * nb) {
            const int img = (NIMG > 1 && bb >= nb) ? 1 : 0;
            const int b = bb - img * nb;
            const LdsArr Xi = X + img * NN;
#pragma unroll
            for (int q = 0; q < R; q++)
                if (need[it] & (1u << q)) Xi[b + q * PP] = v[it][q];
        }
    }
    FM_PASS_SYNC();
}

// The first pass pair (R1, R2; strides 1 and R1) of the INVERSE transform straight from the 204 gathered bins: what
// fm_pass2<R1, R2, NN, 1> computes on the zeroed, conjugated array of FUNcubeBPSKDemod.java:458 -- element i = conj(bin i) for
// i < 204, (0.0, -0.0) elsewhere -- without that array ever being written: item g reads element g + j2 NN/(R1 R2) + j1 NN/R1,
// which is a bin only for j1 = j2 = 0 and g < 204.  The same operations on the same operand values (the zeros included: a
// zero's sign is whatever the general pass would have produced), one LDS write of the image instead of zero-fill + scatter +
// read + write.  src: the spectrum image itself (X + centreBin - 102); every bin is in registers before the first store.
template <int R1, int R2, int NN, class TW2, int NIMG = 1>
__device__ __attribute__((noinline)) void fm_inv_pair_from_bins(LdsArr X, LdsArr src0, TW2 tw2, int tid, int src_d1 = 0)
{
    constexpr int RR = R1 * R2, ng = NN / RR, ITERS = (NIMG * ng + FM_T - 1) / FM_T;
    static_assert(ng >= 204 && FM_T >= 204, "item g < 204 holds bin g, in the first round");
    double2 v[ITERS][R2][R1];
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int gg = it * FM_T + tid;
        const int img = (NIMG > 1 && gg >= ng) ? 1 : 0;
        const int g = gg - img * ng;
        if (gg < NIMG * ng) {
#pragma unroll
            for (int j2 = 0; j2 < R2; j2++) {
#pragma unroll
                for (int j1 = 0; j1 < R1; j1++) v[it][j2][j1] = make_double2(0.0, -0.0);
            }
            if ((NIMG > 1 || it == 0) && g < 204) {
                const double2 a = (src0 + img * src_d1)[g];
                v[it][0][0] = make_double2(a.x, -a.y);
            }
#pragma unroll
            for (int j2 = 0; j2 < R2; j2++) dft_r<R1>(v[it][j2]);  // (stride 1: no twiddles in the first pass)
#pragma unroll
            for (int q1 = 0; q1 < R1; q1++) {
                const int k2 = q1;  // k1 + q1 P with k1 = 0, P = 1
                double2 w[R2];
#pragma unroll
                for (int j2 = 0; j2 < R2; j2++) {
                    w[j2] = v[it][j2][q1];
                    if (j2 >= 1) w[j2] = cdmul(w[j2], tw2[k2 * j2]);
                }
                dft_r<R2>(w);
#pragma unroll
                for (int q2 = 0; q2 < R2; q2++) v[it][q2][q1] = w[q2];
            }
        }
    }
    __syncthreads();  // every bin has been read from the image
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int gg = it * FM_T + tid;
        if (gg < NIMG * ng) {
            const int img = (NIMG > 1 && gg >= ng) ? 1 : 0;
            const int g = gg - img * ng;
            const LdsArr z = X + (img * NN + g * RR);
#pragma unroll
            for (int q2 = 0; q2 < R2; q2++)
#pragma unroll
                for (int q1 = 0; q1 < R1; q1++) z[q1 + q2 * R1] = v[it][q2][q1];
        }
    }
    __syncthreads();
}

__device__ __forceinline__ const double2 *fm_table(const double2 *twL, const FftmArgs &a, int p, int P)
{
    // the narrow tables sit in LDS: a pass that starts with a round trip to L2 for its twiddles costs ~2 us,
    // 14 times per frame, with nothing else to run on the CU
    return (a.tw_off[p] + P * a.rad[p] <= a.lds_tw) ? twL + a.tw_off[p] : a.f.tw + a.tw_off[p];
}

// first_done: the caller has already run the first pass (fm_first_from_*, default frames only)
// mode (default frames): FM_FULL the whole transform; FM_FWD_BAND the forward transform of the front end, last pass
// restricted to the outputs [0, need_end) that are read; FM_INV_REAL the inverse transform after fm_inv_blocks
// (passes 1-3 done), last pass real parts only, scaled by norm
// LDSTW: the tables of the first passes are in LDS (k_front_fftm; a.lds_tw entries) -- for the default frames that is a
// compile-time fact (launch_front_fftm refuses any other split): fftm_twiddles lays the tables out back to back,
//   n = 9600: radices 4,4,4,2,3,5,5, lengths 4,16,64,128,384 | 1920,9600, offsets 0,4,20,84,212 | 596,2516
//   n = 4800: radices 4,4,4,3,5,5,   lengths 4,16,64,192,960 | 4800,      offsets 0,4,20,84,276 | 1236
constexpr int FM_LDS_TW_9600 = 596, FM_LDS_TW_4800 = 1236;
enum { FM_FULL = 0, FM_FWD_BAND = 1, FM_INV_REAL = 2 };
template <bool LDSTW>
__device__ __forceinline__ void fm_forward(LdsArr X, const double2 *twL, const FftmArgs &a, int tid, bool first_done,
                                           int mode = FM_FULL, int need_end = 0, double norm = 1.0, const double *hist = nullptr,
                                           double *gout = nullptr)
{
    const GblArr g = gbl_arr(a.f.tw);
    // the reference's two default frames: the plan is known (fftm_radices: 4,4,4,2,3,5,5 / 4,4,4,3,5,5)
    if (a.f.n == 9600) {
        if constexpr (LDSTW) {
            if (mode == FM_INV_REAL) {
                // k_front_fftm's inverse transform: passes 1-4 came from fm_inv_blocks128_9600
                fm_pass2<3, 5, 9600, 128>(X, lds_arr(twL) + 212, g + 596, 9600, 128, 0u, tid);
                fm_pass5_real<9600, 1920, false, 1>(X, g + 2516, norm, tid, hist);
                return;
            }
            if (mode == FM_FWD_BAND && first_done) {
                // k_front_fftm's forward transform: passes 1-2 came from fm_first2_from_raw (swizzled image); the other
                // five go as 4,2 | 3,5 | 5 -- three LDS round trips, 1200 / 640 / 1920 work items for the 768 threads
                const LdsArr t = lds_arr(twL);
                fm_pass2<4, 2, 9600, 16, true>(X, t + 20, t + 84, 9600, 16, 0u, tid);
                fm_pass2<3, 5, 9600, 128>(X, t + 212, g + 596, 9600, 128, 0u, tid);
                fm_pass5_band<9600, 1920>(X, g + 2516, need_end, tid);
                return;
            }
        } else {
            if (mode == FM_INV_REAL && gout != nullptr) {  // k_front_fft2x's even half after fm_inv_blocks128_9600; samples to gout
                fm_pass2<3, 5, 9600, 128>(X, g + 212, g + 596, 9600, 128, 0u, tid);
                fm_pass5_real<9600, 1920, false, 2>(X, g + 2516, norm, tid, nullptr, gout);
                return;
            }
            if (mode == FM_FWD_BAND && first_done) {  // k_front_fft2x's even half after fm_first2x_from_raw: the same, tables in L2
                fm_pass2<4, 2, 9600, 16, true>(X, g + 20, g + 84, 9600, 16, 0u, tid);
                fm_pass2<3, 5, 9600, 128>(X, g + 212, g + 596, 9600, 128, 0u, tid);
                fm_pass5_band<9600, 1920>(X, g + 2516, need_end, tid);
                return;
            }
        }
        auto head = [&](auto t) {
            if (mode != FM_INV_REAL) {
                if (!first_done) fm_pass<4, 9600, 1>(X, t, 9600, 1, 0u, tid);
                fm_pass2<4, 4, 9600, 4>(X, t + 4, t + 20, 9600, 4, 0u, tid);
            }
            fm_pass2<2, 3, 9600, 64>(X, t + 84, t + 212, 9600, 64, 0u, tid);
        };
        if constexpr (LDSTW)
            head(lds_arr(twL));
        else
            head(g);
        fm_pass<5, 9600, 384>(X, g + 596, 9600, 384, 0u, tid);
        if (mode == FM_FWD_BAND)
            fm_pass5_band<9600, 1920>(X, g + 2516, need_end, tid);
        else if (mode == FM_INV_REAL)
            fm_pass5_real<9600, 1920, false, LDSTW ? 1 : 0>(X, g + 2516, norm, tid, hist);  // (k_front_fftm: compact real samples)
        else
            fm_pass<5, 9600, 1920>(X, g + 2516, 9600, 1920, 0u, tid);
        return;
    }
    if (a.f.n == 4800) {
        auto head = [&](auto t) {
            if (mode != FM_INV_REAL) {
                if (!first_done) fm_pass<4, 4800, 1>(X, t, 4800, 1, 0u, tid);
                fm_pass2<4, 4, 4800, 4>(X, t + 4, t + 20, 4800, 4, 0u, tid);
            }
            fm_pass2<3, 5, 4800, 64>(X, t + 84, t + 276, 4800, 64, 0u, tid);
        };
        if constexpr (LDSTW)
            head(lds_arr(twL));
        else
            head(g);
        if (mode == FM_FWD_BAND)
            fm_pass5_band<4800, 960>(X, g + 1236, need_end, tid);
        else if (mode == FM_INV_REAL)
            fm_pass5_real<4800, 960, false, LDSTW ? 1 : 0>(X, g + 1236, norm, tid, hist);
        else
            fm_pass<5, 4800, 960>(X, g + 1236, 4800, 960, 0u, tid);
        return;
    }
    if (a.f.n == 4410) {
        // radices 2,3,3,5,7,7; table lengths 2,6,18,90,630,4410 at offsets 0,2,8,26,116,746, ALL in LDS (launch_front_fftm
        // checks it): [2,3] [3,5] [7] [7] -- four LDS round trips with compile-time strides instead of the run-time plan's five
        if constexpr (LDSTW) {
            const LdsArr t = lds_arr(twL);
            if (mode != FM_INV_REAL) fm_pass2<2, 3, 4410, 1>(X, t, t + 2, 4410, 1, 0u, tid);  // (the inverse's came from the bins)
            fm_pass2<3, 5, 4410, 6>(X, t + 8, t + 26, 4410, 6, 0u, tid);
            fm_pass<7, 4410, 90>(X, t + 116, 4410, 90, 0u, tid);
            if (mode == FM_FWD_BAND)
                fm_passr_band<7, 4410, 630>(X, t + 746, need_end, tid);
            else if (mode == FM_INV_REAL)
                fm_passr_real<7, 4410, 630, 1>(X, t + 746, norm, tid, hist);
            else
                fm_pass<7, 4410, 630>(X, t + 746, 4410, 630, 0u, tid);
            return;
        }
    }
    // any other frame: run-time plan, tables wherever fm_table finds them (flat accesses)
    int P = 1;
    for (int p = 0; p < a.np;) {
        const int r = a.rad[p];
        const double2 *tw = fm_table(twL, a, p, P);
        // after the first pass (whose stores would all hit one bank group as a pair), consecutive passes go in pairs
        const int r2 = (p >= 1 && p + 1 < a.np) ? a.rad[p + 1] : 0;
        const int pair = r * 8 + r2;
        if (pair == 4 * 8 + 4 || pair == 4 * 8 + 2 || pair == 2 * 8 + 3 || pair == 3 * 8 + 5) {
            const double2 *tw2 = fm_table(twL, a, p + 1, P * r);
            if (pair == 4 * 8 + 4)
                fm_pass2<4, 4>(X, tw, tw2, a.f.n, P, a.pmagic[p], tid);
            else if (pair == 4 * 8 + 2)
                fm_pass2<4, 2>(X, tw, tw2, a.f.n, P, a.pmagic[p], tid);
            else if (pair == 2 * 8 + 3)
                fm_pass2<2, 3>(X, tw, tw2, a.f.n, P, a.pmagic[p], tid);
            else
                fm_pass2<3, 5>(X, tw, tw2, a.f.n, P, a.pmagic[p], tid);
            P *= r * r2;
            p += 2;
            continue;
        }
        if (r > 7)
            fm_pass_generic(X, tw, a.f.tw + a.wr_off[p], a.gscratch + (long long)blockIdx.x * a.gscratch_stride, a.f.n, P, r, tid);
        else if (r == 4)
            fm_pass<4>(X, tw, a.f.n, P, a.pmagic[p], tid);
        else if (r == 2)
            fm_pass<2>(X, tw, a.f.n, P, a.pmagic[p], tid);
        else if (r == 3)
            fm_pass<3>(X, tw, a.f.n, P, a.pmagic[p], tid);
        else if (r == 7)
            fm_pass<7>(X, tw, a.f.n, P, a.pmagic[p], tid);
        else
            fm_pass<5>(X, tw, a.f.n, P, a.pmagic[p], tid);
        P *= r;
        p += 1;
    }
}

template <bool F32IN>
__global__ __launch_bounds__(FM_T) void k_front_fftm(FftmArgs aa)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const FftFrontArgs &a = aa.f;
    const int n = a.n;
    double2 *X = reinterpret_cast<double2 *>(smem);      // [n]
    const LdsArr XL = lds_arr(smem);                     // the same image for the passes (typed view)
    // [32]; during RxDownSample slots 26..51 (over the argmax scratch behind it, dead by then) hold the frame's first 26
    // scaled samples, so that a window reaching back into the previous frame is one contiguous run as well
    double *hist = reinterpret_cast<double *>(X + n);
    double *redv = hist + 32;                            // [16] per-wave best value
    int *redi = reinterpret_cast<int *>(redv + 16);      // [16] per-wave best index
    double2 *twL = reinterpret_cast<double2 *>(redi + 16);  // [lds_tw] twiddle tables of the first passes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < aa.lds_tw; i += FM_T) twL[i] = aa.f.tw[i];
    const int s = blockIdx.x;
    const int beg = a.do_up ? n / 4 : 0;
    const int end = a.do_up ? n / 2 : n / 4;
    // |X| over [beg+24, end-24) and the boxcar sums over [beg+74, end-74) live in the upper part of the image, dead
    // after the forward transform (bins up to n/2+101 are still gathered below); beg is a multiple of 4
    const int pbase = beg + 24;
    double *P = reinterpret_cast<double *>(X + (n / 2 + 104));
    double *A = P + (n / 4 - 48);
    const int abase = beg + 74;
    FftFrontState *sp = &a.st[s];
    if (tid < 26) hist[tid] = sp->hist[tid];
    double avePeakPower = sp->avePeakPower, aveCentreBin = sp->aveCentreBin;
    int centreBin = sp->centreBin;
    // :399-402 -- float expressions widened to double
    const double CFREQ_INV = (double)(1.0F - (2.0F / (1 + 1))), CFREQ_AVG = (double)(2.0F / (1 + 1));
    const double PSD_INV = (double)(1.0F - (2.0F / (10 + 1))), PSD_AVG = (double)(2.0F / (10 + 1));
    const double HOWARD = 0.9 * 32768.0;
    const int D = a.decim;
    const double norm = 1.0 / (double)n;
    const int *__restrict__ raw = a.raw + (long long)s * a.stride_pairs;
    const float2 *__restrict__ rawf = a.rawf + (long long)s * a.stride_pairs;
    double2 *dm = a.dm + (long long)s * a.dm_stride;
    // diagnostics (JSDR_FFT_PHASECLK=1): thread 0 of stream 0 accumulates the clock ticks of every phase
    long long *clk = reinterpret_cast<long long *>(twL + aa.lds_tw);  // the 64 spare bytes behind the tables
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
    __syncthreads();
    if (timing) tprev = (long long)clock64();
    FmRaw16 pre;  // n = 9600: the next frame's samples, requested during RxDownSample
    if (n == 9600) fm_first2_request<9600, F32IN>(pre, raw, rawf, tid);

    for (int f = 0; f < a.nframes; f++) {
        const long long t0 = (long long)f * n;  // call-relative index of the frame's first sample
        // opaque per frame: nothing derived from the thread index is loop invariant, or LLVM hoists every address
        // of every phase out of the frame loop and spills them (the same trap as in k_front_fft)
        int tf = tid;
        asm volatile("" : "+v"(tf));
        const bool fused_first = (n == 9600 || n == 4800);
        const bool compact = fused_first || n == 4410;  // the inverse from the bins, compact real samples, windows as aligned runs
        if (n == 9600) {
            fm_first2_from_regs<9600, F32IN>(XL, pre, a.ic, a.qc, lds_arr(twL) + 4, tf);
        } else if (n == 4800) {
            fm_first_from_raw<4800, F32IN>(XL, raw + t0, rawf + t0, a.ic, a.qc, tf);
        } else {
            // ---- frame -> LDS, natural order (:416-421)
            {
                // all of a thread's samples in flight before the first conversion (a load / convert / store loop pays the
                // HBM latency once per sample: 13 times per frame)
                constexpr int NLD = (FM_NMAX + FM_T - 1) / FM_T;
                int w[NLD];
                float2 wf[NLD];
#pragma unroll
                for (int q = 0; q < NLD; q++) {
                    int t = tf + q * FM_T;
                    t = t < n ? t : n - 1;
                    if (F32IN)
                        wf[q] = rawf[t0 + t];
                    else
                        w[q] = raw[t0 + t];
                }
#pragma unroll
                for (int q = 0; q < NLD; q++) {
                    const int t = tf + q * FM_T;
                    if (t < n) {
                        double di, dq;
                        if (F32IN) {
                            di = (double)wf[q].x;
                            dq = (double)wf[q].y;
                        } else {
                            di = (double)i16_to_float_java(java_short_add((int)(short)(w[q] & 0xffff), a.ic));
                            dq = (double)i16_to_float_java(java_short_add(w[q] >> 16, a.qc));
                        }
                        X[t] = make_double2(di, dq);
                    }
                }
            }
            __syncthreads();
        }
        PHASE(0)
        fm_forward<true>(XL, twL, aa, tf, fused_first, compact ? FM_FWD_BAND : FM_FULL, end + 102);  // :422-423; bins < end + 102 are read
        PHASE(1)
        // ---- |X| (:425-427) over the band the boxcar reads
        for (int i = pbase + tf; i < end - 24; i += FM_T) {
            const double2 v = X[i];
            P[i - pbase] = sqrt(v.x * v.x + v.y * v.y);
        }
        __syncthreads();
        PHASE(6)
        // ---- 100-wide boxcar, summed j ascending for every i (:433-437); first maximum (:439-442).  A thread owns
        // the outputs i (even) and i+1: both windows come out of the same 51 aligned 16-byte reads.
        double bestv = 0.0;  // maxBin starts at 0.0, binPos at -1
        int besti = -1;
        for (int i = beg + 74 + 2 * tf; i < end - 75; i += 2 * FM_T) {
            const double2 *w = reinterpret_cast<const double2 *>(P + (i - 50 - pbase));
            double a0, a1;
            boxcar_pair(w, a0, a1);
            asm volatile("" : "+v"(a0), "+v"(a1));  // due here: sunk into the conditional uses below, the sums drag all 51 reads along
            if (i >= beg + 75) {
                A[i - abase] = a0;
                if (bestv < a0) {  // i ascends within a thread: strict '<' keeps the first maximum
                    bestv = a0;
                    besti = i;
                }
            }
            if (i + 1 < end - 75) {
                A[i + 1 - abase] = a1;
                if (bestv < a1) {
                    bestv = a1;
                    besti = i + 1;
                }
            }
        }
        wave_first_max(bestv, besti);
        if (lane == 0) {
            redv[wave] = bestv;
            redi[wave] = besti;
        }
        __syncthreads();
        PHASE(7)
        // ---- centre-bin rule (:444-453), evaluated by every thread on the same values
        {
            // the twelve per-wave maxima meet in lanes 0..11 of every wave (the same combine as above: larger value,
            // then smaller index; an empty candidate never wins) -- as a 12-step loop run by every thread this cost the
            // SIMDs 3.6k cycles a frame
            double mv = 0.0;
            int mi = -1;
            if (lane < FM_T / 64) {
                mv = redv[lane];
                mi = redi[lane];
            }
            wave_first_max(mv, mi);
            const double maxBin = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(mv)),
                                                   __builtin_amdgcn_readfirstlane(__double2loint(mv)));
            const int binPos = __builtin_amdgcn_readfirstlane(mi);
            if (centreBin < 0) centreBin = 0;
            if (centreBin > end - 1) centreBin = end - 1;
            // aveTemp is cleared per frame (:431) and only [beg+75, end-75) is filled
            const double atc = (centreBin >= beg + 75 && centreBin < end - 75) ? A[centreBin - abase] : 0.0;
            avePeakPower = (PSD_AVG * atc) + (PSD_INV * avePeakPower);
            if (maxBin > (avePeakPower / 4) * 5 && binPos > 0) {
                aveCentreBin = (CFREQ_AVG * (double)(float)binPos) + (CFREQ_INV * aveCentreBin);
                centreBin = (int)(aveCentreBin + (double)1.0F);
            }
            if (centreBin < 102) centreBin = 102;
        }
        PHASE(2)
        // ---- 204 bins around the centre to bin 0 of a zeroed array (:458), inverse transform (:459) as
        // conj o forward o conj; only real parts are read afterwards, so the closing conjugation is dropped
        if (compact) {
            // default frames: passes 1-3 straight from the bins, last pass real parts only and already scaled by 1/n
            const LdsArr t64 = lds_arr(twL) + 20;  // the 64-entry table of pass 3
            if (n == 9600)
                fm_inv_blocks128_9600<true>(XL, XL + (centreBin - 102), t64, lds_arr(twL) + 84, tf);
            else if (n == 4800)
                fm_inv_blocks<4800, true>(XL, XL + (centreBin - 102), t64, 1, t64, 2, tf);
            else
                fm_inv_pair_from_bins<2, 3, 4410>(XL, XL + (centreBin - 102), lds_arr(twL) + 2, tf);  // 4410: the first pair
            PHASE(3)
            fm_forward<true>(XL, twL, aa, tf, true, FM_INV_REAL, 0, norm, hist);
            PHASE(4)
            // ---- RxDownSample(re, re) (:461-463, :470-492) from the compact samples: sample t of the frame at double slot
            // FM_RB0 + t, the previous frame's last 26 in front of them -- every window is one contiguous run
            {
                // (the last frame asks for itself again: a request under a branch would make every later wait the minimum over both paths)
                if (n == 9600) fm_first2_request<9600, F32IN>(pre, raw + (f + 1 < a.nframes ? t0 + n : t0), rawf + (f + 1 < a.nframes ? t0 + n : t0), tf);
                const double *Rb = reinterpret_cast<const double *>(smem);
                long long jlo = (t0 - a.first_out + D - 1) / D;
                if (t0 <= a.first_out) jlo = 0;
                const bool even_d = (D & 1) == 0;                 // then every window of the call ends on the same parity (n is even)
                const int par = (int)((a.first_out - t0) & 1);
                for (long long j = jlo + tf;; j += FM_T) {
                    const long long te = (long long)a.first_out + (long long)D * j;  // window end, call-relative
                    if (te >= t0 + n || j >= a.nds) break;
                    const double2 cs = a.vco_cs[j];
                    const int e = (int)(te - t0);  // 0..n-1 within the frame
                    double fi = 0.0;
                    if (even_d) {
                        // the 27 samples e-26 .. e as 14 aligned 16-byte reads: d[i] = slot ((e + 6) & ~1) + i
                        const double2 *w2 = reinterpret_cast<const double2 *>(Rb + ((e + FM_RB0 - 26) & ~1));
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
                        const double *w = Rb + (FM_RB0 + e);
#pragma unroll
                        for (int k = 0; k < 27; k++) fi += w[-k] * ds_tap(k);
                    }
                    const double o = fi * HOWARD;  // fi == fq: both rails get the same samples
                    dm[64 + j] = make_double2(o * cs.x, o * cs.y);  // :515-516
                }
                if (tf < 26) hist[tf] = Rb[FM_RB0 + n - 26 + tf];  // (nobody reads hist[] before the next frame's last pass)
                __syncthreads();  // every window is read before the next frame's first pass overwrites the image
            }
            PHASE(5)
            continue;
        } else {
            double2 keep = make_double2(0.0, 0.0);
            if (tf < 204) keep = X[centreBin - 102 + tf];
            __syncthreads();
            for (int i = tf; i < n; i += FM_T) X[i] = make_double2(0.0, -0.0);  // conj of the zeroed array: -0.0 imaginary parts
            __syncthreads();
            if (tf < 204) X[tf] = make_double2(keep.x, -keep.y);
            __syncthreads();
            PHASE(3)
            fm_forward<true>(XL, twL, aa, tf, false);
            for (int i = tf; i < n; i += FM_T) X[i].x = X[i].x * norm;  // re = X.x / n (:462)
            __syncthreads();
        }
        PHASE(4)
        if (tf < 26) hist[26 + tf] = X[tf].x;
        __syncthreads();
        // ---- RxDownSample(re, re) (:461-463, :470-492): outputs whose window ends inside this frame
        {
            long long jlo = (t0 - a.first_out + D - 1) / D;
            if (t0 <= a.first_out) jlo = 0;
            for (long long j = jlo + tf;; j += FM_T) {
                const long long te = (long long)a.first_out + (long long)D * j;  // window end, call-relative
                if (te >= t0 + n || j >= a.nds) break;
                const double2 cs = a.vco_cs[j];
                const int e = (int)(te - t0);  // 0..n-1 within the frame
                double fi = 0.0;
                if (e >= 26) {  // all but the first three windows of a frame: no history, constant offsets
                    const double2 *w = X + e;
#pragma unroll
                    for (int k = 0; k < 27; k++) fi += w[-k].x * ds_tap(k);  // newest first (:479-483)
                } else {
                    const double *w = hist + 26 + e;  // the first three windows of a frame: history, then the frame's head
#pragma unroll
                    for (int k = 0; k < 27; k++) fi += w[-k] * ds_tap(k);
                }
                const double o = fi * HOWARD;  // fi == fq: both rails get the same samples
                dm[64 + j] = make_double2(o * cs.x, o * cs.y);  // :515-516
            }
        }
        double hnew = 0.0;
        if (tf < 26) hnew = X[n - 26 + tf].x;
        __syncthreads();
        if (tf < 26) hist[tf] = hnew;
        __syncthreads();
        PHASE(5)
    }
#undef PHASE
    if (tid < 26) sp->hist[tid] = hist[tid];
    if (tid == 0) {
        sp->avePeakPower = avePeakPower;
        sp->aveCentreBin = aveCentreBin;
        sp->centreBin = centreBin;
        if (timing)
            for (int k = 0; k < 8; k++) a.phase_clk[k] = clk[k];
    }
}

// ================================================================================== two frames of a stream at once (round 6)
// n = 4800 (the reference's default frame at 48 kHz) ran 30 % slower per sample than n = 9600: the same passes with half the work
// items -- a pass pair of one frame has 300 or 320 items for 768 threads -- and the same ~900-tick LDS round trip each.  Here a
// workgroup takes frames f and f + 1 of its stream TOGETHER: two images side by side in LDS (2 x 76.8 KB; the tables beyond the
// first 276 entries come from L2 as the 9600 frame's do), every pass over both (the NIMG forms above), |X| / boxcar / argmax for
// both, then the centre-bin rule (:444-453) for f and, with its result, for f + 1 -- the only sequential step -- the inverse
// halves with each frame's own centre bin, and RxDownSample: frame f + 1's windows reach back into frame f's samples, which lie in
// the first image.  A call's odd last frame runs as a pair whose second half is computed and dropped.  Same operations on the same
// operands per frame as k_front_fftm.
template <int NN, bool F32IN>
__global__ __launch_bounds__(FM_T) void k_front_fftm2(FftmArgs aa)
{
    static_assert(NN == 4800 || NN == 4410, "frames of 4800 samples (48 kHz) or 4410 (44.1 kHz: passes [2,3] [3,5] [7] [7])");
    extern __shared__ __align__(16) unsigned char smem[];
    const FftFrontArgs &a = aa.f;
    constexpr int n = NN;
    double2 *X = reinterpret_cast<double2 *>(smem);  // [2 n]
    const LdsArr XL = lds_arr(smem);
    double *hist = reinterpret_cast<double *>(X + 2 * n);    // [32]
    double *redv = hist + 32;                                // [2][16] per-image, per-wave best value
    int *redi = reinterpret_cast<int *>(redv + 32);          // [2][16]
    double2 *twL = reinterpret_cast<double2 *>(redi + 32);   // [lds_tw]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < aa.lds_tw; i += FM_T) twL[i] = aa.f.tw[i];
    const int s = blockIdx.x;
    const int beg = a.do_up ? n / 4 : 0;
    const int end = a.do_up ? n / 2 : n / 4;
    const int pbase = beg + 24;
    const int abase = beg + 74;
    constexpr int POFF = 2 * (NN / 2 + 104);  // |X| over [beg + 24, end - 24) in the dead upper part of each image (double slots)
    constexpr int AOFF = POFF + (NN / 4 - 48);
    FftFrontState *sp = &a.st[s];
    if (tid < 26) hist[tid] = sp->hist[tid];
    double avePeakPower = sp->avePeakPower, aveCentreBin = sp->aveCentreBin;
    int centreBin = sp->centreBin;
    const double CFREQ_INV = (double)(1.0F - (2.0F / (1 + 1))), CFREQ_AVG = (double)(2.0F / (1 + 1));
    const double PSD_INV = (double)(1.0F - (2.0F / (10 + 1))), PSD_AVG = (double)(2.0F / (10 + 1));
    const double HOWARD = 0.9 * 32768.0;
    const int D = a.decim;
    const double norm = 1.0 / (double)n;
    const int *__restrict__ raw = a.raw + (long long)s * a.stride_pairs;
    const float2 *__restrict__ rawf = a.rawf + (long long)s * a.stride_pairs;
    double2 *dm = a.dm + (long long)s * a.dm_stride;
    const GblArr g = gbl_arr(a.tw);
    // diagnostics (JSDR_FFT_PHASECLK=1): thread 0 of stream 0 accumulates the clock ticks of every phase (k_front_fftm's slots)
    long long *clk = reinterpret_cast<long long *>(twL + aa.lds_tw);  // the 64 spare bytes behind the tables
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
    __syncthreads();
    if (timing) tprev = (long long)clock64();
    // the next pair's samples are requested at the top of RxDownSample and found in registers here (see fm_first2_request)
    FmRaw16 pre;
    auto request = [&](long long t0q, bool two_q, int t_) {
        if constexpr (NN == 4800) {
            fm_first_request2<NN, F32IN>(pre, raw + t0q, rawf + t0q, two_q ? NN : 0, t_);
        } else {
            constexpr int NLD = (2 * NN + FM_T - 1) / FM_T;
            static_assert(NLD <= 16, "sixteen samples a thread");
#pragma unroll
            for (int q = 0; q < NLD; q++) {
                int t = t_ + q * FM_T;
                t = t < 2 * NN ? t : 2 * NN - 1;
                if (!two_q && t >= NN) t -= NN;  // (a single last frame: the second image is a copy)
                if (F32IN)
                    pre.wf[q] = reinterpret_cast<const f2v *>(rawf)[t0q + t];
                else
                    pre.w[q] = raw[t0q + t];
            }
        }
    };
    request(0, a.nframes >= 2, tid);

    for (int f = 0; f < a.nframes; f += 2) {
        const bool two = f + 1 < a.nframes;
        int tf = tid;
        asm volatile("" : "+v"(tf));  // (opaque per iteration: see k_front_fftm)
        const long long t0 = (long long)f * n;  // call-relative index of frame f's first sample
        // ---- both frames -> images (:416-421), forward transform (:422-423), the last pass for the bins below end + 102 only
        if constexpr (NN == 4800) {
            fm_first_from_regs2<NN, F32IN>(XL, pre, a.ic, a.qc, tf);  // (with the first pass; a single last frame comes twice)
            const LdsArr t = lds_arr(twL);
            fm_pass2<4, 4, NN, 4, false, LdsArr, LdsArr, 2>(XL, t + 4, t + 20, NN, 4, 0u, tf);
            fm_pass2<3, 5, NN, 64, false, LdsArr, GblArr, 2>(XL, t + 84, g + 276, NN, 64, 0u, tf);
            fm_pass5_band<NN, 960, false, GblArr, 2>(XL, g + 1236, end + 102, tf);
        } else {
            // natural order; frame f + 1 follows frame f in memory as image 1 follows image 0
            constexpr int NLD = (2 * NN + FM_T - 1) / FM_T;
#pragma unroll
            for (int q = 0; q < NLD; q++) {
                const int t = tf + q * FM_T;
                if (t < 2 * n) {
                    double di, dq;
                    if (F32IN) {
                        di = (double)pre.wf[q].x;
                        dq = (double)pre.wf[q].y;
                    } else {
                        di = (double)i16_to_float_java(java_short_add((int)(short)(pre.w[q] & 0xffff), a.ic));
                        dq = (double)i16_to_float_java(java_short_add(pre.w[q] >> 16, a.qc));
                    }
                    X[t] = make_double2(di, dq);
                }
            }
            __syncthreads();
            const LdsArr t = lds_arr(twL);
            fm_pass2<2, 3, NN, 1, false, LdsArr, LdsArr, 2>(XL, t, t + 2, NN, 1, 0u, tf);
            fm_pass2<3, 5, NN, 6, false, LdsArr, LdsArr, 2>(XL, t + 8, t + 26, NN, 6, 0u, tf);
            fm_pass<7, NN, 90, LdsArr, 2>(XL, t + 116, NN, 90, 0u, tf);
            fm_passr_band<7, NN, 630, GblArr, 2>(XL, g + 746, end + 102, tf);
        }
        PHASE(1)  // (with the load: the first pass comes straight from the samples)
        // ---- |X| (:425-427) over the band the boxcar reads, both images
        {
            const int cnt = end - 24 - pbase;
            for (int u = tf; u < 2 * cnt; u += FM_T) {
                const int img = u >= cnt ? 1 : 0;
                const int i = pbase + (u - img * cnt);
                const double2 v = X[img * n + i];
                reinterpret_cast<double *>(X + img * n)[POFF + (i - pbase)] = sqrt(v.x * v.x + v.y * v.y);
            }
        }
        __syncthreads();
        PHASE(6)
        // ---- boxcar + first maximum per image (:433-442): a thread's items of one image ascend
        double bestv[2] = {0.0, 0.0};
        int besti[2] = {-1, -1};
        {
            const int npair = (end - 75 - (beg + 74) + 1) / 2;
            for (int u = tf; u < 2 * npair; u += FM_T) {
                const int img = u >= npair ? 1 : 0;
                const int i = beg + 74 + 2 * (u - img * npair);
                double *Pi = reinterpret_cast<double *>(X + img * n) + POFF;
                double *Ai = reinterpret_cast<double *>(X + img * n) + AOFF;
                const double2 *w = reinterpret_cast<const double2 *>(Pi + (i - 50 - pbase));
                double a0, a1;
                boxcar_pair(w, a0, a1);
                asm volatile("" : "+v"(a0), "+v"(a1));
                double bv = img ? bestv[1] : bestv[0];
                int bi = img ? besti[1] : besti[0];
                if (i >= beg + 75) {
                    Ai[i - abase] = a0;
                    if (bv < a0) {
                        bv = a0;
                        bi = i;
                    }
                }
                if (i + 1 < end - 75) {
                    Ai[i + 1 - abase] = a1;
                    if (bv < a1) {
                        bv = a1;
                        bi = i + 1;
                    }
                }
                if (img) {
                    bestv[1] = bv;
                    besti[1] = bi;
                } else {
                    bestv[0] = bv;
                    besti[0] = bi;
                }
            }
        }
#pragma unroll
        for (int im = 0; im < 2; im++) {
            wave_first_max(bestv[im], besti[im]);
            if (lane == 0) {
                redv[im * 16 + wave] = bestv[im];
                redi[im * 16 + wave] = besti[im];
            }
        }
        __syncthreads();
        PHASE(7)
        // ---- centre-bin rule (:444-453) for frame f, then for frame f + 1, by every thread on the same values
        int cb[2];
#pragma unroll
        for (int im = 0; im < 2; im++) {
            double mv = 0.0;
            int mi = -1;
            if (lane < FM_T / 64) {
                mv = redv[im * 16 + lane];
                mi = redi[im * 16 + lane];
            }
            wave_first_max(mv, mi);
            const double maxBin = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(mv)),
                                                   __builtin_amdgcn_readfirstlane(__double2loint(mv)));
            const int binPos = __builtin_amdgcn_readfirstlane(mi);
            if (im == 0 || two) {
                if (centreBin < 0) centreBin = 0;
                if (centreBin > end - 1) centreBin = end - 1;
                const double *Ai = reinterpret_cast<const double *>(X + im * n) + AOFF;
                const double atc = (centreBin >= beg + 75 && centreBin < end - 75) ? Ai[centreBin - abase] : 0.0;
                avePeakPower = (PSD_AVG * atc) + (PSD_INV * avePeakPower);
                if (maxBin > (avePeakPower / 4) * 5 && binPos > 0) {
                    aveCentreBin = (CFREQ_AVG * (double)(float)binPos) + (CFREQ_INV * aveCentreBin);
                    centreBin = (int)(aveCentreBin + (double)1.0F);
                }
                if (centreBin < 102) centreBin = 102;
            }
            cb[im] = centreBin;
        }
        PHASE(2)
        // ---- 204 bins around each frame's centre to bin 0 of a zeroed array (:458), inverse transform (:459): passes 1-3 from the
        // bins, [3, 5], the last pass real parts only, scaled, compact
        if constexpr (NN == 4800) {
            const LdsArr t64 = lds_arr(twL) + 20;
            fm_inv_blocks<NN, true, LdsArr, LdsArr, 2>(XL, XL + (cb[0] - 102), t64, 1, t64, 2, tf, n + cb[1] - cb[0]);
            fm_pass2<3, 5, NN, 64, false, LdsArr, GblArr, 2>(XL, lds_arr(twL) + 84, g + 276, NN, 64, 0u, tf);
            fm_pass5_real<NN, 960, false, 1, GblArr, 2>(XL, g + 1236, norm, tf, hist);
        } else {
            const LdsArr t = lds_arr(twL);
            fm_inv_pair_from_bins<2, 3, NN, LdsArr, 2>(XL, XL + (cb[0] - 102), t + 2, tf, n + cb[1] - cb[0]);
            fm_pass2<3, 5, NN, 6, false, LdsArr, LdsArr, 2>(XL, t + 8, t + 26, NN, 6, 0u, tf);
            fm_pass<7, NN, 90, LdsArr, 2>(XL, t + 116, NN, 90, 0u, tf);
            fm_passr_real<7, NN, 630, 1, GblArr, 2>(XL, g + 746, norm, tf, hist);
        }
        PHASE(4)
        double *Rb0 = reinterpret_cast<double *>(smem);
        double *Rb1 = Rb0 + 2 * n;
        // (frame f + 1's windows reach back into frame f's last 26 samples)
        if (tf < 26) Rb1[FM_RB0 - 26 + tf] = Rb0[FM_RB0 + n - 26 + tf];
        __syncthreads();
        // ---- RxDownSample(re, re) (:461-463, :470-492) from the compact samples of frame f, then of frame f + 1
        {
            // (the last pair asks for itself again: a request under a branch would make every later wait the minimum over both paths)
            const bool more = f + 2 < a.nframes;
            request(more ? t0 + 2 * (long long)n : t0, more ? f + 3 < a.nframes : two, tf);
        }
#pragma unroll
        for (int im = 0; im < 2; im++) {
            if (im == 1 && !two) break;
            const double *Rb = im ? Rb1 : Rb0;
            const long long tq = t0 + (long long)im * n;
            long long jlo = (tq - a.first_out + D - 1) / D;
            if (tq <= a.first_out) jlo = 0;
            const bool even_d = (D & 1) == 0;
            const int par = (int)((a.first_out - tq) & 1);
            for (long long j = jlo + tf;; j += FM_T) {
                const long long te = (long long)a.first_out + (long long)D * j;  // window end, call-relative
                if (te >= tq + n || j >= a.nds) break;
                const double2 cs = a.vco_cs[j];
                const int e = (int)(te - tq);
                double fi = 0.0;
                if (even_d) {
                    const double2 *w2 = reinterpret_cast<const double2 *>(Rb + ((e + FM_RB0 - 26) & ~1));
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
                    const double *w = Rb + (FM_RB0 + e);
#pragma unroll
                    for (int k = 0; k < 27; k++) fi += w[-k] * ds_tap(k);
                }
                const double o = fi * HOWARD;
                dm[64 + j] = make_double2(o * cs.x, o * cs.y);  // :515-516
            }
        }
        if (tf < 26) hist[tf] = (two ? Rb1 : Rb0)[FM_RB0 + n - 26 + tf];
        __syncthreads();
        PHASE(5)
    }
#undef PHASE
    if (tid < 26) sp->hist[tid] = hist[tid];
    if (tid == 0) {
        sp->avePeakPower = avePeakPower;
        sp->aveCentreBin = aveCentreBin;
        sp->centreBin = centreBin;
        if (timing)
            for (int k = 0; k < 8; k++) a.phase_clk[k] = clk[k];
    }
}

// ================================================================================== the default frames in three phases (round 6)
// k_front_fftm's frame loop cut where the reference's loop carries nothing (bpsk_acq.hip has the whole story): k_acqm_fwd is
// its forward half for ONE (stream, frame) -- transform, |X|, boxcar, first maximum -- and leaves what the scan (k_acq_scan) and
// the inverse half need; k_acqm_inv gathers the frame's 204 bins from that row, runs the inverse half and RxDownSample for the
// windows inside the frame (k_acq_edges does the ones that reach into the frame before).  Same passes, same tables, same
// image -- so a workgroup still owns a CU -- but the grid is frames: a call of few streams fills the chip (one recorded stream
// of 109 frames took 109 frame times in k_front_fftm and takes one here).  Frames of 9600 / 4800 / 4410 samples (the compact
// passes); a workgroup takes its frames one at a time from a ticket counter.
__device__ __forceinline__ long long acqm_ticket(unsigned *ctr, int *tkL, int tid)
{
    if (tid == 0) tkL[0] = (int)atomicAdd(ctr, 1u);
    __syncthreads();
    const long long g = tkL[0];
    __syncthreads();
    return g;
}

template <bool F32IN>
__global__ __launch_bounds__(FM_T) void k_acqm_fwd(FftmArgs aa, AcqArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int n = a.n;
    double2 *X = reinterpret_cast<double2 *>(smem);  // [n]
    const LdsArr XL = lds_arr(smem);
    double *hist = reinterpret_cast<double *>(X + n);  // [32] (unused here; the layout is k_front_fftm's)
    double *redv = hist + 32;                          // [16]
    int *redi = reinterpret_cast<int *>(redv + 16);    // [16]
    double2 *twL = reinterpret_cast<double2 *>(redi + 16);
    int *tkL = reinterpret_cast<int *>(twL + aa.lds_tw);  // (the 64 spare bytes behind the tables)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < aa.lds_tw; i += FM_T) twL[i] = aa.f.tw[i];
    const int beg = a.do_up ? n / 4 : 0;
    const int end = a.do_up ? n / 2 : n / 4;
    const int pbase = beg + 24;
    double *P = reinterpret_cast<double *>(X + (n / 2 + 104));  // (the sums go straight to the frame's row in global memory)
    const long long nfr = (long long)a.S * a.F;
    __syncthreads();
    for (;;) {
        const long long g = acqm_ticket(a.tickets + 0, tkL, tid);
        if (g >= nfr) break;
        int tf = tid;
        asm volatile("" : "+v"(tf));
        const int s = (int)(g / a.F), f = (int)(g - (long long)s * a.F);
        const long long t0 = (long long)s * a.stride_pairs + (long long)(a.f0 + f) * n;
        if (n == 9600)
            fm_first2_from_raw<9600, F32IN>(XL, a.raw + t0, a.rawf + t0, a.ic, a.qc, lds_arr(twL) + 4, tf);
        else if (n == 4800)
            fm_first_from_raw<4800, F32IN>(XL, a.raw + t0, a.rawf + t0, a.ic, a.qc, tf);
        else {
            // n = 4410: the frame to LDS in natural order (:416-421), all of a thread's samples in flight before the first conversion
            constexpr int NLD = (4410 + FM_T - 1) / FM_T;
            int w[NLD];
            float2 wf[NLD];
#pragma unroll
            for (int q = 0; q < NLD; q++) {
                int t = tf + q * FM_T;
                t = t < n ? t : n - 1;
                if (F32IN)
                    wf[q] = a.rawf[t0 + t];
                else
                    w[q] = a.raw[t0 + t];
            }
#pragma unroll
            for (int q = 0; q < NLD; q++) {
                const int t = tf + q * FM_T;
                if (t < n) {
                    double di, dq;
                    if (F32IN) {
                        di = (double)wf[q].x;
                        dq = (double)wf[q].y;
                    } else {
                        di = (double)i16_to_float_java(java_short_add((int)(short)(w[q] & 0xffff), a.ic));
                        dq = (double)i16_to_float_java(java_short_add(w[q] >> 16, a.qc));
                    }
                    X[t] = make_double2(di, dq);
                }
            }
            __syncthreads();
        }
        fm_forward<true>(XL, twL, aa, tf, n != 4410, FM_FWD_BAND, end + 102);  // :422-423; bins < end + 102 are formed
        // ---- the bins a gather can reach, to the frame's row (layout: acq_spec_index), and |X| over the band the boxcar reads
        {
            double2 *specg = a.spec + g * a.nsb;
            const int lo1 = a.do_up ? n / 4 - 26 : 0;
            for (int i = tf; i < a.nsb; i += FM_T) {
                const int b = a.do_up ? (i < 204 ? i : lo1 + (i - 204)) : i;
                specg[i] = X[b];
            }
        }
        for (int i = pbase + tf; i < end - 24; i += FM_T) {
            const double2 v = X[i];
            P[i - pbase] = sqrt(v.x * v.x + v.y * v.y);  // :425-427
        }
        __syncthreads();
        // ---- 100-wide boxcar, summed j ascending for every i (:433-437); first maximum (:439-442)
        double bestv = 0.0;
        int besti = -1;
        double *ab = a.aband + g * a.na;
        for (int i = beg + 74 + 2 * tf; i < end - 75; i += 2 * FM_T) {
            const double2 *w = reinterpret_cast<const double2 *>(P + (i - 50 - pbase));
            double a0, a1;
            boxcar_pair(w, a0, a1);
            asm volatile("" : "+v"(a0), "+v"(a1));
            if (i >= beg + 75) {
                ab[i - (beg + 75)] = a0;
                if (bestv < a0) {
                    bestv = a0;
                    besti = i;
                }
            }
            if (i + 1 < end - 75) {
                ab[i + 1 - (beg + 75)] = a1;
                if (bestv < a1) {
                    bestv = a1;
                    besti = i + 1;
                }
            }
        }
        wave_first_max(bestv, besti);
        if (lane == 0) {
            redv[wave] = bestv;
            redi[wave] = besti;
        }
        __syncthreads();
        if (tid == 0) {
            double mv = 0.0;
            int mi = -1;
            for (int w = 0; w < FM_T / 64; w++) {
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
        __syncthreads();
    }
}

__global__ __launch_bounds__(FM_T) void k_acqm_inv(FftmArgs aa, AcqArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int n = a.n;
    double2 *X = reinterpret_cast<double2 *>(smem);
    const LdsArr XL = lds_arr(smem);
    double *hist = reinterpret_cast<double *>(X + n);  // [32]: zeros (the windows that reach before the frame are k_acq_edges')
    double *redv = hist + 32;
    int *redi = reinterpret_cast<int *>(redv + 16);
    double2 *twL = reinterpret_cast<double2 *>(redi + 16);
    int *tkL = reinterpret_cast<int *>(twL + aa.lds_tw);
    const int tid = threadIdx.x;
    for (int i = tid; i < aa.lds_tw; i += FM_T) twL[i] = aa.f.tw[i];
    if (tid < 32) hist[tid] = 0.0;
    const int D = a.decim;
    const double norm = 1.0 / (double)n;
    const double HOWARD = 0.9 * 32768.0;
    const int lo1 = a.do_up ? n / 4 - 26 : 0;
    const long long nfr = (long long)a.S * a.F;
    __syncthreads();
    for (;;) {
        const long long g = acqm_ticket(a.tickets + 1, tkL, tid);
        if (g >= nfr) break;
        int tf = tid;
        asm volatile("" : "+v"(tf));
        const int s = (int)(g / a.F), f = (int)(g - (long long)s * a.F);
        const long long t0 = (long long)(a.f0 + f) * n;  // call-relative index of the frame's first sample
        // ---- the 204 bins around the centre (:458) to the head of the image
        {
            const int c = a.cbin[g];
            int off = c - 102;
            if (a.do_up) off = (c == 102) ? 0 : 204 + (c - 102 - lo1);
            if (off < 0) off = 0;
            if (off + 204 > a.nsb) off = a.nsb - 204;
            const double2 *src = a.spec + g * a.nsb + off;
            if (tf < 204) X[tf] = src[tf];
        }
        __syncthreads();
        // ---- inverse transform (:459) as conj o forward o conj: the first passes straight from the bins, the last one real parts
        // only, scaled, compact (k_front_fftm's)
        {
            const LdsArr t64 = lds_arr(twL) + 20;
            if (n == 9600)
                fm_inv_blocks128_9600<true>(XL, XL, t64, lds_arr(twL) + 84, tf);
            else if (n == 4800)
                fm_inv_blocks<4800, true>(XL, XL, t64, 1, t64, 2, tf);
            else
                fm_inv_pair_from_bins<2, 3, 4410>(XL, XL, lds_arr(twL) + 2, tf);
        }
        fm_forward<true>(XL, twL, aa, tf, true, FM_INV_REAL, 0, norm, hist);
        // ---- the frame's first and last 26 samples (k_acq_edges), RxDownSample for the windows inside the frame (:461-463, :470-492)
        {
            const double *Rb = reinterpret_cast<const double *>(smem);
            if (tf < 26) {
                double *eg = a.edges + g * 52;
                eg[tf] = Rb[FM_RB0 + tf];
                eg[26 + tf] = Rb[FM_RB0 + n - 26 + tf];
            }
            long long jlo = (t0 - a.first_out + D - 1) / D;
            if (t0 <= a.first_out) jlo = 0;
            const bool even_d = (D & 1) == 0;
            const int par = (int)((a.first_out - t0) & 1);
            for (long long j = jlo + tf;; j += FM_T) {
                const long long te = (long long)a.first_out + (long long)D * j;  // window end, call-relative
                if (te >= t0 + n || j >= a.nds) break;
                const int e = (int)(te - t0);
                if (e < 26) continue;
                const double2 cs = a.vco_cs[j];
                double fi = 0.0;
                if (even_d) {
                    const double2 *w2 = reinterpret_cast<const double2 *>(Rb + ((e + FM_RB0 - 26) & ~1));
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
                    const double *w = Rb + (FM_RB0 + e);
#pragma unroll
                    for (int k = 0; k < 27; k++) fi += w[-k] * ds_tap(k);
                }
                const double o = fi * HOWARD;
                a.dm[(long long)s * a.dm_stride + 64 + j] = make_double2(o * cs.x, o * cs.y);  // :515-516
            }
        }
        __syncthreads();
    }
}

// ============================================================================================== frames of 2 m samples
// n = 19200 (192 kHz with java-sdr's default buffer, JavaAudio.java:58-59: the FUNcube Dongle Pro+ frame): 307 KB as
// double2, twice a workgroup's LDS.  The oracle's transform for n > 9600 starts with ONE radix-2 pass
// (jo_fft_mixed_radices), which splits it into two INDEPENDENT m = n/2 point halves: half c (0 / 1) holds, pass after
// pass, the elements of parity c, and ends as the even / odd output bins.  In half-array coordinates every later
// pass is the m-point Stockham pass; only its twiddles differ: T_{2P'r}[(2k'+c) j] instead of T_{P'r}[k' j] -- for c = 0
// the same numbers (the m-point tables; the long-double arguments scale by an exact 2), for c = 1 a table of its own.
// So a frame is four m-point transforms in the one LDS image (forward c = 0, 1, inverse c = 0, 1); what has to
// outlive the image goes through a per-stream scratch in global memory (L2 sized: 154 KB):
//   ek : the even bins the 204-bin gather may need (bins below n/2 + 102) -- written after the forward c = 0 half
//   r0 : the even output samples re * (1/n) of the inverse -- read by RxDownSample beside the odd ones in LDS
// Same butterflies, same tables, same order as the oracle's fft_f64_mixed with radices 2,4,4,4,2,3,5,5: bit-identical
// centre bins, traces, bits.  Not tuned: single passes for the c = 1 halves, twiddles from L2.
struct Fft2xArgs {
    FftmArgs sub;             // the m-point plan (sub.f.n = m) and everything else of the front end
    int n;                    // 2 m
    int tw1_off[FM_MAXPASS];  // c = 1 tables, [(j-1) P' + k'] = T_{2 P' r}[(2 k' + 1) j], behind the m-point tables in sub.f.tw
    double2 *ek;              // [S][ek_stride]
    double *r0;               // [S][r0_stride]
    long long ek_stride, r0_stride;
};

// one Stockham pass with a two-dimensional twiddle table tw[(j-1) P + k], every input j >= 1 multiplied (also at P = 1)
template <int R>
__device__ __attribute__((noinline)) void fm_pass_t(LdsArr X, GblArr tw, int n, int P, unsigned pmagic, int tid)
{
    constexpr int ITERS = ((FM_NMAX / R) + FM_T - 1) / FM_T;
    const int nb = n / R;
    double2 v[ITERS][R];
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int b = it * FM_T + tid;
        if (b < nb) {
            const int k = (P == 1) ? 0 : b - (int)__umulhi((unsigned)b, pmagic) * P;
#pragma unroll
            for (int j = 0; j < R; j++) {
                v[it][j] = X[b + j * nb];
                if (j >= 1) v[it][j] = cdmul(v[it][j], tw[(j - 1) * P + k]);
            }
            dft_r<R>(v[it]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int b = it * FM_T + tid;
        if (b < nb) {
            const int k = (P == 1) ? 0 : b - (int)__umulhi((unsigned)b, pmagic) * P;
            const int j0 = (b - k) * R + k;
#pragma unroll
            for (int q = 0; q < R; q++) X[j0 + q * P] = v[it][q];
        }
    }
    __syncthreads();
}

// two consecutive passes of an odd half in one LDS round trip: fm_pass2 with the two-dimensional tables
template <int R1, int R2, int NN, int PP, bool SWZ_IN = false>
__device__ __attribute__((noinline)) void fm_pass2_t(LdsArr X, GblArr tw1, GblArr tw2, int tid)
{
    constexpr int RR = R1 * R2;
    constexpr int ITERS = ((NN / RR) + FM_T - 1) / FM_T;
    constexpr int ng = NN / RR, nb1 = NN / R1, P = PP, P2 = PP * R1;
    double2 v[ITERS][R2][R1];
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int g = it * FM_T + tid;
        if (g < ng) {
            const int k1 = g % PP;
#pragma unroll
            for (int j2 = 0; j2 < R2; j2++) {
                const int b1 = g + j2 * ng;
#pragma unroll
                for (int j1 = 0; j1 < R1; j1++) {
                    v[it][j2][j1] = X[SWZ_IN ? fm_swz(b1 + j1 * nb1) : b1 + j1 * nb1];
                    if (j1 >= 1) v[it][j2][j1] = cdmul(v[it][j2][j1], tw1[(j1 - 1) * P + k1]);
                }
                dft_r<R1>(v[it][j2]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q1 = 0; q1 < R1; q1++) {
                const int k2 = k1 + q1 * P;
                double2 w[R2];
#pragma unroll
                for (int j2 = 0; j2 < R2; j2++) {
                    w[j2] = v[it][j2][q1];
                    if (j2 >= 1) w[j2] = cdmul(w[j2], tw2[(j2 - 1) * P2 + k2]);
                }
                dft_r<R2>(w);
#pragma unroll
                for (int q2 = 0; q2 < R2; q2++) v[it][q2][q1] = w[q2];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int g = it * FM_T + tid;
        if (g < ng) {
            const int k1 = g % PP;
            const LdsArr z = X + ((g - k1) * RR + k1);
#pragma unroll
            for (int q2 = 0; q2 < R2; q2++)
#pragma unroll
                for (int q1 = 0; q1 < R1; q1++) z[q1 * P + q2 * P * R1] = v[it][q2][q1];
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void fm_forward_odd(LdsArr X, const Fft2xArgs &a, int tid, int mode = FM_FULL, int need_end = 0,
                                               double norm = 1.0, bool first2_done = false)
{
    const FftmArgs &m = a.sub;
    if (m.f.n == 9600) {  // the 192 kHz default: the 9600-point plan 4 | 4,4 | 2,3 | 5 | 5 with pass pairs, as fm_forward
        const GblArr t = gbl_arr(m.f.tw);
        if (mode == FM_INV_REAL && first2_done) {  // after fm_inv_blocks128_9600 with the odd tables; compact real samples
            fm_pass2_t<3, 5, 9600, 128>(X, t + a.tw1_off[4], t + a.tw1_off[5], tid);
            fm_pass5_real<9600, 1920, true, 1>(X, t + a.tw1_off[6], norm, tid);
            return;
        }
        if (mode == FM_FWD_BAND && first2_done) {  // after fm_first2x_from_raw<.., ODD>: 4,2 | 3,5 | 5 as fm_forward
            fm_pass2_t<4, 2, 9600, 16, true>(X, t + a.tw1_off[2], t + a.tw1_off[3], tid);
            fm_pass2_t<3, 5, 9600, 128>(X, t + a.tw1_off[4], t + a.tw1_off[5], tid);
            fm_pass5_band<9600, 1920, true>(X, t + a.tw1_off[6], need_end, tid);
            return;
        }
        if (mode != FM_INV_REAL) {
            fm_pass_t<4>(X, t + a.tw1_off[0], 9600, 1, 0u, tid);
            fm_pass2_t<4, 4, 9600, 4>(X, t + a.tw1_off[1], t + a.tw1_off[2], tid);
        }
        fm_pass2_t<2, 3, 9600, 64>(X, t + a.tw1_off[3], t + a.tw1_off[4], tid);
        fm_pass_t<5>(X, t + a.tw1_off[5], 9600, 384, m.pmagic[5], tid);
        if (mode == FM_FWD_BAND)
            fm_pass5_band<9600, 1920, true>(X, t + a.tw1_off[6], need_end, tid);
        else if (mode == FM_INV_REAL)
            fm_pass5_real<9600, 1920, true>(X, t + a.tw1_off[6], norm, tid);
        else
            fm_pass_t<5>(X, t + a.tw1_off[6], 9600, 1920, m.pmagic[6], tid);
        return;
    }
    int P = 1;
    for (int p = 0; p < m.np; p++) {
        const int r = m.rad[p];
        const GblArr tw = gbl_arr(m.f.tw + a.tw1_off[p]);
        if (r == 4)
            fm_pass_t<4>(X, tw, m.f.n, P, m.pmagic[p], tid);
        else if (r == 2)
            fm_pass_t<2>(X, tw, m.f.n, P, m.pmagic[p], tid);
        else if (r == 3)
            fm_pass_t<3>(X, tw, m.f.n, P, m.pmagic[p], tid);
        else
            fm_pass_t<5>(X, tw, m.f.n, P, m.pmagic[p], tid);
        P *= r;
    }
}

template <bool F32IN>
__global__ __launch_bounds__(FM_T) void k_front_fft2x(Fft2xArgs aa)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const FftFrontArgs &a = aa.sub.f;
    const int n = aa.n, m = n / 2;
    double2 *X = reinterpret_cast<double2 *>(smem);      // [m]: one half at a time
    const LdsArr XL = lds_arr(smem);                     // the same image for the passes (typed view)
    double *hist = reinterpret_cast<double *>(X + m);    // [64]: [0,26) the previous frame's last 26 scaled samples
    double *redv = hist + 64;
    int *redi = reinterpret_cast<int *>(redv + 16);
    double2 *zb = reinterpret_cast<double2 *>(redi + 16);  // [204] the gathered bins, conjugated (default frame only)
    const bool pruned = (m == 9600);  // the 192 kHz default: pruned passes (fm_inv_blocks, fm_pass5_band / _real)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = blockIdx.x;
    const int beg = a.do_up ? n / 4 : 0;
    const int end = a.do_up ? n / 2 : n / 4;
    const int pbase = beg + 24, abase = beg + 74;
    // |X| and the boxcar sums live behind the odd bins the gather may still need (bins < n/2 + 102 <=> slot < m/2 + 51)
    double *P = reinterpret_cast<double *>(X + (m / 2 + 56));
    double *A = P + (n / 4 - 48);
    FftFrontState *sp = &a.st[s];
    if (tid < 26) hist[tid] = sp->hist[tid];
    double avePeakPower = sp->avePeakPower, aveCentreBin = sp->aveCentreBin;
    int centreBin = sp->centreBin;
    const double CFREQ_INV = (double)(1.0F - (2.0F / (1 + 1))), CFREQ_AVG = (double)(2.0F / (1 + 1));
    const double PSD_INV = (double)(1.0F - (2.0F / (10 + 1))), PSD_AVG = (double)(2.0F / (10 + 1));
    const double HOWARD = 0.9 * 32768.0;
    const int D = a.decim;
    const double norm = 1.0 / (double)n;
    const int *__restrict__ raw = a.raw + (long long)s * a.stride_pairs;
    const float2 *__restrict__ rawf = a.rawf + (long long)s * a.stride_pairs;
    double2 *dm = a.dm + (long long)s * a.dm_stride;
    double2 *ek = aa.ek + (long long)s * aa.ek_stride;
    double *r0 = aa.r0 + (long long)s * aa.r0_stride;
    const int nek = m / 2 + 52;
    __syncthreads();

    // the radix-2 first pass straight from the frame's samples: X[b] = x[b] +/- x[b + m]   (:416-421, dft_r<2>)
    auto load_half = [&](long long t0, int c, int tf) {
        for (int b = tf; b < m; b += FM_T) {
            double2 v0, v1;
            if (F32IN) {
                const float2 f0 = rawf[t0 + b], f1 = rawf[t0 + b + m];
                v0 = make_double2((double)f0.x, (double)f0.y);
                v1 = make_double2((double)f1.x, (double)f1.y);
            } else {
                const int w0 = raw[t0 + b], w1 = raw[t0 + b + m];
                v0 = make_double2((double)i16_to_float_java(java_short_add((int)(short)(w0 & 0xffff), a.ic)),
                                  (double)i16_to_float_java(java_short_add(w0 >> 16, a.qc)));
                v1 = make_double2((double)i16_to_float_java(java_short_add((int)(short)(w1 & 0xffff), a.ic)),
                                  (double)i16_to_float_java(java_short_add(w1 >> 16, a.qc)));
            }
            X[b] = c ? cdsub(v0, v1) : cdadd(v0, v1);
        }
        __syncthreads();
    };
    // bin `bin` of the full spectrum after both forward halves: even bins from the scratch, odd ones from the image
    auto spectrum = [&](int bin) -> double2 { return (bin & 1) ? X[bin >> 1] : ek[bin >> 1]; };

    for (int f = 0; f < a.nframes; f++) {
        const long long t0 = (long long)f * n;
        int tf = tid;
        asm volatile("" : "+v"(tf));
        // ---- forward, even bins
        // bins below end + 102 are read (see fm_pass5_band): even bin 2 i <=> output i of this half
        const int need_even = (end + 102 + 1) / 2, need_odd = (end + 102) / 2;
        if (pruned) {
            const GblArr tg = gbl_arr(aa.sub.f.tw);
            fm_first2x_from_raw<9600, F32IN, false>(XL, raw + t0, rawf + t0, a.ic, a.qc, tg, tg + 4, tf);
            fm_forward<false>(XL, nullptr, aa.sub, tf, true, FM_FWD_BAND, need_even);
        } else {
            load_half(t0, 0, tf);
            fm_forward<false>(XL, nullptr, aa.sub, tf, false, FM_FULL, need_even);
        }
        for (int i = tf; i < (pruned ? need_even : nek); i += FM_T) ek[i] = X[i];
        __threadfence_block();
        __syncthreads();
        // ---- forward, odd bins
        if (pruned) {
            const GblArr tg = gbl_arr(aa.sub.f.tw);
            fm_first2x_from_raw<9600, F32IN, true>(XL, raw + t0, rawf + t0, a.ic, a.qc, tg + aa.tw1_off[0], tg + aa.tw1_off[1], tf);
            fm_forward_odd(XL, aa, tf, FM_FWD_BAND, need_odd, 1.0, true);
        } else {
            load_half(t0, 1, tf);
            fm_forward_odd(XL, aa, tf, FM_FULL, need_odd);
        }
        // ---- |X| over the band the boxcar reads (:425-427)
        // (pbase is even: even bins come from the scratch, odd ones from the image -- one loop each, no per-bin branch)
        for (int i = pbase + 2 * tf; i < end - 24; i += 2 * FM_T) {
            const double2 v = ek[i >> 1];
            P[i - pbase] = sqrt(v.x * v.x + v.y * v.y);
        }
        for (int i = pbase + 1 + 2 * tf; i < end - 24; i += 2 * FM_T) {
            const double2 v = X[i >> 1];
            P[i - pbase] = sqrt(v.x * v.x + v.y * v.y);
        }
        __syncthreads();
        // ---- 100-wide boxcar, first maximum (:433-442)
        double bestv = 0.0;
        int besti = -1;
        for (int i = beg + 74 + 2 * tf; i < end - 75; i += 2 * FM_T) {
            const double2 *w = reinterpret_cast<const double2 *>(P + (i - 50 - pbase));
            double a0, a1;
            boxcar_pair(w, a0, a1);
            asm volatile("" : "+v"(a0), "+v"(a1));
            if (i >= beg + 75) {
                A[i - abase] = a0;
                if (bestv < a0) {
                    bestv = a0;
                    besti = i;
                }
            }
            if (i + 1 < end - 75) {
                A[i + 1 - abase] = a1;
                if (bestv < a1) {
                    bestv = a1;
                    besti = i + 1;
                }
            }
        }
        wave_first_max(bestv, besti);
        if (lane == 0) {
            redv[wave] = bestv;
            redi[wave] = besti;
        }
        __syncthreads();
        // ---- centre-bin rule (:444-453)
        {
            // the twelve per-wave maxima meet in lanes 0..11 of every wave (the same combine as above: larger value,
            // then smaller index; an empty candidate never wins) -- as a 12-step loop run by every thread this cost the
            // SIMDs 3.6k cycles a frame
            double mv = 0.0;
            int mi = -1;
            if (lane < FM_T / 64) {
                mv = redv[lane];
                mi = redi[lane];
            }
            wave_first_max(mv, mi);
            const double maxBin = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(mv)),
                                                   __builtin_amdgcn_readfirstlane(__double2loint(mv)));
            const int binPos = __builtin_amdgcn_readfirstlane(mi);
            if (centreBin < 0) centreBin = 0;
            if (centreBin > end - 1) centreBin = end - 1;
            const double atc = (centreBin >= beg + 75 && centreBin < end - 75) ? A[centreBin - abase] : 0.0;
            avePeakPower = (PSD_AVG * atc) + (PSD_INV * avePeakPower);
            if (maxBin > (avePeakPower / 4) * 5 && binPos > 0) {
                aveCentreBin = (CFREQ_AVG * (double)(float)binPos) + (CFREQ_INV * aveCentreBin);
                centreBin = (int)(aveCentreBin + (double)1.0F);
            }
            if (centreBin < 102) centreBin = 102;
        }
        // ---- 204 bins around the centre to bin 0 of a zeroed array (:458); inverse (:459) as conj o forward o conj
        double2 keep = make_double2(0.0, -0.0);  // conj of the zeroed array
        if (tf < 204) {
            const double2 v = spectrum(centreBin - 102 + tf);
            keep = make_double2(v.x, -v.y);
        }
        __syncthreads();
        const double2 Z = make_double2(0.0, -0.0);
        if (pruned) {
            // The radix-2 first pass leaves z[b] + Z = z[b] in the even half and z[b] - Z = z[b] in the odd one (b < 204, zeros
            // elsewhere): both halves start from the same 204 values, and their first three passes come straight from
            // them (fm_inv_blocks) -- the even half with the 9600-point tables, the odd one with its own
            if (tf < 204) zb[tf] = keep;
            __syncthreads();
            // (four passes, 4,4,4,2: fm_inv_blocks128_9600; the even half's samples re / n (:462) go straight to the scratch, the
            //  odd half's stay in LDS as a compact array of doubles -- sample 2 i + 1 at slot FM_RB0 + i)
            const GblArr tg = gbl_arr(aa.sub.f.tw);
            fm_inv_blocks128_9600<false>(XL, lds_arr(zb), tg + 20, tg + 84, tf);
            fm_forward<false>(XL, nullptr, aa.sub, tf, true, FM_INV_REAL, 0, norm, nullptr, r0);
            fm_inv_blocks128_9600<false>(XL, lds_arr(zb), tg + aa.tw1_off[2], tg + aa.tw1_off[3], tf);  // U_2[k], U_3[kk]
            fm_forward_odd(XL, aa, tf, FM_INV_REAL, 0, norm, true);
        } else {
            // even output samples: first-pass sums z[b] + z[b + m], z[b + m] being the zeroed array's
            for (int b = tf; b < m; b += FM_T) X[b] = cdadd(b < 204 ? keep : Z, Z);
            __syncthreads();
            fm_forward<false>(XL, nullptr, aa.sub, tf, false);
            for (int i = tf; i < m; i += FM_T) r0[i] = X[i].x * norm;  // re = X.x / n (:462), sample 2 i
            __threadfence_block();
            __syncthreads();
            // odd output samples: first-pass differences
            for (int b = tf; b < m; b += FM_T) X[b] = cdsub(b < 204 ? keep : Z, Z);
            __syncthreads();
            fm_forward_odd(XL, aa, tf);
            for (int i = tf; i < m; i += FM_T) X[i].x = X[i].x * norm;  // re = X.x / n (:462), sample 2 i + 1
            __syncthreads();
        }
        // ---- RxDownSample(re, re) (:461-463, :470-492): sample t of the frame = r0[t/2] (t even) or X[t/2].x / n (t odd)
        const double *Ro = reinterpret_cast<const double *>(smem) + FM_RB0;  // pruned: the odd samples, compact
        auto sample = [&](int t) -> double {
            return t < 0 ? hist[26 + t] : ((t & 1) ? (pruned ? Ro[t >> 1] : X[t >> 1].x) : r0[t >> 1]);
        };
        {
            long long jlo = (t0 - a.first_out + D - 1) / D;
            if (t0 <= a.first_out) jlo = 0;
            // with an even decimation (20 at 192 kHz) every window of the call ends on the same parity (t0 is even): which
            // of the 27 taps read the scratch and which the image is then known at compile time
            const bool uniform_parity = (D & 1) == 0;
            const int epar = (int)((a.first_out - t0) & 1);
            for (long long j = jlo + tf;; j += FM_T) {
                const long long te = (long long)a.first_out + (long long)D * j;
                if (te >= t0 + n || j >= a.nds) break;
                const double2 cs = a.vco_cs[j];
                const int e = (int)(te - t0);
                double fi = 0.0;
                if (pruned && (D & 3) == 0 && e >= 27) {
                    // taps of one parity read a run of 14 even samples (scratch), the others a run of 14 odd ones (LDS), both
                    // ending at a slot whose parity is the same for every window of the call (D/2 is even: D = 20): eight aligned
                    // 16-byte reads each, the run picked out of them by a shift that is the same in every lane
                    const int hi_o = epar ? (e >> 1) : (e >> 1) - 1;  // odd samples:  slots hi_o - 13 .. hi_o
                    const int hi_e = epar ? ((e - 1) >> 1) : (e >> 1);  // even samples: slots hi_e - 13 .. hi_e
                    const int bo = (hi_o - 13) & ~1, be = (hi_e - 13) & ~1;
                    const bool sho = ((hi_o - 13) & 1) != 0, she = ((hi_e - 13) & 1) != 0;
                    const double2 *po = reinterpret_cast<const double2 *>(Ro + bo);
                    const double2 *pe = reinterpret_cast<const double2 *>(r0 + be);
                    double dod[16], dev[16];
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const double2 u = pe[i];
                        dev[2 * i] = u.x;
                        dev[2 * i + 1] = u.y;
                    }
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const double2 u = po[i];
                        dod[2 * i] = u.x;
                        dod[2 * i + 1] = u.y;
                    }
                    // sample q slots below the run's end: d[13 - q + shift]
                    if (epar) {
#pragma unroll
                        for (int k = 0; k < 27; k++) {
                            const int q = k >> 1;
                            const double x = (k & 1) ? (she ? dev[14 - q] : dev[13 - q]) : (sho ? dod[14 - q] : dod[13 - q]);
                            fi += x * ds_tap(k);  // newest first (:479-483)
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 27; k++) {
                            const int q = (k & 1) ? ((k + 1) >> 1) - 1 : (k >> 1);
                            const double x = (k & 1) ? (sho ? dod[14 - q] : dod[13 - q]) : (she ? dev[14 - q] : dev[13 - q]);
                            fi += x * ds_tap(k);
                        }
                    }
                } else if (!pruned && uniform_parity && e >= 26) {
                    if (epar) {  // e odd: taps 0,2,.. read odd samples (image), taps 1,3,.. even ones (scratch)
                        const double2 *xo = X + (e >> 1);
                        const double *re = r0 + ((e - 1) >> 1);
#pragma unroll
                        for (int k = 0; k < 27; k++) fi += ((k & 1) ? re[-(k >> 1)] : xo[-(k >> 1)].x) * ds_tap(k);
                    } else {     // e even: taps 0,2,.. read even samples (scratch), taps 1,3,.. odd ones (image)
                        const double *re = r0 + (e >> 1);
                        const double2 *xo = X + ((e - 1) >> 1);
#pragma unroll
                        for (int k = 0; k < 27; k++) fi += ((k & 1) ? xo[-(k >> 1)].x : re[-(k >> 1)]) * ds_tap(k);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 27; k++) fi += sample(e - k) * ds_tap(k);  // newest first (:479-483)
                }
                const double o = fi * HOWARD;
                __builtin_nontemporal_store(o * cs.x, &dm[64 + j].x);  // :515-516 (written once, read by the next kernel)
                __builtin_nontemporal_store(o * cs.y, &dm[64 + j].y);
            }
        }
        double hnew = 0.0;
        if (tf < 26) hnew = sample(n - 26 + tf);
        __syncthreads();
        if (tf < 26) hist[tf] = hnew;
        __syncthreads();
    }
    if (tid < 26) sp->hist[tid] = hist[tid];
    if (tid == 0) {
        sp->avePeakPower = avePeakPower;
        sp->aveCentreBin = aveCentreBin;
        sp->centreBin = centreBin;
    }
}

// radix list for n = 2^a 3^b 5^c (the oracle's jo_fft_mixed_radices); 0: unsupported
int fftm_radices(int n, int *rad)
{
    int c = 0;
    if (n < 2) return 0;
    while (n % 4 == 0 && c < FM_MAXPASS) {
        rad[c++] = 4;
        n /= 4;
    }
    if (n % 2 == 0 && c < FM_MAXPASS) {
        rad[c++] = 2;
        n /= 2;
    }
    while (n % 3 == 0 && c < FM_MAXPASS) {
        rad[c++] = 3;
        n /= 3;
    }
    while (n % 5 == 0 && c < FM_MAXPASS) {
        rad[c++] = 5;
        n /= 5;
    }
    while (n % 7 == 0 && c < FM_MAXPASS) {
        rad[c++] = 7;
        n /= 7;
    }
    // any other prime factor, ascending (the oracle's jo_fft_mixed_radices)
    for (int p = 11; n > 1 && c < FM_MAXPASS; p += 2) {
        if (p * p > n) p = n;
        while (n % p == 0 && c < FM_MAXPASS) {
            rad[c++] = p;
            n /= p;
        }
    }
    return n == 1 ? c : 0;
}

// frames whose plan holds a prime radix above 7 need a per-stream scratch of n elements (0: none)
size_t fftm_scratch(int n)
{
    int rad[FM_MAXPASS];
    const int np = fftm_radices(n, rad);
    for (int p = 0; p < np; p++)
        if (rad[p] > 7) return (size_t)n;
    return 0;
}

bool fftm_supported(int n)
{
    int rad[FM_MAXPASS];
    // the band arithmetic (:429-443: beg = n/4 or 0, end = n/2 or n/4 in Java's integer division, a 100-wide window and 75
    // bins of margin either side) needs n/4 > 150; the image must fit the LDS.  Round 4: any such n = 2^a 3^b 5^c 7^d -- n need
    // not be a multiple of 16 (the 16-byte boxcar reads are aligned relative to the band's own start), so the 4410-sample
    // frame of a 44.1 kHz sound card is in
    // Round 5: any prime factor (a pass of that radix as the DFT's definition: 11.025 kHz gives n = 1102 = 2 19 29), and frames
    // down to 416 samples (8 kHz gives 800): below n = 604 the boxcar range [beg+75, end-75) is empty -- as in the reference,
    // whose loop (:433) then never runs -- and the 204 gathered bins (centreBin <= n/2 - 1) still end inside the frame.
    return n >= 416 && n <= FM_NMAX && (n & (n - 1)) != 0 && fftm_radices(n, rad) > 0;
}

// per-pass tables T[m] = exp(-2 pi i m/(P r)), m < P r: long double + one rounding, exact on the axes -- the same
// values as the oracle's jo_fft_mixed_table
void fftm_twiddles(std::vector<double2> &w, int n, int *np_out, int *rad, int *tw_off, int *wr_off)
{
    const int np = fftm_radices(n, rad);
    *np_out = np;
    w.clear();
    struct Fill {
        static void table(std::vector<double2> &w, size_t o, int len)
        {
            for (int m = 0; m < len; m++) {
                const long double ang = 2.0L * 3.14159265358979323846264338327950288L * (long double)m / (long double)len;
                w[o + m] = make_double2((double)cosl(ang), (double)(-sinl(ang)));
            }
            w[o] = make_double2(1.0, -0.0);
            if (len % 4 == 0) {
                w[o + len / 4] = make_double2(0.0, -1.0);
                w[o + 3 * len / 4] = make_double2(-0.0, 1.0);
            }
            if (len % 2 == 0) w[o + len / 2] = make_double2(-1.0, -0.0);
        }
    };
    int P = 1;
    for (int p = 0; p < np; p++) {
        const int len = P * rad[p];
        tw_off[p] = (int)w.size();
        const size_t o = w.size();
        w.resize(o + (size_t)len);
        for (int m = 0; m < len; m++) {
            const long double ang = 2.0L * 3.14159265358979323846264338327950288L * (long double)m / (long double)len;
            w[o + m] = make_double2((double)cosl(ang), (double)(-sinl(ang)));
        }
        w[o] = make_double2(1.0, -0.0);
        if (len % 4 == 0) {
            w[o + len / 4] = make_double2(0.0, -1.0);
            w[o + 3 * len / 4] = make_double2(-0.0, 1.0);
        }
        if (len % 2 == 0) w[o + len / 2] = make_double2(-1.0, -0.0);
        P *= rad[p];
    }
    // the r-point tables of the prime radices above 7, behind every pass table (so the default frames' offsets stay put)
    for (int p = 0; p < np; p++) {
        if (wr_off) wr_off[p] = 0;
        if (rad[p] > 7 && wr_off) {
            wr_off[p] = (int)w.size();
            const size_t o = w.size();
            w.resize(o + (size_t)rad[p]);
            Fill::table(w, o, rad[p]);
        }
    }
}

// round 6: a call of two or more 4800- or 4410-sample frames a stream takes them in pairs (k_front_fftm2); JSDR_FFTM_PAIR=0: never
bool fftm_pairs(int n, int nframes)
{
    bool pair = (n == 4800 || n == 4410) && nframes >= 2;
    if (const char *e = knob("JSDR_FFTM_PAIR")) pair = pair && atoi(e) != 0;
    return pair;
}

int launch_front_fftm(const FftFrontArgs &a, int np, const int *rad, const int *tw_off, const int *wr_off, double2 *gscratch,
                      int nstreams, hipStream_t st)
{
    FftmArgs aa;
    aa.f = a;
    aa.np = np;
    aa.gscratch = gscratch;
    aa.gscratch_stride = a.n;
    for (int p = 0; p < FM_MAXPASS; p++) {
        aa.wr_off[p] = (p < np && wr_off) ? wr_off[p] : 0;
        if (p < np && rad[p] > 7 && (!gscratch || !wr_off || wr_off[p] <= 0)) {
            set_error("launch_front_fftm: a radix-%d pass without its table or scratch", rad[p]);
            return JSDR_ERR;
        }
    }
    const size_t fixed = sizeof(double2) * (size_t)a.n + sizeof(double) * (32 + 16) + sizeof(int) * 16 + 64;
    const size_t room = (size_t)160 * 1024 - fixed;  // LDS left for twiddle tables (at n = 9600: 596 entries, the first five passes')
    aa.lds_tw = 0;
    int P = 1;
    for (int p = 0; p < FM_MAXPASS; p++) {
        aa.rad[p] = p < np ? rad[p] : 1;
        aa.tw_off[p] = p < np ? tw_off[p] : 0;
        aa.pmagic[p] = (unsigned)(((1ull << 32) + (unsigned)P - 1) / (unsigned)P);  // exact for b < 2^16, P <= 9600
        if (p < np) {
            const size_t end = (size_t)tw_off[p] + (size_t)P * rad[p];
            if (end == (size_t)aa.lds_tw + (size_t)P * rad[p] && end * sizeof(double2) <= room) aa.lds_tw = (int)end;  // a prefix
            P *= rad[p];
        }
    }
    // the default frames' kernels address their tables by compile-time offsets (fm_forward)
    if ((a.n == 9600 && (aa.lds_tw != FM_LDS_TW_9600 || tw_off[5] != 596 || tw_off[6] != 2516)) ||
        (a.n == 4800 && (aa.lds_tw != FM_LDS_TW_4800 || tw_off[4] != 276 || tw_off[5] != 1236)) ||
        (a.n == 4410 && (aa.lds_tw != 5156 || tw_off[1] != 2 || tw_off[2] != 8 || tw_off[3] != 26 || tw_off[4] != 116 || tw_off[5] != 746))) {
        set_error("launch_front_fftm: the twiddle tables of a default frame are not where the kernel expects them");
        return JSDR_ERR;
    }
    const bool f32 = a.rawf != nullptr;
    if (fftm_pairs(a.n, a.nframes)) {
        // the tables of all passes but the last (4800: but the last two) stay in LDS; the rest come from L2
        aa.lds_tw = a.n == 4800 ? 276 : 746;
        const size_t lds2 = sizeof(double2) * 2 * (size_t)a.n + sizeof(double) * (32 + 32) + sizeof(int) * 32 + sizeof(double2) * (size_t)aa.lds_tw + 64;
#define FM_LAUNCH2(NN, F32)                                                                                   \
    do {                                                                                                      \
        JSDR_LDS_ATTR((k_front_fftm2<NN, F32>), lds2);                                                        \
        hipLaunchKernelGGL((k_front_fftm2<NN, F32>), dim3((unsigned)nstreams), dim3(FM_T), lds2, st, aa);     \
    } while (0)
        if (a.n == 4800) {
            if (f32)
                FM_LAUNCH2(4800, true);
            else
                FM_LAUNCH2(4800, false);
        } else {
            if (f32)
                FM_LAUNCH2(4410, true);
            else
                FM_LAUNCH2(4410, false);
        }
#undef FM_LAUNCH2
        JSDR_LAUNCH_CHECK();
        return JSDR_OK;
    }
    const size_t lds = fixed + sizeof(double2) * (size_t)aa.lds_tw;
    if (f32)
        JSDR_LDS_ATTR(k_front_fftm<true>, lds);
    else
        JSDR_LDS_ATTR(k_front_fftm<false>, lds);
    if (f32)
        hipLaunchKernelGGL(k_front_fftm<true>, dim3((unsigned)nstreams), dim3(FM_T), lds, st, aa);
    else
        hipLaunchKernelGGL(k_front_fftm<false>, dim3((unsigned)nstreams), dim3(FM_T), lds, st, aa);
    JSDR_LAUNCH_CHECK();
    return JSDR_OK;
}

// frames of n = 2 m samples with m an LDS-sized mixed-radix frame (n = 19200: m = 9600)
// (the halves' passes are the blocked ones of the 2^a 3^b 5^c frames: m a multiple of 16 without a factor 7)
bool fft2x_supported(int n)
{
    const int m = n / 2;
    return n > FM_NMAX && (n % 2) == 0 && (m % 16) == 0 && (m % 7) != 0 && fftm_supported(m);
}

// the m-point tables (fftm_twiddles) followed by the c = 1 tables U_p[(j-1) P' + k'] = T_{2 P' r}[(2 k' + 1) j]
void fft2x_twiddles(std::vector<double2> &w, int n, int *np_out, int *rad, int *tw_off, int *tw1_off)
{
    const int m = n / 2;
    fftm_twiddles(w, m, np_out, rad, tw_off, nullptr);
    int P = 1;
    for (int p = 0; p < *np_out; p++) {
        const int r = rad[p], len = 2 * P * r;
        tw1_off[p] = (int)w.size();
        const size_t o = w.size();
        w.resize(o + (size_t)(r - 1) * P);
        for (int j = 1; j < r; j++) {
            for (int k = 0; k < P; k++) {
                const int mm = (2 * k + 1) * j;  // < len
                const long double ang = 2.0L * 3.14159265358979323846264338327950288L * (long double)mm / (long double)len;
                double2 t = make_double2((double)cosl(ang), (double)(-sinl(ang)));
                // exact on the axes, as jo_fft_mixed_table
                if (mm == 0) t = make_double2(1.0, -0.0);
                if (len % 4 == 0 && mm == len / 4) t = make_double2(0.0, -1.0);
                if (len % 4 == 0 && mm == 3 * len / 4) t = make_double2(-0.0, 1.0);
                if (len % 2 == 0 && mm == len / 2) t = make_double2(-1.0, -0.0);
                w[o + (size_t)(j - 1) * P + k] = t;
            }
        }
        P *= r;
    }
}

size_t fft2x_scratch_ek(int n) { return (size_t)(n / 4 + 52); }
// (+16: the aligned 16-byte reads of a window run may end two slots past the last sample)
size_t fft2x_scratch_r0(int n) { return (size_t)(n / 2) + 16; }

int launch_front_fft2x(const FftFrontArgs &a, int np, const int *rad, const int *tw_off, const int *tw1_off, double2 *ek,
                       double *r0, int nstreams, hipStream_t st)
{
    Fft2xArgs aa;
    aa.sub.f = a;
    aa.sub.f.n = a.n / 2;
    aa.sub.np = np;
    aa.sub.lds_tw = 0;
    aa.n = a.n;
    int P = 1;
    for (int p = 0; p < FM_MAXPASS; p++) {
        aa.sub.rad[p] = p < np ? rad[p] : 1;
        aa.sub.tw_off[p] = p < np ? tw_off[p] : 0;
        aa.tw1_off[p] = p < np ? tw1_off[p] : 0;
        aa.sub.pmagic[p] = (unsigned)(((1ull << 32) + (unsigned)P - 1) / (unsigned)P);
        if (p < np) P *= rad[p];
    }
    aa.ek = ek;
    aa.r0 = r0;
    aa.ek_stride = (long long)fft2x_scratch_ek(a.n);
    aa.r0_stride = (long long)fft2x_scratch_r0(a.n);
    const size_t lds = sizeof(double2) * (size_t)(a.n / 2) + sizeof(double) * (64 + 16) + sizeof(int) * 16 + sizeof(double2) * 204 + 64;
    const bool f32 = a.rawf != nullptr;
    if (f32)
        JSDR_LDS_ATTR(k_front_fft2x<true>, lds);
    else
        JSDR_LDS_ATTR(k_front_fft2x<false>, lds);
    if (f32)
        hipLaunchKernelGGL(k_front_fft2x<true>, dim3((unsigned)nstreams), dim3(FM_T), lds, st, aa);
    else
        hipLaunchKernelGGL(k_front_fft2x<false>, dim3((unsigned)nstreams), dim3(FM_T), lds, st, aa);
    JSDR_LAUNCH_CHECK();
    return JSDR_OK;
}


// ---- the three-phase form's two frame-parallel kernels for the default frames (bpsk_acq.hip launches the scan and the edges)
bool acqm_supported(int n) { return n == 9600 || n == 4800 || n == 4410; }

int launch_acqm(const AcqArgs &a, const FftFrontArgs &fa, int np, const int *rad, const int *tw_off, const int *wr_off, int num_cu,
                int which, hipStream_t st)
{
    FftmArgs aa;
    aa.f = fa;
    aa.np = np;
    aa.gscratch = nullptr;
    aa.gscratch_stride = 0;
    const size_t fixed = sizeof(double2) * (size_t)a.n + sizeof(double) * (32 + 16) + sizeof(int) * 16 + 64;
    const size_t room = (size_t)160 * 1024 - fixed;
    aa.lds_tw = 0;
    int P = 1;
    for (int p = 0; p < FM_MAXPASS; p++) {
        aa.rad[p] = p < np ? rad[p] : 1;
        aa.tw_off[p] = p < np ? tw_off[p] : 0;
        aa.wr_off[p] = (p < np && wr_off) ? wr_off[p] : 0;
        aa.pmagic[p] = (unsigned)(((1ull << 32) + (unsigned)P - 1) / (unsigned)P);
        if (p < np) {
            const size_t end = (size_t)tw_off[p] + (size_t)P * rad[p];
            if (end == (size_t)aa.lds_tw + (size_t)P * rad[p] && end * sizeof(double2) <= room) aa.lds_tw = (int)end;
            P *= rad[p];
        }
    }
    JSDR_REQUIRE(acqm_supported(a.n), "launch_acqm: frame of %d samples", a.n);
    JSDR_REQUIRE((a.n == 9600 && aa.lds_tw == FM_LDS_TW_9600) || (a.n == 4800 && aa.lds_tw == FM_LDS_TW_4800) || (a.n == 4410 && aa.lds_tw == 5156),
                 "launch_acqm: the twiddle tables of a default frame are not where the kernel expects them");
    const size_t lds = fixed + sizeof(double2) * (size_t)aa.lds_tw;
    const long long nfr = (long long)a.S * a.F;
    const int per_cu = lds > 80 * 1024 ? 1 : 2;
    long long grid = (long long)per_cu * num_cu;
    if (grid > nfr) grid = nfr;
    if (which == 0) {
        if (fa.rawf) {
            JSDR_LDS_ATTR(k_acqm_fwd<true>, lds);
            hipLaunchKernelGGL(k_acqm_fwd<true>, dim3((unsigned)grid), dim3(FM_T), lds, st, aa, a);
        } else {
            JSDR_LDS_ATTR(k_acqm_fwd<false>, lds);
            hipLaunchKernelGGL(k_acqm_fwd<false>, dim3((unsigned)grid), dim3(FM_T), lds, st, aa, a);
        }
    } else {
        JSDR_LDS_ATTR(k_acqm_inv, lds);
        hipLaunchKernelGGL(k_acqm_inv, dim3((unsigned)grid), dim3(FM_T), lds, st, aa, a);
    }
    JSDR_LAUNCH_CHECK();
    return JSDR_OK;
}

}  // namespace jsdr
