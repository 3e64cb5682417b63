// bpsk_acqg.hip -- FUNcubeBPSKDemod FFT-acquire (doBufferFFT, FUNcubeBPSKDemod.java:406-464) for ANY frame (round 6).
//
// JTransforms takes any n (:194, :422-423) and a frame is a tenth of a second of any audio-rate (JavaAudio.java:49,58-59).  The LDS
// front ends hold a frame as a double2 image in ONE workgroup's LDS: powers of two 1024 .. 8192 (bpsk_fft.hip / bpsk_acq.hip), any
// other frame of 416 .. 9600 samples (bpsk_fftm.hip), twice a 16 | m, 2^a 3^b 5^c frame (19200).  Every other frame the oracle
// defines comes here: powers of two below 1024 and above 8192, 17640 (176.4 kHz: 2 x 8820, 7^2 | 8820), 38400 (384 kHz), any
// frame above 9600 samples, and the LDS front ends' frames at decimations they do not take (a 2^k frame below 38.4 kHz).
//
// Phases A and C of the three-phase front end (bpsk_acq.hip: frame-parallel forward half, one scan per stream, frame-parallel
// inverse half, edges) with the frame's image in GLOBAL memory: one launch per pass or pass pair of the transform (per four radix-2 stages), one thread per butterfly or group,
// over all frames of the launch at once.  The transform is the oracle's, operation for operation (oracle/o_fft.c): powers of two
// the radix-2 decimation-in-time network on jo_fft_twiddles_f64's table (inverse: conjugated twiddles), every other n the Stockham
// passes of fft_f64_mixed (radices in jo_fft_mixed_radices' order, per-pass tables, the fixed-order butterflies of bpsk_radix.h, a
// prime radix above 7 as the DFT's definition; inverse = conj o forward o conj) -- so centre bins, traces, bits and FEC bytes are
// bit-identical to the oracle's.  A pass is a round trip through HBM (32 bytes a point): a correctness path for rare rates, ~10x
// the LDS kernels' time per sample; the scan (k_acq_scan) and the windows across frame borders (k_acq_edges) are shared.
#include "bpsk_fft.h"
#include "bpsk_radix.h"
#include <math.h>

namespace jsdr {

constexpr int AG_T = 256;
constexpr int AG_TB = 2048;  // boxcar outputs a workgroup forms per LDS tile of |X|

__device__ __forceinline__ int ag_bitrev(int x, int bits) { return (int)(__brev((unsigned)x) >> (32 - bits)); }

// :416-421 -- the frame to the image, (double) of JavaAudio's float samples; powers of two at the bit-reversed position (the
// permutation jo_fft_f64 starts with)
template <bool F32IN>
__global__ __launch_bounds__(AG_T) void k_acqg_load(AcqArgs a, double2 *img, int logn, int bpf)
{
    const unsigned g = blockIdx.x / (unsigned)bpf;
    const int t = (int)(blockIdx.x - g * (unsigned)bpf) * AG_T + (int)threadIdx.x;
    if (t >= a.n) return;
    const int s = (int)(g / (unsigned)a.F), f = (int)(g - (unsigned)s * (unsigned)a.F);
    const long long src = (long long)s * a.stride_pairs + (long long)(a.f0 + f) * a.n + t;
    double di, dq;
    if (F32IN) {
        const float2 w = a.rawf[src];
        di = (double)w.x;
        dq = (double)w.y;
    } else {
        const int w = a.raw[src];
        di = (double)i16_to_float_java(java_short_add((int)(short)(w & 0xffff), a.ic));
        dq = (double)i16_to_float_java(java_short_add(w >> 16, a.qc));
    }
    img[(long long)g * a.n + (logn ? ag_bitrev(t, logn) : t)] = make_double2(di, dq);
}

// NST consecutive radix-2 stages of jo_fft_f64 (half = half0, 2 half0, ...), in place: a thread holds the 2^NST elements
// base + j + u half0 that those stages connect -- one round trip through memory for up to four stages.  Every butterfly is the
// oracle's: t = w b (four products, one difference, one sum), a' = a + t, b' = a - t, w = the stage table's entry for the pair's
// position in its block (inverse: conjugated).
template <bool INVERSE, int NST>
__global__ __launch_bounds__(AG_T) void k_acqg_stages(double2 *img, int n, int half0, const double2 *tw, int bpf)
{
    constexpr int M = 1 << NST;
    const unsigned g = blockIdx.x / (unsigned)bpf;
    const int i = (int)(blockIdx.x - g * (unsigned)bpf) * AG_T + (int)threadIdx.x;
    if (i >= (n >> NST)) return;
    double2 *X = img + (long long)g * n;
    const int j = i & (half0 - 1);
    const int base = ((i - j) << NST) + j;
    double2 v[M];
#pragma unroll
    for (int u = 0; u < M; u++) v[u] = X[base + u * half0];
#pragma unroll
    for (int t = 0; t < NST; t++) {
        const int h = half0 << t;
#pragma unroll
        for (int u = 0; u < M; u++) {
            if (u & (1 << t)) continue;
            const double2 w = tw[h - 1 + j + (u & ((1 << t) - 1)) * half0];
            const double wr = w.x, wi = INVERSE ? -w.y : w.y;
            const double2 b = v[u | (1 << t)], av = v[u];
            const double p1 = wr * b.x, p2 = wi * b.y, p3 = wr * b.y, p4 = wi * b.x;
            const double tr = p1 - p2, ti = p3 + p4;
            v[u] = make_double2(av.x + tr, av.y + ti);
            v[u | (1 << t)] = make_double2(av.x - tr, av.y - ti);
        }
    }
#pragma unroll
    for (int u = 0; u < M; u++) X[base + u * half0] = v[u];
}

// one Stockham pass of fft_f64_mixed_forward, out of place: butterfly b (k = b mod P) from in[b + j n/r], input j >= 1 times
// T[k j], the r-point butterfly, output q to out[(b - k) r + k + q P]
template <int R>
__global__ __launch_bounds__(AG_T) void k_acqg_pass(const double2 *in, double2 *out, int n, int P, const double2 *tw, int bpf)
{
    const unsigned g = blockIdx.x / (unsigned)bpf;
    const int b = (int)(blockIdx.x - g * (unsigned)bpf) * AG_T + (int)threadIdx.x;
    const int nb = n / R;
    if (b >= nb) return;
    const double2 *x = in + (long long)g * n;
    double2 *y = out + (long long)g * n;
    const int k = b % P;
    double2 v[R];
#pragma unroll
    for (int j = 0; j < R; j++) {
        v[j] = x[b + j * nb];
        if (j >= 1 && P > 1) v[j] = cdmul(v[j], tw[k * j]);
    }
    dft_r<R>(v);
    const int j0 = (b - k) * R + k;
#pragma unroll
    for (int q = 0; q < R; q++) y[j0 + q * P] = v[q];
}

// TWO consecutive Stockham passes (radix R1 at stride P, then R2 at stride P R1) in one round trip through memory: group g = m0 P + k1
// holds the R1 R2 points both passes connect -- the R2 first-pass butterflies b1 = g + j2 n/(R1 R2), whose outputs q1 feed the R1
// second-pass butterflies (k2 = k1 + q1 P), which write out[(g - k1) R1 R2 + k1 + q1 P + q2 P R1].  The operations of
// k_acqg_pass<R1> followed by k_acqg_pass<R2> on the same operands in the same order (bpsk_fftm.hip's fm_pass2, out of place).
template <int R1, int R2>
__global__ __launch_bounds__(AG_T) void k_acqg_pass2(const double2 *in, double2 *out, int n, int P, const double2 *tw1, const double2 *tw2, int bpf)
{
    constexpr int RR = R1 * R2;
    const unsigned fr = blockIdx.x / (unsigned)bpf;
    const int g = (int)(blockIdx.x - fr * (unsigned)bpf) * AG_T + (int)threadIdx.x;
    const int ng = n / RR, nb1 = n / R1;
    if (g >= ng) return;
    const double2 *x = in + (long long)fr * n;
    double2 *y = out + (long long)fr * n;
    const int k1 = g % P;
    double2 v[R2][R1];
#pragma unroll
    for (int j2 = 0; j2 < R2; j2++) {
        const int b1 = g + j2 * ng;
#pragma unroll
        for (int j1 = 0; j1 < R1; j1++) {
            v[j2][j1] = x[b1 + j1 * nb1];
            if (j1 >= 1 && P > 1) v[j2][j1] = cdmul(v[j2][j1], tw1[k1 * j1]);
        }
        dft_r<R1>(v[j2]);
    }
    double2 *z = y + ((g - k1) * RR + k1);
#pragma unroll
    for (int q1 = 0; q1 < R1; q1++) {
        const int k2 = k1 + q1 * P;
        double2 w[R2];
#pragma unroll
        for (int j2 = 0; j2 < R2; j2++) {
            w[j2] = v[j2][q1];
            if (j2 >= 1) w[j2] = cdmul(w[j2], tw2[k2 * j2]);
        }
        dft_r<R2>(w);
#pragma unroll
        for (int q2 = 0; q2 < R2; q2++) z[q1 * P + q2 * P * R1] = w[q2];
    }
}

// a prime radix above 7: out_q = v_0 + v_1 W[q mod r] + ... + v_{r-1} W[(r-1) q mod r], every product a full complex multiply,
// summed left to right (o_fft.c:206-228).  One thread per output; the twiddled inputs are formed again for every q (the same
// operations on the same operands give the same values).
__global__ __launch_bounds__(AG_T) void k_acqg_pass_prime(const double2 *in, double2 *out, int n, int P, int r, const double2 *tw,
                                                          const double2 *wr, int bpf)
{
    const unsigned g = blockIdx.x / (unsigned)bpf;
    const int i = (int)(blockIdx.x - g * (unsigned)bpf) * AG_T + (int)threadIdx.x;
    if (i >= n) return;
    const int nb = n / r;
    const int q = i / nb, b = i - q * nb;
    const double2 *x = in + (long long)g * n;
    double2 *y = out + (long long)g * n;
    const int k = b % P;
    double2 acc = x[b];
    for (int j = 1; j < r; j++) {
        double2 v = x[b + j * nb];
        if (P > 1) v = cdmul(v, tw[k * j]);
        const int m = (int)(((long long)j * q) % r);
        acc = cdadd(acc, cdmul(v, wr[m]));
    }
    y[(b - k) * r + k + q * P] = acc;
}

// :425-443 for one frame per workgroup: the bins a gather can reach to the frame's row (layout: acq_spec_index), |X| over the band
// in LDS tiles, the 100-wide boxcar summed j ascending for every i (:433-437), first maximum (:439-442)
__global__ __launch_bounds__(AG_T) void k_acqg_band(AcqArgs a, const double2 *img)
{
    __shared__ double Pm[AG_TB + 100];
    __shared__ double redv[AG_T / 64];
    __shared__ int redi[AG_T / 64];
    const long long g = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = a.n;
    const double2 *X = img + g * n;
    const int beg = a.do_up ? n / 4 : 0;
    const int end = a.do_up ? n / 2 : n / 4;
    {
        double2 *specg = a.spec + g * a.nsb;
        const int lo1 = n / 4 - 26;
        for (int i = tid; i < a.nsb; i += AG_T) {
            const int b = a.do_up ? (i < 204 ? i : lo1 + (i - 204)) : i;
            specg[i] = X[b];
        }
    }
    double bestv = 0.0;
    int besti = -1;
    double *ab = a.aband + g * a.na;
    for (int i0 = beg + 75; i0 < end - 75; i0 += AG_TB) {
        const int cnt = end - 75 - i0 < AG_TB ? end - 75 - i0 : AG_TB;
        for (int u = tid; u < cnt + 99; u += AG_T) {
            const double2 v = X[i0 - 50 + u];
            Pm[u] = sqrt(v.x * v.x + v.y * v.y);  // :425-427
        }
        __syncthreads();
        for (int u = tid; u < cnt; u += AG_T) {
            double acc = 0.0;
#pragma unroll 4
            for (int j = 0; j < 100; j++) acc += Pm[u + j];
            ab[i0 + u - (beg + 75)] = acc;
            if (bestv < acc) {
                bestv = acc;
                besti = i0 + u;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double ov = __shfl_xor(bestv, off, 64);
        const int oi = __shfl_xor(besti, off, 64);
        if (oi >= 0 && (ov > bestv || (ov == bestv && (besti < 0 || oi < besti)))) {
            bestv = ov;
            besti = oi;
        }
    }
    if (lane == 0) {
        redv[wave] = bestv;
        redi[wave] = besti;
    }
    __syncthreads();
    if (tid == 0) {
        double mv = 0.0;
        int mi = -1;
        for (int w = 0; w < AG_T / 64; w++) {
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
}

// :414-415, :458 -- the zeroed array with the 204 bins around the frame's centre bin at its head, as the inverse transform's input:
// every other n conjugated (fft_f64_mixed: conj o forward o conj; the zeros become (0, -0.0)), powers of two as they are, at the
// bit-reversed position
__global__ __launch_bounds__(AG_T) void k_acqg_gather(AcqArgs a, double2 *img, int logn, int bpf)
{
    const unsigned g = blockIdx.x / (unsigned)bpf;
    const int i = (int)(blockIdx.x - g * (unsigned)bpf) * AG_T + (int)threadIdx.x;
    const int n = a.n;
    if (i >= n) return;
    double2 v = make_double2(0.0, 0.0);
    if (i < 204) {
        const int lo1 = n / 4 - 26;
        const int c = a.cbin[g];
        int off = c - 102;
        if (a.do_up) off = (c == 102) ? 0 : 204 + (c - 102 - lo1);
        if (off < 0) off = 0;
        if (off + 204 > a.nsb) off = a.nsb - 204;
        v = a.spec[(long long)g * a.nsb + off + i];
    }
    if (logn)
        img[(long long)g * n + ag_bitrev(i, logn)] = v;
    else
        img[(long long)g * n + i] = make_double2(v.x, -v.y);
}

// :461-463, :470-492, :511-516 -- re / n of the inverse transform (both definitions scale by the product with 1.0 / n) through
// RxDownSample for the windows that lie inside the frame, VCO mix; the frame's first and last 26 samples for k_acq_edges
__global__ __launch_bounds__(AG_T) void k_acqg_rx(AcqArgs a, const double2 *img, int bpf)
{
    const unsigned g = blockIdx.x / (unsigned)bpf;
    const int r = (int)(blockIdx.x - g * (unsigned)bpf) * AG_T + (int)threadIdx.x;
    const int n = a.n, D = a.decim;
    const int s = (int)(g / (unsigned)a.F), f = (int)(g - (unsigned)s * (unsigned)a.F);
    const double2 *X = img + (long long)g * n;
    const double norm = 1.0 / (double)n;
    const double HOWARD = 0.9 * 32768.0;
    if (r < 52) a.edges[(long long)g * 52 + r] = X[r < 26 ? r : n - 52 + r].x * norm;
    const long long t0 = (long long)(a.f0 + f) * n;  // call-relative index of the frame's first sample
    const long long jlo = t0 <= a.first_out ? 0 : (t0 - a.first_out + D - 1) / D;
    const long long j = jlo + r;
    const long long te = a.first_out + (long long)D * j;  // window end, call-relative
    if (te >= t0 + n || j >= a.nds) return;
    const int e = (int)(te - t0);
    if (e < 26) return;
    double fi = 0.0;
#pragma unroll
    for (int k = 0; k < 27; k++) fi += (X[e - k].x * norm) * ds_tap(k);  // newest first (:479-483)
    const double o = fi * HOWARD;
    const double2 cs = a.vco_cs[j];
    a.dm[(long long)s * a.dm_stride + 64 + j] = make_double2(o * cs.x, o * cs.y);  // :515-516
}

// ============================================================================================================= host
// the oracle's jo_fft_mixed_radices (o_fft.c:151-171), frames above 9600 samples starting with one radix-2 pass
static int acqg_radices(int n, int *rad)
{
    int c = 0;
    if (n < 2) return 0;
    if (n > 9600 && n % 2 == 0) {
        rad[c++] = 2;
        n /= 2;
    }
    while (n % 4 == 0 && c < ACQG_MAXPASS) {
        rad[c++] = 4;
        n /= 4;
    }
    if (n % 2 == 0 && c < ACQG_MAXPASS) {
        rad[c++] = 2;
        n /= 2;
    }
    for (int p = 3; p <= 7; p += 2)
        while (n % p == 0 && c < ACQG_MAXPASS) {
            rad[c++] = p;
            n /= p;
        }
    for (int p = 11; n > 1 && c < ACQG_MAXPASS; p += 2) {
        if (p * p > n) p = n;
        while (n % p == 0 && c < ACQG_MAXPASS) {
            rad[c++] = p;
            n /= p;
        }
    }
    return n == 1 ? c : 0;
}

bool acqg_supported(int n)
{
    // below 416 samples the 204 gathered bins do not end inside the frame (the reference's own arraycopy would throw); the image
    // index arithmetic of the kernels is 32-bit within a frame
    if (n < 416 || n > (1 << 22)) return false;
    if ((n & (n - 1)) == 0) return true;
    int rad[ACQG_MAXPASS];
    const int np = acqg_radices(n, rad);
    if (np <= 0) return false;
    // a prime radix r above 7 is a pass of n r complex multiply-adds (one thread per output, r terms each): bounded, so that a frame
    // with a huge prime factor is refused at create instead of occupying the GPU for minutes per call
    for (int p = 0; p < np; p++)
        if (rad[p] > 7 && (long long)n * rad[p] > (1LL << 31)) return false;
    return true;
}

// images of a frame in the launch's scratch: a power of two is transformed in place, the Stockham passes go between two
size_t acqg_image_bytes(int n) { return sizeof(double2) * (size_t)n * (((n & (n - 1)) == 0) ? 1 : 2); }

void acqg_twiddles(std::vector<double2> &w, int n, AcqgPlan *plan)
{
    plan->on = true;
    plan->logn = 0;
    plan->np = 0;
    if ((n & (n - 1)) == 0) {
        while ((1 << plan->logn) < n) plan->logn++;
        fft_twiddles_f64(w, n);
        return;
    }
    plan->np = acqg_radices(n, plan->rad);
    w.clear();
    auto table = [&](int len) {  // jo_fft_mixed_table: long double + one rounding, exact on the axes
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
        return (int)o;
    };
    int P = 1;
    for (int p = 0; p < plan->np; p++) {
        plan->tw_off[p] = table(P * plan->rad[p]);
        P *= plan->rad[p];
    }
    for (int p = 0; p < plan->np; p++) plan->wr_off[p] = plan->rad[p] > 7 ? table(plan->rad[p]) : 0;
}

// the transform of every frame of the launch: img0 -> wherever the last pass leaves it (returned)
static double2 *acqg_transform(const AcqgPlan &pl, const double2 *tw, double2 *img0, double2 *img1, int n, long long nfr, bool inverse,
                               hipStream_t st, int *rc)
{
    *rc = JSDR_OK;
    if (pl.logn) {
        for (int done = 0; done < pl.logn;) {
            const int nst = pl.logn - done >= 4 ? 4 : pl.logn - done;
            const int half0 = 1 << done;
            const int bpf = ((n >> nst) + AG_T - 1) / AG_T;
            const dim3 grid((unsigned)(nfr * bpf));
#define AG_STAGES(NST)                                                                                              \
    if (inverse)                                                                                                    \
        hipLaunchKernelGGL((k_acqg_stages<true, NST>), grid, dim3(AG_T), 0, st, img0, n, half0, tw, bpf);          \
    else                                                                                                            \
        hipLaunchKernelGGL((k_acqg_stages<false, NST>), grid, dim3(AG_T), 0, st, img0, n, half0, tw, bpf)
            switch (nst) {
                case 4: AG_STAGES(4); break;
                case 3: AG_STAGES(3); break;
                case 2: AG_STAGES(2); break;
                default: AG_STAGES(1); break;
            }
#undef AG_STAGES
            done += nst;
        }
        if (hipGetLastError() != hipSuccess) *rc = JSDR_ERR;
        return img0;
    }
    double2 *in = img0, *out = img1;
    int P = 1;
    for (int p = 0; p < pl.np; p++) {
        const int r = pl.rad[p];
        const double2 *t = tw + pl.tw_off[p];
        // two passes of radices up to 7 and at most 25 points a group in one launch (JSDR_ACQG_PAIRS=0: one pass a launch)
        const int r2 = (p + 1 < pl.np) ? pl.rad[p + 1] : 0;
        static const bool pairs_on = [] { const char *e = knob("JSDR_ACQG_PAIRS"); return !(e && atoi(e) == 0); }();
        if (pairs_on && r <= 7 && r2 >= 2 && r2 <= 7 && r * r2 <= 25) {
            const double2 *t2 = tw + pl.tw_off[p + 1];
            const int bpf = (n / (r * r2) + AG_T - 1) / AG_T;
            const dim3 grid((unsigned)(nfr * bpf));
            bool done = true;
#define AG_P2(A, B)                                                                                      \
    else if (r == A && r2 == B) hipLaunchKernelGGL((k_acqg_pass2<A, B>), grid, dim3(AG_T), 0, st, in, out, n, P, t, t2, bpf)
            if (false) {
            }
            AG_P2(2, 2); AG_P2(2, 3); AG_P2(2, 4); AG_P2(2, 5); AG_P2(2, 7);
            AG_P2(3, 2); AG_P2(3, 3); AG_P2(3, 4); AG_P2(3, 5); AG_P2(3, 7);
            AG_P2(4, 2); AG_P2(4, 3); AG_P2(4, 4); AG_P2(4, 5);
            AG_P2(5, 2); AG_P2(5, 3); AG_P2(5, 4); AG_P2(5, 5);
            AG_P2(7, 2); AG_P2(7, 3);
            else done = false;
#undef AG_P2
            if (done) {
                P *= r * r2;
                p++;
                double2 *tmp = in;
                in = out;
                out = tmp;
                continue;
            }
        }
        const int bpf = ((r > 7 ? n : n / r) + AG_T - 1) / AG_T;
        const dim3 grid((unsigned)(nfr * bpf));
        switch (r) {
            case 2: hipLaunchKernelGGL(k_acqg_pass<2>, grid, dim3(AG_T), 0, st, in, out, n, P, t, bpf); break;
            case 3: hipLaunchKernelGGL(k_acqg_pass<3>, grid, dim3(AG_T), 0, st, in, out, n, P, t, bpf); break;
            case 4: hipLaunchKernelGGL(k_acqg_pass<4>, grid, dim3(AG_T), 0, st, in, out, n, P, t, bpf); break;
            case 5: hipLaunchKernelGGL(k_acqg_pass<5>, grid, dim3(AG_T), 0, st, in, out, n, P, t, bpf); break;
            case 7: hipLaunchKernelGGL(k_acqg_pass<7>, grid, dim3(AG_T), 0, st, in, out, n, P, t, bpf); break;
            default: hipLaunchKernelGGL(k_acqg_pass_prime, grid, dim3(AG_T), 0, st, in, out, n, P, r, t, tw + pl.wr_off[p], bpf); break;
        }
        P *= r;
        double2 *tmp = in;
        in = out;
        out = tmp;
    }
    if (hipGetLastError() != hipSuccess) *rc = JSDR_ERR;
    return in;
}

// phase A (which == 0) or phase C (which == 1) of one launch of the three-phase front end; img: the launch's images (S F frames of
// acqg_image_bytes(n))
int launch_acqg(const AcqArgs &a, const AcqgPlan &pl, double2 *img, int which, hipStream_t st)
{
    const int n = a.n;
    const long long nfr = (long long)a.S * a.F;
    const long long bpf_n = (n + AG_T - 1) / AG_T;
    // (the grids are one dimension of workgroups: frames x blocks a frame; the prime-radix pass has a thread per output)
    JSDR_REQUIRE(nfr * bpf_n < 0x7fffffffLL, "bpsk: an FFT-acquire launch of %d streams x %d frames of %d samples is beyond the any-frame kernels' grid",
                 a.S, a.F, n);
    double2 *img0 = img, *img1 = img + nfr * n;
    int rc = JSDR_OK;
    if (which == 0) {
        const int bpf = (int)bpf_n;
        if (a.rawf)
            hipLaunchKernelGGL(k_acqg_load<true>, dim3((unsigned)(nfr * bpf)), dim3(AG_T), 0, st, a, img0, pl.logn, bpf);
        else
            hipLaunchKernelGGL(k_acqg_load<false>, dim3((unsigned)(nfr * bpf)), dim3(AG_T), 0, st, a, img0, pl.logn, bpf);
        JSDR_LAUNCH_CHECK();
        const double2 *X = acqg_transform(pl, a.tw, img0, img1, n, nfr, false, st, &rc);  // :422-423
        if (rc != JSDR_OK) {
            set_error("bpsk: an any-frame FFT pass failed to launch");
            return JSDR_ERR;
        }
        hipLaunchKernelGGL(k_acqg_band, dim3((unsigned)nfr), dim3(AG_T), 0, st, a, X);
        JSDR_LAUNCH_CHECK();
        return JSDR_OK;
    }
    {
        const int bpf = (int)bpf_n;
        hipLaunchKernelGGL(k_acqg_gather, dim3((unsigned)(nfr * bpf)), dim3(AG_T), 0, st, a, img0, pl.logn, bpf);
        JSDR_LAUNCH_CHECK();
    }
    const double2 *X = acqg_transform(pl, a.tw, img0, img1, n, nfr, true, st, &rc);  // :459
    if (rc != JSDR_OK) {
        set_error("bpsk: an any-frame FFT pass failed to launch");
        return JSDR_ERR;
    }
    {
        // a frame holds at most n / D + 1 window ends; 52 threads of every frame also copy its edge samples
        int per = n / a.decim + 2;
        if (per < 52) per = 52;
        const int bpf = (per + AG_T - 1) / AG_T;
        hipLaunchKernelGGL(k_acqg_rx, dim3((unsigned)(nfr * bpf)), dim3(AG_T), 0, st, a, X, bpf);
        JSDR_LAUNCH_CHECK();
    }
    return JSDR_OK;
}

}  // namespace jsdr
