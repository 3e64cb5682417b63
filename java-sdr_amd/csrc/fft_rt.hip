// fft_rt.hip -- fft.receive (fft.java:190-228) for every frame n = 2^a 3^b 5^c 7^d that has no kernel of its own.
//
// The reference's frame is a tenth of a second of whatever `audio-rate` is (JavaAudio.java:49,58-59; fft.java:67,194
// transforms all of it, JTransforms taking any n): 44.1 kHz gives n = 4410 = 2 3^2 5 7^2 -- the reference's own
// sine4410.wav --, 22.05 kHz 2205, 32 kHz 3200, 88.2 kHz 8820, 24 kHz 2400, 16 kHz 1600, 8 kHz 800.  Round 4 sent all of
// them through the O(n^2) kernel (fft_any.hip: 60 us for one 4410-point frame).  This is the Stockham autosort transform
// with a RUN-TIME radix plan (16, 8, 4, 2 while they divide, then 3s, 5s, 7s): one frame per 256-thread workgroup, two
// LDS images (a pass reads one and writes the other: one barrier a pass, and no bound on the butterflies a thread takes),
// the in-register radix-2^k / 3 / 5 butterflies of the specialised kernels plus a radix-7 one, per-pass tables
// T[m] = exp(-2 pi i m/(P r)) read through L1/L2, int16 -> float fused into the first pass's loads, PSD / first maximum /
// Hz rule in the last pass's epilogue (the same rules as k_fft_mixed).  4 B in + 4 B out per sample.
// Parity: 1e-5 of the frame peak against the exact DFT (JTransforms' own rounding is unknowable: SURVEY 8c).
#include "fft_common.h"
#include <math.h>
#include <vector>

namespace jsdr {

enum { RT_T = 256, RT_MAXPASS = 12, RT_NMAX = 9800 };  // two float2 images of RT_NMAX elements + reduction scratch fit 160 KB

struct RtPlan {
    int n, np;
    int rad[RT_MAXPASS];
    int tw_off[RT_MAXPASS];  // pass p's table at tw + tw_off[p], P_p * r_p entries (pass 0 has none)
    int wr_off[RT_MAXPASS];  // a prime radix above 7: W[m] = exp(-2 pi i m / r), m < r, at tw + wr_off[p]
};

// radices with a butterfly of their own; everything else (a prime above 7: 1102 = 2 . 19 . 29 at 11.025 kHz) takes rt_pass_generic
__host__ __device__ inline bool rt_special(int r)
{
    return r == 2 || r == 3 || r == 4 || r == 5 || r == 6 || r == 7 || r == 8 || r == 9 || r == 10 || r == 14 || r == 15 || r == 16 ||
           r == 21 || r == 25;
}

__device__ __forceinline__ void rt_dft7(float2 *x)
{
    // the radix-5 scheme one size up: a_k = x_k + x_{7-k}, b_k = x_k - x_{7-k};  out_j, out_{7-j} = m_j -/+ i n_j
    constexpr float C1 = (float)0.62348980185873353053, C2 = (float)-0.22252093395631440429, C3 = (float)-0.90096886790241912624;
    constexpr float S1 = (float)0.78183148246802980871, S2 = (float)0.97492791218182360702, S3 = (float)0.43388373911755812048;
    const float2 a1 = cadd(x[1], x[6]), a2 = cadd(x[2], x[5]), a3 = cadd(x[3], x[4]);
    const float2 b1 = csub(x[1], x[6]), b2 = csub(x[2], x[5]), b3 = csub(x[3], x[4]);
    const float2 x0 = x[0];
    const float2 m1 = make_float2(x0.x + C1 * a1.x + C2 * a2.x + C3 * a3.x, x0.y + C1 * a1.y + C2 * a2.y + C3 * a3.y);
    const float2 m2 = make_float2(x0.x + C2 * a1.x + C3 * a2.x + C1 * a3.x, x0.y + C2 * a1.y + C3 * a2.y + C1 * a3.y);
    const float2 m3 = make_float2(x0.x + C3 * a1.x + C1 * a2.x + C2 * a3.x, x0.y + C3 * a1.y + C1 * a2.y + C2 * a3.y);
    const float2 n1 = make_float2(S1 * b1.x + S2 * b2.x + S3 * b3.x, S1 * b1.y + S2 * b2.y + S3 * b3.y);
    const float2 n2 = make_float2(S2 * b1.x - S3 * b2.x - S1 * b3.x, S2 * b1.y - S3 * b2.y - S1 * b3.y);
    const float2 n3 = make_float2(S3 * b1.x - S1 * b2.x + S2 * b3.x, S3 * b1.y - S1 * b2.y + S2 * b3.y);
    x[0] = make_float2(x0.x + a1.x + a2.x + a3.x, x0.y + a1.y + a2.y + a3.y);
    x[1] = make_float2(m1.x + n1.y, m1.y - n1.x);
    x[6] = make_float2(m1.x - n1.y, m1.y + n1.x);
    x[2] = make_float2(m2.x + n2.y, m2.y - n2.x);
    x[5] = make_float2(m2.x - n2.y, m2.y + n2.x);
    x[3] = make_float2(m3.x + n3.y, m3.y - n3.x);
    x[4] = make_float2(m3.x - n3.y, m3.y + n3.x);
}

__device__ __forceinline__ void rt_dft3(float2 *x)
{
    constexpr float S = (float)0.86602540378443864676;
    const float2 t1 = cadd(x[1], x[2]);
    const float2 t2 = make_float2(x[0].x - 0.5f * t1.x, x[0].y - 0.5f * t1.y);
    const float2 d = csub(x[1], x[2]);
    const float2 t3 = make_float2(d.x * S, d.y * S);
    x[0] = cadd(x[0], t1);
    x[1] = make_float2(t2.x + t3.y, t2.y - t3.x);
    x[2] = make_float2(t2.x - t3.y, t2.y + t3.x);
}

__device__ __forceinline__ void rt_dft5(float2 *x)
{
    constexpr float C1 = (float)0.30901699437494742410, C2 = (float)-0.80901699437494742410;
    constexpr float S1 = (float)0.95105651629515357212, S2 = (float)0.58778525229247312917;
    const float2 a1 = cadd(x[1], x[4]), a2 = cadd(x[2], x[3]);
    const float2 b1 = csub(x[1], x[4]), b2 = csub(x[2], x[3]);
    const float2 x0 = x[0];
    const float2 m1 = make_float2(x0.x + C1 * a1.x + C2 * a2.x, x0.y + C1 * a1.y + C2 * a2.y);
    const float2 m2 = make_float2(x0.x + C2 * a1.x + C1 * a2.x, x0.y + C2 * a1.y + C1 * a2.y);
    const float2 n1 = make_float2(S1 * b1.x + S2 * b2.x, S1 * b1.y + S2 * b2.y);
    const float2 n2 = make_float2(S2 * b1.x - S1 * b2.x, S2 * b1.y - S1 * b2.y);
    x[0] = make_float2(x0.x + a1.x + a2.x, x0.y + a1.y + a2.y);
    x[1] = make_float2(m1.x + n1.y, m1.y - n1.x);
    x[4] = make_float2(m1.x - n1.y, m1.y + n1.x);
    x[2] = make_float2(m2.x + n2.y, m2.y - n2.x);
    x[3] = make_float2(m2.x - n2.y, m2.y + n2.x);
}

template <int R>
__device__ __forceinline__ void rt_prime(float2 *x)  // natural order in and out
{
    static_assert(R == 2 || R == 3 || R == 5 || R == 7, "prime radices");
    if constexpr (R == 2) {
        const float2 a = cadd(x[0], x[1]), b = csub(x[0], x[1]);
        x[0] = a;
        x[1] = b;
    } else if constexpr (R == 3) rt_dft3(x);
    else if constexpr (R == 5) rt_dft5(x);
    else rt_dft7(x);
}

// A composite radix R = R1 R2 in registers (Cooley-Tukey, everything a compile-time index): input n = R2 n1 + n2, output
// k = k1 + R1 k2;  R2 transforms of R1 points over n1, the products by exp(-2 pi i n2 k1 / R) as constants (mul_w), R1
// transforms of R2 points over n2.  4410 = 14 . 15 . 21 is three LDS round trips instead of six (2, 3, 3, 5, 7, 7).
template <int R1, int R2, int N2, int K1>
__device__ __forceinline__ void rt_comp_tw(float2 (&y)[R1])
{
    if constexpr (K1 < R1) {
        y[K1] = mul_w<R1 * R2, (N2 * K1) % (R1 * R2)>(y[K1]);
        rt_comp_tw<R1, R2, N2, K1 + 1>(y);
    }
}
template <int R1, int R2, int N2>
__device__ __forceinline__ void rt_comp_cols(float2 *x)
{
    if constexpr (N2 < R2) {
        float2 y[R1];
#pragma unroll
        for (int n1 = 0; n1 < R1; n1++) y[n1] = x[R2 * n1 + N2];
        rt_prime<R1>(y);
        rt_comp_tw<R1, R2, N2, 1>(y);  // (k1 = 0 and n2 = 0: factor 1)
#pragma unroll
        for (int k1 = 0; k1 < R1; k1++) x[R2 * k1 + N2] = y[k1];  // A[k1][n2] kept at R2 k1 + n2
        rt_comp_cols<R1, R2, N2 + 1>(x);
    }
}
template <int R1, int R2>
__device__ __forceinline__ void rt_dft_comp(float2 *x)
{
    rt_comp_cols<R1, R2, 0>(x);
    float2 o[R1 * R2];
#pragma unroll
    for (int k1 = 0; k1 < R1; k1++) {
        float2 z[R2];
#pragma unroll
        for (int n2 = 0; n2 < R2; n2++) z[n2] = x[R2 * k1 + n2];
        rt_prime<R2>(z);
#pragma unroll
        for (int k2 = 0; k2 < R2; k2++) o[k1 + R1 * k2] = z[k2];
    }
#pragma unroll
    for (int i = 0; i < R1 * R2; i++) x[i] = o[i];
}

template <int R>
__device__ __forceinline__ void rt_dft(float2 *x)
{
    if constexpr (R == 3) rt_dft3(x);
    else if constexpr (R == 5) rt_dft5(x);
    else if constexpr (R == 7) rt_dft7(x);
    else if constexpr (R == 6) rt_dft_comp<2, 3>(x);
    else if constexpr (R == 10) rt_dft_comp<2, 5>(x);
    else if constexpr (R == 14) rt_dft_comp<2, 7>(x);
    else if constexpr (R == 9) rt_dft_comp<3, 3>(x);
    else if constexpr (R == 15) rt_dft_comp<3, 5>(x);
    else if constexpr (R == 21) rt_dft_comp<3, 7>(x);
    else if constexpr (R == 25) rt_dft_comp<5, 5>(x);
    else dft_reg<R>(x);  // 2, 4, 8, 16: outputs bit-reversed
}
template <int R>
constexpr int rt_slot(int q)
{
    return (R == 2 || R == 4 || R == 8 || R == 16) ? cx_bitrev(q, R) : q;
}

// One Stockham pass of radix R: butterfly b (k = b mod P) takes in[b + j nb], j < R, multiplies input j >= 1 by T[k j], and
// stores output q at (b - k) R + k + q P.  FIRST: inputs from the frame in global memory (converted); LAST: outputs to the
// PSD (or the spectrum) in global memory.
template <int R, int IN, int OUT, bool FIRST, bool LAST>
__device__ __forceinline__ void rt_pass(const FftArgs &a, long long frame, int n, int P, const float2 *__restrict__ tw, const float2 *src,
                                        float2 *dst, int tid, Best &best)
{
    const int nb = n / R;
    for (int b = tid; b < nb; b += RT_T) {
        float2 v[R];
        if constexpr (FIRST) {
            if constexpr (IN == IN_I16) {
                const int *raw = reinterpret_cast<const int *>(a.in) + frame * n;
                int w[R];
#pragma unroll
                for (int j = 0; j < R; j++) w[j] = raw[b + j * nb];
#pragma unroll
                for (int j = 0; j < R; j++) {
                    const int si = java_short_add((int)(short)(w[j] & 0xffff), a.ic);  // JavaAudio.java:281-288
                    const int sq = java_short_add(w[j] >> 16, a.qc);
                    v[j] = make_float2(i16_to_float_java(si), i16_to_float_java(sq));
                }
            } else {
                const float2 *in = reinterpret_cast<const float2 *>(a.in) + frame * n;
#pragma unroll
                for (int j = 0; j < R; j++) v[j] = in[b + j * nb];
            }
        } else {
#pragma unroll
            for (int j = 0; j < R; j++) v[j] = src[b + j * nb];
        }
        const int k = b % P;
        if (P > 1) {
#pragma unroll
            for (int j = 1; j < R; j++) v[j] = cmul(v[j], tw[k * j]);
        }
        rt_dft<R>(v);
        const int j0 = (b - k) * R + k;
        if constexpr (!LAST) {
#pragma unroll
            for (int q = 0; q < R; q++) dst[j0 + q * P] = v[rt_slot<R>(q)];
        } else if constexpr (OUT == OUT_SPEC) {
            float2 *o = reinterpret_cast<float2 *>(a.out) + frame * n;
#pragma unroll
            for (int q = 0; q < R; q++) o[j0 + q * P] = v[rt_slot<R>(q)];
        } else {
            const float cf = (2.0f / (float)n) * (2.0f / (float)n);  // fft.java:203: cf = 2f / N, squared
            float *o = a.out + frame * (n + 2);
#pragma unroll
            for (int q = 0; q < R; q++) {
                const float2 x = v[rt_slot<R>(q)];
                const float db = 3.0102999566398120f * __log2f((x.x * x.x + x.y * x.y) * cf);  // fft.java:207
                const int bin = j0 + q * P;
                o[bin] = db;
                if (db > best.v || (db == best.v && bin < best.k)) {  // first strict maximum (fft.java:208-211)
                    best.v = db;
                    best.k = bin;
                }
            }
        }
    }
}

// A pass whose radix is a prime above 7 (round 5): one thread per OUTPUT, out[(b - k) r + k + q P] = sum_j v_j W[(j q) mod r] with
// v_j = in[b + j nb] (x T[k j] for j >= 1, P > 1) -- the DFT's definition, O(n r) a pass, image to image.  19 + 29 terms per
// output at n = 1102 instead of the 1102 of the O(n^2) kernel.
__device__ __forceinline__ void rt_pass_generic(int n, int P, int r, const float2 *__restrict__ tw, const float2 *__restrict__ wr,
                                                const float2 *src, float2 *dst, int tid)
{
    const int nb = n / r;
    for (int o = tid; o < n; o += RT_T) {
        const int q = o / nb, b = o - q * nb;
        const int k = b % P;
        // (the r terms are summed in double: a float sum's rounding grows with r -- 2 x 4099 samples: 1.3e-3 dB off in bins 60 dB
        //  below the peak -- and FP64 adds cost this chip what FP32 adds do)
        const float2 x0 = src[b];
        double ax = x0.x, ay = x0.y;
        int m = 0;
        for (int j = 1; j < r; j++) {
            m += q;
            if (m >= r) m -= r;
            float2 v = src[b + j * nb];
            if (P > 1) v = cmul(v, tw[k * j]);
            v = cmul(v, wr[m]);
            ax += (double)v.x;
            ay += (double)v.y;
        }
        dst[(b - k) * r + k + q * P] = make_float2((float)ax, (float)ay);
    }
}

// the frame into an LDS image, converted (a plan whose FIRST radix has no butterfly of its own: 143 = 11 . 13)
template <int IN>
__device__ __forceinline__ void rt_load(const FftArgs &a, long long frame, int n, float2 *dst, int tid)
{
    for (int t = tid; t < n; t += RT_T) {
        if constexpr (IN == IN_I16) {
            const int w = (reinterpret_cast<const int *>(a.in) + frame * n)[t];
            const int si = java_short_add((int)(short)(w & 0xffff), a.ic);
            const int sq = java_short_add(w >> 16, a.qc);
            dst[t] = make_float2(i16_to_float_java(si), i16_to_float_java(sq));
        } else {
            dst[t] = (reinterpret_cast<const float2 *>(a.in) + frame * n)[t];
        }
    }
}

// PSD / first maximum (or the spectrum) from an LDS image (a plan whose LAST radix has no butterfly of its own)
template <int OUT>
__device__ __forceinline__ void rt_epilogue(const FftArgs &a, long long frame, int n, const float2 *src, int tid, Best &best)
{
    const float cf = (2.0f / (float)n) * (2.0f / (float)n);
    for (int bin = tid; bin < n; bin += RT_T) {
        const float2 x = src[bin];
        if constexpr (OUT == OUT_SPEC) {
            (reinterpret_cast<float2 *>(a.out) + frame * n)[bin] = x;
        } else {
            const float db = 3.0102999566398120f * __log2f((x.x * x.x + x.y * x.y) * cf);  // fft.java:207
            (a.out + frame * (n + 2))[bin] = db;
            if (db > best.v || (db == best.v && bin < best.k)) {
                best.v = db;
                best.k = bin;
            }
        }
    }
}

// FULL: the composite radices as well.  They are a kernel of their own: a switch that holds a 25-point butterfly gives EVERY
// plan that kernel's register count (3200 = 16.8.5.5: 5.3 ms per 2^30 samples in the small kernel, 6.8 in the full one).
template <int IN, int OUT, bool FIRST, bool LAST, bool FULL>
__device__ __forceinline__ void rt_pass_any(int r, const FftArgs &a, long long frame, int n, int P, const float2 *tw, const float2 *src,
                                            float2 *dst, int tid, Best &best)
{
    switch (r) {
        case 16: rt_pass<16, IN, OUT, FIRST, LAST>(a, frame, n, P, tw, src, dst, tid, best); break;
        case 8: rt_pass<8, IN, OUT, FIRST, LAST>(a, frame, n, P, tw, src, dst, tid, best); break;
        case 4: rt_pass<4, IN, OUT, FIRST, LAST>(a, frame, n, P, tw, src, dst, tid, best); break;
        case 2: rt_pass<2, IN, OUT, FIRST, LAST>(a, frame, n, P, tw, src, dst, tid, best); break;
        case 3: rt_pass<3, IN, OUT, FIRST, LAST>(a, frame, n, P, tw, src, dst, tid, best); break;
        case 5: rt_pass<5, IN, OUT, FIRST, LAST>(a, frame, n, P, tw, src, dst, tid, best); break;
        case 7: rt_pass<7, IN, OUT, FIRST, LAST>(a, frame, n, P, tw, src, dst, tid, best); break;
        default:
            if constexpr (FULL) {
                switch (r) {
                    case 6: rt_pass<6, IN, OUT, FIRST, LAST>(a, frame, n, P, tw, src, dst, tid, best); break;
                    case 9: rt_pass<9, IN, OUT, FIRST, LAST>(a, frame, n, P, tw, src, dst, tid, best); break;
                    case 10: rt_pass<10, IN, OUT, FIRST, LAST>(a, frame, n, P, tw, src, dst, tid, best); break;
                    case 14: rt_pass<14, IN, OUT, FIRST, LAST>(a, frame, n, P, tw, src, dst, tid, best); break;
                    case 15: rt_pass<15, IN, OUT, FIRST, LAST>(a, frame, n, P, tw, src, dst, tid, best); break;
                    case 21: rt_pass<21, IN, OUT, FIRST, LAST>(a, frame, n, P, tw, src, dst, tid, best); break;
                    default: rt_pass<25, IN, OUT, FIRST, LAST>(a, frame, n, P, tw, src, dst, tid, best); break;
                }
            }
            break;
    }
}

template <int IN, int OUT, bool FULL>
__global__ __launch_bounds__(RT_T) void k_fft_rt(FftArgs a, RtPlan p)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int n = p.n;
    float2 *img0 = reinterpret_cast<float2 *>(smem);
    float2 *img1 = img0 + n;
    float *red_val = reinterpret_cast<float *>(img1 + n);
    int *red_idx = reinterpret_cast<int *>(red_val + RT_T / 64);
    const int tid = threadIdx.x;
    for (long long frame = blockIdx.x; frame < a.nframes; frame += gridDim.x) {
        Best best;
        best.v = -3.402823466e+38f;
        best.k = 0x7fffffff;
        float2 *src = img0, *dst = img1;
        int P = 1, q0 = 0;
        // (a one-pass plan is first and last at once: it is not given to this kernel, rt_plan asks for two)
        if (rt_special(p.rad[0])) {
            rt_pass_any<IN, OUT, true, false, FULL>(p.rad[0], a, frame, n, 1, a.tw, src, dst, tid, best);
            P = p.rad[0];
            q0 = 1;
        } else {
            rt_load<IN>(a, frame, n, dst, tid);
        }
        __syncthreads();
        const bool last_fused = rt_special(p.rad[p.np - 1]);
        const int qend = last_fused ? p.np - 1 : p.np;  // passes [q0, qend) go image to image
        for (int q = q0; q < qend; q++) {
            float2 *t = src;
            src = dst;
            dst = t;
            const int r = p.rad[q];
            if (rt_special(r))
                rt_pass_any<IN, OUT, false, false, FULL>(r, a, frame, n, P, a.tw + p.tw_off[q], src, dst, tid, best);
            else
                rt_pass_generic(n, P, r, a.tw + p.tw_off[q], a.tw + p.wr_off[q], src, dst, tid);
            P *= r;
            __syncthreads();
        }
        if (last_fused)
            rt_pass_any<IN, OUT, false, true, FULL>(p.rad[p.np - 1], a, frame, n, P, a.tw + p.tw_off[p.np - 1], dst, nullptr, tid, best);
        else
            rt_epilogue<OUT>(a, frame, n, dst, tid, best);
        if constexpr (OUT == OUT_PSD) {
            float bestv = best.v;
            int bestk = best.k;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const float ov = __shfl_xor(bestv, off, 64);
                const int ok = __shfl_xor(bestk, off, 64);
                if (ov > bestv || (ov == bestv && ok < bestk)) {
                    bestv = ov;
                    bestk = ok;
                }
            }
            if ((tid & 63) == 0) {
                red_val[tid >> 6] = bestv;
                red_idx[tid >> 6] = bestk;
            }
            __syncthreads();
            if (tid == 0) {
                for (int w = 1; w < RT_T / 64; w++) {
                    const float ov = red_val[w];
                    const int ok = red_idx[w];
                    if (ov > bestv || (ov == bestv && ok < bestk)) {
                        bestv = ov;
                        bestk = ok;
                    }
                }
                // fft.java:201-224: m starts at -Float.MAX_VALUE, p at -1; Hz in wrapping int arithmetic
                int pp = (bestv > -3.402823466e+38f) ? 2 * bestk : -1;
                const float m = (pp >= 0) ? bestv : -3.402823466e+38f;
                const int datlen = 2 * n;
                if (pp >= datlen / 2) pp -= datlen;
                const int hz = (int)((unsigned)pp * (unsigned)a.rate) / datlen;
                float *o = a.out + frame * (n + 2);
                o[n] = (float)hz;
                o[n + 1] = m;
            }
        }
        __syncthreads();  // the images and the reduction scratch are reused by the next frame
    }
}

// radix plan: 16, 8, 4, 2 while they divide (largest first), then 3s, 5s, 7s; false if n has another factor, is too small
// to need two passes, or does not fit two LDS images
bool rt_plan(int n, int *np_out, int *rad, int *tw_off, size_t *tw_count, int *wr_off = nullptr)
{
    if (n < 6 || n > RT_NMAX) return false;
    int m = n, c = 0;
    while (m % 16 == 0 && c < RT_MAXPASS) { rad[c++] = 16; m /= 16; }
    if (m % 8 == 0 && c < RT_MAXPASS) { rad[c++] = 8; m /= 8; }
    if (m % 4 == 0 && c < RT_MAXPASS) { rad[c++] = 4; m /= 4; }
    // the odd prime factors, largest first; a leftover 2 joins the largest, then largest x smallest while the product stays <= 25
    // (25 complex registers a butterfly) AND the pass keeps at least 192 butterflies for the workgroup's 256 threads (measured:
    // 2205 = 21.21.5 leaves 105 butterflies a pass and is slower than 3.3.5.7.7; 4410 = 14.21.15 against 2.3.3.5.7.7: 5.3 vs 8.2 ms
    // per 2^30 samples):  4410 -> 14, 21, 15;  8820 -> 4, 21, 21, 5;  2205 -> 9, 7, 7, 5;  3200 -> 16, 8, 5, 5
    int odd[16], no = 0;
    for (int q : {7, 5, 3})
        while (m % q == 0 && no < 16) { odd[no++] = q; m /= q; }
    const bool two = (m % 2 == 0);
    if (two) m /= 2;
    // what is left are primes above 7 (ascending): passes of that radix by the DFT's definition, behind the others
    int big[8], nbig = 0, big_sum = 0;
    for (int q = 11; m > 1; q += 2) {
        if (q * q > m) q = m;
        while (m % q == 0) {
            if (nbig == 8) return false;
            big[nbig++] = q;
            big_sum += q;
            m /= q;
        }
    }
    // ... while they are small beside n: a term of such a pass costs ~4-7 x a term of k_dft_any's tuned sum (measured per 2^30
    // samples: 2 x 4099: 15.8 s here against 4.3 s there; 97 x 101: 0.62 against ~5.1; 2 x 19 x 29: 0.06 against 0.58)
    if (8 * big_sum > n) return false;
    int lo = 0, hi = no - 1;
    const bool merge = n >= 4000;  // (below, the four- and five-pass plans of the small kernel are the faster ones)
    if (two) {
        if (merge && no > 0 && c < RT_MAXPASS && n / (2 * odd[lo]) >= 192) rad[c++] = 2 * odd[lo++];
        else if (c < RT_MAXPASS) rad[c++] = 2;
    }
    while (lo <= hi && c < RT_MAXPASS) {
        if (merge && lo < hi && odd[lo] * odd[hi] <= 25 && n / (odd[lo] * odd[hi]) >= 192) {
            rad[c++] = odd[lo] * odd[hi];
            lo++;
            hi--;
        } else {
            rad[c++] = odd[lo++];
        }
    }
    if (lo <= hi) return false;
    for (int i = 0; i < nbig; i++) {
        if (c == RT_MAXPASS) return false;
        rad[c++] = big[i];
    }
    if (c < 2) return false;
    size_t o = 0;
    int P = 1;
    for (int p = 0; p < c; p++) {
        tw_off[p] = (int)o;
        if (p >= 1) o += (size_t)P * rad[p];
        P *= rad[p];
    }
    for (int p = 0; p < c; p++) {
        if (wr_off) wr_off[p] = 0;
        if (!rt_special(rad[p])) {
            if (wr_off) wr_off[p] = (int)o;
            o += (size_t)rad[p];
        }
    }
    *np_out = c;
    *tw_count = o;
    return true;
}

// T_p[m] = exp(-2 pi i m/(P r)), m < P r: long double, one rounding to float
void rt_twiddles(int n, std::vector<float2> &w)
{
    int np = 0, rad[RT_MAXPASS], off[RT_MAXPASS], woff[RT_MAXPASS];
    size_t cnt = 0;
    w.clear();
    if (!rt_plan(n, &np, rad, off, &cnt, woff)) return;
    w.resize(cnt);
    for (int p = 0; p < np; p++)
        if (!rt_special(rad[p]))
            for (int m = 0; m < rad[p]; m++) {
                const long double ang = -2.0L * 3.14159265358979323846264338327950288L * (long double)m / (long double)rad[p];
                w[(size_t)woff[p] + m] = make_float2((float)cosl(ang), (float)sinl(ang));
            }
    int P = rad[0];
    for (int p = 1; p < np; p++) {
        const int len = P * rad[p];
        for (int m = 0; m < len; m++) {
            const long double ang = -2.0L * 3.14159265358979323846264338327950288L * (long double)m / (long double)len;
            w[(size_t)off[p] + m] = make_float2((float)cosl(ang), (float)sinl(ang));
        }
        P *= rad[p];
    }
}

int rt_launch(const FftArgs &a, int n, int in_kind, int out_kind, int num_cu, hipStream_t st)
{
    RtPlan p;
    size_t cnt = 0;
    p.n = n;
    if (!rt_plan(n, &p.np, p.rad, p.tw_off, &cnt, p.wr_off)) {
        set_error("fft: no run-time radix plan for n=%d", n);
        return JSDR_ERR;
    }
    for (int i = p.np; i < RT_MAXPASS; i++) p.rad[i] = 1, p.tw_off[i] = 0, p.wr_off[i] = 0;
    const size_t lds = sizeof(float2) * 2 * (size_t)n + (sizeof(float) + sizeof(int)) * (RT_T / 64) + 16;
    const long long per_cu = (long long)(160 * 1024 / lds) < 8 ? (long long)(160 * 1024 / lds) : 8;
    const long long cap = (long long)num_cu * (per_cu < 1 ? 1 : per_cu) * 4;
    const unsigned grid = (unsigned)(a.nframes < cap ? a.nframes : cap);
    auto go = [&](auto kern) -> int {
        JSDR_LDS_ATTR(kern, lds);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(RT_T), lds, st, a, p);
        JSDR_LAUNCH_CHECK();
        return JSDR_OK;
    };
    bool full = false;
    for (int i = 0; i < p.np; i++) full |= rt_special(p.rad[i]) && (p.rad[i] == 6 || p.rad[i] > 8) && p.rad[i] != 16;
    if (full) {
        if (in_kind == IN_I16 && out_kind == OUT_PSD) return go(k_fft_rt<IN_I16, OUT_PSD, true>);
        if (in_kind == IN_F32 && out_kind == OUT_PSD) return go(k_fft_rt<IN_F32, OUT_PSD, true>);
        if (in_kind == IN_F32 && out_kind == OUT_SPEC) return go(k_fft_rt<IN_F32, OUT_SPEC, true>);
        if (in_kind == IN_I16 && out_kind == OUT_SPEC) return go(k_fft_rt<IN_I16, OUT_SPEC, true>);
    } else {
        if (in_kind == IN_I16 && out_kind == OUT_PSD) return go(k_fft_rt<IN_I16, OUT_PSD, false>);
        if (in_kind == IN_F32 && out_kind == OUT_PSD) return go(k_fft_rt<IN_F32, OUT_PSD, false>);
        if (in_kind == IN_F32 && out_kind == OUT_SPEC) return go(k_fft_rt<IN_F32, OUT_SPEC, false>);
        if (in_kind == IN_I16 && out_kind == OUT_SPEC) return go(k_fft_rt<IN_I16, OUT_SPEC, false>);
    }
    set_error("fft: no run-time-plan kernel for in=%d out=%d", in_kind, out_kind);
    return JSDR_ERR;
}

bool rt_supported(int n)
{
    int np, rad[RT_MAXPASS], off[RT_MAXPASS];
    size_t cnt;
    return rt_plan(n, &np, rad, off, &cnt);
}

}  // namespace jsdr
