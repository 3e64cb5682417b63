// fft_any.hip -- fft.receive (fft.java:190-228) for ANY frame size the specialised kernels do not cover.
//
// The reference's audio-rate is a free integer (JavaAudio.java:49,59) and its frame is rate / 10 samples: 44.1 kHz gives
// n = 4410 = 2 . 3^2 . 5 . 7^2 (the reference's own sine4410.wav), 22.05 kHz 2205, 32 kHz 3200, 176.4 kHz 17640 ...
// JTransforms takes any n (fft.java:67,194).  The Stockham kernels exist for the powers of two 64..8192 (fft_psd.hip)
// and for the three default frames 4800 / 9600 / 19200 (fft_mixed.hip); every other n comes here: the DFT itself,
//     X[k] = sum_t x[t] exp(-2 pi i k t / n),
// one output bin per thread, the frame (converted to float as JavaAudio does) in LDS and read as a broadcast, the sum and
// the twiddle in double: w_{t+1} = w_t . w_1 by recurrence, re-seeded from sincospi of the exactly reduced k t mod n every
// 128 steps, so the spectrum is the correctly rounded float of the exact DFT to within a few units of 1e-13 of the frame's
// norm -- far inside the 1e-5 bar, whatever the factors of n.  O(n^2): 10 FP64 operations per (bin, sample); at n = 4410
// about 0.7 Gsamples/s -- a live stream needs 44 ksamples/s.  The PSD, first-maximum and Hz rules are fft.java:196-224 as
// the oracle restates them (float products, log10 in double).
#include "fft_common.h"
#include <math.h>

namespace jsdr {

enum { DFT_T = 256 };

template <int IN, int OUT>
__global__ __launch_bounds__(DFT_T) void k_dft_any(FftArgs a, int n, int bins_per_group)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float2 *xs = reinterpret_cast<float2 *>(smem);
    const long long f = blockIdx.x;
    const int tid = threadIdx.x;
    if constexpr (IN == IN_I16) {
        const int *raw = reinterpret_cast<const int *>(a.in) + f * n;
        for (int t = tid; t < n; t += DFT_T) {
            const int w = raw[t];
            const int si = java_short_add((int)(short)(w & 0xffff), a.ic);  // JavaAudio.java:281-288
            const int sq = java_short_add(w >> 16, a.qc);
            xs[t] = make_float2(i16_to_float_java(si), i16_to_float_java(sq));
        }
    } else {
        const float2 *in = reinterpret_cast<const float2 *>(a.in) + f * n;
        for (int t = tid; t < n; t += DFT_T) xs[t] = in[t];
    }
    __syncthreads();
    const int k0 = blockIdx.y * bins_per_group;
    const int k1 = k0 + bins_per_group < n ? k0 + bins_per_group : n;
    const double inv_n = 1.0 / (double)n;
    for (int k = k0 + tid; k < k1; k += DFT_T) {
        double cr, ci;  // w_1 = exp(-2 pi i k / n)
        sincospi(-2.0 * (double)k * inv_n, &ci, &cr);
        double wr = 1.0, wi = 0.0, re = 0.0, im = 0.0;
        for (int t0 = 0; t0 < n; t0 += 128) {
            {   // re-seed: k t0 mod n is exact, its cosine and sine correctly rounded to ~1 ulp
                const long long m = ((long long)k * t0) % n;
                sincospi(-2.0 * (double)m * inv_n, &wi, &wr);
            }
            const int t1 = t0 + 128 < n ? t0 + 128 : n;
            for (int t = t0; t < t1; t++) {
                const float2 x = xs[t];
                re += (double)x.x * wr - (double)x.y * wi;
                im += (double)x.x * wi + (double)x.y * wr;
                const double nr = wr * cr - wi * ci;
                wi = wr * ci + wi * cr;
                wr = nr;
            }
        }
        const float fr = (float)re, fi = (float)im;
        if constexpr (OUT == OUT_SPEC) {
            reinterpret_cast<float2 *>(a.out)[f * n + k] = make_float2(fr, fi);
        } else {
            float cf = 2.0f / (float)n;  // fft.java:196-197
            cf = __fmul_rn(cf, cf);
            const float pw = __fmul_rn(__fadd_rn(__fmul_rn(fr, fr), __fmul_rn(fi, fi)), cf);  // :207, float products
            a.out[f * (n + 2) + k] = 10.0f * (float)log10((double)pw);
        }
    }
}

// first strictly greater maximum (fft.java:208-211), bin -> Hz in Java int arithmetic (:214-221)
__global__ __launch_bounds__(DFT_T) void k_psd_fin(float *psd, int n, int rate)
{
    __shared__ float bv[DFT_T];
    __shared__ int bk[DFT_T];
    float *p = psd + (long long)blockIdx.x * (n + 2);
    float best = -3.402823466e+38f;  // -Float.MAX_VALUE (:199)
    int kb = -1;
    for (int k = threadIdx.x; k < n; k += DFT_T) {  // ascending k per thread: a later equal value does not replace
        const float v = p[k];
        if (best < v) {
            best = v;
            kb = k;
        }
    }
    bv[threadIdx.x] = best;
    bk[threadIdx.x] = kb;
    __syncthreads();
    for (int off = DFT_T / 2; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) {
            const float v = bv[threadIdx.x + off];
            const int k = bk[threadIdx.x + off];
            // the winner is the larger value; among equals the smaller bin (the serial loop's first strict maximum); a bin of
            // -1 is "no value above -MAX yet"
            const bool take = k >= 0 && (bk[threadIdx.x] < 0 || v > bv[threadIdx.x] || (v == bv[threadIdx.x] && k < bk[threadIdx.x]));
            if (take) {
                bv[threadIdx.x] = v;
                bk[threadIdx.x] = k;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int datlen = 2 * n;
        int pp = bk[0] < 0 ? -1 : 2 * bk[0];  // (-1: no value above -MAX, e.g. an all-zero frame -- the reference's p stays -1, :199)
        int hz;
        if (pp < datlen / 2) {
            hz = (int)((unsigned)pp * (unsigned)rate) / datlen;
        } else {
            pp -= datlen;
            hz = (int)((unsigned)pp * (unsigned)rate) / datlen;
        }
        p[n] = (float)hz;
        p[n + 1] = bv[0];
    }
}

bool dft_any_supported(int n) { return n >= 2 && n <= 20000; }  // (the frame as float pairs fits a workgroup's LDS)

int dft_any_launch(const FftArgs &a, int n, int in_kind, int out_kind, int num_cu, hipStream_t st)
{
    // bins per workgroup: one frame at a time (receive()) spreads over the chip, a batch keeps whole frames together
    long long groups = 1;
    if (a.nframes < 2LL * num_cu) {
        groups = (2LL * num_cu + a.nframes - 1) / a.nframes;
        const long long maxg = (n + DFT_T - 1) / DFT_T;
        if (groups > maxg) groups = maxg;
    }
    const int bins = (int)((n + groups - 1) / groups);
    const size_t lds = sizeof(float2) * (size_t)n;
    const dim3 grid((unsigned)a.nframes, (unsigned)((n + bins - 1) / bins)), block(DFT_T);
#define JSDR_DFT_LAUNCH(IN, OUT)                                                                                      \
    do {                                                                                                              \
        JSDR_LDS_ATTR((k_dft_any<IN, OUT>), lds);                                                                     \
        hipLaunchKernelGGL((k_dft_any<IN, OUT>), grid, block, lds, st, a, n, bins);                                   \
    } while (0)
    JSDR_REQUIRE(a.nframes <= 0x7fffffffLL, "fft: too many frames for one launch");
    if (in_kind == IN_I16 && out_kind == OUT_PSD) JSDR_DFT_LAUNCH(IN_I16, OUT_PSD);
    else if (in_kind == IN_F32 && out_kind == OUT_PSD) JSDR_DFT_LAUNCH(IN_F32, OUT_PSD);
    else if (in_kind == IN_F32 && out_kind == OUT_SPEC) JSDR_DFT_LAUNCH(IN_F32, OUT_SPEC);
    else JSDR_DFT_LAUNCH(IN_I16, OUT_SPEC);
#undef JSDR_DFT_LAUNCH
    JSDR_LAUNCH_CHECK();
    if (out_kind == OUT_PSD) {
        hipLaunchKernelGGL(k_psd_fin, dim3((unsigned)a.nframes), dim3(DFT_T), 0, st, a.out, n, a.rate);
        JSDR_LAUNCH_CHECK();
    }
    return JSDR_OK;
}

}  // namespace jsdr
