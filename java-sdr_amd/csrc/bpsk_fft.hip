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
// state is sequential from frame to frame).  The frame lives in LDS as double2[N] (32 KB at N=2048) next to the
// twiddle table (16 KB) and the |X| / boxcar arrays; every FFT stage is 4 butterflies per thread between two
// barriers.  FP64-issue bound: ~225k FP64 ops per frame.
#include "bpsk_fft.h"
#include <math.h>

namespace jsdr {

// one radix-2 DIT stage over X[n] (in place), `half` = distance between butterfly wings
__device__ __forceinline__ void fft_stage(double2 *X, const double2 *W, int n, int half, int step, bool inverse, int tid)
{
    for (int idx = tid; idx < n / 2; idx += 256) {
        const int j = idx & (half - 1);
        const int ia = ((idx - j) << 1) + j;
        const int ib = ia + half;
        const double2 w = W[j * step];
        const double wr = w.x;
        const double wi = inverse ? -w.y : w.y;
        const double2 b = X[ib];
        const double p1 = wr * b.x, p2 = wi * b.y, p3 = wr * b.y, p4 = wi * b.x;
        const double tr = p1 - p2;
        const double ti = p3 + p4;
        const double2 a = X[ia];
        X[ia] = make_double2(a.x + tr, a.y + ti);
        X[ib] = make_double2(a.x - tr, a.y - ti);
    }
}

__device__ __forceinline__ void fft_inplace(double2 *X, const double2 *W, int n, bool inverse, int tid)
{
    for (int half = 1; half < n; half <<= 1) {
        __syncthreads();
        fft_stage(X, W, n, half, n / (2 * half), inverse, tid);
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_front_fft(FftFrontArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int n = a.n;
    double2 *X = reinterpret_cast<double2 *>(smem);                 // [n]
    double2 *W = X + n;                                              // [n/2]
    double *P = reinterpret_cast<double *>(W + n / 2);               // [n/2]  |X| of the lower half
    double *A = P + n / 2;                                           // [n/2]  boxcar sums (zero outside the searched band)
    double *sc = A + n / 2;                                          // [512]
    double *hist = sc + 512;                                         // [32]
    double *redv = hist + 32;                                        // [4] per-wave best value
    int *redi = reinterpret_cast<int *>(redv + 4);                   // [4] per-wave best index, [4] = centreBin broadcast
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = blockIdx.x;
    for (int i = tid; i < n / 2; i += 256) W[i] = a.tw[i];
    for (int i = tid; i < 512; i += 256) sc[i] = a.sincos[i];
    FftFrontState *sp = &a.st[s];
    if (tid < 26) hist[tid] = sp->hist[tid];
    double avePeakPower = sp->avePeakPower, aveCentreBin = sp->aveCentreBin;
    int centreBin = sp->centreBin;
    // :399-402 -- float expressions widened to double
    const double CFREQ_INV = (double)(1.0F - (2.0F / (1 + 1))), CFREQ_AVG = (double)(2.0F / (1 + 1));
    const double PSD_INV = (double)(1.0F - (2.0F / (10 + 1))), PSD_AVG = (double)(2.0F / (10 + 1));
    const double HOWARD = 0.9 * 32768.0;
    const int beg = a.do_up ? n / 4 : 0;
    const int end = a.do_up ? n / 2 : n / 4;
    const int D = a.decim;
    const double norm = 1.0 / (double)n;
    const int *raw = a.raw + (long long)s * a.stride_pairs;
    const float2 *rawf = a.rawf + (long long)s * a.stride_pairs;
    double2 *dm = a.dm + (long long)s * a.dm_stride;
    __syncthreads();

    for (int f = 0; f < a.nframes; f++) {
        // ---- frame -> LDS in bit-reversed order (:416-421)
        for (int t = tid; t < n; t += 256) {
            double di, dq;
            const long long g = (long long)f * n + t;
            if (a.rawf) {
                const float2 v = rawf[g];
                di = (double)v.x;
                dq = (double)v.y;
            } else {
                const int w = raw[g];
                di = (double)i16_to_float_java(java_short_add((int)(short)(w & 0xffff), a.ic));
                dq = (double)i16_to_float_java(java_short_add(w >> 16, a.qc));
            }
            X[__brev((unsigned)t) >> (32 - a.logn)] = make_double2(di, dq);
        }
        fft_inplace(X, W, n, false, tid);  // :422-423
        // ---- |X| (:425-427), cleared boxcar array
        for (int i = tid; i < n / 2; i += 256) {
            const double2 v = X[i];
            P[i] = sqrt(v.x * v.x + v.y * v.y);
            A[i] = 0.0;
        }
        __syncthreads();
        // ---- 100-wide boxcar, summed j ascending for every i (:433-437); first maximum (:439-442)
        double bestv = 0.0;  // maxBin starts at 0.0, binPos at -1
        int besti = -1;
        for (int i = beg + 75 + tid; i < end - 75; i += 256) {
            double acc = 0.0;
            for (int j = i - 50; j < i + 50; j++) acc += P[j];
            A[i] = acc;
            if (bestv < acc) {  // within a thread i ascends: strict '<' keeps the first maximum
                bestv = acc;
                besti = i;
            }
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
            double maxBin = 0.0;
            int binPos = -1;
            for (int w = 0; w < 4; w++) {
                const double ov = redv[w];
                const int oi = redi[w];
                if (oi >= 0 && (ov > maxBin || (ov == maxBin && (binPos < 0 || oi < binPos)))) {
                    maxBin = ov;
                    binPos = oi;
                }
            }
            // :444-453
            if (centreBin < 0) centreBin = 0;
            if (centreBin > end - 1) centreBin = end - 1;
            avePeakPower = (PSD_AVG * A[centreBin]) + (PSD_INV * avePeakPower);
            if (maxBin > (avePeakPower / 4) * 5 && binPos > 0) {
                aveCentreBin = (CFREQ_AVG * (double)(float)binPos) + (CFREQ_INV * aveCentreBin);
                centreBin = (int)(aveCentreBin + (double)1.0F);
            }
            if (centreBin < 102) centreBin = 102;
            redi[4] = centreBin;
        }
        __syncthreads();
        const int cb = redi[4];
        // ---- 204 bins around the centre move to bin 0 of a zeroed array (:458), bit-reversed for the DIT network
        double2 keep = make_double2(0.0, 0.0);
        if (tid < 204) keep = X[cb - 102 + tid];
        __syncthreads();
        for (int i = tid; i < n; i += 256) X[i] = make_double2(0.0, 0.0);
        __syncthreads();
        if (tid < 204) X[__brev((unsigned)tid) >> (32 - a.logn)] = keep;
        fft_inplace(X, W, n, true, tid);  // :459 complexInverse(fftRev, true)
        for (int i = tid; i < n; i += 256) X[i].x = X[i].x * norm;  // only the real parts are used (:462)
        __syncthreads();
        // ---- RxDownSample(re, re) (:461-463, :470-492): outputs whose window ends inside this frame
        {
            const long long t0 = (long long)f * n;  // call-relative index of the frame's first sample
            // outputs j with t0 <= first_out + D*j < t0 + n
            long long jlo = (t0 - a.first_out + D - 1) / D;
            if (t0 <= a.first_out) jlo = 0;
            for (long long j = jlo + tid;; j += 256) {
                const long long te = (long long)a.first_out + (long long)D * j;  // window end, call-relative
                if (te >= t0 + n || j >= a.nds) break;
                const int e = (int)(te - t0);  // 0..n-1 within the frame
                double fi = 0.0;
#pragma unroll
                for (int k = 0; k < 27; k++) {  // newest first (:479-483)
                    const int idx = e - k;
                    const double v = (idx >= 0) ? X[idx].x : hist[26 + idx];
                    fi += v * a.ds_taps[k];
                }
                const double o = fi * HOWARD;  // fi == fq: both rails get the same samples
                const int kv = a.kvco[j];
                dm[64 + j] = make_double2(o * sc[kv], o * sc[256 + kv]);  // :515-516
            }
        }
        __syncthreads();
        if (tid < 26) hist[tid] = X[n - 26 + tid].x;
        __syncthreads();
    }
    if (tid < 26) sp->hist[tid] = hist[tid];
    if (tid == 0) {
        sp->avePeakPower = avePeakPower;
        sp->aveCentreBin = aveCentreBin;
        sp->centreBin = centreBin;
    }
}

int launch_front_fft(const FftFrontArgs &a, int nstreams, hipStream_t st)
{
    const size_t lds = sizeof(double2) * ((size_t)a.n + a.n / 2) + sizeof(double) * ((size_t)a.n + 512 + 32 + 4) + 64;
    static size_t attr_for = 0;
    if (attr_for < lds) {
        JSDR_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_front_fft),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_for = lds;
    }
    hipLaunchKernelGGL(k_front_fft, dim3((unsigned)nstreams), dim3(256), lds, st, a);
    JSDR_LAUNCH_CHECK();
    return JSDR_OK;
}

// twiddle table identical to oracle/o_fft.c jo_fft_twiddles_f64 (same libm, exact values on the axes)
void fft_twiddles_f64(std::vector<double2> &w, int n)
{
    w.resize((size_t)n / 2);
    // long double + one rounding: independent of sin/cos -> sincos / vector-libm rewrites by the host compiler
    for (int k = 0; k < n / 2; k++) {
        long double ang = 2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)n;
        w[k] = make_double2((double)cosl(ang), (double)(-sinl(ang)));
    }
    w[0] = make_double2(1.0, -0.0);
    if (n >= 4) w[n / 4] = make_double2(0.0, -1.0);
}

}  // namespace jsdr
