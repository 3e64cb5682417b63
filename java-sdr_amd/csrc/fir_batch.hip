// fir_batch.hip -- batched complex FIR + decimate over HBM-resident int16 IQ streams (BASELINE config 3).
//
// The arithmetic of the demodulator's two filter stages, as a stand-alone operator (SURVEY.md 8b's
// `jsdr_fir_* ... batch filter+decimate`):
//   out[s][j] = scale * SUM_{a = 0 .. ntaps-1} x[s][decim*(j+1) - 1 - a] * taps[a]        (I and Q rails separately)
// with x = (double)((float)int16 / 32767f) (JavaAudio.java:281-288, FUNcubeBPSKDemod.java:372-373), samples before the
// start of the batch zero (a cleared delay line, as `new double[27][2]`), the sum taken NEWEST SAMPLE FIRST, every
// product and every addition rounded separately -- i.e. RxDownSample's loop (FUNcubeBPSKDemod.java:466-492: 27 taps,
// decimate rate/9600, x HOWARD_FUDGE_FACTOR) for any tap count <= 128 and any decimation; with the 65 dmFilter taps and
// decimation 1 it is the matched filter's sum in age order (:517-523 adds the same 65 products in ring-slot order; the
// demodulator's own kernels keep that order, this operator is the generic low-pass + decimate), with 21 taps
// fir.java's window (:198-211) on both rails.  Exact-order FP64: compiled with -ffp-contract=off.
//
// Mapping: a lane owns R consecutive outputs and reads ITS OWN window of D*(R-1)+NT samples straight into registers
// (4-byte aligned 16-byte loads, newest quad first), converts each sample once (packed FP32 pipe) and walks newest ->
// oldest with the R accumulator pairs in registers; the tap of (sample, output) is a compile-time index into the
// kernel-argument tap array (scalar operands).  One 16-byte store per output.  Bound: FP64 issue (2*NT/D products and
// sums per input sample and rail); HBM traffic 4 B read + 16/D B written per sample.
#include "common.h"

namespace jsdr {

enum { FIRB_MAX_TAPS = 128 };

struct FirBatchArgs {
    const int *raw;            // int16 pairs as dwords, [S][stride]
    long long stride_pairs;
    int nsamples;
    double2 *out;              // [S][out_stride]
    long long out_stride;
    int nout;
    int ntaps, decim;          // generic kernel only
    double scale;
    double taps[FIRB_MAX_TAPS];
};

template <int NT, int D, int R>
__global__ __launch_bounds__(256) void k_fir_batch(FirBatchArgs a)
{
    constexpr int NS = D * (R - 1) + NT, NSQ = (NS + 3) / 4;
    const int s = blockIdx.y;
    const int *raw = a.raw + (long long)s * a.stride_pairs;
    double2 *out = a.out + (long long)s * a.out_stride;
    const int njobs = (a.nout + R - 1) / R;
    // a wave's 64 jobs produce 64 R consecutive outputs, R consecutive ones per lane: they leave through a wave-private
    // LDS tile, transposed, as R fully coalesced 1 KB stores (one 16-byte store per output and lane at a 16 R-byte lane
    // stride cost the kernel 2.3 of 9.9 ms: 7.5 ms with every lane storing to one place)
    __shared__ double2 tileL[4][64 * R];
    double2 *tile = tileL[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    // (wave-uniform trip count: the lanes of a wave's last round that have no job walk a window of zeros and store nothing)
    for (int jobw = blockIdx.x * blockDim.x + (threadIdx.x & ~63); jobw < njobs; jobw += gridDim.x * blockDim.x) {
        const int job = jobw + lane;
        const int j0 = R * job;                    // first output of the job
        const int n0 = D * (j0 + 1) - NT;          // first window sample: the oldest tap of output j0
        const int last = n0 + 4 * NSQ - 1;
        int4 W[NSQ];
        if (n0 >= 0 && last < a.nsamples) {
#pragma unroll
            for (int q = NSQ - 1; q >= 0; q--) W[q] = *reinterpret_cast<const int4 *>(raw + n0 + 4 * q);
        } else {  // the batch's edges: samples before 0 are zero, reads beyond the end are not used by a stored output
#pragma unroll
            for (int q = NSQ - 1; q >= 0; q--) {
                int w[4];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int n = n0 + 4 * q + t;
                    w[t] = (n >= 0 && n < a.nsamples) ? raw[n] : 0;
                }
                W[q] = make_int4(w[0], w[1], w[2], w[3]);
            }
        }
        double ai[R], aq[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            ai[r] = 0.0;
            aq[r] = 0.0;
        }
#pragma unroll
        for (int q = NSQ - 1; q >= 0; q--) {
            const int4 w4 = W[q];
#pragma unroll
            for (int t = 3; t >= 0; t--) {
                const int m = 4 * q + t;
                if (m < NS) {
                    const int w = (t == 0) ? w4.x : (t == 1) ? w4.y : (t == 2) ? w4.z : w4.w;
                    double di, dq;
                    fm_convert(w, 0, 0, false, di, dq);
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        // output j0+r ends at window sample D*r + NT-1; sample m has age D*r + NT-1 - m there
                        const int age = D * r + NT - 1 - m;
                        if (age >= 0 && age < NT) {
                            const double tp = a.taps[age];
                            ai[r] += di * tp;
                            aq[r] += dq * tp;
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < R; r++) asm volatile("" : "+v"(ai[r]), "+v"(aq[r])::"memory");
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < R; r++) tile[R * lane + r] = make_double2(ai[r] * a.scale, aq[r] * a.scale);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int jw0 = R * jobw;  // the wave's first output
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int e = 64 * r + lane;
            if (jw0 + e < a.nout) out[jw0 + e] = tile[e];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // the tile is rewritten by the wave's next round
    }
}

// any tap count <= 128, any decimation: one output per thread, run-time loops, the same order
__global__ __launch_bounds__(256) void k_fir_batch_generic(FirBatchArgs a)
{
    __shared__ double tp[FIRB_MAX_TAPS];
    for (int i = threadIdx.x; i < a.ntaps; i += blockDim.x) tp[i] = a.taps[i];
    __syncthreads();
    const int s = blockIdx.y;
    const int *raw = a.raw + (long long)s * a.stride_pairs;
    double2 *out = a.out + (long long)s * a.out_stride;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < a.nout; j += gridDim.x * blockDim.x) {
        const long long newest = (long long)a.decim * (j + 1) - 1;
        double fi = 0.0, fq = 0.0;
        for (int age = 0; age < a.ntaps; age++) {
            const long long n = newest - age;
            double di, dq;
            fm_convert(n >= 0 ? raw[n] : 0, 0, 0, false, di, dq);
            fi += di * tp[age];
            fq += dq * tp[age];
        }
        out[j] = make_double2(fi * a.scale, fq * a.scale);
    }
}

template <int NT, int D, int R>
static void launch_fir_batch(const FirBatchArgs &a, int nstreams, hipStream_t st)
{
    const int njobs = (a.nout + R - 1) / R;
    int gx = (njobs + 255) / 256;
    if (gx > 4096) gx = 4096;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL((k_fir_batch<NT, D, R>), dim3((unsigned)gx, (unsigned)nstreams), dim3(256), 0, st, a);
}

}  // namespace jsdr

using namespace jsdr;

extern "C" int jsdr_fir_batch_decimate_i16(const int16_t *raw_dev, int nstreams, int64_t stream_stride_i16, int64_t nsamples,
                                           const double *taps_host, int ntaps, int decim, double scale, double *out_dev,
                                           int64_t out_stride_pairs, int64_t *nout, void *stream)
{
    JSDR_REQUIRE(raw_dev && taps_host && out_dev, "jsdr_fir_batch_decimate_i16: null argument");
    JSDR_REQUIRE(nstreams > 0 && nstreams <= 65535, "jsdr_fir_batch_decimate_i16: %d streams", nstreams);
    JSDR_REQUIRE(ntaps >= 1 && ntaps <= FIRB_MAX_TAPS, "jsdr_fir_batch_decimate_i16: %d taps (1..%d)", ntaps, (int)FIRB_MAX_TAPS);
    JSDR_REQUIRE(decim >= 1, "jsdr_fir_batch_decimate_i16: decimation %d", decim);
    JSDR_REQUIRE(nsamples >= 0 && nsamples <= 0x3fffffffLL, "jsdr_fir_batch_decimate_i16: %lld samples per stream", (long long)nsamples);
    JSDR_REQUIRE((stream_stride_i16 & 1) == 0 && (nstreams == 1 || stream_stride_i16 >= 2 * nsamples),
                 "jsdr_fir_batch_decimate_i16: stream stride %lld too small for %lld samples", (long long)stream_stride_i16,
                 (long long)nsamples);
    const int64_t no = nsamples / decim;  // an output every `decim` inputs, the first after `decim` of them (:476)
    if (nout) *nout = no;
    JSDR_REQUIRE(nstreams == 1 || out_stride_pairs >= no, "jsdr_fir_batch_decimate_i16: output stride %lld < %lld outputs",
                 (long long)out_stride_pairs, (long long)no);
    if (no == 0) return JSDR_OK;
    FirBatchArgs a;
    a.raw = reinterpret_cast<const int *>(raw_dev);
    a.stride_pairs = stream_stride_i16 / 2;
    a.nsamples = (int)nsamples;
    a.out = reinterpret_cast<double2 *>(out_dev);
    a.out_stride = out_stride_pairs;
    a.nout = (int)no;
    a.ntaps = ntaps;
    a.decim = decim;
    a.scale = scale;
    for (int i = 0; i < FIRB_MAX_TAPS; i++) a.taps[i] = i < ntaps ? taps_host[i] : 0.0;
    hipStream_t st = as_stream(stream);
    // the register-blocked kernels exist for exactly these (tap count, decimation) PAIRS; every other shape takes the
    // generic kernel (a combined key such as ntaps * 100 + decim is not unique: (26, 101) would have run <27, 1>)
    bool blocked = true;
    if (ntaps == 27 && decim == 1) launch_fir_batch<27, 1, 8>(a, nstreams, st);
    else if (ntaps == 27 && decim == 10) launch_fir_batch<27, 10, 4>(a, nstreams, st);   // dsFilter at 96 kHz
    else if (ntaps == 27 && decim == 20) launch_fir_batch<27, 20, 4>(a, nstreams, st);   // dsFilter at 192 kHz
    else if (ntaps == 65 && decim == 1) launch_fir_batch<65, 1, 8>(a, nstreams, st);     // dmFilter
    else if (ntaps == 65 && decim == 10) launch_fir_batch<65, 10, 4>(a, nstreams, st);
    else if (ntaps == 65 && decim == 20) launch_fir_batch<65, 20, 2>(a, nstreams, st);
    else if (ntaps == 21 && decim == 1) launch_fir_batch<21, 1, 8>(a, nstreams, st);     // fir.java's 21-tap window
    else if (ntaps == 21 && decim == 10) launch_fir_batch<21, 10, 4>(a, nstreams, st);
    else if (ntaps == 21 && decim == 20) launch_fir_batch<21, 20, 4>(a, nstreams, st);
    else blocked = false;
    if (!blocked) {
        int gx = (int)((no + 255) / 256);
        if (gx > 4096) gx = 4096;
        hipLaunchKernelGGL(k_fir_batch_generic, dim3((unsigned)gx, (unsigned)nstreams), dim3(256), 0, st, a);
    }
    JSDR_LAUNCH_CHECK();
    return JSDR_OK;
}
