// fft_psd.hip -- batched complex forward FFT + PSD + argmax: the fft.java waterfall path.
//
// Replaces fft.receive (fft.java:190-228) and the JTransforms FloatFFT_1D.complexForward call inside
// it (fft.java:194-195), with the int16 -> float rule of JavaAudio.java:276-293 fused into the load.
//
// MI355X mapping (DESIGN.md "fft kernel"): one frame = T threads of a 256/512-thread workgroup; the
// frame is transformed by a 2..4 pass Stockham autosort FFT with radix-16/8/4 butterflies held in
// registers; passes exchange through one padded LDS image per frame (pad 1 float2 per 16 so that the
// stride-16 stores of pass 1 and the stride-N/R loads of the next pass are bank-conflict free).
// Global loads are one dword (I,Q int16 pair) per lane, 256 contiguous bytes per wave instruction;
// PSD stores likewise.  HBM traffic = 4 B in + 4 B out per sample: the kernel is HBM bound.
// Float arithmetic may contract to FMA here: parity for this path is the 1e-5 (peak-normalised)
// tolerance of BASELINE.json, JTransforms' own rounding being unknowable (source absent).
#include "fft_common.h"
#include <math.h>
#include <stdlib.h>
#include <vector>

namespace jsdr {

// ------------------------------------------------------------------ kernel
// Twiddle tables of pass with radix R and P = product of earlier radices (P > 1):
//   P*R <= 512 : "direct"  D[(r-1)*P + k] = (w, w') with w = exp(-2 pi i k r/(P R)), w' = (-w.y, w.x), 1 <= r < R, k < P
//                          (k fastest: conflict-free 16-byte reads; a*w = a.x*w + a.y*w' is two packed instructions)
//   else       : "base"    B[k]       = exp(-2 pi i k  /(P R)), k < P; powers r=2.. by repeated products
constexpr bool tw_direct(int P, int R) { return P * R <= 512; }
constexpr int tw_size(int P, int R) { return P <= 1 ? 0 : (tw_direct(P, R) ? 2 * P * (R - 1) : P); }  // in float2 units
constexpr int tw_total(int R0, int R1, int R2, int R3)
{
    return tw_size(R0, R1) + (R2 > 1 ? tw_size(R0 * R1, R2) : 0) + (R3 > 1 ? tw_size(R0 * R1 * R2, R3) : 0);
}

constexpr int ilog2_c(int n) { return n <= 1 ? 0 : 1 + ilog2_c(n / 2); }

template <int R, int P>
__device__ __forceinline__ void apply_twiddles(float2 *v, int k, const float2 *tab)
{
    if constexpr (P > 1) {
        if constexpr (tw_direct(P, R)) {
#pragma unroll
            for (int r = 1; r < R; r++) {
                const float4 t = reinterpret_cast<const float4 *>(tab)[(r - 1) * P + k];
                v[r] = cmul2(v[r], make_float2(t.x, t.y), make_float2(t.z, t.w));
            }
        } else {
            float2 w1 = tab[k];
            float2 w[R], wq[R];  // wq = (-w.y, w.x): a*w = a.x*w + a.y*wq, two packed instructions
            w[1] = w1;
            wq[1] = cquad(w1);
#pragma unroll
            for (int r = 2; r < R; r++) {
                w[r] = (r & 1) ? cmul2(w[r - 1], w1, wq[1]) : cmul2(w[r / 2], w[r / 2], wq[r / 2]);
                wq[r] = cquad(w[r]);
            }
#pragma unroll
            for (int r = 1; r < R; r++) v[r] = cmul2(v[r], w[r], wq[r]);
        }
    }
}

template <int R, int... Rs>
__device__ __forceinline__ void store_lds(float2 *dst, const float2 *v, int j0, int p, std::integer_sequence<int, Rs...>)
{
    ((dst[lds_pad(j0 + Rs * p)] = v[cx_bitrev(Rs, R)]), ...);
}

// A frame's first-pass inputs as they come from memory: one (I,Q) int16 pair or one float pair per point.  They are
// fetched ONE FRAME AHEAD, just before the previous frame's last pass: that pass holds one small butterfly at a time
// (registers to spare), and a load issued before the PSD stores does not queue behind them -- loads and stores share
// the in-order vmcnt counter, so a frame's loads issued after the previous frame's stores waited for those stores'
// acknowledgements first (measured on 2048 streams: 4.02 ms, without the loads 3.41, without the stores 3.36, without
// both 2.76: memory time simply added to the arithmetic).
template <int IN>
struct RawPoint {
    using type = int;
};
template <>
struct RawPoint<IN_F32> {
    using type = float2;
};
template <int N, int T, int IN, int R>
__device__ __forceinline__ void fft_fetch(const FftArgs &a, long long frame, int tid, typename RawPoint<IN>::type (&w)[R])
{
    constexpr int NB = N / R;
    static_assert(NB == T, "one first-pass butterfly per thread");
    const typename RawPoint<IN>::type *src = reinterpret_cast<const typename RawPoint<IN>::type *>(a.in) +
#ifdef JSDR_X_FFT_SMALLSET  // timing probe (round 5): every frame reads one of 8192 frames -- 64 MB, served by the memory-side cache
        (frame & 8191) * N;
#else
        frame * N;
#endif
#pragma unroll
    for (int r = 0; r < R; r++) {
#ifdef JSDR_X_FFT_NOLOAD  // timing experiment: no input loads (wrong data)
        if constexpr (IN == IN_I16) w[r] = (tid + r * NB) * 2654435761u + (int)(size_t)src;
        else w[r] = make_float2((float)(tid + r), (float)(size_t)src);
#else
        w[r] = src[tid + r * NB];
#endif
    }
}

// One Stockham pass of one frame by T threads.  Each thread owns ITERS = (N/R)/T butterflies, loads
// them all (global for the first pass, LDS otherwise), transforms in registers, and only after a
// barrier (every load of the in-place image has landed) writes its outputs.
template <int N, int T, int IN, int OUT, int R, int P, bool FIRST, bool LAST, class RAW = int>
__device__ __forceinline__ void fft_pass(const FftArgs &a, long long frame, bool active, int tid, float2 *buf,
                                         const float2 *tab, Best &best, const RAW *raw = nullptr)
{
    constexpr int NB = N / R;
    static_assert(NB % T == 0, "butterflies per pass must be a multiple of the threads per frame");
    constexpr int ITERS = NB / T;
    // the last pass writes to global memory, not to the LDS image: its butterflies need not all be in flight
    // before the first store, so they are processed one at a time (half the live registers at ITERS = 2)
    constexpr int GROUP = LAST ? 1 : ITERS;
#pragma unroll
    for (int g0 = 0; g0 < ITERS; g0 += GROUP) {
        float2 v[GROUP][R];
#pragma unroll
        for (int gi = 0; gi < GROUP; gi++) {
            const int b = (g0 + gi) * T + tid;
            if constexpr (FIRST) {
                if (active) {
                    if constexpr (IN == IN_I16) {
                        const RAW *w = raw;
                        if ((a.ic | a.qc) == 0) {  // uniform: no DC correction (the usual case), two adds per sample less
#pragma unroll
                            for (int r = 0; r < R; r++)
                                v[gi][r] = make_float2(i16_to_float_java((int)(short)(w[r] & 0xffff)),
                                                       i16_to_float_java(w[r] >> 16));
                        } else {
#pragma unroll
                            for (int r = 0; r < R; r++) {
                                int si = java_short_add((int)(short)(w[r] & 0xffff), a.ic);
                                int sq = java_short_add(w[r] >> 16, a.qc);
                                v[gi][r] = make_float2(i16_to_float_java(si), i16_to_float_java(sq));
                            }
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < R; r++) v[gi][r] = raw[r];
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < R; r++) v[gi][r] = make_float2(0.f, 0.f);
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; r++) v[gi][r] = buf[lds_pad(b + r * NB)];
            }
            apply_twiddles<R, P>(v[gi], b & (P - 1), tab);
            dft_reg<R>(v[gi]);
        }
        if constexpr (!FIRST && !LAST) __syncthreads();
#pragma unroll
        for (int gi = 0; gi < GROUP; gi++) {
            const int b = (g0 + gi) * T + tid;
            const int k = b & (P - 1);
            const int j0 = (b - k) * R + k;
            if constexpr (!LAST) {
                store_lds<R>(buf, v[gi], j0, P, std::make_integer_sequence<int, R>{});
            } else if (active) {
                if constexpr (OUT == OUT_SPEC) {
                    float2 *dst = reinterpret_cast<float2 *>(a.out) + frame * N;
#pragma unroll
                    for (int r = 0; r < R; r++) dst[j0 + r * P] = v[gi][cx_bitrev(r, R)];
                } else {
                    // fft.java:205-207: 10*log10((re^2+im^2) * (2/N)^2) = 10*log10(2) * (log2(re^2+im^2) + 2 - 2*log2(N)):
                    // N is a power of two, so the scale is an exact integer offset of the base-2 logarithm
                    constexpr float L2 = 3.0102999566398120f;
                    constexpr float OFF = 3.0102999566398120f * (2.0f - 2.0f * (float)ilog2_c(N));
                    float *dst = a.out + frame * (N + 2);
                    float db[R];
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        float2 x = v[gi][cx_bitrev(r, R)];
                        db[r] = L2 * __log2f(x.x * x.x + x.y * x.y) + OFF;
#ifndef JSDR_X_FFT_NOSTORE  // timing experiment: no PSD stores (the argmax keeps the arithmetic alive)
                        dst[j0 + r * P] = db[r];  // (nontemporal stores: 1.99 -> 2.39 ms, measured r02; 8-byte stores through a DPP
                                                  //  exchange between neighbouring lanes: 15.5 -> 16.1 ms at 8192 streams, r03)
#endif
                    }
                    // first strict maximum of this thread's bins (fft.java:208-211): the group's maximum
                    // (fmaxf ignores NaN, like the reference's '>'), then the lowest bin that holds it
                    // (bins ascend with r), folded into the running best once per group
                    float gm = db[0];
#pragma unroll
                    for (int r = 1; r < R; r++) gm = fmaxf(gm, db[r]);
                    int gk = 0x7fffffff;
#pragma unroll
                    for (int r = R - 1; r >= 0; r--) gk = (db[r] == gm) ? j0 + r * P : gk;
                    const bool take = (gm > best.v) | ((gm == best.v) & (gk < best.k));  // branch-free
                    best.v = take ? gm : best.v;
                    best.k = take ? gk : best.k;
                }
            }
        }
    }
    if constexpr (!LAST) __syncthreads();
}

// One frame, executed by T threads (tid in [0,T)).  `raw` holds the frame's first-pass inputs (fft_fetch); before the
// last pass the NEXT frame's inputs are fetched into it.
template <int N, int T, int IN, int OUT, int R0, int R1, int R2, int R3>
__device__ __forceinline__ void fft_frame(const FftArgs &a, long long frame, long long next, int tid, float2 *buf,
                                          const float2 *tw_lds, float *red_val, int *red_idx, int frame_in_block,
                                          typename RawPoint<IN>::type (&raw)[R0])
{
    using RAW = typename RawPoint<IN>::type;
    constexpr bool active = true;
    constexpr int NPASS = (R1 == 1) ? 1 : (R2 == 1) ? 2 : (R3 == 1) ? 3 : 4;
    static_assert(NPASS >= 2, "the first pass must not be the last");
    static_assert(R0 * R1 * R2 * R3 == N, "radix plan must multiply to N");
    Best best;
    best.v = -3.402823466e+38f;
    best.k = 0x7fffffff;
    constexpr int O1 = 0, O2 = tw_size(R0, R1), O3 = O2 + tw_size(R0 * R1, R2);
    fft_pass<N, T, IN, OUT, R0, 1, true, false, RAW>(a, frame, active, tid, buf, tw_lds, best, raw);
    // (unconditional: a workgroup's last frame fetches itself again -- with a branch around the loads the compiler cannot
    //  count what is in flight at the loop head and waits for everything, the previous frame's stores included)
    auto prefetch = [&] { fft_fetch<N, T, IN, R0>(a, next, tid, raw); };
    if constexpr (NPASS == 2) prefetch();
    fft_pass<N, T, IN, OUT, R1, R0, false, NPASS == 2>(a, frame, active, tid, buf, tw_lds + O1, best);
    if constexpr (NPASS >= 3) {
        if constexpr (NPASS == 3) prefetch();
        fft_pass<N, T, IN, OUT, R2, R0 * R1, false, NPASS == 3>(a, frame, active, tid, buf, tw_lds + O2, best);
    }
    if constexpr (NPASS >= 4) {
        prefetch();
        fft_pass<N, T, IN, OUT, R3, R0 * R1 * R2, false, NPASS == 4>(a, frame, active, tid, buf, tw_lds + O3, best);
    }

    if constexpr (OUT == OUT_PSD) {
        // first strict maximum (fft.java:208-211): highest value, lowest bin on ties; none if all -inf
        float bestv = best.v;
        int bestk = best.k;
        constexpr int W = (T < 64) ? T : 64;
#pragma unroll
        for (int off = W / 2; off >= 1; off >>= 1) {
            float ov = __shfl_xor(bestv, off, W);
            int ok = __shfl_xor(bestk, off, W);
            const bool take = (ov > bestv) | ((ov == bestv) & (ok < bestk));  // branch-free
            bestv = take ? ov : bestv;
            bestk = take ? ok : bestk;
        }
        if constexpr (T > 64) {
            constexpr int NW = T / 64;
            if ((tid & 63) == 0) {
                red_val[frame_in_block * NW + (tid >> 6)] = bestv;
                red_idx[frame_in_block * NW + (tid >> 6)] = bestk;
            }
            __syncthreads();
            if (tid == 0) {
                for (int w = 1; w < NW; w++) {
                    float ov = red_val[frame_in_block * NW + w];
                    int ok = red_idx[frame_in_block * NW + w];
                    if (ov > bestv || (ov == bestv && ok < bestk)) {
                        bestv = ov;
                        bestk = ok;
                    }
                }
            }
        }
        if (tid == 0 && active) {
            // fft.java:201-224: m starts at -Float.MAX_VALUE, p at -1; Hz in wrapping int arithmetic
            int p = (bestv > -3.402823466e+38f) ? 2 * bestk : -1;
            float m = (p >= 0) ? bestv : -3.402823466e+38f;
            const int datlen = 2 * N;
            if (p >= datlen / 2) p -= datlen;
            int hz = (int)((unsigned)p * (unsigned)a.rate) / datlen;
            float *dst = a.out + frame * (N + 2);
            dst[N] = (float)hz;
            dst[N + 1] = m;
        }
    }
}

// TWG: the twiddle tables are read from global memory (L1 / L2) instead of a per-workgroup copy in LDS
template <int N, int T, int FPB, int IN, int OUT, int R0, int R1, int R2, int R3, bool TWG = false>
__global__ __launch_bounds__(T *FPB, (N == 2048 ? 4 : 1)) void k_fft(FftArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int FE = lds_frame_elems(N);
    constexpr int TWN = TWG ? 0 : tw_total(R0, R1, R2, R3);
    float2 *tw_own = reinterpret_cast<float2 *>(smem);                    // [TWN]
    float2 *bufs = tw_own + TWN;                                          // [FPB][FE]
    float *red_val = reinterpret_cast<float *>(bufs + (size_t)FPB * FE);  // [FPB*NW]
    int *red_idx = reinterpret_cast<int *>(red_val + FPB * ((T + 63) / 64));
    const float2 *tw_lds = TWG ? a.tw : tw_own;

    const int tid_b = threadIdx.x;
    if constexpr (!TWG) {
        for (int i = tid_b; i < TWN; i += T * FPB) tw_own[i] = a.tw[i];
        __syncthreads();
    }

    const int fib = tid_b / T;
    const int tid = tid_b - fib * T;
    float2 *buf = bufs + (size_t)fib * FE;
    const long long ngroups = (a.nframes + FPB - 1) / FPB;
    // a surplus part-workgroup (frame count not a multiple of FPB) redoes the last frame: identical values to
    // identical addresses, and no predicate on any load, store or barrier
    auto frame_of = [&](long long g) {
        long long f = g * FPB + fib;
        return f < a.nframes ? f : a.nframes - 1;
    };
    typename RawPoint<IN>::type raw[R0];
    if ((long long)blockIdx.x < ngroups) fft_fetch<N, T, IN, R0>(a, frame_of(blockIdx.x), tid, raw);
    for (long long g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const long long gn = g + gridDim.x;
        fft_frame<N, T, IN, OUT, R0, R1, R2, R3>(a, frame_of(g), frame_of(gn < ngroups ? gn : g), tid, buf, tw_lds, red_val,
                                                 red_idx, fib, raw);
        __syncthreads();
    }
}

template <int N, int FPB, int T, int R0, int R1, int R2, int R3, bool TWG = false>
constexpr size_t fft_lds_bytes()
{
    return sizeof(float2) * ((TWG ? (size_t)0 : (size_t)tw_total(R0, R1, R2, R3)) + (size_t)FPB * lds_frame_elems(N)) +
           (sizeof(float) + sizeof(int)) * FPB * ((T + 63) / 64);
}

struct Launcher {
    void (*launch)(const FftArgs &, int grid, hipStream_t) = nullptr;
    int frames_per_block = 0;
    size_t lds_bytes = 0;
    int block = 0;
    int radix[4] = {1, 1, 1, 1};
};

template <int N, int T, int FPB, int IN, int OUT, int R0, int R1, int R2, int R3, bool TWG = false>
static void launch_impl(const FftArgs &a, int grid, hipStream_t s)
{
    constexpr size_t lds = fft_lds_bytes<N, FPB, T, R0, R1, R2, R3, TWG>();
    hipLaunchKernelGGL((k_fft<N, T, FPB, IN, OUT, R0, R1, R2, R3, TWG>), dim3(grid), dim3(T * FPB), lds, s, a);
}

template <int N, int T, int FPB, int R0, int R1, int R2, int R3, bool TWG = false>
static Launcher make_launcher(int in, int out)
{
    Launcher l;
    l.frames_per_block = FPB;
    l.block = T * FPB;
    l.lds_bytes = fft_lds_bytes<N, FPB, T, R0, R1, R2, R3, TWG>();
    l.radix[0] = R0;
    l.radix[1] = R1;
    l.radix[2] = R2;
    l.radix[3] = R3;
    if (in == IN_I16 && out == OUT_PSD) l.launch = launch_impl<N, T, FPB, IN_I16, OUT_PSD, R0, R1, R2, R3, TWG>;
    if (in == IN_F32 && out == OUT_PSD) l.launch = launch_impl<N, T, FPB, IN_F32, OUT_PSD, R0, R1, R2, R3, TWG>;
    if (in == IN_F32 && out == OUT_SPEC) l.launch = launch_impl<N, T, FPB, IN_F32, OUT_SPEC, R0, R1, R2, R3, TWG>;
    return l;
}

static Launcher pick_launcher(int n, int in, int out)
{
    switch (n) {
        case 64: return make_launcher<64, 8, 32, 8, 8, 1, 1>(in, out);
        case 128: return make_launcher<128, 8, 32, 16, 8, 1, 1>(in, out);
        case 256: return make_launcher<256, 16, 16, 16, 16, 1, 1>(in, out);
        case 512: return make_launcher<512, 64, 4, 8, 8, 8, 1>(in, out);
        case 1024: return make_launcher<1024, 64, 4, 16, 8, 8, 1>(in, out);
        // (round 5, measured and not kept -- profiles/r05_experiments.md: one frame per 128-thread workgroup with the tables in LDS
        //  17.0-17.1 ms, one frame with the tables read from global memory (make_launcher<..., true>) 16.7-17.3, two frames with
        //  global tables 17.05, against 15.8 for this form)
        case 2048: return make_launcher<2048, 128, 2, 16, 16, 8, 1>(in, out);
        case 4096: return make_launcher<4096, 256, 1, 16, 16, 16, 1>(in, out);
        case 8192: return make_launcher<8192, 512, 1, 16, 16, 8, 4>(in, out);
        default: return Launcher();
    }
}

}  // namespace jsdr

using namespace jsdr;

struct jsdr_fft {
    int n = 0;
    int rate = 0;
    DevBuf<float2> tw;
    DevBuf<unsigned char> in_stage;  // one frame, for the host-buffer receive() forms
    DevBuf<float> out_stage;
    PinnedStage pin;                 // [frame in (8 n bytes) | psd out (4 (n + 2) bytes)] for the receive() forms
    bool pin_lazy_done = false;      // the stage is allocated at the first receive()
    int num_cu = 256;
    bool mixed = false;  // non power-of-two frame: fft_mixed.hip
    MixedPlan mplan;
    bool rt = false;      // any other 2^a 3^b 5^c 7^d frame up to 9800 samples: fft_rt.hip
    bool direct = false;  // any other frame size: fft_any.hip
    long long last_items = 0, last_grid = 0;  // jsdr_fft_last_launch
    int share_wgs_per_cu = 0;  // jsdr_fft_set_cu_share: workgroups per CU the batch kernel is held to (0: all it can use)
};

static int fft_run(jsdr_fft *h, const void *in_dev, int in_kind, int out_kind, long long nframes, int ic, int qc,
                   float *out_dev, hipStream_t s)
{
    JSDR_REQUIRE(h, "fft: null handle");
    JSDR_REQUIRE(in_dev && out_dev, "fft: null buffer");
    JSDR_REQUIRE(nframes >= 0, "fft: negative frame count");
    if (nframes == 0) return JSDR_OK;
    FftArgs a;
    if (h->rt) {
        a.in = in_dev;
        a.out = out_dev;
        a.tw = h->tw.p;
        a.nframes = nframes;
        a.rate = h->rate;
        a.ic = ic;
        a.qc = qc;
        return rt_launch(a, h->n, in_kind, out_kind, h->num_cu, s);
    }
    if (h->direct) {
        a.in = in_dev;
        a.out = out_dev;
        a.tw = nullptr;
        a.nframes = nframes;
        a.rate = h->rate;
        a.ic = ic;
        a.qc = qc;
        return dft_any_launch(a, h->n, in_kind, out_kind, h->num_cu, s);
    }
    if (h->mixed) {
        a.in = in_dev;
        a.out = out_dev;
        a.tw = h->tw.p;
        a.nframes = nframes;
        a.rate = h->rate;
        a.ic = ic;
        a.qc = qc;
        if (h->mplan.split2)
            return mixed_launch_split2(h->mplan, a, in_kind, out_kind, h->num_cu, s);
        static const int mgrid = [] { const char *e = knob("JSDR_MIXED_GRID"); return e ? atoi(e) : 16; }();  // workgroups per CU: shorter workgroups balance the tail (2: 3.84, 16: 3.56 ms at n = 9600)
        long long cap = (long long)h->num_cu * mgrid;
        return mixed_launch(h->mplan, a, in_kind, out_kind, (int)(nframes < cap ? nframes : cap), s);
    }
    Launcher l = pick_launcher(h->n, in_kind, out_kind);
    JSDR_REQUIRE(l.launch, "fft: no kernel for n=%d in=%d out=%d", h->n, in_kind, out_kind);
    a.in = in_dev;
    a.out = out_dev;
    a.tw = h->tw.p;
    a.nframes = nframes;
    a.rate = h->rate;
    a.ic = ic;
    a.qc = qc;
    long long groups = (nframes + l.frames_per_block - 1) / l.frames_per_block;
    // enough workgroups to fill every CU at the LDS-limited occupancy, grid-stride over the rest
    long long per_cu = (long long)(160 * 1024 / l.lds_bytes);
    if (per_cu < 1) per_cu = 1;
    if (per_cu * l.block > 2048) per_cu = 2048 / l.block;
    static const int mult = [] {
        const char *e = knob("JSDR_FFT_GRID_MULT");  // tuning knob: workgroups per resident slot
        const int v = e ? atoi(e) : 0;
        return v > 0 ? v : 4;
    }();
    long long cap = (long long)h->num_cu * per_cu * mult;
    static const int abs_grid = [] {
        const char *e = knob("JSDR_FFT_GRID_ABS");  // tuning knob: total workgroups (co-residency experiments: 2 per CU = 512)
        return e ? atoi(e) : 0;
    }();
    if (abs_grid > 0) cap = abs_grid;
    // a caller that runs the demodulator BESIDE this kernel (jsdr_fft_set_cu_share): exactly that many persistent workgroups
    // per CU, so that both kernels' workgroups are resident from the start whichever is launched first
    if (h->share_wgs_per_cu > 0) cap = (long long)h->num_cu * h->share_wgs_per_cu;
    int grid = (int)(groups < cap ? groups : cap);
    h->last_items = groups;
    h->last_grid = grid;
    l.launch(a, grid, s);
    JSDR_LAUNCH_CHECK();
    return JSDR_OK;
}

extern "C" {

int jsdr_fft_create(jsdr_fft **out, int n, int rate)
{
    JSDR_REQUIRE(out, "jsdr_fft_create: null handle pointer");
    *out = nullptr;
    MixedPlan mp;
    const bool pow2 = n >= 64 && n <= 8192 && (n & (n - 1)) == 0;
    const bool mixed = !pow2 && mixed_plan(n, mp);
    const bool rt = !pow2 && !mixed && rt_supported(n);
    const bool direct = !pow2 && !mixed && !rt && dft_any_supported(n);
    JSDR_REQUIRE(pow2 || mixed || rt || direct,
                 "jsdr_fft_create: n=%d unsupported (2 .. 20000 samples)", n);
    JSDR_REQUIRE(rate > 0, "jsdr_fft_create: rate must be positive");
    int dev = 0;
    JSDR_HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    JSDR_HIP_TRY(hipGetDeviceProperties(&prop, dev));
    jsdr_fft *h = new jsdr_fft();
    h->n = n;
    h->rate = rate;
    h->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (direct) {  // no tables: the twiddles are computed where they are used
        h->direct = true;
        if (h->in_stage.alloc((size_t)n * 8) != JSDR_OK || h->out_stage.alloc((size_t)n + 2) != JSDR_OK) {
            jsdr_fft_destroy(h);
            return JSDR_ERR;
        }
        *out = h;
        return JSDR_OK;
    }
    if (rt) {
        h->rt = true;
        std::vector<float2> rtw;
        rt_twiddles(n, rtw);
        if (h->tw.alloc(rtw.size() ? rtw.size() : 1) != JSDR_OK || h->in_stage.alloc((size_t)n * 8) != JSDR_OK ||
            h->out_stage.alloc((size_t)n + 2) != JSDR_OK ||
            (rtw.size() && hipMemcpy(h->tw.p, rtw.data(), sizeof(float2) * rtw.size(), hipMemcpyHostToDevice) != hipSuccess)) {
            set_error("jsdr_fft_create: run-time-plan setup failed");
            jsdr_fft_destroy(h);
            return JSDR_ERR;
        }
        *out = h;
        return JSDR_OK;
    }
    if (mixed) {
        h->mixed = true;
        h->mplan = mp;
        std::vector<float2> mtw((size_t)mp.tw_count);
        mixed_twiddles(mp, mtw.data());
        if (h->tw.alloc(mtw.size()) != JSDR_OK || h->in_stage.alloc((size_t)n * 8) != JSDR_OK ||
            h->out_stage.alloc((size_t)n + 2) != JSDR_OK ||
            hipMemcpy(h->tw.p, mtw.data(), sizeof(float2) * mtw.size(), hipMemcpyHostToDevice) != hipSuccess) {
            set_error("jsdr_fft_create: mixed-radix setup failed");
            jsdr_fft_destroy(h);
            return JSDR_ERR;
        }
        *out = h;
        return JSDR_OK;
    }
    // per-pass twiddle tables in the layout fft_pass expects (tw_direct / tw_size above)
    Launcher l = pick_launcher(n, IN_I16, OUT_PSD);
    std::vector<float2> tw;
    int P = l.radix[0];
    for (int pass = 1; pass < 4 && l.radix[pass] > 1; pass++) {
        const int R = l.radix[pass];
        const double base = -2.0 * 3.14159265358979323846 / ((double)P * (double)R);
        if (tw_direct(P, R)) {
            for (int r = 1; r < R; r++)  // row 0 is all ones and never read (and its 256 bytes decide whether a
                                         // fourth 2048-point workgroup fits a CU's LDS)
                for (int k = 0; k < P; k++) {
                    double ang = base * (double)k * (double)r;
                    const float2 w = make_float2((float)cos(ang), (float)sin(ang));
                    tw.push_back(w);
                    tw.push_back(make_float2(-w.y, w.x));
                }
        } else {
            for (int k = 0; k < P; k++) {
                double ang = base * (double)k;
                tw.push_back(make_float2((float)cos(ang), (float)sin(ang)));
            }
        }
        P *= R;
    }
    if ((int)tw.size() != tw_total(l.radix[0], l.radix[1], l.radix[2], l.radix[3])) {
        set_error("jsdr_fft_create: internal twiddle layout mismatch (%d)", (int)tw.size());
        delete h;
        return JSDR_ERR;
    }
    if (h->tw.alloc(tw.size()) != JSDR_OK || h->in_stage.alloc((size_t)n * 8) != JSDR_OK ||
        h->out_stage.alloc((size_t)n + 2) != JSDR_OK) {
        jsdr_fft_destroy(h);
        return JSDR_ERR;
    }
    if (hipMemcpy(h->tw.p, tw.data(), sizeof(float2) * tw.size(), hipMemcpyHostToDevice) != hipSuccess) {
        set_error("jsdr_fft_create: twiddle upload failed");
        jsdr_fft_destroy(h);
        return JSDR_ERR;
    }
    *out = h;
    return JSDR_OK;
}

int jsdr_fft_destroy(jsdr_fft *h)
{
    if (!h) return JSDR_OK;
    h->tw.release();
    h->in_stage.release();
    h->out_stage.release();
    h->pin.release();
    delete h;
    return JSDR_OK;
}

int jsdr_fft_batch_f32(jsdr_fft *h, const float *iq_dev, int64_t nframes, float *psd_dev, void *stream)
{
    return fft_run(h, iq_dev, IN_F32, OUT_PSD, nframes, 0, 0, psd_dev, as_stream(stream));
}

int jsdr_fft_set_cu_share(jsdr_fft *h, int wgs_per_cu)
{
    JSDR_REQUIRE(h && wgs_per_cu >= 0 && wgs_per_cu <= 16, "jsdr_fft_set_cu_share: bad argument");
    h->share_wgs_per_cu = wgs_per_cu;
    return JSDR_OK;
}

const char *jsdr_fft_kernel(jsdr_fft *h)
{
    if (!h) return "";
    return h->rt ? "k_fft_rt" : h->direct ? "k_dft_any" : h->mixed ? (h->mplan.split2 ? "k_fft_mixed_dual" : "k_fft_mixed") : "k_fft";
}

int jsdr_fft_last_launch(jsdr_fft *h, int64_t *work_items, int64_t *workgroups)
{
    JSDR_REQUIRE(h && work_items && workgroups, "jsdr_fft_last_launch: null argument");
    *work_items = h->last_items;
    *workgroups = h->last_grid;
    return JSDR_OK;
}

int jsdr_fft_batch_i16(jsdr_fft *h, const int16_t *raw_dev, int64_t nframes, int ic, int qc, float *psd_dev,
                       void *stream)
{
    return fft_run(h, raw_dev, IN_I16, OUT_PSD, nframes, ic, qc, psd_dev, as_stream(stream));
}

int jsdr_fft_spectrum_f32(jsdr_fft *h, const float *iq_dev, int64_t nframes, float *spec_dev, void *stream)
{
    return fft_run(h, iq_dev, IN_F32, OUT_SPEC, nframes, 0, 0, spec_dev, as_stream(stream));
}

// one frame from / to host buffers: through the handle's pinned stage when it has one (PinnedStage, common.h)
static int fft_receive(jsdr_fft *h, const void *in_host, size_t in_bytes, int in_kind, int ic, int qc, float *psd_host)
{
    const size_t out_bytes = sizeof(float) * ((size_t)h->n + 2), in_room = (size_t)h->n * 8;
    if (h->pin.p && h->pin.bytes >= in_room + out_bytes) {
        memcpy(h->pin.p, in_host, in_bytes);
        SyncOnExit guard;  // (an error exit below must not leave the device reading or writing the stage)
        JSDR_HIP_TRY(hipMemcpyAsync(h->in_stage.p, h->pin.p, in_bytes, hipMemcpyHostToDevice, 0));
        if (fft_run(h, h->in_stage.p, in_kind, OUT_PSD, 1, ic, qc, h->out_stage.p, 0) != JSDR_OK) return JSDR_ERR;
        JSDR_HIP_TRY(hipMemcpyAsync(h->pin.p + in_room, h->out_stage.p, out_bytes, hipMemcpyDeviceToHost, 0));
        JSDR_HIP_TRY(hipStreamSynchronize(0));
        guard.armed = false;
        memcpy(psd_host, h->pin.p + in_room, out_bytes);
        return JSDR_OK;
    }
    JSDR_HIP_TRY(hipMemcpy(h->in_stage.p, in_host, in_bytes, hipMemcpyHostToDevice));
    if (fft_run(h, h->in_stage.p, in_kind, OUT_PSD, 1, ic, qc, h->out_stage.p, 0) != JSDR_OK) return JSDR_ERR;
    JSDR_HIP_TRY(hipMemcpy(psd_host, h->out_stage.p, out_bytes, hipMemcpyDeviceToHost));
    return JSDR_OK;
}

int jsdr_fft_receive_f32(jsdr_fft *h, const float *iq_host, float *psd_host)
{
    JSDR_REQUIRE(h && iq_host && psd_host, "jsdr_fft_receive_f32: null argument");
    if (!h->pin_lazy_done && !h->pin.p) {  // the first receive() of a handle: batch-only handles never pin anything
        h->pin.alloc((size_t)h->n * 8 + sizeof(float) * ((size_t)h->n + 2));
        h->pin_lazy_done = true;
    }
    return fft_receive(h, iq_host, sizeof(float) * 2 * (size_t)h->n, IN_F32, 0, 0, psd_host);
}

int jsdr_fft_receive_i16(jsdr_fft *h, const int16_t *raw_host, int ic, int qc, float *psd_host)
{
    JSDR_REQUIRE(h && raw_host && psd_host, "jsdr_fft_receive_i16: null argument");
    if (!h->pin_lazy_done && !h->pin.p) {
        h->pin.alloc((size_t)h->n * 8 + sizeof(float) * ((size_t)h->n + 2));
        h->pin_lazy_done = true;
    }
    return fft_receive(h, raw_host, sizeof(int16_t) * 2 * (size_t)h->n, IN_I16, ic, qc, psd_host);
}

}  // extern "C"
