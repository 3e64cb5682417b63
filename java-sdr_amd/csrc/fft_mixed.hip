// fft_mixed.hip -- fft.receive for NON power-of-two frames: n = 9600 (java-sdr's default 96 kHz buffer,
// blen = rate*size/10, JavaAudio.java:58-59) and n = 4800 (48 kHz).  JTransforms handles such n with a mixed
// radix 2/3/4/5 decomposition (fft.java:194-195); here: Stockham autosort passes with radices 16/8/4 (the
// in-register radix-2^k kernels of fft_common.h) and hand-written radix-5 / radix-3 butterflies.
//
// One frame per workgroup (n=9600: 512 threads, 81.6 KB padded LDS image + 32 KB of per-pass twiddles); a
// thread owns ceil((n/R)/T) butterflies per pass.  Same PSD / first-maximum / Hz rules as k_fft.
// Parity: 1e-5 of the frame peak against the exact DFT (JTransforms' rounding is unknowable).
#include "fft_common.h"
#include <math.h>
#include <stdlib.h>
#include <vector>

namespace jsdr {

constexpr bool is_pow2(int r) { return (r & (r - 1)) == 0; }
// position of output r inside the register array after dft_any<R>
template <int R>
constexpr int out_slot(int r)
{
    return is_pow2(R) ? cx_bitrev(r, R) : r;
}

__device__ __forceinline__ void dft3(float2 *x)
{
    constexpr float S = (float)0.86602540378443864676;  // sin(2 pi/3)
    const float2 t1 = cadd(x[1], x[2]);
    const float2 t2 = make_float2(x[0].x - 0.5f * t1.x, x[0].y - 0.5f * t1.y);
    const float2 d = csub(x[1], x[2]);
    const float2 t3 = make_float2(d.x * S, d.y * S);
    x[0] = cadd(x[0], t1);
    x[1] = make_float2(t2.x + t3.y, t2.y - t3.x);  // t2 - i*t3
    x[2] = make_float2(t2.x - t3.y, t2.y + t3.x);  // t2 + i*t3
}

__device__ __forceinline__ void dft5(float2 *x)
{
    constexpr float C1 = (float)0.30901699437494742410;   // cos(2 pi/5)
    constexpr float C2 = (float)-0.80901699437494742410;  // cos(4 pi/5)
    constexpr float S1 = (float)0.95105651629515357212;   // sin(2 pi/5)
    constexpr float S2 = (float)0.58778525229247312917;   // sin(4 pi/5)
    const float2 a1 = cadd(x[1], x[4]), a2 = cadd(x[2], x[3]);
    const float2 b1 = csub(x[1], x[4]), b2 = csub(x[2], x[3]);
    const float2 x0 = x[0];
    const float2 m1 = make_float2(x0.x + C1 * a1.x + C2 * a2.x, x0.y + C1 * a1.y + C2 * a2.y);
    const float2 m2 = make_float2(x0.x + C2 * a1.x + C1 * a2.x, x0.y + C2 * a1.y + C1 * a2.y);
    const float2 n1 = make_float2(S1 * b1.x + S2 * b2.x, S1 * b1.y + S2 * b2.y);
    const float2 n2 = make_float2(S2 * b1.x - S1 * b2.x, S2 * b1.y - S1 * b2.y);
    x[0] = make_float2(x0.x + a1.x + a2.x, x0.y + a1.y + a2.y);
    x[1] = make_float2(m1.x + n1.y, m1.y - n1.x);  // m1 - i*n1
    x[4] = make_float2(m1.x - n1.y, m1.y + n1.x);  // m1 + i*n1
    x[2] = make_float2(m2.x + n2.y, m2.y - n2.x);
    x[3] = make_float2(m2.x - n2.y, m2.y + n2.x);
}

template <int R>
__device__ __forceinline__ void dft_any(float2 *x)
{
    if constexpr (R == 3) dft3(x);
    else if constexpr (R == 5) dft5(x);
    else dft_reg<R>(x);
}

// LDS image: n = 9600 must leave room for a SECOND workgroup on the CU (one workgroup per CU marches through its
// load / compute / barrier / store phases alone, and nothing fills the gaps): one pad slot per 32 elements instead of
// per 16 (the first pass's stride-16 stores become 2-way conflicts instead of none) and only the narrow passes'
// twiddle tables in LDS -- the wide ones (stride >= 640: 30 KB) are read through L1/L2, one coalesced read per
// butterfly -- make it 81.2 KB per workgroup.
template <int N>
__device__ __forceinline__ int mpad(int idx)
{
    return (N == 9600 || N == 4800) ? idx + (idx >> 5) : idx + (idx >> 4);
}
constexpr int mixed_frame_elems(int n) { return (n == 9600 || n == 4800) ? n + (n >> 5) + 1 : n + (n >> 4) + 1; }
// passes whose tables are copied to LDS (pass 1 has none); n = 4800: 59 -> 40 KB, four workgroups per CU instead of two
constexpr int mixed_lds_passes(int n) { return n == 4800 ? 2 : 3; }

constexpr bool mtw_direct(int P, int R) { return P * R <= 512; }
constexpr int mtw_size(int P, int R) { return (P <= 1 || R <= 1) ? 0 : (mtw_direct(P, R) ? P * R : P); }

template <int R, int P>
__device__ __forceinline__ void mixed_twiddles_apply(float2 *v, int k, const float2 *tab)
{
    if constexpr (P > 1) {
        if constexpr (mtw_direct(P, R)) {
#pragma unroll
            for (int r = 1; r < R; r++) v[r] = cmul(v[r], tab[r * P + k]);
        } else {
            const float2 w1 = tab[k];
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

template <int N, int T, int IN, int OUT, int R, int P, bool FIRST, bool LAST>
__device__ __forceinline__ void mixed_pass(const FftArgs &a, long long frame, int tid, float2 *buf, const float2 *tab,
                                           Best &best)
{
    constexpr int NB = N / R;
    constexpr int ITERS = (NB + T - 1) / T;
    float2 v[ITERS][R];
    if constexpr (FIRST) {
        // all of a thread's samples are requested before the first one is converted: loads under a per-butterfly
        // guard were issued one at a time, each waiting for the one before (R * ITERS memory latencies per frame)
        // split: this transform is the even / odd half of input frame (frame >> 1), 2N samples long
        const long long fbase = a.split ? (frame >> 1) * (2LL * N) + (frame & 1) : frame * (long long)N;
        const int st = a.split ? 2 : 1;
        if constexpr (IN == IN_I16) {
            const int *src = reinterpret_cast<const int *>(a.in) + fbase;
            int w[ITERS][R];
#pragma unroll
            for (int it = 0; it < ITERS; it++) {
                const int b = (it * T + tid) < NB ? (it * T + tid) : NB - 1;  // clamped: the surplus butterfly is dropped
#pragma unroll
                for (int r = 0; r < R; r++) w[it][r] = src[(b + r * NB) * st];
            }
#pragma unroll
            for (int it = 0; it < ITERS; it++)
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int si = java_short_add((int)(short)(w[it][r] & 0xffff), a.ic);
                    const int sq = java_short_add(w[it][r] >> 16, a.qc);
                    v[it][r] = make_float2(i16_to_float_java(si), i16_to_float_java(sq));
                }
        } else {
            const float2 *src = reinterpret_cast<const float2 *>(a.in) + fbase;
#pragma unroll
            for (int it = 0; it < ITERS; it++) {
                const int b = (it * T + tid) < NB ? (it * T + tid) : NB - 1;
#pragma unroll
                for (int r = 0; r < R; r++) v[it][r] = src[(b + r * NB) * st];
            }
        }
    }
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int b = it * T + tid;
        if (b < NB) {
            if constexpr (!FIRST) {
#pragma unroll
                for (int r = 0; r < R; r++) v[it][r] = buf[mpad<N>(b + r * NB)];
            }
            mixed_twiddles_apply<R, P>(v[it], b % P, tab);
            dft_any<R>(v[it]);
        }
    }
    if constexpr (!FIRST && !LAST) __syncthreads();
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int b = it * T + tid;
        if (b < NB) {
            const int k = b % P;
            const int j0 = (b - k) * R + k;
            if constexpr (!LAST) {
#pragma unroll
                for (int r = 0; r < R; r++) buf[mpad<N>(j0 + r * P)] = v[it][out_slot<R>(r)];
            } else if constexpr (OUT == OUT_SPEC) {
                float2 *dst = reinterpret_cast<float2 *>(a.out) + frame * N;
#pragma unroll
                for (int r = 0; r < R; r++) dst[j0 + r * P] = v[it][out_slot<R>(r)];
            } else {
                const float cf = (2.0f / (float)N) * (2.0f / (float)N);
                float *dst = a.out + frame * (N + 2);
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const float2 x = v[it][out_slot<R>(r)];
                    const float pw = (x.x * x.x + x.y * x.y) * cf;
                    const float db = 3.0102999566398120f * __log2f(pw);  // fft.java:207
                    const int bin = j0 + r * P;
                    dst[bin] = db;
                    if (db > best.v || (db == best.v && bin < best.k)) {
                        best.v = db;
                        best.k = bin;
                    }
                }
            }
        }
    }
    if constexpr (!LAST) __syncthreads();
}

template <int N, int T, int IN, int OUT, int R0, int R1, int R2, int R3, int R4>
__global__ __launch_bounds__(T, (N == 9600 ? 2 : 4)) void k_fft_mixed(FftArgs a)
{
    static_assert(R0 * R1 * R2 * R3 * R4 == N, "radix plan must multiply to N");
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int O2 = 0, O3 = O2 + mtw_size(R0, R1), O4 = O3 + mtw_size(R0 * R1, R2),
                  O5 = O4 + mtw_size(R0 * R1 * R2, R3), TWN = O5 + mtw_size(R0 * R1 * R2 * R3, R4);
    constexpr int TWL = mixed_lds_passes(N) >= 5 ? TWN : (mixed_lds_passes(N) == 3 ? O4 : O3);  // table entries kept in LDS
    float2 *tw = reinterpret_cast<float2 *>(smem);
    float2 *buf = tw + TWL;
    constexpr int NW = (T + 63) / 64;
    float *red_val = reinterpret_cast<float *>(buf + mixed_frame_elems(N));
    int *red_idx = reinterpret_cast<int *>(red_val + NW);
    const int tid = threadIdx.x;
    for (int i = tid; i < TWL; i += T) tw[i] = a.tw[i];
    const float2 *tw3 = (TWL > O3) ? tw + O3 : a.tw + O3;
    const float2 *tw4 = (TWL == TWN) ? tw + O4 : a.tw + O4;
    const float2 *tw5 = (TWL == TWN) ? tw + O5 : a.tw + O5;
    __syncthreads();
    for (long long frame = blockIdx.x; frame < a.nframes; frame += gridDim.x) {
        Best best;
        best.v = -3.402823466e+38f;
        best.k = 0x7fffffff;
        // opaque per frame, so that nothing derived from the thread index is loop invariant -- LLVM otherwise
        // hoists the LDS addresses of all five passes out of the frame loop (158 VGPRs instead of 58) and a CU holds one
        // workgroup less (4.8 -> 4.05 ms per 2^30 samples).  n = 9600: 256 VGPRs + 60 B scratch -> 94, which the second
        // workgroup per CU needs (alone on a CU the hoisted addresses were free and saved arithmetic: 4.0 vs 4.7 ms;
        // 1024-thread workgroups: 4.55 ms opaque, 11 ms with the hoisted addresses spilled under the 128-VGPR cap).
        int tf = tid;
        if constexpr (N <= 9600) asm volatile("" : "+v"(tf));
        mixed_pass<N, T, IN, OUT, R0, 1, true, false>(a, frame, tf, buf, tw, best);
        mixed_pass<N, T, IN, OUT, R1, R0, false, false>(a, frame, tf, buf, tw + O2, best);
        mixed_pass<N, T, IN, OUT, R2, R0 * R1, false, false>(a, frame, tf, buf, tw3, best);
        mixed_pass<N, T, IN, OUT, R3, R0 * R1 * R2, false, false>(a, frame, tf, buf, tw4, best);
        mixed_pass<N, T, IN, OUT, R4, R0 * R1 * R2 * R3, false, true>(a, frame, tf, buf, tw5, best);
        if constexpr (OUT == OUT_PSD) {
            float bestv = best.v;
            int bestk = best.k;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                float ov = __shfl_xor(bestv, off, 64);
                int ok = __shfl_xor(bestk, off, 64);
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
                for (int w = 1; w < NW; w++) {
                    float ov = red_val[w];
                    int ok = red_idx[w];
                    if (ov > bestv || (ov == bestv && ok < bestk)) {
                        bestv = ov;
                        bestk = ok;
                    }
                }
                // fft.java:201-224
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
        __syncthreads();
    }
}

// n = 19200 (192 kHz): the frame does not fit one workgroup's LDS as one transform, but its two halves do --
// X[k] = E[k] + W^k O[k], X[k+n/2] = E[k] - W^k O[k] with E, O the 9600-point transforms of the even / odd samples.
// One workgroup runs both half transforms back to back, each into its own LDS image (2 x 79 KB), and combines them
// from LDS with the PSD epilogue: 4 B in, 4 B out per sample, where the former two-kernel form moved the half
// spectra through HBM (three times the traffic).
template <int T, int IN, int OUT>
__global__ __launch_bounds__(T) void k_fft_mixed_dual(FftArgs a, const float2 *__restrict__ wcomb)
{
    constexpr int N = 9600, R0 = 16, R1 = 8, R2 = 5, R3 = 5, R4 = 3, N2 = 2 * N;
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int O2 = 0, O3 = O2 + mtw_size(R0, R1), O4 = O3 + mtw_size(R0 * R1, R2),
                  O5 = O4 + mtw_size(R0 * R1 * R2, R3);
    constexpr int TWL = O4;
    float2 *tw = reinterpret_cast<float2 *>(smem);
    float2 *bufE = tw + TWL;
    float2 *bufO = bufE + mixed_frame_elems(N);
    constexpr int NW = (T + 63) / 64;
    float *red_val = reinterpret_cast<float *>(bufO + mixed_frame_elems(N));
    int *red_idx = reinterpret_cast<int *>(red_val + NW);
    const int tid = threadIdx.x;
    for (int i = tid; i < TWL; i += T) tw[i] = a.tw[i];
    const float2 *tw4 = a.tw + O4, *tw5 = a.tw + O5;
    __syncthreads();
    for (long long frame = blockIdx.x; frame < a.nframes; frame += gridDim.x) {
        Best best;
        best.v = -3.402823466e+38f;
        best.k = 0x7fffffff;
        int tf = tid;
        asm volatile("" : "+v"(tf));  // (see k_fft_mixed)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            float2 *buf = h ? bufO : bufE;
            const long long hf = 2 * frame + h;  // a.split addressing: even / odd samples of frame `frame`
            int th = tf;
            asm volatile("" : "+v"(th));  // (and per half: shared between the halves, the addresses live through ten passes)
            mixed_pass<N, T, IN, OUT_SPEC, R0, 1, true, false>(a, hf, th, buf, tw, best);
            mixed_pass<N, T, IN, OUT_SPEC, R1, R0, false, false>(a, hf, th, buf, tw + O2, best);
            mixed_pass<N, T, IN, OUT_SPEC, R2, R0 * R1, false, false>(a, hf, th, buf, tw + O3, best);
            mixed_pass<N, T, IN, OUT_SPEC, R3, R0 * R1 * R2, false, false>(a, hf, th, buf, tw4, best);
            mixed_pass<N, T, IN, OUT_SPEC, R4, R0 * R1 * R2 * R3, false, false>(a, hf, th, buf, tw5, best);  // spectrum stays in LDS
        }
        constexpr int ITERS = (N + T - 1) / T;
        float2 wk[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; it++) {  // the combine twiddles of this thread's bins, in flight together
            const int k = it * T + tf;
            wk[it] = wcomb[k < N ? k : N - 1];
        }
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            const int k = it * T + tf;
            if (k < N) {
                const float2 e = bufE[mpad<N>(k)];
                const float2 t = cmul(bufO[mpad<N>(k)], wk[it]);
                const float2 x0 = cadd(e, t), x1 = csub(e, t);
                if constexpr (OUT == OUT_SPEC) {
                    float2 *dst = reinterpret_cast<float2 *>(a.out) + frame * N2;
                    dst[k] = x0;
                    dst[k + N] = x1;
                } else {
                    const float cf = (2.0f / (float)N2) * (2.0f / (float)N2);
                    float *dst = a.out + frame * (N2 + 2);
                    const float d0 = 3.0102999566398120f * __log2f((x0.x * x0.x + x0.y * x0.y) * cf);  // fft.java:207
                    const float d1 = 3.0102999566398120f * __log2f((x1.x * x1.x + x1.y * x1.y) * cf);
                    dst[k] = d0;
                    dst[k + N] = d1;
                    if (d0 > best.v || (d0 == best.v && k < best.k)) {
                        best.v = d0;
                        best.k = k;
                    }
                    if (d1 > best.v || (d1 == best.v && k + N < best.k)) {
                        best.v = d1;
                        best.k = k + N;
                    }
                }
            }
        }
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
                for (int wv = 1; wv < NW; wv++) {
                    const float ov = red_val[wv];
                    const int ok = red_idx[wv];
                    if (ov > bestv || (ov == bestv && ok < bestk)) {
                        bestv = ov;
                        bestk = ok;
                    }
                }
                // fft.java:201-224
                int p = (bestv > -3.402823466e+38f) ? 2 * bestk : -1;
                const float m = (p >= 0) ? bestv : -3.402823466e+38f;
                const int datlen = 2 * N2;
                if (p >= datlen / 2) p -= datlen;
                const int hz = (int)((unsigned)p * (unsigned)a.rate) / datlen;
                float *dst = a.out + frame * (N2 + 2);
                dst[N2] = (float)hz;
                dst[N2 + 1] = m;
            }
        }
        __syncthreads();  // the images are rewritten by the next frame's first passes
    }
}

template <int N, int T, int R0, int R1, int R2, int R3, int R4>
static int mixed_launch_t(const MixedPlan &p, const FftArgs &a, int in_kind, int out_kind, int grid, hipStream_t st)
{
    auto go = [&](auto kern) -> int {
        JSDR_LDS_ATTR(kern, p.lds_bytes);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(T), p.lds_bytes, st, a);
        JSDR_LAUNCH_CHECK();
        return JSDR_OK;
    };
    if (in_kind == IN_I16 && out_kind == OUT_PSD) return go(k_fft_mixed<N, T, IN_I16, OUT_PSD, R0, R1, R2, R3, R4>);
    if (in_kind == IN_F32 && out_kind == OUT_PSD) return go(k_fft_mixed<N, T, IN_F32, OUT_PSD, R0, R1, R2, R3, R4>);
    if (in_kind == IN_F32 && out_kind == OUT_SPEC) return go(k_fft_mixed<N, T, IN_F32, OUT_SPEC, R0, R1, R2, R3, R4>);
    if (in_kind == IN_I16 && out_kind == OUT_SPEC) return go(k_fft_mixed<N, T, IN_I16, OUT_SPEC, R0, R1, R2, R3, R4>);
    set_error("fft: no mixed-radix kernel for in=%d out=%d", in_kind, out_kind);
    return JSDR_ERR;
}

bool mixed_plan(int n, MixedPlan &p)
{
    p = MixedPlan();
    if (n == 9600) {
        const int r[5] = {16, 8, 5, 5, 3};
        p.n = n;
        p.nrad = 5;
        for (int i = 0; i < 5; i++) p.radix[i] = r[i];
        p.threads = 512;
    } else if (n == 4800) {
        const int r[5] = {16, 4, 5, 5, 3};
        p.n = n;
        p.nrad = 5;
        for (int i = 0; i < 5; i++) p.radix[i] = r[i];
        p.threads = 512;
    } else if (n == 19200) {
        // 192 kHz (FUNcube Dongle Pro+): the frame does not fit one workgroup's LDS; X[k] = E[k] + W^k O[k],
        // X[k+n/2] = E[k] - W^k O[k] with E, O the 9600-point transforms of the even / odd samples
        MixedPlan hp;
        if (!mixed_plan(9600, hp)) return false;
        p = hp;
        p.n = n;
        p.split2 = true;
        p.half_tw_count = hp.tw_count;
        p.tw_count = hp.tw_count + n / 2;
        return true;
    } else {
        return false;
    }
    int P = p.radix[0];
    p.tw_count = 0;
    for (int i = 1; i < p.nrad; i++) {
        p.tw_count += mtw_size(P, p.radix[i]);
        P *= p.radix[i];
    }
    const int nw = (p.threads + 63) / 64;
    int lds_tw = 0, Pl = p.radix[0];
    for (int i = 1; i < p.nrad && i < mixed_lds_passes(n); i++) {
        lds_tw += mtw_size(Pl, p.radix[i]);
        Pl *= p.radix[i];
    }
    p.lds_bytes = sizeof(float2) * ((size_t)lds_tw + mixed_frame_elems(n)) + (sizeof(float) + sizeof(int)) * nw + 16;
    return true;
}

void mixed_twiddles(const MixedPlan &p, float2 *out)
{
    if (p.split2) {
        MixedPlan hp;
        mixed_plan(p.n / 2, hp);
        mixed_twiddles(hp, out);
        for (int k = 0; k < p.n / 2; k++) {
            const long double ang = -2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)p.n;
            out[p.half_tw_count + k] = make_float2((float)cosl(ang), (float)sinl(ang));
        }
        return;
    }
    int P = p.radix[0];
    size_t o = 0;
    for (int i = 1; i < p.nrad; i++) {
        const int R = p.radix[i];
        const long double base = -2.0L * 3.14159265358979323846264338327950288L / ((long double)P * (long double)R);
        if (mtw_direct(P, R)) {
            for (int r = 0; r < R; r++)
                for (int k = 0; k < P; k++) {
                    long double ang = base * (long double)k * (long double)r;
                    out[o++] = make_float2((float)cosl(ang), (float)sinl(ang));
                }
        } else {
            for (int k = 0; k < P; k++) {
                long double ang = base * (long double)k;
                out[o++] = make_float2((float)cosl(ang), (float)sinl(ang));
            }
        }
        P *= R;
    }
}

int mixed_launch_split2(const MixedPlan &p, const FftArgs &a, int in_kind, int out_kind, int num_cu, hipStream_t st)
{
    JSDR_REQUIRE(p.n == 19200, "fft: no two-half plan for n=%d", p.n);
    constexpr int T = 512, N = 9600;
    constexpr int O4 = mtw_size(16, 8) + mtw_size(16 * 8, 5);
    const size_t lds = sizeof(float2) * ((size_t)O4 + 2 * (size_t)mixed_frame_elems(N)) + (sizeof(float) + sizeof(int)) * ((T + 63) / 64) + 16;
    FftArgs h = a;
    h.split = 1;  // half transform hf reads the even (hf even) / odd samples of frame hf >> 1
    const float2 *wcomb = a.tw + p.half_tw_count;
    static const int mgrid = [] { const char *e = knob("JSDR_MIXED_GRID"); return e ? atoi(e) : 16; }();
    const long long cap = (long long)num_cu * mgrid;
    const unsigned grid = (unsigned)(a.nframes < cap ? a.nframes : cap);
    auto go = [&](auto kern) -> int {
        JSDR_LDS_ATTR(kern, lds);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(T), lds, st, h, wcomb);
        JSDR_LAUNCH_CHECK();
        return JSDR_OK;
    };
    if (in_kind == IN_I16 && out_kind == OUT_PSD) return go(k_fft_mixed_dual<T, IN_I16, OUT_PSD>);
    if (in_kind == IN_F32 && out_kind == OUT_PSD) return go(k_fft_mixed_dual<T, IN_F32, OUT_PSD>);
    if (in_kind == IN_F32 && out_kind == OUT_SPEC) return go(k_fft_mixed_dual<T, IN_F32, OUT_SPEC>);
    if (in_kind == IN_I16 && out_kind == OUT_SPEC) return go(k_fft_mixed_dual<T, IN_I16, OUT_SPEC>);
    JSDR_REQUIRE(false, "fft: no kernel for in=%d out=%d", in_kind, out_kind);
}

int mixed_launch(const MixedPlan &p, const FftArgs &a, int in_kind, int out_kind, int grid, hipStream_t st)
{
    if (p.n == 9600) return mixed_launch_t<9600, 512, 16, 8, 5, 5, 3>(p, a, in_kind, out_kind, grid, st);
    if (p.n == 4800) return mixed_launch_t<4800, 512, 16, 4, 5, 5, 3>(p, a, in_kind, out_kind, grid, st);
    set_error("fft: no mixed-radix plan for n=%d", p.n);
    return JSDR_ERR;
}

}  // namespace jsdr
