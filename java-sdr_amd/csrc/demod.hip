// demod.hip -- the demod.java AM/FM chain, demod.java:341-483, batched over S streams:
//   21-tap complex float FIR band-pass (filter(), :378-396), down-conversion NCO (:423-434), AM envelope with its
//   running mean (:448-451) / FM quadrature-delay detector (:453-461), per-frame maximum, AGC and the float ->
//   int16 stereo output (:465-481).
// Compiled with -ffp-contract=off: every float multiply and add rounds separately, as Java's do.
//
// What is sequential in the reference and how it is laid out here:
//   * the FIR delay line and the FM detector's previous sample carry over frames and calls: 21 samples of halo
//     are re-read (from the stream buffer, or from a per-stream history of the previous call), the previous
//     mixed sample is recomputed from that halo -- nothing is serial;
//   * the NCO phase `car` is a float accumulator: input independent, so the host steps it in float exactly as
//     the reference does and a kernel turns the call's phases into (cos, sin) pairs once for all streams;
//   * the AM mean avg = (k*avg + a_k)/(k+1) is a data-dependent float recurrence over the frame: one LANE per
//     frame walks it (64 frames per wave, amplitudes transposed through LDS so that global reads stay coalesced);
//   * max / AGC need the whole frame: the output is a second pass over the demodulated floats.
// Kernels: k_demod_front (raw -> demodulated float per sample + per-frame max), k_demod_mean (AM only),
// k_demod_out (-> int16 L,R), k_demod_state (history for the next call), k_demod_nco.
#include "common.h"
#include <math.h>
#include <stdlib.h>
#include <vector>

namespace jsdr {

enum { MODE_OFF = 0, MODE_RAW = 1, MODE_AM = 2, MODE_NFM = 3, MODE_WFM = 4 };  // demod.java:39-43
constexpr int DHALO = 21;   // 20 older samples of the 21-tap filter + the FM detector's previous sample
constexpr int DTILE = 2048;

struct DemodConst {
    float w[21];
    float fmgain;
    int mode, dofir, dodwn, doagc;
};

struct DemodArgs {
    const int *raw;        // int16 pairs [S][stride]
    const float2 *rawf;    // or float pairs
    long long stride_pairs;
    long long L;           // samples per stream in this call
    int n;                 // samples per frame
    int nfr;               // frames per stream in this call
    int ic, qc;
    const float2 *hist;    // [S][21] filter input before the call
    const float2 *lilq;    // [S] FM detector state before the call
    const float2 *nco;     // [L] (cos, sin) of the carrier phase per sample (dodwn)
    float *d;              // [S][L] demodulated value per sample (sam[s] after :441-462)
    unsigned *fmax_bits;   // [S][nfr] bit pattern of max |d| (zeroed before launch)
    DemodConst c;
};

template <bool F32IN>
__device__ __forceinline__ float2 demod_in(const DemodArgs &a, const int *raw, const float2 *rawf, const float2 *hist,
                                           long long g)
{
    if (g < 0) return hist[DHALO + g];
    if (F32IN) return rawf[g];
    const int w = raw[g];
    return make_float2(i16_to_float_java(java_short_add((int)(short)(w & 0xffff), a.ic)),
                       i16_to_float_java(java_short_add(w >> 16, a.qc)));
}

// filter() (:378-396) + mixer (:423-434) at ONE sample whose window sits in xs[first .. first+20] (newest last);
// used for the sample just before a tile and for the state kernel
template <class IDX>
__device__ __forceinline__ float2 demod_mixed_at(const DemodConst &c, const float2 *xs, IDX idx, int newest, const float2 *nco,
                                                 long long g)
{
    float2 v = xs[idx(newest)];
    if (c.dofir) {
        float oi = 0.0f, oq = 0.0f;
#pragma unroll
        for (int k = 0; k < 21; k++) {  // ring order: the newest sample meets w[0]
            const float2 x = xs[idx(newest - k)];
            oi = oi + x.x * c.w[k];
            oq = oq + x.y * c.w[k];
        }
        v = make_float2(oi, oq);
    }
    if (c.dodwn) {
        const float2 cs = nco[g];
        v = make_float2(v.x * cs.x - v.y * cs.y, v.x * cs.y + v.y * cs.x);
    }
    return v;
}

// LDS image of the tile's filter input: one pad slot per 8 so that the 8-sample lane stride of the blocked
// filter below walks distinct banks
__device__ __forceinline__ int xpad8(int i) { return i + (i >> 3); }

typedef float v2f __attribute__((ext_vector_type(2)));

// the sample just before the tile: xs[xpad8(20)] is x(g0 - 1)
__device__ __forceinline__ float2 demod_mixed(const DemodConst &c, const float2 *xs, const float2 *nco, long long g)
{
    return demod_mixed_at(c, xs, [](int i) { return xpad8(i); }, DHALO - 1, nco, g);
}

// One tile = 2048 samples of one frame of one stream; every thread owns 8 CONSECUTIVE samples: their 21-tap
// windows overlap, so 28 LDS reads feed 8 outputs (3.5 per sample instead of 21), and I/Q ride in one packed
// register pair: acc = acc + x*w is v_pk_mul_f32 + v_pk_add_f32, each half rounded separately exactly like the
// reference's two scalar statements (:388-389).
constexpr int DPER = DTILE / 256;  // 8 samples per thread and tile
constexpr int DXS = DTILE + DHALO + (DTILE + DHALO) / 8 + 1;

// tile j of frame f of stream s -> this thread's 8 detected samples dv[] (sam[s] after :441-462) and the running
// maximum of their magnitudes; xs / last are the workgroup's LDS (two barriers inside, none after)
template <bool F32IN, bool CHAINED>
__device__ __forceinline__ void demod_tile(const DemodArgs &a, const int s, const int f, const int j, float2 *xs, float2 *last,
                                           float (&dv)[DPER], unsigned &mbits)
{
    constexpr int PER = DPER;
    const int tid = threadIdx.x;
    const long long g0 = (long long)f * a.n + (long long)j * DTILE;
    const int len = (a.n - j * DTILE) < DTILE ? (a.n - j * DTILE) : DTILE;
    const int *raw = a.raw + (long long)s * a.stride_pairs;
    const float2 *rawf = a.rawf + (long long)s * a.stride_pairs;
    const float2 *hist = a.hist + (long long)s * DHALO;
    // FM detector across tiles of one workgroup: the previous tile's last sample, read before the staging barrier
    // below lets anyone overwrite it
    float2 carried = make_float2(0.0f, 0.0f);
    if (CHAINED && tid == 0) carried = last[255];
    // xs[xpad8(i)] = x(g0 - 21 + i); beyond the tile's end: zeros (their outputs are never stored)
    if (g0 >= DHALO) {
        // every tile but the call's first: all nine loads of a thread are in flight before the first conversion /
        // LDS store (as a plain loop the compiler waits for each load in turn: nine memory latencies per tile)
        // addressing: one uniform base per tile and a 32-bit offset per load, clamped inside the call (values
        // beyond the tile's end are dropped below)
        constexpr int NLD = (DTILE + DHALO + 255) / 256;
        const long long room = a.L - 1 - (g0 - DHALO);
        const unsigned relmax = (unsigned)(room < (long long)(DTILE + DHALO - 1) ? room : (long long)(DTILE + DHALO - 1));
        const int *rb = raw + (g0 - DHALO);
        const float2 *rbf = rawf + (g0 - DHALO);
        int w[NLD];
        float2 wf[NLD];
#pragma unroll
        for (int q = 0; q < NLD; q++) {
            unsigned rel = (unsigned)(tid + 256 * q);
            rel = rel < relmax ? rel : relmax;
            if (F32IN)
                wf[q] = rbf[rel];
            else
                w[q] = rb[rel];
        }
        const bool full = len == DTILE;  // every tile of a frame but a ragged last one
#pragma unroll
        for (int q = 0; q < NLD; q++) {
            const int i = tid + 256 * q;
            float2 v;
            if (F32IN)
                v = wf[q];
            else
                v = make_float2(i16_to_float_java(java_short_add((int)(short)(w[q] & 0xffff), a.ic)),
                                i16_to_float_java(java_short_add(w[q] >> 16, a.qc)));
            // xpad8(tid + 256 q) = xpad8(tid) + 288 q: one address register, constant offsets
            float2 *slot = xs + (tid + (tid >> 3)) + 288 * q;
            if (full) {
                if (q < NLD - 1 || i < DTILE + DHALO) *slot = v;
            } else if (i < DTILE + DHALO) {
                *slot = (i < len + DHALO) ? v : make_float2(0.0f, 0.0f);
            }
        }
    } else {
        for (int i = tid; i < DTILE + DHALO; i += 256)
            xs[xpad8(i)] = (i < len + DHALO) ? demod_in<F32IN>(a, raw, rawf, hist, g0 - DHALO + i) : make_float2(0.0f, 0.0f);
    }
    __syncthreads();
    const int t0 = tid * PER;  // first sample of this thread within the tile
    v2f m[PER];                // filtered + mixed samples
    if (a.c.dofir) {
        v2f x[PER + 20];  // x[q] = input sample t0 - 20 + q
#pragma unroll
        for (int q = 0; q < PER + 20; q++) {
            // xpad8(8 tid + c) = 9 tid + c + (c >> 3)
            const float2 v = xs[9 * tid + (1 + q) + ((1 + q) >> 3)];
            x[q] = (v2f){v.x, v.y};
        }
#pragma unroll
        for (int u = 0; u < PER; u++) {
            v2f acc = (v2f){0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < 21; k++) acc = acc + x[u + 20 - k] * a.c.w[k];  // ring order: newest sample meets w[0]
            m[u] = acc;
        }
    } else {
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const float2 v = xs[9 * tid + (DHALO + u) + ((DHALO + u) >> 3)];
            m[u] = (v2f){v.x, v.y};
        }
    }
    if (a.c.dodwn) {  // :423-434
        const float2 *nc = a.nco + g0 + t0;
        float2 cs[PER];
        if (((g0 + t0) & 1) == 0) {  // 16-byte aligned pairs (always, unless the frame length is odd)
#pragma unroll
            for (int u = 0; u < PER; u += 2) {
                // (an odd tile end reads one entry past its last sample: inside the table, which has spare slots)
                const float4 c2 = (t0 + u < len) ? reinterpret_cast<const float4 *>(nc)[u / 2] : make_float4(1.0f, 0.0f, 1.0f, 0.0f);
                cs[u] = make_float2(c2.x, c2.y);
                cs[u + 1] = make_float2(c2.z, c2.w);
            }
        } else {
#pragma unroll
            for (int u = 0; u < PER; u++) cs[u] = (t0 + u < len) ? nc[u] : make_float2(1.0f, 0.0f);
        }
#pragma unroll
        for (int u = 0; u < PER; u++) {
            // (x*c - y*s, x*s + y*c) as x*(c, s) + (-y, y)*(s, c): two packed products and one packed sum (x - y == x + (-y) exactly)
            const v2f v = m[u];
            const v2f c = (v2f){cs[u].x, cs[u].y};
            const v2f ny = (v2f){-v.y, v.y};
            m[u] = __builtin_shufflevector(v, v, 0, 0) * c + ny * __builtin_shufflevector(c, c, 1, 0);
        }
    }
    // the FM detector's previous sample: the neighbour thread's last one; thread 0 recomputes it from the halo,
    // or takes the carried state at the first sample of the call
    float2 prev = make_float2(0.0f, 0.0f);
    if (a.c.mode == MODE_NFM || a.c.mode == MODE_WFM) {
        last[tid] = make_float2(m[PER - 1].x, m[PER - 1].y);
        __syncthreads();
        if (tid > 0)
            prev = last[tid - 1];
        else if (CHAINED)
            prev = carried;
        else if (g0 > 0)
            prev = demod_mixed(a.c, xs, a.nco, g0 - 1);
        else
            prev = a.lilq[s];
    }
    // max |d| travels as a bit pattern: non-negative floats order like unsigned ints, and any NaN beats every
    // number -- Math.max's NaN propagation (:463) for free
#pragma unroll
    for (int u = 0; u < PER; u++) {
        const v2f mm = m[u];
        if (a.c.mode == MODE_OFF) {
            dv[u] = 0.0f;
        } else if (a.c.mode == MODE_RAW) {
            dv[u] = mm.x;
        } else if (a.c.mode == MODE_AM) {
            // :449 (float)Math.sqrt((double)..): the correctly rounded double root of a float, rounded to float, IS the
            // correctly rounded float root (53 >= 2*24 + 2 bits), which is what sqrtf compiles to here (v_sqrt_f32
            // plus a one-ulp fix-up; __fsqrt_rn is NOT correctly rounded) -- all FP32, no v_rsq_f64 chain
            dv[u] = __builtin_sqrtf(mm.x * mm.x + mm.y * mm.y);
        } else {
            dv[u] = ((prev.x * mm.y) - (prev.y * mm.x)) * a.c.fmgain;
            prev = make_float2(mm.x, mm.y);
        }
        if (t0 + u < len) {
            const unsigned bb = __float_as_uint(dv[u]) & 0x7fffffffu;
            mbits = bb > mbits ? bb : mbits;
        }
    }
}

__device__ __forceinline__ unsigned demod_block_max(unsigned mbits, unsigned *red)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned o = __shfl_xor(mbits, off, 64);
        mbits = o > mbits ? o : mbits;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mbits;
    __syncthreads();
    unsigned mx = red[0];
    for (int w = 1; w < 4; w++) mx = red[w] > mx ? red[w] : mx;
    return mx;
}

// AM (whose running mean needs the whole frame first) and frames longer than 5 tiles: one tile per workgroup, the
// detected samples go to HBM as floats and k_demod_mean / k_demod_out finish the frame
template <bool F32IN>
__global__ __launch_bounds__(256) void k_demod_front(DemodArgs a)
{
    constexpr int PER = DPER;
    __shared__ float2 xs[DXS];
    __shared__ float2 last[256];
    __shared__ unsigned red[4];
    const int tid = threadIdx.x;
    const int s = blockIdx.y;
    const int tiles_per_frame = (a.n + DTILE - 1) / DTILE;
    const int f = blockIdx.x / tiles_per_frame;
    const int j = blockIdx.x - f * tiles_per_frame;
    const long long g0 = (long long)f * a.n + (long long)j * DTILE;
    const int len = (a.n - j * DTILE) < DTILE ? (a.n - j * DTILE) : DTILE;
    const int t0 = tid * PER;
    float dv[PER];
    unsigned mbits = 0;
    demod_tile<F32IN, false>(a, s, f, j, xs, last, dv, mbits);
    float *d = a.d + (long long)s * a.L + g0 + t0;
    if (t0 + PER <= len && (((long long)s * a.L + g0) & 3) == 0) {
        reinterpret_cast<float4 *>(d)[0] = make_float4(dv[0], dv[1], dv[2], dv[3]);
        reinterpret_cast<float4 *>(d)[1] = make_float4(dv[4], dv[5], dv[6], dv[7]);
    } else {
#pragma unroll
        for (int u = 0; u < PER; u++)
            if (t0 + u < len) d[u] = dv[u];
    }
    const unsigned mx = demod_block_max(mbits, red);
    if (tid == 0) atomicMax(&a.fmax_bits[(long long)s * a.nfr + f], mx);
}

// Java's (int) of a float -- NaN -> 0, out of range saturates, else truncation -- is what v_cvt_i32_f32 does by
// itself (C's (int) is undefined out of range, so the compiler may not assume it: spelled out, the three cases
// cost five more instructions per sample)
__device__ __forceinline__ int demod_f2i(float v)
{
    int r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}

// (short)(sam * 32767f) to both channels (:478-481): the low half of the int, twice
__device__ __forceinline__ int demod_lr(float x) { const int sv = demod_f2i(x * 32767.0f); return (int)__builtin_amdgcn_perm((unsigned)sv, (unsigned)sv, 0x01000100u); }

// Every mode but AM, frames of at most NT tiles: one workgroup per frame keeps the detected samples in registers,
// takes the frame maximum itself and writes the int16 audio (:465-481) -- no float round trip through HBM and no
// second pass (4 B in, 4 B out per sample).
template <bool F32IN, int NT>
__global__ __launch_bounds__(256) void k_demod_fused(DemodArgs a, int *__restrict__ out, long long out_stride_pairs,
                                                     float *__restrict__ stats)
{
    constexpr int PER = DPER;
    __shared__ float2 xs[DXS];
    __shared__ float2 last[256];
    __shared__ unsigned red[4];
    const int tid = threadIdx.x;
    const int s = blockIdx.y, f = blockIdx.x;
    const int t0 = tid * PER;
    float dv[NT][PER];
    unsigned mbits = 0;
#pragma unroll
    for (int j = 0; j < NT; j++) {
        if (j) {
            __syncthreads();  // the previous tile's LDS image is still being read
            demod_tile<F32IN, true>(a, s, f, j, xs, last, dv[j], mbits);
        } else {
            demod_tile<F32IN, false>(a, s, f, j, xs, last, dv[j], mbits);
        }
    }
    const float mx = __uint_as_float(demod_block_max(mbits, red));
    const float scale = a.c.doagc ? 1.0f / mx : 1.0f;
    const long long F = (long long)s * a.nfr + f;
    if (stats && tid == 0) {  // the reference's `max` / `avg` fields after the frame
        stats[2 * F] = mx;
        stats[2 * F + 1] = 0.0f;
    }
#pragma unroll
    for (int j = 0; j < NT; j++) {
        const int len = (a.n - j * DTILE) < DTILE ? (a.n - j * DTILE) : DTILE;
        const long long g = (long long)f * a.n + (long long)j * DTILE + t0;
        int *dst = out + (long long)s * out_stride_pairs + g;
        int o[PER];
#pragma unroll
        for (int u = 0; u < PER; u++) {
            o[u] = demod_lr(dv[j][u] * scale);
        }
        if (t0 + PER <= len && ((((long long)s * out_stride_pairs + g) & 3) == 0)) {
            reinterpret_cast<int4 *>(dst)[0] = make_int4(o[0], o[1], o[2], o[3]);
            reinterpret_cast<int4 *>(dst)[1] = make_int4(o[4], o[5], o[6], o[7]);
        } else {
#pragma unroll
            for (int u = 0; u < PER; u++)
                if (t0 + u < len) dst[u] = o[u];
        }
    }
}

// history for the next call: the last 21 filter inputs (only while the filter runs: the reference's ring is not
// written otherwise) and the FM detector's last sample (only in the FM modes, :453-461)
template <bool F32IN>
__global__ __launch_bounds__(64) void k_demod_state(DemodArgs a, float2 *hist_new, float2 *lilq_new, int nstreams)
{
    __shared__ float2 xs[2 * DHALO];
    const int s = blockIdx.x;
    const int tid = threadIdx.x;
    if (s >= nstreams) return;
    const int *raw = a.raw + (long long)s * a.stride_pairs;
    const float2 *rawf = a.rawf + (long long)s * a.stride_pairs;
    const float2 *hist = a.hist + (long long)s * DHALO;
    // xs[i] = x(L - 42 + i): enough for the filter at the last sample even when L < 21
    if (tid < 2 * DHALO) {
        const long long g = a.L - 2 * DHALO + tid;
        xs[tid] = (g >= -DHALO) ? demod_in<F32IN>(a, raw, rawf, hist, g) : make_float2(0.0f, 0.0f);
    }
    __syncthreads();
    if (tid < DHALO) hist_new[(long long)s * DHALO + tid] = a.c.dofir ? xs[DHALO + tid] : hist[tid];
    if (tid == 0) {
        const bool fm = a.c.mode == MODE_NFM || a.c.mode == MODE_WFM;
        lilq_new[s] = (fm && a.L > 0) ? demod_mixed_at(a.c, xs, [](int i) { return i; }, 2 * DHALO - 1, a.nco, a.L - 1) : a.lilq[s];
    }
}

// (float)Math.cos(car), (float)Math.sin(car) (:425-426) for the call's phases, once for all streams
__global__ __launch_bounds__(256) void k_demod_nco(const float *__restrict__ car, long long n, float2 *__restrict__ nco)
{
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= n) return;
    const double c = (double)car[g];
    nco[g] = make_float2((float)cos(c), (float)sin(c));
}

// AM running mean (:450): avg = ((s/2)*avg + sam[s]) / (s/2 + 1), one lane per frame.  64 frames per wave; their
// amplitudes are read row by row (256 contiguous bytes per instruction) and transposed through LDS.
__global__ __launch_bounds__(64) void k_demod_mean(const float *__restrict__ d, long long L, int n, int nfr,
                                                   long long nframes_total, float *__restrict__ favg)
{
    __shared__ float tile[64][65];
    const int lane = threadIdx.x;
    const long long F0 = (long long)blockIdx.x * 64;
    const long long F = F0 + lane;
    const bool valid = F < nframes_total;
    // base of this lane's frame; rows are addressed through readlane below
    const long long base = valid ? (F / nfr) * L + (F % nfr) * (long long)n : 0;
    const int nrows = (int)((nframes_total - F0) < 64 ? (nframes_total - F0) : 64);
    float avg = 0.0f;
    for (int c0 = 0; c0 < n; c0 += 64) {
        // the whole 64 x 64 tile in flight at once: sixteen 16-byte loads per lane (four rows of 64 samples per
        // wave instruction), 16 KB per wave -- the kernel is a 4-byte-per-sample stream read whose only problem
        // is memory-level parallelism (64-thread blocks, 9 per CU)
        if ((n & 3) == 0 && (L & 3) == 0 && c0 + 64 <= n) {
            float4 v[16];
            const int sub = lane >> 4, col = 4 * (lane & 15);
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int r = (4 * u + sub) < nrows ? (4 * u + sub) : (nrows - 1);
                const long long rb = __shfl(base, r, 64);
                v[u] = *reinterpret_cast<const float4 *>(d + rb + c0 + col);
            }
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int r = 4 * u + sub;
                if (r < nrows) {
                    tile[r][col] = v[u].x;
                    tile[r][col + 1] = v[u].y;
                    tile[r][col + 2] = v[u].z;
                    tile[r][col + 3] = v[u].w;
                }
            }
        } else {
            for (int r0 = 0; r0 < nrows; r0 += 16) {
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    const int r = (r0 + u) < nrows ? (r0 + u) : (nrows - 1);
                    const long long rb = __shfl(base, r, 64);
                    v[u] = (c0 + lane < n) ? d[rb + c0 + lane] : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < 16; u++)
                    if (r0 + u < nrows) tile[r0 + u][lane] = v[u];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int cn = (n - c0) < 64 ? (n - c0) : 64;
        if (valid) {
            for (int c = 0; c < cn; c++) {
                const int k = c0 + c;
                avg = ((float)k * avg + tile[lane][c]) / (float)(k + 1);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    if (valid) favg[F] = avg;
}

// :465-481: max -= avg (AM); sam = (AM ? sam - avg : sam) * (doagc ? 1.0f/max : 1); (short)(sam * 32767f) to L and R.
// grid (chunks of 1024 samples within a frame, frame, stream): the frame's statistics are uniform per workgroup
__global__ __launch_bounds__(256) void k_demod_out(const float *__restrict__ d, long long L, int n, int nfr, int mode,
                                                   int doagc, const unsigned *__restrict__ fmax_bits,
                                                   const float *__restrict__ favg, int *__restrict__ out,
                                                   long long out_stride_pairs, float *__restrict__ stats)
{
    const int f = blockIdx.y, s = blockIdx.z;
    const long long F = (long long)s * nfr + f;
    float mx = __uint_as_float(fmax_bits[F]);
    const float avg = (mode == MODE_AM) ? favg[F] : 0.0f;
    if (mode == MODE_AM) mx -= avg;
    const float scale = doagc ? 1.0f / mx : 1.0f;
    if (stats && blockIdx.x == 0 && threadIdx.x == 0) {  // the reference's `max` / `avg` fields after the frame
        stats[2 * F] = mx;
        stats[2 * F + 1] = avg;
    }
    const int t = (blockIdx.x * 256 + threadIdx.x) * 4;  // within the frame
    if (t >= n) return;
    const long long g = (long long)f * n + t;
    const float *src = d + (long long)s * L + g;
    int *dst = out + (long long)s * out_stride_pairs + g;
    float v[4];
    const bool vec = (t + 4 <= n) && ((((long long)s * L + g) & 3) == 0) && ((((long long)s * out_stride_pairs + g) & 3) == 0);
    if (vec) {
        const float4 q = *reinterpret_cast<const float4 *>(src);
        v[0] = q.x, v[1] = q.y, v[2] = q.z, v[3] = q.w;
    } else {
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = (t + u < n) ? src[u] : 0.0f;
    }
    int o[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        float x = v[u];
        if (mode == MODE_AM) x = x - avg;
        o[u] = demod_lr(x * scale);
    }
    if (vec) {
        *reinterpret_cast<int4 *>(dst) = make_int4(o[0], o[1], o[2], o[3]);
    } else {
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (t + u < n) dst[u] = o[u];
    }
}

}  // namespace jsdr

using namespace jsdr;

struct jsdr_demod {
    int rate = 0, n = 0, nstreams = 0;
    long long max_batch = 0;
    int mode = MODE_OFF, dofir = 0, dodwn = 0, doagc = 0;
    int flo = (-2147483647 - 1), fhi = 2147483647;
    float wfir[21] = {0};
    float phi = 0.0f, car = 0.0f;
    DevBuf<float2> hist[2], lilq[2];
    PinnedStage pin;        // receive(): [frame in | audio out]
    bool pin_tried = false;
    int cur = 0;
    // carrier tables, double-buffered: call k fills set k&1 on the copy stream while call k-1's kernels still read
    // the other one (no bubble between calls for the host's phase recurrence and its upload)
    DevBuf<float2> nco[2];
    DevBuf<float> car_dev[2];
    float *car_pinned[2] = {nullptr, nullptr};
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_table[2] = {nullptr, nullptr}, ev_used[2] = {nullptr, nullptr};
    bool used[2] = {false, false};
    unsigned calls = 0;
    DevBuf<float> d, favg, stats;
    DevBuf<unsigned> fmax;
    DevBuf<float> stage_in;  // one frame, receive_f32
    DevBuf<int> stage_out;
    int last_nfr = 0;
    // optional per-kernel HIP-event timing (bench.py's roofline leg), same contract as jsdr_bpsk_profile_*
    bool prof_on = false;
    struct Rec {
        int k;
        hipEvent_t a, b;
    };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
};

enum { DK_NCO = 0, DK_FRONT, DK_STATE, DK_MEAN, DK_OUT, DK_COUNT };
static const char *const kDemodKernels[DK_COUNT] = {"k_demod_nco", "k_demod_front", "k_demod_state", "k_demod_mean",
                                                    "k_demod_out"};
struct DemodProf {
    jsdr_demod *h;
    hipStream_t st;
    hipEvent_t a = nullptr, b = nullptr;
    int k;
    static hipEvent_t get(jsdr_demod *h)
    {
        if (!h->pool.empty()) {
            hipEvent_t e = h->pool.back();
            h->pool.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    DemodProf(jsdr_demod *h_, int k_, hipStream_t st_) : h(h_), st(st_), k(k_)
    {
        if (h->prof_on) {
            a = get(h);
            b = get(h);
            (void)hipEventRecord(a, st);
        }
    }
    ~DemodProf()
    {
        if (h->prof_on && a && b) {
            (void)hipEventRecord(b, st);
            h->recs.push_back({k, a, b});
        }
    }
};

static double dsinl(double x) { return (double)sinl((long double)x); }
static double dcosl(double x) { return (double)cosl((long double)x); }

template <bool F32IN>
static int demod_run(jsdr_demod *h, const int16_t *raw_dev, const float *rawf_dev, int64_t stride_i16, int64_t L, int ic,
                     int qc, int16_t *audio_dev, int64_t audio_stride_i16, hipStream_t st)
{
    JSDR_REQUIRE(h, "demod: null handle");
    JSDR_REQUIRE((raw_dev || rawf_dev) && audio_dev, "demod: null buffer");
    JSDR_REQUIRE(L > 0 && L <= h->max_batch && L % h->n == 0,
                 "demod: nsamples=%lld must be a positive multiple of the frame (%d) and at most max_batch_samples=%lld",
                 (long long)L, h->n, h->max_batch);
    JSDR_REQUIRE((stride_i16 & 1) == 0 && (h->nstreams == 1 || stride_i16 >= 2 * L) && (audio_stride_i16 & 1) == 0 &&
                     (h->nstreams == 1 || audio_stride_i16 >= 2 * L),
                 "demod: stream stride too small for %lld samples", (long long)L);
    const int S = h->nstreams;
    const int nfr = (int)(L / h->n);
    DemodArgs a;
    a.raw = reinterpret_cast<const int *>(raw_dev);
    a.rawf = reinterpret_cast<const float2 *>(rawf_dev);
    a.stride_pairs = stride_i16 / 2;
    a.L = L;
    a.n = h->n;
    a.nfr = nfr;
    a.ic = ic;
    a.qc = qc;
    a.hist = h->hist[h->cur].p;
    a.lilq = h->lilq[h->cur].p;
    const int set = (int)(h->calls++ & 1);
    a.nco = h->nco[set].p;
    a.d = h->d.p;
    a.fmax_bits = h->fmax.p;
    memcpy(a.c.w, h->wfir, sizeof(a.c.w));
    a.c.fmgain = (float)h->rate / (MODE_NFM == h->mode ? 5000.0f : 75000.0f);  // :410
    a.c.mode = h->mode;
    a.c.dofir = h->dofir;
    a.c.dodwn = h->dodwn;
    a.c.doagc = h->doagc;
    if (h->dodwn) {
        // :427-429 in float, exactly as the reference steps it
        if (h->used[set]) JSDR_HIP_TRY(hipEventSynchronize(h->ev_used[set]));  // the call before last has let go of this set
        float *tab = h->car_pinned[set];
        float car = h->car;
        const float phi = h->phi, two_pi = (float)(2 * 3.14159265358979323846);
        // a serial float recurrence, on the host; kept as a (well predicted) branch so that the dependent chain per
        // sample is the one subtraction -- as a select it is subtract + add + blend, four times slower, and longer
        // than the GPU takes for the whole batch
        for (int64_t g = 0; g < L; g++) {
            tab[g] = car;
            car -= phi;
            if (__builtin_expect(car < 0.0f, 0)) {
                asm volatile("" : "+x"(car));
                car += two_pi;
            }
        }
        h->car = car;
        JSDR_HIP_TRY(hipMemcpyAsync(h->car_dev[set].p, tab, sizeof(float) * (size_t)L, hipMemcpyHostToDevice, h->copy_stream));
        {
            DemodProf ps(h, DK_NCO, h->copy_stream);
            hipLaunchKernelGGL(k_demod_nco, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, h->copy_stream, h->car_dev[set].p,
                               (long long)L, h->nco[set].p);
            JSDR_LAUNCH_CHECK();
        }
        JSDR_HIP_TRY(hipEventRecord(h->ev_table[set], h->copy_stream));
        JSDR_HIP_TRY(hipStreamWaitEvent(st, h->ev_table[set], 0));
    }
    const int tiles_per_frame = (h->n + DTILE - 1) / DTILE;
    // every mode but AM, frames of up to 5 tiles (the reference's 9600-sample default included): one kernel from
    // raw samples to int16 audio; JSDR_DEMOD_FUSED=0 keeps the three-kernel path for A/B runs
    static const bool fused_ok = [] {
        const char *e = knob("JSDR_DEMOD_FUSED");
        return !(e && e[0] == '0');
    }();
    const bool fused = fused_ok && h->mode != MODE_AM && tiles_per_frame <= 5;
    if (fused) {
        DemodProf ps(h, DK_FRONT, st);
        int *outp = reinterpret_cast<int *>(audio_dev);
        const long long osp = (long long)(audio_stride_i16 / 2);
        const dim3 grid((unsigned)nfr, (unsigned)S);  // (streams fastest instead, for carrier-table locality: measured, no gain)
        switch (tiles_per_frame) {
            case 1: hipLaunchKernelGGL((k_demod_fused<F32IN, 1>), grid, dim3(256), 0, st, a, outp, osp, h->stats.p); break;
            case 2: hipLaunchKernelGGL((k_demod_fused<F32IN, 2>), grid, dim3(256), 0, st, a, outp, osp, h->stats.p); break;
            case 3: hipLaunchKernelGGL((k_demod_fused<F32IN, 3>), grid, dim3(256), 0, st, a, outp, osp, h->stats.p); break;
            case 4: hipLaunchKernelGGL((k_demod_fused<F32IN, 4>), grid, dim3(256), 0, st, a, outp, osp, h->stats.p); break;
            default: hipLaunchKernelGGL((k_demod_fused<F32IN, 5>), grid, dim3(256), 0, st, a, outp, osp, h->stats.p); break;
        }
        JSDR_LAUNCH_CHECK();
    } else {
        JSDR_HIP_TRY(hipMemsetAsync(h->fmax.p, 0, sizeof(unsigned) * (size_t)S * nfr, st));
        DemodProf ps(h, DK_FRONT, st);
        hipLaunchKernelGGL(k_demod_front<F32IN>, dim3((unsigned)(tiles_per_frame * nfr), (unsigned)S), dim3(256), 0, st, a);
        JSDR_LAUNCH_CHECK();
    }
    {
        DemodProf ps(h, DK_STATE, st);
        hipLaunchKernelGGL(k_demod_state<F32IN>, dim3((unsigned)S), dim3(64), 0, st, a, h->hist[h->cur ^ 1].p,
                           h->lilq[h->cur ^ 1].p, S);
        JSDR_LAUNCH_CHECK();
    }
    h->cur ^= 1;
    const long long nft = (long long)S * nfr;
    if (h->mode == MODE_AM) {
        DemodProf ps(h, DK_MEAN, st);
        hipLaunchKernelGGL(k_demod_mean, dim3((unsigned)((nft + 63) / 64)), dim3(64), 0, st, h->d.p, (long long)L, h->n, nfr,
                           nft, h->favg.p);
        JSDR_LAUNCH_CHECK();
    }
    if (!fused) {
        DemodProf ps(h, DK_OUT, st);
        hipLaunchKernelGGL(k_demod_out, dim3((unsigned)((h->n + 1023) / 1024), (unsigned)nfr, (unsigned)S), dim3(256), 0, st, h->d.p, (long long)L,
                           h->n, nfr, h->mode, h->doagc, h->fmax.p, h->favg.p, reinterpret_cast<int *>(audio_dev),
                           (long long)(audio_stride_i16 / 2), h->stats.p);
        JSDR_LAUNCH_CHECK();
    }
    if (h->dodwn) {
        JSDR_HIP_TRY(hipEventRecord(h->ev_used[set], st));
        h->used[set] = true;
    }
    h->last_nfr = nfr;
    return JSDR_OK;
}

extern "C" {

int jsdr_demod_create(jsdr_demod **out, int rate, int nsamples_per_frame, int nstreams, int64_t max_batch_samples)
{
    JSDR_REQUIRE(out, "jsdr_demod_create: null handle pointer");
    *out = nullptr;
    JSDR_REQUIRE(rate > 0 && nsamples_per_frame > 0 && nstreams > 0, "jsdr_demod_create: rate, frame and nstreams must be positive");
    if (max_batch_samples <= 0) max_batch_samples = nsamples_per_frame;
    JSDR_REQUIRE(max_batch_samples % nsamples_per_frame == 0, "jsdr_demod_create: max_batch_samples must be whole frames");
    JSDR_REQUIRE(nstreams <= 65535 && max_batch_samples / nsamples_per_frame <= 65535,
                 "jsdr_demod_create: at most 65535 streams and 65535 frames per call");
    JSDR_REQUIRE((long long)nstreams * (max_batch_samples / nsamples_per_frame) < (1LL << 31) &&
                     (max_batch_samples / nsamples_per_frame) * ((nsamples_per_frame + DTILE - 1) / DTILE) < (1LL << 31),
                 "jsdr_demod_create: batch too large");
    jsdr_demod *h = new jsdr_demod();
    h->rate = rate;
    h->n = nsamples_per_frame;
    h->nstreams = nstreams;
    h->max_batch = max_batch_samples;
    const size_t S = (size_t)nstreams, L = (size_t)max_batch_samples, nf = S * (L / (size_t)h->n);
    bool ok = true;
    for (int k = 0; k < 2; k++)
        ok = ok && h->hist[k].alloc(S * DHALO) == JSDR_OK && h->lilq[k].alloc(S) == JSDR_OK && h->hist[k].zero() == JSDR_OK &&
             h->lilq[k].zero() == JSDR_OK;
    for (int k = 0; k < 2; k++)
        ok = ok && h->nco[k].alloc(L + 8) == JSDR_OK && h->nco[k].zero() == JSDR_OK && h->car_dev[k].alloc(L) == JSDR_OK &&
             hipHostMalloc(reinterpret_cast<void **>(&h->car_pinned[k]), sizeof(float) * L, hipHostMallocDefault) == hipSuccess &&
             hipEventCreateWithFlags(&h->ev_table[k], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&h->ev_used[k], hipEventDisableTiming) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && h->d.alloc(S * L) == JSDR_OK &&
         h->favg.alloc(nf) == JSDR_OK && h->stats.alloc(2 * nf) == JSDR_OK && h->fmax.alloc(nf) == JSDR_OK &&
         h->stage_in.alloc(2 * (size_t)h->n) == JSDR_OK && h->stage_out.alloc((size_t)h->n) == JSDR_OK &&
         h->favg.zero() == JSDR_OK && hipDeviceSynchronize() == hipSuccess;
    if (!ok) {
        jsdr_demod_destroy(h);
        return JSDR_ERR;
    }
    *out = h;
    return JSDR_OK;
}

int jsdr_demod_destroy(jsdr_demod *h)
{
    if (!h) return JSDR_OK;
    (void)hipDeviceSynchronize();
    h->pin.release();
    for (int k = 0; k < 2; k++) {
        h->hist[k].release();
        h->lilq[k].release();
    }
    for (int k = 0; k < 2; k++) {
        h->nco[k].release();
        h->car_dev[k].release();
        if (h->car_pinned[k]) (void)hipHostFree(h->car_pinned[k]);
        if (h->ev_table[k]) (void)hipEventDestroy(h->ev_table[k]);
        if (h->ev_used[k]) (void)hipEventDestroy(h->ev_used[k]);
    }
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    h->d.release();
    h->favg.release();
    h->stats.release();
    h->fmax.release();
    h->stage_in.release();
    h->stage_out.release();
    for (auto &r : h->recs) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    for (auto e : h->pool) (void)hipEventDestroy(e);
    delete h;
    return JSDR_OK;
}

int jsdr_demod_profile_enable(jsdr_demod *h, int on)
{
    JSDR_REQUIRE(h, "jsdr_demod_profile_enable: null handle");
    h->prof_on = on != 0;
    return JSDR_OK;
}
int jsdr_demod_profile_count(void) { return DK_COUNT; }
const char *jsdr_demod_profile_name(int k) { return (k >= 0 && k < DK_COUNT) ? kDemodKernels[k] : ""; }
int jsdr_demod_profile_read(jsdr_demod *h, double *ms_total, int *launches)
{
    JSDR_REQUIRE(h && ms_total && launches, "jsdr_demod_profile_read: null argument");
    for (int k = 0; k < DK_COUNT; k++) {
        ms_total[k] = 0.0;
        launches[k] = 0;
    }
    for (auto &r : h->recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            ms_total[r.k] += ms;
            launches[r.k]++;
        }
        h->pool.push_back(r.a);
        h->pool.push_back(r.b);
    }
    h->recs.clear();
    return JSDR_OK;
}

int jsdr_demod_configure(jsdr_demod *h, int mode, int dofir, int dodwn, int doagc)
{
    JSDR_REQUIRE(h, "jsdr_demod_configure: null handle");
    JSDR_REQUIRE(mode >= MODE_OFF && mode <= MODE_WFM, "jsdr_demod_configure: mode %d outside 0..4 (demod.java:39-43)", mode);
    h->mode = mode;
    h->dofir = dofir != 0;
    h->dodwn = dodwn != 0;
    h->doagc = doagc != 0;
    return JSDR_OK;
}

// demod.weights() (:341-375) for the given band; clears the delay lines of every stream
int jsdr_demod_weights(jsdr_demod *h, int flo, int fhi, float w_out[21], float *phi_out)
{
    JSDR_REQUIRE(h, "jsdr_demod_weights: null handle");
    const int len = 21;
    h->flo = flo;
    h->fhi = fhi;
    if ((-2147483647 - 1) == flo) {
        for (int i = 0; i < len; i++) h->wfir[i] = 0;
        h->wfir[(len - 1) / 2] = 1;
    } else {
        const float rate = (float)h->rate;
        const float nlo = (float)flo / rate;
        const float nhi = (float)fhi / rate;
        const int ord = len - 1;
        const double PI = 3.14159265358979323846;
        for (int n = 0; n < len; n++) {
            if (n == ord / 2) {
                h->wfir[n] = 2.0f * (nhi - nlo);
            } else {
                h->wfir[n] = (float)((dsinl(2 * PI * nhi * (double)(n - ord / 2)) / (PI * (double)(n - ord / 2))) -
                                     (dsinl(2 * PI * nlo * (double)(n - ord / 2)) / (PI * (double)(n - ord / 2))));
            }
            h->wfir[n] *= (float)(0.54 - 0.46 * dcosl(2 * PI * (double)n / (double)ord));
        }
        h->phi = (float)(2 * PI * nlo);
        h->car = 0.0f;
    }
    JSDR_HIP_TRY(hipDeviceSynchronize());
    if (h->hist[0].zero() != JSDR_OK || h->hist[1].zero() != JSDR_OK) return JSDR_ERR;
    JSDR_HIP_TRY(hipDeviceSynchronize());
    if (w_out) memcpy(w_out, h->wfir, sizeof(h->wfir));
    if (phi_out) *phi_out = h->phi;
    return JSDR_OK;
}

int jsdr_demod_batch_i16(jsdr_demod *h, const int16_t *raw_dev, int64_t stream_stride_i16, int64_t nsamples, int ic, int qc,
                         int16_t *audio_dev, int64_t audio_stride_i16, void *stream)
{
    return demod_run<false>(h, raw_dev, nullptr, stream_stride_i16, nsamples, ic, qc, audio_dev, audio_stride_i16,
                            as_stream(stream));
}

int jsdr_demod_batch_f32(jsdr_demod *h, const float *iq_dev, int64_t stream_stride_f32, int64_t nsamples, int16_t *audio_dev,
                         int64_t audio_stride_i16, void *stream)
{
    return demod_run<true>(h, nullptr, iq_dev, stream_stride_f32, nsamples, 0, 0, audio_dev, audio_stride_i16,
                           as_stream(stream));
}

// IAudioHandler.receive(float[]) of a single-stream handle: one frame in, the frame's audio bytes out
int jsdr_demod_receive_f32(jsdr_demod *h, const float *buf_host, int16_t *audio_host)
{
    JSDR_REQUIRE(h && buf_host && audio_host, "jsdr_demod_receive_f32: null argument");
    JSDR_REQUIRE(h->nstreams == 1, "jsdr_demod_receive_f32: the frame-by-frame form needs a 1-stream handle");
    const size_t in_bytes = sizeof(float) * 2 * (size_t)h->n, out_bytes = sizeof(int) * (size_t)h->n;
    if (!h->pin_tried) {  // the first receive() of the handle (PinnedStage, common.h)
        h->pin.alloc(in_bytes + out_bytes);
        h->pin_tried = true;
    }
    if (h->pin.p) {
        memcpy(h->pin.p, buf_host, in_bytes);
        SyncOnExit guard;  // (an error exit below must not leave the device reading or writing the stage)
        JSDR_HIP_TRY(hipMemcpyAsync(h->stage_in.p, h->pin.p, in_bytes, hipMemcpyHostToDevice, 0));
        if (demod_run<true>(h, nullptr, h->stage_in.p, 2 * (int64_t)h->n, h->n, 0, 0, reinterpret_cast<int16_t *>(h->stage_out.p),
                            2 * (int64_t)h->n, 0) != JSDR_OK)
            return JSDR_ERR;
        JSDR_HIP_TRY(hipMemcpyAsync(h->pin.p + in_bytes, h->stage_out.p, out_bytes, hipMemcpyDeviceToHost, 0));
        JSDR_HIP_TRY(hipStreamSynchronize(0));
        guard.armed = false;
        memcpy(audio_host, h->pin.p + in_bytes, out_bytes);
        return JSDR_OK;
    }
    JSDR_HIP_TRY(hipMemcpy(h->stage_in.p, buf_host, in_bytes, hipMemcpyHostToDevice));
    if (demod_run<true>(h, nullptr, h->stage_in.p, 2 * (int64_t)h->n, h->n, 0, 0, reinterpret_cast<int16_t *>(h->stage_out.p),
                        2 * (int64_t)h->n, 0) != JSDR_OK)
        return JSDR_ERR;
    JSDR_HIP_TRY(hipMemcpy(audio_host, h->stage_out.p, out_bytes, hipMemcpyDeviceToHost));
    return JSDR_OK;
}

// the reference's `max` and `avg` fields as the last frame of the last call left them (:465-467)
int jsdr_demod_frame_stats(jsdr_demod *h, int stream, float *max_out, float *avg_out)
{
    JSDR_REQUIRE(h && max_out && avg_out, "jsdr_demod_frame_stats: null argument");
    JSDR_REQUIRE(stream >= 0 && stream < h->nstreams, "jsdr_demod_frame_stats: stream %d outside 0..%d", stream, h->nstreams - 1);
    JSDR_REQUIRE(h->last_nfr > 0, "jsdr_demod_frame_stats: nothing processed yet");
    float v[2];
    JSDR_HIP_TRY(hipDeviceSynchronize());
    JSDR_HIP_TRY(hipMemcpy(v, h->stats.p + 2 * ((size_t)stream * h->last_nfr + (h->last_nfr - 1)), sizeof(v), hipMemcpyDeviceToHost));
    *max_out = v[0];
    *avg_out = v[1];
    return JSDR_OK;
}

int jsdr_demod_state(jsdr_demod *h, float *car_out, float *phi_out)
{
    JSDR_REQUIRE(h, "jsdr_demod_state: null handle");
    if (car_out) *car_out = h->car;
    if (phi_out) *phi_out = h->phi;
    return JSDR_OK;
}

}  // extern "C"
