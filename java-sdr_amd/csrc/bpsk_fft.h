// bpsk_fft.h -- interface of the FFT-acquire front end (bpsk_fft.hip) used by the pipeline host code (bpsk.hip)
#pragma once
#include "common.h"
#include <vector>

namespace jsdr {

struct FftFrontState {
    double avePeakPower;   // :403
    double aveCentreBin;   // :404
    int centreBin;         // :405
    int pad;
    double hist[26];       // the 26 values fed to RxDownSample before the next frame (re of the inverse FFT)
};

struct FftFrontArgs {
    const int *raw;            // int16 pairs, [S][stride]
    const float2 *rawf;        // or float frames
    long long stride_pairs;
    int nframes;               // frames in this call
    int n, logn;               // samples per frame
    int ic, qc;
    int do_up;
    int decim;
    int first_out;             // input index (within the call) that completes output 0
    const double2 *vco_cs;     // [nds] (cos, sin) of the VCO phase table entry of every decimated sample
    const double2 *tw;         // [n-1] per-stage (cos, -sin), fft_twiddles_f64
    FftFrontState *st;         // [S]
    double2 *dm;               // [S][dm_stride]
    long long dm_stride;
    long long nds;
    const double *ds_taps;     // [27]
    long long *phase_clk;      // diagnostics: [8] cycle counts per phase of stream 0's workgroup, or null
};

// dsFilter (FUNcubeBPSKDemod.java:27-55, symmetric: first 14 of 27), device-side copy with a static initialiser:
// indexed with compile-time constants in the front-end kernels, so the taps fold into the instruction stream (14
// distinct values) instead of occupying 54 SGPRs or an LDS read per multiply
static __constant__ const float kDsHalf[14] = {-6.103515625000e-004F, -1.220703125000e-004F, +2.380371093750e-003F,
                                        +6.164550781250e-003F, +7.324218750000e-003F, +7.629394531250e-004F,
                                        -1.464843750000e-002F, -3.112792968750e-002F, -3.225708007813e-002F,
                                        -1.617431640625e-003F, +6.463623046875e-002F, +1.502380371094e-001F,
                                        +2.231445312500e-001F, +2.518310546875e-001F};
__device__ __forceinline__ double ds_tap(int n) { return (double)kDsHalf[n < 14 ? n : 26 - n]; }

// Two adjacent 100-wide boxcar sums from the same 51 aligned 16-byte reads w[0..50] (w[0].x = P[i-50]):
//   a0 = P[i-50] + ... + P[i+49],  a1 = P[i-49] + ... + P[i+50], each in ascending order (:433-437).
// The window comes in chunks of 8 reads, the next chunk in flight while the current one is summed: as a plain loop
// the compiler waits for every read before its four additions (fifty LDS latencies per pair); left alone with the
// unrolled loop it hoists all 51 reads (204 registers, spilled).  Chunk C is a template parameter so that every
// register index is a compile-time constant.
template <int C>
__device__ __forceinline__ void boxcar_chunk(const double2 *w, const double2 (&cur)[8], double &a0, double &a1, double &prev_y)
{
    constexpr int NCH = 7;  // 7 chunks of 8 cover w[0..50]
    double2 nxt[8];
    if constexpr (C + 1 < NCH) {
#pragma unroll
        for (int u = 0; u < 8; u++)
            if ((C + 1) * 8 + u <= 50) nxt[u] = w[(C + 1) * 8 + u];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const int k = C * 8 + u;  // element w[k]
        if (k <= 50) {
            const double2 e = cur[u];
            if (k >= 1) {
                a0 += prev_y;
                a1 += prev_y;
            }
            if (k < 50) a0 += e.x;
            if (k >= 1) a1 += e.x;
            prev_y = e.y;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (C + 1 < NCH) boxcar_chunk<C + 1>(w, nxt, a0, a1, prev_y);
}

__device__ __forceinline__ void boxcar_pair(const double2 *w, double &a0, double &a1)
{
    double2 first[8];
#pragma unroll
    for (int u = 0; u < 8; u++) first[u] = w[u];
    double prev_y = 0.0;
    a0 = 0.0;
    a1 = 0.0;
    boxcar_chunk<0>(w, first, a0, a1, prev_y);
}

// The wave's FIRST MAXIMUM of the boxcar search (:439-442): every lane brings its own best (value > 0 at index >= 0, or 0.0 at -1);
// out comes, in every lane, the largest value and the LOWEST index that holds it (or 0.0 / -1).  DPP row shifts and row broadcasts
// (an instruction each) instead of __shfl_xor's round trips through the LDS crossbar: a 64-lane reduction of (double, int) by
// shuffles is 18 dependent ds_bpermute, ~2000 cycles with nothing to cover them -- twice a frame in the mixed-radix front ends.
__device__ __forceinline__ void wave_first_max(double &bestv, int &besti)
{
    double v = bestv;
#define JSDR_DPP_MAX(ctrl, rmask)                                                                                \
    {                                                                                                             \
        const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), ctrl, rmask, 0xf, false); \
        const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), ctrl, rmask, 0xf, false); \
        v = fmax(v, __hiloint2double(hi, lo));                                                                    \
    }
    JSDR_DPP_MAX(0x111, 0xf)  // row_shr:1
    JSDR_DPP_MAX(0x112, 0xf)  // row_shr:2
    JSDR_DPP_MAX(0x114, 0xf)  // row_shr:4
    JSDR_DPP_MAX(0x118, 0xf)  // row_shr:8 -- lane 15 of every row holds its row's maximum
    JSDR_DPP_MAX(0x142, 0xa)  // row_bcast:15 into rows 1 and 3
    JSDR_DPP_MAX(0x143, 0xc)  // row_bcast:31 into rows 2 and 3 -- lane 63 holds the wave's
#undef JSDR_DPP_MAX
    const double mv = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
    // the lowest index among the lanes that hold it (a lane's index ascends with its rounds, not with the lane: no "first lane" here)
    int c = (besti >= 0 && bestv == mv) ? besti : 0x7fffffff;
#define JSDR_DPP_MIN(ctrl, rmask)                                                    \
    {                                                                                 \
        const int o = __builtin_amdgcn_update_dpp(c, c, ctrl, rmask, 0xf, false);    \
        c = o < c ? o : c;                                                            \
    }
    JSDR_DPP_MIN(0x111, 0xf)
    JSDR_DPP_MIN(0x112, 0xf)
    JSDR_DPP_MIN(0x114, 0xf)
    JSDR_DPP_MIN(0x118, 0xf)
    JSDR_DPP_MIN(0x142, 0xa)
    JSDR_DPP_MIN(0x143, 0xc)
#undef JSDR_DPP_MIN
    const int mi = __builtin_amdgcn_readlane(c, 63);
    bestv = mi == 0x7fffffff ? 0.0 : mv;
    besti = mi == 0x7fffffff ? -1 : mi;
}

int launch_front_fft(const FftFrontArgs &a, int nstreams, hipStream_t st);
// round 6 (bpsk_acq.hip): the same front end in three phases -- forward transform + boxcar per frame, one scan per stream,
// inverse transform + RxDownSample per frame -- for calls of two or more frames per stream; frames of 1024 .. 8192 samples
bool acq3_supported(int n);
size_t acq3_frame_bytes(int n, int do_up);  // scratch per (stream, frame) of one launch
// what the phases hand each other, per frame id g = s * F + f (F = frames per stream in this launch)
struct AcqPeak {
    double maxBin;
    int binPos;
    int pad;
};

struct AcqArgs {
    const int *raw;        // int16 pairs [S][stride]
    const float2 *rawf;    // or float frames
    long long stride_pairs;
    int ic, qc;
    int S, F, f0;          // streams, frames per stream in this launch, index of its first frame within the call
    int n, do_up, decim;
    long long first_out, nds;
    const double2 *vco_cs;
    const double2 *tw;
    FftFrontState *st;
    double2 *dm;
    long long dm_stride;
    double2 *spec;         // [S F][nsb]: do_up ? bins [0, 204) then [n/4 - 26, n/2 + 28) : bins [0, max(n/4 + 28, 204))
    int nsb;
    double *aband;         // [S F][na]: boxcar sums over [beg + 75, end - 75)
    int na;
    AcqPeak *peak;         // [S F]
    int *cbin;             // [S F] the frame's centre bin (phase B)
    double *edges;         // [S F][52]: the frame's first 26 and last 26 real samples re / n (phase C)
    int nwg;               // persistent workgroups of phases A and C
    unsigned *tickets;     // [2] run counters of k_acq_fwd / k_acq_inv, zero at launch: a workgroup takes its frames in runs of `run`
    int run;               // consecutive frames per ticket (>= 2)
    int rps;               // runs per stream = ceil(F / run): a run lies inside one stream
    long long *clk;        // diagnostics (JSDR_FFT_PHASECLK=1): [16] clock ticks per phase of workgroup 0, k_acq_fwd [0..7], k_acq_inv [8..15]; or null
};

// thread 0 of workgroup 0 accumulates the clock ticks of every phase in LDS (never in the product's default path: clk is null)
#define ACQ_PHASE(k)                                 \
    if (timing) {                                    \
        const long long now_ = (long long)clock64(); \
        clkL[k] += now_ - tprev;                     \
        tprev = now_;                                \
    }

// where bin b of a frame sits in its spec row, or -1
__device__ __forceinline__ int acq_spec_index(int b, int n, int do_up)
{
    if (!do_up) return b < n / 4 + 28 ? b : -1;
    if (b < 204) return b;
    const int lo = n / 4 - 26;
    return (b >= lo && b < n / 2 + 28) ? 204 + (b - lo) : -1;
}

// the default mixed-radix frames (9600 / 4800 / 4410) in the same three phases: bpsk_fftm.hip has the two frame-parallel kernels
// (which 0: k_acqm_fwd, 1: k_acqm_inv), built from k_front_fftm's passes
bool acqm_supported(int n);
int launch_acqm(const AcqArgs &a, const FftFrontArgs &fa, int np, const int *rad, const int *tw_off, const int *wr_off, int num_cu,
                int which, hipStream_t st);
struct AcqmPlan {  // the handle's mixed-radix plan (fftm_twiddles), for launch_acq3
    int np = 0;
    const int *rad = nullptr, *tw_off = nullptr, *wr_off = nullptr;
};
// per-phase timing hook (bench.py's per-kernel figures): phase 0..3 = k_acq_fwd, k_acq_scan, k_acq_inv, k_acq_edges
struct AcqProf {
    void *ctx = nullptr;
    void (*mark)(void *ctx, int phase, bool begin, hipStream_t st) = nullptr;
};
// ANY other frame (bpsk_acqg.hip): phases A and C with the frame's image in global memory, one launch per pass of the oracle's
// transform -- powers of two outside 1024 .. 8192, frames above 9600 samples that are not twice a 16 | m, 2^a 3^b 5^c one
// (17640, 38400), and the LDS front ends' frames at the decimations they do not take
constexpr int ACQG_MAXPASS = 32;
struct AcqgPlan {
    bool on = false;
    int logn = 0;                // a power of two: the radix-2 network on fft_twiddles_f64's table; else the Stockham plan below
    int np = 0;
    int rad[ACQG_MAXPASS] = {0};
    int tw_off[ACQG_MAXPASS] = {0};  // pass p's table T[m] = exp(-2 pi i m / (P r))
    int wr_off[ACQG_MAXPASS] = {0};  // a prime radix above 7: W[m] = exp(-2 pi i m / r)
};
bool acqg_supported(int n);
size_t acqg_image_bytes(int n);  // scratch per (stream, frame) of one launch, on top of acq3_frame_bytes
void acqg_twiddles(std::vector<double2> &w, int n, AcqgPlan *plan);
int launch_acqg(const AcqArgs &a, const AcqgPlan &pl, double2 *img, int which, hipStream_t st);
int launch_acq3(const FftFrontArgs &a, int nstreams, unsigned char *scratch, size_t scratch_bytes, int chunk_frames, int num_cu,
                hipStream_t st, const AcqProf &prof, const AcqmPlan &plan, const AcqgPlan *gen = nullptr);
extern int g_acq_last_grid[4];  // workgroups of the last k_acq_fwd / k_acq_inv launch, and how many of each a CU holds
// frames that are not a power of two (bpsk_fftm.hip): any n with 416 <= n <= 9600 (2^a 3^b 5^c 7^d through the radix passes, any
// other prime factor through a pass that is the DFT's definition and needs fftm_scratch(n) elements of scratch per stream)
bool fftm_supported(int n);
bool fftm_pairs(int n, int nframes);  // launch_front_fftm takes the call's frames two at a time (k_front_fftm2)
size_t fftm_scratch(int n);
void fftm_twiddles(std::vector<double2> &w, int n, int *np_out, int *rad, int *tw_off, int *wr_off);
int launch_front_fftm(const FftFrontArgs &a, int np, const int *rad, const int *tw_off, const int *wr_off, double2 *gscratch,
                      int nstreams, hipStream_t st);
void fft_twiddles_f64(std::vector<double2> &w, int n);
// frames of 2 m samples, m an LDS-sized mixed-radix frame (n = 19200 at 192 kHz): two m-point halves per transform
bool fft2x_supported(int n);
void fft2x_twiddles(std::vector<double2> &w, int n, int *np_out, int *rad, int *tw_off, int *tw1_off);
size_t fft2x_scratch_ek(int n);
size_t fft2x_scratch_r0(int n);
int launch_front_fft2x(const FftFrontArgs &a, int np, const int *rad, const int *tw_off, const int *tw1_off, double2 *ek,
                       double *r0, int nstreams, hipStream_t st);

}  // namespace jsdr
