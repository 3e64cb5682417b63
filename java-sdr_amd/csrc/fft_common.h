// fft_common.h -- device helpers shared by the power-of-two (fft_psd.hip) and mixed-radix (fft_mixed.hip)
// spectrum kernels: compile-time twiddles, in-register radix-2^k DFT, LDS padding.
#pragma once
#include "common.h"
#include <utility>
#include <vector>

namespace jsdr {

// ------------------------------------------------------------------ compile-time twiddles
constexpr double cx_pi = 3.14159265358979323846264338327950288;

constexpr double cx_sin_small(double x)  // |x| <= pi/4
{
    double x2 = x * x, term = x, sum = x;
    for (int i = 1; i < 12; i++) {
        term *= -x2 / ((2 * i) * (2 * i + 1));
        sum += term;
    }
    return sum;
}
constexpr double cx_cos_small(double x)
{
    double x2 = x * x, term = 1.0, sum = 1.0;
    for (int i = 1; i < 12; i++) {
        term *= -x2 / ((2 * i - 1) * (2 * i));
        sum += term;
    }
    return sum;
}
// cos/sin of 2*pi*j/len for 0 <= j < len, exact symmetries first
constexpr double cx_cos_turn(int j, int len)
{
    j %= len;
    if (8 * j <= len) return cx_cos_small(2 * cx_pi * j / len);
    if (8 * j <= 3 * len) return -cx_sin_small(2 * cx_pi * (j - 0.25 * len) / len);
    if (8 * j <= 5 * len) return -cx_cos_small(2 * cx_pi * (j - 0.5 * len) / len);
    if (8 * j <= 7 * len) return cx_sin_small(2 * cx_pi * (j - 0.75 * len) / len);
    return cx_cos_small(2 * cx_pi * (j - len) / len);
}
constexpr double cx_sin_turn(int j, int len) { return cx_cos_turn(4 * j + 3 * len, 4 * len); }

// Complex arithmetic on the native two-float vector type: the IR then holds <2 x float> operations from the
// start, i.e. v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32.  Two rules keep register shuffles out (measured on
// the ISA): a half swap must be a shufflevector (it folds into op_sel), and a sign on ONE half must sit in a
// constant pair such as (1,-1) (a whole-vector negation folds into neg_lo/neg_hi, a partial one becomes
// v_xor + v_mov).
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f to_v(float2 a) { return (v2f){a.x, a.y}; }
__device__ __forceinline__ float2 from_v(v2f v) { return make_float2(v.x, v.y); }
__device__ __forceinline__ v2f vswap(v2f v) { return __builtin_shufflevector(v, v, 1, 0); }
__device__ __forceinline__ v2f vfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return from_v(to_v(a) + to_v(b)); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return from_v(to_v(a) - to_v(b)); }
// a * w with w' = (-w.y, w.x) supplied: (a.x w.x - a.y w.y, a.x w.y + a.y w.x) = a.x * w + a.y * w'
__device__ __forceinline__ float2 cmul2(float2 a, float2 w, float2 wq)
{
    const v2f va = to_v(a);
    return from_v(vfma(__builtin_shufflevector(va, va, 1, 1), to_v(wq), __builtin_shufflevector(va, va, 0, 0) * to_v(w)));
}
// w' = (-w.y, w.x) = swap(w) * (-1, 1)
__device__ __forceinline__ float2 cquad(float2 w) { return from_v(vswap(to_v(w)) * (v2f){-1.0f, 1.0f}); }
__device__ __forceinline__ float2 cmul(float2 a, float2 w) { return cmul2(a, w, cquad(w)); }

// d * exp(-2 pi i J/LEN) with the trivial cases folded
template <int LEN, int J>
__device__ __forceinline__ float2 mul_w(float2 d)
{
    const v2f vd = to_v(d);
    if constexpr (J == 0) {
        return d;
    } else if constexpr (4 * J == LEN) {  // (d.y, -d.x)
        return from_v(vswap(vd) * (v2f){1.0f, -1.0f});
    } else if constexpr (8 * J == LEN) {  // ((d.x + d.y) c, (d.y - d.x) c)
        constexpr float c = (float)0.70710678118654752440;
        return from_v(vfma(vswap(vd), (v2f){c, -c}, vd * c));
    } else if constexpr (8 * J == 3 * LEN) {  // ((d.y - d.x) c, -(d.x + d.y) c)
        constexpr float c = (float)0.70710678118654752440;
        return from_v(vfma(vswap(vd), (v2f){c, -c}, vd * -c));
    } else {  // (d.x c + d.y s, d.y c - d.x s)
        constexpr float c = (float)cx_cos_turn(J, LEN);
        constexpr float s = (float)cx_sin_turn(J, LEN);
        return from_v(vfma(vswap(vd), (v2f){s, -s}, vd * c));
    }
}

template <int LEN, int BASE, int J>
__device__ __forceinline__ void bfly(float2 *x)
{
    constexpr int H = LEN / 2;
    float2 a = x[BASE + J], b = x[BASE + J + H];
    x[BASE + J] = cadd(a, b);
    x[BASE + J + H] = mul_w<LEN, J>(csub(a, b));
}
template <int LEN, int BASE, int... Js>
__device__ __forceinline__ void bfly_group(float2 *x, std::integer_sequence<int, Js...>)
{
    (bfly<LEN, BASE, Js>(x), ...);
}
template <int R, int LEN, int... Bs>
__device__ __forceinline__ void bfly_stage(float2 *x, std::integer_sequence<int, Bs...>)
{
    (bfly_group<LEN, Bs * LEN>(x, std::make_integer_sequence<int, LEN / 2>{}), ...);
}
// in-register radix-R DFT, decimation in frequency: natural order in, BIT-REVERSED order out
template <int R, int LEN = R>
__device__ __forceinline__ void dft_reg(float2 *x)
{
    if constexpr (LEN >= 2) {
        bfly_stage<R, LEN>(x, std::make_integer_sequence<int, R / LEN>{});
        dft_reg<R, LEN / 2>(x);
    }
}
constexpr int cx_bitrev(int v, int r)
{
    int o = 0;
    for (int b = 1; b < r; b <<= 1) {
        o = (o << 1) | (v & 1);
        v >>= 1;
    }
    return o;
}

__device__ __forceinline__ int lds_pad(int idx) { return idx + (idx >> 4); }
constexpr int lds_frame_elems(int n) { return n + (n >> 4) + 1; }


enum { IN_I16 = 0, IN_F32 = 1 };
enum { OUT_PSD = 0, OUT_SPEC = 1 };

struct FftArgs {
    const void *in;      // int16 pairs or float pairs, [nframes][n]
    float *out;          // psd [nframes][n+2] or spectrum [nframes][2n]
    const float2 *tw;    // per-pass twiddle tables, concatenated
    long long nframes;
    int rate;
    int ic, qc;
    int split = 0;  // 1: frame f of the kernel is the even (f&1 == 0) / odd half of input frame f>>1 (2N samples)
};

struct Best {
    float v;
    int k;
};

// mixed-radix path (fft_mixed.hip): n = 4800, 9600 -- java-sdr's default 48 kHz / 96 kHz frames
struct MixedPlan {
    int n = 0, nrad = 0, radix[6] = {1, 1, 1, 1, 1, 1};
    int threads = 0;
    size_t lds_bytes = 0;
    int tw_count = 0;
    // n = 2*half (19200 = 2*9600): two half-size transforms of the even / odd samples + one radix-2 combine pass
    bool split2 = false;
    int half_tw_count = 0;  // the half plan's tables come first; `half` combine twiddles follow
};
bool mixed_plan(int n, MixedPlan &p);
void mixed_twiddles(const MixedPlan &p, float2 *out);
int mixed_launch(const MixedPlan &p, const FftArgs &a, int in_kind, int out_kind, int grid, hipStream_t st);
// split2 plans: half-size spectra of `nframes` frames into tmp [2*nframes][n/2], then the combine pass
int mixed_launch_split2(const MixedPlan &p, const FftArgs &a, int in_kind, int out_kind, int num_cu, hipStream_t st);

// any other composite n up to 9800 (fft_rt.hip): Stockham passes with a run-time radix plan (2^a 3^b 5^c 7^d in registers,
// a larger prime factor as a pass of that radix by the DFT's definition)
bool rt_supported(int n);
void rt_twiddles(int n, std::vector<float2> &w);
int rt_launch(const FftArgs &a, int n, int in_kind, int out_kind, int num_cu, hipStream_t st);

// what is left -- prime frame sizes, frames above 9800 samples (fft_any.hip): the DFT itself, double accumulation
bool dft_any_supported(int n);
int dft_any_launch(const FftArgs &a, int n, int in_kind, int out_kind, int num_cu, hipStream_t st);

}  // namespace jsdr
