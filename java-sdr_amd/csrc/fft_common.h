// fft_common.h -- device helpers shared by the power-of-two (fft_psd.hip) and mixed-radix (fft_mixed.hip)
// spectrum kernels: compile-time twiddles, in-register radix-2^k DFT, LDS padding.
#pragma once
#include "common.h"
#include <utility>

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

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 w)
{
    return make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x);
}

// d * exp(-2 pi i J/LEN) with the trivial cases folded
template <int LEN, int J>
__device__ __forceinline__ float2 mul_w(float2 d)
{
    if constexpr (J == 0) {
        return d;
    } else if constexpr (4 * J == LEN) {
        return make_float2(d.y, -d.x);
    } else if constexpr (8 * J == LEN) {
        constexpr float c = (float)0.70710678118654752440;
        return make_float2((d.x + d.y) * c, (d.y - d.x) * c);
    } else if constexpr (8 * J == 3 * LEN) {
        constexpr float c = (float)0.70710678118654752440;
        return make_float2((d.y - d.x) * c, -(d.x + d.y) * c);
    } else {
        constexpr float c = (float)cx_cos_turn(J, LEN);
        constexpr float s = (float)cx_sin_turn(J, LEN);
        return make_float2(d.x * c + d.y * s, d.y * c - d.x * s);
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
};
bool mixed_plan(int n, MixedPlan &p);
void mixed_twiddles(const MixedPlan &p, float2 *out);
int mixed_launch(const MixedPlan &p, const FftArgs &a, int in_kind, int out_kind, int grid, hipStream_t st);

}  // namespace jsdr
