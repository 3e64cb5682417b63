// bpsk_radix.h -- the fixed-order r-point butterflies of the oracle's mixed-radix transform (oracle/o_fft.c dft_r), shared by the
// LDS front ends (bpsk_fftm.hip) and the any-frame passes through global memory (bpsk_acqg.hip).  Include only from translation
// units compiled with -ffp-contract=off: every product and sum rounds by itself, as Java's do.
#pragma once
#include "common.h"

namespace jsdr {

__device__ __forceinline__ double2 cdadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 cdsub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double2 cdmul(double2 u, double2 w)
{
    return make_double2(u.x * w.x - u.y * w.y, u.x * w.y + u.y * w.x);
}

// the oracle's dft_r(), operation for operation
template <int R>
__device__ __forceinline__ void dft_r(double2 *v)
{
    if constexpr (R == 2) {
        const double2 a = cdadd(v[0], v[1]), b = cdsub(v[0], v[1]);
        v[0] = a;
        v[1] = b;
    } else if constexpr (R == 4) {
        const double2 a = cdadd(v[0], v[2]), b = cdsub(v[0], v[2]), c = cdadd(v[1], v[3]), d = cdsub(v[1], v[3]);
        v[0] = cdadd(a, c);
        v[2] = cdsub(a, c);
        v[1] = make_double2(b.x + d.y, b.y - d.x);  // b - i d
        v[3] = make_double2(b.x - d.y, b.y + d.x);  // b + i d
    } else if constexpr (R == 3) {
        const double S = 0.86602540378443864676;  // sin(2 pi/3)
        const double2 t1 = cdadd(v[1], v[2]);
        const double2 t2 = make_double2(v[0].x - 0.5 * t1.x, v[0].y - 0.5 * t1.y);
        const double2 d = cdsub(v[1], v[2]);
        const double2 t3 = make_double2(S * d.x, S * d.y);
        v[0] = cdadd(v[0], t1);
        v[1] = make_double2(t2.x + t3.y, t2.y - t3.x);
        v[2] = make_double2(t2.x - t3.y, t2.y + t3.x);
    } else if constexpr (R == 7) {
        // (round 4: n = 4410 = 2 3^2 5 7^2, the frame of a 44.1 kHz sound card)
        const double C1 = 0.62348980185873353053, C2 = -0.22252093395631440429, C3 = -0.90096886790241912624;  // cos(2 pi k/7)
        const double S1 = 0.78183148246802980871, S2 = 0.97492791218182360702, S3 = 0.43388373911755812048;   // sin(2 pi k/7)
        const double2 a1 = cdadd(v[1], v[6]), a2 = cdadd(v[2], v[5]), a3 = cdadd(v[3], v[4]);
        const double2 b1 = cdsub(v[1], v[6]), b2 = cdsub(v[2], v[5]), b3 = cdsub(v[3], v[4]);
        const double2 x0 = v[0];
        const double2 m1 = make_double2(((x0.x + C1 * a1.x) + C2 * a2.x) + C3 * a3.x, ((x0.y + C1 * a1.y) + C2 * a2.y) + C3 * a3.y);
        const double2 m2 = make_double2(((x0.x + C2 * a1.x) + C3 * a2.x) + C1 * a3.x, ((x0.y + C2 * a1.y) + C3 * a2.y) + C1 * a3.y);
        const double2 m3 = make_double2(((x0.x + C3 * a1.x) + C1 * a2.x) + C2 * a3.x, ((x0.y + C3 * a1.y) + C1 * a2.y) + C2 * a3.y);
        const double2 n1 = make_double2((S1 * b1.x + S2 * b2.x) + S3 * b3.x, (S1 * b1.y + S2 * b2.y) + S3 * b3.y);
        const double2 n2 = make_double2((S2 * b1.x - S3 * b2.x) - S1 * b3.x, (S2 * b1.y - S3 * b2.y) - S1 * b3.y);
        const double2 n3 = make_double2((S3 * b1.x - S1 * b2.x) + S2 * b3.x, (S3 * b1.y - S1 * b2.y) + S2 * b3.y);
        v[0] = make_double2(((x0.x + a1.x) + a2.x) + a3.x, ((x0.y + a1.y) + a2.y) + a3.y);
        v[1] = make_double2(m1.x + n1.y, m1.y - n1.x);
        v[6] = make_double2(m1.x - n1.y, m1.y + n1.x);
        v[2] = make_double2(m2.x + n2.y, m2.y - n2.x);
        v[5] = make_double2(m2.x - n2.y, m2.y + n2.x);
        v[3] = make_double2(m3.x + n3.y, m3.y - n3.x);
        v[4] = make_double2(m3.x - n3.y, m3.y + n3.x);
    } else {
        static_assert(R == 5, "radices 2, 3, 4, 5, 7");
        const double C1 = 0.30901699437494742410, C2 = -0.80901699437494742410;  // cos(2 pi/5), cos(4 pi/5)
        const double S1 = 0.95105651629515357212, S2 = 0.58778525229247312917;   // sin(2 pi/5), sin(4 pi/5)
        const double2 a1 = cdadd(v[1], v[4]), a2 = cdadd(v[2], v[3]), b1 = cdsub(v[1], v[4]), b2 = cdsub(v[2], v[3]);
        const double2 x0 = v[0];
        const double2 m1 = make_double2((x0.x + C1 * a1.x) + C2 * a2.x, (x0.y + C1 * a1.y) + C2 * a2.y);
        const double2 m2 = make_double2((x0.x + C2 * a1.x) + C1 * a2.x, (x0.y + C2 * a1.y) + C1 * a2.y);
        const double2 n1 = make_double2(S1 * b1.x + S2 * b2.x, S1 * b1.y + S2 * b2.y);
        const double2 n2 = make_double2(S2 * b1.x - S1 * b2.x, S2 * b1.y - S1 * b2.y);
        v[0] = make_double2((x0.x + a1.x) + a2.x, (x0.y + a1.y) + a2.y);
        v[1] = make_double2(m1.x + n1.y, m1.y - n1.x);
        v[4] = make_double2(m1.x - n1.y, m1.y + n1.x);
        v[2] = make_double2(m2.x + n2.y, m2.y - n2.x);
        v[3] = make_double2(m2.x - n2.y, m2.y + n2.x);
    }
}

}  // namespace jsdr
