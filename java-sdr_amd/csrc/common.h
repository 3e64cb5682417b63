// common.h -- internal helpers shared by the libjsdr_hip.so translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include "../../include/jsdr_hip.h"

namespace jsdr {

void set_error(const char *fmt, ...);

#define JSDR_HIP_TRY(expr)                                                                         \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            ::jsdr::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,     \
                              __LINE__);                                                           \
            return JSDR_ERR;                                                                       \
        }                                                                                          \
    } while (0)

#define JSDR_REQUIRE(cond, ...)                                                                    \
    do {                                                                                           \
        if (!(cond)) {                                                                             \
            ::jsdr::set_error(__VA_ARGS__);                                                        \
            return JSDR_ERR;                                                                       \
        }                                                                                          \
    } while (0)

// launch-error check that is safe under stream capture (no sync)
#define JSDR_LAUNCH_CHECK() JSDR_HIP_TRY(hipGetLastError())

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// RAII-less device buffer (handles own them and free in destroy)
template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    int alloc(size_t count)
    {
        release();
        if (count == 0) return JSDR_OK;
        JSDR_HIP_TRY(hipMalloc((void **)&p, count * sizeof(T)));
        n = count;
        return JSDR_OK;
    }
    int zero(hipStream_t s = 0)
    {
        if (p) JSDR_HIP_TRY(hipMemsetAsync(p, 0, n * sizeof(T), s));
        return JSDR_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

// ---- int16 -> float, JavaAudio.java:281-288:  (float)s / (float)Short.MAX_VALUE -----------------
// IEEE-correct float division costs ~10 VALU ops (v_div_scale/fmas/fixup).  For the 65536 possible
// dividends and the constant divisor 32767 the quotient is q = fma(a, rh, a*rl) with rh = RN(1/d) and
// rl = RN(1/d - rh): rh+rl carries 1/d to 2^-48 and no a/32767 lies that close to a rounding boundary.
// Checked for every input in exact rational arithmetic and on the GPU against the CPU's
// division (tests/test_gpu_fft.py, all 65536 inputs).
__device__ __forceinline__ float i16_to_float_java(int s)
{
    const float rh = 0x1.0002p-15f;
    const float rl = 0x1.0002p-45f;
    const float a = (float)s;
    return __builtin_fmaf(a, rh, a * rl);
}

// `short s = getShort(); s += (short)corr;` -- 16-bit wrap-around add
__device__ __forceinline__ int java_short_add(int s, int corr)
{
    return (int)(short)(s + (int)(short)corr);
}

}  // namespace jsdr
