// fir_phase.hip -- fir.java (21-tap int FIR, NCO, int complex mixer) and phase.java reductions.
// Compiled with -ffp-contract=off: fir.filter's double accumulation must round like Java
// (separate multiply and add, fir.java:205).
#include "common.h"
#include <math.h>
#include <vector>

namespace jsdr {

// Math.sin/cos on a double argument, evaluated in long double and rounded once (independent of the host
// compiler's sin/cos -> sincos / vector-libm rewrites; same rule as the oracle)
static double jsin(double x) { return (double)sinl((long double)x); }
static double jcos(double x) { return (double)cosl((long double)x); }

// fir.java:198-211.  Output t = (int) sum_{i=0..20} x[t-i]*w[i], accumulated newest sample first,
// exactly the ring walk `ti=(fof+i)%21`.  xh = 20 history samples followed by the n new ones.
__global__ void k_fir_filter(const int *__restrict__ xh, const double *__restrict__ w, int *__restrict__ out,
                             long long n)
{
    __shared__ double ws[21];
    if (threadIdx.x < 21) ws[threadIdx.x] = w[threadIdx.x];
    __syncthreads();
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (; t < n; t += stride) {
        double o = 0;
#pragma unroll
        for (int i = 0; i < 21; i++) o = o + (double)xh[20 + t - i] * ws[i];
        out[t] = (int)o;  // v_cvt_i32_f64: truncates toward zero, saturates, NaN -> 0 == Java (int)
    }
}

// fir.java:221-228 with the trigonometry tabulated at setup: nco[k] = ((int)(cos(w_k)*4096), (int)(sin(w_k)*4096))
__global__ void k_fir_cgen(const int2 *__restrict__ nco, int period, int start, int2 *__restrict__ out, long long n)
{
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (; t < n; t += stride) out[t] = nco[(int)(((long long)start + t) % period)];
}

// fir.java:214-218: int32 complex multiply, wrapping
__global__ void k_fir_cmod(const int2 *__restrict__ a, const int2 *__restrict__ b, int2 *__restrict__ out, long long n)
{
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (; t < n; t += stride) {
        unsigned ar = (unsigned)a[t].x, ai = (unsigned)a[t].y, br = (unsigned)b[t].x, bi = (unsigned)b[t].y;
        out[t] = make_int2((int)(ar * br - ai * bi), (int)(ar * bi + ai * br));
    }
}

// phase.java:75-80: max = -1; for each float a=|x|: if (max<a) max=a.  One workgroup per frame.
__global__ void k_phase_maxabs(const float *__restrict__ iq, int len, float *__restrict__ out)
{
    const float *p = iq + (long long)blockIdx.x * len;
    float m = -1.0f;
    auto take = [&](float4 v) {
        const float a0 = fabsf(v.x), a1 = fabsf(v.y), a2 = fabsf(v.z), a3 = fabsf(v.w);
        if (m < a0) m = a0;
        if (m < a1) m = a1;
        if (m < a2) m = a2;
        if (m < a3) m = a3;
    };
    // four 16-byte loads of a thread in flight together (the whole 2048-sample frame of a 256-thread workgroup in
    // one go); the max is order independent ('<' never lets a NaN in, whichever way the frame is walked)
    const int stride = blockDim.x * 4;
    int i = threadIdx.x * 4;
    for (; i + 3 * stride + 3 < len; i += 4 * stride) {
        const float4 v0 = *reinterpret_cast<const float4 *>(p + i);
        const float4 v1 = *reinterpret_cast<const float4 *>(p + i + stride);
        const float4 v2 = *reinterpret_cast<const float4 *>(p + i + 2 * stride);
        const float4 v3 = *reinterpret_cast<const float4 *>(p + i + 3 * stride);
        take(v0);
        take(v1);
        take(v2);
        take(v3);
    }
    for (; i < len; i += stride) {
        if (i + 3 < len) {
            take(*reinterpret_cast<const float4 *>(p + i));
        } else {
            for (int j = i; j < len; j++) {
                float a = fabsf(p[j]);
                if (m < a) m = a;
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        float o = __shfl_xor(m, off, 64);
        if (m < o) m = o;
    }
    __shared__ float red[16];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (unsigned w = 1; w < (blockDim.x + 63) / 64; w++)
            if (m < red[w]) m = red[w];
        out[blockIdx.x] = m;
    }
}

// phase.java:93-116: one lane per pixel column; sequential float sums in sample order, then a float
// division by the (int) count.  Column boundaries (float `pos += step` schedule) come from the host.
__global__ void k_phase_columns(const float2 *__restrict__ iq, const int *__restrict__ first, const int *__restrict__ count,
                                int ncol, float *__restrict__ avgi, float *__restrict__ avgq)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncol) return;
    float si = 0.0f, sq = 0.0f;
    int f = first[c], n = count[c];
    for (int s = 0; s < n; s++) {
        float2 v = iq[f + s];
        si += v.x;
        sq += v.y;
    }
    avgi[c] = si / (float)n;
    avgq[c] = sq / (float)n;
}

static inline int grid_for(long long n, int block)
{
    long long g = (n + block - 1) / block;
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace jsdr

using namespace jsdr;

struct jsdr_fir {
    float rate = 44100.0f;
    double wfir[21];
    int hist[20];  // the 20 samples preceding the next input, oldest first (fir.java's ring, unrolled in time)
    DevBuf<double> w_dev;
    DevBuf<int2> nco_dev;
    int nco_period = 0;
    int nco_freq = 0x7fffffff;
};

extern "C" {

int jsdr_fir_create(jsdr_fir **out, float sample_rate)
{
    JSDR_REQUIRE(out, "jsdr_fir_create: null handle pointer");
    *out = nullptr;
    JSDR_REQUIRE(sample_rate >= 1.0f, "jsdr_fir_create: bad sample rate");
    jsdr_fir *h = new jsdr_fir();
    h->rate = sample_rate;
    for (int i = 0; i < 21; i++) h->wfir[i] = 0.0;  // fir.java:30
    for (int i = 0; i < 20; i++) h->hist[i] = 0;
    if (h->w_dev.alloc(21) != JSDR_OK) {
        delete h;
        return JSDR_ERR;
    }
    *out = h;
    return JSDR_OK;
}

int jsdr_fir_destroy(jsdr_fir *h)
{
    if (!h) return JSDR_OK;
    h->w_dev.release();
    h->nco_dev.release();
    delete h;
    return JSDR_OK;
}

// fir.java:169-195 (setup arithmetic, host): Hamming-windowed sinc band-pass, or all-pass when both
// arguments are Integer.MIN_VALUE; clears the delay line.
int jsdr_fir_weights(jsdr_fir *h, int f1, int f2, double w_out[21])
{
    JSDR_REQUIRE(h, "jsdr_fir_weights: null handle");
    const double PI = 3.14159265358979323846;
    const int len = 21;
    if (f1 == (-2147483647 - 1) && f2 == (-2147483647 - 1)) {
        for (int i = 0; i < len; i++) h->wfir[i] = 0;
        h->wfir[(len - 1) / 2] = 1;
    } else {
        double df1 = (double)f1 / h->rate;
        double df2 = (double)f2 / h->rate;
        int ord = len - 1;
        for (int n = 0; n < len; n++) {
            double v;
            if (n == ord / 2)
                v = 2 * (df2 - df1);
            else
                v = (jsin(2 * PI * df2 * (n - ord / 2)) / (PI * (n - ord / 2))) -
                    (jsin(2 * PI * df1 * (n - ord / 2)) / (PI * (n - ord / 2)));
            h->wfir[n] = v * (0.54 - 0.46 * jcos(2 * PI * n / ord));
        }
    }
    for (int i = 0; i < 20; i++) h->hist[i] = 0;
    JSDR_HIP_TRY(hipMemcpy(h->w_dev.p, h->wfir, sizeof(double) * 21, hipMemcpyHostToDevice));
    if (w_out) memcpy(w_out, h->wfir, sizeof(double) * 21);
    return JSDR_OK;
}

int jsdr_fir_filter(jsdr_fir *h, const int32_t *in_host, int32_t *out_host, int64_t n)
{
    JSDR_REQUIRE(h && in_host && out_host, "jsdr_fir_filter: null argument");
    JSDR_REQUIRE(n >= 0, "jsdr_fir_filter: negative length");
    if (n == 0) return JSDR_OK;
    DevBuf<int> xh, out;
    if (xh.alloc((size_t)n + 20) != JSDR_OK || out.alloc((size_t)n) != JSDR_OK) {
        xh.release();
        out.release();
        return JSDR_ERR;
    }
    int rc = JSDR_OK;
    do {
        if (hipMemcpy(xh.p, h->hist, sizeof(int) * 20, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(xh.p + 20, in_host, sizeof(int) * (size_t)n, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(h->w_dev.p, h->wfir, sizeof(double) * 21, hipMemcpyHostToDevice) != hipSuccess) {
            set_error("jsdr_fir_filter: upload failed");
            rc = JSDR_ERR;
            break;
        }
        hipLaunchKernelGGL(k_fir_filter, dim3(grid_for(n, 256)), dim3(256), 0, 0, xh.p, h->w_dev.p, out.p,
                           (long long)n);
        if (hipGetLastError() != hipSuccess ||
            hipMemcpy(out_host, out.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) {
            set_error("jsdr_fir_filter: kernel or download failed");
            rc = JSDR_ERR;
            break;
        }
        // carry the delay line: last 20 inputs, oldest first
        int tmp[20];
        for (int i = 0; i < 20; i++) {
            int64_t src = n - 20 + i;
            tmp[i] = src >= 0 ? in_host[src] : h->hist[20 + src];
        }
        memcpy(h->hist, tmp, sizeof(tmp));
    } while (0);
    xh.release();
    out.release();
    return rc;
}

int jsdr_fir_complex_gen(jsdr_fir *h, int freq, int start, int32_t *sig_host, int64_t n)
{
    JSDR_REQUIRE(h && sig_host, "jsdr_fir_complex_gen: null argument");
    int period = (int)h->rate;  // `if (wav[1]>=(int)fmt.getSampleRate()) wav[1]=0` (fir.java:226)
    JSDR_REQUIRE(period > 0 && start >= 0 && start < period, "jsdr_fir_complex_gen: start outside [0,rate)");
    if (n <= 0) return JSDR_OK;
    if (h->nco_freq != freq || h->nco_period != period) {
        const double PI = 3.14159265358979323846;
        std::vector<int2> tab((size_t)period);
        for (int k = 0; k < period; k++) {
            double w = (2 * PI * freq * k) / h->rate;
            double c = jcos(w) * 4096, s = jsin(w) * 4096;
            tab[k] = make_int2((int)c, (int)s);
        }
        if (h->nco_dev.alloc((size_t)period) != JSDR_OK) return JSDR_ERR;
        JSDR_HIP_TRY(hipMemcpy(h->nco_dev.p, tab.data(), sizeof(int2) * (size_t)period, hipMemcpyHostToDevice));
        h->nco_freq = freq;
        h->nco_period = period;
    }
    DevBuf<int2> out;
    if (out.alloc((size_t)n) != JSDR_OK) return JSDR_ERR;
    hipLaunchKernelGGL(k_fir_cgen, dim3(grid_for(n, 256)), dim3(256), 0, 0, h->nco_dev.p, period, start, out.p,
                       (long long)n);
    int rc = JSDR_OK;
    if (hipGetLastError() != hipSuccess ||
        hipMemcpy(sig_host, out.p, sizeof(int2) * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) {
        set_error("jsdr_fir_complex_gen: kernel or download failed");
        rc = JSDR_ERR;
    }
    out.release();
    return rc;
}

int jsdr_fir_complex_mod(jsdr_fir *h, const int32_t *a_host, const int32_t *b_host, int32_t *out_host, int64_t n)
{
    JSDR_REQUIRE(h && a_host && b_host && out_host, "jsdr_fir_complex_mod: null argument");
    if (n <= 0) return JSDR_OK;
    DevBuf<int2> a, b, o;
    int rc = JSDR_ERR;
    if (a.alloc((size_t)n) == JSDR_OK && b.alloc((size_t)n) == JSDR_OK && o.alloc((size_t)n) == JSDR_OK) {
        if (hipMemcpy(a.p, a_host, sizeof(int2) * (size_t)n, hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(b.p, b_host, sizeof(int2) * (size_t)n, hipMemcpyHostToDevice) == hipSuccess) {
            hipLaunchKernelGGL(k_fir_cmod, dim3(grid_for(n, 256)), dim3(256), 0, 0, a.p, b.p, o.p, (long long)n);
            if (hipGetLastError() == hipSuccess &&
                hipMemcpy(out_host, o.p, sizeof(int2) * (size_t)n, hipMemcpyDeviceToHost) == hipSuccess)
                rc = JSDR_OK;
        }
        if (rc != JSDR_OK) set_error("jsdr_fir_complex_mod: transfer or kernel failed");
    }
    a.release();
    b.release();
    o.release();
    return rc;
}

int jsdr_phase_maxabs(const float *iq_dev, int64_t nframes, int n, float *max_dev, void *stream)
{
    JSDR_REQUIRE(iq_dev && max_dev, "jsdr_phase_maxabs: null buffer");
    JSDR_REQUIRE(n > 0 && nframes >= 0 && nframes < 2147483647LL, "jsdr_phase_maxabs: bad geometry");
    JSDR_REQUIRE(((2 * (long long)n) % 4) == 0, "jsdr_phase_maxabs: frame length must be a multiple of 2 samples");
    if (nframes == 0) return JSDR_OK;
    hipLaunchKernelGGL(k_phase_maxabs, dim3((unsigned)nframes), dim3(256), 0, as_stream(stream), iq_dev, 2 * n,
                       max_dev);
    JSDR_LAUNCH_CHECK();
    return JSDR_OK;
}

int jsdr_phase_columns(const float *iq_dev, int n, int bx, int32_t *pix_host, float *avgi_host, float *avgq_host,
                       int cap, int *ncol_out)
{
    JSDR_REQUIRE(iq_dev && pix_host && avgi_host && avgq_host && ncol_out, "jsdr_phase_columns: null argument");
    JSDR_REQUIRE(n > 0 && bx >= 0, "jsdr_phase_columns: bad geometry");
    // phase.java:81-99: float step, float running position, column closes when (int)pos > last pixel
    const int len = 2 * n;
    float step = (float)(bx * 2) / (float)len;
    float pos = 0;
    int lpix = 0, acnt = 0, start = 0;
    std::vector<int> first, count, pix;
    for (int s = 0; s < len; s += 2) {
        acnt += 1;
        pos += step;
        int p = (int)pos;
        if (p > lpix) {
            first.push_back(start);
            count.push_back(acnt);
            pix.push_back(p);
            lpix = p;
            acnt = 0;
            start = s / 2 + 1;
        }
    }
    int ncol = (int)pix.size();
    JSDR_REQUIRE(ncol <= cap, "jsdr_phase_columns: %d columns exceed the caller's capacity %d", ncol, cap);
    *ncol_out = ncol;
    if (ncol == 0) return JSDR_OK;
    DevBuf<int> dfirst, dcount;
    DevBuf<float> dai, daq;
    int rc = JSDR_ERR;
    if (dfirst.alloc(ncol) == JSDR_OK && dcount.alloc(ncol) == JSDR_OK && dai.alloc(ncol) == JSDR_OK &&
        daq.alloc(ncol) == JSDR_OK) {
        if (hipMemcpy(dfirst.p, first.data(), sizeof(int) * ncol, hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(dcount.p, count.data(), sizeof(int) * ncol, hipMemcpyHostToDevice) == hipSuccess) {
            hipLaunchKernelGGL(k_phase_columns, dim3((ncol + 63) / 64), dim3(64), 0, 0,
                               reinterpret_cast<const float2 *>(iq_dev), dfirst.p, dcount.p, ncol, dai.p, daq.p);
            if (hipGetLastError() == hipSuccess &&
                hipMemcpy(avgi_host, dai.p, sizeof(float) * ncol, hipMemcpyDeviceToHost) == hipSuccess &&
                hipMemcpy(avgq_host, daq.p, sizeof(float) * ncol, hipMemcpyDeviceToHost) == hipSuccess)
                rc = JSDR_OK;
        }
        if (rc != JSDR_OK) set_error("jsdr_phase_columns: transfer or kernel failed");
    }
    dfirst.release();
    dcount.release();
    dai.release();
    daq.release();
    if (rc == JSDR_OK) memcpy(pix_host, pix.data(), sizeof(int) * ncol);
    return rc;
}

// ---- phase.java as a handle (the IAudioHandler drop-in: one frame in per receive(), the two reductions of
// paintComponent out): phase.java:123-128 copies the frame -- here to the device, where max|x| (:75-80) is taken at
// once; the column means (:93-116) depend on the panel width and are computed when the painter asks.
struct jsdr_phase {
    int n = 0;
    DevBuf<float> dpy;   // [2n] the frame
    DevBuf<float> dmax;  // [1]
    float max = -1.0f;   // phase.java:75: `float max = -1` before the first frame
    bool have = false;
};

int jsdr_phase_create(jsdr_phase **out, int n)
{
    JSDR_REQUIRE(out, "jsdr_phase_create: null handle pointer");
    *out = nullptr;
    JSDR_REQUIRE(n > 0 && (n & 1) == 0, "jsdr_phase_create: frame of %d samples (must be positive and even)", n);
    jsdr_phase *h = new jsdr_phase();
    h->n = n;
    if (h->dpy.alloc(2 * (size_t)n) != JSDR_OK || h->dmax.alloc(1) != JSDR_OK ||
        hipMemset(h->dpy.p, 0, sizeof(float) * 2 * (size_t)n) != hipSuccess) {  // `new float[...]` is zero-filled (:23)
        h->dpy.release();
        h->dmax.release();
        delete h;
        set_error("jsdr_phase_create: no HIP device or out of device memory");
        return JSDR_ERR;
    }
    *out = h;
    return JSDR_OK;
}

int jsdr_phase_destroy(jsdr_phase *h)
{
    if (!h) return JSDR_OK;
    h->dpy.release();
    h->dmax.release();
    delete h;
    return JSDR_OK;
}

int jsdr_phase_receive_f32(jsdr_phase *h, const float *iq_host)
{
    JSDR_REQUIRE(h && iq_host, "jsdr_phase_receive_f32: null argument");
    JSDR_HIP_TRY(hipMemcpy(h->dpy.p, iq_host, sizeof(float) * 2 * (size_t)h->n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_phase_maxabs, dim3(1), dim3(256), 0, 0, h->dpy.p, 2 * h->n, h->dmax.p);
    JSDR_LAUNCH_CHECK();
    JSDR_HIP_TRY(hipMemcpy(&h->max, h->dmax.p, sizeof(float), hipMemcpyDeviceToHost));
    h->have = true;
    return JSDR_OK;
}

int jsdr_phase_get_max(jsdr_phase *h, float *max_out)
{
    JSDR_REQUIRE(h && max_out, "jsdr_phase_get_max: null argument");
    // before the first frame the reference paints a zero-filled dpy: max = 0
    *max_out = h->have ? h->max : 0.0f;
    return JSDR_OK;
}

int jsdr_phase_get_columns(jsdr_phase *h, int bx, int32_t *pix_host, float *avgi_host, float *avgq_host, int cap, int *ncol)
{
    JSDR_REQUIRE(h, "jsdr_phase_get_columns: null handle");
    return jsdr_phase_columns(h->dpy.p, h->n, bx, pix_host, avgi_host, avgq_host, cap, ncol);
}

}  // extern "C"
