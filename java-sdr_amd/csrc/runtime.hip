// runtime.hip -- error reporting, device/memory/timer helpers of the C ABI, and the stand-alone
// int16->float conversion kernel (JavaAudio.java:276-293).
#include "common.h"
#include <stdarg.h>
#include <map>
#include <mutex>
#include <utility>

namespace jsdr {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int ensure_dynamic_lds(const void *kernel, size_t bytes)
{
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, size_t> have;
    int dev = 0;
    JSDR_HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    size_t &cur = have[std::make_pair(dev, kernel)];
    if (cur < bytes) {
        JSDR_HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        cur = bytes;
    }
    return JSDR_OK;
}

struct Timer {
    hipEvent_t a, b;
};

// one thread per sample frame: 4 B (or 2 B mono) in, 8 B out
__global__ void k_convert_i16(const int16_t *__restrict__ raw, int64_t nframes, int chns, int ic, int qc,
                              float2 *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < nframes; i += stride) {
        float2 v;
        if (chns > 1) {
            int w = reinterpret_cast<const int *>(raw)[i];
            int si = java_short_add((int)(short)(w & 0xffff), ic);
            int sq = java_short_add(w >> 16, qc);
            v.x = i16_to_float_java(si);
            v.y = i16_to_float_java(sq);
        } else {
            v.x = i16_to_float_java(java_short_add(raw[i], ic));
            v.y = 0.0f;
        }
        out[i] = v;
    }
}

}  // namespace jsdr

using namespace jsdr;

extern "C" {

const char *jsdr_last_error(void) { return g_err; }
int jsdr_version(void) { return 1; }

int jsdr_device_count(int *count)
{
    JSDR_REQUIRE(count, "jsdr_device_count: null argument");
    JSDR_HIP_TRY(hipGetDeviceCount(count));
    return JSDR_OK;
}

int jsdr_set_device(int device)
{
    JSDR_HIP_TRY(hipSetDevice(device));
    return JSDR_OK;
}

int jsdr_get_device(int *device)
{
    JSDR_REQUIRE(device, "jsdr_get_device: null argument");
    JSDR_HIP_TRY(hipGetDevice(device));
    return JSDR_OK;
}

int jsdr_device_name(char *buf, int cap)
{
    JSDR_REQUIRE(buf && cap > 0, "jsdr_device_name: bad buffer");
    int dev = 0;
    JSDR_HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t p;
    JSDR_HIP_TRY(hipGetDeviceProperties(&p, dev));
    snprintf(buf, (size_t)cap, "%s", p.gcnArchName);
    return JSDR_OK;
}

int jsdr_malloc(void **dev, size_t bytes)
{
    JSDR_REQUIRE(dev, "jsdr_malloc: null argument");
    JSDR_HIP_TRY(hipMalloc(dev, bytes ? bytes : 1));
    return JSDR_OK;
}

int jsdr_free(void *dev)
{
    if (dev) JSDR_HIP_TRY(hipFree(dev));
    return JSDR_OK;
}

int jsdr_memset(void *dev, int value, size_t bytes)
{
    JSDR_HIP_TRY(hipMemset(dev, value, bytes));
    return JSDR_OK;
}

int jsdr_memcpy_h2d(void *dev, const void *host, size_t bytes)
{
    JSDR_HIP_TRY(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice));
    return JSDR_OK;
}

int jsdr_memcpy_d2h(void *host, const void *dev, size_t bytes)
{
    JSDR_HIP_TRY(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
    return JSDR_OK;
}

int jsdr_stream_create(void **stream)
{
    JSDR_REQUIRE(stream, "jsdr_stream_create: null argument");
    hipStream_t st = nullptr;
    JSDR_HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    *stream = reinterpret_cast<void *>(st);
    return JSDR_OK;
}

int jsdr_stream_destroy(void *stream)
{
    if (stream) JSDR_HIP_TRY(hipStreamDestroy(as_stream(stream)));
    return JSDR_OK;
}

int jsdr_stream_sync(void *stream)
{
    JSDR_HIP_TRY(hipStreamSynchronize(as_stream(stream)));
    return JSDR_OK;
}

int jsdr_timer_create(void **timer)
{
    JSDR_REQUIRE(timer, "jsdr_timer_create: null argument");
    Timer *t = new Timer();
    JSDR_HIP_TRY(hipEventCreate(&t->a));
    JSDR_HIP_TRY(hipEventCreate(&t->b));
    *timer = t;
    return JSDR_OK;
}

int jsdr_timer_destroy(void *timer)
{
    Timer *t = (Timer *)timer;
    if (!t) return JSDR_OK;
    (void)hipEventDestroy(t->a);
    (void)hipEventDestroy(t->b);
    delete t;
    return JSDR_OK;
}

int jsdr_timer_start(void *timer, void *stream)
{
    JSDR_REQUIRE(timer, "jsdr_timer_start: null timer");
    JSDR_HIP_TRY(hipEventRecord(((Timer *)timer)->a, as_stream(stream)));
    return JSDR_OK;
}

int jsdr_timer_stop(void *timer, void *stream)
{
    JSDR_REQUIRE(timer, "jsdr_timer_stop: null timer");
    JSDR_HIP_TRY(hipEventRecord(((Timer *)timer)->b, as_stream(stream)));
    return JSDR_OK;
}

int jsdr_timer_elapsed_ms(void *timer, float *ms)
{
    JSDR_REQUIRE(timer && ms, "jsdr_timer_elapsed_ms: null argument");
    Timer *t = (Timer *)timer;
    JSDR_HIP_TRY(hipEventSynchronize(t->b));
    JSDR_HIP_TRY(hipEventElapsedTime(ms, t->a, t->b));
    return JSDR_OK;
}

int jsdr_convert_i16(const int16_t *raw_dev, int64_t nframes, int chns, int ic, int qc, float *iq_dev,
                     void *stream)
{
    JSDR_REQUIRE(raw_dev && iq_dev, "jsdr_convert_i16: null buffer");
    JSDR_REQUIRE(chns == 1 || chns == 2, "jsdr_convert_i16: chns must be 1 or 2 (got %d)", chns);
    JSDR_REQUIRE(nframes >= 0, "jsdr_convert_i16: negative frame count");
    if (nframes == 0) return JSDR_OK;
    int64_t blocks = (nframes + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(k_convert_i16, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), raw_dev,
                       nframes, chns, ic, qc, reinterpret_cast<float2 *>(iq_dev));
    JSDR_LAUNCH_CHECK();
    return JSDR_OK;
}

}  // extern "C"
