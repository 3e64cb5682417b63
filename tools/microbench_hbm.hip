// tools/microbench_hbm.hip -- what HBM delivers on this box for k_fft's traffic shape: 4 B read + 4 B written per
// sample, 4 GiB each way.  (a) plain grid-stride dword copy with a convert, (b) the access pattern of k_fft
// (256-thread workgroups, 2 frames of 2048 samples each, 16 strided dwords per thread in flight, LDS-free).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_copy(const int *__restrict__ in, float *__restrict__ out, long long n)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = (float)in[i];
}
template <int UNROLL>
__global__ __launch_bounds__(256) void k_copy_u(const int *__restrict__ in, float *__restrict__ out, long long n)
{
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        int v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = in[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) out[i + u * stride] = (float)v[u];
    }
}
__global__ __launch_bounds__(256, 4) void k_fftshape(const int *__restrict__ in, float *__restrict__ out, long long nframes)
{
    const int fib = threadIdx.x >> 7, tid = threadIdx.x & 127;
    const long long ngroups = nframes / 2;
    for (long long g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const long long frame = 2 * g + fib;
        const int *src = in + frame * 2048;
        float *dst = out + frame * 2050;
        int v[16];
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = src[tid + r * 128];
#pragma unroll
        for (int r = 0; r < 8; r++) dst[tid + r * 256] = (float)v[r];
#pragma unroll
        for (int r = 0; r < 8; r++) dst[128 + tid + r * 256] = (float)v[8 + r];
    }
}
// the same traffic with wide accesses: 16-byte loads (four per thread and frame), 8-byte stores (frames are 2050 floats
// apart: 8-byte aligned only) or 16-byte stores (misaligned on odd frames)
template <int SW>
__global__ __launch_bounds__(256, 4) void k_fftshape_wide(const int *__restrict__ in, float *__restrict__ out, long long nframes)
{
    const int fib = threadIdx.x >> 7, tid = threadIdx.x & 127;
    const long long ngroups = nframes / 2;
    for (long long g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const long long frame = 2 * g + fib;
        const int4 *src = reinterpret_cast<const int4 *>(in + frame * 2048);
        float *dst = out + frame * 2050;
        int4 v[4];
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = src[tid + r * 128];
        if constexpr (SW == 2) {
            float2 *d2 = reinterpret_cast<float2 *>(dst);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                d2[2 * (tid + r * 128)] = make_float2((float)v[r].x, (float)v[r].y);
                d2[2 * (tid + r * 128) + 1] = make_float2((float)v[r].z, (float)v[r].w);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float4 o = make_float4((float)v[r].x, (float)v[r].y, (float)v[r].z, (float)v[r].w);
                __builtin_memcpy(dst + 4 * (tid + r * 128), &o, 16);
            }
        }
    }
}
// dword loads, wide stores / wide loads, dword stores: which side matters
__global__ __launch_bounds__(256, 4) void k_fftshape_wl(const int *__restrict__ in, float *__restrict__ out, long long nframes)
{
    const int fib = threadIdx.x >> 7, tid = threadIdx.x & 127;
    const long long ngroups = nframes / 2;
    for (long long g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const long long frame = 2 * g + fib;
        const int4 *src = reinterpret_cast<const int4 *>(in + frame * 2048);
        float *dst = out + frame * 2050;
        int4 v[4];
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = src[tid + r * 128];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            dst[tid + (4 * r) * 128] = (float)v[r].x;
            dst[tid + (4 * r + 1) * 128] = (float)v[r].y;
            dst[tid + (4 * r + 2) * 128] = (float)v[r].z;
            dst[tid + (4 * r + 3) * 128] = (float)v[r].w;
        }
    }
}
int main()
{
    const long long n = 1ll << 30;
    int *in;
    float *out;
    CK(hipMalloc(&in, n * 4));
    CK(hipMalloc(&out, n * 4 + (n / 2048) * 8));
    CK(hipMemset(in, 1, n * 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    auto time = [&](const char *name, auto launch) {
        for (int i = 0; i < 3; i++) launch();
        hipEventRecord(a);
        for (int i = 0; i < 10; i++) launch();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        ms /= 10;
        printf("%-28s %.3f ms  %.2f TB/s (read+write)\n", name, ms, 8.0 * n / ms / 1e9);
    };
    for (int grid : {2048, 4096, 8192, 16384})
        time(("copy grid " + std::to_string(grid)).c_str(), [&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, in, out, n); });
    time("copy x4 in flight, 4096", [&] { hipLaunchKernelGGL(k_copy_u<4>, dim3(4096), dim3(256), 0, 0, in, out, n); });
    time("copy x16 in flight, 4096", [&] { hipLaunchKernelGGL(k_copy_u<16>, dim3(4096), dim3(256), 0, 0, in, out, n); });
    time("k_fft shape, grid 4096", [&] { hipLaunchKernelGGL(k_fftshape, dim3(4096), dim3(256), 0, 0, in, out, n / 2048); });
    time("k_fft shape, grid 262144", [&] { hipLaunchKernelGGL(k_fftshape, dim3(262144), dim3(256), 0, 0, in, out, n / 2048); });
    time("k_fft shape x4 ld, x2 st, 262144", [&] { hipLaunchKernelGGL(k_fftshape_wide<2>, dim3(262144), dim3(256), 0, 0, in, out, n / 2048); });
    time("k_fft shape x4 ld, x4 st, 262144", [&] { hipLaunchKernelGGL(k_fftshape_wide<4>, dim3(262144), dim3(256), 0, 0, in, out, n / 2048); });
    time("k_fft shape x4 ld, x1 st, 262144", [&] { hipLaunchKernelGGL(k_fftshape_wl, dim3(262144), dim3(256), 0, 0, in, out, n / 2048); });
    time("k_fft shape x4 ld, x2 st, 4096", [&] { hipLaunchKernelGGL(k_fftshape_wide<2>, dim3(4096), dim3(256), 0, 0, in, out, n / 2048); });
    time("k_fft shape x1 ld, x1 st, 16384", [&] { hipLaunchKernelGGL(k_fftshape, dim3(16384), dim3(256), 0, 0, in, out, n / 2048); });
    time("k_fft shape x4 ld, x2 st, 16384", [&] { hipLaunchKernelGGL(k_fftshape_wide<2>, dim3(16384), dim3(256), 0, 0, in, out, n / 2048); });
    return 0;
}
