// microbench_fp64.hip -- issue-rate probes that bound the exact-order FP64 kernels (DESIGN.md roofline):
// v_mul_f64 / v_add_f64 (separately rounded, what the Java-order FIRs need), v_fma_f64, v_cvt_f64_f32,
// and an LDS-fed mul+add loop shaped like the FIR inner loops.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k_rate(double *out, int iters, double seed)
{
    double a[8], b = seed + threadIdx.x * 1e-9, c = 1.0000001;
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0) { a[i] = a[i] * c; a[i] = a[i] + b; }          // mul + add, separately rounded
            if (MODE == 1) { a[i] = __builtin_fma(a[i], c, b); }          // fma
            if (MODE == 2) { a[i] = a[i] * c; }                           // mul only
            if (MODE == 3) { a[i] = a[i] + b; }                           // add only
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_cvt(double *out, int iters, float seed)
{
    float f[8];
    double acc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) f[i] = seed + i + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            double d = (double)f[i];
            asm volatile("" : "+v"(d));
            f[i] = (float)d;
        }
    }
#pragma unroll
    for (int i = 0; i < 8; i++) acc += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// LDS-fed FIR-like loop: one ds_read_b128 (double2) per 4R FP64 ops
template <int R>
__global__ __launch_bounds__(256) void k_ldsfir(double *out, int iters)
{
    __shared__ double2 x[4224];
    for (int i = threadIdx.x; i < 4224; i += 256) x[i] = make_double2(1.0 + i * 1e-6, 2.0 - i * 1e-6);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    double ai[R], aq[R];
#pragma unroll
    for (int r = 0; r < R; r++) { ai[r] = 0; aq[r] = 0; }
    for (int it = 0; it < iters; it++) {
        const double2 *p = x + 64 + 65 * lane;
        for (int i = 0; i < 64; i++) {
            double2 v = p[-i];
#pragma unroll
            for (int r = 0; r < R; r++) {
                double t = 1.0 + (i + r) * 1e-3;
                ai[r] += v.x * t;
                aq[r] += v.y * t;
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int r = 0; r < R; r++) s += ai[r] + aq[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F>
static float time_ms(F f)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    f();
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    printf("device %s, %d CUs, clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    const int blocks = p.multiProcessorCount * 8, threads = 256, iters = 4096;
    double *out;
    CHECK(hipMalloc(&out, sizeof(double) * blocks * threads));
    const double lanes = (double)blocks * threads;
    const char *names[4] = {"mul+add (2 ops)", "fma (1 op)", "mul", "add"};
    const double ops[4] = {2, 1, 1, 1};
    float ms;
    ms = time_ms([&] { hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.5); });
    printf("%-18s %8.3f ms  %7.2f T instr-lanes/s\n", names[0], ms, lanes * iters * 8 * ops[0] / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.5); });
    printf("%-18s %8.3f ms  %7.2f T instr-lanes/s\n", names[1], ms, lanes * iters * 8 * ops[1] / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_rate<2>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.5); });
    printf("%-18s %8.3f ms  %7.2f T instr-lanes/s\n", names[2], ms, lanes * iters * 8 * ops[2] / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_rate<3>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.5); });
    printf("%-18s %8.3f ms  %7.2f T instr-lanes/s\n", names[3], ms, lanes * iters * 8 * ops[3] / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_cvt, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.5f); });
    printf("%-18s %8.3f ms  %7.2f T instr-lanes/s (cvt f32->f64 + f64->f32 pairs: 2 ops)\n", "cvt pair", ms,
           lanes * iters * 8 * 2 / ms / 1e9);
    const int b2 = p.multiProcessorCount * 2;
    ms = time_ms([&] { hipLaunchKernelGGL(k_ldsfir<1>, dim3(b2), dim3(threads), 0, 0, out, 256); });
    printf("%-18s %8.3f ms  %7.2f T instr-lanes/s (R=1: 1 ds_read_b128 per 4 ops)\n", "lds fir R=1", ms,
           (double)b2 * threads * 256 * 64 * 4 * 1 / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_ldsfir<2>, dim3(b2), dim3(threads), 0, 0, out, 256); });
    printf("%-18s %8.3f ms  %7.2f T instr-lanes/s (R=2)\n", "lds fir R=2", ms,
           (double)b2 * threads * 256 * 64 * 4 * 2 / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_ldsfir<8>, dim3(b2), dim3(threads), 0, 0, out, 256); });
    printf("%-18s %8.3f ms  %7.2f T instr-lanes/s (R=8)\n", "lds fir R=8", ms,
           (double)b2 * threads * 256 * 64 * 4 * 8 / ms / 1e9);
    hipFree(out);
    return 0;
}
