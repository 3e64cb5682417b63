// tools/microbench_lds.hip -- does an LDS instruction with few active lanes cost less than a full one?  (k_tail's nine
// chain lanes write and read 8 bytes each per link.)  20 one-wave workgroups per CU, each issuing ds_write_b64 +
// ds_read_b64 pairs with ALL lanes or with lanes 0..8 only; reports LDS instructions per CU per microsecond.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mb_lds tools/microbench_lds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(64) void k_lds(double *out, int iters, double seed)
{
    __shared__ double buf[64 * 9 + 64];
    const int lane = threadIdx.x;
    double e = seed + lane;
    const bool on = MODE == 0 ? true : (lane <= 8);
    for (int it = 0; it < iters; it++) {
        if (on) {
#pragma unroll
            for (int p = 0; p < 16; p++) {
                buf[p * 9 + (lane & 15)] = e;
                asm volatile("" ::: "memory");
                e = buf[((p + 1) & 15) * 9 + (lane & 15)] + 1.0;
            }
        }
    }
    out[blockIdx.x * 64 + lane] = e;
}
template <int MODE>
static void run(const char *name)
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * 20, iters = 4096;
    double *out;
    (void)hipMalloc(&out, sizeof(double) * blocks * 64);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k_lds<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.5);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL(k_lds<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.5);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    const double n = 20.0 * iters * 16 * 2;  // LDS instructions per CU
    printf("%-28s %.3f ms  %.1f LDS instructions per CU per us  (%.1f cycles each at 2.1 GHz)\n", name, ms, n / (ms * 1e3), ms * 1e-3 * 2.1e9 / n);
    (void)hipFree(out);
}
int main()
{
    run<0>("all 64 lanes active");
    run<1>("lanes 0..8 active");
    return 0;
}
