// microbench_lat.hip -- dependent-issue latency and ILP needs of the VALU operations the front end is made of:
// C independent chains per wave, W waves per SIMD; reports SIMD issue slots (4 cycles at the measured clock) per
// instruction.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/mb_lat tools/microbench_lat.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP, int C>
__global__ void k_chain(double *out, int iters, double seed, long long *clk)
{
    double a[C];
    float f[C];
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p[C];
#pragma unroll
    for (int i = 0; i < C; i++) {
        a[i] = seed + i + threadIdx.x * 1e-9;
        f[i] = (float)a[i];
        p[i] = v2f{f[i], f[i] + 1.0f};
    }
    const double c = seed * 1.0000001, b = seed * 1e-3;
    const v2f pc = {(float)c, (float)c};
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int i = 0; i < C; i++) {
                if (OP == 0) a[i] = a[i] + b;                        // v_add_f64
                if (OP == 1) a[i] = a[i] * c;                        // v_mul_f64
                if (OP == 2) a[i] = __builtin_fma(a[i], c, b);       // v_fma_f64
                if (OP == 3) p[i] = __builtin_elementwise_fma(p[i], pc, pc);  // v_pk_fma_f32
                if (OP == 4) f[i] = __builtin_fmaf(f[i], pc.x, pc.y);         // v_fma_f32
                if (OP == 5) { a[i] = (double)f[i]; asm volatile("" : "+v"(a[i])); f[i] = (float)a[i]; }  // cvt pair
                if (OP == 6) { a[i] = a[i] * c; a[i] = a[i] + b; }   // mul -> add dependent pair
            }
        }
    }
    long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < C; i++) s += a[i] + f[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int OP, int C>
static void run(const char *name, int waves_per_simd, int ops_per_it)
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int threads = 64 * 4 * waves_per_simd;  // one workgroup per CU: 4 SIMDs x W waves
    const int blocks = p.multiProcessorCount, iters = 2048;
    double *out;
    long long *clk;
    hipMalloc(&out, sizeof(double) * blocks * threads);
    hipMalloc(&clk, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k_chain<OP, C>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.5, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_chain<OP, C>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.5, clk);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long c;
    hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 8 * C * ops_per_it;  // instructions per wave
    // clock64 = s_memtime: 100 MHz constant on this chip -> use wall time and the nominal clock only as a guide
    printf("%-14s chains %d waves/SIMD %d : %7.3f ms  %6.2f ns per instr per wave  -> %5.2f instr per SIMD per 4 cycles @2.0 GHz  (memtime ticks %lld)\n",
           name, C, waves_per_simd, ms, ms * 1e6 / n, n * waves_per_simd / (ms * 1e-3 * 2.0e9 / 4), c);
    hipFree(out);
    hipFree(clk);
}

int main()
{
#define SWEEP(OP, NAME, OPS)                 \
    run<OP, 1>(NAME, 1, OPS);                \
    run<OP, 2>(NAME, 1, OPS);                \
    run<OP, 4>(NAME, 1, OPS);                \
    run<OP, 8>(NAME, 1, OPS);                \
    run<OP, 1>(NAME, 2, OPS);                \
    run<OP, 2>(NAME, 2, OPS);                \
    run<OP, 1>(NAME, 4, OPS);                \
    run<OP, 2>(NAME, 4, OPS);                \
    run<OP, 4>(NAME, 4, OPS);
    SWEEP(0, "v_add_f64", 1)
    SWEEP(1, "v_mul_f64", 1)
    SWEEP(2, "v_fma_f64", 1)
    SWEEP(3, "v_pk_fma_f32", 1)
    SWEEP(4, "v_fma_f32", 1)
    SWEEP(5, "cvt f32<->f64", 2)
    SWEEP(6, "mul->add f64", 2)
    return 0;
}
