// tools/microbench_sparse.hip -- what a SPARSE read costs on this box: 16 bytes out of every 128 (the differential detector's
// one sample per bit period out of a sample-major (fi,fq) array) against the full read, same footprint (13.7 GB).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mb_sparse tools/microbench_sparse.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
// wave = 8 streams x 8 positions (k_tail8's mapping); lane (s8,c) reads element 8p+c of its stream; SPARSE: only c == v loads
template <int MODE>  // 0: all lanes 16 B; 1: one lane of eight 16 B; 2: all lanes 8 B (an energy array of the same shape)
__global__ __launch_bounds__(64) void k_read8(const int4 *__restrict__ in, long long quads_per_stream, int *out)
{
    const int lane = threadIdx.x, s8 = lane >> 3, c = lane & 7;
    const long long s = (long long)blockIdx.x * 8 + s8;
    int acc = 0;
    if (MODE == 2) {
        const int2 *p = reinterpret_cast<const int2 *>(in) + s * quads_per_stream + c;
        for (long long q = 0; q + 8 * 16 <= quads_per_stream; q += 8 * 16) {
            int2 v[16];
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = p[q + 8 * k];
#pragma unroll
            for (int k = 0; k < 16; k++) acc ^= v[k].x ^ v[k].y;
        }
    } else {
        const int4 *p = in + s * quads_per_stream + c;
        const bool on = MODE == 0 || c == 3;
        for (long long q = 0; q + 8 * 16 <= quads_per_stream; q += 8 * 16) {
            int4 v[16];
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = on ? p[q + 8 * k] : make_int4(0, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 16; k++) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
        }
    }
    if (acc == 0x12345678) out[threadIdx.x] = acc;
}
int main()
{
    const long long streams = 8192, qps = 104858;
    const long long nq = streams * qps;
    int4 *in;
    int *out;
    CK(hipMalloc(&in, nq * 16));
    CK(hipMalloc(&out, 4096));
    CK(hipMemset(in, 1, nq * 16));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    auto time = [&](const char *name, double bytes, auto launch) {
        for (int i = 0; i < 2; i++) launch();
        hipEventRecord(a);
        for (int i = 0; i < 5; i++) launch();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        ms /= 5;
        printf("%-52s %.3f ms  %.2f TB/s of useful bytes\n", name, ms, bytes / ms / 1e9);
    };
    time("8 streams/wave, every lane 16 B (k_tail8 today)", 16.0 * nq, [&] { hipLaunchKernelGGL(k_read8<0>, dim3(streams / 8), dim3(64), 0, 0, in, qps, out); });
    time("8 streams/wave, one lane of eight 16 B (sparse)", 2.0 * nq, [&] { hipLaunchKernelGGL(k_read8<1>, dim3(streams / 8), dim3(64), 0, 0, in, qps, out); });
    time("8 streams/wave, every lane 8 B (energy array)", 8.0 * nq, [&] { hipLaunchKernelGGL(k_read8<2>, dim3(streams / 8), dim3(64), 0, 0, in, qps, out); });
    return 0;
}
