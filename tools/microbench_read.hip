// tools/microbench_read.hip -- what HBM delivers to a READ-ONLY stream on this box, by access pattern:
// (a) grid-stride 16-byte loads over one 13.7 GB buffer, (b) k_tail's pattern: one wave per stream region (1.68 MB apart),
// walking it in 8 KB chunks (eight 16-byte loads per lane in flight), 5 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mb_read tools/microbench_read.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_read_stride(const int4 *__restrict__ in, long long nq, int *out)
{
    int acc = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nq; i += (long long)gridDim.x * 256) {
        const int4 v = in[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678) out[threadIdx.x] = acc;
}
template <int K>
__global__ __launch_bounds__(64) void k_read_streams(const int4 *__restrict__ in, long long quads_per_stream, int *out)
{
    const int4 *p = in + (long long)blockIdx.x * quads_per_stream;
    int acc = 0;
    for (long long c = 0; c + 64 * K <= quads_per_stream; c += 64 * K) {
        int4 v[K];
#pragma unroll
        for (int k = 0; k < K; k++) v[k] = p[c + 64 * k + threadIdx.x];
#pragma unroll
        for (int k = 0; k < K; k++) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    }
    if (acc == 0x12345678) out[threadIdx.x] = acc;
}
int main()
{
    const long long streams = 8192, qps = 104858;  // quads (16 B) per stream = k_tail's (fi,fq) per call
    const long long nq = streams * qps;
    int4 *in;
    int *out;
    CK(hipMalloc(&in, nq * 16));
    CK(hipMalloc(&out, 4096));
    CK(hipMemset(in, 1, nq * 16));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    auto time = [&](const char *name, auto launch) {
        for (int i = 0; i < 2; i++) launch();
        hipEventRecord(a);
        for (int i = 0; i < 5; i++) launch();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        ms /= 5;
        printf("%-44s %.3f ms  %.2f TB/s (read)\n", name, ms, 16.0 * nq / ms / 1e9);
    };
    for (int grid : {2048, 8192, 32768})
        time(("grid-stride x4 loads, grid " + std::to_string(grid)).c_str(), [&] { hipLaunchKernelGGL(k_read_stride, dim3(grid), dim3(256), 0, 0, in, nq, out); });
    time("one wave per stream, 8 loads in flight", [&] { hipLaunchKernelGGL(k_read_streams<8>, dim3(streams), dim3(64), 0, 0, in, qps, out); });
    time("one wave per stream, 16 loads in flight", [&] { hipLaunchKernelGGL(k_read_streams<16>, dim3(streams), dim3(64), 0, 0, in, qps, out); });
    time("one wave per stream, 4 loads in flight", [&] { hipLaunchKernelGGL(k_read_streams<4>, dim3(streams), dim3(64), 0, 0, in, qps, out); });
    return 0;
}
