// synth.hip -- on-device synthetic IQ generators (bench / tests).  NOT part of the reference: java-sdr's
// only signal source is a sound card.  Integer arithmetic on a counter-based hash, bit-identical to
// oracle/o_synth.c by construction (tests/test_gpu_fir_phase_fec.py::test_synth_generators_bit_identical_to_oracle), so
// that multi-GB inputs never cross PCIe.
#include "common.h"

namespace jsdr {

__host__ __device__ __forceinline__ unsigned long long mix64(unsigned long long z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

__device__ __forceinline__ int noise_from_hash(unsigned long long h, int gain)
{
    int c = (int)(h & 0xffff) + (int)((h >> 16) & 0xffff) + (int)((h >> 32) & 0xffff) + (int)(h >> 48) - 131070;
    return (int)(((long long)c * (long long)gain) >> 15);
}

__device__ __forceinline__ int clip16(int v)
{
    v = v > 32767 ? 32767 : v;
    v = v < -32767 ? -32767 : v;
    return v;
}

__global__ void k_synth_payloads(unsigned long long seed, int stream0, int nstreams, int nframes,
                                 unsigned long long *__restrict__ out)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)nstreams * nframes * 32;
    if (i >= total) return;
    int j = (int)(i & 31);
    long long sf = i >> 5;
    int frame = (int)(sf % nframes);
    int stream = stream0 + (int)(sf / nframes);
    unsigned long long key = mix64(seed ^ ((unsigned long long)(unsigned)stream << 20));
    out[i] = mix64(key + (unsigned long long)(unsigned)frame * 32u + (unsigned long long)j);  // little-endian bytes
}

// differential sign: symbol 0 flips the sign, 1 keeps it; one lane per stream (sequential prefix)
__global__ void k_synth_diffsign(const unsigned char *__restrict__ sym, long long nsym, int nstreams,
                                 signed char *__restrict__ dsign)
{
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nstreams) return;
    const unsigned char *p = sym + (long long)s * nsym;
    signed char *o = dsign + (long long)s * nsym;
    int cur = 1;
    for (long long m = 0; m < nsym; m++) {
        if (!p[m]) cur = -cur;
        o[m] = (signed char)cur;
    }
}

struct DbpskArgs {
    int *out;  // int16 pairs as dwords
    long long stream_stride_pairs;
    int nstreams;
    long long n0, n;
    const signed char *dsign;
    long long nsym;
    int sps;
    unsigned phase0, phase_inc;
    const short *cos_tab, *sin_tab;
    int noise_gain;
    const unsigned long long *noise_keys;
};

__global__ void k_synth_dbpsk(DbpskArgs a)
{
    __shared__ short ct[1024], st[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) {
        ct[i] = a.cos_tab[i];
        st[i] = a.sin_tab[i];
    }
    __syncthreads();
    const int s = blockIdx.y;
    const signed char *ds = a.dsign + (long long)s * a.nsym;
    const unsigned long long key = a.noise_keys ? a.noise_keys[s] : 0ull;
    int *o = a.out + (long long)s * a.stream_stride_pairs;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (long long)gridDim.x * blockDim.x) {
        unsigned long long g = (unsigned long long)(a.n0 + i);
        long long m = (long long)(g / (unsigned long long)a.sps) % a.nsym;
        unsigned ph = a.phase0 + (unsigned)(g * (unsigned long long)a.phase_inc);
        unsigned idx = ph >> 22;
        int d = ds[m];
        int vi = d * (int)ct[idx];
        int vq = d * (int)st[idx];
        if (a.noise_gain) {
            vi += noise_from_hash(mix64(key + 2 * g), a.noise_gain);
            vq += noise_from_hash(mix64(key + 2 * g + 1), a.noise_gain);
        }
        o[i] = (clip16(vi) & 0xffff) | (clip16(vq) << 16);
    }
}

__global__ void k_synth_tones(int *__restrict__ out, long long frame0, long long nframes, int n,
                              const short *__restrict__ cos_tab, int noise_gain, unsigned long long key)
{
    __shared__ short ct[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) ct[i] = cos_tab[i];
    __syncthreads();
    const unsigned binstep = (unsigned)(4294967296.0 / (double)n);
    for (long long f = blockIdx.x; f < nframes; f += gridDim.x) {
        unsigned long long fr = (unsigned long long)(frame0 + f);
        unsigned long long h = mix64(key ^ mix64(fr));
        int ntones = 1 + (int)(h % 3u);
        unsigned inc[3], ph0[3];
        int amp[3];
#pragma unroll
        for (int t = 0; t < 3; t++) {
            unsigned long long ht = mix64(h + (unsigned long long)(t + 1));
            inc[t] = (unsigned)(ht % (unsigned long long)n) * binstep;
            ph0[t] = (unsigned)(ht >> 32);
            amp[t] = 32 + (int)((ht >> 24) & 0x7f);
        }
        int *o = out + f * n;
        for (int s = threadIdx.x; s < n; s += blockDim.x) {
            int vi = 0, vq = 0;
#pragma unroll
            for (int t = 0; t < 3; t++) {
                if (t < ntones) {
                    unsigned idx = (ph0[t] + (unsigned)s * inc[t]) >> 22;
                    vi += ((int)ct[idx] * amp[t]) >> 8;
                    vq += ((int)ct[(idx + 768u) & 1023u] * amp[t]) >> 8;
                }
            }
            if (noise_gain) {
                unsigned long long g = fr * (unsigned long long)n + (unsigned long long)s;
                vi += noise_from_hash(mix64(key + 2 * g), noise_gain);
                vq += noise_from_hash(mix64(key + 2 * g + 1), noise_gain);
            }
            o[s] = (clip16(vi) & 0xffff) | (clip16(vq) << 16);
        }
    }
}

}  // namespace jsdr

using namespace jsdr;

extern "C" {

int jsdr_synth_payloads(uint64_t seed, int stream0, int nstreams, int nframes, uint8_t *out_dev, void *stream)
{
    JSDR_REQUIRE(out_dev, "jsdr_synth_payloads: null buffer");
    JSDR_REQUIRE(nstreams > 0 && nframes > 0, "jsdr_synth_payloads: bad geometry");
    long long total = (long long)nstreams * nframes * 32;
    hipLaunchKernelGGL(k_synth_payloads, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                       (unsigned long long)seed, stream0, nstreams, nframes,
                       reinterpret_cast<unsigned long long *>(out_dev));
    JSDR_LAUNCH_CHECK();
    return JSDR_OK;
}

int jsdr_synth_diffsign(const uint8_t *sym_dev, int64_t nsym, int nstreams, int8_t *dsign_dev, void *stream)
{
    JSDR_REQUIRE(sym_dev && dsign_dev, "jsdr_synth_diffsign: null buffer");
    JSDR_REQUIRE(nsym > 0 && nstreams > 0, "jsdr_synth_diffsign: bad geometry");
    hipLaunchKernelGGL(k_synth_diffsign, dim3((nstreams + 63) / 64), dim3(64), 0, as_stream(stream), sym_dev,
                       (long long)nsym, nstreams, dsign_dev);
    JSDR_LAUNCH_CHECK();
    return JSDR_OK;
}

int jsdr_synth_dbpsk(int16_t *out_dev, int64_t stream_stride_i16, int nstreams, int64_t n0, int64_t n,
                     const int8_t *dsign_dev, int64_t nsym, int samples_per_sym, uint32_t phase0, uint32_t phase_inc,
                     const int16_t *cos_tab_dev, const int16_t *sin_tab_dev, int noise_gain,
                     const uint64_t *noise_keys_dev, void *stream)
{
    JSDR_REQUIRE(out_dev && dsign_dev && cos_tab_dev && sin_tab_dev, "jsdr_synth_dbpsk: null buffer");
    JSDR_REQUIRE(nstreams > 0 && nstreams <= 65535 && n > 0 && nsym > 0 && samples_per_sym > 0,
                 "jsdr_synth_dbpsk: bad geometry");
    JSDR_REQUIRE((stream_stride_i16 & 1) == 0 && stream_stride_i16 >= 2 * n, "jsdr_synth_dbpsk: bad stream stride");
    JSDR_REQUIRE(noise_gain == 0 || noise_keys_dev, "jsdr_synth_dbpsk: noise keys missing");
    DbpskArgs a;
    a.out = reinterpret_cast<int *>(out_dev);
    a.stream_stride_pairs = stream_stride_i16 / 2;
    a.nstreams = nstreams;
    a.n0 = n0;
    a.n = n;
    a.dsign = dsign_dev;
    a.nsym = nsym;
    a.sps = samples_per_sym;
    a.phase0 = phase0;
    a.phase_inc = phase_inc;
    a.cos_tab = cos_tab_dev;
    a.sin_tab = sin_tab_dev;
    a.noise_gain = noise_gain;
    a.noise_keys = reinterpret_cast<const unsigned long long *>(noise_keys_dev);
    long long gx = (n + 255) / 256;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(k_synth_dbpsk, dim3((unsigned)gx, (unsigned)nstreams), dim3(256), 0, as_stream(stream), a);
    JSDR_LAUNCH_CHECK();
    return JSDR_OK;
}

int jsdr_synth_tones(int16_t *out_dev, int64_t frame0, int64_t nframes, int n, const int16_t *cos_tab_dev,
                     int noise_gain, uint64_t key, void *stream)
{
    JSDR_REQUIRE(out_dev && cos_tab_dev, "jsdr_synth_tones: null buffer");
    JSDR_REQUIRE(nframes > 0 && n > 0 && (n & (n - 1)) == 0, "jsdr_synth_tones: n must be a power of two");
    long long g = nframes < 65535 ? nframes : 65535;
    hipLaunchKernelGGL(k_synth_tones, dim3((unsigned)g), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<int *>(out_dev), (long long)frame0, (long long)nframes, n, cos_tab_dev,
                       noise_gain, (unsigned long long)key);
    JSDR_LAUNCH_CHECK();
    return JSDR_OK;
}

}  // extern "C"
