/*
 * o_synth.c -- synthetic IQ generators for tests and the CPU baseline.
 * TEST INFRASTRUCTURE.  NOT derived from the reference (which has no signal source but a
 * sound card): the signal definitions are this repo's own, stated in DESIGN.md section
 * "Synthetic inputs".  Everything is integer arithmetic on a counter-based hash so that the
 * HIP generator (csrc/synth.hip, jsdr_synth_*) is bit-identical by construction.
 */
#include "jsdr_oracle.h"
#include <math.h>

uint64_t jo_mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

void jo_synth_payload(uint64_t seed, int stream, int frame, uint8_t out[256])
{
    uint64_t key = jo_mix64(seed ^ ((uint64_t)(uint32_t)stream << 20));
    for (int j = 0; j < 32; j++) {
        uint64_t w = jo_mix64(key + (uint64_t)(uint32_t)frame * 32u + (uint64_t)j);
        for (int b = 0; b < 8; b++) out[8 * j + b] = (uint8_t)(w >> (8 * b));
    }
}

void jo_synth_diffsign(const uint8_t *sym, int64_t nsym, int8_t *dsign, int8_t start)
{
    int8_t s = start;
    for (int64_t m = 0; m < nsym; m++) {
        if (!sym[m]) s = (int8_t)-s; /* symbol 0 = phase reversal, 1 = no change */
        dsign[m] = s;
    }
}

void jo_synth_tables(int amp, int16_t cos_tab[1024], int16_t sin_tab[1024])
{
    for (int k = 0; k < 1024; k++) {
        double a = 2.0 * 3.14159265358979323846 * (double)k / 1024.0;
        cos_tab[k] = (int16_t)lrint((double)amp * cos(a));
        sin_tab[k] = (int16_t)lrint((double)amp * sin(a));
    }
}

static inline int32_t noise_from_hash(uint64_t h, int gain)
{
    int32_t c = (int32_t)(h & 0xffff) + (int32_t)((h >> 16) & 0xffff) + (int32_t)((h >> 32) & 0xffff) +
                (int32_t)(h >> 48) - 131070;
    return (int32_t)(((int64_t)c * (int64_t)gain) >> 15);
}

static inline int16_t clip16(int32_t v)
{
    if (v > 32767) v = 32767;
    if (v < -32767) v = -32767;
    return (int16_t)v;
}

void jo_synth_dbpsk(int16_t *out, int64_t n0, int64_t n, const int8_t *dsign, int64_t nsym,
                    int samples_per_sym, uint32_t phase0, uint32_t phase_inc,
                    const int16_t *cos_tab, const int16_t *sin_tab, int noise_gain, uint64_t noise_key)
{
    for (int64_t i = 0; i < n; i++) {
        uint64_t g = (uint64_t)(n0 + i);
        int64_t m = (int64_t)(g / (uint64_t)samples_per_sym) % nsym;
        uint32_t ph = phase0 + (uint32_t)(g * (uint64_t)phase_inc);
        uint32_t idx = ph >> 22;
        int32_t d = dsign[m];
        int32_t vi = d * (int32_t)cos_tab[idx];
        int32_t vq = d * (int32_t)sin_tab[idx];
        if (noise_gain) {
            vi += noise_from_hash(jo_mix64(noise_key + 2 * g), noise_gain);
            vq += noise_from_hash(jo_mix64(noise_key + 2 * g + 1), noise_gain);
        }
        out[2 * i] = clip16(vi);
        out[2 * i + 1] = clip16(vq);
    }
}

void jo_synth_tones(int16_t *out, int64_t frame0, int64_t nframes, int n, const int16_t *cos_tab,
                    int noise_gain, uint64_t key)
{
    uint32_t binstep = (uint32_t)(4294967296.0 / (double)n); /* n is a power of two */
    for (int64_t f = 0; f < nframes; f++) {
        uint64_t fr = (uint64_t)(frame0 + f);
        uint64_t h = jo_mix64(key ^ jo_mix64(fr));
        int ntones = 1 + (int)(h % 3u);
        uint32_t inc[3], ph0[3];
        int32_t amp[3];
        for (int t = 0; t < 3; t++) {
            uint64_t ht = jo_mix64(h + (uint64_t)(t + 1));
            inc[t] = (uint32_t)(ht % (uint64_t)n) * binstep;
            ph0[t] = (uint32_t)(ht >> 32);
            amp[t] = 32 + (int32_t)((ht >> 24) & 0x7f); /* /256 of table amplitude */
        }
        int16_t *o = out + (size_t)f * 2 * (size_t)n;
        for (int s = 0; s < n; s++) {
            int32_t vi = 0, vq = 0;
            for (int t = 0; t < ntones; t++) {
                uint32_t idx = (ph0[t] + (uint32_t)s * inc[t]) >> 22;
                vi += ((int32_t)cos_tab[idx] * amp[t]) >> 8;
                vq += ((int32_t)cos_tab[(idx + 768u) & 1023u] * amp[t]) >> 8;
            }
            if (noise_gain) {
                uint64_t g = fr * (uint64_t)n + (uint64_t)s;
                vi += noise_from_hash(jo_mix64(key + 2 * g), noise_gain);
                vq += noise_from_hash(jo_mix64(key + 2 * g + 1), noise_gain);
            }
            o[2 * s] = clip16(vi);
            o[2 * s + 1] = clip16(vq);
        }
    }
}
