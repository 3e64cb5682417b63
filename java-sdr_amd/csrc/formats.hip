// formats.hip -- the data formats either side of the hot path:
//   in : IQ recordings -- RIFF/WAVE PCM-16 files as JavaAudio.openFile accepts them (JavaAudio.java:369-395,
//        compareFormat :397-406) and the headerless dumps written by recorder.receive (recorder.java:66-74) and
//        FCD.main (FCD.java:286-303) -- loaded straight into the stream-major device layout raw[S][stride] the
//        batch kernels take;
//   out: the waterfall pixel row of every PSD frame (waterfall.paintLine/getMax, waterfall.java:87-109).
#include "common.h"
#include <errno.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

namespace jsdr {

// Java (int) of a float: truncate toward zero, saturate, NaN -> 0
__device__ __forceinline__ int java_f2i(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}

// One 256-thread workgroup per PSD frame (grid-stride over the frames): the frame is read once with 16-byte loads
// into LDS, every thread then takes the maximum of its pixels' bins from LDS and writes its pixels, a contiguous
// run of the row (rotated by width/2) per wave.  HBM bound: 4 B read per bin + 4 B written per pixel.
__global__ __launch_bounds__(256) void k_waterfall(const float *__restrict__ psd, long long nframes, int n, int width,
                                                   unsigned peak_rgb, unsigned *__restrict__ pix)
{
    extern __shared__ __align__(16) float fr[];  // [n]
    const int tid = threadIdx.x;
    const float step = (float)n / (float)width;
    const float h = -2.55f;
    const int l = java_f2i(step);
    const int pr = (int)((peak_rgb >> 16) & 0xff), pg = (int)((peak_rgb >> 8) & 0xff), pb = (int)(peak_rgb & 0xff);
    for (long long frame = blockIdx.x; frame < nframes; frame += gridDim.x) {
        const float *a = psd + frame * (n + 2);  // rows are n+2 floats: 8-byte aligned only
        const float2 *a2 = reinterpret_cast<const float2 *>(a);
        for (int i = tid; i < n / 2; i += 256) reinterpret_cast<float2 *>(fr)[i] = a2[i];
        if ((n & 1) && tid == 0) fr[n - 1] = a[n - 1];
        __syncthreads();
        unsigned *row = pix + frame * width;
        for (int p = tid; p < width; p += 256) {
            const int o = java_f2i((float)p * step);
            // waterfall.java:102-109 getMax: '>' never replaces with or by a NaN
            float r = fr[o];
            for (int i = o + 1; i < o + l; i++) {
                const float v = fr[i];
                if (v > r) r = v;
            }
            int f = 255 - java_f2i(r * h);
            f = f < 0 ? 0 : f;
            f = f > 255 ? 255 : f;
            const unsigned cr = (unsigned)(pr * f / 256), cg = (unsigned)(pg * f / 256), cb = (unsigned)(pb * f / 256);
            row[(p + width / 2) % width] = 0xff000000u | (cr << 16) | (cg << 8) | cb;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------- recordings
struct Mapped {
    const unsigned char *p = nullptr;
    size_t len = 0;
    int fd = -1;
    ~Mapped()
    {
        if (p) munmap(const_cast<unsigned char *>(p), len);
        if (fd >= 0) close(fd);
    }
};

static int map_file(const char *path, Mapped &m)
{
    m.fd = open(path, O_RDONLY);
    JSDR_REQUIRE(m.fd >= 0, "Not readable: %s (%s)", path, strerror(errno));  // JavaAudio.java:392
    struct stat st;
    JSDR_REQUIRE(fstat(m.fd, &st) == 0 && S_ISREG(st.st_mode), "Not readable: %s (not a regular file)", path);
    m.len = (size_t)st.st_size;
    if (m.len == 0) return JSDR_OK;
    void *p = mmap(nullptr, m.len, PROT_READ, MAP_PRIVATE, m.fd, 0);
    JSDR_REQUIRE(p != MAP_FAILED, "Unable to open file: %s (mmap: %s)", path, strerror(errno));
    m.p = static_cast<const unsigned char *>(p);
    return JSDR_OK;
}

static unsigned le16(const unsigned char *p) { return (unsigned)p[0] | ((unsigned)p[1] << 8); }
static unsigned le32(const unsigned char *p) { return le16(p) | (le16(p + 2) << 16); }

// RIFF/WAVE: 'fmt ' (PCM = 1 or WAVE_FORMAT_EXTENSIBLE with the PCM sub-format) and 'data'; anything else in the
// file is skipped chunk by chunk.  A file without the RIFF magic is a headerless dump.
static int probe_mapped(const char *path, const Mapped &m, jsdr_recording_info *info)
{
    memset(info, 0, sizeof(*info));
    const unsigned char *b = m.p;
    if (m.len >= 12 && memcmp(b, "RIFF", 4) == 0 && memcmp(b + 8, "WAVE", 4) == 0) {
        info->format = JSDR_REC_WAV;
        size_t pos = 12;
        bool have_fmt = false, have_data = false;
        while (pos + 8 <= m.len) {
            const unsigned char *c = b + pos;
            const size_t clen = le32(c + 4);
            const size_t body = pos + 8;
            if (memcmp(c, "fmt ", 4) == 0) {
                JSDR_REQUIRE(clen >= 16 && body + 16 <= m.len, "Unable to open file: %s (short fmt chunk)", path);
                unsigned tag = le16(b + body);
                info->channels = (int)le16(b + body + 2);
                info->rate = (int)le32(b + body + 4);
                info->bits = (int)le16(b + body + 14);
                if (tag == 0xfffe && clen >= 26 && body + 26 <= m.len) tag = le16(b + body + 24);  // extensible
                info->encoding = (int)tag;
                have_fmt = true;
            } else if (memcmp(c, "data", 4) == 0) {
                JSDR_REQUIRE(have_fmt, "Unable to open file: %s (data chunk before fmt)", path);
                size_t avail = m.len - body;
                size_t dlen = clen < avail ? clen : avail;  // a recorder killed mid-file leaves a long length
                info->data_offset = (int64_t)body;
                const int fs = info->channels * (info->bits / 8);
                info->frames = fs > 0 ? (int64_t)(dlen / (size_t)fs) : 0;
                have_data = true;
                break;
            }
            pos = body + clen + (clen & 1);
        }
        JSDR_REQUIRE(have_fmt && have_data, "Unable to open file: %s (no fmt/data chunk)", path);
        return JSDR_OK;
    }
    info->format = JSDR_REC_RAW;  // recorder.java:66-74 writes the audio bytes as they come: no header at all
    info->encoding = 1;
    info->bits = 16;
    info->data_offset = 0;
    info->frames = 0;  // needs the channel count: filled by the caller
    return JSDR_OK;
}

}  // namespace jsdr

using namespace jsdr;

extern "C" {

int jsdr_waterfall_lines(const float *psd_dev, int64_t nframes, int n, int width, uint32_t peak_rgb, uint32_t *pix_dev,
                         void *stream)
{
    JSDR_REQUIRE(psd_dev && pix_dev, "jsdr_waterfall_lines: null buffer");
    JSDR_REQUIRE(n > 0 && width > 0, "jsdr_waterfall_lines: n=%d width=%d must be positive", n, width);
    JSDR_REQUIRE(nframes >= 0, "jsdr_waterfall_lines: negative frame count");
    JSDR_REQUIRE((size_t)n * sizeof(float) <= 128 * 1024, "jsdr_waterfall_lines: n=%d bins exceed the 128 KB LDS image", n);
    if ((size_t)n * sizeof(float) > 64 * 1024) JSDR_LDS_ATTR(k_waterfall, 128 * 1024);  // e.g. n = 19200, the 192 kHz default frame
    // (the reference would throw ArrayIndexOutOfBounds past the bins; (int)((width-1)*step) + (int)step <= n always)
    if (nframes == 0) return JSDR_OK;
    const long long cap = 256LL * 8 * 4;  // 8 workgroups per CU resident, a few rounds
    const unsigned grid = (unsigned)(nframes < cap ? nframes : cap);
    hipLaunchKernelGGL(k_waterfall, dim3(grid), dim3(256), (size_t)n * sizeof(float), as_stream(stream), psd_dev,
                       (long long)nframes, n, width, (unsigned)peak_rgb, reinterpret_cast<unsigned *>(pix_dev));
    JSDR_LAUNCH_CHECK();
    return JSDR_OK;
}

int jsdr_recording_probe(const char *path, int raw_channels, jsdr_recording_info *info)
{
    JSDR_REQUIRE(path && info, "jsdr_recording_probe: null argument");
    Mapped m;
    if (map_file(path, m) != JSDR_OK) return JSDR_ERR;
    if (probe_mapped(path, m, info) != JSDR_OK) return JSDR_ERR;
    if (info->format == JSDR_REC_RAW) {
        JSDR_REQUIRE(raw_channels == 1 || raw_channels == 2, "jsdr_recording_probe: raw_channels must be 1 or 2");
        info->channels = raw_channels;
        info->frames = (int64_t)(m.len / (size_t)(2 * raw_channels));
    }
    return JSDR_OK;
}

int jsdr_recordings_load(const char *const *paths, int nstreams, int channels, int rate, int64_t first_frame,
                         int64_t nframes, int16_t *raw_dev, int64_t stream_stride_i16, int64_t *frames_loaded,
                         void *stream)
{
    JSDR_REQUIRE(paths && raw_dev, "jsdr_recordings_load: null argument");
    JSDR_REQUIRE(nstreams > 0, "jsdr_recordings_load: nstreams must be positive");
    JSDR_REQUIRE(channels == 1 || channels == 2, "jsdr_recordings_load: channels must be 1 or 2 (audio-mode-I/IQ)");
    JSDR_REQUIRE(first_frame >= 0 && nframes >= 0, "jsdr_recordings_load: negative frame range");
    JSDR_REQUIRE(stream_stride_i16 >= 2 * nframes, "jsdr_recordings_load: stream stride %lld too small for %lld frames",
                 (long long)stream_stride_i16, (long long)nframes);
    hipStream_t st = as_stream(stream);
    std::vector<int16_t> widen;
    bool padded = false;
    for (int s = 0; s < nstreams; s++) {
        JSDR_REQUIRE(paths[s], "jsdr_recordings_load: null path for stream %d", s);
        Mapped m;
        jsdr_recording_info info;
        if (map_file(paths[s], m) != JSDR_OK || probe_mapped(paths[s], m, &info) != JSDR_OK) return JSDR_ERR;
        if (info.format == JSDR_REC_RAW) {
            info.channels = channels;
            info.rate = rate;
            info.frames = (int64_t)(m.len / (size_t)(2 * channels));
        }
        // JavaAudio.compareFormat (:397-406): channels, encoding, frame size, rate and sample size must all match;
        // the reference then tries an AudioSystem conversion, which is not reproduced here
        JSDR_REQUIRE(info.encoding == 1 && info.bits == 16 && info.channels == channels && (rate <= 0 || info.rate == rate),
                     "Incompatible audio format: %s: encoding %d, %d Hz, %d bit, %d channel(s); wanted PCM_SIGNED %d Hz, "
                     "16 bit, %d channel(s)", paths[s], info.encoding, info.rate, info.bits, info.channels, rate, channels);
        int64_t have = info.frames - first_frame;
        if (have < 0) have = 0;
        if (have > nframes) have = nframes;
        int16_t *dst = raw_dev + (int64_t)s * stream_stride_i16;
        const unsigned char *src = m.p + info.data_offset + (size_t)first_frame * (size_t)(2 * channels);
        if (have > 0) {
            if (channels == 2) {
                // little-endian 16-bit pairs are the device layout already
                JSDR_HIP_TRY(hipMemcpyAsync(dst, src, (size_t)have * 4, hipMemcpyHostToDevice, st));
            } else {
                // audio-mode-I: a mono recording becomes (I, 0) pairs (JavaAudio.java:286-288 sets Q = 0)
                widen.assign((size_t)have * 2, 0);
                for (int64_t i = 0; i < have; i++) widen[(size_t)2 * i] = (int16_t)le16(src + 2 * i);
                JSDR_HIP_TRY(hipMemcpyAsync(dst, widen.data(), (size_t)have * 4, hipMemcpyHostToDevice, st));
            }
            // the source is unmapped (or reused) when this iteration ends
            JSDR_HIP_TRY(hipStreamSynchronize(st));
        }
        if (have < nframes) {
            JSDR_HIP_TRY(hipMemsetAsync(dst + 2 * have, 0, (size_t)(nframes - have) * 4, st));
            padded = true;
        }
        if (frames_loaded) frames_loaded[s] = have;
    }
    // The zero padding of a short or exhausted recording was only ENQUEUED on `st` (the legacy null stream when the caller
    // passes none): a consumer on a non-blocking stream of its own -- the group's device threads -- is not ordered behind
    // it and would read the previous block's samples.  The copies above are complete when this returns; so is the padding.
    if (padded) JSDR_HIP_TRY(hipStreamSynchronize(st));
    return JSDR_OK;
}

}  // extern "C"
