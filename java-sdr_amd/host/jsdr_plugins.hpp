// jsdr_plugins.hpp -- C++ mirror of java-sdr's plugin surface for the accelerated path, above the C ABI.
//
// The reference's host language is Java; this image has no JDK, so (per the build rules) the host side
// is written in C++ with the reference's own names, argument meaning and error behaviour:
//   IAudioHandler / IRawHandler      IAudioHandler.java:3-6, IRawHandler.java:3-6
//   AudioDescriptor, IAudio          AudioDescriptor.java:3-16, IAudio.java:3-21 (the members the plugins use)
//   IPublish / IPublishListener      IPublish.java:3-8 ("fft-psd", "<name>-bpsk-centre", "<name>-bpsk-tune")
//   IConfig                          IConfig.java:3-7
//   fft, phase, FUNcubeBPSKDemod     constructor argument order of jsdr.java:475-483 (UI host dropped)
//   FECDecoder                       FECDecoder.java:703
// receive() returns void like the reference; a failure of the accelerator throws std::runtime_error, the
// analogue of an unchecked exception escaping a handler (caught by the audio loop, JavaAudio.java:321-328).
// INTEGRATION.md shows the JNI form of the same classes for a real java-sdr build.
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/jsdr_hip.h"

namespace java_sdr {

struct AudioDescriptor {
    int rate, bits, chns, size, blen;
    AudioDescriptor(int r, int b, int c, int s, int l) : rate(r), bits(b), chns(c), size(s), blen(l) {}
};

struct IAudioHandler {
    virtual ~IAudioHandler() = default;
    // sample buffer always contains IQ data (2*floats/sample) in -1.0/1.0 range
    virtual void receive(const float *buf, size_t len) = 0;
};

struct IRawHandler {
    virtual ~IRawHandler() = default;
    // sample buffer contains raw bytes from radio
    virtual void receive(const uint8_t *buf, size_t len) = 0;
};

struct IAudio {
    virtual ~IAudio() = default;
    virtual AudioDescriptor getAudioDescriptor() = 0;
    virtual int getICorrection() = 0;
    virtual int getQCorrection() = 0;
    virtual void addHandler(IAudioHandler *hand) = 0;
    virtual void remHandler(IAudioHandler *hand) = 0;
    virtual void addRawHandler(IRawHandler *hand) = 0;
    virtual void remRawHandler(IRawHandler *hand) = 0;
};

// published values are either an Integer or a float[] in the reference
struct PublishValue {
    bool is_int = true;
    int i = 0;
    std::vector<float> f;
};

struct IPublishListener {
    virtual ~IPublishListener() = default;
    virtual void notify(const std::string &key, const PublishValue &val) = 0;
};

struct IPublish {
    virtual ~IPublish() = default;
    virtual void setPublish(const std::string &key, const PublishValue &val) = 0;
    virtual void listen(IPublishListener *l) = 0;
    virtual void unlisten(IPublishListener *l) = 0;
};

struct IConfig {
    virtual ~IConfig() = default;
    virtual int getIntConfig(const std::string &key, int def) = 0;
    virtual void setIntConfig(const std::string &key, int val) = 0;
};

inline void jsdr_throw(const char *what)
{
    throw std::runtime_error(std::string(what) + ": " + jsdr_last_error());
}

// ------------------------------------------------------------------ fft.java
class fft : public IAudioHandler, public IRawHandler {
  public:
    fft(IConfig *, IPublish *pub, IAudio *aud) : publish(pub) { setup(aud); }
    ~fft() override
    {
        if (audio) audio->remHandler(this);
        jsdr_fft_destroy(h);
    }
    // fft.java:63-77: size everything from the audio descriptor, (re)attach to the audio source
    void setup(IAudio *aud)
    {
        audio = aud;
        AudioDescriptor adsc = audio->getAudioDescriptor();
        if (h) jsdr_fft_destroy(h);
        h = nullptr;
        n = adsc.blen / adsc.size;
        psd.assign((size_t)n + 2, 0.f);
        if (jsdr_fft_create(&h, n, adsc.rate) != JSDR_OK) jsdr_throw("fft.setup");
        audio->remHandler(this);
        audio->addHandler(this);
    }
    // fft.java:190-228
    void receive(const float *buf, size_t len) override
    {
        if (len != (size_t)2 * n) throw std::runtime_error("fft.receive: buffer length != 2*blen/size");
        if (jsdr_fft_receive_f32(h, buf, psd.data()) != JSDR_OK) jsdr_throw("fft.receive");
        publish_psd();
    }
    // raw form: the int16 -> float rule of JavaAudio.java:276-293 runs inside the kernel
    void receive(const uint8_t *buf, size_t len) override
    {
        if (len != (size_t)4 * n) throw std::runtime_error("fft.receive(raw): buffer length != blen");
        if (jsdr_fft_receive_i16(h, reinterpret_cast<const int16_t *>(buf), audio->getICorrection(),
                                 audio->getQCorrection(), psd.data()) != JSDR_OK)
            jsdr_throw("fft.receive(raw)");
        publish_psd();
    }
    const std::vector<float> &getPsd() const { return psd; }

  private:
    void publish_psd()
    {
        if (!publish) return;
        PublishValue v;
        v.is_int = false;
        v.f = psd;  // listeners clone (waterfall.java:33); a fresh vector here
        publish->setPublish("fft-psd", v);
    }
    IPublish *publish = nullptr;
    IAudio *audio = nullptr;
    jsdr_fft *h = nullptr;
    int n = 0;
    std::vector<float> psd;
};

// ------------------------------------------------------------------ waterfall.java (headless: the pixel array)
// Listens for "fft-psd" (waterfall.java:28-36) and keeps the pixel image paintComponent maintains (:62-80): every
// new PSD scrolls the image down one row and paints the top row with paintLine (:87-100) -- computed on the device.
class waterfall : public IPublishListener {
  public:
    waterfall(IPublish *pub, IAudio *, int width_, int height_) : width(width_), height(height_)
    {
        pixs.assign((size_t)width * height, 0xff000000u);  // g.fillRect black (:49-50)
        if (jsdr_malloc(&d_pix, sizeof(uint32_t) * (size_t)width) != JSDR_OK) jsdr_throw("waterfall");
        pub->listen(this);
    }
    ~waterfall() override
    {
        jsdr_free(d_pix);
        jsdr_free(d_psd);
    }
    void notify(const std::string &key, const PublishValue &val) override
    {
        if (key != "fft-psd" || val.is_int || val.f.size() < 3) return;
        const int n = (int)val.f.size() - 2;
        if (n != bins) {  // "audio-change": a new frame size restarts the image (:38-45)
            jsdr_free(d_psd);
            d_psd = nullptr;
            if (jsdr_malloc(&d_psd, sizeof(float) * val.f.size()) != JSDR_OK) jsdr_throw("waterfall");
            bins = n;
            std::fill(pixs.begin(), pixs.end(), 0xff000000u);
        }
        if (jsdr_memcpy_h2d(d_psd, val.f.data(), sizeof(float) * val.f.size()) != JSDR_OK ||
            jsdr_waterfall_lines(static_cast<const float *>(d_psd), 1, n, width, peak, static_cast<uint32_t *>(d_pix),
                                 nullptr) != JSDR_OK)
            jsdr_throw("waterfall.paintLine");
        // System.arraycopy(pixs, 0, pixs, width, len - width) then the new top row (:76-78)
        std::memmove(pixs.data() + width, pixs.data(), sizeof(uint32_t) * (size_t)width * (size_t)(height - 1));
        if (jsdr_memcpy_d2h(pixs.data(), d_pix, sizeof(uint32_t) * (size_t)width) != JSDR_OK) jsdr_throw("waterfall");
        lines++;
    }
    const std::vector<uint32_t> &pixels() const { return pixs; }
    int getWidth() const { return width; }
    int getHeight() const { return height; }
    long linesPainted() const { return lines; }
    uint32_t peak = 0x00ffff;  // Color.CYAN (:15)

  private:
    int width, height, bins = -1;
    long lines = 0;
    std::vector<uint32_t> pixs;
    void *d_psd = nullptr, *d_pix = nullptr;
};

// ------------------------------------------------------------------ FECDecoder.java
class FECDecoder {
  public:
    // FECDecoder.java:703: returns -1 (RS failure) or the channel error count; RSdecdata untouched on failure
    int FECDecode(const uint8_t raw[5200], uint8_t RSdecdata[256])
    {
        int rc = -1;
        if (jsdr_fec_decode(raw, RSdecdata, &rc) != JSDR_OK) jsdr_throw("FECDecoder.FECDecode");
        return rc;
    }
};

// ------------------------------------------------------------------ FUNcubeBPSKDemod.java
class FUNcubeBPSKDemod : public IAudioHandler {
  public:
    FUNcubeBPSKDemod(int idx, IConfig *cfg, IPublish *pub, IAudio *aud)
        : name("FUNcube" + std::to_string(idx)), config(cfg), publish(pub)
    {
        setup(aud);
    }
    ~FUNcubeBPSKDemod() override
    {
        if (audio) audio->remHandler(this);
        jsdr_bpsk_destroy(h);
    }
    // FUNcubeBPSKDemod.java:192-209
    void setup(IAudio *aud)
    {
        audio = aud;
        AudioDescriptor adsc = audio->getAudioDescriptor();
        samples = adsc.blen / adsc.size;
        tuning = config ? config->getIntConfig(name + "-bpsk-tuning", 12000) : 12000;
        int dofft = config ? config->getIntConfig(name + "-bpsk-dofft", 0) : 0;
        int doup = config ? config->getIntConfig(name + "-bpsk-upper", 0) : 0;
        if (h) jsdr_bpsk_destroy(h);
        h = nullptr;
        if (jsdr_bpsk_create(&h, adsc.rate, samples, tuning, dofft, doup, 1, samples) != JSDR_OK)
            jsdr_throw("FUNcubeBPSKDemod.setup");
        doFFT = dofft != 0;
        audio->remHandler(this);
        audio->addHandler(this);
    }
    // FUNcubeBPSKDemod.java:357-379
    void receive(const float *buf, size_t len) override
    {
        if (len != (size_t)2 * samples) throw std::runtime_error("FUNcubeBPSKDemod.receive: bad buffer length");
        if (jsdr_bpsk_receive_f32(h, buf) != JSDR_OK) jsdr_throw("FUNcubeBPSKDemod.receive");
        if (publish && !doFFT) {
            PublishValue c, t;
            c.i = -1;
            t.i = tuning;
            publish->setPublish(name + "-bpsk-centre", c);
            publish->setPublish(name + "-bpsk-tune", t);
        }
    }
    // painted statistics of the reference (FUNcubeBPSKDemod.java:220-228)
    void counters(int32_t out[JSDR_BPSK_NCOUNTERS])
    {
        if (jsdr_bpsk_get_counters(h, 0, out) != JSDR_OK) jsdr_throw("FUNcubeBPSKDemod.counters");
    }
    std::vector<int8_t> lastBits()
    {
        int n = 0;
        if (jsdr_bpsk_get_bits(h, 0, nullptr, 0, &n) != JSDR_OK) jsdr_throw("FUNcubeBPSKDemod.lastBits");
        std::vector<int8_t> b((size_t)(n > 0 ? n : 0));
        if (n > 0 && jsdr_bpsk_get_bits(h, 0, b.data(), n, &n) != JSDR_OK) jsdr_throw("FUNcubeBPSKDemod.lastBits");
        return b;
    }
    void decoded(uint8_t out[256])
    {
        if (jsdr_bpsk_get_decoded(h, 0, out) != JSDR_OK) jsdr_throw("FUNcubeBPSKDemod.decoded");
    }

  private:
    std::string name;
    IConfig *config = nullptr;
    IPublish *publish = nullptr;
    IAudio *audio = nullptr;
    jsdr_bpsk *h = nullptr;
    int samples = 0, tuning = 12000;
    bool doFFT = false;
};

// ------------------------------------------------------------------ demod.java
// The AM/FM IAudioHandler (demod.java:31-483) without its Swing menu and javax.sound output: config keys, the
// filterMove range check (:300-312, the only caller of weights()), the mode / toggle actions (:184-203) and
// receive() -> the frame's audio bytes (what the reference writes to `bbf`, :469-481).
class demod : public IAudioHandler {
  public:
    static constexpr int MODE_OFF = 0, MODE_RAW = 1, MODE_AM = 2, MODE_NFM = 3, MODE_WFM = 4;  // :39-43
    demod(IConfig *cfg, IPublish *pub, IAudio *aud) : config(cfg), publish(pub)
    {
        mode = cfg->getIntConfig("demod-mode", MODE_OFF);                  // :82
        dofir = cfg->getIntConfig("demod-fir-enable", 0) > 0;              // :83
        doagc = cfg->getIntConfig("demod-agc-enable", 0) > 0;              // :84
        flo = cfg->getIntConfig("demod-filter-low", INT32_MIN);            // :86
        fhi = cfg->getIntConfig("demod-filter-high", INT32_MAX);           // :87
        setup(aud);
    }
    ~demod() override
    {
        if (audio) audio->remHandler(this);
        jsdr_demod_destroy(h);
    }
    // demod.java:220-241
    void setup(IAudio *aud)
    {
        audio = aud;
        AudioDescriptor ad = audio->getAudioDescriptor();
        if (h) jsdr_demod_destroy(h);
        h = nullptr;
        n = ad.blen / ad.size;  // sam.length / 2
        rate = ad.rate;
        if (jsdr_demod_create(&h, ad.rate, n, 1, n) != JSDR_OK) jsdr_throw("demod.setup");
        audioOut.assign((size_t)2 * n, 0);
        apply();
        filterMove(0, 0);
        audio->remHandler(this);
        audio->addHandler(this);
    }
    // the menu actions (:184-203)
    void setMode(int m)
    {
        mode = m;
        config->setIntConfig("demod-mode", mode);
        apply();
    }
    void toggleAgc() { doagc = !doagc; config->setIntConfig("demod-agc-enable", doagc ? 1 : 0); apply(); }
    void toggleFir() { dofir = !dofir; config->setIntConfig("demod-fir-enable", dofir ? 1 : 0); apply(); }
    void toggleDown() { dodwn = !dodwn; apply(); }
    // demod.java:300-312: move the band edges; weights() only runs when the new band is ordered and inside
    // (-rate/2, rate/2) -- at the default filter points it never does and the weights stay zero
    bool filterMove(int lo, int hi)
    {
        lo = (int)((uint32_t)lo + (uint32_t)flo);  // Java int arithmetic wraps
        hi = (int)((uint32_t)hi + (uint32_t)fhi);
        if (lo < hi && lo > (-rate / 2) && hi < rate / 2) {
            flo = lo;
            fhi = hi;
            if (jsdr_demod_weights(h, flo, fhi, nullptr, nullptr) != JSDR_OK) jsdr_throw("demod.weights");
            if (publish) {
                PublishValue v;
                v.i = flo;
                publish->setPublish("demod-filter-low", v);
                v.i = fhi;
                publish->setPublish("demod-filter-high", v);
            }
            return true;
        }
        return false;
    }
    // demod.java:398-483
    void receive(const float *buf, size_t len) override
    {
        if (len != (size_t)2 * n) throw std::runtime_error("demod.receive: buffer length != 2*blen/size");
        if (jsdr_demod_receive_f32(h, buf, audioOut.data()) != JSDR_OK) jsdr_throw("demod.receive");
    }
    const std::vector<int16_t> &audioBytes() const { return audioOut; }  // L,R per sample (:473-478)
    void levels(float &max, float &avg)
    {
        if (jsdr_demod_frame_stats(h, 0, &max, &avg) != JSDR_OK) jsdr_throw("demod.levels");
    }
    int filterLow() const { return flo; }
    int filterHigh() const { return fhi; }

  private:
    void apply()
    {
        if (h && jsdr_demod_configure(h, mode, dofir, dodwn, doagc) != JSDR_OK) jsdr_throw("demod.configure");
    }
    IConfig *config = nullptr;
    IPublish *publish = nullptr;
    IAudio *audio = nullptr;
    jsdr_demod *h = nullptr;
    int n = 0, rate = 0, mode = MODE_OFF, flo = INT32_MIN, fhi = INT32_MAX;
    bool dofir = false, dodwn = false, doagc = false;
    std::vector<int16_t> audioOut;
};

// ------------------------------------------------------------------ phase.java
class phase : public IAudioHandler {
  public:
    phase(IConfig *, IPublish *, IAudio *aud) : audio(aud)
    {
        AudioDescriptor ad = audio->getAudioDescriptor();
        n = ad.blen / ad.size;
        if (jsdr_malloc(&dpy, sizeof(float) * 2 * (size_t)n) != JSDR_OK || jsdr_malloc(&dmax, sizeof(float)) != JSDR_OK)
            jsdr_throw("phase");
        audio->addHandler(this);
    }
    ~phase() override
    {
        if (audio) audio->remHandler(this);
        jsdr_free(dpy);
        jsdr_free(dmax);
    }
    // phase.java:123-128: copy the frame (here: to the device)
    void receive(const float *buf, size_t len) override
    {
        if (len != (size_t)2 * n) throw std::runtime_error("phase.receive: bad buffer length");
        if (jsdr_memcpy_h2d(dpy, buf, sizeof(float) * len) != JSDR_OK) jsdr_throw("phase.receive");
    }
    // the two reductions of paintComponent (phase.java:75-80, :93-116)
    float maxAbs()
    {
        float m = 0;
        if (jsdr_phase_maxabs(static_cast<float *>(dpy), 1, n, static_cast<float *>(dmax), nullptr) != JSDR_OK ||
            jsdr_memcpy_d2h(&m, dmax, sizeof(float)) != JSDR_OK)
            jsdr_throw("phase.maxAbs");
        return m;
    }
    int columnMeans(int bx, std::vector<int32_t> &pix, std::vector<float> &avgi, std::vector<float> &avgq)
    {
        pix.resize((size_t)n + 1);
        avgi.resize((size_t)n + 1);
        avgq.resize((size_t)n + 1);
        int ncol = 0;
        if (jsdr_phase_columns(static_cast<float *>(dpy), n, bx, pix.data(), avgi.data(), avgq.data(), n + 1, &ncol) !=
            JSDR_OK)
            jsdr_throw("phase.columnMeans");
        pix.resize((size_t)ncol);
        avgi.resize((size_t)ncol);
        avgq.resize((size_t)ncol);
        return ncol;
    }

  private:
    IAudio *audio = nullptr;
    int n = 0;
    void *dpy = nullptr, *dmax = nullptr;
};

}  // namespace java_sdr
