// harness.cpp -- headless stand-in for JavaAudio.run (JavaAudio.java:195-329) that drives the C++ plugin
// mirror exactly as the reference's audio thread drives its handlers: read blen bytes, hand them to the
// raw handlers, convert int16 -> float with the I/Q corrections (:276-293), hand the float frame to every
// IAudioHandler in registration order.  A RIFF/WAVE recording is accepted like JavaAudio.openFile does
// (:369-395: PCM-16, 2 channels, the configured rate -- otherwise "Incompatible audio format"); anything else
// is a headerless dump (recorder.java:66-74).
//   usage: jsdr_harness <file.raw|file.wav> [rate=96000] [blen=8192] [ic] [qc]
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include "jsdr_plugins.hpp"

using namespace java_sdr;

struct Bus : IPublish {
    std::map<std::string, PublishValue> vals;
    std::vector<IPublishListener *> ls;
    void setPublish(const std::string &k, const PublishValue &v) override
    {
        vals[k] = v;
        for (auto *l : ls) l->notify(k, v);
    }
    void listen(IPublishListener *l) override { ls.push_back(l); }
    void unlisten(IPublishListener *l) override { ls.erase(std::remove(ls.begin(), ls.end(), l), ls.end()); }
};

struct Cfg : IConfig {
    std::map<std::string, int> m;
    int getIntConfig(const std::string &k, int d) override
    {
        auto it = m.find(k);
        if (it == m.end()) {
            m[k] = d;  // jsdr.getIntConfig writes the default back (jsdr.java:87-95)
            return d;
        }
        return it->second;
    }
    void setIntConfig(const std::string &k, int v) override { m[k] = v; }
};

struct FileAudio : IAudio {
    AudioDescriptor ad;
    int ic, qc;
    std::vector<IAudioHandler *> hands;
    std::vector<IRawHandler *> raws;
    FileAudio(int rate, int blen, int ic_, int qc_) : ad(rate, 16, 2, 4, blen), ic(ic_), qc(qc_) {}
    AudioDescriptor getAudioDescriptor() override { return ad; }
    int getICorrection() override { return ic; }
    int getQCorrection() override { return qc; }
    void addHandler(IAudioHandler *h) override { hands.push_back(h); }
    void remHandler(IAudioHandler *h) override { hands.erase(std::remove(hands.begin(), hands.end(), h), hands.end()); }
    void addRawHandler(IRawHandler *h) override { raws.push_back(h); }
    void remRawHandler(IRawHandler *h) override { raws.erase(std::remove(raws.begin(), raws.end(), h), raws.end()); }
};

int main(int argc, char **argv)
{
    if (argc < 2) {
        fprintf(stderr, "usage: %s file.raw [rate] [blen] [ic] [qc]\n", argv[0]);
        return 2;
    }
    const int rate = argc > 2 ? atoi(argv[2]) : 96000, blen = argc > 3 ? atoi(argv[3]) : 8192;
    const int ic = argc > 4 ? atoi(argv[4]) : 0, qc = argc > 5 ? atoi(argv[5]) : 0;
    jsdr_recording_info info;
    if (jsdr_recording_probe(argv[1], 2, &info) != JSDR_OK) {
        fprintf(stderr, "%s\n", jsdr_last_error());
        return 2;
    }
    if (info.format == JSDR_REC_WAV && (info.encoding != 1 || info.bits != 16 || info.channels != 2 || info.rate != rate)) {
        fprintf(stderr, "Incompatible audio format: %s: encoding %d, %d Hz, %d bit, %d channel(s)\n", argv[1], info.encoding,
                info.rate, info.bits, info.channels);
        return 2;
    }
    FILE *fp = fopen(argv[1], "rb");
    if (!fp || fseek(fp, (long)info.data_offset, SEEK_SET) != 0) {
        perror(argv[1]);
        return 2;
    }
    try {
        Bus bus;
        Cfg cfg;
        FileAudio audio(rate, blen, ic, qc);
        phase ph(&cfg, &bus, &audio);
        fft ff(&cfg, &bus, &audio);
        waterfall wf(&bus, &audio, 1024, 64);
        FUNcubeBPSKDemod dem(0, &cfg, &bus, &audio);
        cfg.setIntConfig("demod-mode", demod::MODE_AM);
        cfg.setIntConfig("demod-fir-enable", 1);
        cfg.setIntConfig("demod-agc-enable", 1);
        cfg.setIntConfig("demod-filter-low", -12000);
        cfg.setIntConfig("demod-filter-high", -6000);
        demod am(&cfg, &bus, &audio);
        std::vector<uint8_t> raw((size_t)blen);
        std::vector<float> buf((size_t)2 * blen / 4);
        int frame = 0;
        while (fread(raw.data(), 1, raw.size(), fp) == raw.size()) {
            for (auto *r : audio.raws) r->receive(raw.data(), raw.size());
            const int16_t *s16 = reinterpret_cast<const int16_t *>(raw.data());
            for (size_t sn = 0; sn < buf.size(); sn += 2) {  // JavaAudio.java:279-293
                short s = s16[sn];
                s = (short)(s + (short)ic);
                buf[sn] = (float)s / (float)32767;
                s = s16[sn + 1];
                s = (short)(s + (short)qc);
                buf[sn + 1] = (float)s / (float)32767;
            }
            for (auto *h : audio.hands) h->receive(buf.data(), buf.size());
            const auto &psd = bus.vals["fft-psd"].f;
            int32_t c[JSDR_BPSK_NCOUNTERS];
            dem.counters(c);
            const uint32_t *row = wf.pixels().data();
            int bright = 0;
            for (int p = 1; p < wf.getWidth(); p++)
                if ((row[p] & 0xff) > (row[bright] & 0xff)) bright = p;
            float amax = 0, aavg = 0;
            am.levels(amax, aavg);
            long asum = 0;
            for (int16_t v : am.audioBytes()) asum += v;
            printf("frame %d fft-psd max %.4f dB @ %.1f Hz phase-max %.6f bpsk raw=%d ds=%d bit=%d fec=%d dec=%d tune=%d wf-peak-col=%d "
                   "am-max=%.9g am-avg=%.9g am-sum=%ld\n",
                   frame, psd[psd.size() - 1], psd[psd.size() - 2], ph.maxAbs(), c[0], c[1], c[2], c[3], c[4],
                   bus.vals["FUNcube0-bpsk-tune"].i, bright, amax, aavg, asum);
            frame++;
        }
    } catch (const std::exception &e) {
        // JavaAudio.java:321-323: an exception escaping a handler ends the audio thread with a status message
        fprintf(stderr, "Audio oops: %s\n", e.what());
        fclose(fp);
        return 1;
    }
    fclose(fp);
    return 0;
}
