// harness.cpp -- headless stand-in for JavaAudio.run (JavaAudio.java:195-329) that drives the C++ plugin
// mirror exactly as the reference's audio thread drives its handlers: read blen bytes, hand them to the
// raw handlers, convert int16 -> float with the I/Q corrections (:276-293), hand the float frame to every
// IAudioHandler in registration order.  A RIFF/WAVE recording is accepted like JavaAudio.openFile does
// (:369-395: PCM-16, 2 channels, the configured rate -- otherwise "Incompatible audio format"); anything else
// is a headerless dump (recorder.java:66-74).
//   usage: jsdr_harness <file.raw|file.wav> [rate=96000] [blen=8192] [ic] [qc]
//          jsdr_harness --gpus N [--streams TOTAL=8192] [--samples L=1048576] [--steps K=5] [--warmup W=2] [--psd]
//                       [--copy-gather] [--same-device]
// The second form is BASELINE config 5 behind the C ABI alone (no Python, no torch): ONE process hosts TOTAL lock-step
// demodulators (jsdr.java:479-483 hosts its demodulators in one JVM) on N GPUs through jsdr_group_* -- one host thread per
// device, contiguous shards, one ncclAllGather of the result slots per step (RCCL over xGMI).  It synthesises the
// FEC-carrying DBPSK streams on every device, times K steps, then checks on the host that every device holds the same
// gathered slots and that every stream of every shard decoded the payloads it was sent.  --copy-gather / --same-device:
// the rehearsal a one-GPU box can run (all group members on device 0, device-to-device copies in RCCL's place).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

#define CK(expr)                                                        \
    do {                                                                \
        if ((expr) != JSDR_OK) {                                        \
            fprintf(stderr, "%s: %s\n", #expr, jsdr_last_error());      \
            return 1;                                                   \
        }                                                               \
    } while (0)

static uint64_t mix64(uint64_t z)  // splitmix64 finaliser: the generators' counter hash (csrc/synth.hip)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static int run_group(int argc, char **argv)
{
    int ndev = 1, total = 8192, steps = 5, warmup = 2, flags = 0;
    long long L = 1048576;
    bool same_device = false;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto val = [&](const char *name) -> const char * {
            if (a == name && i + 1 < argc) return argv[++i];
            return nullptr;
        };
        if (const char *v = val("--gpus")) ndev = atoi(v);
        else if (const char *v = val("--streams")) total = atoi(v);
        else if (const char *v = val("--samples")) L = atoll(v);
        else if (const char *v = val("--steps")) steps = atoi(v);
        else if (const char *v = val("--warmup")) warmup = atoi(v);
        else if (a == "--psd") flags |= JSDR_GROUP_WITH_PSD;
        else if (a == "--copy-gather") flags |= JSDR_GROUP_GATHER_COPY;
        else if (a == "--same-device") same_device = true;
        else {
            fprintf(stderr, "unknown argument %s\n", a.c_str());
            return 2;
        }
    }
    const int rate = 96000, frame = 2048, sps = rate / 1200;
    const uint64_t seed = 20020109;
    if (ndev < 1 || total % ndev || L % frame) {
        fprintf(stderr, "--streams must be a multiple of --gpus, --samples of %d\n", frame);
        return 2;
    }
    const int S = total / ndev;
    std::vector<int> devs(ndev);
    for (int d = 0; d < ndev; d++) devs[d] = same_device ? 0 : d;
    jsdr_group *g = nullptr;
    CK(jsdr_group_create(&g, ndev, devs.data(), rate, frame, 12000, 0, 0, total, L, flags));
    int64_t slot_bytes = 0;
    int rccl = 0;
    CK(jsdr_group_info(g, nullptr, nullptr, &slot_bytes, &rccl));
    // inputs: per device, the bench's generator (bench.py make_inputs) for that device's shard of the global stream ids
    const int nfr = (int)((L + 5200LL * sps - 1) / (5200LL * sps)) + 1;
    std::vector<int16_t> ct(1024), st(1024);
    for (int k = 0; k < 1024; k++) {
        const double w = 2.0 * M_PI * k / 1024.0;
        ct[k] = (int16_t)std::nearbyint(3000.0 * std::cos(w));
        st[k] = (int16_t)std::nearbyint(3000.0 * std::sin(w));
    }
    const int gain = (int)std::lround(1500.0 / 37837.0 * 32768.0);
    const uint32_t phase_inc = (uint32_t)(std::llround(13200.0 / rate * 4294967296.0) & 0xffffffffu);
    std::vector<int16_t *> raw(ndev, nullptr);
    std::vector<float *> psd(ndev, nullptr);
    std::vector<std::vector<uint8_t>> payloads(ndev);
    for (int d = 0; d < ndev; d++) {
        CK(jsdr_set_device(devs[d]));
        void *pay = nullptr, *sym = nullptr, *ds = nullptr, *dct = nullptr, *dst = nullptr, *dk = nullptr;
        CK(jsdr_malloc(&pay, (size_t)S * nfr * 256));
        CK(jsdr_malloc(&sym, (size_t)S * nfr * 5200));
        CK(jsdr_malloc(&ds, (size_t)S * nfr * 5200));
        CK(jsdr_malloc(&dct, 2048));
        CK(jsdr_malloc(&dst, 2048));
        CK(jsdr_malloc(&dk, (size_t)S * 8));
        CK(jsdr_malloc((void **)&raw[d], (size_t)S * L * 4));
        if (flags & JSDR_GROUP_WITH_PSD) CK(jsdr_malloc((void **)&psd[d], (size_t)S * (L / frame) * (frame + 2) * 4));
        std::vector<uint64_t> keys(S);
        for (int s2 = 0; s2 < S; s2++) keys[s2] = mix64((seed * 0x9E3779B1ull + (uint64_t)(d * S + s2)) ^ 0xA5A5A5A5ull);
        CK(jsdr_memcpy_h2d(dct, ct.data(), 2048));
        CK(jsdr_memcpy_h2d(dst, st.data(), 2048));
        CK(jsdr_memcpy_h2d(dk, keys.data(), (size_t)S * 8));
        CK(jsdr_synth_payloads(seed, d * S, S, nfr, (uint8_t *)pay, nullptr));
        CK(jsdr_fec_encode_batch((const uint8_t *)pay, (int64_t)S * nfr, (uint8_t *)sym, nullptr));
        CK(jsdr_synth_diffsign((const uint8_t *)sym, (int64_t)nfr * 5200, S, (int8_t *)ds, nullptr));
        CK(jsdr_synth_dbpsk(raw[d], 2 * L, S, 0, L, (const int8_t *)ds, (int64_t)nfr * 5200, sps, 0, phase_inc, (const int16_t *)dct,
                            (const int16_t *)dst, gain, (const uint64_t *)dk, nullptr));
        CK(jsdr_stream_sync(nullptr));
        payloads[d].resize((size_t)S * nfr * 256);
        CK(jsdr_memcpy_d2h(payloads[d].data(), pay, payloads[d].size()));
        for (void *p : {pay, sym, ds, dct, dst, dk}) CK(jsdr_free(p));
    }
    auto step = [&]() { return jsdr_group_batch_i16(g, raw.data(), 2 * L, L, 0, 0, (flags & JSDR_GROUP_WITH_PSD) ? psd.data() : nullptr); };
    for (int i = 0; i < warmup; i++) CK(step());
    CK(jsdr_group_sync(g));
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < steps; i++) CK(step());
    CK(jsdr_group_sync(g));
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    // ---- checks on the host: every device holds the same gathered bytes; every stream decoded what it was sent
    int64_t bits_off = 0, fec_off = 0;
    int slot_bits = 0, nfec_max = 0;
    jsdr_bpsk *dem0 = nullptr;
    CK(jsdr_group_device(g, 0, nullptr, &dem0, nullptr));
    CK(jsdr_bpsk_slot_info(dem0, &slot_bytes, &bits_off, &fec_off, &slot_bits, &nfec_max));
    const size_t all = (size_t)total * (size_t)slot_bytes;
    std::vector<uint8_t> g0(all), gd(all);
    bool same = true;
    for (int d = 0; d < ndev; d++) {
        const uint8_t *p = nullptr;
        CK(jsdr_group_gathered(g, d, &p, nullptr));
        CK(jsdr_set_device(devs[d]));
        CK(jsdr_memcpy_d2h(d == 0 ? g0.data() : gd.data(), p, all));
        if (d > 0 && memcmp(g0.data(), gd.data(), all) != 0) same = false;
    }
    long long decoded = 0, none = 0, wrong = 0;
    for (int s2 = 0; s2 < total; s2++) {
        const uint8_t *slot = g0.data() + (size_t)s2 * slot_bytes;
        const int32_t *hdr = reinterpret_cast<const int32_t *>(slot);
        const int nt = std::min(hdr[1], nfec_max);
        const uint8_t *pay = payloads[s2 / S].data() + (size_t)(s2 % S) * nfr * 256;
        int good = 0;
        for (int t = 0; t < nt; t++) {
            const uint8_t *e = slot + fec_off + (size_t)t * 264;
            int32_t rc;
            memcpy(&rc, e, 4);
            if (rc < 0) continue;  // a sync hit whose decode fails: the reference returns -1 there too (FECDecoder.java:821-824)
            good++;
            bool hit = false;
            for (int f = 0; f < nfr && !hit; f++) hit = memcmp(e + 8, pay + (size_t)f * 256, 256) == 0;
            if (!hit) wrong++;
        }
        decoded += good;
        if (!good) none++;
    }
    const bool ok = same && none == 0 && wrong == 0;
    const double msps = (double)total * (double)L * steps / dt / 1e6;
    printf("{\"harness\": \"jsdr_group\", \"n_gpus\": %d, \"total_streams\": %d, \"samples_per_stream\": %lld, \"steps\": %d, "
           "\"ms_per_step\": %.4f, \"value\": %.3f, \"unit\": \"Msamples/s\", \"gather\": \"%s\", \"rccl_version\": %d, "
           "\"psd\": %s, \"every_device_holds_the_same_slots\": %s, \"decoded_frames\": %lld, \"streams_without_decoded_frame\": %lld, "
           "\"decoded_frames_matching_no_sent_payload\": %lld, \"validated\": %s}\n",
           ndev, total, L, steps, dt / steps * 1e3, msps, (flags & JSDR_GROUP_GATHER_COPY) ? "device-to-device copies" : "ncclAllGather",
           rccl, (flags & JSDR_GROUP_WITH_PSD) ? "true" : "false", same ? "true" : "false", decoded, none, wrong, ok ? "true" : "false");
    for (int d = 0; d < ndev; d++) {
        CK(jsdr_set_device(devs[d]));
        CK(jsdr_free(raw[d]));
        if (psd[d]) CK(jsdr_free(psd[d]));
    }
    CK(jsdr_group_destroy(g));
    return ok ? 0 : 1;
}

int main(int argc, char **argv)
{
    if (argc >= 2 && strncmp(argv[1], "--", 2) == 0) return run_group(argc, argv);
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
