"""GPU parity: FUNcubeBPSKDemod.java (tune mode) + FECDecoder.java, through the C ABI.

Bar (BASELINE.json north_star): slicer bit streams and FECDecoder bytes BIT-EXACT; the exact-order FP64
kernels also reproduce the matched-filter outputs (fi,fq) and every scalar state bit for bit."""
import os

import numpy as np
import pytest

import java_sdr_amd as J
import oracle_lib as O

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_vectors.npz"))
CNAMES = sorted(["cntRaw", "cntDS", "cntBit", "cntFEC", "cntDec", "dmErrBits", "dmCorr", "dmMaxCorr", "decodeOK",
                 "centreBin"])


def same_counters(g, o):
    for k in ("cntRaw", "cntDS", "cntBit", "cntFEC", "cntDec", "dmErrBits", "dmCorr", "dmMaxCorr", "decodeOK"):
        assert g[k] == o[k], (k, g[k], o[k])


def same_state(gs, os_):
    # tuPhase, vcoPhase, dmBitPhase, dmEnergyOut, energy1, energy2, (fft-mode 6,7), dmEnergy[8], dmLastIQ[2]
    for i in (0, 1, 2, 3, 4, 5, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17):
        assert gs[i] == os_[i], (i, gs[i], os_[i])


def test_bpsk_sine4410_fixture_frame_by_frame(golden_dir):
    raw = np.fromfile(os.path.join(golden_dir, "sine4410.raw"), dtype="<i2")
    buf = O.convert_i16(raw)
    d = J.Bpsk(nstreams=1)
    o = O.Bpsk(trace=1024)
    bits, tr = [], []
    for k in range(2):
        d.receive(buf[k * 4096:(k + 1) * 4096])  # IAudioHandler.receive(float[])
        o.receive(buf[k * 4096:(k + 1) * 4096])
        bits.append(d.bits().copy())
        tr.append(d.trace().copy())
    c = d.counters()
    assert (c["cntRaw"], c["cntDS"], c["cntBit"], c["cntFEC"]) == (4096, 409, 50, 0)
    assert np.array_equal(np.concatenate(bits), G["sine_bpsk0_bits"])
    assert np.array_equal(np.concatenate(tr), G["sine_bpsk0_trace"])  # (fi,fq) bit for bit
    same_counters(c, o.counters())
    same_state(d.state(), o.state())
    assert np.all(d.decoded() == 0)
    # raw (IRawHandler) form gives the same stream
    d2 = J.Bpsk(nstreams=1)
    for k in range(2):
        d2.receive_raw(raw[k * 4096:(k + 1) * 4096])
    same_state(d2.state(), o.state())


def run_both(iq_streams, nsamples, chunks, rate=96000, tuning=12000, ic=0, qc=0, do_fft=0, do_up=0, blen=8192):
    """feed S streams to the GPU in the given chunk sizes and to one oracle per stream; compare everything"""
    S = len(iq_streams)
    d = J.Bpsk(rate=rate, blen=blen, tuning=tuning, do_fft=do_fft, do_up=do_up, nstreams=S,
               max_batch_samples=max(chunks))
    stride = 2 * nsamples
    d_iq = J.DeviceBuffer.from_host(np.concatenate(iq_streams))
    oracles = [O.Bpsk(rate=rate, blen=(blen if do_fft else 4), tuning=tuning, do_fft=do_fft, do_up=do_up,
                      trace=nsamples // max(1, rate // 9600) + 8) for _ in range(S)]
    gbits = [[] for _ in range(S)]
    gtrace = [[] for _ in range(S)]
    gfec = [[] for _ in range(S)]
    pos = 0
    for L in chunks:
        d.batch_i16(d_iq.ptr + 4 * pos, stride, L, ic, qc)
        for s in range(S):
            gbits[s].append(d.bits(s).copy())
            gtrace[s].append(d.trace(s).copy())
            gfec[s].extend(d.fec_results(s))
        pos += L
    assert pos == nsamples
    for s in range(S):
        oracles[s].receive_i16(iq_streams[s], ic, qc)
        assert np.array_equal(np.concatenate(gbits[s]), oracles[s].bits()), f"stream {s}: bits differ"
        assert np.array_equal(np.concatenate(gtrace[s]), oracles[s].trace()), f"stream {s}: (fi,fq) differ"
        fo = oracles[s].fec_results()
        assert len(gfec[s]) == len(fo), (s, len(gfec[s]), len(fo))
        for (rc, _, data), (orc, _, odata) in zip(gfec[s], fo):
            assert rc == orc and np.array_equal(data, odata)
        same_counters(d.counters(s), oracles[s].counters())
        same_state(d.state(s), oracles[s].state())
        if do_fft:
            assert d.counters(s)["centreBin"] == oracles[s].counters()["centreBin"]
            assert d.state(s)[6] == oracles[s].state()[6] and d.state(s)[7] == oracles[s].state()[7]
        assert np.array_equal(d.decoded(s), oracles[s].decoded())
    return d, oracles


def test_bpsk_dbpsk_streams_one_batch_bit_exact():
    n = 458752
    streams = [O.make_dbpsk_stream(20020109, s, n, noise_sigma=1500.0 + 700 * s)[0] for s in range(3)]
    d, oracles = run_both(streams, n, [n])
    assert all(o.counters()["cntFEC"] >= 1 for o in oracles)
    # golden vector produced in the build container (stream 3 of the same family)
    iq3, pay3, _ = O.make_dbpsk_stream(20020109, 3, n)
    d3, _ = run_both([iq3], n, [n])
    assert np.array_equal(d3.bits(0), G["dbpsk_bits"])
    assert np.array_equal(d3.fec_results(0)[0][2], pay3[0])


@pytest.mark.parametrize("chunks", [
    [2048] * 40,                                   # the reference's frame cadence
    [77, 1, 2048, 4099, 13, 65536, 9, 10, 11, 40000, 20000],  # ragged, incl. calls that yield 0 outputs
    [131072 - 26, 26],
])
def test_bpsk_state_carries_across_ragged_calls(chunks):
    n = sum(chunks)
    streams = [O.make_dbpsk_stream(7, s, n, noise_sigma=900.0)[0] for s in range(2)]
    run_both(streams, n, chunks)


@pytest.mark.parametrize("rate,tuning", [(192000, 12000), (48000, 9000), (44100, 8000), (96000, 10000),
                                         (96000, -5000), (96000, 0)])
def test_bpsk_other_rates_and_tunings(rate, tuning):
    n = 60000 * (rate // 9600) // 10
    rng = np.random.default_rng(rate + tuning)
    carrier = (tuning if tuning > 0 else 0) + 1200.0
    iq, _, _ = O.make_dbpsk_stream(11, 0, n, rate=rate, carrier_hz=carrier, noise_sigma=500.0)
    noise = rng.integers(-20000, 20000, 2 * n).astype(np.int16)
    run_both([iq, noise], n, [n // 3, n - n // 3], rate=rate, tuning=tuning)


def test_bpsk_dc_correction_is_applied_like_javaaudio():
    n = 40960
    iq, _, _ = O.make_dbpsk_stream(5, 0, n)
    run_both([iq], n, [n], ic=1234, qc=-4321)
    run_both([iq], n, [n], ic=40000, qc=-40000)  # (short) cast of the correction wraps
    # across calls: the 26 history samples are kept corrected, the first windows of a call read them through the
    # edge image with the correction taken off and re-applied (k_fm_edges), also for calls shorter than a window
    run_both([iq], n, [n // 3, 7, 30, n - n // 3 - 37], ic=1234, qc=-4321)
    run_both([iq], n, [4096, 1, 20000, n - 24097], ic=-32768, qc=32767)


@pytest.mark.parametrize("seed", range(16))
def test_bpsk_random_call_patterns_rates_and_corrections(seed):
    """seeded sweep over what decides which samples a call's first and last windows see: call lengths around the
    window sizes (26 / 27-tap window, 57-sample lane window, 60-sample quad span, the 64-sample halo), every rate,
    periodic and non-periodic tunings, DC correction on and off -- bits, (fi,fq), state and FEC against the oracle"""
    rng = np.random.default_rng(1000 + seed)
    rate, tunings = [(96000, [12000, 24000, 10000, 0]), (48000, [12000, 9000]), (44100, [8000]),
                     (192000, [12000, 24000])][seed % 4]
    tuning = tunings[(seed // 4) % len(tunings)]
    n = int(rng.integers(50000, 90000)) * (rate // 9600) // 10
    small = [1, 2, 9, 10, 11, 25, 26, 27, 28, 39, 40, 41, 56, 57, 58, 59, 60, 61, 63, 64, 65, 79, 80, 81, 127, 128, 640, 641]
    chunks, left = [], n
    while left > 0:
        L = int(rng.choice(small)) if rng.random() < 0.6 else int(rng.integers(1000, 30000))
        L = min(L, left)
        chunks.append(L)
        left -= L
    carrier = (tuning if tuning > 0 else 0) + 1200.0
    iq, _, _ = O.make_dbpsk_stream(70 + seed, 0, n, rate=rate, carrier_hz=carrier, noise_sigma=700.0)
    noise = rng.integers(-30000, 30000, 2 * n).astype(np.int16)
    ic, qc = (0, 0) if seed % 2 == 0 else (int(rng.integers(-40000, 40000)), int(rng.integers(-40000, 40000)))
    run_both([iq, noise], n, chunks, rate=rate, tuning=tuning, ic=ic, qc=qc)


def test_bpsk_fec_errors_and_failed_decode_keep_previous_payload():
    """two frames; the second is corrupted beyond repair -> rc=-1, decoded[] keeps frame 0's bytes"""
    n = 2 * 416000 + 30000
    flips = list(range(5200 + 100, 5200 + 5100, 2))  # every other symbol of frame 1
    iq, pay, _ = O.make_dbpsk_stream(31, 0, n, flips=flips, noise_sigma=400.0)
    d, oracles = run_both([iq], n, [n // 2, n - n // 2])
    fo = oracles[0].fec_results()
    assert [r[0] >= 0 for r in fo][:1] == [True]
    assert np.array_equal(d.decoded(0), oracles[0].decoded())


def test_bpsk_result_slots_for_the_all_gather():
    n = 458752
    streams = [O.make_dbpsk_stream(20020109, s, n)[0] for s in range(2)]
    d, oracles = run_both(streams, n, [n])
    from java_sdr_amd import sharding as SH
    info = d.slot_info()
    assert info == SH.slot_layout(info["slot_bits"], info["nfec_max"])  # kernel layout == the documented one
    slots = J.DeviceBuffer(2 * info["slot_bytes"])
    d.pack_slots(slots)
    raw = slots.to_host(np.uint8)
    for s in range(2):
        blob = raw[s * info["slot_bytes"]:(s + 1) * info["slot_bytes"]]
        u = SH.unpack_slot(blob, info)
        assert np.array_equal(u["bits"], oracles[s].bits())
        fo = oracles[s].fec_results()
        assert len(u["fec"]) == len(fo)
        assert u["fec"][0][0] == fo[0][0] and np.array_equal(u["fec"][0][2], fo[0][2])
        c = oracles[s].counters()
        counters = [c[k] for k in ("cntRaw", "cntDS", "cntBit", "cntFEC", "cntDec", "dmErrBits", "dmCorr", "dmMaxCorr",
                                   "decodeOK")]
        want = SH.pack_slot(info, counters, oracles[s].bits(), [(rc, u["fec"][i][1], dat) for i, (rc, _, dat) in enumerate(fo)])
        assert np.array_equal(blob, want)  # byte for byte what the numpy statement of the layout produces


def test_bpsk_roundtrip_property_at_baseline_batch_shape():
    """BASELINE config 4 shape scaled to 64 streams x 1,048,576 samples, generated on the device: every
    stream must decode the payloads it was sent (encode -> modulate -> demodulate -> decode round trip),
    and two sampled streams must match the oracle bit for bit."""
    S, L, sps = 64, 1048576, 80
    nfr = 3
    seed = 20020109
    pay = J.synth_payloads(seed, 0, S, nfr)
    d_sym = J.DeviceBuffer(S * nfr * 5200)
    J.fec_encode_dev(pay, S * nfr, d_sym)
    d_ds = J.DeviceBuffer(S * nfr * 5200)
    J.synth_diffsign(d_sym, nfr * 5200, S, d_ds)
    ct, st = O.synth_tables(3000)
    keys = np.array([O.mix64((seed * 0x9E3779B1 + s) ^ 0xA5A5A5A5) for s in range(S)], np.uint64)
    d_iq = J.DeviceBuffer(S * L * 4)
    gain = int(round(1500.0 / 37837.0 * 32768.0))
    J.synth_dbpsk(d_iq, 2 * L, S, 0, L, d_ds, nfr * 5200, sps, 0, O.phase_inc_u32(13200.0, 96000),
                  J.DeviceBuffer.from_host(ct), J.DeviceBuffer.from_host(st), gain, J.DeviceBuffer.from_host(keys))
    d = J.Bpsk(nstreams=S, max_batch_samples=L)
    d.batch_i16(d_iq, 2 * L, L)
    payloads = pay.to_host(np.uint8).reshape(S, nfr, 256)
    for s in range(S):
        fr = d.fec_results(s)
        assert len(fr) == 2, (s, len(fr))
        for k, (rc, _, data) in enumerate(fr):
            assert rc >= 0 and np.array_equal(data, payloads[s, k]), (s, k, rc)
    for s in (0, 63):
        iq = d_iq.to_host(np.int16, count=2 * L, offset_bytes=4 * L * s)
        ref_iq, _, _ = O.make_dbpsk_stream(seed, s, L, nframes=nfr)
        assert np.array_equal(iq, ref_iq)  # device generator == host generator
        o = O.Bpsk()
        o.receive_i16(iq)
        assert np.array_equal(d.bits(s), o.bits())
        same_counters(d.counters(s), o.counters())


# ------------------------------------------------------------------ FFT-acquire mode (doBufferFFT, :406-464)
def test_bpsk_fft_mode_sine4410_fixture(golden_dir):
    raw = np.fromfile(os.path.join(golden_dir, "sine4410.raw"), dtype="<i2")
    buf = O.convert_i16(raw)
    d = J.Bpsk(nstreams=1, do_fft=1)
    o = O.Bpsk(do_fft=1, trace=1024)
    bits, tr, cb = [], [], []
    for k in range(2):
        d.receive(buf[k * 4096:(k + 1) * 4096])
        o.receive(buf[k * 4096:(k + 1) * 4096])
        bits.append(d.bits().copy())
        tr.append(d.trace().copy())
        cb.append(d.counters()["centreBin"])
    assert cb == [211, 200]  # SURVEY 8c behavioural KAT
    assert d.counters()["cntBit"] == 51
    assert np.array_equal(np.concatenate(bits), G["sine_bpsk1_bits"])
    assert np.array_equal(np.concatenate(tr), G["sine_bpsk1_trace"])
    same_counters(d.counters(), o.counters())
    same_state(d.state(), o.state())


@pytest.mark.parametrize("do_up,carrier", [(0, 13200.0), (0, 6000.0), (1, 13200.0), (1, 30000.0)])
def test_bpsk_fft_mode_streams_bit_exact(do_up, carrier):
    n = 2048 * 120
    rng = np.random.default_rng(int(carrier) + do_up)
    streams = [O.make_dbpsk_stream(41, s, n, carrier_hz=carrier, noise_sigma=600.0 + 900 * s)[0] for s in range(2)]
    streams.append(rng.integers(-12000, 12000, 2 * n).astype(np.int16))  # noise only: centre bin wanders
    run_both(streams, n, [2048 * 7, 2048, 2048 * 100, 2048 * 12], do_fft=1, do_up=do_up)


def test_bpsk_fft_mode_other_frame_sizes_and_rates():
    for blen, rate in ((4096, 96000), (16384, 192000), (32768, 96000)):
        nsf = blen // 4
        n = nsf * 24
        iq = O.make_dbpsk_stream(43, 0, n, rate=rate, carrier_hz=13200.0, noise_sigma=700.0)[0]
        run_both([iq], n, [nsf * 5, nsf * 19], rate=rate, do_fft=1, blen=blen)


def test_bpsk_fft_mode_at_the_reference_default_frames():
    """bpsk-dofft with java-sdr's own default buffers: blen = rate*size/10 -> n = 9600 @96 kHz, 4800 @48 kHz
    (JavaAudio.java:58-59): the mixed-radix exact-order FP64 transform of bpsk_fftm.hip against the oracle's"""
    for blen, rate, do_up, carrier in ((38400, 96000, 0, 13200.0), (38400, 96000, 1, 31200.0), (19200, 48000, 0, 7300.0)):
        nsf = blen // 4
        n = nsf * 26
        iq = O.make_dbpsk_stream(77, 0, n, rate=rate, carrier_hz=carrier, noise_sigma=900.0)[0]
        iq2 = O.make_dbpsk_stream(77, 1, n, rate=rate, carrier_hz=carrier + 350.0, noise_sigma=400.0)[0]
        d, oracles = run_both([iq, iq2], n, [nsf * 7, nsf * 19], rate=rate, do_fft=1, do_up=do_up, blen=blen)
        assert oracles[0].counters()["centreBin"] > 102  # the carrier was acquired, not the clamp value
    # float frames through receive(), one frame at a time
    d = J.Bpsk(rate=96000, blen=38400, nstreams=1, do_fft=1)
    o = O.Bpsk(rate=96000, blen=38400, do_fft=1, trace=4096)
    buf = O.convert_i16(O.make_dbpsk_stream(78, 0, 9600 * 3, carrier_hz=12800.0)[0])
    tr = []
    for k in range(3):
        d.receive(buf[k * 19200:(k + 1) * 19200])
        o.receive(buf[k * 19200:(k + 1) * 19200])
        tr.append(d.trace().copy())
    assert np.array_equal(np.concatenate(tr), o.trace())
    same_counters(d.counters(), o.counters())
    same_state(d.state(), o.state())


def test_bpsk_fft_mode_at_the_fcd_pro_plus_default_frame_19200():
    """bpsk-dofft with java-sdr's default buffer at 192 kHz: blen = rate*size/10 = 76800 -> n = 19200 (JavaAudio.java:58-59),
    the FUNcube Dongle Pro+ frame.  307 KB in FP64: the transform is two 9600-point halves behind one radix-2 pass
    (k_front_fft2x), the order the oracle defines for n > 9600 -- bit-identical centre bins, traces, bits."""
    nsf, rate = 19200, 192000
    n = nsf * 14
    for do_up, carrier in ((0, 13200.0), (1, 61000.0)):
        iq = O.make_dbpsk_stream(91, 0, n, rate=rate, carrier_hz=carrier, noise_sigma=800.0)[0]
        iq2 = O.make_dbpsk_stream(91, 1, n, rate=rate, carrier_hz=carrier + 410.0, noise_sigma=300.0)[0]
        rng = np.random.default_rng(int(carrier))
        noise = rng.integers(-9000, 9000, 2 * n).astype(np.int16)
        d, oracles = run_both([iq, iq2, noise], n, [nsf * 3, nsf, nsf * 10], rate=rate, do_fft=1, do_up=do_up, blen=4 * nsf)
        assert oracles[0].counters()["centreBin"] > 102
    # float frames through receive(), one frame at a time; and 15360 = 2 x 7680 through the same kernel
    d = J.Bpsk(rate=rate, blen=4 * nsf, nstreams=1, do_fft=1)
    o = O.Bpsk(rate=rate, blen=4 * nsf, do_fft=1, trace=8192)
    buf = O.convert_i16(O.make_dbpsk_stream(92, 0, nsf * 3, rate=rate, carrier_hz=12900.0)[0])
    tr = []
    for k in range(3):
        d.receive(buf[k * 2 * nsf:(k + 1) * 2 * nsf])
        o.receive(buf[k * 2 * nsf:(k + 1) * 2 * nsf])
        tr.append(d.trace().copy())
    assert np.array_equal(np.concatenate(tr), o.trace())
    same_counters(d.counters(), o.counters())
    same_state(d.state(), o.state())
    iq = O.make_dbpsk_stream(93, 0, 15360 * 6, rate=rate, carrier_hz=13000.0, noise_sigma=500.0)[0]
    run_both([iq], 15360 * 6, [15360 * 2, 15360 * 4], rate=rate, do_fft=1, blen=4 * 15360)


@pytest.mark.parametrize("blen,rate,do_up,carrier", [(38400, 96000, 0, 23300.0), (38400, 96000, 1, 47100.0), (38400, 96000, 1, 24300.0),
                                                     (19200, 48000, 1, 23300.0), (19200, 48000, 0, 11300.0),
                                                     (76800, 192000, 0, 47300.0), (76800, 192000, 1, 95100.0)])
def test_bpsk_fft_mode_default_frames_carrier_at_the_band_edges(blen, rate, do_up, carrier):
    """round 3's pruned passes (bpsk_fftm.hip: fm_pass5_band forms only the bins below end + 102, fm_inv_blocks builds the
    inverse's first three passes from the 204 gathered bins): a carrier near the edges of the searched band puts the
    gathered bins at the rim of what is formed; a noise-only stream lets the centre bin wander; every trace, state double
    and bit must still equal the oracle's"""
    nsf = blen // 4
    n = nsf * 12
    iq = O.make_dbpsk_stream(83, 0, n, rate=rate, carrier_hz=carrier, noise_sigma=500.0)[0]
    rng = np.random.default_rng(int(carrier) + do_up)
    noise = rng.integers(-15000, 15000, 2 * n).astype(np.int16)
    d, oracles = run_both([iq, noise], n, [nsf * 5, nsf * 7], rate=rate, do_fft=1, do_up=do_up, blen=blen)
    # the carrier at the rim of the band was acquired (the boxcar search stops 75 bins short of the band's edges, :433-443)
    assert abs(oracles[0].counters()["centreBin"] - carrier / (rate / nsf)) <= 60


@pytest.mark.parametrize("nsf,rate", [(9600, 48000), (9600, 192000), (19200, 96000), (19200, 48000), (4800, 96000), (4800, 192000),
                                      (2048, 48000), (2048, 192000), (4096, 48000),
                                      # round 4: frames with factors of 7 and frames that are not multiples of 16 -- 4410 is what a
                                      # 44.1 kHz sound card delivers (JavaAudio.java:59; the reference's own sine4410.wav), 8820 the
                                      # same at 88.2 kHz (decimation 9), 3430 = 2.5.7^3, 7000 = 2^3.5^3.7, 2646 = 2.3^3.7^2
                                      (4410, 44100), (8820, 88200), (3430, 48000), (7000, 96000), (2646, 44100),
                                      # ... and rates below 38.4 kHz (decimation 3, 2, 1), which only the mixed-radix front end takes
                                      (3200, 32000), (2205, 22050), (1200, 9600),
                                      # round 5: ANY frame of 416 .. 9600 samples -- prime factors above 7 through the pass that is
                                      # the DFT's definition (11.025 kHz: 1102 = 2.19.29; 1100 = 4.5.5.11; 1103 is prime; 3146 =
                                      # 2.11.11.13 at a decimation of 3), and frames below 1024 samples (8 kHz: 800; 6 kHz: 600, where
                                      # the boxcar range is empty, as in the reference's loop :433; 418 = 2.11.19)
                                      (1102, 11025), (1100, 11000), (1103, 11030), (3146, 31460), (800, 8000), (600, 9600), (418, 9600)])
def test_bpsk_fft_mode_frames_at_other_decimations(nsf, rate):
    """the front ends' RxDownSample reads the inverse's real samples as a compact array (round 3): 14 aligned 16-byte reads
    per window at an even decimation, single reads at an odd one (48 kHz: 5), two aligned runs at n = 19200 when the
    decimation is a multiple of 4 (192 kHz: 20) and the general loop otherwise -- every frame size at the sample rates that
    are NOT its default, chunked so that calls begin at different frames"""
    n = nsf * 10
    iq = O.make_dbpsk_stream(85, 0, n, rate=rate, carrier_hz=rate / 8.0 + 333.0, noise_sigma=600.0)[0]
    rng = np.random.default_rng(nsf + rate)
    noise = rng.integers(-11000, 11000, 2 * n).astype(np.int16)
    run_both([iq, noise], n, [nsf * 3, nsf, nsf * 6], rate=rate, do_fft=1, blen=4 * nsf)


def test_bpsk_fft_mode_other_mixed_radix_frames():
    """frames of 2^a 3^b 5^c samples other than the two defaults go through the run-time (unspecialised) Stockham passes
    and pass pairs: 7680 = 4.4.4.4.2.3.5 -> [4][4,4][4,2][3,5], 1440 = 4.4.2.3.3.5 -> [4][4,2][3][3,5], 1200 = 4.4.3.5.5"""
    for nsf in (7680, 1440, 1200):
        n = nsf * 12
        iq = O.make_dbpsk_stream(79, 0, n, carrier_hz=13100.0, noise_sigma=700.0)[0]
        iq2 = O.make_dbpsk_stream(79, 1, n, carrier_hz=12650.0, noise_sigma=300.0)[0]
        run_both([iq, iq2], n, [nsf * 5, nsf * 7], do_fft=1, blen=4 * nsf)


def test_bpsk_fft_mode_negative_zero_spectrum_bins():
    """float frames of -0.0 (with a few impulses) leave -0.0 in spectrum bins: there a butterfly with a zero second
    operand does NOT return its first operand ((-0)+(+0) = +0), so the broadcast first inverse pass of
    k_front_fft must take its full-arithmetic branch -- compared including the sign of zeros"""
    rng = np.random.default_rng(5)
    d = J.Bpsk(nstreams=1, do_fft=1)
    o = O.Bpsk(blen=8192, do_fft=1, trace=4096)
    gt = []
    for k in range(5):
        buf = np.full(4096, -0.0, np.float32)
        if k in (1, 2, 4):
            buf[rng.integers(0, 4096, 3)] = (rng.standard_normal(3) * 0.3).astype(np.float32)
        d.receive(buf)
        o.receive(buf)
        gt.append(d.trace().copy())
    gt = np.concatenate(gt)
    ot = o.trace()
    assert np.array_equal(gt, ot)
    assert np.array_equal(np.signbit(gt), np.signbit(ot))
    same_counters(d.counters(), o.counters())
    same_state(d.state(), o.state())


def test_bpsk_fft_mode_rejects_partial_frames_and_odd_sizes():
    d = J.Bpsk(nstreams=1, do_fft=1, max_batch_samples=8192)
    buf = J.DeviceBuffer(4 * 8192)
    buf.zero()
    with pytest.raises(J.JsdrError):
        d.batch_i16(buf, 2 * 8192, 3000)
    with pytest.raises(J.JsdrError):
        J.Bpsk(nstreams=1, do_fft=1, blen=4 * 400)  # below 416 samples the 204 gathered bins would not fit the frame (the reference's arraycopy :458 would throw)
    with pytest.raises(J.JsdrError):
        J.Bpsk(nstreams=1, do_fft=1, blen=4 * 256)
    with pytest.raises(J.JsdrError):
        J.Bpsk(nstreams=1, do_fft=1, blen=4 * 200006, rate=2000060)  # 2 x 100003 (prime): a pass of n r = 2e10 multiply-adds a frame


@pytest.mark.parametrize("nsf,rate,do_up", [
    (512, 48000, 0), (512, 5120, 1),  # powers of two below 1024 (decimations 5 and 1)
    (16384, 163840, 0), (32768, 96000, 1),  # ... and above 8192
    (17640, 176400, 0),  # a 176.4 kHz card's frame: 2 x 8820, 7^2 | 8820
    (38400, 384000, 1),  # a 384 kHz card's
    (9614, 96140, 0),  # 2 . 11 . 19 . 23: above 9600 samples with prime radices above 7
    (11025, 110250, 0),  # odd: 3^2 5^2 7^2
    (2048, 19200, 0), (4096, 32000, 1), (19200, 28800, 0),  # the LDS front ends' own frames at decimations they do not take (2, 3, 3)
])
def test_bpsk_fft_mode_any_frame(nsf, rate, do_up):
    """Round 6: the frames no LDS front end takes -- powers of two outside 1024 .. 8192, frames above 9600 samples other than twice a
    16 | m, 2^a 3^b 5^c frame, power-of-two / 2 m frames below 38.4 kHz -- through the any-frame passes (bpsk_acqg.hip: the oracle's
    transform one launch per pass, the image in global memory) inside the three-phase front end: bits, traces, FEC, counters and
    state against the oracle, chunked so that calls begin at different frames (one call of a single frame among them)"""
    n = nsf * 10
    carrier = rate * (0.375 if do_up else 0.125) + 333.0
    iq = O.make_dbpsk_stream(86, 0, n, rate=rate, carrier_hz=carrier, noise_sigma=600.0)[0]
    rng = np.random.default_rng(nsf + rate)
    noise = rng.integers(-11000, 11000, 2 * n).astype(np.int16)
    d, oracles = run_both([iq, noise], n, [nsf * 3, nsf, nsf * 6], rate=rate, do_fft=1, do_up=do_up, blen=4 * nsf)
    assert d.front_kernel_name() == "k_acqg_pass"
    if nsf >= 604:
        assert oracles[0].counters()["centreBin"] > 102  # the carrier was acquired, not the clamp value


def test_bpsk_fft_mode_any_frame_receive_float_and_int16_frames():
    """the IAudioHandler form through the any-frame passes: one frame a receive(), float and int16 frames in turn (n = 512 and 17640)"""
    for nsf, rate in ((512, 48000), (17640, 176400)):
        nfr = 8
        iq = O.make_dbpsk_stream(94, 0, nsf * nfr, rate=rate, carrier_hz=rate / 8.0 + 150.0, noise_sigma=600.0)[0]
        exact = O.convert_i16(iq)
        d = J.Bpsk(nstreams=1, do_fft=1, rate=rate, blen=4 * nsf)
        o = O.Bpsk(do_fft=1, rate=rate, blen=4 * nsf, trace=nsf * nfr // max(1, rate // 9600) + 8)
        bits, tr = [], []
        for k in range(nfr):
            fr = exact[2 * nsf * k:2 * nsf * (k + 1)].copy()
            if k % 3 == 1:
                d.receive(fr)
            else:
                d.receive_raw(iq[2 * nsf * k:2 * nsf * (k + 1)])
            o.receive(fr)
            bits.append(d.bits().copy())
            tr.append(d.trace().copy())
        assert np.array_equal(np.concatenate(tr), o.trace())
        assert np.array_equal(np.concatenate(bits), o.bits())
        same_counters(d.counters(), o.counters())
        same_state(d.state(), o.state())


def _largest_prime_factor(n):
    p, best = 2, 1
    while p * p <= n:
        while n % p == 0:
            best, n = p, n // p
        p += 1
    return max(best, n) if n > 1 else best


@pytest.mark.parametrize("seed", range(10))
def test_bpsk_fft_mode_any_frame_random_sizes(seed, monkeypatch):
    """random frames of 416 .. 24000 samples at random rates, both band halves, ragged call patterns, all through the any-frame passes
    (JSDR_ACQG=1 also where an LDS front end would take the frame): whatever radix plan the frame has -- powers of two, 2 m, prime
    radices up to 1500 (the oracle's O(n r) pass bounds the test's size, not the kernel) -- against the oracle"""
    monkeypatch.setenv("JSDR_ACQG", "1")
    rng = np.random.default_rng(7000 + seed)
    while True:
        nsf = int(rng.integers(416, 24001)) if seed % 3 else 1 << int(rng.integers(9, 15))
        if _largest_prime_factor(nsf) <= 1500:
            break
    rate = int(rng.choice([8000, 9600, 11025, 22050, 32000, 44100, 48000, 96000, 192000, 250000, 384000]))
    do_up = int(rng.integers(0, 2))
    nfr = int(rng.integers(5, 9))
    cuts = sorted(set(int(c) for c in rng.integers(1, nfr, 2)))
    chunks = [b - a for a, b in zip([0] + cuts, cuts + [nfr])]
    n = nsf * nfr
    iq = O.make_dbpsk_stream(300 + seed, 0, n, rate=rate, carrier_hz=rate * (0.36 if do_up else 0.11) + 97.0, noise_sigma=500.0)[0]
    noise = rng.integers(-9000, 9000, 2 * n).astype(np.int16)
    d, _ = run_both([iq, noise], n, [c * nsf for c in chunks], rate=rate, do_fft=1, do_up=do_up, blen=4 * nsf)
    assert d.front_kernel_name() == "k_acqg_pass"


def test_bpsk_fft_mode_suite_through_the_any_frame_passes():
    """every FFT-mode test once more in a child process with JSDR_ACQG=1: the frames the LDS front ends take (2^k, 9600 / 4800 / 4410,
    19200, the prime-radix ones, -0.0 bins, carriers at the band edges, the fixtures) through the any-frame passes instead -- two
    independent implementations of the oracle's transform against the same expectations"""
    if os.environ.get("JSDR_ACQG") is not None or os.environ.get("JSDR_ACQ3") is not None:
        return
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, JSDR_ACQG="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", "-m", "gpu",
                        os.path.join(here, "test_gpu_bpsk.py"), os.path.join(here, "test_gpu_fixtures.py"),
                        "-k", "fft and not either_front_end and not eight_streams_per_wave and not any_frame"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_bpsk_api_errors():
    with pytest.raises(J.JsdrError):
        J.Bpsk(rate=0)
    d = J.Bpsk(nstreams=2)
    with pytest.raises(J.JsdrError):
        d.receive(np.zeros(4096, np.float32))  # receive() is the 1-stream form
    with pytest.raises(J.JsdrError):
        d.batch_i16(0, 8192, 2048)  # null input
    with pytest.raises(J.JsdrError):
        d.bits(5)


# ------------------------------------------------------------------ result log: concurrent pack, capacity
def test_bpsk_pack_on_a_side_stream_is_ordered_against_the_next_call():
    """k_pack_slots of call k runs on the caller's gather stream; the tail / sync / FEC of call k+1 (the
    handle's side stream) rewrite the same single-buffered result arrays.  Two back-to-back calls with a pack after
    each, on a stream of its own, never synchronising in between: each slot must equal that call's own results."""
    from java_sdr_amd import sharding as SH
    n = 458752
    S = 4
    streams = [O.make_dbpsk_stream(20020109, s, 2 * n, noise_sigma=800.0)[0] for s in range(S)]
    d_iq = J.DeviceBuffer.from_host(np.concatenate(streams))
    # reference run: one call at a time, synchronised, getters after each
    ref = J.Bpsk(nstreams=S, max_batch_samples=n)
    want = []
    for k in range(2):
        ref.batch_i16(d_iq.ptr + 4 * n * k, 4 * n, n)
        want.append([(ref.bits(s).copy(), [(r[0], r[2].copy()) for r in ref.fec_results(s)], ref.counters(s)) for s in range(S)])
    # overlapped run
    d = J.Bpsk(nstreams=S, max_batch_samples=n)
    info = d.slot_info()
    side = J.Stream()
    slots = [J.DeviceBuffer(S * info["slot_bytes"]) for _ in range(2)]
    for k in range(2):
        d.batch_i16(d_iq.ptr + 4 * n * k, 4 * n, n)
        d.pack_slots(slots[k], stream=side.ptr)
    side.sync()
    d.sync()
    for k in range(2):
        blob = slots[k].to_host(np.uint8).reshape(S, info["slot_bytes"])
        for s in range(S):
            u = SH.unpack_slot(blob[s], info)
            bits, fec, cnt = want[k][s]
            assert np.array_equal(u["bits"], bits), (k, s)
            assert [(r[0], bytes(r[2])) for r in u["fec"]] == [(rc, bytes(dat)) for rc, dat in fec], (k, s)
            assert int(u["header"][4]) == cnt["cntBit"] and int(u["header"][5]) == cnt["cntFEC"], (k, s)


def test_bpsk_more_sync_hits_than_the_log_holds_is_flagged_not_truncated_silently():
    """the reference has no limit on FECDecode calls per frame (:560-569); the handle sizes its log from
    max_batch_samples and must (a) keep the FIRST hits, deterministically, (b) count every hit in cntFEC, (c) flag the
    stream so that no getter returns a truncated log as success.  A symbol stream in which every sync bit is held for
    80 symbols correlates at 80 consecutive bit positions per 5200-bit window."""
    from java_sdr_amd import sharding as SH
    sync = (O.bpsk_table(2) > 0).astype(np.uint8)  # SYNC_VECTOR as 1/0
    nsym = 3 * 5200
    sym = np.repeat(sync, 80)[np.arange(nsym) % 5200]
    dsign = O.synth_diffsign(sym, 1)
    ct, st = O.synth_tables(3000)
    n = nsym * 80
    iq = O.synth_dbpsk(0, n, dsign, 80, 0, O.phase_inc_u32(13200.0, 96000), ct, st, 300, O.mix64(99))
    o = O.Bpsk()
    o.receive_i16(iq)
    oc = o.counters()
    d = J.Bpsk(nstreams=1, max_batch_samples=n)
    cap = d.slot_info()["nfec_max"]
    assert oc["cntFEC"] > cap  # the oracle ran FECDecode far more often than the log holds
    d.batch_i16(J.DeviceBuffer.from_host(iq), 2 * n, n)
    info = d.slot_info()
    slots = J.DeviceBuffer(info["slot_bytes"])
    d.pack_slots(slots)
    u = SH.unpack_slot(slots.to_host(np.uint8), info)
    assert int(u["header"][11]) == 1  # overflow flag travels with the slot
    assert int(u["header"][5]) == oc["cntFEC"]  # every hit counted
    assert [r[1] for r in u["fec"]] == [r[1] for r in o.fec_results()[:cap]]  # the first `cap` hits, in order
    for getter in (d.bits, d.fec_results, d.decoded, d.counters):
        with pytest.raises(J.JsdrError):
            getter(0)


def test_bpsk_schedule_look_ahead_for_non_periodic_tuning():
    """a tuning whose phase never repeats (12345 Hz at 192 kHz) needs a fresh tuner / VCO schedule
    every call.  The schedule of call k+1 is stepped on a worker thread while the GPU runs call k; results stay
    bit-identical to the oracle, with and without the look-ahead."""
    rate, n, chunk = 192000, 4 * 131072, 131072
    iq = O.make_dbpsk_stream(61, 0, n, rate=rate, carrier_hz=13545.0, noise_sigma=600.0)[0]
    noise = np.random.default_rng(3).integers(-15000, 15000, 2 * n).astype(np.int16)
    d, _ = run_both([iq, noise], n, [chunk] * 4, rate=rate, tuning=12345)
    st = d.schedule_stats()
    assert st["prefetched"] == 3 and st["computed_inline"] == 1, st
    # a call of another length than the one looked ahead for: computed on the spot, still exact
    d2, _ = run_both([iq], n, [chunk, chunk, 2 * chunk], rate=rate, tuning=12345)
    st = d2.schedule_stats()
    assert st["prefetched"] == 1 and st["computed_inline"] == 2, st
    # the periodic default (12 kHz at 96 kHz) with calls that are whole tuner / decimator periods long: the state comes
    # back to where it was, one schedule (two: the first call's history is the stream start's zeros) serves every call
    c96 = 131040  # a multiple of 8 (tuner cycle) and of 10 (decimation)
    iq96 = O.make_dbpsk_stream(62, 0, 4 * c96, noise_sigma=600.0)[0]
    d3, _ = run_both([iq96], 4 * c96, [c96] * 4)
    st = d3.schedule_stats()
    assert st["prefetched"] + st["computed_inline"] <= 2, st


@pytest.mark.parametrize("route", ["0", "1"])
def test_bpsk_one_stream_fed_through_both_input_forms_alternately(route):
    """the fused kernel (int16 batches) keeps the 64-sample halo of VCO-mixed samples in its own buffer, the generic
    front end (float frames) in the dm array: a handle that is fed through both forms in turn must carry the halo
    across every switch -- compared with one oracle that sees the same samples in order.
    route = "0" (JSDR_F32_AS_I16=0, read once per process: run in a child): float frames take the float kernels, so
    the switches really happen; route = "1" (the default since round 3): receive(float[]) recognises frames that are
    (float)s/32767f values and feeds them to the int16 kernels -- same results, one kernel family"""
    if os.environ.get("JSDR_F32_AS_I16", "1") != route:
        import subprocess
        import sys
        env = dict(os.environ, JSDR_F32_AS_I16=route)
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", "-m", "gpu",
                            __file__ + "::test_bpsk_one_stream_fed_through_both_input_forms_alternately[%s]" % route],
                           env=env, capture_output=True, text=True, timeout=500)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
        return
    n = 2048 * 40
    iq = O.make_dbpsk_stream(71, 0, n, noise_sigma=500.0)[0]
    buf = O.convert_i16(iq)
    d = J.Bpsk(nstreams=1, max_batch_samples=2048 * 8)
    o = O.Bpsk(trace=n // 10 + 8)
    d_iq = J.DeviceBuffer.from_host(iq)
    bits, tr = [], []
    pos = 0
    plan = [("f", 1), ("i", 3), ("f", 2), ("i", 8), ("f", 1), ("f", 1), ("i", 1), ("i", 5), ("f", 3), ("i", 8), ("f", 7)]
    assert sum(k for _, k in plan) == 40
    kernels = set()
    for form, k in plan:
        if form == "f":
            for j in range(k):
                d.receive(buf[2 * (pos + 2048 * j):2 * (pos + 2048 * (j + 1))])
                bits.append(d.bits().copy())
                tr.append(d.trace().copy())
                kernels.add(d.front_kernel_name())
        else:
            d.batch_i16(d_iq.ptr + 4 * pos, 2 * n, 2048 * k)
            bits.append(d.bits().copy())
            tr.append(d.trace().copy())
            kernels.add(d.front_kernel_name())
        pos += 2048 * k
    o.receive_i16(iq)
    assert kernels == ({"k_front", "k_fm"} if route == "0" else {"k_fm"})
    assert np.array_equal(np.concatenate(tr), o.trace())
    assert np.array_equal(np.concatenate(bits), o.bits())
    same_counters(d.counters(), o.counters())
    same_state(d.state(), o.state())


def test_bpsk_switching_to_int16_after_arbitrary_floats_is_refused():
    d = J.Bpsk(nstreams=1, max_batch_samples=2048)
    d.receive(np.full(4096, 0.123456789, np.float32))  # not a (float)s/32767f value
    with pytest.raises(J.JsdrError):
        d.batch_i16(J.DeviceBuffer.from_host(np.zeros(4096, np.int16)), 4096, 2048)


def test_bpsk_fft_mode_one_stream_fed_int16_and_arbitrary_float_frames_in_turn():
    """in FFT-acquire mode a 1-stream handle parks the schedule's VCO factors behind the frame in its staging
    buffer and re-uses them while the schedule cache hits.  A float frame that is NOT a (float)s/32767f image takes the float
    path and is twice as long as the int16 frame the factors were parked behind: they must survive it.  Frames of both
    forms in turn (either order), against one oracle fed the same floats."""
    nsf, nfr = 9600, 8  # the application's default frame: 960 outputs a frame, so EVERY frame hits the schedule cache
    rng = np.random.default_rng(20260104)
    iq = O.make_dbpsk_stream(77, 0, nsf * nfr, noise_sigma=700.0)[0]
    exact = O.convert_i16(iq)  # what JavaAudio delivers
    for first_form in ("i", "f"):
        d = J.Bpsk(nstreams=1, do_fft=1, blen=4 * nsf)
        o = O.Bpsk(do_fft=1, blen=4 * nsf, trace=nsf * nfr // 10 + 8)
        bits, tr = [], []
        for k in range(nfr):
            fr = exact[2 * nsf * k:2 * nsf * (k + 1)].copy()
            as_float = ((k // 2) % 2 == 0) == (first_form == "f")
            if as_float:  # off the short grid (and a silent frame now and then: all zeros pass the grid check)
                fr = (fr * np.float32(0.999) + rng.normal(0, 1e-6, fr.size).astype(np.float32)).astype(np.float32) if k % 5 else np.zeros_like(fr)
                d.receive(fr)
            else:
                d.receive_raw(iq[2 * nsf * k:2 * nsf * (k + 1)])
            o.receive(fr)
            bits.append(d.bits().copy())
            tr.append(d.trace().copy())
        assert np.array_equal(np.concatenate(tr), o.trace()), first_form
        assert np.array_equal(np.concatenate(bits), o.bits()), first_form
        same_counters(d.counters(), o.counters())
        same_state(d.state(), o.state())
        assert d.counters()["centreBin"] == o.counters()["centreBin"]


@pytest.mark.parametrize("rate,tuning", [(32000, 4000), (22050, 3000), (11025, 1500), (64000, 12000), (9600, 1200), (250000, 30000),
                                         (32000, -4000), (8000, 1000)])
def test_bpsk_any_audio_rate_takes_the_generic_front_end(rate, tuning):
    """the reference's audio-rate is a free integer (JavaAudio.java:49,59) and RxDownSample takes whatever rate / 9600 is
    (FUNcubeBPSKDemod.java:476): decimations other than 4 / 5 / 10 / 20 go through k_front_any (one thread per output).
    Same bar as everywhere: (fi,fq), bits, counters, state bit for bit; ragged calls; int16 batches and float frames."""
    D = max(1, rate // 9600)  # (below 9600 Hz the reference's `++dsCnt >= 0` fires at every sample: decimation 1)
    n = 2048 * 24
    streams = [O.make_dbpsk_stream(90 + s, s, n, rate=rate, carrier_hz=abs(tuning) + 1200.0, noise_sigma=500.0 + 300 * s)[0]
               for s in range(3)]
    d, _ = run_both(streams, n, [2048 * 5 + 7, 13, 2048 * 11 - 20, n - (2048 * 16)], rate=rate, tuning=tuning)
    assert d.front_kernel_name() == "k_front_any", D
    # one stream through receive(float[]) with floats off the short grid (the float form of the same kernel)
    buf = (O.convert_i16(streams[0]) * np.float32(0.93)).astype(np.float32)
    d1 = J.Bpsk(rate=rate, tuning=tuning, nstreams=1)
    o1 = O.Bpsk(rate=rate, tuning=tuning, trace=n // D + 8)
    tr, bits = [], []
    for k in range(6):
        d1.receive(buf[4096 * k:4096 * (k + 1)])
        o1.receive(buf[4096 * k:4096 * (k + 1)])
        tr.append(d1.trace().copy())
        bits.append(d1.bits().copy())
    assert d1.front_kernel_name() == "k_front_any"
    assert np.array_equal(np.concatenate(tr), o1.trace())
    assert np.array_equal(np.concatenate(bits), o1.bits())
    same_counters(d1.counters(), o1.counters())
    same_state(d1.state(), o1.state())


def test_bpsk_suite_with_the_eight_streams_per_wave_tail():
    """k_tail8 (eight streams per wave) serves handles of 2048 streams and more; the parity tests use a few streams.  The same
    tests once more in a child process with JSDR_TAIL8=2, which makes every handle take k_tail8 (ragged calls, surplus lane
    groups shadowing the last stream, acquisition and fades through the general path, FFT-acquire frame seams)."""
    if os.environ.get("JSDR_TAIL8") == "2":
        return
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, JSDR_TAIL8="2")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", "-m", "gpu",
                        os.path.join(here, "test_gpu_bpsk.py"), os.path.join(here, "test_gpu_fixtures.py"),
                        "-k", "not eight_streams_per_wave and not alternately"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("mode,extra", [("1", {}), ("1", {"JSDR_ACQ_CHUNK": "3", "JSDR_ACQ_RUN": "2"}), ("0", {})])
def test_bpsk_fft_mode_suite_with_either_front_end(mode, extra):
    """Round 6: FFT-acquire calls of two or more frames per stream take the three-phase front end (bpsk_acq.hip: k_acq_fwd /
    k_acq_scan / k_acq_inv / k_acq_edges; the default mixed-radix frames k_acqm_fwd / k_acqm_inv where frames fill the chip better
    than streams), calls of one frame the fused kernels.  Every FFT-mode test once more in a child process with the choice
    FORCED: JSDR_ACQ3=1 -- the three-phase form wherever the frame size has one, single-frame calls and the 9600 / 4800 / 4410
    frames included, also with the call cut into launches of three frames per stream and tickets of two frames; JSDR_ACQ3=0 --
    the fused kernels everywhere (what the default no longer exercises for 2^k frames)."""
    if os.environ.get("JSDR_ACQ3") is not None:
        return
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, JSDR_ACQ3=mode, **extra)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", "-m", "gpu",
                        os.path.join(here, "test_gpu_bpsk.py"), os.path.join(here, "test_gpu_fixtures.py"),
                        "-k", "fft and not either_front_end and not eight_streams_per_wave"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_bpsk_fft_mode_front_end_choice():
    """which front end serves a call: 2^k frames -- three phases from two frames a call; the default mixed-radix frames -- three
    phases where frames fill the chip better than streams (few streams, many frames), the fused kernel at a full grid of streams
    and for one frame a call"""
    if os.environ.get("JSDR_ACQ3") is not None or os.environ.get("JSDR_ACQG") is not None:
        pytest.skip("the choice is forced")
    iq = O.make_dbpsk_stream(7, 0, 2048 * 6)[0]
    d = J.Bpsk(nstreams=1, do_fft=1, max_batch_samples=2048 * 6)
    buf = J.DeviceBuffer.from_host(iq)
    d.batch_i16(buf, 2 * 2048 * 6, 2048)
    assert d.front_kernel_name() == "k_front_fft"
    bits = [d.bits()]  # (the bits of the call)
    d.batch_i16(buf.ptr + 4 * 2048, 2 * 2048 * 6, 2048 * 5)
    assert d.front_kernel_name() == "k_acq_fwd"
    bits.append(d.bits())
    o = O.Bpsk(do_fft=1)
    o.receive_i16(iq)
    assert np.array_equal(np.concatenate(bits), o.bits()) and d.counters()["centreBin"] == o.counters()["centreBin"]
    iq = O.make_dbpsk_stream(8, 0, 9600 * 4)[0]
    d = J.Bpsk(nstreams=1, do_fft=1, blen=38400, max_batch_samples=9600 * 4)
    buf = J.DeviceBuffer.from_host(iq)
    d.batch_i16(buf, 2 * 9600 * 4, 9600)
    assert d.front_kernel_name() == "k_front_fftm"
    bits = [d.bits()]
    d.batch_i16(buf.ptr + 4 * 9600, 2 * 9600 * 4, 9600 * 3)
    assert d.front_kernel_name() == "k_acqm_fwd"  # one stream, three frames: frames fill the chip, streams do not
    bits.append(d.bits())
    o = O.Bpsk(do_fft=1, blen=38400)
    o.receive_i16(iq)
    assert np.array_equal(np.concatenate(bits), o.bits()) and d.counters()["centreBin"] == o.counters()["centreBin"]
    S = 512
    iqs = np.concatenate([O.make_dbpsk_stream(9, s % 4, 9600 * 2)[0] for s in range(S)])
    d = J.Bpsk(nstreams=S, do_fft=1, blen=38400, max_batch_samples=9600 * 2)
    d.batch_i16(J.DeviceBuffer.from_host(iqs), 2 * 9600 * 2, 9600 * 2)
    assert d.front_kernel_name() == "k_front_fftm"  # a full grid of streams: the fused kernel


@pytest.mark.parametrize("nsf,knob,value,first,second", [
    (2048, "JSDR_ACQ3", "0", "k_acq_fwd", "k_front_fft"),      # three phases (tail on the side stream) against the fused kernel
    (9600, "JSDR_ACQ3", "1", "k_front_fftm", "k_acqm_fwd"),    # the fused kernel against three phases
    (4800, "JSDR_FFTM_PAIR", "0", "k_front_fftm2", "k_front_fftm"),  # two frames at once against one
    (2048, "JSDR_ACQG", "1", "k_acq_fwd", "k_acqg_pass"),      # the LDS kernels against the any-frame passes (radix-2 stages in global memory)
    (9600, "JSDR_ACQG", "1", "k_front_fftm", "k_acqg_pass"),   # ... and the Stockham pass pairs
])
def test_bpsk_fft_mode_front_ends_agree_at_config4_size(nsf, knob, value, first, second, monkeypatch):
    """BASELINE config 4's shape in FFT-acquire mode -- 1024 streams x 1,048,576 samples generated on the device, one call -- through
    the default front end and through the other one the library has for that frame: EVERY stream's result slot (counters, the
    call's bits, every FECDecode result, state) byte for byte equal, and two sampled streams equal to the oracle.  Size-independent
    property at full size: two implementations, one answer."""
    if any(os.environ.get(k) is not None for k in ("JSDR_ACQ3", "JSDR_ACQG", "JSDR_FFTM_PAIR")):
        pytest.skip("a front end is forced")
    S, L, sps = 1024, 1048576, 80
    L = (L // nsf) * nsf
    nfr = 3
    seed = 20020111
    pay = J.synth_payloads(seed, 0, S, nfr)
    d_sym = J.DeviceBuffer(S * nfr * 5200)
    J.fec_encode_dev(pay, S * nfr, d_sym)
    d_ds = J.DeviceBuffer(S * nfr * 5200)
    J.synth_diffsign(d_sym, nfr * 5200, S, d_ds)
    ct, st = O.synth_tables(3000)
    keys = np.array([O.mix64((seed * 0x9E3779B1 + s) ^ 0xA5A5A5A5) for s in range(S)], np.uint64)
    d_iq = J.DeviceBuffer(S * L * 4)
    gain = int(round(1500.0 / 37837.0 * 32768.0))
    J.synth_dbpsk(d_iq, 2 * L, S, 0, L, d_ds, nfr * 5200, sps, 0, O.phase_inc_u32(13200.0, 96000),
                  J.DeviceBuffer.from_host(ct), J.DeviceBuffer.from_host(st), gain, J.DeviceBuffer.from_host(keys))

    def run(expect):
        d = J.Bpsk(nstreams=S, max_batch_samples=L, do_fft=1, blen=4 * nsf)
        d.batch_i16(d_iq, 2 * L, L)
        assert d.front_kernel_name() == expect
        info = d.slot_info()
        slots = J.DeviceBuffer(S * info["slot_bytes"])
        d.pack_slots(slots)
        d.sync()
        return d, slots.to_host(np.uint8).reshape(S, info["slot_bytes"]).copy()

    d1, s1 = run(first)
    monkeypatch.setenv(knob, value)
    d2, s2 = run(second)
    differ = np.flatnonzero(np.any(s1 != s2, axis=1))
    assert differ.size == 0, f"{differ.size} streams differ between {first} and {second}: first {differ[:8]}"
    for s in range(0, S, 37):  # (the state doubles are not in the slot)
        same_state(d1.state(s), d2.state(s))
        assert d1.counters(s)["centreBin"] == d2.counters(s)["centreBin"]
    for s in (0, S - 1):
        iq = d_iq.to_host(np.int16, count=2 * L, offset_bytes=4 * (L * s))
        o = O.Bpsk(do_fft=1, blen=4 * nsf)
        o.receive_i16(iq)
        assert np.array_equal(d1.bits(s), o.bits())
        same_counters(d1.counters(s), o.counters())
        same_state(d1.state(s), o.state())
        assert np.array_equal(d2.bits(s), o.bits())


@pytest.mark.parametrize("nsf,rate", [(2048, 96000), (9600, 96000), (4800, 48000), (4410, 44100), (19200, 192000), (512, 48000), (17640, 176400)])
def test_bpsk_fft_mode_degenerate_inputs(nsf, rate):
    """what the reference's loop does with inputs that carry nothing: all-zero frames (the boxcar finds no maximum: binPos stays -1,
    :439-442, and the centre bin sits at the clamp), full-scale square waves (+32767 / -32768: every FP64 product at its largest),
    a DC offset that wraps the short (:282-288), one impulse in a silent call, and a carrier that switches off half way -- every
    front end family against the oracle, frames in calls of 1, 2 and 3"""
    nfr = 6
    n = nsf * nfr
    rng = np.random.default_rng(nsf)
    zeros = np.zeros(2 * n, np.int16)
    square = np.where((np.arange(2 * n) // 2) % 7 < 3, 32767, -32768).astype(np.int16)
    impulse = zeros.copy()
    impulse[2 * (nsf + 17)] = 32767
    carrier = O.make_dbpsk_stream(401, 0, n, rate=rate, carrier_hz=rate / 8.0 + 50.0, noise_sigma=300.0)[0].copy()
    carrier[2 * (3 * nsf + 11):] = 0
    dc = np.full(2 * n, 30000, np.int16)  # + ic = 5000 wraps past 32767
    for streams, ic, qc in (([zeros, square, impulse, carrier], 0, 0), ([dc, zeros], 5000, -5000)):
        run_both(streams, n, [nsf, 2 * nsf, 3 * nsf], rate=rate, ic=ic, qc=qc, do_fft=1, blen=4 * nsf)
        run_both(streams, n, [3 * nsf, nsf, 2 * nsf], rate=rate, ic=ic, qc=qc, do_fft=1, do_up=1, blen=4 * nsf)


@pytest.mark.parametrize("nsf,rate,do_up", [(4800, 48000, 0), (4800, 96000, 1), (4410, 44100, 0), (4410, 44100, 1)])
def test_bpsk_fft_mode_two_frames_at_once(nsf, rate, do_up, monkeypatch):
    """Round 6: at 4800- and 4410-sample frames a workgroup takes frames f and f + 1 of its stream together (k_front_fftm2: two images
    in LDS, every pass over both, the centre-bin rule for f then f + 1).  Calls of 1 (the one-frame kernel), 2, 3 (a pair and a
    dropped half), 5 and 8 frames, carriers that move the centre bin between the two frames of a pair (a noise stream), both
    band halves -- against the oracle, and the same bits with the pairing switched off"""
    if os.environ.get("JSDR_ACQ3") is not None or os.environ.get("JSDR_ACQG") is not None:
        pytest.skip("another front end is forced")
    nfr = 19
    n = nsf * nfr
    rng = np.random.default_rng(nsf + do_up)
    iq = O.make_dbpsk_stream(88, 0, n, rate=rate, carrier_hz=rate * (0.36 if do_up else 0.12) + 211.0, noise_sigma=700.0)[0]
    noise = rng.integers(-12000, 12000, 2 * n).astype(np.int16)
    S = 256  # a full grid of streams keeps the fused kernel (fewer would take the three-phase form)
    streams = [iq, noise] + [O.make_dbpsk_stream(88, 1 + (s % 3), n, rate=rate, carrier_hz=rate * (0.36 if do_up else 0.12) - 150.0 * (s % 5), noise_sigma=500.0)[0]
                             for s in range(3)]
    # (the oracle replays five streams; the rest of the grid repeats them)
    chunks = [nsf, 2 * nsf, 3 * nsf, 5 * nsf, 8 * nsf]
    d = J.Bpsk(rate=rate, blen=4 * nsf, do_fft=1, do_up=do_up, nstreams=S, max_batch_samples=max(chunks))
    buf = J.DeviceBuffer.from_host(np.concatenate([streams[s % 5] for s in range(S)]))
    bits = [[] for _ in range(5)]
    names = []
    pos = 0
    for L in chunks:
        d.batch_i16(buf.ptr + 4 * pos, 2 * n, L)
        names.append(d.front_kernel_name())
        for s in range(5):
            bits[s].append(d.bits(250 + s).copy())  # (stream 250 + s repeats stream s)
        pos += L
    assert names == ["k_front_fftm", "k_front_fftm2", "k_front_fftm2", "k_front_fftm2", "k_front_fftm2"]
    for s in range(5):
        o = O.Bpsk(rate=rate, blen=4 * nsf, do_fft=1, do_up=do_up)
        o.receive_i16(streams[s])
        assert np.array_equal(np.concatenate(bits[s]), o.bits()), f"stream {s}"
        same_counters(d.counters(s), o.counters())
        same_state(d.state(s), o.state())
        assert d.counters(s)["centreBin"] == o.counters()["centreBin"]
    monkeypatch.setenv("JSDR_FFTM_PAIR", "0")
    d1 = J.Bpsk(rate=rate, blen=4 * nsf, do_fft=1, do_up=do_up, nstreams=S, max_batch_samples=max(chunks))
    pos = 0
    for L in chunks:
        d1.batch_i16(buf.ptr + 4 * pos, 2 * n, L)
        assert d1.front_kernel_name() == "k_front_fftm"
        pos += L
    for s in range(5):
        same_counters(d1.counters(s), d.counters(s))
        same_state(d1.state(s), d.state(s))


@pytest.mark.parametrize("nsf,rate", [(4410, 44100), (3200, 32000), (2205, 22050), (1102, 11025), (800, 8000)])
def test_bpsk_fft_mode_receive_at_consumer_sound_card_rates(nsf, rate):
    """the IAudioHandler form (one stream, one frame per receive()) in FFT-acquire mode at the frames a 44.1 / 32 / 22.05 kHz
    card delivers (JavaAudio.java:59): radix-7 passes, frames that are not multiples of 16, decimations 4 / 3 / 2 -- int16
    and float frames in turn, against the oracle"""
    nfr = 10
    iq = O.make_dbpsk_stream(91, 0, nsf * nfr, rate=rate, carrier_hz=rate / 8.0 + 150.0, noise_sigma=600.0)[0]
    exact = O.convert_i16(iq)
    d = J.Bpsk(nstreams=1, do_fft=1, rate=rate, blen=4 * nsf)
    o = O.Bpsk(do_fft=1, rate=rate, blen=4 * nsf, trace=nsf * nfr // max(1, rate // 9600) + 8)
    bits, tr = [], []
    for k in range(nfr):
        fr = exact[2 * nsf * k:2 * nsf * (k + 1)].copy()
        if k % 3 == 1:
            d.receive(fr)
        else:
            d.receive_raw(iq[2 * nsf * k:2 * nsf * (k + 1)])
        o.receive(fr)
        bits.append(d.bits().copy())
        tr.append(d.trace().copy())
    assert np.array_equal(np.concatenate(tr), o.trace())
    assert np.array_equal(np.concatenate(bits), o.bits())
    same_counters(d.counters(), o.counters())
    same_state(d.state(), o.state())
    assert d.counters()["centreBin"] == o.counters()["centreBin"]
