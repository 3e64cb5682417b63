"""GPU parity for the formats either side of the path: waterfall pixel rows bit-exact vs the
oracle; WAV / headerless recordings land in the stream-major device layout byte for byte (checked against
Python's own `wave` reader) and feed the fft kernel."""
import os
import wave

import numpy as np
import pytest

import java_sdr_amd as J
import oracle_lib as O

pytestmark = pytest.mark.gpu


def test_waterfall_rows_bit_exact():
    rng = np.random.default_rng(3)
    for n, width in ((2048, 1024), (2048, 801), (2048, 2048), (2048, 2500), (9600, 1920), (19200, 1600), (64, 5)):
        psd = (rng.standard_normal((9, n + 2)) * 35 - 70).astype(np.float32)
        psd[2, rng.integers(0, n, 40)] = -np.inf
        psd[3, rng.integers(0, n, 40)] = np.nan
        psd[4, :] = -np.inf  # an all-zero frame (fft.java:207 gives -Infinity everywhere)
        psd[5, rng.integers(0, n, 40)] = 25.0
        for rgb in (0x00FFFF, 0xFFFFFF, 0x123456):
            got = J.waterfall_lines(psd, n, width, rgb)
            for k in range(psd.shape[0]):
                assert np.array_equal(got[k], O.waterfall_line(psd[k], n, width, rgb)), (n, width, k, hex(rgb))


def test_waterfall_of_real_psd_frames(golden_dir):
    raw = np.fromfile(os.path.join(golden_dir, "sine4410.raw"), dtype="<i2")
    f = J.Fft(2048, 96000)
    for k in range(2):
        psd = f.receive_raw(raw[k * 4096:(k + 1) * 4096])
        row = J.waterfall_lines(psd, 2048, 1024)[0]
        assert np.array_equal(row, O.waterfall_line(psd, 2048, 1024))
        # the tone at -9609 Hz shows as the brightest pixel, left of centre after the width/2 rotation
        bright = int(np.argmax(row & 0xFF))
        peak_bin = int(np.argmax(psd[:2048]))
        assert bright == (peak_bin // 2 + 512) % 1024


def test_waterfall_api_errors():
    with pytest.raises(J.JsdrError):
        J.waterfall_lines(np.zeros(66, np.float32), 64, 0)


def test_recordings_load_wav_and_raw_into_stream_major_layout(golden_dir, tmp_path):
    wav = os.path.join(golden_dir, "sine4410.wav")
    rawp = os.path.join(golden_dir, "sine4410.raw")
    info = J.recording_probe(wav)
    with wave.open(wav, "rb") as w:
        assert (info.format, info.encoding, info.channels, info.rate, info.bits) == (1, 1, w.getnchannels(), w.getframerate(), 16)
        assert info.frames == w.getnframes()
        want_wav = np.frombuffer(w.readframes(w.getnframes()), "<i2")
    want_raw = np.fromfile(rawp, "<i2")
    iraw = J.recording_probe(rawp, raw_channels=2)
    assert (iraw.format, iraw.frames, iraw.data_offset) == (0, want_raw.size // 2, 0)

    nframes, first = 6000, 100
    stride = 2 * nframes + 64
    dev = J.DeviceBuffer(2 * stride * 2)
    dev.zero()
    got = J.recordings_load([wav, rawp], 2, 44100, first, nframes, dev, stride)
    host = dev.to_host(np.int16).reshape(2, stride)
    assert got == [nframes, want_raw.size // 2 - first]
    assert np.array_equal(host[0, :2 * nframes], want_wav[2 * first:2 * (first + nframes)])
    n1 = got[1]
    assert np.array_equal(host[1, :2 * n1], want_raw[2 * first:2 * (first + n1)])
    assert np.all(host[1, 2 * n1:2 * nframes] == 0)  # short file: zero fill

    # the loaded frames drive the batch fft exactly like the same samples handed over from the host
    f = J.Fft(2048, 44100)
    d_psd = J.DeviceBuffer(4 * 2 * 2050)
    f.batch_i16(dev, 2, d_psd)  # frames 0,1 of stream 0
    J.lib().jsdr_stream_sync(None)
    psd_dev = d_psd.to_host(np.float32).reshape(2, 2050)
    psd_host = np.stack([f.receive_raw(want_wav[2 * first + 4096 * k:2 * first + 4096 * (k + 1)]) for k in range(2)])
    assert np.array_equal(psd_dev, psd_host)


def test_recordings_mono_and_format_errors(golden_dir, tmp_path):
    rng = np.random.default_rng(9)
    mono = rng.integers(-32768, 32768, 3000).astype("<i2")
    p = str(tmp_path / "mono.wav")
    with wave.open(p, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(48000)
        w.writeframes(mono.tobytes())
    # a LIST chunk between fmt and data must be skipped
    b = open(p, "rb").read()
    i = b.find(b"data")
    extra = b"LIST" + (5).to_bytes(4, "little") + b"abcde" + b"\x00"
    b2 = b[:i] + extra + b[i:]
    b2 = b2[:4] + (len(b2) - 8).to_bytes(4, "little") + b2[8:]
    p2 = str(tmp_path / "mono_list.wav")
    open(p2, "wb").write(b2)
    dev = J.DeviceBuffer(2 * 2 * 4096 * 2)
    got = J.recordings_load([p, p2], 1, 48000, 0, 4096, dev, 2 * 4096)
    host = dev.to_host(np.int16).reshape(2, 2 * 4096)
    assert got == [3000, 3000]
    for s in range(2):
        assert np.array_equal(host[s, 0:6000:2], mono) and np.all(host[s, 1:6000:2] == 0) and np.all(host[s, 6000:] == 0)
    # JavaAudio.compareFormat: wrong channel count, wrong rate, 8-bit samples are "Incompatible audio format"
    with pytest.raises(J.JsdrError, match="Incompatible audio format"):
        J.recordings_load([p], 2, 48000, 0, 16, dev, 2 * 4096)
    with pytest.raises(J.JsdrError, match="Incompatible audio format"):
        J.recordings_load([p], 1, 96000, 0, 16, dev, 2 * 4096)
    p8 = str(tmp_path / "eight.wav")
    with wave.open(p8, "wb") as w:
        w.setnchannels(2)
        w.setsampwidth(1)
        w.setframerate(48000)
        w.writeframes(bytes(64))
    with pytest.raises(J.JsdrError, match="Incompatible audio format"):
        J.recordings_load([p8], 2, 48000, 0, 16, dev, 2 * 4096)
    with pytest.raises(J.JsdrError, match="Not readable"):
        J.recordings_load([str(tmp_path / "missing.wav")], 2, 48000, 0, 16, dev, 2 * 4096)


def test_waterfall_rows_of_a_full_batch():
    """4096 PSD frames straight from the batch fft kernel (device to device), every 97th row against the oracle"""
    n, nframes, width = 2048, 4096, 1280
    ct, _ = O.synth_tables(6000)
    d_ct = J.DeviceBuffer.from_host(ct)
    d_raw = J.DeviceBuffer(nframes * n * 4)
    J.synth_tones(d_raw, 0, nframes, n, d_ct, 250, O.mix64(7))
    f = J.Fft(n, 96000)
    d_psd = J.DeviceBuffer(nframes * (n + 2) * 4)
    f.batch_i16(d_raw, nframes, d_psd)
    d_pix = J.DeviceBuffer(nframes * width * 4)
    J.waterfall_lines_dev(d_psd, nframes, n, width, d_pix)
    J.lib().jsdr_stream_sync(None)
    psd = d_psd.to_host(np.float32).reshape(nframes, n + 2)
    pix = d_pix.to_host(np.uint32).reshape(nframes, width)
    for k in range(0, nframes, 97):
        assert np.array_equal(pix[k], O.waterfall_line(psd[k], n, width)), k
