"""GPU parity against fixtures that do NOT come from the C oracle.

tests/golden/reference_fixtures.npz is the output of the pure-Python restatement of the Java text
(tests/golden/java_restatement.py, run in the build container): slicer bits, counters, every scalar state double,
(fi,fq) trace, FECDecode return codes and bytes.  The HIP path, through the C ABI, must reproduce all of it bit
for bit.  Also here: the BASELINE config-5 per-GPU shard shape (1024 streams x 2^20 samples) on the device.
"""
import hashlib
import os
import sys

import numpy as np
import pytest

import java_sdr_amd as J
from java_sdr_amd import sharding as SH
import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from fixture_cases import FEC_CASES, STREAMS, stream_input  # noqa: E402

pytestmark = pytest.mark.gpu
FX = np.load(os.path.join(HERE, "golden", "reference_fixtures.npz"))
CN = ("cntRaw", "cntDS", "cntBit", "cntFEC", "cntDec", "dmErrBits", "dmCorr", "dmMaxCorr", "decodeOK")
KEEP = [0, 1, 2, 3, 4, 5] + list(range(8, 18))


def _check_stream(name, chunks, variant=None):
    p = STREAMS[name]
    raw = stream_input(name)
    k = "s_" + name + "_"
    assert hashlib.sha256(raw.tobytes()).digest() == FX[k + "sha256"].tobytes()
    kw = {} if variant is None else {"variant": variant}
    d = J.Bpsk(rate=p["rate"], blen=8192, tuning=p["tuning"], nstreams=1, max_batch_samples=max(chunks), **kw)
    d_iq = J.DeviceBuffer.from_host(raw)
    bits, trace, fec = [], [], []
    pos = 0
    for L in chunks:
        d.batch_i16(d_iq.ptr + 4 * pos, 2 * p["n"], L, p["ic"], p["qc"])
        bits.append(d.bits(0).copy())
        trace.append(d.trace(0).copy())
        fec.extend(d.fec_results(0))
        pos += L
    assert pos == p["n"]
    c = d.counters(0)
    assert [c[n] for n in CN] == [int(v) for v in FX[k + "counters"][:9]]
    assert np.array_equal(np.concatenate(bits), FX[k + "bits"])
    assert [r[0] for r in fec] == [int(v) for v in FX[k + "fec_rc"]]
    for r, want in zip(fec, FX[k + "fec_data"]):
        assert np.array_equal(r[2], want)
    assert np.array_equal(d.decoded(0), FX[k + "decoded"])
    if variant in (None, "exact"):  # the exact-order FP64 variant reproduces the doubles as well
        assert d.state(0)[KEEP].tobytes() == FX[k + "state"][KEEP].tobytes()
        assert np.concatenate(trace)[:2048].tobytes() == FX[k + "trace"].tobytes()
    return d


@pytest.mark.parametrize("name", list(STREAMS))
def test_hip_demodulator_equals_python_restatement_one_batch(name):
    _check_stream(name, [STREAMS[name]["n"]])


@pytest.mark.parametrize("name", ["sine", "clean", "r48k_dc", "r192k"])
def test_hip_demodulator_equals_python_restatement_frame_cadence_and_ragged(name):
    n = STREAMS[name]["n"]
    if name == "sine":
        _check_stream(name, [2048, 2048])
    else:
        _check_stream(name, [2048] * 8 + [77, 1, 4099, n - 2048 * 8 - 77 - 1 - 4099 - 65536, 65536])


def test_hip_encoder_equals_python_restatement():
    assert np.array_equal(J.fec_encode(FX["f_payload"]), FX["f_symbols"])


@pytest.mark.parametrize("name", FEC_CASES)
def test_hip_fecdecode_equals_python_restatement(name):
    rc, out = J.fec_decode(FX["f_" + name + "_in"], out_init=np.full(256, 0xEE, np.uint8))
    assert rc == int(FX["f_" + name + "_rc"][0])
    assert np.array_equal(out, FX["f_" + name + "_out"])


def test_hip_fecdecode_batch_equals_python_restatement():
    raws = np.stack([FX["f_" + n + "_in"] for n in FEC_CASES] * 5)
    rcs, outs = J.fec_decode_batch(raws)
    for i, n in enumerate(FEC_CASES * 5):
        assert rcs[i] == int(FX["f_" + n + "_rc"][0])
        if rcs[i] >= 0:
            assert np.array_equal(outs[i], FX["f_" + n + "_out"])


def test_config5_shard_shape_1024_streams_by_2pow20_samples():
    """BASELINE config 5's per-GPU shard: 1024 streams x 1,048,576 int16 IQ samples (4 GiB) resident on the device.
    Every stream must return the payloads it was sent (encode -> modulate -> demodulate -> decode), and eight
    sampled streams must match the oracle's bits, counters and FEC bytes."""
    S, L, sps, nfr, seed = 1024, 1048576, 80, 3, 20020109
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
    info = d.slot_info()
    slots = J.DeviceBuffer(S * info["slot_bytes"])
    d.pack_slots(slots)
    blob = slots.to_host(np.uint8).reshape(S, info["slot_bytes"])
    for s in range(S):
        u = SH.unpack_slot(blob[s], info)
        assert len(u["fec"]) == 2, (s, len(u["fec"]))
        for k2, (rc, _, data) in enumerate(u["fec"]):
            assert rc >= 0 and np.array_equal(data, payloads[s, k2]), (s, k2, rc)
    for s in (0, 1, 127, 128, 511, 512, 777, 1023):
        iq = d_iq.to_host(np.int16, count=2 * L, offset_bytes=4 * L * s)
        o = O.Bpsk()
        o.receive_i16(iq)
        assert np.array_equal(d.bits(s), o.bits()), s
        c, oc = d.counters(s), o.counters()
        assert [c[n] for n in CN] == [oc[n] for n in CN], s
        for (rc, _, data), (orc, _, odata) in zip(d.fec_results(s), o.fec_results()):
            assert rc == orc and np.array_equal(data, odata)


# ------------------------------------------------------------------ the fast variant (FMA-contracted FP64, certified decisions)
@pytest.mark.parametrize("name", list(STREAMS))
def test_fast_variant_bits_and_fec_bytes_equal_python_restatement(name):
    """north_star: bit-exact slicer bits and FECDecoder bytes; the doubles may differ (they do, within the bound)"""
    d = _check_stream(name, [STREAMS[name]["n"]], variant="fast")
    st = d.cert_stats()
    assert st["streams_uncertified"] == 0


@pytest.mark.parametrize("name", ["clean", "noisy", "r48k_dc"])
def test_fast_variant_ragged_calls(name):
    n = STREAMS[name]["n"]
    _check_stream(name, [2048] * 8 + [77, 1, 4099, n - 2048 * 8 - 77 - 1 - 4099 - 65536, 65536], variant="fast")


def test_fast_variant_error_of_fi_fq_is_inside_the_proven_bound():
    p = STREAMS["noisy"]
    raw = stream_input("noisy")
    d_iq = J.DeviceBuffer.from_host(raw)
    tr = {}
    for variant in ("exact", "fast"):
        d = J.Bpsk(rate=p["rate"], blen=8192, tuning=p["tuning"], nstreams=1, max_batch_samples=p["n"], variant=variant)
        d.batch_i16(d_iq, 2 * p["n"], p["n"])
        tr[variant] = d.trace(0).copy()
        if variant == "fast":
            bound = d.cert_stats()["fi_fq_error_bound"]
            assert d.front_kernel_name() == "k_fm"
    err = np.abs(tr["fast"] - tr["exact"]).max()
    assert 0.0 < err <= bound, (err, bound)  # FMA really is in use, and the worst-case bound holds with room
    assert err < 0.05 * bound  # full-scale input would be ~10x larger; this stream's amplitude is 0.1 FS


def test_fast_variant_redoes_unsure_decisions_in_exact_order(monkeypatch):
    """widen the detector margins by 1e14: most decisions fall inside and take the cold path (two samples recomputed
    from the raw input in exact order); bits and bytes must not change"""
    monkeypatch.setenv("JSDR_FAST_MARGIN_SCALE", "1e14")
    d = _check_stream("clean", [STREAMS["clean"]["n"]], variant="fast")
    st = d.cert_stats()
    assert st["decisions_redone_exactly"] > 100 and st["streams_uncertified"] == 0, st
    d = _check_stream("noisy", [200000, STREAMS["noisy"]["n"] - 200000], variant="fast")
    assert d.cert_stats()["decisions_redone_exactly"] > 100


def test_fast_variant_flags_a_stream_it_cannot_certify(monkeypatch):
    monkeypatch.setenv("JSDR_FAST_ARGMAX_SCALE", "1e14")
    p = STREAMS["clean"]
    d = J.Bpsk(rate=p["rate"], blen=8192, tuning=p["tuning"], nstreams=2, max_batch_samples=65536, variant="fast")
    raw = stream_input("clean")[:2 * 65536]
    d.batch_i16(J.DeviceBuffer.from_host(np.concatenate([raw, raw])), 2 * 65536, 65536)
    assert d.cert_stats()["streams_uncertified"] == 2
    for getter in (d.bits, d.fec_results, d.decoded, d.counters):
        with pytest.raises(J.JsdrError):
            getter(0)


def test_fast_variant_uncertified_streams_are_listed_and_recovered_on_an_exact_handle(monkeypatch):
    """the drop-in contract of the fast variant (include/jsdr_hip.h): a stream it cannot certify is LISTED after the call,
    its results are withheld, and running that stream's input on an exact handle gives the reference's bits, counters
    and FEC bytes for the call"""
    monkeypatch.setenv("JSDR_FAST_ARGMAX_SCALE", "1e14")  # every argmax falls inside the (widened) margin
    p = STREAMS["clean"]
    n = 458752
    raw = stream_input("clean")[:2 * n]
    quiet = np.zeros(2 * n, np.int16)  # silence: all eight energies exactly 0.0 in both variants, nothing to certify
    d = J.Bpsk(rate=p["rate"], blen=8192, tuning=p["tuning"], nstreams=3, max_batch_samples=n, variant="fast")
    d_in = J.DeviceBuffer.from_host(np.concatenate([raw, quiet, raw]))
    d.batch_i16(d_in, 2 * n, n)
    flagged = d.uncertified_streams()
    assert flagged == [0, 2] and d.cert_stats()["streams_uncertified"] == 2
    assert len(d.bits(1)) == 0  # the certified stream's results ARE delivered
    monkeypatch.delenv("JSDR_FAST_ARGMAX_SCALE")
    ex = J.Bpsk(rate=p["rate"], blen=8192, tuning=p["tuning"], nstreams=len(flagged), max_batch_samples=n, variant="exact")
    assert ex.uncertified_streams() == []
    ex.batch_i16(J.DeviceBuffer.from_host(np.concatenate([raw] * len(flagged))), 2 * n, n)
    o = O.Bpsk(rate=p["rate"], blen=8192, tuning=p["tuning"])
    o.receive_i16(raw)
    for i in range(len(flagged)):
        assert np.array_equal(ex.bits(i), o.bits())
        assert ex.counters(i)["cntBit"] == o.counters()["cntBit"]
        fg, fo = ex.fec_results(i), o.fec_results()
        assert len(fg) == len(fo) and all(a[0] == b[0] and np.array_equal(a[2], b[2]) for a, b in zip(fg, fo))


def test_fast_variant_recovers_uncertified_streams_by_replay(monkeypatch):
    """a batch over recorded IQ still has its input, so the streams a fast handle could not certify are
    REPLAYED -- jsdr_bpsk_recover_uncertified: every call since creation, on an internal exact handle that serves those
    streams from then on.  After it their bits / FEC bytes / counters / state are the oracle's, the flag no longer withholds
    them, their packed slots are the exact ones, and later calls keep them exact (the shadow runs in lock-step)."""
    monkeypatch.setenv("JSDR_FAST_ARGMAX_SCALE", "1e14")  # every argmax falls inside the (widened) margin: streams 0 and 2 cannot be certified
    p = STREAMS["clean"]
    n = 458752
    raw = stream_input("clean")[:2 * n]
    quiet = np.zeros(2 * n, np.int16)
    cuts = [0, 2048 * 98, 2048 * 175, n]  # (whole frames: the oracle's receive() takes frames)
    d = J.Bpsk(rate=p["rate"], blen=8192, tuning=p["tuning"], nstreams=3, max_batch_samples=2048 * 98, variant="fast")
    d_in = J.DeviceBuffer.from_host(np.concatenate([raw, quiet, raw]))
    o = O.Bpsk(rate=p["rate"], blen=8192, tuning=p["tuning"])
    obits = [0]
    ofec = [0]
    for k in range(3):
        o.receive_i16(raw[2 * cuts[k]:2 * cuts[k + 1]])
        obits.append(len(o.bits()))
        ofec.append(len(o.fec_results()))

    def check_against_oracle(k):  # after call k (0-based) the recovered streams hold the oracle's results of that call
        for s in (0, 2):
            assert np.array_equal(d.bits(s), o_bits_all[obits[k]:obits[k + 1]])
            fg = d.fec_results(s)
            fo = o_fec_all[ofec[k]:ofec[k + 1]]
            assert len(fg) == len(fo) and all(a[0] == b[0] and np.array_equal(a[2], b[2]) for a, b in zip(fg, fo))

    o_bits_all, o_fec_all = o.bits().copy(), o.fec_results()
    ptrs = [d_in.ptr + 4 * cuts[k] for k in range(3)]
    lens = [cuts[k + 1] - cuts[k] for k in range(3)]
    for k in range(2):
        d.batch_i16(ptrs[k], 2 * n, lens[k])
    assert d.uncertified_streams() == [0, 2]
    with pytest.raises(J.JsdrError):
        d.bits(0)
    with pytest.raises(J.JsdrError, match="calls given"):
        d.recover_uncertified(ptrs[:1], lens[:1], 2 * n)  # the whole history, or nothing
    assert d.recover_uncertified(ptrs[:2], lens[:2], 2 * n) == 2
    assert d.uncertified_streams() == [] and d.cert_stats()["streams_uncertified"] == 0
    check_against_oracle(1)
    assert len(d.bits(1)) == 0  # the certified stream is still the fast handle's
    assert d.recover_uncertified(ptrs[:2], lens[:2], 2 * n) == 0  # nothing new
    d.batch_i16(ptrs[2], 2 * n, lens[2])  # the shadow runs beside the fast kernels from now on
    check_against_oracle(2)
    c, oc = d.counters(0), o.counters()
    assert all(c[k] == oc[k] for k in ("cntRaw", "cntDS", "cntBit", "cntFEC", "cntDec", "dmErrBits", "decodeOK")), (c, oc)
    assert d.state(2).tobytes() == o.state().tobytes()
    info = d.slot_info()
    slots = J.DeviceBuffer(3 * info["slot_bytes"])
    d.pack_slots(slots)
    blob = slots.to_host(np.uint8).reshape(3, info["slot_bytes"])
    assert np.array_equal(blob[0], blob[2])
    u = J.sharding.unpack_slot(blob[0], info)
    assert np.array_equal(u["bits"], o_bits_all[obits[2]:obits[3]]) and int(u["header"][4]) == oc["cntBit"] and int(u["header"][12]) == 0


def test_fast_variant_is_tune_mode_only():
    with pytest.raises(J.JsdrError):
        J.Bpsk(nstreams=1, do_fft=1, variant="fast")


def test_fast_variant_takes_int16_input_only():
    d = J.Bpsk(nstreams=1, variant="fast")
    with pytest.raises(J.JsdrError):
        d.receive(np.zeros(4096, np.float32))
    d.receive_raw(np.zeros(4096, np.int16))  # the raw (IRawHandler) form is fine


def test_fast_variant_certifies_silence_and_scales_its_bound_with_the_input():
    """an all-zero stream keeps all eight smoothed energies at exactly 0.0 in both variants: the tie is the reference's
    tie, nothing to certify.  A quiet stream (1 % of full scale) gets margins 100x tighter than the full-scale bound."""
    d = J.Bpsk(nstreams=2, max_batch_samples=65536, variant="fast")
    quiet = (stream_input("clean")[:2 * 65536].astype(np.int32) // 10).astype(np.int16)
    both = np.concatenate([np.zeros(2 * 65536, np.int16), quiet])
    d.batch_i16(J.DeviceBuffer.from_host(both), 2 * 65536, 65536)
    assert d.cert_stats()["streams_uncertified"] == 0
    assert d.counters(0)["cntBit"] == 0
    o = O.Bpsk()
    o.receive_i16(quiet)
    assert np.array_equal(d.bits(1), o.bits())


# ------------------------------------------------------------------ FFT-acquire mode against the Python restatement
from fixture_cases import FFT_STREAMS, fft_stream_input  # noqa: E402

FXF = np.load(os.path.join(HERE, "golden", "fftmode_fixtures.npz"))


@pytest.mark.parametrize("name", list(FFT_STREAMS))
@pytest.mark.parametrize("whole", [True, False])
def test_hip_fft_acquire_equals_python_restatement(name, whole):
    """bpsk-dofft: doBufferFFT restated from the Java text (tests/golden/java_restatement.DemodFFT) around the transform
    this project defines in JTransforms' place -- bits, counters, the centre bin, avePeakPower / aveCentreBin and every
    other state double, the (fi,fq) trace; frame by frame (the plugin's receive cadence) and as one batch call"""
    p = FFT_STREAMS[name]
    raw = fft_stream_input(name)
    k = "x_" + name + "_"
    assert hashlib.sha256(raw.tobytes()).digest() == FXF[k + "sha256"].tobytes()
    n = p["frame"]
    chunks = [p["n"]] if whole else [n] * (p["n"] // n)
    d = J.Bpsk(rate=p["rate"], blen=4 * n, tuning=12000, do_fft=1, do_up=p["do_up"], nstreams=1, max_batch_samples=max(chunks))
    d_iq = J.DeviceBuffer.from_host(raw)
    bits, trace, centre = [], [], []
    pos = 0
    for L in chunks:
        d.batch_i16(d_iq.ptr + 4 * pos, 2 * p["n"], L, 0, 0)
        bits.append(d.bits(0).copy())
        trace.append(d.trace(0).copy())
        centre.append(d.counters(0)["centreBin"])
        pos += L
    c = d.counters(0)
    assert [c[m] for m in CN] == [int(v) for v in FXF[k + "counters"][:9]]
    assert np.array_equal(np.concatenate(bits), FXF[k + "bits"])
    if whole:
        assert centre[-1] == int(FXF[k + "centre"][-1])
    else:
        assert centre == [int(v) for v in FXF[k + "centre"]]
    keep = [1, 2, 3, 4, 5, 6, 7] + list(range(8, 18))
    assert d.state(0)[keep].tobytes() == FXF[k + "state"][keep].tobytes()
    assert np.concatenate(trace)[:1024].tobytes() == FXF[k + "trace"].tobytes()
