"""ctypes binding of libjsdr_hip.so (the C ABI declared in include/jsdr_hip.h).

Plain pointers and sizes only; numpy arrays are used for host buffers, DeviceBuffer (or any integer
device address, e.g. torch.Tensor.data_ptr()) for HBM buffers.
"""
import ctypes as C
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
# JSDR_LIB: developer knob for same-box A/B timing of two builds (tools/build_variant.sh)
_SO = os.environ.get("JSDR_LIB") or os.path.join(HERE, "libjsdr_hip.so")


class JsdrError(RuntimeError):
    pass


def library_path():
    return _SO


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "jsdr_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(jsdr_[a-z0-9_]+)\s*\(", hdr)))


EXPORTED_SYMBOLS = _declared_symbols()

_lib = None


def lib():
    """Load the library (building nothing: run java-sdr_amd/build.py or __graft_entry__.build() first)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise JsdrError(f"{_SO} is missing: build it with `python java-sdr_amd/build.py` "
                            "(there is no CPU fallback)")
        _lib = C.CDLL(_SO)
        _lib.jsdr_last_error.restype = C.c_char_p
        _lib.jsdr_bpsk_profile_name.restype = C.c_char_p
        _lib.jsdr_demod_profile_name.restype = C.c_char_p
        for fn in ("jsdr_bpsk_front_kernel", "jsdr_bpsk_tail_kernel", "jsdr_bpsk_fec_kernel", "jsdr_fft_kernel"):
            getattr(_lib, fn).restype = C.c_char_p
            getattr(_lib, fn).argtypes = [C.c_void_p]
        for name in EXPORTED_SYMBOLS:
            if name not in ("jsdr_last_error", "jsdr_bpsk_profile_name", "jsdr_demod_profile_name", "jsdr_bpsk_front_kernel",
                            "jsdr_bpsk_tail_kernel", "jsdr_bpsk_fec_kernel", "jsdr_fft_kernel"):
                getattr(_lib, name).restype = C.c_int
    return _lib


def _check(rc, what=""):
    if rc != 0:
        msg = lib().jsdr_last_error()
        raise JsdrError(f"{what}: {msg.decode() if msg else 'error'}")


def have_gpu():
    try:
        n = C.c_int(0)
        return lib().jsdr_device_count(C.byref(n)) == 0 and n.value > 0
    except Exception:
        return False


def _addr(x):
    if x is None:
        return None
    if isinstance(x, DeviceBuffer):
        return C.c_void_p(x.ptr)
    if isinstance(x, np.ndarray):
        return C.c_void_p(x.ctypes.data)
    if isinstance(x, int):
        return C.c_void_p(x)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    raise TypeError(type(x))


class DeviceBuffer:
    """HBM allocation owned through jsdr_malloc/jsdr_free."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        _check(lib().jsdr_malloc(C.byref(p), C.c_size_t(self.nbytes)), "jsdr_malloc")
        self.ptr = p.value

    @classmethod
    def from_host(cls, arr):
        arr = np.ascontiguousarray(arr)
        b = cls(arr.nbytes)
        _check(lib().jsdr_memcpy_h2d(C.c_void_p(b.ptr), _addr(arr), C.c_size_t(arr.nbytes)), "h2d")
        return b

    def to_host(self, dtype, count=None, offset_bytes=0):
        dtype = np.dtype(dtype)
        if count is None:
            count = (self.nbytes - offset_bytes) // dtype.itemsize
        out = np.empty(count, dtype)
        _check(lib().jsdr_memcpy_d2h(_addr(out), C.c_void_p(self.ptr + offset_bytes), C.c_size_t(out.nbytes)), "d2h")
        return out

    def zero(self):
        _check(lib().jsdr_memset(C.c_void_p(self.ptr), 0, C.c_size_t(self.nbytes)), "memset")

    def free(self):
        if self.ptr:
            lib().jsdr_free(C.c_void_p(self.ptr))
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Stream:
    """a non-blocking HIP stream (pass `.ptr` as the `stream` argument of the batch calls)"""

    def __init__(self):
        self.h = C.c_void_p()
        _check(lib().jsdr_stream_create(C.byref(self.h)), "jsdr_stream_create")
        self.ptr = self.h.value

    def sync(self):
        stream_sync(self.ptr)

    def __del__(self):
        try:
            if self.h:
                lib().jsdr_stream_destroy(self.h)
        except Exception:
            pass


def stream_sync(stream=None):
    _check(lib().jsdr_stream_sync(C.c_void_p(stream)), "jsdr_stream_sync")


class Timer:
    """HIP-event pair on the stream the kernels are launched on."""

    def __init__(self):
        self.h = C.c_void_p()
        _check(lib().jsdr_timer_create(C.byref(self.h)), "timer_create")

    def start(self, stream=None):
        _check(lib().jsdr_timer_start(self.h, C.c_void_p(stream)), "timer_start")

    def stop(self, stream=None):
        _check(lib().jsdr_timer_stop(self.h, C.c_void_p(stream)), "timer_stop")

    def elapsed_ms(self):
        ms = C.c_float()
        _check(lib().jsdr_timer_elapsed_ms(self.h, C.byref(ms)), "timer_elapsed")
        return ms.value

    def __del__(self):
        try:
            lib().jsdr_timer_destroy(self.h)
        except Exception:
            pass


# ------------------------------------------------------------------ JavaAudio conversion rule
def convert_i16(raw, chns=2, ic=0, qc=0):
    raw = np.ascontiguousarray(raw, np.int16)
    nframes = raw.size // chns
    d_in = DeviceBuffer.from_host(raw)
    d_out = DeviceBuffer(8 * nframes)
    _check(lib().jsdr_convert_i16(_addr(d_in), C.c_int64(nframes), chns, ic, qc, _addr(d_out), None), "convert_i16")
    return d_out.to_host(np.float32)


# ------------------------------------------------------------------ fft.java
class Fft:
    """fft.receive: n = blen/size samples per frame -> psd float[n+2] (fft.java:190-228)."""

    def __init__(self, n, rate):
        self.n, self.rate = n, rate
        self.h = C.c_void_p()
        _check(lib().jsdr_fft_create(C.byref(self.h), n, rate), "jsdr_fft_create")

    def receive(self, buf):
        buf = np.ascontiguousarray(buf, np.float32)
        assert buf.size == 2 * self.n
        psd = np.empty(self.n + 2, np.float32)
        _check(lib().jsdr_fft_receive_f32(self.h, _addr(buf), _addr(psd)), "jsdr_fft_receive_f32")
        return psd

    def receive_raw(self, raw, ic=0, qc=0):
        raw = np.ascontiguousarray(raw, np.int16)
        assert raw.size == 2 * self.n
        psd = np.empty(self.n + 2, np.float32)
        _check(lib().jsdr_fft_receive_i16(self.h, _addr(raw), ic, qc, _addr(psd)), "jsdr_fft_receive_i16")
        return psd

    def set_cu_share(self, wgs_per_cu):
        _check(lib().jsdr_fft_set_cu_share(self.h, int(wgs_per_cu)), "jsdr_fft_set_cu_share")

    def kernel_name(self):
        return lib().jsdr_fft_kernel(self.h).decode()

    def last_launch(self):
        a, b = C.c_int64(), C.c_int64()
        _check(lib().jsdr_fft_last_launch(self.h, C.byref(a), C.byref(b)), "jsdr_fft_last_launch")
        return a.value, b.value

    def batch_i16(self, raw_dev, nframes, psd_dev, ic=0, qc=0, stream=None):
        _check(lib().jsdr_fft_batch_i16(self.h, _addr(raw_dev), C.c_int64(nframes), ic, qc, _addr(psd_dev),
                                        C.c_void_p(stream)), "jsdr_fft_batch_i16")

    def batch_f32(self, iq_dev, nframes, psd_dev, stream=None):
        _check(lib().jsdr_fft_batch_f32(self.h, _addr(iq_dev), C.c_int64(nframes), _addr(psd_dev),
                                        C.c_void_p(stream)), "jsdr_fft_batch_f32")

    def spectrum(self, bufs):
        """complex spectra of a [nframes, 2n] float32 array (for the 1e-5 parity tests)"""
        bufs = np.ascontiguousarray(bufs, np.float32).reshape(-1, 2 * self.n)
        d_in = DeviceBuffer.from_host(bufs)
        d_out = DeviceBuffer(bufs.nbytes)
        _check(lib().jsdr_fft_spectrum_f32(self.h, _addr(d_in), C.c_int64(bufs.shape[0]), _addr(d_out), None),
               "jsdr_fft_spectrum_f32")
        return d_out.to_host(np.float32).reshape(bufs.shape)

    def batch_host_i16(self, raw, ic=0, qc=0):
        raw = np.ascontiguousarray(raw, np.int16).reshape(-1, 2 * self.n)
        d_in = DeviceBuffer.from_host(raw)
        d_out = DeviceBuffer(4 * raw.shape[0] * (self.n + 2))
        self.batch_i16(d_in, raw.shape[0], d_out, ic, qc)
        return d_out.to_host(np.float32).reshape(raw.shape[0], self.n + 2)

    def __del__(self):
        try:
            lib().jsdr_fft_destroy(self.h)
        except Exception:
            pass


# ------------------------------------------------------------------ phase.java
def phase_maxabs(bufs, n):
    bufs = np.ascontiguousarray(bufs, np.float32).reshape(-1, 2 * n)
    d_in = DeviceBuffer.from_host(bufs)
    d_out = DeviceBuffer(4 * bufs.shape[0])
    _check(lib().jsdr_phase_maxabs(_addr(d_in), C.c_int64(bufs.shape[0]), n, _addr(d_out), None), "jsdr_phase_maxabs")
    return d_out.to_host(np.float32)


def phase_columns(buf, bx):
    buf = np.ascontiguousarray(buf, np.float32)
    n = buf.size // 2
    d_in = DeviceBuffer.from_host(buf)
    cap = n + 1
    pix = np.empty(cap, np.int32)
    ai = np.empty(cap, np.float32)
    aq = np.empty(cap, np.float32)
    ncol = C.c_int()
    _check(lib().jsdr_phase_columns(_addr(d_in), n, bx, _addr(pix), _addr(ai), _addr(aq), cap, C.byref(ncol)),
           "jsdr_phase_columns")
    k = ncol.value
    return pix[:k].copy(), ai[:k].copy(), aq[:k].copy()


class Phase:
    """phase.java as a handle (the IAudioHandler drop-in): receive() keeps the frame on the device and takes max|x|
    (phase.java:75-80,123-128); columns(bx) are the painter's per-pixel-column means (:93-116)"""

    def __init__(self, n):
        self.n = n
        self.h = C.c_void_p()
        _check(lib().jsdr_phase_create(C.byref(self.h), n), "jsdr_phase_create")

    def receive(self, buf):
        buf = np.ascontiguousarray(buf, np.float32)
        assert buf.size == 2 * self.n
        _check(lib().jsdr_phase_receive_f32(self.h, _addr(buf)), "jsdr_phase_receive_f32")

    def max(self):
        m = C.c_float()
        _check(lib().jsdr_phase_get_max(self.h, C.byref(m)), "jsdr_phase_get_max")
        return np.float32(m.value)

    def columns(self, bx):
        cap = self.n + 1
        pix = np.empty(cap, np.int32)
        ai = np.empty(cap, np.float32)
        aq = np.empty(cap, np.float32)
        ncol = C.c_int()
        _check(lib().jsdr_phase_get_columns(self.h, bx, _addr(pix), _addr(ai), _addr(aq), cap, C.byref(ncol)),
               "jsdr_phase_get_columns")
        k = ncol.value
        return pix[:k].copy(), ai[:k].copy(), aq[:k].copy()

    def __del__(self):
        try:
            lib().jsdr_phase_destroy(self.h)
        except Exception:
            pass


# ------------------------------------------------------------------ fir.java
class Fir:
    def __init__(self, rate=44100.0):
        self.h = C.c_void_p()
        _check(lib().jsdr_fir_create(C.byref(self.h), C.c_float(rate)), "jsdr_fir_create")

    def weights(self, f1, f2):
        w = np.empty(21, np.float64)
        _check(lib().jsdr_fir_weights(self.h, f1, f2, _addr(w)), "jsdr_fir_weights")
        return w

    def filter_block(self, xs):
        xs = np.ascontiguousarray(xs, np.int32)
        out = np.empty_like(xs)
        _check(lib().jsdr_fir_filter(self.h, _addr(xs), _addr(out), C.c_int64(xs.size)), "jsdr_fir_filter")
        return out

    def complex_gen(self, freq, count, start=0):
        out = np.empty((count, 2), np.int32)
        _check(lib().jsdr_fir_complex_gen(self.h, freq, start, _addr(out), C.c_int64(count)), "jsdr_fir_complex_gen")
        return out

    def complex_mod(self, a, b):
        a = np.ascontiguousarray(a, np.int32)
        b = np.ascontiguousarray(b, np.int32)
        out = np.empty_like(a)
        _check(lib().jsdr_fir_complex_mod(self.h, _addr(a), _addr(b), _addr(out), C.c_int64(a.shape[0])),
               "jsdr_fir_complex_mod")
        return out

    def __del__(self):
        try:
            lib().jsdr_fir_destroy(self.h)
        except Exception:
            pass


def fir_batch_decimate_i16(raw_dev, nstreams, stride_i16, nsamples, taps, decim, scale, out_dev, out_stride_pairs, stream=None):
    """batched FIR + decimate (BASELINE config 3); returns the number of outputs per stream"""
    taps = np.ascontiguousarray(taps, np.float64)
    nout = C.c_int64()
    _check(lib().jsdr_fir_batch_decimate_i16(_addr(raw_dev), nstreams, C.c_int64(stride_i16), C.c_int64(nsamples), _addr(taps),
                                             int(taps.size), int(decim), C.c_double(scale), _addr(out_dev),
                                             C.c_int64(out_stride_pairs), C.byref(nout), C.c_void_p(stream)),
           "jsdr_fir_batch_decimate_i16")
    return nout.value


# ------------------------------------------------------------------ FECDecoder.java
def fec_decode(raw, out_init=None):
    raw = np.ascontiguousarray(raw, np.uint8)
    assert raw.size == 5200
    out = np.zeros(256, np.uint8) if out_init is None else np.array(out_init, np.uint8)
    rc = C.c_int()
    _check(lib().jsdr_fec_decode(_addr(raw), _addr(out), C.byref(rc)), "jsdr_fec_decode")
    return rc.value, out


def fec_encode(data):
    data = np.ascontiguousarray(data, np.uint8)
    assert data.size == 256
    sym = np.empty(5200, np.uint8)
    _check(lib().jsdr_fec_encode(_addr(data), _addr(sym)), "jsdr_fec_encode")
    return sym


def fec_decode_batch(raws):
    raws = np.ascontiguousarray(raws, np.uint8).reshape(-1, 5200)
    nb = raws.shape[0]
    d_raw = DeviceBuffer.from_host(raws)
    d_out = DeviceBuffer(256 * nb)
    d_out.zero()
    d_rc = DeviceBuffer(4 * nb)
    _check(lib().jsdr_fec_decode_batch(_addr(d_raw), C.c_int64(nb), _addr(d_out), _addr(d_rc), None),
           "jsdr_fec_decode_batch")
    return d_rc.to_host(np.int32), d_out.to_host(np.uint8).reshape(nb, 256)


def fec_encode_batch(datas):
    datas = np.ascontiguousarray(datas, np.uint8).reshape(-1, 256)
    nb = datas.shape[0]
    d_in = DeviceBuffer.from_host(datas)
    d_out = DeviceBuffer(5200 * nb)
    _check(lib().jsdr_fec_encode_batch(_addr(d_in), C.c_int64(nb), _addr(d_out), None), "jsdr_fec_encode_batch")
    return d_out.to_host(np.uint8).reshape(nb, 5200)


def fec_encode_dev(data_dev, nblocks, sym_dev, stream=None):
    """device-resident form: data_dev[nblocks][256] -> sym_dev[nblocks][5200]"""
    _check(lib().jsdr_fec_encode_batch(_addr(data_dev), C.c_int64(nblocks), _addr(sym_dev), C.c_void_p(stream)),
           "jsdr_fec_encode_batch")


def fec_decode_dev(raw_dev, nblocks, out_dev, rc_dev, stream=None):
    _check(lib().jsdr_fec_decode_batch(_addr(raw_dev), C.c_int64(nblocks), _addr(out_dev), _addr(rc_dev),
                                       C.c_void_p(stream)), "jsdr_fec_decode_batch")


# ------------------------------------------------------------------ FUNcubeBPSKDemod.java
COUNTER_NAMES = ["cntRaw", "cntDS", "cntBit", "cntFEC", "cntDec", "dmErrBits", "dmCorr", "dmMaxCorr", "decodeOK",
                 "centreBin"]


class BpskSnapshot(C.Structure):
    _fields_ = [("frames", C.c_int64), ("counters", C.c_int32 * 10), ("nbits", C.c_int32), ("state", C.c_double * 18),
                ("decoded", C.c_uint8 * 256), ("bits", C.c_int8 * 512)]


def bpsk_table(which):
    out = np.empty(65, np.float64)
    _check(lib().jsdr_bpsk_table(which, _addr(out), 65), "jsdr_bpsk_table")
    return out[:27 if which == 0 else 65].copy()


class Bpsk:
    """`nstreams` lock-step FUNcubeBPSKDemod instances (FUNcubeBPSKDemod.java:357-595)."""

    def __init__(self, rate=96000, blen=8192, size=4, tuning=12000, do_fft=0, do_up=0, nstreams=1,
                 max_batch_samples=None, variant="exact"):
        self.samples = blen // size
        self.nstreams = nstreams
        self.max_batch = max_batch_samples or self.samples
        self.h = C.c_void_p()
        _check(lib().jsdr_bpsk_create(C.byref(self.h), rate, self.samples, tuning, do_fft, do_up, nstreams,
                                      C.c_int64(self.max_batch)), "jsdr_bpsk_create")
        if variant != "exact":
            _check(lib().jsdr_bpsk_set_variant(self.h, {"exact": 0, "fast": 1}[variant]), "jsdr_bpsk_set_variant")

    def snapshot(self):
        """results of the last completed receive(); callable from any thread while another one is inside receive()"""
        sn = BpskSnapshot()
        _check(lib().jsdr_bpsk_snapshot_read(self.h, C.byref(sn)), "jsdr_bpsk_snapshot_read")
        return sn

    def cert_stats(self):
        r, u, e = C.c_int64(), C.c_int64(), C.c_double()
        _check(lib().jsdr_bpsk_cert_stats(self.h, C.byref(r), C.byref(u), C.byref(e)), "jsdr_bpsk_cert_stats")
        return dict(decisions_redone_exactly=r.value, streams_uncertified=u.value, fi_fq_error_bound=e.value)

    def uncertified_streams(self):
        """fast variant: ids of the streams whose results are withheld (jsdr_hip.h); [] for the exact variant"""
        n = C.c_int()
        _check(lib().jsdr_bpsk_uncertified_streams(self.h, None, 0, C.byref(n)), "jsdr_bpsk_uncertified_streams")
        ids = np.empty(max(n.value, 1), np.int32)
        _check(lib().jsdr_bpsk_uncertified_streams(self.h, _addr(ids), n.value, C.byref(n)), "jsdr_bpsk_uncertified_streams")
        return [int(v) for v in ids[:n.value]]

    def schedule_stats(self):
        a, b = C.c_int64(), C.c_int64()
        _check(lib().jsdr_bpsk_schedule_stats(self.h, C.byref(a), C.byref(b)), "jsdr_bpsk_schedule_stats")
        return dict(computed_inline=a.value, prefetched=b.value)

    def front_kernel_name(self):
        return lib().jsdr_bpsk_front_kernel(self.h).decode()

    def tail_kernel_name(self):
        return lib().jsdr_bpsk_tail_kernel(self.h).decode()

    def fec_kernel_name(self):
        return lib().jsdr_bpsk_fec_kernel(self.h).decode()

    def side_stream(self):
        on = C.c_int()
        _check(lib().jsdr_bpsk_side_stream(self.h, C.byref(on)), "jsdr_bpsk_side_stream")
        return bool(on.value)

    def receive(self, buf):
        buf = np.ascontiguousarray(buf, np.float32)
        assert buf.size == 2 * self.samples
        _check(lib().jsdr_bpsk_receive_f32(self.h, _addr(buf)), "jsdr_bpsk_receive_f32")

    def receive_raw(self, raw, ic=0, qc=0):
        raw = np.ascontiguousarray(raw, np.int16)
        assert raw.size == 2 * self.samples
        _check(lib().jsdr_bpsk_receive_i16(self.h, _addr(raw), ic, qc), "jsdr_bpsk_receive_i16")

    def batch_i16(self, raw_dev, stride_i16, nsamples, ic=0, qc=0, stream=None):
        _check(lib().jsdr_bpsk_batch_i16(self.h, _addr(raw_dev), C.c_int64(stride_i16), C.c_int64(nsamples), ic, qc,
                                         C.c_void_p(stream)), "jsdr_bpsk_batch_i16")

    def is_recovered(self, stream):
        """fast variant: this stream is served by the exact shadow handle (jsdr_bpsk_recover_uncertified replayed it)"""
        v = C.c_int()
        _check(lib().jsdr_bpsk_stream_recovered(self.h, int(stream), C.byref(v)), "jsdr_bpsk_stream_recovered")
        return bool(v.value)

    def recover_uncertified(self, raw_devs, nsamples, stride_i16, ic=0, qc=0, stream=None):
        """fast variant: replay the streams the calls so far left uncertified on an internal exact handle that serves them from
        then on; raw_devs / nsamples: every call since creation.  Returns the number of streams now served in exact order."""
        n = len(raw_devs)
        ptrs = (C.c_void_p * max(n, 1))(*[_addr(r) for r in raw_devs])
        ns = (C.c_int64 * max(n, 1))(*[int(v) for v in nsamples])
        rec = C.c_int()
        _check(lib().jsdr_bpsk_recover_uncertified(self.h, ptrs, ns, n, C.c_int64(stride_i16), ic, qc, C.byref(rec), C.c_void_p(stream)),
               "jsdr_bpsk_recover_uncertified")
        return rec.value

    def set_cu_share(self, wgs_per_cu):
        _check(lib().jsdr_bpsk_set_cu_share(self.h, int(wgs_per_cu)), "jsdr_bpsk_set_cu_share")

    def pair_shares(self):
        """(fft share, bpsk share) the library recommends for running fft.receive beside this demodulator; (0, 0): one after the other"""
        a, b = C.c_int(), C.c_int()
        _check(lib().jsdr_bpsk_pair_shares(self.h, C.byref(a), C.byref(b)), "jsdr_bpsk_pair_shares")
        return a.value, b.value

    def last_launch(self):
        a, b = C.c_int64(), C.c_int64()
        _check(lib().jsdr_bpsk_last_launch(self.h, C.byref(a), C.byref(b)), "jsdr_bpsk_last_launch")
        return a.value, b.value

    def sync(self):
        _check(lib().jsdr_bpsk_sync(self.h), "jsdr_bpsk_sync")

    def counters(self, stream=0):
        out = np.empty(10, np.int32)
        _check(lib().jsdr_bpsk_get_counters(self.h, stream, _addr(out)), "jsdr_bpsk_get_counters")
        return dict(zip(COUNTER_NAMES, (int(v) for v in out)))

    def bits(self, stream=0):
        n = C.c_int()
        _check(lib().jsdr_bpsk_get_bits(self.h, stream, None, 0, C.byref(n)), "jsdr_bpsk_get_bits")
        out = np.empty(max(n.value, 1), np.int8)
        _check(lib().jsdr_bpsk_get_bits(self.h, stream, _addr(out), n.value, C.byref(n)), "jsdr_bpsk_get_bits")
        return out[:n.value]

    def fec_results(self, stream=0):
        cnt = C.c_int()
        _check(lib().jsdr_bpsk_get_fec_count(self.h, stream, C.byref(cnt)), "jsdr_bpsk_get_fec_count")
        res = []
        for i in range(cnt.value):
            rc = C.c_int32()
            bi = C.c_int32()
            out = np.empty(256, np.uint8)
            _check(lib().jsdr_bpsk_get_fec(self.h, stream, i, C.byref(rc), C.byref(bi), _addr(out)),
                   "jsdr_bpsk_get_fec")
            res.append((rc.value, bi.value, out))
        return res

    def decoded(self, stream=0):
        out = np.empty(256, np.uint8)
        _check(lib().jsdr_bpsk_get_decoded(self.h, stream, _addr(out)), "jsdr_bpsk_get_decoded")
        return out

    def trace(self, stream=0):
        n = C.c_int64()
        _check(lib().jsdr_bpsk_get_trace(self.h, stream, None, C.c_int64(0), C.byref(n)), "jsdr_bpsk_get_trace")
        out = np.empty((max(n.value, 1), 2), np.float64)
        _check(lib().jsdr_bpsk_get_trace(self.h, stream, _addr(out), n, C.byref(n)), "jsdr_bpsk_get_trace")
        return out[:n.value]

    def state(self, stream=0):
        out = np.empty(18, np.float64)
        _check(lib().jsdr_bpsk_get_state(self.h, stream, _addr(out)), "jsdr_bpsk_get_state")
        return out

    def profile_enable(self, on=True):
        _check(lib().jsdr_bpsk_profile_enable(self.h, int(on)), "jsdr_bpsk_profile_enable")

    def profile_read(self):
        """{kernel name: (total ms, launches)} since the last read"""
        k = lib().jsdr_bpsk_profile_count()
        ms = np.zeros(k, np.float64)
        cnt = np.zeros(k, np.int32)
        _check(lib().jsdr_bpsk_profile_read(self.h, _addr(ms), _addr(cnt)), "jsdr_bpsk_profile_read")
        return {lib().jsdr_bpsk_profile_name(i).decode(): (float(ms[i]), int(cnt[i])) for i in range(k)}

    def slot_info(self):
        sb, bo, fo = C.c_int64(), C.c_int64(), C.c_int64()
        nb, nf = C.c_int(), C.c_int()
        _check(lib().jsdr_bpsk_slot_info(self.h, C.byref(sb), C.byref(bo), C.byref(fo), C.byref(nb), C.byref(nf)),
               "jsdr_bpsk_slot_info")
        return dict(slot_bytes=sb.value, bits_offset=bo.value, fec_offset=fo.value, slot_bits=nb.value,
                    nfec_max=nf.value)

    def pack_slots(self, slots_dev, stream=None):  # stream: raw hipStream_t value
        _check(lib().jsdr_bpsk_pack_slots(self.h, _addr(slots_dev), C.c_void_p(stream)), "jsdr_bpsk_pack_slots")

    def __del__(self):
        try:
            if getattr(self, "borrowed", False):
                return
            lib().jsdr_bpsk_destroy(self.h)
        except Exception:
            pass


class Group:
    """jsdr_group_*: `total_streams` demodulators of ONE process over `ndev` devices, one host thread per device, the result
    slots gathered to every device after each call (RCCL, or device-to-device copies with gather_copy=True)."""

    def __init__(self, ndev, total_streams, max_batch_samples, devices=None, rate=96000, frame=2048, tuning=12000, do_fft=0,
                 do_up=0, gather_copy=False, with_psd=False):
        self.h = C.c_void_p()
        self.ndev = ndev
        devs = (C.c_int * ndev)(*devices) if devices is not None else None
        flags = (1 if gather_copy else 0) | (2 if with_psd else 0)
        _check(lib().jsdr_group_create(C.byref(self.h), ndev, devs, rate, frame, tuning, do_fft, do_up, total_streams,
                                       C.c_int64(max_batch_samples), flags), "jsdr_group_create")
        n, s, sb, v = C.c_int(), C.c_int(), C.c_int64(), C.c_int()
        _check(lib().jsdr_group_info(self.h, C.byref(n), C.byref(s), C.byref(sb), C.byref(v)), "jsdr_group_info")
        self.streams_per_device, self.slot_bytes, self.rccl_version = s.value, sb.value, v.value
        self.total_streams = total_streams

    def device(self, index):
        """(device ordinal, a borrowed Bpsk view of that device's handle)"""
        d, dem, fft = C.c_int(), C.c_void_p(), C.c_void_p()
        _check(lib().jsdr_group_device(self.h, index, C.byref(d), C.byref(dem), C.byref(fft)), "jsdr_group_device")
        view = Bpsk.__new__(Bpsk)
        view.h, view.nstreams, view.borrowed = dem, self.streams_per_device, True
        return d.value, view

    def batch_i16(self, raw_devs, stride_i16, nsamples, ic=0, qc=0, psd_devs=None):
        raws = (C.c_void_p * self.ndev)(*[_addr(r) for r in raw_devs])
        psds = (C.c_void_p * self.ndev)(*[_addr(p) for p in psd_devs]) if psd_devs is not None else None
        _check(lib().jsdr_group_batch_i16(self.h, raws, C.c_int64(stride_i16), C.c_int64(nsamples), ic, qc, psds),
               "jsdr_group_batch_i16")

    def sync(self):
        _check(lib().jsdr_group_sync(self.h), "jsdr_group_sync")

    def read_slot(self, index, stream):
        out = np.empty(self.slot_bytes, np.uint8)
        _check(lib().jsdr_group_read_slot(self.h, index, stream, _addr(out)), "jsdr_group_read_slot")
        return out

    def gathered(self, index):
        """device `index`'s gathered buffer as a host array [total_streams][slot_bytes] (the device must be current-able)"""
        p, nb = C.c_void_p(), C.c_int64()
        self.sync()
        _check(lib().jsdr_group_gathered(self.h, index, C.byref(p), C.byref(nb)), "jsdr_group_gathered")
        dev, _ = self.device(index)
        out = np.empty(nb.value, np.uint8)
        _check(lib().jsdr_set_device(dev), "jsdr_set_device")
        _check(lib().jsdr_memcpy_d2h(_addr(out), p, C.c_size_t(nb.value)), "jsdr_memcpy_d2h")
        return out.reshape(self.total_streams, self.slot_bytes)

    def __del__(self):
        try:
            lib().jsdr_group_destroy(self.h)
        except Exception:
            pass


def unpack_slot(slot, info):
    """decode one result slot (bytes) -> dict(counters, bits, fec)"""
    hdr = np.frombuffer(slot[:64], np.int32)
    nb, nt = int(hdr[0]), int(hdr[1])
    bits = np.frombuffer(slot[info["bits_offset"]:info["bits_offset"] + nb], np.int8)
    fec = []
    for t in range(nt):
        o = info["fec_offset"] + t * 264
        rc, bi = np.frombuffer(slot[o:o + 8], np.int32)
        fec.append((int(rc), int(bi), np.frombuffer(slot[o + 8:o + 264], np.uint8)))
    names = ["nbits", "nfec"] + COUNTER_NAMES[:9]
    return dict(header=dict(zip(names, (int(v) for v in hdr[:11]))), bits=bits, fec=fec)


# ------------------------------------------------------------------ demod.java
class Demod:
    """demod.receive batched over streams: mode 0 OFF, 1 RAW, 2 AM, 3 NFM, 4 WFM (demod.java:39-43)"""

    def __init__(self, rate=96000, n=2048, nstreams=1, max_batch_samples=None):
        self.rate, self.n, self.S = rate, n, nstreams
        self.max_batch = max_batch_samples or n
        self.h = C.c_void_p()
        _check(lib().jsdr_demod_create(C.byref(self.h), rate, n, nstreams, C.c_int64(self.max_batch)), "jsdr_demod_create")

    def configure(self, mode, dofir=0, dodwn=0, doagc=0):
        _check(lib().jsdr_demod_configure(self.h, mode, dofir, dodwn, doagc), "jsdr_demod_configure")

    def weights(self, flo, fhi):
        w = np.empty(21, np.float32)
        phi = C.c_float()
        _check(lib().jsdr_demod_weights(self.h, flo, fhi, _addr(w), C.byref(phi)), "jsdr_demod_weights")
        return w, np.float32(phi.value)

    def batch_i16(self, raw_dev, stride_i16, nsamples, audio_dev, audio_stride_i16, ic=0, qc=0, stream=None):
        _check(lib().jsdr_demod_batch_i16(self.h, _addr(raw_dev), C.c_int64(stride_i16), C.c_int64(nsamples), ic, qc,
                                          _addr(audio_dev), C.c_int64(audio_stride_i16), C.c_void_p(stream)),
               "jsdr_demod_batch_i16")

    def batch_host_i16(self, raw, nsamples, ic=0, qc=0):
        """raw: int16 [S][2*nsamples] on the host -> audio int16 [S][2*nsamples]"""
        raw = np.ascontiguousarray(raw, np.int16).reshape(self.S, 2 * nsamples)
        d_in = DeviceBuffer.from_host(raw)
        d_out = DeviceBuffer(raw.nbytes)
        self.batch_i16(d_in, 2 * nsamples, nsamples, d_out, 2 * nsamples, ic, qc)
        return d_out.to_host(np.int16).reshape(self.S, 2 * nsamples)

    def receive(self, buf):
        buf = np.ascontiguousarray(buf, np.float32)
        assert buf.size == 2 * self.n
        out = np.empty(2 * self.n, np.int16)
        _check(lib().jsdr_demod_receive_f32(self.h, _addr(buf), _addr(out)), "jsdr_demod_receive_f32")
        return out

    def frame_stats(self, stream=0):
        mx, av = C.c_float(), C.c_float()
        _check(lib().jsdr_demod_frame_stats(self.h, stream, C.byref(mx), C.byref(av)), "jsdr_demod_frame_stats")
        return np.float32(mx.value), np.float32(av.value)

    def state(self):
        car, phi = C.c_float(), C.c_float()
        _check(lib().jsdr_demod_state(self.h, C.byref(car), C.byref(phi)), "jsdr_demod_state")
        return np.float32(car.value), np.float32(phi.value)

    def profile_enable(self, on):
        _check(lib().jsdr_demod_profile_enable(self.h, int(on)), "jsdr_demod_profile_enable")

    def profile_read(self):
        k = lib().jsdr_demod_profile_count()
        ms = np.zeros(k, np.float64)
        cnt = np.zeros(k, np.int32)
        _check(lib().jsdr_demod_profile_read(self.h, _addr(ms), _addr(cnt)), "jsdr_demod_profile_read")
        return {lib().jsdr_demod_profile_name(i).decode(): (float(ms[i]), int(cnt[i])) for i in range(k)}

    def __del__(self):
        try:
            lib().jsdr_demod_destroy(self.h)
        except Exception:
            pass


# ------------------------------------------------------------------ formats either side
class RecordingInfo(C.Structure):
    _fields_ = [("format", C.c_int), ("encoding", C.c_int), ("channels", C.c_int), ("rate", C.c_int),
                ("bits", C.c_int), ("frames", C.c_int64), ("data_offset", C.c_int64)]


def waterfall_lines(psd, n, width, peak_rgb=0x00FFFF):
    """waterfall.paintLine for every frame of psd [nframes][n+2] -> uint32 ARGB [nframes][width]"""
    psd = np.ascontiguousarray(psd, np.float32).reshape(-1, n + 2)
    d_in = DeviceBuffer.from_host(psd)
    d_out = DeviceBuffer(4 * psd.shape[0] * width)
    _check(lib().jsdr_waterfall_lines(_addr(d_in), C.c_int64(psd.shape[0]), n, width, C.c_uint32(peak_rgb),
                                      _addr(d_out), None), "jsdr_waterfall_lines")
    return d_out.to_host(np.uint32).reshape(psd.shape[0], width)


def waterfall_lines_dev(psd_dev, nframes, n, width, pix_dev, peak_rgb=0x00FFFF, stream=None):
    _check(lib().jsdr_waterfall_lines(_addr(psd_dev), C.c_int64(nframes), n, width, C.c_uint32(peak_rgb),
                                      _addr(pix_dev), stream), "jsdr_waterfall_lines")


def recording_probe(path, raw_channels=2):
    info = RecordingInfo()
    _check(lib().jsdr_recording_probe(os.fsencode(path), raw_channels, C.byref(info)), "jsdr_recording_probe")
    return info


def recordings_load(paths, channels, rate, first_frame, nframes, raw_dev, stream_stride_i16):
    """files -> raw_dev[s*stride ...] as int16 (I,Q) pairs; returns the frames really read per stream"""
    arr = (C.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
    got = (C.c_int64 * len(paths))()
    _check(lib().jsdr_recordings_load(arr, len(paths), channels, rate, C.c_int64(first_frame), C.c_int64(nframes),
                                      _addr(raw_dev), C.c_int64(stream_stride_i16), got, None), "jsdr_recordings_load")
    return list(got)


# ------------------------------------------------------------------ synthetic inputs
def synth_mix64(z):
    """splitmix64 finaliser: the counter hash of the synthetic-input generators (csrc/synth.hip)"""
    m = 0xFFFFFFFFFFFFFFFF
    z = (z + 0x9E3779B97F4A7C15) & m
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
    return z ^ (z >> 31)


def synth_carrier_tables(amp):
    """1024-entry int16 cos / sin tables of amplitude amp (round half to even), the carrier of jsdr_synth_dbpsk / _tones"""
    a = 2.0 * np.pi * np.arange(1024, dtype=np.float64) / 1024.0
    return np.rint(amp * np.cos(a)).astype(np.int16), np.rint(amp * np.sin(a)).astype(np.int16)


def synth_phase_inc_u32(freq_hz, rate):
    return int(round(freq_hz / rate * 2 ** 32)) & 0xFFFFFFFF


def synth_payloads(seed, stream0, nstreams, nframes, out_dev=None, stream=None):
    own = out_dev is None
    if own:
        out_dev = DeviceBuffer(nstreams * nframes * 256)
    _check(lib().jsdr_synth_payloads(C.c_uint64(seed), stream0, nstreams, nframes, _addr(out_dev), C.c_void_p(stream)),
           "jsdr_synth_payloads")
    return out_dev


def synth_diffsign(sym_dev, nsym, nstreams, dsign_dev, stream=None):
    _check(lib().jsdr_synth_diffsign(_addr(sym_dev), C.c_int64(nsym), nstreams, _addr(dsign_dev), C.c_void_p(stream)),
           "jsdr_synth_diffsign")


def synth_dbpsk(out_dev, stride_i16, nstreams, n0, n, dsign_dev, nsym, sps, phase0, phase_inc, cos_dev, sin_dev,
                noise_gain, keys_dev, stream=None):
    _check(lib().jsdr_synth_dbpsk(_addr(out_dev), C.c_int64(stride_i16), nstreams, C.c_int64(n0), C.c_int64(n),
                                  _addr(dsign_dev), C.c_int64(nsym), sps, C.c_uint32(phase0), C.c_uint32(phase_inc),
                                  _addr(cos_dev), _addr(sin_dev), noise_gain, _addr(keys_dev), C.c_void_p(stream)),
           "jsdr_synth_dbpsk")


def synth_tones(out_dev, frame0, nframes, n, cos_dev, noise_gain, key, stream=None):
    _check(lib().jsdr_synth_tones(_addr(out_dev), C.c_int64(frame0), C.c_int64(nframes), n, _addr(cos_dev), noise_gain,
                                  C.c_uint64(key), C.c_void_p(stream)), "jsdr_synth_tones")
