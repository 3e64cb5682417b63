"""ctypes binding of oracle/libjsdr_oracle.so -- the CPU oracle (test infrastructure).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
SO = os.path.join(ODIR, "libjsdr_oracle.so")


def build(force=False):
    if os.environ.get("JSDR_ORACLE_SO"):  # another build of the same sources (the sanitizer build: `make -C oracle asan`)
        return os.environ["JSDR_ORACLE_SO"]
    srcs = [os.path.join(ODIR, f) for f in os.listdir(ODIR) if f.endswith((".c", ".h"))]
    stale = force or not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in srcs)
    if stale:
        # into a file of this process's own, then renamed over the target in one step: the ranks of an N > 1 bench run (and tests
        # under pytest-xdist) may all find the library stale at once, and none of them may load a file another is still writing
        tmp = "libjsdr_oracle.%d.tmp.so" % os.getpid()
        try:
            subprocess.check_call(["make", "-s", "-C", ODIR, "-B", "OUT=" + tmp, tmp])
            os.replace(os.path.join(ODIR, tmp), SO)
        finally:
            if os.path.exists(os.path.join(ODIR, tmp)):
                os.remove(os.path.join(ODIR, tmp))
    return SO


_lib = None
_lib_lock = threading.Lock()


def lib():
    """the library, built if stale, with every prototype set BEFORE it becomes visible: bench.py replays streams through the oracle
    from several threads, and a thread that found the handle between CDLL() and _proto() called jo_* with ctypes' default int
    arguments -- 64-bit pointers truncated, a segmentation fault in one run of thirty (round 6)"""
    global _lib
    if _lib is None:
        with _lib_lock:
            if _lib is None:
                L = C.CDLL(build())
                _proto(L)
                _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _proto(L):
    vp = C.c_void_p
    L.jo_bpsk_new.restype = vp
    L.jo_bpsk_new.argtypes = [C.c_int] * 6
    L.jo_bpsk_free.argtypes = [vp]
    L.jo_bpsk_receive.argtypes = [vp, vp]
    L.jo_bpsk_receive_i16.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int]
    L.jo_bpsk_counters.argtypes = [vp, vp]
    L.jo_bpsk_bits.restype = C.c_int64
    L.jo_bpsk_bits.argtypes = [vp, vp, C.c_int64]
    L.jo_bpsk_fec_count.argtypes = [vp]
    L.jo_bpsk_fec_get.argtypes = [vp, C.c_int, vp, vp, vp]
    L.jo_bpsk_decoded.argtypes = [vp, vp]
    L.jo_bpsk_trace_enable.argtypes = [vp, C.c_int64]
    L.jo_bpsk_trace.restype = C.c_int64
    L.jo_bpsk_trace.argtypes = [vp, vp, C.c_int64]
    L.jo_bpsk_trace_ds.restype = C.c_int64
    L.jo_bpsk_trace_ds.argtypes = [vp, vp, C.c_int64]
    L.jo_bpsk_state.argtypes = [vp, vp]
    L.jo_bpsk_istate.argtypes = [vp, vp]
    L.jo_bpsk_table.argtypes = [C.c_int, vp, C.c_int]
    L.jo_bpsk_sincos.argtypes = [vp, vp]
    L.jo_fec_decode.argtypes = [vp, vp]
    L.jo_fec_encode.argtypes = [vp, vp]
    L.jo_fec_table.argtypes = [C.c_char_p, vp, C.c_int]
    L.jo_viterbi27.argtypes = [vp, vp, C.c_int]
    L.jo_decode_rs_8.argtypes = [vp, vp, C.c_int]
    L.jo_convert_i16.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.jo_fft_f32.argtypes = [vp, C.c_int]
    L.jo_fft_f64.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.jo_fft_twiddles_f64.argtypes = [vp, C.c_int]
    L.jo_dft_exact.argtypes = [vp, C.c_int, vp]
    L.jo_fft_receive.argtypes = [vp, C.c_int, C.c_int, vp]
    L.jo_fft_psd_from_spectrum.argtypes = [vp, C.c_int, C.c_int, vp]
    L.jo_fir_init.argtypes = [vp]
    L.jo_fir_weights.argtypes = [vp, C.c_int, C.c_int, C.c_float]
    L.jo_fir_filter.argtypes = [vp, C.c_int]
    L.jo_fir_complex_gen.argtypes = [vp, vp, C.c_float]
    L.jo_fir_complex_mod.argtypes = [vp, vp, vp]
    L.jo_fir_decimate.argtypes = [vp, C.c_int64, vp, C.c_int, C.c_int, C.c_double, vp]
    L.jo_fir_decimate.restype = C.c_int64
    L.jo_phase_maxabs.restype = C.c_float
    L.jo_phase_maxabs.argtypes = [vp, C.c_int]
    L.jo_phase_columns.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp]
    L.jo_waterfall_line.restype = None
    for fn in (L.jo_demod_init, L.jo_demod_weights, L.jo_demod_receive):
        fn.restype = None
    L.jo_demod_init.argtypes = [vp, C.c_int]
    L.jo_demod_weights.argtypes = [vp]
    L.jo_demod_filter_move.argtypes = [vp, C.c_int, C.c_int]
    L.jo_demod_receive.argtypes = [vp, vp, C.c_int, vp]
    L.jo_waterfall_line.argtypes = [vp, C.c_int, C.c_int, C.c_uint, vp]
    L.jo_mix64.restype = C.c_uint64
    L.jo_mix64.argtypes = [C.c_uint64]
    L.jo_synth_payload.argtypes = [C.c_uint64, C.c_int, C.c_int, vp]
    L.jo_synth_diffsign.argtypes = [vp, C.c_int64, vp, C.c_int8]
    L.jo_synth_tables.argtypes = [C.c_int, vp, vp]
    L.jo_synth_dbpsk.argtypes = [vp, C.c_int64, C.c_int64, vp, C.c_int64, C.c_int, C.c_uint32, C.c_uint32,
                                 vp, vp, C.c_int, C.c_uint64]
    L.jo_synth_tones.argtypes = [vp, C.c_int64, C.c_int64, C.c_int, vp, C.c_int, C.c_uint64]
    L.jo_bench_fft.argtypes = [vp, C.c_int64, C.c_int, C.c_int, vp]


def ptr(a):
    return a.ctypes.data if a is not None else None


# ---------------------------------------------------------------- conversion / FFT
def convert_i16(raw, chns=2, ic=0, qc=0):
    raw = np.ascontiguousarray(raw, dtype=np.int16)
    nframes = raw.size // chns
    out = np.empty(2 * nframes, np.float32)
    lib().jo_convert_i16(ptr(raw), nframes, chns, ic, qc, ptr(out))
    return out


def fft_receive(buf, rate):
    buf = np.ascontiguousarray(buf, dtype=np.float32)
    n = buf.size // 2
    psd = np.empty(n + 2, np.float32)
    lib().jo_fft_receive(ptr(buf), n, rate, ptr(psd))
    return psd


def fft_psd_from_spectrum(spec, rate):
    spec = np.ascontiguousarray(spec, dtype=np.float32)
    n = spec.size // 2
    psd = np.empty(n + 2, np.float32)
    lib().jo_fft_psd_from_spectrum(ptr(spec), n, rate, ptr(psd))
    return psd


def fft_f32(a):
    a = np.array(a, dtype=np.float32, copy=True)
    lib().jo_fft_f32(ptr(a), a.size // 2)
    return a


def fft_f64(a, inverse=False, scale=False):
    a = np.array(a, dtype=np.float64, copy=True)
    lib().jo_fft_f64(ptr(a), a.size // 2, int(inverse), int(scale))
    return a


def fft_twiddles_f64(n):
    w = np.empty(n, np.float64)
    lib().jo_fft_twiddles_f64(ptr(w), n)
    return w


def dft_exact(buf):
    buf = np.ascontiguousarray(buf, dtype=np.float32)
    n = buf.size // 2
    out = np.empty(2 * n, np.float64)
    lib().jo_dft_exact(ptr(buf), n, ptr(out))
    return out


# ---------------------------------------------------------------- fir.java
class JoFir(C.Structure):
    _fields_ = [("wfir", C.c_double * 21), ("fir", C.c_int * 21), ("fof", C.c_int)]


class Fir:
    def __init__(self):
        self.s = JoFir()
        lib().jo_fir_init(C.addressof(self.s))

    def weights(self, f1, f2, rate=44100.0):
        lib().jo_fir_weights(C.addressof(self.s), f1, f2, C.c_float(rate))
        return np.array(self.s.wfir[:], dtype=np.float64)

    def filter(self, x):
        return lib().jo_fir_filter(C.addressof(self.s), int(x))

    def filter_block(self, xs):
        f = lib().jo_fir_filter
        a = C.addressof(self.s)
        return np.array([f(a, int(v)) for v in xs], dtype=np.int32)


def fir_complex_gen(freq, count, rate=44100.0, start=0):
    sig = (C.c_int * 2)()
    wav = (C.c_int * 2)(freq, start)
    out = np.empty((count, 2), np.int32)
    for i in range(count):
        lib().jo_fir_complex_gen(sig, wav, C.c_float(rate))
        out[i, 0] = sig[0]
        out[i, 1] = sig[1]
    return out


def fir_decimate(raw, taps, decim, scale):
    """RxDownSample as an operator (FUNcubeBPSKDemod.java:466-492) over int16 IQ: [n // decim, 2] doubles"""
    raw = np.ascontiguousarray(raw, np.int16)
    taps = np.ascontiguousarray(taps, np.float64)
    n = raw.size // 2
    out = np.empty((max(n // decim, 1), 2), np.float64)
    no = lib().jo_fir_decimate(ptr(raw), n, ptr(taps), taps.size, decim, float(scale), ptr(out))
    assert no == n // decim, (no, n, decim)
    return out[:no]


def fir_complex_mod(a, b):
    a = np.ascontiguousarray(a, np.int32)
    b = np.ascontiguousarray(b, np.int32)
    out = np.empty_like(a)
    for i in range(a.shape[0]):
        lib().jo_fir_complex_mod(ptr(a[i]), ptr(b[i]), ptr(out[i]))
    return out


# ---------------------------------------------------------------- demod.java
class _JoDemod(C.Structure):
    _fields_ = [("rate", C.c_int), ("mode", C.c_int), ("dofir", C.c_int), ("dodwn", C.c_int), ("doagc", C.c_int),
                ("flo", C.c_int), ("fhi", C.c_int), ("fof", C.c_int), ("fir", C.c_float * 42), ("wfir", C.c_float * 21),
                ("car", C.c_float), ("phi", C.c_float), ("max", C.c_float), ("avg", C.c_float), ("li", C.c_float),
                ("lq", C.c_float), ("sam", C.c_float * (2 * 32768))]


class Demod:
    """one stream of demod.java (demod.java:341-483)"""

    def __init__(self, rate=96000):
        self.d = _JoDemod()
        lib().jo_demod_init(C.byref(self.d), rate)

    def configure(self, mode, dofir=0, dodwn=0, doagc=0):
        self.d.mode, self.d.dofir, self.d.dodwn, self.d.doagc = mode, dofir, dodwn, doagc

    def weights(self, flo, fhi):
        self.d.flo, self.d.fhi = flo, fhi
        lib().jo_demod_weights(C.byref(self.d))
        return np.array(self.d.wfir[:], np.float32), np.float32(self.d.phi)

    def filter_move(self, lo, hi):
        return bool(lib().jo_demod_filter_move(C.byref(self.d), lo, hi))

    def receive(self, buf):
        buf = np.ascontiguousarray(buf, np.float32)
        out = np.empty(buf.size, np.int16)
        lib().jo_demod_receive(C.byref(self.d), ptr(buf), buf.size, ptr(out))
        return out

    @property
    def max(self):
        return np.float32(self.d.max)

    @property
    def avg(self):
        return np.float32(self.d.avg)

    @property
    def car(self):
        return np.float32(self.d.car)


# ---------------------------------------------------------------- waterfall.java
def waterfall_line(psd, n, width, peak_rgb=0x00FFFF):
    psd = np.ascontiguousarray(psd, np.float32)
    assert psd.size == n + 2
    pix = np.empty(width, np.uint32)
    lib().jo_waterfall_line(ptr(psd), n, width, peak_rgb, ptr(pix))
    return pix


# ---------------------------------------------------------------- phase.java
def phase_maxabs(dpy):
    dpy = np.ascontiguousarray(dpy, np.float32)
    return float(lib().jo_phase_maxabs(ptr(dpy), dpy.size))


def phase_columns(dpy, bx):
    dpy = np.ascontiguousarray(dpy, np.float32)
    cap = dpy.size // 2 + 1
    pix = np.empty(cap, np.int32)
    ai = np.empty(cap, np.float32)
    aq = np.empty(cap, np.float32)
    n = lib().jo_phase_columns(ptr(dpy), dpy.size, bx, ptr(pix), ptr(ai), ptr(aq))
    return pix[:n].copy(), ai[:n].copy(), aq[:n].copy()


# ---------------------------------------------------------------- FEC
def fec_encode(data):
    data = np.ascontiguousarray(data, np.uint8)
    assert data.size == 256
    sym = np.empty(5200, np.uint8)
    lib().jo_fec_encode(ptr(data), ptr(sym))
    return sym


def fec_decode(raw):
    raw = np.ascontiguousarray(raw, np.uint8)
    assert raw.size == 5200
    out = np.zeros(256, np.uint8)
    rc = lib().jo_fec_decode(ptr(raw), ptr(out))
    return rc, out


def fec_table(name):
    out = np.empty(512, np.int32)
    n = lib().jo_fec_table(name.encode(), ptr(out), 512)
    assert n > 0, name
    return out[:n].copy()


def bpsk_table(which):
    out = np.empty(130, np.float64)
    n = lib().jo_bpsk_table(which, ptr(out), 130)
    return out[:n].copy()


def bpsk_sincos():
    s = np.empty(256, np.float64)
    c = np.empty(256, np.float64)
    lib().jo_bpsk_sincos(ptr(s), ptr(c))
    return s, c


# ---------------------------------------------------------------- BPSK demod
class Bpsk:
    def __init__(self, rate=96000, blen=8192, size=4, tuning=12000, do_fft=0, do_up=0, trace=0):
        self.h = lib().jo_bpsk_new(rate, blen, size, tuning, do_fft, do_up)
        self.samples = blen // size
        if trace:
            lib().jo_bpsk_trace_enable(self.h, trace)

    def __del__(self):
        try:
            lib().jo_bpsk_free(self.h)
        except Exception:
            pass

    def receive(self, buf):
        buf = np.ascontiguousarray(buf, np.float32)
        assert buf.size == 2 * self.samples
        lib().jo_bpsk_receive(self.h, ptr(buf))

    def receive_i16(self, raw, ic=0, qc=0):
        raw = np.ascontiguousarray(raw, np.int16)
        lib().jo_bpsk_receive_i16(self.h, ptr(raw), raw.size // 2, ic, qc)

    def counters(self):
        out = np.empty(10, np.int32)
        lib().jo_bpsk_counters(self.h, ptr(out))
        names = ["cntRaw", "cntDS", "cntBit", "cntFEC", "cntDec", "dmErrBits", "dmCorr", "dmMaxCorr",
                 "decodeOK", "centreBin"]
        return dict(zip(names, (int(v) for v in out)))

    def bits(self):
        n = lib().jo_bpsk_bits(self.h, None, 0)
        out = np.empty(max(n, 1), np.int8)
        lib().jo_bpsk_bits(self.h, ptr(out), n)
        return out[:n]

    def fec_results(self):
        res = []
        for i in range(lib().jo_bpsk_fec_count(self.h)):
            rc = C.c_int32()
            bi = C.c_int64()
            out = np.empty(256, np.uint8)
            lib().jo_bpsk_fec_get(self.h, i, C.addressof(rc), C.addressof(bi), ptr(out))
            res.append((rc.value, bi.value, out))
        return res

    def decoded(self):
        out = np.empty(256, np.uint8)
        lib().jo_bpsk_decoded(self.h, ptr(out))
        return out

    def trace(self):
        n = lib().jo_bpsk_trace(self.h, None, 0)
        out = np.empty((max(n, 1), 2), np.float64)
        lib().jo_bpsk_trace(self.h, ptr(out), n)
        return out[:n]

    def fft_probe_enable(self, cap_frames):
        lib().jo_bpsk_fft_probe_enable.argtypes = [C.c_void_p, C.c_int64]
        lib().jo_bpsk_fft_probe_enable(self.h, cap_frames)

    def fft_probe(self):
        """per frame: binPos, maxBin, runner-up, threshold, rule taken, centreBin after, ||x||_2, n"""
        f = lib().jo_bpsk_fft_probe
        f.restype = C.c_int64
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        n = f(self.h, None, 0)
        out = np.empty((max(n, 1), 8), np.float64)
        f(self.h, out.ctypes.data, n)
        return out[:n]

    def decision_margins(self):
        """(fft_probe_enable first) smallest margins of di<0, energy2>100 and the 8-way argmax in units of the error two correct
        double FFTs allow, their counts, and hashes of the threshold outcomes / the dmNewPeak sequence"""
        out = np.zeros(8, np.float64)
        lib().jo_bpsk_decision_margins.argtypes = [C.c_void_p, C.c_void_p]
        lib().jo_bpsk_decision_margins.restype = None
        lib().jo_bpsk_decision_margins(self.h, out.ctypes.data)
        return dict(di=out[0], energy2=out[1], argmax=out[2], n_di=int(out[3]), n_energy2=int(out[4]), n_argmax=int(out[5]),
                    hash_energy2=int(out[6]), hash_peak=int(out[7]))

    def set_sincos(self, sin_tab, cos_tab):
        """instrument: other sin / cos tables (round 6: entries perturbed by an ulp)"""
        st = np.ascontiguousarray(sin_tab, np.float64)
        ct = np.ascontiguousarray(cos_tab, np.float64)
        assert st.size == 256 and ct.size == 256
        lib().jo_bpsk_set_sincos.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib().jo_bpsk_set_sincos.restype = None
        lib().jo_bpsk_set_sincos(self.h, ptr(st), ptr(ct))

    def declog_enable(self, cap):
        lib().jo_bpsk_declog_enable.argtypes = [C.c_void_p, C.c_int64]
        lib().jo_bpsk_declog_enable.restype = None
        lib().jo_bpsk_declog_enable(self.h, int(cap))

    def declog(self, which):
        """which 0: [n, 2] (di, energy2) per detector instant; 1: [n, 2] (argmax gap, dmNewPeak) per bit clock"""
        lib().jo_bpsk_declog.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
        lib().jo_bpsk_declog.restype = C.c_int64
        n = lib().jo_bpsk_declog(self.h, which, None, 0)
        out = np.empty((max(n, 1), 2), np.float64)
        lib().jo_bpsk_declog(self.h, which, ptr(out), n)
        return out[:n]

    def fft_perturb(self, scale, seed):
        lib().jo_bpsk_fft_perturb.argtypes = [C.c_void_p, C.c_double, C.c_uint64]
        lib().jo_bpsk_fft_perturb(self.h, float(scale), int(seed))

    def trace_ds(self):
        n = lib().jo_bpsk_trace_ds(self.h, None, 0)
        out = np.empty((max(n, 1), 2), np.float64)
        lib().jo_bpsk_trace_ds(self.h, ptr(out), n)
        return out[:n]

    def state(self):
        out = np.empty(18, np.float64)
        lib().jo_bpsk_state(self.h, ptr(out))
        return out

    def istate(self):
        out = np.empty(6, np.int32)
        lib().jo_bpsk_istate(self.h, ptr(out))
        return out


# ---------------------------------------------------------------- synthetic inputs
def mix64(z):
    return int(lib().jo_mix64(C.c_uint64(z & 0xFFFFFFFFFFFFFFFF)))


def synth_payload(seed, stream, frame):
    out = np.empty(256, np.uint8)
    lib().jo_synth_payload(C.c_uint64(seed), stream, frame, ptr(out))
    return out


def synth_diffsign(sym, start=1):
    sym = np.ascontiguousarray(sym, np.uint8)
    out = np.empty(sym.size, np.int8)
    lib().jo_synth_diffsign(ptr(sym), sym.size, ptr(out), start)
    return out


def synth_tables(amp):
    c = np.empty(1024, np.int16)
    s = np.empty(1024, np.int16)
    lib().jo_synth_tables(amp, ptr(c), ptr(s))
    return c, s


def phase_inc_u32(freq_hz, rate):
    return int(round(freq_hz / rate * 2 ** 32)) & 0xFFFFFFFF


def synth_dbpsk(n0, n, dsign, sps, phase0, phase_inc, cos_tab, sin_tab, noise_gain, noise_key):
    dsign = np.ascontiguousarray(dsign, np.int8)
    out = np.empty(2 * n, np.int16)
    lib().jo_synth_dbpsk(ptr(out), n0, n, ptr(dsign), dsign.size, sps, C.c_uint32(phase0), C.c_uint32(phase_inc),
                         ptr(cos_tab), ptr(sin_tab), noise_gain, C.c_uint64(noise_key))
    return out


def synth_tones(frame0, nframes, n, cos_tab, noise_gain, key):
    out = np.empty(nframes * 2 * n, np.int16)
    lib().jo_synth_tones(ptr(out), frame0, nframes, n, ptr(cos_tab), noise_gain, C.c_uint64(key))
    return out


def make_dbpsk_stream(seed, stream, nsamples, rate=96000, carrier_hz=13200.0, amp=3000, noise_sigma=1500.0,
                      nframes=None, flips=None):
    """FUNcube-style DBPSK stream carrying FEC frames (payloads = synth_payload(seed, stream, f)).

    Returns (int16 IQ interleaved [2*nsamples], payloads[nframes,256], symbols[nframes*5200]).
    """
    sps = rate // 1200
    if nframes is None:
        nframes = -(-nsamples // (5200 * sps)) + 1
    pay = np.stack([synth_payload(seed, stream, f) for f in range(nframes)])
    sym = np.concatenate([fec_encode(pay[f]) for f in range(nframes)])
    if flips:
        sym = sym.copy()
        for i in flips:
            sym[i] ^= 1
    dsign = synth_diffsign(sym, 1)
    ct, st = synth_tables(amp)
    gain = int(round(noise_sigma / 37837.0 * 32768.0))
    key = mix64((seed * 0x9E3779B1 + stream) ^ 0xA5A5A5A5)
    iq = synth_dbpsk(0, nsamples, dsign, sps, 0, phase_inc_u32(carrier_hz, rate), ct, st, gain, key)
    return iq, pay, sym
