"""CPU: the oracle's demod.java chain (demod.java:341-483) against an independent numpy float32 restatement, plus
its corner semantics (filterMove's range check, AGC on silence, NaN handling of Math.max)."""
import numpy as np

import oracle_lib as O

f32 = np.float32


def java_short_of_float(v):
    v = f32(v)
    if np.isnan(v):
        i = 0
    elif v >= f32(2147483648.0):
        i = 2147483647
    elif v <= f32(-2147483648.0):
        i = -2147483648
    else:
        i = int(np.trunc(v))
    return ((i + 32768) & 0xFFFF) - 32768


class NumpyDemod:
    """written from the Java text, one float32 operation at a time"""

    def __init__(self, rate):
        self.rate = rate
        self.mode = self.dofir = self.dodwn = self.doagc = 0
        self.fir = np.zeros(42, f32)
        self.wfir = np.zeros(21, f32)
        self.fof = 0
        self.car = f32(0)
        self.phi = f32(0)
        self.li = f32(0)
        self.lq = f32(0)

    def weights(self, flo, fhi):
        rate = f32(self.rate)
        nlo = f32(flo) / rate
        nhi = f32(fhi) / rate
        for n in range(21):
            if n == 10:
                self.wfir[n] = f32(2) * (nhi - nlo)
            else:
                k = float(n - 10)
                self.wfir[n] = f32(np.sin(2 * np.pi * float(nhi) * k) / (np.pi * k) - np.sin(2 * np.pi * float(nlo) * k) / (np.pi * k))
            self.wfir[n] = self.wfir[n] * f32(0.54 - 0.46 * np.cos(2 * np.pi * n / 20.0))
        self.phi = f32(2 * np.pi * float(nlo))
        self.car = f32(0)
        self.fir[:] = 0
        self.fof = 40

    def receive(self, buf):
        n2 = buf.size
        sam = np.zeros(n2, f32)
        mx = f32(0)
        avg = f32(0)
        fmgain = f32(self.rate) / (f32(5000) if self.mode == 3 else f32(75000))
        for s in range(0, n2, 2):
            i, q = f32(buf[s]), f32(buf[s + 1])
            if self.dofir:
                self.fir[self.fof] = i
                self.fir[self.fof + 1] = q
                oi = f32(0)
                oq = f32(0)
                for t in range(0, 42, 2):
                    ti = (self.fof + t) % 42
                    oi = oi + self.fir[ti] * self.wfir[t // 2]
                    oq = oq + self.fir[ti + 1] * self.wfir[t // 2]
                i, q = oi, oq
                self.fof -= 2
                if self.fof < 0:
                    self.fof = 40
            if self.dodwn:
                ci = f32(np.cos(float(self.car)))
                cq = f32(np.sin(float(self.car)))
                self.car = self.car - self.phi
                if self.car < 0:
                    self.car = self.car + f32(2 * np.pi)
                i, q = i * ci - q * cq, i * cq + q * ci
            if self.mode == 0:
                i = f32(0)
            elif self.mode == 2:
                i = f32(np.sqrt(float(i * i + q * q)))
                avg = (f32(s // 2) * avg + i) / f32(s // 2 + 1)
            elif self.mode in (3, 4):
                v = ((self.li * q) - (self.lq * i)) * fmgain
                self.li, self.lq = i, q
                i = v
            sam[s] = i
            a = np.abs(i)
            mx = f32(np.nan) if (np.isnan(mx) or np.isnan(a)) else max(mx, a)
        if self.mode == 2:
            mx = mx - avg
        out = np.zeros(n2, np.int16)
        with np.errstate(all="ignore"):
            for s in range(0, n2, 2):
                v = (sam[s] - avg if self.mode == 2 else sam[s]) * (f32(1) / mx if self.doagc else f32(1))
                out[s] = out[s + 1] = java_short_of_float(v * f32(32767))
        return out, mx, avg


def test_demod_oracle_matches_numpy_restatement():
    rng = np.random.default_rng(4)
    n = 256
    for mode in (0, 1, 2, 3, 4):
        for dofir, dodwn, doagc in ((0, 0, 0), (1, 0, 1), (1, 1, 1), (0, 1, 0)):
            o = O.Demod(96000)
            p = NumpyDemod(96000)
            o.configure(mode, dofir, dodwn, doagc)
            p.mode, p.dofir, p.dodwn, p.doagc = mode, dofir, dodwn, doagc
            w, phi = o.weights(3000, 9000)
            p.weights(3000, 9000)
            assert np.array_equal(w, p.wfir) and phi == p.phi
            for k in range(3):
                t = np.arange(k * n, (k + 1) * n)
                buf = np.empty(2 * n, f32)
                buf[0::2] = (0.4 * np.cos(2 * np.pi * 6000 * t / 96000 + 2.0 * np.sin(2 * np.pi * 400 * t / 96000))).astype(f32)
                buf[1::2] = (0.4 * np.sin(2 * np.pi * 6000 * t / 96000 + 2.0 * np.sin(2 * np.pi * 400 * t / 96000))).astype(f32)
                buf += (rng.standard_normal(2 * n) * 0.01).astype(f32)
                got = o.receive(buf)
                want, mx, avg = p.receive(buf)
                assert np.array_equal(got, want), (mode, dofir, dodwn, doagc, k)
                assert (o.max == mx or (np.isnan(o.max) and np.isnan(mx))) and o.avg == avg
                assert o.car == p.car


def test_demod_filter_move_range_check_and_defaults():
    o = O.Demod(96000)
    # demod.java:300-312 with the default filter points (Integer.MIN_VALUE / MAX_VALUE, :85-86): the range check
    # fails, weights() never runs and the weights stay all zero -- with the filter enabled the output is silence
    assert not o.filter_move(0, 0)
    o.configure(1, 1, 0, 0)
    buf = np.full(64, 0.5, f32)
    assert np.all(o.receive(buf) == 0)
    o2 = O.Demod(96000)
    o2.d.flo, o2.d.fhi = 1000, 5000
    assert o2.filter_move(500, 500) and (o2.d.flo, o2.d.fhi) == (1500, 5500)
    assert not o2.filter_move(0, 60000) and (o2.d.flo, o2.d.fhi) == (1500, 5500)  # hi >= rate/2
    assert not o2.filter_move(5000, 0)  # lo >= hi


def test_demod_agc_on_silence_and_nan():
    o = O.Demod(48000)
    o.configure(1, 0, 0, 1)  # RAW + AGC
    out = o.receive(np.zeros(32, f32))
    assert np.all(out == 0)  # 0 * (1/0 = inf) = NaN -> (short)NaN = 0
    buf = np.zeros(32, f32)
    buf[6] = np.nan
    buf[8] = 0.25
    out = o.receive(buf)
    assert np.isnan(o.max) and np.all(out == 0)  # Math.max propagates NaN; x * (1/NaN) = NaN -> 0
    o.configure(1, 0, 0, 0)
    out = o.receive(np.array([2.0, 0, -2.0, 0, 0.5, 0, 70000.0, 0], f32))
    # (short)(int)(2.0*32767) = 65534 -> -2 ; -65534 -> 2 ; 0.5 -> 16383 ; 70000*32767 saturates to INT_MAX -> (short) = -1
    assert list(out[0::2]) == [-2, 2, 16383, -1]
