#!/usr/bin/env python3
"""Independent second restatement, in pure Python, of the part of the reference the C oracle is trusted for.

TEST INFRASTRUCTURE, generation time only.  Written from the Java text (not from oracle/*.c) so that the two
restatements can be held against each other:

  * FUNcubeBPSKDemod.java:366-397,466-595  tune-mode receive chain (RxMixTuner, RxDownSample, RxDemodulate
                                            incl. the 9600 Hz tail :533-594), constants :83-96, tables :159-162
  * JavaAudio.java:276-293                  int16 -> float conversion with DC correction
  * FECDecoder.java:203-278                 viterbi27
  * FECDecoder.java:317-519                 mod255, decode_rs_8
  * FECDecoder.java:538-688                 encode_FEC40 (re-encoder; also builds test frames)
  * FECDecoder.java:703-852                 FECDecode

Every constant table (mettab, Partab, Syms, Scrambler, ALPHA_TO, INDEX_OF, RS_poly, dsFilter, dmFilter,
SYNC_VECTOR) is PARSED FROM THE REFERENCE'S SOURCE TEXT as data when this module is imported, so it only
works where /root/reference exists (the build container).  make_reference_fixtures.py runs it there and commits
inputs' digests + outputs as fixtures; tests compare the C oracle (CPU) and the HIP path (-m gpu) with those.

Python floats are IEEE-754 doubles evaluated without FMA contraction, i.e. Java `double` semantics; float32
steps go through numpy.float32.  sin/cos tables: Java's Math.sin/cos are specified to 1 ulp; here the table is
the correctly rounded value of sin/cos of the double argument, computed in exact rational arithmetic (no libm).
"""
import math
import os
import re
from fractions import Fraction

import numpy as np

REF = os.environ.get("JSDR_REFERENCE", "/root/reference")


# ------------------------------------------------------------------------------------- tables, parsed as data
def _strip_comments(src):
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    return re.sub(r"//[^\n]*", " ", src)


def _array_body(src, name):
    m = re.search(r"\b" + re.escape(name) + r"\b[^=;]*=\s*\{", src)
    if not m:
        raise KeyError(name)
    i = j = m.end()
    depth = 1
    while depth:
        depth += {"{": 1, "}": -1}.get(src[j], 0)
        j += 1
    return src[i:j - 1]


def _ints(body):
    return [int(t, 0) for t in re.findall(r"[-+]?(?:0x[0-9a-fA-F]+|\d+)", body)]


def _floats(body):
    # `double[] x = { 1.5F, ... }` holds float32 values widened to double
    return [float(np.float32(t[:-1])) for t in re.findall(r"[-+]?\d+\.\d+(?:[eE][-+]?\d+)?[fF]", body)]


_fec_src = _strip_comments(open(os.path.join(REF, "FECDecoder.java")).read())
_dem_src = _strip_comments(open(os.path.join(REF, "FUNcubeBPSKDemod.java")).read())

Partab = _ints(_array_body(_fec_src, "Partab"))
_met = _ints(_array_body(_fec_src, "mettab"))
mettab = [_met[:256], _met[256:]]
Syms = _ints(_array_body(_fec_src, "Syms"))
Scrambler = _ints(_array_body(_fec_src, "Scrambler"))
ALPHA_TO = _ints(_array_body(_fec_src, "ALPHA_TO"))
INDEX_OF = _ints(_array_body(_fec_src, "INDEX_OF"))
RS_poly = _ints(_array_body(_fec_src, "RS_poly"))
dsFilter = _floats(_array_body(_dem_src, "dsFilter"))
dmFilter = _floats(_array_body(_dem_src, "dmFilter"))
SYNC_VECTOR = _ints(_array_body(_dem_src, "SYNC_VECTOR"))
assert (len(Partab), len(_met), len(Syms), len(Scrambler), len(ALPHA_TO), len(INDEX_OF), len(RS_poly)) == \
    (256, 512, 128, 320, 256, 256, 16)
assert (len(dsFilter), len(dmFilter), len(SYNC_VECTOR)) == (27, 130, 65)

# FECDecoder.java:12-36
NN, KK, NROOTS, FCR, PRIM, IPRIM = 255, 223, 32, 112, 11, 116
A0 = NN
BLOCKSIZE, RSBLOCKS, RSPAD = 256, 2, 95
K, CPOLYA, CPOLYB = 7, 0x4F, 0x6D
NBITS = (BLOCKSIZE + NROOTS * RSBLOCKS) * 8 + K - 1
ROWS, COLUMNS = 80, 65
SYMPBLOCK = ROWS * COLUMNS
SYNC_POLY = 0x48


def _sbyte(v):
    """Java (byte) cast"""
    v &= 0xFF
    return v - 256 if v >= 128 else v


# ------------------------------------------------------------------------------------- FECDecoder.java
def viterbi27(data, symbols, nbits):
    """:203-278.  data: list of ints (bytes), filled in place."""
    bitcnt = 0
    k = 0
    l = 0
    cmetric = [0] + [-999999] * 63
    nmetric = [0] * 64
    pp = [0] * (nbits * 2)
    mets = [0] * 4
    while True:
        for i in range(4):
            mets[i] = 0
            for j in range(2):
                mets[i] += mettab[(i >> (1 - j)) & 1][symbols[j + k] & 0xFF]
        k += 2
        mask = 1
        for i in range(0, 64, 2):
            b1 = mets[Syms[i]]
            m0 = cmetric[i // 2] + b1
            nmetric[i] = m0
            b2 = mets[Syms[i + 1]]
            b1 -= b2
            m1 = cmetric[(i // 2) + (1 << (K - 2))] + b2
            if m1 > m0:
                nmetric[i] = m1
                pp[l] |= mask
            m0 -= b1
            nmetric[i + 1] = m0
            m1 += b1
            if m1 > m0:
                nmetric[i + 1] = m1
                pp[l] |= mask << 1
            mask <<= 2
            if (mask & 0xFFFFFFFF) == 0:
                mask = 1
                l += 1
        if mask != 1:
            l += 1
        bitcnt += 1
        if bitcnt == nbits:
            beststate = 0
            break
        cmetric = list(nmetric)
    l -= 2
    for i in range(nbits // 8):
        data[i] = 0
    for i in range(nbits - K, -1, -1):
        if pp[l + (beststate >> 5)] & (1 << (beststate & 31)):
            beststate |= 1 << (K - 1)
            data[i >> 3] |= 0x80 >> (i & 7)
        beststate >>= 1
        l -= 2
    return 0


def mod255(x):
    """:317-323"""
    while x >= 255:
        x -= 255
        x = (x >> 8) + (x & 255)
    return x


def decode_rs_8(data):
    """:325-519 with eras_pos == null, no_eras == 0 (the only call, :772).  data: 255 ints 0..255, in place."""
    lam = [0] * (NROOTS + 1)
    s = [0] * NROOTS
    b = [0] * (NROOTS + 1)
    t = [0] * (NROOTS + 1)
    omega = [0] * (NROOTS + 1)
    root = [0] * NROOTS
    reg = [0] * (NROOTS + 1)
    loc = [0] * NROOTS
    for i in range(NROOTS):
        s[i] = _sbyte(data[0])  # `s[i] = data[0]` without & 0xff (:337); column 0 is RS padding (zero)
    for j in range(1, NN):
        for i in range(NROOTS):
            if s[i] == 0:
                s[i] = data[j] & 0xFF
            else:
                s[i] = (data[j] & 0xFF) ^ ALPHA_TO[mod255(INDEX_OF[s[i]] + (FCR + i) * PRIM)]
    syn_error = 0
    for i in range(NROOTS):
        syn_error |= s[i]
        s[i] = INDEX_OF[s[i]]
    if syn_error == 0:
        return 0
    lam[0] = 1
    for i in range(NROOTS + 1):
        b[i] = INDEX_OF[lam[i]]
    r = 0
    el = 0
    while True:
        r += 1
        if r > NROOTS:
            break
        discr_r = 0
        for i in range(r):
            if lam[i] != 0 and s[r - i - 1] != A0:
                discr_r ^= ALPHA_TO[mod255(INDEX_OF[lam[i]] + s[r - i - 1])]
        discr_r = INDEX_OF[discr_r]
        if discr_r == A0:
            b[1:NROOTS + 1] = b[0:NROOTS]
            b[0] = A0
        else:
            t[0] = lam[0]
            for i in range(NROOTS):
                if b[i] != A0:
                    t[i + 1] = lam[i + 1] ^ ALPHA_TO[mod255(discr_r + b[i])]
                else:
                    t[i + 1] = lam[i + 1]
            if 2 * el <= r - 1:
                el = r - el
                for i in range(NROOTS + 1):
                    b[i] = A0 if lam[i] == 0 else mod255(INDEX_OF[lam[i]] - discr_r + NN)
            else:
                b[1:NROOTS + 1] = b[0:NROOTS]
                b[0] = A0
            lam = list(t)
    deg_lambda = 0
    for i in range(NROOTS + 1):
        lam[i] = INDEX_OF[lam[i]]
        if lam[i] != A0:
            deg_lambda = i
    reg[1:NROOTS + 1] = lam[1:NROOTS + 1]
    count = 0
    i = 1
    k = IPRIM - 1
    while i <= NN:
        q = 1
        for j in range(deg_lambda, 0, -1):
            if reg[j] != A0:
                reg[j] = mod255(reg[j] + j)
                q ^= ALPHA_TO[reg[j]]
        if q == 0:
            root[count] = i
            loc[count] = k
            count += 1
            if count == deg_lambda:
                break
        i += 1
        k = mod255(k + IPRIM)
    if deg_lambda != count:
        return -1
    deg_omega = 0
    for i in range(NROOTS):
        tmp = 0
        j = deg_lambda if deg_lambda < i else i
        while j >= 0:
            if s[i - j] != A0 and lam[j] != A0:
                tmp ^= ALPHA_TO[mod255(s[i - j] + lam[j])]
            j -= 1
        if tmp != 0:
            deg_omega = i
        omega[i] = INDEX_OF[tmp]
    omega[NROOTS] = A0
    for j in range(count - 1, -1, -1):
        num1 = 0
        for i in range(deg_omega, -1, -1):
            if omega[i] != A0:
                num1 ^= ALPHA_TO[mod255(omega[i] + i * root[j])]
        num2 = ALPHA_TO[mod255(root[j] * (FCR - 1) + NN)]
        den = 0
        i = min(deg_lambda, NROOTS - 1) & ~1
        while i >= 0:
            if lam[i + 1] != A0:
                den ^= ALPHA_TO[mod255(lam[i + 1] + i * root[j])]
            i -= 2
        if den == 0:
            return -1
        if num1 != 0:
            data[loc[j]] ^= ALPHA_TO[mod255(INDEX_OF[num1] + INDEX_OF[num2] + NN - INDEX_OF[den])]
    return count


class Encoder:
    """:538-688"""

    def __init__(self):
        self.reencode = [0] * SYMPBLOCK

    def interleave_symbol(self, c):
        col = self.Bindex // COLUMNS
        row = self.Bindex % COLUMNS
        if c != 0:
            self.reencode[row * ROWS + col] = 1
        self.Bindex += 1

    def encode_and_interleave(self, c, cnt):
        while cnt != 0:
            cnt -= 1
            self.Conv_sr = ((self.Conv_sr << 1) | (c >> 7)) & 0xFFFFFFFF  # Java int; only the low 7 bits are read
            c <<= 1
            self.interleave_symbol(Partab[self.Conv_sr & CPOLYA])
            self.interleave_symbol(1 - Partab[self.Conv_sr & CPOLYB])

    def scramble_and_encode(self, c):
        c ^= Scrambler[self.Nbytes]
        self.encode_and_interleave(c, 8)

    def local_init_encoder(self):
        self.Nbytes = 0
        self.Conv_sr = 0
        self.Bindex = COLUMNS
        self.RS_block = [[0] * NROOTS for _ in range(RSBLOCKS)]
        for i in range(5200):
            self.reencode[i] = 0
        sr = 0x7F
        for i in range(65):
            if sr & 64:
                self.reencode[ROWS * i] = 1
            sr = (sr << 1) | Partab[sr & SYNC_POLY]

    def local_encode_byte(self, c):
        rsi = self.Nbytes & 1
        blk = self.RS_block[rsi]
        feedback = INDEX_OF[c ^ blk[0]]
        if feedback != A0:
            for j in range(15):
                t = ALPHA_TO[mod255(feedback + RS_poly[j])]
                blk[j + 1] ^= t
                blk[31 - j] ^= t
            blk[16] ^= ALPHA_TO[mod255(feedback + RS_poly[15])]
        for i in range(31):
            blk[i] = blk[i + 1]
        blk[31] = ALPHA_TO[feedback] if feedback != A0 else 0
        self.scramble_and_encode(c)
        self.Nbytes += 1

    def local_encode_parity(self):
        c = self.RS_block[self.Nbytes & 1][(self.Nbytes - 256) >> 1]
        self.scramble_and_encode(c)
        self.Nbytes += 1
        if self.Nbytes == 320:
            self.encode_and_interleave(0, 6)

    def encode_FEC40(self, RSdecdata):
        self.local_init_encoder()
        for i in range(256):
            self.local_encode_byte(RSdecdata[i] & 0xFF)
        for i in range(64):
            self.local_encode_parity()
        return self.reencode


def FECDecode(raw, RSdecdata):
    """:703-852.  raw: 5200 ints 0..255; RSdecdata: 256 ints, written only when both RS words decode."""
    symbols = [0] * (NBITS * 2 + 65 + 3)
    vitdecdata = [0] * ((NBITS - 6) // 8)
    nRC = 0
    coltop = 0
    for col in range(1, ROWS):
        rowstart = 0
        for row in range(COLUMNS):
            symbols[coltop + row] = raw[rowstart + col]
            rowstart += ROWS
        coltop += COLUMNS
    viterbi27(vitdecdata, symbols, NBITS)
    rsblocks = [[0] * NN for _ in range(RSBLOCKS)]
    di = si = 0
    for col in range(RSPAD, NN):
        for row in range(RSBLOCKS):
            rsblocks[row][col] = (vitdecdata[di] ^ Scrambler[si]) & 0xFF
            di += 1
            si += 1
    rserrs = [decode_rs_8(rsblocks[row]) for row in range(RSBLOCKS)]
    if all(e != -1 for e in rserrs):
        j = 0
        for col in range(RSPAD, KK):
            for row in range(RSBLOCKS):
                RSdecdata[j] = rsblocks[row][col]
                j += 1
    for e in rserrs:
        if e == -1:
            nRC = -1
    if nRC >= 0:
        reencode = Encoder().encode_FEC40(RSdecdata)
        errors = 0
        for i in range(SYMPBLOCK):
            if (reencode[i] & 0xFF) != ((raw[i] & 0xFF) >> 7):
                errors += 1
        nRC = errors
    return nRC


# ------------------------------------------------------------------------------------- sin / cos tables
def _round_fraction(q):
    return float(q)  # Fraction -> float is correctly rounded (round-half-even) in CPython


def _sincos_exact(x):
    """correctly rounded (sin x, cos x) of the double x, via Taylor series in exact rationals"""
    q = Fraction(x)
    s = Fraction(0)
    c = Fraction(0)
    term = Fraction(1)
    n = 0
    # |x| <= 2 pi: 80 terms leave a remainder far below 2^-200
    while n < 80:
        if n % 2 == 0:
            c += term if (n // 2) % 2 == 0 else -term
        else:
            s += term if (n // 2) % 2 == 0 else -term
        n += 1
        term = term * q / n
    return _round_fraction(s), _round_fraction(c)


def sincos_tables(size=256):
    sin_tab, cos_tab = [], []
    for n in range(size):
        arg = n * 2.0 * math.pi / size  # :160-161, evaluated left to right in double
        s, c = _sincos_exact(arg)
        sin_tab.append(s)
        cos_tab.append(c)
    return sin_tab, cos_tab


# ------------------------------------------------------------------------------------- JavaAudio.java:276-293
def convert_i16(raw, ic=0, qc=0):
    """raw: numpy int16 interleaved I,Q.  Returns a list of Python floats holding float32 values."""
    raw = np.asarray(raw, dtype=np.int16)
    s = raw.astype(np.int32)
    s[0::2] += _sbyte16(ic)
    s[1::2] += _sbyte16(qc)
    s = ((s + 32768) & 0xFFFF) - 32768  # `short s; s += (short)ic` wraps mod 2^16
    f = s.astype(np.float32) / np.float32(32767)  # (float)s/(float)Short.MAX_VALUE: one correctly rounded division
    return [float(v) for v in f]


def _sbyte16(v):
    v &= 0xFFFF
    return v - 65536 if v >= 32768 else v


# ------------------------------------------------------------------------------------- FUNcubeBPSKDemod.java
class Demod:
    """tune-mode receive chain (:366-397, :466-595); the FFT-acquire front end (:399-464) is DemodFFT below"""

    DOWN_SAMPLE_FILTER_SIZE = 27
    MATCHED_FILTER_SIZE = 65
    SYNC_VECTOR_SIZE = 65
    FEC_BITS_SIZE = 5200
    RX_CARRIER_FREQ = 1200.0
    DOWN_SAMPLE_RATE = 9600
    BIT_RATE = 1200
    SAMPLES_PER_BIT = DOWN_SAMPLE_RATE // BIT_RATE
    VCO_PHASE_INC = 2.0 * math.pi * RX_CARRIER_FREQ / float(DOWN_SAMPLE_RATE)
    BIT_SMOOTH1 = 1.0 / 200.0
    BIT_SMOOTH2 = 1.0 / 800.0
    BIT_PHASE_INC = 1.0 / float(DOWN_SAMPLE_RATE)
    BIT_TIME = 1.0 / float(BIT_RATE)
    SINCOS_SIZE = 256
    EMAX0 = float(np.float32(-1.0e10))  # `double eMax = -1.0e10F`

    def __init__(self, rate=96000, tuning=12000, trace_cap=0):
        self.rate = rate
        self.sinTab, self.cosTab = sincos_tables(self.SINCOS_SIZE)
        self.tuning = float(tuning)
        self.tuPhaseInc = 2.0 * math.pi * self.tuning / float(rate)  # :196
        self.tuPhase = 0.0
        self.dsBuf = [[0.0, 0.0] for _ in range(self.DOWN_SAMPLE_FILTER_SIZE)]
        self.dsPos = self.DOWN_SAMPLE_FILTER_SIZE - 1
        self.dsCnt = 0
        self.HOWARD_FUDGE_FACTOR = 0.9 * 32768.0
        self.vcoPhase = 0.0
        self.dmBuf = [[0.0, 0.0] for _ in range(self.MATCHED_FILTER_SIZE)]
        self.dmPos = self.MATCHED_FILTER_SIZE - 1
        self.dmEnergy = [0.0] * (self.SAMPLES_PER_BIT + 2)
        self.dmBitPos = self.dmPeakPos = self.dmNewPeak = self.dmCorr = self.dmMaxCorr = 0
        self.dmEnergyOut = 1.0
        self.dmHalfTable = [4, 5, 6, 7, 0, 1, 2, 3]
        self.dmBitPhase = 0.0
        self.dmLastIQ = [0.0, 0.0]
        self.dmFECCorr = [0] * self.FEC_BITS_SIZE
        self.decoded = [0] * 256
        self.decodeOK = False
        self.cntRaw = self.cntDS = self.cntBit = self.cntFEC = self.cntDec = self.dmErrBits = 0
        self.energy1 = self.energy2 = 0.0
        # observation only (not in the reference): sliced bits, FECDecode log, (fi,fq) trace
        self.bits = []
        self.fec_log = []
        self.trace = []
        self.trace_cap = trace_cap

    def receive(self, buf):
        """doBufferTune (:366-379): buf = interleaved I,Q floats"""
        two_pi = 2.0 * math.pi
        for n in range(len(buf) // 2):
            i = buf[2 * n]
            q = buf[2 * n + 1]
            # RxMixTuner (:382-397)
            self.tuPhase += self.tuPhaseInc
            if self.tuPhase > two_pi:
                self.tuPhase -= two_pi
            if self.tuPhase > 0.0:
                k = int(self.tuPhase * float(self.SINCOS_SIZE) / two_pi) % self.SINCOS_SIZE
                self.RxDownSample(i * self.cosTab[k], q * self.sinTab[k])
            else:
                self.RxDownSample(i, q)

    def RxDownSample(self, i, q):
        """:470-492"""
        N = self.DOWN_SAMPLE_FILTER_SIZE
        self.dsBuf[self.dsPos][0] = i
        self.dsBuf[self.dsPos][1] = q
        self.dsCnt += 1
        if self.dsCnt >= self.rate // self.DOWN_SAMPLE_RATE:
            fi = 0.0
            fq = 0.0
            for n in range(N):
                dsi = (n + self.dsPos) % N
                fi += self.dsBuf[dsi][0] * dsFilter[n]
                fq += self.dsBuf[dsi][1] * dsFilter[n]
            self.dsCnt = 0
            self.RxDemodulate(fi * self.HOWARD_FUDGE_FACTOR, fq * self.HOWARD_FUDGE_FACTOR)
        self.dsPos -= 1
        if self.dsPos < 0:
            self.dsPos = N - 1
        self.cntRaw += 1

    def RxDemodulate(self, i, q):
        """:505-595"""
        M = self.MATCHED_FILTER_SIZE
        two_pi = 2.0 * math.pi
        self.vcoPhase += self.VCO_PHASE_INC
        if self.vcoPhase > two_pi:
            self.vcoPhase -= two_pi
        k = int(self.vcoPhase * float(self.SINCOS_SIZE) / two_pi) % self.SINCOS_SIZE
        self.dmBuf[self.dmPos][0] = i * self.cosTab[k]
        self.dmBuf[self.dmPos][1] = q * self.sinTab[k]
        fi = 0.0
        fq = 0.0
        for n in range(M):
            dmi = M - self.dmPos + n
            fi += self.dmBuf[n][0] * dmFilter[dmi]
            fq += self.dmBuf[n][1] * dmFilter[dmi]
        self.dmPos -= 1
        if self.dmPos < 0:
            self.dmPos = M - 1
        if len(self.trace) < self.trace_cap:
            self.trace.append((fi, fq))
        self.energy1 = fi * fi + fq * fq
        bp = self.dmBitPos
        self.dmEnergy[bp] = (self.dmEnergy[bp] * (1.0 - self.BIT_SMOOTH1)) + (self.energy1 * self.BIT_SMOOTH1)
        if self.dmBitPos == self.dmPeakPos:
            self.dmEnergyOut = (self.dmEnergyOut * (1.0 - self.BIT_SMOOTH2)) + (self.energy1 * self.BIT_SMOOTH2)
            di = -(self.dmLastIQ[0] * fi + self.dmLastIQ[1] * fq)
            dq = self.dmLastIQ[0] * fq - self.dmLastIQ[1] * fi
            self.dmLastIQ[0] = fi
            self.dmLastIQ[1] = fq
            self.energy2 = math.sqrt(di * di + dq * dq)
            if self.energy2 > 100.0:
                bit = di < 0.0
                self.dmFECCorr = self.dmFECCorr[1:] + [1 if bit else -1]
                self.bits.append(1 if bit else -1)
                self.dmCorr = 0
                for n in range(self.SYNC_VECTOR_SIZE):
                    self.dmCorr += self.dmFECCorr[n * 80] * SYNC_VECTOR[n]
                if self.dmCorr >= 45:
                    dmFECBits = [0xC0 if v == 1 else 0x40 for v in self.dmFECCorr]
                    self.dmErrBits = FECDecode(dmFECBits, self.decoded)
                    self.cntFEC += 1
                    self.dmMaxCorr = 0
                    self.decodeOK = not (self.dmErrBits < 0)
                    self.cntDec += 1 if self.decodeOK else 0
                    self.fec_log.append((self.dmErrBits, len(self.bits), list(self.decoded)))
                if self.dmCorr > self.dmMaxCorr:
                    self.dmMaxCorr = self.dmCorr
                self.cntBit += 1
        if self.dmBitPos == self.dmHalfTable[self.dmPeakPos]:
            self.dmPeakPos = self.dmNewPeak
        self.dmBitPos = (self.dmBitPos + 1) % self.SAMPLES_PER_BIT
        self.dmBitPhase += self.BIT_PHASE_INC
        if self.dmBitPhase >= self.BIT_TIME:
            self.dmBitPhase -= self.BIT_TIME
            self.dmBitPos = 0
            eMax = self.EMAX0
            for n in range(self.SAMPLES_PER_BIT):
                if self.dmEnergy[n] > eMax:
                    self.dmNewPeak = n
                    eMax = self.dmEnergy[n]
        self.cntDS += 1

    # ---- snapshots in the layout of jo_bpsk_counters / jo_bpsk_state / jo_bpsk_istate
    def counters(self):
        return [self.cntRaw, self.cntDS, self.cntBit, self.cntFEC, self.cntDec, self.dmErrBits, self.dmCorr,
                self.dmMaxCorr, 1 if self.decodeOK else 0, 0]

    def state(self):
        return [self.tuPhase, self.vcoPhase, self.dmBitPhase, self.dmEnergyOut, self.energy1, self.energy2, 0.0, 0.0] + \
            self.dmEnergy[:8] + self.dmLastIQ

    def istate(self):
        return [self.dsPos, self.dsCnt, self.dmPos, self.dmBitPos, self.dmPeakPos, self.dmNewPeak]


# ------------------------------------------------------------------------------------- doBufferFFT (:399-464)
def fft_radix2_dit(a, n, inverse, scale, w):
    """The transform this project DEFINES in place of JTransforms' DoubleFFT_1D.complexForward / complexInverse(a, true)
    (FUNcubeBPSKDemod.java:422-423, :459; JTransforms 2.4 is a Maven dependency whose source is not in the reference
    tree, so its own rounding is unknowable): radix-2 decimation in time on interleaved (re, im) doubles --
      bit-reversal permutation, then log2(n) stages of butterflies  t = w b ; a' = a + t ; b' = a - t
      with  tr = wr br - wi bi ,  ti = wr bi + wi br  (every product and every sum rounded on its own),
      w = table entry j n/(2 half) for butterfly j of a stage of half-width `half`, conjugated for the inverse,
      the inverse scaled by the product with 1/n afterwards.
    Written from that definition (oracle/o_fft.c's header comment), not from the C code.  `w` = the twiddle table as DATA."""
    bits = n.bit_length() - 1
    assert 1 << bits == n
    for i in range(n):
        j = int(format(i, "0%db" % bits)[::-1], 2) if bits else 0
        if j > i:
            a[2 * i], a[2 * j] = a[2 * j], a[2 * i]
            a[2 * i + 1], a[2 * j + 1] = a[2 * j + 1], a[2 * i + 1]
    half = 1
    while half < n:
        step = n // (2 * half)
        for base in range(0, n, 2 * half):
            for j in range(half):
                wr = w[2 * (j * step)]
                wi = w[2 * (j * step) + 1]
                if inverse:
                    wi = -wi
                ia = base + j
                ib = ia + half
                br = a[2 * ib]
                bi = a[2 * ib + 1]
                tr = wr * br - wi * bi
                ti = wr * bi + wi * br
                ar = a[2 * ia]
                ai = a[2 * ia + 1]
                a[2 * ia] = ar + tr
                a[2 * ia + 1] = ai + ti
                a[2 * ib] = ar - tr
                a[2 * ib + 1] = ai - ti
        half *= 2
    if inverse and scale:
        norm = 1.0 / float(n)
        for i in range(2 * n):
            a[i] *= norm


def _f32(x):
    return float(np.float32(x))


class DemodFFT(Demod):
    """receive() with bpsk-dofft set: doBufferFFT (:399-464) in front of the same RxDownSample / RxDemodulate chain.
    `samples` complex samples per call (blen/4 of the reference's byte buffer), `twiddles` the table of fft_radix2_dit."""

    # :399-402 -- float expressions, widened to double when the static finals are initialised
    CFREQ_INV_AVERAGE_FACTOR = _f32(np.float32(1.0) - (np.float32(2.0) / np.float32(1 + 1)))
    CFREQ_AVERAGE_FACTOR = _f32(np.float32(2.0) / np.float32(1 + 1))
    PSD_INV_AVERAGE_FACTOR = _f32(np.float32(1.0) - (np.float32(2.0) / np.float32(10 + 1)))
    PSD_AVERAGE_FACTOR = _f32(np.float32(2.0) / np.float32(10 + 1))

    def __init__(self, samples, twiddles, rate=96000, tuning=12000, do_up=False, trace_cap=0):
        super().__init__(rate=rate, tuning=tuning, trace_cap=trace_cap)
        self.samples = samples
        self.w = twiddles
        self.doUp = do_up
        self.avePeakPower = 0.0
        self.aveCentreBin = 0.0
        self.centreBin = 0
        self.centre_log = []  # observation only: centreBin after every call

    def receive(self, buf):
        """:357-363 with doFFT: one call = one frame of `samples` complex samples"""
        assert len(buf) == 2 * self.samples
        self.doBufferFFT(buf)

    def doBufferFFT(self, buf):
        samples = self.samples
        fftFwd = [0.0] * (2 * samples)
        fftRev = [0.0] * (2 * samples)
        psd = [0.0] * samples
        avePsd = [0.0] * samples
        for n in range(samples):
            fftFwd[2 * n] = buf[2 * n]      # (double)buf[n*2]: the float's value
            fftFwd[2 * n + 1] = buf[2 * n + 1]
        fft_radix2_dit(fftFwd, samples, False, False, self.w)  # :422-423
        for i in range(samples // 2):  # :425-427
            psd[i] = math.sqrt(fftFwd[2 * i] * fftFwd[2 * i] + fftFwd[2 * i + 1] * fftFwd[2 * i + 1])
        maxBin = 0.0
        binPos = -1
        beg = samples // 4 if self.doUp else 0
        end = samples // 2 if self.doUp else samples // 4
        for i in range(beg + 75, end - 75):  # :433-443
            acc = 0.0
            for j in range(i - 50, i + 50):
                acc += psd[j]
            avePsd[i] = acc
            if maxBin < acc:
                maxBin = acc
                binPos = i
        if self.centreBin < 0:
            self.centreBin = 0
        if self.centreBin > end - 1:
            self.centreBin = end - 1
        self.avePeakPower = (self.PSD_AVERAGE_FACTOR * avePsd[self.centreBin]) + (self.PSD_INV_AVERAGE_FACTOR * self.avePeakPower)
        if maxBin > (self.avePeakPower / 4) * 5 and binPos > 0:  # :447
            self.aveCentreBin = (self.CFREQ_AVERAGE_FACTOR * float(binPos)) + (self.CFREQ_INV_AVERAGE_FACTOR * self.aveCentreBin)
            self.centreBin = int(self.aveCentreBin + 1.0)  # (int) of a positive double truncates
        if self.centreBin < 102:
            self.centreBin = 102
        self.centre_log.append(self.centreBin)
        lo = 2 * (self.centreBin - 102)
        fftRev[0:2 * 204] = fftFwd[lo:lo + 2 * 204]  # :458
        fft_radix2_dit(fftRev, samples, True, True, self.w)  # :459
        for i in range(samples):
            self.RxDownSample(fftRev[2 * i], fftRev[2 * i])  # "yes it drops Q component.."
