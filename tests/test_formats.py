"""CPU: the oracle's waterfall row (waterfall.java:87-109) against an independent numpy restatement and hand
values.  (The recording loader needs a device buffer: tests/test_gpu_formats.py.)"""
import numpy as np

import oracle_lib as O


def java_f2i(v):
    v = np.float32(v)
    if np.isnan(v):
        return 0
    if v >= np.float32(2147483648.0):
        return 2147483647
    if v <= np.float32(-2147483648.0):
        return -2147483648
    return int(np.trunc(v))


def paint_line_numpy(psd, n, width, peak=(0, 255, 255)):
    """written from the Java text, float32 arithmetic step by step"""
    step = np.float32(n) / np.float32(width)
    off = width // 2
    out = np.zeros(width, np.uint32)
    for p in range(width):
        o = java_f2i(np.float32(p) * step)
        l = java_f2i(step)
        r = psd[o]
        for i in range(o + 1, o + l):
            if psd[i] > r:
                r = psd[i]
        f = 255 - java_f2i(np.float32(r) * np.float32(-2.55))
        f = min(max(f, 0), 255)
        c = [peak[0] * f // 256, peak[1] * f // 256, peak[2] * f // 256]
        out[(p + off) % width] = 0xFF000000 | (c[0] << 16) | (c[1] << 8) | c[2]
    return out


def test_waterfall_line_matches_numpy_restatement():
    rng = np.random.default_rng(11)
    for n, width in ((2048, 1024), (2048, 800), (2048, 2048), (2048, 3000), (9600, 1280), (64, 7)):
        psd = (rng.standard_normal(n + 2) * 30 - 60).astype(np.float32)
        psd[rng.integers(0, n, 5)] = -np.inf  # empty bins (log10 of 0, fft.java:207)
        psd[rng.integers(0, n, 3)] = np.nan
        psd[rng.integers(0, n, 3)] = 12.5     # above full scale: clamps at 255
        assert np.array_equal(O.waterfall_line(psd, n, width), paint_line_numpy(psd, n, width)), (n, width)


def test_waterfall_line_known_values():
    n, width = 8, 4
    psd = np.array([-100, -50, 0, -200, -10.1, 3, np.nan, -20, 0, 0], np.float32)
    pix = O.waterfall_line(psd, n, width, peak_rgb=0x00FFFF)
    # pixel p shows max(psd[2p], psd[2p+1]); f = 255-(int)(m*-2.55) clamped; cyan*f/256; column (p+2)%4
    #   p0: max(-100,-50) = -50 -> 255-127 = 128 -> 255*128/256 = 127
    #   p1: max(0,-200) = 0 -> 255 -> 254;  p2: max(-10.1, 3) = 3 -> 255+7 -> clamp 255 -> 254
    #   p3: a[6] = NaN stays (comparisons with NaN are false) -> (int)NaN = 0 -> 255 -> 254
    want = {2: 127, 3: 254, 0: 254, 1: 254}
    for col, v in want.items():
        assert pix[col] == (0xFF000000 | (v << 8) | v), (col, hex(pix[col]))
    # other peak colours scale per channel
    pix = O.waterfall_line(psd, n, width, peak_rgb=0x804020)
    assert pix[2] == (0xFF000000 | ((0x80 * 128 // 256) << 16) | ((0x40 * 128 // 256) << 8) | (0x20 * 128 // 256))
