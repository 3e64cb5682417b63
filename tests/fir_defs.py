"""The FIR + decimate operator's definition in Python floats (IEEE doubles, no FMA = Java `double`), shared by the CPU
test that pins oracle/o_fir_phase.c:jo_fir_decimate and by the GPU parity tests (FUNcubeBPSKDemod.java:466-492)."""
import numpy as np

import oracle_lib as O

HOWARD = 0.9 * 32768.0  # HOWARD_FUDGE_FACTOR, FUNcubeBPSKDemod.java:469


def py_fir(iq, taps, decim, scale):
    """newest sample first; samples before the batch are zero (a cleared delay line)"""
    x = O.convert_i16(iq).astype(np.float64).reshape(-1, 2)
    n = x.shape[0]
    out = []
    for j in range(n // decim):
        newest = decim * (j + 1) - 1
        fi = fq = 0.0
        for a, t in enumerate(taps):
            k = newest - a
            if k >= 0:
                fi += float(x[k, 0]) * float(t)
                fq += float(x[k, 1]) * float(t)
            else:
                fi += 0.0 * float(t)
                fq += 0.0 * float(t)
        out.append((fi * scale, fq * scale))
    return np.array(out, np.float64).reshape(-1, 2)
