import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import java_sdr_amd as J
import oracle_lib as O
for blen, rate in ((4096, 96000), (16384, 192000), (8192, 96000), (8192, 192000), (16384, 96000)):
    nsf = blen // 4
    n = nsf * 24
    iq = O.make_dbpsk_stream(43, 0, n, rate=rate, carrier_hz=13200.0, noise_sigma=700.0)[0]
    d = J.Bpsk(rate=rate, blen=blen, do_fft=1, nstreams=1, max_batch_samples=n)
    d.batch_i16(J.DeviceBuffer.from_host(iq), 2 * n, n)
    o = O.Bpsk(rate=rate, blen=blen, do_fft=1, trace=n)
    o.receive_i16(iq)
    tg, to = d.trace(), o.trace()
    diff = np.nonzero((tg != to).any(axis=1))[0]
    print(blen, rate, "trace len", len(tg), len(to), "ndiff", len(diff), "first", diff[:5], "centreBin", d.counters()["centreBin"], o.counters()["centreBin"])
    if len(diff):
        i = diff[0]
        print("   gpu", tg[i], "ora", to[i], "rel", np.abs(tg[i] - to[i]) / (np.abs(to[i]) + 1e-300))
