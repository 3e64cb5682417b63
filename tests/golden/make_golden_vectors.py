#!/usr/bin/env python3
"""Generate tests/golden/oracle_vectors.npz: outputs of the CPU oracle on small, seeded inputs.

The reference holds no golden vectors and cannot be executed here (no JVM),
so these vectors pin the ORACLE (regression) and let the GPU box check the HIP path against
numbers produced in the build container.  Behavioural KATs that SURVEY.md section 8c derived
independently (sine4410.raw counters / bits / argmax bins) are asserted in tests/test_oracle_kat.py.

Run:  python tests/golden/make_golden_vectors.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402


def main():
    out = {}
    raw = np.fromfile(os.path.join(HERE, "sine4410.raw"), dtype="<i2")
    buf = O.convert_i16(raw)
    out["sine_psd96k"] = np.stack([O.fft_receive(buf[f * 4096:(f + 1) * 4096], 96000) for f in range(2)])
    out["sine_psd44k"] = np.stack([O.fft_receive(buf[f * 4096:(f + 1) * 4096], 44100) for f in range(2)])
    for mode in (0, 1):
        d = O.Bpsk(do_fft=mode, trace=1024)
        d.receive(buf[:4096])
        d.receive(buf[4096:])
        c = d.counters()
        out[f"sine_bpsk{mode}_counters"] = np.array([c[k] for k in sorted(c)], np.int32)
        out[f"sine_bpsk{mode}_bits"] = d.bits()
        out[f"sine_bpsk{mode}_trace"] = d.trace()
    # FEC known-answer: payload -> symbols, and decode of a corrupted block
    pay = O.synth_payload(20020109, 0, 0)
    sym = O.fec_encode(pay)
    out["fec_payload"] = pay
    out["fec_symbols"] = sym
    soft = np.where(sym == 1, 0xC0, 0x40).astype(np.uint8)
    rng = np.random.default_rng(7)
    pos = rng.choice(5200, 200, replace=False)
    soft[pos] ^= 0x80
    out["fec_soft_200err"] = soft
    rc, dec = O.fec_decode(soft)
    out["fec_rc_200err"] = np.array([rc], np.int32)
    out["fec_dec_200err"] = dec
    # synthetic DBPSK: 1 stream, 450k samples (one full FEC frame + change), tune mode
    iq, payloads, _ = O.make_dbpsk_stream(20020109, 3, 458752)
    d = O.Bpsk()
    d.receive_i16(iq)
    out["dbpsk_iq_head"] = iq[:4096]
    out["dbpsk_iq_sha"] = np.frombuffer(__import__("hashlib").sha256(iq.tobytes()).digest(), np.uint8)
    out["dbpsk_bits"] = d.bits()
    c = d.counters()
    out["dbpsk_counters"] = np.array([c[k] for k in sorted(c)], np.int32)
    fr = d.fec_results()
    out["dbpsk_fec_rc"] = np.array([r[0] for r in fr], np.int32)
    out["dbpsk_fec_bitidx"] = np.array([r[1] for r in fr], np.int64)
    out["dbpsk_fec_data"] = np.stack([r[2] for r in fr]) if fr else np.zeros((0, 256), np.uint8)
    out["dbpsk_payloads"] = payloads
    # tones frames for the FFT path
    ct, _ = O.synth_tables(8000)
    tones = O.synth_tones(0, 4, 2048, ct, 300, O.mix64(20020107))
    out["tones_iq"] = tones
    tb = O.convert_i16(tones)
    out["tones_psd"] = np.stack([O.fft_receive(tb[f * 4096:(f + 1) * 4096], 96000) for f in range(4)])
    # fir.java
    f = O.Fir()
    out["fir_w_500_1500"] = f.weights(500, 1500, 44100.0)
    rng = np.random.default_rng(11)
    xs = rng.integers(-4096, 4096, 512).astype(np.int32)
    out["fir_in"] = xs
    out["fir_out"] = f.filter_block(xs)
    out["fir_cgen_1000"] = O.fir_complex_gen(1000, 256)
    # phase.java
    out["phase_maxabs"] = np.array([O.phase_maxabs(buf[:4096])], np.float32)
    pix, ai, aq = O.phase_columns(buf[:4096], 300)
    out["phase_pix"] = pix
    out["phase_avgi"] = ai
    out["phase_avgq"] = aq
    path = os.path.join(HERE, "oracle_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
