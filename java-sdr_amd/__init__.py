"""java-sdr_amd -- MI355X (gfx950) implementation of java-sdr's FFT / FIR / BPSK-demod hot path.

The product is the C-ABI library libjsdr_hip.so (include/jsdr_hip.h, sources in csrc/).  This
package is the thin ctypes binding the tests and bench.py drive it through, plus (host/) the C++
mirror of the reference's plugin interface.  There is no CPU fallback: importing works anywhere (so
that symbols can be checked), every compute call needs a HIP device and fails loudly without one.

The directory name has a hyphen (the repo's naming rule); import it as `java_sdr_amd` through the
shim module of that name at the repo root.
"""
from . import sharding  # noqa: F401
from .binding import (  # noqa: F401
    JsdrError, lib, library_path, have_gpu, DeviceBuffer, Fft, Fir, Bpsk, Group, Timer, Stream, Phase,
    convert_i16, fir_batch_decimate_i16, bpsk_table, phase_maxabs, phase_columns, fec_decode, fec_encode, fec_decode_batch, fec_encode_batch, fec_encode_dev, fec_decode_dev,
    synth_payloads, synth_diffsign, synth_dbpsk, synth_tones, EXPORTED_SYMBOLS,
    Demod, waterfall_lines, waterfall_lines_dev, recording_probe, recordings_load, RecordingInfo,
)
