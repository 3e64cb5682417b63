"""Host-side helpers of the package that need no GPU."""


def test_synthetic_input_helpers_match_the_oracle_generators():
    """bench.py builds its carrier tables / noise keys with the package's own helpers (it may touch oracle/ only in
    its cpu_baseline leg): they must be the tables the oracle's generator uses, or validation would compare apples
    with pears"""
    import numpy as np
    import oracle_lib as O
    import java_sdr_amd as J
    for amp in (1, 300, 3000, 8000, 32767):
        c, s = J.binding.synth_carrier_tables(amp)
        oc, os_ = O.synth_tables(amp)
        assert np.array_equal(c, oc) and np.array_equal(s, os_)
    for z in (0, 1, 42, 20020107, 2 ** 63 + 12345, 2 ** 64 - 1):
        assert J.binding.synth_mix64(z) == O.mix64(z)
    assert J.binding.synth_phase_inc_u32(13200.0, 96000) == O.phase_inc_u32(13200.0, 96000)
