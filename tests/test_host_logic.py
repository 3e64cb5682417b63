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


def test_bench_refuses_experiment_knobs_and_reports_missing_gpu():
    """bench.py must not measure a product that skips work (JSDR_EXPERIMENT_*), and must fail loudly without a device"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, JSDR_EXPERIMENT_SKIP_FEC="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True)
    assert r.returncode != 0 and "JSDR_EXPERIMENT_SKIP_FEC" in (r.stderr + r.stdout)
    import java_sdr_amd as J
    if not J.have_gpu():
        env.pop("JSDR_EXPERIMENT_SKIP_FEC")
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0"], env=env,
                           capture_output=True, text=True)
        assert r.returncode != 0 and "no HIP device" in (r.stderr + r.stdout)


def test_strong_scaling_shards_are_contiguous_and_cover_every_stream():
    from java_sdr_amd import sharding as SH
    for total, n in ((8192, 1), (8192, 2), (8192, 4), (8192, 8), (1000, 8)):
        seen = []
        for rank in range(n):
            s0, cnt = SH.shard_streams(total, n, rank)
            seen.extend(range(s0, s0 + cnt))
        assert seen == list(range(total)), (total, n)
