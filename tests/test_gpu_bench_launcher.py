"""GPU: `python bench.py --gpus 2` from the BARE command (no torch.distributed.run) -- the launcher inside bench.py starts
one fresh child per rank.  On the one-GPU box both ranks share device 0 and gloo stands in for RCCL (the rehearsal knobs
bench.py documents); what is checked is the N > 1 code path itself: shard -> demodulate -> pack -> all-gather, every
rank's segment of `gathered` equal to that rank's own slots, sampled slots equal to the handle's getters, and every
stream's payload validated."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(extra, env_extra):
    env = dict(os.environ, **env_extra)
    env.pop("JSDR_KNOBS", None)  # (conftest switches the library's test knobs on for the test process; bench.py refuses to measure with them)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True,
                       timeout=560)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    return r, lines


def test_bare_command_two_ranks_gather_is_checked():
    r, lines = run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--streams", "48", "--samples", "1048576",
                          "--no-cpu-baseline"], {"JSDR_BENCH_SAME_DEVICE": "1", "JSDR_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, (r.stdout[-2000:], r.stderr[-2000:])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["total_streams"] == 96 and out["scaling"] == "weak"
    assert out["validated"] is True, out
    g = out["gather_check"]
    assert g["every_rank_segment_equals_that_ranks_slots"] and g["sampled_slots_equal_getters"] and g["all_ranks_ok"], g
    assert out["roofline"]["per_kernel"]["k_fm"]["bound"] == "fp64-issue"
    assert 0.0 < out["roofline"]["per_kernel"]["k_fm"]["fp64_issue_frac"] < 1.0
    assert g["validated_per_rank"] == [True, True] and len(out["ms_per_step_per_rank"]) == 2


def test_bare_command_four_ranks_strong_scaling_shape():
    """the shape of the driver's N > 2 lines: --total-streams fixed, S/N per rank, contiguous shards, one gather a step.
    Four ranks, not eight: the GPU box allows six processes on the card at once (this test process is one of them)."""
    r, lines = run_bench(["--gpus", "4", "--steps", "2", "--warmup", "1", "--total-streams", "32", "--samples", "524288",
                          "--no-cpu-baseline"], {"JSDR_BENCH_SAME_DEVICE": "1", "JSDR_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, (r.stdout[-2000:], r.stderr[-2000:])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["config"]["total_streams"] == 32 and out["config"]["streams_per_gpu"] == 8
    assert out["scaling"] == "strong" and out["validated"] is True, out
    g = out["gather_check"]
    assert g["all_ranks_ok"] and g["validated_per_rank"] == [True] * 4 and len(out["ms_per_step_per_rank"]) == 4, g


def test_bare_command_ends_a_run_whose_rank_hangs():
    """a rank that never finishes (here: asked to sleep) must not hang the launcher: after JSDR_BENCH_LAUNCH_TIMEOUT the
    survivors are terminated, then killed, and the command exits non-zero"""
    import time
    t0 = time.time()
    r, lines = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--streams", "8", "--samples", "262144", "--no-cpu-baseline"],
                         {"JSDR_BENCH_SAME_DEVICE": "1", "JSDR_BENCH_BACKEND": "gloo", "JSDR_BENCH_LAUNCH_TIMEOUT": "20",
                          "JSDR_BENCH_TEST_HANG_RANK": "1"})
    assert r.returncode != 0 and time.time() - t0 < 120, (r.returncode, time.time() - t0)


def test_bare_command_fails_when_a_rank_fails():
    # 7 streams do not split over 2 ranks: every rank exits with an error, and so must the launcher
    r, lines = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--total-streams", "7", "--no-cpu-baseline"],
                         {"JSDR_BENCH_SAME_DEVICE": "1", "JSDR_BENCH_BACKEND": "gloo"})
    assert r.returncode != 0 and not lines


def test_distributed_path_with_the_real_backend_and_one_rank():
    """`--gpus 1 --dist-single`: torch.distributed with the RCCL backend and ONE rank -- process-group init on the device, the
    gather stream, all_gather_into_tensor, the gather check -- the closest a one-GPU box gets to the driver's N > 1 run.  Rank 0's
    stdout must hold the JSON line and nothing else (RCCL prints a banner on stdout when its first communicator comes up)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("JSDR_KNOBS", None)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--dist-single", "--streams", "256", "--steps", "3",
                        "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j["validated"] is True and j["n_gpus"] == 1
    assert j["gather_check"]["all_ranks_ok"] is True and j["gather_check"]["validated_per_rank"] == [True]
    assert "RCCL" in j["config"]["parallelism"]


def test_roofline_traffic_is_counted_in_the_run_itself():
    """`python bench.py` at N = 1 counts the HBM bytes it reports: the same command as a child process under rocprofv3 --pmc
    FETCH_SIZE and again under --pmc WRITE_SIZE before the timed run ('roofline.traffic is a stored constant').
    A small pipeline batch: the source says so, the per-kernel table holds the step's kernels, and the PSD kernel's bytes are its
    algorithmic ones (4 B in + 4 B out per sample) within what the counters' granularity allows."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("JSDR_KNOBS", "JSDR_BENCH_LIVE_TRAFFIC", "JSDR_BENCH_ALLOW_KNOBS"):
        env.pop(k, None)
    S, L = 256, 1048576
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--streams", str(S), "--samples", str(L), "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[:2000]
    out = json.loads(lines[0])
    rf = out["roofline"]
    assert out["validated"] is True
    assert rf["traffic_source"].startswith("counted in THIS run"), rf["traffic_source"]
    assert rf["traffic"] and rf["traffic"] > 0
    # the same command with the counting switched off: the stored table, and the line says that instead
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--streams", str(S), "--samples", str(L), "--steps", "2",
                         "--warmup", "1", "--no-cpu-baseline", "--no-live-traffic", "--serial"], env=env, capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stderr[-2000:]
    rf2 = json.loads([ln for ln in r2.stdout.splitlines() if ln.strip()][0])["roofline"]
    assert "not counted in this run" in rf2["traffic_source"]
    # one after the other the line prices its longest kernel: k_fm (4 B of int16 IQ in, 16 B of (fi,fq) out per 9600 Hz sample) or
    # k_fft (4 B in, 4 B of PSD out per sample)
    alg = S * L * (5.6 if rf2["kernel"] == "k_fm" else 8.0)
    assert rf2["kernel"] in ("k_fm", "k_fft") and 0.9 * alg < rf2["traffic"] < 1.3 * alg, (rf2["kernel"], rf2["traffic"], alg)
    # ... and the counted table of the first run agrees with it kernel by kernel
    per = rf["traffic_per_kernel"]
    assert 0.9 * S * L * 8.0 < per["k_fft"] < 1.3 * S * L * 8.0 and 0.9 * S * L * 5.6 < per["k_fm"] < 1.3 * S * L * 5.6, per
