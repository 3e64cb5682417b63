#!/usr/bin/env python3
"""bench.py -- IQ Msamples/s through FFT + FIR + BPSK demod on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload pipeline|fft|bpsk] [--streams S] [--samples L]

One STEP = one pass of the hot path over one HBM-resident batch of synthetic input:
    S streams x L int16 IQ samples (96 kHz FUNcube-style DBPSK carrying valid FEC frames, generated on
    the device) -> fft.java waterfall PSD of every 2048-sample frame  +  FUNcubeBPSKDemod (tuner, 27-tap
    /10, VCO, 65-tap matched filter, slicer, sync correlation) + FECDecoder of every synchronised frame.
N > 1: one rank per GPU (torch.distributed, backend nccl == RCCL); started either under torch.distributed.run (WORLD_SIZE
set) or from the bare command, in which case this process spawns one fresh child per rank before it touches the GPU.
Default = BASELINE config 5 as SURVEY.md 8d
defines it: --total-streams 8192 FIXED and sharded contiguously over the ranks (8192 / N streams per rank,
"scaling": "strong"; the N=1 line runs all 8192 streams, 32 GiB of IQ, on one GPU).  --streams S gives S streams
per rank instead ("scaling": "weak").  Every step ends with one all-gather of the fixed-size per-stream result
slots over xGMI.  value = samples processed by all ranks / max-over-ranks time.

The JSON line also carries
    roofline     : dominant kernel (HIP events recorded on the launch stream inside the timed region); at N = 1 its `traffic`
                   (HBM bytes per launch) is COUNTED IN THE RUN: before its first GPU call this process runs its own command
                   line (two steps, no validation) as a child under `rocprofv3 --pmc FETCH_SIZE` and again under `--pmc
                   WRITE_SIZE` (live_traffic_table; ~6 s; --no-live-traffic, or a failing profiler: the stored table under profiles/)
    cpu_baseline : the plain-C oracle (oracle/, a restatement of the Java arithmetic: kind "port") on the
                   host cores, bounded sample, rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# algorithmic HBM bytes per IQ sample: 4 B int16 pair read; fft writes 4 B psd (+8 B/frame);
# the demodulator writes ~0.0125 B of bits
BYTES_PER_SAMPLE = {"fft": 4.0 + 4.0 * 2050.0 / 2048.0, "bpsk": 4.0125, "pipeline": 4.0 + 4.0 * 2050.0 / 2048.0 + 0.0125,
                    "demod": 8.0, "fir": 4.0 + 1.6}  # fir (config 3): one double2 per 10 input samples  # demod.java: 4 B read, one (L,R) int16 pair written per sample
DEMOD_MODES = {"raw": 1, "am": 2, "nfm": 3, "wfm": 4}
N_FFT = 2048
RATE = 96000  # --rate overrides it (192000 = the FUNcube Dongle Pro+ default, JavaAudio.java:59, with frames of 19200)
SEED = 20020109
SEED_TONES = 20020107
# measured FP64 issue rate of the box, separately rounded v_mul_f64 + v_add_f64 (tools/microbench_fp64.hip,
# profiles/r01_fp64_microbench.txt: 31.22 T lane-ops/s); the exact-order demodulator's floor
FP64_ISSUE_TOPS = 31.2
# separately rounded FP64 operations per 9600 Hz sample of the exact-order chain: 27 taps x 2 rails x (mul + add) = 108
# (FUNcubeBPSKDemod.java:479-483) + 65 x 2 x 2 = 260 (:519-523)
FP64_OPS_PER_DS_SAMPLE = {"k_fm": 368.0, "k_front_reg": 108.0, "k_front": 108.0, "k_matched": 260.0}
# what binds each kernel, from the measurements cited in DESIGN.md 3 (not from the roofline arithmetic)
KERNEL_BOUND = {"k_fft": "hbm", "k_fir_batch": "hbm", "k_waterfall": "hbm", "k_fm": "fp64-issue", "k_front_reg": "fp64-issue",
                "k_front": "fp64-issue", "k_matched": "fp64-issue", "k_front_fft": "valu-issue", "k_front_fftm": "valu-issue", "k_front_fftm2": "valu-issue",
                "k_front_fft2x": "valu-issue", "k_acq_fwd": "valu-issue", "k_acq_inv": "valu-issue", "k_tail": "lds-issue/latency", "k_tail8": "hbm (16 B per 9600 Hz sample; a wave's latency below ~2000 streams)",
                "k_fec_bits+k_vitq+k_fec_rs": "valu-issue",
                "k_sync_t": "latency", "k_sync": "latency", "k_sync_fin": "latency", "k_fec_bpsk": "latency",
                "k_demod_front": "valu-issue", "k_demod_fused": "valu-issue", "k_demod_out": "hbm", "k_demod_mean": "latency"}


# what the counters say limits a kernel where that is not the roofline it is priced against ("bound" stays the roofline)
KERNEL_LIMIT = {"k_fft": "latency at 4 waves/SIMD: VALU issues 61 % of the cycles, HBM at 4.35 of the ~5 TB/s a copy of this shape "
                         "reaches, 46 % of the wave-cycles in s_waitcnt (profiles/r04_sq_counters_k_fft_*.txt)"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="pipeline", choices=["pipeline", "fft", "bpsk", "demod", "fir"])
    ap.add_argument("--total-streams", type=int, default=8192,
                    help="streams of the whole job, sharded over the GPUs (strong scaling, BASELINE config 5)")
    ap.add_argument("--streams", type=int, default=0, help="streams PER GPU (weak scaling) instead of --total-streams")
    ap.add_argument("--variant", default="exact", choices=["exact", "fast"],
                    help="demodulator arithmetic: exact-order FP64 (default) or FMA-contracted FP64 with margin-certified decisions")
    ap.add_argument("--side-by-side", action="store_true", help="pipeline workload: the CU split below 8192 streams per GPU as well")
    ap.add_argument("--dist-single", action="store_true",
                    help="rehearsal: --gpus 1 through the DISTRIBUTED code path -- torch.distributed with the RCCL backend and ONE rank "
                         "(process-group init on the device, gather stream, all_gather_into_tensor, the gather check): the closest a "
                         "one-GPU box gets to the driver's N > 1 run with the real backend")
    ap.add_argument("--no-autotune", action="store_true",
                    help="pipeline workload where the CU split may pay (8192 streams per GPU): run side by side without first timing three "
                         "steps of each form after the warm-up (profile runs: one form only)")
    ap.add_argument("--no-compare-serial", dest="compare_serial", action="store_false",
                    help="pipeline workload, side by side: skip the five steps of the one-after-the-other form that are otherwise timed on "
                         "the same box right after the timed region (roofline.one_after_the_other_ms_per_step: boxes differ by more "
                         "than the two forms do, so the pair belongs in one record); profile runs pass it so that per-kernel averages "
                         "are those of ONE form")
    ap.add_argument("--serial", action="store_true",
                    help="pipeline workload: PSD then demodulator on ONE stream, each kernel with the whole chip (the default until round 4). "
                         "Default now, from 8192 streams per GPU: side by side on two streams with the CU shares set (jsdr_fft_set_cu_share 2, jsdr_bpsk_set_cu_share 1)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="N = 1: do not count the HBM traffic in this run (two child runs of this command under rocprofv3 --pmc, "
                         "about a minute); roofline.traffic then comes from the stored table under profiles/")
    ap.add_argument("--live-traffic-timeout", type=float, default=240.0, help="seconds one counter pass may take")
    ap.add_argument("--psd-stream", action="store_true",
                    help="PSD kernel on a HIP stream of its own beside the demodulator (measured: within 2 %% of the default, "
                         "one after the other on one stream, whose per-kernel times are not inflated by the overlap)")
    ap.add_argument("--samples", type=int, default=1048576, help="IQ samples per stream per step")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-validate", action="store_true")
    ap.add_argument("--demod-mode", default="am", choices=["am", "nfm", "wfm", "raw"],
                    help="--workload demod: the demod.java chain (8f next-3) with filter, down-conversion and AGC on")
    ap.add_argument("--waterfall-width", type=int, default=0,
                    help="also paint every PSD frame as a waterfall pixel row of this width (waterfall.java:87-109)")
    ap.add_argument("--bpsk-frame", type=int, default=N_FFT,
                    help="--workload bpsk: samples per demodulator frame (blen/size); 9600 = the reference's default, the "
                         "only size at which its FFT-acquire mode (10 Hz bins) brings FEC blocks through")
    ap.add_argument("--fir-taps", type=int, default=27, choices=[21, 27, 65],
                    help="--workload fir: 27 = dsFilter (FUNcubeBPSKDemod.java:27-55), 65 = dmFilter (:58-77), 21 = fir.java weights(500,1500)")
    ap.add_argument("--fir-decim", type=int, default=10, choices=[1, 10, 20], help="--workload fir: decimation")
    ap.add_argument("--fft-acquire", action="store_true", help="demodulator in FFT-acquire mode (bpsk-dofft=1) instead of tune mode")
    ap.add_argument("--rate", type=int, default=96000, choices=[44100, 48000, 96000, 176400, 192000, 384000],
                    help="--workload bpsk: sample rate of the synthetic streams and of the demodulator (audio-rate); 192000 with "
                         "--bpsk-frame 19200 --fft-acquire is the FUNcube Dongle Pro+ default configuration")
    return ap.parse_args()


def make_inputs(J, S, L, stream0):
    """DBPSK streams stream0..stream0+S-1 on the device; returns (d_iq, payloads_dev, nframes)"""
    sps = RATE // 1200
    nfr = -(-L // (5200 * sps)) + 1
    pay = J.synth_payloads(SEED, stream0, S, nfr)
    d_sym = J.DeviceBuffer(S * nfr * 5200)
    J.fec_encode_dev(pay, S * nfr, d_sym)
    d_ds = J.DeviceBuffer(S * nfr * 5200)
    J.synth_diffsign(d_sym, nfr * 5200, S, d_ds)
    ct, st = J.binding.synth_carrier_tables(3000)
    keys = np.array([J.binding.synth_mix64((SEED * 0x9E3779B1 + stream0 + s) ^ 0xA5A5A5A5) for s in range(S)], np.uint64)
    d_iq = J.DeviceBuffer(S * L * 4)
    gain = int(round(1500.0 / 37837.0 * 32768.0))
    J.synth_dbpsk(d_iq, 2 * L, S, 0, L, d_ds, nfr * 5200, sps, 0, J.binding.synth_phase_inc_u32(13200.0, RATE),
                  J.DeviceBuffer.from_host(ct), J.DeviceBuffer.from_host(st), gain, J.DeviceBuffer.from_host(keys))
    J.binding.stream_sync(None)
    return d_iq, pay, nfr


def make_tone_frames(J, nframes, frame0=0):
    """per 2048-sample frame 1-3 complex tones (hash-chosen bin, amplitude 0.08-0.4 FS, random
    phase) + noise of sigma 0.01 FS (4-term Irwin-Hall on a counter hash), clipped to int16, seed 20020107; on the device"""
    ct, _ = J.binding.synth_carrier_tables(21100)  # x amp/256, amp in 32..159: 0.08 .. 0.4 of full scale
    gain = int(round(0.01 * 32767.0 * 32768.0 / 37837.0))  # noise_from_hash: sigma = 37837 * gain / 32768 int16 units
    d_iq = J.DeviceBuffer(nframes * N_FFT * 4)
    key = J.binding.synth_mix64(SEED_TONES)
    J.synth_tones(d_iq, frame0, nframes, N_FFT, J.DeviceBuffer.from_host(ct), gain, key)
    J.binding.stream_sync(None)
    return d_iq


def validate_fft(J, d_iq, d_psd, nframes, rate):
    """sampled frames of the batch against the oracle's fft.receive (fft.java:190-228): amplitude within 1e-5 of the frame
    peak, and the reported maximum is the first strict maximum of the kernel's own PSD with the reference's Hz rule
    (:208-221).  Outside the timed region; the oracle only checks."""
    import oracle_lib as O
    idx = sorted(set(int(v) for v in np.linspace(0, nframes - 1, 48)))
    worst = 0.0
    for f in idx:
        raw = d_iq.to_host(np.int16, 2 * N_FFT, offset_bytes=f * N_FFT * 4)
        psd = d_psd.to_host(np.float32, N_FFT + 2, offset_bytes=f * (N_FFT + 2) * 4)
        ref = O.fft_receive(O.convert_i16(raw), rate)
        lin_g = 10.0 ** (psd[:N_FFT].astype(np.float64) / 20)
        lin_r = 10.0 ** (ref[:N_FFT].astype(np.float64) / 20)
        worst = max(worst, float(np.abs(lin_g - lin_r).max() / lin_r.max()))
        k = int(np.argmax(psd[:N_FFT]))  # numpy: first maximal element == strict '<' running maximum (fft.java:208-211)
        pbin = 2 * k
        hz = pbin * rate // (2 * N_FFT) if pbin < N_FFT else -((2 * N_FFT - pbin) * rate // (2 * N_FFT))
        if psd[N_FFT + 1] != psd[k] or psd[N_FFT] != np.float32(hz):
            return False, {"frames_checked": len(idx), "first_bad_frame": f, "why": "argmax / Hz rule"}
        # the oracle's maximum is the same bin, or a bin within float32 rounding of it (mirror tones: SURVEY 7 hard part 6)
        kr = int(np.argmax(ref[:N_FFT]))
        if kr != k and abs(float(ref[kr]) - float(ref[k])) > 1e-3:
            return False, {"frames_checked": len(idx), "first_bad_frame": f, "why": "maximum bin differs from the oracle's"}
    return worst <= 1e-5, {"frames_checked": len(idx), "worst_amplitude_error_over_frame_peak": worst}


def validate_acquire(J, dem, d_iq, S, L, ncalls, frame, rate, prefer=(), do_fft=1, nsample=6, compare_state=True):
    """Sampled streams of the measured handle against the oracle replaying the SAME calls (both modes; the tune-mode pipeline
    line uses it with do_fft=0: FUNcubeBPSKDemod.java:366-397,466-595 -- the headline's persistent-stride k_fm, k_tail8 and
    the batch FEC at full width, compared bit for bit with the reference's restatement).
    FFT-acquire mode (FUNcubeBPSKDemod.java:406-464) cannot be checked by payload at every frame size -- the block-wise FFT
    filter puts a seam into the signal every frame and the reference itself loses frames.  So sampled streams of the
    measured handle are compared with the oracle replaying the SAME calls (ncalls x the step's input): every counter incl.
    centreBin, every state double, the last call's bits, the last call's FECDecode results.  `prefer`: streams to include
    (those the payload check found without a decoded frame: the oracle must lose them too).  Outside the timed region."""
    import threading
    import oracle_lib as O
    idx = sorted(set(list(prefer)[:3]) | set(int(v) for v in np.linspace(0, S - 1, nsample)))
    res = {}
    O.lib()  # (loaded, and rebuilt if stale, before the threads start)

    def one(st):
        raw = d_iq.to_host(np.int16, 2 * L, offset_bytes=st * L * 4)
        o = O.Bpsk(rate=rate, blen=4 * frame, tuning=12000, do_fft=do_fft)
        nb0 = nf0 = 0
        for k in range(ncalls):
            if k == ncalls - 1:
                nb0, nf0 = len(o.bits()), len(o.fec_results())
            o.receive_i16(raw)
        res[st] = (o.counters(), o.state(), o.bits()[nb0:], o.fec_results()[nf0:])

    th = [threading.Thread(target=one, args=(st,)) for st in idx]
    for t in th:
        t.start()
    for t in th:
        t.join()
    bad = []
    lost = 0
    for st in idx:
        oc, os_, obits, ofec = res[st]
        gc, gs, gbits, gfec = dem.counters(st), dem.state(st), dem.bits(st), dem.fec_results(st)
        sidx = list(range(18)) if do_fft else [0, 1, 2, 3, 4, 5] + list(range(8, 18))  # 6, 7: the FFT-acquire mode's own state
        ckeys = [k for k in oc if do_fft or k != "centreBin"]
        # (compare_state False: the fast variant's doubles differ from the reference's in their last bits by design -- its bits,
        #  counters and FECDecode bytes must not)
        same = all(gc[k] == oc[k] for k in ckeys) and (not compare_state or gs[sidx].tobytes() == os_[sidx].tobytes()) and np.array_equal(gbits, obits) and \
            len(gfec) == len(ofec) and all(x[0] == y[0] and np.array_equal(x[2], y[2]) for x, y in zip(gfec, ofec))
        if not same:
            bad.append(st)
        lost += 1 if oc["cntDec"] == 0 else 0
    return not bad, {"streams_compared_with_the_oracle": len(idx), "calls_replayed": ncalls, "streams_that_differ": bad,
                     "of_them_without_a_decoded_frame_in_the_oracle_too": lost}


def validate_demod(J, d_iq, d_audio, S, L, ncalls, mode, rate):
    """demod.java's chain (demod.java:398-483) is stateful from frame to frame (filter ring, carrier phase, the running mean of the AM
    detector): sampled streams are replayed through the oracle over ALL calls of the run, frame by frame, and the LAST call's int16
    audio must be bit-identical.  Outside the timed region."""
    import threading
    import oracle_lib as O
    idx = sorted(set(int(v) for v in np.linspace(0, S - 1, 4)))
    bad = []

    def one(st):
        raw = d_iq.to_host(np.int16, 2 * L, offset_bytes=st * L * 4)
        buf = O.convert_i16(raw)
        o = O.Demod(rate)
        o.configure(mode, 1, 1, 1)
        o.weights(3000, 15000)
        nfr = L // N_FFT
        last = None
        for k in range(ncalls):
            outs = [o.receive(buf[2 * f * N_FFT:2 * (f + 1) * N_FFT]) for f in range(nfr)]
            if k == ncalls - 1:
                last = np.concatenate(outs)
        got = d_audio.to_host(np.int16, 2 * L, offset_bytes=st * L * 4)
        if not np.array_equal(got, last):
            bad.append(st)

    th = [threading.Thread(target=one, args=(st,)) for st in idx]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return not bad, {"streams_compared_with_the_oracle": len(idx), "calls_replayed": ncalls, "streams_that_differ": sorted(bad),
                     "int16_audio_samples_per_stream": 2 * L}


def validate_fir(J, d_iq, d_fir, S, L, taps, decim, scale):
    """sampled streams of the batch against the oracle's RxDownSample operator (FUNcubeBPSKDemod.java:466-492): every
    output of the stream bit-identical.  Outside the timed region."""
    import oracle_lib as O
    no = L // decim
    idx = sorted(set(int(v) for v in np.linspace(0, S - 1, 6)))
    for st in idx:
        raw = d_iq.to_host(np.int16, 2 * L, offset_bytes=st * L * 4)
        got = d_fir.to_host(np.float64, 2 * no, offset_bytes=st * no * 16).reshape(no, 2)
        want = O.fir_decimate(raw, taps, decim, scale)
        if got.tobytes() != want.tobytes():
            return False, {"streams_checked": len(idx), "first_bad_stream": st}
    return True, {"streams_checked": len(idx), "outputs_per_stream": no}


def usable_cores():
    """threads this process may really run at once: affinity mask, capped by the cgroup CPU quota (a GPU box
    shows all host cores but grants a share of them)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except Exception:
            continue
    return max(1, min(n, int(os.environ.get("JSDR_CPU_THREADS", "64"))))


def cpu_baseline(workload, L, seconds):
    """the oracle on the host cores: one stream per thread (ctypes releases the GIL), repeated until
    ~`seconds` of wall time; same signal family, same frame size.  The only place bench.py touches oracle/."""
    import oracle_lib as O
    cores = usable_cores()
    Lc = min(L, 1048576)
    nframes = Lc // N_FFT
    if workload == "fft":  # config 2's tone frames, the generator's CPU twin
        ct = O.synth_tables(21100)[0]
        gain = int(round(0.01 * 32767.0 * 32768.0 / 37837.0))
        streams = [O.synth_tones(s * nframes, nframes, N_FFT, ct, gain, O.mix64(SEED_TONES)) for s in range(cores)]
    else:
        streams = [O.make_dbpsk_stream(SEED, s, Lc)[0] for s in range(cores)]
    psd = np.empty(N_FFT + 2, np.float32)

    def one_pass(s, dem):
        if workload.startswith("demod"):
            buf = streams_f[s]
            for f in range(nframes):
                dem.receive(buf[2 * f * N_FFT:2 * (f + 1) * N_FFT])
            return
        if workload in ("pipeline", "fft"):
            O.lib().jo_bench_fft(streams[s].ctypes.data, nframes, N_FFT, RATE, psd.ctypes.data if s == 0 else None)
        if workload in ("pipeline", "bpsk"):
            dem.receive_i16(streams[s])

    def make_dem():
        if not workload.startswith("demod"):
            return O.Bpsk()
        d = O.Demod(RATE)
        d.configure(DEMOD_MODES[workload.split(":")[1]], 1, 1, 1)
        d.weights(3000, 15000)
        return d

    streams_f = [O.convert_i16(x) for x in streams] if workload.startswith("demod") else None
    # calibrate on one thread
    dem0 = make_dem()
    t0 = time.perf_counter()
    one_pass(0, dem0)
    t1 = time.perf_counter() - t0
    dems = [make_dem() for _ in range(cores)]
    # one parallel round to see what a pass costs with every thread busy (shared caches, CPU quota)
    th = [threading.Thread(target=one_pass, args=(s, dems[s])) for s in range(cores)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    tpar = time.perf_counter() - t0
    reps = max(1, min(int(seconds / max(tpar, 1e-3)), 400))

    def worker(s):
        for _ in range(reps):
            one_pass(s, dems[s])

    th = [threading.Thread(target=worker, args=(s,)) for s in range(cores)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    total = cores * reps * Lc
    return {"value": round(total / dt / 1e6, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"{cores} streams x {reps} passes x {Lc} samples ({workload}), 1 thread/stream, "
                      f"{dt:.1f} s wall; single-thread {Lc / t1 / 1e6:.2f} Msamples/s"}


def workload_text(a):
    mode = "FFT-acquire mode (bpsk-dofft)" if a.fft_acquire else "tune mode"
    return {"pipeline": f"fft.java PSD of every 2048-sample frame + FUNcubeBPSKDemod ({mode}) + FECDecoder, same HBM-resident IQ",
            "fft": "batched 2048-pt waterfall FFT+PSD (BASELINE config 2)",
            "bpsk": f"FUNcubeBPSKDemod {mode} + FECDecoder (BASELINE config 4), {a.bpsk_frame}-sample frames",
            "demod": f"demod.java {a.demod_mode.upper()} chain (8f next-3): 21-tap complex FIR + NCO + detector + AGC -> "
                     "int16 stereo, 2048-sample frames",
            "fir": f"batched {a.fir_taps}-tap complex FIR + decimate by {a.fir_decim} over int16 IQ (BASELINE config 3), "
                   "double2 outputs"}[a.workload]


def variant_text(a):
    if a.workload == "fft":
        return "float32 Stockham FFT + PSD (1e-5 of frame peak)"
    if a.workload == "demod":
        return "exact-order float32 (bit-exact int16 audio)"
    if a.workload == "fir":
        return "exact-order FP64 (newest sample first, products and sums rounded separately: bit-identical to the Java loop)"
    front = "FFT-acquire front end" if a.fft_acquire else "tuner front end"
    if a.variant == "fast":
        return f"FMA-contracted FP64, every slicer decision margin-certified or recomputed in exact order (bit-exact bits/bytes), {front}"
    return f"exact-order FP64 (bit-exact bits/bytes and doubles), {front}"


def stream_text(a, psd_own_stream, dem):
    """what ran where; the side-stream part is the handle's own answer (jsdr_bpsk_side_stream), not a copy of its rule"""
    side = "tail/sync/FEC on the handle's side stream" if (dem is not None and dem.side_stream()) else "tail/sync/FEC on the same stream"
    if a.workload == "pipeline":
        if psd_own_stream and not a.psd_stream:
            return ("PSD kernel (2 persistent workgroups per CU) and the demodulator's front-end kernel (1 per CU) side by side on two "
                    "streams; ") + side
        return ("PSD on its own HIP stream beside the demodulator; " if psd_own_stream else "PSD then demodulator on one stream; ") + side
    if a.workload == "bpsk":
        return ("front end, matched filter on the caller's stream; " if a.fft_acquire else "") + side
    return "one stream"


def backend_text():
    b = os.environ.get("JSDR_BENCH_BACKEND", "nccl")
    same = os.environ.get("JSDR_BENCH_SAME_DEVICE", "0") == "1"
    return ("RCCL" if b == "nccl" else b) + (", rehearsal: all ranks on device 0" if same else "")


def launch_ranks(N):
    """`python bench.py --gpus N` without a launcher: start N fresh child processes, one per rank (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* as torch.distributed.run would set them), BEFORE anything in this process has touched the GPU
    (no exec from a process that has).  Rank 0's stdout (the JSON line) is relayed; every other rank's goes to stderr.
    Returns the exit code: 0 only when every rank ended with 0."""
    import subprocess
    import tempfile

    # rendezvous through a file store (JSDR_BENCH_RDZV_FILE -> init_method=file://...): no port to pick, so none that
    # another process can take between picking and binding
    fd, rdzv = tempfile.mkstemp(prefix="jsdr_bench_rdzv_")
    os.close(fd)
    os.unlink(rdzv)  # the FileStore creates it
    procs = []
    for r in range(N):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(N), LOCAL_WORLD_SIZE=str(N),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT="0", JSDR_BENCH_RDZV_FILE=rdzv)
        # dmabuf IPC: this image's host driver supports no other kind, and without the setting RCCL's buffer exchange between
        # the ranks' processes fails with `hipIpcGetMemHandle: invalid argument` (the environment's note on multi-process
        # GPU work; the image exports it already -- setdefault only covers a caller that cleaned the environment)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    pending = dict(enumerate(procs))
    limit = float(os.environ.get("JSDR_BENCH_LAUNCH_TIMEOUT", "1500"))  # a rank stuck in a collective must not hang the driver
    t_start = time.monotonic()
    t_term = None  # when the survivors were told to end
    while pending:
        for r, p in list(pending.items()):
            code = p.poll()
            if code is None:
                continue
            del pending[r]
            if code != 0:
                print(f"[bench] rank {r} exited with {code}", file=sys.stderr)
                rc = rc or (code if code > 0 else 1)
                if t_term is None:
                    for q in pending.values():  # a rank is gone: the others would wait in a collective for ever
                        q.terminate()
                    t_term = time.monotonic()
        now = time.monotonic()
        if pending and t_term is None and now - t_start > limit:
            print(f"[bench] ranks {sorted(pending)} still running after {limit:.0f} s: ending the run", file=sys.stderr)
            rc = rc or 124
            for q in pending.values():
                q.terminate()
            t_term = now
        if pending and t_term is not None and now - t_term > 10.0:  # a rank that ignores SIGTERM
            for q in pending.values():
                q.kill()
            t_term = now + 1e9
        time.sleep(0.05)
    try:
        os.unlink(rdzv)
    except OSError:
        pass
    return rc


def live_traffic_table(a, samples_per_launch):
    """HBM bytes per kernel launch COUNTED IN THIS RUN: this very command (two steps after one warm-up, no validation, no CPU leg)
    as a child process under `rocprofv3 --pmc FETCH_SIZE` and once more under `--pmc WRITE_SIZE` -- separate passes, as
    MI355X_MICROARCH.md prescribes -- before this process makes its first GPU call (so the child has the whole HBM, and nothing
    here is an exec from a process that holds the GPU).  Returns (table, None), or (None, why) when the profiler is missing,
    fails or takes too long: the line then carries the stored table (profiles/pmc_traffic*.json) and says why.  A child that
    overruns is asked to stop (SIGTERM to the process group this function started) and waited for; if it has to be KILLED it may
    have died in the middle of a kernel, and this process then does no GPU work at all (the repository's own session rule:
    tools/gpu_session.sh) -- it exits non-zero."""
    import shutil
    import signal
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import pmc_traffic
    except Exception as ex:
        return None, f"tools/pmc_traffic.py not importable ({ex})"
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    # the child runs under the PARENT's interpreter (a real ELF binary: the program right after `--` must not be a launcher that
    # re-execs, and a different python3 on PATH may lack this process's packages)
    py = os.path.realpath(sys.executable) if sys.executable else "python3"
    child = [x for x in sys.argv[1:]] + ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-validate", "--no-compare-serial",
                                         "--no-autotune", "--no-live-traffic"]
    tmp = tempfile.mkdtemp(prefix="jsdr_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", JSDR_BENCH_CHILD="1")
    t0 = time.perf_counter()
    try:
        dirs = {}
        # (third pass: the SQ counters of roofline.valu_issue_frac -- the limit that binds the exact-order step is VALU issue, not HBM)
        for counter in ("FETCH_SIZE", "WRITE_SIZE", "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU"):
            d = os.path.join(tmp, counter.split()[0])
            with open(os.path.join(tmp, counter.split()[0] + ".log"), "w") as log:
                pr = subprocess.Popen([prof, "--pmc"] + counter.split() + ["--output-format", "csv", "-d", d, "--", py,
                                       os.path.join(ROOT, "bench.py")] + child, cwd="/tmp", env=env, stdout=log, stderr=log,
                                      start_new_session=True)
                try:
                    rc = pr.wait(timeout=a.live_traffic_timeout)
                except subprocess.TimeoutExpired:
                    os.killpg(pr.pid, signal.SIGTERM)  # the process group this function started, nothing else
                    try:
                        pr.wait(timeout=30)
                    except subprocess.TimeoutExpired:
                        os.killpg(pr.pid, signal.SIGKILL)
                        pr.wait()
                        raise SystemExit(f"bench.py: the --pmc {counter} child neither finished in {a.live_traffic_timeout:.0f} s nor "
                                         "stopped on SIGTERM and had to be killed, possibly inside a kernel: no measurement on "
                                         "this GPU from this process (re-run with --no-live-traffic)")
                    return None, f"the --pmc {counter} pass took more than {a.live_traffic_timeout:.0f} s and was stopped (SIGTERM, exited)"
            if rc != 0:
                tail = ""
                try:
                    tail = open(os.path.join(tmp, counter.split()[0] + ".log")).read()[-300:].replace("\n", " | ")
                except OSError:
                    pass
                return None, f"the --pmc {counter} pass exited with {rc}: {tail}"
            dirs[counter.split()[0]] = d
        tab = pmc_traffic.table(dirs["FETCH_SIZE"], dirs["WRITE_SIZE"], samples_per_launch)
        try:
            tab["_valu"] = pmc_traffic.valu_table(dirs["SQ_ACTIVE_INST_VALU"], 3)  # (the child runs --steps 2 --warmup 1)
        except Exception as ex:
            tab["_valu"] = {"_error": f"{type(ex).__name__}: {ex}"}
        if not any(k.startswith("k_") for k in tab):
            return None, "the counter output names no kernel of this library"
        tab["_seconds"] = round(time.perf_counter() - t0, 1)
        return tab, None
    except Exception as ex:
        return None, f"{type(ex).__name__}: {ex}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    global RATE
    a = parse()
    if a.rate != RATE:
        if a.workload != "bpsk":
            raise SystemExit("--rate needs --workload bpsk (the pipeline line is BASELINE's 96 kHz configuration)")
        RATE = a.rate
    knobs = sorted(k for k in os.environ if k.startswith("JSDR_EXPERIMENT_") or k.startswith("JSDR_FAST_"))
    if os.environ.get("JSDR_KNOBS") and os.environ.get("JSDR_BENCH_ALLOW_KNOBS") != "1":
        # the library only listens to its tuning knobs with JSDR_KNOBS=1; a measured line must say so itself (tools/ab_*.sh do)
        knobs.append("JSDR_KNOBS (without JSDR_BENCH_ALLOW_KNOBS=1)")
    if knobs:
        raise SystemExit(f"bench.py: {', '.join(knobs)} set -- experiment knobs make the product skip work / change what the fast "
                         "variant certifies; refusing to measure")
    N = a.gpus
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    torch = None
    D = N > 1 or a.dist_single  # the distributed code path
    if a.dist_single and N == 1 and "WORLD_SIZE" not in os.environ:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        world = 1
    saved_stdout = None
    if D and "WORLD_SIZE" in os.environ:
        # RCCL prints a banner ("RCCL version : ...", "Librccl path : ...") on STDOUT when its first communicator comes up, from every
        # rank; the contract is ONE JSON line on rank 0's stdout.  So in the distributed path file descriptor 1 points at stderr for
        # the whole run and is put back only around rank 0's JSON line.
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
    if D:
        if "WORLD_SIZE" not in os.environ:
            # the bare command (`python bench.py --gpus N`): this process has made no GPU call yet and never will --
            # it starts one fresh child per rank and relays rank 0's JSON line
            raise SystemExit(launch_ranks(N))
        if world != N:
            raise SystemExit(f"--gpus {N} under a launcher needs WORLD_SIZE={N}; got {world}")
        if os.environ.get("JSDR_BENCH_TEST_HANG_RANK") == str(rank):  # tests/test_gpu_bench_launcher.py: a stuck rank
            time.sleep(3600)
        import torch  # noqa: F811  (first, so libjsdr_hip.so binds to the same HIP runtime)
        import torch.distributed as dist  # noqa: F811
        # rehearsal knobs for a 1-GPU box (never set by the driver): JSDR_BENCH_SAME_DEVICE=1 maps every rank to
        # device 0 and JSDR_BENCH_BACKEND=gloo swaps RCCL for gloo, so the N>1 code path can be exercised there
        same_dev = os.environ.get("JSDR_BENCH_SAME_DEVICE", "0") == "1"
        backend = os.environ.get("JSDR_BENCH_BACKEND", "nccl")
        dev_index = 0 if same_dev else local_rank
        torch.cuda.set_device(dev_index)
        rdzv = os.environ.get("JSDR_BENCH_RDZV_FILE")  # set by launch_ranks(); under torch.distributed.run: env://
        kw = dict(init_method="file://" + rdzv, rank=rank, world_size=world) if rdzv else {}
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index), **kw)
            else:
                dist.init_process_group(backend, **kw)
        except BaseException as ex:
            # ONE readable line that names the rank (torch.distributed.run and launch_ranks() relay every rank's stderr; a rank that
            # dies here leaves the others waiting in their own init until the rendezvous times out, and each of them says so too)
            print(f"[bench] rank {rank} of {world} (device {dev_index}, {os.uname().nodename}): init_process_group({backend!r}) failed: "
                  f"{type(ex).__name__}: {str(ex).splitlines()[0] if str(ex) else ''} -- HSA_ENABLE_IPC_MODE_LEGACY="
                  f"{os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}, MASTER_ADDR={os.environ.get('MASTER_ADDR')}", file=sys.stderr, flush=True)
            raise
    live_tab, live_why = None, None
    under_profiler = any(k.startswith("ROCPROF") or k.startswith("ROCP_") for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    if (N == 1 and not D and not a.no_live_traffic and os.environ.get("JSDR_BENCH_CHILD") != "1" and not under_profiler and
            os.environ.get("JSDR_BENCH_LIVE_TRAFFIC", "1") != "0" and os.path.exists("/dev/kfd")):
        # (before this process's first GPU call; not when this run is itself being profiled, not for the A/B tooling's runs)
        S_ = a.streams if a.streams > 0 else a.total_streams
        L_ = (a.samples // a.bpsk_frame) * a.bpsk_frame if a.bpsk_frame != N_FFT else a.samples
        live_tab, live_why = live_traffic_table(a, S_ * L_)
        if live_tab is None:
            print(f"[bench] roofline.traffic falls back to the stored table: {live_why}", file=sys.stderr)
    import java_sdr_amd as J
    from java_sdr_amd import sharding as SH

    if not J.have_gpu():
        raise SystemExit("bench.py: no HIP device (libjsdr_hip.so has no CPU fallback)")
    if J.lib().jsdr_set_device((0 if os.environ.get("JSDR_BENCH_SAME_DEVICE", "0") == "1" else local_rank) if D else 0) != 0:
        raise SystemExit(J.lib().jsdr_last_error())

    if a.streams > 0:
        S, scaling = a.streams, "weak"
    else:
        if a.total_streams % N:
            raise SystemExit(f"--total-streams {a.total_streams} is not a multiple of --gpus {N}")
        S, scaling = a.total_streams // N, "strong"
    L = a.samples
    if a.bpsk_frame != N_FFT:
        if a.workload != "bpsk":
            raise SystemExit("--bpsk-frame needs --workload bpsk")
        L = (L // a.bpsk_frame) * a.bpsk_frame
    elif L % N_FFT:
        raise SystemExit("--samples must be a multiple of 2048")
    stream0, _ = SH.shard_streams(N * S, N, rank)  # contiguous shards: rank order == global stream order
    nframes = S * L // N_FFT
    if a.workload == "fft":
        if N * nframes < 262144:
            print(f"[bench] note: {N * nframes} frames; SURVEY 8d config 2 asks for >= 262144", file=sys.stderr)
        d_iq, pay, nfr = make_tone_frames(J, nframes, rank * nframes), None, 0
    else:
        d_iq, pay, nfr = make_inputs(J, S, L, stream0)
    fft = J.Fft(N_FFT, RATE) if a.workload in ("pipeline", "fft") else None
    d_psd = J.DeviceBuffer(nframes * (N_FFT + 2) * 4) if fft else None
    dem = J.Bpsk(rate=RATE, blen=4 * a.bpsk_frame, tuning=12000, do_fft=int(a.fft_acquire), nstreams=S, max_batch_samples=L,
                 variant=a.variant) if a.workload in ("pipeline", "bpsk") else None
    # the PSD kernel is HBM-bound, the demodulator FP64-issue bound: on streams of their own they share the CUs
    # every kernel of the timed loop goes to an explicit stream (the NULL stream would serialise with blocking streams)
    main_stream = J.Stream()
    ms_ = main_stream.ptr
    # pipeline default: the PSD kernel (memory-latency bound) and the demodulator's front-end kernel (FP64-issue bound) side by
    # side, each held to its share of every CU (2 + 1 persistent workgroups: what one CU's registers and LDS take) -- left
    # alone each fills the chip and the two streams just run one after the other
    # (measured, one session each: 34.1 against 35.2 ms at 8192 streams -- but 19.8 against 17.9 at 4096, 10.2 / 9.5 at 2048,
    #  5.9 / 5.1 at 1024, and 35.7 against 33.1 with the fast variant's kernels: the split pays only for the full single-GPU
    #  batch of the exact variant, which is where it is used)
    # WHERE the split pays is the library's knowledge (jsdr_bpsk_pair_shares: exact variant, tune mode, 96 kHz, 2048-sample frames, from
    # 8192 streams per device -- ADVICE r4: the policy lives in the library, jsdr_group_* asks the same function); --side-by-side
    # forces the 2 + 1 split on a smaller batch for A/B runs
    shares = dem.pair_shares() if (a.workload == "pipeline" and fft is not None and dem is not None) else (0, 0)
    if a.side_by_side and shares == (0, 0) and not a.fft_acquire and a.variant == "exact":
        shares = (2, 1)
    side_by_side = shares != (0, 0) and not a.serial and not a.psd_stream
    if side_by_side:
        fft.set_cu_share(shares[0])
        dem.set_cu_share(shares[1])
    psd_stream = J.Stream() if (fft is not None and dem is not None and (a.psd_stream or side_by_side)) else None
    form = {"ps": psd_stream.ptr if psd_stream else ms_, "sbs": side_by_side}

    def set_form(sbs):
        """side by side (the CU shares set, PSD on its own stream) or one after the other (no shares, one stream)"""
        fft.set_cu_share(shares[0] if sbs else 0)
        dem.set_cu_share(shares[1] if sbs else 0)
        form["ps"] = psd_stream.ptr if sbs else ms_
        form["sbs"] = sbs
    fir_taps = d_fir = None
    if a.workload == "fir":
        fir_taps = J.bpsk_table(0) if a.fir_taps == 27 else (J.bpsk_table(1) if a.fir_taps == 65 else J.Fir(44100.0).weights(500, 1500))
        d_fir = J.DeviceBuffer(S * (L // a.fir_decim) * 16)
        BYTES_PER_SAMPLE["fir"] = 4.0 + 16.0 / a.fir_decim
    fir_timer = [J.Timer() for _ in range(a.steps)] if a.workload == "fir" else []
    amfm = d_audio = None
    if a.workload == "demod":
        amfm = J.Demod(rate=RATE, n=N_FFT, nstreams=S, max_batch_samples=L)
        amfm.configure(DEMOD_MODES[a.demod_mode], 1, 1, 1)
        amfm.weights(3000, 15000)
        d_audio = J.DeviceBuffer(S * L * 4)
    slots = gathered = gstream = None
    if dem is not None and D:
        # the gather runs on its own stream: packing waits there for the side-stream tail of this step, so the
        # next step's FFT / front end on the main stream is not serialised behind it
        gstream = torch.cuda.Stream()
        info = dem.slot_info()
        slots = torch.empty(S * info["slot_bytes"], dtype=torch.uint8, device="cuda")
        gathered = torch.empty(N * S * info["slot_bytes"], dtype=torch.uint8, device="cuda")

    fft_timer = [J.Timer() for _ in range(a.steps)] if fft else []
    wf = a.waterfall_width if fft else 0
    d_pix = J.DeviceBuffer(nframes * wf * 4) if wf else None
    wf_timer = [J.Timer() for _ in range(a.steps)] if wf else []

    dem_calls = {"n": 0, "recovered": 0, "recover_ms": 0.0}

    def step(i, timed):
        ps = form["ps"]
        if fft is not None:
            if timed:
                fft_timer[i].start(ps)
            fft.batch_i16(d_iq, nframes, d_psd, stream=ps)
            if timed:
                fft_timer[i].stop(ps)
            if wf:
                if timed:
                    wf_timer[i].start(ps)
                J.waterfall_lines_dev(d_psd, nframes, N_FFT, wf, d_pix, stream=ps)
                if timed:
                    wf_timer[i].stop(ps)
        if fir_taps is not None:
            if timed:
                fir_timer[i].start(ms_)
            J.fir_batch_decimate_i16(d_iq, S, 2 * L, L, fir_taps, a.fir_decim, 0.9 * 32768.0, d_fir, L // a.fir_decim, stream=ms_)
            if timed:
                fir_timer[i].stop(ms_)
        if amfm is not None:
            amfm.batch_i16(d_iq, 2 * L, L, d_audio, 2 * L, stream=ms_)
        if dem is not None:
            dem.batch_i16(d_iq, 2 * L, L, stream=ms_)
            dem_calls["n"] += 1
            if D:
                dem.pack_slots(slots.data_ptr(), stream=gstream.cuda_stream)
                with torch.cuda.stream(gstream):
                    dist.all_gather_into_tensor(gathered, slots)  # == sharding.all_gather_slots, reused buffer

    def sync():
        if psd_stream is not None:
            psd_stream.sync()
        main_stream.sync()
        if dem is not None:
            dem.sync()  # the tail / FEC of the last step run on the handle's side stream
        if D:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
        else:
            J.binding.stream_sync(None)

    for i in range(a.warmup):
        step(i, False)
    sync()
    # Which form is faster depends on the BOX (same-session pairs of rounds 4 and 5: side by side 33.5 against 35.1, 34.2 / 35.7,
    # 34.8 / 36.0 -- but 37.2-39.1 against 37.0 on a slow one), and the gain is 0 +- 1 ms.  A few probe steps cannot decide that
    # (round 5 timed three steps of each and BENCH_r05 ran the form that was 0.6 ms slower than the rejected one; alternating
    # single steps measure a step WITHOUT the overlap of its tail with the next step's kernels and chose wrongly as well): where the
    # library says the split may pay, BOTH forms run the full timed region -- K steps each, from their own steady state -- and the
    # line is the faster one's; both figures and the choice go into the record.
    extra_calls = 0
    autotune = None

    def timed_region():
        if dem is not None:
            dem.profile_read()
            dem.profile_enable(True)
        if amfm is not None:
            amfm.profile_read()
            amfm.profile_enable(True)
        t0 = time.perf_counter()
        for i in range(a.steps):
            step(i, True)
        if dem is not None and a.variant == "fast" and not a.fft_acquire:
            # INSIDE the timed region: the streams the fast kernels could not certify are replayed -- every call
            # of the run, on the library's internal exact handle -- so that the line delivers every stream's results
            tr0 = time.perf_counter()
            dem_calls["recovered"] += dem.recover_uncertified([d_iq] * dem_calls["n"], [L] * dem_calls["n"], 2 * L, stream=ms_)
            dem_calls["recover_ms"] += (time.perf_counter() - tr0) * 1e3  # (host time to enqueue; the replay itself runs on the stream)
        sync()
        dt_ = time.perf_counter() - t0
        if dem is not None:
            dem.profile_enable(False)
        return dt_, {"fft_ms": [t.elapsed_ms() for t in fft_timer] if fft is not None else None,
                     "dem": dem.profile_read() if dem is not None else None}

    if side_by_side and N == 1 and not D and not a.no_autotune:
        runs = {}
        for sbs in (True, False):
            set_form(sbs)
            for _ in range(2):  # into this form's steady state
                step(0, False)
            sync()
            runs[sbs] = timed_region()
        extra_calls = 4 + a.steps
        best = runs[True][0] <= runs[False][0]
        set_form(best)
        dt, snap = runs[best]
        autotune = {"side_by_side_ms": round(runs[True][0] / a.steps * 1e3, 4), "one_after_the_other_ms": round(runs[False][0] / a.steps * 1e3, 4),
                    "chosen": "side by side" if best else "one after the other", "steps_each": a.steps,
                    "decided_on": "both forms ran the full timed region (the same K steps, each from its own steady state, side by side first); "
                                  "the line is the faster one's"}
        side_by_side = best
    else:
        dt, snap = timed_region()
    per_rank_ms = None
    if D:
        per_rank = [None] * N
        dist.all_gather_object(per_rank, dt)
        per_rank_ms = [round(v / a.steps * 1e3, 4) for v in per_rank]
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # ---- per-kernel durations measured inside the timed region
    kern = {}
    if fft is not None:
        ms = snap["fft_ms"]
        kern["k_fft"] = (float(np.sum(ms)), len(ms), BYTES_PER_SAMPLE["fft"])
    if fir_timer:
        ms = [t.elapsed_ms() for t in fir_timer]
        kern["k_fir_batch"] = (float(np.sum(ms)), len(ms), BYTES_PER_SAMPLE["fir"])
    if wf:
        ms = [t.elapsed_ms() for t in wf_timer]
        kern["k_waterfall"] = (float(np.sum(ms)), len(ms), 4.0 + 4.0 * wf / N_FFT)  # reads the PSD, writes the pixels
    if amfm is not None:
        amfm.profile_enable(False)
        for name, (ms, cnt) in amfm.profile_read().items():
            if cnt:
                kern[name] = (ms, cnt, BYTES_PER_SAMPLE["demod"])
    if dem is not None:
        # which kernels ran under the library's timing scopes: the front end (k_front_reg / k_front_fft / k_front_fftm / ...),
        # the tail (k_tail / k_tail8), the FEC form (k_fec_bpsk, or the batch form's three kernels under one scope)
        rename = {"k_front": dem.front_kernel_name(), "k_tail": dem.tail_kernel_name(), "k_fec_bpsk": dem.fec_kernel_name()}
        for name, (ms, cnt) in snap["dem"].items():
            if cnt:
                kern[rename.get(name, name)] = (ms, cnt, BYTES_PER_SAMPLE["bpsk"])
    # the dominant kernel is taken on the critical path: the tail / sync / FEC kernels run on the handle's side
    # stream under the next step's throughput kernels (their times are listed, they do not bound the step)
    SIDE = ("k_tail", "k_tail8", "k_sync", "k_sync_t", "k_sync_fin", "k_fec_bpsk", "k_fec_fin", "k_fec_bits+k_vitq+k_fec_rs")
    main = {k: v for k, v in kern.items() if k not in SIDE or a.fft_acquire}
    dom = max(main, key=lambda k: main[k][0])
    dom_ms = kern[dom][0] / kern[dom][1]
    alg_bytes = kern[dom][2] * S * L
    achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if a.fft_acquire and os.path.exists(os.path.join(ROOT, "profiles", f"pmc_traffic_acq{a.bpsk_frame}.json")):
        pmc = os.path.join(ROOT, "profiles", f"pmc_traffic_acq{a.bpsk_frame}.json")
    stored_tab = None
    if os.path.exists(pmc):
        try:
            stored_tab = json.load(open(pmc))
        except Exception:
            stored_tab = None
    tab = live_tab if live_tab is not None else stored_tab
    if tab:
        per = tab.get(dom, {}).get("hbm_bytes_per_launch")  # measured at tab["_samples_per_launch"] samples a launch
        traffic = int(per * (S * L) / tab.get("_samples_per_launch", 1024 * 1048576)) if per else None
    if live_tab is not None:
        traffic_source = (f"counted in THIS run ({live_tab['_seconds']} s): this command (--steps 2 --warmup 1) as a child process under "
                          "rocprofv3 --pmc FETCH_SIZE and again under --pmc WRITE_SIZE (separate passes; gfx950 correction 2 x FETCH + WRITE), "
                          "mean per launch")
    else:
        traffic_source = (os.path.relpath(pmc, ROOT) + ": separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                          "workload (gfx950 correction 2 x FETCH + WRITE), scaled by streams x samples; not counted in this run" +
                          (f" ({live_why})" if live_why else ""))
    nds_per_launch = S * (L // (RATE // 9600))  # 9600 Hz samples one launch of the demodulator's kernels covers

    def kernel_entry(k, v):
        ms = v[0] / v[1]
        if k in SIDE and not a.fft_acquire:
            # the 9600 Hz tail / sync / FEC kernels on the side stream: their HBM traffic is a few percent of the step's and
            # their times are stretched by whatever they run beside -- listed, not priced against the HBM roofline
            return {"avg_launch_ms": round(ms, 4), "frac": None, "bound": KERNEL_BOUND.get(k, "latency"),
                    "note": "side stream, measured while sharing the CUs with the main stream's kernels"}
        e = {"avg_launch_ms": round(ms, 4), "algorithmic_bytes_per_launch": int(v[2] * S * L),
             "frac": round(v[2] * S * L / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "bound": KERNEL_BOUND.get(k, "hbm")}
        if k in KERNEL_LIMIT:
            e["binding_limit"] = KERNEL_LIMIT[k]
        if k in FP64_OPS_PER_DS_SAMPLE and a.variant == "exact":
            # the exact-order FIRs cannot be contracted: their separately rounded FP64 operations against the box's
            # measured issue rate (tools/microbench_fp64.hip, profiles/r01_fp64_microbench.txt)
            e["fp64_issue_frac"] = round(FP64_OPS_PER_DS_SAMPLE[k] * nds_per_launch / (ms * 1e-3) / (FP64_ISSUE_TOPS * 1e12), 4)
        return e

    # "bound" stays the contract's hbm|mfma word for the roofline the fraction is taken against; what really limits
    # the dominant kernel is "binding_limit" (and per kernel in per_kernel[...]["bound"])
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_source": traffic_source if traffic is not None else None,
                "binding_limit": KERNEL_LIMIT.get(dom, KERNEL_BOUND.get(dom, "hbm")),
                "avg_launch_ms": round(dom_ms, 4), "algorithmic_bytes_per_launch": int(alg_bytes),
                "kernels_ms_per_step": {k: round(v[0] / v[1], 4) for k, v in sorted(kern.items())},
                # every critical-path kernel against the same roofline (the dominant one is the slowest AS MEASURED, i.e.
                # with whatever the side stream's kernels took from it)
                "per_kernel": {k: kernel_entry(k, v) for k, v in sorted(kern.items())
                               if v[0] / v[1] > 0.02 * dom_ms}}  # (not the helper kernels)
    if tab:
        # HBM bytes per launch of every kernel of a step (not the input generator's: k_synth_*, k_fec_encode run once, before the timed region)
        roofline["traffic_per_kernel"] = {k: int(v["hbm_bytes_per_launch"] * (S * L) / tab.get("_samples_per_launch", 1024 * 1048576))
                                          for k, v in sorted(tab.items()) if isinstance(v, dict) and "hbm_bytes_per_launch" in v
                                          and not k.startswith("k_synth") and k != "k_fec_encode"}
    if side_by_side:
        # k_fft and k_fm run CONCURRENTLY: each one's launch duration is the time it shared the chip with the other, not a time
        # it had the HBM to itself -- so the PAIR is the unit that is priced: the step's algorithmic bytes
        # (PSD in + out, demodulator in + bits) against the step's duration; the two kernels' own launch times are listed as
        # concurrent, without a fraction
        step_ms = dt / a.steps * 1e3
        step_bytes = BYTES_PER_SAMPLE["pipeline"] * S * L
        fm = dem.front_kernel_name()
        roofline["kernel"] = f"k_fft + {fm} (concurrent, two streams)"
        roofline["concurrent_kernels"] = ["k_fft", fm]
        roofline["achieved"] = round(step_bytes / (step_ms * 1e-3) / 1e9, 2)
        roofline["frac"] = round(step_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        roofline["avg_launch_ms"] = round(step_ms, 4)
        roofline["algorithmic_bytes_per_launch"] = int(step_bytes)
        roofline["step_algorithmic_bytes"] = int(step_bytes)
        roofline["step_frac"] = roofline["frac"]
        roofline["binding_limit"] = ("VALU issue of the pair: k_fm's separately rounded FP64 operations and k_fft's packed FP32 butterflies "
                                     "share every SIMD (profiles/r05_*: k_fm alone issues VALU for 90 % of its cycles)")
        if tab:
            # every kernel of a step (not the input generator's: k_synth_*, k_fec_encode run once, before the timed region)
            per = sum(v.get("hbm_bytes_per_launch", 0) for k, v in tab.items()
                      if isinstance(v, dict) and not k.startswith("k_synth") and k != "k_fec_encode")
            roofline["traffic"] = int(per * (S * L) / tab.get("_samples_per_launch", 1024 * 1048576)) if per else None
            roofline["traffic_source"] = traffic_source if roofline["traffic"] is not None else None
        for k in ("k_fft", fm):
            if k in roofline["per_kernel"]:
                e = roofline["per_kernel"][k]
                e["frac"] = None
                e.pop("fp64_issue_frac", None)
                e.pop("binding_limit", None)
                e["note"] = "concurrent with the other kernel of the pair: its launch time is time it SHARED the chip"
    if a.workload in ("pipeline", "bpsk") and not a.fft_acquire:
        roofline["note"] = ("the exact-order demodulator is FP64-issue bound, not HBM bound: " +
                            ("its per-kernel figures exist when it runs alone (bench.py --serial: per_kernel[k_fm].fp64_issue_frac) "
                             if side_by_side else "see per_kernel[k_fm].fp64_issue_frac ") +
                            f"(368 separately rounded operations per 9600 Hz sample against {FP64_ISSUE_TOPS} T lane-ops/s measured)")

    # ---- --compare-serial: the same pipeline one after the other, on this box, right after the timed region (boxes differ by
    # more than the two forms do): a short leg outside the timed region, reported beside the line's own number
    calls_made = a.warmup + a.steps + extra_calls
    if autotune is not None:
        roofline["autotune"] = autotune
    # ---- the limit that binds the exact-order step: VALU issue.  From the SQ counters of this run's own profiled child (third --pmc
    # pass): the step's VALU issue cycles per SIMD over the cycles the timed step lasted, at the engine clock the counters imply
    # (busy cycles per shader engine over the kernels' time under the profiler).
    vt = (live_tab or {}).get("_valu") if live_tab is not None else None
    if vt and "_error" not in vt:
        # (the step's kernels: launched at least once a step -- not the synthesis / encoder kernels of the set-up)
        kk = {k: v for k, v in vt.items() if v.get("launches_per_step", 0) >= 0.99}
        valu = sum(v["valu_cycles_per_simd"] for v in kk.values())
        busy = sum(v["busy_cycles_per_se"] for v in kk.values())
        ms_prof = sum(v["ms_under_profiler"] for v in kk.values())
        if valu > 0 and ms_prof > 0:
            clock = busy / (ms_prof * 1e-3)
            vfrac = valu / (dt / a.steps * clock)
            # (a short, VALU-bound step can come out above 1: the unprofiled run clocks higher than the serialised kernels of the
            #  counter pass did -- the fraction is capped and the raw ratio kept beside it)
            roofline["valu_issue_frac"] = round(min(1.0, vfrac), 4)
            roofline["valu_issue"] = {"valu_cycles_per_simd_per_step": int(valu), "engine_clock_ghz": round(clock * 1e-9, 3),
                                      "ratio_at_the_profiled_clock": round(vfrac, 4),
                                      "ms_of_pure_issue": round(valu / clock * 1e3, 3),
                                      "per_kernel": {k: {"valu_issue_frac_alone": round(v["valu_cycles_per_simd"] / v["busy_cycles_per_se"], 4) if v["busy_cycles_per_se"] else None,
                                                         "valu_wave_instructions_per_step": int(v["valu_wave_instructions"])}
                                                     for k, v in sorted(kk.items(), key=lambda kv: -kv[1]["valu_cycles_per_simd"])[:8]},
                                      "source": "this run's child under rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU "
                                                "(kernels serialised by the profiler; 1024 SIMDs, 32 shader engines)"}
    elif vt:
        roofline["valu_issue_frac"] = None
        roofline["valu_issue"] = vt
    if autotune is not None:
        # (both forms ran the full timed region: the other form's figure is that run's)
        roofline["side_by_side_ms_per_step" if not form["sbs"] else "one_after_the_other_ms_per_step"] = \
            autotune["side_by_side_ms" if not form["sbs"] else "one_after_the_other_ms"]
    elif shares != (0, 0) and psd_stream is not None and not a.serial and not a.psd_stream and N == 1 and not D and a.compare_serial:
        # the OTHER form, five steps right after the timed region
        calls_made += 7
        ran = form["sbs"]
        set_form(not ran)
        for _ in range(2):
            step(0, False)
        sync()
        ts = time.perf_counter()
        for _ in range(5):
            step(0, False)
        sync()
        roofline["side_by_side_ms_per_step" if not ran else "one_after_the_other_ms_per_step"] = round((time.perf_counter() - ts) / 5 * 1e3, 4)
        set_form(ran)

    # ---- validation outside the timed region: every sampled stream decodes what it was sent
    validated = None
    # (FFT-acquire mode is not validated by payload: the reference's block-wise FFT filter puts a seam into the
    #  signal every frame, and at 2048-sample frames its own demodulator -- the oracle bit for bit, see
    #  tests/test_gpu_bpsk.py -- rarely brings a 5200-bit FEC block through; parity for that mode is the tests')
    cert = None
    vstats = None
    none_streams = []
    if dem is not None and not a.no_validate and not (a.fft_acquire and (a.bpsk_frame != 9600 or RATE != 96000)):
        payloads = pay.to_host(np.uint8).reshape(S, nfr, 256)
        info = dem.slot_info()
        vslots = J.DeviceBuffer(S * info["slot_bytes"])
        dem.pack_slots(vslots)
        blob = vslots.to_host(np.uint8).reshape(S, info["slot_bytes"])
        # EVERY stream of this rank: at least one FEC frame decoded, and every decoded frame equals a payload that was
        # sent (a sync hit on the seam where the step's input repeats may fail to decode: rc = -1, as in the reference)
        n_none = n_wrong = n_failed = n_good = 0
        for s in range(S):
            fr = SH.unpack_slot(blob[s], info)["fec"]
            good = [r for r in fr if r[0] >= 0]
            n_good += len(good)
            n_failed += len(fr) - len(good)
            n_none += 0 if good else 1
            if not good and len(none_streams) < 8:
                none_streams.append(s)
            n_wrong += sum(0 if any(np.array_equal(r[2], payloads[s, f]) for f in range(nfr)) else 1 for r in good)
        # tune mode: every stream must bring a frame through.  FFT-acquire mode puts a seam into the signal every frame
        # (FUNcubeBPSKDemod.java:458-463) and loses some frames by design -- the oracle does, bit for bit (tests); there only a
        # WRONG payload invalidates the run
        ok = n_wrong == 0 and (n_none == 0 or a.fft_acquire)
        vstats = {"streams": S, "streams_without_decoded_frame": n_none, "decoded_frames_matching_no_sent_payload": n_wrong,
                  "sync_hits_whose_fec_decode_failed": n_failed, "decoded_frames": n_good}
        if not ok or n_failed:
            print(f"[bench] validation rank {rank}: {n_none} streams without a decoded frame, {n_wrong} decoded frames that "
                  f"match no sent payload, {n_failed} sync hits whose FEC decode failed (of {S} streams)", file=sys.stderr)
        validated = bool(ok)
        if a.fft_acquire and n_good < S // 2:
            # FFT-acquire mode may lose frames by design, but a run in which (almost) nothing decodes has checked nothing
            validated = None if n_good == 0 else False
    if dem is not None and not a.no_validate:
        # any frame size: sampled streams against the oracle replaying the same calls (the payload check above only exists
        # for the 9600-sample frame, and passes over streams without a decoded frame: here the oracle must lose those too).
        # Tune mode (the default line): 8 sampled streams, every call of the run, bit for bit -- counters, state doubles,
        # the last call's bits and FECDecode bytes (the fast variant's doubles differ by design: its bar is the payload check)
        # The fast variant: the same replay, its doubles excepted; the streams it recovered by replay are among the sampled ones.
        rec_ids = []
        if a.variant == "fast" and dem_calls["recovered"]:
            rec_ids = [st for st in range(S) if dem.is_recovered(st)][:3]
        ok_o, st_o = validate_acquire(J, dem, d_iq, S, L, dem_calls["n"], a.bpsk_frame, RATE, prefer=list(none_streams)[:3 - len(rec_ids)] + rec_ids,
                                      do_fft=int(a.fft_acquire), nsample=6 if a.fft_acquire else 8, compare_state=a.variant == "exact")
        if rec_ids:
            st_o["recovered_streams_among_them"] = rec_ids
        vstats = dict(vstats or {}, **st_o)
        validated = bool(ok_o) if validated is None else bool(validated and ok_o)
    if fft is not None and not a.no_validate:
        # the PSD the step wrote (also on the pipeline line: it used to go unread there)
        ok_f, st_f = validate_fft(J, d_iq, d_psd, nframes, RATE)
        vstats = dict(vstats or {}, psd=st_f)
        validated = bool(ok_f) if dem is None else bool(validated and ok_f)
    if fir_taps is not None and not a.no_validate:
        validated, vstats = validate_fir(J, d_iq, d_fir, S, L, fir_taps, a.fir_decim, 0.9 * 32768.0)
    if amfm is not None and not a.no_validate:
        J.binding.stream_sync(None)
        validated, vstats = validate_demod(J, d_iq, d_audio, S, L, a.warmup + a.steps, DEMOD_MODES[a.demod_mode], RATE)
    if dem is not None and a.variant == "fast":
        cert = dem.cert_stats()
        cert["streams_recovered_by_replay"] = dem_calls["recovered"]
        cert["calls_replayed_for_them"] = dem_calls["n"] if dem_calls["recovered"] else 0
        if cert["streams_uncertified"] > 0 and validated:
            # an uncertified stream's getters fail for good (include/jsdr_hip.h): its results were NOT delivered
            validated = False
            print(f"[bench] fast variant: {cert['streams_uncertified']} stream(s) ended uncertified -- validated: false",
                  file=sys.stderr)

    # ---- N > 1: the gathered buffer of the LAST timed step against this rank's own results
    gather_check = None
    if dem is not None and D:
        import hashlib
        info = dem.slot_info()
        sb = info["slot_bytes"]
        own = J.DeviceBuffer(S * sb)
        dem.pack_slots(own)
        J.binding.stream_sync(None)
        own_blob = own.to_host(np.uint8)
        g = gathered.cpu().numpy().reshape(N, S * sb)
        digests = [None] * N
        dist.all_gather_object(digests, hashlib.sha256(own_blob.tobytes()).hexdigest())
        seg_ok = [hashlib.sha256(g[r].tobytes()).hexdigest() == digests[r] for r in range(N)]
        # and sampled streams of this rank's segment against the handle's getters (bits, counters, FEC results)
        get_ok = True
        for s_ in sorted(set(int(v) for v in np.linspace(0, S - 1, 8))):
            u = SH.unpack_slot(g[rank][s_ * sb:(s_ + 1) * sb], info)
            bits = dem.bits(s_)
            c = dem.counters(s_)
            fr = dem.fec_results(s_)
            nb = min(len(bits), info["slot_bits"])
            get_ok &= int(u["header"][0]) == len(bits) and np.array_equal(u["bits"][:nb], bits[:nb])
            get_ok &= all(int(u["header"][2 + i]) == c[k] for i, k in enumerate(
                ("cntRaw", "cntDS", "cntBit", "cntFEC", "cntDec", "dmErrBits", "dmCorr", "dmMaxCorr", "decodeOK")))
            get_ok &= len(u["fec"]) == min(len(fr), info["nfec_max"]) and all(
                x[0] == y[0] and x[1] == y[1] and np.array_equal(x[2], y[2]) for x, y in zip(u["fec"], fr))
        flags = [None] * N
        dist.all_gather_object(flags, bool(all(seg_ok) and get_ok))
        # ... and every rank's own verdict on ITS streams (payloads / oracle): rank 0's line vouches for all of them
        rank_valid = [None] * N
        dist.all_gather_object(rank_valid, validated)
        gather_check = {"every_rank_segment_equals_that_ranks_slots": bool(all(seg_ok)),
                        "sampled_slots_equal_getters": bool(get_ok), "all_ranks_ok": bool(all(flags)),
                        "validated_per_rank": rank_valid}
        if not all(flags) or any(v is False for v in rank_valid):
            validated = False
        elif validated is not None and any(v is None for v in rank_valid):
            validated = None

    if rank == 0:
        total = float(N) * S * L * a.steps
        out = {
            "metric": "IQ Msamples/s through FFT+FIR+BPSK demod, 2048-pt frames",
            "value": round(total / dt / 1e6, 3),
            "unit": "Msamples/s",
            "n_gpus": N,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f64" if a.workload in ("pipeline", "bpsk", "fir") else "f32",  # the arithmetic type of the dominant path
            "data": "synthetic",
            "config": {"workload": workload_text(a), "streams_per_gpu": S, "total_streams": N * S, "samples_per_stream": L,
                       "rate_hz": RATE, "frame": a.bpsk_frame if a.workload == "bpsk" else N_FFT, "input_bytes_per_gpu": S * L * 4,
                       "variant": variant_text(a), "streams": stream_text(a, (psd_stream is not None) and (form["sbs"] or a.psd_stream), dem),
                       "parallelism": f"{N * S} streams sharded contiguously over {N} GPU(s)" +
                                      (f", one all-gather of result slots per step ({backend_text()})" if D else "")},
            "roofline": roofline,
            "hbm_read_roofline_frac": round(total / dt * 4.0 / (N * HBM_PEAK_GBS * 1e9), 4),
            "validated": validated,
        }
        if vstats is not None:
            out["validation"] = vstats
        if gather_check is not None:
            out["gather_check"] = gather_check
        if per_rank_ms is not None:
            out["ms_per_step_per_rank"] = per_rank_ms
            try:  # (torch's RCCL: what moved the slots)
                out["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version()) if os.environ.get("JSDR_BENCH_BACKEND", "nccl") == "nccl" else None
            except Exception:
                out["rccl_version"] = None
        if cert is not None:
            out["certification"] = cert
        if N == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.workload if a.workload != "demod" else "demod:" + a.demod_mode, L, a.cpu_seconds)
        sys.stdout.flush()
        if saved_stdout is not None:
            os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        if saved_stdout is not None:
            os.dup2(2, 1)
    if D:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
