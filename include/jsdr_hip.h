/*
 * jsdr_hip.h -- C ABI of libjsdr_hip.so: the MI355X (gfx950) implementation of java-sdr's
 * FFT / FIR / BPSK-demod hot path.
 *
 * This is the drop-in boundary.  java-sdr itself defines no native ABI (it is pure Java); each
 * entry point below replaces the arithmetic of one reference method and is what a JNI shim
 * for that class binds (INTEGRATION.md shows the stubs).  Citations are file:line in the
 * reference tree.
 *
 * Conventions:
 *   - every function returns JSDR_OK (0) or JSDR_ERR (-1)  (cf. FCD.OK/ERR, FCD.java:46-47);
 *     on error jsdr_last_error() returns a thread-local message.  Nothing throws or aborts.
 *   - handles are opaque; one handle is used by one thread at a time (the reference calls
 *     every handler from its single audio thread, JavaAudio.java:298-304).
 *   - "_host" buffers are caller-owned host memory, read/written before the call returns (the
 *     reference re-uses its frame buffers every iteration, JavaAudio.java:220-224).
 *   - "_dev" buffers are device (HBM) pointers valid on the current device; `stream` is a
 *     hipStream_t passed as void* (NULL = default stream); batch calls are asynchronous on it.
 *   - there is NO CPU fallback: without a usable HIP device every compute call fails.
 */
#ifndef JSDR_HIP_H
#define JSDR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JSDR_OK 0
#define JSDR_ERR (-1)

/* ------------------------------------------------------------------ runtime */
const char *jsdr_last_error(void);
int jsdr_version(void);                       /* ABI version, currently 1 */
int jsdr_device_count(int *count);
int jsdr_set_device(int device);
int jsdr_get_device(int *device);            /* the calling thread's current device (to restore after a per-device loop) */
int jsdr_device_name(char *buf, int cap);     /* gcnArchName of the current device */
int jsdr_malloc(void **dev, size_t bytes);
int jsdr_free(void *dev);
int jsdr_memset(void *dev, int value, size_t bytes);
int jsdr_memcpy_h2d(void *dev, const void *host, size_t bytes);
int jsdr_memcpy_d2h(void *host, const void *dev, size_t bytes);
/* a non-blocking HIP stream for the `stream` argument of the batch calls (independent handles on different streams
 * run concurrently: the HBM-bound PSD kernel beside the FP64-bound demodulator) */
int jsdr_stream_create(void **stream);
int jsdr_stream_destroy(void *stream);
int jsdr_stream_sync(void *stream);
/* HIP-event timing on `stream` (what bench.py brackets kernels with) */
int jsdr_timer_create(void **timer);
int jsdr_timer_destroy(void *timer);
int jsdr_timer_start(void *timer, void *stream);
int jsdr_timer_stop(void *timer, void *stream);
int jsdr_timer_elapsed_ms(void *timer, float *ms); /* synchronises on the stop event */

/* ------------------------------------------------------------------ sample conversion
 * JavaAudio.java:276-293: short s = LE16; s += (short)ic (wraps); f = (float)s/32767f; mono => Q=0.
 * Fused into every *_i16 kernel below; exposed on its own for parity tests.                */
int jsdr_convert_i16(const int16_t *raw_dev, int64_t nframes, int chns, int ic, int qc,
                     float *iq_dev, void *stream);

/* ------------------------------------------------------------------ fft.java
 * fft.receive (fft.java:190-228): complex forward FFT (replaces JTransforms FloatFFT_1D,
 * fft.java:194-195), psd[k]=10*log10((re^2+im^2)*(2/n)^2), first strict maximum, bin->Hz in
 * Java int arithmetic; output float[n+2] = what the reference publishes as "fft-psd".
 * n = blen/size (fft.java:67); supported: powers of two 64..8192 (2048 is the tuned size) and the
 * reference's default frames n = 4800 / 9600 (blen = rate*size/10 at 48 / 96 kHz, JavaAudio.java:58-59).  */
typedef struct jsdr_fft jsdr_fft;
int jsdr_fft_create(jsdr_fft **h, int n, int rate);
int jsdr_fft_destroy(jsdr_fft *h);
int jsdr_fft_receive_f32(jsdr_fft *h, const float *iq_host, float *psd_host);      /* IAudioHandler.receive(float[]) */
int jsdr_fft_receive_i16(jsdr_fft *h, const int16_t *raw_host, int ic, int qc,
                         float *psd_host);                                          /* IRawHandler.receive(byte[]) + a0 */
int jsdr_fft_batch_f32(jsdr_fft *h, const float *iq_dev, int64_t nframes, float *psd_dev, void *stream);
/* Running fft.receive and FUNcubeBPSKDemod.receive over the same batch SIDE BY SIDE (jsdr.java:476,482 feeds both from
 * one audio buffer): the PSD kernel is bound by memory latency, the demodulator's front-end kernel by FP64 issue, and
 * each, left alone, fills every CU -- launched on two streams they simply run one after the other.  With a share set on
 * both handles the batch kernels are launched as that many PERSISTENT workgroups per CU (0 = the default: all a CU takes):
 * jsdr_fft_set_cu_share(f, 2) + jsdr_bpsk_set_cu_share(h, 1) is the split that fits one CU's registers and LDS (2 x 128 +
 * 2 x 120 VGPRs per SIMD, 2 x 40 + 69.5 KB), both kernels resident from the start whichever stream gets there first.
 * Results are unchanged (bit for bit); measured on 8192 streams x 2^20 samples: 33.8 ms a step against 35.0 one after the
 * other (DESIGN.md "the step") -- and SLOWER than one after the other on batches of 4096 streams and below, and with the
 * fast variant: it is for the full-size batch of the exact variant.  A handle used on its own keeps the default. */
int jsdr_fft_set_cu_share(jsdr_fft *h, int wgs_per_cu);
/* diagnostics: what the handle's last power-of-two batch launch covered -- work items (groups of frames) and the
 * workgroups that strode over them.  workgroups < work_items means the persistent-stride path ran (the tests assert it). */
int jsdr_fft_last_launch(jsdr_fft *h, int64_t *work_items, int64_t *workgroups);
/* which kernel serves the handle's frame size: "k_fft" (powers of two 64 .. 8192), "k_fft_mixed" (9600 / 4800), "k_fft_mixed_dual"
 * (19200), "k_fft_rt" (any other composite frame up to 9800: 4410, 2205, 3200, 800, 1102 ...), "k_dft_any" (primes, and everything else up to 20 000) */
const char *jsdr_fft_kernel(jsdr_fft *h);
int jsdr_fft_batch_i16(jsdr_fft *h, const int16_t *raw_dev, int64_t nframes, int ic, int qc,
                       float *psd_dev, void *stream);
/* complex spectrum only (float2[n] per frame), for the 1e-5 FFT parity tests */
int jsdr_fft_spectrum_f32(jsdr_fft *h, const float *iq_dev, int64_t nframes, float *spec_dev, void *stream);

/* ------------------------------------------------------------------ phase.java
 * phase.java:75-80 max|x| over the 2n floats of a frame; :93-116 per-pixel-column means of I and Q
 * for a panel `bx` pixels wide.  cols_dev = int32[ncol_cap] pixel indices, avg*_dev float[ncol_cap].  */
int jsdr_phase_maxabs(const float *iq_dev, int64_t nframes, int n, float *max_dev, void *stream);
int jsdr_phase_columns(const float *iq_dev, int n, int bx, int32_t *pix_host, float *avgi_host,
                       float *avgq_host, int cap, int *ncol);
/* the IAudioHandler drop-in for phase.java: receive() copies the frame (phase.java:123-128) to the device and takes
 * max|x| at once; the painter asks for `max` (:75-80) and, for its panel width, the column means (:93-116). */
typedef struct jsdr_phase jsdr_phase;
int jsdr_phase_create(jsdr_phase **h, int n);
int jsdr_phase_destroy(jsdr_phase *h);
int jsdr_phase_receive_f32(jsdr_phase *h, const float *iq_host);   /* 2n floats */
int jsdr_phase_get_max(jsdr_phase *h, float *max_out);
int jsdr_phase_get_columns(jsdr_phase *h, int bx, int32_t *pix_host, float *avgi_host, float *avgq_host, int cap,
                           int *ncol);

/* ------------------------------------------------------------------ fir.java
 * weights (fir.java:169-195) is setup arithmetic (host); filter (:198-211) runs on the GPU over a
 * block of samples with carried delay line; complex_gen/complex_mod (:214-228) likewise.      */
typedef struct jsdr_fir jsdr_fir;
int jsdr_fir_create(jsdr_fir **h, float sample_rate);
int jsdr_fir_destroy(jsdr_fir *h);
int jsdr_fir_weights(jsdr_fir *h, int f1, int f2, double w_out[21]);
int jsdr_fir_filter(jsdr_fir *h, const int32_t *in_host, int32_t *out_host, int64_t n);
int jsdr_fir_complex_gen(jsdr_fir *h, int freq, int start, int32_t *sig_host /*[n][2]*/, int64_t n);
int jsdr_fir_complex_mod(jsdr_fir *h, const int32_t *a_host, const int32_t *b_host, int32_t *out_host, int64_t n);
/* batched complex FIR + decimate over device-resident int16 IQ streams (BASELINE config 3), exact-order FP64:
 *   out[s][j] = scale * SUM_{a<ntaps} x[s][decim*(j+1)-1-a] * taps[a]   per rail, newest sample first, every product
 *   and sum rounded separately; x = (double)((float)int16/32767f); samples before the batch are zero (cleared delay
 *   line).  = RxDownSample (FUNcubeBPSKDemod.java:466-492) for taps = dsFilter, decim = rate/9600, scale = 0.9*32768;
 *   any ntaps <= 128 and decim >= 1 (27/65/21 taps x decimation 1/10/20 have register-blocked kernels).
 * out_dev: double[nstreams][out_stride_pairs][2]; *nout = nsamples / decim outputs per stream.  Asynchronous. */
int jsdr_fir_batch_decimate_i16(const int16_t *raw_dev, int nstreams, int64_t stream_stride_i16, int64_t nsamples,
                                const double *taps_host, int ntaps, int decim, double scale, double *out_dev,
                                int64_t out_stride_pairs, int64_t *nout, void *stream);

/* ------------------------------------------------------------------ FECDecoder.java
 * FECDecode (FECDecoder.java:703-852): 5200 soft symbols -> 256 bytes, return -1 or channel errors.
 * encode_FEC40 (:677-688) is the (private) re-encoder, exposed for test-vector generation.     */
int jsdr_fec_decode(const uint8_t raw_host[5200], uint8_t out_host[256], int *rc);
int jsdr_fec_encode(const uint8_t data_host[256], uint8_t sym_host[5200]);
int jsdr_fec_decode_batch(const uint8_t *raw_dev, int64_t nblocks, uint8_t *out_dev, int32_t *rc_dev, void *stream);
int jsdr_fec_encode_batch(const uint8_t *data_dev, int64_t nblocks, uint8_t *sym_dev, void *stream);

/* ------------------------------------------------------------------ FUNcubeBPSKDemod.java
 * One handle = `nstreams` independent demodulators with identical configuration that advance in
 * lock-step (stream-major batches).  nstreams=1 + receive_* is the IAudioHandler drop-in.
 *   rate, blen/size : AudioDescriptor (FUNcubeBPSKDemod.java:193-194)
 *   tuning_hz       : "bpsk-tuning" (:195), do_fft "bpsk-dofft" (:199), do_up "bpsk-upper" (:200)
 * Frames: tune mode takes any frame; FFT-acquire mode (do_fft) frames of 1024 / 2048 / 4096 / 8192 samples, any other
 * n = 2^a 3^b 5^c 7^d from 1025 to 9600 (java-sdr's defaults 9600 / 4800 at 96 / 48 kHz, 4410 at 44.1 kHz, 3200 at 32 kHz)
 * and twice a 2^a 3^b 5^c frame that is a multiple of 16 (19200, the 192 kHz default) through kernels that hold the frame
 * in LDS; every other frame of 416 .. 4194304 samples (and power-of-two / 2 m frames below 38400 Hz) through passes in
 * global memory -- the same results, about ten times the time per sample.  Below 416 samples: refused (the reference's
 * arraycopy of 204 bins, :458, would run off the frame).
 * max_batch_samples bounds one batch call; the per-call result log holds max_batch/40 + 16 bits and
 * max(8, bits/2600 + 4) FECDecode calls per stream -- a call that exceeds either flags the stream (getters fail).  */
typedef struct jsdr_bpsk jsdr_bpsk;
int jsdr_bpsk_create(jsdr_bpsk **h, int rate, int nsamples_per_frame, int tuning_hz, int do_fft,
                     int do_up, int nstreams, int64_t max_batch_samples);
int jsdr_bpsk_destroy(jsdr_bpsk *h);
/* the constant tables as the library holds them (host side): which = 0 dsFilter[27] (:27-55), 1 dmFilter[65] (:58-77),
 * 2 SYNC_VECTOR[65] (:79-81) as +1/-1 */
int jsdr_bpsk_table(int which, double *out, int cap);
/* arithmetic of the demodulator (before the first sample): EXACT = every double product and sum rounded separately in
 * the reference's order (bits, bytes AND doubles identical to the Java arithmetic); FAST = fused multiply-adds in the
 * two FIR stages, every slicer decision certified by a proven error margin or recomputed in exact order (bits and
 * bytes identical, doubles within 1e-12 relative).                                                          */
enum { JSDR_VARIANT_EXACT = 0, JSDR_VARIANT_FAST = 1 };
int jsdr_bpsk_set_variant(jsdr_bpsk *h, int variant);
/* FAST: decisions recomputed in exact order so far (all streams), streams that could not be certified (their getters
 * fail), and the bound on |fi' - fi|, |fq' - fq| the margins are built on.  In FAST mode the input buffer of a batch
 * call must stay unmodified until the next jsdr_bpsk_sync() / getter: the certification pass may re-read it. */
int jsdr_bpsk_cert_stats(jsdr_bpsk *h, int64_t *redone, int64_t *uncertified_streams, double *ey);
/* FAST, the contract for a drop-in: a stream whose 8-way energy argmax (FUNcubeBPSKDemod.java:586-592) falls inside the
 * error margin cannot be decided without the exact IIR history and is marked UNCERTIFIED by the call in which that
 * happens.  The mark is sticky: from that call on every getter of that stream (counters, bits, fec, decoded, trace,
 * state), the snapshot and receive_*() of a 1-stream handle return JSDR_ERR with a message that says so, and the
 * result slot's header[12] is 1 -- a wrong bit is never delivered, and neither is a silent gap.  Measured rate on the
 * benchmark's streams: about 1 stream in 8192 per 1.4e9 bit periods; a stream that carried signal and then
 * goes silent for ~5 s also ends here (its energies decay below the margin, which scales with the largest sample seen).
 * What the caller does: jsdr_bpsk_uncertified_streams() lists the stream ids after a call (count > 0 <=> some results
 * of that call were withheld); those streams are re-run on a JSDR_VARIANT_EXACT handle -- from the start of the stream,
 * or from the flagged call on an exact handle that was fed the same earlier input -- which decides the same near-tie by
 * the reference's own arithmetic (tests/test_gpu_fixtures.py::test_fast_variant_uncertified_streams_are_listed_and_
 * recovered_on_an_exact_handle).  There is no automatic fallback: it would need the exact state kept beside the fast
 * one for every stream, i.e. the exact variant's cost.  ids: ascending, at most cap; *count: all of them. */
int jsdr_bpsk_uncertified_streams(jsdr_bpsk *h, int32_t *ids, int cap, int *count);
/* Fast variant, batch use over retained input (round 6): REPLAYS every stream the calls so far left uncertified, from the
 * handle's first call, on an internal exact handle that serves those streams from then on -- their getters, their packed
 * slots; the flag no longer withholds them; jsdr_bpsk_uncertified_streams / _cert_stats stop counting them; every later
 * batch call advances the exact shadow in lock-step.  raw_dev_calls[k] / nsamples_calls[k]: the device pointer and sample
 * count jsdr_bpsk_batch_i16 was given at call k, ALL calls since creation (checked against the handle's own counts), the
 * buffers still holding those samples; ic / qc as in those calls.  *recovered = streams now served in exact order (0: none
 * was uncertified).  Cost: one small exact handle's call per call replayed.  Mirrors what a caller of
 * FUNcubeBPSKDemod.receive would do with a recording: run the doubtful streams again in the reference's own arithmetic
 * (FUNcubeBPSKDemod.java:533-594). */
int jsdr_bpsk_stream_recovered(jsdr_bpsk *h, int stream, int *recovered); /* 1: served by the exact shadow (replayed) */
int jsdr_bpsk_recover_uncertified(jsdr_bpsk *h, const int16_t *const *raw_dev_calls, const int64_t *nsamples_calls, int ncalls,
                                  int64_t stream_stride_i16, int ic, int qc, int *recovered, void *stream);
/* receive(float[]) / raw form for stream 0 of a 1-stream handle (:357-364) */
int jsdr_bpsk_receive_f32(jsdr_bpsk *h, const float *iq_host);
int jsdr_bpsk_receive_i16(jsdr_bpsk *h, const int16_t *raw_host, int ic, int qc);
/* batched: raw_dev[s*stream_stride + 2*t .. +1] = I,Q of sample t of stream s; nsamples must be a
 * multiple of nsamples_per_frame in FFT mode; asynchronous on `stream`.                        */
int jsdr_bpsk_batch_i16(jsdr_bpsk *h, const int16_t *raw_dev, int64_t stream_stride_i16,
                        int64_t nsamples, int ic, int qc, void *stream);
int jsdr_bpsk_set_cu_share(jsdr_bpsk *h, int wgs_per_cu); /* see jsdr_fft_set_cu_share; applies to the tune-mode front-end kernel */
/* The library's own answer to "should fft.receive and this demodulator, fed the same batch on two streams, share the CUs?":
 * the shares to pass to jsdr_fft_set_cu_share / jsdr_bpsk_set_cu_share (2 and 1 where the split was measured to pay: the
 * exact variant's tune-mode kernel, 96 kHz, 2048-sample frames, 8192 streams and more per device), 0 and 0 everywhere else.
 * bench.py and jsdr_group_* ask this instead of carrying the rule themselves. */
int jsdr_bpsk_pair_shares(jsdr_bpsk *h, int *fft_wgs_per_cu, int *bpsk_wgs_per_cu);
/* diagnostics: tiles x streams of the last tune-mode front-end launch (k_fm) and the workgroups that strode over them */
int jsdr_bpsk_last_launch(jsdr_bpsk *h, int64_t *work_items, int64_t *workgroups);
/* wait until every kernel of the calls made so far has finished (the 9600 Hz tail and the FEC decoder run on
 * an internal side stream so that they overlap the next call's front end; the getters below call this). */
int jsdr_bpsk_sync(jsdr_bpsk *h);
/* results */
enum { JSDR_BPSK_NCOUNTERS = 10 };
/* cntRaw,cntDS,cntBit,cntFEC,cntDec,dmErrBits,dmCorr,dmMaxCorr,decodeOK,centreBin (:110-115,:405,:498) */
int jsdr_bpsk_get_counters(jsdr_bpsk *h, int stream, int32_t out[JSDR_BPSK_NCOUNTERS]);
/* bits sliced during the LAST batch/receive call, +1/-1 each (:545-554) */
int jsdr_bpsk_get_bits(jsdr_bpsk *h, int stream, int8_t *bits_host, int cap, int *nbits);
/* FECDecode calls made during the last call: rc + index (1-based, within the call) of the bit that triggered */
int jsdr_bpsk_get_fec(jsdr_bpsk *h, int stream, int idx, int32_t *rc, int32_t *bit_index, uint8_t out_host[256]);
int jsdr_bpsk_get_fec_count(jsdr_bpsk *h, int stream, int *count);
int jsdr_bpsk_get_decoded(jsdr_bpsk *h, int stream, uint8_t out_host[256]);   /* decoded[] (:111) */
/* matched-filter outputs (fi,fq) of the last call, double[npairs][2] (:518-531) */
int jsdr_bpsk_get_trace(jsdr_bpsk *h, int stream, double *out_host, int64_t cap_pairs, int64_t *npairs);
/* scalar state, same 18-value layout as the oracle: tuPhase,vcoPhase,dmBitPhase,dmEnergyOut,energy1,
 * energy2,avePeakPower,aveCentreBin,dmEnergy[8],dmLastIQ[2]                                       */
int jsdr_bpsk_get_state(jsdr_bpsk *h, int stream, double out[18]);
/* Results of the last completed receive_f32 / receive_i16 of a 1-stream handle, for a reader on ANOTHER thread (the
 * reference's Swing thread paints these fields while the audio thread is inside receive(), FUNcubeBPSKDemod.java:
 * 220-228,331-337): double-buffered on the host, published after every receive; the read takes no lock, makes no
 * device call and never sees a half-written frame.  The getters above belong to the thread that calls receive. */
typedef struct jsdr_bpsk_snapshot {
    int64_t frames;                        /* receive() calls completed */
    int32_t counters[JSDR_BPSK_NCOUNTERS]; /* as jsdr_bpsk_get_counters */
    int32_t nbits;                         /* bits sliced during that frame (first 512 kept) */
    double state[18];                      /* as jsdr_bpsk_get_state */
    uint8_t decoded[256];                  /* decoded[] (:111) */
    int8_t bits[512];
} jsdr_bpsk_snapshot;
int jsdr_bpsk_snapshot_read(jsdr_bpsk *h, jsdr_bpsk_snapshot *out);
/* per-kernel HIP-event timing of the batch pipeline (events recorded on the caller's stream around each
 * launch while enabled).  profile_read sums and clears what was recorded since the last read.          */
int jsdr_bpsk_profile_enable(jsdr_bpsk *h, int on);
int jsdr_bpsk_profile_count(void);                 /* number of kernels in the pipeline */
const char *jsdr_bpsk_front_kernel(jsdr_bpsk *h);  /* name of the front-end kernel the last call launched */
const char *jsdr_bpsk_tail_kernel(jsdr_bpsk *h);   /* ... of its tail kernel: k_tail (one wave per stream) or k_tail8 (eight streams per wave) */
const char *jsdr_bpsk_fec_kernel(jsdr_bpsk *h);    /* ... of its FEC form: k_fec_bpsk, or the batch form's k_fec_bits+k_vitq+k_fec_rs */
/* 1 when the handle runs its tail / sync / FEC on a side stream of its own (batch handles), 0 when on the caller's stream
 * (1-stream handles; FFT-acquire with a mixed-radix frame; JSDR_NO_OVERLAP): *on receives it.  What a benchmark reports
 * instead of re-deriving the library's rule.                                                                        */
int jsdr_bpsk_side_stream(jsdr_bpsk *h, int *on);
/* the input-independent tuner / VCO schedules built so far: on the calling thread / taken from the look-ahead worker
 * (periodic configurations reuse one schedule and build none after the first calls) */
int jsdr_bpsk_schedule_stats(jsdr_bpsk *h, int64_t *computed_inline, int64_t *prefetched);
const char *jsdr_bpsk_profile_name(int k);
int jsdr_bpsk_profile_read(jsdr_bpsk *h, double *ms_total, int *launches);
/* device-resident result slots for the multi-GPU all-gather: per stream
 * int32 header[16] = {nbits, nfec, counters[10], 0...} followed by int8 bits[slot_bits] and
 * nfec_max x {int32 rc, int32 bit_index, uint8 data[256]}.  Layout constants via slot_info.      */
int jsdr_bpsk_slot_info(jsdr_bpsk *h, int64_t *slot_bytes, int64_t *bits_offset, int64_t *fec_offset,
                        int *slot_bits, int *nfec_max);
int jsdr_bpsk_pack_slots(jsdr_bpsk *h, uint8_t *slots_dev, void *stream);

/* ------------------------------------------------------------------ synthetic inputs (bench / tests)
 * Integer-only generators, bit-identical to oracle/o_synth.c.  Not part of the reference.        */
int jsdr_synth_diffsign(const uint8_t *sym_dev, int64_t nsym, int nstreams, int8_t *dsign_dev, void *stream);
int jsdr_synth_dbpsk(int16_t *out_dev, int64_t stream_stride_i16, int nstreams, int64_t n0, int64_t n,
                     const int8_t *dsign_dev, int64_t nsym, int samples_per_sym, uint32_t phase0,
                     uint32_t phase_inc, const int16_t *cos_tab_dev, const int16_t *sin_tab_dev,
                     int noise_gain, const uint64_t *noise_keys_dev, void *stream);
int jsdr_synth_tones(int16_t *out_dev, int64_t frame0, int64_t nframes, int n, const int16_t *cos_tab_dev,
                     int noise_gain, uint64_t key, void *stream);
int jsdr_synth_payloads(uint64_t seed, int stream0, int nstreams, int nframes, uint8_t *out_dev, void *stream);

/* ------------------------------------------------------------------ demod.java
 * The AM/FM IAudioHandler, demod.receive (demod.java:398-483), batched over `nstreams` independent streams that
 * share one set of controls: 21-tap complex float band-pass (filter(), :378-396), down-conversion NCO
 * (:423-434), AM envelope minus its running mean (:448-451) or FM quadrature-delay detector (:453-461), the
 * frame maximum, AGC and the (short)(x*32767f) stereo output (:465-481).  Float arithmetic in the reference's
 * order (no FMA), so int16 outputs are bit-identical to the CPU restatement; cos/sin of the NCO phase come
 * from the device math library.  A frame is blen/size samples (`sam.length/2`, :229-231); AGC and
 * the AM mean are per frame, the filter, NCO and FM states run on across frames and calls.                    */
typedef struct jsdr_demod jsdr_demod;
#define JSDR_DEMOD_OFF 0 /* demod.java:39-43 */
#define JSDR_DEMOD_RAW 1
#define JSDR_DEMOD_AM 2
#define JSDR_DEMOD_NFM 3
#define JSDR_DEMOD_WFM 4
int jsdr_demod_create(jsdr_demod **h, int rate, int nsamples_per_frame, int nstreams, int64_t max_batch_samples);
int jsdr_demod_destroy(jsdr_demod *h);
/* "demod-mode", "demod-fir-enable", the down-conversion toggle, "demod-agc-enable" (:36-38,198-203) */
int jsdr_demod_configure(jsdr_demod *h, int mode, int dofir, int dodwn, int doagc);
/* demod.weights() (:341-375) for the band [flo, fhi] Hz (flo = INT_MIN: the all-pass impulse); clears every
 * stream's delay line and (band-pass) restarts the carrier phase; returns the 21 weights and phi if asked.
 * The range check of filterMove (:300-312) is the caller's (host mirror: java_sdr::demod).                     */
int jsdr_demod_weights(jsdr_demod *h, int flo, int fhi, float w_out[21], float *phi_out);
/* whole frames of every stream; audio_dev receives one (L,R) int16 pair per input sample (:473-478) */
int jsdr_demod_batch_i16(jsdr_demod *h, const int16_t *raw_dev, int64_t stream_stride_i16, int64_t nsamples,
                         int ic, int qc, int16_t *audio_dev, int64_t audio_stride_i16, void *stream);
int jsdr_demod_batch_f32(jsdr_demod *h, const float *iq_dev, int64_t stream_stride_f32, int64_t nsamples,
                         int16_t *audio_dev, int64_t audio_stride_i16, void *stream);
/* IAudioHandler.receive(float[]) for a 1-stream handle: one frame (2n floats) in, 2n int16 out */
int jsdr_demod_receive_f32(jsdr_demod *h, const float *buf_host, int16_t *audio_host);
/* the reference's `max` / `avg` fields after the last frame of the last call (:465-467) */
int jsdr_demod_frame_stats(jsdr_demod *h, int stream, float *max_out, float *avg_out);
int jsdr_demod_state(jsdr_demod *h, float *car_out, float *phi_out);
/* per-kernel HIP-event timing of the batch calls (bench.py), as jsdr_bpsk_profile_* */
int jsdr_demod_profile_enable(jsdr_demod *h, int on);
int jsdr_demod_profile_count(void);
const char *jsdr_demod_profile_name(int kernel);
int jsdr_demod_profile_read(jsdr_demod *h, double *ms_total, int *launches);

/* ------------------------------------------------------------------ formats either side of the path
 * waterfall.paintLine / getMax (waterfall.java:87-109): one pixel row per PSD frame as fft.receive publishes it
 * ("fft-psd", n bins + 2).  step = (float)n/width; pixel p = max of bins [(int)(p*step), +(int)step), mapped
 * 255-(int)(m*-2.55f), clamped to 0..255, scaled into the peak colour (0xRRGGBB, Color.CYAN = 0x00ffff in the
 * reference) with /256, stored at column (p + width/2) % width as ARGB with alpha 0xff.  Bit-exact.          */
int jsdr_waterfall_lines(const float *psd_dev /*[nframes][n+2]*/, int64_t nframes, int n, int width,
                         uint32_t peak_rgb, uint32_t *pix_dev /*[nframes][width]*/, void *stream);

/* IQ recordings -> the stream-major device layout raw[S][stride] of the batch calls.
 * JavaAudio.openFile (JavaAudio.java:369-395): a file AudioSystem reads as PCM_SIGNED 16-bit little-endian with
 * the configured channel count and rate (compareFormat :397-406); RIFF/WAVE is parsed here, other containers and
 * the AudioSystem format conversion are not.  A file without a RIFF header is taken as the headerless dump
 * recorder.receive (recorder.java:66-74) / FCD.main (FCD.java:286-303) write.                                  */
#define JSDR_REC_RAW 0
#define JSDR_REC_WAV 1
typedef struct jsdr_recording_info {
    int format;          /* JSDR_REC_RAW / JSDR_REC_WAV */
    int encoding;        /* WAVE format tag: 1 = PCM */
    int channels, rate, bits;
    int64_t frames;      /* sample frames (one per IQ pair) in the file */
    int64_t data_offset; /* byte offset of the first sample */
} jsdr_recording_info;
int jsdr_recording_probe(const char *path, int raw_channels, jsdr_recording_info *info);
/* loads frames [first_frame, first_frame+nframes) of every file to raw_dev + s*stream_stride_i16 as int16 (I,Q)
 * pairs; a mono file (channels = 1, audio-mode-I) becomes (I,0) pairs -- equal to JavaAudio.java:286-288 as long
 * as the Q correction passed to the kernels is 0; a short file is zero-filled and frames_loaded[s] (optional)
 * says how many frames were real.  Format mismatch -> JSDR_ERR "Incompatible audio format" (:388).             */
int jsdr_recordings_load(const char *const *paths, int nstreams, int channels, int rate, int64_t first_frame,
                         int64_t nframes, int16_t *raw_dev, int64_t stream_stride_i16, int64_t *frames_loaded,
                         void *stream);

/* ------------------------------------------------------------------ one process, several GPUs
 * The reference hosts all its demodulators in one JVM (jsdr.java:479-483: `new FUNcubeBPSKDemod(i, ...)` in a loop) and one
 * audio thread feeds them (JavaAudio.java:298-304).  A group is that across `ndev` GPUs: `total_streams` lock-step
 * demodulators split into contiguous equal shards (device index g owns streams [g S, (g+1) S), S = total_streams / ndev),
 * ONE host thread per device that owns the device's jsdr_bpsk (and, with JSDR_GROUP_WITH_PSD, jsdr_fft) handle and its
 * HIP streams, and after every batch call ONE ncclAllGather per device (RCCL over xGMI, librccl.so loaded at the first
 * create) of the fixed-size per-stream result slots (jsdr_bpsk_pack_slots' layout), on a gather stream of its own beside
 * the next call's kernels.  There is no other exchange: streams are independent.
 *   devices: ndev device ordinals, or NULL for 0 .. ndev-1.
 *   JSDR_GROUP_GATHER_COPY: gather with device-to-device copies instead of RCCL (a host without librccl; or several
 *     group members on ONE device, which RCCL refuses -- the rehearsal a one-GPU box can run).
 * jsdr_group_batch_i16: raw_dev[g] = device g's [S][stream_stride] int16 IQ (on THAT device); psd_dev[g] (optional,
 * WITH_PSD) = float[S * nsamples / n][n + 2] on that device.  Returns when every device thread has enqueued its work;
 * a failure on any device abandons the step's gather on ALL of them (nobody waits in a collective for a rank that is
 * gone) and is reported here.  jsdr_group_sync waits for kernels, tails and gathers of every device.
 * jsdr_group_gathered: device g's copy of ALL total_streams slots, ordered by global stream id.                  */
typedef struct jsdr_group jsdr_group;
#define JSDR_GROUP_GATHER_COPY 1
#define JSDR_GROUP_WITH_PSD 2
int jsdr_group_create(jsdr_group **g, int ndev, const int *devices, int rate, int nsamples_per_frame, int tuning_hz,
                      int do_fft, int do_up, int total_streams, int64_t max_batch_samples, int flags);
int jsdr_group_destroy(jsdr_group *g);
int jsdr_group_info(jsdr_group *g, int *ndev, int *streams_per_device, int64_t *slot_bytes, int *rccl_version);
int jsdr_group_device(jsdr_group *g, int index, int *device, jsdr_bpsk **dem, jsdr_fft **fft);
int jsdr_group_batch_i16(jsdr_group *g, const int16_t *const *raw_dev, int64_t stream_stride_i16, int64_t nsamples,
                         int ic, int qc, float *const *psd_dev);
int jsdr_group_sync(jsdr_group *g);
int jsdr_group_gathered(jsdr_group *g, int index, const uint8_t **slots_dev, int64_t *bytes);
int jsdr_group_read_slot(jsdr_group *g, int index, int stream, uint8_t *slot_host);

#ifdef __cplusplus
}
#endif
#endif /* JSDR_HIP_H */
