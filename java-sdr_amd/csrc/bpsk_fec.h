// bpsk_fec.h -- the hook between the demodulator (bpsk.hip) and the FEC decoder (fec.hip).
#pragma once
#include "common.h"

namespace jsdr {

// One wave per stream that has sync hits in this call.  For each hit (in bit order) build the soft block
// from the +1/-1/0 bit history (FUNcubeBPSKDemod.java:562-564), decode, keep the stream's decoded[] on
// success, log {rc, decoded[]} per call.
struct BpskFecArgs {
    const signed char *bitlog;  // [nstreams][bitlog_stride]; the window of bit b (0-based in this call) starts at b+1
    long long bitlog_stride;
    const int *trig_count;      // [nstreams]
    const int *trig_bits;       // [nstreams][max_trig], ascending
    int max_trig;
    unsigned char *decoded;     // [nstreams][256], persistent decoded[] (FUNcubeBPSKDemod.java:111)
    int *fec_rc;                // [nstreams][max_trig]
    unsigned char *fec_data;    // [nstreams][max_trig][256]
    int *last;                  // [nstreams][2]: dmErrBits, decodeOK (:565-568)
    int *cnt_dec;               // [nstreams] cntDec (:569)
    int nstreams;
    unsigned long long *dec_scratch;  // [nstreams][max_trig][fec_dec_scratch_words()] Viterbi decision words
    int *done;                  // [nstreams] blocks of the stream that have finished in this launch; zero before and after
    int fuse;                   // 1: the stream's last block runs stage 2 itself (one launch less: the 1-stream receive() form);
                                // 0: k_fec_fin follows -- the hand-over needs a DEVICE-scope release per block, which on this
                                // multi-XCD part writes the XCD's L2 back: 20 000 of them beside the PSD kernel cost the
                                // 8192-stream step 1.8 ms (measured, one session)
    // fuse only: byte copies the block that completes the stream's FEC work performs afterwards (the 1-stream receive()
    // packs its result snapshot this way: what used to be one more dependent launch)
    int ncopy;
    const unsigned char *csrc[8];
    unsigned char *cdst[8];
    int cbytes[8];
    // the batch form (k_vitq, lane-per-block Viterbi): set vit to select it; dec_scratch then holds
    // fec_vitq_scratch_words(nstreams, max_trig) words
    unsigned char *vit;         // [nstreams][max_trig][320] Viterbi output bytes, or null
    int *work_list;             // [nstreams * max_trig]
    int *work_count;            // [1]
};

int launch_fec_bpsk(const BpskFecArgs &a, hipStream_t st);
int fec_prepare();
int fec_dec_scratch_words();
long long fec_vitq_scratch_words(int nstreams, int max_trig);

}  // namespace jsdr
