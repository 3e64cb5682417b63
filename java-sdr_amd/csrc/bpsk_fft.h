// bpsk_fft.h -- interface of the FFT-acquire front end (bpsk_fft.hip) used by the pipeline host code (bpsk.hip)
#pragma once
#include "common.h"
#include <vector>

namespace jsdr {

struct FftFrontState {
    double avePeakPower;   // :403
    double aveCentreBin;   // :404
    int centreBin;         // :405
    int pad;
    double hist[26];       // the 26 values fed to RxDownSample before the next frame (re of the inverse FFT)
};

struct FftFrontArgs {
    const int *raw;            // int16 pairs, [S][stride]
    const float2 *rawf;        // or float frames
    long long stride_pairs;
    int nframes;               // frames in this call
    int n, logn;               // samples per frame
    int ic, qc;
    int do_up;
    int decim;
    int first_out;             // input index (within the call) that completes output 0
    const double2 *vco_cs;     // [nds] (cos, sin) of the VCO phase table entry of every decimated sample
    const double2 *tw;         // [n-1] per-stage (cos, -sin), fft_twiddles_f64
    FftFrontState *st;         // [S]
    double2 *dm;               // [S][dm_stride]
    long long dm_stride;
    long long nds;
    const double *ds_taps;     // [27]
    long long *phase_clk;      // diagnostics: [8] cycle counts per phase of stream 0's workgroup, or null
};

int launch_front_fft(const FftFrontArgs &a, int nstreams, hipStream_t st);
// frames that are not a power of two (bpsk_fftm.hip): n = 2^a 3^b 5^c, 1024 < n <= 9600, n % 16 == 0
bool fftm_supported(int n);
void fftm_twiddles(std::vector<double2> &w, int n, int *np_out, int *rad, int *tw_off);
int launch_front_fftm(const FftFrontArgs &a, int np, const int *rad, const int *tw_off, int nstreams, hipStream_t st);
void fft_twiddles_f64(std::vector<double2> &w, int n);

}  // namespace jsdr
