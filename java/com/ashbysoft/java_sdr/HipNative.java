// HipNative.java -- the native methods of libjsdr_jni.so (jni/jsdr_jni.c), which forwards to libjsdr_hip.so
// (include/jsdr_hip.h).  One class holds them all; the plugin classes Hip* call nothing else.
//
// Not part of the reference: these sources are what a java-sdr maintainer adds next to fft.java / demod.java /
// FUNcubeBPSKDemod.java to put the MI355X path behind the unchanged IAudioHandler / IRawHandler surface
// (IAudioHandler.java:3-6, IRawHandler.java:3-6, jsdr.java:475-483).  Compiled where a JDK exists (see jni/Makefile);
// the build container of this repo has none, so tests/test_jni_sources.py checks them structurally instead.
package com.ashbysoft.java_sdr;

final class HipNative {
    static {
        System.loadLibrary("jsdr_jni");
    }

    private HipNative() {
    }

    // ---- fft.java:63-77,190-228
    static native long fftCreate(int n, int rate);
    static native void fftDestroy(long h);
    /** IAudioHandler form: buf = 2n floats (I,Q), psd = n+2 floats (fft.java:203-227) */
    static native void fftReceive(long h, float[] buf, float[] psd);
    /** IRawHandler form: raw = 4n bytes, little-endian int16 I,Q; ic/qc = IAudio.getICorrection()/getQCorrection() */
    static native void fftReceiveRaw(long h, byte[] raw, int ic, int qc, float[] psd);

    // ---- FUNcubeBPSKDemod.java:127-209,357-595
    static native long bpskCreate(int rate, int samples, int tuning, int doFFT, int doUp);
    static native void bpskDestroy(long h);
    static native void bpskReceive(long h, float[] buf);
    static native void bpskReceiveRaw(long h, byte[] raw, int ic, int qc);
    /** cntRaw,cntDS,cntBit,cntFEC,cntDec,dmErrBits,dmCorr,dmMaxCorr,decodeOK,centreBin (:110-115,:405,:498) */
    static native void bpskCounters(long h, int[] out10);
    /** decoded[] (:111): the last successfully decoded 256-byte frame */
    static native void bpskDecoded(long h, byte[] out256);
    /** bits sliced during the last receive(), +1/-1 each; returns their number (at most out.length are copied) */
    static native int bpskBits(long h, byte[] out);
    /** tuPhase,vcoPhase,dmBitPhase,dmEnergyOut,energy1,energy2,avePeakPower,aveCentreBin,dmEnergy[8],dmLastIQ[2] */
    static native void bpskState(long h, double[] out18);

    // ---- FECDecoder.java:703-852
    /** returns -1 or the channel error count; out256 is written only when both RS words decode (:780) */
    static native int fecDecode(byte[] raw5200, byte[] out256);

    // ---- demod.java:229-231,300-312,341-483
    static native long demodCreate(int rate, int n);
    static native void demodDestroy(long h);
    static native void demodConfigure(long h, int mode, boolean fir, boolean down, boolean agc);
    /** weights() for the band [flo, fhi]; the range check of filterMove stays in Java */
    static native void demodWeights(long h, int flo, int fhi);
    /** receive(): buf = 2n floats in, bbf = 4n bytes out (n little-endian (L,R) int16 pairs, :473-478) */
    static native void demodReceive(long h, float[] buf, byte[] bbf);
    /** the `max` and `avg` fields after the last frame (:465-467) */
    static native void demodFrameStats(long h, float[] out2);
}
