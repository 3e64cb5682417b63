// HipNative.java -- the native methods of libjsdr_jni.so (jni/jsdr_jni.c), which forwards to libjsdr_hip.so
// (include/jsdr_hip.h).  One class holds them all; the plugin classes Hip* call nothing else.
//
// Every receive checks the array lengths against the frame size its handle was created with and throws
// IllegalArgumentException on a mismatch (a buffer of the old geometry around an "audio-change"); a failed native call
// throws IllegalStateException with the library's message.  Handle 0 is "no handle" and is rejected everywhere.
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
    /** Results of the last COMPLETED receive(), callable from any thread (no lock, no device call): the Swing thread
     *  paints these while the audio thread is inside the next receive (:220-228,331-337).
     *  counters11 = cntRaw,cntDS,cntBit,cntFEC,cntDec,dmErrBits,dmCorr,dmMaxCorr,decodeOK,centreBin (:110-115,:405,:498)
     *  + the number of bits sliced in that frame; state18 = tuPhase,vcoPhase,dmBitPhase,dmEnergyOut,energy1,energy2,
     *  avePeakPower,aveCentreBin,dmEnergy[8],dmLastIQ[2]; decoded256 = decoded[] (:111); bits512 = the frame's first 512
     *  bits, +1/-1.  Returns the number of frames received so far; 0 = none yet, arrays untouched. */
    static native long bpskSnapshot(long h, int[] counters11, double[] state18, byte[] decoded256, byte[] bits512);
    /** every bit sliced during the last receive(), +1/-1 each; returns their number (at most out.length are copied).
     *  A device call: for the thread that calls receive(). */
    static native int bpskBits(long h, byte[] out);

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

    // ---- phase.java:75-80,93-116,123-128
    static native long phaseCreate(int n);
    static native void phaseDestroy(long h);
    /** receive(): buf = 2n floats; the frame is kept on the device and max|x| (:75-80) taken at once */
    static native void phaseReceive(long h, float[] buf);
    /** `max` of the last frame (:75-80); 0 before the first one (a zero-filled dpy) */
    static native float phaseMaxabs(long h);
    /** per-pixel-column means of I and Q for a panel bx pixels wide (:93-116): pix[c], avgi[c], avgq[c]; arrays of at
     *  least n+1 elements; returns the number of columns */
    static native int phaseColumns(long h, int bx, int[] pix, float[] avgi, float[] avgq);

    // ---- one JVM, several GPUs: jsdr.java:479-483 hosts all its demodulators in one process (jsdr_group_*, include/jsdr_hip.h)
    /** totalStreams lock-step FUNcubeBPSKDemod instances split into contiguous shards over ndev devices (0 .. ndev-1), one
     *  native host thread per device; every device gets an input buffer of totalStreams/ndev x maxBatch samples.
     *  flags: 1 = gather with device copies instead of RCCL, 2 = fft.receive's PSD beside the demodulators */
    static native long groupCreate(int ndev, int rate, int samples, int tuning, int doFFT, int doUp, int totalStreams,
                                   long maxBatch, int flags);
    static native void groupDestroy(long g);
    /** info7 = devices, streams per device, bytes per result slot, RCCL version (0: copies), then the slot's layout
     *  (jsdr_bpsk_slot_info): offset of the bits, offset of the FECDecode entries, how many entries a slot holds */
    static native void groupInfo(long g, long[] info7);
    /** recordings (WAV or headerless int16 IQ dumps, JavaAudio.java:369-395 / recorder.java:66-74), one per stream, frames
     *  [firstFrame, firstFrame + nframes) into the devices' input buffers; returns the frames really read, summed */
    static native long groupLoadRecordings(long g, String[] paths, int channels, int rate, long firstFrame, long nframes);
    /** FUNcubeBPSKDemod.receive over nsamples of every stream's loaded input, asynchronously on every device; the result
     *  slots of all streams are gathered to every device (ncclAllGather over xGMI) */
    static native void groupBatch(long g, long nsamples, int ic, int qc);
    static native void groupSync(long g);
    /** the result slot of global stream `stream` (counters, the call's bits, every FECDecode result: jsdr_bpsk_pack_slots'
     *  layout) from device 0's gathered copy; slot.length >= info7[2] */
    static native void groupReadSlot(long g, int stream, byte[] slot);
}
