// HipFUNcubeBPSKDemod.java -- FUNcubeBPSKDemod's receive chain (FUNcubeBPSKDemod.java:357-595) and FECDecoder on the
// MI355X.  Constructor as jsdr.java:479-483 builds the reference class; the same three configuration keys
// (:102-104,195-200); the same publications (:377-378,455-456); the painted fields (:220-228,331-337) through getters.
package com.ashbysoft.java_sdr;

public class HipFUNcubeBPSKDemod implements IAudioHandler, IRawHandler, IPublishListener {
    private static final String CFG_TUNING = "bpsk-tuning";
    private static final String CFG_DOFFT = "bpsk-dofft";
    private static final String CFG_UPPER = "bpsk-upper";

    private final String name;
    private final IConfig config;
    private final IPublish publish;
    private final ILogger logger;
    private final boolean rawPath;
    private IAudio audio;
    private long handle;
    private int tuning;
    private boolean doFFT, doUp;
    private final int[] counters = new int[10];
    private final byte[] decoded = new byte[256];
    private final double[] state = new double[18];

    public HipFUNcubeBPSKDemod(int idx, IConfig cfg, IPublish pub, ILogger log, IUIHost hst, IAudio aud) {
        this(idx, cfg, pub, log, hst, aud, false);
    }

    public HipFUNcubeBPSKDemod(int idx, IConfig cfg, IPublish pub, ILogger log, IUIHost hst, IAudio aud, boolean rawPath) {
        this.name = "FUNcube" + idx;
        this.config = cfg;
        this.publish = pub;
        this.logger = log;
        this.rawPath = rawPath;
        setup(aud);
        pub.listen(this);
    }

    public void notify(String key, Object val) {
        if ("audio-change".equals(key) && val instanceof IAudio)
            setup((IAudio) val);
    }

    /** the menu actions of :177-190 that change the demodulator: a new tuning or mode restarts it, as setup() does */
    public synchronized void retune(int newTuning, boolean fft, boolean upper) {
        config.setIntConfig(name + "-" + CFG_TUNING, newTuning);
        config.setIntConfig(name + "-" + CFG_DOFFT, fft ? 1 : 0);
        config.setIntConfig(name + "-" + CFG_UPPER, upper ? 1 : 0);
        setup(audio);
    }

    private synchronized void setup(IAudio aud) {
        if (audio != null) {
            audio.remHandler(this);
            audio.remRawHandler(this);
        }
        audio = aud;
        AudioDescriptor ad = aud.getAudioDescriptor();
        tuning = config.getIntConfig(name + "-" + CFG_TUNING, 12000);
        doFFT = 0 != config.getIntConfig(name + "-" + CFG_DOFFT, 0);
        doUp = 0 != config.getIntConfig(name + "-" + CFG_UPPER, 0);
        if (handle != 0)
            HipNative.bpskDestroy(handle);
        handle = HipNative.bpskCreate(ad.rate, ad.blen / ad.size, tuning, doFFT ? 1 : 0, doUp ? 1 : 0);
        if (rawPath)
            audio.addRawHandler(this);
        else
            audio.addHandler(this);
    }

    public synchronized void receive(float[] buf) {
        HipNative.bpskReceive(handle, buf);
        afterFrame();
    }

    public synchronized void receive(byte[] raw) {
        HipNative.bpskReceiveRaw(handle, raw, audio.getICorrection(), audio.getQCorrection());
        afterFrame();
    }

    private void afterFrame() {
        HipNative.bpskCounters(handle, counters);
        if (doFFT) {  // :455-456
            publish.setPublish(name + "-bpsk-tune", -1);
            publish.setPublish(name + "-bpsk-centre", counters[9]);
        } else {      // :377-378
            publish.setPublish(name + "-bpsk-centre", -1);
            publish.setPublish(name + "-bpsk-tune", tuning);
        }
        if (counters[8] != 0)
            HipNative.bpskDecoded(handle, decoded);
    }

    /** cntRaw,cntDS,cntBit,cntFEC,cntDec,dmErrBits,dmCorr,dmMaxCorr,decodeOK,centreBin */
    public synchronized int[] getCounters() {
        return counters.clone();
    }

    public synchronized boolean isDecodeOK() {
        return counters[8] != 0;
    }

    public synchronized byte[] getDecoded() {
        return decoded.clone();
    }

    /** bits sliced during the last frame, +1/-1 */
    public synchronized byte[] getBits() {
        byte[] tmp = new byte[4096];
        int n = HipNative.bpskBits(handle, tmp);
        byte[] out = new byte[Math.min(n, tmp.length)];
        System.arraycopy(tmp, 0, out, 0, out.length);
        return out;
    }

    public synchronized double[] getState() {
        HipNative.bpskState(handle, state);
        return state.clone();
    }

    public synchronized void close() {
        if (audio != null) {
            audio.remHandler(this);
            audio.remRawHandler(this);
        }
        publish.unlisten(this);
        if (handle != 0)
            HipNative.bpskDestroy(handle);
        handle = 0;
        logger.statusMsg(name + ": closed");
    }
}
