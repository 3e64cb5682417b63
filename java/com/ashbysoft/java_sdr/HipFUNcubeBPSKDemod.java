// HipFUNcubeBPSKDemod.java -- FUNcubeBPSKDemod's receive chain (FUNcubeBPSKDemod.java:357-595) and FECDecoder on the
// MI355X.  Constructor as jsdr.java:479-483 builds the reference class; the same three configuration keys
// (:102-104,195-200); the same publications (:377-378,455-456); the painted fields (:220-228,331-337) through getters.
// It is a tab like the class it replaces (jsdr.java:479-483: tabs.add(nm, new FUNcubeBPSKDemod(...)) takes a Swing
// component): it extends IUIComponent as FUNcubeBPSKDemod.java:24 does, with hotKey and a paintComponent that draws the
// text statistics of :220-228 and the decoded frame of :331-337 from the lock-free snapshot.
package com.ashbysoft.java_sdr;

import java.awt.Color;
import java.awt.Graphics;

public class HipFUNcubeBPSKDemod extends IUIComponent implements IAudioHandler, IRawHandler, IPublishListener {
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
    // results of the last completed receive(), refreshed through HipNative.bpskSnapshot -- a lock-free host-side read
    // (no device call), so the painting thread never waits for the GPU and never blocks the audio thread
    private final Object resultLock = new Object();
    private final int[] counters = new int[11];   // ten counters + the number of bits sliced in that frame
    private final byte[] decoded = new byte[256];
    private final double[] state = new double[18];
    private final byte[] bits = new byte[512];

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
        long old = handle;
        handle = 0;  // bpskCreate throws when it fails (a frame FFT-acquire mode does not support, no device memory):
                     // the freed pointer must not stay behind for a later getter, close() or setup()
        if (old != 0)
            HipNative.bpskDestroy(old);
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
        int centre;
        synchronized (resultLock) {
            HipNative.bpskSnapshot(handle, counters, state, decoded, bits);
            centre = counters[9];
        }
        if (doFFT) {  // :455-456
            publish.setPublish(name + "-bpsk-tune", -1);
            publish.setPublish(name + "-bpsk-centre", centre);
        } else {      // :377-378
            publish.setPublish(name + "-bpsk-centre", -1);
            publish.setPublish(name + "-bpsk-tune", tuning);
        }
        repaint();
    }

    public void hotKey(char c) {
    }

    /** the statistics of FUNcubeBPSKDemod.java:220-228 and, after a successful decode, the frame's bytes as :331-337
     *  prints them.  From the getters: takes resultLock only, never the receive() monitor. */
    public void paintComponent(Graphics g) {
        if (!isVisible())
            return;
        int[] c = getCounters();
        double[] st = getState();
        byte[] dec = getDecoded();
        g.setColor(Color.BLACK);
        g.fillRect(0, 0, getWidth(), getHeight());
        g.setColor(Color.GREEN);
        g.drawString("decodeOK=" + (c[8] != 0) + " dmErrBits=" + c[5] + " raw=" + c[0] + " ds=" + c[1] + " bit=" + c[2]
                     + " fec=" + c[3] + " dec=" + c[4], 10, 20);
        if (doFFT)
            g.drawString("centreBin=" + c[9], getWidth() - 250, 20);
        else
            g.drawString("tuning=" + tuning, getWidth() - 250, 20);
        g.drawString("e1=" + st[4], getWidth() - 250, 36);
        g.drawString("e2=" + st[5], getWidth() - 250, 52);
        g.drawString("eO=" + st[3], getWidth() - 250, 68);
        g.drawString("com=" + c[7] + " co=" + c[6], getWidth() - 250, 84);
        if (c[8] != 0) {
            for (int n = 0; n < dec.length; n += 16)
                for (int l = 0; l < 16 && n + l < dec.length; l++)
                    g.drawString(String.format("%02x ", dec[n + l] & 0xff), 10 + (20 * l), 40 + n);
        }
    }

    // The getters below serve the painting thread (FUNcubeBPSKDemod.java:220-228,331-337).  They take only resultLock,
    // never this object's monitor, which the audio thread holds for the whole of receive(): a repaint never waits for
    // the GPU, and never delays the audio loop.

    /** cntRaw,cntDS,cntBit,cntFEC,cntDec,dmErrBits,dmCorr,dmMaxCorr,decodeOK,centreBin */
    public int[] getCounters() {
        synchronized (resultLock) {
            return java.util.Arrays.copyOf(counters, 10);
        }
    }

    public boolean isDecodeOK() {
        synchronized (resultLock) {
            return counters[8] != 0;
        }
    }

    /** decoded[] (:111): the last successfully decoded frame */
    public byte[] getDecoded() {
        synchronized (resultLock) {
            return decoded.clone();
        }
    }

    /** bits sliced during the last frame, +1/-1 -- the first 512 of them (a 2048-sample frame slices about 25, the
     *  default 9600-sample frame 120; getAllBits() has no such limit) */
    public byte[] getBits() {
        synchronized (resultLock) {
            return java.util.Arrays.copyOf(bits, Math.min(counters[10], bits.length));
        }
    }

    /** every bit of the last frame whatever its size; a device call, so it takes the receive() monitor: for the audio
     *  thread or a test, not for the painter */
    public synchronized byte[] getAllBits() {
        int n;
        synchronized (resultLock) {
            n = counters[10];
        }
        byte[] out = new byte[Math.max(n, 1)];
        int got = HipNative.bpskBits(handle, out);
        return java.util.Arrays.copyOf(out, Math.max(0, Math.min(got, out.length)));
    }

    /** tuPhase,vcoPhase,dmBitPhase,dmEnergyOut,energy1,energy2,avePeakPower,aveCentreBin,dmEnergy[8],dmLastIQ[2] */
    public double[] getState() {
        synchronized (resultLock) {
            return state.clone();
        }
    }

    public synchronized void close() {
        if (audio != null) {
            audio.remHandler(this);
            audio.remRawHandler(this);
        }
        publish.unlisten(this);
        long old = handle;
        handle = 0;
        if (old != 0)
            HipNative.bpskDestroy(old);
        logger.statusMsg(name + ": closed");
    }
}
