// HipDemod.java -- the arithmetic of demod.receive (demod.java:398-483) on the MI355X: same constructor arguments as
// jsdr.java:477 passes, same configuration keys (:27-32), same filterMove rule (:300-312) and publications; the
// demodulated frame lands in a little-endian byte array of (L,R) int16 pairs exactly as `bbf` holds it (:469-481) and is
// written to the SourceDataLine the caller opened (demod.java's own output thread and device dialog stay in demod.java).
// It is a tab like the class it replaces (jsdr.java:477: tabs.add("Demod", new demod(...)) takes a Swing component): it
// extends IUIComponent as demod.java does; paintComponent shows mode, band edges and the frame's max / avg (:465-467).
package com.ashbysoft.java_sdr;

import java.awt.Color;
import java.awt.Graphics;
import javax.sound.sampled.SourceDataLine;

public class HipDemod extends IUIComponent implements IAudioHandler, IPublishListener {
    private static final String CFG_DEMOD_FLOW = "demod-filter-low";
    private static final String CFG_DEMOD_FHGH = "demod-filter-high";
    private static final String CFG_DEMOD_MODE = "demod-mode";
    private static final String CFG_DEMOD_FIRE = "demod-fir-enable";
    private static final String CFG_DEMOD_AGCE = "demod-agc-enable";
    public static final int MODE_OFF = 0, MODE_RAW = 1, MODE_AM = 2, MODE_NFM = 3, MODE_WFM = 4;

    private final IConfig config;
    private final IPublish publish;
    private final ILogger logger;
    private IAudio audio;
    private long handle;
    private volatile int mode, flo, fhi;          // (volatile: the painter reads them without the receive() monitor)
    private volatile boolean dofir, dodwn, doagc;
    private byte[] bbf;
    private final float[] stats = new float[2];
    private final float[] shownStats = new float[2];  // max, avg of the last frame, for the painter (its own lock)
    private volatile SourceDataLine sdl;

    public HipDemod(IConfig cfg, IPublish pub, ILogger log, IUIHost hst, IAudio aud) {
        config = cfg;
        publish = pub;
        logger = log;
        publish.listen(this);
        mode = cfg.getIntConfig(CFG_DEMOD_MODE, MODE_OFF);
        dofir = cfg.getIntConfig(CFG_DEMOD_FIRE, 0) > 0;
        doagc = cfg.getIntConfig(CFG_DEMOD_AGCE, 0) > 0;
        flo = cfg.getIntConfig(CFG_DEMOD_FLOW, Integer.MIN_VALUE);
        fhi = cfg.getIntConfig(CFG_DEMOD_FHGH, Integer.MAX_VALUE);
        setup(aud);
    }

    public void notify(String key, Object val) {
        if ("audio-change".equals(key) && val instanceof IAudio)
            setup((IAudio) val);
    }

    private synchronized void setup(IAudio aud) {
        if (audio != null)
            audio.remHandler(this);
        audio = aud;
        AudioDescriptor ad = aud.getAudioDescriptor();
        int n = ad.blen / ad.size;
        long old = handle;
        handle = 0;  // a failed create below (exception) must not leave the freed pointer behind
        if (old != 0)
            HipNative.demodDestroy(old);
        handle = HipNative.demodCreate(ad.rate, n);
        bbf = new byte[4 * n];
        HipNative.demodConfigure(handle, mode, dofir, dodwn, doagc);
        filterMove(0, 0);
        audio.addHandler(this);
    }

    /** demod-off / -raw / -am / -nfm / -wfm and the three toggles (:184-203) */
    public synchronized void configure(int mode, boolean fir, boolean down, boolean agc) {
        this.mode = mode;
        this.dofir = fir;
        this.dodwn = down;
        this.doagc = agc;
        HipNative.demodConfigure(handle, mode, fir, down, agc);
        config.setIntConfig(CFG_DEMOD_MODE, mode);
        config.setIntConfig(CFG_DEMOD_FIRE, fir ? 1 : 0);
        config.setIntConfig(CFG_DEMOD_AGCE, agc ? 1 : 0);
    }

    /** demod.java:300-312: move the band edges if the result is a band inside (-rate/2, rate/2) */
    public synchronized void filterMove(int lo, int hi) {
        lo += flo;
        hi += fhi;
        int rate = audio.getAudioDescriptor().rate;
        if (lo < hi && lo > (-rate / 2) && hi < rate / 2) {
            flo = lo;
            fhi = hi;
            HipNative.demodWeights(handle, flo, fhi);
            publish.setPublish(CFG_DEMOD_FLOW, this.flo);
            publish.setPublish(CFG_DEMOD_FHGH, this.fhi);
            config.setIntConfig(CFG_DEMOD_FLOW, flo);
            config.setIntConfig(CFG_DEMOD_FHGH, fhi);
        }
    }

    public void setOutput(SourceDataLine line) {
        sdl = line;
    }

    public synchronized void receive(float[] buf) {
        if (MODE_OFF == mode)
            return;
        HipNative.demodReceive(handle, buf, bbf);
        if (isVisible()) {
            // max / avg feed the painter only (demod.java:465-467): a second device round trip per frame on the live audio
            // thread is paid only while the tab is showing
            HipNative.demodFrameStats(handle, stats);
            synchronized (shownStats) {
                shownStats[0] = stats[0];
                shownStats[1] = stats[1];
            }
        }
        SourceDataLine line = sdl;
        if (line != null)
            line.write(bbf, 0, bbf.length);
        repaint();
    }

    public void hotKey(char c) {
    }

    /** mode, band edges, and max / avg of the last frame (demod.java:465-467); never takes the receive() monitor */
    public void paintComponent(Graphics g) {
        if (!isVisible())
            return;
        float mx, av;
        synchronized (shownStats) {
            mx = shownStats[0];
            av = shownStats[1];
        }
        g.setColor(Color.BLACK);
        g.fillRect(0, 0, getWidth(), getHeight());
        g.setColor(Color.GREEN);
        String[] names = {"off", "raw", "am", "nfm", "wfm"};
        g.drawString("mode=" + names[mode < 0 || mode > 4 ? 0 : mode] + " fir=" + dofir + " agc=" + doagc + " band=" + flo + ".." + fhi + " Hz", 10, 20);
        g.drawString("max=" + mx + " avg=" + av, 10, 36);
    }

    /** the frame's audio as demod.java's bbf holds it */
    public synchronized byte[] getAudioFrame() {
        return bbf.clone();
    }

    /** max, avg after the last frame (:465-467) */
    public synchronized float[] getFrameStats() {
        HipNative.demodFrameStats(handle, stats);
        return stats.clone();
    }

    public synchronized void close() {
        if (audio != null)
            audio.remHandler(this);
        publish.unlisten(this);
        long old = handle;
        handle = 0;
        if (old != 0)
            HipNative.demodDestroy(old);
        logger.statusMsg("demod: closed");
    }
}
