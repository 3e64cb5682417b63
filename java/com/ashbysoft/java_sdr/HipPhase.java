// HipPhase.java -- the data path of phase.java on the MI355X: receive() (phase.java:123-128) hands the frame to the
// device, where the maximum of |x| over its 2n floats (:75-80) is taken at once; the per-pixel-column means of I and Q
// (:93-116) are computed when the painter asks, for the panel width it has.  Constructor as jsdr.java:475 builds the
// reference class; the same registration with IAudio and the same reaction to "audio-change" (:30-43).  It is a tab like
// the class it replaces (jsdr.java:475: tabs.add("Phase", new phase(...)) takes a Swing component): it extends
// IUIComponent as phase.java:10 does; paintComponent draws the I and Q strips of phase.java:93-116 from the column means
// the device reduced, for the width the panel has.
package com.ashbysoft.java_sdr;

import java.awt.Color;
import java.awt.Graphics;

public class HipPhase extends IUIComponent implements IAudioHandler, IPublishListener {
    /** what one repaint needs from the last frame: max (:75-80) and, per pixel column, pix / avgi / avgq (:93-116) */
    public static final class Columns {
        public final float max;
        public final int[] pix;
        public final float[] avgi, avgq;

        Columns(float max, int[] pix, float[] avgi, float[] avgq) {
            this.max = max;
            this.pix = pix;
            this.avgi = avgi;
            this.avgq = avgq;
        }
    }

    private final IPublish publish;
    private final ILogger logger;
    private IAudio audio;
    private long handle;
    private int n;

    public HipPhase(IConfig cfg, IPublish pub, ILogger log, IUIHost hst, IAudio aud) {
        this.publish = pub;
        this.logger = log;
        setup(aud);
        pub.listen(this);
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
        n = ad.blen / ad.size;
        long old = handle;
        handle = 0;  // a failed create (exception) must not leave the freed pointer behind
        if (old != 0)
            HipNative.phaseDestroy(old);
        handle = HipNative.phaseCreate(n);
        logger.logMsg("phase: dpy.length=" + 2 * n);
        audio.addHandler(this);
    }

    public synchronized void receive(float[] buf) {
        HipNative.phaseReceive(handle, buf);
        repaint();  // phase.java:127
    }

    public void hotKey(char c) {
    }

    /** the I (red) and Q (blue) strips of phase.java:106-110 over the panel's width; the phase-space dots (:101-102) need
     *  every sample and stay with the reference class */
    public void paintComponent(Graphics g) {
        if (!isVisible())
            return;
        int w = getWidth(), h = getHeight();
        g.setColor(Color.BLACK);
        g.fillRect(0, 0, w, h);
        if (w < 2)
            return;
        Columns col = getColumns(w);
        float max = col.max > 0f ? col.max : 1f;
        float hiq = (float) (h / 4) / max;
        float lsti = 0, lstq = 0;
        for (int k = 0; k < col.pix.length; k++) {
            int pix = col.pix[k];
            g.setColor(Color.RED);
            g.drawLine(pix, h / 4 - (int) (lsti * hiq), pix, h / 4 - (int) (col.avgi[k] * hiq));
            g.setColor(Color.BLUE);
            g.drawLine(pix, h * 3 / 4 - (int) (lstq * hiq), pix, h * 3 / 4 - (int) (col.avgq[k] * hiq));
            lsti = col.avgi[k];
            lstq = col.avgq[k];
        }
        g.setColor(Color.GREEN);
        g.drawString("max: " + col.max, w / 2, 12);
    }

    /** max of |x| over the last frame (phase.java:75-80) */
    public synchronized float getMax() {
        return HipNative.phaseMaxabs(handle);
    }

    /** the reductions one paintComponent makes, for a panel whose I/Q strip is bx pixels wide (phase.java:52-54,81-116) */
    public synchronized Columns getColumns(int bx) {
        int[] pix = new int[n + 1];
        float[] avgi = new float[n + 1];
        float[] avgq = new float[n + 1];
        int ncol = HipNative.phaseColumns(handle, bx, pix, avgi, avgq);
        return new Columns(HipNative.phaseMaxabs(handle), java.util.Arrays.copyOf(pix, ncol),
                           java.util.Arrays.copyOf(avgi, ncol), java.util.Arrays.copyOf(avgq, ncol));
    }

    public synchronized void close() {
        if (audio != null)
            audio.remHandler(this);
        publish.unlisten(this);
        long old = handle;
        handle = 0;
        if (old != 0)
            HipNative.phaseDestroy(old);
    }
}
