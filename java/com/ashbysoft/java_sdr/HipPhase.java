// HipPhase.java -- the data path of phase.java on the MI355X: receive() (phase.java:123-128) hands the frame to the
// device, where the maximum of |x| over its 2n floats (:75-80) is taken at once; the per-pixel-column means of I and Q
// (:93-116) are computed when the painter asks, for the panel width it has.  Constructor as jsdr.java:475 builds the
// reference class; the same registration with IAudio and the same reaction to "audio-change" (:30-43).  Drawing stays
// in Java: phase.java's paintComponent can draw from getMax() / getColumns().
package com.ashbysoft.java_sdr;

public class HipPhase implements IAudioHandler, IPublishListener {
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
