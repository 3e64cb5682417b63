// HipFft.java -- the data path of fft.java (fft.java:63-77,190-228) on the MI355X: same constructor arguments as
// jsdr.java:476 passes, same registration with IAudio, same "fft-psd" publication (float[n+2], listeners clone:
// waterfall.java:33).  It IS a tab like the class it replaces: jsdr.java:476 does tabs.add("FFT", new fft(...)), and
// tabs.add takes a Swing component -- the reference's plugins extend IUIComponent (IUIComponent.java:5-7, fft.java:19), so
// does this one: hotKey as fft.java:230-231, and a paintComponent that draws the last PSD and the reported maximum.
package com.ashbysoft.java_sdr;

import java.awt.Color;
import java.awt.Graphics;

public class HipFft extends IUIComponent implements IAudioHandler, IRawHandler, IPublishListener {
    private final IPublish publish;
    private final ILogger logger;
    private final boolean rawPath;
    private IAudio audio;
    private long handle;
    private float[] psd;
    private final Object paintLock = new Object();
    private float[] shown = new float[0];  // the painter's copy of the last PSD (the published array is reused every frame)

    public HipFft(IConfig cfg, IPublish pub, ILogger log, IUIHost host, IAudio aud) {
        this(cfg, pub, log, host, aud, false);
    }

    /** rawPath: register as IRawHandler, so that the int16 -> float rule (JavaAudio.java:276-293) runs on the GPU too */
    public HipFft(IConfig cfg, IPublish pub, ILogger log, IUIHost host, IAudio aud, boolean rawPath) {
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

    private synchronized void setup(IAudio aud) {
        if (audio != null) {
            audio.remHandler(this);
            audio.remRawHandler(this);
        }
        audio = aud;
        AudioDescriptor ad = aud.getAudioDescriptor();
        int n = ad.blen / ad.size;
        long old = handle;
        handle = 0;  // a failed create below (exception) must not leave the freed pointer behind
        if (old != 0)
            HipNative.fftDestroy(old);
        handle = HipNative.fftCreate(n, ad.rate);
        psd = new float[n + 2];
        if (rawPath)
            audio.addRawHandler(this);
        else
            audio.addHandler(this);
        logger.statusMsg("fft: " + n + " bins on the GPU");
    }

    public synchronized void receive(float[] buf) {
        HipNative.fftReceive(handle, buf, psd);
        publish.setPublish("fft-psd", psd);
        afterFrame();
    }

    public synchronized void receive(byte[] raw) {
        HipNative.fftReceiveRaw(handle, raw, audio.getICorrection(), audio.getQCorrection(), psd);
        publish.setPublish("fft-psd", psd);
        afterFrame();
    }

    private void afterFrame() {
        synchronized (paintLock) {
            if (shown.length != psd.length)
                shown = new float[psd.length];
            System.arraycopy(psd, 0, shown, 0, psd.length);
        }
        repaint();
    }

    public void hotKey(char c) {
    }

    /** the last PSD, negative frequencies left of centre as fft.java draws them (:150-170), and the reported maximum
     *  (psd[n] = Hz, psd[n+1] = dB: fft.java:214-223).  Takes paintLock only, never the receive() monitor. */
    public void paintComponent(Graphics g) {
        if (!isVisible())
            return;
        float[] p;
        synchronized (paintLock) {
            p = shown.clone();
        }
        int w = getWidth(), h = getHeight();
        g.setColor(Color.BLACK);
        g.fillRect(0, 0, w, h);
        int n = p.length - 2;
        if (n <= 0 || w < 2)
            return;
        g.setColor(Color.GREEN);
        g.drawString("max: " + p[n + 1] + " dB @ " + (int) p[n] + " Hz  (" + n + " bins, GPU)", 10, 12);
        g.setColor(Color.YELLOW);
        int ly = 0;
        for (int x = 0; x < w; x++) {
            int k = (int) ((long) x * n / w);   // 0 .. n-1 left to right = -rate/2 .. +rate/2
            int bin = (k + n / 2) % n;          // FFT order: 0,+f .. ,-f
            float db = p[bin];
            int y = (int) (-db * h / 120f);     // 0 dB at the top, -120 dB at the bottom
            y = y < 0 ? 0 : (y >= h ? h - 1 : y);
            if (x > 0)
                g.drawLine(x - 1, ly, x, y);
            ly = y;
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
            HipNative.fftDestroy(old);
    }
}
