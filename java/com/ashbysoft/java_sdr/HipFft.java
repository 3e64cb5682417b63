// HipFft.java -- the data path of fft.java (fft.java:63-77,190-228) on the MI355X: same constructor arguments as
// jsdr.java:476 passes, same registration with IAudio, same "fft-psd" publication (float[n+2], listeners clone:
// waterfall.java:33).  Painting stays where it is: fft.java's paintComponent can draw the published array.
package com.ashbysoft.java_sdr;

public class HipFft implements IAudioHandler, IRawHandler, IPublishListener {
    private final IPublish publish;
    private final ILogger logger;
    private final boolean rawPath;
    private IAudio audio;
    private long handle;
    private float[] psd;

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
    }

    public synchronized void receive(byte[] raw) {
        HipNative.fftReceiveRaw(handle, raw, audio.getICorrection(), audio.getQCorrection(), psd);
        publish.setPublish("fft-psd", psd);
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
