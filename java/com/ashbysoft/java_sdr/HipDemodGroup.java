// HipDemodGroup.java -- many FUNcubeBPSKDemod instances over recorded IQ files, on every GPU of the machine.
//
// The reference hosts its demodulators in one JVM (jsdr.java:479-483: `for (...) new FUNcubeBPSKDemod(fc, ...)`), each fed
// by the one audio thread.  Decoding a library of recordings is the same thing at scale: `paths.length` lock-step
// demodulators, split into contiguous shards over the machine's GPUs (one native host thread per device), the recordings
// mapped straight into the devices' input buffers (JavaAudio.openFile's formats, :369-395; recorder.java's headerless
// dumps, :66-74), one call per block of samples, and after every call the result slots of ALL streams gathered to every
// device with one RCCL all-gather over xGMI (jsdr_group_*, include/jsdr_hip.h).  Not a Swing component: a batch tool.
package com.ashbysoft.java_sdr;

import java.nio.ByteBuffer;
import java.nio.ByteOrder;

public class HipDemodGroup implements AutoCloseable {
    public static final int GATHER_COPY = 1;

    /** one decoded frame of one stream: FECDecode's return value (-1 or the channel error count) and its 256 bytes */
    public static final class Frame {
        public final int stream, errors, bitIndex;
        public final byte[] data;

        Frame(int stream, int errors, int bitIndex, byte[] data) {
            this.stream = stream;
            this.errors = errors;
            this.bitIndex = bitIndex;
            this.data = data;
        }
    }

    private long handle;
    private final int streams;
    private final long block;
    private final int slotBytes, devices, rcclVersion, fecOffset, fecMax;

    /** rate / samples as an AudioDescriptor gives them (samples = blen / size); block = samples per stream and call */
    public HipDemodGroup(int devices, String[] paths, int rate, int samples, int tuning, boolean doFFT, boolean doUp, long block, int flags) {
        this.streams = paths.length;
        this.block = block;
        handle = HipNative.groupCreate(devices, rate, samples, tuning, doFFT ? 1 : 0, doUp ? 1 : 0, paths.length, block, flags);
        long[] info = new long[7];
        HipNative.groupInfo(handle, info);
        this.devices = (int) info[0];
        this.slotBytes = (int) info[2];
        this.rcclVersion = (int) info[3];
        this.fecOffset = (int) info[5];
        this.fecMax = (int) info[6];
    }

    public int getDevices() {
        return devices;
    }

    public int getRcclVersion() {
        return rcclVersion;
    }

    /** block number `k` of every recording through the demodulators; returns the frames FECDecode produced in this block */
    public java.util.List<Frame> decodeBlock(String[] paths, int channels, int rate, long k, int ic, int qc) {
        long frames = HipNative.groupLoadRecordings(handle, paths, channels, rate, k * block, block);
        java.util.ArrayList<Frame> out = new java.util.ArrayList<Frame>();
        if (frames <= 0)
            return out;
        HipNative.groupBatch(handle, block, ic, qc);
        HipNative.groupSync(handle);
        byte[] slot = new byte[slotBytes];
        for (int s = 0; s < streams; s++) {
            HipNative.groupReadSlot(handle, s, slot);
            ByteBuffer b = ByteBuffer.wrap(slot).order(ByteOrder.LITTLE_ENDIAN);
            int nfec = b.getInt(4);  // header: nbits, nfec, counters ... (jsdr_bpsk_pack_slots)
            for (int t = 0; t < nfec && t < fecMax; t++) {
                int o = fecOffset + 264 * t;
                byte[] data = new byte[256];
                System.arraycopy(slot, o + 8, data, 0, 256);
                out.add(new Frame(s, b.getInt(o), b.getInt(o + 4), data));
            }
        }
        return out;
    }

    public synchronized void close() {
        long old = handle;
        handle = 0;
        if (old != 0)
            HipNative.groupDestroy(old);
    }
}
