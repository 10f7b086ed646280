package com.datacomp.service.hip;

import com.datacomp.service.FrequencyService;

/**
 * FrequencyService (service/FrequencyService.java:6-27) on the HIP histogram kernel.
 * Drop-in for GpuFrequencyService (service/gpu/GpuFrequencyService.java:87-149).
 */
public class HipFrequencyService implements FrequencyService, AutoCloseable {
    private final long ctx;

    public HipFrequencyService() {
        this(0);
    }

    public HipFrequencyService(int device) {
        this.ctx = HipNative.deviceCount() > 0 ? HipNative.ctxCreate(device) : 0L;
    }

    @Override
    public long[] computeHistogram(byte[] data, int offset, int length) {
        if (ctx == 0L) throw new IllegalStateException("HIP device not available");
        if (offset < 0 || length < 0 || offset + length > data.length) {
            throw new ArrayIndexOutOfBoundsException("offset/length outside the array");
        }
        long[] hist = new long[256];
        int st = HipNative.histogram(ctx, data, offset, length, hist);
        if (st != HipNative.DCZ_OK) throw new RuntimeException("HIP histogram failed: " + HipNative.strerror(st));
        return hist;
    }

    @Override
    public String getServiceName() {
        return "HIP (MI355X gfx950)";
    }

    @Override
    public boolean isAvailable() {
        return ctx != 0L; // no probe kernel per call (GpuFrequencyService.java:255-283 launches one every time)
    }

    @Override
    public void close() {
        if (ctx != 0L) HipNative.ctxDestroy(ctx);
    }
}
