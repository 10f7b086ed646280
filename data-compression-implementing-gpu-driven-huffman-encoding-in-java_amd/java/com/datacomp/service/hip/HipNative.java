package com.datacomp.service.hip;

/**
 * JNI binding of include/dcz.h (libdczhip.so) -- one static native per C entry point.
 * The shim is csrc/jni/dcz_jni.c (libdczjni.so).  Status codes are the DCZ_* constants of dcz.h.
 * NOT COMPILED in the authoring image (no JDK there); see INTEGRATION.md.
 */
public final class HipNative {
    static {
        System.loadLibrary("dczjni"); // links libdczhip.so
    }

    private HipNative() {}

    public static final int DCZ_OK = 0;
    public static final int DCZ_E_BADSTREAM = -5;

    /** dcz_device_count */
    public static native int deviceCount();

    /** dcz_ctx_create / dcz_ctx_destroy: the handle is the dcz_ctx pointer. */
    public static native long ctxCreate(int device);
    public static native void ctxDestroy(long ctx);
    public static native String strerror(int status);

    /** dcz_histogram(ctx, data, offset, length, hist[256]) */
    public static native int histogram(long ctx, byte[] data, int offset, int length, long[] hist256);

    /** dcz_encode_block: returns compressed size (>= 0) or a negative status; fills lengths256. */
    public static native int encodeBlock(long ctx, byte[] data, int length, int[] lengths256, byte[] out);

    /** dcz_decode_block: returns DCZ_OK or a negative status; errPos[0] = symbol index on DCZ_E_BADSTREAM. */
    public static native int decodeBlock(long ctx, byte[] comp, int compSize, int[] lengths256, byte[] out,
                                         int outSize, long[] errPos);
}
