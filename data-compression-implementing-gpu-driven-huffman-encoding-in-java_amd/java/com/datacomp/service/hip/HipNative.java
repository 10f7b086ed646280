package com.datacomp.service.hip;

import java.nio.ByteBuffer;

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
    public static final int DCZ_E_INVALID = -1;
    public static final int DCZ_E_BADSTREAM = -5;
    /** dcz_decompress_host: decoded, but the per-chunk digests were not computed on the device (ragged strides). */
    public static final int DCZ_NO_DIGESTS = 1;

    /** dcz_device_count */
    public static native int deviceCount();

    /** dcz_ctx_create / dcz_ctx_destroy: the handle is the dcz_ctx pointer. */
    public static native long ctxCreate(int device);
    public static native void ctxDestroy(long ctx);
    /** dcz_ctx_reserve: device workspace for batches of n bytes in chunks of blockBytes. */
    public static native int ctxReserve(long ctx, long n, long blockBytes);
    public static native String strerror(int status);

    /** dcz_host_register / dcz_host_unregister on a direct buffer (page-locks it for full PCIe rate). */
    public static native int hostRegister(ByteBuffer direct);
    public static native int hostUnregister(ByteBuffer direct);

    /** dcz_histogram(ctx, data, offset, length, hist[256]) */
    public static native int histogram(long ctx, byte[] data, int offset, int length, long[] hist256);

    /** dcz_encode_block: returns compressed size (>= 0) or a negative status; fills lengths256. */
    public static native int encodeBlock(long ctx, byte[] data, int length, int[] lengths256, byte[] out);

    /** dcz_decode_block: returns DCZ_OK or a negative status; errPos[0] = symbol index on DCZ_E_BADSTREAM. */
    public static native int decodeBlock(long ctx, byte[] comp, int compSize, int[] lengths256, byte[] out,
                                         int outSize, long[] errPos);

    /**
     * dcz_compress_host: one batch of K = ceil(n / blockBytes) chunks from a direct buffer.  Returns the payload bytes
     * written to {@code out} (>= 0) or a negative status.  compSize/compOff/status have K entries, lens K*256 code
     * lengths, sha (nullable) K*32 SHA-256 bytes of the original chunks.
     */
    public static native long compressBlocks(long ctx, ByteBuffer in, long n, int blockBytes, ByteBuffer out,
                                             int[] compSize, long[] compOff, byte[] lens, int[] status, byte[] sha);

    /**
     * dcz_decompress_host: K chunks whose payloads start at compOff[k] inside {@code comp}; chunk k is written at
     * k * outStride of {@code out}.  Returns DCZ_OK, DCZ_NO_DIGESTS or a negative status; per-chunk status / errPos
     * are always filled; sha (nullable) receives the SHA-256 of the decoded chunks.
     */
    public static native int decompressBlocks(long ctx, ByteBuffer comp, long compBytes, long[] compOff, int[] compSize,
                                              int[] origSize, byte[] lens, int K, long outStride, ByteBuffer out,
                                              int[] status, long[] errPos, byte[] sha);
}
