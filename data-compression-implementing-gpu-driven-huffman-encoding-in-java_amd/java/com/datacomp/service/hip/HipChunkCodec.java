package com.datacomp.service.hip;

/**
 * The per-chunk seam of the orchestrators (CpuCompressionService.java:233-258 compress side,
 * :513-532 decompress side) on the HIP kernels.  One instance per worker thread: a dcz_ctx serves
 * one caller at a time (include/dcz.h), and the reference calls the seam from 1..8 pool threads.
 */
public final class HipChunkCodec implements AutoCloseable {
    private final long ctx;

    public HipChunkCodec(int device) {
        this.ctx = HipNative.ctxCreate(device);
        if (ctx == 0L) throw new IllegalStateException("HIP device not available");
    }

    /** Result of histogram -> buildCanonicalCodes -> encodeChunk for one chunk. */
    public static final class Encoded {
        public final byte[] payload;
        public final int[] codeLengths; // what ChunkMetadata stores (CompressionHeader.java:80-83)

        Encoded(byte[] payload, int[] codeLengths) {
            this.payload = payload;
            this.codeLengths = codeLengths;
        }
    }

    public Encoded encode(byte[] chunk, int length) {
        int[] lengths = new int[256];
        byte[] out = new byte[Math.max(length, 1)]; // a Huffman payload never exceeds the input
        int n = HipNative.encodeBlock(ctx, chunk, length, lengths, out);
        if (n < 0) throw new RuntimeException("GPU compression failed: " + HipNative.strerror(n));
        return new Encoded(java.util.Arrays.copyOf(out, n), lengths);
    }

    /** rebuildCodes + TableBasedHuffmanDecoder.decode; same exception text as TableBasedHuffmanDecoder.java:109-111. */
    public byte[] decode(byte[] compressed, int[] codeLengths, int originalSize) {
        byte[] out = new byte[originalSize];
        long[] errPos = new long[1];
        int st = HipNative.decodeBlock(ctx, compressed, compressed.length, codeLengths, out, originalSize, errPos);
        if (st == HipNative.DCZ_E_BADSTREAM) {
            throw new RuntimeException("Huffman decode error at position " + errPos[0]);
        }
        if (st != HipNative.DCZ_OK) throw new RuntimeException("HIP decode failed: " + HipNative.strerror(st));
        return out;
    }

    @Override
    public void close() {
        HipNative.ctxDestroy(ctx);
    }
}
