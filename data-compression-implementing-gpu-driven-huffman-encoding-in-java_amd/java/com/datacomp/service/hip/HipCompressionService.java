package com.datacomp.service.hip;

import com.datacomp.core.ChunkMetadata;
import com.datacomp.core.CompressionHeader;
import com.datacomp.model.StageMetrics;
import com.datacomp.service.CompressionService;
import com.datacomp.service.cpu.CpuCompressionService;
import com.datacomp.util.ChecksumUtil;

import java.io.ByteArrayInputStream;
import java.io.ByteArrayOutputStream;
import java.io.DataInputStream;
import java.io.DataOutputStream;
import java.io.IOException;
import java.nio.ByteBuffer;
import java.nio.channels.FileChannel;
import java.nio.file.Files;
import java.nio.file.Path;
import java.nio.file.StandardOpenOption;
import java.security.MessageDigest;
import java.util.ArrayList;
import java.util.List;
import java.util.function.Consumer;

/**
 * CompressionService (service/CompressionService.java:11-66) on the HIP kernels: the slot
 * GpuCompressionService occupies today (service/gpu/GpuCompressionService.java:143-168, :834-862),
 * including its fallback-to-CPU contract (:145-149, :160-167).  The footer layout and the digest of
 * digests reuse the reference's own classes, so files are interchangeable with CpuCompressionService
 * in both directions.
 *
 * Chunks are processed in BATCHES through HipNative.compressBlocks / decompressBlocks (one JNI crossing,
 * one H2D / kernel / D2H sequence per batch on page-locked direct buffers; per-chunk SHA-256 on the device),
 * the way csrc/host/dcz_service.cpp does -- not one chunk per call like the TornadoVM path this replaces
 * (GpuCompressionService.java:389-460).  A batch is at least one chunk, so a file of few 16-32 MiB chunks
 * still gets whole-chip kernels (the decoder splits a chunk over many workgroups, csrc/k4_split.hip).
 *
 * NOT COMPILED in the authoring image (no JDK there).  See INTEGRATION.md.
 */
public class HipCompressionService implements CompressionService, AutoCloseable {
    private static final long DEFAULT_BATCH_BYTES = 256L << 20;

    private final int chunkSizeBytes;
    private final boolean fallbackOnError;
    private final CpuCompressionService cpuFallback;
    private final long ctx;          // dcz_ctx*, 0 = no usable device
    private final int chunksPerBatch;
    private ByteBuffer rawBuf;       // batch of original bytes   (direct, page-locked)
    private ByteBuffer payBuf;       // batch of payload bytes    (direct, page-locked)
    private StageMetrics lastStageMetrics = new StageMetrics();

    public HipCompressionService(int chunkSizeMB, boolean fallbackOnError) {
        this.chunkSizeBytes = chunkSizeMB * 1024 * 1024;
        this.fallbackOnError = fallbackOnError;
        this.cpuFallback = fallbackOnError ? new CpuCompressionService(chunkSizeMB) : null;
        this.chunksPerBatch = (int) Math.max(1, DEFAULT_BATCH_BYTES / chunkSizeBytes);
        long c = 0L;
        try {
            if (HipNative.deviceCount() > 0) c = HipNative.ctxCreate(0);
        } catch (Throwable t) { // UnsatisfiedLinkError included
            c = 0L;
        }
        this.ctx = c;
    }

    /** The GUI reaches this by instanceof (ui/CompressController.java:292-298), like CpuCompressionService.java:52. */
    public StageMetrics getLastStageMetrics() {
        return lastStageMetrics;
    }

    private void ensureBuffers() {
        long need = (long) chunksPerBatch * chunkSizeBytes + 64;
        if (rawBuf == null || rawBuf.capacity() < need) {
            releaseBuffers();
            rawBuf = ByteBuffer.allocateDirect((int) need);
            payBuf = ByteBuffer.allocateDirect((int) need);
            HipNative.hostRegister(rawBuf); // best effort: unregistered buffers still work, at pageable-copy speed
            HipNative.hostRegister(payBuf);
            HipNative.ctxReserve(ctx, need, chunkSizeBytes);
        }
    }

    private void releaseBuffers() {
        if (rawBuf != null) HipNative.hostUnregister(rawBuf);
        if (payBuf != null) HipNative.hostUnregister(payBuf);
        rawBuf = null;
        payBuf = null;
    }

    @Override
    public void compress(Path in, Path out, Consumer<Double> progress) throws IOException {
        if (!isAvailable()) {
            if (fallbackOnError) { cpuFallback.compress(in, out, progress); return; }
            throw new IOException("GPU compression failed", new IllegalStateException("HIP device not available"));
        }
        try {
            compressHip(in, out, progress);
        } catch (RuntimeException e) {
            if (fallbackOnError) { cpuFallback.compress(in, out, progress); return; }
            throw new IOException("GPU compression failed", e);
        }
    }

    private void compressHip(Path in, Path out, Consumer<Double> progress) throws IOException {
        lastStageMetrics = new StageMetrics();
        final StageMetrics m = lastStageMetrics;
        long size = Files.size(in);
        int numChunks = (int) ((size + chunkSizeBytes - 1) / chunkSizeBytes);
        MessageDigest global = ChecksumUtil.createSha256();
        List<ChunkMetadata> metas = new ArrayList<>();
        if (numChunks > 0) ensureBuffers();
        try (FileChannel src = FileChannel.open(in, StandardOpenOption.READ);
             FileChannel dst = FileChannel.open(out, StandardOpenOption.CREATE, StandardOpenOption.WRITE,
                     StandardOpenOption.TRUNCATE_EXISTING)) {
            long compressedOffset = 0;
            int done = 0;
            for (int c0 = 0; c0 < numChunks; c0 += chunksPerBatch) {
                int k = Math.min(chunksPerBatch, numChunks - c0);
                long off = (long) c0 * chunkSizeBytes;
                long n = Math.min((long) k * chunkSizeBytes, size - off);
                long t0 = System.nanoTime();
                rawBuf.clear();
                rawBuf.limit((int) n);
                while (rawBuf.hasRemaining()) {
                    if (src.read(rawBuf, off + rawBuf.position()) < 0) throw new IOException("Unexpected end of " + in);
                }
                m.recordStage(StageMetrics.Stage.FILE_IO, System.nanoTime() - t0, n);

                int[] compSize = new int[k];
                long[] compOff = new long[k];
                byte[] lens = new byte[k * 256];
                int[] status = new int[k];
                byte[] sha = new byte[k * 32];
                t0 = System.nanoTime();
                long total = HipNative.compressBlocks(ctx, rawBuf, n, chunkSizeBytes, payBuf, compSize, compOff, lens,
                        status, sha);
                if (total < 0) throw new RuntimeException("GPU compression failed: " + HipNative.strerror((int) total));
                // histogram, code build, encode and the per-chunk SHA-256 all ran inside this one device call
                m.recordStage(StageMetrics.Stage.ENCODING, System.nanoTime() - t0, n);
                m.recordStage(StageMetrics.Stage.FREQUENCY_ANALYSIS, 0, n);
                m.recordStage(StageMetrics.Stage.HUFFMAN_TREE_BUILD, 0, 0);
                m.recordStage(StageMetrics.Stage.CHECKSUM_COMPUTE, 0, n);

                t0 = System.nanoTime();
                payBuf.clear();
                payBuf.limit((int) total);
                while (payBuf.hasRemaining()) dst.write(payBuf);
                m.recordStage(StageMetrics.Stage.FILE_IO, System.nanoTime() - t0, total);

                for (int i = 0; i < k; i++) {
                    int idx = c0 + i;
                    long o = (long) idx * chunkSizeBytes;
                    int len = (int) Math.min(chunkSizeBytes, size - o);
                    byte[] digest = java.util.Arrays.copyOfRange(sha, i * 32, i * 32 + 32);
                    int[] codeLengths = new int[256];
                    for (int s = 0; s < 256; s++) codeLengths[s] = lens[i * 256 + s] & 0xFF;
                    global.update(digest); // digest of digests, CpuCompressionService.java:106-109
                    metas.add(new ChunkMetadata(idx, o, len, compressedOffset + compOff[i], compSize[i], digest, codeLengths));
                    done++;
                    if (progress != null) progress.accept((double) done / numChunks);
                }
                compressedOffset += total;
            }
            long t0 = System.nanoTime();
            CompressionHeader header = new CompressionHeader(in.getFileName().toString(), size,
                    Files.getLastModifiedTime(in).toMillis(), global.digest(), chunkSizeBytes);
            for (ChunkMetadata c : metas) header.addChunk(c);
            long footerStart = dst.position();
            ByteArrayOutputStream bos = new ByteArrayOutputStream();
            DataOutputStream dos = new DataOutputStream(bos);
            header.writeTo(dos);
            dos.writeLong(footerStart); // CpuCompressionService.java:174
            dos.flush();
            ByteBuffer fb = ByteBuffer.wrap(bos.toByteArray());
            while (fb.hasRemaining()) dst.write(fb);
            m.recordStage(StageMetrics.Stage.HEADER_WRITE, System.nanoTime() - t0, fb.capacity());
        }
    }

    /** Header of a container: the old header-first probe, then the footer pointer (CpuCompressionService.java:338-393). */
    private static final class Located {
        CompressionHeader header;
        long dataStart;
    }

    private static Located locateHeader(FileChannel src) throws IOException {
        Located r = new Located();
        long total = src.size();
        try { // header-first ("old") format: parse the first <= 4096 bytes
            int n = (int) Math.min(4096, total);
            ByteBuffer hb = ByteBuffer.allocate(n);
            while (hb.hasRemaining() && src.read(hb, hb.position()) >= 0) { }
            r.header = CompressionHeader.readFrom(new DataInputStream(new ByteArrayInputStream(hb.array(), 0, hb.position())));
            long sum = 0;
            for (ChunkMetadata c : r.header.getChunks()) sum += c.getCompressedSize();
            if (sum > total) throw new IOException("Compressed sizes exceed the file");
            r.dataStart = total - sum;
            return r;
        } catch (Exception e) {
            r.header = null;
        }
        if (total < 8) throw new IOException("Invalid footer position: file too small");
        ByteBuffer pb = ByteBuffer.allocate(8);
        while (pb.hasRemaining() && src.read(pb, total - 8 + pb.position()) >= 0) { }
        pb.flip();
        long footerStart = pb.getLong();
        if (footerStart < 0 || footerStart >= total - 8) throw new IOException("Invalid footer position: " + footerStart);
        ByteBuffer fb = ByteBuffer.allocate((int) (total - footerStart - 8));
        while (fb.hasRemaining() && src.read(fb, footerStart + fb.position()) >= 0) { }
        r.header = CompressionHeader.readFrom(new DataInputStream(new ByteArrayInputStream(fb.array())));
        r.dataStart = 0;
        return r;
    }

    @Override
    public void decompress(Path in, Path out, Consumer<Double> progress) throws IOException {
        if (!isAvailable()) {
            if (fallbackOnError) { cpuFallback.decompress(in, out, progress); return; }
            throw new IOException("GPU decompression failed", new IllegalStateException("HIP device not available"));
        }
        try (FileChannel dst = FileChannel.open(out, StandardOpenOption.CREATE, StandardOpenOption.WRITE,
                StandardOpenOption.TRUNCATE_EXISTING)) {
            decodeAll(in, dst, progress);
        }
    }

    /** Decodes and verifies every chunk; writes to {@code dst} unless it is null.  Returns the header. */
    private CompressionHeader decodeAll(Path in, FileChannel dst, Consumer<Double> progress) throws IOException {
        lastStageMetrics = new StageMetrics();
        final StageMetrics m = lastStageMetrics;
        try (FileChannel src = FileChannel.open(in, StandardOpenOption.READ)) {
            long t0 = System.nanoTime();
            Located loc = locateHeader(src);
            m.recordStage(StageMetrics.Stage.FILE_IO, System.nanoTime() - t0, 0);
            CompressionHeader header = loc.header;
            List<ChunkMetadata> chunks = header.getChunks();
            int numChunks = chunks.size();
            final int stride = header.getChunkSizeBytes();
            if (numChunks > 0) {
                if (stride <= 0) throw new IOException("Chunk decompression failed: bad chunk size in the header");
                ensureBuffers();
            }
            int perBatch = (int) Math.max(1, Math.min(chunksPerBatch, (long) rawCapacity() / Math.max(stride, 1)));
            int done = 0;
            for (int c0 = 0; c0 < numChunks; c0 += perBatch) {
                int k = Math.min(perBatch, numChunks - c0);
                long spanStart = chunks.get(c0).getCompressedOffset();
                long spanEnd = spanStart;
                long[] compOff = new long[k];
                int[] compSize = new int[k];
                int[] origSize = new int[k];
                byte[] lens = new byte[k * 256];
                for (int i = 0; i < k; i++) {
                    ChunkMetadata c = chunks.get(c0 + i);
                    if (c.getOriginalSize() < 0 || c.getOriginalSize() > stride || c.getCompressedSize() < 0
                            || c.getCompressedOffset() < spanStart) {
                        throw new IOException("Chunk decompression failed: bad metadata in chunk " + c.getChunkIndex());
                    }
                    compOff[i] = c.getCompressedOffset() - spanStart;
                    compSize[i] = c.getCompressedSize();
                    origSize[i] = c.getOriginalSize();
                    int[] cl = c.getCodeLengths();
                    for (int s = 0; s < 256; s++) {
                        if (cl[s] < 0 || cl[s] > 32) throw new IOException("Chunk decompression failed: bad code length table");
                        lens[i * 256 + s] = (byte) cl[s];
                    }
                    spanEnd = Math.max(spanEnd, c.getCompressedOffset() + c.getCompressedSize());
                }
                long span = spanEnd - spanStart;
                if (span + 16 > payBuf.capacity()) throw new IOException("Chunk decompression failed: payload larger than its chunks");
                t0 = System.nanoTime();
                payBuf.clear();
                payBuf.limit((int) span);
                while (payBuf.hasRemaining()) {
                    if (src.read(payBuf, loc.dataStart + spanStart + payBuf.position()) < 0)
                        throw new IOException("Chunk decompression failed: truncated payload");
                }
                m.recordStage(StageMetrics.Stage.FILE_IO, System.nanoTime() - t0, span);

                int[] status = new int[k];
                long[] errPos = new long[k];
                byte[] sha = new byte[k * 32];
                t0 = System.nanoTime();
                int st = HipNative.decompressBlocks(ctx, payBuf, span, compOff, compSize, origSize, lens, k, stride, rawBuf,
                        status, errPos, sha);
                long bytes = 0;
                for (int i = 0; i < k; i++) bytes += origSize[i];
                m.recordStage(StageMetrics.Stage.DECODING, System.nanoTime() - t0, bytes);
                for (int i = 0; i < k; i++) {
                    if (status[i] == HipNative.DCZ_E_BADSTREAM) { // CpuCompressionService.java:469-471
                        throw new IOException("Chunk decompression failed",
                                new RuntimeException("Huffman decode error at position " + errPos[i]));
                    }
                    if (status[i] != HipNative.DCZ_OK)
                        throw new IOException("Chunk decompression failed", new RuntimeException(HipNative.strerror(status[i])));
                }
                if (st < 0) throw new IOException("Chunk decompression failed", new RuntimeException(HipNative.strerror(st)));

                t0 = System.nanoTime();
                for (int i = 0; i < k; i++) { // CpuCompressionService.java:536-550
                    ChunkMetadata c = chunks.get(c0 + i);
                    byte[] actual;
                    if (st == HipNative.DCZ_NO_DIGESTS) {
                        byte[] dec = new byte[origSize[i]];
                        rawBuf.position(i * stride);
                        rawBuf.get(dec);
                        actual = ChecksumUtil.computeSha256(dec);
                    } else {
                        actual = java.util.Arrays.copyOfRange(sha, i * 32, i * 32 + 32);
                    }
                    if (!MessageDigest.isEqual(actual, c.getSha256Checksum())) {
                        throw new IOException("Checksum mismatch in chunk " + c.getChunkIndex()
                                + ":\n  Expected: " + ChecksumUtil.toHexString(c.getSha256Checksum())
                                + "\n  Actual:   " + ChecksumUtil.toHexString(actual));
                    }
                }
                m.recordStage(StageMetrics.Stage.CHECKSUM_VERIFY, System.nanoTime() - t0, bytes);

                if (dst != null) {
                    t0 = System.nanoTime();
                    for (int i = 0; i < k; i++) {
                        rawBuf.clear();
                        rawBuf.position(i * stride);
                        rawBuf.limit(i * stride + origSize[i]);
                        while (rawBuf.hasRemaining()) dst.write(rawBuf);
                    }
                    m.recordStage(StageMetrics.Stage.FILE_IO, System.nanoTime() - t0, bytes);
                }
                done += k;
                if (progress != null) progress.accept((double) done / numChunks);
            }
            return header;
        }
    }

    private int rawCapacity() {
        return rawBuf == null ? 0 : rawBuf.capacity() - 64;
    }

    @Override
    public void resumeCompression(Path in, Path out, int lastCompletedChunk, Consumer<Double> progress) {
        throw new UnsupportedOperationException("Resume not yet implemented"); // CpuCompressionService.java:636-641
    }

    /**
     * A real check (every chunk decoded and hashed, digest of digests compared); the reference only scans the last
     * 64 KiB for a header and never hashes (CpuCompressionService.java:652-694; SURVEY.md appendix D).
     */
    @Override
    public boolean verifyIntegrity(Path compressed) throws IOException {
        if (!isAvailable()) return fallbackOnError && cpuFallback.verifyIntegrity(compressed);
        try {
            CompressionHeader header = decodeAll(compressed, null, null);
            MessageDigest g = ChecksumUtil.createSha256();
            for (ChunkMetadata c : header.getChunks()) g.update(c.getSha256Checksum());
            return MessageDigest.isEqual(g.digest(), header.getGlobalChecksum());
        } catch (IOException | RuntimeException e) {
            return false;
        }
    }

    @Override
    public String getServiceName() {
        return "HIP Compression (MI355X)";
    }

    @Override
    public boolean isAvailable() {
        return ctx != 0L;
    }

    @Override
    public void close() {
        releaseBuffers();
        if (ctx != 0L) HipNative.ctxDestroy(ctx);
        if (cpuFallback != null) cpuFallback.close();
    }
}
