package com.datacomp.service.hip;

import com.datacomp.core.ChunkMetadata;
import com.datacomp.core.CompressionHeader;
import com.datacomp.service.CompressionService;
import com.datacomp.service.cpu.CpuCompressionService;
import com.datacomp.util.ChecksumUtil;

import java.io.ByteArrayInputStream;
import java.io.ByteArrayOutputStream;
import java.io.DataInputStream;
import java.io.DataOutputStream;
import java.io.IOException;
import java.io.RandomAccessFile;
import java.nio.file.Files;
import java.nio.file.Path;
import java.security.MessageDigest;
import java.util.function.Consumer;

/**
 * CompressionService (service/CompressionService.java:11-66) on the HIP kernels: the slot
 * GpuCompressionService occupies today (service/gpu/GpuCompressionService.java:143-168, :834-862),
 * including its fallback-to-CPU contract (:145-149, :160-167).  Container I/O, SHA-256 and the
 * footer layout reuse the reference's own classes, so files are interchangeable with
 * CpuCompressionService in both directions.
 * NOT COMPILED in the authoring image (no JDK there); kept deliberately small.  See INTEGRATION.md.
 */
public class HipCompressionService implements CompressionService, AutoCloseable {
    private final int chunkSizeBytes;
    private final boolean fallbackOnError;
    private final CpuCompressionService cpuFallback;
    private final HipChunkCodec codec; // one dcz_ctx; this service streams chunks sequentially

    public HipCompressionService(int chunkSizeMB, boolean fallbackOnError) {
        this.chunkSizeBytes = chunkSizeMB * 1024 * 1024;
        this.fallbackOnError = fallbackOnError;
        this.cpuFallback = fallbackOnError ? new CpuCompressionService(chunkSizeMB) : null;
        HipChunkCodec c = null;
        try {
            if (HipNative.deviceCount() > 0) c = new HipChunkCodec(0);
        } catch (Throwable t) { // UnsatisfiedLinkError included
            c = null;
        }
        this.codec = c;
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
        long size = Files.size(in);
        int numChunks = (int) ((size + chunkSizeBytes - 1) / chunkSizeBytes);
        MessageDigest global = ChecksumUtil.createSha256();
        CompressionHeader header;
        java.util.List<ChunkMetadata> metas = new java.util.ArrayList<>();
        try (RandomAccessFile src = new RandomAccessFile(in.toFile(), "r");
             RandomAccessFile dst = new RandomAccessFile(out.toFile(), "rw")) {
            dst.setLength(0);
            byte[] buf = new byte[(int) Math.min(chunkSizeBytes, Math.max(size, 1))];
            long compressedOffset = 0;
            for (int k = 0; k < numChunks; k++) {
                long off = (long) k * chunkSizeBytes;
                int len = (int) Math.min(chunkSizeBytes, size - off);
                src.seek(off);
                src.readFully(buf, 0, len);
                byte[] sha = ChecksumUtil.computeSha256(buf, 0, len);
                global.update(sha);
                HipChunkCodec.Encoded e = codec.encode(buf, len);
                dst.write(e.payload);
                metas.add(new ChunkMetadata(k, off, len, compressedOffset, e.payload.length, sha, e.codeLengths));
                compressedOffset += e.payload.length;
                if (progress != null) progress.accept((double) (k + 1) / numChunks);
            }
            header = new CompressionHeader(in.getFileName().toString(), size,
                    Files.getLastModifiedTime(in).toMillis(), global.digest(), chunkSizeBytes);
            for (ChunkMetadata m : metas) header.addChunk(m);
            long footerStart = dst.getFilePointer();
            ByteArrayOutputStream bos = new ByteArrayOutputStream();
            header.writeTo(new DataOutputStream(bos));
            dst.write(bos.toByteArray());
            dst.writeLong(footerStart); // CpuCompressionService.java:174
        }
    }

    @Override
    public void decompress(Path in, Path out, Consumer<Double> progress) throws IOException {
        if (!isAvailable()) {
            if (fallbackOnError) { cpuFallback.decompress(in, out, progress); return; }
            throw new IOException("GPU decompression failed", new IllegalStateException("HIP device not available"));
        }
        try (RandomAccessFile src = new RandomAccessFile(in.toFile(), "r");
             RandomAccessFile dst = new RandomAccessFile(out.toFile(), "rw")) {
            dst.setLength(0);
            long total = src.length();
            src.seek(total - 8);
            long footerStart = src.readLong();
            if (footerStart < 0 || footerStart >= total - 8) throw new IOException("Invalid footer position: " + footerStart);
            byte[] footer = new byte[(int) (total - footerStart - 8)];
            src.seek(footerStart);
            src.readFully(footer);
            CompressionHeader header = CompressionHeader.readFrom(new DataInputStream(new ByteArrayInputStream(footer)));
            int n = header.getNumChunks(), done = 0;
            for (ChunkMetadata c : header.getChunks()) {
                byte[] comp = new byte[c.getCompressedSize()];
                src.seek(c.getCompressedOffset());
                src.readFully(comp);
                byte[] dec;
                try {
                    dec = codec.decode(comp, c.getCodeLengths(), c.getOriginalSize());
                } catch (RuntimeException e) {
                    throw new IOException("Chunk decompression failed", e); // CpuCompressionService.java:469-471
                }
                if (!MessageDigest.isEqual(ChecksumUtil.computeSha256(dec), c.getSha256Checksum())) {
                    throw new IOException("Checksum mismatch in chunk " + c.getChunkIndex());
                }
                dst.write(dec);
                if (progress != null) progress.accept((double) (++done) / n);
            }
        }
    }

    @Override
    public void resumeCompression(Path in, Path out, int lastCompletedChunk, Consumer<Double> progress) {
        throw new UnsupportedOperationException("Resume not yet implemented"); // CpuCompressionService.java:636-641
    }

    @Override
    public boolean verifyIntegrity(Path compressed) throws IOException {
        Path tmp = Files.createTempFile("dcz-verify", ".bin");
        try {
            decompress(compressed, tmp, null);
            return true;
        } catch (IOException e) {
            return false;
        } finally {
            Files.deleteIfExists(tmp);
        }
    }

    @Override
    public String getServiceName() {
        return "HIP Compression (MI355X)";
    }

    @Override
    public boolean isAvailable() {
        return codec != null;
    }

    @Override
    public void close() {
        if (codec != null) codec.close();
        if (cpuFallback != null) cpuFallback.close();
    }
}
