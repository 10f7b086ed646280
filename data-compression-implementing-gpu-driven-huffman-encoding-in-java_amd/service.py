"""Host-side mirror of the reference's service seam over the C ABI (include/dcz.h).

  HipFrequencyService   <-> com.datacomp.service.FrequencyService   (service/FrequencyService.java:6-27)
  HipCompressionService <-> com.datacomp.service.CompressionService (service/CompressionService.java:11-66)

Same method names (snake_case), argument meaning and error behaviour as the Java interfaces; the Java
classes that bind the same C entry points through JNI are in java/ (see INTEGRATION.md).  torch is
used for device buffers, streams and torch.distributed only.  Every hot stage runs in HIP kernels;
there is no CPU fallback here (the reference's fallback-to-CpuCompressionService lives one level up,
in ServiceFactory, service/ServiceFactory.java:21-48).
"""
import ctypes as C
import hashlib
import os
import time

import numpy as np

from . import container as fmt
from . import native as nv

# model/StageMetrics.java:11-20 stage names
STAGES = ("Frequency Analysis", "Huffman Tree Build", "Encoding", "Checksum Computation", "File I/O", "Header Write",
          "Decoding", "Checksum Verification")


class StageMetrics:
    """model/StageMetrics.java:45-49 recordStage accumulators (ns, count, bytes)."""

    def __init__(self):
        self.times = {s: 0 for s in STAGES}
        self.counts = {s: 0 for s in STAGES}
        self.bytes = {s: 0 for s in STAGES}

    def record(self, stage, ns, nbytes=0):
        self.times[stage] += int(ns)
        self.counts[stage] += 1
        self.bytes[stage] += int(nbytes)

    def summary(self):
        tot = sum(self.times.values()) or 1
        lines = ["Stage Performance Breakdown:"]
        for s in STAGES:
            if self.counts[s]:
                lines.append("%-25s: %8.2f ms (%5.1f%%) [%d runs]" % (s, self.times[s] / 1e6,
                                                                     100.0 * self.times[s] / tot, self.counts[s]))
        return "\n".join(lines)


class HuffmanDecodeError(RuntimeError):
    """RuntimeException("Huffman decode error at position i"), core/TableBasedHuffmanDecoder.java:109-111."""

    def __init__(self, position):
        super().__init__("Huffman decode error at position %d" % position)
        self.position = position


def _np_u8(data):
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.uint8)
    return np.frombuffer(bytes(data), dtype=np.uint8)


class HipFrequencyService:
    """FrequencyService on the HIP histogram kernel (K1)."""

    def __init__(self, device=0, ctx=None):
        self._own = ctx is None
        self.ctx = ctx if ctx is not None else nv.Context(device)

    def compute_histogram(self, data, offset=0, length=None):
        """computeHistogram(byte[] data, int offset, int length) -> long[256] (FrequencyService.java:16)."""
        d = _np_u8(data)
        if length is None:
            length = d.size - offset
        if offset < 0 or length < 0 or offset + length > d.size:
            raise IndexError("offset/length outside the array")  # ArrayIndexOutOfBounds in the reference loop
        hist = np.zeros(256, dtype=np.int64)
        self.ctx.check(nv.lib().dcz_histogram(self.ctx.handle, d.ctypes.data, offset, length, hist.ctypes.data))
        return hist

    def get_service_name(self):
        return "HIP (MI355X gfx950)"

    def is_available(self):
        return nv.lib().dcz_device_count() > 0

    def close(self):
        if self._own:
            self.ctx.close()


class DeviceBlocks:
    """Result of a device-resident compress: tensors named after the footer fields (CompressionHeader.java:71-84)."""

    def __init__(self, payload, comp_size, comp_off, code_lengths, status, total, n, block_bytes):
        self.payload, self.comp_size, self.comp_off = payload, comp_size, comp_off
        self.code_lengths, self.status, self.total = code_lengths, status, total
        self.n, self.block_bytes = n, block_bytes

    @property
    def num_chunks(self):
        return int(self.comp_size.numel())


class HipCompressionService:
    """CompressionService on the HIP pipeline (K1 histogram -> K2 code build -> K3 encode; K4 decode)."""

    def __init__(self, chunk_size_mb=16, device=0, batch_bytes=1 << 30):
        import torch  # device memory + streams only
        self.torch = torch
        if chunk_size_mb <= 0 or chunk_size_mb > 2047:
            raise ValueError("chunk size must be 1..2047 MB")  # int chunkSizeBytes, CpuCompressionService.java:38
        self.chunk_size_bytes = int(chunk_size_mb) * 1024 * 1024
        self.device = device
        self.ctx = nv.Context(device)
        self.batch_bytes = max(int(batch_bytes), self.chunk_size_bytes)
        self.last_stage_metrics = StageMetrics()
        # The library runs on the stream it is handed; a NULL handle means the context's private non-blocking stream,
        # which is NOT ordered with torch's streams.  The service wraps that stream for torch and orders it before and
        # after every call with the caller's current torch stream, so that tensors produced by torch ops (fills,
        # collectives) and by the kernels see each other in program order without host synchronisation.
        self._dev = torch.device("cuda", device)
        self.stream = torch.cuda.ExternalStream(self.ctx.stream_handle, device=self._dev)

    # ---- interface methods -------------------------------------------------------------------
    def get_service_name(self):
        return "HIP Compression (MI355X)"

    def is_available(self):
        return nv.lib().dcz_device_count() > 0

    def get_last_stage_metrics(self):
        return self.last_stage_metrics

    def close(self):
        self.ctx.close()

    def _enter(self, stream, dev):
        """-> (raw hipStream_t handle to launch on, caller's torch stream or None).  With stream=None the service
        stream first waits for everything the caller's current stream has queued."""
        if stream is not None:
            return stream, None
        cur = self.torch.cuda.current_stream(dev)
        if cur.cuda_stream != 0:
            # a stream of the caller's own: the kernels are queued on it like the caller's torch work, nothing to order
            # (two cross-stream events per call cost config 3's step 0.36 -> 0.28 ms)
            return cur.cuda_stream, None
        # torch's default stream has handle 0, which the C ABI reads as "the context's stream": order the two by events
        self.stream.wait_stream(cur)
        return self.stream.cuda_stream, cur

    def _leave(self, cur):
        if cur is not None:
            cur.wait_stream(self.stream)  # later torch work on the caller's stream sees the kernels' results

    def resume_compression(self, input_path, output_path, last_completed_chunk, progress_callback=None):
        # UnsupportedOperationException in both reference services (CpuCompressionService.java:636-641)
        raise NotImplementedError("Resume not yet implemented")

    # ---- device-resident hot path (what bench.py times) --------------------------------------
    def compress_device(self, t_in, block_bytes=None, out=None, stream=None):
        """t_in: 1-D uint8 tensor on this service's device.  Asynchronous; returns DeviceBlocks."""
        torch = self.torch
        bb = int(block_bytes or self.chunk_size_bytes)
        n = int(t_in.numel())
        K = (n + bb - 1) // bb
        dev = t_in.device
        if out is None:
            out = DeviceBlocks(torch.empty(max(n, 16), dtype=torch.uint8, device=dev),
                               torch.empty(max(K, 1), dtype=torch.int32, device=dev)[:K],
                               torch.empty(max(K, 1), dtype=torch.int64, device=dev)[:K],
                               torch.empty((max(K, 1), 256), dtype=torch.uint8, device=dev)[:K],
                               torch.empty(max(K, 1), dtype=torch.int32, device=dev)[:K],
                               torch.zeros(1, dtype=torch.int64, device=dev), n, bb)
        s, cur = self._enter(stream, dev)
        self.ctx.check(nv.lib().dcz_compress_blocks(
            self.ctx.handle, t_in.data_ptr(), n, bb, out.payload.data_ptr(), int(out.payload.numel()),
            out.comp_size.data_ptr(), out.comp_off.data_ptr(), out.code_lengths.data_ptr(), out.status.data_ptr(),
            out.total.data_ptr(), s))
        self._leave(cur)
        out.n, out.block_bytes = n, bb
        return out

    def decompress_device(self, payload, comp_off, comp_size, orig_size, code_lengths, out_stride, t_out=None,
                          status=None, errpos=None, stream=None):
        torch = self.torch
        K = int(comp_size.numel())
        dev = payload.device
        if t_out is None:
            t_out = torch.empty(max(K * out_stride, 16), dtype=torch.uint8, device=dev)
        if status is None:
            status = torch.zeros(max(K, 1), dtype=torch.int32, device=dev)
        if errpos is None:
            errpos = torch.zeros(max(K, 1), dtype=torch.int64, device=dev)
        s, cur = self._enter(stream, dev)  # after the allocations / fills above, which run on the caller's stream
        self.ctx.check(nv.lib().dcz_decompress_blocks(
            self.ctx.handle, payload.data_ptr(), int(payload.numel()), comp_off.data_ptr(), comp_size.data_ptr(),
            orig_size.data_ptr(), code_lengths.data_ptr(), K, int(out_stride), t_out.data_ptr(), status.data_ptr(),
            errpos.data_ptr(), s))
        self._leave(cur)
        return t_out, status, errpos

    def sha256_device(self, t, block_bytes, stream=None):
        """SHA-256 of every block of a device buffer (ChecksumUtil.computeSha256 per chunk) -> uint8 tensor [K, 32]."""
        torch = self.torch
        n = int(t.numel())
        bb = int(block_bytes)
        K = (n + bb - 1) // bb
        out = torch.empty((max(K, 1), 32), dtype=torch.uint8, device=t.device)
        s, cur = self._enter(stream, t.device)
        self.ctx.check(nv.lib().dcz_sha256_blocks(self.ctx.handle, t.data_ptr(), n, bb, out.data_ptr(), s))
        self._leave(cur)
        return out[:K]

    # one lane per chunk: below this many chunks in a batch the host's hashlib is faster
    SHA_GPU_MIN_CHUNKS = 512

    # ---- single chunk, host bytes (processChunk / decodeChunkParallel hot stages) ------------
    def encode_chunk(self, data):
        """-> (payload ndarray, code_lengths int32[256])."""
        d = _np_u8(data)
        out = np.zeros(max(d.size, 1), dtype=np.uint8)
        lens = np.zeros(256, dtype=np.int32)
        n_out = C.c_size_t()
        self.ctx.check(nv.lib().dcz_encode_block(self.ctx.handle, d.ctypes.data, d.size, lens.ctypes.data,
                                                 out.ctypes.data, out.size, C.byref(n_out)))
        return out[:n_out.value].copy(), lens

    def decode_chunk(self, comp, code_lengths, original_size):
        c = _np_u8(comp)
        lens = np.ascontiguousarray(code_lengths, dtype=np.int32)
        out = np.zeros(max(original_size, 1), dtype=np.uint8)
        ep = C.c_int64()
        st = nv.lib().dcz_decode_block(self.ctx.handle, c.ctypes.data, c.size, lens.ctypes.data, out.ctypes.data,
                                       original_size, C.byref(ep))
        if st == nv.DCZ_E_BADSTREAM:
            raise HuffmanDecodeError(ep.value)
        self.ctx.check(st)
        return out[:original_size]

    def build_codes(self, hist):
        h = np.ascontiguousarray(hist, dtype=np.int64)
        lens = np.zeros(256, dtype=np.int32)
        codes = np.zeros(256, dtype=np.uint32)
        self.ctx.check(nv.lib().dcz_build_codes(self.ctx.handle, h.ctypes.data, lens.ctypes.data, codes.ctypes.data))
        return lens, codes

    def codes_from_lengths(self, lens):
        ln = np.ascontiguousarray(lens, dtype=np.int32)
        codes = np.zeros(256, dtype=np.uint32)
        self.ctx.check(nv.lib().dcz_codes_from_lengths(self.ctx.handle, ln.ctypes.data, codes.ctypes.data))
        return codes

    # ---- file -> file (CompressionService.compress / decompress / verifyIntegrity) -----------
    def compress(self, input_path, output_path, progress_callback=None):
        """CompressionService.compress (CpuCompressionService.java:57-205 layout: payloads, footer, pointer)."""
        torch = self.torch
        self.last_stage_metrics = m = StageMetrics()
        size = os.path.getsize(input_path)
        cb = self.chunk_size_bytes
        num_chunks = (size + cb - 1) // cb
        header_chunks = []
        digests = []
        comp_offset = 0
        done = 0
        dev = torch.device("cuda", self.device)
        chunks_per_batch = max(1, self.batch_bytes // cb)
        with open(input_path, "rb") as fin, open(output_path, "wb") as fout:
            for c0 in range(0, num_chunks, chunks_per_batch):
                c1 = min(num_chunks, c0 + chunks_per_batch)
                t0 = time.perf_counter_ns()
                raw = fin.read((c1 - c0) * cb)
                m.record("File I/O", time.perf_counter_ns() - t0, len(raw))
                host = np.frombuffer(raw, dtype=np.uint8)
                t_in = torch.from_numpy(host.copy()).to(dev)
                t0 = time.perf_counter_ns()
                if c1 - c0 >= self.SHA_GPU_MIN_CHUNKS:  # K5 on the device-resident batch
                    dg = self.sha256_device(t_in, cb).cpu().numpy()
                    digests.extend(dg[i].tobytes() for i in range(c1 - c0))
                else:
                    for k in range(c0, c1):
                        digests.append(hashlib.sha256(raw[(k - c0) * cb:(k - c0 + 1) * cb]).digest())
                m.record("Checksum Computation", time.perf_counter_ns() - t0, len(raw))
                t0 = time.perf_counter_ns()
                blocks = self.compress_device(t_in, cb)
                torch.cuda.synchronize(dev)
                status = blocks.status.cpu().numpy()
                if (status != 0).any():
                    k = int(np.nonzero(status)[0][0])
                    raise IOError("GPU compression failed: chunk %d: %s" % (
                        c0 + k, nv.lib().dcz_strerror(int(status[k])).decode()))
                sizes = blocks.comp_size.cpu().numpy().astype(np.int64)
                lens = blocks.code_lengths.cpu().numpy()
                total = int(blocks.total.item())
                payload = blocks.payload[:total].cpu().numpy()
                m.record("Encoding", time.perf_counter_ns() - t0, len(raw))
                t0 = time.perf_counter_ns()
                fout.write(payload.tobytes())
                m.record("File I/O", time.perf_counter_ns() - t0, total)
                for k in range(c0, c1):
                    osz = min(cb, size - k * cb)
                    header_chunks.append(fmt.ChunkMetadata(k, k * cb, osz, comp_offset, int(sizes[k - c0]),
                                                           digests[k], lens[k - c0]))
                    comp_offset += int(sizes[k - c0])
                    done += 1
                    if progress_callback is not None:
                        progress_callback(done / num_chunks)
            t0 = time.perf_counter_ns()
            g = hashlib.sha256()
            for dg in digests:  # digest of digests, CpuCompressionService.java:106-109,126
                g.update(dg)
            header = fmt.CompressionHeader(os.path.basename(str(input_path)), size,
                                           int(os.path.getmtime(input_path) * 1000), g.digest(), cb)
            for ch in header_chunks:
                header.add_chunk(ch)
            footer_start = fout.tell()
            fout.write(header.write())
            fout.write(fmt.footer_pointer(footer_start))
            m.record("Header Write", time.perf_counter_ns() - t0, 0)

    def _decode_all(self, input_path, sink, progress_callback=None):
        torch = self.torch
        self.last_stage_metrics = m = StageMetrics()
        t0 = time.perf_counter_ns()
        with open(input_path, "rb") as f:
            data = f.read()
        m.record("File I/O", time.perf_counter_ns() - t0, len(data))
        header, data_start = fmt.locate_header(data)
        chunks = header.chunks
        num_chunks = len(chunks)
        # The metadata is untrusted: sizes that cannot belong to this file are rejected before they size any buffer
        # (the reference is protected by Java's int / byte[] semantics, CompressionHeader.java:71-84).
        for c in chunks:
            if (header.chunk_size_bytes <= 0 or not 0 <= c.original_size <= header.chunk_size_bytes
                    or not 0 <= c.compressed_size <= len(data) or not 0 <= c.compressed_offset <= len(data)):
                raise IOError("Chunk decompression failed: metadata of chunk %d does not fit the file" % c.chunk_index)
        dev = torch.device("cuda", self.device)
        per = max(1, self.batch_bytes // max(1, header.chunk_size_bytes))
        done = 0
        for c0 in range(0, num_chunks, per):
            batch = chunks[c0:c0 + per]
            t0 = time.perf_counter_ns()
            spans = [data[data_start + c.compressed_offset:data_start + c.compressed_offset + c.compressed_size]
                     for c in batch]
            for c, sp in zip(batch, spans):
                if len(sp) != c.compressed_size:
                    raise IOError("Chunk decompression failed: truncated payload in chunk %d" % c.chunk_index)
            sizes = np.array([c.compressed_size for c in batch], dtype=np.int64)  # (device columns are read as u32)
            offs = np.zeros(len(batch), dtype=np.int64)
            offs[1:] = np.cumsum(sizes)[:-1]
            blob = np.frombuffer(b"".join(spans) + b"\0" * 16, dtype=np.uint8)
            lens = np.array([c.code_lengths for c in batch], dtype=np.int64)
            if ((lens < 0) | (lens > 32)).any():
                raise IOError("Chunk decompression failed: bad code length table")
            stride = max(max(c.original_size for c in batch), 16)
            stride = (stride + 15) & ~15
            t_out, status, errpos = self.decompress_device(
                torch.from_numpy(blob.copy()).to(dev), torch.from_numpy(offs).to(dev),
                torch.from_numpy(sizes.astype(np.int32)).to(dev),
                torch.from_numpy(np.array([c.original_size for c in batch], dtype=np.int32)).to(dev),
                torch.from_numpy(lens.astype(np.uint8)).to(dev), stride)
            torch.cuda.synchronize(dev)
            st = status.cpu().numpy()[:len(batch)]
            if (st != 0).any():
                k = int(np.nonzero(st)[0][0])
                cause = (HuffmanDecodeError(int(errpos[k].item())) if st[k] == nv.DCZ_E_BADSTREAM
                         else nv.DczError(int(st[k])))
                raise IOError("Chunk decompression failed") from cause  # CpuCompressionService.java:469-471
            gpu_digests = None
            if (len(batch) >= self.SHA_GPU_MIN_CHUNKS and all(c.original_size == stride for c in batch[:-1])
                    and batch[-1].original_size <= stride):  # decoded chunks are contiguous: hash them where they are
                n_dec = (len(batch) - 1) * stride + batch[-1].original_size
                gpu_digests = self.sha256_device(t_out[:n_dec], stride).cpu().numpy()
            out = t_out.cpu().numpy()
            m.record("Decoding", time.perf_counter_ns() - t0, int(sum(c.original_size for c in batch)))
            for i, c in enumerate(batch):
                t0 = time.perf_counter_ns()
                dec = out[i * stride:i * stride + c.original_size]
                actual = gpu_digests[i].tobytes() if gpu_digests is not None else hashlib.sha256(dec.tobytes()).digest()
                if actual != c.sha256:  # CpuCompressionService.java:536-550
                    raise IOError("Checksum mismatch in chunk %d:\n  Expected: %s\n  Actual:   %s\n"
                                  "  Chunk size: %d bytes\n  Compressed size: %d bytes\n  Compressed offset: %d" % (
                                      c.chunk_index, c.sha256.hex(), actual.hex(), c.original_size, c.compressed_size,
                                      c.compressed_offset))
                m.record("Checksum Verification", time.perf_counter_ns() - t0, c.original_size)
                sink(dec)
                done += 1
                if progress_callback is not None:
                    progress_callback(done / num_chunks)
        return header

    def decompress(self, input_path, output_path, progress_callback=None):
        """CompressionService.decompress (CpuCompressionService.java:318-506)."""
        with open(output_path, "wb") as fout:
            self._decode_all(input_path, lambda a: fout.write(a.tobytes()), progress_callback)

    def verify_integrity(self, compressed_path):
        """verifyIntegrity, implemented as a real check (the reference only scans the last 64 KiB for a
        header and never hashes: CpuCompressionService.java:652-694)."""
        try:
            g = hashlib.sha256()
            header = self._decode_all(compressed_path, lambda a: None)
            for c in header.chunks:
                g.update(c.sha256)
            return g.digest() == header.global_checksum
        except (IOError, OSError, ValueError):
            return False
