"""DCZF container (".dcz"): byte-exact mirror of core/CompressionHeader.java:51-144 and
core/ChunkMetadata.java:20-30 (big-endian DataOutputStream fields), laid out as
service/cpu/CpuCompressionService.java:155-177 writes it: payloads, footer, 8-byte footer pointer.
Format logic only -- no compression happens here."""
import struct

MAGIC = 0x44435A46  # "DCZF", CompressionHeader.java:15
VERSION = 1         # CompressionHeader.java:16
CHUNK_META_BYTES = 4 + 8 + 4 + 8 + 4 + 32 + 256 * 2  # 572, CompressionHeader.java:71-84


class ChunkMetadata:
    """core/ChunkMetadata.java:20-30."""
    __slots__ = ("chunk_index", "original_offset", "original_size", "compressed_offset", "compressed_size",
                 "sha256", "code_lengths")

    def __init__(self, chunk_index, original_offset, original_size, compressed_offset, compressed_size, sha256,
                 code_lengths):
        self.chunk_index = int(chunk_index)
        self.original_offset = int(original_offset)
        self.original_size = int(original_size)
        self.compressed_offset = int(compressed_offset)
        self.compressed_size = int(compressed_size)
        self.sha256 = bytes(sha256)
        self.code_lengths = [int(x) for x in code_lengths]
        if len(self.sha256) != 32 or len(self.code_lengths) != 256:
            raise ValueError("chunk metadata needs a 32-byte digest and 256 code lengths")


class CompressionHeader:
    """core/CompressionHeader.java:18-48."""

    def __init__(self, original_file_name, original_file_size, original_timestamp, global_checksum, chunk_size_bytes):
        self.original_file_name = original_file_name
        self.original_file_size = int(original_file_size)
        self.original_timestamp = int(original_timestamp)
        self.global_checksum = bytes(global_checksum)
        self.chunk_size_bytes = int(chunk_size_bytes)
        self.chunks = []

    def add_chunk(self, chunk):
        self.chunks.append(chunk)

    def write(self):
        """CompressionHeader.writeTo (CompressionHeader.java:51-85)."""
        name = self.original_file_name.encode("utf-8")
        out = [struct.pack(">iii", MAGIC, VERSION, len(name)), name,
               struct.pack(">qqi", self.original_file_size, self.original_timestamp, self.chunk_size_bytes),
               self.global_checksum, struct.pack(">i", len(self.chunks))]
        for c in self.chunks:
            out.append(struct.pack(">iqiqi", c.chunk_index, c.original_offset, _i32(c.original_size),
                                   c.compressed_offset, _i32(c.compressed_size)))
            out.append(c.sha256)
            out.append(struct.pack(">256h", *c.code_lengths))
        return b"".join(out)

    @staticmethod
    def read(buf, pos=0):
        """CompressionHeader.readFrom (CompressionHeader.java:90-144). Raises IOError like the reference."""
        def need(n):
            if pos + n > len(buf):
                raise IOError("Unexpected end of header")  # EOFException in the reference
        need(12)
        magic, version, name_len = struct.unpack_from(">iii", buf, pos)
        if magic != MAGIC:
            raise IOError("Invalid file format: bad magic number")
        if version != VERSION:
            raise IOError("Unsupported version: %d" % version)
        pos += 12
        if name_len < 0:
            raise IOError("Invalid file format: negative name length")
        need(name_len + 8 + 8 + 4 + 32 + 4)
        name = bytes(buf[pos:pos + name_len]).decode("utf-8", errors="replace")
        pos += name_len
        size, ts, chunk = struct.unpack_from(">qqi", buf, pos)
        pos += 20
        gsum = bytes(buf[pos:pos + 32])
        pos += 32
        (k,) = struct.unpack_from(">i", buf, pos)
        pos += 4
        h = CompressionHeader(name, size, ts, gsum, chunk)
        for _ in range(k):
            need(CHUNK_META_BYTES)
            idx, ooff, osz, coff, csz = struct.unpack_from(">iqiqi", buf, pos)
            pos += 28
            sha = bytes(buf[pos:pos + 32])
            pos += 32
            lens = struct.unpack_from(">256h", buf, pos)
            pos += 512
            h.add_chunk(ChunkMetadata(idx, ooff, osz & 0xFFFFFFFF, coff, csz & 0xFFFFFFFF, sha, lens))
        return h


def _i32(v):
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v >= (1 << 31) else v


def footer_pointer(footer_start):
    """raf.writeLong(footerStart), CpuCompressionService.java:174."""
    return struct.pack(">q", footer_start)


def locate_header(data):
    """Reader probe order of CpuCompressionService.decompress: header at offset 0 from the first <= 4096
    bytes (CpuCompressionService.java:338-358), else the footer through the trailing pointer with the
    0 <= ptr < size-8 check (:366-388).  Returns (header, data_start)."""
    size = len(data)
    try:
        # CpuCompressionService.java:340-341: a min(64 KiB, size) buffer of which only the first 4096 bytes are read
        probe = bytearray(min(64 * 1024, size))
        k = min(4096, len(probe))
        probe[:k] = data[:k]
        h = CompressionHeader.read(probe, 0)
        # header-first ("old") format: compressedDataStart = fileSize - sum(compressedSize)  (:349-353)
        total = sum(c.compressed_size for c in h.chunks)
        if total > size:
            raise IOError("compressed sizes exceed the file")
        return h, size - total
    except (IOError, struct.error):
        pass
    if size < 8:
        raise IOError("Invalid file format: file too small")
    (ptr,) = struct.unpack_from(">q", data, size - 8)
    if ptr < 0 or ptr >= size - 8:
        raise IOError("Invalid footer position: %d" % ptr)
    h = CompressionHeader.read(data[ptr:size - 8], 0)
    return h, 0
