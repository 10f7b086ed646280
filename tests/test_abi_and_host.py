"""CPU-side tests: the C-ABI library loads and exports every symbol include/dcz.h declares (no compute
calls without a GPU), the product never touches the oracle, and the DCZF container logic is byte-exact."""
import ctypes
import os
import re
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "data-compression-implementing-gpu-driven-huffman-encoding-in-java_amd")


def test_library_exports_every_declared_symbol(pkg):
    with open(os.path.join(ROOT, "include", "dcz.h")) as f:
        hdr = f.read()
    declared = set(re.findall(r"\b(dczu?_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(pkg.native.SYMBOLS), declared ^ set(pkg.native.SYMBOLS)
    lib = ctypes.CDLL(pkg.native.SO_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), "libdczhip.so does not export %s" % name
    # calls that need no device
    L = pkg.lib()
    assert L.dcz_strerror(0) == b"ok" and b"decode" in L.dcz_strerror(pkg.native.DCZ_E_BADSTREAM)
    assert L.dcz_device_count() >= 0


def test_context_creation_fails_loudly_without_a_gpu(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.DczError) as e:
        pkg.Context(0)
    assert e.value.status == pkg.native.DCZ_E_NODEVICE  # no CPU fallback inside the product


def test_product_never_references_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(PKG):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp", ".c", ".java")):
                with open(os.path.join(dirpath, fn), errors="replace") as f:
                    text = f.read()
                if re.search(r"load_oracle|dczoracle|orc_[a-z_]+\(|from oracle|import oracle|oracle/_", text):
                    bad.append(os.path.join(dirpath, fn))
    assert not bad, "product files reference the oracle: %s" % bad
    with open(os.path.join(ROOT, "bench.py")) as f:
        bench = f.read()
    # bench.py may use the oracle only inside its cpu_baseline leg
    assert bench.count("load_oracle") == 1 and "def cpu_baseline" in bench
    assert bench.index("load_oracle") > bench.index("def cpu_baseline")


def test_missing_library_is_an_import_error(pkg, monkeypatch):
    monkeypatch.setattr(pkg.native, "_lib", None)
    monkeypatch.setattr(pkg.native, "SO_PATH", os.path.join(PKG, "no_such_lib.so"))
    with pytest.raises(ImportError):
        pkg.native.lib()


# ---- DCZF container (CompressionHeader.java:51-144) -------------------------------------------------
def _header(pkg, orc, data, chunk, name, ts=1700000000000):
    fmt = pkg.container
    pay, sizes, offs, lens = orc.compress_blocks(data, chunk)
    import hashlib
    g = hashlib.sha256()
    h = fmt.CompressionHeader(name, data.size, ts, b"\0" * 32, chunk)
    for k in range(sizes.size):
        blk = data[k * chunk:(k + 1) * chunk]
        d = hashlib.sha256(blk.tobytes()).digest()
        g.update(d)
        h.add_chunk(fmt.ChunkMetadata(k, k * chunk, blk.size, int(offs[k]), int(sizes[k]), d, lens[k]))
    h.global_checksum = g.digest()
    return h, pay


def test_container_sizes_match_what_the_reference_logged(pkg, orc, vectors):
    from conftest import pin_input
    fmt = pkg.container
    assert fmt.CHUNK_META_BYTES == vectors["container"]["chunk_meta_bytes"] == 572
    for pin in vectors["payload_pins"]:
        if "file_size" not in pin:
            continue
        data = pin_input(orc, pin)
        h, pay = _header(pkg, orc, data, 16 << 20, pin["file_name"])
        blob = pay.tobytes() + h.write() + fmt.footer_pointer(pay.size)
        assert len(blob) == pin["file_size"], pin["src"]
    e = vectors["container"]["empty_file"]
    h = fmt.CompressionHeader(e["file_name"], 0, 0, b"\0" * 32, 1 << 20)
    assert len(h.write()) + 8 == e["file_size"]


def test_container_field_layout_is_big_endian_dataoutputstream(pkg, orc):
    fmt = pkg.container
    data = orc.gen_text(3, 0, 3000)
    h, pay = _header(pkg, orc, data, 1024, "a.txt", ts=0x0102030405060708)
    raw = h.write()
    assert raw[:4] == b"DCZF" and raw[4:8] == b"\0\0\0\1"
    assert struct.unpack(">i", raw[8:12])[0] == 5 and raw[12:17] == b"a.txt"
    assert struct.unpack(">q", raw[17:25])[0] == 3000
    assert raw[25:33] == bytes([1, 2, 3, 4, 5, 6, 7, 8])
    assert struct.unpack(">i", raw[33:37])[0] == 1024
    assert raw[37:69] == h.global_checksum
    assert struct.unpack(">i", raw[69:73])[0] == 3
    assert len(raw) == 68 + 5 + 3 * 572
    c1 = raw[73 + 572:73 + 2 * 572]
    idx, ooff, osz, coff, csz = struct.unpack(">iqiqi", c1[:28])
    assert (idx, ooff, osz) == (1, 1024, 1024) and coff == h.chunks[0].compressed_size
    assert struct.unpack(">256h", c1[60:]) == tuple(h.chunks[1].code_lengths)
    back = fmt.CompressionHeader.read(raw)
    assert back.write() == raw and back.original_file_name == "a.txt"


def test_reader_probe_order_and_errors(pkg, orc):
    fmt = pkg.container
    data = orc.java_random_bytes(5, 5000)
    h, pay = _header(pkg, orc, data, 2048, "r.bin")
    blob = pay.tobytes() + h.write() + fmt.footer_pointer(pay.size)
    got, start = fmt.locate_header(blob)  # footer format: data starts at 0 (CpuCompressionService.java:386)
    assert start == 0 and got.write() == h.write()
    old = h.write() + pay.tobytes()       # header-first format (:338-358)
    got, start = fmt.locate_header(old)
    assert start == len(h.write()) and len(got.chunks) == 3
    with pytest.raises(IOError):
        fmt.locate_header(b"\0" * 100)    # Invalid footer position
    with pytest.raises(IOError):
        fmt.CompressionHeader.read(b"XXXX" + bytes(100))
    with pytest.raises(IOError):
        fmt.CompressionHeader.read(struct.pack(">iii", fmt.MAGIC, 2, 0) + bytes(100))  # Unsupported version
    bad_ptr = pay.tobytes() + h.write() + struct.pack(">q", len(blob))
    with pytest.raises(IOError):
        fmt.locate_header(bad_ptr)


def test_stage_metrics_and_service_surface(pkg):
    m = pkg.StageMetrics()
    m.record("Encoding", 2_000_000, 100)
    m.record("Encoding", 1_000_000, 50)
    assert m.counts["Encoding"] == 2 and m.bytes["Encoding"] == 150 and "Encoding" in m.summary()
    for name in ("compress", "decompress", "resume_compression", "verify_integrity", "get_service_name",
                 "is_available", "close", "get_last_stage_metrics"):  # CompressionService.java:11-66 (+ AutoCloseable)
        assert hasattr(pkg.HipCompressionService, name)
    for name in ("compute_histogram", "get_service_name", "is_available"):  # FrequencyService.java:6-27
        assert hasattr(pkg.HipFrequencyService, name)
