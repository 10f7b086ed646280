"""GPU parity tests: the HIP path (through the C ABI, libdczhip.so) against the CPU oracle, bit for bit.

Integer/byte work: the bar is exact equality of histograms, code lengths, codewords, payload bytes,
sizes, offsets and decoded bytes.  Full-size cases use size-independent properties (round trip on
device, payload == input for 8-bit codes, sizes from histograms)."""
import ctypes as C
import hashlib

import numpy as np
import pytest

from conftest import pin_input

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    return torch


def hip_compress(svc, data, block_bytes):
    torch = _torch()
    t = torch.from_numpy(np.ascontiguousarray(data)).cuda()
    blk = svc.compress_device(t, block_bytes)
    torch.cuda.synchronize()
    total = int(blk.total.item())
    return (blk, blk.payload[:total].cpu().numpy(), blk.comp_size.cpu().numpy().astype(np.uint32),
            blk.comp_off.cpu().numpy().astype(np.uint64), blk.code_lengths.cpu().numpy().astype(np.int32),
            blk.status.cpu().numpy())


def hip_decompress(svc, blk, n, block_bytes):
    torch = _torch()
    K = blk.num_chunks
    orig = torch.tensor([min(block_bytes, n - k * block_bytes) for k in range(K)], dtype=torch.int32, device="cuda")
    stride = (block_bytes + 15) & ~15
    out, status, errpos = svc.decompress_device(blk.payload, blk.comp_off, blk.comp_size, orig, blk.code_lengths,
                                                stride)
    torch.cuda.synchronize()
    st = status.cpu().numpy()[:K]
    o = out.cpu().numpy()
    dec = np.concatenate([o[k * stride:k * stride + int(orig[k])] for k in range(K)]) if K else np.zeros(0, np.uint8)
    return dec, st, errpos.cpu().numpy()[:K]


def assert_parity(svc, orc, data, block_bytes):
    data = np.ascontiguousarray(data, dtype=np.uint8)
    blk, pay, sizes, offs, lens, status = hip_compress(svc, data, block_bytes)
    opay, osizes, ooffs, olens = orc.compress_blocks(data, block_bytes)
    assert (status == 0).all()
    assert (lens == olens).all(), "code lengths differ from the oracle"
    assert (sizes == osizes).all(), "compressed sizes differ"
    assert (offs == ooffs).all(), "payload offsets differ"
    assert pay.size == opay.size
    if not (pay == opay).all():
        i = int(np.nonzero(pay != opay)[0][0])
        raise AssertionError("payload differs at byte %d of %d (hip %02x oracle %02x)" % (i, pay.size, pay[i], opay[i]))
    dec, st, _ = hip_decompress(svc, blk, data.size, block_bytes)
    assert (st == 0).all()
    assert dec.size == data.size
    if not (dec == data).all():
        bad = np.nonzero(dec != data)[0]
        b = int(bad[0]) // block_bytes
        l = lens[b]
        nb = min(block_bytes, data.size - b * block_bytes)
        raise AssertionError(
            "decode differs from the input: %d bytes in blocks %s of %d; block %d: %d bytes, payload %d at offset %d, %d symbols, "
            "lengths %s, first bad byte %d (got %s want %s); launch shapes %s" % (
                bad.size, np.unique(bad // block_bytes)[:8].tolist(), sizes.size, b, nb, sizes[b], offs[b], int((l > 0).sum()),
                sorted(set(l[l > 0].tolist())), int(bad[0]) - b * block_bytes, dec[bad[0]:bad[0] + 8].tolist(),
                data[bad[0]:bad[0] + 8].tolist(), svc.ctx.launch_shapes()))
    return blk


# ---------------------------------------------------------------------------------------------------
def test_library_is_native_and_loaded(pkg, svc):
    assert pkg.lib().dcz_device_count() >= 1
    assert svc.is_available()
    with open("/proc/self/maps") as f:
        assert "libdczhip.so" in f.read()  # the in-tree HIP library is what this process runs


def test_histogram_kats_through_abi(pkg, vectors):
    fs = pkg.HipFrequencyService(device=0)
    for kat in vectors["histogram_kats"]:
        if "data" in kat:
            data = np.array(kat["data"], dtype=np.uint8)
        elif kat["recipe"] == "identity256":
            data = np.arange(256, dtype=np.uint8)
        else:
            data = np.array([5] * 50 + [10] * 50, dtype=np.uint8)
        h = fs.compute_histogram(data, kat["offset"], kat["length"])
        assert h.sum() == kat["length"]
        if "expect_all" in kat:
            assert (h == kat["expect_all"]).all()
        for k, v in kat.get("expect", {}).items():
            assert h[int(k)] == v, kat["src"]
    # CpuFrequencyServiceTest.java:52-67: 128 KiB of i % 10
    d = (np.arange(128 * 1024) % 10).astype(np.uint8)
    assert (fs.compute_histogram(d) == np.bincount(d, minlength=256)).all()
    assert fs.compute_histogram(np.zeros(0, np.uint8)).sum() == 0
    fs.close()


@pytest.mark.parametrize("n,off,ln", [(1, 0, 1), (15, 3, 9), (1000, 1, 998), (32768, 0, 32768), (32769, 0, 32769),
                                      (100000, 7, 99990), (3 * 32768 + 17, 5, 3 * 32768)])
def test_histogram_windows(pkg, orc, n, off, ln):
    fs = pkg.HipFrequencyService(device=0)
    def runs():  # runs of random bytes (both halves of the packed counters), 1..200 long: K1 adds a 16-byte unit of
        rng = np.random.default_rng(n)  # equal bytes as one add of 16, unit by unit, next to units that are not runs
        return np.repeat(rng.integers(0, 256, size=n, dtype=np.uint8), rng.integers(1, 200, size=n))[:n]

    for gen in (lambda: orc.java_random_bytes(n, n), lambda: orc.gen_lowentropy(n, 0, n),
                lambda: np.zeros(n, np.uint8), lambda: np.full(n, 255, np.uint8), runs):
        d = gen()
        assert (fs.compute_histogram(d, off, ln) == orc.histogram(d, off, ln)).all()
    fs.close()


def _tie_histograms():
    hs = []
    for m in (2, 3, 5, 6, 7, 9, 26, 100, 255, 256):  # all-equal counts, non-powers of two included
        h = np.zeros(256, np.int64)
        h[:m] = 7
        hs.append(h)
    h = np.zeros(256, np.int64)
    h[:40] = [1 << (i // 2) for i in range(40)]  # 1,1,2,2,4,4,...
    hs.append(h)
    fib = [1, 1]
    while len(fib) < 30:
        fib.append(fib[-1] + fib[-2])
    h = np.zeros(256, np.int64)
    h[100:130] = fib  # depth 29
    hs.append(h)
    h = np.zeros(256, np.int64)
    h[[3, 200]] = [5, 1 << 36]
    hs.append(h)
    h = np.zeros(256, np.int64)
    h[0x41] = 2048  # single symbol
    hs.append(h)
    hs.append(np.zeros(256, np.int64))  # no symbol
    # speed_test_input.bin counts (Phase3IntegrationTest.java:99-142): 16 x 20200, 20188, 9 x 20100
    h = np.zeros(256, np.int64)
    h[0x41:0x41 + 16] = 20200
    h[0x41 + 16] = 20188
    h[0x41 + 17:0x41 + 26] = 20100
    hs.append(h)
    rng = np.random.default_rng(11)
    for _ in range(60):  # jqwik generator of HuffmanPropertyTest.java:80-92
        hs.append(rng.integers(0, 1001, 256).astype(np.int64))
    for _ in range(20):  # tie-heavy: tiny alphabets of counts
        hs.append(rng.integers(0, 4, 256).astype(np.int64))
    for _ in range(20):
        hs.append((rng.integers(1, 3, 256) * 4096).astype(np.int64))
    return hs


def test_code_build_matches_priority_queue_oracle(svc, orc):
    for h in _tie_histograms():
        lens, codes = svc.build_codes(h)
        olens, ocodes = orc.build_canonical_codes(h)
        assert (lens == olens).all(), "code lengths differ for hist %s" % h[h > 0][:12]
        assert (codes == ocodes).all()
        assert (svc.codes_from_lengths(olens) == ocodes).all()


def test_code_length_over_32_is_an_error(pkg, svc):
    fib = [1, 1]
    while len(fib) < 36:
        fib.append(fib[-1] + fib[-2])
    h = np.zeros(256, np.int64)
    h[:36] = fib  # depth 35 -> ArrayIndexOutOfBounds in CanonicalHuffman.java:106
    with pytest.raises(pkg.DczError) as e:
        svc.build_codes(h)
    assert e.value.status == pkg.native.DCZ_E_CODELEN
    bad = np.zeros(256, np.int32)
    bad[0] = 33
    with pytest.raises(pkg.DczError) as e:
        svc.codes_from_lengths(bad)
    assert e.value.status == pkg.native.DCZ_E_BADTABLE


@pytest.mark.parametrize("idx", range(11))
def test_reference_payload_pins(svc, orc, vectors, idx):
    pin = vectors["payload_pins"][idx]
    data = pin_input(orc, pin)
    chunk = 1 << 20 if pin["name"] == "mod256_3mib" else max(data.size, 1)
    blk, pay, sizes, offs, lens, status = hip_compress(svc, data, chunk)
    assert pay.size == pin["payload_size"], pin["src"]
    if "payload_hex" in pin:
        assert pay.tobytes().hex() == pin["payload_hex"]
    if pin.get("payload_all_zero"):
        assert not pay.any()
    if pin.get("payload_equals_input"):
        assert (pay == data).all() and (lens == 8).all()
    assert_parity(svc, orc, data, chunk)
    # and through the single-chunk host-pointer ABI (processChunk / decodeChunkParallel seam)
    p2, l2 = svc.encode_chunk(data[:chunk])
    op, ol = orc.encode_block(data[:chunk])
    assert (l2 == ol).all() and p2.size == op.size and (p2 == op).all()
    assert (svc.decode_chunk(p2, l2, min(chunk, data.size)) == data[:chunk]).all()


SIZES = [1, 2, 7, 16, 17, 63, 64, 65, 1000, 1023, 1024, 1025, 4095, 16384, 32767, 32768, 32769, 65536 + 3,
         200000, 3 * 32768 + 31, 1 << 20]


@pytest.mark.parametrize("n", SIZES)
def test_parity_sizes_text(svc, orc, n):
    assert_parity(svc, orc, orc.gen_text(0xD0C2, 0, n), max(n, 1))


@pytest.mark.parametrize("n", SIZES)
def test_parity_sizes_lowentropy(svc, orc, n):
    assert_parity(svc, orc, orc.gen_lowentropy(0xD0C5, 1000, n), max(n, 1))


@pytest.mark.parametrize("n", [1, 5, 64, 4097, 32768, 100001, 1 << 20])
def test_parity_sizes_random(svc, orc, n):
    assert_parity(svc, orc, orc.java_random_bytes(42, n), max(n, 1))


@pytest.mark.parametrize("n,bb", [(10 * 65536, 65536), (10 * 65536 + 777, 65536), (5 * 100000 + 1, 100000),
                                  (7 * 4096 + 5, 4096), (40 * 1000, 1000), (3 * 333 + 1, 333), (64, 1), (1 << 21, 1 << 20),
                                  (33 * 32768, 8 * 32768), (9 * 49152 + 100, 49152)])
def test_parity_multi_block_ragged(svc, orc, n, bb):
    # several chunks with a short last chunk, chunk sizes that are not multiples of 16 or of the segment
    rng = np.random.default_rng(n + bb)
    parts = [orc.gen_text(1, 0, n // 3), orc.gen_lowentropy(2, 0, n // 3), orc.java_random_bytes(3, n - 2 * (n // 3))]
    data = np.concatenate(parts)
    rng.shuffle(data[: n // 2])
    assert_parity(svc, orc, data, bb)


def _fib_data(depth, seed):
    fib = [1, 1]
    while len(fib) < depth + 1:
        fib.append(fib[-1] + fib[-2])
    data = np.repeat(np.arange(40, 40 + len(fib), dtype=np.uint8), fib)
    np.random.default_rng(seed).shuffle(data)
    return data


@pytest.mark.parametrize("depth", [12, 16, 17, 20, 26, 27, 30])
def test_parity_long_codes(svc, orc, depth):
    # maxlen = depth: crosses the reference's 10-bit table, our 11-bit table, the G=4 (<=16), G=2 (<=26)
    # and wide (<=32) encode paths
    data = _fib_data(depth, depth)
    lens, _ = orc.build_canonical_codes(orc.histogram(data))
    assert lens.max() == depth
    assert_parity(svc, orc, data, data.size)
    if data.size > 200000:
        assert_parity(svc, orc, data[: (data.size // 3) * 3], data.size // 3)


def test_decode_errors_match_reference_semantics(pkg, svc, orc):
    lens = np.zeros(256, np.int32)
    lens[0x41] = 1
    comp = np.zeros(4096, np.uint8)
    comp[1234] = 0x04  # a set bit has no code in a single-symbol table
    with pytest.raises(orc.DecodeError) as oe:
        orc.decode_block(comp, lens, 20000)
    with pytest.raises(pkg.HuffmanDecodeError) as he:
        svc.decode_chunk(comp, lens, 20000)
    assert he.value.position == oe.value.position == 1234 * 8 + 5
    assert str(he.value) == str(oe.value) == "Huffman decode error at position %d" % (1234 * 8 + 5)
    # an error beyond the requested symbols is not an error
    assert (svc.decode_chunk(comp, lens, 1234 * 8 + 5) == 0x41).all()
    # bits past the end of the payload read as zero (TableBasedHuffmanDecoder.java:204-208)
    assert (svc.decode_chunk(np.zeros(1, np.uint8), lens, 5000) == orc.decode_block(np.zeros(1, np.uint8), lens, 5000)).all()
    # truncated payload of a real stream decodes like the reference does (zero fill), no crash
    data = orc.gen_text(5, 0, 50000)
    pay, l = orc.encode_block(data)
    cut = pay[: pay.size // 2]
    try:
        want = orc.decode_block(cut, l, data.size)
        got = svc.decode_chunk(cut, l, data.size)
        assert (got == want).all()
    except orc.DecodeError as e:
        with pytest.raises(pkg.HuffmanDecodeError) as he:
            svc.decode_chunk(cut, l, data.size)
        assert he.value.position == e.position
    # empty table: error at position 0
    with pytest.raises(pkg.HuffmanDecodeError) as he:
        svc.decode_chunk(np.zeros(8, np.uint8), np.zeros(256, np.int32), 10)
    assert he.value.position == 0
    # oversubscribed table is rejected (documented deviation: the reference would decode garbage)
    bad = np.zeros(256, np.int32)
    bad[:3] = 1
    with pytest.raises(pkg.DczError) as e:
        svc.decode_chunk(np.zeros(8, np.uint8), bad, 10)
    assert e.value.status == pkg.native.DCZ_E_BADTABLE


def test_decode_foreign_streams(svc, orc):
    # streams produced by the ORACLE encoder with arbitrary (non-Huffman but prefix-free, even incomplete)
    # length tables decode identically: canonical codes from lengths are all the decoder knows
    rng = np.random.default_rng(3)
    for trial in range(6):
        lens = np.zeros(256, np.int32)
        syms = rng.choice(256, size=20, replace=False)
        lens[syms] = rng.integers(5, 14, size=20)  # Kraft sum well below 1
        codes, _ = orc.canonical_codes(lens)
        data = rng.choice(syms, size=30000 + trial).astype(np.uint8)
        pay, _ = orc.encode_block(data, lens, codes)
        assert (svc.decode_chunk(pay, lens, data.size) == data).all()


def test_generators_match_oracle(pkg, svc, orc):
    torch = _torch()
    lib, h = pkg.lib(), svc.ctx.handle
    for n, start in [(1, 0), (17, 4), (4096, 0), (100003, 1 << 20), (1 << 20, 12)]:
        t = torch.zeros(n, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()  # (torch's fill runs on torch's stream, the generators on the context's)
        assert lib.dczu_fill_java_random(h, t.data_ptr(), n, 42, start, None) == 0
        torch.cuda.synchronize()
        assert (t.cpu().numpy() == orc.java_random_bytes(42, start + n)[start:]).all()
        assert lib.dczu_fill_text(h, t.data_ptr(), n, 0xD0C2, start, None) == 0
        torch.cuda.synchronize()
        assert (t.cpu().numpy() == orc.gen_text(0xD0C2, start, n)).all()
        assert lib.dczu_fill_lowentropy(h, t.data_ptr(), n, 0xD0C5, start, None) == 0
        torch.cuda.synchronize()
        assert (t.cpu().numpy() == orc.gen_lowentropy(0xD0C5, start, n)).all()


def test_capacity_error_is_reported_per_chunk(pkg, svc, orc):
    torch = _torch()
    data = orc.java_random_bytes(9, 4 * 65536)
    t = torch.from_numpy(data).cuda()
    out = pkg.DeviceBlocks(torch.zeros(2 * 65536 + 100, dtype=torch.uint8, device="cuda"),
                           torch.zeros(4, dtype=torch.int32, device="cuda"), torch.zeros(4, dtype=torch.int64, device="cuda"),
                           torch.zeros((4, 256), dtype=torch.uint8, device="cuda"),
                           torch.zeros(4, dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.int64, device="cuda"),
                           data.size, 65536)
    svc.compress_device(t, 65536, out=out)
    torch.cuda.synchronize()
    st = out.status.cpu().numpy()
    assert list(st) == [0, 0, pkg.native.DCZ_E_CAPACITY, pkg.native.DCZ_E_CAPACITY]
    assert (out.payload[: 2 * 65536].cpu().numpy() == data[: 2 * 65536]).all()  # chunks that fit are written
    assert not out.payload[2 * 65536:].cpu().numpy().any()                      # nothing beyond them


# ---- BASELINE.json configs at full size: size-independent properties --------------------------------
def _roundtrip_device(svc, t_in, bb):
    torch = _torch()
    n = t_in.numel()
    blk = svc.compress_device(t_in, bb)
    K = blk.num_chunks
    orig = torch.full((K,), bb, dtype=torch.int32, device="cuda")
    if n % bb:
        orig[-1] = n % bb
    out, status, _ = svc.decompress_device(blk.payload, blk.comp_off, blk.comp_size, orig, blk.code_lengths, bb)
    torch.cuda.synchronize()
    assert int(blk.status.abs().sum().item()) == 0 and int(status.abs().sum().item()) == 0
    assert torch.equal(out[:n], t_in), "device round trip is not the identity"
    return blk


def test_config2_test_2mb_single_block(svc, orc, vectors):
    data = np.full(2 * 1024 * 1024, 0x41, np.uint8)  # test_2mb.bin (sha pinned in the golden file)
    pin = [p for p in vectors["payload_pins"] if p["name"] == "a2mib"][0]
    assert hashlib.sha256(data.tobytes()).hexdigest() == pin["sha256"]
    torch = _torch()
    blk = _roundtrip_device(svc, torch.from_numpy(data).cuda(), 32 << 20)  # CLI default chunk 32 MB -> 1 block
    assert blk.num_chunks == 1 and int(blk.total.item()) == 262144
    assert not blk.payload[:262144].any().item()


def test_config3_256mib_uniform_random_1mib_blocks(pkg, svc, orc):
    torch = _torch()
    n, bb = 256 << 20, 1 << 20
    t = torch.empty(n, dtype=torch.uint8, device="cuda")
    assert pkg.lib().dczu_fill_java_random(svc.ctx.handle, t.data_ptr(), n, 42, 0, None) == 0
    blk = _roundtrip_device(svc, t, bb)
    # all 256 code lengths are 8 in every block, codeword(s) = s, payload == input
    assert bool((blk.code_lengths == 8).all().item())
    assert int(blk.total.item()) == n and torch.equal(blk.payload[:n], t)
    # the first MiB is bit-exact against the oracle encoder too
    first = t[:bb].cpu().numpy()
    assert (first == orc.java_random_bytes(42, bb)).all()
    op, ol = orc.encode_block(first)
    assert (blk.payload[:bb].cpu().numpy() == op).all()


@pytest.mark.parametrize("kind", ["text", "lowentropy"])
def test_config4_5_distributions_4mib_blocks(pkg, svc, orc, kind):
    torch = _torch()
    n, bb = (1 << 30) + 12345, 4 << 20  # 1 GiB slice of the 8 / 64 GiB streams, ragged last block
    t = torch.empty(n, dtype=torch.uint8, device="cuda")
    fill = pkg.lib().dczu_fill_text if kind == "text" else pkg.lib().dczu_fill_lowentropy
    seed = 0xD0C2 if kind == "text" else 0xD0C5
    assert fill(svc.ctx.handle, t.data_ptr(), n, seed, 0, None) == 0
    blk = _roundtrip_device(svc, t, bb)
    # compressed sizes equal sum(hist * len) computed independently with torch on the device
    for k in (0, blk.num_chunks // 2, blk.num_chunks - 1):
        chunk = t[k * bb:(k + 1) * bb]
        hist = torch.bincount(chunk.to(torch.int64), minlength=256)
        bits = int((hist * blk.code_lengths[k].to(torch.int64)).sum().item())
        assert int(blk.comp_size[k].item()) == (bits + 7) // 8
    # offsets are the exclusive scan of sizes (payloads concatenated without gaps)
    sizes = blk.comp_size.to(torch.int64)
    assert torch.equal(blk.comp_off, torch.cumsum(sizes, 0) - sizes)
    # first and last block bit-exact against the oracle
    for k in (0, blk.num_chunks - 1):
        chunk = t[k * bb:(k + 1) * bb].cpu().numpy()
        op, ol = orc.encode_block(chunk)
        off, sz = int(blk.comp_off[k].item()), int(blk.comp_size[k].item())
        assert sz == op.size and (blk.code_lengths[k].cpu().numpy() == ol).all()
        assert (blk.payload[off:off + sz].cpu().numpy() == op).all()
    ratio = int(blk.total.item()) / n
    assert (0.55 < ratio < 0.70) if kind == "text" else (0.125 < ratio < 0.15)


def test_file_service_round_trip_container(svc, orc, pkg, tmp_path):
    # CpuCompressionServiceTest.java:98-127: 3 MiB of i % 256 -> 3 chunks, progress in [0, 1], verifyIntegrity
    data = (np.arange(3 * 1024 * 1024 + 999) % 256).astype(np.uint8)
    src, dcz, back = tmp_path / "in.bin", tmp_path / "in.dcz", tmp_path / "back.bin"
    src.write_bytes(data.tobytes())
    prog = []
    svc.compress(src, dcz, prog.append)
    assert prog == sorted(prog) and prog[-1] == 1.0 and all(0 <= p <= 1 for p in prog) and len(prog) == 4
    raw = dcz.read_bytes()
    header, start = pkg.container.locate_header(raw)
    assert start == 0 and len(header.chunks) == 4 and header.original_file_size == data.size
    pay, sizes, offs, lens = orc.compress_blocks(data, 1 << 20)
    assert raw[: pay.size] == pay.tobytes()
    for k, c in enumerate(header.chunks):
        assert (c.compressed_size, c.compressed_offset) == (int(sizes[k]), int(offs[k]))
        assert c.code_lengths == lens[k].tolist()
        assert c.sha256 == orc.sha256(data[k << 20:(k + 1) << 20])
    assert len(raw) == pay.size + 68 + len("in.bin") + 572 * 4 + 8
    svc.decompress(dcz, back)
    assert back.read_bytes() == data.tobytes()
    assert svc.verify_integrity(dcz)
    corrupt = bytearray(raw)
    corrupt[100] ^= 0x40
    (tmp_path / "bad.dcz").write_bytes(bytes(corrupt))
    assert not svc.verify_integrity(tmp_path / "bad.dcz")
    with pytest.raises(IOError):
        svc.decompress(tmp_path / "bad.dcz", back)
    # empty file (CpuCompressionServiceTest.java:81-95): no chunks, 68 + nameLen + 8 bytes
    (tmp_path / "empty.txt").write_bytes(b"")
    svc.compress(tmp_path / "empty.txt", tmp_path / "empty.dcz")
    assert (tmp_path / "empty.dcz").stat().st_size == 85
    svc.decompress(tmp_path / "empty.dcz", back)
    assert back.read_bytes() == b""


def test_zero_padding_semantics_match_reference(pkg, svc, orc):
    """Bits past the end of the payload read as zero (TableBasedHuffmanDecoder.java:204-208): a chunk that asks
    for more symbols than its payload holds keeps decoding the all-zero codeword.  The HIP decoder keeps that
    padding out of its sync fixed point and fills analytically; both must agree with the oracle."""
    data = orc.gen_text(9, 0, 70000)
    pay, lens = orc.encode_block(data)
    for cut, want in [(pay.size, data.size + 5000), (pay.size // 3, data.size), (0, 4000), (1, 300), (17, 70000)]:
        comp = pay[:cut] if cut else np.zeros(0, np.uint8)
        assert (svc.decode_chunk(comp, lens, want) == orc.decode_block(comp, lens, want)).all(), (cut, want)
    # shortest code of length 3 (256 % 3 != 0): padding is a periodic stream that never self-synchronises
    l3 = np.zeros(256, np.int32)
    l3[[10, 20, 30, 40, 50, 60, 70, 80]] = 3
    codes, _ = orc.canonical_codes(l3)
    rng = np.random.default_rng(5)
    msg = rng.choice([10, 20, 30, 40, 50, 60, 70, 80], size=50000).astype(np.uint8)
    p3, _ = orc.encode_block(msg, l3, codes)
    assert (svc.decode_chunk(p3, l3, 120000) == orc.decode_block(p3, l3, 120000)).all()


@pytest.mark.parametrize("period_len", [3, 5, 7])
def test_periodic_runs_inside_a_block(svc, orc, period_len):
    """A long run of one symbol whose code length does not divide the 256-bit subsequence never self-synchronises
    from a wrong phase; the decoder must still converge (slowly) to the exact parse."""
    rng = np.random.default_rng(period_len)
    nsym = 1 << period_len  # complete fixed-length code of `period_len` bits
    alphabet = rng.choice(256, size=nsym, replace=False).astype(np.uint8)
    body = np.full(300000, alphabet[0], np.uint8)          # the run: a periodic bit pattern
    noise = rng.choice(alphabet, size=40000).astype(np.uint8)
    data = np.concatenate([noise, body, noise, alphabet])  # every symbol present -> all lengths equal
    lens, _ = orc.build_canonical_codes(orc.histogram(data))
    assert_parity(svc, orc, data, data.size)
    assert_parity(svc, orc, data[: (data.size // 5) * 5], data.size // 5)


# ---------------------------------------------------------------------------------------------------
# K4 decodes a block by one of three strategies chosen from its average code length (>= 6.5 bits: 48 parked
# symbols per subsequence; <= 72 expected symbols per subsequence: 96 parked symbols; shorter: multi-symbol
# tables) and by one of two launch shapes (>= 1024 blocks / fewer).  These streams sit inside and on the edges
# of every class, include codewords longer than the 11-bit table, and make windows overflow one tile flush.
def _skewed(kind, n, seed):
    rng = np.random.default_rng(seed)
    if kind == "hi7":        # ~7 bits/symbol + rare symbols (long codes)
        p = np.r_[np.full(128, 1.0), np.full(128, 2e-4)]
    elif kind == "hi7_5":    # between 7 and 8 bits: a window's output exceeds one flush of the tile
        p = np.r_[np.full(180, 1.0), np.full(76, 1e-3)]
    elif kind == "edge6_5":  # blocks fall on both sides of the 6.5-bit class boundary
        p = np.r_[np.full(90, 1.0), np.full(166, 3e-4)]
    elif kind == "mid6":     # 6 bits + rare symbols
        p = np.r_[np.full(64, 1.0), np.full(192, 5e-5)]
    elif kind == "mid4":     # geometric, ~4.1 bits
        p = 0.85 ** np.arange(256)
    elif kind == "edge3_5":  # geometric, ~3.5 bits: blocks on both sides of the 72-symbols-per-subsequence boundary
        p = 0.785 ** np.arange(256)
    elif kind == "low_nz":   # ~1.2 bits: long runs of a NON-zero dominant byte (run entries of the short-code tables)
        p = np.full(256, 0.03 / 255)
        p[0xAA] = 0.97
    elif kind == "low_2":    # two frequent symbols, ~1.6 bits: short runs, multi-symbol lookups
        p = np.full(256, 0.04 / 254)
        p[0x00] = 0.60
        p[0x20] = 0.36
    else:
        raise ValueError(kind)
    p = p / p.sum()
    return rng.choice(256, size=n, p=p).astype(np.uint8)


@pytest.mark.parametrize("kind", ["hi7", "hi7_5", "edge6_5", "mid6", "mid4", "edge3_5", "low_nz", "low_2"])
@pytest.mark.parametrize("shape", ["many_blocks", "few_blocks"])
def test_parity_code_length_classes(svc, orc, kind, shape):
    if shape == "many_blocks":
        bb, n = 65536, 1100 * 65536 + 4321   # 1101 blocks: the 256-thread kernels
    else:
        bb, n = 262144, 40 * 262144 + 12345  # 41 blocks: the 1024-thread kernels
    data = _skewed(kind, n, seed=len(kind) * 131 + len(shape))
    blk = assert_parity(svc, orc, data, bb)
    sizes = blk.comp_size.cpu().numpy().astype(np.float64)
    bits = 8.0 * sizes[:-1].mean() / bb
    lo, hi = {"hi7": (6.9, 7.2), "hi7_5": (7.3, 7.8), "edge6_5": (6.4, 6.6), "mid6": (5.9, 6.2), "mid4": (3.9, 4.4),
              "edge3_5": (3.4, 3.7), "low_nz": (1.1, 1.5), "low_2": (1.3, 1.9)}[kind]
    assert lo < bits < hi, "generator drifted away from the class it is meant to exercise: %.3f bits/symbol" % bits


def test_corrupt_and_foreign_blocks_in_the_many_blocks_regime(pkg, svc, orc):
    """Per-block status / error position / bytes against the oracle's decoder when some of >= 1024 blocks are damaged.
    The table is prefix-free but incomplete (Kraft sum < 1), so damaged streams do hit bit patterns without a codeword
    ('Huffman decode error at position i', TableBasedHuffmanDecoder.java:109-111)."""
    torch = _torch()
    rng = np.random.default_rng(17)
    lens = np.zeros(256, np.int32)
    syms = rng.choice(256, size=48, replace=False)
    lens[syms] = rng.integers(6, 14, size=48)
    codes, _ = orc.canonical_codes(lens)
    K, nsym = 1040, 6000
    pays, want = [], []
    for k in range(K):
        data = rng.choice(syms, size=nsym).astype(np.uint8)
        pay, _ = orc.encode_block(data, lens, codes)
        pay = pay.copy()
        if k % 5 == 2:  # flip one bit somewhere
            i = int(rng.integers(0, pay.size))
            pay[i] ^= np.uint8(1 << int(rng.integers(0, 8)))
        if k % 11 == 7:  # truncate: the reference reads zeros past the end
            pay = pay[: pay.size // 2]
        try:
            want.append((0, 0, orc.decode_block(pay, lens, nsym)))
        except orc.DecodeError as e:
            want.append((pkg.native.DCZ_E_BADSTREAM, e.position, None))
        pays.append(pay)
    sizes = np.array([p.size for p in pays], dtype=np.int32)
    offs = np.concatenate([[0], np.cumsum(sizes[:-1], dtype=np.int64)]).astype(np.int64)
    payload = torch.from_numpy(np.concatenate(pays + [np.zeros(16, np.uint8)])).cuda()
    stride = (nsym + 15) & ~15
    out, st, ep = svc.decompress_device(payload, torch.from_numpy(offs).cuda(), torch.from_numpy(sizes).cuda(),
                                        torch.full((K,), nsym, dtype=torch.int32, device="cuda"),
                                        torch.from_numpy(np.tile(lens.astype(np.uint8), (K, 1))).cuda(), stride)
    torch.cuda.synchronize()
    st, ep, out = st.cpu().numpy(), ep.cpu().numpy(), out.cpu().numpy()
    nerr = 0
    for k, (wst, wpos, wdata) in enumerate(want):
        assert st[k] == wst, "block %d: status %d, oracle %d" % (k, st[k], wst)
        if wst:
            nerr += 1
            assert ep[k] == wpos, "block %d: error position %d, oracle %d" % (k, ep[k], wpos)
        else:
            assert (out[k * stride:k * stride + nsym] == wdata).all(), "block %d decodes differently" % k
    assert nerr > 20  # the case is only meaningful if damage does produce decode errors


# ---------------------------------------------------------------------------------------------------
# K5: per-chunk SHA-256 (ChecksumUtil.computeSha256, util/ChecksumUtil.java:11-27) against hashlib and the FIPS 180-4
# example messages.
def test_sha256_known_answers(svc):
    torch = _torch()
    kats = [(b"abc", "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad"),
            (b"abcdbcdecdefdefgefghfghighijhijkijkljklmklmnlmnomnopnopq",
             "248d6a61d20638b8e5c026930c3e6039a33ce45964ff2167f6ecedd419db06c1"),
            (b"a" * 1000000, "cdc76e5c9914fb9281a1c7e284d73e67f1809a48a497200e046d39ccc7112cd0")]
    for msg, want in kats:
        t = torch.from_numpy(np.frombuffer(msg, dtype=np.uint8).copy()).cuda()
        got = svc.sha256_device(t, len(msg)).cpu().numpy()
        assert got.shape == (1, 32) and got[0].tobytes().hex() == want


@pytest.mark.parametrize("n,bb", [(1, 1), (55, 55), (56, 56), (63, 64), (64, 64), (65, 64), (119, 120), (1000, 7),
                                  (3 * 4096 + 17, 4096), (2000 * 1024 + 1, 1024), (33 * 100003, 100003)])
def test_sha256_blocks_match_hashlib(svc, n, bb):
    torch = _torch()
    data = np.random.default_rng(n * 31 + bb).integers(0, 256, size=n, dtype=np.uint8)
    for shift in (0, 3):  # 16-byte aligned and unaligned block starts
        t = torch.zeros(n + shift, dtype=torch.uint8, device="cuda")
        t[shift:] = torch.from_numpy(data).cuda()
        got = svc.sha256_device(t[shift:], bb).cpu().numpy()
        K = (n + bb - 1) // bb
        assert got.shape == (K, 32)
        for k in range(K):
            assert got[k].tobytes() == hashlib.sha256(data[k * bb:(k + 1) * bb].tobytes()).digest(), (n, bb, shift, k)


def test_file_service_uses_device_checksums_for_many_chunks(svc, pkg, orc, tmp_path):
    """Above SHA_GPU_MIN_CHUNKS chunks per batch the container's per-chunk digests come from K5; the file must still be
    the oracle's, byte for byte, and must verify."""
    old_cs, old_min = svc.chunk_size_bytes, svc.SHA_GPU_MIN_CHUNKS
    try:
        svc.chunk_size_bytes = 4096
        svc.SHA_GPU_MIN_CHUNKS = 8
        data = orc.gen_text(99, 0, 600 * 4096 + 1234)
        src = tmp_path / "many.bin"
        src.write_bytes(data.tobytes())
        out = tmp_path / "many.dcz"
        svc.compress(str(src), str(out))
        blob = out.read_bytes()
        header, start = pkg.container.locate_header(blob)
        for c in header.chunks:
            assert c.sha256 == hashlib.sha256(data[c.original_offset:c.original_offset + c.original_size].tobytes()).digest()
        back = tmp_path / "many.out"
        svc.decompress(str(out), str(back))
        assert back.read_bytes() == data.tobytes()
        assert svc.verify_integrity(str(out))
        # a damaged payload byte must be caught by the (device) checksum verification
        bad = bytearray(blob)
        bad[100] ^= 0x40
        (tmp_path / "bad.dcz").write_bytes(bytes(bad))
        with pytest.raises(IOError):
            svc.decompress(str(tmp_path / "bad.dcz"), str(tmp_path / "bad.out"))
    finally:
        svc.chunk_size_bytes, svc.SHA_GPU_MIN_CHUNKS = old_cs, old_min


@pytest.mark.parametrize("kind", ["text", "lowentropy", "random"])
def test_parity_cli_default_32mib_blocks(svc, orc, kind):
    """The reference CLI's default chunk size (cli/DataCompCLI.java: 32 MB): two full blocks and a ragged third."""
    n = 2 * (32 << 20) + 1234567
    data = {"text": lambda: orc.gen_text(0xD0C2, 0, n), "lowentropy": lambda: orc.gen_lowentropy(0xD0C5, 0, n),
            "random": lambda: orc.java_random_bytes(42, n)}[kind]()
    assert_parity(svc, orc, data, 32 << 20)


# ---------------------------------------------------------------------------------------------------
# Seeded fuzz: arbitrary sizes, block sizes and byte distributions (sparse alphabets, heavy skew, long codes), every
# artefact compared with the oracle.  Deterministic (fixed seeds), 300 cases.
def _fuzz_case(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([rng.integers(1, 300), rng.integers(300, 70000), rng.integers(70000, 3000000)]))
    kind = int(rng.integers(0, 6))
    if kind == 0:    # Dirichlet over a random alphabet size
        k = int(rng.integers(1, 257))
        p = np.zeros(256)
        p[rng.choice(256, size=k, replace=False)] = rng.dirichlet(np.full(k, float(rng.choice([0.05, 0.3, 1.0, 10.0]))))
    elif kind == 1:  # geometric with random ratio over a random permutation of the bytes
        p = np.zeros(256)
        p[rng.permutation(256)] = float(rng.uniform(0.3, 0.99)) ** np.arange(256)
    elif kind == 2:  # Fibonacci-like weights: very long codes
        d = int(rng.integers(8, 31))
        w = np.ones(d)
        for i in range(2, d):
            w[i] = w[i - 1] + w[i - 2]
        p = np.zeros(256)
        p[rng.choice(256, size=d, replace=False)] = w
    elif kind == 3:  # runs: piecewise constant data with random run lengths
        vals = rng.integers(0, 256, size=max(1, n // int(rng.integers(1, 200)) + 1), dtype=np.uint8)
        data = np.repeat(vals, int(rng.integers(1, 200)))[:n]
        if data.size < n:
            data = np.resize(data, n)
        p = None
    elif kind == 4:  # two-level mix: mostly one byte, the rest uniform
        p = np.full(256, (1 - float(rng.uniform(0.5, 0.999))) / 255)
        p[int(rng.integers(0, 256))] = 0
        p[int(rng.integers(0, 256))] += 1 - p.sum()
    else:            # uniform over all bytes
        p = np.ones(256)
    if p is not None:
        p = np.maximum(p, 0)
        data = rng.choice(256, size=n, p=p / p.sum()).astype(np.uint8)
    bb_kind = int(rng.integers(0, 4))
    if bb_kind == 0:
        bb = n
    elif bb_kind == 1:
        bb = int(rng.integers(1, n + 1))
    elif bb_kind == 2:
        bb = 1 << int(rng.integers(6, 21))
    else:
        bb = max(1, n // int(rng.integers(1, 40)))
    if (n + bb - 1) // bb > 5000:  # keep the number of blocks (oracle time) bounded
        bb = (n + 4999) // 5000
    return np.ascontiguousarray(data, dtype=np.uint8), int(bb)


@pytest.mark.parametrize("seed", range(300))
def test_fuzz_parity(svc, orc, seed):
    data, bb = _fuzz_case(1000 + seed)
    assert_parity(svc, orc, data, bb)


@pytest.mark.parametrize("seed", [149, 267])
def test_repeated_decodes_of_many_small_blocks_are_identical(svc, orc, seed):
    """Thousands of blocks smaller than one decoder window (one partly filled wave per workgroup, repair rounds, slots that
    run over into the zero padding): the same payload decoded 25 times must give the input 25 times.  Two builds of
    k4_dfa's recording walk passed every single-shot test and got the last symbols of some subsequences wrong in 60 % of
    repeated decodes of these two inputs (round 3; the subsequence registers were marked as modified inside the round loop)."""
    data, bb = _fuzz_case(1000 + seed)
    blk, pay, sizes, offs, lens, status = hip_compress(svc, data, bb)
    assert (status == 0).all()
    for rep in range(25):
        dec, st, _ = hip_decompress(svc, blk, data.size, bb)
        assert (st == 0).all()
        bad = np.nonzero(dec != data)[0]
        assert bad.size == 0, "decode %d differs in %d bytes, first at %d (block %d)" % (rep, bad.size, bad[0], bad[0] // bb)


@pytest.mark.parametrize("seed", range(100))
def test_fuzz_decode_foreign_and_damaged(pkg, svc, orc, seed):
    """Single-block decoder (dcz_decode_block) on streams of arbitrary prefix-free tables (complete or not), intact,
    bit-flipped or truncated: bytes, or the reference's error position, must equal the oracle decoder's."""
    rng = np.random.default_rng(5000 + seed)
    k = int(rng.integers(1, 60))
    lens = np.zeros(256, np.int32)
    syms = rng.choice(256, size=k, replace=False)
    budget = 1.0  # Kraft budget: assign random lengths while the sum of 2^-len stays <= 1
    for s_ in syms:
        lo = 1
        while 2.0 ** -lo > budget and lo < 24:
            lo += 1
        if 2.0 ** -lo > budget:
            break
        ln = int(rng.integers(lo, min(24, lo + 12) + 1))
        lens[s_] = ln
        budget -= 2.0 ** -ln
    used = np.nonzero(lens)[0]
    codes, _ = orc.canonical_codes(lens)
    n = int(rng.integers(1, 60000))
    data = rng.choice(used, size=n).astype(np.uint8)
    pay, _ = orc.encode_block(data, lens, codes)
    pay = pay.copy()
    mode = int(rng.integers(0, 3))
    if mode == 1 and pay.size:
        for _ in range(int(rng.integers(1, 4))):
            pay[int(rng.integers(0, pay.size))] ^= np.uint8(1 << int(rng.integers(0, 8)))
    elif mode == 2:
        pay = pay[: int(rng.integers(0, pay.size + 1))]
        if pay.size == 0:
            pay = np.zeros(1, np.uint8)
    try:
        want = orc.decode_block(pay, lens, n)
    except orc.DecodeError as e:
        with pytest.raises(pkg.HuffmanDecodeError) as he:
            svc.decode_chunk(pay, lens, n)
        assert he.value.position == e.position
        return
    assert (svc.decode_chunk(pay, lens, n) == want).all()


def test_contexts_are_independent_across_threads(pkg, orc):
    """The reference calls the seam from up to 8 pool threads (CpuCompressionService.java:42-44, :94-96): one context per
    thread must work concurrently on one device (ctypes releases the GIL inside the C ABI)."""
    import threading
    results, errors = {}, []

    def work(i):
        try:
            svc = pkg.HipCompressionService(1, 0)
            try:
                for rep in range(3):
                    data = orc.gen_text(100 + i, rep * 1000, 300000 + 7919 * i + rep)
                    pay, lens = svc.encode_chunk(data)
                    opay, olens = orc.encode_block(data)
                    assert (lens == olens).all() and pay.size == opay.size and (pay == opay).all()
                    assert (svc.decode_chunk(pay, lens, data.size) == data).all()
                    h = svc.frequency_service().compute_histogram(data, 0, data.size) if hasattr(svc, "frequency_service") else None
                    if h is not None:
                        assert (np.asarray(h) == np.bincount(data, minlength=256)).all()
                results[i] = True
            finally:
                svc.close()
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert len(results) == 6


@pytest.mark.parametrize("seed", range(120))
def test_code_build_properties_and_oracle_on_random_histograms(svc, orc, seed):
    """HuffmanPropertyTest.java:11-78 (jqwik: 256 counts in 0..1000) through dcz_build_codes: codes of one length are
    distinct, a more frequent symbol never has a longer code, every non-zero count has a code -- and lengths and
    codewords equal the oracle's PriorityQueue emulation exactly (small ranges make equal weights, hence ties, common)."""
    rng = np.random.default_rng(9000 + seed)
    hi = int(rng.choice([2, 5, 50, 1000, 1000000]))
    h = rng.integers(0, hi + 1, size=256).astype(np.int64)
    if seed % 7 == 0:
        h[rng.choice(256, size=int(rng.integers(1, 250)), replace=False)] = 0
    lens, codes = svc.build_codes(h)
    olens, ocodes = orc.build_canonical_codes(h)
    assert (lens == olens).all() and (codes == ocodes).all()
    nz = np.nonzero(h)[0]
    assert (lens[nz] > 0).all() and (lens[h == 0] == 0).all()
    seen = set()
    for s_ in nz:
        assert (int(lens[s_]), int(codes[s_])) not in seen
        seen.add((int(lens[s_]), int(codes[s_])))
    if nz.size > 1:
        order = nz[np.argsort(h[nz], kind="stable")]
        # strictly more frequent => not longer
        for a, b in zip(order[:-1], order[1:]):
            if h[b] > h[a]:
                assert lens[b] <= lens[a]
        assert sum(2.0 ** -int(l) for l in lens[nz]) <= 1.0 + 1e-12  # prefix-free (Kraft)


def test_pipelined_compress_on_a_caller_stream(svc, orc):
    """>= 2048 blocks take the two-half pipelined path inside dcz_compress_blocks (second HIP stream + events); here it
    runs on a caller-provided non-default stream, twice back to back, and is followed by the decode on the same stream
    without any host synchronisation in between."""
    torch = _torch()
    bb, K = 16384, 2100
    data = orc.gen_text(31, 0, bb * K - 777)
    t = torch.from_numpy(data).cuda()
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        blk = svc.compress_device(t, bb, stream=s.cuda_stream)
        blk = svc.compress_device(t, bb, out=blk, stream=s.cuda_stream)
        orig = torch.tensor([min(bb, data.size - k * bb) for k in range(blk.num_chunks)], dtype=torch.int32, device="cuda")
        out, st, ep = svc.decompress_device(blk.payload, blk.comp_off, blk.comp_size, orig, blk.code_lengths, bb,
                                            stream=s.cuda_stream)
    s.synchronize()
    torch.cuda.synchronize()
    opay, osizes, ooffs, olens = orc.compress_blocks(data, bb)
    total = int(blk.total.item())
    assert total == opay.size and (blk.payload[:total].cpu().numpy() == opay).all()
    assert (blk.comp_size.cpu().numpy().astype(np.uint32) == osizes).all()
    assert (st.cpu().numpy()[:blk.num_chunks] == 0).all()
    assert (out.cpu().numpy()[:data.size] == data).all()


# ---------------------------------------------------------------------------------------------------
# Streams that do not self-synchronise: (nearly) fixed-length codes whose length does not divide the 256-bit
# subsequence keep a wrong phase for ever.  A probe launch finds such blocks and the exact-entry instantiations decode
# them (k4_exact_entries).  Parity with the oracle in both launch shapes, with ragged tails and damaged streams.
def _fixed_length_like(kind, n, seed):
    rng = np.random.default_rng(seed)
    if kind == "sym64":      # 64 equiprobable symbols: 6-bit codes (base64-like)
        return rng.integers(0, 64, size=n).astype(np.uint8) + 32
    if kind == "sym128":     # 128 equiprobable symbols + rare others: 7-bit codes, one 8-bit code, long tail
        p = np.r_[np.full(128, 1.0), np.full(128, 2e-4)]
    elif kind == "ripple":   # near-uniform bytes: 7/8/9-bit codes mixed
        p = 1.0 + 0.3 * np.sin(np.arange(256))
    elif kind == "sym3":     # 3 equiprobable symbols: lengths 1/2/2
        return (rng.integers(0, 3, size=n) * 7).astype(np.uint8)
    elif kind == "sym32":    # 5-bit codes
        return (rng.integers(0, 32, size=n) * 3).astype(np.uint8)
    else:
        raise ValueError(kind)
    return rng.choice(256, size=n, p=p / p.sum()).astype(np.uint8)


@pytest.mark.parametrize("kind", ["sym64", "sym128", "ripple", "sym3", "sym32"])
@pytest.mark.parametrize("shape", ["many_blocks", "few_blocks"])
def test_parity_streams_that_do_not_self_synchronise(svc, orc, kind, shape):
    if shape == "many_blocks":
        bb, n = 65536, 1060 * 65536 + 777
    else:
        bb, n = 1 << 20, 9 * (1 << 20) + 54321
    data = _fixed_length_like(kind, n, seed=len(kind) * 17 + len(shape))
    assert_parity(svc, orc, data, bb)


def test_exact_entry_decoder_reports_damage_like_the_reference(pkg, svc, orc):
    """Incomplete table of equal-length codewords (so the stream does not self-synchronise and goes through the
    exact-entry decoder) with bit flips: status, error position and bytes per block against the oracle decoder."""
    torch = _torch()
    rng = np.random.default_rng(23)
    lens = np.zeros(256, np.int32)
    syms = rng.choice(256, size=100, replace=False)
    lens[syms] = 7  # 100 of the 128 seven-bit patterns are codewords
    codes, _ = orc.canonical_codes(lens)
    K, nsym = 1030, 40000
    pays, want = [], []
    for k in range(K):
        data = rng.choice(syms, size=nsym).astype(np.uint8)
        pay, _ = orc.encode_block(data, lens, codes)
        pay = pay.copy()
        if k % 4 == 1:
            i = int(rng.integers(0, pay.size))
            pay[i] ^= np.uint8(1 << int(rng.integers(0, 8)))
        try:
            want.append((0, 0, orc.decode_block(pay, lens, nsym)))
        except orc.DecodeError as e:
            want.append((pkg.native.DCZ_E_BADSTREAM, e.position, None))
        pays.append(pay)
    sizes = np.array([p.size for p in pays], dtype=np.int32)
    offs = np.concatenate([[0], np.cumsum(sizes[:-1], dtype=np.int64)]).astype(np.int64)
    payload = torch.from_numpy(np.concatenate(pays + [np.zeros(16, np.uint8)])).cuda()
    stride = (nsym + 15) & ~15
    out, st, ep = svc.decompress_device(payload, torch.from_numpy(offs).cuda(), torch.from_numpy(sizes).cuda(),
                                        torch.full((K,), nsym, dtype=torch.int32, device="cuda"),
                                        torch.from_numpy(np.tile(lens.astype(np.uint8), (K, 1))).cuda(), stride)
    torch.cuda.synchronize()
    st, ep, out = st.cpu().numpy(), ep.cpu().numpy(), out.cpu().numpy()
    nerr = 0
    for k, (wst, wpos, wdata) in enumerate(want):
        assert st[k] == wst, "block %d: status %d, oracle %d" % (k, st[k], wst)
        if wst:
            nerr += 1
            assert ep[k] == wpos, "block %d: error position %d, oracle %d" % (k, ep[k], wpos)
        else:
            assert (out[k * stride:k * stride + nsym] == wdata).all(), "block %d decodes differently" % k
    assert nerr > 20


# ---------------------------------------------------------------------------------------------------
# Fixed-length complete codes (2^L symbols, every code L bits): decoded analytically by k4_fixed over many workgroups
# per block; K3 copies blocks of 256 8-bit symbols.  Reference semantics: TableBasedHuffmanDecoder.java:78-88, 103-134.
def _fixed_data(rng, L, n):
    syms = np.sort(rng.choice(256, size=1 << L, replace=False)).astype(np.uint8)
    # a balanced multiset (every symbol floor/ceil(n / 2^L) times), shuffled: max count < 2 * min count => all lengths L
    reps = -(-n // (1 << L))
    data = np.tile(syms, reps)[:n].copy()
    rng.shuffle(data)
    return data


@pytest.mark.parametrize("L", [1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("n,bb", [(5 * 65536 + 4321, 65536), (3 * 100000, 100000), (1 << 21, 1 << 21), (40000, 70001)])
def test_parity_fixed_length_codes(svc, orc, L, n, bb):
    rng = np.random.default_rng(1000 * L + n % 977)
    data = _fixed_data(rng, L, n)
    blk = assert_parity(svc, orc, data, bb)
    lens = blk.code_lengths.cpu().numpy()
    fixed = [((l == 0) | (l == L)).all() and (l == L).sum() == (1 << L) for l in lens]
    assert any(fixed)  # (a short last block may have a skewed histogram and an ordinary Huffman code)


def test_fixed_length_blocks_with_truncated_and_oversized_requests(pkg, svc, orc):
    """Zero bits past the payload (TableBasedHuffmanDecoder.java:204-208) and symbol counts beyond the payload, for
    the analytic decoder, against the oracle's decoder; payload offsets of every alignment."""
    rng = np.random.default_rng(77)
    for L in (1, 3, 4, 7, 8):
        data = _fixed_data(rng, L, 50000 + L)
        pay, lens = orc.encode_block(data)
        assert set(np.unique(lens)) <= {0, L}
        for cut, want in [(pay.size, data.size), (pay.size, data.size + 777), (pay.size // 2, data.size), (3, 100), (0, 50)]:
            comp = pay[:cut] if cut else np.zeros(0, np.uint8)
            assert (svc.decode_chunk(comp, lens, want) == orc.decode_block(comp, lens, want)).all(), (L, cut, want)
    # many blocks at once, payload offsets at every byte alignment, mixed with table-walk blocks
    torch = _torch()
    K = 70
    pays, lens_all, origs, want = [], [], [], []
    for k in range(K):
        L = 1 + k % 8
        n = 30000 + 17 * k
        d = _fixed_data(rng, L, n) if k % 5 else orc.gen_text(k, 0, n)
        p, l = orc.encode_block(d)
        pays.append(np.concatenate([p, np.zeros(k % 16, np.uint8)]))  # slack bytes shift the next payload's alignment
        lens_all.append(l.astype(np.uint8))
        origs.append(n)
        want.append(d)
    sizes = np.array([p.size - (k % 16) for k, p in enumerate(pays)], dtype=np.int32)
    offs = np.concatenate([[0], np.cumsum([p.size for p in pays][:-1])]).astype(np.int64)
    payload = torch.from_numpy(np.concatenate(pays + [np.zeros(16, np.uint8)])).cuda()
    stride = (max(origs) + 15) & ~15
    out, st, _ = svc.decompress_device(payload, torch.from_numpy(offs).cuda(), torch.from_numpy(sizes).cuda(),
                                       torch.tensor(origs, dtype=torch.int32, device="cuda"),
                                       torch.from_numpy(np.stack(lens_all)).cuda(), stride)
    torch.cuda.synchronize()
    assert (st.cpu().numpy()[:K] == 0).all()
    o = out.cpu().numpy()
    for k in range(K):
        assert (o[k * stride:k * stride + origs[k]] == want[k]).all(), "block %d" % k


def test_footer_fields_outside_the_buffers_are_rejected(pkg, svc, orc):
    """dcz_decompress_blocks takes untrusted footer fields: offset/size beyond the payload buffer or an original size
    beyond the output stride give DCZ_E_INVALID for that block and nothing is read or written for it."""
    torch = _torch()
    data = orc.gen_text(3, 0, 4 * 20000)
    blk, pay, sizes, offs, lens, status = hip_compress(svc, data, 20000)
    K = 4
    total = int(pay.size)
    payload = blk.payload[:total + 16].clone()
    orig = np.full(K, 20000, np.int32)
    stride = 20000 + 16
    cases = [("offset", 1, offs.astype(np.int64) + np.array([0, total, 0, 0]), sizes, orig),
             ("size", 2, offs.astype(np.int64), sizes.astype(np.int64) + np.array([0, 0, total, 0]), orig),
             ("orig", 3, offs.astype(np.int64), sizes, orig + np.array([0, 0, 0, 64], np.int32))]
    for name, bad, o_, s_, g_ in cases:
        sentinel = torch.full((K * stride,), 0xA5, dtype=torch.uint8, device="cuda")
        out, st, ep = svc.decompress_device(payload[:total], torch.from_numpy(np.asarray(o_, np.int64)).cuda(),
                                            torch.from_numpy(np.asarray(s_, np.int64).astype(np.int32)).cuda(),
                                            torch.from_numpy(np.asarray(g_, np.int32)).cuda(), blk.code_lengths, stride,
                                            t_out=sentinel)
        torch.cuda.synchronize()
        st = st.cpu().numpy()[:K]
        assert st[bad] == pkg.native.DCZ_E_INVALID, name
        assert (np.delete(st, bad) == 0).all(), name
        o = out.cpu().numpy()
        assert (o[bad * stride:(bad + 1) * stride] == 0xA5).all(), name + ": the rejected block was written"
        for k in range(K):
            if k != bad:
                assert (o[k * stride:k * stride + 20000] == data[k * 20000:(k + 1) * 20000]).all()


def test_torch_ops_and_kernels_are_ordered_without_host_sync(pkg, svc, orc):
    """Tensors written by the kernels (on the service's stream) and torch ops queued right after on the caller's stream
    -- what sharding.gather_chunk_sizes does between compress and decompress -- must see each other in program order."""
    torch = _torch()
    from dcz_amd import sharding
    n, bb = 64 << 20, 1 << 20
    t = torch.empty(n, dtype=torch.uint8, device="cuda")
    assert pkg.lib().dczu_fill_text(svc.ctx.handle, t.data_ptr(), n, 5, 0, svc.stream.cuda_stream) == 0
    torch.cuda.synchronize()
    blk = svc.compress_device(t, bb)
    ref_sizes = blk.comp_size.clone()
    torch.cuda.synchronize()
    want = ref_sizes.cpu().numpy().astype(np.int64)
    for _ in range(5):
        blk.comp_size.zero_()  # queued on the caller's stream: must land BEFORE the kernels rewrite the sizes
        blk = svc.compress_device(t, bb, out=blk)
        all_sizes, offsets, base = sharding.gather_chunk_sizes(blk.comp_size, blk.num_chunks)  # no host sync in between
        got = all_sizes.cpu().numpy()
        assert (got == want).all()
        assert (offsets.cpu().numpy() == np.concatenate([[0], np.cumsum(want)[:-1]])).all()


def test_kernels_run_on_the_callers_own_stream_in_program_order(pkg, svc, orc):
    """A caller whose current torch stream is not the default one gets the kernels queued on THAT stream (no service
    stream, no cross-stream events: what bench.py does): torch ops before and after a call see its tensors in program
    order, a whole round trip included, with no host synchronisation in between."""
    torch = _torch()
    from dcz_amd import sharding
    n, bb = 48 << 20, 1 << 20
    t = _gen_device(pkg, svc, "text", n, seed=11)
    want_sizes = svc.compress_device(t, bb).comp_size.clone()
    torch.cuda.synchronize()
    ts = torch.cuda.Stream()
    ts.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(ts):
        assert svc._enter(None, t.device) == (ts.cuda_stream, None)
        blk = svc.compress_device(t, bb)
        orig = torch.full((blk.num_chunks,), bb, dtype=torch.int32, device="cuda")
        for _ in range(4):
            blk.comp_size.zero_()  # queued on ts: must land BEFORE the kernels rewrite the sizes
            blk = svc.compress_device(t, bb, out=blk)
            all_sizes, _, _ = sharding.gather_chunk_sizes(blk.comp_size, blk.num_chunks)
            out, status, _ = svc.decompress_device(blk.payload, blk.comp_off, blk.comp_size, orig, blk.code_lengths, bb)
            same = torch.equal(out[:n], t) and bool((status == 0).all().item()) and torch.equal(all_sizes.to(torch.int32), want_sizes)
            assert same
    torch.cuda.synchronize()


# ---------------------------------------------------------------------------------------------------
# Few large blocks: one block decoded by many workgroups (k4_split.hip: counting pass over regions, proven entries, the
# table-walk kernels once per region).  The reference's own chunk sizes: 32 MiB (cli/DataCompCLI.java:35) and 16 MiB
# (application.conf:10); its decodeChunkParallel (CpuCompressionService.java:511-556) is what a single call replaces.
def _gen_device(pkg, svc, kind, n, seed=None):
    torch = _torch()
    t = torch.empty(n, dtype=torch.uint8, device="cuda")
    lib, h = pkg.lib(), svc.ctx.handle
    if kind == "text":
        assert lib.dczu_fill_text(h, t.data_ptr(), n, seed or 0xD0C2, 0, None) == 0
    elif kind == "lowentropy":
        assert lib.dczu_fill_lowentropy(h, t.data_ptr(), n, seed or 0xD0C5, 0, None) == 0
    else:
        assert lib.dczu_fill_java_random(h, t.data_ptr(), n, seed or 42, 0, None) == 0
    torch.cuda.synchronize()
    return t


@pytest.mark.parametrize("kind", ["text", "lowentropy", "random", "near_uniform"])
@pytest.mark.parametrize("shape", ["1x32MiB", "3x16MiB", "1x2MiB+", "5x3MiB_ragged", "1x100MiB"])
def test_split_decode_of_few_large_blocks(pkg, svc, orc, kind, shape):
    """(1x32MiB and 1x100MiB reach payload offsets beyond 2^24 inside one chunk: a build of k4_dfa with one more subsequence
    dword hoisted decoded exactly that range wrong, the last symbols of some subsequences, while every smaller case passed.)"""
    torch = _torch()
    n, bb = {"1x32MiB": (32 << 20, 32 << 20), "3x16MiB": (48 << 20, 16 << 20), "1x2MiB+": ((2 << 20) + 12345, 4 << 20),
             "5x3MiB_ragged": (4 * (3 << 20) + 777777, 3 << 20), "1x100MiB": (100 << 20, 100 << 20)}[shape]
    if kind == "near_uniform":  # 7/8/9-bit codes: the long-code class, not fixed-length
        rng = np.random.default_rng(3)
        p = 1.0 + 0.3 * np.sin(np.arange(256))
        base = rng.choice(256, size=1 << 22, p=p / p.sum()).astype(np.uint8)
        t = torch.from_numpy(np.resize(base, n)).cuda()
    else:
        t = _gen_device(pkg, svc, kind, n)
    blk = _roundtrip_device(svc, t, bb)
    # first and last block against the oracle (payload bytes and code lengths), the rest through the round trip
    data = t.cpu().numpy()
    K = blk.num_chunks
    for k in {0, K - 1}:
        seg = data[k * bb:min(n, (k + 1) * bb)]
        if seg.size <= (4 << 20):
            pay, lens = orc.encode_block(seg)
            o, c = int(blk.comp_off[k]), int(blk.comp_size[k])
            assert c == pay.size and (blk.payload[o:o + c].cpu().numpy() == pay).all()
            assert (blk.code_lengths[k].cpu().numpy().astype(np.int32) == lens).all()


def test_split_decode_falls_back_on_damaged_and_foreign_streams(pkg, svc, orc):
    """Blocks that the split decoder cannot prove (damage, truncation, requests beyond the payload) must give exactly what
    the one-workgroup-per-block path gives: the oracle decoder's bytes, status and error position."""
    torch = _torch()
    rng = np.random.default_rng(23)
    lens = np.zeros(256, np.int32)
    syms = rng.choice(256, size=48, replace=False)
    lens[syms] = rng.integers(4, 12, size=48)  # prefix-free but incomplete: damage hits patterns without a codeword
    codes, _ = orc.canonical_codes(lens)
    nsym = 1500000
    cases = []
    for k in range(6):
        data = rng.choice(syms, size=nsym + 1000 * k).astype(np.uint8)
        pay, _ = orc.encode_block(data, lens, codes)
        pay = pay.copy()
        want_n = data.size
        if k == 1:
            pay[pay.size // 2] ^= 0x10  # one flipped bit in the middle
        if k == 2:
            pay = pay[: pay.size * 2 // 3]  # truncated: zero bits past the end
        if k == 3:
            want_n += 5000  # more symbols than the payload holds
        if k == 4:
            pay[100] ^= 0x01  # damage inside the first region
        try:
            want = (0, 0, orc.decode_block(pay, lens, want_n))
        except orc.DecodeError as e:
            want = (pkg.native.DCZ_E_BADSTREAM, e.position, None)
        cases.append((pay, want_n, want))
    sizes = np.array([c[0].size for c in cases], dtype=np.int32)
    offs = np.concatenate([[0], np.cumsum(sizes[:-1], dtype=np.int64)]).astype(np.int64)
    origs = np.array([c[1] for c in cases], dtype=np.int32)
    payload = torch.from_numpy(np.concatenate([c[0] for c in cases] + [np.zeros(16, np.uint8)])).cuda()
    stride = (int(origs.max()) + 15) & ~15
    K = len(cases)
    out, st, ep = svc.decompress_device(payload, torch.from_numpy(offs).cuda(), torch.from_numpy(sizes).cuda(),
                                        torch.from_numpy(origs).cuda(),
                                        torch.from_numpy(np.tile(lens.astype(np.uint8), (K, 1))).cuda(), stride)
    torch.cuda.synchronize()
    st, ep, out = st.cpu().numpy(), ep.cpu().numpy(), out.cpu().numpy()
    for k, (pay, want_n, (wst, wpos, wdata)) in enumerate(cases):
        assert st[k] == wst, "block %d: status %d, oracle %d" % (k, st[k], wst)
        if wst:
            assert ep[k] == wpos, "block %d: error position %d, oracle %d" % (k, ep[k], wpos)
        else:
            assert (out[k * stride:k * stride + want_n] == wdata).all(), "block %d decodes differently" % k


def test_compress_and_decompress_are_capturable_in_a_hip_graph(pkg, svc, orc):
    """include/dcz.h: after dcz_ctx_reserve the batched entry points allocate nothing and never synchronise with the host,
    so a compress + decompress pair can be captured once and replayed on new data (pipelined halves on two streams, the
    classify / fixed / table-walk launches of the decoder and, with few large chunks, the split decoder included)."""
    torch = _torch()
    lib, h = pkg.lib(), svc.ctx.handle
    for n, bb, fill, seed in [(2048 * 32768, 32768, lib.dczu_fill_text, 11), (3 * (8 << 20), 8 << 20, lib.dczu_fill_text, 12),
                              (2048 * 32768, 32768, lib.dczu_fill_java_random, 13)]:
        K = n // bb
        svc.ctx.check(lib.dcz_ctx_reserve(h, n, bb))
        t = torch.empty(n, dtype=torch.uint8, device="cuda")
        assert fill(h, t.data_ptr(), n, seed, 0, None) == 0
        blk = svc.compress_device(t, bb)  # allocates the outputs (and warms every kernel up)
        orig = torch.full((K,), bb, dtype=torch.int32, device="cuda")
        out, st, ep = svc.decompress_device(blk.payload, blk.comp_off, blk.comp_size, orig, blk.code_lengths, bb)
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            svc.compress_device(t, bb, out=blk, stream=s.cuda_stream)
            svc.decompress_device(blk.payload, blk.comp_off, blk.comp_size, orig, blk.code_lengths, bb, t_out=out, status=st,
                                  errpos=ep, stream=s.cuda_stream)
        for rep in range(2):  # new input, replay only
            assert fill(h, t.data_ptr(), n, seed + 100 + rep, 0, None) == 0
            out.zero_()
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            assert int(blk.status.abs().sum()) == 0 and int(st.abs().sum()) == 0
            assert torch.equal(out[:n], t), "graph replay did not reproduce the new input"
        data = t[:bb].cpu().numpy()
        if bb <= (1 << 20):
            pay, lens = orc.encode_block(data)
            assert (blk.payload[:int(blk.comp_size[0])].cpu().numpy() == pay).all()


def test_host_batch_entry_points_match_the_oracle(pkg, svc, orc):
    """dcz_compress_host / dcz_decompress_host (what the JNI twins bind): footer columns, payload bytes and per-chunk
    SHA-256 against the oracle and hashlib, round trip through pinned and registered host buffers."""
    lib, h = pkg.lib(), svc.ctx.handle
    for n, bb, gen in [(10 * 65536 + 777, 65536, lambda n: orc.gen_text(4, 0, n)), (3 << 20, 1 << 20, lambda n: orc.java_random_bytes(7, n)),
                       (5 * 100000 + 1, 100000, lambda n: orc.gen_lowentropy(9, 0, n)), ((8 << 20) + 5, 8 << 20, lambda n: orc.gen_text(5, 0, n))]:
        data = np.ascontiguousarray(gen(n))
        K = (n + bb - 1) // bb
        p_in = lib.dcz_ctx_pinned(h, 0, n)
        assert p_in
        C.memmove(p_in, data.ctypes.data, n)
        out = np.zeros(n + 16, np.uint8)
        assert lib.dcz_host_register(out.ctypes.data, out.nbytes) == 0
        sizes, offs = np.zeros(K, np.uint32), np.zeros(K, np.uint64)
        lens, status, sha = np.zeros((K, 256), np.uint8), np.zeros(K, np.int32), np.zeros((K, 32), np.uint8)
        total = C.c_uint64()
        st = lib.dcz_compress_host(h, p_in, n, bb, out.ctypes.data, n, sizes.ctypes.data, offs.ctypes.data, lens.ctypes.data,
                                   status.ctypes.data, C.byref(total), sha.ctypes.data)
        assert st == 0 and (status == 0).all()
        opay, osizes, ooffs, olens = orc.compress_blocks(data, bb)
        assert total.value == opay.size and (sizes == osizes).all() and (offs == ooffs).all()
        assert (lens.astype(np.int32) == olens).all() and (out[:opay.size] == opay).all()
        for k in range(K):
            assert sha[k].tobytes() == hashlib.sha256(data[k * bb:(k + 1) * bb].tobytes()).digest()
        # and back
        origs = np.array([min(bb, n - k * bb) for k in range(K)], np.uint32)
        dec = np.zeros(K * bb, np.uint8)
        dst, dep, dsha = np.zeros(K, np.int32), np.zeros(K, np.int64), np.zeros((K, 32), np.uint8)
        st = lib.dcz_decompress_host(h, out.ctypes.data, int(total.value), offs.ctypes.data, sizes.ctypes.data,
                                     origs.ctypes.data, lens.ctypes.data, K, bb, dec.ctypes.data, dst.ctypes.data,
                                     dep.ctypes.data, dsha.ctypes.data)
        assert st == 0 and (dst == 0).all()
        assert (dec[:n] == data).all() and (dsha == sha).all()
        assert lib.dcz_host_unregister(out.ctypes.data) == 0
    # untrusted columns: an original size beyond the stride is rejected before anything runs
    bad = origs.copy()
    bad[0] = bb + 1
    assert lib.dcz_decompress_host(h, out.ctypes.data, int(total.value), offs.ctypes.data, sizes.ctypes.data,
                                   bad.ctypes.data, lens.ctypes.data, K, bb, dec.ctypes.data, dst.ctypes.data,
                                   dep.ctypes.data, None) == pkg.native.DCZ_E_INVALID


# ---------------------------------------------------------------------------------------------------
# BASELINE.json configs 4 and 5 at the size ONE GPU holds when the stream is sharded over eight (8 GiB): K = 2048 chunks,
# the pipelined two-half compress and the many-blocks decode kernels (the 1 GiB cases above run the few-blocks kernels).
@pytest.mark.parametrize("kind", ["lowentropy", "text"])
def test_config4_5_at_8gib_per_gpu(pkg, svc, orc, kind):
    torch = _torch()
    n, bb = 8 << 30, 4 << 20
    t = _gen_device(pkg, svc, kind, n)
    blk = _roundtrip_device(svc, t, bb)  # asserts every status and out == in over all 8 GiB
    K = blk.num_chunks
    assert K == 2048
    sizes = blk.comp_size.to(torch.int64)
    assert torch.equal(blk.comp_off, torch.cumsum(sizes, 0) - sizes) and int(blk.total.item()) == int(sizes.sum().item())
    for k in (0, K // 2 - 1, K // 2, K - 1):  # both halves of the pipelined compress, first and last block of each
        chunk = t[k * bb:(k + 1) * bb]
        hist = torch.bincount(chunk.to(torch.int64), minlength=256)
        bits = int((hist * blk.code_lengths[k].to(torch.int64)).sum().item())
        assert int(blk.comp_size[k].item()) == (bits + 7) // 8
    for k in (0, K - 1):  # bit-exact against the oracle encoder
        op, ol = orc.encode_block(t[k * bb:(k + 1) * bb].cpu().numpy())
        off, sz = int(blk.comp_off[k].item()), int(blk.comp_size[k].item())
        assert sz == op.size and (blk.code_lengths[k].cpu().numpy() == ol).all()
        assert (blk.payload[off:off + sz].cpu().numpy() == op).all()
    ratio = int(blk.total.item()) / n
    assert (0.55 < ratio < 0.70) if kind == "text" else (0.125 < ratio < 0.15)
    del blk, t
    torch.cuda.empty_cache()


def test_north_star_8gib_uniform_random_1mib_chunks(pkg, svc, orc):
    """The north-star workload at full size: every chunk has 256 codes of 8 bits, the payload is the input (K3's copy
    path), the decoder's fixed-length class reproduces it (k4_fixed)."""
    torch = _torch()
    n, bb = 8 << 30, 1 << 20
    t = _gen_device(pkg, svc, "random", n)
    blk = _roundtrip_device(svc, t, bb)
    assert blk.num_chunks == 8192 and bool((blk.code_lengths == 8).all().item())
    assert int(blk.total.item()) == n and torch.equal(blk.payload[:n], t)
    assert (t[:bb].cpu().numpy() == orc.java_random_bytes(42, bb)).all()
    del blk, t
    torch.cuda.empty_cache()


# ---------------------------------------------------------------------------------------------------
# The nibble automaton (k4_dfa.hip) on what only it sees differently: incomplete tables (the error state), damage,
# truncation and requests past the payload in the medium class, in both launch shapes; tables it must hand over (a 1-bit
# codeword; more than 255 internal nodes) decode through the other kernels with the same results.
def _decode_table_case(pkg, svc, orc, rng, lens, syms, probs, K, nsym):
    """K blocks of nsym (+k) symbols drawn from `syms` with a hand-made length table, some of them damaged, cut short or
    asked for more symbols than they hold: status, error position and bytes against the oracle."""
    torch = _torch()
    codes, _ = orc.canonical_codes(lens)
    p = None if probs is None else probs / probs.sum()
    pays, origs, want = [], [], []
    for k in range(K):
        data = rng.choice(syms, size=nsym + k, p=p).astype(np.uint8)
        pay, _ = orc.encode_block(data, lens, codes)
        pay = pay.copy()
        n = data.size
        if k % 7 == 3:
            pay[int(rng.integers(0, pay.size))] ^= np.uint8(1 << int(rng.integers(0, 8)))
        if k % 11 == 5:
            pay = pay[: pay.size // 2]
        if k % 5 == 1:  # cut on a 32-byte boundary, usually inside a codeword: the reference completes it with zero bits
            pay = pay[: (pay.size * 3 // 4) & ~31]
        if k % 13 == 6:
            n += 300
        try:
            want.append((0, 0, orc.decode_block(pay, lens, n)))
        except orc.DecodeError as e:
            want.append((pkg.native.DCZ_E_BADSTREAM, e.position, None))
        pays.append(pay)
        origs.append(n)
    sizes = np.array([q.size for q in pays], dtype=np.int32)
    offs = np.concatenate([[0], np.cumsum(sizes[:-1], dtype=np.int64)]).astype(np.int64)
    payload = torch.from_numpy(np.concatenate(pays + [np.zeros(16, np.uint8)])).cuda()
    stride = (max(origs) + 15) & ~15
    out, st, ep = svc.decompress_device(payload, torch.from_numpy(offs).cuda(), torch.from_numpy(sizes).cuda(),
                                        torch.tensor(origs, dtype=torch.int32, device="cuda"),
                                        torch.from_numpy(np.tile(lens.astype(np.uint8), (K, 1))).cuda(), stride)
    torch.cuda.synchronize()
    st, ep, out = st.cpu().numpy(), ep.cpu().numpy(), out.cpu().numpy()
    for k, (wst, wpos, wdata) in enumerate(want):
        assert st[k] == wst, "block %d: status %d, oracle %d" % (k, st[k], wst)
        if wst:
            assert ep[k] == wpos, "block %d: error position %d, oracle %d" % (k, ep[k], wpos)
        else:
            assert (out[k * stride:k * stride + origs[k]] == wdata).all(), "block %d decodes differently" % k


@pytest.mark.parametrize("shape", ["many_blocks", "few_blocks", "one_large_block"])
def test_recording_walk_slots_that_run_over_and_blocks_that_do_not_record(pkg, svc, orc, shape):
    """k4_dfa's recording walk gives every 32-byte subsequence a slot of 80 symbols (256-thread workgroups; 128 with 1024
    threads).  Text with bursts of its shortest codeword completes up to 128 symbols per subsequence: the windows that hold
    a burst are redone by the output walk, the others are not.  Blocks that average more than 60 symbols per subsequence
    (< 4.27 bits per symbol, still the medium class) do not record at all.  All three launch shapes of the automaton."""
    rng = np.random.default_rng(17)
    bb, K = {"many_blocks": (16384, 900), "few_blocks": (1 << 20, 5), "one_large_block": (8 << 20, 1)}[shape]
    n = bb * K - 123
    text = orc.gen_text(23, 0, n).copy()
    top = int(np.bincount(text, minlength=256).argmax())
    for pos in rng.integers(0, n - 3000, size=max(8, n // 200000)):  # bursts inside blocks and across block boundaries
        text[int(pos):int(pos) + int(rng.integers(200, 2500))] = top
    blk = assert_parity(svc, orc, text, bb)
    lens = blk.code_lengths.cpu().numpy()
    assert lens[:, top].min() <= 3, "the burst symbol must be short enough to overrun a slot of 80"
    bits = 8.0 * blk.comp_size.cpu().numpy().sum() / n
    assert bits >= 4.27, "these blocks are meant to record (%.2f bits per symbol)" % bits
    # ~3.9 bits per symbol: medium class, more than 60 symbols per subsequence on average
    pr = np.r_[np.full(8, 3.0), np.full(8, 1.0)]
    dense = rng.choice(16, size=n, p=pr / pr.sum()).astype(np.uint8) + 40
    blk = assert_parity(svc, orc, dense, bb)
    bits = 8.0 * blk.comp_size.cpu().numpy().sum() / n
    assert 3.6 < bits < 4.27, "these blocks are meant to stay with the three walks (%.2f bits per symbol)" % bits


@pytest.mark.parametrize("shape", ["many_blocks", "few_blocks"])
@pytest.mark.parametrize("table", ["incomplete", "complete", "one_bit", "deep_chain"])
def test_medium_class_tables_and_damage_against_the_oracle(pkg, svc, orc, shape, table):
    torch = _torch()
    rng = np.random.default_rng({"incomplete": 1, "complete": 2, "one_bit": 3, "deep_chain": 4}[table])
    lens = np.zeros(256, np.int32)
    if table == "incomplete":
        syms = rng.choice(256, size=24, replace=False)
        lens[syms] = rng.integers(3, 9, size=24)  # Kraft sum < 1: damage meets patterns without a codeword
        while sum(2.0 ** -l for l in lens[lens > 0]) > 1.0:
            lens[syms[np.argmin(lens[syms])]] += 1
        probs = None
    elif table == "complete":
        data0 = orc.gen_text(7, 0, 200000)
        _, l0 = orc.encode_block(data0)
        lens[:] = l0
        syms = np.nonzero(lens)[0]
        probs = np.bincount(data0, minlength=256)[syms].astype(float)
    elif table == "one_bit":
        syms = np.array([5, 9, 17, 33, 65, 129, 200, 201, 202])
        lens[syms] = [1, 2, 3, 4, 5, 6, 7, 8, 8]  # complete, with a 1-bit codeword: not the automaton's
        probs = np.array([2, 2, 3, 6, 10, 16, 20, 20, 21], float)  # skewed to ~5 bits per symbol (medium class)
    else:  # a few very long codewords: > 255 internal nodes in the code tree
        syms = rng.choice(256, size=30, replace=False)
        lens[syms[:12]] = 32
        lens[syms[12:]] = rng.integers(3, 7, size=18)
        while sum(2.0 ** -l for l in lens[lens > 0]) > 1.0:
            lens[syms[12 + int(rng.integers(0, 18))]] += 1
        probs = np.r_[np.full(12, 1e-4), np.full(18, 1.0)]
    K, nsym = (800, 5000) if shape == "many_blocks" else (12, 70000)
    _decode_table_case(pkg, svc, orc, rng, lens, syms, probs, K, nsym)


@pytest.mark.parametrize("shape", ["many_blocks", "few_blocks"])
@pytest.mark.parametrize("table", ["pairs", "incomplete", "deep", "many_symbols", "not_sparse"])
def test_sparse_class_tables_and_damage_against_the_oracle(pkg, svc, orc, shape, table):
    """Blocks dominated by a 1-bit symbol (k4_dfa's SPARSE instantiation: up to four symbols per nibble, two of them
    other than the 1-bit one) and the tables it leaves to k4_decode's short-code kernel."""
    rng = np.random.default_rng({"pairs": 11, "incomplete": 12, "deep": 13, "many_symbols": 14, "not_sparse": 15}[table])
    lens = np.zeros(256, np.int32)
    if table == "pairs":       # 2-bit and 3-bit neighbours: nibbles that complete two symbols other than the 1-bit one
        syms = np.array([0x41, 0x00, 0x7F, 0xFE])
        lens[syms] = [2, 1, 3, 3]
        probs = np.array([0.05, 0.90, 0.025, 0.025])
    elif table == "incomplete":  # Kraft sum < 1: damage meets patterns without a codeword
        syms = np.array([0xAA, 3, 200])
        lens[syms] = [1, 3, 4]
        probs = np.array([0.96, 0.03, 0.01])
    elif table == "deep":      # a chain of codewords down to 24 bits
        syms = np.arange(10, 10 + 24)
        lens[syms] = np.r_[np.arange(1, 24), 23]
        probs = np.r_[0.97, np.full(23, 0.03 / 23)]
    elif table == "many_symbols":  # the 1-bit symbol + 255 codewords of 9 bits: > 255 internal nodes, not the automaton's
        syms = np.arange(256)
        lens[:] = 9
        lens[0] = 1
        probs = np.r_[0.97, np.full(255, 0.03 / 255)]
    else:                      # a 1-bit symbol that is not dominant enough (>= 1.3 bits per symbol): multi-symbol tables
        syms = np.array([0, 1, 2, 3])
        lens[syms] = [1, 2, 3, 3]
        probs = np.array([0.6, 0.2, 0.1, 0.1])
    K, nsym = (800, 60000) if shape == "many_blocks" else (12, 700000)
    _decode_table_case(pkg, svc, orc, rng, lens, syms, probs, K, nsym)


@pytest.mark.parametrize("shape", ["many_blocks", "few_blocks"])
def test_sparse_windows_composed_in_lds_and_windows_whose_lists_run_over(pkg, svc, orc, shape):
    """k4_dfa's SPARSE instantiation records the symbols other than the 1-bit one in per-lane lists of 12 entries and
    composes the window's output in LDS, written once; a window in which a list runs over is filled and patched in global
    memory instead.  Zero pages with 1 % noise and bursts of 30 % noise: both kinds of window, in either order, with the
    16-byte carry between them, in both launch shapes."""
    rng = np.random.default_rng(29)
    bb, K = (32768, 800) if shape == "many_blocks" else (2 << 20, 6)
    n = bb * K - 77
    data = orc.gen_lowentropy(0xD0C5, 0, n).copy()
    for pos in rng.integers(0, n - 5000, size=max(10, n // 300000)):
        ln = int(rng.integers(300, 4000))
        burst = rng.integers(1, 256, size=ln).astype(np.uint8)
        burst[rng.random(ln) > 0.3] = 0
        data[int(pos):int(pos) + ln] = burst
    blk = assert_parity(svc, orc, data, bb)
    bits = 8.0 * blk.comp_size.cpu().numpy().astype(np.float64) / bb
    assert (bits < 1.3).mean() > 0.8, "most blocks are meant to stay in the sparse class (%.0f %% do)" % (100 * (bits < 1.3).mean())


@pytest.mark.parametrize("seed", range(160))
def test_fuzz_decode_random_tables(pkg, svc, orc, seed):
    """Random length tables (Huffman codes of random histograms, complete or made incomplete by lengthening codewords) and
    data of matching statistics, decoded through the batched entry point with damaged, cut and over-asked blocks mixed in:
    whichever kernel the classification hands a block to, status, error position and bytes are the oracle's."""
    rng = np.random.default_rng(7000 + seed)
    nsyms = int(rng.choice([2, 3, 5, 9, 17, 40, 100, 200, 256]))
    syms = np.sort(rng.choice(256, size=nsyms, replace=False))
    shape = rng.choice(["flat", "geometric", "one_dominant", "two_level"])
    if shape == "flat":
        p = rng.random(nsyms) + 0.2
    elif shape == "geometric":
        p = float(rng.uniform(0.55, 0.97)) ** np.arange(nsyms)
    elif shape == "one_dominant":
        p = np.full(nsyms, (1.0 - float(rng.uniform(0.5, 0.995))) / max(1, nsyms - 1))
        p[0] = 1.0 - p[1:].sum() if nsyms > 1 else 1.0
    else:
        p = np.r_[np.full(nsyms // 2 + 1, 1.0), np.full(nsyms - nsyms // 2 - 1, float(rng.uniform(1e-4, 0.1)))][:nsyms]
    p = p / p.sum()
    freq = np.zeros(256, np.int64)
    freq[syms] = np.maximum(1, (p * 1e6).astype(np.int64))
    lens = np.asarray(orc.build_canonical_codes(freq)[0], np.int32).copy()
    if seed % 3 == 1:  # incomplete: Kraft sum < 1
        for s_ in rng.choice(syms, size=max(1, nsyms // 4), replace=False):
            if lens[s_] < 30:
                lens[s_] += int(rng.integers(1, 3))
    K, nsym = (24, 60000) if seed % 2 else (300, 6000)
    _decode_table_case(pkg, svc, orc, rng, lens, syms, p, K, nsym)


def test_launch_shape_hint_both_shapes_on_both_kinds_of_input(pkg, orc):
    """k4_fixed and k3_copy_identity get a flat grid when a recent call met a block for them and a small persistent grid
    otherwise (ShapeHint); either shape must handle either kind of input, whatever the calls before were."""
    torch = _torch()
    svc = pkg.HipCompressionService(1, 0)  # a context of its own: the hint is per context
    text = orc.gen_text(5, 0, 6 * 65536 + 100)
    rnd = orc.java_random_bytes(77, 40 * 65536 + 4321)   # 256 symbols of 8 bits in every full block: identity / fixed
    six = np.random.default_rng(5).integers(0, 64, size=9 * 65536 + 17).astype(np.uint8) + 32  # fixed-length, 6 bits
    seq = [text] * 2 + [rnd] * 3 + [text] * 10 + [rnd, six, rnd] + [text] * 10 + [six] * 2
    for i, data in enumerate(seq):
        assert_parity(svc, orc, data, 65536)
    svc.close()


def _queue_pairs(pkg, svc, datas, bb, reps):
    """Queue reps compress + decompress pairs per input with NO host synchronisation in between (every pair has its
    own output tensors), then synchronise once -> list of (input, DeviceBlocks, decoded tensor, status)."""
    torch = _torch()
    jobs = []
    # (growing the context's workspace frees the old one, which waits for the device: size it before anything is queued)
    svc.ctx.check(pkg.lib().dcz_ctx_reserve(svc.ctx.handle, max(d.size for d in datas), bb))
    for r in range(reps):
        data = datas[r % len(datas)]
        t = torch.from_numpy(data).cuda()
        K = (data.size + bb - 1) // bb
        orig = torch.tensor([min(bb, data.size - k * bb) for k in range(K)], dtype=torch.int32, device="cuda")
        out = pkg.DeviceBlocks(torch.empty(data.size + 16, dtype=torch.uint8, device="cuda"),
                               torch.zeros(K, dtype=torch.int32, device="cuda"), torch.zeros(K, dtype=torch.int64, device="cuda"),
                               torch.zeros((K, 256), dtype=torch.uint8, device="cuda"),
                               torch.zeros(K, dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.int64, device="cuda"),
                               data.size, bb)
        jobs.append((data, t, orig, torch.empty(K * bb, dtype=torch.uint8, device="cuda"),
                     torch.zeros(K, dtype=torch.int32, device="cuda"), torch.zeros(K, dtype=torch.int64, device="cuda"), out))
    torch.cuda.synchronize()
    res = []
    first_done = torch.cuda.Event()
    torch.cuda._sleep(int(1.5e9))  # hold the device back (~0.6 s) so that every call below is issued before the first one runs
    for data, t, orig, t_out, st, ep, out in jobs:  # nothing below allocates or waits for the device
        blk = svc.compress_device(t, bb, out=out)
        svc.decompress_device(blk.payload, blk.comp_off, blk.comp_size, orig, blk.code_lengths, bb, t_out=t_out, status=st,
                              errpos=ep)
        if not res:
            first_done.record()
        res.append((data, blk, t_out, st))
    assert not first_done.query(), "the queue was not deep: the first pair completed while the others were being issued"
    torch.cuda.synchronize()
    return res


def _check_pairs(orc, res, bb):
    for data, blk, t_out, st in res:
        opay, osizes, ooffs, olens = orc.compress_blocks(data, bb)
        assert (blk.status.cpu().numpy() == 0).all() and (st.cpu().numpy() == 0).all()
        assert (blk.code_lengths.cpu().numpy().astype(np.int32) == olens).all()
        assert (blk.comp_size.cpu().numpy().astype(np.uint32) == osizes).all()
        assert (blk.comp_off.cpu().numpy().astype(np.uint64) == ooffs).all()
        assert int(blk.total.item()) == opay.size
        assert (blk.payload[: opay.size].cpu().numpy() == opay).all(), "payload differs from the oracle"
        assert (t_out[: data.size].cpu().numpy() == data).all(), "round trip differs"


def test_queued_calls_keep_the_launch_shape_of_the_last_completed_calls(pkg, orc):
    """The back-to-back call pattern of benchmark/BenchmarkSuite.java:77-101 (and of bench.py's timed loop): many calls
    queued with no host synchronisation.  The launch shapes must follow what the last COMPLETED calls met, however deep
    the queue is -- round 2 compared with the number of calls ISSUED, and every call after the 8th queued one fell back to
    the persistent shapes and the unfused K1 (1409 -> 590 GB/s under the driver's 20-step loop)."""
    torch = _torch()
    svc = pkg.HipCompressionService(1, 0)  # a context of its own: the hint is per context
    bb, nv = 65536, pkg.native
    rnd = [orc.java_random_bytes(900 + i, 48 * bb) for i in range(3)]  # every block: 256 symbols of 8 bits
    text = [orc.gen_text(40 + i, 0, 48 * bb) for i in range(3)]
    for i in range(10):  # completed all-identity calls: the state bench.py's warm-up leaves behind
        assert_parity(svc, orc, rnd[i % 3], bb)
    svc.ctx.reset_profiling()
    svc.ctx.set_profiling(True)
    res = _queue_pairs(pkg, svc, rnd, bb, 24)
    sh = svc.ctx.launch_shapes()
    assert sh == {"decode_flat": 24, "decode_persistent": 0, "encode_flat": 0, "encode_persistent": 24}, sh
    assert svc.ctx.kernel_time(nv.K_HISTOGRAM)[1] == 0, "a queued call fell back to the unfused K1"
    assert svc.ctx.kernel_time(nv.K_HISTOGRAM_COPY)[1] == 24
    _check_pairs(orc, res, bb)
    # the other way round: completed text calls, then a deep queue of text calls -> persistent grids throughout
    for i in range(10):
        assert_parity(svc, orc, text[i % 3], bb)
    svc.ctx.reset_profiling()
    res = _queue_pairs(pkg, svc, text, bb, 24)
    sh = svc.ctx.launch_shapes()
    assert sh == {"decode_flat": 0, "decode_persistent": 24, "encode_flat": 0, "encode_persistent": 24}, sh
    assert svc.ctx.kernel_time(nv.K_HISTOGRAM_COPY)[1] == 0
    _check_pairs(orc, res, bb)
    svc.close()


def test_a_wrong_launch_shape_still_gives_the_oracles_bytes(pkg, orc):
    """Completed text calls, then a deep queue of all-identity / fixed-length inputs: every queued call gets the
    persistent shapes of k4_fixed and k3_copy_identity WITH work for them (more flagged blocks than one range of the
    compaction holds, several tiles per block, a ragged last block), and the other way round: in-place speculation and
    flat grids on text."""
    torch = _torch()
    svc = pkg.HipCompressionService(1, 0)
    bb = 32768
    rng = np.random.default_rng(11)
    rnd = orc.java_random_bytes(5, 1300 * bb + 777)  # 1300 identity blocks (> 2 ranges of 512) + a ragged one
    six = (rng.integers(0, 64, size=700 * bb + 5).astype(np.uint8) + 32)  # fixed-length complete code, 6 bits
    text = orc.gen_text(9, 0, 40 * bb)
    for i in range(10):
        assert_parity(svc, orc, text, bb)
    svc.ctx.reset_profiling()
    res = _queue_pairs(pkg, svc, [rnd, six], bb, 6)
    sh = svc.ctx.launch_shapes()
    assert sh["decode_persistent"] == 6 and sh["encode_persistent"] == 6, sh
    _check_pairs(orc, res, bb)
    small = orc.java_random_bytes(6, 64 * bb)
    for i in range(10):
        assert_parity(svc, orc, small, bb)
    svc.ctx.reset_profiling()
    res = _queue_pairs(pkg, svc, [text, rnd[: 100 * bb], text], bb, 6)
    sh = svc.ctx.launch_shapes()
    assert sh["decode_flat"] == 6 and sh["encode_persistent"] == 6, sh  # (in place: speculated, wrong for the text calls)
    _check_pairs(orc, res, bb)
    svc.close()


def test_identity_blocks_are_stored_in_place_by_the_histogram_pass(pkg, orc):
    """After calls whose blocks all had the identity code (256 symbols of 8 bits), K1 also stores the input at the same
    offsets of the output and K3 leaves those blocks alone (DCZ_K_HISTOGRAM_COPY).  The decision rests on what earlier
    calls found, so every way it can be wrong must still give the oracle's bytes: a call with other blocks mixed in, a
    ragged last block, an input that is not aligned like the output, an output that is too small for the copy."""
    torch = _torch()
    svc = pkg.HipCompressionService(1, 0)  # a context of its own: the hint is per context
    svc.ctx.set_profiling(True)
    K_COPY, bb = pkg.native.K_HISTOGRAM_COPY, 65536
    rnd = orc.java_random_bytes(31, 24 * bb)
    for i in range(14):  # all blocks identity: the later calls run the fused kernel
        assert_parity(svc, orc, rnd if i % 2 else orc.java_random_bytes(100 + i, 24 * bb), bb)
        torch.cuda.synchronize()
    assert svc.ctx.kernel_time(K_COPY)[1] > 0, "the fused K1 never ran on fourteen all-identity calls"
    mixed = np.concatenate([rnd[:8 * bb], orc.gen_text(3, 0, 4 * bb), rnd[8 * bb:14 * bb]])
    assert_parity(svc, orc, mixed, bb)                        # speculated, wrong for four blocks: K3 writes everything that moved
    torch.cuda.synchronize()
    before = svc.ctx.kernel_time(K_COPY)[1]
    assert_parity(svc, orc, rnd, bb)                          # the call before had other blocks: no speculation now
    torch.cuda.synchronize()
    assert svc.ctx.kernel_time(K_COPY)[1] == before
    for i in range(12):
        assert_parity(svc, orc, rnd, bb)
        torch.cuda.synchronize()
    assert svc.ctx.kernel_time(K_COPY)[1] > before             # ... and back again
    assert_parity(svc, orc, rnd[: 20 * bb + 4321], bb)        # ragged last block (not identity)
    for i in range(10):
        assert_parity(svc, orc, rnd, bb)
        torch.cuda.synchronize()
    # input misaligned against the output: no copy, same bytes
    t = torch.from_numpy(np.concatenate([np.zeros(3, np.uint8), rnd])).cuda()
    n0 = svc.ctx.kernel_time(K_COPY)[1]
    blk = svc.compress_device(t[3:], bb)
    torch.cuda.synchronize()
    assert svc.ctx.kernel_time(K_COPY)[1] == n0
    opay, _, _, _ = orc.compress_blocks(rnd, bb)
    assert (blk.payload[: int(blk.total.item())].cpu().numpy() == opay).all()
    # output too small for the copy (and for the payload): per-chunk capacity errors as ever, nothing past the capacity
    out = pkg.DeviceBlocks(torch.zeros(10 * bb + 100, dtype=torch.uint8, device="cuda"),
                           torch.zeros(24, dtype=torch.int32, device="cuda"), torch.zeros(24, dtype=torch.int64, device="cuda"),
                           torch.zeros((24, 256), dtype=torch.uint8, device="cuda"),
                           torch.zeros(24, dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.int64, device="cuda"),
                           rnd.size, bb)
    guard = torch.full((4096,), 0x5A, dtype=torch.uint8, device="cuda")
    svc.compress_device(torch.from_numpy(rnd).cuda(), bb, out=out)
    torch.cuda.synchronize()
    st = out.status.cpu().numpy()
    assert list(st[:10]) == [0] * 10 and (st[10:] == pkg.native.DCZ_E_CAPACITY).all()
    assert (out.payload[: 10 * bb].cpu().numpy() == rnd[: 10 * bb]).all()
    assert bool((guard == 0x5A).all())


def test_bench_line_keeps_the_drivers_contract():
    """bench.py on a small slice of the default workload, run the way the driver runs it (its own process, one JSON line on
    stdout): the keys the driver and the judge read are there, the arithmetic inside the line is consistent (frac =
    achieved / peak, value = bytes / time), the round trip was verified, and no figure exceeds the HBM peak."""
    import json
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--bytes-per-gpu",
           str(256 << 20), "--cpu-sample-mib", "16"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["unit"] == "GB/s" and d["dtype"] == "u8"
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["verified_bit_exact_round_trip"] is True
    n = d["config"]["bytes_per_gpu"]
    assert n == 256 << 20
    assert abs(d["value"] - n / (d["ms_per_step"] * 1e-3) / 1e9) <= 0.01 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0.0 < rf["frac"] < 1.0
    assert abs(rf["achieved"] - rf["alg_bytes_per_launch"] / (rf["avg_launch_ms"] * 1e-3) / 1e9) <= 0.01 * rf["achieved"]
    assert "traffic" in rf  # HBM bytes from the offline PMC pass, or null when the launch mix differs
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 2 and cb["value"] > 0 and "sample" in cb and cb["unit"] == "GB/s"
    for kname, kv in d["kernels"].items():
        assert (kv.get("gbps") or 0.0) <= 8000.0, (kname, kv)
