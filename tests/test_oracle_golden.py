"""Pins the CPU oracle (oracle/dcz_oracle.c) against every known-answer vector the reference holds for
the hot path (tests/golden/reference_vectors.json) and against the properties its own tests assert
(test/.../core/CanonicalHuffmanTest.java, test/.../core/HuffmanPropertyTest.java)."""
import hashlib

import numpy as np
import pytest

from conftest import pin_input


def test_histogram_kats(orc, vectors):
    for kat in vectors["histogram_kats"]:
        if "data" in kat:
            data = np.array(kat["data"], dtype=np.uint8)
        elif kat["recipe"] == "identity256":
            data = np.arange(256, dtype=np.uint8)
        else:  # CpuFrequencyServiceTest.java:70-80
            data = np.array([5] * 50 + [10] * 50, dtype=np.uint8)
        h = orc.histogram(data, kat["offset"], kat["length"])
        assert h.sum() == kat["length"]
        if "expect_all" in kat:
            assert (h == kat["expect_all"]).all()
        for k, v in kat.get("expect", {}).items():
            assert h[int(k)] == v, kat["src"]


def test_concat_kats(orc, vectors):
    kat = vectors["concat_kats"][0]
    lens = np.zeros(256, dtype=np.int32)
    for k, v in kat["lengths"].items():
        lens[int(k)] = v
    codes, maxlen = orc.canonical_codes(lens)
    for k, v in kat["codes"].items():
        assert codes[int(k)] == v
    payload, _ = orc.encode_block(np.frombuffer(kat["text"].encode(), dtype=np.uint8), lens, codes)
    assert payload.tobytes().hex() == kat["payload_hex"]
    # merge = (a << lenB) | b, ReductionBasedEncodingTest.java:27-41
    (va, la), (vb, lb) = vectors["concat_kats"][1]["merge"]
    assert ((va << lb) | vb) == vectors["concat_kats"][1]["value"]


@pytest.mark.parametrize("idx", range(11))
def test_payload_pins(orc, vectors, idx):
    pin = vectors["payload_pins"][idx]
    data = pin_input(orc, pin)
    if "sha256" in pin:
        assert hashlib.sha256(data.tobytes()).hexdigest() == pin["sha256"]
        assert orc.sha256(data).hex() == pin["sha256"]  # the oracle's own SHA-256 too
    chunk = 1 << 20 if pin["name"] == "mod256_3mib" else 1 << 30
    payload, sizes, offs, lens = orc.compress_blocks(data, chunk)
    assert payload.size == pin["payload_size"], pin["src"]
    if "payload_hex" in pin:
        assert payload.tobytes().hex() == pin["payload_hex"]
    if pin.get("payload_all_zero"):
        assert not payload.any()
        assert (lens.sum(axis=1) == 1).all()  # the single-symbol rule, CanonicalHuffman.java:35-45
    if pin.get("payload_equals_input"):
        assert (payload == data).all()
        assert (lens == 8).all()
    if "file_size" in pin and "file_name" in pin:
        K = sizes.size
        assert payload.size + 68 + len(pin["file_name"]) + 572 * K + 8 == pin["file_size"]
    # decode is the inverse on every pinned input
    for k in range(sizes.size):
        blk = data[k * chunk:(k + 1) * chunk]
        dec = orc.decode_block(payload[int(offs[k]):int(offs[k]) + int(sizes[k])], lens[k], blk.size)
        assert (dec == blk).all()


def test_canonical_huffman_unit_properties(orc):
    # CanonicalHuffmanTest.java:12-27 uniform -> every symbol coded
    lens, _ = orc.build_canonical_codes(np.full(256, 100, dtype=np.int64))
    assert (lens == 8).all()
    # :30-45 skewed: most frequent symbol is not longer than the least frequent
    f = np.array([1000 - 3 * i for i in range(256)], dtype=np.int64)
    lens, _ = orc.build_canonical_codes(f)
    assert lens[0] <= lens[255]
    # :48-57 single symbol 42 -> length 1
    f = np.zeros(256, dtype=np.int64)
    f[42] = 100
    lens, codes = orc.build_canonical_codes(f)
    assert lens[42] == 1 and codes[42] == 0 and lens.sum() == 1
    # :60-66 all-zero -> no codes
    lens, _ = orc.build_canonical_codes(np.zeros(256, dtype=np.int64))
    assert not lens.any()
    # :69-94 canonical consecutiveness for freq[i] = i + 1
    lens, codes = orc.build_canonical_codes(np.arange(1, 257, dtype=np.int64))
    for l in set(lens.tolist()):
        c = codes[lens == l]
        assert (np.diff(c.astype(np.int64)) == 1).all()


def test_huffman_properties_jqwik_style(orc):
    # HuffmanPropertyTest.java:11-78 with its generator (256 ints in 0..1000)
    rng = np.random.default_rng(20251114)
    for _ in range(200):
        f = rng.integers(0, 1001, size=256).astype(np.int64)
        if (f > 0).sum() < 2:
            continue
        lens, codes = orc.build_canonical_codes(f)
        assert ((lens > 0) == (f > 0)).all()
        for l in set(lens[lens > 0].tolist()):
            c = codes[lens == l]
            assert len(set(c.tolist())) == c.size
        hi, lo = int(np.argmax(f)), int(np.argmin(np.where(f > 0, f, 1 << 40)))
        assert lens[hi] <= lens[lo]
        assert sum(2.0 ** -int(l) for l in lens[lens > 0]) == 1.0  # a Huffman code is complete


def test_decoder_table_and_fallback(orc):
    # codes longer than the 10-bit table take the fallback path (TableBasedHuffmanDecoder.java:140-152),
    # which no reference test exercises: Fibonacci counts give lengths up to 24.
    fib = [1, 1]
    while len(fib) < 24:
        fib.append(fib[-1] + fib[-2])
    f = np.zeros(256, dtype=np.int64)
    f[:24] = fib
    data = np.repeat(np.arange(24, dtype=np.uint8), fib)
    np.random.default_rng(1).shuffle(data)
    lens, codes = orc.build_canonical_codes(f)
    assert lens.max() == 23
    tsym, tlen = orc.lookup_table(lens)
    assert (tlen[tsym >= 0] <= 10).all() and set(tlen[tsym == -1].tolist()) == {10}  # long-code prefixes marked
    payload, _ = orc.encode_block(data, lens, codes)
    assert (orc.decode_block(payload, lens, data.size) == data).all()


def test_decode_error_position(orc):
    # single-symbol table: a set bit has no code -> "Huffman decode error at position i"
    lens = np.zeros(256, dtype=np.int32)
    lens[0x41] = 1
    comp = np.zeros(16, dtype=np.uint8)
    comp[5] = 0x10  # bit 43
    with pytest.raises(orc.DecodeError) as e:
        orc.decode_block(comp, lens, 128)
    assert e.value.position == 43
    # bits past the end read as zero (TableBasedHuffmanDecoder.java:204-208)
    assert (orc.decode_block(np.zeros(1, dtype=np.uint8), lens, 100) == 0x41).all()


def test_java_random_stream(orc):
    # java.util.Random(42): first nextInt() is -1170105035 (a widely published value)
    b = orc.java_random_bytes(42, 8)
    assert int.from_bytes(b[:4].tobytes(), "little", signed=True) == -1170105035
    # chunked generation equals one big call (TestDataGenerator.java:30-40 refills a 1 MiB buffer)
    assert (orc.java_random_bytes(42, 4096)[:1000] == orc.java_random_bytes(42, 1000)).all()


def test_synthetic_streams_are_position_keyed(orc):
    a = orc.gen_text(7, 0, 5000)
    assert (orc.gen_text(7, 1234, 100) == a[1234:1334]).all()
    z = orc.gen_lowentropy(7, 0, 200000)
    assert (orc.gen_lowentropy(7, 777, 64) == z[777:841]).all()
    assert 0.985 < (z == 0).mean() < 0.995
    lens, _ = orc.build_canonical_codes(orc.histogram(orc.gen_text(0xD0C2, 0, 1 << 20)))
    assert 17 <= lens.max() <= 26  # long enough to leave the reference's 10-bit table
