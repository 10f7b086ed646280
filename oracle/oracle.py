"""ctypes wrapper over oracle/_build/libdczoracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
The product package never does (tests/test_no_oracle_in_product.py checks that).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libdczoracle.so")


def build(force=False):
    src = os.path.join(_HERE, "dcz_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u8p, i64p, i32p, u32p = (C.POINTER(C.c_uint8), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_uint32))
        L.orc_histogram.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, i64p]
        L.orc_build_canonical_codes.argtypes = [i64p, i32p, u32p]
        L.orc_build_canonical_codes.restype = C.c_int
        L.orc_build_code_lengths.argtypes = [i64p, i32p]
        L.orc_canonical_codes.argtypes = [i32p, u32p]
        L.orc_canonical_codes.restype = C.c_int
        L.orc_encode_block.argtypes = [C.c_void_p, C.c_size_t, i32p, u32p, C.c_void_p, C.c_size_t]
        L.orc_encode_block.restype = C.c_int64
        L.orc_encoded_size.argtypes = [i64p, i32p]
        L.orc_encoded_size.restype = C.c_int64
        L.orc_decode_block.argtypes = [C.c_void_p, C.c_size_t, i32p, C.c_void_p, C.c_size_t]
        L.orc_decode_block.restype = C.c_int64
        L.orc_build_lookup_table.argtypes = [i32p, i32p, i32p]
        L.orc_java_random_bytes.argtypes = [C.c_int64, C.c_void_p, C.c_size_t]
        L.orc_gen_text.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t]
        L.orc_gen_lowentropy.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t]
        L.orc_sha256.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_roundtrip_blocks_mt.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int,
                                              C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        L.orc_roundtrip_blocks_mt.restype = C.c_int
        _lib = L
    return _lib


def _u8(a):
    if isinstance(a, (bytes, bytearray, memoryview)):
        a = np.frombuffer(bytes(a), dtype=np.uint8)
    return np.ascontiguousarray(a, dtype=np.uint8)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def histogram(data, offset=0, length=None):
    d = _u8(data)
    if length is None:
        length = d.size - offset
    h = np.zeros(256, dtype=np.int64)
    lib().orc_histogram(d.ctypes.data, offset, length, _p(h, C.c_int64))
    return h


def build_canonical_codes(freq):
    f = np.ascontiguousarray(freq, dtype=np.int64)
    ln = np.zeros(256, dtype=np.int32)
    cd = np.zeros(256, dtype=np.uint32)
    n = lib().orc_build_canonical_codes(_p(f, C.c_int64), _p(ln, C.c_int32), _p(cd, C.c_uint32))
    if n < 0:
        raise ValueError("code length > 32 (CanonicalHuffman.java:106 would throw)")
    return ln, cd


def canonical_codes(lengths):
    ln = np.ascontiguousarray(lengths, dtype=np.int32)
    cd = np.zeros(256, dtype=np.uint32)
    m = lib().orc_canonical_codes(_p(ln, C.c_int32), _p(cd, C.c_uint32))
    if m < 0:
        raise ValueError("bad code length")
    return cd, m


def encode_block(data, lengths=None, codes=None):
    """Returns (payload bytes ndarray, lengths int32[256])."""
    d = _u8(data)
    if lengths is None:
        lengths, codes = build_canonical_codes(histogram(d))
    ln = np.ascontiguousarray(lengths, dtype=np.int32)
    if codes is None:
        codes, _ = canonical_codes(ln)
    cd = np.ascontiguousarray(codes, dtype=np.uint32)
    cap = int(d.size) * 4 + 8
    out = np.zeros(cap, dtype=np.uint8)
    n = lib().orc_encode_block(d.ctypes.data, d.size, _p(ln, C.c_int32), _p(cd, C.c_uint32), out.ctypes.data, cap)
    assert n >= 0
    return out[:n].copy(), ln


class DecodeError(RuntimeError):
    def __init__(self, position):
        super().__init__("Huffman decode error at position %d" % position)
        self.position = position


def decode_block(comp, lengths, out_size):
    c = _u8(comp)
    ln = np.ascontiguousarray(lengths, dtype=np.int32)
    out = np.zeros(out_size, dtype=np.uint8)
    r = lib().orc_decode_block(c.ctypes.data, c.size, _p(ln, C.c_int32), out.ctypes.data, out_size)
    if r < 0:
        raise DecodeError(-r - 1)
    return out


def lookup_table(lengths):
    ln = np.ascontiguousarray(lengths, dtype=np.int32)
    s = np.zeros(1024, dtype=np.int32)
    l = np.zeros(1024, dtype=np.int32)
    lib().orc_build_lookup_table(_p(ln, C.c_int32), _p(s, C.c_int32), _p(l, C.c_int32))
    return s, l


def java_random_bytes(seed, n):
    out = np.zeros(n, dtype=np.uint8)
    lib().orc_java_random_bytes(seed, out.ctypes.data, n)
    return out


def gen_text(seed, start, n):
    out = np.zeros(n, dtype=np.uint8)
    lib().orc_gen_text(seed, start, out.ctypes.data, n)
    return out


def gen_lowentropy(seed, start, n):
    out = np.zeros(n, dtype=np.uint8)
    lib().orc_gen_lowentropy(seed, start, out.ctypes.data, n)
    return out


def sha256(data):
    d = _u8(data)
    out = np.zeros(32, dtype=np.uint8)
    lib().orc_sha256(d.ctypes.data, d.size, out.ctypes.data)
    return bytes(out)


def roundtrip_blocks_mt(data, block_bytes, threads):
    """CPU baseline: returns (enc_seconds, dec_seconds, comp_total). Raises on mismatch."""
    d = _u8(data)
    e, dd, ct = C.c_double(), C.c_double(), C.c_uint64()
    r = lib().orc_roundtrip_blocks_mt(d.ctypes.data, d.size, block_bytes, threads, C.byref(e), C.byref(dd), C.byref(ct))
    if r != 0:
        raise RuntimeError("oracle round trip failed")
    return e.value, dd.value, ct.value


def compress_blocks(data, block_bytes):
    """Per-block oracle encode: returns (payload, comp_size[K], comp_off[K], len[K,256])."""
    d = _u8(data)
    K = (d.size + block_bytes - 1) // block_bytes
    parts, sizes, lens = [], [], []
    for b in range(K):
        blk = d[b * block_bytes:(b + 1) * block_bytes]
        p, ln = encode_block(blk)
        parts.append(p)
        sizes.append(p.size)
        lens.append(ln)
    sizes = np.array(sizes, dtype=np.uint32)
    offs = np.zeros(K, dtype=np.uint64)
    if K:
        offs[1:] = np.cumsum(sizes.astype(np.uint64))[:-1]
    payload = np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint8)
    lens = np.array(lens, dtype=np.int32).reshape(K, 256)
    return payload, sizes, offs, lens
