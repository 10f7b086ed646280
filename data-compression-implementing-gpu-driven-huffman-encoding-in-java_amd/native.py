"""ctypes binding of include/dcz.h (libdczhip.so).  Fails loudly if the library is missing: there is
no CPU fallback in this package."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("DCZ_LIB") or os.path.join(HERE, "libdczhip.so")  # DCZ_LIB: tuning experiments only

DCZ_OK = 0
DCZ_E_INVALID = -1
DCZ_E_NODEVICE = -2
DCZ_E_HIP = -3
DCZ_E_CAPACITY = -4
DCZ_E_BADSTREAM = -5
DCZ_E_CODELEN = -6
DCZ_E_BADTABLE = -7

K_HISTOGRAM, K_CODEBUILD, K_OFFSETS, K_ENCODE, K_DECODE = range(5)
K_HISTOGRAM_COPY = 5
KERNEL_NAMES = {K_HISTOGRAM: "k1_histogram", K_CODEBUILD: "k2_codebuild", K_OFFSETS: "k2_offsets",
                K_ENCODE: "k3_encode", K_DECODE: "k4_decode", K_HISTOGRAM_COPY: "k1_histogram_copy"}
SEGMENT_BYTES = 32768

# every symbol include/dcz.h declares (tests check the .so exports each one)
SYMBOLS = [
    "dcz_device_count", "dcz_ctx_create", "dcz_ctx_destroy", "dcz_ctx_stream", "dcz_strerror", "dcz_last_error", "dcz_ctx_reserve",
    "dcz_histogram", "dcz_build_codes", "dcz_codes_from_lengths", "dcz_encode_block", "dcz_decode_block",
    "dcz_compress_blocks", "dcz_decompress_blocks", "dcz_host_register", "dcz_host_unregister", "dcz_ctx_pinned",
    "dcz_compress_host", "dcz_decompress_host", "dcz_ctx_set_profiling", "dcz_ctx_reset_profiling",
    "dcz_ctx_kernel_time", "dcz_ctx_launch_shapes", "dcz_sha256_blocks", "dczu_fill_java_random", "dczu_fill_text", "dczu_fill_lowentropy",
]

_lib = None


class DczError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        msg = lib().dcz_strerror(status).decode()
        super().__init__("%s (%d)%s" % (msg, status, (": " + detail) if detail else ""))


def lib():
    """Load libdczhip.so; raises if it has not been built (python __graft_entry__.py / build.py)."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # One HIP runtime per process: torch ships its own libamdhip64.so.7 and dlopens it by path. If ours
        # (DT_NEEDED libamdhip64.so.7 -> /opt/rocm) were loaded first, two runtimes would fight over the device;
        # importing torch first makes the loader hand torch's copy to libdczhip.so by SONAME.
        import torch  # noqa: F401
    except ImportError:
        pass  # no torch in the process (e.g. the JNI host): the ROCm runtime under /opt/rocm is used
    if not os.path.exists(SO_PATH):
        raise ImportError("libdczhip.so is missing at %s -- run build.py (hipcc --offload-arch=gfx950). "
                          "This package has no CPU fallback." % SO_PATH)
    L = C.CDLL(SO_PATH)
    vp, sz, i32, u64 = C.c_void_p, C.c_size_t, C.c_int, C.c_uint64
    L.dcz_device_count.restype = i32
    L.dcz_ctx_create.argtypes = [i32, C.POINTER(vp)]
    L.dcz_ctx_create.restype = i32
    L.dcz_ctx_destroy.argtypes = [vp]
    L.dcz_ctx_destroy.restype = None
    L.dcz_ctx_stream.argtypes = [vp]
    L.dcz_ctx_stream.restype = vp
    L.dcz_strerror.argtypes = [i32]
    L.dcz_strerror.restype = C.c_char_p
    L.dcz_last_error.argtypes = [vp]
    L.dcz_last_error.restype = C.c_char_p
    L.dcz_ctx_reserve.argtypes = [vp, sz, sz]
    L.dcz_ctx_reserve.restype = i32
    L.dcz_histogram.argtypes = [vp, vp, sz, sz, vp]
    L.dcz_histogram.restype = i32
    L.dcz_build_codes.argtypes = [vp, vp, vp, vp]
    L.dcz_build_codes.restype = i32
    L.dcz_codes_from_lengths.argtypes = [vp, vp, vp]
    L.dcz_codes_from_lengths.restype = i32
    L.dcz_encode_block.argtypes = [vp, vp, sz, vp, vp, sz, C.POINTER(sz)]
    L.dcz_encode_block.restype = i32
    L.dcz_decode_block.argtypes = [vp, vp, sz, vp, vp, sz, C.POINTER(C.c_int64)]
    L.dcz_decode_block.restype = i32
    L.dcz_compress_blocks.argtypes = [vp, vp, sz, sz, vp, sz, vp, vp, vp, vp, vp, vp]
    L.dcz_compress_blocks.restype = i32
    L.dcz_decompress_blocks.argtypes = [vp, vp, sz, vp, vp, vp, vp, sz, sz, vp, vp, vp, vp]
    L.dcz_decompress_blocks.restype = i32
    L.dcz_host_register.argtypes = [vp, sz]
    L.dcz_host_register.restype = i32
    L.dcz_host_unregister.argtypes = [vp]
    L.dcz_host_unregister.restype = i32
    L.dcz_ctx_pinned.argtypes = [vp, i32, sz]
    L.dcz_ctx_pinned.restype = vp
    L.dcz_compress_host.argtypes = [vp, vp, sz, sz, vp, sz, vp, vp, vp, vp, C.POINTER(u64), vp]
    L.dcz_compress_host.restype = i32
    L.dcz_decompress_host.argtypes = [vp, vp, sz, vp, vp, vp, vp, sz, sz, vp, vp, vp, vp]
    L.dcz_decompress_host.restype = i32
    L.dcz_ctx_set_profiling.argtypes = [vp, i32]
    L.dcz_ctx_set_profiling.restype = i32
    L.dcz_ctx_reset_profiling.argtypes = [vp]
    L.dcz_ctx_reset_profiling.restype = i32
    L.dcz_ctx_kernel_time.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(u64)]
    L.dcz_ctx_kernel_time.restype = i32
    L.dcz_ctx_launch_shapes.argtypes = [vp, C.POINTER(u64 * 4)]
    L.dcz_ctx_launch_shapes.restype = i32
    L.dcz_sha256_blocks.argtypes = [vp, vp, sz, sz, vp, vp]
    L.dcz_sha256_blocks.restype = i32
    L.dczu_fill_java_random.argtypes = [vp, vp, sz, C.c_int64, u64, vp]
    L.dczu_fill_java_random.restype = i32
    L.dczu_fill_text.argtypes = [vp, vp, sz, u64, u64, vp]
    L.dczu_fill_text.restype = i32
    L.dczu_fill_lowentropy.argtypes = [vp, vp, sz, u64, u64, vp]
    L.dczu_fill_lowentropy.restype = i32
    _lib = L
    return L


class Context:
    """One dcz_ctx (one device, one caller thread at a time)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        st = lib().dcz_ctx_create(device, C.byref(self._h))
        if st != DCZ_OK:
            self._h = None
            raise DczError(st)
        self.device = device

    def close(self):
        if self._h:
            lib().dcz_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream_handle(self):
        """hipStream_t of the context's own stream (what a NULL stream argument means)."""
        return lib().dcz_ctx_stream(self.handle)

    @property
    def handle(self):
        if not self._h:
            raise RuntimeError("context is closed")
        return self._h

    def check(self, st):
        if st != DCZ_OK:
            raise DczError(st, lib().dcz_last_error(self.handle).decode() if st == DCZ_E_HIP else "")

    def set_profiling(self, on):
        self.check(lib().dcz_ctx_set_profiling(self.handle, 1 if on else 0))

    def reset_profiling(self):
        self.check(lib().dcz_ctx_reset_profiling(self.handle))

    def kernel_time(self, kernel):
        ms, n = C.c_double(), C.c_uint64()
        self.check(lib().dcz_ctx_kernel_time(self.handle, kernel, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def launch_shapes(self):
        """-> dict of calls by chosen launch shape since the last reset_profiling (include/dcz.h dcz_ctx_launch_shapes)."""
        a = (C.c_uint64 * 4)()
        self.check(lib().dcz_ctx_launch_shapes(self.handle, C.byref(a)))
        return {"decode_flat": a[0], "decode_persistent": a[1], "encode_flat": a[2], "encode_persistent": a[3]}
