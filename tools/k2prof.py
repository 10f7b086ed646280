# Debug: cycles per phase of k2_codebuild (needs a library built with -DDCZ_K2_PROF=1, passed via DCZ_LIB).
import sys, ctypes, numpy as np, torch
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g.load_package()
svc = pkg.HipCompressionService(1, 0)
lib, h = pkg.lib(), svc.ctx.handle
prof = lib.dcz_debug_k2_prof
prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
names = ["row sums", "lengths (heap)", "canonical codes", "len/size out", "segment offsets"]
for name, fill, seed, n, bb in [("text 1GiB/4MiB", lib.dczu_fill_text, 0xD0C2, 1 << 30, 4 << 20),
                                ("lowentropy 1GiB/4MiB", lib.dczu_fill_lowentropy, 0xD0C5, 1 << 30, 4 << 20),
                                ("text 1GiB/32MiB", lib.dczu_fill_text, 0xD0C2, 1 << 30, 32 << 20)]:
    t = torch.empty(n, dtype=torch.uint8, device="cuda")
    fill(h, t.data_ptr(), n, seed, 0, None)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    prof(buf, 1)
    blk = svc.compress_device(t, bb)
    torch.cuda.synchronize()
    prof(buf, 1)
    v = np.array(list(buf)[:5], dtype=np.float64)
    print(name, "blocks %d, cycles/block %.0f |" % (buf[7], v.sum() / max(1, buf[7])), " ".join("%s %.1f%%" % (a, 100 * x / v.sum()) for a, x in zip(names, v)))
