# Debug: per-phase cycle shares of K4 (needs a library built with -DDCZ_K4_PROF=1, passed via DCZ_LIB).
import sys, ctypes, numpy as np, torch
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g.load_package()
svc = pkg.HipCompressionService(1, 0)
lib, h = pkg.lib(), svc.ctx.handle
prof = lib.dcz_debug_k4_prof
prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
names = ["A decode", "staging+sync", "A barrier/check", "scan+err", "prefetch", "B write", "flush", "tail/end"]
for name, fill, seed, n, bb in [("rand2g", lib.dczu_fill_java_random, 42, 2 << 30, 1 << 20),
                                ("text2g", lib.dczu_fill_text, 0xD0C2, 2 << 30, 1 << 20),
                                ("low2g", lib.dczu_fill_lowentropy, 0xD0C5, 2 << 30, 1 << 20)]:
    t = torch.empty(n, dtype=torch.uint8, device="cuda")
    fill(h, t.data_ptr(), n, seed, 0, None)
    blk = svc.compress_device(t, bb)
    K = blk.num_chunks
    orig = torch.full((K,), bb, dtype=torch.int32, device="cuda")
    buf = (ctypes.c_ulonglong * 12)()
    prof(buf, 1)
    out, st, ep = svc.decompress_device(blk.payload, blk.comp_off, blk.comp_size, orig, blk.code_lengths, bb)
    torch.cuda.synchronize()
    prof(buf, 1)
    v = np.array(list(buf)[:8], dtype=np.float64)
    print("   windows %d, self-sync rounds per window %.2f" % (buf[8], buf[9] / max(1, buf[8])))
    print(name, "ok", bool(torch.equal(out[:n], t)), " ".join("%s %.1f%%" % (nm, 100 * x / v.sum()) for nm, x in zip(names, v)))
