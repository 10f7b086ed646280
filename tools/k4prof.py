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
    if hasattr(lib, "dcz_debug_rw_prof"):
        lib.dcz_debug_rw_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.dcz_debug_rw_prof((ctypes.c_ulonglong * 12)(), 1)
    if hasattr(lib, "dcz_debug_dfa_prof"):
        lib.dcz_debug_dfa_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.dcz_debug_dfa_prof((ctypes.c_ulonglong * 12)(), 1)
    out, st, ep = svc.decompress_device(blk.payload, blk.comp_off, blk.comp_size, orig, blk.code_lengths, bb)
    torch.cuda.synchronize()
    prof(buf, 1)
    v = np.array(list(buf)[:8], dtype=np.float64)
    print("   windows %d, self-sync rounds per window %.2f" % (buf[8], buf[9] / max(1, buf[8])))
    print(name, "ok", bool(torch.equal(out[:n], t)), " ".join("%s %.1f%%" % (nm, 100 * x / max(1, v.sum())) for nm, x in zip(names, v)))
    if hasattr(lib, "dcz_debug_dfa_prof"):  # k4_dfa.hip (medium class)
        dp = lib.dcz_debug_dfa_prof
        dp.argtypes = [ctypes.c_void_p, ctypes.c_int]
        db = (ctypes.c_ulonglong * 12)()
        dp(db, 1)
        dv = np.array(list(db)[:8], dtype=np.float64)
        dn = ["tables", "registers", "A walk", "A barrier", "scan+err", "B walk", "B barrier", "flush"]
        if db[8]:
            print("   dfa: windows %d, rounds/window %.2f, flushes/window %.2f, cycles/window (wave 0) %.0f | %s" % (
                db[8], db[9] / db[8], db[10] / db[8], dv.sum() / db[8], " ".join("%s %.1f%%" % (a, 100 * x / dv.sum()) for a, x in zip(dn, dv))))
    if hasattr(lib, "dcz_debug_rw_prof"):  # k4_regwin.hip (medium class)
        rp = lib.dcz_debug_rw_prof
        rp.argtypes = [ctypes.c_void_p, ctypes.c_int]
        rb = (ctypes.c_ulonglong * 12)()
        rp(rb, 1)
        rv = np.array(list(rb)[:8], dtype=np.float64)
        rn = ["tables", "staging", "A walk", "A barrier", "scan+err+prefetch", "B emit", "B barrier", "flush"]
        if rb[8]:
            print("   regwin: windows %d, rounds/window %.2f, flushes/window %.2f, cycles/window (wave 0) %.0f | %s" % (
                rb[8], rb[9] / rb[8], rb[10] / rb[8], rv.sum() / rb[8], " ".join("%s %.1f%%" % (a, 100 * x / rv.sum()) for a, x in zip(rn, rv))))
