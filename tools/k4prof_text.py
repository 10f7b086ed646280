# Debug: k4_dfa phase shares on text for the 1024-thread shape (256 chunks of 4 MiB) and the 256-thread shape (1024 chunks);
# needs a library built with -DDCZ_K4_PROF=1, passed via DCZ_LIB.
import sys, ctypes, numpy as np, torch
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g.load_package()
svc = pkg.HipCompressionService(4, 0)
lib, h = pkg.lib(), svc.ctx.handle
dp = lib.dcz_debug_dfa_prof
dp.argtypes = [ctypes.c_void_p, ctypes.c_int]
dn = ["tables", "registers", "A walk", "A barrier", "scan+err", "B walk/compact", "B barrier", "flush"]
for name, n, bb in [("256 chunks (W=1024)", 1 << 30, 4 << 20), ("1024 chunks (W=256)", 4 << 30, 4 << 20)]:
    t = torch.empty(n, dtype=torch.uint8, device="cuda")
    lib.dczu_fill_text(h, t.data_ptr(), n, 0xD0C2, 0, None)
    blk = svc.compress_device(t, bb)
    K = blk.num_chunks
    orig = torch.full((K,), bb, dtype=torch.int32, device="cuda")
    db = (ctypes.c_ulonglong * 12)()
    torch.cuda.synchronize()
    dp(db, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out, st, ep = svc.decompress_device(blk.payload, blk.comp_off, blk.comp_size, orig, blk.code_lengths, bb)
    e1.record()
    torch.cuda.synchronize()
    dp(db, 1)
    dv = np.array(list(db)[:8], dtype=np.float64)
    print("%s: ok %s, %.3f ms per GiB, windows %d, rounds/window %.2f, cycles/window (lane 0 of a workgroup) %.0f" % (
        name, bool(torch.equal(out[:n], t)), e0.elapsed_time(e1) / (n / 2**30), db[8], db[9] / db[8], dv.sum() / db[8]))
    print("   " + " ".join("%s %.1f%%" % (a, 100 * x / dv.sum()) for a, x in zip(dn, dv)))
    del t, blk, out
    torch.cuda.empty_cache()
