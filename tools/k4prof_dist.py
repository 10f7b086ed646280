# Debug: share of the exact-entry procedure in K4 on streams that do not self-synchronise (needs -DDCZ_K4_PROF=1 via DCZ_LIB).
import sys, ctypes, numpy as np, torch
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g.load_package()
svc = pkg.HipCompressionService(1, 0)
lib, h = pkg.lib(), svc.ctx.handle
prof = lib.dcz_debug_k4_prof
prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
N, BB = 1 << 28, 1 << 20
rng = np.random.default_rng(1)
for kind, p in [("hi7", np.r_[np.full(128, 1.0), np.full(128, 2e-4)]), ("bin7_8", 1.0 + 0.3 * np.sin(np.arange(256))),
                ("mid6", np.r_[np.full(64, 1.0), np.full(192, 5e-5)])]:
    p = p / p.sum()
    base = rng.choice(256, size=1 << 24, p=p).astype(np.uint8)
    data = np.concatenate([np.roll(base, 4099 * i) for i in range(N >> 24)])
    t = torch.from_numpy(data).cuda()
    blk = svc.compress_device(t, BB)
    K = blk.num_chunks
    orig = torch.full((K,), BB, dtype=torch.int32, device="cuda")
    buf = (ctypes.c_ulonglong * 12)()
    torch.cuda.synchronize()
    prof(buf, 1)
    out, st, ep = svc.decompress_device(blk.payload, blk.comp_off, blk.comp_size, orig, blk.code_lengths, BB)
    torch.cuda.synchronize()
    prof(buf, 1)
    v = np.array(list(buf), dtype=np.float64)
    tot = v[:8].sum()
    print("%-7s ok %s maxlen %d windows %d rounds/window %.2f | cycles/window (wave 0) %.0f: exact-entry procedure %.1f%% (its serial chain %.1f%% of all)" % (
        kind, bool(torch.equal(out[:N], t)), int(blk.code_lengths.max()), buf[8], buf[9] / max(1, buf[8]), tot / max(1, buf[8]),
        100 * v[10] / tot, 100 * v[11] / tot))
