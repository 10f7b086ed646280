# Soak: many compress -> decompress round trips per workload shape, every one compared bit for bit with the input
# (rare races -- fill vs byte store in the sparse decoder, the launch-shape hint, the two-half pipeline -- would show here).
# usage: python tools/soak.py [iterations]
import sys, torch
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g.load_package()
svc = pkg.HipCompressionService(1, 0)
lib, h = pkg.lib(), svc.ctx.handle
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
shapes = [("lowentropy", lib.dczu_fill_lowentropy, 0xD0C5, 2 << 30, 4 << 20), ("text", lib.dczu_fill_text, 0xD0C2, 2 << 30, 4 << 20),
          ("random", lib.dczu_fill_java_random, 42, 2 << 30, 1 << 20), ("text 32MiB chunks", lib.dczu_fill_text, 0xD0C2, 1 << 30, 32 << 20),
          ("lowentropy 32MiB chunks", lib.dczu_fill_lowentropy, 0xD0C5, 1 << 30, 32 << 20), ("random 64KiB chunks", lib.dczu_fill_java_random, 7, 1 << 30, 65536)]
bad = 0
for i in range(iters):
    for name, fill, seed, n, bb in shapes:
        t = torch.empty(n, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        fill(h, t.data_ptr(), n, seed + i, 0, None)
        blk = svc.compress_device(t, bb)
        K = blk.num_chunks
        orig = torch.full((K,), bb, dtype=torch.int32, device="cuda")
        out, st, ep = svc.decompress_device(blk.payload, blk.comp_off, blk.comp_size, orig, blk.code_lengths, bb)
        ok = bool(torch.equal(out[:n], t)) and int(st.abs().max()) == 0
        if not ok:
            bad += 1
            print("MISMATCH", i, name, flush=True)
        del t, blk, out
    if i % 5 == 4:
        print("iteration", i + 1, "failures", bad, flush=True)
# runs of all-identity calls (the fused K1 stores the payload in place) broken by calls of other kinds
name, fill, seed, n, bb = shapes[2]
for i in range(3 * iters):
    kind = shapes[1] if i % 13 == 12 else shapes[2]
    t = torch.empty(n if kind is shapes[2] else kind[3], dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    kind[1](h, t.data_ptr(), t.numel(), kind[2] + 1000 + i, 0, None)
    blk = svc.compress_device(t, kind[4])
    K = blk.num_chunks
    orig = torch.full((K,), kind[4], dtype=torch.int32, device="cuda")
    out, st, ep = svc.decompress_device(blk.payload, blk.comp_off, blk.comp_size, orig, blk.code_lengths, kind[4])
    if not (bool(torch.equal(out[:t.numel()], t)) and int(st.abs().max()) == 0):
        bad += 1
        print("MISMATCH in-place run", i, kind[0], flush=True)
    del t, blk, out
print("soak done:", iters, "iterations x", len(shapes), "shapes +", 3 * iters, "calls in identity runs, failures:", bad)
