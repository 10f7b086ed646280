# K4 time of the sparse decoder on zero pages with p % of other bytes (host-generated, 1 GiB, 4 MiB chunks), for choosing the
# first nibble of its exit-only walk (DCZ_DFA_X_FROM_SPARSE; variants through DCZ_LIB).  usage: sparse_noise.py [p ...]
import sys, numpy as np, torch
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g.load_package()
svc = pkg.HipCompressionService(4, 0)
n, bb = 1 << 30, 4 << 20
for p in [float(x) for x in (sys.argv[1:] or ["0.5", "1", "2", "3"])]:
    rng = np.random.default_rng(int(p * 100))
    d = np.zeros(n, np.uint8)
    m = rng.random(n) < p / 100.0
    d[m] = rng.integers(1, 256, size=int(m.sum()), dtype=np.uint8)
    t = torch.from_numpy(d).cuda()
    blk = svc.compress_device(t, bb)
    K = blk.num_chunks
    orig = torch.full((K,), bb, dtype=torch.int32, device="cuda")
    out, st, _ = svc.decompress_device(blk.payload, blk.comp_off, blk.comp_size, orig, blk.code_lengths, bb)
    torch.cuda.synchronize()
    assert torch.equal(out[:n], t) and bool((st == 0).all().item())
    bits = 8.0 * float(blk.total.item()) / n
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record()
        svc.decompress_device(blk.payload, blk.comp_off, blk.comp_size, orig, blk.code_lengths, bb, t_out=out, status=st)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("noise %.1f %%: %.2f bits per symbol, decode %.3f ms per GiB (best of 5)" % (p, bits, min(ts)))
    del t, blk, out
