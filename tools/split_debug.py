# Debug: region tables of the split decoder (k4_split.hip) for one workload: which regions fail the proof and why.
import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g.load_package()
svc = pkg.HipCompressionService(1, 0)
lib, h = pkg.lib(), svc.ctx.handle
kind = sys.argv[1] if len(sys.argv) > 1 else "lowentropy"
n, bb = (int(sys.argv[2]) << 20 if len(sys.argv) > 2 else 64 << 20), 16 << 20
t = torch.empty(n, dtype=torch.uint8, device="cuda")
{"text": lib.dczu_fill_text, "lowentropy": lib.dczu_fill_lowentropy}[kind](h, t.data_ptr(), n, 0xD0C5, 0, None)
blk = svc.compress_device(t, bb)
K = blk.num_chunks
orig = torch.full((K,), bb, dtype=torch.int32, device="cuda")
out, st, ep = svc.decompress_device(blk.payload, blk.comp_off, blk.comp_size, orig, blk.code_lengths, bb)
torch.cuda.synchronize()
print("ok", bool(torch.equal(out[:n], t)), "status", st.cpu().numpy()[:K])
so, se = C.c_size_t(), C.c_size_t()
lib.dcz_debug_decode_ws_layout.restype = C.c_size_t
lib.dcz_debug_decode_ws_layout.argtypes = [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
tot = lib.dcz_debug_decode_ws_layout(h, C.byref(so), C.byref(se))
E = se.value
buf = np.zeros(4 * E + 128, dtype=np.uint32)
lib.dcz_debug_read_decode_ws.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
assert lib.dcz_debug_read_decode_ws(h, buf.ctypes.data, so.value, buf.nbytes) == 0
entry, count, exitx, off, nreg = buf[:E], buf[E:2*E], buf[2*E:3*E], buf[3*E:4*E], buf[4*E:4*E+128]
comp_bytes = int(blk.payload.numel())
S = max(65536, (comp_bytes + 2047) // 2048); S = (S + 8191) & ~8191
rmax = (comp_bytes + 15) // S + 2
print("comp_bytes", comp_bytes, "S", S, "rmax", rmax, "nreg", nreg[:K], "csize", blk.comp_size.cpu().numpy())
for b in range(K):
    cs = int(blk.comp_size[b]); nr = (cs + S - 1) // S
    e, c, x = entry[b*rmax:b*rmax+nr], count[b*rmax:b*rmax+nr], exitx[b*rmax:b*rmax+nr]
    bad = [r for r in range(1, nr) if x[r-1] != e[r]] + [r for r in range(nr) if x[r] == 0xFFFFFFFF]
    if bad or b < 2: print("block", b, "regions", nr, "mismatches at", bad[:10], "of", len(bad))
    for r in bad[:4]:
        print("   r", r, "exit[r-1]", x[r-1] if r else None, "entry[r]", e[r], "count", c[r], "exit[r]", x[r])
