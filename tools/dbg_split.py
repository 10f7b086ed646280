import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as e
pkg, orc = e.load_package(), e.load_oracle()
svc = pkg.HipCompressionService(32, 0)
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 32) << 20
data = orc.gen_text(3, 0, n)
t = torch.from_numpy(data).cuda()
blk = svc.compress_device(t, n)
torch.cuda.synchronize()
orig = torch.tensor([n], dtype=torch.int32, device="cuda")
out, st, ep = svc.decompress_device(blk.payload, blk.comp_off, blk.comp_size, orig, blk.code_lengths, n)
torch.cuda.synchronize()
dec = out[:n].cpu().numpy()
bad = np.nonzero(dec != data)[0]
import ctypes
if hasattr(pkg.lib(), "dcz_debug_dfa_dbg"):
    f = pkg.lib().dcz_debug_dfa_dbg
    f.argtypes = [ctypes.c_void_p, ctypes.c_int]
    buf = (ctypes.c_ulonglong * 8)()
    f(buf, 1)
    print("dbg: lanes %d, stamp.n != nsym %d, checksum(walk) != checksum(slot at compaction) %d (last: stamp %08x slot %08x)" % (
        buf[0], buf[1], buf[3], buf[2] >> 32, buf[2] & 0xFFFFFFFF))
print("n MiB", n >> 20, "regions of 64 KiB ~", int(blk.comp_size[0]) >> 16)
print("status", st.cpu().numpy()[:1], "payload", int(blk.comp_size[0]), "mismatches", bad.size)
if bad.size:
    print("first", bad[:20].tolist())
    gaps = np.diff(bad)
    starts = np.concatenate([[bad[0]], bad[1:][gaps > 64]])
    print("clusters", starts.size, "first cluster starts", starts[:12].tolist())
    print("cluster spacing", np.diff(starts)[:12].tolist())
    x = int(bad[0])
    print("got ", dec[x - 4:x + 16].tolist())
    print("want", data[x - 4:x + 16].tolist())
    # per-subsequence symbol counts around the first bad position
    l = blk.code_lengths.cpu().numpy()[0].astype(np.int64)
    ends = np.cumsum(l[data[:x + 5000]])
    sub = (ends - 1) // 256
    o = np.concatenate([[0], np.cumsum(np.bincount(sub))])
    k = int(np.searchsorted(o, x, side="right") - 1)
    print("first bad in subsequence", k, "(window", k // 256, "lane", k % 256, ") which starts at symbol", int(o[k]), "count", int(o[k + 1] - o[k]))
