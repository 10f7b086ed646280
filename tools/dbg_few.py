# Where a build of k4_dfa's 1024-thread shape decodes wrong: text in K chunks of 4 MiB (one workgroup per block), positions of
# the mismatches as (block, window of 1024 subsequences, lane, symbol inside the subsequence).  usage: DCZ_LIB=... dbg_few.py [K]
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as e
pkg, orc = e.load_package(), e.load_oracle()
svc = pkg.HipCompressionService(4, 0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
bb = 4 << 20
n = K * bb
data = orc.gen_text(0xD0C2, 0, n)
t = torch.from_numpy(data).cuda()
blk = svc.compress_device(t, bb)
torch.cuda.synchronize()
orig = torch.full((K,), bb, dtype=torch.int32, device="cuda")
for rep in range(3):
    out, st, ep = svc.decompress_device(blk.payload, blk.comp_off, blk.comp_size, orig, blk.code_lengths, bb)
    torch.cuda.synchronize()
    dec = out[:n].cpu().numpy()
    bad = np.nonzero(dec != data)[0]
    print("rep", rep, "status", np.unique(st.cpu().numpy()).tolist(), "mismatches", bad.size, "in blocks", np.unique(bad // bb).tolist()[:10])
import ctypes
if hasattr(pkg.lib(), "dcz_debug_dfa_dbg"):  # -DDCZ_DFA_DBG=1: what the walk stamped into its slot against what the compaction read
    f = pkg.lib().dcz_debug_dfa_dbg
    f.argtypes = [ctypes.c_void_p, ctypes.c_int]
    buf = (ctypes.c_ulonglong * 8)()
    f(buf, 1)
    print("dbg: lanes compared %d, count differs %d, checksum(walk) != checksum(slot at compaction) %d" % (buf[0], buf[1], buf[3]))
if bad.size:
    lens = blk.code_lengths.cpu().numpy().astype(np.int64)
    shown = 0
    for b in np.unique(bad // bb)[:2]:
        d = data[b * bb:(b + 1) * bb]
        ends = np.cumsum(lens[b][d])              # bit position after each symbol
        sub = (ends - 1) // 256                    # subsequence in which a symbol completes
        first = np.concatenate([[0], np.cumsum(np.bincount(sub))])  # first symbol of every subsequence
        bb_bad = bad[(bad // bb) == b] - b * bb
        gaps = np.diff(bb_bad)
        starts = np.concatenate([[bb_bad[0]], bb_bad[1:][gaps > 1]])
        print("block", int(b), "bad bytes", bb_bad.size, "runs", starts.size)
        for x in starts[:12]:
            k = int(np.searchsorted(first, x, side="right") - 1)
            run = int(np.sum((bb_bad >= x) & (bb_bad < x + 300)))
            print("   byte %8d: window %4d lane %4d (wave %2d lane %2d), symbol %3d of %3d in its subsequence; got %s want %s" % (
                x, k // 1024, k % 1024, (k % 1024) // 64, k % 64, x - first[k], first[k + 1] - first[k],
                dec[b * bb + x:b * bb + x + 6].tolist(), d[x:x + 6].tolist()))
    # statistics over all bad blocks: lanes per (window, wave), wave histogram, how many symbols from the end the first bad one is
    from collections import Counter
    groups, waves, fromend, sizes = Counter(), Counter(), Counter(), Counter()
    for b in np.unique(bad // bb)[:24]:
        d = data[b * bb:(b + 1) * bb]
        ends = np.cumsum(lens[b][d])
        sub = (ends - 1) // 256
        first = np.concatenate([[0], np.cumsum(np.bincount(sub))])
        bb_bad = bad[(bad // bb) == b] - b * bb
        ks = np.searchsorted(first, bb_bad, side="right") - 1
        for k in np.unique(ks):
            xs = bb_bad[ks == k]
            groups[(int(b), int(k) // 1024, (int(k) % 1024) // 64)] += 1
            fromend[int(first[k + 1] - xs.min())] += 1
    for (b, w, wv), c in groups.items():
        waves[wv] += 1
        sizes[c] += 1
    print("corrupted (block, window, wave) groups:", len(groups), "| lanes per group:", sorted(sizes.items()))
    print("waves:", sorted(waves.items()))
    print("first bad symbol, counted from the end of its subsequence:", sorted(fromend.items()))
    print("windows of the first groups:", sorted(groups)[:16])
