import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import __graft_entry__ as e
import importlib.util
spec = importlib.util.spec_from_file_location("t", os.path.join(ROOT, "tests/test_gpu_parity.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
pkg, orc = e.load_package(), e.load_oracle()
svc = pkg.HipCompressionService(1, 0)
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    data, bb = m._fuzz_case(seed)
    blk, pay, sizes, offs, lens, status = m.hip_compress(svc, data, bb)
    sh0 = svc.ctx.launch_shapes()
    dec, st, ep = m.hip_decompress(svc, blk, data.size, bb)
    sh1 = svc.ctx.launch_shapes()
    bad = np.nonzero(dec != data)[0]
    if bad.size or st.any():
        print("seed", seed, "n", data.size, "bb", bb, "K", sizes.size, "status nonzero", np.nonzero(st)[0][:10], "mismatches", bad.size,
              "decode shape persistent" if sh1["decode_persistent"] > sh0["decode_persistent"] else "decode shape flat")
        blocks = np.unique(bad // bb)
        print(" bad blocks", blocks[:20], "of", sizes.size)
        for b in blocks[:6]:
            bi = bad[bad // bb == b] - b * bb
            l = lens[b]
            nb = min(bb, data.size - b * bb)
            print("  block", b, "n", nb, "csize", sizes[b], "bits/sym %.2f" % (8.0 * sizes[b] / nb), "nsyms", int((l > 0).sum()),
                  "lens", sorted(set(l[l > 0].tolist())), "skew", int(offs[b]) & 15, "first bad", bi[:6], "last bad", bi[-3:], "count", bi.size)
            i = int(bi[0])
            print("    got ", dec[b * bb + max(0, i - 4): b * bb + i + 12])
            print("    want", data[b * bb + max(0, i - 4): b * bb + i + 12])
print("done")
