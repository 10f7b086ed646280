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
reps = int(sys.argv[1])
for seed in [int(a) for a in sys.argv[2:]]:
    data, bb = m._fuzz_case(seed)
    blk, pay, sizes, offs, lens, status = m.hip_compress(svc, data, bb)
    fails, nbad, detail = 0, 0, None
    for r in range(reps):
        dec, st, ep = m.hip_decompress(svc, blk, data.size, bb)
        bad = np.nonzero(dec != data)[0]
        if bad.size or st.any():
            fails += 1
            nbad += bad.size
            if detail is None and bad.size:
                b = int(bad[0]) // bb
                bi = bad[bad // bb == b] - b * bb
                detail = "block %d csize %d bad bytes %s\n   got  %s\n   want %s" % (b, sizes[b], bi[:16].tolist(), dec[b * bb + bi[:16]].tolist(), data[b * bb + bi[:16]].tolist())
                l = lens[b]
                ends = np.cumsum(l[data[b * bb:(b + 1) * bb]].astype(np.int64))
                o = np.concatenate([[0], np.cumsum(np.bincount((ends - 1) // 256))])
                detail += "\n   lane starts %s" % o[:12].tolist()
                # what the same positions hold in the neighbourhood (is it a shifted copy?)
                x = int(bi[0])
                detail += "\n   around first bad: got %s want %s" % (dec[b * bb + x - 6:b * bb + x + 6].tolist(), data[b * bb + x - 6:b * bb + x + 6].tolist())
    print("seed", seed, "K", sizes.size, "bb", bb, "fails %d/%d" % (fails, reps), "bad bytes", nbad, detail or "", flush=True)
