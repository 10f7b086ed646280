import sys, os, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import __graft_entry__ as g
pkg = g.load_package(); orc = g.load_oracle()
import test_gpu_parity as T
svc = pkg.HipCompressionService(1, 0)
bad = 0
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (5000, 8000)
for seed in range(lo, hi):
    data, bb = T._fuzz_case(seed)
    try:
        T.assert_parity(svc, orc, data, bb)
    except AssertionError as e:
        bad += 1
        print("FAIL seed", seed, data.size, bb, str(e)[:100])
        if bad > 5: break
# fixed-length-like with random alphabet sizes and block sizes (exact-entry path)
rng = np.random.default_rng(99)
for t in range(150):
    k = int(rng.integers(2, 257))
    n = int(rng.integers(50000, 4000000))
    bb = int(rng.choice([n, 65536, 1 << 20, int(rng.integers(1000, n))]))
    if (n + bb - 1) // bb > 3000: bb = (n + 2999) // 3000
    syms = rng.choice(256, size=k, replace=False)
    data = rng.choice(syms, size=n).astype(np.uint8)
    try:
        T.assert_parity(svc, orc, data, bb)
    except AssertionError as e:
        bad += 1
        print("FAIL fixed k", k, n, bb, str(e)[:100])
        if bad > 5: break
print("extended fuzz done, failures:", bad)
