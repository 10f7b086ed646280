# one line per bench JSON file: value, ms per step, average launch time of every kernel family, compression ratio
# (k1c = K1 fused with the identity copy, include/dcz.h DCZ_K_HISTOGRAM_COPY: it runs instead of K1 and K3's copy)
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.load(open(f))
        k=d["kernels"]
        k1c=k.get("k1_histogram_copy",{}).get("avg_ms",0.0)
        print("%-28s %8.1f GB/s %8.2f ms | k1 %.2f%s k2 %.2f k3 %.2f k4 %.2f | C/N %.3f" % (f.split("/")[-1], d["value"], d["ms_per_step"], k["k1_histogram"]["avg_ms"], (" k1c %.2f" % k1c) if k1c else "", k["k2_codebuild"]["avg_ms"], k["k3_encode"]["avg_ms"], k["k4_decode"]["avg_ms"], d["config"]["compressed_bytes_per_gpu"]/d["config"]["bytes_per_gpu"]))
    except Exception as e:
        print(f, "ERR", e)
